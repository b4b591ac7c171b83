// ufm_relax.h -- k_relax (one tile visit: staging, asynchronous in-LDS sweeps, write-back and activations) in its three scheduler forms -- fused triage over a short list (!DYN), cursor hand-out of a ready list (DYN, with k_triage), resident owners (OWNK) --, and k_triage
// (a piece of ufm_engine.hip, the engine's one translation unit: included there, inside its anonymous namespace)
#pragma once

// ---- the hot kernel ------------------------------------------------------------------------
// Launch k reads candidate list k%3 (priorities of parity k&1), appends to list (k+1)%3
// (parity (k+1)&1) and recycles the counter of list (k+2)%3.
//  1. triage (fused, every workgroup redundantly): theta = smallest queued priority + delta.
//     Tiles above theta are carried over untouched -- their inputs are still going to improve
//     (tile-level fast-marching order); delta = +inf is plain FIM (used for invalidation).
//  2. each released tile is staged in LDS (+1 halo), relaxed to its fixed point by 16 waves that
//     sweep 4x4-node patches asynchronously (wake bits in LDS, no workgroup barrier per sweep),
//     written back, and the neighbours whose halo changed are queued with the smallest changed
//     value as priority.
// DYN: the tiles come from the ready list k_triage built (taken through an atomic cursor: perfect
// balance, used while the queue is long); !DYN: triage fused as described above (short queues).
// OWN: the resident form (one launch per lowering phase): no lists at all, every workgroup serves the tiles it owns
// from their queue words (own_push above) until all of them, everywhere, are empty.
template <int ALGO, int MODE, bool DYN, int OWNK = 0>   // OWNK: 0 launch chain, 1 resident with 16 waves per visit, 2 resident with 8
__global__ __launch_bounds__(OWNK == 2 ? NTHR / 2 : NTHR, UFM_RELAX_WAVES) void k_relax(DevParams P, int k_arg, float delta, float rbound, int max_sweeps) {
    constexpr bool OWN = OWNK != 0;
    static_assert(!OWN || (!DYN && MODE == MODE_LOWER), "the resident kernel lowers");
    // The resident kernel can give a tile visit 8 waves instead of 16 and run two visits per CU: during the sweeps about five of a
    // visit's 16 patches are active at a time (the front crosses the tile), so half the waves are idle slots of the SIMDs.
    // A wave then owns the two patches (pr, pc) with (pr + 2 pc) mod 8 = its index -- no two patches of a row, a column or a
    // diagonal, the lines a front lies along, share a wave.
    constexpr int NTH = OWNK == 2 ? NTHR / 2 : NTHR;
    constexpr int NWV = NTH / 64;
    constexpr int PPWK = (PT * PT) / NWV;
    constexpr bool SKEW = (NWV == 8 && PT == 4);
    static_assert(SKEW || PPWK == PPW, "patch-to-wave maps: 4 x 4 waves of PR x PR patches, or the skewed 8-wave one");
    __shared__ float Gs[(T + 2) * GP];
    __shared__ float Cs[(T + 1) * CP];
    __shared__ int s_wake[16];  // per wave: bit j = patch j of the wave has new inputs (PPWK bits)
    __shared__ int s_idle;      // waves currently without work
    __shared__ int s_giveup;    // a wave hit the sweep cap: end the visit, re-queue the tile
    __shared__ int s_misc[4];   // 0: first touch, 1: earlier visits in this step, 2: patch sweeps (sum), 3: (max per wave)
    __shared__ int s_bmin[9];   // per direction: smallest changed value on that border (float bits)
    __shared__ int s_min;
    __shared__ float s_B[64];   // fused triage: start key of the first 64 maps
    __shared__ unsigned long long s_best;   // resident kernel: {priority, slot} of the best tile this workgroup may take / collect checksum
    __shared__ int s_gmin;      // resident kernel: votes of a decision (own_decide), then what thread 0 made of it (1 take, -1 stop)
    __shared__ int s_pf[OWN ? NTH : 1];      // resident kernel: this workgroup's first queue words and the other owners' hints as of the
    __shared__ int s_pfh[OWN ? OWN_NW : 1];   // start of the visit in progress (loaded straight into LDS while it sweeps)
    __shared__ int s_se[OWN ? 256 : 1];       // resident kernel: start elements of the first 64 maps (index into G, -1 unused) ...
    __shared__ float s_sh[OWN ? 256 : 1];     // ... and hm * dist(start, element)
    __shared__ int s_wsel[2];   // resident kernel, UFM_DIRWAKE: [0] the visit ends at the end condition (no sweeps), [1] the tile's last visit converged (its `seen` record is complete)
    __shared__ int s_late;      // resident kernel: thread 0 has seen the time limit pass (no more tiles are taken ahead: the next look leaves)
    __shared__ int s_held;      // resident kernel, P.dag_on: the last decision saw a queued tile of this workgroup that is still waiting for first visits of its neighbours
    __shared__ int s_pd[(OWN && DAG) ? NTH : 1];      // ... and the counts of those (dag_left) for this workgroup's first queue words, loaded with s_pf
    __shared__ int s_own[4];    // resident kernel, thread 0's book-keeping: 0 slot whose mark is still to be taken back, 1 slot being visited, 2 visits
    __shared__ unsigned long long s_stat[3];   // thread 4's per-workgroup statistics (visits, sweeps, evaluations), flushed once at the end
    // resident kernel, node planners: border values are handed to the neighbours DURING the visit (early hand-off, below):
    // Os = what HBM holds for every element of the tile (as staged, then as last written), s_emin = per wave and direction the
    // smallest border value an early write has changed
    // (16 waves per visit only: with 8 waves and two visits per CU -- the form for jobs that are bound by the number of visits, not by
    //  their chain -- the longer visits cost more than the saved ones bring: 8192^2 plan 41.3 -> 44.0 ms)
    constexpr bool EARLY = (OWNK == 1 || (OWNK == 2 && UFM_EARLY_HANDOFF > 1)) && UFM_EARLY_HANDOFF && !is_dfm<ALGO>;
    __shared__ float Os[EARLY ? TT : 1];
    __shared__ int s_emin[EARLY ? 16 * 9 : 1];
#ifdef UFM_TIMING
    __shared__ unsigned int s_misc_vi;
#endif
    __shared__ uint8_t Bs[(MODE == MODE_RAISE && !is_dfm<ALGO>) ? TT : 1];   // the tile's back-pointer bytes (invalidation of the node planners)
    __shared__ int s_qw[2];       // in-visit refresh: [0] this tile's queue word as an idle wave last saw it (loaded straight into LDS), [1] refreshes of this visit

    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int q = lane & 3, nd = lane >> 2;                    // quad lane, node within the 4x4 patch
    if constexpr (NWV == 8) UFM_SETPRIO(UFM_VISIT_PRIO);
    // (wp: the wave's index as an owner of patches -- wave_index16(): which SIMD a patch class runs on)
    const int wp = (NWV == 16 && PR == 1) ? wave_index16(w) : (SKEW ? wave_index8(w) : w);
    const int wr = wp >> 2, wc = wp & 3;                       // the wave's 8x8 region = 2x2 patches
    const int io_r = tid / T, io_c = tid % T;                  // HBM mapping (threads tid < T*T)
    const bool io_on = tid < T * T;
    constexpr int Q = (MODE == MODE_LOWER) ? Q_LOWER : Q_RAISE;
    const int k = launch_index(P, Q, k_arg);
    const int r = k % 3, rz = (k + 2) % 3;
    // (the ready entry at this workgroup's own index is asked for together with the list lengths: it is the right one
    //  whenever the index lies in the front part of the list -- almost always -- and a memory round trip earlier)
    const int spec_first = DYN ? P.ready[blockIdx.x] : 0;
    const int n_long = DYN ? P.ctr->nready[k & 1] : 0;
    const int n = OWN ? 0x7FFFFFFF : (DYN ? n_long + P.ctr->nshort[k & 1] : P.ctr->cnt[Q][r]);
    if (!OWN && blockIdx.x == 0 && tid == 0) {
        P.ctr->cnt[Q][rz] = 0; P.ctr->rel[Q][rz] = 0; P.ctr->lmin[Q][rz] = INFBITS;
        P.ctr->nready[(k + 1) & 1] = 0; P.ctr->nshort[(k + 1) & 1] = 0; P.ctr->rcursor[(k + 1) & 1] = 0;   // for the next triage
        if (DYN) { P.ctr->rel[Q][r] = n; if (n) P.ctr->last_work[Q] = k; }
    }
    if (n == 0) return;
    const int focused = P.dyn->focused, thr = P.dyn->thr;
    UFM_TICK(tkb);
    const int *cand = P.cand + (size_t)(Q * 3 + r) * P.NT;
    constexpr int CROWS = is_dfm<ALGO> ? T : T + 1;
    constexpr int COFF = is_dfm<ALGO> ? 0 : 1;

    if (tid == 0) s_min = INFBITS;
    // the list -> priority loads of the scan below are issued before the start keys are waited for: the two
    // chains of dependent loads (start elements -> G, list -> priority) run side by side instead of in series
    int c_first = 0, pb_first = INFBITS;
    if (!DYN && !OWN && MODE == MODE_LOWER && tid < n) { c_first = cand[tid]; pb_first = prio_read(P, Q, k, c_first); }
    if (!DYN && !OWN && tid < 64) s_B[tid] = (MODE == MODE_LOWER && focused && tid < P.nmaps) ? start_bound(P, tid) : INFINITY;
    __syncthreads();
    if (!DYN && !OWN && MODE == MODE_LOWER) {   // smallest priority among the entries that are not parked beyond their map's start key
        // (invalidation is order-free -- delta = +inf --: no band, no scan, two dependent loads less per launch)
        int lmin = INFBITS;
        for (int i = tid; i < n; i += NTH) {
            const int c = (i == tid) ? c_first : cand[i], pb = (i == tid) ? pb_first : prio_read(P, Q, k, c), mm = c / P.NTm;
            const float Bm = mm < 64 ? s_B[mm] : INFINITY;
            if (__int_as_float(pb) < Bm || Bm == INFINITY) lmin = min(lmin, pb);
        }
        if (lmin != INFBITS) atomicMin(&s_min, lmin);
    }
    __syncthreads();
    const float theta = __int_as_float(s_min) + delta;

    // lanes 0..8 each watch one of the 3x3 patches around the one being swept: the lanes of the
    // patch that border it (wake_sel); lane 4 is the patch itself
    unsigned long long wake_sel = 0ull;
    if (lane < 9) {
        const int dr = lane / 3 - 1, dc = lane % 3 - 1;
        wake_sel = ~0ull;
        if (dr < 0) wake_sel &= 0x000000000000FFFFull; else if (dr > 0) wake_sel &= 0xFFFF000000000000ull;
        if (dc < 0) wake_sel &= 0x000F000F000F000Full; else if (dc > 0) wake_sel &= 0xF000F000F000F000ull;
    }
    const int colour = ((nd >> 2) & 1) | ((nd & 1) << 1);

    // A workgroup's first tile is the one at its own index -- no round trip to the shared cursor
    // before the first visit (256 same-address atomics across 8 XCDs take microseconds); the launch
    // has one workgroup per CU, so all of them start at once and the longest-first order is kept.
    bool first_pop = UFM_STATIC_FIRST;
    if (tid == 4) { s_stat[0] = 0ull; s_stat[1] = 0ull; s_stat[2] = 0ull; }   // (thread 4 alone reads and writes them)
    int st_lmax = 0;
    // resident kernel: this workgroup's queue words.  Slots are kept as indices into the whole array of words (owner * own_slots + slot),
    // because an idle workgroup also takes tiles of other owners (below).  s_own[0] = word of a finished visit whose "being visited" mark
    // is still to be taken back; own_next = word taken (marked) for the next visit while the current one was being written back
    int *const own_q = OWN ? P.own_prio + (size_t)blockIdx.x * P.own_slots : nullptr;
    const int own_base = OWN ? (int)blockIdx.x * P.own_slots : 0;
    int own_next = -1, own_slot_now = -1;   // (the same in every thread)
    int prev_gt = -1;                       // the tile of the visit that has just ended (-1: none yet / it ended at the end condition)
    int own_hrot = 0;                       // which part of the hints the visit in progress has loaded ahead
    [[maybe_unused]] int look_rot = 0;      // ... and which part the look of an idle workgroup loads
    [[maybe_unused]] int held_looks = 0;    // P.dag_on: looks in a row that found only held tiles
    const unsigned long long own_t0 = OWN ? wall_clock64() : 0ull;
    if (OWN && tid == 0) { s_own[0] = -1; s_own[1] = -1; s_own[2] = 0; s_late = 0; }
    if constexpr (OWN) {   // the start elements of the first 64 maps: address in G and the heuristic term of their keys (start_bound())
        const float hm = P.dyn->hm;
        for (int e = tid; e < 4 * min(P.nmaps, 64); e += NTH) {
            const int el = P.start[e], m_ = e >> 2;
            const int x = el / P.EY, y = el - x * P.EY;
            s_se[e] = el >= 0 ? (int)gaddr(P, m_, x, y) : -1;
            s_sh[e] = el >= 0 ? hm * hypotf(P.spos[2 * m_] - (float)x, P.spos[2 * m_ + 1] - (float)y) : 0.0f;
        }
    }
    // the resident scheduler's decisions -- own_decide, own_empty, own_take_issue / _resolve, own_commit, own_steal --: ufm_relax_resident.inc
#include "ufm_relax_resident.inc"
    for (int i = blockIdx.x;; i += gridDim.x) {
        int gt_own = -1;
        if constexpr (OWN) {
            if (own_next >= 0 && s_own[3] >= INFBITS) own_next = -1;   // chosen ahead from the older copy of the words, but the take failed
            if (tid == 0 && s_own[1] >= 0) { s_own[0] = s_own[1]; s_own[1] = -1; }   // (at most one mark waits: own_commit ran since)
            if (NWV == 8 && own_next < 0) UFM_SETPRIO(0);                  // (looking for work: behind the waves of the CU's other visit)
            // Following the front (round 4, second session; own_flags & 64 turns it off): a workgroup that has nothing of its own to go on with takes the neighbour it has
            // just queued with the smallest priority -- the tile the front moves into -- by the ordinary take, instead of leaving it to its owner's next look
            // (word, look, take: ~10 us of a ring's ~30 even with the chip nearly empty).  Inside the ordering band only (unless own_flags & 128).
            if (UFM_FOLLOW && !(P.own_flags & 64) && own_next < 0 && prev_gt >= 0) {
                if (w == 0) {
                    const int pm = prev_gt / P.NTm, pt = prev_gt - pm * P.NTm, ptx = pt / P.TY, pty = pt - ptx * P.TY;
                    unsigned long long mine = ~0ull;
                    if (lane < 9 && lane != 4) {
                        const int ntx = ptx + lane / 3 - 1, nty = pty + lane % 3 - 1, pb = s_bmin[lane];
                        if (pb != INFBITS && ntx >= 0 && ntx < P.TX && nty >= 0 && nty < P.TY) mine = ((unsigned long long)(unsigned int)pb << 32) | (unsigned int)lane;
                    }
                    const int hint = (lane < UFM_HINT_SAMPLE && (lane + own_hrot * UFM_HINT_SAMPLE) % P.own_nw != (int)blockIdx.x) ? s_pfh[lane] : INFBITS;
                    int gw = -1;
#pragma unroll 1
                    for (int attempt = 0; attempt < UFM_FOLLOW && gw < 0; ++attempt) {        // (the best neighbour may be in a visit already: then the next one)
                        unsigned long long key = mine;
                        for (int o_ = 8; o_; o_ >>= 1) key = min(key, (unsigned long long)__shfl_xor((long long)key, o_));
                        key = (unsigned long long)__shfl((long long)key, 0);
                        if (key == ~0ull) break;
                        const int fp = (int)(key >> 32), d = (int)(unsigned int)key;
                        const bool beyond = hint != INFBITS && __int_as_float(fp) > __int_as_float(hint) + delta;
                        if ((__ballot(beyond) != 0ull && !(P.own_flags & 128)) || s_late) break;
                        if (lane == 0) {
                            const int ngt = pm * P.NTm + (ptx + d / 3 - 1) * P.TY + (pty + d % 3 - 1);
                            int o_, sl_;
                            own_locate(P, ngt, o_, sl_);
                            const int cand = o_ * P.own_slots + sl_;
                            int pr = fp, r_old, r_lk;
                            own_take_issue(cand, pr, r_old, r_lk);
                            if (own_take_resolve(cand, pr, r_old, r_lk)) {
                                gw = cand; s_own[3] = pr;
                                __hip_atomic_fetch_min(&P.own_min[blockIdx.x], pr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // what this workgroup holds now
                            }
                        }
                        gw = __shfl(gw, 0);
                        if (lane == d) mine = ~0ull;
                    }
                    if (lane == 0) s_gmin = gw;
                }
                __syncthreads();
                const int fw = s_gmin;
                __syncthreads();                               // (s_gmin is used again)
                if (fw >= 0) own_next = __builtin_amdgcn_readfirstlane(fw);
            }
            while (own_next < 0) {                             // nothing was taken ahead: look, wait, look again
                __syncthreads();                               // LDS of the previous visit / round is free
                // the other owners' hints.  (UFM_LEAN_LOOKS: a sample of them, another one at every look and in an order of this workgroup's own --
                //  the band is a heuristic, as for the visits (UFM_HINT_SAMPLE); workgroup 0, which decides when the phase is over, looks at all.)
                int hint = INFBITS, ho = tid;
#if UFM_LEAN_LOOKS
                const int nh = blockIdx.x == 0 ? P.own_nw : min(P.own_nw, UFM_LOOK_HINTS);
                ++look_rot;
                if (blockIdx.x != 0) ho = (int)((unsigned int)(tid + look_rot * nh + (int)blockIdx.x * 61) % (unsigned int)P.own_nw);
                if (tid < nh && ho != (int)blockIdx.x) hint = __hip_atomic_load(&P.own_min[ho], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
                if (tid < P.own_nw && tid != (int)blockIdx.x) hint = __hip_atomic_load(&P.own_min[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
                const int v0 = tid < P.own_slots ? __hip_atomic_load(&own_q[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : INFBITS;
                const int d0 = (DAG && P.dag_on && tid < P.own_slots) ? __hip_atomic_load(&P.dag_left[own_base + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                const int aborted = tid == 0 ? __hip_atomic_load(&P.ctr->own_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                own_decide(v0, hint, -1, d0, DAG && held_looks >= P.dag_patience);
                const unsigned long long b = s_best;
                const int votes = s_gmin;
                const bool have = b != ~0ull;
                if constexpr (DAG) held_looks = (!have && s_held) ? held_looks + 1 : 0;      // (the same in every thread: s_best / s_held are read behind own_decide's closing barrier)
                const bool take = have && !(votes & 2);        // inside the ordering band
                bool stop = false;
                if (!have && !(votes & 1) && blockIdx.x == 0) {
                    // nobody seems to hold anything: two collects of all queue words; identical and all empty = the phase is over.
                    // (Workgroup 0 alone looks -- it tells the others through own_abort = 2: with every workgroup collecting for itself the
                    //  end of a 4096^2 plan was 512 x 2 x 264 KB of loads.)
                    unsigned long long h0 = 0ull, h1 = 1ull;
                    bool ok = true;
                    const int total = P.own_nw * P.own_slots;
                    for (int pass = 0; pass < 2 && ok; ++pass) {
                        __syncthreads();
                        if (tid == 0) s_best = 0ull;
                        __syncthreads();
                        unsigned long long acc = 0ull;
                        bool mine_ok = true;
                        // (eight loads in flight per thread instead of 153 dependent round trips per pass on a 4096^2 map.  What is left to gain
                        //  here: with no collect at all the product build's 14.2 ms plan kernel is 0.15-0.2 ms shorter -- profiles/r3_end_of_phase.txt)
#pragma unroll 1
                        for (int e0 = tid; e0 < total; e0 += NTH * 8) {
                            int v[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) v[u] = __hip_atomic_load(&P.own_prio[min(e0 + u * NTH, total - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                const int e = e0 + u * NTH;
                                if (e < total) {
                                    if (v[u] < INFBITS || v[u] >= OWN_MARK) mine_ok = false;
                                    acc += ((unsigned long long)(unsigned int)v[u] + 1ull) * (0x9E3779B97F4A7C15ull + 2ull * (unsigned long long)e);
                                }
                            }
                        }
                        for (int o_ = 32; o_; o_ >>= 1) acc += (unsigned long long)__shfl_xor((long long)acc, o_);
                        if (lane == 0) atomicAdd(&s_best, acc);
                        ok = __syncthreads_and(mine_ok) != 0;
                        if (pass == 0) h0 = s_best; else h1 = s_best;
                    }
                    stop = ok && h0 == h1;
                }
                if (tid == 0) {
                    int flag = 0;
                    // hand back to the launch chain (k_own_export): never stay for ever -- and not alone: a workgroup that only became
                    // resident when the others had left (the device was shared) must not wait out a limit of its own
                    if (aborted == 2) stop = true;            // workgroup 0 has seen the end
                    else if (stop) __hip_atomic_store(&P.ctr->own_abort, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef UFM_TIMING
                    if (blockIdx.x == 0) plog(stop ? 1u : 0u, stop ? 0xFFFFFFFFu : 0xFFFFFFFDu, (unsigned int)(b >> 32));
#endif
                    const bool late = !stop && (aborted || wall_clock64() - own_t0 > P.own_limit);
                    if (late) { atomicAdd(&P.ctr->own_stops, 1); __hip_atomic_store(&P.ctr->own_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                    const bool taking = take && !stop && !late;
                    int r_old = INFBITS, r_lk = 1;
                    own_commit(b, taking, true, r_old, r_lk);
                    // (the take is waited for: the visit's loads must not overtake it -- an activation it removes has to be one whose
                    //  values the visit then sees)
                    int pr = (int)(b >> 32);
                    const bool got = taking && own_take_resolve(own_base + (int)(unsigned int)b, pr, r_old, r_lk);
                    s_own[3] = got ? pr : INFBITS;
#ifdef UFM_TIMING
                    atomicAdd(&g_sdiag[0], 1ull);
                    if (!taking) atomicAdd(&g_sdiag[1], 1ull);
                    if (taking && !got) atomicAdd(&g_sdiag[7], 1ull);
#endif
                    if (stop || late) flag = -1;
                    else if (got) flag = 1;
                    s_gmin = flag;
                }
                __syncthreads();
                const int flag = s_gmin;
                if (flag > 0) own_next = own_base + __builtin_amdgcn_readfirstlane((int)(unsigned int)b);
                if (flag < 0) break;
                if (flag == 0) {
                    __syncthreads();                           // (s_gmin is used again)
                    int got = -1;
                    if ((votes & 1) && !(P.own_flags & 32)) got = own_steal(hint, ho);
                    if (got >= 0) own_next = __builtin_amdgcn_readfirstlane(got);
                    else __builtin_amdgcn_s_sleep(UFM_LOOK_SLEEP);
                }
            }
            if (own_next < 0) break;
            if constexpr (NWV == 8) UFM_SETPRIO(UFM_VISIT_PRIO);
            { const int o_ = own_next / P.own_slots; int m_, tx_, ty_; gt_own = own_tile(P, o_, own_next - o_ * P.own_slots, m_, tx_, ty_); }
            if (tid == 0) s_own[1] = own_next;
            own_slot_now = own_next;
            own_next = -1;
            __syncthreads();                                   // LDS of the previous visit is free
            UFM_STRICT_ACQUIRE();                              // (checking build: the take is behind us, the staging loads in front)
        }
        if (DYN && !first_pop) {                           // next ready tile, whoever is free takes it
            __syncthreads();
            if (tid == 0) s_min = (UFM_STATIC_FIRST ? gridDim.x : 0) + atomicAdd(&P.ctr->rcursor[k & 1], 1);
            __syncthreads();
            i = s_min;
        }
        first_pop = false;
        if (i >= n) {
#ifdef UFM_TIMING
            if (MODE == MODE_LOWER && tid == 0) trace_rec(k, 1, tkb, wall_clock64(), 0);
#endif
            break;
        }
        UFM_TICK(tk0);
#ifdef UFM_TIMING
        if (OWN && tid == 0 && gt_own >= 0 && gt_own < TILE_DIAG_MAX) {
            const unsigned int now = (unsigned int)(tk0 - g_tile_t0);
            atomicMin(&g_tile[0][gt_own], now);
            const unsigned int pushed = atomicExch(&g_tile[2][gt_own], 0xFFFFFFFFu);
            if (pushed != 0xFFFFFFFFu && now > pushed) atomicAdd(&g_tile[4][gt_own], now - pushed);
            atomicAdd(&g_tile[3][gt_own], 1u);
            const unsigned long long p64 = atomicExch(&g_push64[gt_own], ~0ull);
            const unsigned int vi = atomicAdd(&g_nvis, 1u);
            s_misc_vi = vi;
            if (vi < VIS_DIAG_MAX) { g_vis[vi][0] = gt_own; g_vis[vi][1] = now; g_vis[vi][2] = 0u; g_vis[vi][3] = (unsigned int)(p64 >> 32); g_vis[vi][4] = (unsigned int)p64; }
        }
#endif
        const int gt = OWN ? gt_own : (DYN ? (i < n_long ? (i == (int)blockIdx.x ? spec_first : P.ready[i]) : P.ready[P.NT - 1 - (i - n_long)]) : cand[i]);
        const int pbits = (DYN || OWN) ? 0 : prio_read(P, Q, k, gt);
        const int m = gt / P.NTm, t = gt - m * P.NTm;
        // lowering: release within the ordering band and below the start's key (end condition);
        // invalidation: release below the bound the host derived from the start's key
        if (!DYN && !OWN) {
            bool release, parked = false;
            if (MODE == MODE_LOWER) {
                // (the start key as of the beginning of the launch, s_B: re-reading the start elements for
                //  every tile put two more dependent memory round trips before each visit)
                const float B = focused ? (m < 64 ? s_B[m] : start_bound(P, m)) : INFINITY;
                const float hd = focused ? tile_heuristic(P, m, t / P.TY, t % P.TY) : 0.0f;
                parked = !(__int_as_float(pbits) + hd < B || B == INFINITY);
                release = !(__int_as_float(pbits) > theta) && !parked;
            } else {
                release = !(__int_as_float(pbits) > (rbound < 0.0f ? P.ctr->rbound : rbound));
            }
            __syncthreads();                               // LDS of the previous tile is free
            if (!release) {                                // not yet: carry over / park beyond the bound
                if (tid == 0) {
                    if (parked || MODE == MODE_RAISE) park_tile(P, Q, gt, pbits);
                    else activate(P, Q, k + 1, gt, pbits);
                }
                continue;
            }
            if (tid == 0) { atomicAdd(&P.ctr->rel[Q][r], 1); atomicMax(&P.ctr->last_work[Q], k); }
        }
        const int tx = t / P.TY, ty = t - tx * P.TY;
        const int x0 = tx * T, y0 = ty * T;
        float *Gt = P.G + (size_t)gt * TT;                       // this tile's values (thread tid owns element tid)
        const float *ring = P.ring + (size_t)gt * RING;          // its neighbours' border values
        const uint8_t *ct = P.costT + (size_t)gt * CTS;          // its cost window

        // All global loads of the staging are issued first, unconditionally (clamped addresses instead of
        // branches), then thread 0's bookkeeping atomics, and only then are the results consumed: one memory
        // round trip in front of the visit.  (Written as guarded blocks -- load, wait, LDS store, each -- the
        // first waves paid three round trips in series, thread 0's wave up to five.)
        const int ht = tid - (NTH - (4 * T + 4));               // halo: the last 4T+4 threads of the workgroup
        constexpr int CN = CROWS * CROWS;
        const float gl0 = ld_f<OWN>(&Gt[io_on ? tid : 0]);
        const float hv = ld_f<OWN>(&ring[ht >= 0 ? ht : 0]);
        constexpr bool DIRWAKE = OWN && UFM_DIRWAKE && MODE == MODE_LOWER;
        float *seen = DIRWAKE ? P.seen + (size_t)gt * RING : nullptr;
        // (thread ht == -1 loads the record's "complete" flag)
        const float sv = (DIRWAKE && ht >= -1) ? ld_f<OWN>(&seen[ht >= 0 ? ht : RING - 1]) : 0.0f;
        const int c0 = ct[tid < CN ? tid : 0];
        constexpr bool BPRAISE = MODE == MODE_RAISE && !is_dfm<ALGO>;   // invalidation along the stored back-pointers
        const int bp0 = BPRAISE ? P.bp[(size_t)gt * TT + (io_on ? tid : 0)] : BP_NONE;
        const int goal_x = P.goal[2 * m], goal_y = P.goal[2 * m + 1];   // (with the rest: read after the barrier they cost two more round trips)
        // resident kernel: the values of the map's start elements, one per lane (for the end condition below)
        const int own_sa = (OWN && focused && tid < 4 && m < 64) ? s_se[m * 4 + tid] : -1;
        const float own_sg = own_sa >= 0 ? ld_f<OWN>(&P.G[own_sa]) : INFINITY;
        // P.dag_on: the arrival estimate of this tile and the thresholds of its eight neighbours (lane 0..8, 4 = this tile): after its FIRST visit the
        // tile counts itself off at every neighbour that was waiting for it (the activations' lanes, below)
        [[maybe_unused]] float dag_mine = 0.0f, dag_nthr = -INFINITY;
        [[maybe_unused]] int dag_nslot = -1;
        if constexpr (OWN && DAG) if (P.dag_on && tid < 9 && tid != 4) {
            const int ntx = tx + tid / 3 - 1, nty = ty + tid % 3 - 1;
            if (ntx >= 0 && ntx < P.TX && nty >= 0 && nty < P.TY) {
                const int ngt = m * P.NTm + ntx * P.TY + nty;
                int o_, s_;
                own_locate(P, ngt, o_, s_);
                dag_nslot = o_ * P.own_slots + s_;
                dag_nthr = P.dag_thr[ngt];
                dag_mine = P.dag_a[gt];
            }
        }
        if (tid == 0) {
            const int seen = atomicAdd(&P.touched[gt], 1);   // visits of this tile in the current step
            const int first = seen == 0;
            if (first) P.tlist[atomicAdd(&P.ctr->tcount, 1)] = gt;
            s_misc[0] = first; s_misc[1] = seen; s_misc[2] = 0; s_misc[3] = 0; s_idle = 0; s_giveup = 0;
        }
        if (tid < NWV) s_wake[tid] = DIRWAKE ? 0 : (1 << PPWK) - 1;      // (DIRWAKE: set behind the staging barrier, when it is known what there is to wake)
        if (DIRWAKE && ht == -1) s_wsel[1] = (sv == 1.0f) ? 1 : 0;
        bool own_parked = false;      // (thread 0)
        if constexpr (OWN) if (focused && w == 0) {
            // End condition: a tile whose priority lies beyond its map's start key (start_bound(): the largest key among the
            // start elements that have been reached) is not relaxed -- it goes to the park list of the launch chain, the
            // counterpart of the entries the reference leaves in its priority queue when end_condition() fires -- and the visit
            // ends without a sweep (no wake bits).  The start key only falls while a phase lowers: beyond it stays beyond it.
            float bq = (own_sa >= 0 && own_sg < INFINITY) ? own_sg + s_sh[m * 4 + (tid & 3)] : 0.0f;
            bq = fmaxf(bq, __shfl_xor(bq, 1));
            bq = fmaxf(bq, __shfl_xor(bq, 2));
            if (tid == 0) {
                const float B = m < 64 ? (bq > 0.0f ? bq : INFINITY) : start_bound(P, m);
                const int kb = s_own[3];                         // the priority the tile was taken with
                if (!(__int_as_float(kb) + tile_heuristic(P, m, tx, ty) < B || B == INFINITY)) {
                    for (int j = 0; j < NWV; ++j) s_wake[j] = 0;
                    park_tile(P, Q_LOWER, gt, kb);
                    own_parked = true;
                }
            }
        }
        if (DIRWAKE && tid == 0) s_wsel[0] = own_parked ? 1 : 0;
        if (tid >= 32 && tid < 41) s_bmin[tid - 32] = INFBITS;

        // the tile (contiguous) and its halo: the ring record, in this order (RING_*)
        if (io_on) Gs[(io_r + 1) * GP + io_c + 1] = gl0;
        if (BPRAISE && io_on) Bs[tid] = (uint8_t)bp0;
        if (ht >= 0) {
            int hr, hc;
            if (ht < T) { hr = -1; hc = ht; }
            else if (ht < 2 * T) { hr = T; hc = ht - T; }
            else if (ht < 3 * T) { hr = ht - 2 * T; hc = -1; }
            else if (ht < 4 * T) { hr = ht - 3 * T; hc = T; }
            else { hr = (ht & 2) ? T : -1; hc = (ht & 1) ? T : -1; }
            Gs[(hr + 1) * GP + hc + 1] = hv;
        }
        // the cost window as float (inf = obstacle / outside: Graph::get_cost, Graph.cpp:262-268)
        for (int e = tid; e < CN; e += NTH) {
            const int cr = e / CROWS, cc = e - cr * CROWS;
            const int cx = x0 + cr - COFF, cy = y0 + cc - COFF;
            const int c = (e == tid) ? c0 : ct[e];
            Cs[cr * CP + cc] = (cx < 0 || cy < 0 || cx >= P.L || cy >= P.W || c >= thr) ? INFINITY : (float)c;
        }
        if constexpr (EARLY) { if (io_on) Os[tid] = gl0; if (tid < NWV * 9) s_emin[tid] = INFBITS; if (tid == 0) { s_qw[0] = OWN_MARK; s_qw[1] = own_parked ? 0x20000 : 0; } }
        __syncthreads();
        // the halo entry of thread ht (>= 0) inside the staged tile: row / column -1 or T
        auto halo_rc = [&](int &hr, int &hc) {
            if (ht < T) { hr = -1; hc = ht; }
            else if (ht < 2 * T) { hr = T; hc = ht - T; }
            else if (ht < 3 * T) { hr = ht - 2 * T; hc = -1; }
            else if (ht < 4 * T) { hr = ht - 3 * T; hc = T; }
            else { hr = (ht & 2) ? T : -1; hc = (ht & 1) ? T : -1; }
        };
        if constexpr (DIRWAKE) {
            // What is there to sweep?  The tile's first visit of the step, or one after a visit that ran into the sweep cap: everything.  Otherwise the
            // tile's own values are a fixed point of the halo its last visit saw (`seen`): only the patches along halo entries that have changed since
            // can have anything to do -- the others are woken by their neighbours if it comes to that.  (Round 3 woke all sixteen at every visit: half of
            // the 7.6 M patch sweeps of a 4096^2 plan that found nothing to do.)  A visit that finds no changed entry at all ends after the vote.
            if (!s_wsel[0]) {
                if (s_misc[0] || !s_wsel[1]) { if (tid < NWV) s_wake[tid] = (1 << PPWK) - 1; }
                else if (ht >= 0 && __float_as_int(hv) != __float_as_int(sv)) {
                    int hr, hc;
                    halo_rc(hr, hc);
                    const int r0 = max(hr - 1, 0) / 4, r1 = min(hr + 1, T - 1) / 4, c0_ = max(hc - 1, 0) / 4, c1 = min(hc + 1, T - 1) / 4;
                    for (int pr_ = r0; pr_ <= r1; ++pr_)
                        for (int pc_ = c0_; pc_ <= c1; ++pc_) {
                            int wv, bit;
                            if constexpr (SKEW) { wv = (pr_ + 2 * pc_) & 7; bit = (pc_ == (wv < 4 ? 0 : (wv >> 1) - 1)) ? 1 : 2; }
                            else { wv = (pr_ / PR) * 4 + (pc_ / PR); bit = 1 << ((pr_ % PR) * PR + (pc_ % PR)); }
                            __hip_atomic_fetch_or(&s_wake[wv], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                }
            }
            lds_barrier();
        }
        UFM_TICK(tk1);
#ifdef UFM_TIMING
        const int dbg_hint = P.hint[gt];
        const int dbg_rank = P.rank[gt];
        const int dbg_ninf0 = __syncthreads_count(io_on && gl0 == INFINITY);
#endif
        // One changed element (r, c) of the tile goes out: its value (was `gref` in HBM) into the tile's own record and into the rings
        // of the neighbours it borders; bm[9] (LDS, float bits, one entry per direction, 4 = this tile itself) notes the smallest changed
        // value each neighbour has to hear of.  Used by the write-back at the end of a visit and by the early hand-off during it.
        auto wb_store = [&](int wb_r, int wb_c, float gf) {
            st_f<OWN>(&Gt[wb_r * T + wb_c], gf);
#ifdef UFM_TIMING
            s_qw[1] |= 0x10000;      // (diagnostics: the visit changed a value)
#endif
            {   // a border value also lives in the rings of the neighbours it borders
                const int er_ = (wb_r == 0) ? -1 : ((wb_r == T - 1) ? 1 : 0);
                const int ec_ = (wb_c == 0) ? -1 : ((wb_c == T - 1) ? 1 : 0);
                const bool rok = er_ && tx + er_ >= 0 && tx + er_ < P.TX, cok = ec_ && ty + ec_ >= 0 && ty + ec_ < P.TY;
                if (rok) st_f<OWN>(&P.ring[(size_t)(gt + er_ * P.TY) * RING + (er_ < 0 ? RING_BOT : RING_TOP) + wb_c], gf);
                if (cok) st_f<OWN>(&P.ring[(size_t)(gt + ec_) * RING + (ec_ < 0 ? RING_RIGHT : RING_LEFT) + wb_r], gf);
                if (rok && cok)   // my corner (er_, ec_) is the opposite corner of the diagonal neighbour's halo
                    st_f<OWN>(&P.ring[(size_t)(gt + er_ * P.TY + ec_) * RING + RING_CORNER + (er_ < 0 ? 2 : 0) + (ec_ < 0 ? 1 : 0)], gf);
            }
        };
        // ... and which neighbours have to hear of it.  Bit 0: the one across this element's row border, 1: across its column border,
        // 2: the diagonal one, 3: this tile itself (a border value that rose)
        auto wb_need = [&](int wb_r, int wb_c, float gf, float gl0) -> int {
            // MS-DFM only: the float fixed point of the upwind quadratic is not unique; two neighbouring tiles can
            // push each other's border values through a finite set of last-bit states for ever (2048^2, seed 1006, seen once).  The
            // livelock guard: in a tile that has come back more than UFM_DFM1_QUIET_VISITS times in one step a change of at most
            // 4 ulp is stored but does not wake the neighbour.
            bool significant = true;
            if (is_dfm<ALGO> && MODE == MODE_LOWER && gf < INFINITY && gl0 < INFINITY) {
                const int du = __float_as_int(gf) - __float_as_int(gl0);
                if (s_misc[1] > UFM_DFM1_QUIET_VISITS && du >= -4 && du <= 4) significant = false;
            }
            const int er = (wb_r == 0) ? -1 : ((wb_r == T - 1) ? 1 : 0);
            const int ec = (wb_c == 0) ? -1 : ((wb_c == T - 1) ? 1 : 0);
            // Causality: every value the update operators produce is larger than each input it
            // depends on (the interpolated cost-to-goal of the far edge plus a positive traversal
            // cost), so an element h of a neighbour tile can neither be lowered by nor have been
            // supported by a border value that is, before and after this visit, not below h: the
            // wake-up -- half of all tile visits used to find nothing to do -- is skipped.  h is read
            // from the halo as staged.  A neighbour that is being visited in this same launch only
            // lowers its border meanwhile, which keeps the test conservative -- except for the
            // ulp-level rises of replace semantics: a tile whose own border ROSE during a visit
            // therefore comes back once more (s_bmin[4]) and re-reads its neighbours' borders.
            bool need_r = true, need_c = true, need_d = true;
            if (UFM_CAUSAL_FILTER && MODE == MODE_LOWER) {
                const float lo = fminf(gf, gl0);
                const int cl = max(wb_c - 1, 0) + 1, ch = min(wb_c + 1, T - 1) + 1;     // halo columns / rows that belong
                const int rl = max(wb_r - 1, 0) + 1, rh = min(wb_r + 1, T - 1) + 1;     // to the edge neighbour itself
                if (er) {
                    const float *h = Gs + (wb_r + 1 + er) * GP;
                    need_r = lo < fmaxf(fmaxf(h[cl], h[wb_c + 1]), h[ch]);
                }
                if (ec) {
                    const int hc = wb_c + 1 + ec;
                    need_c = lo < fmaxf(fmaxf(Gs[rl * GP + hc], Gs[(wb_r + 1) * GP + hc]), Gs[rh * GP + hc]);
                }
                if (er && ec) need_d = lo < Gs[(wb_r + 1 + er) * GP + wb_c + 1 + ec];
                // Node planners, lowered value: sharper.  Whatever a neighbour's border node h can gain from this side
                // comes over the row of cells between the two tiles, from the border nodes next to h: its new value
                // would be at least (the smallest of those nodes) + (the cheaper of the two cells it touches on this side)
                // x (one edge length).  (The nodes next to h in the halo belong to a third tile; if one of them is being
                // lowered in this very launch, this tile sees its old value -- but then it is that tile's visit that
                // holds the edge's cheaper end and makes the same test with the right number.)  A neighbour whose border already lies below that -- a front running beside
                // this tile, a step ahead of it -- has nothing to gain and is not woken (41 % of the plan's tile visits
                // found nothing to do with the test above alone).  Rises keep the test above: an ulp-level correction
                // must reach whoever was computed from the old value.
                if (UFM_STEP_FILTER && !is_dfm<ALGO> && gf < gl0) {
                    const int crow_r = (er < 0) ? 0 : T;                   // cost row / column of the cells between the tiles
                    const int ccol_c = (ec < 0) ? 0 : T;
                    auto gain_r = [&](int hc) {                           // h = halo row, LDS column hc (node column hc - 1 of the tile)
                        const float *mine = Gs + (wb_r + 1) * GP;          // my border row (new values)
                        const float m3 = fminf(fminf(mine[hc - 1], mine[hc]), mine[hc + 1]);       // (halo columns included: a node of the tile beside
                                                                                           //  this one can be the cheaper end of the edge)
                        const float c2 = fminf(Cs[crow_r * CP + hc - 1], Cs[crow_r * CP + hc]);
                        return Gs[(wb_r + 1 + er) * GP + hc] > m3 + c2;
                    };
                    auto gain_c = [&](int hr) {
                        const int mc = wb_c + 1;
                        const float m3 = fminf(fminf(Gs[(hr - 1) * GP + mc], Gs[hr * GP + mc]), Gs[(hr + 1) * GP + mc]);
                        const float c2 = fminf(Cs[(hr - 1) * CP + ccol_c], Cs[hr * CP + ccol_c]);
                        return Gs[hr * GP + mc + ec] > m3 + c2;
                    };
                    if (er && need_r) need_r = gain_r(cl) || gain_r(wb_c + 1) || gain_r(ch);
                    if (ec && need_c) need_c = gain_c(rl) || gain_c(wb_r + 1) || gain_c(rh);
                    if (er && ec && need_d) need_d = Gs[(wb_r + 1 + er) * GP + wb_c + 1 + ec] >
                                                      fminf(gf, fminf(Gs[(wb_r + 1 + er) * GP + wb_c + 1], Gs[(wb_r + 1) * GP + wb_c + 1 + ec])) + Cs[crow_r * CP + ccol_c];
                }
            }
            int need = 0;
            if (UFM_CAUSAL_FILTER && MODE == MODE_LOWER && (er || ec) && significant && gf > gl0) need |= 8;
            if (er && significant && need_r) need |= 1;
            if (ec && significant && need_c) need |= 2;
            if (er && ec && significant && need_d) need |= 4;
            return need;
        };
        // bm[9] (LDS, float bits; one entry per direction, 4 = this tile itself): the smallest changed value each neighbour has to hear
        // of = its priority: the new value (lowering) / the value that was invalidated (raising: the reference's key of an
        // under-consistent element, min(g, rhs) = g)
        auto wb_note = [&](int wb_r, int wb_c, int need, float gf, float gl0, int *bm) {
            const int pb = __float_as_int((MODE == MODE_LOWER) ? gf : gl0);
            const int er = (wb_r == 0) ? -1 : ((wb_r == T - 1) ? 1 : 0);
            const int ec = (wb_c == 0) ? -1 : ((wb_c == T - 1) ? 1 : 0);
            if (need & 8) atomicMin(&bm[4], __float_as_int(gl0));
            if (need & 1) atomicMin(&bm[(er + 1) * 3 + 1], pb);
            if (need & 2) atomicMin(&bm[3 + ec + 1], pb);
            if (need & 4) atomicMin(&bm[(er + 1) * 3 + ec + 1], pb);
        };
        if (s_misc[0]) {   // first touch of the tile in this step: snapshot for num_nodes_expanded -- unless there is nothing
            // to remember (a plan's tiles hold only +inf: 1 KB of writes per tile and as many reads at the end saved)
            const int any = __syncthreads_or(io_on && gl0 != INFINITY);
            if (any && io_on) P.Gprev[(size_t)gt * TT + tid] = gl0;
            if (tid == 0) P.fresh[gt] = any ? 0 : 1;
        }
        if constexpr (OWN) {   // (no register is held for these: the data lands in LDS some time during the sweeps)
            typedef __attribute__((address_space(3))) void *lds_ptr;
            typedef const __attribute__((address_space(1))) void *glb_ptr;
            __builtin_amdgcn_global_load_lds((glb_ptr)(own_q + min(tid, P.own_slots - 1)), (lds_ptr)(s_pf + (tid & ~63)), 4, 0, 16);              // (16: sc1)
            if (DAG && P.dag_on) __builtin_amdgcn_global_load_lds((glb_ptr)(P.dag_left + own_base + min(tid, P.own_slots - 1)), (lds_ptr)(s_pd + (tid & ~63)), 4, 0, 16);
            // (the hints: a different quarter or half of them at every visit -- the band is a heuristic, 2 KB of hints per visit next to
            //  1.8 KB of tile data is not)
            ++own_hrot;
            if (tid < UFM_HINT_SAMPLE) {
                const int ho = (tid + own_hrot * UFM_HINT_SAMPLE) % P.own_nw;
                __builtin_amdgcn_global_load_lds((glb_ptr)(P.own_min + ho), (lds_ptr)(s_pfh + (tid & ~63)), 4, 0, 16);
            }
        }

        // per-lane constants of the wave's four patches
        QuadConsts<ALGO> C[PPWK];
        int off[PPWK], wword[PPWK], wbit[PPWK], bpc[PPWK];
        bool goal[PPWK];
#pragma unroll
        for (int j = 0; j < PPWK; ++j) {
            int pr_, pc_;                                      // the wave's patch j in the PT x PT patch grid
            if constexpr (SKEW) { pc_ = j ? (wp < 2 ? 3 : (wp >> 1)) : (wp < 4 ? 0 : (wp >> 1) - 1); pr_ = (wp - 2 * pc_) & 7; }
            else { pr_ = wr * PR + j / PR; pc_ = wc * PR + j % PR; }
            const int lx = pr_ * 4 + (nd >> 2), ly = pc_ * 4 + (nd & 3);
            C[j].load(Cs, lx, ly, q);
            off[j] = (lx + 1) * GP + ly + 1;
            bpc[j] = BPRAISE ? Bs[lx * T + ly] : BP_NONE;
            goal[j] = (x0 + lx == goal_x) & (y0 + ly == goal_y);
            wword[j] = 0; wbit[j] = 0;
            if (lane < 9) {
                const int gr = pr_ + lane / 3 - 1, gc = pc_ + lane % 3 - 1;
                if (gr >= 0 && gr < PT && gc >= 0 && gc < PT) {
                    if constexpr (SKEW) { const int wv = (gr + 2 * gc) & 7; wword[j] = wv; wbit[j] = (gc == (wv < 4 ? 0 : (wv >> 1) - 1)) ? 1 : 2; }
                    else { wword[j] = (gr / PR) * 4 + (gc / PR); wbit[j] = 1 << ((gr % PR) * PR + (gc % PR)); }
                }
            }
        }

        // Asynchronous in-LDS relaxation.  A wave takes the wake bits of its four patches and
        // sweeps each woken patch in a burst: as long as the patch keeps changing itself it is
        // re-swept back to back with no synchronisation at all (LDS operations of one wave are
        // ordered); neighbouring patches are woken by fire-and-forget ds_or.  A wave without work
        // counts itself idle and polls; when all 16 are idle a two-step barrier vote (arrive, then
        // look at the wake bits, which are stable once everybody has arrived) decides between
        // "converged" and "resume".
        // Increases in the lowering phase are ulp-level corrections of values computed from
        // transient neighbours.  Two neighbours that feed each other can flip-flop forever if they
        // rise in the same sweep, so an element may only rise in sweeps of its own colour
        // (4-colouring: no two 8-neighbours share one).
        int cnt[PPWK] = {};
        int tot = 0;
        int ew_done = 0;      // early hand-off: patches of this wave that have handed their border out once in this visit
        int ew_pend = 0;      // ... border values of this wave are on their way to HBM, the neighbours have not been queued yet
        // early hand-off and in-visit refresh (resident kernel, node planners) -- ew_flush, halo_poll, halo_refresh --: ufm_relax_early.inc
#include "ufm_relax_early.inc"
        const bool lax = is_dfm<ALGO> && s_misc[1] > UFM_DFM1_LAX_VISITS;
        bool conv = false;
#ifdef UFM_TIMING
        const bool wtrace_on = DYN && MODE == MODE_LOWER && k == UFM_TRACE_K0 && i == 0;
        UFM_WREC(0, s_misc[1]);
#endif
        for (;;) {
            int bits = 0;
            if (lane == 0) bits = atomicExch(&s_wake[wp], 0);
            bits = __builtin_amdgcn_readfirstlane(bits);
            bool vote = __hip_atomic_load(&s_giveup, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
            if (bits && !vote) {
                UFM_WREC(1, bits);
#pragma unroll
                for (int j = 0; j < PPWK; ++j) {
                    if (!(bits & (1 << j))) continue;    // wave-uniform
                    float *ctr = Gs + off[j];
                    bool again = true;
                    asm volatile("" ::: "memory");
                    // the node's own value lives in a register during a burst: only this quad writes it, so
                    // re-reading it from LDS after the evaluation only put a second LDS round trip on the
                    // dependent chain of every sweep
                    float g = ctr[0];
#ifdef UFM_SWEEPSTAT
                    const int sst_c0 = cnt[j];
#endif
                    for (int b = 0; b < 16 && again; ++b) {
                        asm volatile("" ::: "memory");   // re-read the LDS tile every sweep (other waves and lanes write it)
                        float rl;                        // this lane's candidate
                        if constexpr (BPRAISE) rl = eval_quad_bp<ALGO>(ctr, q, C[j], bpc[j]);
                        else rl = eval_quad<ALGO>(ctr, q, C[j]);
                        float nv = quad_min(rl);
                        if (goal[j]) nv = 0.0f;          // RHS(goal) = 0, *_impl.h init()
                        bool want, doit;
                        if (MODE == MODE_LOWER) {
                            want = (nv != g);            // replace semantics: G <- F(G)
                            // DFM: the upwind quadratic is not causal at the ulp level -- elements that feed
                            // each other can creep upwards one ulp per sweep for tens of thousands of launches
                            // (seen on 2048^2).  In a tile that keeps coming back (`lax`, > 16 visits in one
                            // step) a rise of 1 ulp is treated as rounding noise and left alone; everywhere
                            // else the relaxation stays exact.
                            if (is_dfm<ALGO> && lax) want = want & !((nv > g) & (nv < INFINITY) & (__float_as_int(nv) - __float_as_int(g) <= 1));
                            doit = want & ((nv < g) | (colour == (cnt[j] & 3)));
                        } else {
                            // value lost its support (DFM: by more than the 8 ulp its neighbours may be stale)
                            if (is_dfm<ALGO>) want = (g < INFINITY) & (nv > g) & ((nv == INFINITY) | (__float_as_int(nv) - __float_as_int(g) > 8));
                            else want = (g < INFINITY) & (nv > g);
                            doit = want;
                            nv = INFINITY;
                        }
                        if (doit && q == 0) ctr[0] = nv;
                        // the lane masks come from float compares (one v_cmp each): a ballot of a combined
                        // predicate costs a v_cndmask + v_cmp to rebuild the mask the compare already was
                        const float gn = doit ? nv : g;
                        const unsigned long long mask = __builtin_amdgcn_ballot_w64(gn != g);   // = doit
                        unsigned long long wanted;                                               // lanes not yet settled
                        if (MODE == MODE_RAISE) wanted = mask;
                        else if (is_dfm<ALGO>) wanted = __builtin_amdgcn_ballot_w64(want);
                        else wanted = __builtin_amdgcn_fcmpf(nv, g, 14);   // lanes with nv != g (14 = FCMP_UNE), as a v_cmp into an SGPR pair
#ifdef UFM_SWEEPSTAT
                        {
                            const unsigned long long chg = __builtin_amdgcn_ballot_w64(gn != g), low = __builtin_amdgcn_ballot_w64(gn < g);
                            if (lane == 0) {
                                atomicAdd(&g_sstat[0], 1ull);
                                if (!chg) atomicAdd(&g_sstat[1], 1ull);
                                atomicAdd(&g_sstat[2], (unsigned long long)__popcll(chg) / 4ull);
                                atomicAdd(&g_sstat[5], (unsigned long long)__popcll(low) / 4ull);
                                if (b == 0) { atomicAdd(&g_sstat[3], 1ull); if (!chg) atomicAdd(&g_sstat[4], 1ull); }
                            }
                        }
#endif
                        g = gn;
                        UFM_SWEEP_FENCE();                               // value before wake bit
                        if ((mask & wake_sel) != 0ull && wbit[j] && lane != 4)
                            __hip_atomic_fetch_or(&s_wake[wword[j]], wbit[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        ++cnt[j];
                        ++tot;
                        again = wanted != 0ull;
                    }
#ifdef UFM_SWEEPSTAT
                    if (lane == 0) atomicAdd(&g_sstat[8 + min(cnt[j] - sst_c0, 16) - 1], 1ull);
#endif
                    if (again && lane == 0)              // burst cap: leave the rest to the next take
                        __hip_atomic_fetch_or(&s_wake[wp], 1 << j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if constexpr (EARLY) if (!(P.own_flags & 2)) {
                        // Early hand-off.  A plan is a chain of dependent tile visits: the next tile on a front's way can
                        // only start when this visit has been written back, although its inputs -- this tile's far border -- are usually
                        // there long before the visit ends (the rest of it is the tile settling behind the front).  So a border patch
                        // whose burst is over and has LOWERED border values writes them out at once (tile record + the neighbours' rings,
                        // the same code and the same wake-up filters as the write-back) and queues the neighbours, in this order:
                        // stores, wait for them, queue words -- per wave what the write-back does per workgroup.  Os remembers what
                        // HBM holds, so the write-back at the end of the visit only handles what has changed since.  Rises (ulp-level
                        // corrections) wait for the write-back.
                        int t_ = tid;
                        asm volatile("" : "+v"(t_));     // (nothing of this is to be computed ahead of the sweeps and carried through them)
                        const int w_ = t_ >> 6, l_ = t_ & 63;
                        int pr_, pc_;
                        if constexpr (SKEW) { const int wq_ = wave_index8(w_); pc_ = j ? (wq_ < 2 ? 3 : (wq_ >> 1)) : (wq_ < 4 ? 0 : (wq_ >> 1) - 1); pr_ = (wq_ - 2 * pc_) & 7; }
                        else { const int wq_ = (NWV == 16 && PR == 1) ? wave_index16(w_) : w_; pr_ = (wq_ >> 2) * PR + j / PR; pc_ = (wq_ & 3) * PR + j % PR; }
                        if ((pr_ == 0 || pr_ == PT - 1 || pc_ == 0 || pc_ == PT - 1) && !((P.own_flags & 4) && (ew_done & (1 << j)))) {      // (wave-uniform)
                            const int lx = pr_ * 4 + (l_ >> 4), ly = pc_ * 4 + ((l_ >> 2) & 3);
                            const bool onb = (lx == 0) | (lx == T - 1) | (ly == 0) | (ly == T - 1);
                            const float o = Os[lx * T + ly];
                            const bool chg = onb & (g < o) & ((l_ & 3) == 0);
                            const int need = chg ? (wb_need(lx, ly, g, o) & 7) : 0;
                            if (__builtin_amdgcn_ballot_w64(need != 0) != 0ull) {        // (a neighbour has something to gain: otherwise nothing is written)
                                int *bm = s_emin + w_ * 9;
                                ew_done |= 1 << j;
                                if (chg) {
                                    Os[lx * T + ly] = g; wb_store(lx, ly, g); wb_note(lx, ly, need, g, o, bm);
                                    atomicMin(&bm[4], __float_as_int(g));     // (the tile's own priority, should the visit end at the sweep cap)
                                }
                                if (P.own_flags & 8) ew_flush();     // (variant: wait for the stores here, in the sweep loop)
                                else ew_pend = 1;                    // the queue words follow when the wave has nothing to sweep (idle loop)
                            }
                        }
                    }
                }
                UFM_WREC(2, tot);
                if (tot >= PPWK * max_sweeps && lane == 0)  // give up this visit; the tile is re-queued
                    __hip_atomic_store(&s_giveup, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                continue;
            }
            if (!vote) {                                 // nothing to do: idle until woken or all idle
                if (bits && lane == 0) __hip_atomic_fetch_or(&s_wake[wp], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (lane == 0) atomicAdd(&s_idle, 1);
                UFM_WREC(3, 0);
                int polls = 0;
                if constexpr (NWV == 8) UFM_SETPRIO(0);         // (an idle wave's looks: behind the waves that sweep)
                for (;;) {
                    __builtin_amdgcn_s_sleep(UFM_IDLE_SLEEP);
                    ++polls;
                    if constexpr (EARLY) {
                        if (ew_pend && polls >= UFM_EARLY_POLLS) ew_flush();   // (the stores are ~1 us old by now: no wait)
                        // (not in a visit that ended at the end condition: an activation taken back there -- its priority may lie below
                        //  the start's key -- would be lost with the sweeps that visit does not make)
                        // (the refresh reloads the 68 floats of a 16 x 16 tile's ring record with one wave)
                        if (T == 16 && !(P.own_flags & 18) && !(__hip_atomic_load(&s_qw[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & 0x20000)) {
                            if (w < 4 && (polls & 31) == 8 + 4 * w) halo_poll();
                            if (halo_refresh()) break;                         // (back to the wake bits: this wave is not idle any more)
                        }
                    }
                    if (__hip_atomic_load(&s_idle, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= NWV ||
                        __hip_atomic_load(&s_giveup, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) { vote = true; break; }
                    if (__hip_atomic_load(&s_wake[wp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) {
                        if (lane == 0) atomicSub(&s_idle, 1);
                        UFM_WREC(4, 0);
                        break;
                    }
                }
                if constexpr (NWV == 8) UFM_SETPRIO(UFM_VISIT_PRIO);
                if (!vote) continue;
            } else if (bits && lane == 0) {
                __hip_atomic_fetch_or(&s_wake[wp], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // put back what was taken
            }
            // vote: everybody arrives first, then the wake bits are stable
            UFM_WREC(5, 0);
            __syncthreads();
            const int work = __syncthreads_or(__hip_atomic_load(&s_wake[wp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0);
            const int gave_up = __hip_atomic_load(&s_giveup, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (gave_up || !work) { conv = !gave_up; break; }
            if (tid == 0) s_idle = 0;
            __syncthreads();
        }
        if (lane == 0 && tot) { atomicAdd(&s_misc[2], tot); atomicMax(&s_misc[3], tot); }
        __syncthreads();
        UFM_TICK(tk2);

        int own_ro = INFBITS, own_rl = 1;     // thread 0: what the two atomics of the take ahead returned (looked at after the write-back)
        if constexpr (OWN) {
            // The next tile is chosen and marked now, from the queue words as they were when this visit began to sweep: the
            // exchange is on its way while this visit is written back, and the next visit's loads follow the write-back with no
            // queue round trip in between.  (A fresh look costs two round trips in a row -- words, then exchange -- per visit.)
            own_decide(tid < P.own_slots ? s_pf[tid] : INFBITS, (tid < UFM_HINT_SAMPLE && (tid + own_hrot * UFM_HINT_SAMPLE) % P.own_nw != (int)blockIdx.x) ? s_pfh[tid] : INFBITS,
                       (own_slot_now >= own_base && own_slot_now < own_base + P.own_slots) ? own_slot_now - own_base : -1, (DAG && P.dag_on && tid < P.own_slots) ? s_pd[tid] : 0, false);
            const unsigned long long b = s_best;
            const bool take = b != ~0ull && !(s_gmin & 2) && !(P.own_flags & 1) && !s_late;
            if (tid == 0) {
                own_commit(b, take, false, own_ro, own_rl);
                if (wall_clock64() - own_t0 > P.own_limit) s_late = 1;   // (a workgroup that is never out of work looks at the clock here)
            }
            own_next = __builtin_amdgcn_readfirstlane(take ? own_base + (int)(unsigned int)b : -1);
        }
        // write back what changed; note which neighbours saw their halo change
        // (this thread's row and column, made opaque once per visit: the compiler otherwise computes the two dozen LDS addresses of
        //  the tests below ahead of the tile loop and carries them through the sweeps -- registers the sweep loop needs)
        int wb_r = io_r, wb_c = io_c;
        asm volatile("" : "+v"(wb_r), "+v"(wb_c));
        if constexpr (DIRWAKE) if (!s_wsel[0]) {      // the halo this visit has converged against (as staged, plus what an in-visit refresh took in), and whether it did converge
            if (ht >= 0) {
                int hr, hc;
                halo_rc(hr, hc);
                const float hcur = Gs[(hr + 1) * GP + hc + 1];
                if (__float_as_int(hcur) != __float_as_int(sv)) st_f<OWN>(&seen[ht], hcur);
            } else if (ht == -1) {
                const float f = conv ? 1.0f : 0.0f;
                if (f != sv) st_f<OWN>(&seen[RING - 1], f);
            }
        }
        const float gref = EARLY ? (io_on ? Os[tid] : 0.0f) : gl0;   // what HBM holds (early hand-off: as last written during the visit)
        const float gf = io_on ? Gs[(wb_r + 1) * GP + wb_c + 1] : gref;
        if (gf != gref) {
            wb_store(wb_r, wb_c, gf);
            wb_note(wb_r, wb_c, wb_need(wb_r, wb_c, gf, gref), gf, gref, s_bmin);
            if (!conv) atomicMin(&s_bmin[4], __float_as_int((MODE == MODE_LOWER) ? gf : gref));
        }
        // (early hand-off: what a wave has written out but not yet told the neighbours goes with the write-back's activations)
        // (... and the smallest value written out early counts for the tile's own priority when the visit ended at the sweep cap, like
        //  every other value the visit changed)
        if constexpr (EARLY) if (tid < NWV * 9) { const int v = s_emin[tid]; if (v != INFBITS && (tid % 9 != 4 || !conv)) atomicMin(&s_bmin[tid % 9], v); }
        // resident kernel: the values must have arrived -- and the next tile's mark -- before a neighbour is told
        if constexpr (OWN) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (tid == 0) {
                int pr = (int)(s_best >> 32);                    // (own_decide's choice: nobody has touched s_best since)
                s_own[3] = (own_next >= 0 && own_take_resolve(own_next, pr, own_ro, own_rl)) ? pr : INFBITS;
#ifdef UFM_TIMING
                if (own_next >= 0 && s_own[3] >= INFBITS) atomicAdd(&g_sdiag[6], 1ull);
#endif
            }
        }
        __syncthreads();
        // the tile is free again: its values are in HBM (an activation that landed meanwhile has re-queued it already)
        if (OWN && tid == 9) { UFM_STRICT_RELEASE(); __hip_atomic_store(&P.own_lock[own_slot_now], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        if (tid < 9) {
            const int dr = tid / 3 - 1, dc = tid % 3 - 1;
            if (tid == 4) {
                if (!conv || s_bmin[4] != INFBITS) {           // sweep cap hit / border rose: come back
                    if (OWN) own_push(P, gt, min(s_bmin[4], INFBITS - 1), gt);
                    else activate(P, Q, k + 1, gt, min(s_bmin[4], INFBITS - 1));
                }
                P.hint[gt] = s_misc[3];
                // statistics: summed in this thread's registers, flushed once when the workgroup is done
                // (same-address atomics from 256 CUs are memory-side operations; five per visit add up)
                st_lmax = max(st_lmax, s_misc[3]);
                s_stat[0] += 1ull;
                s_stat[1] += (unsigned long long)s_misc[3];
                s_stat[2] += 16ull * (unsigned long long)s_misc[2];
            } else if (s_bmin[tid] != INFBITS) {
                const int ntx = tx + dr, nty = ty + dc;
                if (ntx >= 0 && ntx < P.TX && nty >= 0 && nty < P.TY) {
                    if (OWN) own_push(P, m * P.NTm + ntx * P.TY + nty, s_bmin[tid], gt);
                    else activate(P, Q, k + 1, m * P.NTm + ntx * P.TY + nty, s_bmin[tid]);
                }
            }
            if constexpr (OWN && DAG) if (tid != 4 && dag_nslot >= 0 && s_misc[0] && dag_mine < dag_nthr)      // first visit done: whoever waited for it has one less to wait for
                __hip_atomic_fetch_sub(&P.dag_left[dag_nslot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if constexpr (OWN) prev_gt = gt;
#ifdef UFM_TIMING
        if (MODE == MODE_LOWER) {
            const int dbg_ninf1 = __syncthreads_count(io_on && gf == INFINITY);
            if (tid == 0) {
                const unsigned long long tk3 = wall_clock64();
                if (OWN && gt < TILE_DIAG_MAX && (!EARLY || (s_qw[1] & 0x10000))) g_tile[1][gt] = (unsigned int)(tk3 - g_tile_t0);
                if (OWN && s_misc_vi < VIS_DIAG_MAX) g_vis[s_misc_vi][2] = (unsigned int)(tk3 - g_tile_t0);
                if (OWN && gt / P.TY < 2 && gt % P.TY < 2) plog((unsigned int)gt, 0xFFFFFFFBu, (conv ? 1u : 0u) | ((unsigned int)s_misc[3] << 8) | ((unsigned int)blockIdx.x << 20));
                if (EARLY) atomicAdd(&g_tdiag[6], (unsigned long long)(s_qw[1] & 0xFFFF));   // in-visit refreshes
                atomicAdd(&g_tdiag[0], tk1 - tk0); atomicAdd(&g_tdiag[1], tk2 - tk1); atomicAdd(&g_tdiag[2], tk3 - tk2);
                atomicAdd(&g_tdiag[3], 1ull);
                const unsigned long long bin = (tk3 - tk0) / 200;
                atomicAdd(&g_tdiag[8 + (bin < 31 ? bin : 31)], 1ull);
                atomicAdd(&g_tdiag[40 + (s_misc[3] < 23 ? s_misc[3] : 23)], 1ull);   // histogram of per-wave sweep counts / 1
                trace_rec(k, 0, tk0, tk3, (long long)(s_misc[3] & 255) | ((long long)min(s_misc[1], 255) << 8) | ((long long)min(dbg_hint, 255) << 16) | ((long long)dbg_ninf0 << 24) | ((long long)dbg_ninf1 << 40) | ((long long)dbg_rank << 52));
            }
        }
#endif
    }
    if (tid == 4 && s_stat[0]) {
        atomicMax(&P.lmax[k & (LMAX - 1)], st_lmax);
        atomicAdd(&P.ctr->tile_visits, s_stat[0]);
        if (MODE == MODE_RAISE) atomicAdd(&P.ctr->raise_visits, s_stat[0]);
        atomicAdd(&P.ctr->tile_iters, s_stat[1]);
        atomicAdd(&P.ctr->elem_evals, s_stat[2]);
    }
}

// Vectorised triage for long queues: one thread per queued tile decides "release now" (append to
// the ready list of the following relax launch) or "carry over" (same list ring as k_relax).
template <int MODE>
__global__ void k_triage(DevParams P, int k, float delta, float rbound) {
    constexpr int Q = (MODE == MODE_LOWER) ? Q_LOWER : Q_RAISE;
    const int r = k % 3, rn = (k + 1) % 3, pc = k & 1, pn = pc ^ 1;
    const int n = P.ctr->cnt[Q][r];
    const int *cand = P.cand + (size_t)(Q * 3 + r) * P.NT;
    const float theta = __int_as_float(P.ctr->lmin[Q][r]) + delta;
    const float rb = (rbound < 0.0f) ? P.ctr->rbound : rbound;
    // start keys of the first 64 maps once per workgroup: their loads (start elements -> G) then run
    // beside the list -> priority chain instead of behind it
    __shared__ float s_B[64];
    const int focused = P.dyn->focused;
    if (MODE == MODE_LOWER && focused) {
        if (threadIdx.x < 64 && (int)threadIdx.x < P.nmaps) s_B[threadIdx.x] = start_bound(P, threadIdx.x);
        __syncthreads();
    }
    // One list entry per thread; every append goes through one atomic per wave (ballot + popcount):
    // thousands of same-address atomics -- list cursors, the list minimum -- were most of this kernel
    for (int base = blockIdx.x * blockDim.x; base < n; base += gridDim.x * blockDim.x) {
        const int i = base + threadIdx.x;
        const bool valid = i < n;
        const int gt = valid ? cand[i] : 0;
        const int pbits = valid ? prio_read(P, Q, k, gt) : INFBITS;
        const int m = gt / P.NTm, t = gt - m * P.NTm;
        bool release = false, parked = false;
        if (valid) {
            if (MODE == MODE_LOWER) {
                const float B = focused ? (m < 64 ? s_B[m] : start_bound(P, m)) : INFINITY;
                const float hd = focused ? tile_heuristic(P, m, t / P.TY, t % P.TY) : 0.0f;
                parked = !(__int_as_float(pbits) + hd < B || B == INFINITY);
                release = !(__int_as_float(pbits) > theta) && !parked;
            } else {
                release = !(__int_as_float(pbits) > rb);
            }
        }
        // a launch lasts (work per CU) + (its longest visit) when long visits are handed out last;
        // tiles a front is still crossing (first visit of the step, or many sweeps last time) go first
        const bool lng = release && UFM_LPT && (P.touched[gt] == 0 || P.hint[gt] >= UFM_LONG_SWEEPS);
#ifdef UFM_TIMING
        if (release && MODE == MODE_LOWER) {
            const float lo = __int_as_float(P.ctr->lmin[Q][r]);
            P.rank[gt] = (int)fminf(255.0f, fmaxf(0.0f, 255.0f * (__int_as_float(pbits) - lo) / fmaxf(delta, 1e-6f)));
        }
#endif
        // What an entry becomes -- released (long / short), parked, carried -- is decided above; the words that say whether
        // a park / carry is the tile's first are swapped next, all of them in flight together; then the four list cursors
        // are advanced by ONE instruction (lanes 0..3, one cursor each, the counts from ballots) instead of four returning
        // atomics one after the other: this kernel is a chain of dependent memory round trips, once per band step of a plan.
        const bool sht = release && !lng;
        const bool prk = valid && !release && (parked || MODE == MODE_RAISE);
        const bool carry = valid && !release && !prk;
        bool fresh = false, first = false;
        if (prk) {
            atomicMin(&P.pprio[Q * P.NT + gt], pbits);
            fresh = atomicExch(&P.pflag[Q * P.NT + gt], 1) == 0;
        }
        if (carry) {     // (nobody has queued anything for launch k + 1 yet -- its list fills while launch k runs, after this kernel --
            //              and a list holds a tile once: every carry is its tile's first entry there, no need to ask)
            atomicMin(&P.prio[(size_t)(Q * 2 + pn) * P.NT + gt], prio_key(k + 1, pbits));
            P.queued[(size_t)(Q * 2 + pn) * P.NT + gt] = k + 2;
            first = true;
        }
        int wmin = carry ? pbits : INFBITS;
        for (int off = 32; off; off >>= 1) wmin = min(wmin, __shfl_xor(wmin, off));
        const int lane = threadIdx.x & 63;
        if (wmin != INFBITS && lane == 0) atomicMin(&P.ctr->lmin[Q][rn], wmin);
        const unsigned long long m0 = __ballot(lng), m1 = __ballot(sht), m2 = __ballot(fresh), m3 = __ballot(first);
        int slot0 = 0;
        if (lane < 4) {
            const unsigned long long mk = lane == 0 ? m0 : (lane == 1 ? m1 : (lane == 2 ? m2 : m3));
            int *ctr = lane == 0 ? &P.ctr->nready[k & 1] : (lane == 1 ? &P.ctr->nshort[k & 1] : (lane == 2 ? &P.ctr->npark[Q] : &P.ctr->cnt[Q][rn]));
            if (mk) slot0 = atomicAdd(ctr, __popcll(mk));
        }
        const unsigned long long below = (1ull << lane) - 1ull;
        const int b0 = __shfl(slot0, 0), b1 = __shfl(slot0, 1), b2 = __shfl(slot0, 2), b3 = __shfl(slot0, 3);
        if (lng) P.ready[b0 + __popcll(m0 & below)] = gt;
        if (sht) P.ready[P.NT - 1 - (b1 + __popcll(m1 & below))] = gt;
        if (fresh) P.park[(size_t)(Q * 2) * P.NT + b2 + __popcll(m2 & below)] = gt;
        const int sc = b3 + __popcll(m3 & below);
        if (first) P.cand[(size_t)(Q * 3 + rn) * P.NT + sc] = gt;
    }
}

