// ufm_region.h -- region-resident replan (included by ufm_engine.hip; shares its operators, queues and layout).
//
// What it replaces inside the engine: the launch chain of a replan -- k_replan_begin, ~9 invalidation launches,
// k_raise_to_lower, ~9 lowering launches, k_replan_end, each launch a handful of tile visits on a 256-CU chip
// (~230 us per replan, slower than one CPU core) -- i.e. the reference's update() + plan() after one patch_map
// (FieldDPlanner_impl.h:118-163, :23-66; DynamicFastMarching_impl.h:13-132).
//
// How: a replan touches a few thousand elements around the patch.  ONE workgroup stages a block of up to
// RTMAX x RTMAX tiles around the patch (field + 1-element frame + cost bytes, ~135 KB of the CU's 160 KB LDS) and
// runs both phases there to their fixed point -- invalidation of what lost its support, then lowering -- with the
// same per-patch asynchronous sweeps, wake bits and update operators as a tile visit of k_relax, but with no
// kernel boundary, no HBM round trip and no queue traffic between the dependent steps.  The reference's
// end_condition is honoured per element: a lowering result at or beyond the start's key (read from LDS, it moves
// while the phase runs) and an invalidation of a value beyond the invalidation bound are not applied; their tiles
// are parked in the persistent queues exactly as a tile visit would leave them.  Afterwards the block is written
// back (changed values, neighbour rings), tiles outside the block whose halo changed are queued, and the same
// device-side end condition as k_replan_end decides: nothing left below the start's key -> done, counters
// published to the spinning host; otherwise the host carries on with the ordinary launch chain from the queues.
// A sweep budget bounds every wave (livelock guard): what is still dirty when it runs out goes to the queues too.

#ifndef UFM_REGION_IDLE_SLEEP_RAISE
#define UFM_REGION_IDLE_SLEEP_RAISE 8    // ... in the node planners' invalidation phase, whose sweeps are a few loads and a compare
#endif
// Waves that sweep run at a higher issue priority than waves that look for work (s_setprio): the looks of the idle waves of a SIMD do not
// take the issue slots of the one that is on the chain
#ifndef UFM_REGION_PRIO
#define UFM_REGION_PRIO 2
#endif
#if UFM_REGION_PRIO
#define UFM_REGION_SETPRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define UFM_REGION_SETPRIO(x) do {} while (0)
#endif
#ifndef UFM_REGION_IDLE_SLEEP
#define UFM_REGION_IDLE_SLEEP 8          // an idle wave of the block kernel looks at its wake words this often (x 64 clocks): 1 / 4 / 16 / 32 / 64 -> 100 replans 18.9 / 18.0 / 17.6 / 17.6 / 17.8 ms (the looks of twelve idle waves take issue slots and LDS cycles from the four that sweep)
#endif
constexpr int RTMAX = 128 / T;            // block edge in tiles (8 for 16 x 16 tiles: field 71 KB + cost bytes 17 KB + back-pointer codes 16 KB of the CU's 160 KB LDS)
constexpr int RN = RTMAX * T;             // ... in elements (160)
constexpr int RP = RN + 8;                // LDS pitch of the block's field: rows 4 apart on distinct banks (168 = 5*32 + 8)
constexpr int RCP = RN + 4;               // pitch of the cost bytes
constexpr int RBP = RN;                   // pitch of the back-pointer codes
constexpr int TP = T / 4;                 // 4x4-element patches per tile side
constexpr int RPW = RTMAX * TP / 4;       // patches per wave per side: patch (pr, pc) belongs to wave ((pr & 3) << 2 | (pc & 3))
constexpr int RWW = (RPW * RPW + 31) / 32;   // wake words per wave
constexpr int RFRAME = 4 * RTMAX + 4;     // tiles around the block

struct RegionJob {
    ReplanBegin rb;                       // step bookkeeping, consumed patch rectangles, invalidation queue index, margin
    DevDyn dyn;
    int tx0, ty0, ntx, nty;               // the block, in tiles
    int k_lower;                          // index of the next lowering launch (list the activations go to)
    unsigned int seq;                     // sequence number to publish with
    int max_sweeps;                       // sweep budget per wave and phase
    int debug;                            // diagnostics.  bit 0: no end_condition gating inside the block; bit 1: the end check's inputs, event
                                          // counts and a phase timeline go to the diagnostics array (ufm_debug_lmax, tools/replan_timeline.py)
    float slack;                          // lowering results up to this far beyond the start's key are still applied (see Bgate)
    float delta;                          // width of the ordering band of the lowering sub-rounds (cost units; +inf: no ordering)
    int map;                              // the map of a batch this job belongs to
    int batch;                            // part of a batch step: the host has done the step bookkeeping, the last workgroup publishes
    const uint8_t *psrc[4];               // per consumed rectangle: the patch's bytes if it has NOT been applied yet (a patch handed over from host memory:
                                          // the bytes sit in host-coherent pinned memory and this kernel does Graph::update + the seeding itself, no patch
                                          // kernel, no copy, no stream synchronisation in front of the replan); nullptr: applied by k_patch_small at the call
};
// A replan round of a batch (BASELINE config 4: every map of the GPU's share gets its own patch): one workgroup per
// map, all in one launch.
constexpr int RJOBS = 8;
struct RegionJobs { int n; RegionJob j[RJOBS]; };

// a priority this very kernel may have written (agent-scope load: past the CU's L1)
__device__ __forceinline__ int prio_read_fresh(const DevParams &P, int qz, int kk, int gt) {
    const unsigned long long v = __hip_atomic_load(&P.prio[(size_t)(qz * 2 + (kk & 1)) * P.NT + gt], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (int)(v >> 32) == 0x7FFFFFFF - kk ? (int)(unsigned int)v : INFBITS;
}

// counters the block kernel shares between its phases (LDS)
struct RegionShared {
    int wake[16][RWW];                    // per wave: patches with new inputs
    int ever[16][RWW];                    // patches swept by the invalidation phase (they are re-lowered)
    int renew[16][RWW];                   // patches whose back-pointers are renewed after the phases: those in which a value changed, and the patches around them
    int tbp[RTMAX * RTMAX];               // tile holds renewed back-pointers (they are written back with the tile)
    int seed[16][RWW];                    // node planners: patches holding elements of the consumed rectangles whose parent triangle has not been re-evaluated yet
    int defer[2][16][RWW];                // patches holding results beyond the bound (lower / raise)
    int dprio[2][RTMAX * RTMAX];          // per tile: smallest deferred priority (float bits)
    int tflag[RTMAX * RTMAX];             // tile has changed values
    int traised[RTMAX * RTMAX];           // the invalidation took a value away in this tile: then it takes away EVERY unsupported one of the tile
    int actL[RFRAME], actR[RFRAME];       // per tile of the frame: smallest lowered / invalidated border value next to it
    int soff[4];                          // start elements inside the block: LDS offset (-1: outside / unused)
    float sdist[4];                       // hm * dist(start, element)
    float B0, rbound;
    float theta;                          // ordering: lowering results at / above it wait for a later sub-round (tile-engine band, per patch)
    float Bgate;                          // min(Bsub + slack, theta): what the bursts compare with.  The slack is one maximal
                                          // traversal step: every element NEXT to one inside the bound still gets its value, as in the
                                          // reference, where expanding an element computes the RHS of all its neighbours (the path
                                          // extractor reads RHS around the start: FieldDPlanner_impl.h:44-52, PathExtraction impl:76-77)
    int rmin;                             // smallest value the invalidation took away (float bits)
    float Bsub;                           // the bound lowering results are held against during the current sub-round
    int swas[4];                          // start element held a finite value when the kernel started
    int idle, giveup, again, any_start_in;
    unsigned long long sweeps;
    int expanded;
    int m_r, m_l, done;
    int dkey;                             // smallest KEY (value + hm * dist) lowering held back in the current sub-round (float bits)
    int dbg[8];                           // diagnostics
    int dbg2[8];                          // diagnostics: [0] lowering bursts that changed nothing, [1] ... whose patch and its 1-element surround held nothing but +inf, [2] invalidation bursts that changed nothing
    unsigned long long tstamp[12];        // diagnostics: wall_clock64 at the phase boundaries
};

// the start's key from the block (elements outside the block cannot change while the kernel runs: B0 covers them)
__device__ __forceinline__ float region_start_key(const float *Gs, const RegionShared &S) {
    if (!S.any_start_in) return S.B0;
    float b = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int o = S.soff[i];
        if (o < 0) continue;
        const float g = Gs[o];
        if (g < INFINITY) b = fmaxf(b, g + S.sdist[i]);
    }
    return b > 0.0f ? b : INFINITY;
}

// The bound of a lowering sub-round.  It is FIXED while the sub-round runs -- so that what gets applied does not depend
// on which wave looked at the start's key when (results reproducible from run to run) -- and it is the start's key
// unless a start element that had a value when the kernel began has lost it to the invalidation and not yet got it
// back: then nobody knows the key yet, and the invalidation bound (old key + margin) stands in for it, instead of
// "no bound at all" (which floods the whole block, every time the patch lies on the start -- as it does in the
// harness's reveal-around-the-robot patches).  The check at quiescence re-opens what a too small bound held back.
__device__ __forceinline__ float region_lower_bound(const float *Gs, const RegionShared &S) {
    if (!S.any_start_in) return S.B0;
    float b = 0.0f;
    bool pending = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int o = S.soff[i];
        if (o < 0) continue;
        const float g = Gs[o];
        if (g < INFINITY) b = fmaxf(b, g + S.sdist[i]);
        else if (S.swas[i]) pending = true;
    }
    if (pending) return fmaxf(b, S.rbound);
    return b > 0.0f ? b : INFINITY;
}

// One phase (MODE_RAISE: invalidation, MODE_LOWER: lowering) of the block to quiescence.  All 16 waves call.
template <int ALGO, int MODE>
__device__ void region_phase(const DevParams &P, const RegionJob &J, float *Gs, const uint8_t *Cb, uint8_t *Bb, RegionShared &S,
                             int thr, float hm, int focused, int goal_lx, int goal_ly) {
    constexpr bool BPRAISE = MODE == MODE_RAISE && !is_dfm<ALGO>;   // invalidation along the stored back-pointers (k_relax)
    const int tid = threadIdx.x, w = wave_index16(tid >> 6), lane = tid & 63;
    const int q = lane & 3, nd = lane >> 2;
    const int nprow = J.ntx * TP, npcol = J.nty * TP;
    unsigned long long wake_sel = 0ull;
    if (lane < 9) {
        const int dr = lane / 3 - 1, dc = lane % 3 - 1;
        wake_sel = ~0ull;
        if (dr < 0) wake_sel &= 0x000000000000FFFFull; else if (dr > 0) wake_sel &= 0xFFFF000000000000ull;
        if (dc < 0) wake_sel &= 0x000F000F000F000Full; else if (dc > 0) wake_sel &= 0xF000F000F000F000ull;
    }
    const int colour = ((nd >> 2) & 1) | ((nd & 1) << 1);
    // a patch's 3 x 3 neighbours: patch (pr, pc) belongs to wave ((pr & 3) << 2) | (pc & 3) and is bit (pr >> 2) * RPW + (pc >> 2) of that wave's words; this
    // wave's patches all have (pr & 3, pc & 3) = (w >> 2, w & 3), so the neighbour's wave and the distance of its bit from this patch's bit are per-lane constants
    const bool wt_lane = lane < 9;
    const int wt_dr = wt_lane ? lane / 3 - 1 : 0, wt_dc = wt_lane ? lane % 3 - 1 : 0;
    const int wt_ar = (w >> 2) + wt_dr, wt_ac = (w & 3) + wt_dc;                       // -1 .. 4
    const int wt_wave = ((wt_ar & 3) << 2) | (wt_ac & 3), wt_di = (wt_ar >> 2) * RPW + (wt_ac >> 2);
    const float sx = J.rb.sb.sx, sy = J.rb.sb.sy;
    const int rx0 = J.tx0 * T, ry0 = J.ty0 * T;
    int budget = J.max_sweeps;               // (per wave and phase)
    unsigned int my_sweeps = 0;
    auto cost_at = [=](int r, int c) { const int b = Cb[r * RCP + c]; return b >= thr ? INFINITY : (float)b; };
    // what does not change while a phase runs (thread 0 sets it between the phases): read once, not at every burst
    const int so0 = __builtin_amdgcn_readfirstlane(S.soff[0]), so1 = __builtin_amdgcn_readfirstlane(S.soff[1]);
    const int so2 = __builtin_amdgcn_readfirstlane(S.soff[2]), so3 = __builtin_amdgcn_readfirstlane(S.soff[3]);
    // (J lives in the kernel's argument segment: a field of it read inside the sweep loop is a scalar memory load, waited for, in
    //  every sweep -- J.debug was)
    const bool dbg_nogate = (J.debug & 1) != 0, dbg_count = (J.debug & 2) != 0;
    const float Bphase = (MODE == MODE_LOWER) ? S.Bgate : INFINITY;
    const float rb_phase = S.rbound + J.slack;
    // words of this wave that can hold a bit at all (a block smaller than RTMAX x RTMAX leaves the upper ones empty)
    const int nwords = min(RWW, (((nprow + 3) / 4 - 1) * RPW + (npcol + 3) / 4 + 31) / 32);
    UFM_REGION_SETPRIO(UFM_REGION_PRIO);
    for (;;) {
        bool took = false;
        bool vote = __hip_atomic_load(&S.giveup, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
        for (int wd = 0; wd < nwords && !vote; ++wd) {
            // (looked at with a plain load first: a returning LDS atomic per word and look, most of them on empty words, was a fifth
            //  of the time between two bursts)
            if (__hip_atomic_load(&S.wake[w][wd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) continue;
            int bits = 0;
            if (lane == 0) bits = atomicExch(&S.wake[w][wd], 0);
            bits = __builtin_amdgcn_readfirstlane(bits);
            if (!bits) continue;
            // (Round 4: re-lowering only the patches that LOST a value, and the seeded ones, starts the lowering phase with a quarter fewer patches and
            //  ends with as many bursts -- the neighbours' wake bits bring the others in anyway: 16.4 ms per 100 replans either way.)
            if (MODE == MODE_RAISE && lane == 0) S.ever[w][wd] |= bits;      // only this wave writes its own words
            // Invalidation of the node planners follows the stored back-pointers: an element is gone when a vertex its value depends on is gone
            // (the byte's dep bits: two LDS loads and a compare, no operator, no per-patch constants).  Only where cell costs changed -- the
            // patches of the consumed rectangles, once -- is the parent triangle itself evaluated with the new costs (eval_quad_bp).
            int seedbits = 0;
            if (BPRAISE) seedbits = __builtin_amdgcn_readfirstlane(S.seed[w][wd]) & bits;
            while (bits) {
                const int j = __ffs(bits) - 1;
                if (budget <= 0) {                                           // out of budget: leave the rest for the queues
                    if (lane == 0) {
                        __hip_atomic_fetch_or(&S.wake[w][wd], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_store(&S.giveup, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    vote = true;
                    break;
                }
                bits &= bits - 1;
                took = true;
                const int idx = wd * 32 + j;
                const int pr = (idx / RPW) * 4 + (w >> 2), pc = (idx % RPW) * 4 + (w & 3);
                if (pr >= nprow || pc >= npcol) continue;                    // (never woken; a block smaller than RTMAX)
                const int lx = pr * 4 + (nd >> 2), ly = pc * 4 + (nd & 3);
                const bool arith = BPRAISE && ((seedbits >> j) & 1);            // (wave-uniform)
                QuadConsts<ALGO> C;
                if (!BPRAISE || arith) C.load_at(cost_at, lx, ly, q, RP);
                float *ctr = Gs + (lx + 1) * RP + ly + 1;
                uint8_t *bpp = Bb + lx * RBP + ly;
                const int bpb = BPRAISE ? *bpp : BP_NONE;
                // the two vertices of the element's parent triangle (LDS offsets) and whether its value depends on them
                const int bqc = bpb >> 3, bsx = (bqc & 2) ? RP : -RP, bsy = (bqc & 1) ? 1 : -1;
                const int bo1 = (bpb & 4) ? bsy : bsx, bo2 = bsx + bsy;
                const bool bd1 = bpb & 1, bd2 = bpb & 2;
                const bool goal = (lx == goal_lx) & (ly == goal_ly);
                // wake targets: lane 0..8 = the 3x3 patches around this one (lane 4: itself).  Which wave a neighbouring patch belongs to and how far its
                // bit lies from this patch's own do not depend on the patch (wt_*, above the loop): per burst only the position and the range test are left
                const int npr = pr + wt_dr, npc = pc + wt_dc, ni = idx + wt_di;
                const bool wt_ok = wt_lane & ((unsigned)npr < (unsigned)nprow) & ((unsigned)npc < (unsigned)npcol);
                const int nwave = wt_wave, nword = wt_ok ? (ni >> 5) : -1, nbit = 1 << (ni & 31);
                // admissible part of the key that does not depend on the value: hm * dist(start, patch)
                float hd = 0.0f;
                if (hm != 0.0f) {
                    const float x0 = (float)(rx0 + pr * 4), x1 = x0 + 3.0f, y0 = (float)(ry0 + pc * 4), y1 = y0 + 3.0f;
                    const float dx = fmaxf(fmaxf(x0 - sx, sx - x1), 0.0f), dy = fmaxf(fmaxf(y0 - sy, sy - y1), 0.0f);
                    hd = hm * hypotf(dx, dy) * 0.999f;
                }
                const int tl = (pr / TP) * J.nty + (pc / TP);
                // the sub-round's bound (region_lower_bound); a start element itself is never held back
                const float B = Bphase;
                const int my_off = (lx + 1) * RP + ly + 1;
                const bool is_start = (my_off == so0) | (my_off == so1) | (my_off == so2) | (my_off == so3);
                // Invalidation is held back beyond the bound per TILE, like a tile visit of k_relax: once a tile has lost a value,
                // every unsupported value in it goes, whatever its size.  (Holding back single elements would leave unsupported
                // values NEXT to invalidated ones: the lowering phase -- this one with its one-move slack, or the launch chain's
                // tile visits, which do not look at single values -- would then rebuild the hole from them.)
                const float rb = (MODE == MODE_RAISE && S.traised[tl]) ? INFINITY : rb_phase;
                asm volatile("" ::: "memory");
                float g = ctr[0];
                if (dbg_count && MODE == MODE_LOWER) {
                    const int sx_ = (q & 2) ? RP : -RP, sy_ = (q & 1) ? 1 : -1;
                    const bool fin = (g < INFINITY) | (ctr[sx_] < INFINITY) | (ctr[sy_] < INFINITY) | (ctr[sx_ + sy_] < INFINITY);
                    if (__builtin_amdgcn_ballot_w64(fin) == 0ull && lane == 0) atomicAdd(&S.dbg2[1], 1);
                }
                float dmin = INFINITY;                                        // smallest priority this lane deferred
                float rmin_l = INFINITY;                                      // smallest value this lane invalidated
                bool again = true, pch = false;
                [[maybe_unused]] bool held_last = false;
                [[maybe_unused]] float nv_last = INFINITY;
                [[maybe_unused]] const bool gate_free = is_start || B == INFINITY || dbg_nogate;      // lowering: a start element itself is never held back
                unsigned long long chg = 0ull;                              // lanes whose node the burst changed
                int cnt = 0;
                for (int b = 0; b < 16 && again; ++b) {
                    asm volatile("" ::: "memory");                          // re-read the block every sweep (other waves write it)
                    float nv;
                    if constexpr (BPRAISE) {
                        if (arith && b == 0) nv = quad_min(eval_quad_bp<ALGO, RP>(ctr, q, C, bpb));
                        else { const float p1 = ctr[bo1], p2 = ctr[bo2]; nv = ((bd1 && p1 == INFINITY) || (bd2 && p2 == INFINITY)) ? INFINITY : g; }   // (both loads asked for at once)
                    }
                    else nv = quad_min(eval_quad<ALGO, RP>(ctr, q, C));
                    if (goal) nv = 0.0f;
                    bool want, gate, doit;
                    if (MODE == MODE_LOWER) {
                        // (lane conditions as `&&` / `||` of comparisons: written with `&` / `|` next to the wave-uniform flags they were materialised as
                        //  0 / 1 integers and combined by vector instructions, a dozen per sweep)
                        want = (nv != g);
                        gate = (nv + hd < B) || gate_free;                     // end_condition: results at / beyond the start's key wait
                        doit = want && gate && ((nv < g) || (colour == (cnt & 3)));
                        held_last = want && !gate; nv_last = nv;              // (what the LAST sweep holds back is what the burst leaves pending: dmin below)
                    } else {
                        if (is_dfm<ALGO>) want = (g < INFINITY) && (nv > g) && ((nv == INFINITY) || (__float_as_int(nv) - __float_as_int(g) > 8));
                        else want = (g < INFINITY) && (nv > g);
                        gate = !(g > rb) || dbg_nogate;                                              // beyond the invalidation bound: wait
                        doit = want && gate;
                        if (want && !gate) dmin = fminf(dmin, g);
                        rmin_l = doit ? fminf(rmin_l, g) : rmin_l;        // (what the burst took away: into S.rmin / S.traised after it, not per sweep)
                        nv = INFINITY;
                    }
                    if (doit && q == 0) ctr[0] = nv;
                    const float gn = doit ? nv : g;
                    const unsigned long long mask = __builtin_amdgcn_ballot_w64(doit);            // (= gn != g: what is applied differs from g in either mode)
                    const unsigned long long wanted = __builtin_amdgcn_ballot_w64(want && gate);  // lanes not yet settled (a colour-gated rise waits for its sweep)
                    g = gn;
                    pch |= mask != 0ull;
                    chg |= mask;
                    UFM_SWEEP_FENCE();                                        // value before wake bit
                    if ((mask & wake_sel) != 0ull && nword >= 0 && lane != 4)
                        __hip_atomic_fetch_or(&S.wake[nwave][nword], nbit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    ++cnt;
                    again = wanted != 0ull;
                }
                budget -= cnt; my_sweeps += cnt;
                if (pch && lane == 0) S.tflag[tl] = 1;                      // the tile holds changed values (write-back looks at these tiles only)
                if (MODE == MODE_LOWER && held_last) dmin = nv_last;
                // (the patch and the eight around it: lanes 0..8 hold their wake words anyway)
                // (... of the patches around it only those that border a node the burst changed -- the wake-up's own test: a node none of whose eight
                //  neighbours has changed keeps its parent triangle.  Round 4; before, all eight: a third more patches to renew.)
                if ((chg & wake_sel) != 0ull && nword >= 0) __hip_atomic_fetch_or(&S.renew[nwave][nword], nbit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                // (a seeded patch that had to hold a result back is evaluated once more when it is woken again)
                if (BPRAISE && arith) {
                    const unsigned long long held = __builtin_amdgcn_ballot_w64(dmin < INFINITY);
                    if (lane == 0 && held == 0ull) S.seed[w][wd] &= ~(1 << j);
                }
                if (MODE == MODE_RAISE && rmin_l < INFINITY && q == 0) { atomicMin(&S.rmin, __float_as_int(rmin_l)); S.traised[tl] = 1; }
                if (dbg_count && lane == 0) { atomicAdd(&S.dbg[MODE == MODE_LOWER ? 5 : 4], 1); atomicAdd(&S.dbg[MODE == MODE_LOWER ? 7 : 6], cnt); if (!pch) atomicAdd(&S.dbg2[MODE == MODE_LOWER ? 0 : 2], 1); }
                if (again && lane == 0) __hip_atomic_fetch_or(&S.wake[w][wd], 1 << j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // burst cap
                if (dmin < INFINITY && q == 0) {
                    atomicAdd(&S.dbg[MODE == MODE_LOWER ? 0 : 1], 1);
                    atomicMin(&S.dprio[MODE == MODE_LOWER ? 0 : 1][tl], __float_as_int(dmin));      // the tile's priority when it is parked
                    if (MODE == MODE_LOWER) atomicMin(&S.dkey, __float_as_int(dmin + hd));          // what the gate compared with the start's key
                    __hip_atomic_fetch_or(&S.defer[MODE == MODE_LOWER ? 0 : 1][w][wd], 1 << j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        if (took && !vote) continue;
        if (!vote) {                                                         // nothing to do: idle until woken or all idle
            if (lane == 0) atomicAdd(&S.idle, 1);
            UFM_REGION_SETPRIO(0);
            for (;;) {
                __builtin_amdgcn_s_sleep((MODE == MODE_RAISE && !is_dfm<ALGO>) ? UFM_REGION_IDLE_SLEEP_RAISE : UFM_REGION_IDLE_SLEEP);
                // (all loads first, then the decisions: one LDS round trip per look)
                const int idle_now = __hip_atomic_load(&S.idle, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int gave_up_now = __hip_atomic_load(&S.giveup, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                int any = 0;
#pragma unroll
                for (int wd = 0; wd < RWW; ++wd) any |= __hip_atomic_load(&S.wake[w][wd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (idle_now >= 16 || gave_up_now != 0) { vote = true; break; }
                if (any) { if (lane == 0) atomicSub(&S.idle, 1); break; }
            }
            UFM_REGION_SETPRIO(UFM_REGION_PRIO);
            if (!vote) continue;
        }
        // vote: everybody arrives first, then the wake bits are stable
        __syncthreads();
        int mine = 0;
#pragma unroll
        for (int wd = 0; wd < RWW; ++wd) mine |= __hip_atomic_load(&S.wake[w][wd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const int work = __syncthreads_or(mine != 0);
        const int gave_up = __hip_atomic_load(&S.giveup, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (gave_up || !work) break;
        if (tid == 0) S.idle = 0;
        __syncthreads();
    }
    UFM_REGION_SETPRIO(0);
    if (lane == 0 && my_sweeps) atomicAdd(&S.sweeps, (unsigned long long)my_sweeps);
    __syncthreads();
    if (tid == 0) S.idle = 0;
    __syncthreads();
}

template <int ALGO>
__global__ __launch_bounds__(NTHR) void k_replan_region(DevParams P, RegionJobs JS, DevCounters *host, unsigned int *flag) {
    __shared__ float Gs[(RN + 2) * RP];
    __shared__ uint8_t Cb[(RN + 1) * RCP];
    __shared__ __attribute__((aligned(16))) uint8_t Bb[RN * RBP];
    __shared__ RegionShared S;
    __shared__ int s_last;
#ifdef UFM_REGION_NG0
    constexpr int NG0 = UFM_REGION_NG0;              // (test builds: a small copy, so that the path of the tiles beyond it runs)
#else
    constexpr int NG0 = (46 * 1024) / (TT * 4);      // tiles of the write-back's LDS copy of what HBM holds: what the CU's 160 KB leave (46 KB; a 6 x 6 block needs 36)
#endif
    __shared__ float G0c[NG0 * TT];
    __shared__ int s_wbl[RTMAX * RTMAX];  // write-back: the tiles that have something to write
    __shared__ int s_nwb;
    __shared__ uint8_t s_pmask[4096];     // change mask of a patch that comes with the job (<= 64 x 64 cells)
    __shared__ int s_simd[16];               // diagnostics: the SIMD each wave runs on
    __shared__ unsigned long long s_tb[8];   // diagnostics: the prologue's timeline
    constexpr bool CELLS = is_dfm<ALGO>;
    constexpr int COFF = CELLS ? 0 : 1;
    const int tid = threadIdx.x;
    const RegionJob &J = JS.j[blockIdx.x];
    const int m = J.map, gt0 = m * P.NTm;                     // the job's map and its first tile
    const int ntl = J.ntx * J.nty;
    // tile index of the block -> its row: tl / nty as a multiplication (exact for tl < 256, nty <= 8): the staging and write-back loops divided once per
    // iteration and thread, ~25 instructions each
    const int nty_magic = (65536 + J.nty - 1) / J.nty;
    const int rx0 = J.tx0 * T, ry0 = J.ty0 * T, rnx = J.ntx * T, rny = J.nty * T;

    const unsigned long long t_begin = wall_clock64();
#define STAMP(k) do { if (tid == 0 && (J.debug & 2)) s_tb[k] = wall_clock64(); } while (0)
    STAMP(0);
    if ((tid & 63) == 0 && (J.debug & 2)) s_simd[tid >> 6] = (int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);   // HW_ID.SIMD_ID
    const float hm = J.dyn.hm;
    const int thr = J.dyn.thr, focused = J.dyn.focused;
    // ---- 0. Everything the kernel needs from memory is asked for FIRST, before anything is waited for (round 4: as five dependent steps --
    // cost bytes, patch bytes from host memory, the seeding's returning atomics, the start elements' values behind the step bookkeeping's
    // stores, the tiles -- the prologue took 14 us; `tools/replan_timeline.py`): the bytes of the first patch that comes with the job (host
    // memory: the longest trip), the start elements' values (for the start's key before the patch), the count of pending seeds.
    int r_first = -1;                                       // the first rectangle whose patch this kernel applies
    for (int r = J.rb.nrect - 1; r >= 0; --r) if (J.psrc[r]) r_first = r;
    // (one patch, applied here, nothing else pending: the common replan.  Its seeding needs no marks and no seed list -- see 0b)
    const bool one_fused = J.rb.nrect == 1 && r_first == 0 && !J.batch;
    int pb0[4] = {0, 0, 0, 0};
    if (r_first >= 0) {
        const int n0 = J.rb.rect[r_first][3] * J.rb.rect[r_first][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { const int e = tid + c * NTHR; if (e < n0) pb0[c] = J.psrc[r_first][e]; }
    }
    float sg = INFINITY;
    int se_x = -1, se_y = -1;             // threads 0..3: this thread's start element (-1: none)
    {   // (not J.rb.sb.start[tid]: indexing the kernel's arguments by thread is a memory load, waited for with the host's bytes in front of it)
        const int e = tid == 0 ? J.rb.sb.start[0] : (tid == 1 ? J.rb.sb.start[1] : (tid == 2 ? J.rb.sb.start[2] : (tid == 3 ? J.rb.sb.start[3] : -1)));
        if (e >= 0) { se_x = e / P.EY; se_y = e - se_x * P.EY; sg = P.G[gaddr(P, m, se_x, se_y)]; }
    }
    const int n_pending = P.ctr->scount;
    const int goal_x = P.goal[2 * m], goal_y = P.goal[2 * m + 1];      // (asked for here: read where it is used, in front of the phases, it was a memory round trip of its own)
    // ---- 0a. stage the block: the cost bytes (a patch that comes with the job is applied on top of them), the tiles (contiguous 1 KB each)
    // with their back-pointer bytes, the 1-element frame around the block ----
    // (Work is dealt out by wave -- a row piece of 64 cost bytes, a quarter KB of a tile -- so that which row / which tile is the wave's scalar
    //  arithmetic, and every load of a thread is asked for before the first one is waited for: dealt out element by element, a division and
    //  a memory round trip per element and thread, this staging took 12 us of the kernel's 160.)
    STAMP(1);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63;
    {
        const int crow = rnx + COFF, ccol = rny + COFF;
        constexpr int KP = (RN + 64) / 64, KR = (RN + 16) / 16;          // 64-byte pieces of a row (3), rows per wave (9)
        const uint8_t *cbase = P.cost + (size_t)m * P.cstride;
        uint8_t cv[KP * KR];
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            const int c = j * 64 + ln, cy = ry0 + c - COFF;
            const bool cok = c < ccol && cy >= 0 && cy < P.W;
#pragma unroll
            for (int k = 0; k < KR; ++k) {
                const int r = wv + 16 * k, cx = rx0 + r - COFF;
                cv[j * KR + k] = 255;
                if (j * 64 < ccol && r < crow && cx >= 0 && cx < P.L && cok) cv[j * KR + k] = cbase[(size_t)cx * P.W + cy];
            }
        }
        constexpr int CPT = TT / 256, KT = RTMAX * RTMAX * CPT / 16;                 // CPT: 64-lane float4 pieces per tile
        float4 gv[KT];
        unsigned int bv[KT];
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            const int pc = wv + 16 * k, tl = pc / CPT;
            if (tl >= ntl) continue;
            const int ti = (tl * nty_magic) >> 16, tj = tl - ti * J.nty, v = (pc - tl * CPT) * 64 + ln;
            const size_t gt = (size_t)(gt0 + (J.tx0 + ti) * P.TY + J.ty0 + tj);
            gv[k] = reinterpret_cast<const float4 *>(P.G + gt * TT)[v];
            bv[k] = reinterpret_cast<const unsigned int *>(P.bp + gt * TT)[v];
        }
        // the frame (at most 4 * RN + 4 elements: one per thread)
        static_assert(4 * RN + 4 <= NTHR, "one frame element per thread");
        float fv = INFINITY;
        int fo = -1;
        if (tid < 2 * (rny + 2) + 2 * rnx) {
            int hx, hy;
            if (tid < rny + 2) { hx = -1; hy = tid - 1; }
            else if (tid < 2 * (rny + 2)) { hx = rnx; hy = tid - (rny + 2) - 1; }
            else if (tid < 2 * (rny + 2) + rnx) { hx = tid - 2 * (rny + 2); hy = -1; }
            else { hx = tid - 2 * (rny + 2) - rnx; hy = rny; }
            const int x = rx0 + hx, y = ry0 + hy;
            fo = (hx + 1) * RP + hy + 1;
            if (x >= 0 && y >= 0 && x < P.TX * T && y < P.TY * T) fv = P.G[gaddr(P, m, x, y)];
        }
        STAMP(7);
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            const int c = j * 64 + ln;
#pragma unroll
            for (int k = 0; k < KR; ++k) {
                const int r = wv + 16 * k;
                if (j * 64 < ccol && r < crow && c < ccol) Cb[r * RCP + c] = cv[j * KR + k];
            }
        }
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            const int pc = wv + 16 * k, tl = pc / CPT;
            if (tl >= ntl) continue;
            const int ti = (tl * nty_magic) >> 16, tj = tl - ti * J.nty, e = ((pc - tl * CPT) * 64 + ln) * 4;
            float *g = &Gs[(ti * T + e / T + 1) * RP + tj * T + e % T + 1];
            g[0] = gv[k].x; g[1] = gv[k].y; g[2] = gv[k].z; g[3] = gv[k].w;
            *reinterpret_cast<unsigned int *>(&Bb[(ti * T + e / T) * RBP + tj * T + e % T]) = bv[k];
        }
        if (fo >= 0) Gs[fo] = fv;
    }
    // ---- 0b. Graph::update (Graph.cpp:36-51) + the seeding of update() for the patches that come with the job -- what k_patch_small does for a
    // patch applied at the call.  Every rectangle lies inside the block (place_job), so the old bytes are in Cb: compare there, write the changed
    // ones to the raster, to the cost windows and to Cb; the change mask stays in LDS.  In the order the patches were handed over (they may overlap).
    for (int r = 0; r < J.rb.nrect; ++r) {
        if (!J.psrc[r]) continue;
        const int *qr = J.rb.rect[r];                       // {map, x, y, w, h}
        const int px = qr[1], py = qr[2], pw = qr[3], ph = qr[4];
        const unsigned pw_magic = ((1u << 20) + (unsigned)pw - 1u) / (unsigned)pw;      // (patches are at most 64 x 64 cells: host side)
        __syncthreads();                                    // Cb staged / the previous rectangle's seeds read
        STAMP(2);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int e = tid + c * NTHR;
            if (e >= pw * ph) continue;
            const int i = (int)(((unsigned)e * pw_magic) >> 20), j = e - i * pw;      // (e / pw: exact for e < 4400, pw <= 65)
            uint8_t *cb = &Cb[(px + i - rx0 + COFF) * RCP + (py + j - ry0 + COFF)];
            const uint8_t nv = (r == r_first) ? (uint8_t)pb0[c] : J.psrc[r][e];
            const uint8_t ch = *cb != nv;
            s_pmask[e] = ch;
            if (ch) { *cb = nv; P.cost[(size_t)m * P.cstride + (size_t)(px + i) * P.W + (py + j)] = nv; cost_window_store(P, m, px + i, py + j, nv); }
        }
        __syncthreads();
        STAMP(3);
        const int ne = CELLS ? pw * ph : (pw + 1) * (ph + 1);
        if (one_fused) {
            // num_nodes_updated (FD impl:138, DFM impl:109): the elements a changed cell touches.  One patch, nothing else pending: no element can
            // have been counted already (the marks, which are for that, are all clear between steps and stay clear), and every seeded tile lies
            // inside the block (no seed list).  Counted per wave, one atomic each.
            const int ew = CELLS ? pw : pw + 1;
            const unsigned ew_magic = ((1u << 20) + (unsigned)ew - 1u) / (unsigned)ew;
            int cnt = 0;
            for (int e = tid; e < ne; e += NTHR) {
                const int i = (int)(((unsigned)e * ew_magic) >> 20), j = e - i * ew;
                bool ch;
                if (!CELLS) ch = (i > 0 && j > 0 && s_pmask[(i - 1) * pw + j - 1]) || (i > 0 && j < pw && s_pmask[(i - 1) * pw + j]) ||
                                 (i < ph && j > 0 && s_pmask[i * pw + j - 1]) || (i < ph && j < pw && s_pmask[i * pw + j]);
                else ch = s_pmask[i * pw + j];
                cnt += ch ? 1 : 0;
            }
            for (int o = 32; o; o >>= 1) cnt += __shfl_xor(cnt, o);
            if ((tid & 63) == 0 && cnt) atomicAdd(&P.num_updated[m], (unsigned int)cnt);
        } else {
            for (int base = 0; base < ne; base += NTHR) patch_seed<!CELLS>(P, m, s_pmask, px, py, pw, ph, base + tid);
        }
    }
    if (J.rb.nrect && J.psrc[J.rb.nrect - 1]) __syncthreads();      // (the seeds' list entries and counts are read below)
    STAMP(4);
    // ---- 0c. what k_replan_begin does: step bookkeeping, mark reset, the seeds, the invalidation bound ----
    if (tid == 0 && !J.batch) *P.dyn = J.dyn;      // (a batch: the host has put them in place before the launch -- other workgroups read them too)
    if (!J.batch) step_begin(P, J.rb.sb);
    if (!one_fused)
        for (int r = 0; r < J.rb.nrect; ++r) {
            const int *qr = J.rb.rect[r];
            for (int e = tid; e < (qr[3] + 1) * (qr[4] + 1); e += NTHR) clear_mark(P, qr[0], qr[1], qr[2], qr[3], qr[4], e);
        }
    for (int i = tid; i < (int)(sizeof(RegionShared) / sizeof(int)); i += NTHR) reinterpret_cast<int *>(&S)[i] = 0;
    __syncthreads();
    STAMP(5);
    for (int i = tid; i < 2 * RTMAX * RTMAX; i += NTHR) (&S.dprio[0][0])[i] = INFBITS;
    for (int i = tid; i < RFRAME; i += NTHR) { S.actL[i] = INFBITS; S.actR[i] = INFBITS; }
    if (tid == 0) { S.rmin = INFBITS; S.m_r = INFBITS; }
    {   // pending seeds of this map (all consumed): tiles inside the block are handled here, any other goes to the queue
        // (n_pending was read before the patches of this job were seeded: the general seeding above appends to the list)
        const int n = one_fused ? n_pending : P.ctr->scount;
        for (int i = tid; i < n; i += NTHR) {
            const int gt = P.slist[i];
            if (gt / P.NTm != m) continue;
            const int tx = (gt - gt0) / P.TY, ty = (gt - gt0) - tx * P.TY;
            P.sflag[gt] = 0;
            if (tx < J.tx0 || tx >= J.tx0 + J.ntx || ty < J.ty0 || ty >= J.ty0 + J.nty) activate(P, Q_RAISE, J.rb.k_raise, gt, 0);
        }
        __syncthreads();
        STAMP(6);
        if (tid == 0 && !J.batch) { P.ctr->scount = 0; P.ctr->done = 0; }   // (a batch: the last workgroup, when everybody has read the list)
        if (tid < 64) {
            // the start's key before the patch (start_bound(), from the values asked for at the top): one start element per lane 0..3 (one thread
            // after the other -- four divisions, four hypotf -- this was a microsecond in front of the phases), the largest key by two shuffles
            float key = 0.0f, dist = 0.0f;
            int off = -1;
            if (se_x >= 0) {
                dist = hm * hypotf(J.rb.sb.sx - (float)se_x, J.rb.sb.sy - (float)se_y);
                if (sg < INFINITY) key = sg + dist;
                if (se_x >= rx0 && se_x < rx0 + rnx && se_y >= ry0 && se_y < ry0 + rny) off = (se_x - rx0 + 1) * RP + (se_y - ry0 + 1);
            }
            if (tid < 4) { S.soff[tid] = off; S.sdist[tid] = dist; }
            const bool in = __builtin_amdgcn_ballot_w64(off >= 0) != 0ull;
            float b0 = fmaxf(key, __shfl_xor(key, 1));
            b0 = fmaxf(b0, __shfl_xor(b0, 2));
            if (tid == 0) {
                b0 = b0 > 0.0f ? b0 : INFINITY;
                S.B0 = b0;
                S.rbound = focused ? b0 + J.rb.band : INFINITY;
                S.any_start_in = in ? 1 : 0;
            }
        }
    }

    if (tid == 0) { S.tstamp[0] = t_begin; S.tstamp[1] = wall_clock64(); S.tstamp[2] = S.tstamp[1]; }
    // the seeds: every patch that holds an element of a consumed rectangle (a superset of the changed cells' elements;
    // a sweep that finds nothing to do costs a fraction of a microsecond)
    __syncthreads();
    for (int r = 0; r < J.rb.nrect; ++r) {
        const int *qr = J.rb.rect[r];                       // {map, x, y, w, h}: cells x..x+h-1, y..y+w-1
        const int ex0 = qr[1] - rx0, ex1 = qr[1] + qr[4] - (CELLS ? 1 : 0) - rx0, ey0 = qr[2] - ry0, ey1 = qr[2] + qr[3] - (CELLS ? 1 : 0) - ry0;
        const int p0 = max(ex0, 0) / 4, p1 = min(ex1, rnx - 1) / 4, c0 = max(ey0, 0) / 4, c1 = min(ey1, rny - 1) / 4;
        const int np = (p1 - p0 + 1) * (c1 - c0 + 1);
        for (int i = tid; i < np; i += NTHR) {
            const int pr = p0 + i / (c1 - c0 + 1), pc = c0 + i % (c1 - c0 + 1);
            const int ni = (pr >> 2) * RPW + (pc >> 2);
            atomicOr(&S.wake[((pr & 3) << 2) | (pc & 3)][ni >> 5], 1 << (ni & 31));
            atomicOr(&S.seed[((pr & 3) << 2) | (pc & 3)][ni >> 5], 1 << (ni & 31));
            S.traised[(pr / TP) * J.nty + pc / TP] = 1;      // a seeded tile is invalidated without a bound (the launch chain queues seeds with priority 0)
        }
    }
    if (tid < 4) S.swas[tid] = (S.soff[tid] >= 0 && Gs[S.soff[tid]] < INFINITY) ? 1 : 0;
    __syncthreads();
    const int goal_lx = goal_x - rx0, goal_ly = goal_y - ry0;

    // ---- 2. invalidate, lower; again while an invalidation that was held back lies below the start's new key ----
    for (int round = 0;; ++round) {
        for (;;) {
            region_phase<ALGO, MODE_RAISE>(P, J, Gs, Cb, Bb, S, thr, hm, focused, goal_lx, goal_ly);
            // a tile that lost a value after some of its patches had been held back: those patches again
            const int again_t = __syncthreads_or(tid < ntl && S.traised[tid] && S.dprio[1][tid] != INFBITS) && !S.giveup;
            __syncthreads();
            if (!again_t) break;
            for (int i = tid; i < 16 * RWW; i += NTHR) { (&S.wake[0][0])[i] |= (&S.defer[1][0][0])[i]; (&S.defer[1][0][0])[i] = 0; }
            for (int i = tid; i < RTMAX * RTMAX; i += NTHR) S.dprio[1][i] = INFBITS;
            __syncthreads();
        }
        if (tid == 0 && round == 0) S.tstamp[3] = wall_clock64();
        // everything the invalidation swept is re-lowered (plus, in later rounds, what lowering had to hold back)
        for (int i = tid; i < 16 * RWW; i += NTHR) {
            (&S.wake[0][0])[i] |= (&S.ever[0][0])[i] | (&S.defer[0][0][0])[i];
            (&S.ever[0][0])[i] = 0; (&S.defer[0][0][0])[i] = 0;
        }
        for (int i = tid; i < RTMAX * RTMAX; i += NTHR) S.dprio[0][i] = INFBITS;
        if (tid == 0) {
            S.dkey = INFBITS;
            S.Bsub = focused ? region_lower_bound(Gs, S) : INFINITY;
            // the first band: everything below the smallest value that was taken away, plus one band width (a step without
            // invalidations is not ordered: its few lowered values sit around the patch)
            S.theta = (round == 0) ? ((S.rmin != INFBITS) ? __int_as_float(S.rmin) + J.delta : INFINITY) : S.theta;
            S.Bgate = fminf(S.Bsub + J.slack, S.theta);
        }
        __syncthreads();
        for (int sub = 0;; ++sub) {
            region_phase<ALGO, MODE_LOWER>(P, J, Gs, Cb, Bb, S, thr, hm, focused, goal_lx, goal_ly);
            // the start's key may have risen while results were being held back against an earlier value of it
            if (tid == 0) {
                const float B = focused ? region_lower_bound(Gs, S) : INFINITY;
                const int m = S.dkey;
                bool again = !S.giveup && m != INFBITS && (__int_as_float(m) < B || B == INFINITY);
                if (again && sub >= 4096) { again = false; S.giveup = 1; }   // (never seen; what is dirty goes to the queues)
                S.again = again ? 1 : 0;
                S.Bsub = B;
                if (again) S.theta = fmaxf(S.theta, __int_as_float(m)) + J.delta;    // the next band starts where work is waiting
                S.Bgate = fminf(S.Bsub + J.slack, S.theta);
            }
            __syncthreads();
            const int again_l = S.again;
            __syncthreads();                    // (thread 0 overwrites the word below: everyone has read it first)
            if (!again_l) break;
            if (tid == 0) ++S.dbg[2];
            for (int i = tid; i < 16 * RWW; i += NTHR) { (&S.wake[0][0])[i] |= (&S.defer[0][0][0])[i]; (&S.defer[0][0][0])[i] = 0; }
            for (int i = tid; i < RTMAX * RTMAX; i += NTHR) S.dprio[0][i] = INFBITS;
            if (tid == 0) S.dkey = INFBITS;
            __syncthreads();
        }
        if (tid < ntl && S.dprio[1][tid] != INFBITS) atomicMin(&S.m_r, S.dprio[1][tid]);     // (S.m_r: INFBITS here, reset below)
        __syncthreads();
        if (tid == 0) {
            const float B = focused ? region_lower_bound(Gs, S) : INFINITY;
            const int m = S.m_r;
            S.m_r = INFBITS;
            bool again = !S.giveup && focused && m != INFBITS && __int_as_float(m) < B + J.slack;   // lowering reaches the key plus one move
            if (again && round >= 14) { again = false; S.giveup = 1; }
            if (again) S.rbound = fmaxf(B, S.rbound) + J.rb.band;
            S.again = again ? 1 : 0;
        }
        __syncthreads();
        const int again_r = S.again;
        __syncthreads();
        if (!again_r) break;
        if (tid == 0) ++S.dbg[3];
        for (int i = tid; i < 16 * RWW; i += NTHR) { (&S.wake[0][0])[i] |= (&S.defer[1][0][0])[i]; (&S.defer[1][0][0])[i] = 0; }
        for (int i = tid; i < RTMAX * RTMAX; i += NTHR) S.dprio[1][i] = INFBITS;
        __syncthreads();
    }

    // ---- 2b. the back-pointers (what k_finalize_bp does for the launch chain's steps): every node of a patch in which a value changed, and of
    // the patches around it -- a neighbour's value may be what it was and still come from another triangle now --, evaluated once more on the
    // values as they stand, the arg-min kept.  (MS-DFM level 0 has none.)
    {
        const int w = wave_index16(tid >> 6), lane = tid & 63, q = lane & 3, nd = lane >> 2;
        const int nprow = J.ntx * TP, npcol = J.nty * TP;
        auto cost_at = [=](int r, int c) { const int b = Cb[r * RCP + c]; return b >= thr ? INFINITY : (float)b; };
        for (int wd = 0; wd < RWW; ++wd) {
            int bits = __builtin_amdgcn_readfirstlane(S.renew[w][wd]);
            while (bits) {
                const int j = __ffs(bits) - 1;
                bits &= bits - 1;
                const int idx = wd * 32 + j;
                const int pr = (idx / RPW) * 4 + (w >> 2), pc = (idx % RPW) * 4 + (w & 3);
                if (pr >= nprow || pc >= npcol) continue;
                const int lx = pr * 4 + (nd >> 2), ly = pc * 4 + (nd & 3);
                QuadConsts<ALGO> C;
                C.load_at(cost_at, lx, ly, q, RP);
                const LaneEval le = eval_quad_w<ALGO, RP>(Gs + (lx + 1) * RP + ly + 1, q, C);
                const float nv = quad_min(le.r);
                const int bq = quad_min_int(bp_byte<ALGO>(le, q, C, le.r == nv)), b = bq == 0x3FF ? BP_NONE : (bq & 0x1F);
                if (q == 0) Bb[lx * RBP + ly] = (uint8_t)((lx == goal_lx && ly == goal_ly) ? BP_NONE : b);
                if (lane == 0) S.tbp[(pr / TP) * J.nty + pc / TP] = 1;
            }
        }
        __syncthreads();
    }
    if (tid == 0) S.tstamp[4] = wall_clock64();
    // ---- 3. write back: changed values, the rings of the neighbours they border, what the frame has to hear ----
    // frame tile index of the tile at block coordinates (ti, tj), ti in -1..ntx, tj in -1..nty (one of them outside)
    auto frame_index = [&](int ti, int tj) {
        if (ti < 0) return tj + 1;                                   // top row: 0 .. nty+1
        if (ti >= J.ntx) return (J.nty + 2) + tj + 1;                // bottom row
        if (tj < 0) return 2 * (J.nty + 2) + ti;                     // left column
        return 2 * (J.nty + 2) + J.ntx + ti;                         // right column
    };
    static_assert(NTHR == 1024 && TT % 64 == 0 && RBP % 4 == 0, "the block kernel deals its staging and write-back out by wave");
    // (Round 4: this used to be two loops of 16 iterations per thread over ALL tiles of the largest block, unrolled so that the values HBM holds could wait in
    //  registers between them -- 12 000 instructions of which a replan executes a sixth, fetched cold at every launch.  Now: the tiles that have something to
    //  write are listed, what HBM holds of them goes straight into LDS (no register held, every load in flight at once), and one loop walks the list.)
    constexpr int UPT = TT / 64;                                        // units (64 consecutive elements = one wave) per tile
    const int wvw = __builtin_amdgcn_readfirstlane(tid >> 6), lnw = tid & 63;
    if (tid < 64) {
        const bool has = tid < ntl && (S.tflag[tid] | S.tbp[tid]);
        const unsigned long long hm_ = __builtin_amdgcn_ballot_w64(has);
        if (has) s_wbl[__popcll(hm_ & ((1ull << tid) - 1ull))] = tid;
        if (tid == 0) s_nwb = __popcll(hm_);
    }
    __syncthreads();
    const int nunits = s_nwb * UPT;
    auto unit_tile = [&](int u, int &k, int &part, int &tl, int &ti, int &tj) {      // (all wave-uniform)
        k = u / UPT; part = u - k * UPT; tl = __builtin_amdgcn_readfirstlane(s_wbl[k]);
        ti = (tl * nty_magic) >> 16; tj = tl - ti * J.nty;
    };
    {
        typedef __attribute__((address_space(3))) void *lds_ptr;
        typedef const __attribute__((address_space(1))) void *glb_ptr;
#pragma unroll 1
        for (int u = wvw; u < nunits; u += 16) {
            int k, part, tl, ti, tj;
            unit_tile(u, k, part, tl, ti, tj);
            if (!S.tflag[tl] || k >= NG0) continue;
            const size_t gt = (size_t)(gt0 + (J.tx0 + ti) * P.TY + J.ty0 + tj);
            __builtin_amdgcn_global_load_lds((glb_ptr)(P.G + gt * TT + part * 64 + lnw), (lds_ptr)(G0c + k * TT + part * 64), 4, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    int n_exp = 0;
#pragma unroll 1
    for (int u = wvw; u < nunits; u += 16) {
        int k, part, tl, ti, tj;
        unit_tile(u, k, part, tl, ti, tj);
        const int e = part * 64 + lnw;
        // the renewed back-pointers of the tile (a patch next to a changed one may lie in an unchanged tile)
        if (S.tbp[tl]) P.bp[(size_t)(gt0 + (J.tx0 + ti) * P.TY + J.ty0 + tj) * TT + e] = Bb[(ti * T + e / T) * RBP + tj * T + e % T];
        if (!S.tflag[tl]) continue;                                         // nothing was applied in this tile
        const int tx = J.tx0 + ti, ty = J.ty0 + tj, gt = gt0 + tx * P.TY + ty;
        const int io_r = e / T, io_c = e % T;
        float gl0;
        if (k < NG0) gl0 = G0c[k * TT + e];
        else { gl0 = P.G[(size_t)gt * TT + e]; P.Gprev[(size_t)gt * TT + e] = gl0; }      // (more changed tiles than the LDS copy holds: their old values go to the snapshot at once)
        const float gf = Gs[(ti * T + io_r + 1) * RP + tj * T + io_c + 1];
        if (gf == gl0) continue;
        ++n_exp;
        P.G[(size_t)gt * TT + e] = gf;
        const int er = (io_r == 0) ? -1 : ((io_r == T - 1) ? 1 : 0);
        const int ec = (io_c == 0) ? -1 : ((io_c == T - 1) ? 1 : 0);
        const bool rok = er && tx + er >= 0 && tx + er < P.TX, cok = ec && ty + ec >= 0 && ty + ec < P.TY;
        if (rok) P.ring[(size_t)(gt + er * P.TY) * RING + (er < 0 ? RING_BOT : RING_TOP) + io_c] = gf;
        if (cok) P.ring[(size_t)(gt + ec) * RING + (ec < 0 ? RING_RIGHT : RING_LEFT) + io_r] = gf;
        if (rok && cok) P.ring[(size_t)(gt + er * P.TY + ec) * RING + RING_CORNER + (er < 0 ? 2 : 0) + (ec < 0 ? 1 : 0)] = gf;
        // neighbours outside the block: lowered value -> lowering queue, lost / raised value -> invalidation queue,
        // each only if an element of that neighbour (the frame, as staged) can be concerned at all (causality, see k_relax)
        const bool out_r = rok && (ti + er < 0 || ti + er >= J.ntx), out_c = cok && (tj + ec < 0 || tj + ec >= J.nty);
        if (!(out_r || out_c)) continue;
        const int lxr = ti * T + io_r + 1, lyc = tj * T + io_c + 1;          // position in Gs
        const int cl = max(io_c - 1, 0) - io_c, ch = min(io_c + 1, T - 1) - io_c;   // columns / rows of the edge neighbour itself
        const int rl = max(io_r - 1, 0) - io_r, rh = min(io_r + 1, T - 1) - io_r;
        const float lo = fminf(gf, gl0);
        const bool lowered = gf < gl0;
        const int pb = __float_as_int(lowered ? gf : gl0);
        auto concerned = [&](float a, float b, float c) {
            if (lowered) return lo < fmaxf(fmaxf(a, b), c);                  // someone above the new value
            const float fa = a < INFINITY ? a : -INFINITY, fb = b < INFINITY ? b : -INFINITY, fc = c < INFINITY ? c : -INFINITY;
            return gl0 < fmaxf(fmaxf(fa, fb), fc);                          // someone (finite) above the old value may have leaned on it
        };
        if (out_r) {
            const float *h = Gs + (lxr + er) * RP + lyc;
            if (concerned(h[cl], h[0], h[ch])) atomicMin(lowered ? &S.actL[frame_index(ti + er, tj)] : &S.actR[frame_index(ti + er, tj)], pb);
        }
        if (out_c) {
            const float *h = Gs + lxr * RP + lyc + ec;
            if (concerned(h[rl * RP], h[0], h[rh * RP])) atomicMin(lowered ? &S.actL[frame_index(ti, tj + ec)] : &S.actR[frame_index(ti, tj + ec)], pb);
        }
        if (rok && cok && (out_r || out_c)) {                               // the diagonal neighbour
            const int di = ti + er, dj = tj + ec;
            if (di < 0 || di >= J.ntx || dj < 0 || dj >= J.nty) {
                const float hv = Gs[(lxr + er) * RP + lyc + ec];
                if (concerned(hv, hv, hv)) atomicMin(lowered ? &S.actL[frame_index(di, dj)] : &S.actR[frame_index(di, dj)], pb);
            }
        }
    }
    if (n_exp) atomicAdd(&S.expanded, n_exp);
    __syncthreads();
    // frame tiles -> queues; held-back results -> park lists; what a budget overrun left dirty -> queues
    for (int i = tid; i < 2 * (J.nty + 2) + 2 * J.ntx; i += NTHR) {
        int ti, tj;
        if (i < J.nty + 2) { ti = -1; tj = i - 1; }
        else if (i < 2 * (J.nty + 2)) { ti = J.ntx; tj = i - (J.nty + 2) - 1; }
        else if (i < 2 * (J.nty + 2) + J.ntx) { ti = i - 2 * (J.nty + 2); tj = -1; }
        else { ti = i - 2 * (J.nty + 2) - J.ntx; tj = J.nty; }
        const int tx = J.tx0 + ti, ty = J.ty0 + tj;
        if (tx < 0 || ty < 0 || tx >= P.TX || ty >= P.TY) continue;
        const int gt = gt0 + tx * P.TY + ty;
        if (S.actR[i] != INFBITS) activate(P, Q_RAISE, J.rb.k_raise, gt, S.actR[i]);
        if (S.actL[i] != INFBITS) activate(P, Q_LOWER, J.k_lower, gt, S.actL[i]);
    }
    for (int tl = tid; tl < ntl; tl += NTHR) {
        const int gt = gt0 + (J.tx0 + tl / J.nty) * P.TY + J.ty0 + tl % J.nty;
        if (S.dprio[0][tl] != INFBITS) park_tile(P, Q_LOWER, gt, S.dprio[0][tl]);
        if (S.dprio[1][tl] != INFBITS) park_tile(P, Q_RAISE, gt, S.dprio[1][tl]);
    }
    if (S.giveup) {
        for (int i = tid; i < 16 * RWW * 32; i += NTHR) {
            const int w = i / (RWW * 32), r = i % (RWW * 32);
            if (!(S.wake[w][r >> 5] & (1 << (r & 31)))) continue;
            const int pr = (r / RPW) * 4 + (w >> 2), pc = (r % RPW) * 4 + (w & 3);
            if (pr >= J.ntx * TP || pc >= J.nty * TP) continue;
            const int gt = gt0 + (J.tx0 + pr / TP) * P.TY + J.ty0 + pc / TP;
            activate(P, Q_RAISE, J.rb.k_raise, gt, 0);      // invalidation first, then (k_touched_to_active) lowering
            activate(P, Q_LOWER, J.k_lower, gt, 0);
        }
    }
    // (one workgroup of a batch reads what another one queued: agent scope; a single map's end check only reads what this very
    //  workgroup wrote -- its loads are agent-scope loads from the XCD's L2, where its stores are once they have been acknowledged)
    if (J.batch) __threadfence(); else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();

    if (tid == 0) S.tstamp[5] = wall_clock64();
    // ---- 4. the device-side end condition (as k_replan_end): is anything left below the start's key? ----
    if (tid == 0) { S.m_r = INFBITS; S.m_l = INFBITS; }
    __syncthreads();
    {
        const float B = focused ? region_start_key(Gs, S) : INFINITY;
        int mr = INFBITS, ml = INFBITS;
        // (the four list lengths together, before the first list is walked)
        const int nr = __hip_atomic_load(&P.ctr->cnt[Q_RAISE][J.rb.k_raise % 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int npr = __hip_atomic_load(&P.ctr->npark[Q_RAISE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int nl = __hip_atomic_load(&P.ctr->cnt[Q_LOWER][J.k_lower % 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int npl = __hip_atomic_load(&P.ctr->npark[Q_LOWER], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (A batch: the other maps' workgroups append to these lists while this one reads them -- a list's length is advanced BEFORE the entry is
        //  stored (activate(), park_tile()), so a slot below the length may still hold whatever the memory held: an entry is only followed if it
        //  is a tile of THIS map.  Round 4: the lowering loops used to load the priority of any entry first -- with recycled device memory behind
        //  the lists that was an intermittent "Memory access fault" of test_batch_with_heuristic_keys..., never seen with fresh (zeroed) memory.
        //  This map's own entries are complete: its workgroup appended them before the fence above.)
        auto mine = [&](int gt) { return (unsigned)gt < (unsigned)P.NT && gt / P.NTm == m; };
        for (int i = tid; i < nr; i += NTHR) {
            const int gt = __hip_atomic_load(&P.cand[(size_t)(Q_RAISE * 3 + J.rb.k_raise % 3) * P.NT + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (mine(gt)) mr = min(mr, prio_read_fresh(P, Q_RAISE, J.rb.k_raise, gt));
        }
        for (int i = tid; i < npr; i += NTHR) {
            const int gt = __hip_atomic_load(&P.park[(size_t)(Q_RAISE * 2) * P.NT + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (mine(gt)) mr = min(mr, __hip_atomic_load(&P.pprio[Q_RAISE * P.NT + gt], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
        auto lower_key = [&](int gt, int pbits) {      // smallest key an element of the tile can have: priority + hm * dist(start, tile)
            if (pbits == INFBITS) return INFINITY;
            return __int_as_float(pbits) + (focused ? tile_heuristic(P, m, (gt - gt0) / P.TY, (gt - gt0) % P.TY) : 0.0f);
        };
        float kl = INFINITY;
        for (int i = tid; i < nl; i += NTHR) {
            const int gt = __hip_atomic_load(&P.cand[(size_t)(Q_LOWER * 3 + J.k_lower % 3) * P.NT + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (mine(gt)) kl = fminf(kl, lower_key(gt, prio_read_fresh(P, Q_LOWER, J.k_lower, gt)));
        }
        for (int i = tid; i < npl; i += NTHR) {
            const int gt = __hip_atomic_load(&P.park[(size_t)(Q_LOWER * 2) * P.NT + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (mine(gt)) kl = fminf(kl, lower_key(gt, __hip_atomic_load(&P.pprio[Q_LOWER * P.NT + gt], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
        }
        ml = kl < INFINITY ? __float_as_int(kl) : INFBITS;
        if (mr != INFBITS) atomicMin(&S.m_r, mr);
        if (ml != INFBITS) atomicMin(&S.m_l, ml);
        __syncthreads();
        if (tid == 0) {
            bool done;
            if (focused && B < INFINITY) done = !(__int_as_float(S.m_r) < B) && !(__int_as_float(S.m_l) < B);
            else done = (S.m_r == INFBITS && S.m_l == INFBITS) && (J.batch || (nr == 0 && nl == 0));   // (a batch: the other maps'
            //                                           workgroups add to the same lists meanwhile; this map's entries are what counts)
            if (S.giveup) done = false;
            S.done = done ? 1 : 0;
            unsigned int upd = 0;
            if (P.consume[m]) { upd = P.num_updated[m]; P.num_updated[m] = 0; }
            if (!J.batch) {
                P.ctr->done = S.done;
                P.ctr->rbound = S.rbound;
                P.ctr->qmin[Q_RAISE] = S.m_r;
                P.ctr->updated = upd;
                P.ctr->tile_visits += (unsigned long long)ntl;
                P.ctr->raise_visits += (unsigned long long)ntl / 2;
                P.ctr->tile_iters += S.sweeps / 16;
                P.ctr->elem_evals += 16ull * S.sweeps;
                if (done) P.ctr->expanded = (unsigned long long)S.expanded;
            } else {           // shared counters (the host zeroed them for the step); the last workgroup draws the conclusion
                if (!done) atomicAdd(&P.ctr->done_fail, 1);
                // (does any map of the round have an invalidation left below ITS bound?  0 = yes: the host compares with the round's bound)
                if (S.m_r != INFBITS && __int_as_float(S.m_r) < S.rbound) atomicMin(&P.ctr->qmin[Q_RAISE], 0);
                atomicMax(reinterpret_cast<int *>(&P.ctr->rbound), __float_as_int(S.rbound));   // (positive floats order like their bits)
                atomicAdd(&P.ctr->updated, upd);
                atomicAdd(&P.ctr->tile_visits, (unsigned long long)ntl);
                atomicAdd(&P.ctr->raise_visits, (unsigned long long)ntl / 2);
                atomicAdd(&P.ctr->tile_iters, S.sweeps / 16);
                atomicAdd(&P.ctr->elem_evals, 16ull * S.sweeps);
                if (done) atomicAdd(&P.ctr->expanded, (unsigned long long)S.expanded);
            }
            if (J.debug & 2) {   // diagnostics: the end check's inputs, readable through ufm_debug_lmax
                int nd0 = 0, nd1 = 0, m0 = INFBITS, m1 = INFBITS;
                for (int i = 0; i < ntl; ++i) { if (S.dprio[0][i] != INFBITS) { ++nd0; m0 = min(m0, S.dprio[0][i]); } if (S.dprio[1][i] != INFBITS) { ++nd1; m1 = min(m1, S.dprio[1][i]); } }
                int *d = P.lmax;
                d[0] = __float_as_int(B); d[1] = __float_as_int(S.B0); d[2] = __float_as_int(S.rbound); d[3] = S.m_r; d[4] = S.m_l;
                d[5] = nr; d[6] = nl; d[7] = npr; d[8] = npl; d[9] = S.done; d[10] = S.giveup; d[11] = (int)S.sweeps; d[12] = S.expanded;
                d[13] = nd0; d[14] = m0; d[15] = nd1; d[16] = m1; d[17] = S.any_start_in;
                for (int i = 0; i < 4; ++i) d[18 + i] = S.soff[i] >= 0 ? __float_as_int(Gs[S.soff[i]]) : -1;
                for (int i = 0; i < 8; ++i) d[22 + i] = S.dbg[i];
                S.tstamp[6] = wall_clock64();
                for (int i = 1; i < 7; ++i) d[30 + i] = (int)(S.tstamp[i] - S.tstamp[0]);
                for (int i = 1; i < 8; ++i) d[40 + i] = (int)(s_tb[i] - s_tb[0]);
                for (int i = 0; i < 16; ++i) d[50 + i] = s_simd[i];
                for (int i = 0; i < 8; ++i) d[70 + i] = S.dbg2[i];
            }
        }
        __syncthreads();
    }
    // not done: the launch chain takes over; the tiles changed here join the step's touched list with their values
    // as of the start of the step (num_nodes_expanded is counted against that snapshot at the end of the step)
    if (!S.done) {
#pragma unroll 1
        for (int u = wvw; u < nunits; u += 16) {
            int k, part, tl, ti, tj;
            unit_tile(u, k, part, tl, ti, tj);
            if (!S.tflag[tl]) continue;
            const int e = part * 64 + lnw;
            const int gt = gt0 + (J.tx0 + ti) * P.TY + J.ty0 + tj;
            if (k < NG0) P.Gprev[(size_t)gt * TT + e] = G0c[k * TT + e];
            if (e == 0) {
                P.fresh[gt] = 0;
                if (atomicAdd(&P.touched[gt], 1) == 0) P.tlist[atomicAdd(&P.ctr->tcount, 1)] = gt;
            }
        }
    }
    if (J.batch) __threadfence();          // (a single map: the kernel's end, or thread 0's system-scope release below, publishes)
    __syncthreads();
    if (J.batch) {             // the last workgroup to get here publishes for all of them
        if (tid == 0) s_last = (atomicAdd(&P.ctr->fin_blocks, 1) == (int)gridDim.x - 1);
        __syncthreads();
        if (!s_last) return;
        __threadfence();
        if (tid == 0) {
            P.ctr->fin_blocks = 0;
            P.ctr->scount = 0;
            P.ctr->done = __hip_atomic_load(&P.ctr->done_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 ? 1 : 0;
        }
        __threadfence();
        __syncthreads();
    }
    const int *src = reinterpret_cast<const int *>(P.ctr);
    int *dst = reinterpret_cast<int *>(host);
    for (int i = tid; i < (int)(sizeof(DevCounters) / sizeof(int)); i += NTHR)
        dst[i] = __hip_atomic_load(&src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (no fence of every wave's own here: the barrier orders the copies before thread 0's store, whose system-scope release writes
    //  the L2 back once)
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flag, J.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
