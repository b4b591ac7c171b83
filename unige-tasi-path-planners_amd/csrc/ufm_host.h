// ufm_host.h -- the host side: Engine (allocation, launch chain, resident phase, replan submission, ReplannerBase::step)
// (a piece of ufm_engine.hip, the engine's one translation unit: included there, inside its anonymous namespace)
#pragma once

// ---- host side -------------------------------------------------------------------
#define HIPCHK(expr)                                                      \
    do {                                                                  \
        hipError_t _e = (expr);                                           \
        if (_e != hipSuccess) return UFM_ERR_HIP_BASE - (int)_e;          \
    } while (0)

struct PatchRect { int m, x, y, w, h; };

#include "ufm_path.h"

struct MapState {
    bool initialize_search = true;   // ReplannerBase.h:149
    bool goal_set = false;           // :150
    bool new_goal = false;           // :151
    bool new_start = false;          // :152
    bool have_map = false;           // !initialize_graph :148
    bool start_set = false;
    float start_x = 0, start_y = 0, goal_x = 0, goal_y = 0;
    int goal_ex = 0, goal_ey = 0;    // Node()/Cell() of the goal
    bool goal_elem_valid = false;
};

struct Engine {
    int algo = 0, opt_lvl = 0, heur = 0, device = 0, nmaps = 1;
    float heuristic_multiplier = 1.0f;
    int thr_uchar = 254;             // Graph.h:34
    DevDyn dyn_dev{-1.0f, -1, -1, 0};  // what *P.dyn holds (as far as the host knows)
    uint32_t graphs_made = 0;        // replan graphs instantiated so far (ufm_stats::graphs_instantiated)
    int W = 0, L = 0;
    DevParams P{};
    bool allocated = false;
    hipStream_t stream = nullptr;
    DevCounters *h_ctr = nullptr;    // pinned, host-coherent: k_publish writes it, the host spins on h_flag
    unsigned int *h_flag = nullptr;  // sequence number of the last published copy (same allocation)
    DevCounters *h_pipe_ctr[2] = {nullptr, nullptr};   // run_phase keeps one batch of launches in flight ahead of the one whose
    unsigned int *h_pipe_flag[2] = {nullptr, nullptr}; // counters it is looking at: two more published copies, used alternately
    bool pipeline_batches = true;
    unsigned int pub_seq = 0;
    bool spin_wait = true;           // false: hipMemcpyAsync + hipStreamSynchronize instead
    bool fuse_control = true;        // replans: fused control kernels (k_replan_begin / _raise_to_lower / _end)
    bool use_graph = true;           // replans: the whole submission replayed as one captured hipGraph
    bool use_owned = true;           // plans: the lowering phase as ONE resident launch (k_relax<.,LOWER,false,1|2>) instead of a launch per band step
    float owned_limit_ms = -1.0f;    // ... which hands back to the launch chain after this long, whatever happens (< 0: by the size of the job,
                                     //     ~15 x what a plan of that many tiles takes: a device shared with another long-running kernel)
    float owned_band = -1.0f;        // ... ordering band in tile crossings (< 0: by the job -- 2.5 for a single map of a node planner, 3 for MS-DFM, 2 for a batch: owned_phase())
    uint32_t owned_launches = 0;
    hipEvent_t own_ev[2] = {nullptr, nullptr};
    hipEvent_t reg_ev[2] = {nullptr, nullptr};     // profiling: around the block kernel of a replan
    bool own_timed = false;
    int owned_flags = 0;             // resident kernel, diagnostics and variants.  1: no tile taken ahead; 2: no early hand-off (FD / SG); 4: early hand-off once per patch and visit;
                                     // 8: early hand-off waits for its stores inside the sweep loop; 16: no in-visit halo refresh; 32: idle workgroups do not help out;
                                     // 64: a workgroup does not follow the front (the neighbour it has just queued); 128: ... follows it beyond the ordering band too
    int owned_waves = 0;             // waves per tile visit of the resident kernel: 16 (256 workgroups), 8 (512), 0 = by the size of the job
    int dag_mode = 0;                // round 4 experiment: first visits gated by an arrival estimate.  0 off; 1: the estimate handed in through ufm_debug_set_tile_order
    bool dag_have = false;           // ... an estimate is in P.dag_a
    float dag_kappa = 0.5f;          // ... a neighbour counts as clearly earlier below own estimate - kappa x (own - earliest neighbour's)
    int dag_patience = 16;           // ... looks without an eligible tile before a workgroup takes a held one anyway
    bool use_region = true;          // replans: one workgroup runs both phases in LDS on the block around the patch (ufm_region.h);
                                     // the launch chain only takes over when work is left outside the block
    int region_ahead = 2;            // block placement: tiles kept between the patches' centre and the block's goal-side edge
    int region_tiles = 6;            // block edge in tiles (<= RTMAX; measured on the headline replans: 10 -> 6 tiles: 20.0 -> 18.6 ms per 100, same completion rate)
    int region_sweeps = 4096;        // sweep budget per wave and phase
    int region_debug = 0;
    float region_band = 1.5f;        // ordering band of the block's lowering sub-rounds, in patch crossings at the mean cost (0: unordered)
    uint32_t region_runs = 0, region_done = 0;   // replans submitted to the block kernel / completed by it alone
    uint32_t region_cont = 0, region_cont_done = 0;   // ... continued by one blind submission of the launch chain / completed by that
    int cont_raise = 4, cont_lower = 8;          // launches of that submission per phase (0: straight to the adaptive loop -- MS-DFM, whose leftovers are
                                                 // chains of 20-40 band steps, mostly invalidation: measured, config 4, 903 us per such round either way;
                                                 // sending the leftover lowering through the resident kernel instead: 1 111 us)
    int batch_margin = 1;            // replans: launches per phase = most that the last 6 replans needed + this
    float raise_margin = 0.25f;      // invalidation bound = start key + this many ordering bands (a miss costs a second round)
    ReplanJob *h_job = nullptr;      // host-coherent pinned: per-replan inputs of the graph's first node
    struct GraphSig { DevParams P; float band, delta; int max_iters, grid; };
    GraphSig graph_sig{};
    std::vector<std::pair<int, hipGraphExec_t>> graphs;   // key nr * 256 + nl
    int relax_kernel(int mode, int k_arg, float rbound, int grid);
    void finalize_bp(int only_if_done) {   // the back-pointers of the tiles the step touched (k_finalize_bp); MS-DFM level 0 has none (its map has no Info)
        if (algo == UFM_ALGO_FD) k_finalize_bp<UFM_ALGO_FD><<<2048, 256, 0, stream>>>(P, only_if_done);
        else if (algo == UFM_ALGO_SG) k_finalize_bp<UFM_ALGO_SG><<<2048, 256, 0, stream>>>(P, only_if_done);
        else if (opt_lvl >= 1) k_finalize_bp<ALGO_DFM1><<<2048, 256, 0, stream>>>(P, only_if_done);
    }
    int tail_grid = 96;              // replan graph: workgroups of the later launches of a phase (few tiles left)
    int replan_graph(int nr, int nl, float band, hipGraphExec_t *out);
    void drop_graphs() { for (auto &g : graphs) hipGraphExecDestroy(g.second); graphs.clear(); }
    int *h_scratch = nullptr;        // pinned, nmaps*4 ints
    int *d_scratch = nullptr;
    uint8_t *d_patch = nullptr;      // staging for host patches
    size_t d_patch_cap = 0;
    uint8_t *h_patch = nullptr;      // pinned staging
    float *d_field = nullptr;        // ufm_read_field: the requested window, dense
    size_t d_field_cap = 0;
    int32_t *d_info = nullptr;       // ufm_read_info: back-pointers of the requested window
    size_t d_info_cap = 0;           // (int32 entries)
    uint8_t *d_pmask = nullptr;      // changed-cell mask of the patch being applied
    size_t d_pmask_cap = 0;
    PathJob *d_jobs = nullptr, *h_jobs = nullptr;     // path extraction: per-map start / goal (h_: pinned)
    float *d_path = nullptr, *h_path = nullptr;       // per-map output records
    size_t path_cap = 0;                              // floats per buffer
    std::vector<MapState> maps;
    std::vector<PatchRect> pending;
    std::vector<PatchRect> region_rects;   // the rectangles the current step consumes (jobs of the block kernel)
    int iter[2] = {0, 0};            // index k of the next relax launch of each queue (never reset: the queues persist)
    bool focused = true;             // stop at the start's key like the reference (end_condition)
    bool start_cell_floor = false;   // start_cell_ = the cell that contains the start position (floor) instead of Cell(Position)'s roundf (Cell.cpp:20-21): what the
                                     // revision of the reference that wrote its two recorded mission logs did (tests/test_reference_mission.py; oracle: ORC_REV_LOG)
    bool dynamic_mode = true;        // long queues: k_triage + cursor hand-out
    float *h_bnd = nullptr;          // pinned [nmaps]
    int last_active = 1;             // queue length at the last host check: long queues go through k_triage
    int grid_relax = 512;
    int dyn_grid = 256;              // workgroups of a cursor hand-out launch: the number of CUs
    int small_grid = 1 << 30;        // workgroups of a relax launch over a short queue (measured: no gain, off)
    int max_iters = 32;              // sweep cap per tile visit (x4 patch sweeps per wave): a tile that needs more is
                                     // re-queued instead of holding the whole launch (measured optimum on 4096^2)
    float delta_abs = -1.0f;         // ordering band; < 0: delta_scale * T * mean traversable cost
    float delta_scale = 1.5f;
    float delta_scale_long = 2.0f;   // ... for long queues (the plans' cursor hand-out launches): a wider band, fewer band steps
                                     // (tools/sweep.py on the final scheduler: plan 24.2 ms at 1.5, 23.6 at 2.0, 23.9 at 2.5;
                                     //  the replans' short launches are best at 1.5)
    float mean_cost = 1.0f;
    int batch_fixed = 0;
    bool profiling = false;
    std::vector<hipEvent_t> ev;
    ufm_stats last{};

    int alloc(int width, int length);
    void release();
    int launch_relax(int mode, float rbound, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
    int fetch_counters();
    int wait_published();
    int wait_flag(const unsigned int *flag, unsigned int seq);
    int win_raise[6] = {8, 8, 8, 8, 8, 8}, win_lower[6] = {8, 8, 8, 8, 8, 8}, win_pos = 0;   // launches recent replans needed
    int run_phase(int mode, float rbound, uint32_t *launches, float *kernel_ms, uint32_t *timed);
    int owned_phase();
    void own_layout(int ys) {            // the owner pattern: 16 x (1 << ys) tiles per block, one word per block and map for each owner
        P.own_ys = ys; P.own_nw = 16 << ys;
        P.own_sx = (P.TX + 15) / 16; P.own_sy = (P.TY + (1 << ys) - 1) >> ys;
        P.own_slots = nmaps * P.own_sx * P.own_sy;
    }
    size_t own_words() const {           // words of the queue array: enough for either pattern
        const size_t sx = (size_t)(P.TX + 15) / 16;
        return (size_t)nmaps * sx * std::max<size_t>(256 * (size_t)((P.TY + 15) / 16), 512 * (size_t)((P.TY + 31) / 32));
    }
    int profile_stride = 4;          // profiling: every n-th launch of a plan is bracketed by events
    int reset_queues();
    int read_bounds(float *bmax);
    int step(ufm_stats *out);
    int patch(int m, const uint8_t *dev_patch, int x, int y, int w, int h, bool may_defer = false);
    // OPT-IN (ufm_batch_set_param "defer_patches", 1; round 4: it was the default, which silently extended the lifetime the ABI asks of
    // a patch buffer): small patches handed to a batch as device pointers are held back until something needs them applied (the next
    // step, a read of the raster, a path extraction) and applied by ONE launch -- the caller then keeps each buffer valid and unchanged
    // until that call has returned.  Off: every patch is applied at the call, stream-ordered, like a single planner's.
    struct DeferredPatch { int m, x, y, w, h; const uint8_t *ptr; };
    std::vector<DeferredPatch> deferred;
    bool defer_patches = false;
    int flush_deferred();
    int flush_deferred_only();
    // A small patch of a SINGLE planner handed over from HOST memory (ufm_patch_map) is not uploaded and applied at the call: its bytes are
    // copied into a slot of host-coherent pinned memory (the caller's buffer is free again when the call returns, as before) and the replan's
    // block kernel reads them from there and does Graph::update + the seeding itself (RegionJob::psrc) -- no staging copy, no stream
    // synchronisation, no patch kernel in front of the replan.  Whatever else needs the raster first (a read of the map, a path extraction,
    // another kind of patch, a step that does not go through the block kernel) applies the held patches the ordinary way: flush_lazy().
    static constexpr int LAZY_SLOTS = 4;                  // = the most rectangles a block-kernel job takes
    struct LazyPatch { int m, x, y, w, h, slot; };
    std::vector<LazyPatch> lazy;
    uint8_t *h_lazy = nullptr;       // LAZY_SLOTS x 4096 bytes, pinned + mapped
    int lazy_next = 0;               // slot after the last one handed out
    bool lazy_dirty[LAZY_SLOTS] = {false, false, false, false};   // a kernel queued on the stream may still read the slot
    bool lazy_patches = true;
    int patch_lazy(int m, const uint8_t *host_patch, int x, int y, int w, int h, bool *taken);
    int flush_lazy();
    int ensure_pmask(size_t n);
    bool region_fits(const int (*rects)[5], int nrect, int m, int *tx0, int *ntx, int *ty0, int *nty) const;
    bool lazy_region_ok() const;
};

void Engine::release() {
    if (!allocated) return;
    if (stream) hipStreamSynchronize(stream);
    drop_graphs();                       // captured kernel arguments hold these pointers
    deferred.clear();
    std::memset(&graph_sig, 0, sizeof(graph_sig));
    void *ptrs[] = {P.G, P.Gprev, P.bp, P.ring, P.seen, P.cost, P.costT, P.goal, P.cand, P.ready, P.hint, P.rank, P.park, P.pflag, P.pprio,
                    P.queued, P.prio, P.start, P.bnd, P.dyn, P.spos, P.touched, P.fresh, P.tlist, P.sflag, P.slist, P.slist2,
                    P.mark, P.num_updated, P.consume, P.lmax, P.own_prio, P.own_lock, P.own_min, P.dag_a, P.dag_thr, P.dag_left, P.ctr, d_scratch};
    for (void *q : ptrs) if (q) hipFree(q);
    P = DevParams{};                     // every pointer null again: a failed alloc() can be released, and released twice
    d_scratch = nullptr;
    allocated = false;
}

int Engine::alloc(int width, int length) {
    release();
    W = width; L = length;
    const bool nodes = algo != UFM_ALGO_DFM;
    P.W = W; P.L = L;
    P.EX = nodes ? L + 1 : L;
    P.EY = nodes ? W + 1 : W;
    P.TX = (P.EX + T - 1) / T;
    P.TY = (P.EY + T - 1) / T;
    // tile ids are ints; the element count of a map must fit one as well (start elements, marks)
    if ((long long)P.TX * P.TY * nmaps > (long long)INT32_MAX / 4 || (long long)P.EX * P.EY > INT32_MAX) return UFM_ERR_NOMEM;
    P.NTm = P.TX * P.TY;
    P.nmaps = nmaps;
    P.NT = P.NTm * nmaps;
    P.cells = nodes ? 0 : 1;
    P.gstride = (size_t)P.NTm * TT;
    P.cstride = (size_t)L * W;
    P.mstride = (size_t)P.EX * P.EY;
    own_layout(4);
    allocated = true;                    // from here on release() has something to free, also after a failure half way
    const size_t gbytes = P.gstride * nmaps * sizeof(float);
    int rc = UFM_OK;
    auto dmalloc = [&](auto *&ptr, size_t bytes) {
        if (rc != UFM_OK) return;
        void *q = nullptr;
        const hipError_t e = hipMalloc(&q, bytes ? bytes : 1);
        if (e != hipSuccess) { (void)hipGetLastError(); rc = (e == hipErrorOutOfMemory) ? UFM_ERR_NOMEM : UFM_ERR_HIP_BASE - (int)e; return; }
        ptr = static_cast<std::remove_reference_t<decltype(ptr)>>(q);
    };
    dmalloc(P.G, gbytes);
    dmalloc(P.Gprev, gbytes);
    dmalloc(P.bp, P.gstride * nmaps);
    dmalloc(P.ring, (size_t)P.NT * RING * sizeof(float));
    dmalloc(P.seen, UFM_DIRWAKE ? (size_t)P.NT * RING * sizeof(float) : sizeof(float));
    dmalloc(P.cost, P.cstride * nmaps);
    dmalloc(P.costT, (size_t)P.NT * CTS);
    dmalloc(P.goal, sizeof(int) * 2 * nmaps);
    dmalloc(P.cand, sizeof(int) * 6 * P.NT);
    dmalloc(P.ready, sizeof(int) * P.NT);
    dmalloc(P.hint, sizeof(int) * P.NT);
    dmalloc(P.rank, sizeof(int) * P.NT);
    dmalloc(P.park, sizeof(int) * 4 * P.NT);
    dmalloc(P.pflag, sizeof(int) * 2 * P.NT);
    dmalloc(P.pprio, sizeof(int) * 2 * P.NT);
    dmalloc(P.queued, sizeof(int) * 4 * P.NT);
    dmalloc(P.prio, sizeof(unsigned long long) * 4 * P.NT);
    dmalloc(P.start, sizeof(int) * 4 * nmaps);
    dmalloc(P.bnd, sizeof(float) * nmaps);
    dmalloc(P.dyn, sizeof(DevDyn));
    dmalloc(P.spos, sizeof(float) * 2 * nmaps);
    dmalloc(P.touched, sizeof(int) * P.NT);
    dmalloc(P.fresh, (size_t)P.NT);
    dmalloc(P.tlist, sizeof(int) * P.NT);
    dmalloc(P.sflag, sizeof(int) * P.NT);
    dmalloc(P.slist, sizeof(int) * P.NT);
    dmalloc(P.slist2, sizeof(int) * P.NT);
    dmalloc(P.mark, P.mstride * nmaps);
    dmalloc(P.num_updated, sizeof(unsigned int) * nmaps);
    dmalloc(P.consume, sizeof(int) * nmaps);
    dmalloc(P.lmax, sizeof(int) * LMAX);
    dmalloc(P.own_prio, sizeof(int) * own_words());
    dmalloc(P.own_lock, sizeof(int) * own_words());
    dmalloc(P.own_min, sizeof(int) * OWN_NW);
    dag_have = false;
    dmalloc(P.dag_a, sizeof(float) * (size_t)P.NT);
    dmalloc(P.dag_thr, sizeof(float) * (size_t)P.NT);
    dmalloc(P.dag_left, sizeof(int) * own_words());
    dmalloc(P.ctr, sizeof(DevCounters));
    dmalloc(d_scratch, sizeof(int) * (4 * nmaps + 16));
    if (rc != UFM_OK) { release(); return rc; }
    rc = [&]() -> int {
        // (the lists: a slot that was never written must still read as a tile id -- an in-launch reader may look at a slot its writer has claimed
        //  but not yet stored, see the end check of k_replan_region)
        HIPCHK(hipMemsetAsync(P.cand, 0, sizeof(int) * 6 * P.NT, stream));
        HIPCHK(hipMemsetAsync(P.park, 0, sizeof(int) * 4 * P.NT, stream));
        HIPCHK(hipMemsetAsync(P.ready, 0, sizeof(int) * P.NT, stream));
        HIPCHK(hipMemsetAsync(P.tlist, 0, sizeof(int) * P.NT, stream));
        HIPCHK(hipMemsetAsync(P.slist, 0, sizeof(int) * P.NT, stream));
        HIPCHK(hipMemsetAsync(P.slist2, 0, sizeof(int) * P.NT, stream));
        HIPCHK(hipMemsetAsync(P.rank, 0, sizeof(int) * P.NT, stream));
        HIPCHK(hipMemsetAsync(P.hint, 0, sizeof(int) * P.NT, stream));
        HIPCHK(hipMemsetAsync(P.spos, 0, sizeof(float) * 2 * nmaps, stream));
        HIPCHK(hipMemsetAsync(P.start, 0xFF, sizeof(int) * 4 * nmaps, stream));
        HIPCHK(hipMemsetAsync(P.ctr, 0, sizeof(DevCounters), stream));
        { int rq = reset_queues(); if (rq != UFM_OK) return rq; }
        HIPCHK(hipMemsetAsync(P.touched, 0, sizeof(int) * P.NT, stream));
        HIPCHK(hipMemsetAsync(P.fresh, 0, (size_t)P.NT, stream));
        HIPCHK(hipMemsetAsync(P.sflag, 0, sizeof(int) * P.NT, stream));
        HIPCHK(hipMemsetAsync(P.mark, 0, P.mstride * nmaps, stream));
        HIPCHK(hipMemsetAsync(P.num_updated, 0, sizeof(unsigned int) * nmaps, stream));
        HIPCHK(hipMemsetAsync(P.goal, 0xFF, sizeof(int) * 2 * nmaps, stream));
        k_fill<<<1024, 256, 0, stream>>>(P.G, P.gstride * nmaps, INFINITY);
        k_fill<<<1024, 256, 0, stream>>>(P.Gprev, P.gstride * nmaps, INFINITY);
        HIPCHK(hipMemsetAsync(P.bp, BP_NONE, P.gstride * nmaps, stream));
        k_fill<<<1024, 256, 0, stream>>>(P.ring, (size_t)P.NT * RING, INFINITY);
        dyn_dev = DevDyn{heur ? heuristic_multiplier : 0.0f, thr_uchar, focused ? 1 : 0, 0};
        k_set_dyn<<<1, 1, 0, stream>>>(P.dyn, dyn_dev);
        HIPCHK(hipGetLastError());
        return UFM_OK;
    }();
    if (rc != UFM_OK) { release(); return rc; }
    pending.clear();
    for (auto &ms : maps) { ms.have_map = false; ms.initialize_search = true; }
    return UFM_OK;
}

// drop every queued tile (full re-initialisation: nothing of the old search survives)
int Engine::reset_queues() {
    HIPCHK(hipMemsetAsync(P.queued, 0, sizeof(int) * 4 * P.NT, stream));
    HIPCHK(hipMemsetAsync(P.prio, 0xFF, sizeof(unsigned long long) * 4 * P.NT, stream));   // tag of no launch, larger than any key
    HIPCHK(hipMemsetAsync(P.pflag, 0, sizeof(int) * 2 * P.NT, stream));
    k_fill<<<64, 256, 0, stream>>>(reinterpret_cast<float *>(P.pprio), (size_t)2 * P.NT, INFINITY);
    k_fill<<<64, 256, 0, stream>>>(reinterpret_cast<float *>(P.own_prio), own_words(), INFINITY);   // (+inf = INFBITS: empty)
    HIPCHK(hipMemsetAsync(P.own_lock, 0, sizeof(int) * own_words(), stream));
    k_fill<<<2, 256, 0, stream>>>(reinterpret_cast<float *>(P.own_min), (size_t)OWN_NW, INFINITY);
    // the queue state at the head of DevCounters: cnt, rel, lmin, npark, nready, rcursor, nshort, last_work, fin_blocks
    static_assert(offsetof(DevCounters, cnt) == 0, "queue state leads the counter block");
    HIPCHK(hipMemsetAsync(P.ctr, 0, offsetof(DevCounters, kbase), stream));
    k_fill<<<1, 64, 0, stream>>>(reinterpret_cast<float *>(&P.ctr->lmin[0][0]), (size_t)6, INFINITY);
    last_active = 1;
    iter[0] = iter[1] = 0;
    return UFM_OK;
}
// largest start key over the maps (+inf if some map's start is not reached yet)
int Engine::read_bounds(float *bmax) {
    k_start_bound<<<(nmaps + 63) / 64, 64, 0, stream>>>(P);
    HIPCHK(hipMemcpyAsync(h_bnd, P.bnd, sizeof(float) * nmaps, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    float b = 0.0f;
    for (int m = 0; m < nmaps; ++m) b = std::fmax(b, h_bnd[m]);
    *bmax = b;
    return UFM_OK;
}

// Counters to the host.  A D2H copy + hipStreamSynchronize costs ~40 us of wake-up latency per
// host round trip (measured: copy done at 289 us, host running again at 327 us); a replan has one
// round trip, a plan one per batch of launches.  Instead the last kernel of a submission writes
// the counters into host-coherent pinned memory, fences, and bumps a sequence number the host
// spins on (the reference's driver owns its core anyway, main.cpp:36-47).
__global__ void k_publish(const DevCounters *src, DevCounters *dst, unsigned int *flag, unsigned int seq) {
    const int *s = reinterpret_cast<const int *>(src);
    int *d = reinterpret_cast<int *>(dst);
    for (int i = threadIdx.x; i < (int)(sizeof(DevCounters) / sizeof(int)); i += blockDim.x) d[i] = s[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
int Engine::fetch_counters() {
    if (!spin_wait) {
        HIPCHK(hipMemcpyAsync(h_ctr, P.ctr, sizeof(DevCounters), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        return UFM_OK;
    }
    ++pub_seq;
    k_publish<<<1, 64, 0, stream>>>(P.ctr, h_ctr, h_flag, pub_seq);
    HIPCHK(hipGetLastError());
    return wait_published();
}
// spin until the device has published copy number pub_seq
int Engine::wait_published() { return wait_flag(h_flag, pub_seq); }
int Engine::wait_flag(const unsigned int *flag, unsigned int seq) {
    const auto t0 = std::chrono::steady_clock::now();
    auto next_query = t0 + std::chrono::milliseconds(200);
    for (unsigned int spins = 1;; ++spins) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return UFM_OK;
        __builtin_ia32_pause();
        if ((spins & 0xFFF) != 0) continue;
        // a faulted kernel never publishes: every 200 ms ask the runtime whether the stream is still alive
        const auto now = std::chrono::steady_clock::now();
        if (now < next_query) continue;
        next_query = now + std::chrono::milliseconds(200);
        const hipError_t q = hipStreamQuery(stream);
        if (q == hipErrorNotReady) continue;
        if (q != hipSuccess) return UFM_ERR_HIP_BASE - (int)q;
        break;                          // the stream has drained: the flag is there by now, or it never will be
    }
    return __atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq ? UFM_OK : UFM_ERR_HIP_BASE;
}

// one relax launch over a short queue (fused triage) with an explicit launch-index argument
int Engine::relax_kernel(int mode, int k_arg, float rbound, int grid) {
    const dim3 g(grid), b(NTHR);
    const float delta = (mode == MODE_RAISE) ? INFINITY : (delta_abs >= 0.0f ? delta_abs : delta_scale * T * mean_cost);
#define UFM_LAUNCH(A, M) k_relax<A, M, false><<<g, b, 0, stream>>>(P, k_arg, delta, rbound, max_iters)
    if (mode == MODE_LOWER) {
        if (algo == UFM_ALGO_FD) UFM_LAUNCH(UFM_ALGO_FD, MODE_LOWER);
        else if (algo == UFM_ALGO_SG) UFM_LAUNCH(UFM_ALGO_SG, MODE_LOWER);
        else UFM_LAUNCH(ALGO_DFM1, MODE_LOWER);
    } else {
        if (algo == UFM_ALGO_FD) UFM_LAUNCH(UFM_ALGO_FD, MODE_RAISE);
        else if (algo == UFM_ALGO_SG) UFM_LAUNCH(UFM_ALGO_SG, MODE_RAISE);
        else UFM_LAUNCH(ALGO_DFM1, MODE_RAISE);
    }
#undef UFM_LAUNCH
    return UFM_OK;
}
// The whole replan submission -- begin, nr invalidation launches, transition, nl lowering
// launches, end -- captured once per (nr, nl) and replayed: the host enqueues one graph instead of
// ~20 kernels (2.9 us of host time each, measured; the kernels of a replan are that short).  The
// graph is static: launch indices are offsets to a base the first node takes, with the rest of
// the per-replan inputs, from host-coherent memory (h_job).
int Engine::replan_graph(int nr, int nl, float band, hipGraphExec_t *out) {
    GraphSig sig{};
    sig.P = P; sig.band = band; sig.delta = delta_abs >= 0.0f ? delta_abs : delta_scale * T * mean_cost;
    sig.max_iters = max_iters; sig.grid = grid_relax * 4096 + tail_grid;
    if (std::memcmp(&sig, &graph_sig, sizeof(GraphSig)) != 0) { drop_graphs(); std::memcpy(&graph_sig, &sig, sizeof(GraphSig)); }
    const int key = nr * 256 + nl;
    for (auto &g : graphs) if (g.first == key) { *out = g.second; return UFM_OK; }
    if (graphs.size() >= 64) drop_graphs();
    HIPCHK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
    k_replan_begin_job<<<1, 1024, 0, stream>>>(P, h_job);
    // a phase starts with its largest launches; what is left after a few of them fits a small
    // grid, which starts -- and, when the queue has run dry, ends -- sooner (an empty 512-workgroup
    // launch lasts 4.6 us)
    auto grid_of = [&](int i) { return i < 2 ? grid_relax : (i < 4 ? std::max(tail_grid, grid_relax / 2) : tail_grid); };
    for (int i = 0; i < nr; ++i) relax_kernel(MODE_RAISE, -1 - i, -1.0f, std::min(grid_relax, grid_of(i)));
    k_raise_to_lower<<<1, 1024, 0, stream>>>(P, -1);
    for (int i = 0; i < nl; ++i) relax_kernel(MODE_LOWER, -1 - i, INFINITY, std::min(grid_relax, grid_of(i)));
    k_replan_end<<<64, T * T, 0, stream>>>(P, -1 - nr, -1 - nl, band, h_ctr, h_flag, 0u);
    finalize_bp(1);     // (behind the publication: the host does not wait for it, the next step's kernels do)
    hipGraph_t g = nullptr;
    HIPCHK(hipStreamEndCapture(stream, &g));
    hipGraphExec_t ge = nullptr;
    const hipError_t err = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphDestroy(g);
    HIPCHK(err);
    graphs.emplace_back(key, ge);
    ++graphs_made;
    *out = ge;
    return UFM_OK;
}

// e0 / e1 (profiling): HIP events recorded on the engine's stream right around the relax kernel
int Engine::launch_relax(int mode, float rbound, hipEvent_t e0, hipEvent_t e1) {
    dim3 g(grid_relax), b(NTHR);
    const int q = (mode == MODE_LOWER) ? Q_LOWER : Q_RAISE;
    // long queue: vectorised triage + balanced hand-out of the released tiles; short queue: fused
    const bool dyn = dynamic_mode && last_active > grid_relax / 4;
    // invalidation is order-free; lowering releases tiles in bands of `delta`
    const float delta = (mode == MODE_RAISE) ? INFINITY : (delta_abs >= 0.0f ? delta_abs : (dyn ? delta_scale_long : delta_scale) * T * mean_cost);
    // a short queue (replans: a handful of tiles per launch) does not need the whole chip: a small
    // grid starts, and when there is nothing left to do ends, sooner
    if (!dyn && last_active <= small_grid / 2 && small_grid < grid_relax) g = dim3(small_grid);
    if (dyn && UFM_STATIC_FIRST) g = dim3(std::min(grid_relax, dyn_grid));   // one resident workgroup per CU
    if (dyn) {
        if (mode == MODE_LOWER) k_triage<MODE_LOWER><<<64, 256, 0, stream>>>(P, iter[q], delta, rbound);
        else k_triage<MODE_RAISE><<<64, 256, 0, stream>>>(P, iter[q], delta, rbound);
    }
    // timed launch: the events are attached to the dispatch itself (start / stop time stamps of the
    // kernel, what rocprofv3 reports too), not recorded around it as separate packets
    const bool timed = e0 && e1;
    const int kk = iter[q];
    const int ms_ = max_iters;
#define UFM_LAUNCH(A, M) do { \
        if (timed) { if (dyn) hipExtLaunchKernelGGL((k_relax<A, M, true>), g, b, 0, stream, e0, e1, 0, P, kk, delta, rbound, ms_); \
                     else hipExtLaunchKernelGGL((k_relax<A, M, false>), g, b, 0, stream, e0, e1, 0, P, kk, delta, rbound, ms_); } \
        else if (dyn) k_relax<A, M, true><<<g, b, 0, stream>>>(P, kk, delta, rbound, ms_); \
        else k_relax<A, M, false><<<g, b, 0, stream>>>(P, kk, delta, rbound, ms_); } while (0)
    if (mode == MODE_LOWER) {
        if (algo == UFM_ALGO_FD) UFM_LAUNCH(UFM_ALGO_FD, MODE_LOWER);
        else if (algo == UFM_ALGO_SG) UFM_LAUNCH(UFM_ALGO_SG, MODE_LOWER);
        else UFM_LAUNCH(ALGO_DFM1, MODE_LOWER);
    } else {
        if (algo == UFM_ALGO_FD) UFM_LAUNCH(UFM_ALGO_FD, MODE_RAISE);
        else if (algo == UFM_ALGO_SG) UFM_LAUNCH(UFM_ALGO_SG, MODE_RAISE);
        else UFM_LAUNCH(ALGO_DFM1, MODE_RAISE);
    }
#undef UFM_LAUNCH
    ++iter[q];
    return UFM_OK;
}

// Launch relax kernels until the active list runs dry.  The list lengths live on
// the device; the host peeks at them once per batch of launches (an empty launch
// costs a few microseconds, a host round trip more).
// A phase also ends when a launch released nothing: everything still queued lies beyond the bound
// (the start's key) and stays queued for a later step.
// The host does not wait for a batch before it submits the next one: while it reads the counters batch
// b published, batch b+1 is already running (a host round trip -- publish, PCIe, decision, first
// dispatch -- left the GPU idle for ~15 us, 68 times per 4096^2 plan).  The price: when batch b turns
// out to have drained the queue, batch b+1 consists of launches that find nothing to do (a few us each).
// A whole lowering phase in one launch: the resident kernel (k_relax<., LOWER, false, 1 | 2>) between the two kernels that
// move the queue into and out of its per-owner words.  What it leaves behind is an ordinary (short or empty) list
// for launch iter + 1, which run_phase() then finds.
int Engine::owned_phase() {
    const int k = iter[Q_LOWER];
    // the ordering band in tile crossings.  Round 2: 4 (a workgroup that finds nothing inside the band idles, so wider paid).  With the idle
    // workgroups' looks made cheap (round 3) the optimum moved down -- 4096^2 FD 14.7 / 14.2 / 14.2 / 14.2 / 14.6 ms for 1.5 / 2 / 2.5 / 3 / 4 with 230 k ... 350 k
    // visits, 8192^2 36.7 / 35.8 / 35.5 / 36.0 / 37.4, SG 2048^2 5.85 / 5.4 / 5.3 / 5.4 / 5.7; MS-DFM 2048^2 12.0 / 11.8 / 11.4 / 11.7 (2 ... 4); the 8-map
    // MS-DFM batch, bound by the number of visits: 34.0 / 34.3 / 34.9 / 35.8 (2 ... 4)
    const float band_auto = nmaps > 1 ? 2.0f : (algo == UFM_ALGO_DFM ? 3.0f : 2.5f);
    const float delta = delta_abs >= 0.0f ? delta_abs : (owned_band >= 0.0f ? owned_band : band_auto) * T * mean_cost;
    const double limit_ms = owned_limit_ms >= 0.0f ? (double)owned_limit_ms : 200.0 + (double)P.NT / 250.0;   // (4096^2: 0.46 s; its plan takes 17 ms)
    P.own_limit = (unsigned long long)(limit_ms * 1e5);   // 100 MHz ticks
    P.own_flags = owned_flags;
    // (measured with the helping workgroups in place: FD 4096^2 15.9-16.3 ms with 8 waves against 16.5-16.8 with 16, 2048^2 7.25 against 6.44,
    //  SG 2048^2 7.08 against 6.55, 1024^2 3.40 against 2.84; MS-DFM, whose visits are longer and which has no early hand-off, 2048^2 13.1 against 15.0)
    const bool half = T == 16 && (owned_waves == 8 || (owned_waves == 0 && (nmaps > 1 || P.NTm > (algo == UFM_ALGO_DFM ? 12000 : 50000))));
    own_layout(half ? 5 : 4);
    P.dag_on = (DAG && dag_mode != 0 && dag_have) ? 1 : 0;
    P.dag_patience = dag_patience;
    if (P.dag_on) k_dag_setup<<<256, 256, 0, stream>>>(P, dag_kappa);
    k_own_import<<<64, 256, 0, stream>>>(P, k);
    // 16 waves per tile visit, one visit per CU -- or 8 and two: a visit is then ~17 % longer and a CU makes 1.7 x as many.  That pays
    // where there are always more tiles to visit than workgroups (several maps, or a front as long as that of an 8192^2 map); a single
    // 4096^2 plan is bound by the chain of dependent visits along the front's way, not by their number
    const dim3 g(P.own_nw), b(half ? NTHR / 2 : NTHR);
    const int ms_ = max_iters;
    own_timed = false;
    if (profiling) {
        for (auto &e : own_ev) if (!e) HIPCHK(hipEventCreate(&e));
        own_timed = true;
    }
#if UFM_TILE == 16
#define UFM_LAUNCH(A) do { if (own_timed) { if (half) hipExtLaunchKernelGGL((k_relax<A, MODE_LOWER, false, 2>), g, b, 0, stream, own_ev[0], own_ev[1], 0, P, k, delta, INFINITY, ms_); \
                                                else hipExtLaunchKernelGGL((k_relax<A, MODE_LOWER, false, 1>), g, b, 0, stream, own_ev[0], own_ev[1], 0, P, k, delta, INFINITY, ms_); } \
                           else if (half) k_relax<A, MODE_LOWER, false, 2><<<g, b, 0, stream>>>(P, k, delta, INFINITY, ms_); \
                           else k_relax<A, MODE_LOWER, false, 1><<<g, b, 0, stream>>>(P, k, delta, INFINITY, ms_); } while (0)
#else   // 32 x 32 tiles: the 16-wave form only (the skewed 8-wave patch map is written for 4 x 4 patches per tile)
#define UFM_LAUNCH(A) do { if (own_timed) hipExtLaunchKernelGGL((k_relax<A, MODE_LOWER, false, 1>), g, b, 0, stream, own_ev[0], own_ev[1], 0, P, k, delta, INFINITY, ms_); \
                           else k_relax<A, MODE_LOWER, false, 1><<<g, b, 0, stream>>>(P, k, delta, INFINITY, ms_); } while (0)
#endif
    if (algo == UFM_ALGO_FD) UFM_LAUNCH(UFM_ALGO_FD);
    else if (algo == UFM_ALGO_SG) UFM_LAUNCH(UFM_ALGO_SG);
    else UFM_LAUNCH(ALGO_DFM1);
#undef UFM_LAUNCH
    k_own_export<<<256, 256, 0, stream>>>(P, k + 1);
    HIPCHK(hipGetLastError());
    ++iter[Q_LOWER];
    last_active = 1;
    ++owned_launches;
    return UFM_OK;
}

int Engine::run_phase(int mode, float rbound, uint32_t *launches, float *kernel_ms, uint32_t *timed) {
    const int q = (mode == MODE_LOWER) ? Q_LOWER : Q_RAISE;
    int batch = batch_fixed > 0 ? batch_fixed : 4;
    const long cap = 64L * (P.TX + P.TY) * T + 4096;   // generous bound on sweeps
    long total = 0;
    if (spin_wait && pipeline_batches && h_pipe_ctr[0]) {
        struct InFlight { unsigned int seq; int slot, ns, iter_after; };
        const int EVSLOT = 2 * std::max(32, batch_fixed);   // events per slot: two per launch of a batch (adaptive batches: <= 32 launches)
        while (profiling && ev.size() < (size_t)(4 + 2 * EVSLOT)) { hipEvent_t a; HIPCHK(hipEventCreate(&a)); ev.push_back(a); }
        auto collect = [&](const InFlight &f) -> int {  // wait for the batch, add its timed launches
            int rc = wait_flag(h_pipe_flag[f.slot], f.seq);
            if (rc != UFM_OK) return rc;
            for (int k = 0; k < f.ns; ++k) {
                float ms = 0;
                HIPCHK(hipEventElapsedTime(&ms, ev[4 + f.slot * EVSLOT + 2 * k], ev[4 + f.slot * EVSLOT + 2 * k + 1]));
                *kernel_ms += ms;
            }
            *timed += (uint32_t)f.ns;
            return UFM_OK;
        };
        InFlight prev{}, cur{};
        bool have_prev = false;
        int slot = 0;
        for (;;) {
            int ns = 0;
            for (int k = 0; k < batch; ++k) {
                const bool timed_k = profiling && ((total + k) % profile_stride == 0);
                const int eb = 4 + slot * EVSLOT + 2 * ns;
                launch_relax(mode, rbound, timed_k ? ev[eb] : nullptr, timed_k ? ev[eb + 1] : nullptr);
                if (timed_k) ++ns;
            }
            ++pub_seq;
            k_publish<<<1, 64, 0, stream>>>(P.ctr, h_pipe_ctr[slot], h_pipe_flag[slot], pub_seq);
            HIPCHK(hipGetLastError());
            cur = {pub_seq, slot, ns, iter[q]};
            *launches += (uint32_t)batch;
            total += batch;
            if (have_prev) {
                int rc = collect(prev);
                if (rc != UFM_OK) return rc;
                const DevCounters *c = h_pipe_ctr[prev.slot];
                const int active = c->cnt[q][prev.iter_after % 3];
                const bool done = active == 0 || c->rel[q][(prev.iter_after + 2) % 3] == 0;   // drained / nothing released: the rest lies beyond the bound
                if (done || total > cap) {
                    rc = collect(cur);                  // the batch submitted meanwhile found nothing to do
                    if (rc != UFM_OK) return rc;
                    last_active = h_pipe_ctr[cur.slot]->cnt[q][cur.iter_after % 3];
                    return done ? UFM_OK : UFM_ERR_NOT_CONVERGED;
                }
                last_active = active;
                batch = batch_fixed > 0 ? batch_fixed : (active > 512 ? 32 : (active > 256 ? 16 : (active > 32 ? 8 : 4)));
            }
            prev = cur; have_prev = true; slot ^= 1;
        }
    }
    for (;;) {
        int ns = 0;   // launches of this batch that are timed: a sample, the event packets cost ~4 us each
        for (int k = 0; k < batch; ++k) {
            const bool timed_k = profiling && ((total + k) % profile_stride == 0);
            if (timed_k) {
                while (ev.size() < (size_t)(2 * (ns + 1) + 4)) {   // ev[0..3] belong to the replan path
                    hipEvent_t a;
                    HIPCHK(hipEventCreate(&a));
                    ev.push_back(a);
                }
            }
            launch_relax(mode, rbound, timed_k ? ev[4 + 2 * ns] : nullptr, timed_k ? ev[4 + 2 * ns + 1] : nullptr);
            if (timed_k) ++ns;
        }
        HIPCHK(hipGetLastError());
        *launches += (uint32_t)batch;
        total += batch;
        { int rc = fetch_counters(); if (rc != UFM_OK) return rc; }
        for (int k = 0; k < ns; ++k) {
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, ev[4 + 2 * k], ev[4 + 2 * k + 1]));
            *kernel_ms += ms;
        }
        *timed += (uint32_t)ns;
        const int active = h_ctr->cnt[q][iter[q] % 3];
        last_active = active;
        if (active == 0) return UFM_OK;
        if (h_ctr->rel[q][(iter[q] + 2) % 3] == 0) return UFM_OK;   // the last launch released nothing
        if (total > cap) return UFM_ERR_NOT_CONVERGED;
        batch = batch_fixed > 0 ? batch_fixed : (active > 512 ? 32 : (active > 256 ? 16 : (active > 32 ? 8 : 4)));
    }
}

int Engine::flush_deferred() {       // every patch that is being held -- a single planner's host patches, a batch's deferred device patches -- applied
    { int rc = flush_lazy(); if (rc != UFM_OK) return rc; }
    return flush_deferred_only();
}
int Engine::flush_deferred_only() {
    if (deferred.empty()) return UFM_OK;
    PatchMulti a{};
    a.n = (int)deferred.size();
    for (int i = 0; i < a.n; ++i) {
        const DeferredPatch &d = deferred[i];
        int *q = a.rect[i]; q[0] = d.m; q[1] = d.x; q[2] = d.y; q[3] = d.w; q[4] = d.h;
        a.ptr[i] = d.ptr;
    }
    deferred.clear();
    if (algo == UFM_ALGO_DFM) k_patch_multi<false><<<a.n, 1024, 0, stream>>>(P, a, d_pmask);
    else k_patch_multi<true><<<a.n, 1024, 0, stream>>>(P, a, d_pmask);
    HIPCHK(hipGetLastError());
    return UFM_OK;
}

int Engine::ensure_pmask(size_t n) {   // room for the masks of PATCH_MULTI small patches, or of one large one
    const size_t need = std::max((size_t)PATCH_MULTI * 4096, n);
    if (need > d_pmask_cap) {
        { int rc = flush_deferred_only(); if (rc != UFM_OK) return rc; }      // (their launch writes the old buffer)
        if (d_pmask) { HIPCHK(hipStreamSynchronize(stream)); hipFree(d_pmask); d_pmask = nullptr; d_pmask_cap = 0; }
        HIPCHK(hipMalloc(&d_pmask, need));
        d_pmask_cap = need;
    }
    return UFM_OK;
}
// the held host patches, applied the ordinary way (one k_patch_small each, reading the pinned slot), in the order they came
int Engine::flush_lazy() {
    if (lazy.empty()) return UFM_OK;
    { int rc = ensure_pmask(4096); if (rc != UFM_OK) return rc; }
    for (const LazyPatch &p : lazy) {
        const uint8_t *src = h_lazy + (size_t)p.slot * 4096;
        if (algo == UFM_ALGO_DFM) k_patch_small<false><<<1, 1024, 0, stream>>>(P, p.m, src, d_pmask, p.x, p.y, p.w, p.h);
        else k_patch_small<true><<<1, 1024, 0, stream>>>(P, p.m, src, d_pmask, p.x, p.y, p.w, p.h);
        lazy_dirty[p.slot] = true;         // (read by a kernel that is only queued: patch_lazy waits for the stream before it writes the slot again)
    }
    lazy.clear();
    HIPCHK(hipGetLastError());
    return UFM_OK;
}
// ufm_patch_map of a small patch, single planner: hold it (see the member's comment); *taken = false: the caller goes the ordinary way
int Engine::patch_lazy(int m, const uint8_t *host_patch, int x, int y, int w, int h, bool *taken) {
    *taken = false;
    if (m < 0 || m >= nmaps || !allocated || !maps[m].have_map) return UFM_ERR_INVALID;
    if (x < 0 || y < 0 || w <= 0 || h <= 0 || x + h > L || y + w > W) return UFM_ERR_INVALID;   // Graph.cpp:38-41
    if (!(lazy_patches && nmaps == 1 && use_region && fuse_control && spin_wait && w <= 64 && h <= 64)) return UFM_OK;
    if ((int)lazy.size() >= LAZY_SLOTS || pending.size() != lazy.size()) return UFM_OK;      // (only behind other held patches: the block kernel applies them in order)
    if (!h_lazy) HIPCHK(hipHostMalloc(&h_lazy, (size_t)LAZY_SLOTS * 4096, hipHostMallocMapped));
    // (a slot is free again when the step that consumed its patch has returned -- step() is synchronous -- or when flush_lazy() has run and the
    //  stream has been waited for; slots are handed out in order, so the one after the last held patch's is the oldest)
    const int slot = lazy.empty() ? (lazy_next % LAZY_SLOTS) : ((lazy.back().slot + 1) % LAZY_SLOTS);
    if (lazy_dirty[slot]) { HIPCHK(hipStreamSynchronize(stream)); for (bool &d : lazy_dirty) d = false; }
    std::memcpy(h_lazy + (size_t)slot * 4096, host_patch, (size_t)w * h);
    __atomic_thread_fence(__ATOMIC_RELEASE);
    lazy.push_back({m, x, y, w, h, slot});
    lazy_next = slot + 1;
    pending.push_back({m, x, y, w, h});
    *taken = true;
    return UFM_OK;
}
// the block of the replan kernel around a set of consumed rectangles {map, x, y, w, h}: its goal-side edge `region_ahead` tiles beyond the
// rectangles' centre, the rest of its extent behind it -- where the elements that lean on the patched cells are; false if they do not fit into one block
bool Engine::region_fits(const int (*rects)[5], int nrect, int m, int *tx0, int *ntx, int *ty0, int *nty) const {
    if (nrect <= 0) return false;
    const bool nodes = algo != UFM_ALGO_DFM;
    int ex0 = INT32_MAX, ex1 = -1, ey0 = INT32_MAX, ey1 = -1;
    for (int r = 0; r < nrect; ++r) {
        const int *qr = rects[r];
        ex0 = std::min(ex0, qr[1]); ex1 = std::max(ex1, qr[1] + qr[4] - (nodes ? 0 : 1));
        ey0 = std::min(ey0, qr[2]); ey1 = std::max(ey1, qr[2] + qr[3] - (nodes ? 0 : 1));
    }
    auto place = [&](int e0, int e1, int goal_e, int ntiles_map, int *t0, int *nt) {
        *nt = std::min(std::min(region_tiles, RTMAX), ntiles_map);
        const int tc = ((e0 + e1) / 2) / T;
        int lo = (goal_e >= (e0 + e1) / 2) ? tc + region_ahead - *nt + 1 : tc - region_ahead;
        lo = std::max(0, std::min(lo, ntiles_map - *nt));
        *t0 = lo;
        return e0 / T >= lo && e1 / T <= lo + *nt - 1;      // every consumed rectangle inside the block
    };
    const bool okx = place(ex0, ex1, maps[m].goal_ex, P.TX, tx0, ntx);
    const bool oky = place(ey0, ey1, maps[m].goal_ey, P.TY, ty0, nty);
    return okx && oky;
}
// will the next step() send the pending patches (the held ones among them) through the block kernel?  The conditions of step()'s fast path.
bool Engine::lazy_region_ok() const {
    if (!(nmaps == 1 && fuse_control && spin_wait && use_region)) return false;
    const MapState &ms = maps[0];
    if (!ms.have_map || !ms.goal_set || ms.initialize_search || ms.new_goal || !ms.new_start) return false;
    if (pending.empty() || pending.size() > 4) return false;
    int rects[4][5], n = 0;
    for (const PatchRect &r : pending) {
        if ((r.w + 1) * (r.h + 1) > 65 * 65) return false;
        int *q = rects[n++]; q[0] = r.m; q[1] = r.x; q[2] = r.y; q[3] = r.w; q[4] = r.h;
    }
    int a, b, c, d;
    return region_fits(rects, n, 0, &a, &b, &c, &d);
}

int Engine::patch(int m, const uint8_t *dev_patch, int x, int y, int w, int h, bool may_defer) {
    if (m < 0 || m >= nmaps || !allocated || !maps[m].have_map) return UFM_ERR_INVALID;
    if (x < 0 || y < 0 || w <= 0 || h <= 0 || x + h > L || y + w > W) return UFM_ERR_INVALID;   // Graph.cpp:38-41
    const int n = w * h;
    { int rc = flush_lazy(); if (rc != UFM_OK) return rc; }       // (held host patches come first)
    { int rc = ensure_pmask((size_t)n); if (rc != UFM_OK) return rc; }
    // (a batch only: the patch kernel of a single map runs while the host prepares the step -- applying it inside the
    //  replan's block kernel instead was tried and saved nothing, it only made that kernel longer)
    if (may_defer && defer_patches && nmaps > 1 && n <= 4096) {
        // (one per map at a time: two patches of one map may overlap, and then their order counts)
        bool clash = (int)deferred.size() >= PATCH_MULTI;
        for (const DeferredPatch &d : deferred) clash = clash || d.m == m;
        if (clash) { int rc = flush_deferred(); if (rc != UFM_OK) return rc; }
        deferred.push_back({m, x, y, w, h, dev_patch});
        pending.push_back({m, x, y, w, h});
        return UFM_OK;
    }
    { int rc = flush_deferred(); if (rc != UFM_OK) return rc; }   // keep the order of the patches
    if (n <= 4096) {
        if (algo == UFM_ALGO_DFM) k_patch_small<false><<<1, 1024, 0, stream>>>(P, m, dev_patch, d_pmask, x, y, w, h);
        else k_patch_small<true><<<1, 1024, 0, stream>>>(P, m, dev_patch, d_pmask, x, y, w, h);
    } else {
        k_patch_apply<<<(n + 255) / 256, 256, 0, stream>>>(P, m, dev_patch, d_pmask, x, y, w, h);
        const int ne = (w + 1) * (h + 1);
        if (algo == UFM_ALGO_DFM) k_patch_seed<false><<<(ne + 255) / 256, 256, 0, stream>>>(P, m, d_pmask, x, y, w, h);
        else k_patch_seed<true><<<(ne + 255) / 256, 256, 0, stream>>>(P, m, d_pmask, x, y, w, h);
    }
    HIPCHK(hipGetLastError());
    pending.push_back({m, x, y, w, h});
    return UFM_OK;
}

int Engine::step(ufm_stats *out) {
    // ReplannerBase.h:44-45
    for (int m = 0; m < nmaps; ++m) if (!maps[m].have_map) return UFM_LOOP_FAILURE_NO_GRAPH;
    for (int m = 0; m < nmaps; ++m) if (!maps[m].goal_set) return UFM_LOOP_FAILURE_NO_GOAL;
    ufm_stats st{};
    const auto t0 = std::chrono::steady_clock::now();
    const bool single = (nmaps == 1);
    // held host patches: the replan's block kernel applies them itself if this step goes that way; otherwise now, the ordinary way
    const bool lazy_in_kernel = !lazy.empty() && lazy_region_ok();
    if (!lazy_in_kernel) { int rc = flush_deferred(); if (rc != UFM_OK) return rc; }   // (and a batch's deferred device patches: one launch)

    if (!single) {
        HIPCHK(hipMemsetAsync(&P.ctr->tcount, 0, sizeof(int), stream));
        HIPCHK(hipMemsetAsync(&P.ctr->expanded, 0, 4 * sizeof(unsigned long long), stream));
        HIPCHK(hipMemsetAsync(&P.ctr->raise_visits, 0, sizeof(unsigned long long), stream));
        if (profiling) HIPCHK(hipMemsetAsync(P.lmax, 0, sizeof(int) * LMAX, stream));
    }
    // heuristic multiplier / threshold / focused flag live in device memory (DevDyn); a changed value reaches the
    // device with the first kernel of the step: through the replan graph's job record, or by k_set_dyn
    const DevDyn dyn_now{heur ? heuristic_multiplier : 0.0f, thr_uchar, focused ? 1 : 0, 0};
    bool dyn_pending = std::memcmp(&dyn_now, &dyn_dev, sizeof(DevDyn)) != 0;
    auto flush_dyn = [&]() {
        if (!dyn_pending) return;
        k_set_dyn<<<1, 1, 0, stream>>>(P.dyn, dyn_now);
        dyn_dev = dyn_now; dyn_pending = false;
    };

    // classify maps: (re)initialise, propagate pending patches, or idle  (ReplannerBase.h:48-59)
    int n_init = 0, n_upd = 0;
    int *consume = h_scratch, *init_tiles = h_scratch + nmaps, *goals = h_scratch + 2 * nmaps;
    {   // a full re-initialisation drops whatever the old search left queued
        bool all_init = true;
        for (int m = 0; m < nmaps; ++m) all_init = all_init && (maps[m].initialize_search || maps[m].new_goal);
        if (all_init) { int rc = reset_queues(); if (rc != UFM_OK) return rc; }
    }
    for (int m = 0; m < nmaps; ++m) {
        MapState &ms = maps[m];
        consume[m] = 0;
        if (ms.initialize_search || ms.new_goal) {
            consume[m] = 1;
            goals[2 * m] = ms.goal_elem_valid ? ms.goal_ex : -1;
            goals[2 * m + 1] = ms.goal_elem_valid ? ms.goal_ey : -1;
            k_fill<<<1024, 256, 0, stream>>>(P.G + (size_t)m * P.gstride, P.gstride, INFINITY);
            HIPCHK(hipMemsetAsync(P.bp + (size_t)m * P.gstride, BP_NONE, P.gstride, stream));
            k_fill<<<256, 256, 0, stream>>>(P.ring + (size_t)m * P.NTm * RING, (size_t)P.NTm * RING, INFINITY);
            if (ms.goal_elem_valid) init_tiles[n_init++] = m * P.NTm + (ms.goal_ex / T) * P.TY + (ms.goal_ey / T);
            else ++n_init;   // nothing reachable: field stays +inf
        } else if (ms.new_start) {
            ms.new_start = false;
            consume[m] = 1;
            ++n_upd;
        }
    }
    // (goal array upload is per map to keep untouched maps' goals)
    for (int m = 0; m < nmaps; ++m) {
        MapState &ms = maps[m];
        if (ms.initialize_search || ms.new_goal)
            HIPCHK(hipMemcpyAsync(P.goal + 2 * m, goals + 2 * m, 2 * sizeof(int), hipMemcpyHostToDevice, stream));
    }
    // replan of a single map with a few small pending patches: the control steps run fused
    // (k_replan_begin / k_raise_to_lower / k_replan_end) instead of as ten separate launches
    ReplanBegin rb{};
    bool fused = single && fuse_control && spin_wait && n_init == 0 && n_upd > 0 && !pending.empty() && pending.size() <= 4;
    if (fused)
        for (const PatchRect &r : pending) fused = fused && consume[r.m] && (r.w + 1) * (r.h + 1) <= 65 * 65;
    {   // start elements: the 4 corners of the start cell (FD impl:9-13, Cell.cpp:48-60) / the start cell (DFM)
        int *st_el = h_scratch + 5 * nmaps + 4;
        float *sp = reinterpret_cast<float *>(h_scratch + 9 * nmaps + 8);
        for (int m = 0; m < nmaps; ++m) {
            const MapState &ms = maps[m];
            for (int i = 0; i < 4; ++i) st_el[4 * m + i] = -1;
            sp[2 * m] = sp[2 * m + 1] = 0.0f;
            if (!ms.start_set) continue;
            const int cx = (int)(start_cell_floor ? std::floor(ms.start_x) : std::roundf(ms.start_x)), cy = (int)(start_cell_floor ? std::floor(ms.start_y) : std::roundf(ms.start_y));
            // keys measure from start_pos_ (FD/SG, Position::distance) or from start_cell_ (DFM, Cell::distance)
            sp[2 * m] = (algo == UFM_ALGO_DFM) ? (float)cx : ms.start_x;
            sp[2 * m + 1] = (algo == UFM_ALGO_DFM) ? (float)cy : ms.start_y;
            const int ncorner = (algo == UFM_ALGO_DFM) ? 1 : 4;
            for (int i = 0; i < ncorner; ++i) {
                const int ex = cx + (i & 1), ey = cy + (i >> 1);
                if (ex >= 0 && ey >= 0 && ex < P.EX && ey < P.EY) st_el[4 * m + i] = ex * P.EY + ey;
            }
        }
        if (single) {
            for (int i = 0; i < 4; ++i) rb.sb.start[i] = st_el[i];
            rb.sb.consume = consume[0];
            rb.sb.clear_lmax = profiling ? 1 : 0;
            rb.sb.sx = sp[0]; rb.sb.sy = sp[1];
            if (!fused) k_step_begin<<<1, 256, 0, stream>>>(P, rb.sb);
        } else {
            HIPCHK(hipMemcpyAsync(P.start, st_el, sizeof(int) * 4 * nmaps, hipMemcpyHostToDevice, stream));
            HIPCHK(hipMemcpyAsync(P.spos, sp, sizeof(float) * 2 * nmaps, hipMemcpyHostToDevice, stream));
        }
    }
    uint64_t updated = 0;
    bool have_seeds = false;
    bool fast_done = false;
    bool skip_raise = false;      // the block kernel has left nothing to invalidate below its bound (only lowering work beyond the block)
    // margin of the invalidation bound above the start's current key (the key may rise through the patch)
    const float band = raise_margin * (delta_abs >= 0.0f ? delta_abs : delta_scale * T * mean_cost);
    if (n_upd > 0 || n_init > 0) {
        if (!single) HIPCHK(hipMemcpyAsync(P.consume, consume, sizeof(int) * nmaps, hipMemcpyHostToDevice, stream));
        // consume pending patch rectangles of the participating maps
        std::vector<PatchRect> keep;
        region_rects.clear();
        for (const PatchRect &r : pending) {
            if (!consume[r.m]) { keep.push_back(r); continue; }
            have_seeds = true;
            region_rects.push_back(r);
            const int cnt = (r.h + 1) * (r.w + 1);
            if (fused) { int *q = rb.rect[rb.nrect++]; q[0] = r.m; q[1] = r.x; q[2] = r.y; q[3] = r.w; q[4] = r.h; }
            else k_clear_marks<<<(cnt + 255) / 256, 256, 0, stream>>>(P, r.m, r.x, r.y, r.w, r.h);
        }
        pending.swap(keep);
    }
    const auto t_seed = std::chrono::steady_clock::now();
    if (have_seeds && n_init == 0 && n_upd > 0) {
        // Replan fast path: one submission, one host round trip.  Seeds -> invalidation bound ->
        // a blind batch of invalidation launches -> re-lower what they touched -> a blind batch of
        // lowering launches -> device-side check -> finalise if the check says "done".  (An empty
        // launch costs a few microseconds; a host round trip costs more.)  If the batches were too
        // short the general adaptive loop below takes over.
        // blind batch sizes: what the recent replans needed, plus one
        int nr = 1, nl = 1;
        for (int i = 0; i < 6; ++i) { nr = std::max(nr, win_raise[i] + batch_margin); nl = std::max(nl, win_lower[i] + batch_margin); }
        const int k0_raise = iter[Q_RAISE], k0_lower = iter[Q_LOWER];
        if (profiling) while (ev.size() < 4) { hipEvent_t a; HIPCHK(hipEventCreate(&a)); ev.push_back(a); }
        // The block around the patches (ufm_region.h): its goal-side edge `region_ahead` tiles beyond the patches' centre,
        // the rest of its extent behind it -- where the elements that lean on the patched cells are.
        RegionJobs rjs{};
        bool regioned = false;
        {
            const bool nodes = algo != UFM_ALGO_DFM;
            // the block of one map: around its consumed rectangles; false if they do not fit into one block
            auto place_job = [&](RegionJob &j, const ReplanBegin &b, int m) {
                if (!region_fits(b.rect, b.nrect, m, &j.tx0, &j.ntx, &j.ty0, &j.nty)) return false;
                j.rb = b; j.rb.k_raise = iter[Q_RAISE]; j.rb.band = band;
                j.dyn = dyn_now; j.k_lower = iter[Q_LOWER]; j.max_sweeps = region_sweeps; j.debug = region_debug;
                j.slack = 255.0f * SQRT2F + 1.0f;     // the largest cost of one move (a diagonal through the most expensive cell)
                j.delta = region_band > 0.0f ? region_band * 4.0f * mean_cost : INFINITY;
                j.map = m;
                for (int r = 0; r < 4; ++r) j.psrc[r] = nullptr;
                return true;
            };
            if (fused && use_region) {                       // one map, a few small patches
                regioned = place_job(rjs.j[0], rb, 0);
                rjs.n = 1; rjs.j[0].batch = 0;
                if (lazy_in_kernel) {                        // the held host patches are the last rectangles: the kernel applies them
                    if (!regioned || lazy.size() > (size_t)rb.nrect) return UFM_ERR_INVALID;      // (lazy_region_ok() said otherwise: cannot happen)
                    for (size_t i = 0; i < lazy.size(); ++i) {
                        rjs.j[0].psrc[rb.nrect - (int)lazy.size() + (int)i] = h_lazy + (size_t)lazy[i].slot * 4096;
                    }       // (the kernel has read them when this step returns: the slots are free again then)
                    lazy.clear();
                }
            } else if (!single && use_region && spin_wait && nmaps <= RJOBS && !region_rects.empty()) {
                // a batch: one job per consuming map, every one of them with 1..4 small rectangles of its own
                bool ok = true;
                rjs.n = 0;
                const int *st_el = h_scratch + 5 * nmaps + 4;
                const float *sp = reinterpret_cast<const float *>(h_scratch + 9 * nmaps + 8);
                for (int m = 0; m < nmaps && ok; ++m) {
                    if (!consume[m]) continue;
                    ReplanBegin b{};
                    for (const PatchRect &r : region_rects) {
                        if (r.m != m) continue;
                        if (b.nrect >= 4 || (r.w + 1) * (r.h + 1) > 65 * 65) { ok = false; break; }
                        int *q = b.rect[b.nrect++]; q[0] = r.m; q[1] = r.x; q[2] = r.y; q[3] = r.w; q[4] = r.h;
                    }
                    for (int i = 0; i < 4; ++i) b.sb.start[i] = st_el[4 * m + i];
                    b.sb.consume = 1; b.sb.sx = sp[2 * m]; b.sb.sy = sp[2 * m + 1];
                    RegionJob &j = rjs.j[rjs.n];
                    ok = ok && place_job(j, b, m);
                    j.batch = 1;
                    ++rjs.n;
                }
                regioned = ok && rjs.n > 0;
            }
        }
        const bool graphed = !regioned && fused && use_graph && nr < 250 && nl < 250;
        if (regioned) {
            const unsigned int seq = ++pub_seq;
            for (int i = 0; i < rjs.n; ++i) rjs.j[i].seq = seq;
            if (rjs.j[0].batch) {   // the counters the maps' workgroups add to
                HIPCHK(hipMemsetAsync(&P.ctr->rbound, 0, offsetof(DevCounters, done_fail) + sizeof(int) - offsetof(DevCounters, rbound), stream));
                k_fill<<<1, 64, 0, stream>>>(reinterpret_cast<float *>(&P.ctr->qmin[Q_RAISE]), (size_t)1, INFINITY);
            }
            // the per-step scalars: a single map's workgroup stores the job's copy itself (it is the only reader before the next launch); a batch's
            // workgroups read *P.dyn side by side (start_bound, tile_heuristic), so there it is in place before the launch
            if (rjs.j[0].batch) flush_dyn();
            else { dyn_dev = dyn_now; dyn_pending = false; }
            const dim3 g(rjs.n), b(NTHR);
            const bool reg_timed = profiling && (region_runs & 7u) == 0u;     // a sample: the event packets cost a few microseconds each
            if (reg_timed) for (auto &e : reg_ev) if (!e) HIPCHK(hipEventCreate(&e));
#define UFM_LAUNCH(A) do { if (reg_timed) hipExtLaunchKernelGGL((k_replan_region<A>), g, b, 0, stream, reg_ev[0], reg_ev[1], 0, P, rjs, h_ctr, h_flag); \
                           else k_replan_region<A><<<g, b, 0, stream>>>(P, rjs, h_ctr, h_flag); } while (0)
            if (algo == UFM_ALGO_FD) UFM_LAUNCH(UFM_ALGO_FD);
            else if (algo == UFM_ALGO_SG) UFM_LAUNCH(UFM_ALGO_SG);
                    else UFM_LAUNCH(ALGO_DFM1);
#undef UFM_LAUNCH
            HIPCHK(hipGetLastError());
            last_active = 1;
            int rc = wait_published();
            if (rc != UFM_OK) return rc;
            st.region_launches = 1u;
            for (int i = 0; i < rjs.n; ++i) st.region_tiles += (uint32_t)(rjs.j[i].ntx * rjs.j[i].nty);
            if (reg_timed) {      // (the kernel has published its result: its stop event follows within microseconds -- spin, do not sleep)
                hipError_t q;
                while ((q = hipEventQuery(reg_ev[1])) == hipErrorNotReady) __builtin_ia32_pause();
                HIPCHK(q);
                HIPCHK(hipEventElapsedTime(&st.region_kernel_ms, reg_ev[0], reg_ev[1]));
                st.region_timed = 1u;
            }
            region_runs += (uint32_t)rjs.n;
            if (h_ctr->done) region_done += (uint32_t)rjs.n;
            else if (focused) {
                // (its end check has the smallest invalidation priority of the map -- of any map of a batch --, queued or parked: at or
                //  beyond the bound -- a batch: the largest of the maps' bounds -- means the launch chain's invalidation phase, two batches
                //  of launches and two host round trips, would release nothing)
                float qm;
                std::memcpy(&qm, &h_ctr->qmin[Q_RAISE], sizeof(float));
                skip_raise = !(qm < h_ctr->rbound);
            }
        } else if (graphed) {
            rb.k_raise = iter[Q_RAISE]; rb.band = band;
            hipGraphExec_t ge = nullptr;
            int rc = replan_graph(nr, nl, band, &ge);
            if (rc != UFM_OK) return rc;
            h_job->rb = rb; h_job->k_lower = iter[Q_LOWER]; h_job->seq = ++pub_seq;
            h_job->dyn = dyn_now; dyn_dev = dyn_now; dyn_pending = false;
            __atomic_thread_fence(__ATOMIC_RELEASE);
            HIPCHK(hipGraphLaunch(ge, stream));
            iter[Q_RAISE] += nr; iter[Q_LOWER] += nl;
            last_active = 1;
            rc = wait_published();
            if (rc != UFM_OK) return rc;
        } else {
        flush_dyn();
        if (fused) {
            rb.k_raise = iter[Q_RAISE]; rb.band = band;
            k_replan_begin<<<1, 1024, 0, stream>>>(P, rb);
        } else {
            k_seeds_to_active<<<1, 1024, 0, stream>>>(P, Q_RAISE, iter[Q_RAISE]);
            k_prepare_bound<<<1, 64, 0, stream>>>(P, band);
            k_unpark<<<1, 1024, 0, stream>>>(P, Q_RAISE, iter[Q_RAISE], -1.0f);
        }
        hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr, e3 = nullptr;
        if (profiling) {
            while (ev.size() < 4) { hipEvent_t a; HIPCHK(hipEventCreate(&a)); ev.push_back(a); }
            e0 = ev[0]; e1 = ev[1]; e2 = ev[2]; e3 = ev[3];
            HIPCHK(hipEventRecord(e0, stream));
        }
        last_active = 1;             // replans touch a handful of tiles: fused triage
        for (int i = 0; i < nr; ++i) launch_relax(MODE_RAISE, -1.0f);
        if (profiling) HIPCHK(hipEventRecord(e1, stream));
        if (fused) {
            k_raise_to_lower<<<1, 1024, 0, stream>>>(P, iter[Q_LOWER]);
        } else {
            k_touched_to_active<<<64, 256, 0, stream>>>(P, Q_LOWER, iter[Q_LOWER]);
            k_unpark<<<1, 1024, 0, stream>>>(P, Q_LOWER, iter[Q_LOWER], INFINITY);
        }
        if (profiling) HIPCHK(hipEventRecord(e2, stream));
        for (int i = 0; i < nl; ++i) launch_relax(MODE_LOWER, INFINITY);
        if (profiling) HIPCHK(hipEventRecord(e3, stream));
        if (fused) {
            ++pub_seq;
            k_replan_end<<<64, T * T, 0, stream>>>(P, iter[Q_RAISE], iter[Q_LOWER], band, h_ctr, h_flag, pub_seq);
            finalize_bp(1);
            HIPCHK(hipGetLastError());
            int rc = wait_published();
            if (rc != UFM_OK) return rc;
        } else {
            k_check<<<1, 1024, 0, stream>>>(P, iter[Q_RAISE], iter[Q_LOWER], band);
            finalize_bp(1);
            k_finalize<<<2048, 256, 0, stream>>>(P, 1);
            HIPCHK(hipGetLastError());
            int rc = fetch_counters();
            if (rc != UFM_OK) return rc;
        }
        if (profiling) {
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, e0, e1)); st.kernel_ms += ms; st.raise_kernel_ms += ms;
            HIPCHK(hipEventElapsedTime(&ms, e2, e3)); st.kernel_ms += ms;
        }
        }   // !graphed
        updated += h_ctr->updated;
        fast_done = h_ctr->done != 0;
        if (regioned) {
            st.launches += 1u;
            if (!fast_done && spin_wait && cont_lower > 0) {
                // The block kernel has left work beyond its block (3 of the headline's 100 replans; a round of a batch as soon as ONE of its maps
                // has): the launch chain takes over from the queues -- first as ONE blind submission ending in the device-side end check, like
                // the fused replan path (a few launches that may find nothing to do are cheaper than the adaptive loop's host round trips:
                // that loop cost a batch round of config 4 ~0.6 ms), and only if that was not enough through the adaptive loop below.
                // (one invalidation launch even when the block kernel saw nothing to invalidate below its bound: the end check reads what the
                //  last launch of each phase released)
                const int nr2 = skip_raise ? 1 : std::max(1, cont_raise), nl2 = cont_lower;
                last_active = 1;
                k_unpark<<<1, 1024, 0, stream>>>(P, Q_RAISE, iter[Q_RAISE], -1.0f);       // (-1: the bound the block kernel left in the counters)
                for (int i = 0; i < nr2; ++i) launch_relax(MODE_RAISE, -1.0f);
                k_raise_to_lower<<<1, 1024, 0, stream>>>(P, iter[Q_LOWER]);
                for (int i = 0; i < nl2; ++i) launch_relax(MODE_LOWER, INFINITY);
                ++pub_seq;
                k_replan_end<<<64, T * T, 0, stream>>>(P, iter[Q_RAISE], iter[Q_LOWER], band, h_ctr, h_flag, pub_seq);
                finalize_bp(1);
                HIPCHK(hipGetLastError());
                int rc = wait_published();
                if (rc != UFM_OK) return rc;
                fast_done = h_ctr->done != 0;
                st.raise_launches += (uint32_t)nr2;
                st.launches += (uint32_t)(nr2 + nl2);
                skip_raise = false;
                ++region_cont; if (fast_done) ++region_cont_done;
            }
        } else {
        st.raise_launches += (uint32_t)nr;
        st.launches += (uint32_t)(nr + nl);
        // (launches replayed from the graph are not event-timed: HIP cannot read events recorded by graph nodes)
        if (profiling && !graphed) { st.timed_launches += (uint32_t)(nr + nl); st.timed_raise_launches += (uint32_t)nr; }
        // launches the batches actually needed (for the next steps' batch sizes); a batch that was
        // too short costs a host round trip and the adaptive loop, so err on the long side after one
        {
            const int need_r = std::max(0, h_ctr->last_work[Q_RAISE] - k0_raise + 1);
            const int need_l = std::max(0, h_ctr->last_work[Q_LOWER] - k0_lower + 1);
            win_raise[win_pos] = fast_done ? need_r : nr + 2;
            win_lower[win_pos] = fast_done ? need_l : nl + 2;
            win_pos = (win_pos + 1) % 6;
        }
        }
    } else if (have_seeds) {
        flush_dyn();
        // num_nodes_updated (FD impl:138, DFM impl:109) of the participating maps
        HIPCHK(hipMemcpyAsync(h_scratch + 2 * nmaps + 2 * nmaps, P.num_updated, sizeof(unsigned int) * nmaps, hipMemcpyDeviceToHost, stream));
        // patches enter an existing field through the invalidation queue, a fresh one directly
        const int sq = (n_upd > 0) ? Q_RAISE : Q_LOWER;
        k_seeds_to_active<<<1, 1024, 0, stream>>>(P, sq, iter[sq]);
        HIPCHK(hipStreamSynchronize(stream));
        const unsigned int *nu = reinterpret_cast<const unsigned int *>(h_scratch + 4 * nmaps);
        for (int m = 0; m < nmaps; ++m) {
            MapState &ms = maps[m];
            if (!consume[m]) continue;
            if (!(ms.initialize_search || ms.new_goal)) updated += nu[m];
            HIPCHK(hipMemsetAsync(P.num_updated + m, 0, sizeof(unsigned int), stream));
        }
    }
    if (n_init > 0) {
        int k = 0;
        for (int m = 0; m < nmaps; ++m) {
            MapState &ms = maps[m];
            if ((ms.initialize_search || ms.new_goal) && ms.goal_elem_valid) ++k;
        }
        if (k > 0) {
            HIPCHK(hipMemcpyAsync(d_scratch, init_tiles, sizeof(int) * k, hipMemcpyHostToDevice, stream));
            k_activate_list<<<1, 64, 0, stream>>>(P, Q_LOWER, iter[Q_LOWER], d_scratch, k);
        }
    }
    // ReplannerBase.h:65-69: plan() only if something was (re)initialised or updated
    const bool do_plan = (n_init > 0 || updated > 0 || (have_seeds && n_upd > 0)) && !fast_done;
    const bool do_raise = have_seeds && n_upd > 0;
    auto t1 = std::chrono::steady_clock::now();
    double u_acc = std::chrono::duration<double, std::milli>(t1 - t0).count(), p_acc = 0.0;
    if (do_plan) {
        flush_dyn();
        // Invalidate, then lower, both only as far as the start's key (the reference's
        // end_condition).  The invalidation bound must reach the key the start ends up with, which
        // is only known afterwards: start from the current key plus one ordering band and repeat
        // while invalidations below the new key are still queued.
        float rbound = INFINITY;
        if (focused && do_raise) {
            if (h_ctr->rbound > 0.0f && n_init == 0 && n_upd > 0) {
                rbound = h_ctr->rbound;      // continue from the fast path's (possibly enlarged) bound
            } else {
                float b0 = 0.0f;
                int rc = read_bounds(&b0);
                if (rc != UFM_OK) return rc;
                rbound = b0 + band;
            }
        }
        for (int round = 0; round < 64; ++round) {
            const auto ta = std::chrono::steady_clock::now();
            if (do_raise && !(skip_raise && round == 0)) {
                uint32_t rl = 0;
                float rk = 0.0f;
                k_unpark<<<1, 1024, 0, stream>>>(P, Q_RAISE, iter[Q_RAISE], rbound);
                uint32_t rt = 0;
                int rc = run_phase(MODE_RAISE, rbound, &rl, &rk, &rt);
                if (rc != UFM_OK) return rc;
                st.kernel_ms += rk; st.raise_kernel_ms += rk;
                st.timed_launches += rt; st.timed_raise_launches += rt;
                st.raise_launches += rl;
                st.launches += rl;
                // everything invalidation touched must be re-lowered
                k_touched_to_active<<<64, 256, 0, stream>>>(P, Q_LOWER, iter[Q_LOWER]);
            }
            const auto tb = std::chrono::steady_clock::now();
            uint32_t ll = 0;
            k_unpark<<<1, 1024, 0, stream>>>(P, Q_LOWER, iter[Q_LOWER], INFINITY);
            int owned_left = -1;
            if (use_owned && n_init > 0 && round == 0 && dyn_grid >= 256) {
                int rc = owned_phase();
                if (rc != UFM_OK) return rc;
                st.launches += 1u;
                st.resident_launches += 1u;
                // what it handed back (nothing, unless it ran into its time limit): no need to send launches after an empty list
                rc = fetch_counters();
                if (rc != UFM_OK) return rc;
                owned_left = h_ctr->cnt[Q_LOWER][iter[Q_LOWER] % 3];
            }
            int rc = owned_left == 0 ? UFM_OK : run_phase(MODE_LOWER, INFINITY, &ll, &st.kernel_ms, &st.timed_launches);
            if (rc != UFM_OK) return rc;
            st.launches += ll;
            bool again = false;
            if (focused && do_raise) {
                float bnew = 0.0f;
                rc = read_bounds(&bnew);
                if (rc != UFM_OK) return rc;
                k_queue_min<<<1, 1024, 0, stream>>>(P, Q_RAISE, iter[Q_RAISE]);
                rc = fetch_counters();
                if (rc != UFM_OK) return rc;
                float qm;
                std::memcpy(&qm, &h_ctr->qmin[Q_RAISE], sizeof(float));
                if (qm < bnew) { again = true; rbound = std::fmax(bnew, rbound) + band; }
            }
            const auto tc = std::chrono::steady_clock::now();
            u_acc += std::chrono::duration<double, std::milli>(tb - ta).count();
            p_acc += std::chrono::duration<double, std::milli>(tc - tb).count();
            if (!again) break;
        }
        const auto td = std::chrono::steady_clock::now();
        finalize_bp(0);
        k_finalize<<<2048, 256, 0, stream>>>(P, 0);
        { int rc = fetch_counters(); if (rc != UFM_OK) return rc; }
        st.expanded = h_ctr->expanded;
        st.tile_visits = h_ctr->tile_visits;
        st.tile_iters = h_ctr->tile_iters;
        st.elem_evals = h_ctr->elem_evals;
        st.raise_tile_visits = h_ctr->raise_visits;
        if (st.resident_launches) {
            st.resident_tile_visits = h_ctr->own_vis1 - h_ctr->own_vis0;
            st.resident_stops = (uint32_t)h_ctr->own_stops;
            if (own_timed) HIPCHK(hipEventElapsedTime(&st.resident_kernel_ms, own_ev[0], own_ev[1]));
        }
        if (profiling) {   // diagnostics: sum over launches of the slowest tile's sweep count
            std::vector<int> lm(LMAX);
            HIPCHK(hipMemcpy(lm.data(), P.lmax, sizeof(int) * LMAX, hipMemcpyDeviceToHost));
            for (int v : lm) st.crit_sweeps += (uint64_t)v;
        }
        p_acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - td).count();
    } else if (fast_done) {
        st.expanded = h_ctr->expanded;
        st.tile_visits = h_ctr->tile_visits;
        st.tile_iters = h_ctr->tile_iters;
        st.elem_evals = h_ctr->elem_evals;
        st.raise_tile_visits = h_ctr->raise_visits;
        const double dt = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_seed).count();
        u_acc += 0.5 * dt;   // invalidation and lowering ran in one submission: split evenly
        p_acc += 0.5 * dt;
    } else {
        HIPCHK(hipStreamSynchronize(stream));
    }
    for (int m = 0; m < nmaps; ++m) maps[m].new_goal = maps[m].initialize_search = false;
    st.updated = updated;
    st.queued_lower = (uint32_t)(h_ctr->cnt[Q_LOWER][iter[Q_LOWER] % 3] + h_ctr->npark[Q_LOWER]);   // parked beyond the start's key
    st.queued_raise = (uint32_t)(h_ctr->cnt[Q_RAISE][iter[Q_RAISE] % 3] + h_ctr->npark[Q_RAISE]);
    st.graphs_instantiated = graphs_made;
    st.region_replans = region_runs; st.region_replans_done = region_done;
    st.u_ms = (float)u_acc;   // seeding + invalidation (the reference's update())
    st.p_ms = (float)p_acc;   // propagation + finalisation (the reference's plan())
    last = st;
    if (out) *out = st;
    return UFM_OK;
}

int engine_create(Engine **out, int n_maps, int algo, int opt_lvl, int use_heuristic, int device_id) {
    if (!out || n_maps < 1 || algo < 0 || algo > 2 || opt_lvl < 0 || opt_lvl > 2) return UFM_ERR_INVALID;
    if (algo != UFM_ALGO_SG && opt_lvl > 1) return UFM_ERR_INVALID;   // only ShiftedGridPlanner has level 2
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) return UFM_ERR_INVALID;
    HIPCHK(hipSetDevice(device_id));
    Engine *e = new (std::nothrow) Engine();
    if (!e) return UFM_ERR_NOMEM;
    e->algo = algo; e->opt_lvl = opt_lvl; e->heur = use_heuristic; e->device = device_id; e->nmaps = n_maps;
    // scheduling defaults per planner family (tools/sweep.py, 4096^2): DFM's two-stencil operator needs about
    // twice the sweeps per tile; a wider band and an earlier re-queue suit it better (plan 60 -> 52 ms)
    // -- for a single map; a batch is throughput-bound and keeps the less redundant setting (8 x 2048^2: 484 vs 476 M cells/s)
    if (algo == UFM_ALGO_DFM && n_maps == 1) { e->delta_scale = e->delta_scale_long = 2.5f; e->max_iters = 16; }
    if (algo == UFM_ALGO_DFM) { e->cont_lower = 0; e->region_tiles = 8; e->region_ahead = 3; }   // (config 4: the 8 x 8 block finishes 78 % of the rounds alone, 6 x 6: 71 %)
    e->maps.resize(n_maps);
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device_id));
    e->grid_relax = prop.multiProcessorCount * 2;
    e->dyn_grid = prop.multiProcessorCount;
    HIPCHK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    {   // counters + sequence flag in one host-coherent, device-mapped allocation
        const size_t flag_off = (sizeof(DevCounters) + 63) / 64 * 64;
        void *pub = nullptr;
        HIPCHK(hipHostMalloc(&pub, flag_off + 64, hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(pub, 0, flag_off + 64);
        e->h_ctr = static_cast<DevCounters *>(pub);
        e->h_flag = reinterpret_cast<unsigned int *>(static_cast<char *>(pub) + flag_off);
        for (int i = 0; i < 2; ++i) {
            void *pp = nullptr;
            HIPCHK(hipHostMalloc(&pp, flag_off + 64, hipHostMallocMapped | hipHostMallocCoherent));
            std::memset(pp, 0, flag_off + 64);
            e->h_pipe_ctr[i] = static_cast<DevCounters *>(pp);
            e->h_pipe_flag[i] = reinterpret_cast<unsigned int *>(static_cast<char *>(pp) + flag_off);
        }
        void *job = nullptr;
        HIPCHK(hipHostMalloc(&job, sizeof(ReplanJob), hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(job, 0, sizeof(ReplanJob));
        e->h_job = static_cast<ReplanJob *>(job);
    }
    {   // The first graph a process captures and instantiates costs ~8 ms of one-time set-up inside the
        // runtime; pay it here, not in the first replan (a planner is created outside any timed step).
        hipGraph_t g = nullptr;
        hipGraphExec_t ge = nullptr;
        if (hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            k_fill<<<1, 64, 0, e->stream>>>(reinterpret_cast<float *>(e->h_job), 0, 0.0f);
            if (hipStreamEndCapture(e->stream, &g) == hipSuccess && g) {
                if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) == hipSuccess && ge) {
                    (void)hipGraphLaunch(ge, e->stream);
                    (void)hipStreamSynchronize(e->stream);
                    (void)hipGraphExecDestroy(ge);
                }
                (void)hipGraphDestroy(g);
            }
        }
        (void)hipGetLastError();
    }
    HIPCHK(hipHostMalloc(&e->h_scratch, sizeof(int) * (11 * n_maps + 16)));
    HIPCHK(hipHostMalloc(&e->h_bnd, sizeof(float) * n_maps));
    *out = e;
    return UFM_OK;
}

int engine_destroy(Engine *e) {
    if (!e) return UFM_ERR_INVALID;
    hipSetDevice(e->device);
    if (e->stream) hipStreamSynchronize(e->stream);
    e->release();
    for (hipEvent_t v : e->ev) hipEventDestroy(v);
    for (hipEvent_t v : e->own_ev) if (v) hipEventDestroy(v);
    for (hipEvent_t v : e->reg_ev) if (v) hipEventDestroy(v);
    if (e->d_patch) hipFree(e->d_patch);
    if (e->d_pmask) hipFree(e->d_pmask);
    if (e->d_field) hipFree(e->d_field);
    if (e->d_info) hipFree(e->d_info);
    if (e->d_jobs) hipFree(e->d_jobs);
    if (e->h_jobs) hipHostFree(e->h_jobs);
    if (e->d_path) hipFree(e->d_path);
    if (e->h_path) hipHostFree(e->h_path);
    if (e->h_patch) hipHostFree(e->h_patch);
    if (e->h_lazy) hipHostFree(e->h_lazy);
    e->drop_graphs();
    if (e->h_ctr) hipHostFree(e->h_ctr);
    for (int i = 0; i < 2; ++i) if (e->h_pipe_ctr[i]) hipHostFree(e->h_pipe_ctr[i]);
    if (e->h_job) hipHostFree(e->h_job);
    if (e->h_scratch) hipHostFree(e->h_scratch);
    if (e->h_bnd) hipHostFree(e->h_bnd);
    if (e->stream) hipStreamDestroy(e->stream);
    delete e;
    return UFM_OK;
}

int engine_set_map(Engine *e, int m, const uint8_t *src, bool on_device, int width, int length) {
    if (!e || !src || m < 0 || m >= e->nmaps || width <= 0 || length <= 0) return UFM_ERR_INVALID;
    HIPCHK(hipSetDevice(e->device));
    if (e->allocated) { int rc = e->flush_deferred(); if (rc != UFM_OK) return rc; }   // (patches held back belong before the new raster)
    if (!e->allocated || width != e->W || length != e->L) {
        bool others = false;
        for (int k = 0; k < e->nmaps; ++k) if (k != m && e->maps[k].have_map) others = true;
        if (e->allocated && others) return UFM_ERR_INVALID;   // all maps of a batch share one size
        int rc = e->alloc(width, length);
        if (rc != UFM_OK) return rc;
    }
    HIPCHK(hipMemcpyAsync(e->P.cost + (size_t)m * e->P.cstride, src, (size_t)width * length,
                          on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, e->stream));
    k_cost_windows<<<2048, 256, 0, e->stream>>>(e->P, m);
    {   // mean traversable cost -> default ordering band
        unsigned long long *d_acc = reinterpret_cast<unsigned long long *>(e->d_scratch);
        HIPCHK(hipMemsetAsync(d_acc, 0, 2 * sizeof(unsigned long long), e->stream));
        k_cost_stats<<<512, 256, 0, e->stream>>>(e->P.cost + (size_t)m * e->P.cstride, (size_t)width * length, e->thr_uchar, d_acc);
        unsigned long long *h_acc = reinterpret_cast<unsigned long long *>(e->h_scratch);
        HIPCHK(hipMemcpyAsync(h_acc, d_acc, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        if (h_acc[1] > 0) e->mean_cost = (float)((double)h_acc[0] / (double)h_acc[1]);
    }
    e->maps[m].have_map = true;   // initialize_graph = false, ReplannerBase.h:87
    // goal element validity depends on the map size
    MapState &ms = e->maps[m];
    if (ms.goal_set) ms.goal_elem_valid = ms.goal_ex >= 0 && ms.goal_ey >= 0 && ms.goal_ex < e->P.EX && ms.goal_ey < e->P.EY;
    return UFM_OK;
}

int engine_patch(Engine *e, int m, const uint8_t *src, bool on_device, int x, int y, int w, int h) {
    if (!e || !src) return UFM_ERR_INVALID;
    HIPCHK(hipSetDevice(e->device));
    if (on_device) return e->patch(m, src, x, y, w, h, true);
    if (w <= 0 || h <= 0) return UFM_ERR_INVALID;
    {   // a small patch of a single planner: held in pinned memory for the replan's block kernel (Engine::patch_lazy)
        bool taken = false;
        const int rc = e->patch_lazy(m, src, x, y, w, h, &taken);
        if (rc != UFM_OK || taken) return rc;
    }
    const size_t n = (size_t)w * h;
    if (n > e->d_patch_cap) {
        if (e->d_patch || e->h_patch) HIPCHK(hipStreamSynchronize(e->stream));
        if (e->d_patch) hipFree(e->d_patch);
        if (e->h_patch) hipHostFree(e->h_patch);
        e->d_patch = nullptr; e->h_patch = nullptr; e->d_patch_cap = 0;   // nothing dangling if an allocation below fails
        const size_t cap = n < 4096 ? 4096 : n;
        HIPCHK(hipMalloc(&e->d_patch, cap));
        HIPCHK(hipHostMalloc(&e->h_patch, cap));
        e->d_patch_cap = cap;
    } else {
        HIPCHK(hipStreamSynchronize(e->stream));   // staging buffers are reused
    }
    std::memcpy(e->h_patch, src, n);
    HIPCHK(hipMemcpyAsync(e->d_patch, e->h_patch, n, hipMemcpyHostToDevice, e->stream));
    return e->patch(m, e->d_patch, x, y, w, h);
}

int engine_set_goal(Engine *e, int m, float x, float y) {
    if (!e || m < 0 || m >= e->nmaps) return UFM_ERR_INVALID;
    MapState &ms = e->maps[m];
    // Node(Position)/Cell(Position) round (Node.cpp:14-17, Cell.cpp:20-21); ReplannerBase.h:99-108
    const int ex = (int)std::roundf(x), ey = (int)std::roundf(y);
    ms.new_goal = !ms.goal_set ? true : (ex != ms.goal_ex || ey != ms.goal_ey);
    ms.goal_x = x; ms.goal_y = y; ms.goal_ex = ex; ms.goal_ey = ey;
    ms.goal_set = true;
    ms.goal_elem_valid = e->allocated && ex >= 0 && ey >= 0 && ex < e->P.EX && ey < e->P.EY;
    return UFM_OK;
}

int engine_read_field(Engine *e, int m, int x0, int y0, int nx, int ny, float *g, float *rhs) {
    if (!e || m < 0 || m >= e->nmaps || !e->allocated) return UFM_ERR_INVALID;
    if (x0 < 0 || y0 < 0 || nx <= 0 || ny <= 0 || x0 + nx > e->P.EX || y0 + ny > e->P.EY) return UFM_ERR_INVALID;
    HIPCHK(hipSetDevice(e->device));
    float *dst = g ? g : rhs;
    if (!dst) return UFM_OK;
    // the field is tile-major on the device: gather the window into a dense buffer, then one copy
    const size_t n = (size_t)nx * ny;
    if (n > e->d_field_cap) {
        if (e->d_field) { HIPCHK(hipStreamSynchronize(e->stream)); hipFree(e->d_field); e->d_field = nullptr; e->d_field_cap = 0; }
        HIPCHK(hipMalloc(&e->d_field, n * sizeof(float)));
        e->d_field_cap = n;
    }
    k_gather_field<<<(unsigned)std::min<size_t>((n + 255) / 256, 65535), 256, 0, e->stream>>>(e->P, m, x0, y0, nx, ny, e->d_field);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(dst, e->d_field, n * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    // at the fixed point RHS(s) = F(G)(s) = G(s) for every element (goal: 0 = 0)
    if (g && rhs) std::memcpy(rhs, g, (size_t)nx * ny * sizeof(float));
    return UFM_OK;
}

// Path extraction for all maps of the engine in one launch (one wavefront per map).
// path_xy: [nmaps][cap_pts][2], step_costs: [nmaps][cap_costs], info: [nmaps].
int engine_extract_path(Engine *e, int max_steps, int lookahead, int allow_indirect,
                        float *path_xy, int cap_pts, float *step_costs, int cap_costs, ufm_path_info *info) {
    if (!e || !e->allocated || !info || max_steps < 1 || cap_pts < 0 || cap_costs < 0) return UFM_ERR_INVALID;
    if ((cap_pts > 0 && !path_xy) || (cap_costs > 0 && !step_costs)) return UFM_ERR_INVALID;
    for (const MapState &ms : e->maps) if (!ms.have_map || !ms.start_set || !ms.goal_set) return UFM_ERR_INVALID;
    HIPCHK(hipSetDevice(e->device));
    { int rc = e->flush_deferred(); if (rc != UFM_OK) return rc; }   // (the walk reads the raster)
    const auto t0 = std::chrono::steady_clock::now();
    const int n = e->nmaps;
    // the device keeps what the caller has room for, at most what max_steps moves can produce
    const int dev_pts = std::min(cap_pts, 3 * max_steps + 1), dev_cst = std::min(cap_costs, 2 * max_steps);
    const size_t ostride = PATH_HDR + 2 * (size_t)dev_pts + dev_cst;
    if (ostride * n > e->path_cap) {
        HIPCHK(hipStreamSynchronize(e->stream));
        if (e->d_path) hipFree(e->d_path);
        if (e->h_path) hipHostFree(e->h_path);
        e->d_path = nullptr; e->h_path = nullptr; e->path_cap = 0;
        HIPCHK(hipMalloc(&e->d_path, ostride * n * sizeof(float)));
        HIPCHK(hipHostMalloc(&e->h_path, ostride * n * sizeof(float)));
        e->path_cap = ostride * n;
    }
    if (!e->d_jobs) {
        HIPCHK(hipMalloc(&e->d_jobs, sizeof(PathJob) * n));
        HIPCHK(hipHostMalloc(&e->h_jobs, sizeof(PathJob) * n));
    }
    for (int m = 0; m < n; ++m) e->h_jobs[m] = PathJob{e->maps[m].start_x, e->maps[m].start_y, e->maps[m].goal_x, e->maps[m].goal_y};
    HIPCHK(hipMemcpyAsync(e->d_jobs, e->h_jobs, sizeof(PathJob) * n, hipMemcpyHostToDevice, e->stream));
    PathField F{};
    F.G = e->P.G; F.cost = e->P.cost;
    F.EX = e->P.EX; F.EY = e->P.EY; F.L = e->P.L; F.W = e->P.W; F.TY = e->P.TY; F.thr = e->thr_uchar;
    F.cells = (e->algo == UFM_ALGO_DFM); F.indirect = allow_indirect != 0;
    k_extract_path<<<n, 64, 0, e->stream>>>(F, e->P.gstride, e->P.cstride, e->d_jobs, e->d_path, ostride,
                                            dev_pts, dev_cst, lookahead != 0, max_steps);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(e->h_path, e->d_path, ostride * n * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int m = 0; m < n; ++m) {
        const float *o = e->h_path + ostride * m;
        ufm_path_info &pi = info[m];
        std::memcpy(&pi.n_points, &o[0], 4);
        std::memcpy(&pi.n_costs, &o[1], 4);
        pi.total_cost = o[2];
        pi.total_dist = o[3];
        std::memcpy(&pi.steps, &o[4], 4);
        const int np = std::min(pi.n_points, dev_pts), nc = std::min(pi.n_costs, dev_cst);
        if (np > 0) std::memcpy(path_xy + (size_t)m * cap_pts * 2, o + PATH_HDR, sizeof(float) * 2 * np);
        if (nc > 0) std::memcpy(step_costs + (size_t)m * cap_costs, o + PATH_HDR + 2 * (size_t)dev_pts, sizeof(float) * nc);
    }
    const float ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (int m = 0; m < n; ++m) info[m].e_ms = ms;
    return UFM_OK;
}

// The elements the reference would hold in its priority queue after the step: G != RHS (k_queue_scan, ufm_path.h).
int engine_read_queue(Engine *e, int m, int cap, int32_t *xy, float *g_rhs, int *total) {
    if (!e || m < 0 || m >= e->nmaps || !e->allocated || !total || cap < 0 || (cap > 0 && (!xy || !g_rhs))) return UFM_ERR_INVALID;
    if (!e->maps[m].goal_set) return UFM_ERR_INVALID;
    HIPCHK(hipSetDevice(e->device));
    { int rc = e->flush_deferred(); if (rc != UFM_OK) return rc; }   // (the derived RHS / the stored bytes' view read the raster)
    const size_t words = (size_t)std::max(cap, 1) * 4 + 1;          // [cap][2] int32, [cap][2] float, the count
    if (words > e->d_info_cap) {
        if (e->d_info) { HIPCHK(hipStreamSynchronize(e->stream)); hipFree(e->d_info); e->d_info = nullptr; e->d_info_cap = 0; }
        HIPCHK(hipMalloc(&e->d_info, words * sizeof(int32_t)));
        e->d_info_cap = words;
    }
    int32_t *d_xy = e->d_info;
    float *d_gr = reinterpret_cast<float *>(e->d_info + (size_t)std::max(cap, 1) * 2);
    unsigned int *d_cnt = reinterpret_cast<unsigned int *>(e->d_info + (size_t)std::max(cap, 1) * 4);
    PathField F{};
    F.G = e->P.G + (size_t)m * e->P.gstride; F.cost = e->P.cost + (size_t)m * e->P.cstride;
    F.EX = e->P.EX; F.EY = e->P.EY; F.L = e->P.L; F.W = e->P.W; F.TY = e->P.TY; F.thr = e->thr_uchar;
    F.cells = (e->algo == UFM_ALGO_DFM); F.indirect = (e->algo == UFM_ALGO_FD);
    const MapState &ms = e->maps[m];
    HIPCHK(hipMemsetAsync(d_cnt, 0, sizeof(unsigned int), e->stream));
    const size_t n = (size_t)e->P.EX * e->P.EY;
    k_queue_scan<<<(unsigned)std::min<size_t>((n + 255) / 256, 4096), 256, 0, e->stream>>>(F, e->opt_lvl, ms.goal_ex, ms.goal_ey, cap, d_xy, d_gr, d_cnt);
    hipError_t err = hipGetLastError();
    unsigned int cnt = 0;
    if (err == hipSuccess) err = hipMemcpyAsync(&cnt, d_cnt, sizeof(cnt), hipMemcpyDeviceToHost, e->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
    HIPCHK(err);
    *total = (int)cnt;
    const size_t k = std::min<size_t>(cnt, (size_t)cap);
    if (k) {
        HIPCHK(hipMemcpyAsync(xy, d_xy, k * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipMemcpyAsync(g_rhs, d_gr, k * 2 * sizeof(float), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
    }
    return UFM_OK;
}

// Back-pointers of a window of elements: the stored codes in the reference's format (k_info_stored), or derived from the field alone
// (k_info, the checker), ufm_path.h.
int engine_read_info(Engine *e, int m, int x0, int y0, int nx, int ny, int32_t *info, bool derived) {
    if (!e || m < 0 || m >= e->nmaps || !e->allocated || !info) return UFM_ERR_INVALID;
    if (e->opt_lvl == 0) return UFM_ERR_INVALID;            // level 0: the map has no Info member (void)
    if (x0 < 0 || y0 < 0 || nx <= 0 || ny <= 0 || x0 + nx > e->P.EX || y0 + ny > e->P.EY) return UFM_ERR_INVALID;
    HIPCHK(hipSetDevice(e->device));
    { int rc = e->flush_deferred(); if (rc != UFM_OK) return rc; }   // (the derived RHS / the stored bytes' view read the raster)
    const size_t n = (size_t)nx * ny;
    if (n * 2 > e->d_info_cap) {        // device buffer kept between calls (a consumer asks window after window)
        if (e->d_info) { HIPCHK(hipStreamSynchronize(e->stream)); hipFree(e->d_info); e->d_info = nullptr; e->d_info_cap = 0; }
        HIPCHK(hipMalloc(&e->d_info, n * 2 * sizeof(int32_t)));
        e->d_info_cap = n * 2;
    }
    int32_t *d_out = e->d_info;
    PathField F{};
    F.G = e->P.G + (size_t)m * e->P.gstride; F.cost = e->P.cost + (size_t)m * e->P.cstride;
    F.EX = e->P.EX; F.EY = e->P.EY; F.L = e->P.L; F.W = e->P.W; F.TY = e->P.TY; F.thr = e->thr_uchar;
    F.cells = (e->algo == UFM_ALGO_DFM); F.indirect = (e->algo == UFM_ALGO_FD);   // FD: all five cost cases; SG: B / II / A
    if (derived) k_info<<<(unsigned)((n + 255) / 256), 256, 0, e->stream>>>(F, e->opt_lvl, x0, y0, nx, ny, d_out);
    else k_info_stored<<<(unsigned)((n + 255) / 256), 256, 0, e->stream>>>(F, e->P.bp + (size_t)m * e->P.gstride, x0, y0, nx, ny, d_out);
    hipError_t err = hipGetLastError();
    if (err == hipSuccess) err = hipMemcpyAsync(info, d_out, n * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
    HIPCHK(err);
    return UFM_OK;
}

