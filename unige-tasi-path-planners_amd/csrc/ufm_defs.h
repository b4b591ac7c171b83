// ufm_defs.h -- compile-time knobs, the HBM layout (DevParams), the tile queues of the launch chain and of the resident kernel, diagnostics
// (a piece of ufm_engine.hip, the engine's one translation unit: included there, inside its anonymous namespace)
#pragma once


#ifndef UFM_TILE
#define UFM_TILE 16
#endif
constexpr int T = UFM_TILE;    // tile edge (elements): 32 or 16
static_assert(T == 32 || T == 16, "tile edge must be 32 or 16");
constexpr int GP = T + 8;      // LDS pitch of the G tile: rows 4 apart land on distinct banks
constexpr int CP = T + 2;      // LDS pitch of the cost tile
constexpr int PT = T / 4;      // 4x4-node patches per tile side
constexpr int PR = PT / 4;     // patches per wave per side (the 16 waves form a 4x4 grid)
constexpr int PPW = PR * PR;   // patches per wave: 4 (T = 32) or 1 (T = 16)
constexpr int NTHR = 1024;     // 16 waves; four lanes per node, each wave owns PPW 4x4-node patches
#ifndef UFM_RELAX_WAVES
#define UFM_RELAX_WAVES 4      // waves per SIMD the relax kernel is compiled for (4: one 1024-thread workgroup per CU)
#endif
#ifndef UFM_CAUSAL_FILTER
#define UFM_CAUSAL_FILTER 1    // do not wake a neighbour tile that a changed border value cannot influence
#endif
#ifndef UFM_STEP_FILTER
#define UFM_STEP_FILTER 1        // do not wake a neighbour tile whose border is less than one step above this tile's (see k_relax write-back)
#endif
// MS-DFM (operator ALGO_DFM1 below) converges without cut-offs on almost every map; these are its livelock guard only: block
// Gauss-Seidel between two tiles can cycle through a finite set of last-bit states (2048^2, seed 1006): after LAX visits of a tile
// in one step a 1-ulp rise is left alone, after QUIET visits a change of <= 4 ulp no longer wakes the neighbours
#ifndef UFM_DFM1_LAX_VISITS
#define UFM_DFM1_LAX_VISITS 64
#endif
#ifndef UFM_DFM1_QUIET_VISITS
#define UFM_DFM1_QUIET_VISITS 96
#endif
#ifndef UFM_DPP_MIN_ASM
#define UFM_DPP_MIN_ASM 1
#endif
#ifndef UFM_STATIC_FIRST
#define UFM_STATIC_FIRST 1     // cursor hand-out: first tile of a workgroup by index, the rest through the cursor
#endif
#ifndef UFM_LPT
#define UFM_LPT 1              // longest-expected-first hand-out of the ready list
#endif
#ifndef UFM_LONG_SWEEPS
#define UFM_LONG_SWEEPS 8      // a visit that took at least this many sweeps per wave counts as long
#endif
#ifndef UFM_LDS_FENCE
#define UFM_LDS_FENCE 1        // 1: workgroup-scope release fence between a sweep's value write and its wake bits
                               // (0: compiler-only ordering, relying on the LDS executing one wave's DS instructions in
                               //  issue order -- all tests pass and nothing measurable is gained, so the fence stays)
#endif
#if UFM_LDS_FENCE
#define UFM_SWEEP_FENCE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup")
#else
#define UFM_SWEEP_FENCE() asm volatile("" ::: "memory")
#endif
#ifndef UFM_EARLY_HANDOFF
#define UFM_EARLY_HANDOFF 1    // resident kernel, FD / SG: a border patch that has gone quiet writes its lowered border values out and
#endif                         // queues the neighbours at once, while the rest of the tile is still being swept (k_relax)
#ifndef UFM_HINT_SAMPLE
#define UFM_HINT_SAMPLE 64     // hints (other owners' smallest priorities) a visit loads ahead for the choice of the next tile
#endif
#ifndef UFM_LEAN_LOOKS
#define UFM_LEAN_LOOKS 1       // looks of an idle workgroup: a sample of the hints, other owners' words only where a hint lies inside the band, no lock scan
#endif
#ifndef UFM_LOOK_HINTS
#define UFM_LOOK_HINTS 128     // hints an idle workgroup's look loads (UFM_LEAN_LOOKS; workgroup 0 loads all)
#endif
#ifndef UFM_LOOK_SLEEP
#define UFM_LOOK_SLEEP (UFM_LEAN_LOOKS ? 127 : 32)   // pause of a workgroup that found nothing to visit before it looks again (x 64 clocks; measured r2: 32 / 64 / 127 -> 15.0 / 14.9 / 14.9 ms)
#endif
#ifndef UFM_STEAL_VICTIMS
#ifndef UFM_FOLLOW
#define UFM_FOLLOW 3             // resident kernel: a workgroup with nothing of its own to go on with takes the neighbour it has just queued with the smallest priority (k_relax); the
                                 // number of neighbours it tries (the best one is often in a visit already).  FD 4096^2 plan kernel 13.84 (0) / 13.6 (1) / 13.43 (2) / 13.36 (3) / 13.38 ms (8)
#endif
#define UFM_STEAL_VICTIMS 4    // owners whose words an idle workgroup looks at per look (k_relax, own_steal)
#endif
#ifndef UFM_EARLY_POLLS
#define UFM_EARLY_POLLS 2      // ... its queue words follow after this many looks of the idle wave at its wake bits (round 4, looks 128 clocks apart: 1 / 2 / 6 / 10 / 16
                               // looks -> SG 2048^2 plan kernel 5.00 / 5.00 / 5.15 / 5.21 / 5.32 ms; the wave waits for its stores first in any case)
#endif
// Issue priority (s_setprio): waves that sweep, stage or write back run above waves that only look -- an idle wave of a visit at its wake bits,
// a workgroup without a tile at the queue words -- so that on a SIMD the looks do not take issue slots from the wave on the chain.  In the 8-wave form
// (two visits per CU; FD 4096^2 plan kernel 14.0 -> 13.6 ms, MS-DFM 2048^2 11.9 -> 11.7); with 16 waves per visit, whose idle waves also fetch the tile's
// new inputs during the visit (in-visit refresh), it costs visits: SG 2048^2 51 k -> 55 k, 5.25 -> 5.4 ms -- not there.
#ifndef UFM_VISIT_PRIO
#define UFM_VISIT_PRIO 3
#endif
#if UFM_VISIT_PRIO
#define UFM_SETPRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define UFM_SETPRIO(x) do {} while (0)
#endif
#ifndef UFM_IDLE_SLEEP
#define UFM_IDLE_SLEEP 2      // an idle wave of a tile visit looks at its wake bits this often (x 64 clocks).  Round 4, with the issue priorities in place: 1 / 2 / 4 / 8 ->
#endif                       // FD 4096^2 plan kernel 13.62 / 13.53 / 13.63 / 13.8 ms, SG 2048^2 (16 waves, in-visit refresh) 5.07 / 5.10 / 5.25 / 5.46 ms
#ifndef UFM_DIRWAKE
#define UFM_DIRWAKE 0          // resident kernel: a re-visit wakes the patches along the halo entries that changed since the tile's last visit, not all sixteen
#endif                         // (round 4, measured, diagnostic builds only: FD 4096^2 evaluations per element 29.3 -> 27.4, plan kernel 14.10 -> 14.50 ms; 8192^2 24.3 -> 22.9,
                               //  35.1 -> 35.7 ms; MS-DFM 2048^2 100 -> 97, 11.35 -> 11.6 ms: the kernel is bound by its chain of dependent visits, not by its work -- DESIGN.md 11)
#ifndef UFM_DAG
#define UFM_DAG 0              // round 4 experiment, diagnostic builds only (tools/dag_probe.py, lib build/exp/libufm_dag.so): first visits of the resident
#endif                         // kernel gated by an arrival estimate (DevParams::dag_*).  Measured: -28 % tile visits, -36 % evaluations, +10 % time.
constexpr bool DAG = UFM_DAG != 0;
// HBM layout of the field: tile-major.  A tile's T x T values are contiguous
// (1 KB for T = 16: eight 128-B lines); next to them every tile keeps a *ring*: copies of the border
// values of its eight neighbours (top row, bottom row, left column, right column, four corners --
// contiguous, three lines), which the neighbours' visits keep up to date when they write their own
// borders back.  A visit therefore reads tile + ring + cost window = 14 lines where the row-major
// layout touched ~60 (two lines per field row, one or two per cost row).
constexpr int TT = T * T;                                        // floats per tile
// Which patches a wave of a 16-wave workgroup owns (the block kernel of ufm_region.h; the 16-wave tile visits of k_relax with 16 x 16 tiles).
// Patch (pr, pc) belongs to the wave with the index ((pr & 3) << 2) | (pc & 3); hardware wave p of the workgroup
// runs on SIMD p & 3 (read back from HW_ID: tools/replan_timeline.py prints it).  With index = p the four waves of a patch COLUMN class share one
// SIMD: a front that runs along the rows keeps one SIMD busy and three idle.  UFM_WAVE_PERM: hardware wave p takes the index whose column
// class is p >> 2 and whose row class is ((p & 3) - 2 * (p >> 2)) & 3, i.e. the patch class (r, c) runs on SIMD (r + 2c) & 3 -- neighbours along
// a row class two SIMDs apart, along a column one, along either diagonal one or three: no direction of a front lands on a single SIMD.
#ifndef UFM_WAVE_PERM
#define UFM_WAVE_PERM 1
#endif
// ... and of an 8-wave visit (the wave with index w owns the two patches with (pr + 2 pc) mod 8 = w): UFM_WAVE_PERM8
#ifndef UFM_WAVE_PERM8
#define UFM_WAVE_PERM8 1
#endif
__device__ __forceinline__ int wave_index8(int p) {
#if UFM_WAVE_PERM8 == 1      // index w on SIMD ((w >> 1) + 2 (w & 1)) & 3: the patches of a row AND of a column on four SIMDs
    const int s = p & 3;
    return (p >> 2) ? 2 * ((s + 2) & 3) + 1 : 2 * s;
#else
    return p;
#endif
}
__device__ __forceinline__ int wave_index16(int p) {
#if UFM_WAVE_PERM == 1      // class (r, c) on SIMD (r + 2c) & 3
    return ((((p & 3) - 2 * (p >> 2)) & 3) << 2) | (p >> 2);
#elif UFM_WAVE_PERM == 2    // (2r + c) & 3
    return ((p >> 2) << 2) | (((p & 3) - 2 * (p >> 2)) & 3);
#elif UFM_WAVE_PERM == 3    // (r + c) & 3
    return ((((p & 3) - (p >> 2)) & 3) << 2) | (p >> 2);
#elif UFM_WAVE_PERM == 4    // (r - c) & 3
    return ((((p & 3) + (p >> 2)) & 3) << 2) | (p >> 2);
#else
    return p;
#endif
}
constexpr int RING = (4 * T + 4 + 31) / 32 * 32;                 // floats per ring record (4T+4 used)
constexpr int RING_TOP = 0, RING_BOT = T, RING_LEFT = 2 * T, RING_RIGHT = 3 * T, RING_CORNER = 4 * T;   // corner order: TL TR BL BR
constexpr int CTS = ((T + 1) * (T + 1) + 127) / 128 * 128;       // bytes per cost-window record
constexpr float SQRT2F = 1.41421356237309504880168872420969807856967187537694f;  // Macros.cpp:2

enum { MODE_LOWER = 0, MODE_RAISE = 1 };
// Kernel-side operator ids: FD and SG as in include/ufm.h, and ONE operator for MS-DFM, the level-1 form (id 3; UFM_ALGO_DFM = 2 is the
// family's id at the C ABI and never a template argument).  DFMPlanner<1> never evaluates min_rhs<0>'s "best cell of each pair, then one
// quadratic per stencil" while it lowers: every expansion offers each neighbour ONE candidate built on the expanded cell itself
// (min_rhs_decreased_neighbor, DynamicFastMarching_impl.h:270-313) and RHS keeps the smallest (plan<1> :79-86).  Its consistent field is the
// fixed point of "min over the eight per-neighbour candidates" -- not the level-0 operator at the ulp level: the float quadratic is not
// monotone, so Q(min(a,b), .) and min(Q(a, .), Q(b, .)) differ in the last bit where two fronts meet (oracle's 1024^2 field: 0 of 1.02 M
// interior elements violate G = F1(G), 259 violate G = F0(G)).  A level-0 planner is served by the same operator: iterating F0 itself needs
// creep cut-offs to terminate (rounds 1-2) and lands FARTHER from the reference's level-0 field than the fixed point of F1 does -- 24 maps,
// 256^2..1024^2: <= 2.45e-6 / 27 ulp with 60-90 % of the elements off, against <= 1.03e-6 / 10 ulp with 0.3-33 %.
constexpr int ALGO_DFM1 = 3;
template <int ALGO> constexpr bool is_dfm = (ALGO == ALGO_DFM1);

constexpr int LMAX = 8192;
constexpr int INFBITS = 0x7F800000;   // +inf as int: non-negative floats order like their bits

// Two work queues: Q_LOWER (value propagation, ordered by value) and Q_RAISE (invalidation,
// keyed by the value an element had before it lost its support).  Entries that lie beyond the
// current bound (the start's key, D*-Lite's end condition) stay queued across steps -- the
// counterpart of the reference's persistent priority queue.
enum { Q_LOWER = 0, Q_RAISE = 1 };

struct DevCounters {
    int cnt[2][3];              // [queue][ring]: candidate-list lengths (ring of three, see k_relax)
    int rel[2][3];              // [queue][ring]: tiles released (relaxed) by the launch that read the list
    int lmin[2][3];             // [queue][ring]: smallest priority ever queued in the list (float bits)
    int npark[2];               // [queue]: tiles parked beyond the bound (not re-examined by every launch)
    int nready[2];              // [launch parity]: ready list k_triage built: entries expected to take long (front of the array)
    int rcursor[2];             // [launch parity]: next ready entry to hand to a workgroup
    int nshort[2];              // [launch parity]: ready entries expected to be short (filled from the back of the array)
    int last_work[2];           // [queue]: index of the last launch that released a tile (sizes the replan batches)
    int fin_blocks;             // k_replan_end: workgroups that have finished (the last one publishes)
    int kbase[2];               // [queue]: launch index at the start of a replan graph (its kernels carry offsets)
    unsigned int pubseq;        // sequence number the replan graph publishes with
    int tcount;                 // touched-list length
    int scount;                 // pending-seed-list length (survives steps)
    unsigned long long expanded;
    unsigned long long tile_visits;
    unsigned long long tile_iters;
    unsigned long long elem_evals;
    int qmin[2];                // k_queue_min: smallest priority queued
    float rbound;               // invalidation bound computed on the device (k_prepare_bound / k_check)
    int done;                   // k_check: both queues drained below the start's key
    unsigned int updated;       // k_check: num_nodes_updated summed over the consuming maps
    int done_fail;              // batch replan round in the block kernel: maps whose workgroup could not finish the replan alone
    unsigned long long raise_visits;   // tile visits of the invalidation kernel (subset of tile_visits)
    int own_stops;              // resident kernel: workgroups that left on the time limit instead of on an empty queue (cumulative)
    int own_abort;              // resident kernel: a workgroup has left on the time limit -- everybody else follows at its next decision
    unsigned long long own_vis0, own_vis1;   // tile_visits before / after the step's resident launch
};

// Per-step scalars the kernels read from memory, not from their by-value parameter block: the replans are
// replayed from captured graphs whose kernel arguments are frozen, and the reference's harness sends a new
// heuristic multiplier with every move (Tests/Planners/DFM/main.cpp:111-112).
struct DevDyn {
    float hm;                   // heuristic multiplier of the keys (0 when built like -DNO_HEURISTIC)
    int thr;                    // Graph::occupancy_threshold_uchar_
    int focused;                // honour the reference's end condition (stop at the start's key)
    int pad;
};

struct DevParams {
    float *G;                   // [NT][T][T] tile-major; elements of a tile beyond the map stay +inf
    float *Gprev;               // snapshot of a tile at its first touch in a step (same layout)
    uint8_t *bp;                // [NT][T][T] back-pointers (the level-1/2 planners' INFO, FD impl:86-111, SG :131-166, DFM :73-99), same layout as G: which of
                                // the operator's candidates gives the element's value, and which of its inputs that leans on (bp_byte); BP_NONE: goal / never set
    float *ring;                // [NT][RING] border values of each tile's eight neighbours (+inf where there is none)
    float *seen;                // [NT][RING] resident kernel: the halo values a tile's last visit converged against (entries as `ring`), [RING - 1] = 1.0f if that
                                // visit did converge: a later visit of the same step wakes only the patches along halo entries that have changed since (UFM_DIRWAKE)
    uint8_t *cost;              // [nmaps][L][W] the raster (Graph::map_)
    uint8_t *costT;             // [NT][CTS] per tile, the cost bytes its visit needs: cells (x0-1..x0+T-1, y0-1..y0+T-1) of a
                                // node tile, (x0..x0+T-1, y0..y0+T-1) of a cell tile (DFM), row-major; 255 outside the map
    int *goal;                  // [nmaps][2]
    int *cand;                  // [2 queues][3][NT] queued tiles (global tile ids), ring of three lists
    int *ready;                 // [NT] tiles released by k_triage for the following relax launch
    int *hint;                  // [NT] sweeps the tile's last visit took (longest-first hand-out)
    int *rank;                  // [NT] diagnostics (UFM_TIMING): position of the tile's priority inside the band, 0..255
    int *park;                  // [2 queues][2][NT] parked tiles (list + scratch for compaction)
    int *pflag;                 // [2 queues][NT] tile is in the park list
    int *pprio;                 // [2 queues][NT] its priority (float bits)
    int *queued;                // [2 queues][2][NT] launch index + 1 the tile was last queued for (list of that launch parity)
    unsigned long long *prio;   // [2 queues][2][NT] {tag of the launch it is queued for, float bits}: smallest value that entered the tile
                                // since its last visit (prio_key / prio_read); entries of earlier launches are stale by their tag, nobody resets them
    int *start;                 // [nmaps][4] start elements (linear index in the map, -1 unused)
    float *bnd;                 // [nmaps] k_start_bound output
    DevDyn *dyn;                // heuristic multiplier, occupancy threshold, focused flag (see DevDyn)
    float *spos;                // [nmaps][2] start position (FD/SG: Position; DFM: start cell indices)
    int *touched;               // [NT] visits of the tile in the current step
    uint8_t *fresh;             // [NT] the tile held nothing but +inf when the step first touched it (no Gprev snapshot taken)
    int *tlist;                 // [NT]
    int *sflag;                 // [NT] pending seeds (from patches)
    int *slist;                 // [NT]
    int *slist2;                // [NT] scratch
    uint8_t *mark;              // [nmaps][EX*EY] element already counted in num_updated this round
    unsigned int *num_updated;  // [nmaps]
    int *consume;               // [nmaps]
    int *lmax;                  // [LMAX] diagnostics: per launch, the largest per-wave sweep count of any tile
    int *own_prio;              // [OWN_NW][own_slots] resident lowering kernel (k_relax<.,LOWER,false,1|2>): the queue, one word per tile, grouped
                                // by the workgroup that owns the tile -- float bits of its priority, >= INFBITS = not queued (see own_push)
    int *own_lock;              // [OWN_NW][own_slots] 1 while the tile is being visited: whoever takes a tile (its owner, or an idle workgroup helping
                                // out) needs both the queue word AND this lock -- an activation that lands during a visit re-queues the tile at once
    int *own_min;               // [OWN_NW] smallest priority each owner holds (queued or in flight): a hint for the ordering band, not exact
    // Round 4 experiment: first visits gated by an approximate arrival order.  dag_a[tile] = estimate of when the front reaches
    // the tile (any monotone proxy); a tile's FIRST visit waits until every neighbour that the estimate puts clearly before it (dag_a < dag_thr
    // of the tile) has had its first visit: dag_left counts those neighbours down (grouped like own_prio).  Later visits are not gated.
    float *dag_a;               // [NT]
    float *dag_thr;             // [NT]
    int *dag_left;              // [OWN_NW][own_slots]
    int dag_on;                 // 0: off (ordering band only)
    int dag_patience;           // looks without an eligible tile after which a workgroup takes a held one anyway (an estimate may name a neighbour that never comes)
    unsigned long long own_limit;   // wall-clock ticks (100 MHz) after which the resident kernel hands back to the launch chain
    int own_flags;              // diagnostics: 1 = no tile taken ahead (every visit starts with a fresh look at the queue)
    int own_slots, own_sx, own_sy;  // words per owner = nmaps * own_sx * own_sy; blocks of 16 x (1 << own_ys) tiles per map side
    int own_nw, own_ys;             // owners (= workgroups of the resident launch) = 16 << own_ys: 256 (own_ys 4) or 512 (5)
    DevCounters *ctr;
    int EX, EY;                 // elements per map (nodes or cells)
    int L, W;                   // cells per map
    int TX, TY, NTm, NT, nmaps;
    int cells;                  // elements are cells (DFM), not nodes
    size_t gstride;             // floats per map in G (= NTm * T * T)
    size_t cstride;             // bytes per map in cost
    size_t mstride;             // bytes per map in mark
};

// Launch index: a kernel launched directly carries it; a kernel inside the captured replan graph
// carries -1 - offset and adds the base k_replan_begin_job stored (the graph is replayed unchanged).
__device__ __forceinline__ int launch_index(const DevParams &P, int qz, int k_arg) {
    return k_arg >= 0 ? k_arg : P.ctr->kbase[qz] + (-1 - k_arg);
}
// Priorities carry the launch they were queued for in their upper half -- newer launches compare smaller, so an
// atomicMin of a fresh key always beats what an earlier launch left in the word, and a reader that finds another
// launch's tag knows the word is stale.  Nothing ever has to be reset between launches, and during launch k nobody
// writes the words of launch k: every workgroup that scans the list sees the same priorities.
__device__ __forceinline__ unsigned long long prio_key(int kk, int pbits) {
    return ((unsigned long long)(unsigned int)(0x7FFFFFFF - kk) << 32) | (unsigned int)pbits;
}
__device__ __forceinline__ int prio_read(const DevParams &P, int qz, int kk, int gt) {
    const unsigned long long v = P.prio[(size_t)(qz * 2 + (kk & 1)) * P.NT + gt];
    return (int)(v >> 32) == 0x7FFFFFFF - kk ? (int)(unsigned int)v : INFBITS;
}
// queue tile gt in queue qz for launch kk (list kk % 3, priority words of parity kk & 1)
// (`banded` = false for an entry that is only parked beyond the start's key: it must not hold the
// ordering band of the other entries -- of other maps in a batch -- down)
__device__ __forceinline__ void activate(const DevParams &P, int qz, int kk, int gt, int pbits, bool banded = true) {
    const int lst = kk % 3;
    const size_t w = (size_t)(qz * 2 + (kk & 1)) * P.NT + gt;
    atomicMin(&P.prio[w], prio_key(kk, pbits));
    if (banded) atomicMin(&P.ctr->lmin[qz][lst], pbits);
    if (atomicExch(&P.queued[w], kk + 1) != kk + 1) {
        const int k = atomicAdd(&P.ctr->cnt[qz][lst], 1);
        P.cand[(qz * 3 + lst) * P.NT + k] = gt;
    }
}
// A tile whose priority lies beyond the bound (the start's key for lowering, the invalidation
// bound for raising) is parked: it leaves the launch-to-launch candidate ring -- carrying hundreds
// of such entries through every launch cost ~4 us per launch -- and is looked at again by k_unpark
// when a phase starts (the bound only matters then).  Counterpart of the entries the reference
// leaves in its priority queue when end_condition() fires.
__device__ __forceinline__ void park_tile(const DevParams &P, int qz, int gt, int pbits) {
    atomicMin(&P.pprio[qz * P.NT + gt], pbits);
    if (atomicExch(&P.pflag[qz * P.NT + gt], 1) == 0) P.park[(size_t)(qz * 2) * P.NT + atomicAdd(&P.ctr->npark[qz], 1)] = gt;
}
// address of element (x, y) of map m in the tile-major field
__device__ __host__ __forceinline__ size_t gaddr(const DevParams &P, int m, int x, int y) {
    return ((size_t)m * P.NTm + (size_t)(x / T) * P.TY + (y / T)) * TT + (size_t)(x % T) * T + (y % T);
}
// D*-Lite end condition as a bound on useful work (FieldDPlanner_impl.h:225-256,
// ShiftedGridPlanner_impl.h:355-386, DynamicFastMarching_impl.h:315-320): the largest key
// among the start elements that have been reached; +inf while none has.
__device__ __forceinline__ float start_bound(const DevParams &P, int m) {
    float b = 0.0f;
    const float sx = P.spos[2 * m], sy = P.spos[2 * m + 1], hm = P.dyn->hm;
    for (int i = 0; i < 4; ++i) {
        const int e = P.start[4 * m + i];
        if (e < 0) continue;
        const int x = e / P.EY, y = e - x * P.EY;
        const float g = __hip_atomic_load(&P.G[gaddr(P, m, x, y)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // heuristic keys (FD impl:178-186, DFM impl:146-155): first component k + hm * dist(start, s)
        if (g < INFINITY) b = fmaxf(b, g + hm * hypotf(sx - (float)x, sy - (float)y));
    }
    return b > 0.0f ? b : INFINITY;
}
// admissible lower bound of hm * dist(start, s) over the elements s of a tile: with heuristic keys
// an element is only worth relaxing while value + hm * dist < the start's key
__device__ __forceinline__ float tile_heuristic(const DevParams &P, int m, int tx, int ty) {
    const float hm = P.dyn->hm;
    if (hm == 0.0f) return 0.0f;
    const float sx = P.spos[2 * m], sy = P.spos[2 * m + 1];
    const float x0 = (float)(tx * T), x1 = (float)(tx * T + T - 1), y0 = (float)(ty * T), y1 = (float)(ty * T + T - 1);
    const float dx = fmaxf(fmaxf(x0 - sx, sx - x1), 0.0f), dy = fmaxf(fmaxf(y0 - sy, sy - y1), 0.0f);
    return hm * hypotf(dx, dy) * 0.999f;   // (0.999: stay below the reference's own float rounding of the distance)
}

// ---- the queue of the resident lowering kernel ---------------------------------------------------------
// One launch runs a whole lowering phase: OWN_NW workgroups, one per CU, stay resident and each owns the tiles
// (tx, ty) with (tx mod 16, ty mod 16) = its index -- any stretch of a front is spread over all of them.  A tile's
// queue entry is ONE word that only its owner ever removes:
//     key (< INFBITS)  queued with that priority          -- neighbours lower it with atomicMin, fire and forget
//     OWN_MARK + w     being visited by workgroup w        -- an atomicMin of a key re-queues it meanwhile
//     other >= INFBITS empty (INFBITS + 1 + a per-owner visit count: the word never returns to an earlier empty value)
// so a tile is never visited twice at once, no list is appended to and no cursor is shared.  The owner resets
// MARK -> empty only after the activations of that visit have been performed: the words of all owners are non-empty
// as long as anything is queued, in flight, or about to be queued, and two identical all-empty collects of them in a
// row mean the phase is over (an empty value never repeats, so identical collects are a true snapshot).
constexpr int OWN_NW = 512;    // most owners a launch of the resident kernel has (P.own_nw: 256 or 512)
constexpr int OWN_MARK = 0x7FFFFE00;     // + the visiting workgroup (< OWN_NW): a visitor takes back its own mark only
constexpr unsigned int OWN_EMPTIES = 0x7FFDFEu;   // empty values: INFBITS + 1 + (0 .. OWN_EMPTIES - 1), all below the marks
__device__ __forceinline__ void own_locate(const DevParams &P, int gt, int &o, int &s) {
    const int m = gt / P.NTm, t = gt - m * P.NTm, tx = t / P.TY, ty = t - tx * P.TY;
    o = ((tx & 15) << P.own_ys) | (ty & ((1 << P.own_ys) - 1));
    s = (m * P.own_sx + (tx >> 4)) * P.own_sy + (ty >> P.own_ys);
}
// tile of slot s of owner o; -1 if that position lies outside the map
__device__ __forceinline__ int own_tile(const DevParams &P, int o, int s, int &m, int &tx, int &ty) {
    const int per = P.own_sx * P.own_sy;
    m = s / per;
    const int r = s - m * per, bx = r / P.own_sy, by = r - bx * P.own_sy;
    tx = bx * 16 + (o >> P.own_ys); ty = (by << P.own_ys) + (o & ((1 << P.own_ys) - 1));
    return (tx < P.TX && ty < P.TY) ? m * P.NTm + tx * P.TY + ty : -1;
}
#ifdef UFM_TIMING
// per tile (resident kernel): [0] first visit start, [1] end of the last visit that changed a value, [2] earliest activation not yet
// taken, [3] visits, [4] sum of activation -> visit start waits; 100 MHz ticks since the launch's first visit (g_tile_t0)
constexpr int TILE_DIAG_MAX = 1 << 19;
__device__ unsigned int g_tile[5][TILE_DIAG_MAX];
__device__ unsigned long long g_tile_t0;
__device__ unsigned long long g_sdiag[16];   // looks of idle workgroups: [0] looks, [1] with nothing to take, [2] helping attempts, [3] a victim's word found,
                                             // [4] inside the band, [5] taken, [6] takes ahead that failed, [7] fresh takes that failed
// ... and the visits themselves, for the critical path: {tile, start, end, earliest activation taken: time, tile that sent it}
constexpr int VIS_DIAG_MAX = 1 << 20;
__device__ unsigned long long g_push64[TILE_DIAG_MAX];     // per tile: {time, sender} of the earliest activation not yet taken
__device__ unsigned int g_vis[VIS_DIAG_MAX][5];
__device__ unsigned int g_nvis;
// events around the corner tiles (0..1, 0..1) of map 0: {time, a, b, c}; b = 0xFFFFFFFF: workgroup 0 saw the end; 0xFFFFFFFE: in-visit refresh
// (a = tile, c = what the exchange returned); 0xFFFFFFFD: a look of the owner (a = workgroup, c = its best priority); else an activation a -> b with priority c
constexpr int PLOG_MAX = 1 << 14;
__device__ unsigned int g_plog[PLOG_MAX][4];
__device__ unsigned int g_nplog;
__device__ __forceinline__ void plog(unsigned int a, unsigned int b, unsigned int c) {
    const unsigned int i = atomicAdd(&g_nplog, 1u);
    if (i < PLOG_MAX) { g_plog[i][0] = (unsigned int)(wall_clock64() - g_tile_t0); g_plog[i][1] = a; g_plog[i][2] = b; g_plog[i][3] = c; }
}
#endif
// -DUFM_STRICT_FENCES (a checking build, libufm_strict.so: tests/test_strict_fences.py holds the product build to it bit for bit): the
// textbook form of the protocol -- an agent-scope release fence in front of every activation and of every lock release, an agent-scope
// acquire fence behind every take -- next to the product's argued one (sc1 stores and loads, s_waitcnt vmcnt(0), relaxed atomics; the table
// in DESIGN.md section 4.7).  A release here writes the XCD's L2 back, an acquire invalidates it: several times slower, same results.
#ifdef UFM_STRICT_FENCES
#define UFM_STRICT_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent")
#define UFM_STRICT_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#else
#define UFM_STRICT_RELEASE()
#define UFM_STRICT_ACQUIRE()
#endif
__device__ __forceinline__ void own_push(const DevParams &P, int gt, int pbits, int from = -1) {
    int o, s;
    own_locate(P, gt, o, s);
    UFM_STRICT_RELEASE();
    __hip_atomic_fetch_min(&P.own_prio[(size_t)o * P.own_slots + s], pbits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_min(&P.own_min[o], pbits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef UFM_TIMING
    if (gt < TILE_DIAG_MAX) {
        const unsigned int now = (unsigned int)(wall_clock64() - g_tile_t0);
        atomicMin(&g_tile[2][gt], now);
        atomicMin(&g_push64[gt], ((unsigned long long)now << 32) | (unsigned int)from);
        if (gt / P.TY < 2 && gt % P.TY < 2) plog((unsigned int)from, (unsigned int)gt, (unsigned int)pbits);
    }
#endif
}
// Values other workgroups write while the resident kernel runs are read and written past the per-XCD L2
// (agent-scope accesses); the launch-per-band-step kernels rely on the kernel boundaries instead.
// A workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every global access in flight, which
// is exactly what the resident kernel's decision -- made while its stores and its prefetches are on their way -- must not do.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <bool COH> __device__ __forceinline__ float ld_f(const float *p) {
    if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <bool COH> __device__ __forceinline__ void st_f(float *p, float v) {
    if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// ---- optional in-kernel timing of tile visits (-DUFM_TIMING, diagnostic builds only) -------
// g_tdiag: [0] sum of pop->staged, [1] sum of sweep phases, [2] sum of write-back/activation,
// [3] visits, [4] sum of per-block busy time, [5] blocks, [8..39] histogram of visit times (2 us bins)
// all in 10 ns ticks of the constant 100 MHz counter
#ifdef UFM_TIMING
__device__ unsigned long long g_tdiag[64];
// trace of the lowering launches UFM_TRACE_K0 .. +7: per record {launch | block<<16 | kind<<40, t0, t1, sweeps};
// kind 0 = tile visit (pop .. end), 1 = block lifetime (entry .. exit)
#ifndef UFM_TRACE_K0
#define UFM_TRACE_K0 300
#endif
__device__ unsigned long long g_trace[4 * 16384];
__device__ unsigned int g_ntrace;
// per-wave timeline of ONE tile visit (the first long-list visit of workgroup 0 in launch UFM_TRACE_K0):
// records {type, t, value}; 1 burst start (wake bits), 2 burst end (sweeps in it), 3 idle, 4 woken, 5 vote
__device__ unsigned long long g_wtrace[16 * 256 * 2];
__device__ unsigned int g_nw[16];
#define UFM_WREC(type, val) do { if (wtrace_on && lane == 0) { const unsigned int i_ = g_nw[w]++; if (i_ < 256) { \
    g_wtrace[(w * 256 + i_) * 2] = ((unsigned long long)(type) << 32) | (unsigned int)(val); g_wtrace[(w * 256 + i_) * 2 + 1] = wall_clock64(); } } } while (0)
__device__ __forceinline__ void trace_rec(int k, int kind, unsigned long long t0, unsigned long long t1, long long sw) {
    if (k < UFM_TRACE_K0 || k >= UFM_TRACE_K0 + 8) return;
    const unsigned int i = atomicAdd(&g_ntrace, 1u);
    if (i >= 16384) return;
    g_trace[4 * i] = (unsigned long long)k | ((unsigned long long)blockIdx.x << 16) | ((unsigned long long)kind << 40);
    g_trace[4 * i + 1] = t0; g_trace[4 * i + 2] = t1; g_trace[4 * i + 3] = (unsigned long long)sw;
}
#define UFM_TICK(v) const unsigned long long v = wall_clock64()
#else
#define UFM_TICK(v)
#define UFM_WREC(type, val)
#endif

