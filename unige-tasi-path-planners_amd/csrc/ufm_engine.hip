// ufm_engine.hip -- MI355X (gfx950) cost-propagation engine behind include/ufm.h.
//
// What it replaces: the priority-queue driven G/RHS wavefront of the reference
// replanners -- ReplannerBase::step (ProjectToolkit/include/ReplannerBase.h:43-75)
// with FieldDPlanner / ShiftedGridPlanner / DFMPlanner init/update/plan
// (FieldDStar/FieldDPlanner_impl.h:15-163, ShiftedGridFastMarching/
// ShiftedGridPlanner_impl.h:9-231, DynamicFastMarching/DynamicFastMarching_impl.h:6-132).
//
// How: the serial D*-Lite expansion order is replaced by a block Fast Iterative Method ordered
// like fast marching at tile granularity.  The field G lives densely in HBM; the domain is cut
// into UFM_TILE x UFM_TILE-element tiles (16 by default); per-tile priorities (the smallest value
// that entered a tile since its last visit) drive launches of k_relax, which releases the tiles of
// the current band, stages each (+1 halo) in LDS, sweeps it to its local fixed point with
// asynchronous waves (one 4x4-node patch per wave, four lanes per node), writes it back and queues
// the neighbours whose halo changed.  The fixed point G = F(G) is unique (costs >= 1), so it equals
// the reference's consistent field.  Map patches (cost increases) are handled by an invalidation
// ("raise") phase -- an element whose value is no longer supported by its neighbours is reset to
// +inf, transitively -- followed by the usual lowering phase; both stop at the start's key like the
// reference's end_condition and keep the rest queued across steps.  DESIGN.md has the full story.
//
// Arithmetic contract (bit parity with oracle/ufm_oracle.c): IEEE fp32, one
// rounding per operation (-ffp-contract=off), correctly rounded sqrt.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <chrono>
#include <climits>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>
#include <thread>
#include <vector>

#include "../../include/ufm.h"

namespace {

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
#ifndef UFM_DFM_LAX_VISITS
#define UFM_DFM_LAX_VISITS 16     // DFM: after this many visits of a tile in one step a 1-ulp rise is rounding noise
#endif
#ifndef UFM_DFM_QUIET_VISITS
#define UFM_DFM_QUIET_VISITS 24   // DFM: ... and a decrease of <= 4 ulp no longer wakes the neighbours
#endif
// Level 1 (ALGO_DFM1) converges without such cut-offs on almost every map; they are its livelock guard only: block
// Gauss-Seidel between two tiles can cycle through a finite set of last-bit states (2048^2, seed 1006)
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
#ifndef UFM_LOOK_SLEEP
#define UFM_LOOK_SLEEP 32      // pause of a workgroup that found nothing to visit before it looks again (x 64 clocks)
#endif
#ifndef UFM_STEAL_VICTIMS
#define UFM_STEAL_VICTIMS 4    // owners whose words an idle workgroup looks at per look (k_relax, own_steal)
#endif
#ifndef UFM_EARLY_POLLS
#define UFM_EARLY_POLLS 6      // ... its queue words follow after this many looks of the idle wave at its wake bits (the stores have landed by then)
#endif
#ifndef UFM_IDLE_SLEEP
#define UFM_IDLE_SLEEP 4
#endif
// HBM layout of the field: tile-major.  A tile's T x T values are contiguous
// (1 KB for T = 16: eight 128-B lines); next to them every tile keeps a *ring*: copies of the border
// values of its eight neighbours (top row, bottom row, left column, right column, four corners --
// contiguous, three lines), which the neighbours' visits keep up to date when they write their own
// borders back.  A visit therefore reads tile + ring + cost window = 14 lines where the row-major
// layout touched ~60 (two lines per field row, one or two per cost row).
constexpr int TT = T * T;                                        // floats per tile
constexpr int RING = (4 * T + 4 + 31) / 32 * 32;                 // floats per ring record (4T+4 used)
constexpr int RING_TOP = 0, RING_BOT = T, RING_LEFT = 2 * T, RING_RIGHT = 3 * T, RING_CORNER = 4 * T;   // corner order: TL TR BL BR
constexpr int CTS = ((T + 1) * (T + 1) + 127) / 128 * 128;       // bytes per cost-window record
constexpr float SQRT2F = 1.41421356237309504880168872420969807856967187537694f;  // Macros.cpp:2

enum { MODE_LOWER = 0, MODE_RAISE = 1 };
// Kernel-side operator ids: the three planner families of include/ufm.h plus the level-1 form of MS-DFM.
// DFMPlanner<1> never evaluates min_rhs<0>'s "best cell of each pair, then one quadratic per stencil" while it
// lowers: every expansion offers each neighbour ONE candidate built on the expanded cell itself
// (min_rhs_decreased_neighbor, DynamicFastMarching_impl.h:270-313) and RHS keeps the smallest (plan<1> :79-86).
// Its consistent field is therefore the fixed point of "min over the eight per-neighbour candidates" -- which
// is not the level-0 operator at the ulp level: the float quadratic is not monotone, so Q(min(a,b), .) and
// min(Q(a, .), Q(b, .)) differ in the last bit where two fronts meet.  Measured on the oracle's 1024^2 field:
// 0 of 1.02 M interior elements violate G = F1(G), 259 violate G = F0(G).
constexpr int ALGO_DFM1 = 3;
template <int ALGO> constexpr bool is_dfm = (ALGO == UFM_ALGO_DFM || ALGO == ALGO_DFM1);

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

#ifdef UFM_SWEEPSTAT
// diagnostics (-DUFM_SWEEPSTAT, tools/sweep_stats.py): what the patch sweeps of k_relax find.  [0] sweeps, [1] sweeps that changed no node,
// [2] node values changed, [3] bursts, [4] bursts whose first sweep changed nothing, [5] node values lowered, [8..23] histogram of sweeps per burst (1..16)
__device__ unsigned long long g_sstat[32];
#endif
// ---- update operators -------------------------------------------------------
// Correctly rounded fp32 square root (std::sqrt of the reference, Macros.h:12):
// v_sqrt_f32 is good to 1 ulp; two fused residuals pick the neighbour that is
// the round-to-nearest result.  Arguments here are never denormal, so the
// scaling steps of the generic library routine are omitted.
__device__ __forceinline__ float sqrt_rn(float x) {
    float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __int_as_float(__float_as_int(s) - 1);
    const float sp = __int_as_float(__float_as_int(s) + 1);
    const float rm = __builtin_fmaf(-sm, s, x);
    const float rp = __builtin_fmaf(-sp, s, x);
    s = (rm <= 0.0f) ? sm : s;
    s = (rp > 0.0f) ? sp : s;
    return s;
}
// The traversal-cost case analyses below are evaluated branch-free: every quantity that
// depends only on the cell costs is folded, once per tile visit, into per-lane constants chosen
// so that IEEE comparisons/selects reproduce the reference's if/else chain exactly, including
// the +inf (obstacle / unreached) cases.  `|` and `&` on bools are deliberate (no short-circuit
// control flow in the sweep loop).

// ShiftedGridPlanner_impl.h:422-436 (+ InterpolatedTraversal.cpp:125-127,324-326,403-405):
//   g1,g2 both inf -> inf ; c inf -> inf ; f = g1-g2
//   f <= 0 -> g1 + c ; f*SQRT2 <= c -> g1 + sqrt(c^2-f^2) ; else g2 + c*SQRT2
struct CellSG {          // per cell c
    float cadd;          // c (inf for an obstacle)
    float ccmp;          // c, or -1 for an obstacle so that "f*SQRT2 <= c" fails and Type A (= inf) is taken
    float c2, cs2;       // c*c, c*SQRT2
    __device__ __forceinline__ void set(float c) {
        cadd = c; ccmp = (c == INFINITY) ? -1.0f : c; c2 = c * c; cs2 = c * SQRT2F;
    }
};
__device__ __forceinline__ float tri_sg(float g1, float g2, const CellSG &K) {
    const float f = g1 - g2;                       // NaN when both inf -> every test false -> tA = inf
    const float tII = g1 + sqrt_rn(K.c2 - f * f);
    const float tA = g2 + K.cs2;
    float r = (f * SQRT2F <= K.ccmp) ? tII : tA;
    r = (f <= 0.0f) ? (g1 + K.cadd) : r;
    return r;
}

// FieldDPlanner_impl.h:292-319 (+ InterpolatedTraversal.cpp:8-10,125-127,236-238,324-326,403-405):
//   c > b : f<=0 or f^2 <= CATH(c,b) -> g1+b (III) ; f<=b and c > f*SQRT2 -> g1+CATH(c,f) (II) ;
//           f>b and c > b*SQRT2 -> g2+b+CATH(c,b) (I) ; else g2+c*SQRT2 (A)
//   c <= b: f<=0 -> g1+c (B) ; f*SQRT2 < c -> g1+CATH(c,f) (II) ; else (A)
// Unified: with bp = (c>b ? b : c), cbp = (c>b ? CATH(c,b) : -1), bI = (c > b*SQRT2 ? b : inf)
//   r = A ; if (f > bI) r = (g2+bp)+cbp ; if (f <= bp & c > f*SQRT2) r = II ; if (f<=0 | f^2 <= cbp) r = g1+bp
// (for c <= b the test f <= bp=c is implied by f*SQRT2 < c; c = inf is folded as bp = inf.)
struct CellFD { float c, c2, cs2; };
struct TriFD {
    float bp, cbp, bI;
    __device__ __forceinline__ void set(float c, float b) {
        const bool cgb = c > b;
        bp = cgb ? b : c;
        cbp = cgb ? sqrt_rn(c * c - b * b) : -1.0f;          // CATH(c,b), Macros.h:12
        bI = (c > b * SQRT2F) ? b : INFINITY;
        if (c == INFINITY) { bp = INFINITY; cbp = -1.0f; }
    }
};
__device__ __forceinline__ float tri_fd(float g1, float g2, const CellFD &K, const TriFD &Q) {
    const float f = g1 - g2;
    const float ff = f * f;
    const float tII = g1 + sqrt_rn(K.c2 - ff);
    float r = g2 + K.cs2;                                                  // A
    r = (f > Q.bI) ? ((g2 + Q.bp) + Q.cbp) : r;                            // I
    r = ((f <= Q.bp) & (K.c > f * SQRT2F)) ? tII : r;                      // II
    r = ((f <= 0.0f) | (ff <= Q.cbp)) ? (g1 + Q.bp) : r;                   // III / B
    return r;
}
// DynamicFastMarching_impl.h:322-342
__device__ __forceinline__ float q_dfm(float a, float b, float th) {
    const float ga = fminf(a, b), gb = fmaxf(a, b);
    const float d = gb - ga;
    const float s = ((ga + gb) + sqrt_rn(2.0f * (th * th) - d * d)) * 0.5f;
    return (th > d) ? s : (ga + th);        // both inf -> d NaN -> ga + th = inf
}

// ---- quad evaluation ---------------------------------------------------------------------
// The eight triangles around a node split naturally by the cell they lie in.  Four adjacent
// lanes (a DPP quad) own one node; lane q evaluates the two triangles of cell q (same c, same
// diagonal neighbour) and a two-step quad_perm min gives RHS to all four lanes.  The dependent
// instruction chain of one sweep -- what the critical path of a tile visit is made of -- is a
// quarter of the one-lane-per-node form.  Cell q of node (x,y): (x-1+dx, y-1+dy), dx=q>>1, dy=q&1;
// its triangles: (p1 = vertical neighbour, p2 = diagonal) and (p1 = horizontal neighbour, p2).
template <int ALGO> struct QuadConsts;
// load_at: `cost(r, c)` returns the traversal cost (float, +inf = obstacle / outside) of entry (r, c) of the staged
// cost window -- row r, column c <-> cell (x0 + r - off, y0 + c - off), off = 1 for node planners; (lx, ly) is the
// element inside the staged block.  load(): the tile kernel's float window Cs with pitch CP.
template <> struct QuadConsts<UFM_ALGO_DFM> {
    float th;   // lane 0: cost (orthogonal stencil, h = 1); lane 1: cost*SQRT2 (diagonal stencil)
    template <class CostAt> __device__ __forceinline__ void load_at(CostAt cost, int lx, int ly, int q, int /*gpitch*/) {
        const float tau = cost(lx, ly);
        th = (q & 1) ? tau * SQRT2F : tau;
    }
    __device__ __forceinline__ void load(const float *Cs, int lx, int ly, int q) { load_at([=](int r, int c) { return Cs[r * CP + c]; }, lx, ly, q, GP); }
};
template <> struct QuadConsts<ALGO_DFM1> {
    float th;   // lanes 0, 1: cost (h = 1); lanes 2, 3: cost*SQRT2 (= cost * HYPOT(+-1, +-1))
    int so, po; // LDS offsets: +-so = the lane's two neighbours, +-po = the perpendicular pair of the same stencil
    template <class CostAt> __device__ __forceinline__ void load_at(CostAt cost, int lx, int ly, int q, int gpitch) {
        const float tau = cost(lx, ly);
        th = (q & 2) ? tau * SQRT2F : tau;
        so = (q == 0) ? gpitch : (q == 1) ? 1 : (q == 2) ? gpitch - 1 : gpitch + 1;   // vertical | horizontal | TR-BL | TL-BR
        po = (q == 0) ? 1 : (q == 1) ? gpitch : (q == 2) ? gpitch + 1 : gpitch - 1;   // (dx != dy -> TL/BR pair, dx == dy -> BL/TR pair, impl:284-296)
    }
    __device__ __forceinline__ void load(const float *Cs, int lx, int ly, int q) { load_at([=](int r, int c) { return Cs[r * CP + c]; }, lx, ly, q, GP); }
};
template <> struct QuadConsts<UFM_ALGO_SG> {
    CellSG k;
    template <class CostAt> __device__ __forceinline__ void load_at(CostAt cost, int lx, int ly, int q, int /*gpitch*/) {
        k.set(cost(lx + (q >> 1), ly + (q & 1)));
    }
    __device__ __forceinline__ void load(const float *Cs, int lx, int ly, int q) { load_at([=](int r, int c) { return Cs[r * CP + c]; }, lx, ly, q, GP); }
};
template <> struct QuadConsts<UFM_ALGO_FD> {
    CellFD k;
    TriFD tv, th;   // b = the cell across the edge s-p1 (FieldDPlanner_impl.h:322-337)
    template <class CostAt> __device__ __forceinline__ void load_at(CostAt cost, int lx, int ly, int q, int /*gpitch*/) {
        const int dx = q >> 1, dy = q & 1;
        const float c = cost(lx + dx, ly + dy);
        const float bv = cost(lx + dx, ly + 1 - dy);   // across the vertical edge s-p1
        const float bh = cost(lx + 1 - dx, ly + dy);   // across the horizontal edge s-p1
        k = {c, c * c, c * SQRT2F};
        tv.set(c, bv);
        th.set(c, bh);
    }
    __device__ __forceinline__ void load(const float *Cs, int lx, int ly, int q) { load_at([=](int r, int c) { return Cs[r * CP + c]; }, lx, ly, q, GP); }
};

__device__ __forceinline__ float quad_min(float v) {
#if UFM_DPP_MIN_ASM
    // v_min_f32 with a DPP source operand: one instruction per step instead of mov_dpp + canonicalise + min
    // (IEEE mode: v_min_f32 returns the non-NaN operand like fminf; the values here are never NaN).
    // The s_nop covers the VALU-write -> DPP-read hazard of the second step.
    float r;
    asm volatile("s_nop 1\n\t"      // the compiler does not see a DPP read of %1 in here: cover its hazard too
                 "v_min_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
                 : "=&v"(r) : "v"(v));
    return r;
#else
    int x = __float_as_int(v);
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false)));   // quad_perm [1,0,3,2]
    x = __float_as_int(v);
    return fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false)));   // quad_perm [2,3,0,1]
#endif
}

// Back-pointers (DevParams::bp), one byte per element: (code << 2) | dep.  Written once per step, when it has converged, for every tile the step
// touched (k_finalize_bp; the block kernel of a replan does it for the tiles it changed, ufm_region.h): one evaluation of the operator on the
// final values with the arg-min kept -- not in the sweeps, where tracking the winner cost the plan 9 % and still left bytes behind whose
// triangle no longer gave the value (the operator's case analysis is not monotone).
// code, node planners: (q << 1) | h -- the triangle of cell q (the quad lane that evaluated it) whose edge neighbour p1 is the vertical (h = 0) or the
// horizontal one (h = 1); its other vertex p2 is the diagonal node of that cell.  MS-DFM level 0: the stencil (0 orthogonal, 1 diagonal); level 1:
// (q << 1) | which of the axis's two neighbours (0: -so, 1: +so).  Of several candidates that tie, the lowest code.
// dep, node planners: which of the two vertices the value depends on -- bit 0: G(p1), bit 1: G(p2) -- by the case compute_optimal_cost took
// (FD impl:292-319, SG :422-436): "g1 + ..." (III, B) leans on p1 alone, "g2 + ..." (I, A) on p2 alone, the interpolated case (II) on both.  With the other
// vertex at +inf the same case is taken and gives the same value, so: an element is gone exactly when a vertex it depends on is gone (the invalidation of
// ufm_region.h follows these bits without evaluating anything).  MS-DFM: 3.
constexpr int BP_NONE = 0xFF;
__device__ __forceinline__ int dep_sg(float g1, float g2, const CellSG &K) {
    const float f = g1 - g2;
    return (f <= 0.0f) ? 1 : ((f * SQRT2F <= K.ccmp) ? 3 : 2);
}
__device__ __forceinline__ int dep_fd(float g1, float g2, const CellFD &K, const TriFD &Q) {
    const float f = g1 - g2, ff = f * f;
    return ((f <= 0.0f) | (ff <= Q.cbp)) ? 1 : (((f <= Q.bp) & (K.c > f * SQRT2F)) ? 3 : ((f > Q.bI) ? 2 : 2));
}
// A lane's evaluation with what the back-pointer needs: r = the smaller of the lane's candidates, h = it was the second one, and (node planners) the
// three neighbour values it was computed from.
struct LaneEval { float r; bool h; float gV, gH, gD; };
template <int ALGO, int GPITCH = GP>
__device__ __forceinline__ LaneEval eval_quad_w(const float *ctr, int q, const QuadConsts<ALGO> &C) {
    LaneEval e;
    e.gV = e.gH = e.gD = 0.0f;
    if constexpr (ALGO == UFM_ALGO_DFM) {
        e.r = INFINITY;
        if (q < 2) {
            const int a = q ? -GPITCH - 1 : -GPITCH, b = q ? GPITCH + 1 : GPITCH, c = q ? GPITCH - 1 : -1, d = q ? -GPITCH + 1 : 1;
            e.r = q_dfm(fminf(ctr[a], ctr[b]), fminf(ctr[c], ctr[d]), C.th);
        }
        e.h = false;
    } else if constexpr (ALGO == ALGO_DFM1) {
        const float pm = fminf(ctr[-C.po], ctr[C.po]);
        const float a = q_dfm(ctr[-C.so], pm, C.th), b = q_dfm(ctr[C.so], pm, C.th);
        e.h = b < a;
        e.r = e.h ? b : a;
    } else {
        const int sx = (q & 2) ? GPITCH : -GPITCH, sy = (q & 1) ? 1 : -1;
        e.gD = ctr[sx + sy]; e.gV = ctr[sx]; e.gH = ctr[sy];
        float tV, tH;
        if constexpr (ALGO == UFM_ALGO_SG) { tV = tri_sg(e.gV, e.gD, C.k); tH = tri_sg(e.gH, e.gD, C.k); }
        else { tV = tri_fd(e.gV, e.gD, C.k, C.tv); tH = tri_fd(e.gH, e.gD, C.k, C.th); }
        e.h = tH < tV;
        e.r = e.h ? tH : tV;
    }
    return e;
}
// ... and the byte for a lane that holds the quad's minimum (0x3FF for one that does not: the quad's smallest is the lowest winning code)
template <int ALGO>
__device__ __forceinline__ int bp_byte(const LaneEval &e, int q, const QuadConsts<ALGO> &C, bool winner) {
    int code, dep = 3;
    if constexpr (ALGO == UFM_ALGO_DFM) code = q;
    else code = (q << 1) | (e.h ? 1 : 0);
    if constexpr (ALGO == UFM_ALGO_SG) dep = dep_sg(e.h ? e.gH : e.gV, e.gD, C.k);
    if constexpr (ALGO == UFM_ALGO_FD) {
        TriFD t;
        t.bp = e.h ? C.th.bp : C.tv.bp; t.cbp = e.h ? C.th.cbp : C.tv.cbp; t.bI = e.h ? C.th.bI : C.tv.bI;
        dep = dep_fd(e.h ? e.gH : e.gV, e.gD, C.k, t);
    }
    return winner ? ((code << 2) | dep) : 0x3FF;
}
__device__ __forceinline__ int quad_min_int(int v) {
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
    return min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false)); // quad_perm [2,3,0,1]
}
// Node planners, invalidation: the value the element's OWN parent triangle (stored byte bpb) gives now -- what the reference's level-1/2 planners
// look at when a neighbour is raised (FD impl:100-110: only elements whose back-pointer involves the raised node are recomputed).  +inf from the
// lanes of the other cells (and from every lane when there is no parent: the quad's min is then +inf, and a finite value without a parent goes).
template <int ALGO, int GPITCH = GP>
__device__ __forceinline__ float eval_quad_bp(const float *ctr, int q, const QuadConsts<ALGO> &C, int bpb) {
    static_assert(ALGO == UFM_ALGO_FD || ALGO == UFM_ALGO_SG, "node planners");
    const int bpc = bpb >> 2;
    const int sx = (q & 2) ? GPITCH : -GPITCH, sy = (q & 1) ? 1 : -1;
    const bool h = bpc & 1;
    const float gD = ctr[sx + sy], g1 = ctr[h ? sy : sx];
    float r;
    if constexpr (ALGO == UFM_ALGO_SG) r = tri_sg(g1, gD, C.k);
    else {
        TriFD t;
        t.bp = h ? C.th.bp : C.tv.bp; t.cbp = h ? C.th.cbp : C.tv.cbp; t.bI = h ? C.th.bI : C.tv.bI;
        r = tri_fd(g1, gD, C.k, t);
    }
    return (bpc >> 1) == q ? r : INFINITY;
}
// ctr points at the node inside the LDS tile; returns this lane's share of RHS(node)
template <int ALGO, int GPITCH = GP>
__device__ __forceinline__ float eval_quad(const float *ctr, int q, const QuadConsts<ALGO> &C) {
    if constexpr (ALGO == UFM_ALGO_DFM) {
        // DynamicFastMarching_impl.h:157-210: best_cell (:344-351) is a min on values, and
        // "diag < ortho ? diag : ortho" is the quad min of the two stencil solves
        float r = INFINITY;
        if (q < 2) {
            const int a = q ? -GPITCH - 1 : -GPITCH, b = q ? GPITCH + 1 : GPITCH, c = q ? GPITCH - 1 : -1, d = q ? -GPITCH + 1 : 1;
            r = q_dfm(fminf(ctr[a], ctr[b]), fminf(ctr[c], ctr[d]), C.th);
        }
        return r;
    } else if constexpr (ALGO == ALGO_DFM1) {
        // DynamicFastMarching_impl.h:270-313 for the two neighbours of this lane's axis: g_a = G(nbr), g_b = the
        // better cell of the perpendicular pair; RHS = the smallest of the eight candidates (plan<1> :79-86)
        const float pm = fminf(ctr[-C.po], ctr[C.po]);
        return fminf(q_dfm(ctr[-C.so], pm, C.th), q_dfm(ctr[C.so], pm, C.th));
    } else {
        const int sx = (q & 2) ? GPITCH : -GPITCH, sy = (q & 1) ? 1 : -1;
        const float gD = ctr[sx + sy], gV = ctr[sx], gH = ctr[sy];
        if constexpr (ALGO == UFM_ALGO_SG)   // ShiftedGridPlanner_impl.h:258-264
            return fminf(tri_sg(gV, gD, C.k), tri_sg(gH, gD, C.k));
        else                                 // FieldDPlanner_impl.h:188-194
            return fminf(tri_fd(gV, gD, C.k, C.tv), tri_fd(gH, gD, C.k, C.th));
    }
}

// The cost windows (DevParams::costT).  Entry (cr, cc) of tile (tx, ty) is cell (tx*T + cr - off, ty*T + cc - off),
// off = 1 and T+1 rows for node tiles (a node's four cells), off = 0 and T rows for cell tiles (DFM).
__device__ __forceinline__ void cost_window_store(const DevParams &P, int m, int cx, int cy, uint8_t v) {
    const int bx = cx / T, by = cy / T, rx = cx % T, ry = cy % T;
    if (P.cells) {
        P.costT[((size_t)m * P.NTm + (size_t)bx * P.TY + by) * CTS + rx * T + ry] = v;
        return;
    }
    // a cell is read by the node tile that holds its lower-right corner nodes and, on a tile edge, by the next one
#pragma unroll
    for (int dx = 0; dx < 2; ++dx)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            if ((dx && rx != T - 1) || (dy && ry != T - 1)) continue;
            const int tx = bx + dx, ty = by + dy;
            if (tx >= P.TX || ty >= P.TY) continue;
            const int cr = dx ? 0 : rx + 1, cc = dy ? 0 : ry + 1;
            P.costT[((size_t)m * P.NTm + (size_t)tx * P.TY + ty) * CTS + cr * (T + 1) + cc] = v;
        }
}
// all windows of map m from its raster (set_map)
__global__ void k_cost_windows(DevParams P, int m) {
    const int crows = P.cells ? T : T + 1, off = P.cells ? 0 : 1, per = crows * crows;
    const uint8_t *cm = P.cost + (size_t)m * P.cstride;
    const size_t n = (size_t)P.NTm * per;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(i / per), e = (int)(i - (size_t)t * per);
        const int tx = t / P.TY, ty = t - tx * P.TY, cr = e / crows, cc = e - cr * crows;
        const int cx = tx * T + cr - off, cy = ty * T + cc - off;
        const bool in = cx >= 0 && cy >= 0 && cx < P.L && cy < P.W;
        P.costT[((size_t)m * P.NTm + t) * CTS + e] = in ? cm[(size_t)cx * P.W + cy] : (uint8_t)255;
    }
}

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
    __shared__ int s_late;      // resident kernel: thread 0 has seen the time limit pass (no more tiles are taken ahead: the next look leaves)
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
    const int wr = w >> 2, wc = w & 3;                         // the wave's 8x8 region = 2x2 patches
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
    int own_hrot = 0;                       // which part of the hints the visit in progress has loaded ahead
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
    // The decision: s_best = {priority, slot} of my best queued tile; s_gmin bit 0 = some other owner holds something, bit 1 =
    // ... something more than an ordering band below my best (which then has to wait).  From this thread's first queue word
    // and one other owner's hint.  Opens and closes with a barrier.  (No reduction tree, no same-address atomics from whole
    // waves -- either costs more than a microsecond here: a handful of lanes have a queued word, and the hints only vote.)
    auto own_decide = [&](int v0, int hint, int skip) {   // skip: the slot of the visit in progress (its turn comes again later)
        if (tid == 0) { s_best = ~0ull; s_gmin = 0; }
        lds_barrier();
        unsigned long long best = ~0ull;
#pragma unroll 1
        for (int sl = tid; sl < P.own_slots; sl += NTH) {
            const int v = (sl == tid) ? v0 : __hip_atomic_load(&own_q[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v < INFBITS && sl != skip) best = min(best, ((unsigned long long)(unsigned int)v << 32) | (unsigned int)sl);
        }
        if (best != ~0ull) atomicMin(&s_best, best);
        lds_barrier();
        const unsigned long long b = s_best;
        const bool any = hint != INFBITS;
        const bool below = any && b != ~0ull && __int_as_float((int)(b >> 32)) > __int_as_float(hint) + delta;
        const unsigned long long ba = __ballot(any), bb = __ballot(below);
        if (lane == 0 && ba) atomicOr(&s_gmin, bb ? 3 : 1);
        lds_barrier();
    };
    // An "empty" value no word has held before in this launch (the end-of-phase test relies on it): visit count x workgroup.  (A workgroup
    // that runs out of values -- 16 k of them -- leaves like one that runs out of time.)
    auto own_empty = [&]() -> int {
        const unsigned int c = (unsigned int)s_own[2]++;
        return INFBITS + 1 + (int)((c * (unsigned int)OWN_NW + blockIdx.x) % OWN_EMPTIES);
    };
    // Taking a tile (thread 0): the queue word goes from the priority it was seen with to this workgroup's mark (compare-and-swap: a mark
    // or an empty value of somebody else is never overwritten) and the tile's lock from 0 to 1 -- two atomics issued together, looked
    // at together.  With both, the tile is this workgroup's until own_release().  With the activation but not the lock (the tile is
    // being visited: the activation landed during that visit) the activation goes back into the word; with the lock but no
    // activation (the word had changed since it was looked at) the lock is released.
    auto own_take_issue = [&](int gw, int prio, int &r_old, int &r_lk) {
        int expect = prio;
        __hip_atomic_compare_exchange_strong(&P.own_prio[gw], &expect, OWN_MARK + (int)blockIdx.x, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        r_old = expect;
        r_lk = __hip_atomic_exchange(&P.own_lock[gw], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto own_take_resolve = [&](int gw, int &prio, int r_old, int r_lk) -> bool {
        bool got = r_old == prio;
        if (!got && r_lk == 0 && r_old < prio) {   // still queued, only lower meanwhile (the word was chosen from an older copy): once more
            int expect = r_old;
            got = __hip_atomic_compare_exchange_strong(&P.own_prio[gw], &expect, OWN_MARK + (int)blockIdx.x, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (got) prio = r_old;
        }
        if (got && r_lk == 0) return true;
        if (got) __hip_atomic_fetch_min(&P.own_prio[gw], prio, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // (mark -> queued again)
        else if (r_lk == 0) __hip_atomic_store(&P.own_lock[gw], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
    };
    // thread 0, after a decision: take back the mark of the visit before the last (its activations have long been performed),
    // say what this workgroup holds, mark the chosen tile.  The atomics of the take are not waited for here.
    auto own_commit = [&](unsigned long long b, bool take, bool wait, int &r_old, int &r_lk) {   // wait: that visit's activations have only just been issued
        const int own_prev = s_own[0];
        if (own_prev >= 0) {
            if (wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            int expect = OWN_MARK + (int)blockIdx.x;
            __hip_atomic_compare_exchange_strong(&P.own_prio[own_prev], &expect, own_empty(), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_own[0] = -1;
        }
        __hip_atomic_store(&P.own_min[blockIdx.x], b != ~0ull ? (int)(b >> 32) : INFBITS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (take) own_take_issue(own_base + (int)(unsigned int)b, (int)(b >> 32), r_old, r_lk);
    };
    // An idle workgroup helps out.  Ownership spreads a front over the workgroups only on average: the critical path of a 4096^2 plan
    // (tools/front_timing.py) is ~330 visits long, and on it an activation waits 45 us for its visit -- 31 us of them while its owner
    // is busy with other tiles, next to workgroups that find nothing of their own inside the band.  So a workgroup with nothing to take
    // looks at the owner that holds the smallest priority (the hints), at that owner's words, and takes its best queued tile if
    // that lies inside the ordering band -- by the same take the owner uses (own_take_issue), so a tile still has one visitor at a time.
    // All threads call; returns the word taken (its priority in s_own[3]) or -1.
    int steal_seq = 0;
    auto own_steal = [&](int hint) -> int {
        unsigned long long hk = hint != INFBITS ? (((unsigned long long)(unsigned int)hint << 32) | (unsigned int)tid) : ~0ull;
        for (int o_ = 32; o_; o_ >>= 1) hk = min(hk, (unsigned long long)__shfl_xor((long long)hk, o_));
        if (tid == 0) { s_best = ~0ull; s_gmin = -1; }
        lds_barrier();
        if (lane == 0 && hk != ~0ull) atomicMin(&s_best, hk);
        lds_barrier();
        const unsigned long long vk = s_best;
        lds_barrier();
#ifdef UFM_TIMING
        if (tid == 0) atomicAdd(&g_sdiag[2], 1ull);
#endif
        if (vk == ~0ull) return -1;
        // (whose words: not the holder of the smallest priority -- its best tile is as a rule the one it is visiting, and every idle
        //  workgroup would go for the same word -- but a different owner at every look; the smallest hint is the floor of the band)
        ++steal_seq;
        unsigned long long bb = ~0ull;         // {priority, which of the owners looked at, slot}
        int vos[UFM_STEAL_VICTIMS];
#pragma unroll
        for (int vi = 0; vi < UFM_STEAL_VICTIMS; ++vi) {
            const int vo = ((int)blockIdx.x + 1 + (int)((unsigned int)((steal_seq * UFM_STEAL_VICTIMS + vi) * 61 + (int)blockIdx.x * 17) % (unsigned int)(P.own_nw - 1))) % P.own_nw;
            vos[vi] = vo;
            const int *vq = P.own_prio + (size_t)vo * P.own_slots;
#pragma unroll 1
            for (int sl = tid; sl < P.own_slots; sl += NTH) {
                const int v = __hip_atomic_load(&vq[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int lk = __hip_atomic_load(&P.own_lock[(size_t)vo * P.own_slots + sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v < INFBITS && lk == 0) bb = min(bb, ((unsigned long long)(unsigned int)v << 32) | ((unsigned int)vi << 28) | (unsigned int)sl);
            }
        }
        for (int o_ = 32; o_; o_ >>= 1) bb = min(bb, (unsigned long long)__shfl_xor((long long)bb, o_));
        if (tid == 0) s_best = ~0ull;
        lds_barrier();
        if (lane == 0 && bb != ~0ull) atomicMin(&s_best, bb);
        lds_barrier();
        if (tid == 0) {
            const unsigned long long b2 = s_best;
#ifdef UFM_TIMING
            if (b2 != ~0ull) atomicAdd(&g_sdiag[3], 1ull);
            if (b2 != ~0ull && !(__int_as_float((int)(b2 >> 32)) > __int_as_float((int)(vk >> 32)) + delta)) atomicAdd(&g_sdiag[4], 1ull);
#endif
            if (b2 != ~0ull && !(__int_as_float((int)(b2 >> 32)) > __int_as_float((int)(vk >> 32)) + delta)) {
                int vo = vos[0];
#pragma unroll
                for (int vi = 1; vi < UFM_STEAL_VICTIMS; ++vi) if ((int)(((unsigned int)b2 >> 28) & 15u) == vi) vo = vos[vi];
                const int gw = vo * P.own_slots + (int)((unsigned int)b2 & 0x0FFFFFFFu);
                int pr = (int)(b2 >> 32);
                int r_old, r_lk;
                own_take_issue(gw, pr, r_old, r_lk);
                if (own_take_resolve(gw, pr, r_old, r_lk)) {
                    s_own[3] = pr; s_gmin = gw;
#ifdef UFM_TIMING
                    atomicAdd(&g_sdiag[5], 1ull);
#endif
                    __hip_atomic_fetch_min(&P.own_min[blockIdx.x], pr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // what this workgroup holds now
                }
            }
        }
        __syncthreads();
        return s_gmin;
    };
    for (int i = blockIdx.x;; i += gridDim.x) {
        int gt_own = -1;
        if constexpr (OWN) {
            if (own_next >= 0 && s_own[3] >= INFBITS) own_next = -1;   // chosen ahead from the older copy of the words, but the take failed
            if (tid == 0 && s_own[1] >= 0) { s_own[0] = s_own[1]; s_own[1] = -1; }   // (at most one mark waits: own_commit ran since)
            while (own_next < 0) {                             // nothing was taken ahead: look, wait, look again
                __syncthreads();                               // LDS of the previous visit / round is free
                int hint = INFBITS;
                if (tid < P.own_nw && tid != (int)blockIdx.x) hint = __hip_atomic_load(&P.own_min[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int v0 = tid < P.own_slots ? __hip_atomic_load(&own_q[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : INFBITS;
                const int aborted = tid == 0 ? __hip_atomic_load(&P.ctr->own_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                own_decide(v0, hint, -1);
                const unsigned long long b = s_best;
                const int votes = s_gmin;
                const bool have = b != ~0ull;
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
#pragma unroll 1
                        for (int e = tid; e < total; e += NTH) {
                            const int v = __hip_atomic_load(&P.own_prio[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (v < INFBITS || v >= OWN_MARK) mine_ok = false;
                            acc += ((unsigned long long)(unsigned int)v + 1ull) * (0x9E3779B97F4A7C15ull + 2ull * (unsigned long long)e);
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
                    const bool late = !stop && (aborted || wall_clock64() - own_t0 > P.own_limit || s_own[2] > 16000);
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
                    if ((votes & 1) && !(P.own_flags & 32)) got = own_steal(hint);
                    if (got >= 0) own_next = __builtin_amdgcn_readfirstlane(got);
                    else __builtin_amdgcn_s_sleep(UFM_LOOK_SLEEP);
                }
            }
            if (own_next < 0) break;
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
        const int c0 = ct[tid < CN ? tid : 0];
        constexpr bool BPRAISE = MODE == MODE_RAISE && !is_dfm<ALGO>;   // invalidation along the stored back-pointers
        const int bp0 = BPRAISE ? P.bp[(size_t)gt * TT + (io_on ? tid : 0)] : BP_NONE;
        const int goal_x = P.goal[2 * m], goal_y = P.goal[2 * m + 1];   // (with the rest: read after the barrier they cost two more round trips)
        // resident kernel: the values of the map's start elements, one per lane (for the end condition below)
        const int own_sa = (OWN && focused && tid < 4 && m < 64) ? s_se[m * 4 + tid] : -1;
        const float own_sg = own_sa >= 0 ? ld_f<OWN>(&P.G[own_sa]) : INFINITY;
        if (tid == 0) {
            const int seen = atomicAdd(&P.touched[gt], 1);   // visits of this tile in the current step
            const int first = seen == 0;
            if (first) P.tlist[atomicAdd(&P.ctr->tcount, 1)] = gt;
            s_misc[0] = first; s_misc[1] = seen; s_misc[2] = 0; s_misc[3] = 0; s_idle = 0; s_giveup = 0;
        }
        if (tid < NWV) s_wake[tid] = (1 << PPWK) - 1;
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
            // DFM only: the float fixed point of the upwind quadratic is not unique;
            // neighbouring tiles can push each other's border values up one ulp at a time for tens of
            // thousands of launches.  An INCREASE of at most 4 ulp (a rounding-level correction, never
            // new information) is stored but does not wake the neighbour; after 24 visits of a tile in
            // one step the same holds for decreases.  Well inside DFM's 1e-6 tolerance.
            bool significant = true;
            if (is_dfm<ALGO> && MODE == MODE_LOWER && gf < INFINITY && gl0 < INFINITY) {
                const int du = __float_as_int(gf) - __float_as_int(gl0);
                // (level 1: only in a tile that keeps coming back -- the rises of its operator are corrections of
                //  values latched from transient neighbours and have to travel)
                if (ALGO == UFM_ALGO_DFM ? (du > 0 ? du <= 4 : (s_misc[1] > UFM_DFM_QUIET_VISITS && du >= -4))
                                         : (s_misc[1] > UFM_DFM1_QUIET_VISITS && du >= -4 && du <= 4)) significant = false;
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
            if constexpr (SKEW) { pc_ = j ? (w < 2 ? 3 : (w >> 1)) : (w < 4 ? 0 : (w >> 1) - 1); pr_ = (w - 2 * pc_) & 7; }
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
        // the wave's early stores have to have arrived before a neighbour is told (as in the write-back: stores, wait, queue words)
        auto ew_flush = [&]() {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            int t_ = tid;
            asm volatile("" : "+v"(t_));
            const int l_ = t_ & 63;
            if (l_ < 9 && l_ != 4) {
                int *bm = s_emin + (t_ >> 6) * 9;
                const int v = __hip_atomic_exchange(&bm[l_], INFBITS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (v != INFBITS) {
                    const int ntx = tx + l_ / 3 - 1, nty = ty + l_ % 3 - 1;
                    if (ntx >= 0 && ntx < P.TX && nty >= 0 && nty < P.TY) own_push(P, m * P.NTm + ntx * P.TY + nty, v, gt);
                }
            }
            ew_pend = 0;
        };
        // In-visit refresh, the other half of the early hand-off: a neighbour that hands its border out while this tile is being
        // visited lowers this tile's queue word (MARK -> a priority).  Idle waves look at the word now and then (a load straight into
        // LDS, nobody waits for it); the one that finds it changed takes the activation back (exchange -> MARK, waited for: its values
        // are then visible), reloads the ring record into the LDS halo and wakes the border patches on the sides that changed --
        // the visit carries on with the new inputs instead of ending, being written back, queued, taken and staged again.
        auto halo_poll = [&]() {
            typedef __attribute__((address_space(3))) void *lds_ptr;
            typedef const __attribute__((address_space(1))) void *glb_ptr;
            if (lane == 0) __builtin_amdgcn_global_load_lds((glb_ptr)(P.own_prio + own_slot_now), (lds_ptr)s_qw, 4, 0, 16);
        };
        auto halo_refresh = [&]() -> bool {      // true: this wave did a refresh (it has left the idle count meanwhile)
            int seen = __hip_atomic_load(&s_qw[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (seen >= INFBITS) return false;           // (still my mark)
            int mine = 0;
            if (lane == 0) mine = atomicCAS(&s_qw[0], seen, OWN_MARK) == seen;
            if (!__builtin_amdgcn_readfirstlane(mine)) return false;
            if (lane == 0) { atomicSub(&s_idle, 1); s_qw[1] += 1; }
#ifdef UFM_TIMING
            if (lane == 0 && gt < TILE_DIAG_MAX) atomicExch(&g_tile[2][gt], 0xFFFFFFFFu);
#endif
            int was = OWN_MARK;
            if (lane == 0) was = __hip_atomic_exchange(&P.own_prio[own_slot_now], OWN_MARK + (int)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            was = __builtin_amdgcn_readfirstlane(was);
            UFM_STRICT_ACQUIRE();
            if (was < INFBITS) {                 // (an activation: its values were stored before it was queued)
                if (lane == 0) atomicMin(&s_emin[w * 9 + 4], was);   // (its priority counts for the tile's own, should the visit end at the sweep cap)
                int t_ = tid;
                asm volatile("" : "+v"(t_));
                const int l_ = t_ & 63;
                const float h0 = ld_f<true>(&ring[l_]);
                const float h1 = ld_f<true>(&ring[64 + (l_ & 3)]);
                int sides = 0;                   // 1 top, 2 bottom, 4 left, 8 right
                auto put = [&](int ht, float v) {
                    int hr, hc, sd;
                    if (ht < T) { hr = -1; hc = ht; sd = 1; }
                    else if (ht < 2 * T) { hr = T; hc = ht - T; sd = 2; }
                    else if (ht < 3 * T) { hr = ht - 2 * T; hc = -1; sd = 4; }
                    else if (ht < 4 * T) { hr = ht - 3 * T; hc = T; sd = 8; }
                    else { hr = (ht & 2) ? T : -1; hc = (ht & 1) ? T : -1; sd = ((ht & 2) ? 2 : 1) | ((ht & 1) ? 8 : 4); }
                    float *d = &Gs[(hr + 1) * GP + hc + 1];
                    if (*d != v) { *d = v; sides |= sd; }
                };
                put(l_, h0);
                if (l_ < 4) put(64 + l_, h1);
                for (int o_ = 32; o_; o_ >>= 1) sides |= __shfl_xor(sides, o_);
                UFM_SWEEP_FENCE();               // values before wake bits
                if (sides && l_ < PT * PT) {
                    const int pr_ = l_ / PT, pc_ = l_ % PT;
                    const bool hit = ((sides & 1) && pr_ == 0) || ((sides & 2) && pr_ == PT - 1) || ((sides & 4) && pc_ == 0) || ((sides & 8) && pc_ == PT - 1);
                    if (hit) {
                        int wv, bit;
                        if constexpr (SKEW) { wv = (pr_ + 2 * pc_) & 7; bit = (pc_ == (wv < 4 ? 0 : (wv >> 1) - 1)) ? 1 : 2; }
                        else { wv = (pr_ / PR) * 4 + (pc_ / PR); bit = 1 << ((pr_ % PR) * PR + (pc_ % PR)); }
                        __hip_atomic_fetch_or(&s_wake[wv], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
            return true;
        };
        const bool lax = is_dfm<ALGO> && (s_misc[1] > (ALGO == ALGO_DFM1 ? UFM_DFM1_LAX_VISITS : UFM_DFM_LAX_VISITS));
        bool conv = false;
#ifdef UFM_TIMING
        const bool wtrace_on = DYN && MODE == MODE_LOWER && k == UFM_TRACE_K0 && i == 0;
        UFM_WREC(0, s_misc[1]);
#endif
        for (;;) {
            int bits = 0;
            if (lane == 0) bits = atomicExch(&s_wake[w], 0);
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
                        __hip_atomic_fetch_or(&s_wake[w], 1 << j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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
                        if constexpr (SKEW) { pc_ = j ? (w_ < 2 ? 3 : (w_ >> 1)) : (w_ < 4 ? 0 : (w_ >> 1) - 1); pr_ = (w_ - 2 * pc_) & 7; }
                        else { pr_ = (w_ >> 2) * PR + j / PR; pc_ = (w_ & 3) * PR + j % PR; }
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
                if (bits && lane == 0) __hip_atomic_fetch_or(&s_wake[w], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (lane == 0) atomicAdd(&s_idle, 1);
                UFM_WREC(3, 0);
                int polls = 0;
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
                    if (__hip_atomic_load(&s_wake[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) {
                        if (lane == 0) atomicSub(&s_idle, 1);
                        UFM_WREC(4, 0);
                        break;
                    }
                }
                if (!vote) continue;
            } else if (bits && lane == 0) {
                __hip_atomic_fetch_or(&s_wake[w], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // put back what was taken
            }
            // vote: everybody arrives first, then the wake bits are stable
            UFM_WREC(5, 0);
            __syncthreads();
            const int work = __syncthreads_or(__hip_atomic_load(&s_wake[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0);
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
                       (own_slot_now >= own_base && own_slot_now < own_base + P.own_slots) ? own_slot_now - own_base : -1);
            const unsigned long long b = s_best;
            const bool take = b != ~0ull && !(s_gmin & 2) && !(P.own_flags & 1) && !s_late;
            if (tid == 0) {
                own_commit(b, take, false, own_ro, own_rl);
                if (wall_clock64() - own_t0 > P.own_limit || s_own[2] > 16000) s_late = 1;   // (a workgroup that is never out of work looks at the clock here)
            }
            own_next = __builtin_amdgcn_readfirstlane(take ? own_base + (int)(unsigned int)b : -1);
        }
        // write back what changed; note which neighbours saw their halo change
        // (this thread's row and column, made opaque once per visit: the compiler otherwise computes the two dozen LDS addresses of
        //  the tests below ahead of the tile loop and carries them through the sweeps -- registers the sweep loop needs)
        int wb_r = io_r, wb_c = io_c;
        asm volatile("" : "+v"(wb_r), "+v"(wb_c));
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
        }
#ifdef UFM_TIMING
        if (MODE == MODE_LOWER) {
            const int dbg_ninf1 = __syncthreads_count(io_on && gf == INFINITY);
            if (tid == 0) {
                const unsigned long long tk3 = wall_clock64();
                if (OWN && gt < TILE_DIAG_MAX && (!EARLY || (s_qw[1] & 0x10000))) g_tile[1][gt] = (unsigned int)(tk3 - g_tile_t0);
                if (OWN && s_misc_vi < VIS_DIAG_MAX) g_vis[s_misc_vi][2] = (unsigned int)(tk3 - g_tile_t0);
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

// ---- small control kernels -----------------------------------------------------
__global__ void k_fill(float *p, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

// Graph::update (Graph.cpp:36-51) on the device: overwrite the rectangle, remember which cells
// changed (one byte per patch cell in `pmask`).
__device__ __forceinline__ void patch_apply(const DevParams &P, int m, const uint8_t *patch, uint8_t *pmask, int x, int y, int w, int h, int e) {
    if (e >= w * h) return;
    const int i = e / w, j = e - i * w;
    uint8_t *cm = P.cost + (size_t)m * P.cstride;
    const size_t ci = (size_t)(x + i) * P.W + (y + j);
    const uint8_t nv = patch[e];
    const uint8_t ch = cm[ci] != nv;
    pmask[e] = ch;
    if (ch) { cm[ci] = nv; cost_window_store(P, m, x + i, y + j, nv); }
}
__global__ void k_patch_apply(DevParams P, int m, const uint8_t *patch, uint8_t *pmask, int x, int y, int w, int h) {
    patch_apply(P, m, patch, pmask, x, y, w, h, blockIdx.x * blockDim.x + threadIdx.x);
}
// Seeding of update(): the corner nodes of the changed cells (FD impl:127-136, Cell.cpp:48-60) or
// the changed cells themselves (DFM impl:106-112).  One thread per element of the patch's
// element rectangle, so every element has one owner: plain byte marks, no atomics per element;
// the counter and the tile seeds are aggregated per wave.
template <bool NODES>   // every lane of a wave must call (ballots)
__device__ __forceinline__ void patch_seed(const DevParams &P, int m, const uint8_t *pmask, int x, int y, int w, int h, int e) {
    const int ew = NODES ? w + 1 : w, eh = NODES ? h + 1 : h;
    bool hit = false;
    int gt = -1;
    if (e < ew * eh) {
        const int i = e / ew, j = e - i * ew;
        bool ch;
        if (NODES) {   // node (x+i, y+j) touches patch cells (i-1..i, j-1..j)
            ch = (i > 0 && j > 0 && pmask[(i - 1) * w + j - 1]) || (i > 0 && j < w && pmask[(i - 1) * w + j]) ||
                 (i < h && j > 0 && pmask[i * w + j - 1]) || (i < h && j < w && pmask[i * w + j]);
        } else {
            ch = pmask[i * w + j];
        }
        if (ch) {
            const int ex = x + i, ey = y + j;
            uint8_t *mk = P.mark + (size_t)m * P.mstride + (size_t)ex * P.EY + ey;
            hit = (*mk == 0);
            *mk = 1;
            gt = m * P.NTm + (ex / T) * P.TY + (ey / T);
        }
    }
    const unsigned long long hm = __ballot(hit);
    const int lane = threadIdx.x & 63;
    if (hm && lane == 0) atomicAdd(&P.num_updated[m], (unsigned int)__popcll(hm));
    unsigned long long todo = __ballot(gt >= 0);
    while (todo) {                       // one seed attempt per distinct tile per wave
        const int leader = __ffsll((long long)todo) - 1;
        const int t = __shfl(gt, leader);
        if (lane == leader && atomicExch(&P.sflag[t], 1) == 0) P.slist[atomicAdd(&P.ctr->scount, 1)] = t;
        todo &= ~__ballot(gt == t);
    }
}
template <bool NODES>
__global__ void k_patch_seed(DevParams P, int m, const uint8_t *pmask, int x, int y, int w, int h) {
    patch_seed<NODES>(P, m, pmask, x, y, w, h, blockIdx.x * blockDim.x + threadIdx.x);
}
// a patch of at most 64 x 64 cells: Graph::update and the seeding of update() in one workgroup
// (each separate launch costs ~5 us of dispatch latency)
template <bool NODES>
__global__ __launch_bounds__(1024) void k_patch_small(DevParams P, int m, const uint8_t *patch, uint8_t *pmask, int x, int y, int w, int h) {
    for (int e = threadIdx.x; e < w * h; e += blockDim.x) patch_apply(P, m, patch, pmask, x, y, w, h, e);
    __syncthreads();
    const int ne = NODES ? (w + 1) * (h + 1) : w * h;
    for (int base = 0; base < ne; base += blockDim.x) patch_seed<NODES>(P, m, pmask, x, y, w, h, base + threadIdx.x);
}
// Small patches of several maps, handed over as device pointers and held back until the step that consumes them: one
// launch, one workgroup per patch (a batch's eight patch kernels in a row were 60 us of every replan round).
constexpr int PATCH_MULTI = 16;
struct PatchMulti { int n; int rect[PATCH_MULTI][5]; const uint8_t *ptr[PATCH_MULTI]; };
template <bool NODES>
__global__ __launch_bounds__(1024) void k_patch_multi(DevParams P, PatchMulti a, uint8_t *pmask) {
    const int *q = a.rect[blockIdx.x];
    const int m = q[0], x = q[1], y = q[2], w = q[3], h = q[4];
    const uint8_t *patch = a.ptr[blockIdx.x];
    uint8_t *pm = pmask + (size_t)blockIdx.x * 4096;
    for (int e = threadIdx.x; e < w * h; e += blockDim.x) patch_apply(P, m, patch, pm, x, y, w, h, e);
    __syncthreads();
    const int ne = NODES ? (w + 1) * (h + 1) : w * h;
    for (int base = 0; base < ne; base += blockDim.x) patch_seed<NODES>(P, m, pm, x, y, w, h, base + threadIdx.x);
}
__device__ __forceinline__ void clear_mark(const DevParams &P, int m, int x, int y, int w, int h, int e) {
    const int r = e / (w + 1), c = e - r * (w + 1);
    if (r > h) return;
    const int ex = x + r, ey = y + c;
    if (ex >= P.EX || ey >= P.EY) return;
    P.mark[(size_t)m * P.mstride + (size_t)ex * P.EY + ey] = 0;
}
__global__ void k_clear_marks(DevParams P, int m, int x, int y, int w, int h) {
    clear_mark(P, m, x, y, w, h, blockIdx.x * blockDim.x + threadIdx.x);
}
// pending seeds of consuming maps -> candidate list of launch k; others stay pending. One block.
// (device bodies: run by ONE workgroup; s_keep is a shared counter of the calling kernel)
__device__ void seeds_to_active(const DevParams &P, int qz, int k, int &s_keep) {
    if (threadIdx.x == 0) s_keep = 0;
    __syncthreads();
    const int n = P.ctr->scount;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int gt = P.slist[i];
        if (P.consume[gt / P.NTm]) { P.sflag[gt] = 0; activate(P, qz, k, gt, 0); }
        else P.slist2[atomicAdd(&s_keep, 1)] = gt;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < s_keep; i += blockDim.x) P.slist[i] = P.slist2[i];
    if (threadIdx.x == 0) P.ctr->scount = s_keep;
    __syncthreads();
}
__global__ void k_seeds_to_active(DevParams P, int qz, int k) {
    __shared__ int s_keep;
    seeds_to_active(P, qz, k, s_keep);
}
__global__ void k_touched_to_active(DevParams P, int qz, int k) {
    const int n = P.ctr->tcount;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) activate(P, qz, k, P.tlist[i], 0);
}
__global__ void k_activate_list(DevParams P, int qz, int k, const int *tiles, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) activate(P, qz, k, tiles[i], 0);
}
// Phase start: parked tiles whose priority is now inside the bound go back to the candidate list
// of launch k; the others stay parked.  One workgroup.
__device__ void unpark(const DevParams &P, int qz, int k, float rbound, int &s_keep) {
    if (threadIdx.x == 0) s_keep = 0;
    __syncthreads();
    const int n = P.ctr->npark[qz];
    int *list = P.park + (size_t)(qz * 2) * P.NT, *tmp = list + P.NT;
    const float rb = (rbound < 0.0f) ? P.ctr->rbound : rbound;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int gt = list[i];
        const int pbits = P.pprio[qz * P.NT + gt];
        const int m = gt / P.NTm, t = gt - m * P.NTm;
        bool in;
        if (qz == Q_LOWER) {
            const float B = P.dyn->focused ? start_bound(P, m) : INFINITY;
            const float hd = P.dyn->focused ? tile_heuristic(P, m, t / P.TY, t % P.TY) : 0.0f;
            in = (__int_as_float(pbits) + hd < B || B == INFINITY);
        } else {
            in = !(__int_as_float(pbits) > rb);
        }
        if (in) { P.pflag[qz * P.NT + gt] = 0; P.pprio[qz * P.NT + gt] = INFBITS; activate(P, qz, k, gt, pbits); }
        else tmp[atomicAdd(&s_keep, 1)] = gt;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < s_keep; i += blockDim.x) list[i] = tmp[i];
    if (threadIdx.x == 0) P.ctr->npark[qz] = s_keep;
    __syncthreads();
}
// The resident lowering kernel stands in for launch k of the lowering queue and everything after it: this kernel hands
// it the entries of list k % 3 and does the list bookkeeping a launch does for its successors (k_relax, block 0) ...
__global__ void k_own_import(DevParams P, int k) {
#ifdef UFM_TIMING
    if (blockIdx.x == 0 && threadIdx.x == 0) g_tile_t0 = wall_clock64();
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < TILE_DIAG_MAX; t += gridDim.x * blockDim.x) {
        g_tile[0][t] = 0xFFFFFFFFu; g_tile[1][t] = 0u; g_tile[2][t] = 0xFFFFFFFFu; g_tile[3][t] = 0u; g_tile[4][t] = 0u;
        g_push64[t] = ~0ull;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) g_nvis = 0u;
#endif
    const int r = k % 3, rz = (k + 2) % 3;
    const int n = P.ctr->cnt[Q_LOWER][r];
    const int *cand = P.cand + (size_t)(Q_LOWER * 3 + r) * P.NT;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int gt = cand[i];
        own_push(P, gt, min(prio_read(P, Q_LOWER, k, gt), INFBITS - 1));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        P.ctr->cnt[Q_LOWER][rz] = 0; P.ctr->rel[Q_LOWER][rz] = 0; P.ctr->lmin[Q_LOWER][rz] = INFBITS;
        P.ctr->nready[(k + 1) & 1] = 0; P.ctr->nshort[(k + 1) & 1] = 0; P.ctr->rcursor[(k + 1) & 1] = 0;
        P.ctr->rel[Q_LOWER][r] = n;
        if (n) P.ctr->last_work[Q_LOWER] = k;
        P.ctr->own_vis0 = P.ctr->tile_visits;
        P.ctr->own_abort = 0;
    }
}
// ... and this one gives what it left queued -- tiles beyond the start's key; everything, had it run into its time
// limit -- back to the launch chain as the list of launch k1 = k + 1, and leaves all words empty.
__global__ void k_own_export(DevParams P, int k1) {
    const int total = P.own_nw * P.own_slots;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int v = P.own_prio[e];
        P.own_lock[e] = 0;
        if (v == INFBITS) continue;
        P.own_prio[e] = INFBITS;
        if (v < INFBITS) {
            int m, tx, ty;
            const int o = e / P.own_slots, gt = own_tile(P, o, e - o * P.own_slots, m, tx, ty);
            if (gt >= 0) activate(P, Q_LOWER, k1, gt, v);
        }
    }
    if (blockIdx.x == 0) for (int i = threadIdx.x; i < OWN_NW; i += blockDim.x) P.own_min[i] = INFBITS;
    if (blockIdx.x == 0 && threadIdx.x == 0) P.ctr->own_vis1 = P.ctr->tile_visits;
}
__global__ void k_unpark(DevParams P, int qz, int k, float rbound) {
    __shared__ int s_keep;
    unpark(P, qz, k, rbound, s_keep);
}
// smallest priority waiting in queue qz (list of launch k); one workgroup
__global__ void k_queue_min(DevParams P, int qz, int k) {
    __shared__ int s_m;
    if (threadIdx.x == 0) s_m = INFBITS;
    __syncthreads();
    const int n = P.ctr->cnt[qz][k % 3];
    int lmin = INFBITS;
    for (int i = threadIdx.x; i < n; i += blockDim.x)
        lmin = min(lmin, prio_read(P, qz, k, P.cand[(size_t)(qz * 3 + k % 3) * P.NT + i]));
    {
        const int np = P.ctr->npark[qz];
        for (int i = threadIdx.x; i < np; i += blockDim.x) lmin = min(lmin, P.pprio[qz * P.NT + P.park[(size_t)(qz * 2) * P.NT + i]]);
    }
    if (lmin != INFBITS) atomicMin(&s_m, lmin);
    __syncthreads();
    if (threadIdx.x == 0) P.ctr->qmin[qz] = s_m;
}
// invalidation bound for this step: the current start key plus one ordering band
// start of a step with a single map: counters, start elements and the consume flag in one launch
struct StepBegin { int start[4]; int consume; int clear_lmax; float sx, sy; };
__device__ __forceinline__ void step_begin(const DevParams &P, const StepBegin &a) {
    const int t = threadIdx.x;
    if (t == 0) {
        P.ctr->tcount = 0; P.ctr->expanded = 0; P.ctr->tile_visits = 0; P.ctr->tile_iters = 0; P.ctr->elem_evals = 0;
        P.ctr->raise_visits = 0;
        P.consume[0] = a.consume;
    }
    if (t < 4) P.start[t] = a.start[t];
    if (t == 0) { P.spos[0] = a.sx; P.spos[1] = a.sy; }
    if (a.clear_lmax) for (int i = t; i < LMAX; i += blockDim.x) P.lmax[i] = 0;
}
__global__ void k_step_begin(DevParams P, StepBegin a) { step_begin(P, a); }
__device__ __forceinline__ void prepare_bound(const DevParams &P, float margin) {
    float b = 0.0f;
    for (int m = 0; m < P.nmaps; ++m) b = fmaxf(b, start_bound(P, m));
    P.ctr->rbound = P.dyn->focused ? b + margin : INFINITY;
    P.ctr->done = 0;
}
__global__ void k_prepare_bound(DevParams P, float margin) {
    if (threadIdx.x || blockIdx.x) return;
    prepare_bound(P, margin);
}
// After a blind batch of invalidation + lowering launches: are both queues drained below the
// start's key, and did the invalidation bound reach the key the start ended up with?
// kr / kl: index of the next launch of the raise / lower queue.  One workgroup.
// (s_m, s_done: shared words of the calling kernel; `record`: this workgroup writes the verdict)
__device__ int replan_check(const DevParams &P, int kr, int kl, float margin, bool record, int &s_m, int &s_done) {
    if (threadIdx.x == 0) s_m = INFBITS;
    __syncthreads();
    const int n = P.ctr->cnt[Q_RAISE][kr % 3];
    int lmin = INFBITS;
    for (int i = threadIdx.x; i < n; i += blockDim.x)
        lmin = min(lmin, prio_read(P, Q_RAISE, kr, P.cand[(size_t)(Q_RAISE * 3 + kr % 3) * P.NT + i]));
    {   // ... and the parked invalidations
        const int np = P.ctr->npark[Q_RAISE];
        for (int i = threadIdx.x; i < np; i += blockDim.x) lmin = min(lmin, P.pprio[Q_RAISE * P.NT + P.park[(size_t)(Q_RAISE * 2) * P.NT + i]]);
    }
    if (lmin != INFBITS) atomicMin(&s_m, lmin);
    __syncthreads();
    if (threadIdx.x == 0) {
        float bnew = 0.0f;
        for (int m = 0; m < P.nmaps; ++m) bnew = fmaxf(bnew, start_bound(P, m));
        const bool raise_done = P.ctr->cnt[Q_RAISE][kr % 3] == 0 || P.ctr->rel[Q_RAISE][(kr + 2) % 3] == 0;
        const bool lower_done = P.ctr->cnt[Q_LOWER][kl % 3] == 0 || P.ctr->rel[Q_LOWER][(kl + 2) % 3] == 0;
        const bool again = P.dyn->focused && (__int_as_float(s_m) < bnew);
        const int done = (raise_done && lower_done && !again) ? 1 : 0;
        s_done = done;
        if (record) {
            if (again) P.ctr->rbound = fmaxf(bnew, P.ctr->rbound) + margin;
            P.ctr->qmin[Q_RAISE] = s_m;
            P.ctr->done = done;
            unsigned int upd = 0;
            for (int m = 0; m < P.nmaps; ++m) if (P.consume[m]) { upd += P.num_updated[m]; P.num_updated[m] = 0; }
            P.ctr->updated = upd;
        }
    }
    __syncthreads();
    return s_done;
}
__global__ void k_check(DevParams P, int kr, int kl, float margin) {
    __shared__ int s_m, s_done;
    replan_check(P, kr, kl, margin, true, s_m, s_done);
}
__global__ void k_set_dyn(DevDyn *dst, DevDyn v) { *dst = v; }
__global__ void k_start_bound(DevParams P) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < P.nmaps) P.bnd[m] = start_bound(P, m);
}
// window [x0, x0+nx) x [y0, y0+ny) of map m's field, dense row-major (ufm_read_field)
__global__ void k_gather_field(DevParams P, int m, int x0, int y0, int nx, int ny, float *out) {
    const size_t n = (size_t)nx * ny;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / ny), c = (int)(i - (size_t)r * ny);
        out[i] = P.G[gaddr(P, m, x0 + r, y0 + c)];
    }
}
// Self-check of the layout's redundancy (ufm_check_layout): every ring entry must equal the border
// value of the neighbour it copies (+inf where there is no neighbour), every cost-window byte the
// raster cell it copies.  out[0] / out[1]: mismatching ring entries / window bytes.
__global__ void k_check_layout(DevParams P, unsigned long long *out) {
    const int crows = P.cells ? T : T + 1, off = P.cells ? 0 : 1;
    unsigned long long bad_ring = 0, bad_cost = 0;
    const size_t nr = (size_t)P.NT * (4 * T + 4);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nr; i += (size_t)gridDim.x * blockDim.x) {
        const int gt = (int)(i / (4 * T + 4)), h = (int)(i - (size_t)gt * (4 * T + 4));
        const int m = gt / P.NTm, t = gt - m * P.NTm, tx = t / P.TY, ty = t - tx * P.TY;
        int hr, hc;   // halo position relative to the tile, as k_relax stages it
        if (h < T) { hr = -1; hc = h; }
        else if (h < 2 * T) { hr = T; hc = h - T; }
        else if (h < 3 * T) { hr = h - 2 * T; hc = -1; }
        else if (h < 4 * T) { hr = h - 3 * T; hc = T; }
        else { hr = ((h - 4 * T) & 2) ? T : -1; hc = ((h - 4 * T) & 1) ? T : -1; }
        const int x = tx * T + hr, y = ty * T + hc;
        const bool in = x >= 0 && y >= 0 && x < P.TX * T && y < P.TY * T;
        const float want = in ? P.G[gaddr(P, m, x, y)] : INFINITY;
        const float have = P.ring[(size_t)gt * RING + h];
        if (__float_as_int(want) != __float_as_int(have)) ++bad_ring;
    }
    const size_t nc = (size_t)P.NT * crows * crows;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nc; i += (size_t)gridDim.x * blockDim.x) {
        const int gt = (int)(i / (crows * crows)), e = (int)(i - (size_t)gt * crows * crows);
        const int m = gt / P.NTm, t = gt - m * P.NTm, tx = t / P.TY, ty = t - tx * P.TY;
        const int cx = tx * T + e / crows - off, cy = ty * T + e % crows - off;
        if (cx < 0 || cy < 0 || cx >= P.L || cy >= P.W) continue;
        if (P.costT[(size_t)gt * CTS + e] != P.cost[(size_t)m * P.cstride + (size_t)cx * P.W + cy]) ++bad_cost;
    }
    if (bad_ring) atomicAdd(&out[0], bad_ring);
    if (bad_cost) atomicAdd(&out[1], bad_cost);
}
// Self-check of the stored back-pointers (ufm_check_info), node planners: every element that holds a value (but the goal) must name a parent
// triangle, that triangle must give the element's value when it is evaluated on the field as it stands -- bit for bit, with the operator the
// sweeps use -- and the dep bits must say which vertices that evaluation leans on.  The invalidation follows these bytes blindly (ufm_region.h),
// so this is the invariant it rests on.  out[0]: elements with a value, out[1]: without a parent, out[2]: whose parent gives a LARGER value
// (an unsupported element) although the element lies below its map's start key: never; out[3]: whose dep bits differ from the case the evaluation
// takes; out[4]: whose parent gives a smaller value (an element waiting to be lowered: beyond the start's key in a focused search, nowhere
// otherwise); out[5]: unsupported elements at or beyond the start's key (invalidations a focused search keeps queued, like the reference's
// under-consistent queue entries beyond its end condition).
template <int ALGO>
__global__ void k_check_bp(DevParams P, unsigned long long *out) {
    unsigned long long n_val = 0, n_none = 0, n_bad = 0, n_dep = 0, n_low = 0, n_parked = 0;
    const int thr = P.dyn->thr;
    const size_t n = (size_t)P.nmaps * P.EX * P.EY;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / ((size_t)P.EX * P.EY)), e = (int)(i - (size_t)m * P.EX * P.EY), x = e / P.EY, y = e - x * P.EY;
        const float g = P.G[gaddr(P, m, x, y)];
        if (!(g < INFINITY) || (x == P.goal[2 * m] && y == P.goal[2 * m + 1])) continue;
        ++n_val;
        const int b = P.bp[gaddr(P, m, x, y)];
        if (b == BP_NONE) { ++n_none; continue; }
        const int q = (b >> 3) & 3, h = (b >> 2) & 1, dx = (q & 2) ? 1 : -1, dy = (q & 1) ? 1 : -1;
        auto val = [&](int ex, int ey) { return (ex >= 0 && ey >= 0 && ex < P.EX && ey < P.EY) ? P.G[gaddr(P, m, ex, ey)] : INFINITY; };
        auto cst = [&](int cx, int cy) {
            if (cx < 0 || cy < 0 || cx >= P.L || cy >= P.W) return INFINITY;
            const int c = P.cost[(size_t)m * P.cstride + (size_t)cx * P.W + cy];
            return c >= thr ? INFINITY : (float)c;
        };
        const int cx = x - 1 + (q >> 1), cy = y - 1 + (q & 1);                  // the triangle's cell
        const float c = cst(cx, cy), g1 = h ? val(x, y + dy) : val(x + dx, y), g2 = val(x + dx, y + dy);
        float r; int dep;
        if constexpr (ALGO == UFM_ALGO_SG) { CellSG k; k.set(c); r = tri_sg(g1, g2, k); dep = dep_sg(g1, g2, k); }
        else {
            const float bb = h ? cst(x - 1 + (1 - (q >> 1)), cy) : cst(cx, y - 1 + (1 - (q & 1)));   // the cell across the edge s-p1
            CellFD k{c, c * c, c * SQRT2F}; TriFD t; t.set(c, bb);
            r = tri_fd(g1, g2, k, t); dep = dep_fd(g1, g2, k, t);
        }
        if (r > g || r != r) {
            const float B = P.dyn->focused ? start_bound(P, m) : INFINITY;
            if (g + tile_heuristic(P, m, x / T, y / T) < B || B == INFINITY) ++n_bad; else ++n_parked;
        }
        else if (r < g) ++n_low;
        else if (dep != (b & 3)) ++n_dep;
    }
    if (n_val) atomicAdd(&out[0], n_val);
    if (n_none) atomicAdd(&out[1], n_none);
    if (n_bad) atomicAdd(&out[2], n_bad);
    if (n_dep) atomicAdd(&out[3], n_dep);
    if (n_low) atomicAdd(&out[4], n_low);
    if (n_parked) atomicAdd(&out[5], n_parked);
}
// mean traversable cost of a raster (sets the default ordering band)
__global__ void k_cost_stats(const uint8_t *cm, size_t n, int thr, unsigned long long *out) {
    unsigned long long s = 0, c = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int v = cm[i];
        if (v < thr) { s += v; ++c; }
    }
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_down(s, o); c += __shfl_down(c, o); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], s); atomicAdd(&out[1], c); }
}
// count elements whose G differs from the snapshot taken at first touch; release the tiles
__device__ __forceinline__ void finalize_tiles(const DevParams &P) {
    const int n = P.ctr->tcount;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const int gt = P.tlist[i];
        const size_t gidx = (size_t)gt * TT + threadIdx.x;
        const int diff = (threadIdx.x < T * T) && (P.fresh[gt] ? (P.G[gidx] != INFINITY) : (P.G[gidx] != P.Gprev[gidx]));
        const int c = __syncthreads_count(diff);
        if (threadIdx.x == 0) {
            if (c) atomicAdd(&P.ctr->expanded, (unsigned long long)c);
            P.touched[gt] = 0;
        }
    }
}
// The same for long touched lists (the end of a plan: every tile of the map): one wave per tile, no
// workgroup barrier, the count summed in registers -- the block-per-tile loop above took 0.8 ms for
// the 65 k tiles of a 4096^2 plan (128 barrier-separated iterations per block), this takes ~0.1 ms.
__global__ __launch_bounds__(256) void k_finalize(DevParams P, int only_if_done) {
    if (only_if_done && !P.ctr->done) return;
    const int n = P.ctr->tcount;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    unsigned long long total = 0;
    for (int i = wave; i < n; i += nwaves) {
        const int gt = P.tlist[i];
        const float *g = P.G + (size_t)gt * TT, *g0 = P.Gprev + (size_t)gt * TT;
        int c = 0;
        if (P.fresh[gt]) {
#pragma unroll
            for (int e = lane; e < TT; e += 64) c += (g[e] != INFINITY) ? 1 : 0;
        } else {
#pragma unroll
            for (int e = lane; e < TT; e += 64) c += (g[e] != g0[e]) ? 1 : 0;
        }
        total += (unsigned long long)c;
        if (lane == 0) P.touched[gt] = 0;
    }
    for (int o = 32; o > 0; o >>= 1) total += __shfl_down(total, o);
    if (lane == 0 && total) atomicAdd(&P.ctr->expanded, total);
}
// The back-pointers of the tiles a step touched, once it has converged (see bp_byte): per tile one workgroup of four waves stages the tile, its
// ring and its cost window as a visit does and evaluates every node once, four lanes per node, keeping the arg-min.
template <int ALGO>
__device__ void tile_bp(const DevParams &P, int gt, int thr, float *Gs, float *Cs) {      // all threads of a 256-thread workgroup call
    constexpr int CROWS = is_dfm<ALGO> ? T : T + 1, COFF = is_dfm<ALGO> ? 0 : 1, CN = CROWS * CROWS;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, q = lane & 3, nd = lane >> 2;
    const int m = gt / P.NTm, t = gt - m * P.NTm, tx = t / P.TY, ty = t - tx * P.TY, x0 = tx * T, y0 = ty * T;
    const float *Gt = P.G + (size_t)gt * TT, *ring = P.ring + (size_t)gt * RING;
    const uint8_t *ct = P.costT + (size_t)gt * CTS;
    for (int e = tid; e < TT; e += 256) Gs[(e / T + 1) * GP + e % T + 1] = Gt[e];
    for (int ht = tid; ht < 4 * T + 4; ht += 256) {
        int hr, hc;
        if (ht < T) { hr = -1; hc = ht; }
        else if (ht < 2 * T) { hr = T; hc = ht - T; }
        else if (ht < 3 * T) { hr = ht - 2 * T; hc = -1; }
        else if (ht < 4 * T) { hr = ht - 3 * T; hc = T; }
        else { hr = (ht & 2) ? T : -1; hc = (ht & 1) ? T : -1; }
        Gs[(hr + 1) * GP + hc + 1] = ring[ht];
    }
    for (int e = tid; e < CN; e += 256) {
        const int cr = e / CROWS, cc = e - cr * CROWS, cx = x0 + cr - COFF, cy = y0 + cc - COFF, c = ct[e];
        Cs[cr * CP + cc] = (cx < 0 || cy < 0 || cx >= P.L || cy >= P.W || c >= thr) ? INFINITY : (float)c;
    }
    __syncthreads();
    const int goal_x = P.goal[2 * m], goal_y = P.goal[2 * m + 1];
    for (int p = w; p < PT * PT; p += 4) {
        const int lx = (p / PT) * 4 + (nd >> 2), ly = (p % PT) * 4 + (nd & 3);
        QuadConsts<ALGO> C;
        C.load(Cs, lx, ly, q);
        const LaneEval le = eval_quad_w<ALGO>(Gs + (lx + 1) * GP + ly + 1, q, C);
        const float nv = quad_min(le.r);
        const int b = quad_min_int(bp_byte<ALGO>(le, q, C, le.r == nv));
        if (q == 0) P.bp[(size_t)gt * TT + lx * T + ly] = (uint8_t)((x0 + lx == goal_x && y0 + ly == goal_y) ? BP_NONE : b);
    }
    __syncthreads();
}
template <int ALGO>
__global__ __launch_bounds__(256) void k_finalize_bp(DevParams P, int only_if_done) {
    __shared__ float Gs[(T + 2) * GP];
    __shared__ float Cs[(T + 1) * CP];
    if (only_if_done && !P.ctr->done) return;
    const int n = P.ctr->tcount, thr = P.dyn->thr;
    for (int i = blockIdx.x; i < n; i += gridDim.x) tile_bp<ALGO>(P, P.tlist[i], thr, Gs, Cs);
}
// Replan, end of the submission in one launch instead of three: every workgroup evaluates the
// device-side end condition (the queues are short; workgroup 0 records the verdict), finalises its
// share of the touched tiles if the replan is complete, and the last workgroup to finish writes the
// counters into host-coherent memory and bumps the sequence number the host spins on.
__global__ __launch_bounds__(T * T) void k_replan_end(DevParams P, int kr, int kl, float margin,
                                                      DevCounters *host, unsigned int *flag, unsigned int seq) {
    __shared__ int s_m, s_done, s_last;
    kr = launch_index(P, Q_RAISE, kr);
    kl = launch_index(P, Q_LOWER, kl);
    if (seq == 0) seq = P.ctr->pubseq;
    const int done = replan_check(P, kr, kl, margin, blockIdx.x == 0, s_m, s_done);
    if (done) finalize_tiles(P);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(&P.ctr->fin_blocks, 1) == (int)gridDim.x - 1);
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (threadIdx.x == 0) P.ctr->fin_blocks = 0;
    const int *src = reinterpret_cast<const int *>(P.ctr);
    int *dst = reinterpret_cast<int *>(host);
    for (int i = threadIdx.x; i < (int)(sizeof(DevCounters) / sizeof(int)); i += blockDim.x)
        dst[i] = __hip_atomic_load(&src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Replan, single map: everything between "the patches are in" and the first invalidation launch
// in one workgroup -- step bookkeeping, mark reset of the consumed patch rectangles, seeds into the
// invalidation queue, the invalidation bound, parked invalidations below it.  (Each separate
// launch costs ~5 us of dispatch latency; a replan used to begin with five of them.)
struct ReplanBegin { StepBegin sb; int nrect; int rect[4][5]; int k_raise; float band; };
__device__ void replan_begin(const DevParams &P, const ReplanBegin &a, int &s_keep) {
    step_begin(P, a.sb);
    for (int r = 0; r < a.nrect; ++r) {
        const int *q = a.rect[r];
        for (int e = threadIdx.x; e < (q[3] + 1) * (q[4] + 1); e += blockDim.x) clear_mark(P, q[0], q[1], q[2], q[3], q[4], e);
    }
    __syncthreads();
    seeds_to_active(P, Q_RAISE, a.k_raise, s_keep);
    if (threadIdx.x == 0) prepare_bound(P, a.band);
    __syncthreads();
    unpark(P, Q_RAISE, a.k_raise, -1.0f, s_keep);
}
__global__ __launch_bounds__(1024) void k_replan_begin(DevParams P, ReplanBegin a) {
    __shared__ int s_keep;
    replan_begin(P, a, s_keep);
}
// first node of the replan graph: the per-replan inputs come from host-coherent memory
struct ReplanJob { ReplanBegin rb; int k_lower; unsigned int seq; DevDyn dyn; };
__global__ __launch_bounds__(1024) void k_replan_begin_job(DevParams P, const ReplanJob *job) {
    __shared__ int s_keep;
    __shared__ ReplanJob s_job;
    for (int i = threadIdx.x; i < (int)(sizeof(ReplanJob) / sizeof(int)); i += blockDim.x)
        reinterpret_cast<int *>(&s_job)[i] = __hip_atomic_load(reinterpret_cast<const int *>(job) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    if (threadIdx.x == 0) {
        P.ctr->kbase[Q_RAISE] = s_job.rb.k_raise; P.ctr->kbase[Q_LOWER] = s_job.k_lower; P.ctr->pubseq = s_job.seq;
        *P.dyn = s_job.dyn;          // this workgroup reads it back below (prepare_bound, unpark), the later kernels from memory
    }
    __syncthreads();
    replan_begin(P, s_job.rb, s_keep);
}
// ... and between the invalidation and the lowering launches
__global__ __launch_bounds__(1024) void k_raise_to_lower(DevParams P, int k_lower) {
    __shared__ int s_keep;
    k_lower = launch_index(P, Q_LOWER, k_lower);
    const int n = P.ctr->tcount;
    for (int i = threadIdx.x; i < n; i += blockDim.x) activate(P, Q_LOWER, k_lower, P.tlist[i], 0);
    unpark(P, Q_LOWER, k_lower, INFINITY, s_keep);
}

#include "ufm_region.h"

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
    float owned_band = -1.0f;        // ... ordering band in tile crossings (< 0: twice delta_scale_long -- nobody waits for a launch to end
                                     //     here, and a workgroup that finds nothing inside the band idles: wider pays)
    uint32_t owned_launches = 0;
    hipEvent_t own_ev[2] = {nullptr, nullptr};
    hipEvent_t reg_ev[2] = {nullptr, nullptr};     // profiling: around the block kernel of a replan
    bool own_timed = false;
    int owned_flags = 0;             // resident kernel, diagnostics and variants.  1: no tile taken ahead; 2: no early hand-off (FD / SG); 4: early hand-off once per patch and visit;
                                     // 8: early hand-off waits for its stores inside the sweep loop; 16: no in-visit halo refresh; 32: idle workgroups do not help out
    int owned_waves = 0;             // waves per tile visit of the resident kernel: 16 (256 workgroups), 8 (512), 0 = by the size of the job
    bool use_region = true;          // replans: one workgroup runs both phases in LDS on the block around the patch (ufm_region.h);
                                     // the launch chain only takes over when work is left outside the block
    int region_ahead = 2;            // block placement: tiles kept between the patches' centre and the block's goal-side edge
    int region_tiles = 6;            // block edge in tiles (<= RTMAX; measured on the headline replans: 10 -> 6 tiles: 20.0 -> 18.6 ms per 100, same completion rate)
    int region_sweeps = 4096;        // sweep budget per wave and phase
    int region_debug = 0;
    float region_band = 1.5f;        // ordering band of the block's lowering sub-rounds, in patch crossings at the mean cost (0: unordered)
    uint32_t region_runs = 0, region_done = 0;   // replans submitted to the block kernel / completed by it alone
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
    // Small patches handed over as device pointers are held back until something needs them applied (the next step, a read
    // of the raster, a path extraction): the caller keeps such a buffer valid until the next step() has returned anyway.
    struct DeferredPatch { int m, x, y, w, h; const uint8_t *ptr; };
    std::vector<DeferredPatch> deferred;
    bool defer_patches = true;
    int flush_deferred();
};

void Engine::release() {
    if (!allocated) return;
    if (stream) hipStreamSynchronize(stream);
    drop_graphs();                       // captured kernel arguments hold these pointers
    deferred.clear();
    std::memset(&graph_sig, 0, sizeof(graph_sig));
    void *ptrs[] = {P.G, P.Gprev, P.bp, P.ring, P.cost, P.costT, P.goal, P.cand, P.ready, P.hint, P.rank, P.park, P.pflag, P.pprio,
                    P.queued, P.prio, P.start, P.bnd, P.dyn, P.spos, P.touched, P.fresh, P.tlist, P.sflag, P.slist, P.slist2,
                    P.mark, P.num_updated, P.consume, P.lmax, P.own_prio, P.own_lock, P.own_min, P.ctr, d_scratch};
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
    dmalloc(P.ctr, sizeof(DevCounters));
    dmalloc(d_scratch, sizeof(int) * (4 * nmaps + 16));
    if (rc != UFM_OK) { release(); return rc; }
    rc = [&]() -> int {
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
        else if (opt_lvl == 0) UFM_LAUNCH(UFM_ALGO_DFM, MODE_LOWER);
        else UFM_LAUNCH(ALGO_DFM1, MODE_LOWER);
    } else {
        if (algo == UFM_ALGO_FD) UFM_LAUNCH(UFM_ALGO_FD, MODE_RAISE);
        else if (algo == UFM_ALGO_SG) UFM_LAUNCH(UFM_ALGO_SG, MODE_RAISE);
        else if (opt_lvl == 0) UFM_LAUNCH(UFM_ALGO_DFM, MODE_RAISE);
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
        else if (opt_lvl == 0) UFM_LAUNCH(UFM_ALGO_DFM, MODE_LOWER);
        else UFM_LAUNCH(ALGO_DFM1, MODE_LOWER);
    } else {
        if (algo == UFM_ALGO_FD) UFM_LAUNCH(UFM_ALGO_FD, MODE_RAISE);
        else if (algo == UFM_ALGO_SG) UFM_LAUNCH(UFM_ALGO_SG, MODE_RAISE);
        else if (opt_lvl == 0) UFM_LAUNCH(UFM_ALGO_DFM, MODE_RAISE);
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
    const float delta = delta_abs >= 0.0f ? delta_abs : (owned_band >= 0.0f ? owned_band : 2.0f * delta_scale_long) * T * mean_cost;
    const double limit_ms = owned_limit_ms >= 0.0f ? (double)owned_limit_ms : 200.0 + (double)P.NT / 250.0;   // (4096^2: 0.46 s; its plan takes 17 ms)
    P.own_limit = (unsigned long long)(limit_ms * 1e5);   // 100 MHz ticks
    P.own_flags = owned_flags;
    // (measured with the helping workgroups in place: FD 4096^2 15.9-16.3 ms with 8 waves against 16.5-16.8 with 16, 2048^2 7.25 against 6.44,
    //  SG 2048^2 7.08 against 6.55, 1024^2 3.40 against 2.84; MS-DFM, whose visits are longer and which has no early hand-off, 2048^2 13.1 against 15.0)
    const bool half = T == 16 && (owned_waves == 8 || (owned_waves == 0 && (nmaps > 1 || P.NTm > (algo == UFM_ALGO_DFM ? 12000 : 50000))));
    own_layout(half ? 5 : 4);
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
    else if (opt_lvl == 0) UFM_LAUNCH(UFM_ALGO_DFM);
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

int Engine::flush_deferred() {
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

int Engine::patch(int m, const uint8_t *dev_patch, int x, int y, int w, int h, bool may_defer) {
    if (m < 0 || m >= nmaps || !allocated || !maps[m].have_map) return UFM_ERR_INVALID;
    if (x < 0 || y < 0 || w <= 0 || h <= 0 || x + h > L || y + w > W) return UFM_ERR_INVALID;   // Graph.cpp:38-41
    const int n = w * h;
    {   // room for the masks of PATCH_MULTI small patches, or of one large one
        const size_t need = std::max((size_t)PATCH_MULTI * 4096, (size_t)n);
        if (need > d_pmask_cap) {
            { int rc = flush_deferred(); if (rc != UFM_OK) return rc; }
            if (d_pmask) { HIPCHK(hipStreamSynchronize(stream)); hipFree(d_pmask); d_pmask = nullptr; d_pmask_cap = 0; }
            HIPCHK(hipMalloc(&d_pmask, need));
            d_pmask_cap = need;
        }
    }
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
    { int rc = flush_deferred(); if (rc != UFM_OK) return rc; }   // the patches held back: applied now, in one launch

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
            const int cx = (int)std::roundf(ms.start_x), cy = (int)std::roundf(ms.start_y);
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
                if (b.nrect <= 0) return false;
                int ex0 = INT32_MAX, ex1 = -1, ey0 = INT32_MAX, ey1 = -1;
                for (int r = 0; r < b.nrect; ++r) {
                    const int *qr = b.rect[r];
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
                const bool okx = place(ex0, ex1, maps[m].goal_ex, P.TX, &j.tx0, &j.ntx);
                const bool oky = place(ey0, ey1, maps[m].goal_ey, P.TY, &j.ty0, &j.nty);
                if (!(okx && oky)) return false;
                j.rb = b; j.rb.k_raise = iter[Q_RAISE]; j.rb.band = band;
                j.dyn = dyn_now; j.k_lower = iter[Q_LOWER]; j.max_sweeps = region_sweeps; j.debug = region_debug;
                j.slack = 255.0f * SQRT2F + 1.0f;     // the largest cost of one move (a diagonal through the most expensive cell)
                j.delta = region_band > 0.0f ? region_band * 4.0f * mean_cost : INFINITY;
                j.map = m;
                return true;
            };
            if (fused && use_region) {                       // one map, a few small patches
                regioned = place_job(rjs.j[0], rb, 0);
                rjs.n = 1; rjs.j[0].batch = 0;
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
            else if (opt_lvl == 0) UFM_LAUNCH(UFM_ALGO_DFM);
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

// Back-pointers of a window of elements: the stored codes in the reference's format (k_info_stored), or derived from the field alone
// (k_info, the checker), ufm_path.h.
int engine_read_info(Engine *e, int m, int x0, int y0, int nx, int ny, int32_t *info, bool derived) {
    if (!e || m < 0 || m >= e->nmaps || !e->allocated || !info) return UFM_ERR_INVALID;
    if (e->opt_lvl == 0) return UFM_ERR_INVALID;            // level 0: the map has no Info member (void)
    if (x0 < 0 || y0 < 0 || nx <= 0 || ny <= 0 || x0 + nx > e->P.EX || y0 + ny > e->P.EY) return UFM_ERR_INVALID;
    HIPCHK(hipSetDevice(e->device));
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

}  // namespace

// ---- C ABI ---------------------------------------------------------------------------
struct ufm_planner { Engine *e; };
struct ufm_batch { std::vector<Engine *> shards; int n_maps = 0, per = 1; };

extern "C" {

#ifdef UFM_SWEEPSTAT
int ufm_debug_sstat(unsigned long long *out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sstat), sizeof(unsigned long long) * 32) != hipSuccess) return UFM_ERR_HIP_BASE;
    if (reset) { unsigned long long z[32] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_sstat), z, sizeof(z)) != hipSuccess) return UFM_ERR_HIP_BASE; }
    return UFM_OK;
}
#endif
#ifdef UFM_TIMING
int ufm_debug_trace(unsigned long long *out, int cap) {     // returns the number of records copied (4 words each)
    unsigned int n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_ntrace), sizeof(n)) != hipSuccess) return UFM_ERR_HIP_BASE;
    if ((int)n > cap) n = cap;
    if (n > 16384) n = 16384;
    if (n && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), sizeof(unsigned long long) * 4 * n) != hipSuccess) return UFM_ERR_HIP_BASE;
    const unsigned int z = 0;
    hipMemcpyToSymbol(HIP_SYMBOL(g_ntrace), &z, sizeof(z));
    return (int)n;
}
int ufm_debug_visits(unsigned int *out, int cap) {   // out[cap][5]; returns the number of visits recorded
    unsigned int n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_nvis), sizeof(n)) != hipSuccess) return UFM_ERR_HIP_BASE;
    if ((int)n > cap) n = cap;
    if (n > VIS_DIAG_MAX) n = VIS_DIAG_MAX;
    if (n && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_vis), sizeof(unsigned int) * 5 * n) != hipSuccess) return UFM_ERR_HIP_BASE;
    return (int)n;
}
int ufm_debug_sdiag(unsigned long long *out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sdiag), sizeof(unsigned long long) * 16) != hipSuccess) return UFM_ERR_HIP_BASE;
    if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_sdiag), z, sizeof(z)) != hipSuccess) return UFM_ERR_HIP_BASE; }
    return UFM_OK;
}
int ufm_debug_tiles(unsigned int *out, int n) {     // out[5][n]
    if (n > TILE_DIAG_MAX) n = TILE_DIAG_MAX;
    for (int r = 0; r < 5; ++r)
        if (hipMemcpyFromSymbol(out + (size_t)r * n, HIP_SYMBOL(g_tile), sizeof(unsigned int) * n, sizeof(unsigned int) * (size_t)r * TILE_DIAG_MAX) != hipSuccess) return UFM_ERR_HIP_BASE;
    return n;
}
int ufm_debug_wtrace(unsigned long long *out, unsigned int *counts) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wtrace), sizeof(unsigned long long) * 16 * 256 * 2) != hipSuccess) return UFM_ERR_HIP_BASE;
    if (hipMemcpyFromSymbol(counts, HIP_SYMBOL(g_nw), sizeof(unsigned int) * 16) != hipSuccess) return UFM_ERR_HIP_BASE;
    unsigned int z[16] = {};
    hipMemcpyToSymbol(HIP_SYMBOL(g_nw), z, sizeof(z));
    return UFM_OK;
}
int ufm_debug_tdiag(unsigned long long *out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tdiag), sizeof(unsigned long long) * 64) != hipSuccess) return UFM_ERR_HIP_BASE;
    if (reset) { unsigned long long z[64] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_tdiag), z, sizeof(z)) != hipSuccess) return UFM_ERR_HIP_BASE; }
    return UFM_OK;
}
#endif
int ufm_debug_lmax(ufm_t *p, int32_t *out, int n) {      // diagnostics array of the device (not part of include/ufm.h)
    if (!p || !p->e->allocated || n > LMAX) return UFM_ERR_INVALID;
    hipStreamSynchronize(p->e->stream);
    return hipMemcpy(out, p->e->P.lmax, sizeof(int) * n, hipMemcpyDeviceToHost) == hipSuccess ? UFM_OK : UFM_ERR_HIP_BASE;
}
int ufm_tile_edge(void) { return T; }
#ifdef UFM_STRICT_FENCES
const char *ufm_version(void) { return "ufm-gfx950 0.1 (block-FIM, strict-fence checking build)"; }
#else
const char *ufm_version(void) { return T == 32 ? "ufm-gfx950 0.1 (block-FIM, tile 32)" : "ufm-gfx950 0.1 (block-FIM, tile 16)"; }
#endif

int ufm_create(ufm_t **out, int algo, int opt_lvl, int use_heuristic, int device_id) {
    if (!out) return UFM_ERR_INVALID;
    Engine *e = nullptr;
    int rc = engine_create(&e, 1, algo, opt_lvl, use_heuristic, device_id);
    if (rc != UFM_OK) return rc;
    *out = new ufm_planner{e};
    return UFM_OK;
}
int ufm_destroy(ufm_t *p) {
    if (!p) return UFM_ERR_INVALID;
    int rc = engine_destroy(p->e);
    delete p;
    return rc;
}
int ufm_reset(ufm_t *p) { if (!p) return UFM_ERR_INVALID; p->e->maps[0].initialize_search = true; return UFM_OK; }
int ufm_set_occupancy_threshold(ufm_t *p, float thr) {
    if (!p) return UFM_ERR_INVALID;
    p->e->thr_uchar = (int)(thr * 255.0f);   // Graph.cpp:18-20
    return UFM_OK;
}
int ufm_set_heuristic_multiplier(ufm_t *p, float mult) { if (!p) return UFM_ERR_INVALID; p->e->heuristic_multiplier = mult; return UFM_OK; }
int ufm_set_map(ufm_t *p, const uint8_t *host_map, int width, int length) { return p ? engine_set_map(p->e, 0, host_map, false, width, length) : UFM_ERR_INVALID; }
int ufm_set_map_device(ufm_t *p, const uint8_t *dev_map, int width, int length) { return p ? engine_set_map(p->e, 0, dev_map, true, width, length) : UFM_ERR_INVALID; }
int ufm_patch_map(ufm_t *p, const uint8_t *host_patch, int x, int y, int w, int h) { return p ? engine_patch(p->e, 0, host_patch, false, x, y, w, h) : UFM_ERR_INVALID; }
int ufm_patch_map_device(ufm_t *p, const uint8_t *dev_patch, int x, int y, int w, int h) { return p ? engine_patch(p->e, 0, dev_patch, true, x, y, w, h) : UFM_ERR_INVALID; }
int ufm_set_start(ufm_t *p, float x, float y) {
    if (!p) return UFM_ERR_INVALID;
    MapState &ms = p->e->maps[0];
    ms.start_x = x; ms.start_y = y; ms.new_start = true; ms.start_set = true;   // ReplannerBase.h:94-97
    return UFM_OK;
}
int ufm_set_goal(ufm_t *p, float x, float y) { return p ? engine_set_goal(p->e, 0, x, y) : UFM_ERR_INVALID; }
int ufm_step(ufm_t *p, ufm_stats *stats) {
    if (!p) return UFM_ERR_INVALID;
    if (hipSetDevice(p->e->device) != hipSuccess) return UFM_ERR_HIP_BASE;
    return p->e->step(stats);
}
int ufm_field_dims(const ufm_t *p, int *nx, int *ny) {
    if (!p || !p->e->allocated) return UFM_ERR_INVALID;
    if (nx) *nx = p->e->P.EX;
    if (ny) *ny = p->e->P.EY;
    return UFM_OK;
}
int ufm_read_field(ufm_t *p, int x0, int y0, int nx, int ny, float *g, float *rhs) { return p ? engine_read_field(p->e, 0, x0, y0, nx, ny, g, rhs) : UFM_ERR_INVALID; }
static int engine_check_layout(Engine *e, uint64_t *bad_ring, uint64_t *bad_cost) {
    if (!e || !e->allocated) return UFM_ERR_INVALID;
    for (const MapState &ms : e->maps) if (!ms.have_map) return UFM_ERR_INVALID;
    HIPCHK(hipSetDevice(e->device));
    { int rc = e->flush_deferred(); if (rc != UFM_OK) return rc; }
    unsigned long long *d_acc = reinterpret_cast<unsigned long long *>(e->d_scratch);
    HIPCHK(hipMemsetAsync(d_acc, 0, 2 * sizeof(unsigned long long), e->stream));
    k_check_layout<<<1024, 256, 0, e->stream>>>(e->P, d_acc);
    unsigned long long h[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(h, d_acc, sizeof(h), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (bad_ring) *bad_ring = h[0];
    if (bad_cost) *bad_cost = h[1];
    return UFM_OK;
}
int ufm_check_layout(ufm_t *p, uint64_t *bad_ring, uint64_t *bad_cost) { return p ? engine_check_layout(p->e, bad_ring, bad_cost) : UFM_ERR_INVALID; }
static int engine_check_info(Engine *e, uint64_t out[6]) {
    if (!e || !e->allocated || !out) return UFM_ERR_INVALID;
    if (e->algo == UFM_ALGO_DFM) return UFM_ERR_INVALID;       // (MS-DFM invalidates by evaluation: its bytes are for ufm_read_info only)
    for (const MapState &ms : e->maps) if (!ms.have_map) return UFM_ERR_INVALID;
    HIPCHK(hipSetDevice(e->device));
    { int rc = e->flush_deferred(); if (rc != UFM_OK) return rc; }
    unsigned long long *d_acc = reinterpret_cast<unsigned long long *>(e->d_scratch);
    HIPCHK(hipMemsetAsync(d_acc, 0, 6 * sizeof(unsigned long long), e->stream));
    if (e->algo == UFM_ALGO_SG) k_check_bp<UFM_ALGO_SG><<<1024, 256, 0, e->stream>>>(e->P, d_acc);
    else k_check_bp<UFM_ALGO_FD><<<1024, 256, 0, e->stream>>>(e->P, d_acc);
    unsigned long long h[6] = {0, 0, 0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(h, d_acc, sizeof(h), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int i = 0; i < 6; ++i) out[i] = h[i];
    return UFM_OK;
}
int ufm_check_info(ufm_t *p, uint64_t out[6]) { return p ? engine_check_info(p->e, out) : UFM_ERR_INVALID; }
int ufm_batch_check_info(ufm_batch_t *b, uint64_t out[6]) {
    if (!b || !out) return UFM_ERR_INVALID;
    uint64_t acc[6] = {0, 0, 0, 0, 0, 0};
    for (Engine *e : b->shards) {
        uint64_t o[6];
        const int rc = engine_check_info(e, o);
        if (rc != UFM_OK) return rc;
        for (int i = 0; i < 6; ++i) acc[i] += o[i];
    }
    for (int i = 0; i < 6; ++i) out[i] = acc[i];
    return UFM_OK;
}
int ufm_batch_check_layout(ufm_batch_t *b, uint64_t *bad_ring, uint64_t *bad_cost) {
    if (!b) return UFM_ERR_INVALID;
    uint64_t r = 0, c = 0;
    for (Engine *e : b->shards) {
        uint64_t a = 0, d = 0;
        const int rc = engine_check_layout(e, &a, &d);
        if (rc != UFM_OK) return rc;
        r += a; c += d;
    }
    if (bad_ring) *bad_ring = r;
    if (bad_cost) *bad_cost = c;
    return UFM_OK;
}
static int engine_read_map(Engine *e, int m, uint8_t *host_map) {
    if (!e || !host_map || !e->allocated || m < 0 || m >= e->nmaps) return UFM_ERR_INVALID;
    HIPCHK(hipSetDevice(e->device));
    { int rc = e->flush_deferred(); if (rc != UFM_OK) return rc; }
    HIPCHK(hipMemcpyAsync(host_map, e->P.cost + (size_t)m * e->P.cstride, e->P.cstride, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return UFM_OK;
}
int ufm_read_map(ufm_t *p, uint8_t *host_map) { return p ? engine_read_map(p->e, 0, host_map) : UFM_ERR_INVALID; }
static int engine_set_param(Engine *e, const char *name, double value) {
    if (!e || !name) return UFM_ERR_INVALID;
    if (!std::strcmp(name, "delta")) e->delta_abs = (float)value;
    else if (!std::strcmp(name, "delta_scale")) { e->delta_scale = e->delta_scale_long = (float)value; e->delta_abs = -1.0f; }
    else if (!std::strcmp(name, "delta_scale_long")) { e->delta_scale_long = (float)value; e->delta_abs = -1.0f; }
    else if (!std::strcmp(name, "max_iters")) e->max_iters = value < 1 ? 1 : (int)value;
    else if (!std::strcmp(name, "batch")) e->batch_fixed = value < 0 ? 0 : (value > 1024 ? 1024 : (int)value);
    else if (!std::strcmp(name, "pipeline_batches")) e->pipeline_batches = value != 0;
    else if (!std::strcmp(name, "grid")) e->grid_relax = value < 1 ? 1 : (int)value;
    else if (!std::strcmp(name, "small_grid")) e->small_grid = value < 1 ? 1 : (int)value;
    else if (!std::strcmp(name, "dyn_grid")) e->dyn_grid = value < 1 ? 1 : (int)value;
    else if (!std::strcmp(name, "spin_wait")) e->spin_wait = value != 0;
    else if (!std::strcmp(name, "fuse_control")) e->fuse_control = value != 0;
    else if (!std::strcmp(name, "graph")) e->use_graph = value != 0;
    else if (!std::strcmp(name, "region")) e->use_region = value != 0;
    else if (!std::strcmp(name, "owned")) e->use_owned = value != 0;
    else if (!std::strcmp(name, "owned_limit_ms")) e->owned_limit_ms = (float)value;
    else if (!std::strcmp(name, "owned_band")) e->owned_band = (float)value;
    else if (!std::strcmp(name, "owned_flags")) e->owned_flags = (int)value;
    else if (!std::strcmp(name, "owned_waves")) e->owned_waves = (int)value;
    else if (!std::strcmp(name, "defer_patches")) { if (e->allocated) { int rc = e->flush_deferred(); if (rc != UFM_OK) return rc; } e->defer_patches = value != 0; }
    else if (!std::strcmp(name, "region_debug")) e->region_debug = (int)value;
    else if (!std::strcmp(name, "region_band")) e->region_band = (float)value;
    else if (!std::strcmp(name, "region_ahead")) e->region_ahead = (int)value;
    else if (!std::strcmp(name, "region_tiles")) e->region_tiles = value < 3 ? 3 : (value > RTMAX ? RTMAX : (int)value);
    else if (!std::strcmp(name, "region_sweeps")) e->region_sweeps = value < 16 ? 16 : (int)value;
    else if (!std::strcmp(name, "batch_margin")) e->batch_margin = (int)value;
    else if (!std::strcmp(name, "raise_margin")) e->raise_margin = (float)value;
    else if (!std::strcmp(name, "tail_grid")) e->tail_grid = value < 1 ? 1 : (int)value;
    else if (!std::strcmp(name, "profile_stride")) e->profile_stride = value < 1 ? 1 : (int)value;
    else if (!std::strcmp(name, "focused")) e->focused = value != 0.0;
    else if (!std::strcmp(name, "dynamic")) e->dynamic_mode = value != 0.0;
    else return UFM_ERR_INVALID;
    return UFM_OK;
}
int ufm_set_param(ufm_t *p, const char *name, double value) { return p ? engine_set_param(p->e, name, value) : UFM_ERR_INVALID; }
int ufm_batch_set_param(ufm_batch_t *b, const char *name, double value) {
    if (!b) return UFM_ERR_INVALID;
    for (Engine *e : b->shards) { const int rc = engine_set_param(e, name, value); if (rc != UFM_OK) return rc; }
    return UFM_OK;
}
int ufm_set_profiling(ufm_t *p, int enable) { if (!p) return UFM_ERR_INVALID; p->e->profiling = enable != 0; return UFM_OK; }
void *ufm_stream(ufm_t *p) { return p ? (void *)p->e->stream : nullptr; }

// A batch is one engine per device; map i lives in shard i / per (contiguous blocks of maps).
static int batch_locate(const ufm_batch *b, int i, Engine **e, int *local) {
    if (!b || i < 0 || i >= b->n_maps) return UFM_ERR_INVALID;
    const int s = i / b->per;
    *e = b->shards[s];
    *local = i - s * b->per;
    return UFM_OK;
}
int ufm_batch_create_sharded(ufm_batch_t **out, int n_maps, int algo, int opt_lvl, int use_heuristic, const int *devices, int n_devices) {
    if (!out || !devices || n_devices < 1 || n_maps < n_devices) return UFM_ERR_INVALID;
    ufm_batch *b = new (std::nothrow) ufm_batch();
    if (!b) return UFM_ERR_NOMEM;
    b->n_maps = n_maps;
    b->per = (n_maps + n_devices - 1) / n_devices;
    for (int s = 0; s * b->per < n_maps; ++s) {
        Engine *e = nullptr;
        const int cnt = std::min(b->per, n_maps - s * b->per);
        const int rc = engine_create(&e, cnt, algo, opt_lvl, use_heuristic, devices[s]);
        if (rc != UFM_OK) { ufm_batch_destroy(b); return rc; }
        b->shards.push_back(e);
    }
    *out = b;
    return UFM_OK;
}
int ufm_batch_create(ufm_batch_t **out, int n_maps, int algo, int opt_lvl, int use_heuristic, int device_id) {
    return ufm_batch_create_sharded(out, n_maps, algo, opt_lvl, use_heuristic, &device_id, 1);
}
int ufm_batch_destroy(ufm_batch_t *b) {
    if (!b) return UFM_ERR_INVALID;
    int rc = UFM_OK;
    for (Engine *e : b->shards) { const int r = engine_destroy(e); if (rc == UFM_OK) rc = r; }
    delete b;
    return rc;
}
int ufm_batch_size(const ufm_batch_t *b) { return b ? b->n_maps : UFM_ERR_INVALID; }
int ufm_batch_shards(const ufm_batch_t *b) { return b ? (int)b->shards.size() : UFM_ERR_INVALID; }
int ufm_batch_set_occupancy_threshold(ufm_batch_t *b, float thr) {
    if (!b) return UFM_ERR_INVALID;
    for (Engine *e : b->shards) e->thr_uchar = (int)(thr * 255.0f);
    return UFM_OK;
}
int ufm_batch_set_heuristic_multiplier(ufm_batch_t *b, float mult) {
    if (!b) return UFM_ERR_INVALID;
    for (Engine *e : b->shards) e->heuristic_multiplier = mult;
    return UFM_OK;
}
#define UFM_BATCH_MAP(b, i) Engine *e = nullptr; int li = 0; { const int rc_ = batch_locate(b, i, &e, &li); if (rc_ != UFM_OK) return rc_; }
int ufm_batch_set_map(ufm_batch_t *b, int i, const uint8_t *host_map, int width, int length) { UFM_BATCH_MAP(b, i); return engine_set_map(e, li, host_map, false, width, length); }
int ufm_batch_set_map_device(ufm_batch_t *b, int i, const uint8_t *dev_map, int width, int length) { UFM_BATCH_MAP(b, i); return engine_set_map(e, li, dev_map, true, width, length); }
int ufm_batch_patch_map(ufm_batch_t *b, int i, const uint8_t *host_patch, int x, int y, int w, int h) { UFM_BATCH_MAP(b, i); return engine_patch(e, li, host_patch, false, x, y, w, h); }
int ufm_batch_patch_map_device(ufm_batch_t *b, int i, const uint8_t *dev_patch, int x, int y, int w, int h) { UFM_BATCH_MAP(b, i); return engine_patch(e, li, dev_patch, true, x, y, w, h); }
int ufm_batch_set_start(ufm_batch_t *b, int i, float x, float y) {
    UFM_BATCH_MAP(b, i);
    MapState &ms = e->maps[li];
    ms.start_x = x; ms.start_y = y; ms.new_start = true; ms.start_set = true;
    return UFM_OK;
}
int ufm_batch_set_goal(ufm_batch_t *b, int i, float x, float y) { UFM_BATCH_MAP(b, i); return engine_set_goal(e, li, x, y); }
int ufm_batch_reset(ufm_batch_t *b, int i) {
    UFM_BATCH_MAP(b, i);
    e->maps[li].initialize_search = true;
    return UFM_OK;
}
// One step of every map.  Shards on different devices advance side by side, one host thread each (a step is
// synchronous: its host thread spins on the device's published counters); the statistics are summed, the times
// are those of the slowest shard.
int ufm_batch_step(ufm_batch_t *b, ufm_stats *stats) {
    if (!b || b->shards.empty()) return UFM_ERR_INVALID;
    const size_t n = b->shards.size();
    std::vector<ufm_stats> st(n);
    std::vector<int> rcs(n, UFM_OK);
    auto run = [&](size_t s) {
        if (hipSetDevice(b->shards[s]->device) != hipSuccess) { rcs[s] = UFM_ERR_HIP_BASE; return; }
        rcs[s] = b->shards[s]->step(&st[s]);
    };
    if (n == 1) run(0);
    else {
        std::vector<std::thread> th;
        for (size_t s = 1; s < n; ++s) th.emplace_back(run, s);
        run(0);
        for (auto &t : th) t.join();
    }
    for (size_t s = 0; s < n; ++s) if (rcs[s] != UFM_OK) return rcs[s];
    if (stats) {
        ufm_stats a = st[0];
        for (size_t s = 1; s < n; ++s) {
            const ufm_stats &c = st[s];
            a.u_ms = std::max(a.u_ms, c.u_ms); a.p_ms = std::max(a.p_ms, c.p_ms);
            a.updated += c.updated; a.expanded += c.expanded; a.tile_visits += c.tile_visits; a.tile_iters += c.tile_iters;
            a.elem_evals += c.elem_evals; a.launches += c.launches; a.raise_launches += c.raise_launches; a.kernel_ms += c.kernel_ms;
            a.crit_sweeps += c.crit_sweeps; a.raise_tile_visits += c.raise_tile_visits; a.raise_kernel_ms += c.raise_kernel_ms;
            a.queued_lower += c.queued_lower; a.queued_raise += c.queued_raise; a.timed_launches += c.timed_launches;
            a.timed_raise_launches += c.timed_raise_launches; a.graphs_instantiated += c.graphs_instantiated;
            a.region_replans += c.region_replans; a.region_replans_done += c.region_replans_done;
            a.resident_launches += c.resident_launches; a.resident_kernel_ms += c.resident_kernel_ms;
            a.resident_stops += c.resident_stops; a.resident_tile_visits += c.resident_tile_visits;
            a.region_launches += c.region_launches; a.region_timed += c.region_timed; a.region_kernel_ms += c.region_kernel_ms; a.region_tiles += c.region_tiles;
        }
        *stats = a;
    }
    return UFM_OK;
}
int ufm_batch_read_field(ufm_batch_t *b, int i, int x0, int y0, int nx, int ny, float *g, float *rhs) { UFM_BATCH_MAP(b, i); return engine_read_field(e, li, x0, y0, nx, ny, g, rhs); }
int ufm_batch_read_map(ufm_batch_t *b, int i, uint8_t *host_map) { UFM_BATCH_MAP(b, i); return engine_read_map(e, li, host_map); }
int ufm_batch_set_profiling(ufm_batch_t *b, int enable) {
    if (!b) return UFM_ERR_INVALID;
    for (Engine *e : b->shards) e->profiling = enable != 0;
    return UFM_OK;
}
void *ufm_batch_stream(ufm_batch_t *b, int shard) { return (b && shard >= 0 && shard < (int)b->shards.size()) ? (void *)b->shards[shard]->stream : nullptr; }

int ufm_read_info(ufm_t *p, int x0, int y0, int nx, int ny, int32_t *info) { return p ? engine_read_info(p->e, 0, x0, y0, nx, ny, info, false) : UFM_ERR_INVALID; }
int ufm_read_info_derived(ufm_t *p, int x0, int y0, int nx, int ny, int32_t *info) { return p ? engine_read_info(p->e, 0, x0, y0, nx, ny, info, true) : UFM_ERR_INVALID; }
int ufm_extract_path(ufm_t *p, int max_steps, int lookahead, int allow_indirect,
                     float *path_xy, int cap_points, float *step_costs, int cap_costs, ufm_path_info *info) {
    return p ? engine_extract_path(p->e, max_steps, lookahead, allow_indirect, path_xy, cap_points, step_costs, cap_costs, info) : UFM_ERR_INVALID;
}
int ufm_batch_extract_path(ufm_batch_t *b, int max_steps, int lookahead, int allow_indirect,
                           float *path_xy, int cap_points, float *step_costs, int cap_costs, ufm_path_info *info) {
    if (!b) return UFM_ERR_INVALID;
    int first = 0;                      // shard by shard (one launch each), outputs in map order
    for (Engine *e : b->shards) {
        const int rc = engine_extract_path(e, max_steps, lookahead, allow_indirect,
                                           path_xy ? path_xy + (size_t)first * cap_points * 2 : nullptr, cap_points,
                                           step_costs ? step_costs + (size_t)first * cap_costs : nullptr, cap_costs, info ? info + first : nullptr);
        if (rc != UFM_OK) return rc;
        first += e->nmaps;
    }
    return UFM_OK;
}

}  // extern "C"
