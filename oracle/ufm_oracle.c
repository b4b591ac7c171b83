/*
 * ufm_oracle.c -- TEST INFRASTRUCTURE ONLY (see ufm_oracle.h).
 *
 * Plain-C restatement of the reference's three replanners on dense arrays
 * with an indexed binary heap.  Every function cites the reference source
 * (paths relative to the reference tree) whose behaviour it follows.
 * The reference keeps its search state in hash maps and a Fibonacci heap;
 * those are containers only (no arithmetic), so dense arrays + a binary
 * heap give the same G/RHS values.  Tie-breaking among equal keys is not
 * defined by the reference either (heap implementation detail).
 *
 * Floating point: strict IEEE single precision, one rounding per
 * operation (build with -ffp-contract=off, no fast-math), sqrtf correctly
 * rounded -- the same arithmetic contract the HIP kernels are built to.
 *
 * PARITY PINNING: see ufm_oracle.h -- FD with heuristic keys, replans and
 * extraction included, is pinned by the reference's two recorded mission logs
 * (both replayed in full in their revision, tests/test_reference_mission.py);
 * MS-DFM, FD's Type III and whole fields are "parity unpinned" (no golden
 * vectors exist in the reference; the reference is unbuildable in this image),
 * cross-checked against SURVEY.md App. E known answers in tests/test_oracle.py.
 */
#include "ufm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ProjectToolkit/Macros.cpp:2 */
static const float SQRT2 = 1.41421356237309504880168872420969807856967187537694f;

typedef struct { float k1, k2; } okey;

struct orc_planner {
    int algo, lvl, heur;
    /* Graph (ProjectToolkit/include/Graph.h:20-66) */
    uint8_t *map;
    int W, L;
    int thr_uchar; /* Graph.h:34 default 254 */
    int *upd_cells; /* Graph::updated_cells_ as linear cell indices */
    int n_upd, cap_upd;
    float start_px, start_py, goal_px, goal_py;
    int start_cx, start_cy, goal_cx, goal_cy;
    int start_nx, start_ny, goal_nx, goal_ny;
    /* ReplannerBase.h:147-152 */
    int initialize_graph, initialize_search, goal_set, new_goal, new_start;
    float hm;
    unsigned long num_updated, num_expanded;
    float u_time, p_time;
    /* ExpandedMap as dense arrays */
    int nx, ny;
    size_t n;
    float *g, *rhs;
    uint8_t *inmap;
    int32_t *bptr; /* lvl>=1: FD/SG one index per elem; DFM two */
    unsigned long map_size;
    /* PriorityQueue as indexed binary heap */
    int *heap;
    okey *hkey;
    int *hpos;
    int hn;
    /* FieldDPlanner::start_nodes (FieldDPlanner.h:62) */
    int start_nodes[4];
    int n_start_nodes;
    int start_set;
    int runaway;   /* the last plan() hit expansion_cap() */
    int revision;  /* ORC_REV_CURRENT (the sources as they stand) or bits of ORC_REV_LOG (see orc_set_revision) */
    /* orc_track_changes(): which elements' G a step touched, with the value each had before -- to count the elements whose G
     * DIFFERS after a step, the engine's definition of "cells updated" (num_nodes_expanded counts queue pops) */
    int chg_on;
    uint32_t chg_step, *chg_stamp;
    float *chg_g0;
    int *chg_list, chg_n;
};

/* ---------------- keys: std::pair lexicographic / float --------------- */
static inline int key_lt(okey a, okey b) {
    return a.k1 < b.k1 || (!(b.k1 < a.k1) && a.k2 < b.k2);
}
static inline int key_le(okey a, okey b) { return !key_lt(b, a); }

/* ---------------- heap (PriorityQueue.h:7-62, impl) ------------------- */
static void heap_swap(orc_t *p, int i, int j) {
    int ei = p->heap[i], ej = p->heap[j];
    okey ki = p->hkey[i];
    p->heap[i] = ej; p->hkey[i] = p->hkey[j]; p->hpos[ej] = i;
    p->heap[j] = ei; p->hkey[j] = ki; p->hpos[ei] = j;
}
static void heap_up(orc_t *p, int i) {
    while (i > 0) {
        int par = (i - 1) >> 1;
        if (key_lt(p->hkey[i], p->hkey[par])) { heap_swap(p, i, par); i = par; }
        else break;
    }
}
static void heap_down(orc_t *p, int i) {
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = i;
        if (l < p->hn && key_lt(p->hkey[l], p->hkey[m])) m = l;
        if (r < p->hn && key_lt(p->hkey[r], p->hkey[m])) m = r;
        if (m == i) break;
        heap_swap(p, i, m);
        i = m;
    }
}
/* PriorityQueue_impl.h:30-36 insert_or_update */
static void pq_insert_or_update(orc_t *p, int e, okey k) {
    int pos = p->hpos[e];
    if (pos >= 0) {
        okey old = p->hkey[pos];
        p->hkey[pos] = k;
        if (key_lt(k, old)) heap_up(p, pos); else heap_down(p, pos);
    } else {
        pos = p->hn++;
        p->heap[pos] = e; p->hkey[pos] = k; p->hpos[e] = pos;
        heap_up(p, pos);
    }
}
/* PriorityQueue_impl.h:21-28 remove_if_present */
static void pq_remove_if_present(orc_t *p, int e) {
    int pos = p->hpos[e];
    if (pos < 0) return;
    int last = --p->hn;
    p->hpos[e] = -1;
    if (pos != last) {
        okey old = p->hkey[pos];
        p->heap[pos] = p->heap[last]; p->hkey[pos] = p->hkey[last];
        p->hpos[p->heap[pos]] = pos;
        if (key_lt(p->hkey[pos], old)) heap_up(p, pos); else heap_down(p, pos);
    }
}
/* PriorityQueue_impl.h:38-42 pop */
static void pq_pop(orc_t *p) { pq_remove_if_present(p, p->heap[0]); }
static void pq_clear(orc_t *p) {
    for (int i = 0; i < p->hn; ++i) p->hpos[p->heap[i]] = -1;
    p->hn = 0;
}

/* ---------------- Graph ------------------------------------------------ */
/* Graph.cpp:53-55 / 61-63 : validity of a Node / Cell */
static inline int valid_elem(const orc_t *p, int x, int y) {
    return x >= 0 && y >= 0 && x < p->nx && y < p->ny;
}
static inline int valid_cell(const orc_t *p, int x, int y) {
    return x >= 0 && x < p->L && y >= 0 && y < p->W;
}
/* Graph.cpp:262-268 get_cost */
static inline float get_cost(const orc_t *p, int cx, int cy) {
    if (!valid_cell(p, cx, cy)) return INFINITY;
    int c = p->map[(size_t)cx * p->W + cy];
    return (c >= p->thr_uchar) ? INFINITY : (float)c;
}
/* ExpandedMap_impl.h:65-74 get_g : missing / out of range => inf */
static inline float get_g(const orc_t *p, int x, int y) {
    if (!valid_elem(p, x, y)) return INFINITY;
    return p->g[(size_t)x * p->ny + y];
}
static inline int eidx(const orc_t *p, int x, int y) { return x * p->ny + y; }
/* ExpandedMap_impl.h:5-14 find_or_init */
static inline void find_or_init(orc_t *p, int e) {
    if (!p->inmap[e]) { p->inmap[e] = 1; p->map_size++; }
}
/* G(s) = v of plan() (FD impl:44,58,84,105 ...), with the bookkeeping of orc_track_changes() */
static inline void set_g(orc_t *p, int s, float v) {
    if (p->chg_on && p->chg_n >= 0 && p->chg_stamp[s] != p->chg_step) { p->chg_stamp[s] = p->chg_step; p->chg_g0[s] = p->g[s]; p->chg_list[p->chg_n++] = s; }
    p->g[s] = v;
}
/* ExpandedMap_impl.h:16-28 insert_or_assign */
static inline void insert_or_assign(orc_t *p, int e, float g, float rhs) {
    find_or_init(p, e);
    p->g[e] = g; p->rhs[e] = rhs;
}

/* 8-neighbourhood ring in counter-clockwise order as defined by the LUTs of
 * Graph.cpp:232-260: top(-1,0) -> top_right(-1,+1) -> right(0,+1) ->
 * bottom_right -> bottom -> bottom_left -> left -> top_left -> top. */
static const int RING_DX[8] = {-1, -1, 0, 1, 1, 1, 0, -1};
static const int RING_DY[8] = {0, 1, 1, 1, 0, -1, -1, -1};
static inline int ring_index(int dx, int dy) {
    static const int idx[3][3] = {{7, 0, 1}, {6, -1, 2}, {5, 4, 3}};
    return idx[dx + 1][dy + 1];
}
/* Graph.cpp:232-245 ccw_neighbor / 247-260 cw_neighbor; returns 0 if invalid */
static inline int ccw_neighbor(const orc_t *p, int sx, int sy, int qx, int qy, int *ox, int *oy) {
    int r = (ring_index(qx - sx, qy - sy) + 1) & 7;
    *ox = sx + RING_DX[r]; *oy = sy + RING_DY[r];
    return valid_elem(p, *ox, *oy);
}
static inline int cw_neighbor(const orc_t *p, int sx, int sy, int qx, int qy, int *ox, int *oy) {
    int r = (ring_index(qx - sx, qy - sy) + 7) & 7;
    *ox = sx + RING_DX[r]; *oy = sy + RING_DY[r];
    return valid_elem(p, *ox, *oy);
}
/* Graph.cpp:71-85 neighbors_8 enumeration order:
 * top, top_left, left, bottom_left, bottom, bottom_right, right, top_right */
static const int N8_DX[8] = {-1, -1, 0, 1, 1, 1, 0, -1};
static const int N8_DY[8] = {0, -1, -1, -1, 0, 1, 1, 1};
/* Graph.cpp:87-101 neighbors_4: top, left, bottom, right */
static const int N4_DX[4] = {-1, 0, 1, 0};
static const int N4_DY[4] = {0, -1, 0, 1};
/* Graph.cpp:103-117 neighbors_diag_4: top_left, bottom_left, top_right, bottom_right */
static const int ND_DX[4] = {-1, 1, -1, 1};
static const int ND_DY[4] = {-1, -1, 1, 1};

/* ---------------- keys -------------------------------------------------- */
/* FieldDPlanner_impl.h:178-186, ShiftedGridPlanner_impl.h:247-256,
 * DynamicFastMarching_impl.h:146-155 */
static okey calc_key_k(const orc_t *p, int x, int y, float cost_so_far) {
    okey k;
    if (!p->heur) { k.k1 = cost_so_far; k.k2 = 0.0f; return k; }
    float dist;
    if (p->algo == ORC_ALGO_DFM) /* Cell::distance, Cell.cpp:67-69 (double hypot of ints) */
        dist = (float)hypot((double)(p->start_cx - x), (double)(p->start_cy - y));
    else /* Position::distance, Position.cpp:29-31 (float hypot) */
        dist = hypotf(p->start_px - (float)x, p->start_py - (float)y);
    k.k1 = cost_so_far + p->hm * dist;
    k.k2 = cost_so_far;
    return k;
}
static okey calc_key(const orc_t *p, int e, float g, float rhs) {
    int x = e / p->ny, y = e % p->ny;
    return calc_key_k(p, x, y, g < rhs ? g : rhs); /* std::min(g, rhs) */
}
/* ReplannerBase.h:110-115 */
static void enqueue_if_inconsistent(orc_t *p, int e) {
    if (!(p->g[e] == p->rhs[e])) pq_insert_or_update(p, e, calc_key(p, e, p->g[e], p->rhs[e]));
    else pq_remove_if_present(p, e);
}

/* ---------------- node planners: traversal cost ------------------------ */
/* CATH / SQUARE, Macros.h:9-12 */
static inline float cath(float x, float y) { return sqrtf((float)(x * x) - (float)(y * y)); }

/* Which case of compute_optimal_cost an evaluation took (ORC_CASE_*, ufm_oracle.h) -- bookkeeping for the question "which
 * branches of the operator does a given run exercise, and which of them ever DECIDE an RHS" (tests/test_reference_mission.py:
 * what the reference's recorded missions pin and what they do not).  Process-wide, not thread-safe: test infrastructure. */
static unsigned long g_case_eval[ORC_NCASES], g_case_won[ORC_NCASES];
static int g_last_case;
/* Ablations of FD's compute_optimal_cost (test hook, tests/test_reference_mission.py: a branch is PINNED by the reference's mission logs
 * if the logs are no longer reproduced without it).  bit 0: without the `f^2 <= CATH(c, b)` clause of Type III; bit 1: without Type I
 * (falls through to A); bit 2: without the whole c > b chain (every triangle through B / II / A as for c <= b); bit 3: Type III pays c instead
 * of b (= Type B: no walking along the edge in the cheaper cell across it). */
static int g_fd_ablate;
void orc_set_fd_ablation(int mask) { g_fd_ablate = mask; }
#define CASE_RET(k, v) do { g_last_case = (k); ++g_case_eval[(k)]; return (v); } while (0)
void orc_case_counts(unsigned long *evaluated, unsigned long *won) {
    for (int i = 0; i < ORC_NCASES; ++i) { if (evaluated) evaluated[i] = g_case_eval[i]; if (won) won[i] = g_case_won[i]; }
}
void orc_case_counts_reset(void) {
    for (int i = 0; i < ORC_NCASES; ++i) g_case_eval[i] = g_case_won[i] = 0;
}

/* FieldDPlanner_impl.h:292-319 + InterpolatedTraversal.cpp:8-10,125-127,
 * 236-238,324-326,403-405 */
static float cost_fd(float g1, float g2, float b, float c) {
    if (g1 == INFINITY && g2 == INFINITY) return INFINITY;
    if (c == INFINITY) return INFINITY;
    float f = g1 - g2;
    if (c > b && !(g_fd_ablate & 4)) {
        if (f <= 0) CASE_RET(ORC_CASE_FD_III, g1 + ((g_fd_ablate & 8) ? c : b));                     /* III */
        else if (!(g_fd_ablate & 1) && (float)(f * f) <= cath(c, b)) CASE_RET(ORC_CASE_FD_III_SQ, g1 + b);   /* III by the f^2 <= CATH(c,b) test [sic] */
        else if ((f <= b) && (c > (f * SQRT2))) CASE_RET(ORC_CASE_FD_II_CGB, g1 + cath(c, f));      /* II  */
        else if (!(g_fd_ablate & 2) && (f > b) && (c > (b * SQRT2))) CASE_RET(ORC_CASE_FD_I, g2 + b + cath(c, b));   /* I   */
        else CASE_RET(ORC_CASE_FD_A_CGB, g2 + c * SQRT2);                                           /* A   */
    } else {
        if (f <= 0) CASE_RET(ORC_CASE_FD_B, g1 + c);                                                /* B   */
        else if ((f * SQRT2) < c) CASE_RET(ORC_CASE_FD_II, g1 + cath(c, f));                        /* II  */
        else CASE_RET(ORC_CASE_FD_A, g2 + c * SQRT2);                                               /* A   */
    }
}
/* ShiftedGridPlanner_impl.h:422-436 */
static float cost_sg(float g1, float g2, float c) {
    if (g1 == INFINITY && g2 == INFINITY) return INFINITY;
    if (c == INFINITY) return INFINITY;
    float f = g1 - g2;
    if (f <= 0) CASE_RET(ORC_CASE_SG_B, g1 + c);
    else if ((f * SQRT2) <= c) CASE_RET(ORC_CASE_SG_II, g1 + cath(c, f));
    else CASE_RET(ORC_CASE_SG_A, g2 + c * SQRT2);
}

/* compute_optimal_cost(n, p_a, p_b, ga, gb): FieldDPlanner_impl.h:269-320 /
 * ShiftedGridPlanner_impl.h:399-437, with fill_traversal_costs
 * (FD :322-337, SG :439-451; Node.cpp:44-57).  One of p_a/p_b is aligned
 * with n (p1), the other is diagonal (p2). */
static float coc_g(const orc_t *p, int nx_, int ny_, int ax, int ay, int bx, int by, float ga, float gb) {
    int cond = (nx_ == ax) || (ny_ == ay); /* Position::aligned */
    int p1x = cond ? ax : bx, p1y = cond ? ay : by;
    int p2x = cond ? bx : ax, p2y = cond ? by : ay;
    float g1 = cond ? ga : gb, g2 = cond ? gb : ga;
    if (g1 == INFINITY && g2 == INFINITY) return INFINITY;
    int ccx, ccy, cbx, cby;
    /* neighbor_cell(bottom_TOP, left_RIGHT) of p1: TOP -> x-1 else x; RIGHT -> y else y-1 */
    if (nx_ == p1x) {
        int lr = ny_ > p1y;
        cbx = (p2x > p1x) ? p1x - 1 : p1x; cby = lr ? p1y : p1y - 1;
        ccx = (p2x < p1x) ? p1x - 1 : p1x; ccy = cby;
    } else {
        int tb = nx_ < p1x;
        cbx = tb ? p1x - 1 : p1x; cby = (p2y < p1y) ? p1y : p1y - 1;
        ccx = cbx;                ccy = (p2y > p1y) ? p1y : p1y - 1;
    }
    float c = get_cost(p, ccx, ccy);
    if (p->algo == ORC_ALGO_SG) return cost_sg(g1, g2, c);
    float b = get_cost(p, cbx, cby);
    return cost_fd(g1, g2, b, c);
}
static float coc(const orc_t *p, int nx_, int ny_, int ax, int ay, int bx, int by) {
    return coc_g(p, nx_, ny_, ax, ay, bx, by, get_g(p, ax, ay), get_g(p, bx, by));
}

/* min_rhs<0>: FieldDPlanner_impl.h:188-194, ShiftedGridPlanner_impl.h:258-264
 * over Graph::consecutive_neighbors(Node), Graph.cpp:202-230 */
static float min_rhs0_node(const orc_t *p, int x, int y) {
    static const int cdx[8] = {1, 1, 0, -1, -1, -1, 0, 1};
    static const int cdy[8] = {0, 1, 1, 1, 0, -1, -1, -1};
    float rhs = INFINITY;
    int win = -1;
    for (int i = 0; i < 8; ++i) {
        int ax = x + cdx[i], ay = y + cdy[i];
        if (valid_elem(p, ax, ay)) {
            int j = (i + 1) & 7;
            int bx = x + cdx[j], by = y + cdy[j];
            if (valid_elem(p, bx, by)) {
                float c = coc(p, x, y, ax, ay, bx, by);
                if (c < rhs) { rhs = c; win = g_last_case; }
            } else ++i; /* Graph.cpp:224-226 */
        }
    }
    if (win >= 0) ++g_case_won[win];
    return rhs;
}
/* min_rhs<1>(s, bptr): FieldDPlanner_impl.h:196-208, ShiftedGridPlanner_impl.h:266-278 */
static float min_rhs1_node(const orc_t *p, int x, int y, int *bptr) {
    float rhs = INFINITY;
    int win = -1;
    for (int i = 0; i < 8; ++i) {
        int qx = x + N8_DX[i], qy = y + N8_DY[i];
        if (!valid_elem(p, qx, qy)) continue;
        int cx, cy;
        if (ccw_neighbor(p, x, y, qx, qy, &cx, &cy)) {
            float cost = coc(p, x, y, qx, qy, cx, cy);
            if (cost < rhs) { rhs = cost; win = g_last_case; }
            if (rhs == cost) *bptr = eidx(p, qx, qy);
        }
    }
    if (win >= 0) ++g_case_won[win];
    return rhs;
}
/* ShiftedGridPlanner_impl.h:280-303 min_rhs<2> */
static float min_rhs2_node(const orc_t *p, int x, int y, int *bptr) {
    float rhs = INFINITY;
    for (int i = 0; i < 4; ++i) {
        int qx = x + ND_DX[i], qy = y + ND_DY[i];
        if (!valid_elem(p, qx, qy)) continue;
        int ccx, ccy, cwx, cwy;
        int v1 = ccw_neighbor(p, x, y, qx, qy, &ccx, &ccy);
        int v2 = cw_neighbor(p, x, y, qx, qy, &cwx, &cwy);
        float ccn_g = v1 ? get_g(p, ccx, ccy) : INFINITY;
        float cn_g = v2 ? get_g(p, cwx, cwy) : INFINITY;
        float sp_g = get_g(p, qx, qy);
        if (v1 && (!v2 || ccn_g <= cn_g)) {
            float cost = coc_g(p, x, y, qx, qy, ccx, ccy, sp_g, ccn_g);
            if (cost < rhs) rhs = cost;
            if (rhs == cost) *bptr = eidx(p, qx, qy);
        } else if (v2 && (!v1 || ccn_g > cn_g)) {
            float cost = coc_g(p, x, y, qx, qy, cwx, cwy, sp_g, cn_g);
            if (cost < rhs) rhs = cost;
            if (rhs == cost) *bptr = eidx(p, cwx, cwy);
        }
    }
    return rhs;
}
/* min_rhs_decreased_neighbor: FieldDPlanner_impl.h:210-223,
 * ShiftedGridPlanner_impl.h:305-333 (<1> and <2>ortho are identical) */
static float min_rhs_decreased_neighbor_node(const orc_t *p, int spx, int spy, int sx, int sy, int *bptr) {
    int ccx, ccy, cwx, cwy;
    int v1 = ccw_neighbor(p, spx, spy, sx, sy, &ccx, &ccy);
    int v2 = cw_neighbor(p, spx, spy, sx, sy, &cwx, &cwy);
    float cost1 = v1 ? coc(p, spx, spy, sx, sy, ccx, ccy) : INFINITY;
    float cost2 = v2 ? coc(p, spx, spy, cwx, cwy, sx, sy) : INFINITY;
    if (cost1 <= cost2) { *bptr = eidx(p, sx, sy); return cost1; }
    *bptr = eidx(p, cwx, cwy);
    return cost2;
}
/* ShiftedGridPlanner_impl.h:335-353 */
static float min_rhs_decreased_diag_neighbor(const orc_t *p, int spx, int spy, int sx, int sy, int *bptr) {
    int ccx, ccy, cwx, cwy;
    int v1 = ccw_neighbor(p, spx, spy, sx, sy, &ccx, &ccy);
    int v2 = cw_neighbor(p, spx, spy, sx, sy, &cwx, &cwy);
    float g_ccn = v1 ? get_g(p, ccx, ccy) : INFINITY;
    float g_cn = v2 ? get_g(p, cwx, cwy) : INFINITY;
    float g_s = get_g(p, sx, sy);
    if (v1 && (!v2 || g_ccn <= g_cn)) {
        *bptr = eidx(p, sx, sy);
        return coc_g(p, spx, spy, sx, sy, ccx, ccy, g_s, g_ccn);
    } else if (v2 && (!v1 || g_ccn > g_cn)) {
        *bptr = eidx(p, cwx, cwy);
        return coc_g(p, spx, spy, sx, sy, cwx, cwy, g_s, g_cn);
    }
    return INFINITY;
}

/* ---------------- DFM --------------------------------------------------- */
/* DynamicFastMarching_impl.h:344-351 best_cell: ties -> b */
static inline void best_cell(const orc_t *p, int ax, int ay, int bx, int by, int *ox, int *oy, float *og) {
    float ca = get_g(p, ax, ay), cb = get_g(p, bx, by);
    if (ca < cb) { *ox = ax; *oy = ay; *og = ca; } else { *ox = bx; *oy = by; *og = cb; }
}
/* DynamicFastMarching_impl.h:322-342; cells given as linear idx or -1 ( Cell() ) */
static float dfm_coc(const orc_t *p, int ca, int cb, float ga1, float gb1, float tau, float h, int *b0, int *b1) {
    (void)p;
    if (ga1 > gb1) { float t = ga1; ga1 = gb1; gb1 = t; int tc = ca; ca = cb; cb = tc; }
    if (ga1 == INFINITY && gb1 == INFINITY) { *b0 = -1; *b1 = -1; return INFINITY; }
    else if ((tau * h) > (gb1 - ga1)) {
        *b0 = ca; *b1 = cb;
        float th = tau * h;
        float d = gb1 - ga1;
        return (ga1 + gb1 + sqrtf((float)(2 * (float)(th * th) - (float)(d * d)))) * 0.5f;
    } else { *b0 = ca; *b1 = -1; return ga1 + tau * h; }
}
/* linear index of a cell used as back-pointer; -2 for an out-of-grid cell (never equals a real one) */
static inline int cidx(const orc_t *p, int x, int y) { return valid_elem(p, x, y) ? eidx(p, x, y) : -2; }
/* min_rhs<0>/<1>: DynamicFastMarching_impl.h:157-210 / 212-268 */
static float dfm_min_rhs(const orc_t *p, int x, int y, int *b0, int *b1) {
    float tau = get_cost(p, x, y);
    *b0 = -1; *b1 = -1;
    if (tau == INFINITY) return INFINITY;
    int ax, ay, bx, by; float ga, gb;
    int o0, o1, d0, d1;
    best_cell(p, x - 1, y, x + 1, y, &ax, &ay, &ga);       /* top, bottom */
    best_cell(p, x, y - 1, x, y + 1, &bx, &by, &gb);       /* left, right */
    float so = dfm_coc(p, cidx(p, ax, ay), cidx(p, bx, by), ga, gb, tau, 1.0f, &o0, &o1);
    best_cell(p, x - 1, y - 1, x + 1, y + 1, &ax, &ay, &ga); /* top_left, bottom_right */
    best_cell(p, x + 1, y - 1, x - 1, y + 1, &bx, &by, &gb); /* bottom_left, top_right */
    float sd = dfm_coc(p, cidx(p, ax, ay), cidx(p, bx, by), ga, gb, tau, SQRT2, &d0, &d1);
    if (sd < so) { *b0 = d0; *b1 = d1; return sd; }
    *b0 = o0; *b1 = o1;
    return so;
}
/* DynamicFastMarching_impl.h:270-313 */
static float dfm_min_rhs_decreased_neighbor(const orc_t *p, int x, int y, int nx_, int ny_, int *b0, int *b1) {
    float tau = get_cost(p, x, y);
    *b0 = -1; *b1 = -1;
    if (tau == INFINITY) return INFINITY;
    float ga = get_g(p, nx_, ny_), gb;
    int bx, by;
    int dx = nx_ - x, dy = ny_ - y;
    if (dx * dy == 0) {
        if (dx != 0) best_cell(p, x, y - 1, x, y + 1, &bx, &by, &gb);
        else best_cell(p, x - 1, y, x + 1, y, &bx, &by, &gb);
    } else {
        if (dx != dy) best_cell(p, x - 1, y - 1, x + 1, y + 1, &bx, &by, &gb);
        else best_cell(p, x + 1, y - 1, x - 1, y + 1, &bx, &by, &gb);
    }
    return dfm_coc(p, cidx(p, nx_, ny_), cidx(p, bx, by), ga, gb, tau, hypotf((float)dx, (float)dy), b0, b1);
}

/* ---------------- end conditions --------------------------------------- */
/* FieldDPlanner_impl.h:225-256 / ShiftedGridPlanner_impl.h:355-386 */
static int end_condition_node(const orc_t *p) {
    okey top_key = p->hkey[0];
    okey max_start_key = {0.0f, 0.0f};
    for (int i = 0; i < p->n_start_nodes; ++i) {
        int e = p->start_nodes[i];
        okey key = calc_key(p, e, p->g[e], p->rhs[e]);
        float k = key.k1;
        if (p->rhs[e] != INFINITY && k != INFINITY) {
            if (key_lt(max_start_key, key)) max_start_key = key; /* std::max */
            if (p->rhs[e] > p->g[e]) return 0;
        }
    }
    if (max_start_key.k1 == 0) return 0;
    return key_le(max_start_key, top_key);
}
/* DynamicFastMarching_impl.h:315-320 */
static int end_condition_dfm(const orc_t *p) {
    int e = eidx(p, p->start_cx, p->start_cy);
    okey top_key = p->hkey[0];
    return (p->g[e] == p->rhs[e]) && key_le(calc_key(p, e, p->g[e], p->rhs[e]), top_key);
}

/* ---------------- init / plan / update --------------------------------- */
/* FieldDPlanner_impl.h:15-21, ShiftedGridPlanner_impl.h:9-15, DynamicFastMarching_impl.h:6-11 */
static void planner_init(orc_t *p) {
    if (p->algo == ORC_ALGO_DFM) {
        insert_or_assign(p, eidx(p, p->start_cx, p->start_cy), INFINITY, INFINITY);
        int ge = eidx(p, p->goal_cx, p->goal_cy);
        insert_or_assign(p, ge, INFINITY, 0.0f);
        pq_insert_or_update(p, ge, calc_key_k(p, p->goal_cx, p->goal_cy, 0.0f));
    } else {
        for (int i = 0; i < p->n_start_nodes; ++i) insert_or_assign(p, p->start_nodes[i], INFINITY, INFINITY);
        int ge = eidx(p, p->goal_nx, p->goal_ny);
        insert_or_assign(p, ge, INFINITY, 0.0f);
        pq_insert_or_update(p, ge, calc_key_k(p, p->goal_nx, p->goal_ny, 0.0f));
    }
}
static inline int goal_elem(const orc_t *p) {
    return p->algo == ORC_ALGO_DFM ? eidx(p, p->goal_cx, p->goal_cy) : eidx(p, p->goal_nx, p->goal_ny);
}
static float elem_min_rhs0(const orc_t *p, int x, int y) {
    int b0, b1;
    return p->algo == ORC_ALGO_DFM ? dfm_min_rhs(p, x, y, &b0, &b1) : min_rhs0_node(p, x, y);
}

/* plan<0>: FieldDPlanner_impl.h:23-66, ShiftedGridPlanner_impl.h:17-60,
 * DynamicFastMarching_impl.h:13-54 */
/* Safety net of the TEST INFRASTRUCTURE, not of the algorithm: the reference's loop has no bound, and
 * with DFM's upwind quadratic -- not monotone at the ulp level -- an element can flip between over-
 * and under-consistent for ever on noise-like maps (seen: 432x113 white noise, 20 % obstacles).  A
 * step that expands more than 256 x the number of elements stops and orc_step returns -75. */
static unsigned long expansion_cap(const orc_t *p) { return 256ul * (unsigned long)p->n + 100000ul; }
static void plan0(orc_t *p) {
    unsigned long expanded = 0;
    const int goal = goal_elem(p);
    if (p->algo == ORC_ALGO_DFM) find_or_init(p, eidx(p, p->start_cx, p->start_cy));
    else for (int i = 0; i < p->n_start_nodes; ++i) find_or_init(p, p->start_nodes[i]);
    while (p->hn > 0 && !(p->algo == ORC_ALGO_DFM ? end_condition_dfm(p) : end_condition_node(p))) {
        int s = p->heap[0];
        pq_pop(p);
        if (++expanded > expansion_cap(p)) { p->runaway = 1; break; }
        int sx = s / p->ny, sy = s % p->ny;
        int under = !(p->g[s] > p->rhs[s]);
        set_g(p, s, under ? INFINITY : p->rhs[s]);
        for (int i = 0; i < 8; ++i) {
            int qx = sx + N8_DX[i], qy = sy + N8_DY[i];
            if (!valid_elem(p, qx, qy)) continue;
            int q = eidx(p, qx, qy);
            find_or_init(p, q);
            if (q != goal) p->rhs[q] = elem_min_rhs0(p, qx, qy);
            enqueue_if_inconsistent(p, q);
        }
        if (under) {
            if (s != goal) p->rhs[s] = elem_min_rhs0(p, sx, sy);
            enqueue_if_inconsistent(p, s);
        }
    }
    p->num_expanded = expanded;
}

/* plan<1>/<2> for FD / SG: FieldDPlanner_impl.h:68-116, ShiftedGridPlanner_impl.h:62-170 */
static void plan12_node(orc_t *p) {
    unsigned long expanded = 0;
    int bptr = 0;
    for (int i = 0; i < p->n_start_nodes; ++i) find_or_init(p, p->start_nodes[i]);
    while (p->hn > 0 && !end_condition_node(p)) {
        if (expanded > expansion_cap(p)) { p->runaway = 1; break; }
        int s = p->heap[0];
        ++expanded;
        int sx = s / p->ny, sy = s % p->ny;
        if (p->g[s] > p->rhs[s]) {
            set_g(p, s, p->rhs[s]);
            pq_pop(p);
            if (p->lvl == 2) { /* SG<2>: diagonal then orthogonal neighbours */
                for (int i = 0; i < 4; ++i) {
                    int qx = sx + ND_DX[i], qy = sy + ND_DY[i];
                    if (!valid_elem(p, qx, qy)) continue;
                    int q = eidx(p, qx, qy);
                    find_or_init(p, q);
                    float rhs = min_rhs_decreased_diag_neighbor(p, qx, qy, sx, sy, &bptr);
                    if (rhs < p->rhs[q]) { p->rhs[q] = rhs; p->bptr[q] = bptr; }
                    enqueue_if_inconsistent(p, q);
                }
                for (int i = 0; i < 4; ++i) {
                    int qx = sx + N4_DX[i], qy = sy + N4_DY[i];
                    if (!valid_elem(p, qx, qy)) continue;
                    int q = eidx(p, qx, qy);
                    find_or_init(p, q);
                    float rhs = min_rhs_decreased_neighbor_node(p, qx, qy, sx, sy, &bptr);
                    if (rhs < p->rhs[q]) { p->rhs[q] = rhs; p->bptr[q] = bptr; }
                    enqueue_if_inconsistent(p, q);
                }
            } else {
                for (int i = 0; i < 8; ++i) {
                    int qx = sx + N8_DX[i], qy = sy + N8_DY[i];
                    if (!valid_elem(p, qx, qy)) continue;
                    int q = eidx(p, qx, qy);
                    find_or_init(p, q);
                    float rhs = min_rhs_decreased_neighbor_node(p, qx, qy, sx, sy, &bptr);
                    if (rhs < p->rhs[q]) { p->rhs[q] = rhs; p->bptr[q] = bptr; }
                    enqueue_if_inconsistent(p, q);
                }
            }
        } else {
            set_g(p, s, INFINITY);
            for (int i = 0; i < 8; ++i) {
                int qx = sx + N8_DX[i], qy = sy + N8_DY[i];
                if (!valid_elem(p, qx, qy)) continue;
                int q = eidx(p, qx, qy);
                /* reference asserts the neighbour exists (map.find); a missing one has
                 * a default INFO = Node(0,0) which we reproduce via bptr[] init 0 */
                int cwx, cwy;
                int has_cw = cw_neighbor(p, qx, qy, sx, sy, &cwx, &cwy);
                if (p->bptr[q] == s || (has_cw && p->bptr[q] == eidx(p, cwx, cwy))) {
                    find_or_init(p, q);
                    p->rhs[q] = (p->lvl == 2) ? min_rhs2_node(p, qx, qy, &bptr) : min_rhs1_node(p, qx, qy, &bptr);
                    if (p->rhs[q] < INFINITY) p->bptr[q] = bptr;
                    enqueue_if_inconsistent(p, q);
                }
            }
            enqueue_if_inconsistent(p, s);
        }
    }
    p->num_expanded = expanded;
}

/* DynamicFastMarching_impl.h:56-104 */
static void plan1_dfm(orc_t *p) {
    unsigned long expanded = 0;
    const int goal = goal_elem(p);
    int b0, b1;
    find_or_init(p, eidx(p, p->start_cx, p->start_cy));
    while (p->hn > 0 && !end_condition_dfm(p)) {
        if (expanded > expansion_cap(p)) { p->runaway = 1; break; }
        int s = p->heap[0];
        ++expanded;
        int sx = s / p->ny, sy = s % p->ny;
        if (p->g[s] > p->rhs[s]) {
            set_g(p, s, p->rhs[s]);
            pq_pop(p);
            for (int i = 0; i < 8; ++i) {
                int qx = sx + N8_DX[i], qy = sy + N8_DY[i];
                if (!valid_elem(p, qx, qy)) continue;
                int q = eidx(p, qx, qy);
                find_or_init(p, q);
                if (q != goal) {
                    float rhs = dfm_min_rhs_decreased_neighbor(p, qx, qy, sx, sy, &b0, &b1);
                    if (rhs < p->rhs[q]) { p->rhs[q] = rhs; p->bptr[2 * q] = b0; p->bptr[2 * q + 1] = b1; }
                }
                enqueue_if_inconsistent(p, q);
            }
        } else {
            set_g(p, s, INFINITY);
            for (int i = 0; i < 8; ++i) {
                int qx = sx + N8_DX[i], qy = sy + N8_DY[i];
                if (!valid_elem(p, qx, qy)) continue;
                int q = eidx(p, qx, qy);
                if (p->bptr[2 * q] == s || p->bptr[2 * q + 1] == s) {
                    find_or_init(p, q);
                    if (q != goal) p->rhs[q] = dfm_min_rhs(p, qx, qy, &p->bptr[2 * q], &p->bptr[2 * q + 1]);
                    enqueue_if_inconsistent(p, q);
                }
            }
            enqueue_if_inconsistent(p, s);
        }
    }
    p->num_expanded = expanded;
}

/* update(): FieldDPlanner_impl.h:118-163, ShiftedGridPlanner_impl.h:172-231,
 * DynamicFastMarching_impl.h:106-132 */
static void update_elem(orc_t *p, int e) {
    const int goal = goal_elem(p);
    int x = e / p->ny, y = e % p->ny;
    find_or_init(p, e);
    if (p->algo == ORC_ALGO_DFM) {
        if (p->lvl == 0) { if (e != goal) p->rhs[e] = elem_min_rhs0(p, x, y); }
        else if (e != goal) p->rhs[e] = dfm_min_rhs(p, x, y, &p->bptr[2 * e], &p->bptr[2 * e + 1]);
        enqueue_if_inconsistent(p, e);
        return;
    }
    if (p->lvl == 0) {
        if (e != goal) p->rhs[e] = min_rhs0_node(p, x, y);
        enqueue_if_inconsistent(p, e);
    } else if (e != goal) {
        int bptr = 0;
        p->rhs[e] = (p->lvl == 2) ? min_rhs2_node(p, x, y, &bptr) : min_rhs1_node(p, x, y, &bptr);
        if (p->rhs[e] < INFINITY) p->bptr[e] = bptr;
        enqueue_if_inconsistent(p, e);
    }
}
static void planner_update(orc_t *p) {
    if (p->algo == ORC_ALGO_DFM) {
        for (int i = 0; i < p->n_upd; ++i) update_elem(p, p->upd_cells[i]); /* cell idx == elem idx */
        p->num_updated = (unsigned long)p->n_upd;
        return;
    }
    if (p->heur) { /* FD impl:119-126 re-key the whole queue on start move */
        for (int i = 0; i < p->hn; ++i) {
            int e = p->heap[i];
            p->hkey[i] = calc_key_k(p, e / p->ny, e % p->ny, p->hkey[i].k2);
        }
        for (int i = p->hn / 2 - 1; i >= 0; --i) heap_down(p, i);
    }
    /* distinct corner nodes of the updated cells (Cell.cpp:48-60) */
    unsigned long cnt = 0;
    int *list = (int *)malloc(sizeof(int) * (size_t)(4 * p->n_upd + 1));
    uint8_t *mark = p->inmap; /* reuse bit 1 as a scratch 'seen' mark */
    for (int i = 0; i < p->n_upd; ++i) {
        int cx = p->upd_cells[i] / p->W, cy = p->upd_cells[i] % p->W;
        int cs[4] = {eidx(p, cx, cy), eidx(p, cx + 1, cy), eidx(p, cx, cy + 1), eidx(p, cx + 1, cy + 1)};
        for (int k = 0; k < 4; ++k) {
            /* the recorded missions' revision left out the corner nodes on the map's far borders (x == L, y == W): orc_set_revision */
            if ((p->revision & ORC_REV_UPDATE_SKIPS_FAR_BORDER) && (cs[k] / p->ny == p->nx - 1 || cs[k] % p->ny == p->ny - 1)) continue;
            if (!(mark[cs[k]] & 2)) { mark[cs[k]] |= 2; list[cnt++] = cs[k]; }
        }
    }
    for (unsigned long i = 0; i < cnt; ++i) mark[list[i]] &= 1;
    for (unsigned long i = 0; i < cnt; ++i) update_elem(p, list[i]);
    free(list);
    p->num_updated = cnt;
}

/* ---------------- public API ------------------------------------------- */
orc_t *orc_create(int algo, int opt_lvl, int use_heuristic) {
    if (algo < 0 || algo > 2 || opt_lvl < 0 || opt_lvl > 2) return NULL;
    if (algo != ORC_ALGO_SG && opt_lvl > 1) return NULL;
    orc_t *p = (orc_t *)calloc(1, sizeof(orc_t));
    p->algo = algo; p->lvl = opt_lvl; p->heur = use_heuristic ? 1 : 0;
    p->thr_uchar = 254;
    p->hm = 1.0f;
    p->initialize_graph = 1; p->initialize_search = 1;
    /* Cell() default is (-1,-1) (Cell.cpp:10), Node() is (0,0) */
    p->start_cx = p->start_cy = p->goal_cx = p->goal_cy = -1;
    return p;
}
static void free_state(orc_t *p) {
    free(p->g); free(p->rhs); free(p->inmap); free(p->bptr);
    free(p->heap); free(p->hkey); free(p->hpos);
    free(p->chg_stamp); free(p->chg_g0); free(p->chg_list); p->chg_stamp = NULL; p->chg_g0 = NULL; p->chg_list = NULL; p->chg_on = 0;
    p->g = p->rhs = NULL; p->inmap = NULL; p->bptr = NULL; p->heap = NULL; p->hkey = NULL; p->hpos = NULL;
}
void orc_destroy(orc_t *p) {
    if (!p) return;
    free_state(p);
    free(p->map); free(p->upd_cells);
    free(p);
}
void orc_reset(orc_t *p) { p->initialize_search = 1; }                         /* ReplannerBase.h:39-41 */
void orc_set_occupancy_threshold(orc_t *p, float thr) { p->thr_uchar = (int)(thr * 255.0f); } /* Graph.cpp:18-20 */
void orc_set_heuristic_multiplier(orc_t *p, float m) { p->hm = m; }           /* ReplannerBase.h:81-83 */

void orc_set_map(orc_t *p, const uint8_t *map, int width, int length) {       /* ReplannerBase.h:85-88, Graph.cpp:22-29 */
    free(p->map);
    p->W = width; p->L = length;
    p->map = (uint8_t *)malloc((size_t)width * length);
    memcpy(p->map, map, (size_t)width * length);
    p->initialize_graph = 0;
    int nx = p->algo == ORC_ALGO_DFM ? length : length + 1;
    int ny = p->algo == ORC_ALGO_DFM ? width : width + 1;
    if (nx != p->nx || ny != p->ny || !p->g) {
        free_state(p);
        p->nx = nx; p->ny = ny; p->n = (size_t)nx * ny;
        p->g = (float *)malloc(sizeof(float) * p->n);
        p->rhs = (float *)malloc(sizeof(float) * p->n);
        p->inmap = (uint8_t *)calloc(p->n, 1);
        if (p->lvl > 0) p->bptr = (int32_t *)malloc(sizeof(int32_t) * p->n * (p->algo == ORC_ALGO_DFM ? 2 : 1));
        p->heap = (int *)malloc(sizeof(int) * p->n);
        p->hkey = (okey *)malloc(sizeof(okey) * p->n);
        p->hpos = (int *)malloc(sizeof(int) * p->n);
        for (size_t i = 0; i < p->n; ++i) { p->g[i] = INFINITY; p->rhs[i] = INFINITY; p->hpos[i] = -1; }
        p->hn = 0; p->map_size = 0;
        p->initialize_search = 1;
    }
}
void orc_patch_map(orc_t *p, const uint8_t *patch, int x, int y, int w, int h) { /* Graph.cpp:36-51 */
    p->n_upd = 0;
    for (int i = 0; i < h; ++i)
        for (int j = 0; j < w; ++j) {
            size_t ci = (size_t)(i + x) * p->W + (j + y);
            uint8_t pv = patch[(size_t)i * w + j];
            if (p->map[ci] != pv) {
                if (p->n_upd == p->cap_upd) {
                    p->cap_upd = p->cap_upd ? 2 * p->cap_upd : 1024;
                    p->upd_cells = (int *)realloc(p->upd_cells, sizeof(int) * (size_t)p->cap_upd);
                }
                p->upd_cells[p->n_upd++] = (int)ci;
            }
            p->map[ci] = pv;
        }
}
static void refresh_start_nodes(orc_t *p) {
    /* FieldDPlanner_impl.h:9-13 start_nodes = start_cell_.corners();
     * Cell::corners(), Cell.cpp:56-58: (x,y) (x+1,y) (x,y+1) (x+1,y+1) */
    p->n_start_nodes = 0;
    if (p->algo == ORC_ALGO_DFM || !p->start_set) return;
    const int cx = p->start_cx, cy = p->start_cy;
    const int xs[4] = {cx, cx + 1, cx, cx + 1}, ys[4] = {cy, cy, cy + 1, cy + 1};
    for (int k = 0; k < 4; ++k)
        if (valid_elem(p, xs[k], ys[k])) p->start_nodes[p->n_start_nodes++] = eidx(p, xs[k], ys[k]);
}
/* Which revision of the reference the restatement follows.  ORC_REV_CURRENT (default): the sources as they stand under
 * /root/reference.  ORC_REV_LOG: the revision that wrote the two mission logs the reference holds (Tests/Results/{noise-trap,wall-b}/
 * planner_opt0.log -- they print lines the current sources have commented out, FieldDPlanner_impl.h:65,139).  Two differences were
 * identified by search against the logs (tools/mission_revision_probe.py), both outside the operators:
 *   (1) ORC_REV_START_CELL_FLOOR: start_cell_ = Cell(floor(x), floor(y)) -- the cell that CONTAINS the position -- where Cell(const Position&) now rounds
 *       (Cell.cpp:20-21): the four start nodes of end_condition() (FD impl:225-256) differ whenever a coordinate's fraction is >= 0.5;
 *       with it the restatement's "nodes expanded" equals the log's in 104 of 105 + 73 of 73 replans (with roundf: 84 + 8);
 *   (2) ORC_REV_UPDATE_SKIPS_FAR_BORDER: update() (FD impl:118-140) did not take the corner nodes that lie on the map's far borders (x == length, y == width) into its
 *       set: "nodes updated" is 2 smaller there in exactly the steps whose changed cells touch the bottom row / right column; with it
 *       133 of 133 + 88 of 88 agree.
 * With both, the restatement replays BOTH logs closed-loop to the last printed digit -- 134 + 89 steps: position, path cost, path
 * length, nodes updated, and every "nodes expanded" but one (noise-trap step 119: 272 against 273). */
void orc_set_revision(orc_t *p, int revision) { p->revision = revision; }
void orc_set_start(orc_t *p, float x, float y) { /* Graph.cpp:6-10, ReplannerBase.h:94-97 */
    p->start_px = x; p->start_py = y;
    p->start_cx = (int)roundf(x); p->start_cy = (int)roundf(y);            /* Cell(Position), Cell.cpp:20-21 */
    if (p->revision & ORC_REV_START_CELL_FLOOR) { p->start_cx = (int)floorf(x); p->start_cy = (int)floorf(y); }   /* the recorded missions' revision: see orc_set_revision */
    p->start_nx = (int)roundf(x); p->start_ny = (int)roundf(y);
    p->new_start = 1;
    p->start_set = 1;
}
void orc_set_goal(orc_t *p, float x, float y) { /* ReplannerBase.h:99-108, Graph.cpp:12-16 */
    int rx = (int)roundf(x), ry = (int)roundf(y);
    if (p->algo == ORC_ALGO_DFM) p->new_goal = (p->goal_cx != rx || p->goal_cy != ry);
    else p->new_goal = (p->goal_nx != rx || p->goal_ny != ry);
    p->goal_px = x; p->goal_py = y;
    p->goal_cx = p->goal_nx = rx; p->goal_cy = p->goal_ny = ry;
    p->goal_set = 1;
}
static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
/* Test / benchmark bookkeeping, not part of the reference: from now on every step records which elements' G it assigns; orc_num_changed() = how
 * many of them hold another value after the step than before it (for a step that (re)initialises the search: how many hold a value). */
void orc_track_changes(orc_t *p, int on) {
    p->chg_on = on ? 1 : 0;
    if (p->chg_on && !p->chg_stamp && p->n) {
        p->chg_stamp = (uint32_t *)calloc(p->n, sizeof(uint32_t));
        p->chg_g0 = (float *)malloc(p->n * sizeof(float));
        p->chg_list = (int *)malloc(p->n * sizeof(int));
    }
}
unsigned long orc_num_changed(const orc_t *p) {
    unsigned long c = 0;
    if (!p->chg_on) return 0;
    if (p->chg_n < 0) { for (size_t i = 0; i < p->n; ++i) c += p->g[i] < INFINITY; return c; }      /* the step initialised the search */
    for (int i = 0; i < p->chg_n; ++i) { const int s = p->chg_list[i]; c += memcmp(&p->g[s], &p->chg_g0[s], sizeof(float)) != 0; }
    return c;
}
int orc_step(orc_t *p) { /* ReplannerBase.h:43-75 */
    if (p->initialize_graph) return ORC_LOOP_FAILURE_NO_GRAPH;
    if (!p->goal_set) return ORC_LOOP_FAILURE_NO_GOAL;
    double t0 = now_ms();
    if (p->chg_on) { ++p->chg_step; p->chg_n = (p->initialize_search || p->new_goal) ? -1 : 0; }
    refresh_start_nodes(p);
    if (p->initialize_search || p->new_goal) {
        p->num_updated = 0; p->num_expanded = 0;
        pq_clear(p);
        p->n_upd = 0;
        if (p->map_size) {
            for (size_t i = 0; i < p->n; ++i) { p->g[i] = INFINITY; p->rhs[i] = INFINITY; }
            memset(p->inmap, 0, p->n);
            p->map_size = 0;
        }
        if (p->bptr) { /* default INFO: Node() = (0,0) -> index 0; pair<Cell,Cell>() = (-1,-1) */
            size_t nb = p->n * (p->algo == ORC_ALGO_DFM ? 2 : 1);
            int32_t dv = p->algo == ORC_ALGO_DFM ? -1 : 0;
            for (size_t i = 0; i < nb; ++i) p->bptr[i] = dv;
        }
        planner_init(p);
    } else if (p->new_start) {
        p->new_start = 0;
        planner_update(p);
    }
    double t1 = now_ms();
    p->u_time = (float)(t1 - t0);
    p->runaway = 0;
    if (p->new_goal || p->initialize_search || p->num_updated > 0) {
        if (p->lvl == 0) plan0(p);
        else if (p->algo == ORC_ALGO_DFM) plan1_dfm(p);
        else plan12_node(p);
    } else p->num_expanded = 0;
    p->new_goal = p->initialize_search = 0;
    p->p_time = (float)(now_ms() - t1);
    return p->runaway ? ORC_LOOP_RUNAWAY : ORC_LOOP_OK;
}

int orc_field_dims(const orc_t *p, int *nx, int *ny) { *nx = p->nx; *ny = p->ny; return 0; }
const float *orc_g(const orc_t *p) { return p->g; }
const float *orc_rhs(const orc_t *p) { return p->rhs; }
const uint8_t *orc_inmap(const orc_t *p) { return p->inmap; }
const int32_t *orc_bptr(const orc_t *p) { return p->bptr; }
unsigned long orc_num_expanded(const orc_t *p) { return p->num_expanded; }
unsigned long orc_num_updated(const orc_t *p) { return p->num_updated; }
unsigned long orc_map_size(const orc_t *p) { return p->map_size; }
unsigned long orc_queue_size(const orc_t *p) { return (unsigned long)p->hn; }
float orc_u_time_ms(const orc_t *p) { return p->u_time; }
float orc_p_time_ms(const orc_t *p) { return p->p_time; }
void orc_top_key(const orc_t *p, float *k1, float *k2) {
    if (p->hn > 0) { *k1 = p->hkey[0].k1; *k2 = p->hkey[0].k2; }
    else { *k1 = INFINITY; *k2 = INFINITY; }
}

/* path extraction on this planner's field (LinearInterpolationPathExtractor reads map.get_interp_rhs
 * and grid.get_cost / start_pos_ / goal_pos_, PathExtraction impl:76-77, 183, 216, 232-233) */
int orc_threshold_uchar(const orc_t *p) { return p->thr_uchar; }
int orc_extract_path(const orc_t *p, int lookahead, int max_steps, int allow_indirect,
                     float *path_xy, int cap_pts, float *costs, int cap_costs,
                     int *n_costs, float *total_cost, float *total_dist) {
    return orc_extract_path_field(p->rhs, p->nx, p->ny, p->algo == ORC_ALGO_DFM, p->map, p->W, p->L, p->thr_uchar,
                                  p->start_px, p->start_py, p->goal_px, p->goal_py, lookahead, max_steps,
                                  allow_indirect, path_xy, cap_pts, costs, cap_costs, n_costs, total_cost, total_dist);
}

/* ---- back-pointer (INFO) of an element as the level-1/2 min_rhs computes it from the current G
 * field (FieldDPlanner_impl.h:196-208, ShiftedGridPlanner_impl.h:266-303,
 * DynamicFastMarching_impl.h:212-268): a pure function of the field, used to check the engine's
 * ufm_read_info.  Node planners: *b0 = linear index of the node bptr (RHS = cost over the edge
 * bptr -> ccw_neighbor(bptr)), *b1 = -1; DFM: the two cells of the winning stencil (-1: none, -2:
 * outside the grid).  Returns the RHS. */
float orc_min_rhs_info(const orc_t *p, int x, int y, int32_t *b0, int32_t *b1) {
    int a = -1, b = -1;
    float rhs;
    if (p->algo == ORC_ALGO_DFM) rhs = dfm_min_rhs(p, x, y, &a, &b);
    else if (p->algo == ORC_ALGO_SG && p->lvl == 2) rhs = min_rhs2_node(p, x, y, &a);
    else rhs = min_rhs1_node(p, x, y, &a);
    *b0 = a; *b1 = b;
    return rhs;
}
/* The cost through a GIVEN back-pointer on the current G field (checker of the engine's stored back-pointers, which may name
 * another candidate than min_rhs where two of them tie).  Node planners: compute_optimal_cost(s, b, ccw_neighbor(s, b))
 * (FieldDPlanner_impl.h:200-205); *b0 = b, *b1 = -1.  DFM: the level-1 candidate built on the neighbour cell (bx, by),
 * min_rhs_decreased_neighbor (DynamicFastMarching_impl.h:270-313), with the pair of cells it leaves. */
float orc_cost_via(const orc_t *p, int x, int y, int bx, int by, int32_t *b0, int32_t *b1) {
    int a = -1, b = -1;
    float c = INFINITY;
    if (p->algo == ORC_ALGO_DFM) c = dfm_min_rhs_decreased_neighbor(p, x, y, bx, by, &a, &b);
    else if (valid_elem(p, bx, by)) {
        int cx, cy;
        if (ccw_neighbor(p, x, y, bx, by, &cx, &cy)) { c = coc(p, x, y, bx, by, cx, cy); a = eidx(p, bx, by); }
    }
    *b0 = a; *b1 = b;
    return c;
}
/* test hook: overwrite the G field (e.g. with one read back from the engine) */
void orc_load_g(orc_t *p, const float *g) { memcpy(p->g, g, p->n * sizeof(float)); }
