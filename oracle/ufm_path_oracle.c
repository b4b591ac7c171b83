/*
 * ufm_path_oracle.c -- TEST INFRASTRUCTURE ONLY (parity checker), see ufm_oracle.h.
 *
 * CPU restatement, in plain C, of the reference's path extraction (the consumer of the
 * RHS field): PathExtraction/LinearInterpolationPathExtractor_impl.h and the traversal
 * case tables of ProjectToolkit/InterpolatedTraversal.cpp.  Every function cites the
 * reference lines it follows.  It works on plain arrays (a dense RHS field, the cost
 * raster) so that it can be run on the oracle planner's field and on a field read back
 * from the GPU engine alike.
 *
 * Pinning: the known answers SURVEY.md App. E recorded from the reference for its own
 * noise-trap bitmap (points / total_cost / total_dist for DFM, SG, FD with
 * max_steps = 800) are reproduced by tests/test_oracle.py::test_noise_trap_path_known_answers.
 */
#include "ufm_oracle.h"

#include <math.h>
#include <stddef.h>
#ifdef ORC_PATH_TRACE   /* diagnostics (tools/wallb_probe.py): every candidate of getPathAdditions on stderr */
#include <stdio.h>
#endif

/* ProjectToolkit/Macros.cpp:2 */
static const float SQRT2 = 1.41421356237309504880168872420969807856967187537694f;

typedef struct { float x, y; } pos_t;   /* Position.h */
typedef struct { int x, y; } nd_t;      /* Node.h */

/* InterpolatedTraversal.h:11-26 */
typedef struct {
    pos_t p0;
    nd_t p1, p2;
    float b, c, f, g1, g2, p, q;
} tparams;

/* InterpolatedTraversal.h:32-39 ; at most 3 positions / 2 step costs per traversal */
typedef struct {
    pos_t steps[3];
    float costs[2];
    int ns, nc;
    float cost_to_goal;
} padd;

typedef struct {
    const float *rhs;
    int nx, ny, cells;
    const uint8_t *map;
    int W, L, thr;
    pos_t start, goal;
    int indirect;
} pctx;

enum { K_CORNER = 0, K_CONTIG = 1, K_OPP = 2 };
enum { T_I = 0, T_II = 1, T_III = 2, T_A = 3, T_B = 4 };

/* Macros.h:9-12,18,24 */
static inline float sq(float x) { return x * x; }
static inline float cath(float x, float y) { return sqrtf((float)(sq(x) - sq(y))); }
static inline float hyp(float x, float y) { return hypotf(x, y); }
static inline float interp1(float from, float to, float d) { return from + (to - from) * d; }
static inline float interp_abs_f(float from, float to, float d) { return from + (to - from) / fabsf(to - from) * d; }
/* integer operands: (to-from)/abs(to-from) is an integer division (+-1) before the product */
static inline float interp_abs_i(int from, int to, float d) {
    int s = (to - from) / (to - from < 0 ? from - to : to - from);
    return (float)from + (float)s * d;
}
static inline float interp1_i(int from, int to, float d) { return (float)from + (float)(to - from) * d; }
static inline pos_t mkpos(float x, float y) { pos_t r = {x, y}; return r; }
static inline pos_t npos(nd_t n) { pos_t r = {(float)n.x, (float)n.y}; return r; }

/* Graph.cpp:262-268 */
static float p_cost(const pctx *c, int cx, int cy) {
    if (cx < 0 || cx >= c->L || cy < 0 || cy >= c->W) return INFINITY;
    int v = c->map[(size_t)cx * c->W + cy];
    return (v >= c->thr) ? INFINITY : (float)v;
}
/* ExpandedMap_impl.h:76-85 get_rhs: unknown / out of range => inf */
static float p_rhs(const pctx *c, int x, int y) {
    if (x < 0 || y < 0 || x >= c->nx || y >= c->ny) return INFINITY;
    return c->rhs[(size_t)x * c->ny + y];
}
/* ExpandedMap_impl.h:87-97 get_interp_rhs: node maps read the node; cell maps average the
 * four cells around the node in the order bottom, self, bottom-right, right of (x-1,y-1) */
static float p_interp_rhs(const pctx *c, nd_t s) {
    if (!c->cells) return p_rhs(c, s.x, s.y);
    const int px = s.x - 1, py = s.y - 1;
    float a = p_rhs(c, px + 1, py), b = p_rhs(c, px, py), cc = p_rhs(c, px + 1, py + 1), d = p_rhs(c, px, py + 1);
    return (a + b + cc + d) * 0.25f;
}
/* Graph.cpp:53-55, 57-59, 65-69 */
static int p_valid_node(const pctx *c, int x, int y) { return x <= c->L && y <= c->W && x >= 0 && y >= 0; }
static int p_valid_vertex(const pctx *c, pos_t p) {
    return ceilf(p.x) == p.x && ceilf(p.y) == p.y && p.x >= 0.0f && p.x <= (float)c->L && p.y >= 0.0f && p.y <= (float)c->W;
}

/* Graph.cpp:151-200 consecutive_neighbors(Position): ring of 6 (point on an edge) or 8 nodes,
 * pairs of consecutive valid nodes; a valid node followed by an invalid one skips a slot */
static int p_edges(const pctx *c, pos_t p, nd_t ea[8], nd_t eb[8]) {
    float ipx, ipy;
    const float dx = modff(p.x, &ipx), dy = modff(p.y, &ipy);
    const int X = (int)ipx, Y = (int)ipy;
    nd_t r[8];
    int n;
    if (0.0f < dx && dx < 1.0f) {
        n = 6;
        r[0] = (nd_t){X, Y}; r[1] = (nd_t){X, Y - 1}; r[2] = (nd_t){X + 1, Y - 1};
        r[3] = (nd_t){X + 1, Y}; r[4] = (nd_t){X + 1, Y + 1}; r[5] = (nd_t){X, Y + 1};
    } else if (0.0f < dy && dy < 1.0f) {
        n = 6;
        r[0] = (nd_t){X, Y}; r[1] = (nd_t){X + 1, Y}; r[2] = (nd_t){X + 1, Y + 1};
        r[3] = (nd_t){X, Y + 1}; r[4] = (nd_t){X - 1, Y + 1}; r[5] = (nd_t){X - 1, Y};
    } else {
        n = 8;
        r[0] = (nd_t){X + 1, Y}; r[1] = (nd_t){X + 1, Y + 1}; r[2] = (nd_t){X, Y + 1}; r[3] = (nd_t){X - 1, Y + 1};
        r[4] = (nd_t){X - 1, Y}; r[5] = (nd_t){X - 1, Y - 1}; r[6] = (nd_t){X, Y - 1}; r[7] = (nd_t){X + 1, Y - 1};
    }
    int m = 0;
    for (int i = 0; i < n; ++i) {
        if (p_valid_node(c, r[i].x, r[i].y)) {
            const nd_t nxt = r[(i + 1) % n];
            if (p_valid_node(c, nxt.x, nxt.y)) { ea[m] = r[i]; eb[m] = nxt; ++m; }
            else ++i;
        }
    }
    return m;
}

/* LinearInterpolationPathExtractor_impl.h:221-235 ; Node.cpp:44-57 neighbor_cell(top, right) */
static void p_fill_costs(const pctx *c, tparams *t) {
    int btop, bright, ctop, cright;
    if (t->p0.x == (float)t->p1.x) {
        btop = t->p2.x > t->p1.x; bright = t->p0.y > (float)t->p1.y;
        ctop = t->p2.x < t->p1.x; cright = bright;
    } else {
        btop = t->p0.x < (float)t->p1.x; bright = t->p2.y < t->p1.y;
        ctop = btop; cright = t->p2.y > t->p1.y;
    }
    t->b = p_cost(c, t->p1.x - (btop ? 1 : 0), t->p1.y - (bright ? 0 : 1));
    t->c = p_cost(c, t->p1.x - (ctop ? 1 : 0), t->p1.y - (cright ? 0 : 1));
}

/* ---- traversal case tables (InterpolatedTraversal.cpp:6-478) ---------------------- */

/* cond() of each case */
static int tt_cond(int kind, int type, const tparams *t) {
    switch (type) {
    case T_I:
        if (kind == K_CORNER) return t->c > (t->b * SQRT2);                         /* :23-25 */
        if (kind == K_CONTIG) return t->c > (t->b * hyp(1, 1 / (1 - t->q)));        /* :60-62 */
        return t->c > (t->b * hyp(1, 1 + t->p));                                    /* :81-83 */
    case T_II:
        if (kind == K_CORNER) return t->c > (t->f * SQRT2);                         /* :129-131 */
        if (kind == K_CONTIG) return (t->f > 0) && (t->c > t->f * hyp(1, 1 - t->q)); /* :175-177 */
        return (t->f > 0) && (t->c > (t->f * hyp(1, 1 / (1 - t->p))));              /* :212-214 */
    case T_III:
        if (kind == K_OPP) return t->c > t->b * hyp(1, t->p);                       /* :304-306 */
        return t->c > t->b;                                                         /* :248-250, 274-276 */
    default:
        return 1;                                                                   /* A, B */
    }
}

/* cost() of each case: cost to goal through this traversal */
static float tt_cost(int kind, int type, const tparams *t) {
    switch (type) {
    case T_I:
        if (kind == K_CORNER) return t->g2 + t->b + cath(t->c, t->b);                       /* :8-10 */
        if (kind == K_CONTIG) return t->g2 + (1 - t->q) * t->b + cath(t->c, t->b);          /* :45-47 */
        return t->g2 + t->b + (1 + t->p) * cath(t->c, t->b);                                /* :85-87 */
    case T_II:
        if (kind == K_CORNER) return t->g1 + cath(t->c, t->f);                              /* :125-127 */
        if (kind == K_CONTIG) return t->g1 + (1 - t->q) * cath(t->c, t->f);                 /* :160-162 */
        return t->g2 + cath(t->c, t->f) + (1 - t->p) * t->f;                                /* :197-199 */
    case T_III:
        if (kind == K_CORNER) return t->g1 + t->b;                                          /* :236-238 */
        if (kind == K_CONTIG) return t->g1 + (1 - t->q) * t->b;                             /* :262-264 */
        return t->g1 + t->b + t->p * cath(t->c, t->b);                                      /* :288-290 */
    case T_A:
        if (kind == K_CORNER) return t->g2 + t->c * SQRT2;                                  /* :324-326 */
        if (kind == K_CONTIG) return t->g2 + t->c * hyp(1, 1 - t->q);                       /* :351-353 */
        return t->g2 + t->c * hyp(1 - t->p, 1);                                             /* :376-378 */
    default:
        if (kind == K_CORNER) return t->g1 + t->c;                                          /* :403-405 */
        if (kind == K_CONTIG) return t->g1 + t->c * (1 - t->q);                             /* :429-431 */
        return t->g1 + t->c * hyp(t->p, 1);                                                 /* :454-456 */
    }
}
static float tt_condcost(int kind, int type, const tparams *t) {
    return tt_cond(kind, type, t) ? tt_cost(kind, type, t) : INFINITY;
}

/* additions() + stepcosts() of each case */
static void tt_emit(int kind, int type, const tparams *t, padd *o) {
    const int vert = (t->p0.x == (float)t->p1.x); /* "p lies on a vertical edge" */
    o->ns = o->nc = 0;
    switch (type) {
    case T_I: {
        if (kind == K_CORNER) {                                                      /* :12-41 */
            float x = 1 - t->b / cath(t->c, t->b);
            o->costs[0] = x * t->b; o->costs[1] = hyp(1 - x, 1) * t->c; o->nc = 2;
            o->steps[0] = vert ? mkpos(t->p0.x, interp1(t->p0.y, (float)t->p1.y, x))
                               : mkpos(interp1(t->p0.x, (float)t->p1.x, x), t->p0.y);
            o->steps[1] = npos(t->p2); o->ns = 2;
        } else if (kind == K_CONTIG) {                                               /* :49-77 */
            float x = 1 - t->q - t->b / cath(t->c, t->b);
            o->costs[0] = x * t->b; o->costs[1] = hyp(1 - t->q - x, 1) * t->c; o->nc = 2;
            o->steps[0] = vert ? mkpos(t->p0.x, interp_abs_f(t->p0.y, (float)t->p1.y, x))
                               : mkpos(interp_abs_f(t->p0.x, (float)t->p1.x, x), t->p0.y);
            o->steps[1] = npos(t->p2); o->ns = 2;
        } else {                                                                     /* :89-119 */
            float x = 1 - (1 + t->p) * t->b / cath(t->c, t->b);
            float v = (1 - x) * t->p / (t->p + 1);
            o->costs[0] = x * t->b; o->costs[1] = hyp(1 - x, 1 + t->p) * t->c; o->nc = 2;
            if (vert) {
                o->steps[0] = mkpos(t->p0.x, interp1(t->p0.y, (float)t->p1.y, v));
                o->steps[1] = mkpos(t->p0.x, interp1(t->p0.y, (float)t->p1.y, v + x));
            } else {
                o->steps[0] = mkpos(interp1(t->p0.x, (float)t->p1.x, v), t->p0.y);
                o->steps[1] = mkpos(interp1(t->p0.x, (float)t->p1.x, v + x), t->p0.y);
            }
            o->steps[2] = npos(t->p2); o->ns = 3;
        }
        break;
    }
    case T_II: {
        float y;
        if (kind == K_CORNER) {                                                      /* :133-157 */
            y = t->f / cath(t->c, t->f);
            o->costs[0] = hyp(1, y) * t->c;
            o->steps[0] = vert ? mkpos(interp1_i(t->p1.x, t->p2.x, y), (float)t->p1.y)
                               : mkpos((float)t->p1.x, interp1_i(t->p1.y, t->p2.y, y));
        } else if (kind == K_CONTIG) {                                               /* :164-193 */
            y = (1 - t->q) * t->f / cath(t->c, t->f);
            o->costs[0] = hyp(1 - t->q, y) * t->c;
            o->steps[0] = vert ? mkpos(interp_abs_i(t->p1.x, t->p2.x, y), (float)t->p1.y)
                               : mkpos((float)t->p1.x, interp_abs_i(t->p1.y, t->p2.y, y));
        } else {                                                                     /* :201-230 */
            y = t->p + t->f / cath(t->c, t->f);
            o->costs[0] = hyp(1, y - t->p) * t->c;
            o->steps[0] = vert ? mkpos(interp1_i(t->p1.x, t->p2.x, y), (float)t->p1.y)
                               : mkpos((float)t->p1.x, interp1_i(t->p1.y, t->p2.y, y));
        }
        o->nc = 1; o->ns = 1;
        break;
    }
    case T_III: {
        if (kind == K_CORNER) {                                                      /* :240-258 */
            o->costs[0] = t->b; o->nc = 1;
            o->steps[0] = npos(t->p1); o->ns = 1;
        } else if (kind == K_CONTIG) {                                               /* :266-284 */
            o->costs[0] = (1 - t->q) * t->b; o->nc = 1;
            o->steps[0] = npos(t->p1); o->ns = 1;
        } else {                                                                     /* :292-320 */
            float x = t->p * t->b / cath(t->c, t->b);
            o->costs[0] = hyp(x, t->p) * t->c; o->costs[1] = (1 - x) * t->b; o->nc = 2;
            o->steps[0] = vert ? mkpos(t->p0.x, interp1(t->p0.y, (float)t->p1.y, x))
                               : mkpos(interp1(t->p0.x, (float)t->p1.x, x), t->p0.y);
            o->steps[1] = npos(t->p1); o->ns = 2;
        }
        break;
    }
    case T_A:                                                                        /* :328-346, 355-373, 380-398 */
        o->costs[0] = (kind == K_CORNER) ? t->c * SQRT2
                    : (kind == K_CONTIG) ? t->c * hyp(1, 1 - t->q) : t->c * hyp(1 - t->p, 1);
        o->nc = 1; o->steps[0] = npos(t->p2); o->ns = 1;
        break;
    default:                                                                         /* :407-425, 433-451, 458-476 */
        o->costs[0] = (kind == K_CORNER) ? t->c
                    : (kind == K_CONTIG) ? (1 - t->q) * t->c : t->c * hyp(t->p, 1);
        o->nc = 1; o->steps[0] = npos(t->p1); o->ns = 1;
        break;
    }
}

/* std::accumulate(step_costs, .0f) */
static float sum_costs(const padd *o) {
    float s = .0f;
    for (int i = 0; i < o->nc; ++i) s += o->costs[i];
    return s;
}
static padd empty_additions(void) {
    padd o = {0}; o.cost_to_goal = INFINITY; return o;
}

/* InterpolatedTraversal.cpp:482-534 (all cases) and :658-695 (direct cases only) */
static padd trav_corner(const pctx *c, tparams *t, float *step_cost) {
    if (t->g1 == INFINITY && t->g2 == INFINITY) return empty_additions();
    if (t->c == INFINITY) return empty_additions();
    t->f = t->g1 - t->g2;
    int type;
    if (c->indirect && t->c > t->b) {
        if ((t->f <= 0) || (sq(t->f) <= cath(t->c, t->b))) type = T_III;
        else if ((t->f <= t->b) && (t->c > (t->f * SQRT2))) type = T_II;
        else if ((t->f > t->b) && (t->c > (t->b * SQRT2))) type = T_I;
        else type = T_A;
    } else {
        if (t->f <= 0) type = T_B;
        else if ((t->f * SQRT2) < t->c) type = T_II;
        else type = T_A;
    }
    padd o = {0};
    tt_emit(K_CORNER, type, t, &o);
    o.cost_to_goal = tt_cost(K_CORNER, type, t);
    *step_cost = sum_costs(&o);
    return o;
}

/* first minimum of an array (std::min_element) */
static int argmin(const float *v, int n) {
    int k = 0;
    for (int i = 1; i < n; ++i) if (v[i] < v[k]) k = i;
    return k;
}

/* InterpolatedTraversal.cpp:535-579 and :697-733 */
static padd trav_contiguous(const pctx *c, tparams *t, float *step_cost) {
    if (t->g1 == INFINITY && t->g2 == INFINITY) return empty_additions();
    if (t->c == INFINITY) return empty_additions();
    t->f = t->g1 - t->g2;
    float costs[5];
    int types[5], n = 0;
    if (c->indirect) {
        types[n] = T_I;   costs[n++] = tt_condcost(K_CONTIG, T_I, t);
        types[n] = T_II;  costs[n++] = tt_condcost(K_CONTIG, T_II, t);
        types[n] = T_III; costs[n++] = tt_condcost(K_CONTIG, T_III, t);
    } else {
        types[n] = T_II;  costs[n++] = tt_condcost(K_CONTIG, T_II, t);
    }
    types[n] = T_A; costs[n++] = tt_cost(K_CONTIG, T_A, t);
    types[n] = T_B; costs[n++] = tt_cost(K_CONTIG, T_B, t);
    const int k = argmin(costs, n);
    padd o = {0};
    tt_emit(K_CONTIG, types[k], t, &o);
    o.cost_to_goal = costs[k];
    *step_cost = sum_costs(&o);
    return o;
}

/* InterpolatedTraversal.cpp:580-656 and :735-778 ; the emptiness test reads g1 of the first
 * and g2 of the second parameter set, which are the same node (p_a) */
static padd trav_opposite(const pctx *c, tparams *t1, tparams *t2, float *step_cost) {
    if (t1->g1 == INFINITY && t2->g2 == INFINITY) return empty_additions();
    if (t1->c == INFINITY) return empty_additions();
    t1->f = t1->g1 - t1->g2;
    t2->f = -t1->f;
    float costs[8];
    int types[8], n = 0;
    tparams *who[8];
    static const int ind[3] = {T_I, T_II, T_III};
    if (c->indirect) {
        for (int k = 0; k < 3; ++k) {
            types[n] = ind[k]; who[n] = t1; costs[n++] = tt_condcost(K_OPP, ind[k], t1);
            types[n] = ind[k]; who[n] = t2; costs[n++] = tt_condcost(K_OPP, ind[k], t2);
        }
    } else {
        types[n] = T_II; who[n] = t1; costs[n++] = tt_condcost(K_OPP, T_II, t1);
        types[n] = T_II; who[n] = t2; costs[n++] = tt_condcost(K_OPP, T_II, t2);
    }
    types[n] = T_A; who[n] = t1; costs[n++] = tt_cost(K_OPP, T_A, t1);
    types[n] = T_A; who[n] = t2; costs[n++] = tt_cost(K_OPP, T_A, t2);
    const int k = argmin(costs, n);
    padd o = {0};
    tt_emit(K_OPP, types[k], who[k], &o);
    o.cost_to_goal = costs[k];
    *step_cost = sum_costs(&o);
    return o;
}

/* Position::aligned, Position.cpp:32-34 */
static int aligned(pos_t p, nd_t n) { return p.x == (float)n.x || p.y == (float)n.y; }

/* LinearInterpolationPathExtractor_impl.h:60-84 */
static padd from_corner(const pctx *c, pos_t p, nd_t a, nd_t b, float *step_cost) {
    tparams t = {0};
    t.p0 = p;
    const int al = aligned(p, a);
    t.p1 = al ? a : b;
    t.p2 = al ? b : a;
    t.g1 = p_interp_rhs(c, t.p1);
    t.g2 = p_interp_rhs(c, t.p2);
    p_fill_costs(c, &t);
    return trav_corner(c, &t, step_cost);
}
/* :86-110 */
static padd from_contiguous_edge(const pctx *c, pos_t p, nd_t a, nd_t b, float *step_cost) {
    tparams t = {0};
    const int al = aligned(p, a);
    t.p0 = p;
    t.p1 = al ? a : b;
    t.p2 = al ? b : a;
    t.g1 = p_interp_rhs(c, t.p1);
    t.g2 = p_interp_rhs(c, t.p2);
    p_fill_costs(c, &t);
    t.q = 1 - fabsf((float)t.p1.y - p.y) - fabsf((float)t.p1.x - p.x);
    return trav_contiguous(c, &t, step_cost);
}
/* :112-144 */
static padd from_opposite_edge(const pctx *c, pos_t p, nd_t a, nd_t b, float *step_cost) {
    tparams t1 = {0}, t2 = {0};
    t1.p1 = t2.p2 = a;
    t1.p2 = t2.p1 = b;
    t1.p0 = t2.p0 = p;
    if (a.x == b.x) { t1.p0.y = (float)a.y; t2.p0.y = (float)b.y; }
    else            { t1.p0.x = (float)a.x; t2.p0.x = (float)b.x; }
    t1.g1 = t2.g2 = p_interp_rhs(c, a);
    t1.g2 = t2.g1 = p_interp_rhs(c, b);
    p_fill_costs(c, &t1);
    p_fill_costs(c, &t2);
    t1.p = fabsf(p.y - t1.p0.y) + fabsf(p.x - t1.p0.x);
    t2.p = 1 - t1.p;
    return trav_opposite(c, &t1, &t2, step_cost);
}
/* :146-163 */
static padd from_edge(const pctx *c, pos_t p, nd_t a, nd_t b, float *step_cost) {
    const int c1 = (p.x == (float)a.x || p.y == (float)a.y);
    const int c2 = (p.x == (float)b.x || p.y == (float)b.y);
    if (c1 || c2) return from_contiguous_edge(c, p, a, b, step_cost);
    return from_opposite_edge(c, p, a, b, step_cost);
}

/* :165-213 getPathAdditions. A position without any usable edge returns the value-initialised
 * additions (no steps, cost_to_goal 0) and leaves step_cost untouched, as the reference does. */
static padd path_additions(const pctx *c, pos_t p, int do_lookahead, float *step_cost) {
    float min_cost = INFINITY;
    padd best = {0};
    nd_t ea[8], eb[8];
    const int ne = p_edges(c, p, ea, eb);
    const int vertex = p_valid_vertex(c, p);
    for (int e = 0; e < ne; ++e) {
        float cur = INFINITY;
        padd t = vertex ? from_corner(c, p, ea[e], eb[e], &cur) : from_edge(c, p, ea[e], eb[e], &cur);
        if (t.ns == 0) continue;
        float la = -1.0f;
        if (do_lookahead && !p_valid_vertex(c, t.steps[t.ns - 1])) {
            float dummy = 0.0f;
            la = path_additions(c, t.steps[t.ns - 1], 0, &dummy).cost_to_goal;
        }
#ifdef ORC_PATH_TRACE
        fprintf(stderr, "%s(%.9g,%.9g) edge %d (%d,%d)-(%d,%d): ns %d end (%.9g,%.9g) cost_to_goal %.9g step %.9g lookahead %.9g%s\n", do_lookahead ? "" : "      la ",
                p.x, p.y, e, ea[e].x, ea[e].y, eb[e].x, eb[e].y, t.ns, t.steps[t.ns - 1].x, t.steps[t.ns - 1].y, t.cost_to_goal, cur, la,
                (la > t.cost_to_goal) ? " REJECTED" : (t.cost_to_goal < min_cost ? " best" : (t.cost_to_goal == min_cost ? " TIE" : "")));
#endif
        if (la > t.cost_to_goal) continue;
        if (t.cost_to_goal < min_cost) {
            min_cost = t.cost_to_goal;
            best = t;
            *step_cost = cur;
        }
    }
    return best;
}

/* :11-58 extract_path.  Returns the number of path points (0 when "no valid path exists");
 * *n_costs step costs; points beyond cap_pts / costs beyond cap_costs are counted, not stored. */
int orc_extract_path_field(const float *rhs, int nx, int ny, int cells,
                           const uint8_t *map, int width, int length, int thr_uchar,
                           float start_x, float start_y, float goal_x, float goal_y,
                           int lookahead, int max_steps, int allow_indirect,
                           float *path_xy, int cap_pts, float *costs, int cap_costs,
                           int *n_costs, float *total_cost, float *total_dist) {
    pctx c = {rhs, nx, ny, cells, map, width, length, thr_uchar, {start_x, start_y}, {goal_x, goal_y}, allow_indirect};
    int npts = 0, ncost = 0, curr_step = 0;
    float tcost = 0, tdist = 0, min_cost, step_cost = 0.0f;
    pos_t last = c.start;
    if (npts < cap_pts) { path_xy[2 * npts] = last.x; path_xy[2 * npts + 1] = last.y; }
    ++npts;
    do {
        padd pa = path_additions(&c, last, lookahead, &step_cost);
        float step_dist = 0;
        pos_t prev = last;
        for (int i = 0; i < pa.ns; ++i) {
            if (npts < cap_pts) { path_xy[2 * npts] = pa.steps[i].x; path_xy[2 * npts + 1] = pa.steps[i].y; }
            ++npts;
            step_dist += hypotf(prev.x - pa.steps[i].x, prev.y - pa.steps[i].y); /* Position.cpp:29-31 */
            prev = pa.steps[i];
        }
        for (int i = 0; i < pa.nc; ++i) {
            if (ncost < cap_costs) costs[ncost] = pa.costs[i];
            ++ncost;
        }
        min_cost = pa.cost_to_goal;
        tcost += step_cost;
        tdist += step_dist;
        curr_step += 1;
        last = prev;
    } while (!(c.goal.x == last.x && c.goal.y == last.y) && (min_cost != INFINITY) && (curr_step < max_steps));
    if (min_cost == INFINITY) npts = 0;
    *n_costs = ncost;
    *total_cost = tcost;
    *total_dist = tdist;
    return npts;
}
