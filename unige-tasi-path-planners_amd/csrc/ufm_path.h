// ufm_path.h -- path extraction on the device (included by ufm_engine.hip; gfx950 only).
//
// Counterpart of the reference's LinearInterpolationPathExtractor (PathExtraction/
// LinearInterpolationPathExtractor_impl.h:11-235) and of the traversal case tables it calls
// (ProjectToolkit/InterpolatedTraversal.cpp:6-778).  The reference walks the field on the CPU:
// per step it tries the <= 8 edges around the current position and, for each candidate that ends
// on an edge, runs the same search once more from there (the "lookahead" of Perkins et al.).
// That is an 8 x 8 fan-out of independent evaluations, each a handful of dependent field / raster
// reads -- exactly one 64-lane wavefront: lane = 8 * e + l evaluates primary edge e and, when
// needed, lookahead edge l of e's end point; two 8-lane reductions and one 8-group reduction pick
// the step.  One wavefront per map, so a batch of maps is extracted in a single launch and the
// field never leaves HBM.  All arithmetic is the reference's, in fp32 with correctly rounded
// sqrt / divide and hypotf evaluated in fp64 as glibc does, so decisions (argmin, lookahead
// rejections) fall the same way as on the CPU.
#pragma once

struct PathField {
    const float *G;        // this map's field, tile-major: element (x,y) in tile (x/T, y/T) of TY per row, T*T floats each
    const uint8_t *cost;   // this map's raster [L][W]
    int EX, EY, L, W, TY, thr;
    int cells;             // cell-centred field (DFM): node values are 4-cell averages
    int indirect;          // allow_indirect_traversals
};

// one candidate move: <= 3 way points, <= 2 step costs (InterpolatedTraversal.h:32-39)
struct Move {
    float x[3], y[3];
    float sc[2];
    int ns, nc;
    float ctg;             // cost to goal through this move
};

constexpr float PATH_SQRT2 = 1.41421356237309504880168872420969807856967187537694f;   // Macros.cpp:2

// std::hypotf of glibc >= 2.35: the fp64 evaluation, narrowed once
__device__ __forceinline__ float hyp_rn(float a, float b) {
    return (float)__builtin_sqrt((double)a * (double)a + (double)b * (double)b);
}
// CATH (Macros.h:12): products rounded separately
__device__ __forceinline__ float cath_rn(float a, float b) { return sqrt_rn(a * a - b * b); }

__device__ __forceinline__ float field_at(const PathField &F, int x, int y) {      // ExpandedMap::get_rhs
    if (x < 0 || y < 0 || x >= F.EX || y >= F.EY) return INFINITY;
    return F.G[((size_t)(x / T) * F.TY + (y / T)) * (T * T) + (size_t)(x % T) * T + (y % T)];
}
// ExpandedMap::get_interp_rhs (ExpandedMap_impl.h:87-97); summation order of the cell variant:
// (x,y-1) + (x-1,y-1) + (x,y) + (x-1,y)
__device__ __forceinline__ float node_value(const PathField &F, int x, int y) {
    if (!F.cells) return field_at(F, x, y);
    return (field_at(F, x, y - 1) + field_at(F, x - 1, y - 1) + field_at(F, x, y) + field_at(F, x - 1, y)) * 0.25f;
}
__device__ __forceinline__ float raster_cost(const PathField &F, int cx, int cy) {  // Graph::get_cost
    if (cx < 0 || cy < 0 || cx >= F.L || cy >= F.W) return INFINITY;
    const int v = F.cost[(size_t)cx * F.W + cy];
    return v >= F.thr ? INFINITY : (float)v;
}
__device__ __forceinline__ bool node_ok(const PathField &F, int x, int y) { return x >= 0 && y >= 0 && x <= F.L && y <= F.W; }
__device__ __forceinline__ bool is_vertex(const PathField &F, float x, float y) {   // Graph::is_valid_vertex
    return ceilf(x) == x && ceilf(y) == y && x >= 0.0f && y >= 0.0f && x <= (float)F.L && y <= (float)F.W;
}

// k-th pair of Graph::consecutive_neighbors(Position) (Graph.cpp:151-200); false when there are
// fewer pairs.  The ring around a point on an edge has 6 nodes, around a vertex 8.
__device__ bool ring_pair(const PathField &F, float px, float py, int k, int &ax, int &ay, int &bx, int &by) {
    const float ix = truncf(px), iy = truncf(py);
    const bool fx = px != ix, fy = py != iy;
    const int X = (int)ix, Y = (int)iy;
    // ring offsets, packed 2 bits per coordinate (+1 biased), slot i at bits 4i..4i+3 (dx | dy<<2)
    //   x fractional: (0,0)(0,-1)(1,-1)(1,0)(1,1)(0,1)
    //   y fractional: (0,0)(1,0)(1,1)(0,1)(-1,1)(-1,0)
    //   vertex      : (1,0)(1,1)(0,1)(-1,1)(-1,0)(-1,-1)(0,-1)(1,-1)
    const int n = (fx || fy) ? 6 : 8;
    int m = 0;
    for (int i = 0; i < n; ++i) {
        int dx0, dy0, dx1, dy1;
        auto slot = [&](int s, int &dx, int &dy) {
            if (fx) { const int tx[6] = {0, 0, 1, 1, 1, 0}, ty[6] = {0, -1, -1, 0, 1, 1}; dx = tx[s]; dy = ty[s]; }
            else if (fy) { const int tx[6] = {0, 1, 1, 0, -1, -1}, ty[6] = {0, 0, 1, 1, 1, 0}; dx = tx[s]; dy = ty[s]; }
            else { const int tx[8] = {1, 1, 0, -1, -1, -1, 0, 1}, ty[8] = {0, 1, 1, 1, 0, -1, -1, -1}; dx = tx[s]; dy = ty[s]; }
        };
        slot(i, dx0, dy0);
        if (!node_ok(F, X + dx0, Y + dy0)) continue;
        slot((i + 1) % n, dx1, dy1);
        if (!node_ok(F, X + dx1, Y + dy1)) { ++i; continue; }     // the next slot starts at an invalid node: skipped too
        if (m == k) { ax = X + dx0; ay = Y + dy0; bx = X + dx1; by = Y + dy1; return true; }
        ++m;
    }
    return false;
}

// geometry of one parameter set (TraversalParams, InterpolatedTraversal.h:11-26)
struct TSet {
    float p0x, p0y;        // point aligned with p1
    int p1x, p1y, p2x, p2y;
    float g1, g2, b, c, f;
    bool samerow;          // p0 and p1 share x: the reference's "p lies on a vertical edge" branch
};
// costs of the traversed cell (c: holds p0, p1, p2) and of its mirror image across p0-p1 (b)
// (LinearInterpolationPathExtractor_impl.h:221-235)
__device__ void set_costs(const PathField &F, TSet &t) {
    t.samerow = (t.p0x == (float)t.p1x);
    int cx, cy, bx, by;
    if (t.samerow) {
        cy = by = (t.p0y > (float)t.p1y) ? t.p1y : t.p1y - 1;
        cx = (t.p2x < t.p1x) ? t.p1x - 1 : t.p1x;
        bx = (t.p2x > t.p1x) ? t.p1x - 1 : t.p1x;
    } else {
        cx = bx = (t.p0x < (float)t.p1x) ? t.p1x - 1 : t.p1x;
        cy = (t.p2y > t.p1y) ? t.p1y : t.p1y - 1;
        by = (t.p2y < t.p1y) ? t.p1y : t.p1y - 1;
    }
    t.c = raster_cost(F, cx, cy);
    t.b = raster_cost(F, bx, by);
}
// way points: a point at parameter s on the p0 -> p1 side, on the p1 -> p2 side
__device__ __forceinline__ void on_p0p1(const TSet &t, float s, bool unit, float &ox, float &oy) {
    // INTERP_1 / INTERP_ABS (Macros.h:18,24) between a float and an int coordinate
    if (t.samerow) { const float d = (float)t.p1y - t.p0y; ox = t.p0x; oy = t.p0y + (unit ? d : d / fabsf(d)) * s; }
    else           { const float d = (float)t.p1x - t.p0x; ox = t.p0x + (unit ? d : d / fabsf(d)) * s; oy = t.p0y; }
}
__device__ __forceinline__ void on_p1p2(const TSet &t, float s, float &ox, float &oy) {
    // both operands are node coordinates one apart: INTERP_1 and INTERP_ABS coincide ((to-from)/|to-from| == to-from)
    if (t.samerow) { ox = (float)t.p1x + (float)(t.p2x - t.p1x) * s; oy = (float)t.p1y; }
    else           { ox = (float)t.p1x; oy = (float)t.p1y + (float)(t.p2y - t.p1y) * s; }
}
__device__ __forceinline__ void put(Move &m, int i, float x, float y) { m.x[i] = x; m.y[i] = y; }

enum { TY_I = 0, TY_II = 1, TY_III = 2, TY_A = 3, TY_B = 4 };

// ---- corner: p is a grid vertex (InterpolatedTraversal.cpp:482-534 / 658-695) --------------
__device__ void move_corner(const PathField &F, TSet &t, Move &m) {
    m.ns = m.nc = 0; m.ctg = INFINITY;
    if ((t.g1 == INFINITY && t.g2 == INFINITY) || t.c == INFINITY) return;
    const float f = t.f = t.g1 - t.g2, b = t.b, c = t.c;
    int ty;
    if (F.indirect && c > b) {
        if (f <= 0 || f * f <= cath_rn(c, b)) ty = TY_III;
        else if (f <= b && c > f * PATH_SQRT2) ty = TY_II;
        else if (f > b && c > b * PATH_SQRT2) ty = TY_I;
        else ty = TY_A;
    } else {
        ty = (f <= 0) ? TY_B : ((f * PATH_SQRT2 < c) ? TY_II : TY_A);
    }
    if (ty == TY_III) {                                   // :236-258
        m.ctg = t.g1 + b; m.sc[0] = b; m.nc = 1;
        put(m, 0, (float)t.p1x, (float)t.p1y); m.ns = 1;
    } else if (ty == TY_II) {                             // :125-157
        const float cf = cath_rn(c, f), y = f / cf;
        m.ctg = t.g1 + cf; m.sc[0] = hyp_rn(1, y) * c; m.nc = 1;
        on_p1p2(t, y, m.x[0], m.y[0]); m.ns = 1;
    } else if (ty == TY_I) {                              // :8-41
        const float cb = cath_rn(c, b), x = 1 - b / cb;
        m.ctg = t.g2 + b + cb; m.sc[0] = x * b; m.sc[1] = hyp_rn(1 - x, 1) * c; m.nc = 2;
        on_p0p1(t, x, true, m.x[0], m.y[0]); put(m, 1, (float)t.p2x, (float)t.p2y); m.ns = 2;
    } else if (ty == TY_A) {                              // :324-346
        m.ctg = t.g2 + c * PATH_SQRT2; m.sc[0] = c * PATH_SQRT2; m.nc = 1;
        put(m, 0, (float)t.p2x, (float)t.p2y); m.ns = 1;
    } else {                                              // :403-425
        m.ctg = t.g1 + c; m.sc[0] = c; m.nc = 1;
        put(m, 0, (float)t.p1x, (float)t.p1y); m.ns = 1;
    }
}

// ---- contiguous edge: p lies on the edge that ends in p1 (:535-579 / 697-733) --------------
__device__ void move_contiguous(const PathField &F, TSet &t, float q, Move &m) {
    m.ns = m.nc = 0; m.ctg = INFINITY;
    if ((t.g1 == INFINITY && t.g2 == INFINITY) || t.c == INFINITY) return;
    const float f = t.f = t.g1 - t.g2, b = t.b, c = t.c, r = 1 - q;
    // candidate costs in the reference's order; first minimum wins (std::min_element)
    float best = INFINITY; int ty = -1;
    auto offer = [&](int type, bool ok, float cost) {
        const float v = ok ? cost : INFINITY;
        if (ty < 0 || v < best) { best = v; ty = type; }
    };
    if (F.indirect) offer(TY_I, c > b * hyp_rn(1, 1 / r), t.g2 + r * b + cath_rn(c, b));             // :45-62
    offer(TY_II, (f > 0) && (c > f * hyp_rn(1, r)), t.g1 + r * cath_rn(c, f));                        // :160-177
    if (F.indirect) offer(TY_III, c > b, t.g1 + r * b);                                               // :262-276
    offer(TY_A, true, t.g2 + c * hyp_rn(1, r));                                                       // :351-353
    offer(TY_B, true, t.g1 + c * r);                                                                  // :429-431
    m.ctg = best;
    if (ty == TY_I) {                                     // :49-77
        const float x = r - b / cath_rn(c, b);
        m.sc[0] = x * b; m.sc[1] = hyp_rn(r - x, 1) * c; m.nc = 2;
        on_p0p1(t, x, false, m.x[0], m.y[0]); put(m, 1, (float)t.p2x, (float)t.p2y); m.ns = 2;
    } else if (ty == TY_II) {                             // :164-193
        const float y = r * f / cath_rn(c, f);
        m.sc[0] = hyp_rn(r, y) * c; m.nc = 1;
        on_p1p2(t, y, m.x[0], m.y[0]); m.ns = 1;
    } else if (ty == TY_III) {                            // :266-284
        m.sc[0] = r * b; m.nc = 1; put(m, 0, (float)t.p1x, (float)t.p1y); m.ns = 1;
    } else if (ty == TY_A) {                              // :355-373
        m.sc[0] = c * hyp_rn(1, r); m.nc = 1; put(m, 0, (float)t.p2x, (float)t.p2y); m.ns = 1;
    } else {                                              // :433-451
        m.sc[0] = r * c; m.nc = 1; put(m, 0, (float)t.p1x, (float)t.p1y); m.ns = 1;
    }
}

// ---- opposite edge: p lies on the cell side facing the edge a-b (:580-656 / 735-778) -------
// two parameter sets, one per end of the edge; set 2 mirrors set 1 (f2 = -f1, p2 = 1 - p1)
__device__ float opp_cost(int type, const TSet &t, float p, bool &ok) {
    const float b = t.b, c = t.c, f = t.f;
    switch (type) {
    case TY_I:   ok = c > b * hyp_rn(1, 1 + p);                 return t.g2 + b + (1 + p) * cath_rn(c, b);   // :81-87
    case TY_II:  ok = (f > 0) && (c > f * hyp_rn(1, 1 / (1 - p))); return t.g2 + cath_rn(c, f) + (1 - p) * f; // :197-214
    case TY_III: ok = c > b * hyp_rn(1, p);                     return t.g1 + b + p * cath_rn(c, b);         // :288-306
    default:     ok = true;                                     return t.g2 + c * hyp_rn(1 - p, 1);          // :376-378
    }
}
__device__ void move_opposite(const PathField &F, TSet &t1, TSet &t2, float p1, Move &m) {
    m.ns = m.nc = 0; m.ctg = INFINITY;
    // the reference tests g1 of the first and g2 of the second set: the same node twice
    if ((t1.g1 == INFINITY && t2.g2 == INFINITY) || t1.c == INFINITY) return;
    t1.f = t1.g1 - t1.g2;
    t2.f = -t1.f;
    const float p2 = 1 - p1;
    float best = INFINITY; int ty = -1, which = 0;
    for (int type = F.indirect ? TY_I : TY_II; type <= TY_A; ++type) {
        if (!F.indirect && type == TY_III) continue;
        for (int w = 0; w < 2; ++w) {
            bool ok;
            const float cost = opp_cost(type, w ? t2 : t1, w ? p2 : p1, ok);
            const float v = ok ? cost : INFINITY;
            if (ty < 0 || v < best) { best = v; ty = type; which = w; }
        }
    }
    const TSet &t = which ? t2 : t1;
    const float p = which ? p2 : p1, b = t.b, c = t.c, f = t.f;
    m.ctg = best;
    if (ty == TY_I) {                                     // :89-119
        const float x = 1 - (1 + p) * b / cath_rn(c, b), v = (1 - x) * p / (p + 1);
        m.sc[0] = x * b; m.sc[1] = hyp_rn(1 - x, 1 + p) * c; m.nc = 2;
        on_p0p1(t, v, true, m.x[0], m.y[0]); on_p0p1(t, v + x, true, m.x[1], m.y[1]);
        put(m, 2, (float)t.p2x, (float)t.p2y); m.ns = 3;
    } else if (ty == TY_II) {                             // :201-230
        const float y = p + f / cath_rn(c, f);
        m.sc[0] = hyp_rn(1, y - p) * c; m.nc = 1;
        on_p1p2(t, y, m.x[0], m.y[0]); m.ns = 1;
    } else if (ty == TY_III) {                            // :292-320
        const float x = p * b / cath_rn(c, b);
        m.sc[0] = hyp_rn(x, p) * c; m.sc[1] = (1 - x) * b; m.nc = 2;
        on_p0p1(t, x, true, m.x[0], m.y[0]); put(m, 1, (float)t.p1x, (float)t.p1y); m.ns = 2;
    } else {                                              // :380-398
        m.sc[0] = c * hyp_rn(1 - p, 1); m.nc = 1; put(m, 0, (float)t.p2x, (float)t.p2y); m.ns = 1;
    }
}

// the move from position (px,py) across the edge (a,b) of its ring
// (LinearInterpolationPathExtractor_impl.h:60-163)
__device__ void move_across(const PathField &F, float px, float py, bool vertex, int ax, int ay, int bx, int by, Move &m) {
    const bool al_a = (px == (float)ax) || (py == (float)ay);     // Position::aligned
    const bool al_b = (px == (float)bx) || (py == (float)by);
    if (vertex || al_a || al_b) {
        TSet t;
        t.p0x = px; t.p0y = py;
        t.p1x = al_a ? ax : bx; t.p1y = al_a ? ay : by;
        t.p2x = al_a ? bx : ax; t.p2y = al_a ? by : ay;
        t.g1 = node_value(F, t.p1x, t.p1y);
        t.g2 = node_value(F, t.p2x, t.p2y);
        set_costs(F, t);
        if (vertex) move_corner(F, t, m);
        else move_contiguous(F, t, 1 - fabsf((float)t.p1y - py) - fabsf((float)t.p1x - px), m);
    } else {
        TSet t1, t2;
        t1.p1x = t2.p2x = ax; t1.p1y = t2.p2y = ay;
        t1.p2x = t2.p1x = bx; t1.p2y = t2.p1y = by;
        t1.p0x = t2.p0x = px; t1.p0y = t2.p0y = py;
        if (ax == bx) { t1.p0y = (float)ay; t2.p0y = (float)by; }    // slide p onto the line through each end
        else          { t1.p0x = (float)ax; t2.p0x = (float)bx; }
        t1.g1 = t2.g2 = node_value(F, ax, ay);
        t1.g2 = t2.g1 = node_value(F, bx, by);
        set_costs(F, t1);
        set_costs(F, t2);
        move_opposite(F, t1, t2, fabsf(py - t1.p0y) + fabsf(px - t1.p0x), m);
    }
}

// Output record per map (floats): [0] points (int bits) [1] step costs (int bits) [2] total_cost
// [3] total_dist [4] steps taken (int bits) [5..7] reserved, then cap_pts (x,y) pairs, then
// cap_costs step costs.  Counts beyond the capacities are counted, not stored.
constexpr int PATH_HDR = 8;

struct PathJob {
    float sx, sy, gx, gy;     // Graph::start_pos_, goal_pos_
};

__global__ __launch_bounds__(64) void k_extract_path(PathField F0, size_t gstride, size_t cstride, const PathJob *jobs,
                                                     float *out, size_t ostride, int cap_pts, int cap_costs,
                                                     int lookahead, int max_steps) {
    const int m = blockIdx.x, lane = threadIdx.x, e = lane >> 3, l = lane & 7;
    PathField F = F0;
    F.G += (size_t)m * gstride;
    F.cost += (size_t)m * cstride;
    const PathJob job = jobs[m];
    float *o = out + (size_t)m * ostride;
    float *opts = o + PATH_HDR, *ocst = opts + 2 * (size_t)cap_pts;

    float lx = job.sx, ly = job.sy;               // `last`
    int npts = 1, ncst = 0, step = 0;
    float total_cost = 0, total_dist = 0, step_cost = 0, min_cost;
    if (lane == 0 && cap_pts > 0) { opts[0] = lx; opts[1] = ly; }
    do {
        // primary move across edge e of the ring around `last`
        Move mv; mv.ns = mv.nc = 0; mv.ctg = INFINITY;
        int ax, ay, bx, by;
        const bool vertex = is_vertex(F, lx, ly);
        if (ring_pair(F, lx, ly, e, ax, ay, bx, by)) move_across(F, lx, ly, vertex, ax, ay, bx, by, mv);
        bool usable = mv.ns > 0;
        // lookahead from the end point, when that is not a vertex: cost of its best move
        float ex = lx, ey = ly;
        for (int i = 0; i < 3; ++i) if (i == mv.ns - 1) { ex = mv.x[i]; ey = mv.y[i]; }
        const bool need = lookahead && usable && !is_vertex(F, ex, ey);
        float la = INFINITY;
        if (need) {
            Move l2; l2.ns = 0; l2.ctg = INFINITY;
            if (ring_pair(F, ex, ey, l, ax, ay, bx, by)) move_across(F, ex, ey, false, ax, ay, bx, by, l2);
            if (l2.ns > 0) la = l2.ctg;
        }
        la = fminf(la, __shfl_xor(la, 1));
        la = fminf(la, __shfl_xor(la, 2));
        la = fminf(la, __shfl_xor(la, 4));
        if (la == INFINITY) la = 0.0f;             // no move was promoted: value-initialised additions
        if (need && la > mv.ctg) usable = false;    // lookahead test failed
        // first strictly smallest cost among the usable edges, in ring order
        float kc = (usable && mv.ctg < INFINITY) ? mv.ctg : INFINITY;
        int ke = (kc < INFINITY) ? e : 8;
        for (int s = 8; s < 64; s <<= 1) {
            const float oc = __shfl_xor(kc, s);
            const int oe = __shfl_xor(ke, s);
            if (oc < kc || (oc == kc && oe < ke)) { kc = oc; ke = oe; }
        }
        // broadcast the winner (lanes 8*ke..): every lane continues with the same state
        float bxs[3] = {0, 0, 0}, bys[3] = {0, 0, 0}, bsc[2] = {0, 0};
        int bns = 0, bnc = 0;
        min_cost = 0.0f;
        if (ke < 8) {
            const int src = ke * 8;
            bns = __shfl(mv.ns, src); bnc = __shfl(mv.nc, src);
            for (int i = 0; i < 3; ++i) { bxs[i] = __shfl(mv.x[i], src); bys[i] = __shfl(mv.y[i], src); }
            bsc[0] = __shfl(mv.sc[0], src); bsc[1] = __shfl(mv.sc[1], src);
            min_cost = __shfl(mv.ctg, src);
            float s = .0f;                          // std::accumulate(step_costs, .0f)
            for (int i = 0; i < 2; ++i) if (i < bnc) s += bsc[i];
            step_cost = s;
        }
        float step_dist = 0;
        for (int i = 0; i < 3; ++i) if (i < bns) {
            if (lane == 0 && npts < cap_pts) { opts[2 * npts] = bxs[i]; opts[2 * npts + 1] = bys[i]; }
            ++npts;
            step_dist += hyp_rn(lx - bxs[i], ly - bys[i]);      // Position::distance
            lx = bxs[i]; ly = bys[i];
        }
        for (int i = 0; i < 2; ++i) if (i < bnc) {
            if (lane == 0 && ncst < cap_costs) ocst[ncst] = bsc[i];
            ++ncst;
        }
        total_cost += step_cost;
        total_dist += step_dist;
        ++step;
    } while (!(job.gx == lx && job.gy == ly) && min_cost != INFINITY && step < max_steps);
    if (lane == 0) {
        o[0] = __int_as_float(min_cost == INFINITY ? 0 : npts);
        o[1] = __int_as_float(ncst);
        o[2] = total_cost;
        o[3] = total_dist;
        o[4] = __int_as_float(step);
    }
}

// ---- back-pointers (the reference's INFO member of a level-1/2 map element) ----------------------
// The reference stores, per expanded element, which neighbours its RHS came from
// (FieldDPlanner_impl.h:86-111,196-208; ShiftedGridPlanner_impl.h:131-166,266-303;
// DynamicFastMarching_impl.h:73-99,212-268) and recomputes, when a neighbour is raised, only the elements
// that point at it.  The engine stores one code per element (DevParams::bp, written with every value:
// k_relax, ufm_region.h) and invalidates along it; k_info_stored turns the codes into the reference's
// format.  k_info derives the same from the field alone, exactly as min_rhs<level>() would: the checker
// of the stored ones (they may differ where two candidates tie).
// Node planners: out[0] = linear index (x * EY + y) of the node b with RHS(s) = cost over the edge
// (b, ccw_neighbor(s, b)), out[1] = -1.  DFM: the two cells of the winning stencil (-1: none,
// -2: outside the grid).
__device__ __forceinline__ int ring_of(int dx, int dy) {          // Graph.cpp:232-260 rotation order
    const int idx[3][3] = {{7, 0, 1}, {6, -1, 2}, {5, 4, 3}};
    return idx[dx + 1][dy + 1];
}
__device__ __forceinline__ void ring_at(int r, int &dx, int &dy) {
    const int RX[8] = {-1, -1, 0, 1, 1, 1, 0, -1}, RY[8] = {0, 1, 1, 1, 0, -1, -1, -1};
    dx = RX[r & 7]; dy = RY[r & 7];
}
__device__ __forceinline__ bool elem_ok(const PathField &F, int x, int y) { return x >= 0 && y >= 0 && x < F.EX && y < F.EY; }

// compute_optimal_cost(s, a, b) with the given neighbour values: the corner traversal cost
__device__ float node_cost(const PathField &F, int sx, int sy, int ax, int ay, int bx, int by, float ga, float gb) {
    TSet t;
    const bool al = (sx == ax) || (sy == ay);
    t.p0x = (float)sx; t.p0y = (float)sy;
    t.p1x = al ? ax : bx; t.p1y = al ? ay : by;
    t.p2x = al ? bx : ax; t.p2y = al ? by : ay;
    t.g1 = al ? ga : gb; t.g2 = al ? gb : ga;
    if (t.g1 == INFINITY && t.g2 == INFINITY) return INFINITY;
    set_costs(F, t);
    Move m;
    move_corner(F, t, m);
    return m.ctg;
}

// DynamicFastMarching_impl.h:322-351
__device__ __forceinline__ void dfm_best(const PathField &F, int ax, int ay, int bx, int by, int &ox, int &oy, float &og) {
    const float ca = field_at(F, ax, ay), cb = field_at(F, bx, by);
    if (ca < cb) { ox = ax; oy = ay; og = ca; } else { ox = bx; oy = by; og = cb; }
}
__device__ float dfm_stencil(int ca, int cb, float ga, float gb, float tau, float h, int &b0, int &b1) {
    if (ga > gb) { const float tg = ga; ga = gb; gb = tg; const int tc = ca; ca = cb; cb = tc; }
    if (ga == INFINITY && gb == INFINITY) { b0 = -1; b1 = -1; return INFINITY; }
    if (tau * h > gb - ga) {
        b0 = ca; b1 = cb;
        const float th = tau * h, d = gb - ga;
        return (ga + gb + sqrt_rn(2 * (th * th) - d * d)) * 0.5f;
    }
    b0 = ca; b1 = -1;
    return ga + tau * h;
}

// The stored bytes ((code << 2) | dep, ufm_engine.hip; BP_NONE: the goal, or an element that never got a value -> -1, -1; likewise an
// element whose value is +inf).  Node planners, code (q << 1) | h: the triangle of cell q -- (x-1+dx, y-1+dy), dx = q >> 1, dy = q & 1 -- over the vertical (h = 0) or the
// horizontal (h = 1) neighbour p1 and the diagonal node p2 of that cell; b is the one of the two whose ccw_neighbor is the other.
// MS-DFM level 1, code (q << 1) | w: the candidate of min_rhs_decreased_neighbor (impl:270-313) built on the neighbour w of axis q (vertical,
// horizontal, TR-BL, TL-BR) and the better cell of the perpendicular pair; out = the pair compute_optimal_cost leaves (impl:322-342).
__global__ void k_info_stored(PathField F, const uint8_t *bp, int x0, int y0, int nx, int ny, int32_t *out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nx * ny) return;
    const int x = x0 + e / ny, y = y0 + e % ny;
    int b0 = -1, b1 = -1;
    const int code = bp[((size_t)(x / T) * F.TY + (y / T)) * (T * T) + (size_t)(x % T) * T + (y % T)];
#ifdef UFM_BPDEBUG
    if (code == 0xFD) { out[2 * e] = -3; out[2 * e + 1] = -3; return; }
#endif
    if (code != 0xFF && field_at(F, x, y) < INFINITY) {
        const int q = (code >> 3) & 3, w = (code >> 2) & 1;      // the byte: (((q << 1) | w) << 2) | dep
        if (F.cells) {
            const int NX[4][2] = {{-1, 1}, {0, 0}, {-1, 1}, {-1, 1}}, NY[4][2] = {{0, 0}, {-1, 1}, {1, -1}, {-1, 1}};
            // the perpendicular pair in the reference's argument order (best_cell: a tie goes to the second)
            const int PAX[4] = {0, -1, -1, 1}, PAY[4] = {-1, 0, -1, -1}, PBX[4] = {0, 1, 1, -1}, PBY[4] = {1, 0, 1, 1};
            const int qx = x + NX[q][w], qy = y + NY[q][w];
            int px, py; float gp;
            dfm_best(F, x + PAX[q], y + PAY[q], x + PBX[q], y + PBY[q], px, py, gp);
            auto lin = [&](int ax, int ay) { return elem_ok(F, ax, ay) ? ax * F.EY + ay : -2; };
            dfm_stencil(lin(qx, qy), lin(px, py), field_at(F, qx, qy), gp, raster_cost(F, x, y), q < 2 ? 1.0f : PATH_SQRT2, b0, b1);
        } else {
            const int dx = (q & 2) ? 1 : -1, dy = (q & 1) ? 1 : -1;
            const int p1x = w ? x : x + dx, p1y = w ? y + dy : y, p2x = x + dx, p2y = y + dy;
            int cx, cy;
            ring_at(ring_of(p1x - x, p1y - y) + 1, cx, cy);
            const bool p1_first = (x + cx == p2x) && (y + cy == p2y);
            b0 = p1_first ? p1x * F.EY + p1y : p2x * F.EY + p2y;
        }
    }
    out[2 * e] = b0;
    out[2 * e + 1] = b1;
}
// RHS of an element as the reference's min_rhs<level>() derives it from the field (and, b0 / b1, the back-pointer(s) it would store).
// The goal's RHS is 0 by definition (init(), impl:17-21): the caller's business.
__device__ float derived_rhs(const PathField &F, int lvl, int x, int y, int &b0, int &b1) {
    b0 = -1; b1 = -1;
    float out_rhs = INFINITY;
    auto lin = [&](int qx, int qy) { return elem_ok(F, qx, qy) ? qx * F.EY + qy : -2; };
    if (F.cells) {                                        // min_rhs<1>, DynamicFastMarching_impl.h:212-268
        const float tau = raster_cost(F, x, y);
        if (tau != INFINITY) {
            int ax, ay, bx, by, o0, o1, d0, d1; float ga, gb;
            dfm_best(F, x - 1, y, x + 1, y, ax, ay, ga);
            dfm_best(F, x, y - 1, x, y + 1, bx, by, gb);
            const float so = dfm_stencil(lin(ax, ay), lin(bx, by), ga, gb, tau, 1.0f, o0, o1);
            dfm_best(F, x - 1, y - 1, x + 1, y + 1, ax, ay, ga);
            dfm_best(F, x + 1, y - 1, x - 1, y + 1, bx, by, gb);
            const float sd = dfm_stencil(lin(ax, ay), lin(bx, by), ga, gb, tau, PATH_SQRT2, d0, d1);
            if (sd < so) { b0 = d0; b1 = d1; } else { b0 = o0; b1 = o1; }
            out_rhs = fminf(sd, so);
        }
    } else if (lvl == 2) {                                // ShiftedGridPlanner_impl.h:280-303
        const int DX[4] = {-1, 1, -1, 1}, DY[4] = {-1, -1, 1, 1};        // Graph::neighbors_diag_4
        float rhs = INFINITY;
        for (int i = 0; i < 4; ++i) {
            const int qx = x + DX[i], qy = y + DY[i];
            if (!elem_ok(F, qx, qy)) continue;
            const int r = ring_of(qx - x, qy - y);
            int dx, dy;
            ring_at(r + 1, dx, dy); const int ccx = x + dx, ccy = y + dy;
            ring_at(r + 7, dx, dy); const int cwx = x + dx, cwy = y + dy;
            const bool v1 = elem_ok(F, ccx, ccy), v2 = elem_ok(F, cwx, cwy);
            const float g_cc = v1 ? field_at(F, ccx, ccy) : INFINITY, g_cw = v2 ? field_at(F, cwx, cwy) : INFINITY;
            const float g_q = field_at(F, qx, qy);
            if (v1 && (!v2 || g_cc <= g_cw)) {
                const float c = node_cost(F, x, y, qx, qy, ccx, ccy, g_q, g_cc);
                if (c < rhs) rhs = c;
                if (rhs == c) b0 = qx * F.EY + qy;
            } else if (v2 && (!v1 || g_cc > g_cw)) {
                const float c = node_cost(F, x, y, qx, qy, cwx, cwy, g_q, g_cw);
                if (c < rhs) rhs = c;
                if (rhs == c) b0 = cwx * F.EY + cwy;
            }
        }
        out_rhs = rhs;
    } else {                                              // FieldDPlanner_impl.h:196-208, ShiftedGridPlanner_impl.h:266-278
        const int DX[8] = {-1, -1, 0, 1, 1, 1, 0, -1}, DY[8] = {0, -1, -1, -1, 0, 1, 1, 1};   // Graph::neighbors_8
        float rhs = INFINITY;
        for (int i = 0; i < 8; ++i) {
            const int qx = x + DX[i], qy = y + DY[i];
            if (!elem_ok(F, qx, qy)) continue;
            int dx, dy;
            ring_at(ring_of(qx - x, qy - y) + 1, dx, dy);
            const int ccx = x + dx, ccy = y + dy;
            if (!elem_ok(F, ccx, ccy)) continue;
            const float c = node_cost(F, x, y, qx, qy, ccx, ccy, field_at(F, qx, qy), field_at(F, ccx, ccy));
            if (c < rhs) rhs = c;
            if (rhs == c) b0 = qx * F.EY + qy;
        }
        out_rhs = rhs;
    }
    return out_rhs;
}
__global__ void k_info(PathField F, int lvl, int x0, int y0, int nx, int ny, int32_t *out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nx * ny) return;
    int b0, b1;
    derived_rhs(F, lvl, x0 + e / ny, y0 + e % ny, b0, b1);
    out[2 * e] = b0;
    out[2 * e + 1] = b1;
}
// What the reference would hold in its priority queue: the elements that are not consistent (G != RHS, ReplannerBase.h:110-115), with both
// values -- the caller makes the keys (calculate_key: min(g, rhs) [+ heuristic]).  In no particular order; *count = how many there are
// (the first `cap` of them are stored).
__global__ void k_queue_scan(PathField F, int lvl, int gx, int gy, int cap, int32_t *xy, float *gr, unsigned int *count) {
    const int n = F.EX * F.EY;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const int x = e / F.EY, y = e - x * F.EY;
        const float g = field_at(F, x, y);
        int b0, b1;
        const float r = (x == gx && y == gy) ? 0.0f : derived_rhs(F, lvl, x, y, b0, b1);
        if (g == r || (g != g && r != r)) continue;
        // MS-DFM: the float fixed point of the upwind quadratic is not unique -- the field is a fixed point of the
        // relaxation's evaluation of the candidates, min_rhs<1>() as restated here may round the last bits the other way: within the
        // planner's tolerance (UFM_DFM_RTOL, include/ufm.h) a cell counts as consistent
        // (both values finite: with G = +inf the bound is +inf too and `inf <= inf` would drop every never-expanded frontier cell -- the bulk of the
        //  reference's queue after a focused step)
        if (F.cells && g < INFINITY && r < INFINITY && fabsf(g - r) <= UFM_DFM_RTOL * fmaxf(fabsf(g), 1.0f)) continue;
        const unsigned int i = atomicAdd(count, 1u);
        if (i < (unsigned int)cap) { xy[2 * i] = x; xy[2 * i + 1] = y; gr[2 * i] = g; gr[2 * i + 1] = r; }
    }
}
