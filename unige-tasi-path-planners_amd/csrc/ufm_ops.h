// ufm_ops.h -- the update operators of the three planner families, the four-lanes-per-node evaluation, back-pointer bytes, cost windows
// (a piece of ufm_engine.hip, the engine's one translation unit: included there, inside its anonymous namespace)
#pragma once

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
    // both triangles of a lane at once (k_replan_region sets these up at every burst): the two square roots side by side and unconditionally -- as two
    // `cgb ? sqrt_rn(..) : -1` they became two exec-mask branches one after the other, each a dozen dependent instructions long
    static __device__ __forceinline__ void set2(TriFD &tv, TriFD &th, float c, float bv, float bh) {
        const float c2 = c * c;
        float sv = sqrt_rn(fmaxf(c2 - bv * bv, 0.0f)), sh = sqrt_rn(fmaxf(c2 - bh * bh, 0.0f));
        asm volatile("" : "+v"(sv), "+v"(sh));
        const bool gv = c > bv, gh = c > bh, cinf = c == INFINITY;
        tv.bp = cinf ? INFINITY : (gv ? bv : c); th.bp = cinf ? INFINITY : (gh ? bh : c);
        tv.cbp = (gv && !cinf) ? sv : -1.0f; th.cbp = (gh && !cinf) ? sh : -1.0f;
        tv.bI = (c > bv * SQRT2F) ? bv : INFINITY; th.bI = (c > bh * SQRT2F) ? bh : INFINITY;
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
        TriFD::set2(tv, th, c, bv, bh);
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
// Tied candidates (round 4): the one the reference's min_rhs<1>() would keep -- it walks Graph::neighbors_8 (top, top-left, left, bottom-left, bottom,
// bottom-right, right, top-right: Graph.cpp:71-85) and lets the LAST tied neighbour win (`if (rhs == cost) bptr = ...`, FD impl:196-208) --, so that the
// stored byte names the neighbour min_rhs<1>() on the same field names (ufm_read_info == ufm_read_info_derived; round 3 kept the lowest code and the two
// views parted on the 7-8 % of the nodes whose triangles over one grid edge tie).  bp_ref(code) = that neighbour's position in neighbors_8.
// (FD impl:292-319, SG :422-436): "g1 + ..." (III, B) leans on p1 alone, "g2 + ..." (I, A) on p2 alone, the interpolated case (II) on both.  With the other
// vertex at +inf the same case is taken and gives the same value, so: an element is gone exactly when a vertex it depends on is gone (the invalidation of
// ufm_region.h follows these bits without evaluating anything).  MS-DFM: 3.
constexpr int BP_NONE = 0xFF;
// code (q << 1) | h -> index in Graph::neighbors_8 of the node b with RHS(s) = cost(s, b, ccw_neighbor(s, b)): codes 0..7 -> 1 2 0 7 4 3 5 6
__device__ __forceinline__ int bp_ref(int code) { return (0x65347021u >> (4 * code)) & 7; }
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
    if constexpr (ALGO == ALGO_DFM1) {
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
        e.h = (tH < tV) | ((tH == tV) & (q != 2));      // (a tie inside the lane: the later one in neighbors_8 order -- h = 1 except for cell q = 2)
        e.r = e.h ? tH : tV;
    }
    return e;
}
// ... and the byte for a lane that holds the quad's minimum (0x3FF for one that does not: the quad's smallest is the lowest winning code)
template <int ALGO>
__device__ __forceinline__ int bp_byte(const LaneEval &e, int q, const QuadConsts<ALGO> &C, bool winner) {
    int code, dep = 3;
    code = (q << 1) | (e.h ? 1 : 0);
    if constexpr (ALGO == UFM_ALGO_SG) dep = dep_sg(e.h ? e.gH : e.gV, e.gD, C.k);
    if constexpr (ALGO == UFM_ALGO_FD) {
        TriFD t;
        t.bp = e.h ? C.th.bp : C.tv.bp; t.cbp = e.h ? C.th.cbp : C.tv.cbp; t.bI = e.h ? C.th.bI : C.tv.bI;
        dep = dep_fd(e.h ? e.gH : e.gV, e.gD, C.k, t);
    }
    if constexpr (ALGO == ALGO_DFM1) return winner ? ((code << 2) | dep) : 0x3FF;          // (MS-DFM: the lowest code)
    return winner ? (((7 - bp_ref(code)) << 5) | (code << 2) | dep) : 0x3FF;      // (quad_min_int: the tied candidate latest in neighbors_8 order; the byte = its low 5 bits)
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
    if constexpr (ALGO == ALGO_DFM1) {
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

