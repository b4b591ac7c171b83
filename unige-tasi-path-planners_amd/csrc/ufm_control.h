// ufm_control.h -- the small kernels around the hot one: patches and seeds, queue moves, step begin / end, finalisation, self-checks
// (a piece of ufm_engine.hip, the engine's one translation unit: included there, inside its anonymous namespace)
#pragma once

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
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_nvis = 0u; g_nplog = 0u; }
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
// Round 4 experiment (P.dag_on): from the arrival estimates dag_a, per tile the threshold below which a neighbour counts as "clearly before it"
// -- halfway (kappa) between the tile's own estimate and its earliest neighbour's -- and the number of such neighbours (dag_left, in the owners' layout).
// A tile without an estimate (+inf / NaN: the estimate does not reach it) waits for nobody.
__global__ void k_dag_setup(DevParams P, float kappa) {
    for (int gt = blockIdx.x * blockDim.x + threadIdx.x; gt < P.NT; gt += gridDim.x * blockDim.x) {
        const int m = gt / P.NTm, t = gt - m * P.NTm, tx = t / P.TY, ty = t - tx * P.TY;
        const float a = P.dag_a[gt];
        float amin = INFINITY;
        for (int d = 0; d < 9; ++d) {
            const int ntx = tx + d / 3 - 1, nty = ty + d % 3 - 1;
            if (d == 4 || ntx < 0 || nty < 0 || ntx >= P.TX || nty >= P.TY) continue;
            amin = fminf(amin, P.dag_a[m * P.NTm + ntx * P.TY + nty]);
        }
        const float thr = (a < INFINITY && amin < a) ? a - kappa * (a - amin) : -INFINITY;
        int n = 0;
        for (int d = 0; d < 9; ++d) {
            const int ntx = tx + d / 3 - 1, nty = ty + d % 3 - 1;
            if (d == 4 || ntx < 0 || nty < 0 || ntx >= P.TX || nty >= P.TY) continue;
            n += P.dag_a[m * P.NTm + ntx * P.TY + nty] < thr ? 1 : 0;
        }
        P.dag_thr[gt] = thr;
        int o, s_;
        own_locate(P, gt, o, s_);
        P.dag_left[(size_t)o * P.own_slots + s_] = n;
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
        const int bq = quad_min_int(bp_byte<ALGO>(le, q, C, le.r == nv)), b = bq == 0x3FF ? BP_NONE : (bq & 0x1F);
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

