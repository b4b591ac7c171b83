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

#include "ufm_defs.h"
#include "ufm_ops.h"
#include "ufm_relax.h"
#include "ufm_control.h"

#include "ufm_region.h"

#include "ufm_host.h"

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
int ufm_debug_plog(unsigned int *out, int cap) {     // out[cap][4]
    unsigned int n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_nplog), sizeof(n)) != hipSuccess) return UFM_ERR_HIP_BASE;
    if ((int)n > cap) n = cap;
    if (n > PLOG_MAX) n = PLOG_MAX;
    if (n && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_plog), sizeof(unsigned int) * 4 * n) != hipSuccess) return UFM_ERR_HIP_BASE;
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
// round 4 experiment: an arrival estimate per tile ([TX][TY] floats of map 0, +inf = no estimate) for the gated first visits of the resident kernel
// (ufm_set_param "dag", 1); tools/dag_probe.py hands in the true first-arrival values of an earlier plan.  Not part of include/ufm.h.
int ufm_debug_set_tile_order(ufm_t *p, const float *a, int n) {
    if (!p || !p->e->allocated || !a || n != p->e->P.NT) return UFM_ERR_INVALID;
    hipStreamSynchronize(p->e->stream);
    if (hipMemcpy(p->e->P.dag_a, a, sizeof(float) * (size_t)n, hipMemcpyHostToDevice) != hipSuccess) return UFM_ERR_HIP_BASE;
    p->e->dag_have = true;
    return UFM_OK;
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
    else if (!std::strcmp(name, "cont_raise")) e->cont_raise = value < 0 ? 0 : (int)value;
    else if (!std::strcmp(name, "cont_lower")) e->cont_lower = value < 0 ? 0 : (int)value;
    else if (!std::strcmp(name, "region_sweeps")) e->region_sweeps = value < 16 ? 16 : (int)value;
    else if (!std::strcmp(name, "batch_margin")) e->batch_margin = (int)value;
    else if (!std::strcmp(name, "raise_margin")) e->raise_margin = (float)value;
    else if (!std::strcmp(name, "tail_grid")) e->tail_grid = value < 1 ? 1 : (int)value;
    else if (!std::strcmp(name, "profile_stride")) e->profile_stride = value < 1 ? 1 : (int)value;
    else if (!std::strcmp(name, "focused")) e->focused = value != 0.0;
    else if (!std::strcmp(name, "start_cell_floor")) e->start_cell_floor = value != 0.0;
    else if (!std::strcmp(name, "lazy_patches")) { if (e->allocated) { int rc = e->flush_deferred(); if (rc != UFM_OK) return rc; } e->lazy_patches = value != 0; }
    else if (!std::strcmp(name, "dag")) e->dag_mode = (int)value;
    else if (!std::strcmp(name, "dag_kappa")) e->dag_kappa = (float)value;
    else if (!std::strcmp(name, "dag_patience")) e->dag_patience = value < 1 ? 1 : (int)value;
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
int ufm_read_queue(ufm_t *p, int cap, int32_t *xy, float *g_rhs, int *total) { return p ? engine_read_queue(p->e, 0, cap, xy, g_rhs, total) : UFM_ERR_INVALID; }
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
