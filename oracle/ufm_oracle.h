/*
 * ufm_oracle.h -- TEST INFRASTRUCTURE ONLY (parity checker + CPU baseline).
 *
 * CPU restatement, in plain C, of the reference's priority-queue driven
 * D*-Lite style replanners (Field D*, Shifted-Grid FM, Multi-Stencil DFM).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product path (libufm.so, HIP) never links it.
 *
 * PARITY PINNING STATUS: pinned, for Field D* with heuristic keys, on the two outputs of the reference itself that its tree holds: the
 * console logs of two whole missions of its Field D* planner process (level 0) on its own bitmaps (Tests/Results/{noise-trap,wall-b}/
 * planner_opt0.log, written by an older revision of the sources).  In that revision (orc_set_revision(ORC_REV_LOG): start cell by
 * truncation, update() without the corner nodes on the far map borders -- both identified by search against the logs) this restatement --
 * planner, replans under a moving start, path extractor -- replays BOTH logs closed-loop in full: 134 + 89 steps, every position, path cost,
 * path length and "nodes updated" to the last printed digit, and 177 of 178 "nodes expanded"; as the current sources stand it replays the
 * first log's paths (tests/test_reference_mission.py).  Ablations (orc_set_fd_ablation) say which branches of compute_optimal_cost the logs
 * cover: B, II, A and Type I; NOT Type III.  Everything else -- MS-DFM, SG's own comparisons, full fields -- is "parity unpinned":
 * cross-checked against the numbers SURVEY.md App. E recorded from a build of the reference with stand-in headers (tests/test_oracle.py),
 * which pins nothing.  The reference ships no golden vectors and cannot be compiled in this image (its three header-only dependencies are
 * un-vendored empty submodules; writing stand-ins for them is not allowed), so there is no oracle/_ref build.
 * The restatement follows the reference sources function by function (citations in
 * ufm_oracle.c / ufm_path_oracle.c).
 */
#ifndef UFM_ORACLE_H
#define UFM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_planner orc_t;

enum { ORC_ALGO_FD = 0, ORC_ALGO_SG = 1, ORC_ALGO_DFM = 2 };

#define ORC_LOOP_OK 0
#define ORC_LOOP_FAILURE_NO_GRAPH (-1)
#define ORC_LOOP_FAILURE_NO_GOAL (-2)
#define ORC_LOOP_RUNAWAY (-75)   /* safety net of the tests: plan() expanded > 256 x elements (see expansion_cap) */

orc_t *orc_create(int algo, int opt_lvl, int use_heuristic);
void orc_destroy(orc_t *p);

/* ORC_REV_CURRENT: the reference's sources as they stand (default).  ORC_REV_LOG: the older revision that wrote the reference's two
 * recorded mission logs -- start cell by truncation, update() without the corner nodes on the far map borders (ufm_oracle.c:
 * orc_set_revision).  Only tests/test_reference_mission.py uses ORC_REV_LOG. */
enum { ORC_REV_CURRENT = 0, ORC_REV_START_CELL_FLOOR = 1, ORC_REV_UPDATE_SKIPS_FAR_BORDER = 2, ORC_REV_LOG = 3 };
void orc_set_revision(orc_t *p, int revision);
/* which case of compute_optimal_cost (FD impl:292-319, SG :422-436) the evaluations took / which case gave the minimum of a
 * min_rhs<0/1>() call, since the last reset; process-wide */
enum { ORC_CASE_FD_III = 0, ORC_CASE_FD_III_SQ, ORC_CASE_FD_II_CGB, ORC_CASE_FD_I, ORC_CASE_FD_A_CGB, ORC_CASE_FD_B, ORC_CASE_FD_II, ORC_CASE_FD_A,
       ORC_CASE_SG_B, ORC_CASE_SG_II, ORC_CASE_SG_A, ORC_NCASES };
void orc_case_counts(unsigned long *evaluated, unsigned long *won);
void orc_case_counts_reset(void);
/* test hook: FD's compute_optimal_cost without one of its branches (bit 0: the f^2 <= CATH(c,b) clause, 1: Type I, 2: the whole c > b chain); 0 = the operator */
void orc_set_fd_ablation(int mask);

void orc_reset(orc_t *p);
void orc_set_occupancy_threshold(orc_t *p, float thr);
void orc_set_heuristic_multiplier(orc_t *p, float m);
/* copies the raster (the reference shares and mutates the caller's) */
void orc_set_map(orc_t *p, const uint8_t *map, int width, int length);
void orc_patch_map(orc_t *p, const uint8_t *patch, int x, int y, int w, int h);
void orc_set_start(orc_t *p, float x, float y);
void orc_set_goal(orc_t *p, float x, float y);
int orc_step(orc_t *p);

/* field view: dense row-major [nx][ny]; nodes: (L+1)x(W+1), cells: LxW.
 * Elements never inserted into the reference's ExpandedMap read (inf,inf)
 * and have inmap == 0. */
int orc_field_dims(const orc_t *p, int *nx, int *ny);
const float *orc_g(const orc_t *p);
const float *orc_rhs(const orc_t *p);
const uint8_t *orc_inmap(const orc_t *p);
const int32_t *orc_bptr(const orc_t *p); /* level>=1: linear index (DFM: 2 per elem), else NULL */

/* bookkeeping for the benchmark's CPU leg: elements whose G differs after a step (the engine's "cells updated"), see ufm_oracle.c */
void orc_track_changes(orc_t *p, int on);
unsigned long orc_num_changed(const orc_t *p);
unsigned long orc_num_expanded(const orc_t *p);
unsigned long orc_num_updated(const orc_t *p);
unsigned long orc_map_size(const orc_t *p);
unsigned long orc_queue_size(const orc_t *p);
float orc_u_time_ms(const orc_t *p);
float orc_p_time_ms(const orc_t *p);
/* key at the top of the queue after the last step (inf,inf when empty) */
void orc_top_key(const orc_t *p, float *k1, float *k2);

/* ---- path extraction (consumer of the RHS field), ufm_path_oracle.c ----------------------
 * Restatement of LinearInterpolationPathExtractor::extract_path on plain arrays: `rhs` is a
 * dense row-major [nx][ny] field (cells != 0: cell-centred, as DFM's), `map` the [length][width]
 * raster.  Returns the number of path points (0: no valid path); points are written as (x,y)
 * pairs up to cap_pts, step costs up to cap_costs (*n_costs is the full count). */
int orc_extract_path_field(const float *rhs, int nx, int ny, int cells,
                           const uint8_t *map, int width, int length, int thr_uchar,
                           float start_x, float start_y, float goal_x, float goal_y,
                           int lookahead, int max_steps, int allow_indirect,
                           float *path_xy, int cap_pts, float *costs, int cap_costs,
                           int *n_costs, float *total_cost, float *total_dist);
/* the same on the oracle planner's own RHS field, raster, threshold, start and goal */
int orc_extract_path(const orc_t *p, int lookahead, int max_steps, int allow_indirect,
                     float *path_xy, int cap_pts, float *costs, int cap_costs,
                     int *n_costs, float *total_cost, float *total_dist);
int orc_threshold_uchar(const orc_t *p);
/* back-pointer(s) min_rhs<1/2> derives from the current G field (a pure function of it), and a
 * test hook that loads a G field */
float orc_min_rhs_info(const orc_t *p, int x, int y, int32_t *b0, int32_t *b1);
/* cost through a given back-pointer (node planners: node b; DFM: neighbour cell of the level-1 candidate) on the current G field */
float orc_cost_via(const orc_t *p, int x, int y, int bx, int by, int32_t *b0, int32_t *b1);
void orc_load_g(orc_t *p, const float *g);

#ifdef __cplusplus
}
#endif
#endif
