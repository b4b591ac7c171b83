// ReplannerBase.h -- the planner driver surface of the reference
// (ProjectToolkit/include/ReplannerBase.h:29-163) on top of the MI355X engine (libufm.so).
//
// Same template signature, typedefs, public data members and methods, so a driver written for
// the reference (Tests/Planners/*/main.cpp) compiles against it: reset(), step(),
// set_occupancy_threshold(), set_heuristic_multiplier(), set_map(), patch_map(), set_start(),
// set_goal(), get_expanded_map(), get_grid(); u_time / p_time, num_nodes_updated /
// num_nodes_expanded, grid, map, priority_queue -- the last as a read view (PriorityQueue.h: size / empty /
// top_key / top_value / ordered iteration over the elements that are not consistent, derived from the field;
// the engine has no heap, its tile queues live on the device).  What is not here: enqueue_if_inconsistent()
// and the queue's mutators -- nothing outside the device decides what is relaxed next.
// step() returns LOOP_OK / LOOP_FAILURE_NO_GRAPH / LOOP_FAILURE_NO_GOAL like the reference; a
// device error is returned as the negative ufm code and kept in last_error.
#ifndef UFM_REPLANNER_BASE_H
#define UFM_REPLANNER_BASE_H

#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>

#include "ExpandedMap.h"
#include "Graph.h"
#include "GridTypes.h"
#include "PriorityQueue.h"
#include "ufm.h"

#define LOOP_OK 0
#define LOOP_FAILURE_NO_GRAPH -1
#define LOOP_FAILURE_NO_GOAL -2

template <typename Derived, typename MapElem_, typename MapInfo_, typename QueueKey_>
class ReplannerBase {
 public:
  typedef QueueKey_ Key;
  typedef MapElem_ Elem;
  typedef MapInfo_ Info;
  typedef ExpandedMap<MapElem_, MapInfo_> Map;
  typedef PriorityQueue<QueueKey_, MapElem_> Queue;
  float u_time = 0, p_time = 0;

  void reset() { initialize_search = true; check(ufm_reset(handle_)); map.invalidate(); priority_queue.invalidate(); }

  int step() {
    if (initialize_graph) return LOOP_FAILURE_NO_GRAPH;
    if (!goal_set) return LOOP_FAILURE_NO_GOAL;
    ufm_stats st{};
    const int rc = ufm_step(handle_, &st);
    map.invalidate();
    priority_queue.invalidate();
    if (rc != UFM_OK) { last_error = rc; return rc; }
    u_time = st.u_ms; p_time = st.p_ms;
    num_nodes_updated = st.updated; num_nodes_expanded = st.expanded;
    stats = st;
    new_goal = initialize_search = new_start = false;
    return LOOP_OK;
  }

  void set_occupancy_threshold(float threshold) { grid.set_occupancy_threshold(threshold); check(ufm_set_occupancy_threshold(handle_, threshold)); }
  void set_heuristic_multiplier(float mult) { heuristic_multiplier = mult; check(ufm_set_heuristic_multiplier(handle_, mult)); }

  void set_map(const std::shared_ptr<uint8_t> &new_map, int w, int h) {
    grid.init(new_map, w, h);
    check(ufm_set_map(handle_, new_map.get(), w, h));
    int nx = 0, ny = 0;
    check(ufm_field_dims(handle_, &nx, &ny));
    map.set_dims(nx, ny);
    initialize_graph = false;
  }
  void patch_map(const std::shared_ptr<uint8_t> &patch, int x, int y, int w, int h) {
    grid.update(patch, x, y, w, h);
    check(ufm_patch_map(handle_, patch.get(), x, y, w, h));
  }
  void set_start(const Position &pos) { grid.set_start(pos); new_start = true; check(ufm_set_start(handle_, pos.x, pos.y)); }
  void set_goal(const Position &point) {
    if constexpr (std::is_same<MapElem_, Node>::value) new_goal = grid.goal_node_ != Node(point);
    else new_goal = grid.goal_cell_ != Cell(point);
    grid.set_goal(point);
    goal_set = true;
    check(ufm_set_goal(handle_, point.x, point.y));
  }

  const Map &get_expanded_map() { return map; }
  const Graph &get_grid() { return grid; }

  /** engine knobs without a reference counterpart (ufm_set_param) */
  void set_engine_param(const char *name, double v) { check(ufm_set_param(handle_, name, v)); }
  ufm_t *native_handle() { return handle_; }

  unsigned long num_nodes_updated = 0;
  unsigned long num_nodes_expanded = 0;
  float heuristic_multiplier = 1;
  bool initialize_graph = true;
  bool initialize_search = true;
  bool goal_set = false;
  bool new_goal = false;
  bool new_start = false;
  int last_error = 0;
  ufm_stats stats{};

  Queue priority_queue;
  Graph grid;
  Map map;

 protected:
  ReplannerBase(int algo, int opt_lvl, bool use_heuristic, int device = 0) {
    const int rc = ufm_create(&handle_, algo, opt_lvl, use_heuristic ? 1 : 0, device);
    if (rc != UFM_OK) throw std::runtime_error("ufm_create failed with code " + std::to_string(rc) + " (no MI355X / libufm.so?)");
    map.attach(handle_);
    priority_queue.attach(handle_, [this](const MapElem_ &s, float cost_so_far) { return calculate_key(s, cost_so_far); });
  }

  /** FieldDPlanner_impl.h:177-186, ShiftedGridPlanner_impl.h (same), DynamicFastMarching_impl.h:146-155 */
  Key calculate_key(const MapElem_ &s, float cost_so_far) const {
    if constexpr (std::is_same<Key, float>::value) {
      (void)s;
      return cost_so_far;
    } else {
      float dist;
      if constexpr (std::is_same<MapElem_, Node>::value) dist = s.distance(grid.start_pos_);
      else dist = grid.start_cell_.distance(s);
      return {cost_so_far + heuristic_multiplier * dist, cost_so_far};
    }
  }
  ~ReplannerBase() { if (handle_) ufm_destroy(handle_); }
  ReplannerBase(const ReplannerBase &) = delete;
  ReplannerBase &operator=(const ReplannerBase &) = delete;

 private:
  void check(int rc) { if (rc != UFM_OK) last_error = rc; }
  ufm_t *handle_ = nullptr;
};

namespace ufm_detail {
#ifdef NO_HEURISTIC
using key_type = float;
constexpr bool kHeuristic = false;
#else
using key_type = std::pair<float, float>;
constexpr bool kHeuristic = true;
#endif
}  // namespace ufm_detail

#endif  // UFM_REPLANNER_BASE_H
