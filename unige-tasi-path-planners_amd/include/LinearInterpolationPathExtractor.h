// LinearInterpolationPathExtractor.h -- the reference's path extractor
// (PathExtraction/LinearInterpolationPathExtractor.h:8-40) on the MI355X engine.
//
// Same template signature, constructor and public members (path_, cost_, total_cost, total_dist,
// lookahead, max_steps, e_time, allow_indirect_traversals, extract_path()), so the reference
// drivers' use of it (Tests/Planners/FDSTAR/main.cpp:78-82,116-136,158-160) compiles unchanged.
// extract_path() is one call of ufm_extract_path: the walk over the RHS field, its lookahead and
// the traversal case tables run on the device (one wavefront), the field stays in HBM; start and
// goal are the positions last given to the planner (Graph::start_pos_, goal_pos_).
#ifndef UFM_LINEAR_INTERPOLATION_PATH_EXTRACTOR_H
#define UFM_LINEAR_INTERPOLATION_PATH_EXTRACTOR_H
#include <iostream>
#include <vector>

#include "ExpandedMap.h"
#include "Graph.h"
#include "InterpolatedTraversal.h"
#include "ufm.h"

template <typename E, typename T>
class LinearInterpolationPathExtractor {
 public:
  LinearInterpolationPathExtractor(const ExpandedMap<E, T> &map, const Graph &grid) : map(map), grid(grid) {}

  void extract_path() {
    path_.clear();
    cost_.clear();
    total_cost = 0;
    total_dist = 0;
    e_time = 0;
    const int cap_p = 3 * max_steps + 1, cap_c = 2 * max_steps;
    std::vector<float> xy(2 * static_cast<size_t>(cap_p)), sc(static_cast<size_t>(cap_c));
    ufm_path_info info{};
    last_error = ufm_extract_path(map.native_handle(), max_steps, lookahead ? 1 : 0, allow_indirect_traversals ? 1 : 0,
                                  xy.data(), cap_p, sc.data(), cap_c, &info);
    if (last_error != UFM_OK) return;
    total_cost = info.total_cost;
    total_dist = info.total_dist;
    e_time = info.e_ms;
    if (info.n_points == 0) {                       // impl:49-51
      std::cerr << "[Extraction] No valid path exists" << std::endl;
      return;
    }
    path_.reserve(info.n_points);
    for (int i = 0; i < info.n_points; ++i) path_.emplace_back(xy[2 * i], xy[2 * i + 1]);
    cost_.assign(sc.begin(), sc.begin() + info.n_costs);
  }

  std::vector<Position> path_{};
  std::vector<float> cost_{};
  float total_cost = 0;
  float total_dist = 0;
  bool lookahead = true;
  int max_steps = 20;
  float e_time = 0;
  bool allow_indirect_traversals = true;
  int last_error = 0;       // ufm code of the last call (not in the reference)

 private:
  const ExpandedMap<E, T> &map;
  const Graph &grid;
};
#endif
