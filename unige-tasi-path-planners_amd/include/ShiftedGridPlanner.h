// ShiftedGridPlanner.h -- Shifted-Grid Fast Marching / MFD* planner (reference:
// ShiftedGridFastMarching/ShiftedGridPlanner.h:29-88) on the MI355X engine; levels 0/1/2.
#ifndef UFM_SHIFTEDGRIDPLANNER_H
#define UFM_SHIFTEDGRIDPLANNER_H
#include "ReplannerBase.h"

template <int OptimizationLevel>
class ShiftedGridPlanner
    : public ReplannerBase<ShiftedGridPlanner<OptimizationLevel>, Node,
                           typename std::conditional<OptimizationLevel == 0, void, Node>::type, ufm_detail::key_type> {
  static_assert(OptimizationLevel >= 0 && OptimizationLevel <= 2, "ShiftedGridPlanner has levels 0, 1 and 2");
 public:
  typedef ReplannerBase<ShiftedGridPlanner<OptimizationLevel>, Node,
                        typename std::conditional<OptimizationLevel == 0, void, Node>::type, ufm_detail::key_type> Base;
  typedef typename Base::Key Key;
  typedef typename Base::Map Map;
  explicit ShiftedGridPlanner(int device = 0) : Base(UFM_ALGO_SG, OptimizationLevel, ufm_detail::kHeuristic, device) {}
};
#endif
