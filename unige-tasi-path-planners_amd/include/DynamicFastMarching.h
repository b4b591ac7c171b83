// DynamicFastMarching.h -- Multi-Stencil Dynamic Fast Marching planner on cells (reference:
// DynamicFastMarching/DynamicFastMarching.h:29-81) on the MI355X engine; levels 0/1.
#ifndef UFM_DYNAMICFASTMARCHING_H
#define UFM_DYNAMICFASTMARCHING_H
#include "ReplannerBase.h"

template <int OptimizationLevel>
class DFMPlanner
    : public ReplannerBase<DFMPlanner<OptimizationLevel>, Cell,
                           typename std::conditional<OptimizationLevel == 0, void, std::pair<Cell, Cell>>::type, ufm_detail::key_type> {
  static_assert(OptimizationLevel == 0 || OptimizationLevel == 1, "DFMPlanner has levels 0 and 1");
 public:
  typedef ReplannerBase<DFMPlanner<OptimizationLevel>, Cell,
                        typename std::conditional<OptimizationLevel == 0, void, std::pair<Cell, Cell>>::type, ufm_detail::key_type> Base;
  typedef typename Base::Key Key;
  typedef typename Base::Map Map;
  explicit DFMPlanner(int device = 0) : Base(UFM_ALGO_DFM, OptimizationLevel, ufm_detail::kHeuristic, device) {}
};
#endif
