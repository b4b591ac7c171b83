// FieldDPlanner.h -- Field D* planner (reference: FieldDStar/FieldDPlanner.h:29-83) on the
// MI355X engine.  OptimizationLevel 0/1 select the reference's plain / back-pointer variants;
// both compute the same consistent field, which is what the engine produces.
#ifndef UFM_FIELDDPLANNER_H
#define UFM_FIELDDPLANNER_H
#include "ReplannerBase.h"

template <int OptimizationLevel>
class FieldDPlanner
    : public ReplannerBase<FieldDPlanner<OptimizationLevel>, Node,
                           typename std::conditional<OptimizationLevel == 0, void, Node>::type, ufm_detail::key_type> {
  static_assert(OptimizationLevel == 0 || OptimizationLevel == 1, "FieldDPlanner has levels 0 and 1");
 public:
  typedef ReplannerBase<FieldDPlanner<OptimizationLevel>, Node,
                        typename std::conditional<OptimizationLevel == 0, void, Node>::type, ufm_detail::key_type> Base;
  typedef typename Base::Key Key;
  typedef typename Base::Map Map;
  explicit FieldDPlanner(int device = 0) : Base(UFM_ALGO_FD, OptimizationLevel, ufm_detail::kHeuristic, device) {}
};
#endif
