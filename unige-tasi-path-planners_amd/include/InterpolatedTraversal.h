// InterpolatedTraversal.h -- the value types of the reference's traversal tables
// (ProjectToolkit/include/InterpolatedTraversal.h:11-39), kept for source compatibility.
// The case tables themselves (TraversalTypeI..B x Corner / ContiguousEdge / OppositeEdge,
// InterpolatedTraversal.cpp:6-778) are evaluated on the device by ufm_extract_path
// (csrc/ufm_path.h); there is no host copy of them.
#ifndef UFM_INTERP_TRAVERSAL_H
#define UFM_INTERP_TRAVERSAL_H
#include <vector>

#include "GridTypes.h"
#include "Macros.h"

/** Parameters of one linear-interpolation traversal (labels of the Field D* paper). */
struct TraversalParams {
  Position p0;      // point aligned with p1, but not with p2
  Node p1, p2;      // the edge to traverse
  float b;          // cost of the cell across p0-p1
  float c;          // cost of the traversed cell
  float f;          // g(p1) - g(p2)
  float g1, g2;
  float p, q;       // offsets of the starting point inside the (unit) cell
};

/** A piece of trajectory: way points, their costs, remaining cost to goal. */
struct PathAdditions {
  std::vector<Position> steps;
  std::vector<float> stepcosts;
  float cost_to_goal;
};
#endif
