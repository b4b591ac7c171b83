// Macros.h -- the subset of ProjectToolkit/include/Macros.h a driver of the planner surface uses.
#pragma once
#include <optional>
using std::nullopt;
using std::optional;
/** sqrt(2) rounded to float, ProjectToolkit/Macros.cpp:2 */
static const float SQRT2 = 1.41421356237309504880168872420969807856967187537694f;
