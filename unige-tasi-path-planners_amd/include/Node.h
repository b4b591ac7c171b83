// Node.h -- same header name as the reference's ProjectToolkit/include/Node.h; the type lives in
// GridTypes.h (the three value types refer to each other).
#pragma once
#include "GridTypes.h"
