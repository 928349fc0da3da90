// shape_group.hip -- one group of scenario shapes as a translation unit of its own (-DMATE_SHAPE_GROUP=k; mate_amd/build.py).
#include <hip/hip_runtime.h>
#include "shape_groups.hpp"
#include "shape_group.inc"
