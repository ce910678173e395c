// The reference's host-side launch-size helpers (render/renderutils/c_src/common.cpp: getLaunchBlockSize, getLaunchGridSize),
// compiled unchanged as their own translation unit, as in the reference's build (TEST INFRASTRUCTURE).
#include <common.cpp>
