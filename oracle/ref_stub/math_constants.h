// <math_constants.h> stand-in.  CUDA defines CUDART_PI as a DOUBLE literal and CUDART_PI_F as the float one; kernel.cu uses
// CUDART_PI, so `2.0 * CUDART_PI * u`, `costheta / CUDART_PI`, `atan2f(..) / (2.0f * CUDART_PI)` are double arithmetic on
// the device too (fast-math does not touch doubles).
#pragma once
#define CUDART_PI 3.1415926535897931e+0
#define CUDART_PI_F 3.141592654f
