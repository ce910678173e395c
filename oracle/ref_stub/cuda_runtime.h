// <cuda_runtime.h> stand-in (render/renderutils/c_src/common.cpp:12): the host shim.
#pragma once
#include "cuda_host_shim.h"
