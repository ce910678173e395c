// <cuda.h> as the reference's renderutils sources include it: everything comes from the host shim.
#pragma once
#include "cuda_host_shim.h"
