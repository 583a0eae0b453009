// libgru4rec_hip.so -- host side of the C ABI declared in include/gru4rec_hip.h.
// Owns device memory, the HIP stream, the captured step graph and the (optional) RCCL communicator.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdarg>
#include <string>
#include <vector>

#include "g4r_eval_kernels.cuh"
#include "g4r_sync_kernels.cuh"
#include "g4r_micro_kernels.cuh"
#include "g4r_wide_kernels.cuh"

// Host code below is compiled in the host pass only: on the device pass the descriptor pointer fields are
// address-space qualified (g4r_device.cuh) and the template kernels are instantiated explicitly.
#if !defined(__HIP_DEVICE_COMPILE__)
#include "g4r_host_model.hpp"

extern "C" {
#include "g4r_host_create.hpp"
#include "g4r_host_plan.hpp"
#include "g4r_host_step.hpp"
#include "g4r_host_predict.hpp"
#include "g4r_host_comm.hpp"
#include "g4r_host_sync.hpp"
#include "g4r_host_debug.hpp"
}  // extern "C"
#endif  // !__HIP_DEVICE_COMPILE__
