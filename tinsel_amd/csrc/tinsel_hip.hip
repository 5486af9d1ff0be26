// tinsel_hip.hip -- host side of the C-ABI (include/tinsel_hip.h): scene flattening/upload,
// camera set-up, batch scheduling of the streaming pipeline, statistics and timing.
//
// Replaces the reference's GpuRenderer (src/render.cu:978-1110).  Built only with
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off  (tinsel_amd/build.py)

#include "../../include/tinsel_hip.h"

#include "tn_launch.h"
#include "tn_lbvh.h"
#include "tn_ubench.h"
#include "tn_selftest.h"

#include "tn_sort.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>          // types only: the library is dlopen'ed by the first multi-GPU group (no link-time dependency)

#include <dlfcn.h>

#include <cmath>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace tn;

namespace {

thread_local std::string g_error;

int fail(const std::string& msg)
{
    g_error = msg;
    fprintf(stderr, "tinsel_hip: %s\n", msg.c_str());
    return -1;
}

#define HIP_TRY(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess)                                                                               \
            return fail(std::string(#expr) + ": " + hipGetErrorString(_e));                                 \
    } while (0)

#define HIP_TRY_NULL(expr)                                                                                  \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess) {                                                                             \
            fail(std::string(#expr) + ": " + hipGetErrorString(_e));                                        \
            return nullptr;                                                                                 \
        }                                                                                                   \
    } while (0)

} // namespace

// The host side by concern, one translation unit (the kernels and everything above are shared):
#include "tn_host_layout.h"       // BVH re-layout, arena
#include "tn_host_state.h"        // struct tinsel_hip
#include "tn_host_batch.h"        // launches, grids and regions, render_batch / render_impl
#include "tn_host_lookahead.h"    // the one-pass-per-call pattern
#include "tn_host_bvh_build.h"    // device BVH build

// ===========================================================================
// C-ABI
#include "tn_host_create.h"       // create / destroy
#include "tn_host_api.h"          // everything else of include/tinsel_hip.h for one device
#include "tn_host_group.h"        // tinsel_hip_group
