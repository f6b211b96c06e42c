// Optional per-kernel timing for bench.py's roofline line: when enabled, the instrumented launches go through
// hipExtLaunchKernelGGL with a (start, stop) event pair that brackets exactly the kernel on its own stream, and the
// algorithmic work (flops or bytes) of each launch is recorded next to it.  Disabled (the default) it costs nothing.
#pragma once
#include <hip/hip_ext.h>

#include "common.h"

namespace cf {

enum ProfileKernel {
    PK_CONV_MT1 = 0, PK_CONV_MT2 = 1, PK_CONV_MT4 = 2,      // work = flops
    PK_CORRVOL_S1 = 3, PK_CORRVOL_S2 = 4, PK_CORRVOL_S4 = 5, // work = algorithmic bytes
    PK_CONV_F16S = 6,                                        // work = flops
    PK_ALLPAIRS = 7,                                         // RAFT all-pairs volume + pyramid pooling; work = algorithmic bytes
    PK_CORR_LOOKUP = 8,                                      // RAFT correlation lookup; work = algorithmic bytes
    PK_CONVEX_UP = 9,                                        // RAFT convex upsampling; work = algorithmic bytes
    PK_GN_APPLY = 10,                                        // GroupNorm apply passes; work = algorithmic bytes
    PK_WARP = 11, PK_WARP_LABELS = 12, PK_JACOBIAN = 13,     // VoxelMorph warp family (2-D); work = algorithmic bytes
    PK_CONV_SMALL_COUT = 14,                                 // direct 3x3 convolution to <= 4 output channels (flow heads); work = algorithmic bytes
    PK_CONV_STREAM = 15,                                     // conv_stream_kernel (same arithmetic as PK_CONV_F16S); work = flops
    PK_CONV_WINO = 16,                                       // conv_wino_kernel (row Winograd F(2,3), 3-term split); work = direct-form flops
    PK_COUNT = 17
};

bool profile_on();
// returns false when the event pool is exhausted (the launch then goes out un-timed)
bool profile_events(int kid, double work, hipEvent_t* start, hipEvent_t* stop);

template <typename K, typename... Args>
static inline void launch_profiled(int kid, double work, K kernel, dim3 grid, dim3 block, hipStream_t s, Args... args) {
    hipEvent_t e0, e1;
    if (profile_on() && profile_events(kid, work, &e0, &e1)) {
        hipExtLaunchKernelGGL(kernel, grid, block, 0, s, e0, e1, 0, args...);
    } else {
        hipLaunchKernelGGL(kernel, grid, block, 0, s, args...);
    }
}

}  // namespace cf
