// Per-dispatch timing of the conv-stage kernels for bench.py's roofline figure (mcav_kernel_timer_* in mcav_conv.h).
// While the timer is on, a conv kernel is launched with hipExtLaunchKernelGGL and a start/stop event pair bound to THAT dispatch, so the
// elapsed time is the kernel's own (what rocprofv3 --kernel-trace reports), free of the dispatch gap that an event recorded in front of a
// launch includes.  Off (the default), launches are plain hipLaunchKernelGGL: the product path is unchanged.
#pragma once
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

namespace mcav {

bool kernel_timer_on();
void kernel_timer_add(hipEvent_t e0, hipEvent_t e1);

template <class K, class... A>
inline void timed_launch(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t s, A... args) {
    if (!kernel_timer_on()) {
        hipLaunchKernelGGL(kernel, grid, block, (unsigned)lds, s, args...);
        return;
    }
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        hipLaunchKernelGGL(kernel, grid, block, (unsigned)lds, s, args...);
        return;
    }
    hipExtLaunchKernelGGL(kernel, grid, block, (unsigned)lds, s, e0, e1, 0, args...);
    kernel_timer_add(e0, e1);
}

}  // namespace mcav
