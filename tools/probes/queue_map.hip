// Which hardware queue does HIP give a stream?  Creates streams of every priority in a fixed order, launches one labelled kernel on
// each (grid.x = label) and leaves the answer to `rocprofv3 --kernel-trace` (Queue_Id, Stream_Id, Grid_Size_X per dispatch).
// build: hipcc --offload-arch=gfx950 -o build/queue_map tools/probes/queue_map.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_label(int* p) { if (p) p[0] = 1; }
int main()
{
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    std::printf("priority range: least %d .. greatest %d\n", lo, hi);
    hipStream_t s[16];
    int prio[16];
    int n = 0;
    s[n] = nullptr; prio[n++] = 99;                                  // the NULL stream
    for (int k = 0; k < 5; ++k) { hipStreamCreateWithFlags(&s[n], hipStreamNonBlocking); prio[n++] = 0; }
    for (int k = 0; k < 4; ++k) { hipStreamCreateWithPriority(&s[n], hipStreamNonBlocking, hi); prio[n++] = hi; }
    for (int k = 0; k < 3; ++k) { hipStreamCreateWithPriority(&s[n], hipStreamNonBlocking, lo); prio[n++] = lo; }
    for (int rep = 0; rep < 2; ++rep)
        for (int i = 0; i < n; ++i) {
            hipLaunchKernelGGL(k_label, dim3(i + 1), dim3(64), 0, s[i], nullptr);
            hipStreamSynchronize(s[i]);
        }
    for (int i = 0; i < n; ++i) std::printf("label %d: priority %d\n", i + 1, prio[i]);
    return 0;
}
