// queue_probe.hip -- which hardware queue does HIP give the n-th stream of a process?  Creates N streams (optionally with a
// priority), launches one tiny kernel on each; run under `rocprofv3 --kernel-trace` and read Stream_Id -> Queue_Id from the trace.
// usage: queue_probe N [priority|none] [extra streams created first and kept, as a context's own streams would be]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
__global__ void k_probe(int *p) { if (threadIdx.x == 0) atomicAdd(p, 1); }
int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 32;
    const bool prio = argc > 2 && strcmp(argv[2], "none") != 0;
    const int p = prio ? atoi(argv[2]) : 0;
    const int extra = argc > 3 ? atoi(argv[3]) : 0;
    int *d = nullptr;
    if (hipMalloc(&d, 4) != hipSuccess) return 1;
    hipMemset(d, 0, 4);
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    printf("priority range: least %d greatest %d\n", lo, hi);
    std::vector<hipStream_t> keep(extra), s(n);
    for (auto &x : keep) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    for (auto &x : keep) hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, x, d);
    for (auto &x : s) {
        if (prio) hipStreamCreateWithPriority(&x, hipStreamNonBlocking, p);
        else hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    }
    for (int rep = 0; rep < 2; ++rep)
        for (auto &x : s) hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, x, d);
    hipDeviceSynchronize();
    printf("done %d streams\n", n);
    return 0;
}
