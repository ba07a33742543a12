// graph_probe.hip -- is replaying a captured 5-kernel chain (hipGraphLaunch) cheaper than five launches, per "frame" with a
// stream synchronisation after each, on T host threads with a stream each?  Kernels spin for a fixed number of clock ticks.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void spin(unsigned long long ticks, int *sink)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (sink && threadIdx.x == 9999) *sink = 1;
}
static void chain(hipStream_t s, const unsigned long long *us)
{
    for (int k = 0; k < 5; ++k) spin<<<64, 64, 0, s>>>(us[k] * 100ull, nullptr);   // wall clock: 100 MHz
}
int main()
{
    const unsigned long long us[5] = {4, 9, 35, 12, 7};   // colour convert, pyramid, PyrLK, Harris, merge (isolated times)
    for (int threads : {1, 4, 16, 32}) {
        for (int mode = 0; mode < 2; ++mode) {
            const int frames = 2000;
            std::vector<std::thread> th;
            const auto t0 = std::chrono::steady_clock::now();
            for (int t = 0; t < threads; ++t)
                th.emplace_back([&, mode] {
                    hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
                    hipGraphExec_t exec = nullptr;
                    if (mode == 1) {
                        hipGraph_t g;
                        (void)hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
                        chain(s, us);
                        (void)hipStreamEndCapture(s, &g);
                        if (hipGraphInstantiate(&exec, g, nullptr, nullptr, 0) != hipSuccess) { printf("instantiate failed\n"); return; }
                    }
                    for (int f = 0; f < frames; ++f) {
                        if (mode == 1) (void)hipGraphLaunch(exec, s); else chain(s, us);
                        (void)hipStreamSynchronize(s);
                    }
                    (void)hipStreamDestroy(s);
                });
            for (auto &x : th) x.join();
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("%2d threads, %s: %.1f us per frame per thread, %.0f frames/s in all (kernel time 67 us per frame)\n", threads,
                   mode ? "graph replay " : "five launches", dt / frames * 1e6, threads * frames / dt);
        }
    }
    return 0;
}
