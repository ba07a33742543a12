// valu_rate.hip -- measures the wave64 VALU issue rate per SIMD on this GPU (f32 FMA, int add3, DPP mov),
// at 1, 2 and 4 waves per SIMD.  Prints cycles per wave-instruction per SIMD (shader clock from s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ __launch_bounds__(256) void rate(float *out, unsigned long long *cyc, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0001f, c = 0.5f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = fmaf(a0, m, c); a1 = fmaf(a1, m, c); a2 = fmaf(a2, m, c); a3 = fmaf(a3, m, c);
                a4 = fmaf(a4, m, c); a5 = fmaf(a5, m, c); a6 = fmaf(a6, m, c); a7 = fmaf(a7, m, c);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                int b0 = __float_as_int(a0), b1 = __float_as_int(a1), b2 = __float_as_int(a2), b3 = __float_as_int(a3);
                int b4 = __float_as_int(a4), b5 = __float_as_int(a5), b6 = __float_as_int(a6), b7 = __float_as_int(a7);
                b0 = b0 * 3 + i; b1 = b1 * 3 + i; b2 = b2 * 3 + i; b3 = b3 * 3 + i;   // v_mad_u32_u24 / lshl_add forms
                b4 = b4 * 3 + i; b5 = b5 * 3 + i; b6 = b6 * 3 + i; b7 = b7 * 3 + i;
                a0 = __int_as_float(b0); a1 = __int_as_float(b1); a2 = __int_as_float(b2); a3 = __int_as_float(b3);
                a4 = __int_as_float(b4); a5 = __int_as_float(b5); a6 = __int_as_float(b6); a7 = __int_as_float(b7);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
    float *out; unsigned long long *cyc, h[4096];
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&cyc, sizeof(h));
    const int iters = 2000;
    for (int kind = 0; kind < 2; ++kind)
        for (int wps = 1; wps <= 4; wps *= 2) {   // waves per SIMD: blocks of 256 threads = 1 wave on each SIMD of a CU
            const int blocks = 256 * wps;
            for (int rep = 0; rep < 2; ++rep) {
                if (kind == 0) rate<0><<<blocks, 256>>>(out, cyc, iters); else rate<1><<<blocks, 256>>>(out, cyc, iters);
                if (hipDeviceSynchronize() != hipSuccess) { printf("fail\n"); return 1; }
            }
            hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
            double mean = 0; for (int i = 0; i < blocks; ++i) mean += h[i]; mean /= blocks;
            const double instr = (double)iters * 64;   // per wave
            printf("%s waves/SIMD %d: wave lifetime %.0f cyc, %.2f cyc per wave-instr per wave, %.2f cyc per wave-instr per SIMD\n",
                   kind ? "int mad" : "v_fma_f32", wps, mean, mean / instr, mean / (instr * wps));
        }
    return 0;
}
