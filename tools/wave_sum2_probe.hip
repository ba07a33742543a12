// wave_sum2_probe.hip -- checks on the GPU that the merged two-value wave reduction (lane-parity interleave, row rotations,
// gfx950 v_permlane16/32_swap for the cross-row steps) is bit-identical to two separate butterfly reductions of
// pyrlk.hip, and times both (all SIMDs busy, 4 waves each).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v)
{
    v += dpp_mov<0xB1, 0xf>(v);
    v += dpp_mov<0x4E, 0xf>(v);
    v += dpp_mov<0x141, 0xf>(v);
    v += dpp_mov<0x140, 0xf>(v);
    v += dpp_mov<0x142, 0xa>(v);
    v += dpp_mov<0x143, 0xc>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ void wave_sum2(float a, float b, bool odd, float &sa, float &sb)
{
    a += dpp_mov<0xB1, 0xf>(a);
    b += dpp_mov<0xB1, 0xf>(b);
    float z = odd ? b : a;
    z += dpp_mov<0x4E, 0xf>(z);
    z += dpp_mov<0x124, 0xf>(z);    // row_ror:4
    z += dpp_mov<0x128, 0xf>(z);    // row_ror:8
    v2u s = __builtin_amdgcn_permlane16_swap(__float_as_uint(z), __float_as_uint(z), false, false);
    z = __uint_as_float(s.x) + __uint_as_float(s.y);
    s = __builtin_amdgcn_permlane32_swap(__float_as_uint(z), __float_as_uint(z), false, false);
    z = __uint_as_float(s.x) + __uint_as_float(s.y);
    sa = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z), 62));
    sb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z), 63));
}
__global__ __launch_bounds__(64) void check(const float *in, float *out, int trials)
{
    const int lane = threadIdx.x;
    for (int t = 0; t < trials; ++t) {
        const float a = in[(2 * t) * 64 + lane], b = in[(2 * t + 1) * 64 + lane];
        float n1, n2;
        const float o1 = wave_sum(a), o2 = wave_sum(b);
        wave_sum2(a, b, lane & 1, n1, n2);
        if (lane == 0) { out[4 * t] = o1; out[4 * t + 1] = o2; out[4 * t + 2] = n1; out[4 * t + 3] = n2; }
    }
}
template <int NEW>
__global__ __launch_bounds__(256) void timing(float *out, int iters)
{
    float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f;
    const bool odd = threadIdx.x & 1;
    for (int i = 0; i < iters; ++i) {
        float s1, s2;
        if (NEW) wave_sum2(a, b, odd, s1, s2);
        else { s1 = wave_sum(a); s2 = wave_sum(b); }
        a = a * 0.5f + s1 * 1e-3f; b = b * 0.5f + s2 * 1e-3f;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b;
}
int main()
{
    const int trials = 4096;
    float *h = (float *)malloc(trials * 128 * 4), *ho = (float *)malloc(trials * 16);
    srand(7);
    for (int i = 0; i < trials * 128; ++i) h[i] = ((rand() % 200001) - 100000) * 0.37f * ((i % 7) ? 1.0f : 1e-3f);
    float *d, *dout, *tout;
    (void)hipMalloc(&d, trials * 128 * 4); (void)hipMalloc(&dout, trials * 16); (void)hipMalloc(&tout, 4096 * 256 * 4);
    (void)hipMemcpy(d, h, trials * 128 * 4, hipMemcpyHostToDevice);
    check<<<1, 64>>>(d, dout, trials);
    (void)hipMemcpy(ho, dout, trials * 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < trials; ++t)
        if (memcmp(&ho[4 * t], &ho[4 * t + 2], 8) != 0) { if (bad < 5) printf("trial %d: old %a %a new %a %a\n", t, ho[4*t], ho[4*t+1], ho[4*t+2], ho[4*t+3]); ++bad; }
    printf("%d of %d trials differ\n", bad, trials);
    for (int nw = 0; nw < 2; ++nw) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        const int iters = 20000, blocks = 4096;
        if (nw) timing<1><<<blocks, 256>>>(tout, iters); else timing<0><<<blocks, 256>>>(tout, iters);
        (void)hipEventRecord(e0);
        if (nw) timing<1><<<blocks, 256>>>(tout, iters); else timing<0><<<blocks, 256>>>(tout, iters);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.3f ms -> %.1f ns (%.1f cycles at 2.4 GHz) of SIMD time per pair of sums\n", nw ? "merged" : "two butterflies", ms,
               ms * 1e6 / ((double)iters * 16), ms * 1e6 / ((double)iters * 16) * 2.4);
    }
    return bad != 0;
}
