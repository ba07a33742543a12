// pk_rate.hip -- wall-clock issue rate of scalar-f32 against packed-f32 VALU instructions on the whole GPU:
// every SIMD holds WPS resident waves that run 8 independent chains of one instruction kind.  Prints ns per
// wave-instruction per SIMD (x clock = cycles) so that v_fma_f32 / v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32
// and a DPP add can be compared directly.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(256) void rate(float *out, int iters)
{
    const float t = threadIdx.x;
    float a0 = t, a1 = t + 1, a2 = t + 2, a3 = t + 3, a4 = t + 4, a5 = t + 5, a6 = t + 6, a7 = t + 7;
    v2f p0 = {t, t + 8}, p1 = {t + 1, t + 9}, p2 = {t + 2, t + 10}, p3 = {t + 3, t + 11}, p4 = {t + 4, t + 12}, p5 = {t + 5, t + 13},
        p6 = {t + 6, t + 14}, p7 = {t + 7, t + 15};
    const float m = 1.0001f, c = 0.5f;
    const v2f pm = {1.0001f, 0.9999f}, pc = {0.5f, 0.25f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (KIND == 1) {
                asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                             "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm), "v"(pc));
            } else if (KIND == 2) {
                asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                             "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc));
            } else if (KIND == 3) {
                asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                             "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm));
            } else if (KIND == 4) {
                asm volatile("v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n"
                             "v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
            } else {
                // packed fma with different sources per half (op_sel free form): a = a*m + c with m,c from distinct pairs
                asm volatile("v_pk_fma_f32 %0, %1, %8, %0\n v_pk_fma_f32 %1, %2, %8, %1\n v_pk_fma_f32 %2, %3, %8, %2\n v_pk_fma_f32 %3, %4, %8, %3\n"
                             "v_pk_fma_f32 %4, %5, %8, %4\n v_pk_fma_f32 %5, %6, %8, %5\n v_pk_fma_f32 %6, %7, %8, %6\n v_pk_fma_f32 %7, %0, %8, %7\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y +
                                                 p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
}
template <int KIND>
double run(float *out, int blocks, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    rate<KIND><<<blocks, 256>>>(out, iters);
    hipEventRecord(e0);
    rate<KIND><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    if (hipDeviceSynchronize() != hipSuccess) { printf("fail\n"); exit(1); }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main()
{
    float *out; hipMalloc(&out, 8192 * 256 * 4);
    const int iters = 4000;
    const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_sub_f32", "v_pk_fma_f32 (cross operands)"};
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int blocks = 256 * wps * 4;   // 4 rounds of `wps` waves per SIMD (a block = one wave on each SIMD of a CU)
        double ms[6] = { run<0>(out, blocks, iters), run<1>(out, blocks, iters), run<2>(out, blocks, iters), run<3>(out, blocks, iters),
                         run<4>(out, blocks, iters), run<5>(out, blocks, iters) };
        for (int k = 0; k < 6; ++k) {
            const double instr_per_simd = (double)iters * 64 * wps * 4;
            printf("launch of %d blocks (%d resident waves/SIMD if evenly placed) %-32s %.3f ms  %.3f ns per wave-instr per SIMD (= %.2f cyc at 2.4 GHz)\n",
                   blocks, wps, names[k], ms[k], ms[k] * 1e6 / instr_per_simd, ms[k] * 1e6 / instr_per_simd * 2.4);
        }
    }
    return 0;
}
