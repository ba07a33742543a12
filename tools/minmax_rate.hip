// minmax_rate.hip -- wall-clock issue rate of the minimum / maximum instruction forms the FAST cell pass could be built from
// (fast_cells.hpp), same method as valu_kinds.hip: all SIMDs busy with 4 resident waves each, 8 independent chains per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(OP, TAIL) \
    asm volatile(OP " %0, %0" TAIL "\n" OP " %1, %1" TAIL "\n" OP " %2, %2" TAIL "\n" OP " %3, %3" TAIL "\n" \
                 OP " %4, %4" TAIL "\n" OP " %5, %5" TAIL "\n" OP " %6, %6" TAIL "\n" OP " %7, %7" TAIL "\n" \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c))
template <int KIND>
__global__ __launch_bounds__(256) void rate(unsigned *out, int iters)
{
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const unsigned m = 0x3f800347u, c = 3u;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) REP8("v_min_i32", ", %8");
            else if (KIND == 1) REP8("v_max_u32", ", %8");
            else if (KIND == 2) REP8("v_min3_i32", ", %8, %9");
            else if (KIND == 3) REP8("v_min_f32", ", %8");
            else if (KIND == 4) REP8("v_max_f32", ", %8");
            else if (KIND == 5) REP8("v_min3_f32", ", %8, %9");
            else if (KIND == 6) REP8("v_pk_min_i16", ", %8");
            else if (KIND == 7) REP8("v_pk_max_u16", ", %8");
            else if (KIND == 8) REP8("v_pk_min_f16", ", %8");
            else if (KIND == 9) REP8("v_pk_max_f16", ", %8");
            else if (KIND == 10) REP8("v_min_u16", ", %8");
            else if (KIND == 11) REP8("v_min_f16", ", %8");
            else if (KIND == 12) REP8("v_alignbit_b32", ", %8, %9");
            else if (KIND == 13) REP8("v_pk_add_u16", ", %8");
            else if (KIND == 14) REP8("v_pk_sub_i16", ", %8");
            else if (KIND == 15) REP8("v_med3_i32", ", %8, %9");
            else if (KIND == 16) REP8("v_pk_add_f16", ", %8");
            else if (KIND == 17) REP8("v_pk_fma_f16", ", %8, %9");
            else if (KIND == 18) REP8("v_add_f32", ", %8");
            else if (KIND == 19) REP8("v_sub_u32", ", %8");
            else if (KIND == 20) REP8("v_cmp_gt_i32 vcc", ", %8");
            else REP8("v_mov_b32", "");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int KIND>
void run(unsigned *out, const char *name)
{
    const int iters = 2000, wps = 4, blocks = 256 * wps * 4;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    rate<KIND><<<blocks, 256>>>(out, iters);
    (void)hipEventRecord(e0);
    rate<KIND><<<blocks, 256>>>(out, iters);
    (void)hipEventRecord(e1);
    if (hipDeviceSynchronize() != hipSuccess) { printf("fail %s\n", name); exit(1); }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / ((double)iters * 64 * wps * 4);
    printf("%-26s %.3f ns per wave-instr per SIMD\n", name, ns);
}
int main()
{
    unsigned *out; (void)hipMalloc(&out, 4096 * 256 * 4);
    run<18>(out, "v_add_f32 (yardstick)"); run<19>(out, "v_sub_u32");
    run<0>(out, "v_min_i32"); run<1>(out, "v_max_u32"); run<2>(out, "v_min3_i32"); run<15>(out, "v_med3_i32");
    run<3>(out, "v_min_f32"); run<4>(out, "v_max_f32"); run<5>(out, "v_min3_f32");
    run<6>(out, "v_pk_min_i16"); run<7>(out, "v_pk_max_u16"); run<13>(out, "v_pk_add_u16"); run<14>(out, "v_pk_sub_i16");
    run<8>(out, "v_pk_min_f16"); run<9>(out, "v_pk_max_f16"); run<16>(out, "v_pk_add_f16"); run<17>(out, "v_pk_fma_f16");
    run<10>(out, "v_min_u16"); run<11>(out, "v_min_f16"); run<12>(out, "v_alignbit_b32");
    return 0;
}
