// dpp_probe.hip -- prints what the DPP controls used by the kernels do on this GPU (lane i holds value i+100)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL2 = 0> struct unused_ {};
#define DPP(old, v, CTRL, RM, BC) __builtin_amdgcn_update_dpp(old, v, CTRL, RM, 0xf, BC)
__global__ void probe(int *out)
{
    const int l = threadIdx.x, v = l + 100;
    out[0 * 64 + l] = DPP(0, v, 0x138, 0xf, true);    // wave_shr:1
    out[1 * 64 + l] = DPP(0, v, 0x130, 0xf, true);    // wave_shl:1
    out[2 * 64 + l] = DPP(-1, v, 0x142, 0xa, false);   // row_bcast:15 rows 1,3
    out[3 * 64 + l] = DPP(-1, v, 0x143, 0xc, false);   // row_bcast:31 rows 2,3
    out[4 * 64 + l] = DPP(0, v, 0x111, 0xf, true);    // row_shr:1
    out[5 * 64 + l] = DPP(0, v, 0x101, 0xf, true);    // row_shl:1
}
int main()
{
    int *d, h[6 * 64];
    hipMalloc(&d, sizeof(h));
    probe<<<1, 64>>>(d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { printf("fail\n"); return 1; }
    const char *nm[6] = { "wave_shr1", "wave_shl1", "row_bcast15(rm a)", "row_bcast31(rm c)", "row_shr1", "row_shl1" };
    for (int k = 0; k < 6; ++k) {
        printf("%s:", nm[k]);
        for (int l = 0; l < 64; ++l) printf(" %d", h[k * 64 + l]);
        printf("\n");
    }
    return 0;
}
