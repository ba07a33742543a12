// calib_fetch.hip -- calibrates rocprofv3's FETCH_SIZE on gfx950 for the access patterns our kernels use
// (MI355X_MICROARCH.md, HBM section: widths other than 16 B/lane are uncalibrated).  Each kernel touches a
// known set of bytes once; the host prints how many unique bytes / 32-, 64- and 128-B sectors that is.
// Run: rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- ./calib_fetch
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <set>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void calib_b16(const uint4 *p, size_t n, uint32_t *out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345u) out[0] = acc;
}
__global__ void calib_b4(const uint32_t *p, size_t n, uint32_t *out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x12345u) out[0] = acc;
}
__global__ void calib_b1(const uint8_t *p, size_t n, uint32_t *out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x12345u) out[0] = acc;
}
// one wave per tile: rows x nd aligned dwords at (x0,y0), lane -> (row, dword) like pyrlk.hip's tile_issue
__global__ void calib_tile(const uint8_t *img, uint32_t pitch, const int2 *origins, int rows, int nd, uint32_t *out)
{
    const int2 o = origins[blockIdx.x];
    uint32_t acc = 0;
    for (int idx = threadIdx.x; idx < rows * nd; idx += 64) {
        const int r = idx / nd, d = idx - r * nd;
        acc += *reinterpret_cast<const uint32_t *>(img + (size_t)(o.y + r) * pitch + (o.x & ~3) + 4 * d);
    }
    if (acc == 0x12345u) out[0] = acc;
}

int main()
{
    const size_t BYTES = 64u << 20;
    uint8_t *buf; uint32_t *out;
    CK(hipMalloc(&buf, BYTES)); CK(hipMalloc(&out, 256));
    CK(hipMemset(buf, 1, BYTES)); CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; ++rep) {
        calib_b16<<<2048, 256>>>((const uint4 *)buf, BYTES / 16, out);
        calib_b4<<<2048, 256>>>((const uint32_t *)buf, BYTES / 4, out);
        calib_b1<<<2048, 256>>>(buf, BYTES / 4, out);   // first 16 MiB, one byte per lane
    }
    CK(hipDeviceSynchronize());
    printf("calib_b16 unique_bytes %zu\ncalib_b4 unique_bytes %zu\ncalib_b1 unique_bytes %zu\n", BYTES, BYTES, BYTES / 4);

    // LK-like tiles: 30 rows x 9 dwords, pitch 2048, 20000 tiles at pseudo-random origins in a 2048 x 32768 image
    const uint32_t pitch = 2048; const int H = (int)(BYTES / pitch), rows = 30, nd = 9, NT = 20000;
    std::vector<int2> org(NT);
    uint32_t s = 12345u;
    std::set<uint64_t> s32, s64, s128; size_t req = 0;
    for (int i = 0; i < NT; ++i) {
        s = s * 1664525u + 1013904223u; int x = (s >> 8) % (pitch - 4 * nd - 4);
        s = s * 1664525u + 1013904223u; int y = (s >> 8) % (H - rows);
        org[i] = make_int2(x, y);
        for (int r = 0; r < rows; ++r)
            for (int d = 0; d < nd; ++d) {
                uint64_t a = (uint64_t)(y + r) * pitch + (x & ~3) + 4 * d;
                s32.insert(a / 32); s64.insert(a / 64); s128.insert(a / 128); req += 4;
            }
    }
    int2 *dorg; CK(hipMalloc(&dorg, NT * sizeof(int2)));
    CK(hipMemcpy(dorg, org.data(), NT * sizeof(int2), hipMemcpyHostToDevice));
    for (int rep = 0; rep < 3; ++rep) {
        // evict: stream the whole buffer through so tiles are not L2-resident from the previous rep
        calib_b16<<<2048, 256>>>((const uint4 *)buf, BYTES / 16, out);
        calib_tile<<<NT, 64>>>(buf, pitch, dorg, rows, nd, out);
    }
    CK(hipDeviceSynchronize());
    printf("calib_tile requested_bytes %zu unique_32B %zu unique_64B %zu unique_128B %zu\n", req, s32.size() * 32, s64.size() * 64,
           s128.size() * 128);
    return 0;
}
