// corner_track.hip -- nvxHarrisTrack / nvxFastTrack: corner score, 3x3 non-maximum suppression,
// strongest corner per detector cell, merge with the tracked points, deterministic output order.
//
// Reference call sites replaced: nvxHarrisTrackNode / nvxcuHarrisTrack and nvxFastTrackNode /
// nvxcuFastTrack (demos/feature_tracker/feature_tracker.cpp:418-430,
// demos/feature_tracker_nvxcu/feature_tracker_nvxcu.cpp:307-317).  Semantics: oracle/vxoracle.c
// (cell selection/merge, Harris response and FAST strength functions).
//
// Two launches, no atomics on global memory, no host round trip:
//   k_harris_cells /     one workgroup OWNS a rectangle of whole detector cells (image read once +
//   k_fast_cells         halo), computes the score, the NMS and each cell's best (score, lowest
//                        raster index) as a 64-bit key, and stores one key per cell (0 = no corner).
//                        Harris: one wave per rectangle, lane = image column, rows streamed through
//                        register sliding windows, horizontal neighbours by DPP wave shifts.
//                        FAST: the same layout; the 16 circle pixels come from a register ring of the last
//                        seven rows and their DPP-shifted copies (fast_cells.hpp).
//   k_merge_compact      one workgroup marks cells that hold a surviving tracked point, then
//                        compacts survivors (input order) and new corners (cell raster order)
//                        with ballot/popcount prefix sums (merge_cells.hpp; in a graph with a FindHomography
//                        node on the same matches the pass is workgroup 0 of that node's launch, homography.hip).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "kernels.hpp"
#include "harris_cells.hpp"
#include "fast_cells.hpp"
#include "merge_cells.hpp"


namespace vsa {
namespace {

__global__ __launch_bounds__(64) void k_harris_cells(CornerArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    VSA_SHORT_KERNEL_PRIORITY();
    harris_cells_wave(a, blockIdx.x, gridDim.x, smem);
}


// ---- FAST: the same shape -- one wave per rectangle of cells, lane = image column (fast_cells.hpp) ----------------
template <int N>
__global__ __launch_bounds__(64) void k_fast_cells(CornerArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    VSA_SHORT_KERNEL_PRIORITY();
    fast_cells_wave<N>(a, blockIdx.x, gridDim.x, smem);
}

// ---- merge + compaction: merge_cells.hpp ------------------------------------------------------
// At most 48 registers, whatever the compiler would like (merge_cells.hpp: two of its waves per SIMD must fit in the 96 registers
// that four PyrLK waves leave; at 52 the 32-stream bench loses 17 %).  The attribute counts the unified register file in halves.
template <uint32_t KT, uint32_t KC>
__global__ __launch_bounds__(MG_THREADS) void k_merge_compact(MergeArgs a)
{
    VSA_SHORT_KERNEL_PRIORITY();
    merge_compact_wg<MG_THREADS, KT, KC>(a);
}

}  // namespace

size_t corner_track_scratch_bytes(uint32_t width, uint32_t height, uint32_t cell_size)
{
    if (cell_size == 0) return 0;
    const size_t ncx = (width + cell_size - 1) / cell_size, ncy = (height + cell_size - 1) / cell_size;
    return ncx * ncy * sizeof(uint64_t);
}

// the cell pass: one key per detector cell into p.scratch
hipError_t launch_corner_cells(const CornerTrackParams &p, hipStream_t stream)
{
    const uint32_t cs = p.cell_size;
    if (cs == 0u || p.image.width > 65535u || p.image.height > 65535u) return hipErrorInvalidValue;   // cell keys hold (y << 16) | x
    CornerArgs a{};
    uint32_t blocks = 0;
    size_t lds = 0;
    harris_cells_setup(p, a, blocks, lds);
    if (p.detector == kHarris) {
        hipLaunchKernelGGL(k_harris_cells, dim3(blocks), dim3(64), lds, stream, a);
    } else {
        fast_cells_setup(p, a, blocks, lds, FS_BAND);
        switch (p.fast_n) {
        case 9: hipLaunchKernelGGL(k_fast_cells<9>, dim3(blocks), dim3(64), lds, stream, a); break;
        case 10: hipLaunchKernelGGL(k_fast_cells<10>, dim3(blocks), dim3(64), lds, stream, a); break;
        case 11: hipLaunchKernelGGL(k_fast_cells<11>, dim3(blocks), dim3(64), lds, stream, a); break;
        case 12: hipLaunchKernelGGL(k_fast_cells<12>, dim3(blocks), dim3(64), lds, stream, a); break;
        default: return hipErrorInvalidValue;
        }
    }
    return hipGetLastError();
}

// the merge pass: surviving tracked points (input order), then the cells' new corners (cell raster order)
hipError_t launch_corner_merge(const CornerTrackParams &p, hipStream_t stream)
{
    MergeArgs m{};
    size_t lds = 0;
    if (!merge_setup(p, m, lds)) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_merge_compact<MG_KT, MG_KC>), dim3(1), dim3(MG_THREADS), lds, stream, m);
    return hipGetLastError();
}

hipError_t launch_corner_track(const CornerTrackParams &p, hipStream_t stream)
{
    const hipError_t e = launch_corner_cells(p, stream);
    return e != hipSuccess ? e : launch_corner_merge(p, stream);
}

}  // namespace vsa

#ifdef VSA_MG_STAMPS
extern "C" int visualslam_amd_debug_mg_stamps(uint64_t *host, size_t n_u64)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(vsa::g_mg_stamps), n_u64 * sizeof(uint64_t));
}
#endif

