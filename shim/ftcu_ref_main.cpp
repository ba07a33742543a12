// ftcu_ref_main.cpp -- headless frame loop around the REFERENCE's own thin-layer tracker class, nvxcu::FeatureTracker of
// demos/feature_tracker_nvxcu/feature_tracker_nvxcu.{cpp,hpp}, which oracle/Makefile.ref compiles where it lies (runtime calls
// renamed cuda* -> hip*) against include/NVX/nvxcu.h.  This file stands in for the demo's
// main_feature_tracker_nvxcu.cpp, whose frame source and renderer are GStreamer / OpenGL code: RGBX frames from hipMallocPitch
// (:128-143), init() on the first frame (:287), track() per frame (:324), getPrevFeatures() / getCurrFeatures() after each
// (:341-342) -- written to a file instead of drawn.  Test infrastructure (tests/test_gpu_thin_layer.py); same command line and
// output format as visualslam_amd/lib/ftcu_hip:
//   ftcu_headless frames.rgbx W H N out.bin [key=value ...] [mask=mask.u8]      (keys of feature_tracker_demo_config.ini)
//   out.bin: per tracked frame  u32 n_prev, n_prev x {f32 x,y}, u32 n_curr, n_curr x {f32 x,y}
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "feature_tracker_nvxcu.hpp"

#define HIP_OK(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { std::cerr << #call << ": " << hipGetErrorString(e__) << "\n"; std::exit(1); } } while (0)

static nvxcu_pitch_linear_image_t make_image(uint32_t w, uint32_t h, nvxcu_df_image_e fmt, const unsigned char *src)
{
    nvxcu_pitch_linear_image_t im = {};
    size_t pitch = 0;
    const size_t row = (size_t)w * (fmt == NVXCU_DF_IMAGE_RGBX ? 4 : 1);
    HIP_OK(hipMallocPitch(&im.planes[0].dev_ptr, &pitch, row, h));
    im.base.image_type = NVXCU_PITCH_LINEAR_IMAGE;
    im.base.format = fmt;
    im.base.width = w;
    im.base.height = h;
    im.planes[0].pitch_in_bytes = (uint32_t)pitch;
    if (src) HIP_OK(hipMemcpy2D(im.planes[0].dev_ptr, pitch, src, row, row, h, hipMemcpyHostToDevice));
    return im;
}

static void dump(std::ofstream &out, const nvxcu_array_t *arr)
{
    const nvxcu_plain_array_t *a = reinterpret_cast<const nvxcu_plain_array_t *>(arr);   // main_feature_tracker_nvxcu.cpp:344-346
    uint32_t n = 0;
    HIP_OK(hipMemcpy(&n, a->num_items_dev_ptr, sizeof(n), hipMemcpyDeviceToHost));
    if (n > a->base.capacity) { std::cerr << "item count beyond the capacity\n"; std::exit(1); }
    std::vector<nvxcu_point2f_t> v(n);
    if (n) HIP_OK(hipMemcpy(v.data(), a->dev_ptr, n * sizeof(nvxcu_point2f_t), hipMemcpyDeviceToHost));
    out.write(reinterpret_cast<const char *>(&n), 4);
    out.write(reinterpret_cast<const char *>(v.data()), (std::streamsize)(n * sizeof(nvxcu_point2f_t)));
}

int main(int argc, char **argv)
{
    if (argc < 6) { std::cerr << "usage: ftcu_headless frames.rgbx W H N out.bin [key=value ...]\n"; return 2; }
    const uint32_t W = (uint32_t)atoi(argv[2]), H = (uint32_t)atoi(argv[3]);
    const int N = atoi(argv[4]);
    nvxcu::FeatureTracker::Params params;      // the reference's defaults (feature_tracker_nvxcu.cpp:487-506)
    std::string mask_path;
    for (int i = 6; i < argc; ++i) {
        const std::string kv = argv[i];
        const size_t eq = kv.find('=');
        const std::string k = kv.substr(0, eq), v = eq == std::string::npos ? "" : kv.substr(eq + 1);
        if (k == "pyr_levels") params.pyr_levels = (uint32_t)atoi(v.c_str());
        else if (k == "lk_num_iters") params.lk_num_iters = (uint32_t)atoi(v.c_str());
        else if (k == "lk_win_size") params.lk_win_size = (uint32_t)atoi(v.c_str());
        else if (k == "array_capacity") params.array_capacity = (uint32_t)atoi(v.c_str());
        else if (k == "detector_cell_size") params.detector_cell_size = (uint32_t)atoi(v.c_str());
        else if (k == "detector") params.use_harris_detector = (v == "harris");
        else if (k == "harris_k") params.harris_k = (float)atof(v.c_str());
        else if (k == "harris_thresh") params.harris_thresh = (float)atof(v.c_str());
        else if (k == "fast_type") params.fast_type = (uint32_t)atoi(v.c_str());
        else if (k == "fast_thresh") params.fast_thresh = (uint32_t)atoi(v.c_str());
        else if (k == "mask") mask_path = v;
        else { std::cerr << "unknown key " << k << "\n"; return 2; }
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) { std::cerr << "no GPU\n"; return 4; }
    std::ifstream in(argv[1], std::ios::binary);
    if (!in) { std::cerr << "cannot open " << argv[1] << "\n"; return 3; }
    std::vector<unsigned char> host((size_t)W * H * 4);
    nvxcu_pitch_linear_image_t frame = make_image(W, H, NVXCU_DF_IMAGE_RGBX, nullptr);
    nvxcu_pitch_linear_image_t mask_img = {};
    const nvxcu_image_t *mask = nullptr;
    if (!mask_path.empty()) {
        std::ifstream mi(mask_path, std::ios::binary);
        std::vector<unsigned char> m((size_t)W * H);
        mi.read(reinterpret_cast<char *>(m.data()), (std::streamsize)m.size());
        if (!mi) { std::cerr << "short mask file\n"; return 3; }
        mask_img = make_image(W, H, NVXCU_DF_IMAGE_U8, m.data());
        mask = &mask_img.base;
    }
    std::ofstream out(argv[5], std::ios::binary);
    try {
        std::unique_ptr<nvxcu::FeatureTracker> tracker(nvxcu::FeatureTracker::create(params));
        for (int t = 0; t < N; ++t) {
            in.read(reinterpret_cast<char *>(host.data()), (std::streamsize)host.size());
            if (!in) { std::cerr << "short frame file\n"; return 3; }
            HIP_OK(hipMemcpy2D(frame.planes[0].dev_ptr, frame.planes[0].pitch_in_bytes, host.data(), (size_t)W * 4, (size_t)W * 4, H, hipMemcpyHostToDevice));
            if (t == 0) { tracker->init(&frame.base, mask); continue; }
            tracker->track(&frame.base, mask);
            dump(out, tracker->getPrevFeatures());
            dump(out, tracker->getCurrFeatures());
        }
        std::cout << "ftcu_headless: " << (N - 1) << " frames tracked by the reference's nvxcu::FeatureTracker" << std::endl;
    } catch (const std::exception &e) {
        std::cerr << "Error: " << e.what() << std::endl;
        return 1;
    }
    (void)hipFree(frame.planes[0].dev_ptr);
    if (mask) (void)hipFree(mask_img.planes[0].dev_ptr);
    return 0;
}
