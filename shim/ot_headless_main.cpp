// ot_headless_main.cpp -- headless driver for the reference's keypoint object tracker on the thin layer
// (samples/object_tracker_nvxcu/object_tracker_keypoint_nvxcu.cpp, nvxcu::ObjectTracker), following the frame loop of
// samples/object_tracker_nvxcu/main_object_tracker_nvxcu.cpp:560-735 without window, video decoder or mouse: frames come from a
// raw U8 file, the object's rectangle is given on the command line, per-frame status and location go to stdout.
//
// The tracker's sources are compiled where they lie with the one edit INTEGRATION.md section 2 describes, the mechanical cuda* -> hip*
// rename of the application's own runtime calls (oracle/Makefile.ref does it with sed into a temporary directory): every pyramid,
// FAST / Harris track and forward-backward PyrLK call they make goes through libvisualslam_amd's nvxcu entry points.
//
//   ot_headless frames.u8 W H N x0 y0 x1 y1 [key=value ...]
//     keys: pyr_levels lk_num_iters lk_win_size max_corners detector_cell_size detector(fast|harris) fast_type fast_threshold
//           harris_k harris_threshold                          (fields of nvxcu::KeypointObjectTrackerParams)
//   stdout: one line per frame after the first:  frame k status S x0 y0 x1 y1     (S: 0 tracked, 1 temporarily lost, 2 lost)
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include <NVX/nvxcu.h>

#include "object_tracker_nvxcu.hpp"   // the reference's own header

#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 3; } } while (0)

int main(int argc, char **argv)
{
    if (argc < 9) { std::cerr << "usage: ot_headless frames.u8 W H N x0 y0 x1 y1 [key=value ...]\n"; return 2; }
    const std::string path = argv[1];
    const uint32_t W = (uint32_t)atoi(argv[2]), H = (uint32_t)atoi(argv[3]);
    const int N = atoi(argv[4]);
    nvxcu_rectangle_t rect = { (uint32_t)atoi(argv[5]), (uint32_t)atoi(argv[6]), (uint32_t)atoi(argv[7]), (uint32_t)atoi(argv[8]) };
    nvxcu::KeypointObjectTrackerParams params;
    for (int i = 9; i < argc; ++i) {
        const std::string kv = argv[i];
        const size_t eq = kv.find('=');
        const std::string k = kv.substr(0, eq), v = eq == std::string::npos ? "" : kv.substr(eq + 1);
        if (k == "pyr_levels") params.pyr_levels = (uint32_t)atoi(v.c_str());
        else if (k == "lk_num_iters") params.lk_num_iters = (uint32_t)atoi(v.c_str());
        else if (k == "lk_win_size") params.lk_win_size = (uint32_t)atoi(v.c_str());
        else if (k == "max_corners") params.max_corners = (uint32_t)atoi(v.c_str());
        else if (k == "detector_cell_size") params.detector_cell_size = (uint32_t)atoi(v.c_str());
        else if (k == "detector") params.use_fast_detector = (v != "harris");
        else if (k == "fast_type") params.fast_type = (uint32_t)atoi(v.c_str());
        else if (k == "fast_threshold") params.fast_threshold = (uint32_t)atoi(v.c_str());
        else if (k == "harris_k") params.harris_k = (float)atof(v.c_str());
        else if (k == "harris_threshold") params.harris_threshold = (float)atof(v.c_str());
        else { std::cerr << "unknown key " << k << "\n"; return 2; }
    }
    std::ifstream in(path, std::ios::binary);
    if (!in) { std::cerr << "cannot open " << path << "\n"; return 2; }
    std::vector<unsigned char> host((size_t)W * H);

    // the frame as the sample's createImage makes it (main_object_tracker_nvxcu.cpp:57-90): a pitched device allocation
    nvxcu_pitch_linear_image_t frame = {};
    frame.base.image_type = NVXCU_PITCH_LINEAR_IMAGE;
    frame.base.format = NVXCU_DF_IMAGE_U8;
    frame.base.width = W;
    frame.base.height = H;
    void *dev = nullptr;
    size_t pitch = 0;
    HIP_OK(hipMallocPitch(&dev, &pitch, W, H));
    frame.planes[0].dev_ptr = dev;
    frame.planes[0].pitch_in_bytes = (uint32_t)pitch;

    std::unique_ptr<nvxcu::ObjectTracker> tracker(nvxcuCreateKeypointObjectTracker(params));
    if (!tracker) { std::cerr << "nvxcuCreateKeypointObjectTracker failed\n"; return 3; }
    nvxcu::ObjectTracker::TrackedObject *obj = nullptr;
    for (int k = 0; k < N; ++k) {
        in.read(reinterpret_cast<char *>(host.data()), (std::streamsize)host.size());
        if (!in) { std::cerr << "short read at frame " << k << "\n"; return 2; }
        HIP_OK(hipMemcpy2D(dev, pitch, host.data(), W, W, H, hipMemcpyHostToDevice));
        if (tracker->process(&frame) != NVXCU_SUCCESS) { std::cerr << "process failed at frame " << k << "\n"; return 3; }
        if (k == 0) {
            obj = tracker->addObject(rect);      // on the last processed frame (object_tracker_nvxcu.hpp: addObject)
            if (!obj) { std::cerr << "addObject failed\n"; return 3; }
            continue;
        }
        const nvxcu_rectangle_t r = obj->getLocation();
        printf("frame %d status %d %u %u %u %u\n", k, (int)obj->getStatus(), r.start_x, r.start_y, r.end_x, r.end_y);
    }
    tracker.reset();
    HIP_OK(hipFree(dev));
    return 0;
}
