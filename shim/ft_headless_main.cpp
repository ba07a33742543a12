// ft_headless_main.cpp -- headless driver for the UNMODIFIED reference wrapper
// demos/feature_tracker/feature_tracker.cpp (nvx::FeatureTracker), following the frame loop of
// demos/feature_tracker/main_feature_tracker.cpp:233-298,315-392 without window, video decoder or
// FPS limiter: frames come from a raw RGBX file, per-frame feature lists go to a binary file.
//
//   ft_headless frames.rgbx W H N out.bin [key=value ...] [mask=mask.u8]
//     keys: pyr_levels lk_num_iters lk_win_size array_capacity detector_cell_size detector(harris|fast)
//           harris_k harris_thresh fast_type fast_thresh            (names of feature_tracker_demo_config.ini)
//   out.bin: per tracked frame  u32 n_prev, n_prev x {f32 x,y}, u32 n_curr, n_curr x {f32 x,y}
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include <NVX/nvx.h>
#include <NVX/nvx_timer.hpp>
#include <OVX/UtilityOVX.hpp>

#include "feature_tracker.hpp"   // the reference's own header, included where it lies

static std::vector<nvx_point2f_t> readPoints(vx_array arr)
{
    vx_size n = 0;
    NVXIO_SAFE_CALL( vxQueryArray(arr, VX_ARRAY_ATTRIBUTE_NUMITEMS, &n, sizeof(n)) );
    std::vector<nvx_point2f_t> pts(n);
    if (n) NVXIO_SAFE_CALL( vxCopyArrayRange(arr, 0, n, sizeof(nvx_point2f_t), pts.data(), VX_READ_ONLY, VX_MEMORY_TYPE_HOST) );
    return pts;
}

static void upload(vx_image img, const unsigned char *src, vx_uint32 w, vx_uint32 h, vx_uint32 bpp)
{
    vx_imagepatch_addressing_t addr = {};
    addr.dim_x = w; addr.dim_y = h; addr.stride_x = (vx_int32)bpp; addr.stride_y = (vx_int32)(w * bpp);
    NVXIO_SAFE_CALL( vxCopyImagePatch(img, nullptr, 0, &addr, const_cast<unsigned char *>(src), VX_WRITE_ONLY, VX_MEMORY_TYPE_HOST) );
}

int main(int argc, char **argv)
{
    try {
        if (argc < 6) { std::cerr << "usage: ft_headless frames.rgbx W H N out.bin [key=value ...]\n"; return 2; }
        const std::string framesPath = argv[1], outPath = argv[5];
        const vx_uint32 W = (vx_uint32)atoi(argv[2]), H = (vx_uint32)atoi(argv[3]);
        const int N = atoi(argv[4]);
        nvx::FeatureTracker::Params params;
        std::string maskPath;
        bool verbose = false;
        for (int i = 6; i < argc; ++i) {
            const std::string kv = argv[i];
            const size_t eq = kv.find('=');
            const std::string k = kv.substr(0, eq), v = eq == std::string::npos ? "" : kv.substr(eq + 1);
            if (k == "pyr_levels") params.pyr_levels = (vx_uint32)atoi(v.c_str());
            else if (k == "lk_num_iters") params.lk_num_iters = (vx_uint32)atoi(v.c_str());
            else if (k == "lk_win_size") params.lk_win_size = (vx_uint32)atoi(v.c_str());
            else if (k == "array_capacity") params.array_capacity = (vx_uint32)atoi(v.c_str());
            else if (k == "detector_cell_size") params.detector_cell_size = (vx_uint32)atoi(v.c_str());
            else if (k == "detector") params.use_harris_detector = (v == "harris");
            else if (k == "harris_k") params.harris_k = (vx_float32)atof(v.c_str());
            else if (k == "harris_thresh") params.harris_thresh = (vx_float32)atof(v.c_str());
            else if (k == "fast_type") params.fast_type = (vx_uint32)atoi(v.c_str());
            else if (k == "fast_thresh") params.fast_thresh = (vx_uint32)atoi(v.c_str());
            else if (k == "mask") maskPath = v;
            else if (k == "verbose") verbose = true;
            else { std::cerr << "unknown key " << k << "\n"; return 2; }
        }

        std::ifstream in(framesPath, std::ios::binary);
        if (!in) { std::cerr << "cannot open " << framesPath << "\n"; return 3; }
        std::vector<unsigned char> frameBuf((size_t)W * H * 4);

        ovxio::ContextGuard context;
        vxDirective(context, VX_DIRECTIVE_ENABLE_PERFORMANCE);
        vxRegisterLogCallback(context, &ovxio::stdoutLogCallback, vx_false_e);

        // main_feature_tracker.cpp:233-246
        vx_image frameExemplar = vxCreateImage(context, W, H, VX_DF_IMAGE_RGBX);
        NVXIO_CHECK_REFERENCE(frameExemplar);
        vx_delay frame_delay = vxCreateDelay(context, (vx_reference)frameExemplar, 2);
        NVXIO_CHECK_REFERENCE(frame_delay);
        vxReleaseImage(&frameExemplar);
        vx_image frame = (vx_image)vxGetReferenceFromDelay(frame_delay, 0);

        vx_image mask = NULL;
        std::vector<unsigned char> maskBuf;
        if (!maskPath.empty()) {
            std::ifstream mi(maskPath, std::ios::binary);
            maskBuf.resize((size_t)W * H);
            mi.read(reinterpret_cast<char *>(maskBuf.data()), (std::streamsize)maskBuf.size());
            mask = vxCreateImage(context, W, H, VX_DF_IMAGE_U8);
            NVXIO_CHECK_REFERENCE(mask);
            upload(mask, maskBuf.data(), W, H, 1);
        }

        std::unique_ptr<nvx::FeatureTracker> tracker(nvx::FeatureTracker::create(context, params));
        std::ofstream out(outPath, std::ios::binary);

        in.read(reinterpret_cast<char *>(frameBuf.data()), (std::streamsize)frameBuf.size());
        upload(frame, frameBuf.data(), W, H, 4);
        tracker->init(frame, mask);                 // :296
        vxAgeDelay(frame_delay);                    // :298

        double proc_ms = 0;
        for (int t = 1; t < N; ++t) {
            in.read(reinterpret_cast<char *>(frameBuf.data()), (std::streamsize)frameBuf.size());
            if (!in) { std::cerr << "short frame file\n"; return 3; }
            upload(frame, frameBuf.data(), W, H, 4);
            nvx::Timer procTimer;
            procTimer.tic();
            tracker->track(frame, mask);            // :339
            proc_ms += procTimer.toc();
            if (verbose) tracker->printPerfs();     // :347
            const std::vector<nvx_point2f_t> prev = readPoints(tracker->getPrevFeatures());
            const std::vector<nvx_point2f_t> curr = readPoints(tracker->getCurrFeatures());
            const vx_uint32 np = (vx_uint32)prev.size(), nc = (vx_uint32)curr.size();
            out.write(reinterpret_cast<const char *>(&np), 4);
            out.write(reinterpret_cast<const char *>(prev.data()), (std::streamsize)(np * sizeof(nvx_point2f_t)));
            out.write(reinterpret_cast<const char *>(&nc), 4);
            out.write(reinterpret_cast<const char *>(curr.data()), (std::streamsize)(nc * sizeof(nvx_point2f_t)));
            vxAgeDelay(frame_delay);                // :392
        }
        std::cout << "ft_headless: " << (N - 1) << " frames tracked, " << proc_ms / (N > 1 ? N - 1 : 1) << " ms/frame (host wall)" << std::endl;
        tracker.reset();
        vxReleaseImage(&mask);
        vxReleaseDelay(&frame_delay);
    } catch (const std::exception &e) {
        std::cerr << "Error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
