// ftcu_hip_main.cpp -- a C++ caller of the thin layer (include/NVX/nvxcu.h), written the way a maintainer of
// demos/feature_tracker_nvxcu/feature_tracker_nvxcu.cpp would have it after swapping the CUDA runtime for HIP:
// pitch-linear images from hipMallocPitch, plain arrays with a device-resident u32 item count, scratch sized by the
// *_GetBufSize queries, nvxcu_stream_exec_target_t filled from hipGetDeviceProperties, every primitive enqueued on one
// caller-owned stream and ONE hipStreamSynchronize per frame (feature_tracker_nvxcu.cpp:102-179 allocation helpers,
// :191-222 execution target and border, :255-321 track(), :387-450 data objects and scratch, :457-484 first frame).
// It proves the L1 ABI from C++ (struct layouts as the compiler lays them out, hipDeviceProp_t embedded in the target),
// not only through the ctypes mirrors the Python tests use.
//
//   ftcu_hip frames.rgbx W H N out.bin [key=value ...] [mask=mask.u8] [noalloc=1]
//     keys: pyr_levels lk_num_iters lk_win_size array_capacity detector_cell_size detector(harris|fast)
//           harris_k harris_thresh fast_type fast_thresh            (names of feature_tracker_demo_config.ini)
//     noalloc=1: the Harris detector goes through nvxcuHarrisTrack_NoAlloc (no tmp_buf)
//   out.bin (same as ft_headless): per tracked frame  u32 n_prev, n_prev x {f32 x,y}, u32 n_curr, n_curr x {f32 x,y}
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include <hip/hip_runtime_api.h>

#include <NVX/nvxcu.h>

#define HIP_OK(call) do { hipError_t err__ = (call); if (err__ != hipSuccess) throw std::runtime_error(std::string(#call) + ": " + hipGetErrorString(err__)); } while (0)
#define NVXCU_OK(call) do { nvxcu_error_status_e st__ = (call); if (st__ != NVXCU_SUCCESS) throw std::runtime_error(std::string(#call) + " -> " + std::to_string((int)st__)); } while (0)

namespace {

struct Settings {
    uint32_t pyr_levels = 6, lk_num_iters = 5, lk_win_size = 10, array_capacity = 2000, detector_cell_size = 18;
    bool harris = true, noalloc = false;
    float harris_k = 0.04f, harris_thresh = 100.0f;
    uint32_t fast_type = 9, fast_thresh = 25;
};

nvxcu_pitch_linear_image_t make_image(uint32_t w, uint32_t h, nvxcu_df_image_e fmt)
{
    nvxcu_pitch_linear_image_t im{};
    size_t pitch = 0;
    HIP_OK(hipMallocPitch(&im.planes[0].dev_ptr, &pitch, (size_t)w * (fmt == NVXCU_DF_IMAGE_RGBX ? 4 : 1), h));
    im.base.image_type = NVXCU_PITCH_LINEAR_IMAGE;
    im.base.format = fmt;
    im.base.width = w;
    im.base.height = h;
    im.planes[0].pitch_in_bytes = (uint32_t)pitch;
    return im;
}

struct Pyramid {
    std::vector<nvxcu_pitch_linear_image_t> levels;
    nvxcu_pitch_linear_pyramid_t desc{};
    void create(uint32_t w, uint32_t h, uint32_t n)
    {
        float scale = 1.0f;
        uint32_t cw = w, ch = h;
        for (uint32_t i = 0; i < n; ++i) {          // level i is ceil(base * 0.5^i) in float arithmetic (:164-176)
            levels.push_back(make_image(cw, ch, NVXCU_DF_IMAGE_U8));
            scale *= NVXCU_SCALE_PYRAMID_HALF;
            cw = (uint32_t)ceilf((float)w * scale);
            ch = (uint32_t)ceilf((float)h * scale);
        }
        desc.base.pyramid_type = NVXCU_PITCH_LINEAR_PYRAMID;
        desc.base.num_levels = n;
        desc.base.scale = NVXCU_SCALE_PYRAMID_HALF;
        desc.levels = levels.data();
    }
    void destroy() { for (auto &l : levels) (void)hipFree(l.planes[0].dev_ptr); levels.clear(); }
};

nvxcu_plain_array_t make_points(uint32_t capacity)
{
    nvxcu_plain_array_t a{};
    HIP_OK(hipMalloc(&a.dev_ptr, (size_t)capacity * sizeof(nvxcu_point2f_t)));
    HIP_OK(hipMalloc(reinterpret_cast<void **>(&a.num_items_dev_ptr), sizeof(uint32_t)));
    HIP_OK(hipMemset(a.num_items_dev_ptr, 0, sizeof(uint32_t)));
    a.base.array_type = NVXCU_PLAIN_ARRAY;
    a.base.item_type = NVXCU_TYPE_POINT2F;
    a.base.capacity = capacity;
    return a;
}

nvxcu_tmp_buf_t make_scratch(nvxcu_tmp_buf_size_t size)
{
    nvxcu_tmp_buf_t b{ nullptr, nullptr };
    if (size.dev_buf_size) HIP_OK(hipMalloc(&b.dev_ptr, size.dev_buf_size));
    if (size.host_buf_size) HIP_OK(hipHostMalloc(&b.host_ptr, size.host_buf_size, hipHostMallocDefault));
    return b;
}

class ThinTracker {
public:
    ThinTracker(const Settings &s, uint32_t w, uint32_t h) : s_(s)
    {
        int dev = -1;
        HIP_OK(hipGetDevice(&dev));
        target_.base.exec_target_type = NVXCU_STREAM_EXEC_TARGET;
        HIP_OK(hipGetDeviceProperties(&target_.dev_prop, dev));
        HIP_OK(hipStreamCreate(&target_.stream));
        border_.mode = NVXCU_BORDER_MODE_UNDEFINED;
        border_.constant_value.U8 = 0;
        prev_pyr_.create(w, h, s.pyr_levels);
        curr_pyr_.create(w, h, s.pyr_levels);
        prev_pts_ = make_points(s.array_capacity);
        curr_pts_ = make_points(s.array_capacity);
        tracked_ = make_points(s.array_capacity);
        HIP_OK(hipMalloc(reinterpret_cast<void **>(&num_corners_dev_), sizeof(size_t)));
        pyr_buf_ = make_scratch(nvxcuGaussianPyramid_GetBufSize(w, h, s.pyr_levels, NVXCU_SCALE_PYRAMID_HALF, &border_, &target_.dev_prop));
        lk_buf_ = make_scratch(nvxcuOpticalFlowPyrLK_GetBufSize(w, h, s.pyr_levels, NVXCU_SCALE_PYRAMID_HALF, NVXCU_TYPE_POINT2F,
                                                                s.array_capacity, s.lk_win_size, 0, &border_, &target_.dev_prop));
        det_buf_ = make_scratch(s.harris ? nvxcuHarrisTrack_GetBufSize(w, h, NVXCU_DF_IMAGE_U8, NVXCU_TYPE_POINT2F, s.array_capacity,
                                                                       s.detector_cell_size, &target_.dev_prop)
                                         : nvxcuFastTrack_GetBufSize(w, h, NVXCU_DF_IMAGE_U8, NVXCU_TYPE_POINT2F, s.array_capacity,
                                                                     s.detector_cell_size, &target_.dev_prop));
    }
    ~ThinTracker()
    {
        (void)hipStreamSynchronize(target_.stream);
        for (nvxcu_tmp_buf_t *b : { &pyr_buf_, &lk_buf_, &det_buf_ }) { (void)hipFree(b->dev_ptr); if (b->host_ptr) (void)hipHostFree(b->host_ptr); }
        for (nvxcu_plain_array_t *a : { &prev_pts_, &curr_pts_, &tracked_ }) { (void)hipFree(a->dev_ptr); (void)hipFree(a->num_items_dev_ptr); }
        (void)hipFree(num_corners_dev_);
        prev_pyr_.destroy();
        curr_pyr_.destroy();
        (void)hipStreamDestroy(target_.stream);
    }

    void first_frame(const nvxcu_image_t *frame, const nvxcu_image_t *mask)
    {
        front_end(frame);
        detect(mask, nullptr);
        HIP_OK(hipStreamSynchronize(target_.stream));
    }

    void track(const nvxcu_image_t *frame, const nvxcu_image_t *mask)
    {
        std::swap(curr_pts_, prev_pts_);
        std::swap(curr_pyr_, prev_pyr_);
        front_end(frame);
        // the output array of PyrLK holds the initial estimates on entry: count + payload of the previous points (:288-295)
        HIP_OK(hipMemcpyAsync(tracked_.num_items_dev_ptr, prev_pts_.num_items_dev_ptr, sizeof(uint32_t), hipMemcpyDeviceToDevice, target_.stream));
        HIP_OK(hipMemcpyAsync(tracked_.dev_ptr, prev_pts_.dev_ptr, (size_t)prev_pts_.base.capacity * sizeof(nvxcu_point2f_t),
                              hipMemcpyDeviceToDevice, target_.stream));
        NVXCU_OK(nvxcuOpticalFlowPyrLK(&prev_pyr_.desc.base, &curr_pyr_.desc.base, &prev_pts_.base, &tracked_.base, NVXCU_TERM_CRITERIA_BOTH,
                                       0.01f, s_.lk_num_iters, s_.lk_win_size, 0, &lk_buf_, &border_, &target_.base));
        detect(mask, &tracked_.base);
        HIP_OK(hipStreamSynchronize(target_.stream));    // the one synchronisation of the frame (:320)
    }

    std::vector<nvxcu_point2f_t> download(const nvxcu_plain_array_t &a) const
    {
        uint32_t n = 0;
        HIP_OK(hipMemcpy(&n, a.num_items_dev_ptr, sizeof(n), hipMemcpyDeviceToHost));
        if (n > a.base.capacity) throw std::runtime_error("item count beyond the capacity");
        std::vector<nvxcu_point2f_t> v(n);
        if (n) HIP_OK(hipMemcpy(v.data(), a.dev_ptr, n * sizeof(nvxcu_point2f_t), hipMemcpyDeviceToHost));
        return v;
    }
    const nvxcu_plain_array_t &prev_points() const { return prev_pts_; }
    const nvxcu_plain_array_t &tracked_points() const { return tracked_; }
    size_t corners_before_truncation() const
    {
        size_t n = 0;
        HIP_OK(hipMemcpy(&n, num_corners_dev_, sizeof(n), hipMemcpyDeviceToHost));
        return n;
    }

private:
    void front_end(const nvxcu_image_t *frame)
    {
        // the colour-converted frame IS level 0 of the current pyramid (:277-285)
        NVXCU_OK(nvxcuColorConvert(frame, &curr_pyr_.levels[0].base, NVXCU_COLOR_SPACE_DEFAULT, NVXCU_CHANNEL_RANGE_FULL, &target_.base));
        NVXCU_OK(nvxcuGaussianPyramid(&curr_pyr_.desc.base, &pyr_buf_, &border_, &target_.base));
    }
    void detect(const nvxcu_image_t *mask, const nvxcu_array_t *tracked)
    {
        const nvxcu_image_t *gray = &curr_pyr_.levels[0].base;
        if (s_.harris && s_.noalloc)
            NVXCU_OK(nvxcuHarrisTrack_NoAlloc(gray, &curr_pts_.base, mask, tracked, s_.harris_k, s_.harris_thresh, s_.detector_cell_size,
                                              num_corners_dev_, &target_.base));
        else if (s_.harris)
            NVXCU_OK(nvxcuHarrisTrack(gray, &curr_pts_.base, mask, tracked, s_.harris_k, s_.harris_thresh, s_.detector_cell_size,
                                      num_corners_dev_, &det_buf_, &target_.base));
        else
            NVXCU_OK(nvxcuFastTrack(gray, &curr_pts_.base, mask, tracked, s_.fast_type, s_.fast_thresh, s_.detector_cell_size,
                                    num_corners_dev_, &det_buf_, &target_.base));
    }

    Settings s_;
    nvxcu_stream_exec_target_t target_{};
    nvxcu_border_t border_{};
    Pyramid prev_pyr_, curr_pyr_;
    nvxcu_plain_array_t prev_pts_{}, curr_pts_{}, tracked_{};
    nvxcu_tmp_buf_t pyr_buf_{}, lk_buf_{}, det_buf_{};
    size_t *num_corners_dev_ = nullptr;
};

void upload(const nvxcu_pitch_linear_image_t &im, const unsigned char *src, size_t row_bytes)
{
    HIP_OK(hipMemcpy2D(im.planes[0].dev_ptr, im.planes[0].pitch_in_bytes, src, row_bytes, row_bytes, im.base.height, hipMemcpyHostToDevice));
}

}  // namespace

int main(int argc, char **argv)
{
    try {
        if (argc < 6) { std::cerr << "usage: ftcu_hip frames.rgbx W H N out.bin [key=value ...]\n"; return 2; }
        const std::string frames_path = argv[1], out_path = argv[5];
        const uint32_t W = (uint32_t)atoi(argv[2]), H = (uint32_t)atoi(argv[3]);
        const int N = atoi(argv[4]);
        Settings s;
        std::string mask_path;
        for (int i = 6; i < argc; ++i) {
            const std::string kv = argv[i];
            const size_t eq = kv.find('=');
            const std::string k = kv.substr(0, eq), v = eq == std::string::npos ? "" : kv.substr(eq + 1);
            if (k == "pyr_levels") s.pyr_levels = (uint32_t)atoi(v.c_str());
            else if (k == "lk_num_iters") s.lk_num_iters = (uint32_t)atoi(v.c_str());
            else if (k == "lk_win_size") s.lk_win_size = (uint32_t)atoi(v.c_str());
            else if (k == "array_capacity") s.array_capacity = (uint32_t)atoi(v.c_str());
            else if (k == "detector_cell_size") s.detector_cell_size = (uint32_t)atoi(v.c_str());
            else if (k == "detector") s.harris = (v == "harris");
            else if (k == "harris_k") s.harris_k = (float)atof(v.c_str());
            else if (k == "harris_thresh") s.harris_thresh = (float)atof(v.c_str());
            else if (k == "fast_type") s.fast_type = (uint32_t)atoi(v.c_str());
            else if (k == "fast_thresh") s.fast_thresh = (uint32_t)atoi(v.c_str());
            else if (k == "mask") mask_path = v;
            else if (k == "noalloc") s.noalloc = atoi(v.c_str()) != 0;
            else { std::cerr << "unknown key " << k << "\n"; return 2; }
        }
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count == 0) { std::cerr << "no GPU\n"; return 4; }

        std::ifstream in(frames_path, std::ios::binary);
        if (!in) { std::cerr << "cannot open " << frames_path << "\n"; return 3; }
        std::vector<unsigned char> host((size_t)W * H * 4);
        nvxcu_pitch_linear_image_t frame = make_image(W, H, NVXCU_DF_IMAGE_RGBX);     // main_feature_tracker_nvxcu.cpp:128-143
        nvxcu_pitch_linear_image_t mask_img{};
        const nvxcu_image_t *mask = nullptr;
        if (!mask_path.empty()) {
            std::ifstream mi(mask_path, std::ios::binary);
            std::vector<unsigned char> m((size_t)W * H);
            mi.read(reinterpret_cast<char *>(m.data()), (std::streamsize)m.size());
            if (!mi) { std::cerr << "short mask file\n"; return 3; }
            mask_img = make_image(W, H, NVXCU_DF_IMAGE_U8);
            upload(mask_img, m.data(), W);
            mask = &mask_img.base;
        }
        std::ofstream out(out_path, std::ios::binary);
        {
            ThinTracker tracker(s, W, H);
            in.read(reinterpret_cast<char *>(host.data()), (std::streamsize)host.size());
            if (!in) { std::cerr << "short frame file\n"; return 3; }
            upload(frame, host.data(), (size_t)W * 4);
            tracker.first_frame(&frame.base, mask);
            size_t total = tracker.corners_before_truncation();
            for (int t = 1; t < N; ++t) {
                in.read(reinterpret_cast<char *>(host.data()), (std::streamsize)host.size());
                if (!in) { std::cerr << "short frame file\n"; return 3; }
                upload(frame, host.data(), (size_t)W * 4);
                tracker.track(&frame.base, mask);
                const auto prev = tracker.download(tracker.prev_points()), curr = tracker.download(tracker.tracked_points());
                const uint32_t np = (uint32_t)prev.size(), nc = (uint32_t)curr.size();
                out.write(reinterpret_cast<const char *>(&np), 4);
                out.write(reinterpret_cast<const char *>(prev.data()), (std::streamsize)(np * sizeof(nvxcu_point2f_t)));
                out.write(reinterpret_cast<const char *>(&nc), 4);
                out.write(reinterpret_cast<const char *>(curr.data()), (std::streamsize)(nc * sizeof(nvxcu_point2f_t)));
                total = tracker.corners_before_truncation();
            }
            std::cout << "ftcu_hip: " << (N - 1) << " frames tracked, " << total << " corners before truncation on the last frame, sizeof(target) = "
                      << sizeof(nvxcu_stream_exec_target_t) << std::endl;
        }
        (void)hipFree(frame.planes[0].dev_ptr);
        if (mask) (void)hipFree(mask_img.planes[0].dev_ptr);
    } catch (const std::exception &e) {
        std::cerr << "Error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
