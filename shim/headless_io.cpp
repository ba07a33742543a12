// headless_io.cpp -- the two OVXIO factories the reference's demo main needs, for a machine with no
// window system, video decoder or camera ("headless shim").  Everything else of
// NVXIO that demos/feature_tracker/main_feature_tracker.cpp uses (Application, ConfigParser,
// OptionHandler, SyncTimer, UtilityOVX) is compiled from the reference's own sources where they lie;
// the reference's FrameSource/Render implementations are CUDA + GStreamer + OpenGL code and are out of
// scope, so this file supplies
//
//   ovxio::createDefaultFrameSource(context, uri)     nvxio/include/OVX/FrameSourceOVX.hpp:305
//       uri = rawrgbx:///path/to/frames.rgbx?width=W&height=H[&fps=F]
//       a file of packed RGBX frames, read in order; CLOSED at end of file (the demo re-opens a closed
//       source, main_feature_tracker.cpp:324-330, so the sequence loops like a video file would)
//
//   ovxio::createDefaultRender(context, title, w, h, ...)   nvxio/include/OVX/RenderOVX.hpp:380
//       records what the demo draws instead of drawing it: every putArrows(old, new) appends
//       u32 n_old, n_old x {f32 x,y}, u32 n_new, n_new x {f32 x,y} to $VSA_HEADLESS_OUT (if set);
//       every putImage(image) appends u32 width, u32 height and the packed RGBX (or U8) pixels to
//       $VSA_HEADLESS_IMAGES (if set) -- what demos/video_stabilizer/main_video_stabilizer.cpp:291 would show;
//       flush() returns false -- which ends the demo's loop (main_feature_tracker.cpp:385-388) --
//       after $VSA_HEADLESS_FRAMES flushes (default 10).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include <NVX/nvx.h>
#include <OVX/FrameSourceOVX.hpp>
#include <OVX/RenderOVX.hpp>
#include <OVX/UtilityOVX.hpp>

namespace {

struct Uri {
    std::string protocol, path;
    std::map<std::string, std::string> query;
};

bool parseUri(const std::string &uri, Uri *out)
{
    const size_t p = uri.find("://");
    if (p == std::string::npos) return false;
    out->protocol = uri.substr(0, p);
    std::string rest = uri.substr(p + 3);
    const size_t q = rest.find('?');
    out->path = rest.substr(0, q);
    if (q != std::string::npos) {
        std::string qs = rest.substr(q + 1);
        size_t pos = 0;
        while (pos < qs.size()) {
            size_t amp = qs.find('&', pos);
            if (amp == std::string::npos) amp = qs.size();
            const std::string kv = qs.substr(pos, amp - pos);
            const size_t eq = kv.find('=');
            if (eq != std::string::npos) out->query[kv.substr(0, eq)] = kv.substr(eq + 1);
            pos = amp + 1;
        }
    }
    return true;
}

class RawRgbxSource : public ovxio::FrameSource {
public:
    RawRgbxSource(const std::string &path, vx_uint32 w, vx_uint32 h, vx_uint32 fps)
        : ovxio::FrameSource(ovxio::FrameSource::VIDEO_SOURCE, "RawRgbxSource"), path_(path), buf_((size_t)w * h * 4)
    {
        cfg_.frameWidth = w;
        cfg_.frameHeight = h;
        cfg_.format = VX_DF_IMAGE_RGBX;
        cfg_.fps = fps;
    }
    bool open() override
    {
        in_.close();
        in_.clear();
        in_.open(path_, std::ios::binary);
        return (bool)in_;
    }
    FrameStatus fetch(vx_image image, vx_uint32 /*timeout*/) override
    {
        if (!in_.is_open()) return CLOSED;
        in_.read(reinterpret_cast<char *>(buf_.data()), (std::streamsize)buf_.size());
        if ((size_t)in_.gcount() != buf_.size()) {
            in_.close();
            return CLOSED;
        }
        vx_imagepatch_addressing_t addr = {};
        addr.dim_x = cfg_.frameWidth;
        addr.dim_y = cfg_.frameHeight;
        addr.stride_x = 4;
        addr.stride_y = (vx_int32)(cfg_.frameWidth * 4);
        NVXIO_SAFE_CALL( vxCopyImagePatch(image, nullptr, 0, &addr, buf_.data(), VX_WRITE_ONLY, VX_MEMORY_TYPE_HOST) );
        return OK;
    }
    Parameters getConfiguration() override { return cfg_; }
    bool setConfiguration(const Parameters &) override { return false; }
    void close() override { in_.close(); }

private:
    std::string path_;
    std::ifstream in_;
    std::vector<unsigned char> buf_;
    Parameters cfg_;
};

class RecordingRender : public ovxio::Render {
public:
    RecordingRender(vx_uint32 w, vx_uint32 h) : ovxio::Render(ovxio::Render::UNKNOWN_RENDER, "RecordingRender"), w_(w), h_(h)
    {
        if (const char *e = getenv("VSA_HEADLESS_FRAMES")) frames_ = atoi(e);
        if (const char *o = getenv("VSA_HEADLESS_OUT")) out_.open(o, std::ios::binary);
        if (const char *o = getenv("VSA_HEADLESS_IMAGES")) images_.open(o, std::ios::binary);
    }
    void setOnKeyboardEventCallback(OnKeyboardEventCallback, void *) override {}
    void setOnMouseEventCallback(OnMouseEventCallback, void *) override {}
    void putImage(vx_image image) override
    {
        if (!images_.is_open()) return;
        vx_uint32 w = 0, h = 0;
        vx_df_image fmt = VX_DF_IMAGE_VIRT;
        NVXIO_SAFE_CALL( vxQueryImage(image, VX_IMAGE_ATTRIBUTE_WIDTH, &w, sizeof(w)) );
        NVXIO_SAFE_CALL( vxQueryImage(image, VX_IMAGE_ATTRIBUTE_HEIGHT, &h, sizeof(h)) );
        NVXIO_SAFE_CALL( vxQueryImage(image, VX_IMAGE_ATTRIBUTE_FORMAT, &fmt, sizeof(fmt)) );
        const vx_uint32 bpp = fmt == VX_DF_IMAGE_RGBX ? 4u : 1u;
        std::vector<unsigned char> px((size_t)w * h * bpp);
        vx_imagepatch_addressing_t addr = {};
        addr.dim_x = w; addr.dim_y = h; addr.stride_x = (vx_int32)bpp; addr.stride_y = (vx_int32)(w * bpp);
        NVXIO_SAFE_CALL( vxCopyImagePatch(image, nullptr, 0, &addr, px.data(), VX_READ_ONLY, VX_MEMORY_TYPE_HOST) );
        images_.write(reinterpret_cast<const char *>(&w), 4);
        images_.write(reinterpret_cast<const char *>(&h), 4);
        images_.write(reinterpret_cast<const char *>(px.data()), (std::streamsize)px.size());
    }
    void putTextViewport(const std::string &, const TextBoxStyle &) override {}
    void putFeatures(vx_array, const FeatureStyle &) override {}
    void putFeatures(vx_array, vx_array) override {}
    void putLines(vx_array, const LineStyle &) override {}
    void putConvexPolygon(vx_array, const LineStyle &) override {}
    void putMotionField(vx_image, const MotionFieldStyle &) override {}
    void putObjectLocation(const vx_rectangle_t &, const DetectedObjectStyle &) override {}
    void putCircles(vx_array, const CircleStyle &) override {}
    void putArrows(vx_array old_points, vx_array new_points, const LineStyle &) override
    {
        if (!out_.is_open()) return;
        writePoints(old_points);
        writePoints(new_points);
    }
    bool flush() override { return ++flushed_ < frames_; }
    void close() override { out_.close(); images_.close(); }
    vx_uint32 getViewportWidth() const override { return w_; }
    vx_uint32 getViewportHeight() const override { return h_; }

private:
    void writePoints(vx_array arr)
    {
        vx_size n = 0;
        NVXIO_SAFE_CALL( vxQueryArray(arr, VX_ARRAY_ATTRIBUTE_NUMITEMS, &n, sizeof(n)) );
        std::vector<nvx_point2f_t> pts(n);
        if (n) NVXIO_SAFE_CALL( vxCopyArrayRange(arr, 0, n, sizeof(nvx_point2f_t), pts.data(), VX_READ_ONLY, VX_MEMORY_TYPE_HOST) );
        const vx_uint32 n32 = (vx_uint32)n;
        out_.write(reinterpret_cast<const char *>(&n32), 4);
        out_.write(reinterpret_cast<const char *>(pts.data()), (std::streamsize)(n * sizeof(nvx_point2f_t)));
    }
    vx_uint32 w_, h_;
    int frames_ = 10, flushed_ = 0;
    std::ofstream out_, images_;
};

}  // namespace

namespace ovxio {

std::unique_ptr<FrameSource> createDefaultFrameSource(vx_context, const std::string &uri)
{
    Uri u;
    if (!parseUri(uri, &u) || u.protocol != "rawrgbx") return nullptr;   // the reference returns nullptr for unknown URIs too
    const int w = atoi(u.query["width"].c_str()), h = atoi(u.query["height"].c_str());
    const int fps = u.query.count("fps") ? atoi(u.query["fps"].c_str()) : 30;
    if (w <= 0 || h <= 0) return nullptr;
    return std::unique_ptr<FrameSource>(new RawRgbxSource(u.path, (vx_uint32)w, (vx_uint32)h, (vx_uint32)fps));
}

std::unique_ptr<Render> createDefaultRender(vx_context, const std::string &, vx_uint32 width, vx_uint32 height, vx_uint32, bool, bool)
{
    return std::unique_ptr<Render>(new RecordingRender(width, height));
}

}  // namespace ovxio
