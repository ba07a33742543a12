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
//     uri = /path/frame_%04d.pgm | .ppm         (the reference's rule: a '%' in the path = image sequence,
//       nvxio/src/NVX/FrameSource/FrameSource.cpp:198-328) binary PGM (P5) / PPM (P6) files numbered from the first
//       of 0 or 1 that exists, delivered as RGBX (grey replicated, X = 255); CLOSED after the last file
//     uri = /path/picture.pgm | .ppm            one image: SINGLE_IMAGE_SOURCE, the same frame on every fetch
//
//   ovxio::createDefaultRender(context, title, w, h, ...)   nvxio/include/OVX/RenderOVX.hpp:380
//       records what the demo draws instead of drawing it: every putArrows(old, new) appends
//       u32 n_old, n_old x {f32 x,y}, u32 n_new, n_new x {f32 x,y} to $VSA_HEADLESS_OUT (if set);
//       every putImage(image) appends u32 width, u32 height and the packed RGBX (or U8) pixels to
//       $VSA_HEADLESS_IMAGES (if set) -- what demos/video_stabilizer/main_video_stabilizer.cpp:291 would show;
//       every call is also written to $VSA_HEADLESS_LOG (if set) as one line of the reference's EventLogger text
//       format (nvxio/src/NVX/Render/EventLogger.cpp:186-196,203-209,255-265,271-324,484-532: "<frame>: image(..)",
//       "features(color(..), n,ftr(x,y)..)", "arrows(..)", "textBox(..)", "object(..)", "circles(..)", "lines(..)") --
//       a regression log two runs can be diffed by;
//       flush() returns false -- which ends the demo's loop (main_feature_tracker.cpp:385-388) --
//       after $VSA_HEADLESS_FRAMES flushes (default 10).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

// binary PGM / PPM reader: "P5|P6 <w> <h> <maxval<=255>" with '#' comments, then the pixels
bool readPnm(const std::string &path, vx_uint32 *w, vx_uint32 *h, std::vector<unsigned char> *rgbx)
{
    std::ifstream in(path, std::ios::binary);
    if (!in) return false;
    std::string magic;
    in >> magic;
    if (magic != "P5" && magic != "P6") return false;
    int vals[3], got = 0;
    while (got < 3) {
        in >> std::ws;
        if (in.peek() == '#') { std::string c; std::getline(in, c); continue; }
        if (!(in >> vals[got])) return false;
        ++got;
    }
    in.get();                                   // the single whitespace byte after maxval
    if (vals[0] <= 0 || vals[1] <= 0 || vals[2] <= 0 || vals[2] > 255) return false;
    const size_t ch = magic == "P5" ? 1 : 3, n = (size_t)vals[0] * vals[1];
    std::vector<unsigned char> raw(n * ch);
    in.read(reinterpret_cast<char *>(raw.data()), (std::streamsize)raw.size());
    if ((size_t)in.gcount() != raw.size()) return false;
    rgbx->resize(n * 4);
    for (size_t i = 0; i < n; ++i) {
        unsigned char *o = rgbx->data() + 4 * i;
        o[0] = raw[ch * i]; o[1] = raw[ch * i + (ch == 3 ? 1 : 0)]; o[2] = raw[ch * i + (ch == 3 ? 2 : 0)]; o[3] = 255;
    }
    *w = (vx_uint32)vals[0]; *h = (vx_uint32)vals[1];
    return true;
}

class PnmSource : public ovxio::FrameSource {
public:
    PnmSource(const std::string &pattern, bool sequence)
        : ovxio::FrameSource(sequence ? ovxio::FrameSource::IMAGE_SEQUENCE_SOURCE : ovxio::FrameSource::SINGLE_IMAGE_SOURCE, "PnmSource"),
          pattern_(pattern), sequence_(sequence)
    {
        cfg_.frameWidth = cfg_.frameHeight = 0;
        cfg_.format = VX_DF_IMAGE_RGBX;
        cfg_.fps = 30;
    }
    bool open() override
    {
        opened_ = false;
        first_ = 0;
        vx_uint32 w = 0, h = 0;
        if (sequence_ && !readPnm(name(0), &w, &h, &buf_)) first_ = 1;
        if (!readPnm(name(first_), &w, &h, &buf_)) return false;
        cfg_.frameWidth = w; cfg_.frameHeight = h;
        next_ = first_;
        opened_ = true;
        return true;
    }
    FrameStatus fetch(vx_image image, vx_uint32 /*timeout*/) override
    {
        if (!opened_) return CLOSED;
        vx_uint32 w = 0, h = 0;
        if (!readPnm(name(next_), &w, &h, &buf_) || w != cfg_.frameWidth || h != cfg_.frameHeight) { opened_ = false; return CLOSED; }
        if (sequence_) ++next_;
        vx_imagepatch_addressing_t addr = {};
        addr.dim_x = w; addr.dim_y = h; addr.stride_x = 4; addr.stride_y = (vx_int32)(w * 4);
        NVXIO_SAFE_CALL( vxCopyImagePatch(image, nullptr, 0, &addr, buf_.data(), VX_WRITE_ONLY, VX_MEMORY_TYPE_HOST) );
        return OK;
    }
    Parameters getConfiguration() override { return cfg_; }
    bool setConfiguration(const Parameters &) override { return false; }
    void close() override { opened_ = false; }

private:
    std::string name(int i) const
    {
        if (!sequence_) return pattern_;
        char b[4096];
        snprintf(b, sizeof(b), pattern_.c_str(), i);
        return b;
    }
    std::string pattern_;
    bool sequence_, opened_ = false;
    int first_ = 0, next_ = 0;
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
        if (const char *o = getenv("VSA_HEADLESS_LOG")) log_ = fopen(o, "w");
    }
    ~RecordingRender() { if (log_) fclose(log_); }
    void setOnKeyboardEventCallback(OnKeyboardEventCallback, void *) override {}
    void setOnMouseEventCallback(OnMouseEventCallback, void *) override {}
    void putImage(vx_image image) override
    {
        vx_uint32 w = 0, h = 0;
        vx_df_image fmt = VX_DF_IMAGE_VIRT;
        NVXIO_SAFE_CALL( vxQueryImage(image, VX_IMAGE_ATTRIBUTE_WIDTH, &w, sizeof(w)) );
        NVXIO_SAFE_CALL( vxQueryImage(image, VX_IMAGE_ATTRIBUTE_HEIGHT, &h, sizeof(h)) );
        NVXIO_SAFE_CALL( vxQueryImage(image, VX_IMAGE_ATTRIBUTE_FORMAT, &fmt, sizeof(fmt)) );
        if (log_) fprintf(log_, "%d: image(%d, %dx%d)\n", frame_, (int)fmt, w, h);
        if (!images_.is_open()) return;
        const vx_uint32 bpp = fmt == VX_DF_IMAGE_RGBX ? 4u : 1u;
        std::vector<unsigned char> px((size_t)w * h * bpp);
        vx_imagepatch_addressing_t addr = {};
        addr.dim_x = w; addr.dim_y = h; addr.stride_x = (vx_int32)bpp; addr.stride_y = (vx_int32)(w * bpp);
        NVXIO_SAFE_CALL( vxCopyImagePatch(image, nullptr, 0, &addr, px.data(), VX_READ_ONLY, VX_MEMORY_TYPE_HOST) );
        images_.write(reinterpret_cast<const char *>(&w), 4);
        images_.write(reinterpret_cast<const char *>(&h), 4);
        images_.write(reinterpret_cast<const char *>(px.data()), (std::streamsize)px.size());
    }
    void putTextViewport(const std::string &text, const TextBoxStyle &style) override
    {
        if (!log_) return;
        std::string filtered;
        for (char c : text) { if (c == '\n') filtered += "\\n"; else filtered += c; }
        fprintf(log_, "%d: textBox(color(%u,%u,%u,%u), bkcolor(%u,%u,%u,%u), origin(%u,%u), \"%s\")\n", frame_,
                style.color[0], style.color[1], style.color[2], style.color[3],
                style.bgcolor[0], style.bgcolor[1], style.bgcolor[2], style.bgcolor[3], style.origin.x, style.origin.y, filtered.c_str());
    }
    void putFeatures(vx_array arr, const FeatureStyle &style) override
    {
        if (!log_) return;
        vx_size n = 0, isz = 0;
        NVXIO_SAFE_CALL( vxQueryArray(arr, VX_ARRAY_ATTRIBUTE_NUMITEMS, &n, sizeof(n)) );
        NVXIO_SAFE_CALL( vxQueryArray(arr, VX_ARRAY_ATTRIBUTE_ITEMSIZE, &isz, sizeof(isz)) );
        fprintf(log_, "%d: features(color(%u,%u,%u,%u), %u", frame_, style.color[0], style.color[1], style.color[2], style.color[3], (unsigned)n);
        if (n) {   // POINT2F and KEYPOINTF both start with {f32 x, y}
            std::vector<unsigned char> raw(n * isz);
            NVXIO_SAFE_CALL( vxCopyArrayRange(arr, 0, n, isz, raw.data(), VX_READ_ONLY, VX_MEMORY_TYPE_HOST) );
            for (vx_size i = 0; i < n; ++i) {
                float xy[2];
                memcpy(xy, raw.data() + i * isz, 8);
                fprintf(log_, ",ftr(%.1f,%.1f)", xy[0], xy[1]);
            }
        }
        fprintf(log_, ")\n");
    }
    void putFeatures(vx_array arr, vx_array) override
    {
        if (!log_) return;
        vx_size n = 0;
        NVXIO_SAFE_CALL( vxQueryArray(arr, VX_ARRAY_ATTRIBUTE_NUMITEMS, &n, sizeof(n)) );
        fprintf(log_, "%d: features(%u)\n", frame_, (unsigned)n);     // per-feature styles are not recorded
    }
    void putLines(vx_array arr, const LineStyle &style) override { logCounted("lines", arr, style.color, style.thickness); }
    void putConvexPolygon(vx_array arr, const LineStyle &style) override { logCounted("polygon", arr, style.color, style.thickness); }
    void putMotionField(vx_image, const MotionFieldStyle &) override {}
    void putObjectLocation(const vx_rectangle_t &r, const DetectedObjectStyle &style) override
    {
        if (log_) fprintf(log_, "%d: object(color(%u,%u,%u,%u), location(%u,%u,%u,%u), \"%s\")\n", frame_, style.color[0], style.color[1],
                          style.color[2], style.color[3], r.start_x, r.start_y, r.end_x, r.end_y, style.label.c_str());
    }
    void putCircles(vx_array arr, const CircleStyle &style) override { logCounted("circles", arr, style.color, style.thickness); }
    void putArrows(vx_array old_points, vx_array new_points, const LineStyle &style) override
    {
        if (log_) {
            vx_size a = 0, b = 0;
            NVXIO_SAFE_CALL( vxQueryArray(old_points, VX_ARRAY_ATTRIBUTE_NUMITEMS, &a, sizeof(a)) );
            NVXIO_SAFE_CALL( vxQueryArray(new_points, VX_ARRAY_ATTRIBUTE_NUMITEMS, &b, sizeof(b)) );
            fprintf(log_, "%d: arrows(color(%u,%u,%u,%u), thickness(%d), %u)\n", frame_, style.color[0], style.color[1], style.color[2],
                    style.color[3], style.thickness, (unsigned)std::min(a, b));
        }
        if (!out_.is_open()) return;
        writePoints(old_points);
        writePoints(new_points);
    }
    bool flush() override
    {
        ++frame_;                                   // EventLogger.cpp:534-540
        if (log_) fflush(log_);
        return ++flushed_ < frames_;
    }
    void close() override { out_.close(); images_.close(); if (log_) { fclose(log_); log_ = nullptr; } }
    vx_uint32 getViewportWidth() const override { return w_; }
    vx_uint32 getViewportHeight() const override { return h_; }

private:
    void logCounted(const char *what, vx_array arr, const vx_uint8 *color, int thickness)
    {
        if (!log_) return;
        vx_size n = 0;
        NVXIO_SAFE_CALL( vxQueryArray(arr, VX_ARRAY_ATTRIBUTE_NUMITEMS, &n, sizeof(n)) );
        fprintf(log_, "%d: %s(color(%u,%u,%u,%u), thickness(%d), %u)\n", frame_, what, color[0], color[1], color[2], color[3], thickness, (unsigned)n);
    }
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
    int frames_ = 10, flushed_ = 0, frame_ = 0;
    std::ofstream out_, images_;
    FILE *log_ = nullptr;
};

}  // namespace

namespace ovxio {

std::unique_ptr<FrameSource> createDefaultFrameSource(vx_context, const std::string &uri)
{
    Uri u;
    if (uri.find("://") == std::string::npos) {     // a plain path: image or image sequence, by extension
        const size_t dot = uri.rfind('.');
        std::string ext = dot == std::string::npos ? "" : uri.substr(dot + 1);
        std::transform(ext.begin(), ext.end(), ext.begin(), ::tolower);
        if (ext != "pgm" && ext != "ppm") return nullptr;
        return std::unique_ptr<FrameSource>(new PnmSource(uri, uri.find('%') != std::string::npos));
    }
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
