// stream_driver.cpp -- host side of the multi-camera bench: the per-frame call sequence of
// nvx::FeatureTracker::track (demos/feature_tracker/feature_tracker.cpp:216-224)
//
//     vxSetParameterByIndex(cvt, 0, frame); vxSetParameterByIndex(detector, 2, mask);
//     vxAgeDelay(pyramids); vxAgeDelay(points); vxProcessGraph(graph);
//
// driven on N camera streams by N native host threads, one per stream, exactly as N instances of the reference
// application (a C++ program) would drive it.  It uses nothing but the public C API of include/VX/vx.h and is built
// into its own small library (libvsa_stream_driver.so) next to libvisualslam_amd.so: bench.py creates the objects
// through the same C API and hands the handles over, so the timed loop has no interpreter in it (16 Python threads
// taking turns on the interpreter lock capped the whole bench at ~25 k frames/s whatever the GPU did).
#include <VX/vx.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

extern "C" {

// One stream's loop over frames k0 .. k0+n_frames-1: frame k is ring[(phase + 1 + k) % ring_len].  Returns the first
// status that is not VX_SUCCESS.
int vsa_drive_stream(vx_graph graph, vx_node cvt, vx_node det, vx_delay pyr_delay, vx_delay pts_delay, const vx_image *ring,
                     int ring_len, int phase, int k0, int n_frames)
{
    if (!graph || !cvt || !det || !pyr_delay || !pts_delay || !ring || ring_len <= 0 || n_frames < 0) return VX_ERROR_INVALID_PARAMETERS;
    for (int k = k0; k < k0 + n_frames; ++k) {
        vx_image frame = ring[(unsigned)(phase + 1 + k) % (unsigned)ring_len];
        vx_status s = vxSetParameterByIndex(cvt, 0, (vx_reference)frame);
        if (s == VX_SUCCESS) s = vxSetParameterByIndex(det, 2, nullptr);     // no mask (feature_tracker.cpp:217 passes the caller's, may be NULL)
        if (s == VX_SUCCESS) s = vxAgeDelay(pyr_delay);
        if (s == VX_SUCCESS) s = vxAgeDelay(pts_delay);
        if (s == VX_SUCCESS) s = vxProcessGraph(graph);
        if (s != VX_SUCCESS) return (int)s;
    }
    return VX_SUCCESS;
}

// n_streams independent streams, one host thread each; returns when all have finished their n_frames frames.
int vsa_drive_streams(int n_streams, const vx_graph *graphs, const vx_node *cvt, const vx_node *det, const vx_delay *pyr_delay,
                      const vx_delay *pts_delay, const vx_image *ring, int ring_len, const int *phase, int k0, int n_frames)
{
    if (n_streams <= 0 || !graphs || !cvt || !det || !pyr_delay || !pts_delay || !phase) return VX_ERROR_INVALID_PARAMETERS;
    std::atomic<int> first_error{VX_SUCCESS};
    std::vector<std::thread> threads;
    threads.reserve((size_t)n_streams);
    for (int i = 0; i < n_streams; ++i)
        threads.emplace_back([&, i] {
            const int s = vsa_drive_stream(graphs[i], cvt[i], det[i], pyr_delay[i], pts_delay[i], ring, ring_len, phase[i], k0, n_frames);
            int expected = VX_SUCCESS;
            if (s != VX_SUCCESS) first_error.compare_exchange_strong(expected, s);
        });
    for (auto &t : threads) t.join();
    return first_error.load();
}

// ---- persistent driver -------------------------------------------------------------------------------------------
// The same loops on threads that outlive a run: vsa_driver_create starts one thread per stream, vsa_driver_run releases
// them together for n_frames frames and returns when the last one is done, so a timed run contains the per-frame calls
// and nothing else -- no thread creation or join (a 20-step run of 32 streams is ~20 ms; spawning 32 threads inside it
// cost ~5 %).  Handles are borrowed: the caller keeps the graphs, nodes, delays and frames alive until vsa_driver_destroy.
struct vsa_driver {
    std::vector<vx_graph> graphs;
    std::vector<vx_node> cvt, det;
    std::vector<vx_delay> pyr_delay, pts_delay;
    std::vector<vx_image> ring;
    std::vector<int> phase;
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable go, done;
    std::atomic<unsigned long> generation{0};     // bumped by every run (and by destroy)
    int k0 = 0, n_frames = 0, running = 0, first_error = VX_SUCCESS;
    bool quit = false;
    // VSA_DRIVER_TRACE=<prefix>: the time every frame of a run completed on the host (microseconds from the run's start),
    // one line per stream, into <prefix>.<streams>_streams_<frames>_frames -- where a short run loses its time (tools/exp_trace_run.sh)
    const char *trace_path = nullptr;
    std::vector<double> trace, trace_start;   // trace_start: when the stream's thread began the run
    std::chrono::steady_clock::time_point t_run;
};

static void vsa_driver_thread(vsa_driver *d, int i)
{
    unsigned long seen = 0;
    for (;;) {
        int k0, n;
        {
            std::unique_lock<std::mutex> lk(d->mu);
            d->go.wait(lk, [&] { return d->generation.load(std::memory_order_acquire) != seen; });
            seen = d->generation.load(std::memory_order_acquire);
            if (d->quit) return;
            k0 = d->k0; n = d->n_frames;
        }
        int s = VX_SUCCESS;
        if (!d->trace_path) {
            s = vsa_drive_stream(d->graphs[i], d->cvt[i], d->det[i], d->pyr_delay[i], d->pts_delay[i], d->ring.data(),
                                 (int)d->ring.size(), d->phase[i], k0, n);
        } else {
            d->trace_start[(size_t)i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - d->t_run).count();
            for (int j = 0; j < n && s == VX_SUCCESS; ++j) {
                s = vsa_drive_stream(d->graphs[i], d->cvt[i], d->det[i], d->pyr_delay[i], d->pts_delay[i], d->ring.data(),
                                     (int)d->ring.size(), d->phase[i], k0 + j, 1);
                d->trace[(size_t)i * (size_t)n + (size_t)j] =
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - d->t_run).count();
            }
        }
        {
            std::lock_guard<std::mutex> lk(d->mu);
            if (s != VX_SUCCESS && d->first_error == VX_SUCCESS) d->first_error = s;
            if (--d->running == 0) d->done.notify_all();
        }
    }
}

vsa_driver *vsa_driver_create(int n_streams, const vx_graph *graphs, const vx_node *cvt, const vx_node *det, const vx_delay *pyr_delay,
                              const vx_delay *pts_delay, const vx_image *ring, int ring_len, const int *phase)
{
    if (n_streams <= 0 || !graphs || !cvt || !det || !pyr_delay || !pts_delay || !ring || ring_len <= 0 || !phase) return nullptr;
    auto *d = new vsa_driver();
    d->graphs.assign(graphs, graphs + n_streams);
    d->cvt.assign(cvt, cvt + n_streams);
    d->det.assign(det, det + n_streams);
    d->pyr_delay.assign(pyr_delay, pyr_delay + n_streams);
    d->pts_delay.assign(pts_delay, pts_delay + n_streams);
    d->ring.assign(ring, ring + ring_len);
    d->phase.assign(phase, phase + n_streams);
    d->trace_path = getenv("VSA_DRIVER_TRACE");
    for (int i = 0; i < n_streams; ++i) d->threads.emplace_back(vsa_driver_thread, d, i);
    return d;
}

// frames k0 .. k0+n_frames-1 on every stream, concurrently; returns the first status that is not VX_SUCCESS
int vsa_driver_run(vsa_driver *d, int k0, int n_frames)
{
    if (!d || n_frames < 0) return VX_ERROR_INVALID_PARAMETERS;
    std::unique_lock<std::mutex> lk(d->mu);
    d->k0 = k0; d->n_frames = n_frames;
    d->running = (int)d->threads.size();
    d->first_error = VX_SUCCESS;
    if (d->trace_path) { d->trace_start.assign(d->threads.size(), 0.0); d->trace.assign(d->threads.size() * (size_t)n_frames, 0.0); }
    d->t_run = std::chrono::steady_clock::now();
    d->generation.fetch_add(1, std::memory_order_release);   // k0 / n_frames / running are published with it
    d->go.notify_all();
    d->done.wait(lk, [&] { return d->running == 0; });
    if (d->trace_path) {
        char path[1024];
        snprintf(path, sizeof(path), "%s.%zu_streams_%d_frames", d->trace_path, d->threads.size(), n_frames);
        if (FILE *f = fopen(path, "w")) {   // one file per (stream count, run length): a later run of the same shape replaces it
            for (size_t i = 0; i < d->threads.size(); ++i) {
                fprintf(f, "%.1f ", d->trace_start[i]);   // first column: the thread's start
                for (int j = 0; j < n_frames; ++j) fprintf(f, "%.1f ", d->trace[i * (size_t)n_frames + (size_t)j]);
                fputc('\n', f);
            }
            fclose(f);
        }
    }
    return d->first_error;
}

void vsa_driver_destroy(vsa_driver *d)
{
    if (!d) return;
    {
        std::lock_guard<std::mutex> lk(d->mu);
        d->quit = true;
        d->generation.fetch_add(1, std::memory_order_release);
    }
    d->go.notify_all();
    for (auto &t : d->threads) t.join();
    delete d;
}

}  // extern "C"
