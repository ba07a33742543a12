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

}  // extern "C"
