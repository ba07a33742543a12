// OVX/UtilityOVX.hpp -- headless stand-in for the few OVXIO helpers that the VisionWorks algorithm
// wrappers use ("headless shim").  Authored for this repository; it mirrors the
// interface of nvxio/include/OVX/UtilityOVX.hpp:115-287 (macros, ContextGuard, printPerf,
// stdoutLogCallback, printVersionInfo) so that demos/feature_tracker/feature_tracker.cpp compiles
// UNMODIFIED against libvisualslam_amd.  NVXIO itself (GUI, GStreamer, GL) is out of scope.
#ifndef OVX_UTILITY_OVX_HPP
#define OVX_UTILITY_OVX_HPP

#include <cstring>
#include <iostream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>

#include <NVX/nvx.h>
#include <VX/vx.h>

#define NVXIO_THROW_EXCEPTION(msg) \
    do { std::ostringstream ostr_; ostr_ << msg; throw std::runtime_error(ostr_.str()); } while (0)

#define NVXIO_SAFE_CALL(vxOp) \
    do { \
        vx_status status = (vxOp); \
        if (status != VX_SUCCESS) \
            NVXIO_THROW_EXCEPTION(#vxOp << " failure [status = " << status << "]" << " in file " << __FILE__ << " line " << __LINE__); \
    } while (0)

#define NVXIO_ASSERT(cond) \
    do { if (!(cond)) NVXIO_THROW_EXCEPTION(#cond << " failure in file " << __FILE__ << " line " << __LINE__); } while (0)

#define NVXIO_CHECK_REFERENCE(ref) NVXIO_ASSERT(ref != 0 && vxGetStatus((vx_reference)ref) == VX_SUCCESS)

namespace ovxio {

inline void VX_CALLBACK stdoutLogCallback(vx_context, vx_reference, vx_status, const vx_char string[])
{
    std::cout << "[NVX LOG] " << string << std::endl;
}

inline bool &perfPrinting() { static bool on = false; return on; }   // the shim stays quiet unless asked

inline void printPerf(vx_graph graph, const char *label)
{
    vx_perf_t perf;
    NVXIO_SAFE_CALL( vxQueryGraph(graph, VX_GRAPH_ATTRIBUTE_PERFORMANCE, &perf, sizeof(perf)) );
    if (perfPrinting()) std::cout << label << " Graph Time : " << perf.tmp / 1000000.0 << " ms" << std::endl;
}

inline void printPerf(vx_node node, const char *label)
{
    vx_perf_t perf;
    NVXIO_SAFE_CALL( vxQueryNode(node, VX_NODE_ATTRIBUTE_PERFORMANCE, &perf, sizeof(perf)) );
    if (perfPrinting()) std::cout << "\t " << label << " Time : " << perf.tmp / 1000000.0 << " ms" << std::endl;
}

inline void printVersionInfo()
{
    nvx_version_info_t info;
    nvxGetVersionInfo(&info);
    std::cout << "library version : " << info.visionworks_version.major << "." << info.visionworks_version.minor << "."
              << info.visionworks_version.patch << info.visionworks_version.suffix << ", OpenVX "
              << info.openvx_major_version << "." << info.openvx_minor_version << std::endl;
}

inline void checkIfContextIsValid(vx_context context)
{
    vx_status status = vxGetStatus((vx_reference)context);
    if (status == NVX_ERROR_NO_GPU) NVXIO_THROW_EXCEPTION("HIP-capable GPU was not found.");
    if (status == NVX_ERROR_UNSUPPORTED_GPU) NVXIO_THROW_EXCEPTION("Unsupported GPU.");
    NVXIO_CHECK_REFERENCE(context);
}

struct ContextGuard
{
    ContextGuard() : context(vxCreateContext()) { checkIfContextIsValid(context); }
    ContextGuard(const ContextGuard &) = delete;
    ContextGuard &operator=(const ContextGuard &) = delete;
    ~ContextGuard() { vxReleaseContext(&context); }
    operator vx_context() const { return context; }
    operator vx_reference() const { return (vx_reference)context; }

private:
    vx_context context;
};

}  // namespace ovxio

#endif
