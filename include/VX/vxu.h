/*
 * VX/vxu.h -- immediate-mode OpenVX calls used on the first frame of the feature tracker
 * (demos/feature_tracker/feature_tracker.cpp:322-323).  Synchronous: results are readable on return.
 */
#ifndef VX_VXU_H
#define VX_VXU_H

#include <VX/vx.h>

#ifdef __cplusplus
extern "C" {
#endif

/* feature_tracker.cpp:322 */
VX_API_ENTRY vx_status VX_API_CALL vxuColorConvert(vx_context context, vx_image input, vx_image output);
/* feature_tracker.cpp:323 -- the input is copied into level 0 unless it already is level 0 */
VX_API_ENTRY vx_status VX_API_CALL vxuGaussianPyramid(vx_context context, vx_image input, vx_pyramid gaussian);
VX_API_ENTRY vx_status VX_API_CALL vxuOpticalFlowPyrLK(vx_context context, vx_pyramid old_images, vx_pyramid new_images,
                                                       vx_array old_points, vx_array new_points_estimates,
                                                       vx_array new_points, vx_enum termination, vx_scalar epsilon,
                                                       vx_scalar num_iterations, vx_scalar use_initial_estimate,
                                                       vx_size window_dimension);

/* immediate form of vxWarpPerspectiveNode (VX/vx.h); border = constant 0 */
VX_API_ENTRY vx_status VX_API_CALL vxuWarpPerspective(vx_context context, vx_image input, vx_matrix matrix, vx_enum type, vx_image output);

#ifdef __cplusplus
}
#endif
#endif
