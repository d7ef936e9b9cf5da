// reference_glue.hpp — what a maintainer of Artemarius/cuda-gaussian-splatting adds to the tree
// (see INTEGRATION.md).  Header-only; needs the reference's headers (core/gaussian.hpp,
// core/types.hpp -> Eigen3) and is therefore NOT compiled in this repository's build image: it has been
// checked against the reference's declarations by reading only.  cugs::FusedAdam's member bodies are in
// reference_fused_adam.cpp (same status).
//
// It defines the reference's own entry points - namespace cugs, identical signatures
// (rasterizer/rasterizer.hpp:57,88; projection.hpp:39; sorting.hpp:41; forward.hpp:41;
// backward.hpp:39; projection_backward.hpp:44; core/sh.hpp:29; core/sh_backward.hpp:25) - on top of
// cugs_hip_torch.hpp, so that src/rasterizer/*.cu, src/core/sh*.cu and src/optimizer/fused_adam.cu
// drop out of the build and apps/train_main.cpp / tests/test_rasterizer.cpp keep their calls.
#pragma once

#include "core/gaussian.hpp"
#include "core/types.hpp"
#include "rasterizer/rasterizer.hpp"
#include "rasterizer/projection.hpp"
#include "rasterizer/sorting.hpp"
#include "rasterizer/forward.hpp"
#include "rasterizer/backward.hpp"
#include "rasterizer/projection_backward.hpp"

#include "cugs_hip_torch.hpp"

namespace cugs {

/// CameraInfo -> POD: the float[16] built at projection.cu:228-233 and the camera centre of :273-275.
inline cugs_camera to_pod(const CameraInfo& camera) {
    cugs_camera c{};
    const Eigen::Matrix4f w2c = camera.world_to_camera();
    for (int r = 0; r < 4; ++r)
        for (int col = 0; col < 4; ++col) c.view[r * 4 + col] = w2c(r, col);     // Eigen is column-major
    c.fx = camera.intrinsics.fx; c.fy = camera.intrinsics.fy; c.cx = camera.intrinsics.cx; c.cy = camera.intrinsics.cy;
    c.width = camera.width; c.height = camera.height;
    const Eigen::Vector3f cc = camera.camera_center();
    c.cam_center[0] = cc.x(); c.cam_center[1] = cc.y(); c.cam_center[2] = cc.z();
    return c;
}
inline cugs_hip::ModelTensors tensors_of(const GaussianModel& m) {
    return {m.positions, m.sh_coeffs, m.opacities, m.rotations, m.scales};
}
inline cugs_hip::RenderSettings settings_of(const RenderSettings& s) {
    cugs_hip::RenderSettings o;
    for (int i = 0; i < 3; ++i) o.background[i] = s.background[i];
    o.active_sh_degree = s.active_sh_degree; o.scale_modifier = s.scale_modifier;
    return o;
}

// The scratch the HIP blend kernels want (packed records, [N,12]) is kept alive between render() and
// render_backward() in a small side table keyed by the colour tensor's storage, because RenderOutput
// (rasterizer.hpp:27-46) has no spare field.  A miss (a RenderOutput that did not come from render(), or one
// older than the table's eight entries) is not an error: cugs_hip::render_backward rebuilds the records from
// the four reference-layout arrays with one extra launch, so the benchmarked packed kernels run either way.
namespace glue_detail {
struct PackedTable {
    static constexpr int kSlots = 8;
    const void* key[kSlots] = {};
    torch::Tensor packed[kSlots];
    torch::Tensor gate[kSlots];        // [N] ReLU gate bits of the SH backward, from the projection
    torch::Tensor accum[kSlots];       // the backward's accumulator, cleared by the forward blend: newest render only, taken once
    int next = 0;
    void put(const torch::Tensor& color, const torch::Tensor& p, const torch::Tensor& zeroed_accum = {},
             const torch::Tensor& colour_gate = {}) {
        for (auto& a : accum) a = torch::Tensor();                // at most one 64 B/Gaussian buffer is kept alive
        key[next] = color.defined() ? color.data_ptr() : nullptr;
        packed[next] = p;
        gate[next] = colour_gate;
        accum[next] = zeroed_accum;
        next = (next + 1) % kSlots;
    }
    torch::Tensor take_accum(const torch::Tensor& color, int64_t n) {
        if (!color.defined()) return {};
        for (int i = 0; i < kSlots; ++i)
            if (key[i] == color.data_ptr() && accum[i].defined() && accum[i].size(0) == n) {
                torch::Tensor a = accum[i];
                accum[i] = torch::Tensor();
                return a;
            }
        return {};
    }
    torch::Tensor get_gate(const torch::Tensor& color, int64_t n) const {
        if (!color.defined()) return {};
        for (int i = 0; i < kSlots; ++i)
            if (key[i] == color.data_ptr() && gate[i].defined() && gate[i].size(0) == n) return gate[i];
        return {};
    }
    torch::Tensor get(const torch::Tensor& color, int64_t n) const {
        if (!color.defined()) return {};
        for (int i = 0; i < kSlots; ++i)
            if (key[i] == color.data_ptr() && packed[i].defined() && packed[i].size(0) == n) return packed[i];
        return {};
    }
};
inline PackedTable& packed_table() { static thread_local PackedTable t; return t; }   // host code is single-threaded
}  // namespace glue_detail

inline RenderOutput render(const GaussianModel& model, const CameraInfo& camera, const RenderSettings& settings) {
    TORCH_CHECK(model.is_valid(), "GaussianModel is not valid");                           // rasterizer.cpp:27
    auto r = cugs_hip::render(tensors_of(model), to_pod(camera), settings_of(settings));
    glue_detail::packed_table().put(r.color, r.packed, r.zeroed_accum, r.colour_gate);
    return RenderOutput{r.color, r.final_T, r.n_contrib, r.means_2d, r.depths, r.cov_2d_inv, r.radii, r.rgb,
                        r.opacities_act, r.gaussian_indices, r.tile_ranges};
}

inline BackwardOutput render_backward(const torch::Tensor& dL_dcolor, const RenderOutput& ro, const GaussianModel& model,
                                      const CameraInfo& camera, const RenderSettings& settings) {
    cugs_hip::RenderOutput h{ro.color, ro.final_T, ro.n_contrib, ro.means_2d, ro.depths, ro.cov_2d_inv, ro.radii, ro.rgb,
                             ro.opacities_act, ro.gaussian_indices, ro.tile_ranges,
                             glue_detail::packed_table().get(ro.color, model.num_gaussians()),
                             glue_detail::packed_table().get_gate(ro.color, model.num_gaussians())};
    h.zeroed_accum = glue_detail::packed_table().take_accum(ro.color, model.num_gaussians());
    auto b = cugs_hip::render_backward(dL_dcolor, h, tensors_of(model), to_pod(camera), settings_of(settings));
    return BackwardOutput{b.dL_dpositions, b.dL_drotations, b.dL_dscales, b.dL_dopacities, b.dL_dsh_coeffs, b.dL_dmeans_2d};
}

// ---- stage functions, reference signatures (forward.hpp:41, backward.hpp:39, projection_backward.hpp:44) ----
inline ForwardOutput rasterize_forward(const torch::Tensor& means_2d, const torch::Tensor& cov_2d_inv,
                                       const torch::Tensor& rgb, const torch::Tensor& opacities,
                                       const torch::Tensor& tile_ranges, const torch::Tensor& gaussian_indices,
                                       int img_w, int img_h, const float background[3]) {
    auto f = cugs_hip::rasterize_forward(means_2d, cov_2d_inv, rgb, opacities, tile_ranges, gaussian_indices, img_w, img_h,
                                         background);
    return ForwardOutput{f.color, f.final_T, f.n_contrib};
}

inline RasterizeBackwardOutput rasterize_backward(const torch::Tensor& dL_dcolor, const torch::Tensor& means_2d,
                                                  const torch::Tensor& cov_2d_inv, const torch::Tensor& rgb,
                                                  const torch::Tensor& opacities, const torch::Tensor& tile_ranges,
                                                  const torch::Tensor& gaussian_indices, const torch::Tensor& final_T,
                                                  const torch::Tensor& n_contrib, int img_w, int img_h,
                                                  const float background[3], int n_gaussians) {
    auto b = cugs_hip::rasterize_backward(dL_dcolor, means_2d, cov_2d_inv, rgb, opacities, tile_ranges, gaussian_indices,
                                          final_T, n_contrib, img_w, img_h, background, n_gaussians);
    return RasterizeBackwardOutput{b.dL_drgb, b.dL_dopacity_act, b.dL_dmeans_2d, b.dL_dcov_2d_inv};
}

inline ProjectionBackwardOutput project_backward(const torch::Tensor& dL_dmeans_2d, const torch::Tensor& dL_dcov_2d_inv,
                                                 const torch::Tensor& dL_drgb, const torch::Tensor& dL_dopacity_act,
                                                 const torch::Tensor& positions, const torch::Tensor& rotations,
                                                 const torch::Tensor& scales, const torch::Tensor& opacities,
                                                 const torch::Tensor& sh_coeffs, const torch::Tensor& radii,
                                                 const CameraInfo& camera, int active_sh_degree,
                                                 float scale_modifier = 1.0f) {
    auto p = cugs_hip::project_backward(dL_dmeans_2d, dL_dcov_2d_inv, dL_drgb, dL_dopacity_act, positions, rotations, scales,
                                        opacities, sh_coeffs, radii, to_pod(camera), active_sh_degree, scale_modifier);
    return ProjectionBackwardOutput{p.dL_dpositions, p.dL_drotations, p.dL_dscales, p.dL_dopacities, p.dL_dsh_coeffs};
}

inline ProjectionOutput project_gaussians(const torch::Tensor& positions, const torch::Tensor& rotations,
                                          const torch::Tensor& scales, const torch::Tensor& opacities,
                                          const torch::Tensor& sh_coeffs, const CameraInfo& camera, int active_sh_degree,
                                          float scale_modifier = 1.0f) {
    auto p = cugs_hip::project_gaussians(positions, rotations, scales, opacities, sh_coeffs, to_pod(camera),
                                         active_sh_degree, scale_modifier);
    return ProjectionOutput{p.means_2d, p.depths, p.cov_2d_inv, p.radii, p.tiles_touched, p.rgb, p.opacities_act};
}

inline SortingOutput sort_gaussians(const torch::Tensor& means_2d, const torch::Tensor& depths, const torch::Tensor& radii,
                                    const torch::Tensor& tiles_touched, int img_w, int img_h) {
    auto s = cugs_hip::sort_gaussians(means_2d, depths, radii, tiles_touched, img_w, img_h);
    return SortingOutput{s.gaussian_keys_sorted, s.gaussian_values_sorted, s.tile_ranges, s.total_pairs};
}

inline torch::Tensor evaluate_sh_cuda(int degree, const torch::Tensor& sh, const torch::Tensor& dirs) {
    return cugs_hip::evaluate_sh_cuda(degree, sh, dirs);
}
inline torch::Tensor evaluate_sh_backward_cuda(int degree, const torch::Tensor& sh, const torch::Tensor& dirs,
                                               const torch::Tensor& dL_dcolor) {
    return cugs_hip::evaluate_sh_backward_cuda(degree, sh, dirs, dL_dcolor);
}

}  // namespace cugs
