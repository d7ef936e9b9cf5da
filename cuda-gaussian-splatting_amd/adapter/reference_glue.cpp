// reference_glue.cpp — the translation unit a maintainer of Artemarius/cuda-gaussian-splatting adds to the
// `cugs_rasterizer` target in place of src/rasterizer/{projection,sorting,forward,backward,projection_backward}.cu,
// rasterizer.cpp and src/core/sh.cu, sh_backward.cu.  It DEFINES the reference's own entry
// points, non-inline, with exactly the signatures its headers declare, on top of libcugs_hip_torch.so - so
// apps/train_main.cpp, src/training/trainer.cpp, tests/test_rasterizer.cpp, tests/test_projection.cpp ... keep their
// calls and link against these symbols.  Default arguments are NOT repeated here: the reference's declarations carry
// them (projection.hpp:47, projection_backward.hpp:57), a second specification is ill-formed.
//
// Compile-checked in this repository against the reference's headers with a stand-in for Eigen
// (tests/test_reference_glue_compiles.py: -fsyntax-only, then a two-TU link of a declaration-only caller against
// this file's object); built for real only in the maintainer's tree.
#include "reference_glue.hpp"

#include "core/sh.hpp"
#include "core/sh_backward.hpp"

#include <type_traits>

namespace cugs {

RenderOutput render(const GaussianModel& model, const CameraInfo& camera, const RenderSettings& settings) {
    TORCH_CHECK(model.is_valid(), "GaussianModel is not valid");                           // rasterizer.cpp:27
    auto r = cugs_hip::render(tensors_of(model), to_pod(camera), settings_of(settings));
    glue_detail::packed_table().put(r.color, r.packed, r.zeroed_accum, r.colour_gate);
    return RenderOutput{r.color, r.final_T, r.n_contrib, r.means_2d, r.depths, r.cov_2d_inv, r.radii, r.rgb,
                        r.opacities_act, r.gaussian_indices, r.tile_ranges};
}

BackwardOutput render_backward(const torch::Tensor& dL_dcolor, const RenderOutput& ro, const GaussianModel& model,
                                      const CameraInfo& camera, const RenderSettings& settings) {
    cugs_hip::RenderOutput h{ro.color, ro.final_T, ro.n_contrib, ro.means_2d, ro.depths, ro.cov_2d_inv, ro.radii, ro.rgb,
                             ro.opacities_act, ro.gaussian_indices, ro.tile_ranges,
                             glue_detail::packed_table().get(ro.color, model.num_gaussians()),
                             glue_detail::packed_table().get_gate(ro.color, model.num_gaussians())};
    h.zeroed_accum = glue_detail::packed_table().take_accum(ro.color, model.num_gaussians());
    // the reference's struct has no field for the blend kernels' tile order either: one small launch rebuilds it
    h.tile_order = cugs_hip::tile_order_of(ro.tile_ranges, camera.width, camera.height);
    auto b = cugs_hip::render_backward(dL_dcolor, h, tensors_of(model), to_pod(camera), settings_of(settings));
    return BackwardOutput{b.dL_dpositions, b.dL_drotations, b.dL_dscales, b.dL_dopacities, b.dL_dsh_coeffs, b.dL_dmeans_2d};
}

// ---- stage functions, reference signatures (forward.hpp:41, backward.hpp:39, projection_backward.hpp:44) ----
ForwardOutput rasterize_forward(const torch::Tensor& means_2d, const torch::Tensor& cov_2d_inv,
                                       const torch::Tensor& rgb, const torch::Tensor& opacities,
                                       const torch::Tensor& tile_ranges, const torch::Tensor& gaussian_indices,
                                       int img_w, int img_h, const float background[3]) {
    auto f = cugs_hip::rasterize_forward(means_2d, cov_2d_inv, rgb, opacities, tile_ranges, gaussian_indices, img_w, img_h,
                                         background);
    return ForwardOutput{f.color, f.final_T, f.n_contrib};
}

RasterizeBackwardOutput rasterize_backward(const torch::Tensor& dL_dcolor, const torch::Tensor& means_2d,
                                                  const torch::Tensor& cov_2d_inv, const torch::Tensor& rgb,
                                                  const torch::Tensor& opacities, const torch::Tensor& tile_ranges,
                                                  const torch::Tensor& gaussian_indices, const torch::Tensor& final_T,
                                                  const torch::Tensor& n_contrib, int img_w, int img_h,
                                                  const float background[3], int n_gaussians) {
    auto b = cugs_hip::rasterize_backward(dL_dcolor, means_2d, cov_2d_inv, rgb, opacities, tile_ranges, gaussian_indices,
                                          final_T, n_contrib, img_w, img_h, background, n_gaussians);
    return RasterizeBackwardOutput{b.dL_drgb, b.dL_dopacity_act, b.dL_dmeans_2d, b.dL_dcov_2d_inv};
}

ProjectionBackwardOutput project_backward(const torch::Tensor& dL_dmeans_2d, const torch::Tensor& dL_dcov_2d_inv,
                                                 const torch::Tensor& dL_drgb, const torch::Tensor& dL_dopacity_act,
                                                 const torch::Tensor& positions, const torch::Tensor& rotations,
                                                 const torch::Tensor& scales, const torch::Tensor& opacities,
                                                 const torch::Tensor& sh_coeffs, const torch::Tensor& radii,
                                                 const CameraInfo& camera, int active_sh_degree,
                                                 float scale_modifier) {
    auto p = cugs_hip::project_backward(dL_dmeans_2d, dL_dcov_2d_inv, dL_drgb, dL_dopacity_act, positions, rotations, scales,
                                        opacities, sh_coeffs, radii, to_pod(camera), active_sh_degree, scale_modifier);
    return ProjectionBackwardOutput{p.dL_dpositions, p.dL_drotations, p.dL_dscales, p.dL_dopacities, p.dL_dsh_coeffs};
}

ProjectionOutput project_gaussians(const torch::Tensor& positions, const torch::Tensor& rotations,
                                          const torch::Tensor& scales, const torch::Tensor& opacities,
                                          const torch::Tensor& sh_coeffs, const CameraInfo& camera, int active_sh_degree,
                                          float scale_modifier) {
    auto p = cugs_hip::project_gaussians(positions, rotations, scales, opacities, sh_coeffs, to_pod(camera),
                                         active_sh_degree, scale_modifier);
    return ProjectionOutput{p.means_2d, p.depths, p.cov_2d_inv, p.radii, p.tiles_touched, p.rgb, p.opacities_act};
}

SortingOutput sort_gaussians(const torch::Tensor& means_2d, const torch::Tensor& depths, const torch::Tensor& radii,
                                    const torch::Tensor& tiles_touched, int img_w, int img_h) {
    auto s = cugs_hip::sort_gaussians(means_2d, depths, radii, tiles_touched, img_w, img_h);
    return SortingOutput{s.gaussian_keys_sorted, s.gaussian_values_sorted, s.tile_ranges, s.total_pairs};
}

torch::Tensor evaluate_sh_cuda(int degree, const torch::Tensor& sh, const torch::Tensor& dirs) {
    return cugs_hip::evaluate_sh_cuda(degree, sh, dirs);
}
torch::Tensor evaluate_sh_backward_cuda(int degree, const torch::Tensor& sh, const torch::Tensor& dirs,
                                               const torch::Tensor& dL_dcolor) {
    return cugs_hip::evaluate_sh_backward_cuda(degree, sh, dirs, dL_dcolor);
}

// The definitions above must BE the declared functions, not overloads beside them: taking the address of an
// overloaded name with a mismatching definition is ambiguous or fails the comparison below.
static_assert(std::is_same_v<decltype(&render),
                             RenderOutput (*)(const GaussianModel&, const CameraInfo&, const RenderSettings&)>);
static_assert(std::is_same_v<decltype(&render_backward),
                             BackwardOutput (*)(const torch::Tensor&, const RenderOutput&, const GaussianModel&,
                                                const CameraInfo&, const RenderSettings&)>);
static_assert(std::is_same_v<decltype(&project_gaussians),
                             ProjectionOutput (*)(const torch::Tensor&, const torch::Tensor&, const torch::Tensor&,
                                                  const torch::Tensor&, const torch::Tensor&, const CameraInfo&, int, float)>);
static_assert(std::is_same_v<decltype(&sort_gaussians),
                             SortingOutput (*)(const torch::Tensor&, const torch::Tensor&, const torch::Tensor&,
                                               const torch::Tensor&, int, int)>);
static_assert(std::is_same_v<decltype(&rasterize_forward),
                             ForwardOutput (*)(const torch::Tensor&, const torch::Tensor&, const torch::Tensor&,
                                               const torch::Tensor&, const torch::Tensor&, const torch::Tensor&, int, int,
                                               const float*)>);
static_assert(std::is_same_v<decltype(&rasterize_backward),
                             RasterizeBackwardOutput (*)(const torch::Tensor&, const torch::Tensor&, const torch::Tensor&,
                                                         const torch::Tensor&, const torch::Tensor&, const torch::Tensor&,
                                                         const torch::Tensor&, const torch::Tensor&, const torch::Tensor&,
                                                         int, int, const float*, int)>);
static_assert(std::is_same_v<decltype(&project_backward),
                             ProjectionBackwardOutput (*)(const torch::Tensor&, const torch::Tensor&, const torch::Tensor&,
                                                          const torch::Tensor&, const torch::Tensor&, const torch::Tensor&,
                                                          const torch::Tensor&, const torch::Tensor&, const torch::Tensor&,
                                                          const torch::Tensor&, const CameraInfo&, int, float)>);
static_assert(std::is_same_v<decltype(&evaluate_sh_cuda),
                             torch::Tensor (*)(int, const torch::Tensor&, const torch::Tensor&)>);
static_assert(std::is_same_v<decltype(&evaluate_sh_backward_cuda),
                             torch::Tensor (*)(int, const torch::Tensor&, const torch::Tensor&, const torch::Tensor&)>);

}  // namespace cugs
