// cugs_hip_torch.hpp — libtorch layer over the C ABI (include/cugs_hip.h).
//
// C++ host side of the drop-in: the reference's stage functions and render()/render_backward()
// with torch::Tensor arguments and results (same names, shapes, dtypes and zero/empty-case
// behaviour as namespace cugs), the camera as the POD `cugs_camera`.  It depends on libtorch
// only; `reference_glue.hpp` adds the few lines that map cugs::CameraInfo / cugs::GaussianModel
// (Eigen + the reference's headers) onto it.  Errors: non-zero C-ABI codes become
// std::runtime_error like CUDA_CHECK (utils/cuda_utils.cuh:12-20); argument checks are TORCH_CHECK.
#pragma once

#include <torch/torch.h>

#include <atomic>
#include <memory>

#include <array>
#include <string>

#include "../../include/cugs_hip.h"

namespace cugs_hip {

struct ProjectionOutput {          // rasterizer/projection.hpp
    torch::Tensor means_2d, depths, cov_2d_inv, radii, tiles_touched, rgb, opacities_act;
    torch::Tensor packed;          // [N,12] scratch for the blend kernels (not in the reference)
    torch::Tensor colour_gate;     // [N] uint8: ReLU gate bits of the SH backward (cugs_hip.h; not in the reference)
};
struct SortingOutput {             // rasterizer/sorting.hpp:18-24
    torch::Tensor gaussian_keys_sorted, gaussian_values_sorted, tile_ranges;
    int total_pairs = 0;
    torch::Tensor tile_order;      // [tiles,4] int32, optional (not in the reference): {tile, first, end, 0}, longest list first (cugs_tile_order)
};
struct ForwardOutput { torch::Tensor color, final_T, n_contrib; };
struct RasterizeBackwardOutput {
    torch::Tensor dL_drgb, dL_dopacity_act, dL_dmeans_2d, dL_dcov_2d_inv;
    torch::Tensor grad_accum;      // [N,16] packed rows (not in the reference)
};
struct ProjectionBackwardOutput { torch::Tensor dL_dpositions, dL_drotations, dL_dscales, dL_dopacities, dL_dsh_coeffs; };

struct ModelTensors {              // the five tensors of cugs::GaussianModel (core/gaussian.hpp:34-40)
    torch::Tensor positions, sh_coeffs, opacities, rotations, scales;
};
struct RenderSettings { float background[3] = {0.f, 0.f, 0.f}; int active_sh_degree = 3; float scale_modifier = 1.f; };
struct RenderOutput {              // rasterizer/rasterizer.hpp:27-46
    torch::Tensor color, final_T, n_contrib, means_2d, depths, cov_2d_inv, radii, rgb, opacities_act,
        gaussian_indices, tile_ranges;
    torch::Tensor packed;
    torch::Tensor colour_gate;     // from the projection; undefined = render_backward recomputes the gate from the coefficients
    // [N, 16] accumulator of the blend backward, already cleared by the forward blend (which leaves HBM idle); handed
    // to ONE render_backward, which takes it out of the struct (hence mutable).  COPIES of a RenderOutput share the
    // buffer; `accum_used` (shared by the copies) makes "one" hold across them: the second backward - through whichever
    // copy - finds the flag set and fills a fresh accumulator instead of adding onto the first one's rows.
    mutable torch::Tensor zeroed_accum;
    std::shared_ptr<std::atomic<bool>> accum_used;
    torch::Tensor tile_order;      // [tiles,4] int32: the order the blend kernels' workgroups take the tiles in (undefined: spatial)
};
struct BackwardOutput { torch::Tensor dL_dpositions, dL_drotations, dL_dscales, dL_dopacities, dL_dsh_coeffs, dL_dmeans_2d; };

ProjectionOutput project_gaussians(const torch::Tensor& positions, const torch::Tensor& rotations,
                                   const torch::Tensor& scales, const torch::Tensor& opacities,
                                   const torch::Tensor& sh_coeffs, const cugs_camera& camera,
                                   int active_sh_degree, float scale_modifier = 1.0f);
SortingOutput sort_gaussians(const torch::Tensor& means_2d, const torch::Tensor& depths, const torch::Tensor& radii,
                             const torch::Tensor& tiles_touched, int img_w, int img_h);
ForwardOutput rasterize_forward(const torch::Tensor& means_2d, const torch::Tensor& cov_2d_inv,
                                const torch::Tensor& rgb, const torch::Tensor& opacities,
                                const torch::Tensor& tile_ranges, const torch::Tensor& gaussian_indices,
                                int img_w, int img_h, const float background[3],
                                const torch::Tensor& packed = {}, const torch::Tensor& zero_buf = {},
                                const torch::Tensor& tile_order = {});
// the tiles ordered by the length of their lists, longest first, from any valid tile_ranges (cugs_tile_order)
torch::Tensor tile_order_of(const torch::Tensor& tile_ranges, int img_w, int img_h);
RasterizeBackwardOutput rasterize_backward(const torch::Tensor& dL_dcolor, const torch::Tensor& means_2d,
                                           const torch::Tensor& cov_2d_inv, const torch::Tensor& rgb,
                                           const torch::Tensor& opacities, const torch::Tensor& tile_ranges,
                                           const torch::Tensor& gaussian_indices, const torch::Tensor& final_T,
                                           const torch::Tensor& n_contrib, int img_w, int img_h,
                                           const float background[3], int n_gaussians,
                                           const torch::Tensor& packed = {}, bool unpack = true,
                                           const torch::Tensor& zeroed_accum = {}, const torch::Tensor& tile_order = {});
ProjectionBackwardOutput project_backward(const torch::Tensor& dL_dmeans_2d, const torch::Tensor& dL_dcov_2d_inv,
                                          const torch::Tensor& dL_drgb, const torch::Tensor& dL_dopacity_act,
                                          const torch::Tensor& positions, const torch::Tensor& rotations,
                                          const torch::Tensor& scales, const torch::Tensor& opacities,
                                          const torch::Tensor& sh_coeffs, const torch::Tensor& radii,
                                          const cugs_camera& camera, int active_sh_degree,
                                          float scale_modifier = 1.0f);
torch::Tensor evaluate_sh_cuda(int degree, const torch::Tensor& sh_coeffs, const torch::Tensor& directions);
torch::Tensor evaluate_sh_backward_cuda(int degree, const torch::Tensor& sh_coeffs, const torch::Tensor& directions,
                                        const torch::Tensor& dL_dcolor);

// `for_backward` = false (evaluation, viewer; not in the reference): no accumulator is prepared for a backward pass.
RenderOutput render(const ModelTensors& model, const cugs_camera& camera, const RenderSettings& settings,
                    bool for_backward = true);
class FusedAdam;
// `fused` (optional, not in the reference; single-GPU training): the projection backward applies the optimizer step to
// the model in place (cugs_project_backward_adam) and the five parameter gradients are never materialised (undefined
// in the result; dL_dmeans_2d is returned) - bit for bit render_backward + apply_gradients + step.
BackwardOutput render_backward(const torch::Tensor& dL_dcolor, const RenderOutput& render_out,
                               const ModelTensors& model, const cugs_camera& camera, const RenderSettings& settings,
                               FusedAdam* fused = nullptr);

// training/loss.hpp:21-52 + the autograd step of trainer.cpp:214-217 in two launches.
// Scalars are 0-dim device tensors, as in the reference.
struct LossAndGrad { torch::Tensor loss, l1, ssim_mean, dL_dcolor; };
LossAndGrad combined_loss_and_grad(const torch::Tensor& rendered, const torch::Tensor& target, float lambda_ = 0.2f,
                                   bool want_grad = true);
torch::Tensor combined_loss(const torch::Tensor& rendered, const torch::Tensor& target, float lambda_ = 0.2f);
torch::Tensor ssim(const torch::Tensor& rendered, const torch::Tensor& target, int window_size = 11);

// optimizer/fused_adam.hpp:29-106 on raw tensors (group order: positions, sh, opacities, scales, rotations)
struct AdamHyper { float beta1 = 0.9f, beta2 = 0.999f, eps = 1e-15f; };
class FusedAdam {
public:
    FusedAdam(std::array<torch::Tensor, 5> params, std::array<float, 5> lrs, AdamHyper h = {});
    void apply_gradients(const BackwardOutput& grads);
    void zero_grad();
    void set_lr(int group, float lr) { lrs_[group] = lr; }
    float get_lr(int group) const { return lrs_[group]; }
    void step();
    // The optimizer half of cugs_project_backward_adam: counts the step, returns moments / learning rates / bias
    // corrections for the launch that updates the parameters in place (they must be contiguous float32).
    cugs_adam_fused begin_fused_step();
    const std::array<torch::Tensor, 5>& params() const { return params_; }
private:
    friend class DensificationController;      // carries m_/v_ through clone/split/prune
    friend bool write_gaussian_ply(const std::string&, const ModelTensors&, const FusedAdam*);
    friend ModelTensors read_gaussian_ply(const std::string&, const torch::Device&, FusedAdam*);
    std::array<torch::Tensor, 5> params_, m_, v_, grads_;
    std::array<float, 5> lrs_;
    AdamHyper h_;
    int step_count_ = 0;
};

// optimizer/densification.hpp:23-168 over csrc/densify.hip.  Same schedule, thresholds and
// result order as the reference; the split noise is an argument ([2, N, 3] standard normal, drawn from the
// device generator when undefined) and the VRAM guards are not mirrored.
struct DensificationConfig {
    int densify_from = 500, densify_until = 15000, densify_every = 100, opacity_reset_every = 3000;
    float grad_threshold = 0.0002f, opacity_threshold = 0.005f, percent_dense = 0.01f;
    int max_screen_size = 20, max_gaussians = 0;
};
struct DensificationStats { int num_cloned = 0, num_split = 0, num_pruned = 0, num_before = 0, num_after = 0; };
class DensificationController {
public:
    DensificationController(const DensificationConfig& config, float scene_extent)
        : config_(config), scene_extent_(scene_extent) {}
    void accumulate_gradients(const torch::Tensor& dL_dmeans_2d, const torch::Tensor& radii);
    bool should_densify(int step) const;
    bool should_reset_opacity(int step) const;
    // `optimizer` (optional): its parameter tensors are re-pointed at the new model and its moments carried over
    // (survivors keep theirs, new Gaussians start at zero) instead of the reference's optimizer rebuild.
    DensificationStats densify(ModelTensors& model, int step, const torch::Tensor& noise = {},
                               FusedAdam* optimizer = nullptr);
    void reset_opacity(ModelTensors& model);
private:
    void reset_accumulators(int64_t n, const torch::Device& device);
    DensificationConfig config_;
    float scene_extent_;
    torch::Tensor grad_accum_, grad_count_, max_radii_2d_;
};

// the float [height, width, 3] training target from a device-resident uint8 [h, w, 3] view - x 1/255
// and, if the sizes differ, the reference's resize_image (data/image_io.cpp:35-39, 47-100), bit for bit what
// trainer.cpp:186-198 builds on the CPU and uploads.
torch::Tensor image_to_float(const torch::Tensor& view_u8, int width, int height);

// utils/ply_io.hpp:53-64 over csrc/ply.hip: the reference's binary PLY layout, records packed and
// unpacked on the device.  `optimizer` (optional): the Adam moments and step count ride along as extra properties
// m_*, v_* and a header comment (the reference's reader skips both); read_gaussian_ply restores them into
// `optimizer` when the file has them.  Errors as in the reference: write returns false, read throws.
bool write_gaussian_ply(const std::string& path, const ModelTensors& model, const FusedAdam* optimizer = nullptr);
ModelTensors read_gaussian_ply(const std::string& path, const torch::Device& device = torch::kCPU,
                               FusedAdam* optimizer = nullptr);

}  // namespace cugs_hip
