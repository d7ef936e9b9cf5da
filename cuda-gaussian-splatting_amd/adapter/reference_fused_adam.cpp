// reference_fused_adam.cpp — drop-in replacement for src/optimizer/fused_adam.cu of
// Artemarius/cuda-gaussian-splatting: the six member functions of cugs::FusedAdam (+ its private
// launch_kernel) over the C ABI of libcugs_hip.so, with the class declaration
// (src/optimizer/fused_adam.hpp:29-106) left exactly as it is - same members, same semantics:
//   * the constructor turns on requires_grad and zero-initialises the moments (fused_adam.cu:82-111);
//   * apply_gradients stores references (:113-120); update_lr applies position_lr (:122-124);
//   * step() computes the bias corrections in double on the host (:145-148) and skips groups whose
//     gradient is undefined (:156); here ALL defined groups go into ONE launch (cugs_fused_adam_groups)
//     instead of one kernel per group;
//   * the optimizer keeps a REFERENCE to the model and must be rebuilt when N changes, as
//     trainer.cpp:283 already does after densification.
// tests/test_fused_adam.cpp constructs FusedAdam(model, config) directly and keeps working unchanged.
//
// Status: compile- and link-checked in this repository's image against the reference's headers with a test-only
// stand-in for Eigen (fused_adam.hpp pulls in core/types.hpp -> Eigen3, which is absent):
// tests/test_reference_glue_compiles.py.  Built for real only in the maintainer's tree.
// The arithmetic underneath (cugs_fused_adam_groups) is what tests/test_gpu_parity.py and
// tests/test_gpu_configs.py check bit for bit.
#include "optimizer/fused_adam.hpp"

#include <c10/hip/HIPStream.h>

#include <cmath>
#include <stdexcept>
#include <string>

#include "cugs_hip.h"

namespace cugs {

namespace {
void check_cugs(int code, const char* what) {                       // CUDA_CHECK's behaviour (utils/cuda_utils.cuh:12-20)
    if (code != 0)
        throw std::runtime_error(std::string("HIP error in ") + what + " — " + cugs_error_string(code));
}
void* current_stream(const torch::Tensor& t) {
    return static_cast<void*>(c10::hip::getCurrentHIPStream(t.device().index()).stream());
}
}  // namespace

FusedAdam::FusedAdam(GaussianModel& model, const AdamConfig& config)
    : model_(model), config_(config), step_count_(0) {
    model_.positions.requires_grad_(true);
    model_.sh_coeffs.requires_grad_(true);
    model_.opacities.requires_grad_(true);
    model_.scales.requires_grad_(true);
    model_.rotations.requires_grad_(true);
    torch::Tensor* params[kNumGroups] = {&model_.positions, &model_.sh_coeffs, &model_.opacities, &model_.scales,
                                         &model_.rotations};                  // ParamGroup order
    for (int i = 0; i < kNumGroups; ++i) {
        m_[i] = torch::zeros_like(*params[i]);
        v_[i] = torch::zeros_like(*params[i]);
        grads_[i] = torch::Tensor();
    }
    learning_rates_[0] = config.position_lr_config.lr_init;
    learning_rates_[1] = config.lr_sh_coeffs;
    learning_rates_[2] = config.lr_opacities;
    learning_rates_[3] = config.lr_scales;
    learning_rates_[4] = config.lr_rotations;
}

void FusedAdam::apply_gradients(const BackwardOutput& grads) {
    grads_[0] = grads.dL_dpositions;
    grads_[1] = grads.dL_dsh_coeffs;
    grads_[2] = grads.dL_dopacities;
    grads_[3] = grads.dL_dscales;
    grads_[4] = grads.dL_drotations;
}

void FusedAdam::update_lr(int step) { learning_rates_[0] = position_lr(step, config_.position_lr_config); }

void FusedAdam::zero_grad() {
    for (int i = 0; i < kNumGroups; ++i) grads_[i] = torch::Tensor();
    if (model_.positions.grad().defined()) model_.positions.mutable_grad().zero_();
    if (model_.sh_coeffs.grad().defined()) model_.sh_coeffs.mutable_grad().zero_();
    if (model_.opacities.grad().defined()) model_.opacities.mutable_grad().zero_();
    if (model_.scales.grad().defined()) model_.scales.mutable_grad().zero_();
    if (model_.rotations.grad().defined()) model_.rotations.mutable_grad().zero_();
}

void FusedAdam::step() {
    step_count_++;
    float bc1 = 0.0f, bc2 = 0.0f;                                   // 1/(1-beta^t) in double, then float (:145-148,161-162)
    cugs_adam_bias_correction(config_.beta1, config_.beta2, step_count_, &bc1, &bc2);
    torch::Tensor* params[kNumGroups] = {&model_.positions, &model_.sh_coeffs, &model_.opacities, &model_.scales,
                                         &model_.rotations};
    cugs_adam_group groups[kNumGroups];
    torch::Tensor pc[kNumGroups], gc[kNumGroups], mc[kNumGroups], vc[kNumGroups];   // contiguous views kept alive
    void* stream = nullptr;
    torch::NoGradGuard no_grad;                                      // the parameters carry requires_grad
    for (int i = 0; i < kNumGroups; ++i) {
        groups[i] = cugs_adam_group{nullptr, nullptr, nullptr, nullptr, 0, learning_rates_[i], 0.0f};
        if (!grads_[i].defined()) continue;                          // :156
        TORCH_CHECK(params[i]->is_cuda(), "FusedAdam: param must be on CUDA");
        TORCH_CHECK(grads_[i].is_cuda(), "FusedAdam: grad must be on CUDA");
        TORCH_CHECK(params[i]->numel() == grads_[i].numel(), "FusedAdam: param/grad size mismatch: ",
                    params[i]->numel(), " vs ", grads_[i].numel());
        pc[i] = params[i]->detach().contiguous();
        gc[i] = grads_[i].contiguous();
        mc[i] = m_[i].contiguous();
        vc[i] = v_[i].contiguous();
        groups[i].param = pc[i].data_ptr<float>();
        groups[i].grad = gc[i].data_ptr<float>();
        groups[i].m = mc[i].data_ptr<float>();
        groups[i].v = vc[i].data_ptr<float>();
        groups[i].n = pc[i].numel();
        stream = current_stream(*params[i]);
    }
    check_cugs(cugs_fused_adam_groups(groups, kNumGroups, config_.beta1, config_.beta2, config_.eps, bc1, bc2, stream),
               "cugs_fused_adam_groups");
    for (int i = 0; i < kNumGroups; ++i) {                           // write-back if contiguous() copied (:216-218)
        if (!grads_[i].defined()) continue;
        if (!params[i]->is_contiguous()) params[i]->detach().copy_(pc[i]);
        if (!m_[i].is_contiguous()) m_[i].copy_(mc[i]);
        if (!v_[i].is_contiguous()) v_[i].copy_(vc[i]);
    }
}

float FusedAdam::get_lr(ParamGroup group) const { return learning_rates_[static_cast<int>(group)]; }

// The private per-group launcher of the reference (fused_adam.hpp:78-85) is still declared in the header, so it
// stays defined; step() above does not use it.
void FusedAdam::launch_kernel(torch::Tensor& param, const torch::Tensor& grad, torch::Tensor& m, torch::Tensor& v,
                              float lr, float bc1, float bc2) {
    TORCH_CHECK(param.is_cuda(), "FusedAdam: param must be on CUDA");
    TORCH_CHECK(grad.is_cuda(), "FusedAdam: grad must be on CUDA");
    TORCH_CHECK(param.numel() == grad.numel(), "FusedAdam: param/grad size mismatch: ", param.numel(), " vs ",
                grad.numel());
    torch::NoGradGuard no_grad;
    auto param_c = param.detach().contiguous();
    auto grad_c = grad.contiguous();
    auto m_c = m.contiguous();
    auto v_c = v.contiguous();
    if (param_c.numel() == 0) return;
    check_cugs(cugs_fused_adam(param_c.data_ptr<float>(), grad_c.data_ptr<float>(), m_c.data_ptr<float>(),
                               v_c.data_ptr<float>(), param_c.numel(), lr, config_.beta1, config_.beta2, config_.eps, bc1,
                               bc2, current_stream(param)),
               "cugs_fused_adam");
    if (!param.is_contiguous()) param.detach().copy_(param_c);
    if (!m.is_contiguous()) m.copy_(m_c);
    if (!v.is_contiguous()) v.copy_(v_c);
}

}  // namespace cugs
