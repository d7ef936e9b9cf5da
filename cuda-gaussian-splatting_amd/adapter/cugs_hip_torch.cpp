// cugs_hip_torch.cpp — see cugs_hip_torch.hpp.  Each function: validate like the reference's
// launcher, allocate outputs with torch (the C ABI allocates nothing), pass raw pointers and
// torch's current HIP stream, turn a non-zero return into std::runtime_error.
#include "cugs_hip_torch.hpp"

#include <c10/hip/HIPFunctions.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

namespace cugs_hip {
namespace {

void check(int code, const char* what) {
    if (code != 0)
        throw std::runtime_error(std::string("HIP error in ") + what + " - " + cugs_error_string(code));
}
void* stream_of(const torch::Tensor& t) {
    return static_cast<void*>(c10::hip::getCurrentHIPStream(t.device().index()).stream());
}
torch::Tensor f32c(const torch::Tensor& t) { return t.contiguous().to(torch::kFloat32); }   // projection.cu:240-243
template <typename T> T* ptr(const torch::Tensor& t) {
    return (t.defined() && t.numel() > 0) ? t.data_ptr<T>() : nullptr;
}
torch::TensorOptions fopt(const torch::Tensor& like) { return torch::TensorOptions().dtype(torch::kFloat32).device(like.device()); }
torch::TensorOptions iopt(const torch::Tensor& like) { return torch::TensorOptions().dtype(torch::kInt32).device(like.device()); }

// Returns a handle BY VALUE: a reference into the pool would dangle when a later call grows the vector.
// Scratch is per (device, CURRENT STREAM, kind): two renders queued on two streams of one device run concurrently and
// must not share a sort workspace (cugs_project_forward_keyed writes the sort keys into it).
torch::Tensor workspace(const torch::Device& dev, size_t bytes, int kind) {      // grow-only
    // intentionally leaked: device tensors must not be destroyed during static destruction, after the
    // HIP caching allocator has gone away
    static thread_local auto& pool = *new std::vector<std::tuple<torch::Device, void*, int, torch::Tensor>>();
    void* st = static_cast<void*>(c10::hip::getCurrentHIPStream(dev.index()).stream());
    auto make = [&] { return torch::empty({static_cast<int64_t>(bytes + bytes / 4 + 4096)},
                                          torch::TensorOptions().dtype(torch::kUInt8).device(dev)); };
    for (auto& e : pool)
        if (std::get<0>(e) == dev && std::get<1>(e) == st && std::get<2>(e) == kind) {
            if (static_cast<size_t>(std::get<3>(e).numel()) < bytes) std::get<3>(e) = make();
            return std::get<3>(e);
        }
    pool.emplace_back(dev, st, kind, make());
    return std::get<3>(pool.back());
}

// Host-side state of render()'s predicted sort, one per (device, stream): the running pair-count estimate, the held
// capacity, the pinned word the count lands in, and how many more sorts stay on the general depth route after one
// reported a depth outside the three-pass range (-1): sticky, so that such a view does not pay a wasted sort + blend
// and a blocking re-sort on every frame; probed again after kWideDepthHold sorts.
struct SortState {
    int64_t last_pairs = -1;
    int64_t held_cap = 0;
    int wide_left = 0;
    torch::Tensor pinned;
};
constexpr int kWideDepthHold = 256;
constexpr int64_t kTileOrderMinPairs = 2000000;
SortState& sort_state(const torch::Device& dev) {
    static thread_local auto& states = *new std::map<std::pair<int, void*>, SortState>();
    return states[{dev.index(), static_cast<void*>(c10::hip::getCurrentHIPStream(dev.index()).stream())}];
}

// key_sort: cugs_project_forward_keyed - the kernel also leaves the sort's depth keys and tile rectangles in this
// device's N-level sort workspace; the caller's next sort on the stream must be cugs_sort_pairs_predicted_keyed
ProjectionOutput project_impl(const torch::Tensor& positions, const torch::Tensor& rotations,
                              const torch::Tensor& scales, const torch::Tensor& opacities,
                              const torch::Tensor& sh_coeffs, const cugs_camera& camera,
                              int active_sh_degree, float scale_modifier, bool key_sort) {
    TORCH_CHECK(positions.is_cuda(), "positions must be on CUDA");
    TORCH_CHECK(positions.dim() == 2 && positions.size(1) == 3);
    const int64_t n = positions.size(0);
    ProjectionOutput o;
    o.means_2d = torch::empty({n, 2}, fopt(positions));
    o.depths = torch::empty({n}, fopt(positions));
    o.cov_2d_inv = torch::empty({n, 3}, fopt(positions));
    o.radii = torch::empty({n}, iopt(positions));
    o.tiles_touched = torch::empty({n}, iopt(positions));
    o.opacities_act = torch::empty({n}, fopt(positions));
    o.rgb = torch::empty({n, 3}, fopt(positions));
    o.packed = torch::empty({n, CUGS_PACKED_STRIDE}, fopt(positions));
    o.colour_gate = torch::empty({n}, torch::TensorOptions().dtype(torch::kUInt8).device(positions.device()));
    if (n == 0) return o;
    auto pos = f32c(positions), rot = f32c(rotations), scl = f32c(scales), opa = f32c(opacities), sh = f32c(sh_coeffs);
    TORCH_CHECK(sh.dim() == 3 && sh.size(0) == n && sh.size(1) == 3, "sh_coeffs must be [N, 3, C]");
    if (key_sort) {
        auto ws = workspace(positions.device(), cugs_sort_workspace_bytes(n), 0);
        check(cugs_project_forward_keyed(n, static_cast<int>(sh.size(2)), active_sh_degree, ptr<float>(pos), ptr<float>(rot),
                                         ptr<float>(scl), ptr<float>(opa), ptr<float>(sh), &camera, scale_modifier,
                                         ptr<float>(o.means_2d), ptr<float>(o.depths), ptr<float>(o.cov_2d_inv),
                                         ptr<int32_t>(o.radii), ptr<int32_t>(o.tiles_touched), ptr<float>(o.opacities_act),
                                         ptr<float>(o.rgb), ptr<float>(o.packed), ptr<uint8_t>(o.colour_gate), ws.data_ptr(),
                                         ws.numel(), stream_of(positions)),
              "cugs_project_forward_keyed");
        return o;
    }
    check(cugs_project_forward(n, static_cast<int>(sh.size(2)), active_sh_degree, ptr<float>(pos), ptr<float>(rot),
                               ptr<float>(scl), ptr<float>(opa), ptr<float>(sh), &camera, scale_modifier,
                               ptr<float>(o.means_2d), ptr<float>(o.depths), ptr<float>(o.cov_2d_inv),
                               ptr<int32_t>(o.radii), ptr<int32_t>(o.tiles_touched), ptr<float>(o.opacities_act),
                               ptr<float>(o.rgb), ptr<float>(o.packed), ptr<uint8_t>(o.colour_gate), stream_of(positions)),
          "cugs_project_forward");
    return o;
}

}  // namespace

ProjectionOutput project_gaussians(const torch::Tensor& positions, const torch::Tensor& rotations,
                                   const torch::Tensor& scales, const torch::Tensor& opacities,
                                   const torch::Tensor& sh_coeffs, const cugs_camera& camera,
                                   int active_sh_degree, float scale_modifier) {
    return project_impl(positions, rotations, scales, opacities, sh_coeffs, camera, active_sh_degree, scale_modifier, false);
}

namespace {
SortingOutput sort_gaussians_impl(const torch::Tensor& means_2d, const torch::Tensor& depths, const torch::Tensor& radii,
                                  const torch::Tensor& tiles_touched, int img_w, int img_h, bool wide_depth) {
    TORCH_CHECK(means_2d.is_cuda(), "means_2d must be on CUDA");
    const int64_t n = means_2d.size(0);
    const int num_tiles = ((img_w + CUGS_TILE - 1) / CUGS_TILE) * ((img_h + CUGS_TILE - 1) / CUGS_TILE);
    SortingOutput o;
    o.tile_ranges = torch::empty({num_tiles, 2}, iopt(means_2d));
    void* st = stream_of(means_2d);
    auto tiles = tiles_touched.contiguous().to(torch::kInt32);
    auto m = means_2d.contiguous(), d = depths.contiguous(), r = radii.contiguous();
    auto ws = workspace(means_2d.device(), cugs_sort_workspace_bytes(n), 0);
    int64_t total = 0;
    if (n > 0)      // wide_depth: the general depth route at once (the caller has seen this view report -1)
        check((wide_depth ? cugs_sort_count_pairs_wide : cugs_sort_count_pairs)(
                  n, ptr<float>(m), ptr<float>(d), ptr<int32_t>(r), ptr<int32_t>(tiles), img_w, img_h, ws.data_ptr(),
                  ws.numel(), &total, st), "cugs_sort_count_pairs");
    o.total_pairs = static_cast<int>(total);
    o.gaussian_keys_sorted = torch::empty({total}, torch::TensorOptions().dtype(torch::kInt64).device(means_2d.device()));
    o.gaussian_values_sorted = torch::empty({total}, iopt(means_2d));
    if (num_tiles > 0) {
        auto wp = workspace(means_2d.device(), cugs_sort_pair_workspace_bytes(total), 1);
        check(cugs_sort_pairs(n, total, ptr<float>(m), ptr<float>(d), ptr<int32_t>(r), ptr<int32_t>(tiles), img_w, img_h,
                              ws.data_ptr(), ws.numel(), wp.data_ptr(), wp.numel(),
                              reinterpret_cast<uint64_t*>(ptr<int64_t>(o.gaussian_keys_sorted)),
                              ptr<int32_t>(o.gaussian_values_sorted), ptr<int32_t>(o.tile_ranges), st),
              "cugs_sort_pairs");
    }
    return o;
}
}  // namespace

SortingOutput sort_gaussians(const torch::Tensor& means_2d, const torch::Tensor& depths, const torch::Tensor& radii,
                             const torch::Tensor& tiles_touched, int img_w, int img_h) {
    return sort_gaussians_impl(means_2d, depths, radii, tiles_touched, img_w, img_h, false);
}

ForwardOutput rasterize_forward(const torch::Tensor& means_2d, const torch::Tensor& cov_2d_inv, const torch::Tensor& rgb,
                                const torch::Tensor& opacities, const torch::Tensor& tile_ranges,
                                const torch::Tensor& gaussian_indices, int img_w, int img_h,
                                const float background[3], const torch::Tensor& packed, const torch::Tensor& zero_buf,
                                const torch::Tensor& tile_order) {
    TORCH_CHECK(means_2d.is_cuda(), "means_2d must be on CUDA");
    ForwardOutput o;
    o.color = torch::empty({img_h, img_w, 3}, fopt(means_2d));
    o.final_T = torch::empty({img_h, img_w}, fopt(means_2d));
    o.n_contrib = torch::empty({img_h, img_w}, iopt(means_2d));
    if (img_w == 0 || img_h == 0) {
        if (zero_buf.defined()) zero_buf.zero_();             // the promise holds without a blend launch too
        return o;
    }
    auto m = means_2d.contiguous(), c = cov_2d_inv.contiguous(), r = rgb.contiguous(), op = opacities.contiguous();
    auto tr = tile_ranges.contiguous(), gi = gaussian_indices.contiguous();
    if (zero_buf.defined())                                     // the blend also clears the backward's accumulator
        TORCH_CHECK(zero_buf.is_contiguous() && zero_buf.scalar_type() == torch::kFloat32 && zero_buf.numel() % 4 == 0,
                    "zero_buf must be a contiguous float32 tensor of a multiple of four elements");
    if (tile_order.defined()) {                                 // workgroups handed out longest tile list first
        TORCH_CHECK(tile_order.is_contiguous() && tile_order.scalar_type() == torch::kInt32 &&
                    tile_order.numel() == 4 * tr.size(0), "tile_order must be a contiguous [tiles, 4] int32 tensor");
        check(cugs_rasterize_forward_ordered(img_w, img_h, background, ptr<int32_t>(tr), ptr<int32_t>(gi), ptr<float>(m),
                                             ptr<float>(c), ptr<float>(r), ptr<float>(op), ptr<float>(packed), ptr<float>(o.color),
                                             ptr<float>(o.final_T), ptr<int32_t>(o.n_contrib),
                                             zero_buf.defined() ? zero_buf.data_ptr() : nullptr,
                                             zero_buf.defined() ? static_cast<size_t>(zero_buf.numel()) * sizeof(float) : 0,
                                             reinterpret_cast<const uint32_t*>(tile_order.data_ptr<int32_t>()),
                                             stream_of(means_2d)),
              "cugs_rasterize_forward_ordered");
        return o;
    }
    if (zero_buf.defined()) {
        check(cugs_rasterize_forward_zero(img_w, img_h, background, ptr<int32_t>(tr), ptr<int32_t>(gi), ptr<float>(m),
                                          ptr<float>(c), ptr<float>(r), ptr<float>(op), ptr<float>(packed), ptr<float>(o.color),
                                          ptr<float>(o.final_T), ptr<int32_t>(o.n_contrib), zero_buf.data_ptr(),
                                          static_cast<size_t>(zero_buf.numel()) * sizeof(float), stream_of(means_2d)),
              "cugs_rasterize_forward_zero");
        return o;
    }
    check(cugs_rasterize_forward(img_w, img_h, background, ptr<int32_t>(tr), ptr<int32_t>(gi), ptr<float>(m), ptr<float>(c),
                                 ptr<float>(r), ptr<float>(op), ptr<float>(packed), ptr<float>(o.color),
                                 ptr<float>(o.final_T), ptr<int32_t>(o.n_contrib), stream_of(means_2d)),
          "cugs_rasterize_forward");
    return o;
}

torch::Tensor tile_order_of(const torch::Tensor& tile_ranges, int img_w, int img_h) {
    auto tr = tile_ranges.contiguous();
    auto order = torch::empty({tr.size(0), 4}, iopt(tile_ranges));     // {tile, first pair, one past the last, 0}
    if (tr.size(0) > 0)
        check(cugs_tile_order(img_w, img_h, ptr<int32_t>(tr), reinterpret_cast<uint32_t*>(order.data_ptr<int32_t>()),
                              stream_of(tile_ranges)), "cugs_tile_order");
    return order;
}

RasterizeBackwardOutput rasterize_backward(const torch::Tensor& dL_dcolor, const torch::Tensor& means_2d,
                                           const torch::Tensor& cov_2d_inv, const torch::Tensor& rgb,
                                           const torch::Tensor& opacities, const torch::Tensor& tile_ranges,
                                           const torch::Tensor& gaussian_indices, const torch::Tensor& final_T,
                                           const torch::Tensor& n_contrib, int img_w, int img_h,
                                           const float background[3], int n_gaussians, const torch::Tensor& packed,
                                           bool unpack, const torch::Tensor& zeroed_accum, const torch::Tensor& tile_order) {
    TORCH_CHECK(dL_dcolor.is_cuda(), "dL_dcolor must be on CUDA");
    const int64_t n = n_gaussians;
    RasterizeBackwardOutput o;
    const bool prezeroed = zeroed_accum.defined();
    if (prezeroed)
        TORCH_CHECK(zeroed_accum.is_contiguous() && zeroed_accum.dim() == 2 && zeroed_accum.size(0) == n &&
                    zeroed_accum.size(1) == CUGS_GRAD_STRIDE, "zeroed_accum must be a contiguous [N, 16] float32 tensor");
    o.grad_accum = prezeroed ? zeroed_accum : torch::empty({n, CUGS_GRAD_STRIDE}, fopt(dL_dcolor));
    if (unpack) {
        o.dL_drgb = torch::empty({n, 3}, fopt(dL_dcolor));
        o.dL_dopacity_act = torch::empty({n}, fopt(dL_dcolor));
        o.dL_dmeans_2d = torch::empty({n, 2}, fopt(dL_dcolor));
        o.dL_dcov_2d_inv = torch::empty({n, 3}, fopt(dL_dcolor));
    }
    if (n == 0) return o;
    auto g = dL_dcolor.contiguous(), m = means_2d.contiguous(), c = cov_2d_inv.contiguous(), r = rgb.contiguous();
    auto op = opacities.contiguous(), tr = tile_ranges.contiguous(), gi = gaussian_indices.contiguous();
    auto ft = final_T.contiguous(), nc = n_contrib.contiguous();
    if (tile_order.defined()) {
        TORCH_CHECK(tile_order.is_contiguous() && tile_order.scalar_type() == torch::kInt32 &&
                    tile_order.numel() == 4 * tr.size(0), "tile_order must be a contiguous [tiles, 4] int32 tensor");
        check(cugs_rasterize_backward_ordered(img_w, img_h, background, ptr<int32_t>(tr), ptr<int32_t>(gi), ptr<float>(m),
                                              ptr<float>(c), ptr<float>(r), ptr<float>(op), ptr<float>(packed), ptr<float>(g),
                                              ptr<float>(ft), ptr<int32_t>(nc), n, ptr<float>(o.grad_accum),
                                              ptr<float>(o.dL_drgb), ptr<float>(o.dL_dopacity_act), ptr<float>(o.dL_dmeans_2d),
                                              ptr<float>(o.dL_dcov_2d_inv), prezeroed ? 1 : 0,
                                              reinterpret_cast<const uint32_t*>(tile_order.data_ptr<int32_t>()),
                                              stream_of(dL_dcolor)),
              "cugs_rasterize_backward_ordered");
        return o;
    }
    auto entry = prezeroed ? cugs_rasterize_backward_prezeroed : cugs_rasterize_backward;
    check(entry(img_w, img_h, background, ptr<int32_t>(tr), ptr<int32_t>(gi), ptr<float>(m), ptr<float>(c),
                ptr<float>(r), ptr<float>(op), ptr<float>(packed), ptr<float>(g), ptr<float>(ft),
                ptr<int32_t>(nc), n, ptr<float>(o.grad_accum), ptr<float>(o.dL_drgb),
                ptr<float>(o.dL_dopacity_act), ptr<float>(o.dL_dmeans_2d), ptr<float>(o.dL_dcov_2d_inv),
                stream_of(dL_dcolor)),
          prezeroed ? "cugs_rasterize_backward_prezeroed" : "cugs_rasterize_backward");
    return o;
}

namespace {
ProjectionBackwardOutput project_backward_impl(const torch::Tensor* accum, const torch::Tensor* colour_gate,
                                               torch::Tensor* d_means_out, const torch::Tensor& gm, const torch::Tensor& gc,
                                               const torch::Tensor& gr, const torch::Tensor& go,
                                               const torch::Tensor& positions, const torch::Tensor& rotations,
                                               const torch::Tensor& scales, const torch::Tensor& opacities,
                                               const torch::Tensor& sh_coeffs, const torch::Tensor& radii,
                                               const cugs_camera& camera, int degree, float scale_modifier) {
    TORCH_CHECK(positions.is_cuda(), "positions must be on CUDA");
    const int64_t n = positions.size(0);
    ProjectionBackwardOutput o;
    o.dL_dpositions = torch::empty({n, 3}, fopt(positions));
    o.dL_drotations = torch::empty({n, 4}, fopt(positions));
    o.dL_dscales = torch::empty({n, 3}, fopt(positions));
    o.dL_dopacities = torch::empty({n, 1}, fopt(positions));
    auto sh = f32c(sh_coeffs);
    o.dL_dsh_coeffs = torch::empty_like(sh);
    if (n == 0) return o;
    auto pos = f32c(positions), rot = f32c(rotations), scl = f32c(scales), opa = f32c(opacities), rad = radii.contiguous();
    auto cm = gm.defined() ? gm.contiguous() : gm, cc = gc.defined() ? gc.contiguous() : gc;
    auto cr = gr.defined() ? gr.contiguous() : gr, co = go.defined() ? go.contiguous() : go;
    check(cugs_project_backward(n, static_cast<int>(sh.size(2)), degree, ptr<float>(pos), ptr<float>(rot), ptr<float>(scl),
                                ptr<float>(opa), ptr<float>(sh), ptr<int32_t>(rad),
                                colour_gate ? ptr<uint8_t>(*colour_gate) : nullptr, &camera, scale_modifier,
                                accum ? ptr<float>(*accum) : nullptr, ptr<float>(cm), ptr<float>(cc), ptr<float>(cr),
                                ptr<float>(co), ptr<float>(o.dL_dpositions), ptr<float>(o.dL_drotations),
                                ptr<float>(o.dL_dscales), ptr<float>(o.dL_dopacities), ptr<float>(o.dL_dsh_coeffs),
                                d_means_out ? ptr<float>(*d_means_out) : nullptr, /*dL_drgb_gated_out=*/nullptr,
                                stream_of(positions)),
          "cugs_project_backward");
    return o;
}
}  // namespace

ProjectionBackwardOutput project_backward(const torch::Tensor& dL_dmeans_2d, const torch::Tensor& dL_dcov_2d_inv,
                                          const torch::Tensor& dL_drgb, const torch::Tensor& dL_dopacity_act,
                                          const torch::Tensor& positions, const torch::Tensor& rotations,
                                          const torch::Tensor& scales, const torch::Tensor& opacities,
                                          const torch::Tensor& sh_coeffs, const torch::Tensor& radii,
                                          const cugs_camera& camera, int active_sh_degree, float scale_modifier) {
    return project_backward_impl(nullptr, nullptr, nullptr, dL_dmeans_2d, dL_dcov_2d_inv, dL_drgb, dL_dopacity_act,
                                 positions, rotations, scales, opacities, sh_coeffs, radii, camera, active_sh_degree,
                                 scale_modifier);
}

torch::Tensor evaluate_sh_cuda(int degree, const torch::Tensor& sh_coeffs, const torch::Tensor& directions) {
    TORCH_CHECK(degree >= 0 && degree <= 3, "SH degree must be 0..3, got ", degree);                 // core/sh.cu:84-97
    TORCH_CHECK(sh_coeffs.is_cuda(), "sh_coeffs must be on CUDA device");
    TORCH_CHECK(directions.is_cuda(), "directions must be on CUDA device");
    TORCH_CHECK(sh_coeffs.dim() == 3 && sh_coeffs.size(1) == 3, "sh_coeffs must be [N, 3, C]");
    TORCH_CHECK(directions.dim() == 2 && directions.size(1) == 3, "directions must be [N, 3]");
    TORCH_CHECK(sh_coeffs.size(0) == directions.size(0), "Batch size mismatch");
    TORCH_CHECK(sh_coeffs.size(2) >= (degree + 1) * (degree + 1), "Need at least ", (degree + 1) * (degree + 1),
                " coefficients for degree ", degree);
    auto c = f32c(sh_coeffs), d = f32c(directions);
    auto out = torch::empty({c.size(0), 3}, fopt(c));
    if (c.size(0) == 0) return out;
    check(cugs_evaluate_sh(degree, c.size(0), static_cast<int>(c.size(2)), ptr<float>(c), ptr<float>(d), ptr<float>(out),
                           stream_of(c)), "cugs_evaluate_sh");
    return out;
}

torch::Tensor evaluate_sh_backward_cuda(int degree, const torch::Tensor& sh_coeffs, const torch::Tensor& directions,
                                        const torch::Tensor& dL_dcolor) {
    TORCH_CHECK(degree >= 0 && degree <= 3, "SH degree must be 0..3, got ", degree);                 // sh_backward.cu:120-130
    TORCH_CHECK(sh_coeffs.is_cuda() && directions.is_cuda() && dL_dcolor.is_cuda(), "inputs must be on CUDA device");
    TORCH_CHECK(sh_coeffs.dim() == 3 && sh_coeffs.size(1) == 3, "sh_coeffs must be [N, 3, C]");
    TORCH_CHECK(directions.dim() == 2 && directions.size(1) == 3, "directions must be [N, 3]");
    TORCH_CHECK(dL_dcolor.dim() == 2 && dL_dcolor.size(1) == 3, "dL_dcolor must be [N, 3]");
    auto c = f32c(sh_coeffs), d = f32c(directions), g = f32c(dL_dcolor);
    auto out = torch::empty_like(c);
    if (c.size(0) == 0) return out;
    check(cugs_evaluate_sh_backward(degree, c.size(0), static_cast<int>(c.size(2)), ptr<float>(c), ptr<float>(d),
                                    ptr<float>(g), ptr<float>(out), stream_of(c)), "cugs_evaluate_sh_backward");
    return out;
}

static int max_sh_degree(const torch::Tensor& sh) {                                                    // gaussian.hpp:47-54
    return sh.defined() ? static_cast<int>(std::sqrt(static_cast<float>(sh.size(2)))) - 1 : 0;
}

RenderOutput render(const ModelTensors& model, const cugs_camera& camera, const RenderSettings& settings,
                    bool for_backward) {
    TORCH_CHECK(model.positions.defined() && model.positions.is_cuda(), "GaussianModel must be on CUDA device");
    const int64_t n = model.positions.size(0);
    const int w = camera.width, h = camera.height;
    RenderOutput o;
    if (n == 0) {                                                                                     // rasterizer.cpp:36-55
        o.color = torch::empty({h, w, 3}, fopt(model.positions));
        for (int ch = 0; ch < 3; ++ch) o.color.select(2, ch).fill_(settings.background[ch]);
        o.final_T = torch::ones({h, w}, fopt(model.positions));
        o.n_contrib = torch::zeros({h, w}, iopt(model.positions));
        o.means_2d = torch::empty({0, 2}, fopt(model.positions)); o.depths = torch::empty({0}, fopt(model.positions));
        o.cov_2d_inv = torch::empty({0, 3}, fopt(model.positions)); o.radii = torch::empty({0}, iopt(model.positions));
        o.rgb = torch::empty({0, 3}, fopt(model.positions)); o.opacities_act = torch::empty({0}, fopt(model.positions));
        o.gaussian_indices = torch::empty({0}, iopt(model.positions)); o.tile_ranges = torch::empty({0, 2}, iopt(model.positions));
        return o;
    }
    const int degree = std::min(settings.active_sh_degree, max_sh_degree(model.sh_coeffs));
    SortState& state = sort_state(model.positions.device());
    const int num_tiles = ((w + CUGS_TILE - 1) / CUGS_TILE) * ((h + CUGS_TILE - 1) / CUGS_TILE);
    const bool predicted = state.last_pairs >= 0 && num_tiles > 0;
    const bool wide = state.wide_left > 0;       // this stream's views leave the three-pass depth range: general route
    if (wide) --state.wide_left;                 // at 0 the next sort probes the three-pass route again
    // with a prediction to sort on, the projection keys the sort's workspace in passing (one launch and 40 MB per million
    // Gaussians less; the fallbacks below, and the general depth route, rebuild the keys from the arrays)
    const bool keyed = predicted && !wide;
    auto proj = project_impl(model.positions, model.rotations, model.scales, model.opacities, model.sh_coeffs, camera,
                             degree, settings.scale_modifier, keyed);
    // The sort runs on the pair count predicted from this stream's previous frame (cugs_sort_pairs_predicted) and
    // the forward blend is queued behind it before the host looks at the true count: no idle device while the
    // host waits.  A prediction that was too small is detected afterwards and the exact path re-run.
    // the backward blend's accumulator, cleared in passing by the forward blend (issue-bound, HBM idle)
    torch::Tensor accum = for_backward ? torch::empty({n, CUGS_GRAD_STRIDE}, fopt(model.positions)) : torch::Tensor();
    auto blend = [&](const SortingOutput& s) {
        return rasterize_forward(proj.means_2d, proj.cov_2d_inv, proj.rgb, proj.opacities_act, s.tile_ranges,
                                 s.gaussian_values_sorted, w, h, settings.background, proj.packed, accum, s.tile_order);
    };
    SortingOutput srt;
    ForwardOutput fwd;
    const int64_t prev = state.last_pairs < 0 ? 0 : state.last_pairs;
    // longest-list-first order for the blend kernels' workgroups: worth making for large frames (~8 us of one workgroup
    // inside the sort's last kernel, which a small frame does not hide)
    const bool ordered = prev >= kTileOrderMinPairs;
    if (!predicted) {
        srt = sort_gaussians_impl(proj.means_2d, proj.depths, proj.radii, proj.tiles_touched, w, h, wide);
        fwd = blend(srt);
    } else {
        // capacity = estimate * 1.10 + 64 Ki, HELD while the estimate drifts below it (down to 80 %) and grown with 5 %
        // to spare: buffer sizes that follow a slowly moving pair count fragment the caching allocator
        const int64_t needed = std::min<int64_t>(prev + prev / 10 + 65536, 2147483647ll);
        int64_t cap = needed;
        if (state.held_cap > 0 && needed <= state.held_cap && state.held_cap <= needed + needed / 4) cap = state.held_cap;
        else if (state.held_cap > 0 && needed > state.held_cap) cap = std::min<int64_t>(needed + needed / 20, 2147483647ll);
        state.held_cap = cap;
        if (!state.pinned.defined()) state.pinned = torch::zeros({1}, torch::kInt64).pin_memory();
        auto total = state.pinned;
        void* st = stream_of(proj.means_2d);
        auto tiles = proj.tiles_touched.contiguous().to(torch::kInt32);
        srt.tile_ranges = torch::empty({num_tiles, 2}, iopt(proj.means_2d));
        srt.gaussian_values_sorted = torch::empty({cap}, iopt(proj.means_2d));
        auto ws = workspace(proj.means_2d.device(), cugs_sort_workspace_bytes(n), 0);
        auto wp = workspace(proj.means_2d.device(), cugs_sort_pair_workspace_bytes(cap), 1);
        if (wide) {
            check(cugs_sort_pairs_predicted_wide(
                      n, cap, ptr<float>(proj.means_2d), ptr<float>(proj.depths), ptr<int32_t>(proj.radii), ptr<int32_t>(tiles), w, h,
                      ws.data_ptr(), ws.numel(), wp.data_ptr(), wp.numel(), nullptr, ptr<int32_t>(srt.gaussian_values_sorted),
                      ptr<int32_t>(srt.tile_ranges), total.data_ptr<int64_t>(), st), "cugs_sort_pairs_predicted_wide");
            if (ordered) srt.tile_order = tile_order_of(srt.tile_ranges, w, h);
        } else if (!ordered) {
            check(cugs_sort_pairs_predicted_keyed(
                      n, cap, ptr<float>(proj.means_2d), ptr<float>(proj.depths), ptr<int32_t>(proj.radii), ptr<int32_t>(tiles), w, h,
                      ws.data_ptr(), ws.numel(), wp.data_ptr(), wp.numel(), nullptr, ptr<int32_t>(srt.gaussian_values_sorted),
                      ptr<int32_t>(srt.tile_ranges), total.data_ptr<int64_t>(), st), "cugs_sort_pairs_predicted_keyed");
        } else {
            // the sort also leaves the order the blend kernels hand their workgroups out in (longest tile list first)
            srt.tile_order = torch::empty({num_tiles, 4}, iopt(proj.means_2d));
            check(cugs_sort_pairs_predicted_keyed_ordered(
                      n, cap, ptr<float>(proj.means_2d), ptr<float>(proj.depths), ptr<int32_t>(proj.radii), ptr<int32_t>(tiles), w, h,
                      ws.data_ptr(), ws.numel(), wp.data_ptr(), wp.numel(), nullptr, ptr<int32_t>(srt.gaussian_values_sorted),
                      ptr<int32_t>(srt.tile_ranges), total.data_ptr<int64_t>(),
                      reinterpret_cast<uint32_t*>(srt.tile_order.data_ptr<int32_t>()), st), "cugs_sort_pairs_predicted_keyed_ordered");
        }
        hipEvent_t ev;
        TORCH_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess, "hipEventCreate failed");
        TORCH_CHECK(hipEventRecord(ev, static_cast<hipStream_t>(st)) == hipSuccess, "hipEventRecord failed");
        fwd = blend(srt);                                   // queued; the host has not waited yet
        const bool ok = hipEventSynchronize(ev) == hipSuccess;
        (void)hipEventDestroy(ev);
        TORCH_CHECK(ok, "hipEventSynchronize failed");
        const int64_t p = total.data_ptr<int64_t>()[0];
        TORCH_CHECK(p >= -1 && p <= 2147483647ll, "pair count exceeds the reference's int indexing");
        if (p >= 0 && p <= cap) {
            srt.total_pairs = static_cast<int>(p);
            srt.gaussian_values_sorted = srt.gaussian_values_sorted.slice(0, 0, p);
        } else {
            // prediction too small, or -1: a depth outside the three-pass sort's range - remembered for the following
            // sorts on this stream.  Exact path, blend again.
            if (p == -1) state.wide_left = kWideDepthHold;
            srt = sort_gaussians_impl(proj.means_2d, proj.depths, proj.radii, proj.tiles_touched, w, h,
                                      p == -1 || state.wide_left > 0);
            if (ordered) srt.tile_order = tile_order_of(srt.tile_ranges, w, h);
            fwd = blend(srt);
        }
    }
    // running maximum with a slow decay: views differ by tens of percent, spare capacity is cheap, a miss is not
    state.last_pairs = std::max<int64_t>(srt.total_pairs, prev - prev / 32);
    o.color = fwd.color; o.final_T = fwd.final_T; o.n_contrib = fwd.n_contrib;
    o.means_2d = proj.means_2d; o.depths = proj.depths; o.cov_2d_inv = proj.cov_2d_inv; o.radii = proj.radii;
    o.rgb = proj.rgb; o.opacities_act = proj.opacities_act;
    o.gaussian_indices = srt.gaussian_values_sorted; o.tile_ranges = srt.tile_ranges; o.packed = proj.packed;
    o.tile_order = srt.tile_order;
    o.colour_gate = proj.colour_gate;
    o.zeroed_accum = accum;
    if (accum.defined()) o.accum_used = std::make_shared<std::atomic<bool>>(false);
    return o;
}

BackwardOutput render_backward(const torch::Tensor& dL_dcolor, const RenderOutput& ro, const ModelTensors& model,
                               const cugs_camera& camera, const RenderSettings& settings, FusedAdam* fused) {
    TORCH_CHECK(dL_dcolor.is_cuda(), "dL_dcolor must be on CUDA device");                             // rasterizer.cpp:122-124
    TORCH_CHECK(dL_dcolor.dim() == 3 && dL_dcolor.size(2) == 3, "dL_dcolor must be [H, W, 3]");
    const int64_t n = model.positions.size(0);
    BackwardOutput o;
    if (n == 0) {                                                                                     // rasterizer.cpp:130-139
        o.dL_dpositions = torch::zeros({0, 3}, fopt(dL_dcolor)); o.dL_drotations = torch::zeros({0, 4}, fopt(dL_dcolor));
        o.dL_dscales = torch::zeros({0, 3}, fopt(dL_dcolor)); o.dL_dopacities = torch::zeros({0, 1}, fopt(dL_dcolor));
        o.dL_dsh_coeffs = torch::zeros_like(model.sh_coeffs); o.dL_dmeans_2d = torch::zeros({0, 2}, fopt(dL_dcolor));
        return o;
    }
    const int degree = std::min(settings.active_sh_degree, max_sh_degree(model.sh_coeffs));
    // a RenderOutput that lost its packed records (e.g. it went through the reference's struct, which has no
    // field for them) gets them rebuilt - one 48 B/Gaussian launch - so the packed blend kernel runs either way
    torch::Tensor packed = ro.packed;
    if (!packed.defined() || packed.size(0) != n) {
        packed = torch::empty({n, CUGS_PACKED_STRIDE}, fopt(dL_dcolor));
        auto m = ro.means_2d.contiguous(), c = ro.cov_2d_inv.contiguous(), r = ro.rgb.contiguous(),
             op = ro.opacities_act.contiguous();
        check(cugs_pack_projected(n, ptr<float>(m), ptr<float>(c), ptr<float>(r), ptr<float>(op), ptr<float>(packed),
                                  stream_of(dL_dcolor)), "cugs_pack_projected");
    }
    // the accumulator render() had the forward blend clear is good for ONE backward
    torch::Tensor zeroed = ro.zeroed_accum;
    ro.zeroed_accum = torch::Tensor();
    if (zeroed.defined() && ro.accum_used && ro.accum_used->exchange(true)) zeroed = torch::Tensor();   // a copy used it
    if (zeroed.defined() && (zeroed.dim() != 2 || zeroed.size(0) != n)) zeroed = torch::Tensor();
    auto rb = rasterize_backward(dL_dcolor, ro.means_2d, ro.cov_2d_inv, ro.rgb, ro.opacities_act, ro.tile_ranges,
                                 ro.gaussian_indices, ro.final_T, ro.n_contrib, camera.width, camera.height,
                                 settings.background, static_cast<int>(n), packed, /*unpack=*/false, zeroed, ro.tile_order);
    o.dL_dmeans_2d = torch::empty({n, 2}, fopt(dL_dcolor));
    if (fused) {                                                // a8 + a9 + a11 in one launch, parameters updated in place
        const auto& pr = fused->params();
        TORCH_CHECK(pr[0].data_ptr() == model.positions.data_ptr() && pr[1].data_ptr() == model.sh_coeffs.data_ptr(),
                    "the fused optimizer step needs the FusedAdam that was built on this model");
        TORCH_CHECK(ro.colour_gate.defined() && ro.colour_gate.size(0) == n,
                    "the fused optimizer step needs the colour_gate of cugs_hip::render");
        const cugs_adam_fused adam = fused->begin_fused_step();
        auto radii = ro.radii.contiguous(), gate = ro.colour_gate.contiguous();
        check(cugs_project_backward_adam(n, static_cast<int>(model.sh_coeffs.size(2)), degree, ptr<float>(model.positions),
                                         ptr<float>(model.rotations), ptr<float>(model.scales), ptr<float>(model.opacities),
                                         ptr<float>(model.sh_coeffs), ptr<int32_t>(radii), ptr<uint8_t>(gate), &camera,
                                         settings.scale_modifier, ptr<float>(rb.grad_accum), &adam,
                                         ptr<float>(o.dL_dmeans_2d), stream_of(dL_dcolor)),
              "cugs_project_backward_adam");
        return o;
    }
    // without the gate bits (a RenderOutput that went through the reference's struct) the kernel recomputes the gate
    // from the coefficients as the reference does: the same bits, 12 C more bytes read per Gaussian
    const bool have_gate = ro.colour_gate.defined() && ro.colour_gate.dim() == 1 && ro.colour_gate.size(0) == n;
    auto pb = project_backward_impl(&rb.grad_accum, have_gate ? &ro.colour_gate : nullptr, &o.dL_dmeans_2d, {}, {}, {}, {}, model.positions,
                                    model.rotations, model.scales, model.opacities, model.sh_coeffs, ro.radii, camera, degree,
                                    settings.scale_modifier);
    o.dL_dpositions = pb.dL_dpositions; o.dL_drotations = pb.dL_drotations; o.dL_dscales = pb.dL_dscales;
    o.dL_dopacities = pb.dL_dopacities; o.dL_dsh_coeffs = pb.dL_dsh_coeffs;
    return o;
}

namespace {
void validate_pair(const torch::Tensor& rendered, const torch::Tensor& target) {               // loss.cpp:14-34
    for (const torch::Tensor* img : {&rendered, &target}) {
        TORCH_CHECK(img->dim() == 3, "image must be 3-dimensional [H, W, 3], got ", img->dim(), " dims");
        TORCH_CHECK(img->size(2) == 3, "image must have 3 channels, got ", img->size(2));
        TORCH_CHECK(img->dtype() == torch::kFloat32, "image must be float32, got ", img->dtype());
        TORCH_CHECK(img->is_cuda(), "image must be on a CUDA device");
    }
    TORCH_CHECK(rendered.sizes() == target.sizes(), "rendered and target must have the same shape, got ",
                rendered.sizes(), " vs ", target.sizes());
}
torch::Tensor run_loss(const torch::Tensor& rendered, const torch::Tensor& target, float lambda_, int window_size,
                       torch::Tensor* ssim_map, torch::Tensor* grad) {
    validate_pair(rendered, target);
    TORCH_CHECK(window_size % 2 == 1, "window_size must be odd, got ", window_size);               // loss.cpp:96-97
    TORCH_CHECK(window_size >= 3, "window_size must be >= 3, got ", window_size);
    const int h = static_cast<int>(rendered.size(0)), w = static_cast<int>(rendered.size(1));
    auto r = rendered.contiguous(), t = target.contiguous();
    auto out = torch::empty({4}, fopt(rendered));
    if (ssim_map) *ssim_map = torch::empty({h, w}, fopt(rendered));
    if (grad) *grad = torch::empty({h, w, 3}, fopt(rendered));
    auto ws = workspace(rendered.device(), cugs_loss_workspace_bytes(w, h), 2);
    check(cugs_combined_loss(w, h, ptr<float>(r), ptr<float>(t), lambda_, window_size, ws.data_ptr(), ws.numel(),
                             ptr<float>(out), ssim_map ? ptr<float>(*ssim_map) : nullptr,
                             grad ? ptr<float>(*grad) : nullptr, stream_of(rendered)), "cugs_combined_loss");
    return out;
}
}  // namespace

LossAndGrad combined_loss_and_grad(const torch::Tensor& rendered, const torch::Tensor& target, float lambda_, bool want_grad) {
    LossAndGrad o;
    auto out = run_loss(rendered, target, lambda_, 11, nullptr, want_grad ? &o.dL_dcolor : nullptr);
    o.loss = out[0]; o.l1 = out[1]; o.ssim_mean = out[2];
    return o;
}
torch::Tensor combined_loss(const torch::Tensor& rendered, const torch::Tensor& target, float lambda_) {
    return run_loss(rendered, target, lambda_, 11, nullptr, nullptr)[0];
}
torch::Tensor ssim(const torch::Tensor& rendered, const torch::Tensor& target, int window_size) {
    torch::Tensor map;
    run_loss(rendered, target, 0.2f, window_size, &map, nullptr);
    return map;
}

FusedAdam::FusedAdam(std::array<torch::Tensor, 5> params, std::array<float, 5> lrs, AdamHyper h)
    : params_(std::move(params)), lrs_(lrs), h_(h) {
    for (int i = 0; i < 5; ++i) { m_[i] = torch::zeros_like(params_[i]); v_[i] = torch::zeros_like(params_[i]); }
}
void FusedAdam::apply_gradients(const BackwardOutput& g) {        // references, no copy (fused_adam.cu:113-120)
    grads_ = {g.dL_dpositions, g.dL_dsh_coeffs, g.dL_dopacities, g.dL_dscales, g.dL_drotations};
}
void FusedAdam::zero_grad() { for (auto& g : grads_) g = torch::Tensor(); }
cugs_adam_fused FusedAdam::begin_fused_step() {
    ++step_count_;
    cugs_adam_fused a{};
    cugs_adam_bias_correction(h_.beta1, h_.beta2, step_count_, &a.bc1, &a.bc2);
    for (int i = 0; i < 5; ++i) {
        TORCH_CHECK(params_[i].is_cuda() && params_[i].is_contiguous() && params_[i].scalar_type() == torch::kFloat32 &&
                    m_[i].is_contiguous() && v_[i].is_contiguous(),
                    "FusedAdam: the fused step needs contiguous float32 CUDA parameters and moments");
        a.m[i] = m_[i].data_ptr<float>(); a.v[i] = v_[i].data_ptr<float>(); a.lr[i] = lrs_[i];
    }
    a.beta1 = h_.beta1; a.beta2 = h_.beta2; a.eps = h_.eps;
    zero_grad();
    return a;
}
void FusedAdam::step() {
    ++step_count_;
    float bc1, bc2;
    cugs_adam_bias_correction(h_.beta1, h_.beta2, step_count_, &bc1, &bc2);
    cugs_adam_group groups[5];
    std::array<torch::Tensor, 5> pc, gc, mc, vc;
    void* st = nullptr;
    for (int i = 0; i < 5; ++i) {
        groups[i] = cugs_adam_group{nullptr, nullptr, nullptr, nullptr, 0, lrs_[i], 0.f};
        if (!grads_[i].defined()) continue;                       // fused_adam.cu:156
        TORCH_CHECK(params_[i].is_cuda(), "FusedAdam: param must be on CUDA");
        TORCH_CHECK(grads_[i].is_cuda(), "FusedAdam: grad must be on CUDA");
        TORCH_CHECK(params_[i].numel() == grads_[i].numel(), "FusedAdam: param/grad size mismatch: ", params_[i].numel(),
                    " vs ", grads_[i].numel());
        pc[i] = params_[i].contiguous(); gc[i] = grads_[i].contiguous(); mc[i] = m_[i].contiguous(); vc[i] = v_[i].contiguous();
        groups[i].param = pc[i].data_ptr<float>(); groups[i].grad = gc[i].data_ptr<float>();
        groups[i].m = mc[i].data_ptr<float>(); groups[i].v = vc[i].data_ptr<float>(); groups[i].n = pc[i].numel();
        st = stream_of(params_[i]);
    }
    check(cugs_fused_adam_groups(groups, 5, h_.beta1, h_.beta2, h_.eps, bc1, bc2, st), "cugs_fused_adam_groups");
    for (int i = 0; i < 5; ++i) {                                 // fused_adam.cu:216-218
        if (!grads_[i].defined()) continue;
        if (!params_[i].is_contiguous()) params_[i].copy_(pc[i]);
        if (!m_[i].is_contiguous()) m_[i].copy_(mc[i]);
        if (!v_[i].is_contiguous()) v_[i].copy_(vc[i]);
    }
}

// ---------------------------------------------------------------------------------------------
// N2: adaptive density control (optimizer/densification.cpp)
// ---------------------------------------------------------------------------------------------
bool DensificationController::should_densify(int step) const {             // densification.cpp:42-46
    return step >= config_.densify_from && step <= config_.densify_until && step % config_.densify_every == 0;
}
bool DensificationController::should_reset_opacity(int step) const {       // :48-52
    return config_.opacity_reset_every > 0 && step >= config_.densify_from && step % config_.opacity_reset_every == 0;
}
void DensificationController::reset_accumulators(int64_t n, const torch::Device& device) {   // :344-349
    auto o = torch::TensorOptions().dtype(torch::kFloat32).device(device);
    grad_accum_ = torch::zeros({n}, o); grad_count_ = torch::zeros({n}, o); max_radii_2d_ = torch::zeros({n}, o);
}
void DensificationController::accumulate_gradients(const torch::Tensor& dL_dmeans_2d, const torch::Tensor& radii) {
    TORCH_CHECK(dL_dmeans_2d.is_cuda() && radii.is_cuda(), "accumulate_gradients: tensors must be on CUDA");
    TORCH_CHECK(dL_dmeans_2d.dim() == 2 && dL_dmeans_2d.size(1) == 2, "dL_dmeans_2d must be [N, 2]");
    const int64_t n = dL_dmeans_2d.size(0);
    TORCH_CHECK(radii.numel() == n, "radii must be [N]");
    if (!grad_accum_.defined() || grad_accum_.size(0) != n) reset_accumulators(n, dL_dmeans_2d.device());
    auto g = dL_dmeans_2d.contiguous().to(torch::kFloat32);
    auto r = radii.contiguous().to(torch::kInt32);
    check(cugs_densify_accumulate(n, ptr<float>(g), ptr<int32_t>(r), ptr<float>(grad_accum_), ptr<float>(grad_count_),
                                  ptr<float>(max_radii_2d_), stream_of(g)), "cugs_densify_accumulate");
}
void DensificationController::reset_opacity(ModelTensors& model) {          // :331-334
    torch::NoGradGuard no_grad;
    model.opacities.fill_(-4.59511985013459f);
}
DensificationStats DensificationController::densify(ModelTensors& model, int step, const torch::Tensor& noise_in,
                                                    FusedAdam* optimizer) {
    torch::NoGradGuard no_grad;
    DensificationStats stats;
    const int64_t n = model.positions.size(0);
    stats.num_before = stats.num_after = static_cast<int>(n);
    if (n == 0) return stats;
    TORCH_CHECK(model.positions.is_cuda(), "densify: model must be on CUDA");
    const auto dev = model.positions.device();
    if (!grad_accum_.defined() || grad_accum_.size(0) != n) reset_accumulators(n, dev);
    void* st = stream_of(model.positions);
    auto scales_c = model.scales.contiguous(), opa_c = model.opacities.contiguous();
    auto flags = torch::empty({n}, torch::TensorOptions().dtype(torch::kUInt8).device(dev));
    auto avg = torch::empty({n}, fopt(model.positions));
    const float size_thr = config_.percent_dense * scene_extent_;           // :367, :394
    const float ws_thr = 0.1f * scene_extent_;                              // :436
    const int size_pruning = config_.opacity_reset_every > 0 && step > config_.opacity_reset_every;   // :415-416
    check(cugs_densify_classify(n, ptr<float>(grad_accum_), ptr<float>(grad_count_), ptr<float>(max_radii_2d_),
                                ptr<float>(scales_c), ptr<float>(opa_c), config_.grad_threshold, size_thr,
                                config_.opacity_threshold, size_pruning, static_cast<float>(config_.max_screen_size),
                                ws_thr, flags.data_ptr<uint8_t>(), ptr<float>(avg), st), "cugs_densify_classify");
    if (config_.max_gaussians > 0) {                                        // :121-139, :184-209 (rare path: libtorch topk)
        auto clone = flags.bitwise_and(1).to(torch::kBool);
        int64_t num_clone = clone.sum().item<int64_t>();
        const int64_t budget = config_.max_gaussians - n;
        if (num_clone > budget) {
            auto keep_clone = torch::zeros_like(clone);
            if (budget > 0) keep_clone.index_fill_(0, std::get<1>(avg.masked_fill(~clone, -1.0f).topk(budget)), true);
            clone = keep_clone;
            num_clone = budget > 0 ? budget : 0;
        }
        auto split = flags.bitwise_right_shift(1).bitwise_and(1).to(torch::kBool);
        const int64_t num_split = split.sum().item<int64_t>();
        const int64_t sbudget = (config_.max_gaussians - (n + num_clone)) / 2;
        if (num_split > sbudget) {
            auto keep_split = torch::zeros_like(split);
            if (sbudget > 0) keep_split.index_fill_(0, std::get<1>(avg.masked_fill(~split, -1.0f).topk(sbudget)), true);
            split = keep_split;
        }
        flags = flags.bitwise_and(4).bitwise_or(clone.to(torch::kUInt8)).bitwise_or(split.to(torch::kUInt8).bitwise_left_shift(1)).contiguous();
    }
    auto ws = workspace(dev, cugs_densify_workspace_bytes(n), 3);
    int64_t counts[4];
    check(cugs_densify_plan(n, flags.data_ptr<uint8_t>(), ws.data_ptr(), ws.numel(), counts, st), "cugs_densify_plan");
    const int64_t n_out = counts[3];
    torch::Tensor noise = noise_in.defined() ? noise_in.contiguous().to(torch::kFloat32)
                                             : (counts[2] > 0 ? torch::randn({2, n, 3}, fopt(model.positions))
                                                              : torch::zeros({2, n, 3}, fopt(model.positions)));
    TORCH_CHECK(noise.is_cuda() && noise.dim() == 3 && noise.size(0) == 2 && noise.size(1) == n && noise.size(2) == 3,
                "noise must be [2, N, 3] on CUDA");
    std::vector<torch::Tensor> srcs, dsts;
    std::vector<cugs_densify_array> desc;
    auto add = [&](const torch::Tensor& src, int mode) {
        auto s = src.contiguous();
        auto sizes = s.sizes().vec();
        sizes[0] = n_out;
        auto d = torch::empty(sizes, fopt(model.positions));
        srcs.push_back(s); dsts.push_back(d);
        desc.push_back(cugs_densify_array{s.data_ptr<float>(), d.data_ptr<float>(), static_cast<int32_t>(s.numel() / n), mode});
        return d;
    };
    auto new_pos = add(model.positions, CUGS_DENSIFY_POSITIONS);
    auto new_sh = add(model.sh_coeffs, CUGS_DENSIFY_COPY);
    auto new_opa = add(model.opacities, CUGS_DENSIFY_COPY);
    auto new_rot = add(model.rotations, CUGS_DENSIFY_COPY);
    auto new_scl = add(model.scales, CUGS_DENSIFY_SCALES);
    std::array<torch::Tensor, 5> nm, nv;
    if (optimizer) for (int i = 0; i < 5; ++i) { nm[i] = add(optimizer->m_[i], CUGS_DENSIFY_STATE); nv[i] = add(optimizer->v_[i], CUGS_DENSIFY_STATE); }
    if (n_out > 0)
        check(cugs_densify_apply(n, n_out, ws.data_ptr(), ws.numel(), ptr<float>(noise), ptr<float>(scales_c), desc.data(),
                                 static_cast<int>(desc.size()), st), "cugs_densify_apply");
    model.positions = new_pos; model.sh_coeffs = new_sh; model.opacities = new_opa; model.rotations = new_rot; model.scales = new_scl;
    if (optimizer) {
        optimizer->params_ = {model.positions, model.sh_coeffs, model.opacities, model.scales, model.rotations};
        optimizer->m_ = nm; optimizer->v_ = nv;
        optimizer->zero_grad();
    }
    stats.num_cloned = static_cast<int>(counts[1]);
    stats.num_split = static_cast<int>(counts[2]);
    stats.num_pruned = static_cast<int>(n + counts[1] + 2 * counts[2] - n_out);   // :313-316
    stats.num_after = static_cast<int>(n_out);
    reset_accumulators(n_out, dev);                                               // :321
    return stats;
}

// N4: training target from a cached 8-bit view
torch::Tensor image_to_float(const torch::Tensor& view_u8, int width, int height) {
    TORCH_CHECK(view_u8.is_cuda() && view_u8.dtype() == torch::kUInt8 && view_u8.dim() == 3 && view_u8.size(2) == 3,
                "image must be a uint8 [H, W, 3] CUDA tensor");
    if (width <= 0 || height <= 0) throw std::runtime_error("Invalid target dimensions for resize");   // image_io.cpp:48-50
    auto src = view_u8.contiguous();
    auto dst = torch::empty({height, width, 3}, torch::TensorOptions().dtype(torch::kFloat32).device(src.device()));
    check(cugs_image_to_float(static_cast<int>(src.size(1)), static_cast<int>(src.size(0)), src.data_ptr<uint8_t>(), width,
                              height, dst.data_ptr<float>(), stream_of(src)), "cugs_image_to_float");
    return dst;
}

// ---------------------------------------------------------------------------------------------
// N3: PLY checkpoints (utils/ply_io.cpp:98-196, 258-351)
// ---------------------------------------------------------------------------------------------
namespace {
std::vector<std::string> ply_model_names(int c) {                 // record order without the normals
    std::vector<std::string> n = {"x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2"};
    for (int i = 0; i < 3 * (c - 1); ++i) n.push_back("f_rest_" + std::to_string(i));
    for (const char* p : {"opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"}) n.push_back(p);
    return n;
}
}  // namespace

bool write_gaussian_ply(const std::string& path, const ModelTensors& model, const FusedAdam* optimizer) {
    const auto& p0 = model.positions;
    if (!p0.defined() || p0.dim() != 2 || p0.size(1) != 3 || !model.sh_coeffs.defined() || model.sh_coeffs.dim() != 3 ||
        model.sh_coeffs.size(0) != p0.size(0) || model.sh_coeffs.size(1) != 3 || model.opacities.numel() != p0.size(0) ||
        model.scales.numel() != 3 * p0.size(0) || model.rotations.numel() != 4 * p0.size(0))
        return false;                                             // ply_io.cpp:100-103 (is_valid)
    const auto dev = p0.is_cuda() ? p0.device() : torch::Device(torch::kCUDA, c10::hip::current_device());
    auto f32 = [&](const torch::Tensor& t) { return t.to(dev).contiguous().to(torch::kFloat32); };
    std::array<torch::Tensor, 5> pr = {f32(model.positions), f32(model.sh_coeffs), f32(model.opacities), f32(model.scales),
                                       f32(model.rotations)}, mm, vv;
    const int64_t n = pr[0].size(0);
    const int c = static_cast<int>(pr[1].size(2));
    const bool state = optimizer != nullptr;
    const float *pp[5], *pm[5], *pv[5];
    for (int i = 0; i < 5; ++i) {
        pp[i] = pr[i].data_ptr<float>();
        if (state) { mm[i] = f32(optimizer->m_[i]); vv[i] = f32(optimizer->v_[i]); pm[i] = mm[i].data_ptr<float>(); pv[i] = vv[i].data_ptr<float>(); }
    }
    const int row = cugs_ply_vertex_floats(c, state ? 1 : 0);
    auto verts = torch::empty({n, row}, fopt(pr[0]));
    check(cugs_ply_pack(n, c, pp, state ? pm : nullptr, state ? pv : nullptr, ptr<float>(verts), stream_of(pr[0])), "cugs_ply_pack");
    auto host = verts.cpu();                                      // one device-to-host copy of the finished records
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    std::string head = "ply\nformat binary_little_endian 1.0\n";
    if (state) head += "comment cugs_adam_step " + std::to_string(optimizer->step_count_) + "\n";
    head += "element vertex " + std::to_string(n) + "\n";
    auto names = ply_model_names(c);
    auto emit = [&](const std::string& prefix, bool normals) {
        for (size_t i = 0; i < names.size(); ++i) {
            head += "property float " + prefix + names[i] + "\n";
            if (normals && i == 2) head += "property float nx\nproperty float ny\nproperty float nz\n";
        }
    };
    emit("", true);
    if (state) { emit("m_", false); emit("v_", false); }
    head += "end_header\n";
    bool ok = fwrite(head.data(), 1, head.size(), f) == head.size();
    const size_t bytes = static_cast<size_t>(n) * row * sizeof(float);
    ok = ok && (bytes == 0 || fwrite(host.data_ptr<float>(), 1, bytes, f) == bytes);
    return (fclose(f) == 0) && ok;
}

ModelTensors read_gaussian_ply(const std::string& path, const torch::Device& device, FusedAdam* optimizer) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("Failed to open PLY file: " + path);
    std::string buf;
    {
        char chunk[1 << 16];
        size_t got;
        while ((got = fread(chunk, 1, sizeof(chunk), f)) > 0) buf.append(chunk, got);
        fclose(f);
    }
    // parse_ply_header (ply_io.cpp:211-250)
    size_t pos = 0;
    std::vector<std::string> lines;
    while (true) {
        const size_t end = buf.find('\n', pos);
        if (end == std::string::npos) throw std::runtime_error("Not a PLY file");
        std::string line = buf.substr(pos, end - pos);
        if (!line.empty() && line.back() == '\r') line.pop_back();
        pos = end + 1;
        lines.push_back(line);
        if (line == "end_header") break;
    }
    if (lines[0].find("ply") == std::string::npos) throw std::runtime_error("Not a PLY file");
    if (lines.size() < 2 || lines[1].find("binary_little_endian") == std::string::npos)
        throw std::runtime_error("Only binary_little_endian PLY is supported");
    int64_t n = 0;
    long step = 0;
    std::vector<std::string> names;
    for (size_t i = 2; i < lines.size(); ++i) {
        char a[64] = {0}, b[64] = {0}, c3[64] = {0};
        const int got = sscanf(lines[i].c_str(), "%63s %63s %63s", a, b, c3);
        if (got >= 3 && std::string(a) == "element" && std::string(b) == "vertex") n = atoll(c3);
        else if (got >= 3 && std::string(a) == "property") names.push_back(c3);
        else if (got == 3 && std::string(a) == "comment" && std::string(b) == "cugs_adam_step") step = atol(c3);
    }
    std::map<std::string, int> index;
    for (size_t i = 0; i < names.size(); ++i) index[names[i]] = static_cast<int>(i);
    int num_rest = 0;
    while (index.count("f_rest_" + std::to_string(num_rest))) ++num_rest;
    const int c = 1 + num_rest / 3;                               // :283
    const bool state = optimizer != nullptr && index.count("m_x") && index.count("v_x");
    auto canon = ply_model_names(c);
    std::vector<int32_t> col_of;
    for (const char* prefix : {"", "m_", "v_"}) {
        if (prefix[0] && !state) break;
        for (auto& nm : canon) {
            auto it = index.find(std::string(prefix) + nm);
            if (it == index.end()) throw std::runtime_error("Missing PLY property: " + std::string(prefix) + nm);
            col_of.push_back(it->second);
        }
    }
    const int num_props = static_cast<int>(names.size());
    if (buf.size() - pos < static_cast<size_t>(n) * num_props * sizeof(float)) throw std::runtime_error("Failed to read PLY binary data");
    const auto work = device.is_cuda() ? device : torch::Device(torch::kCUDA, c10::hip::current_device());
    auto data = torch::from_blob(const_cast<char*>(buf.data() + pos), {n * num_props}, torch::kFloat32).to(work);
    auto cols = torch::from_blob(col_of.data(), {static_cast<int64_t>(col_of.size())}, torch::kInt32).to(work);
    auto o = torch::TensorOptions().dtype(torch::kFloat32).device(work);
    auto mk = [&] { return std::array<torch::Tensor, 5>{torch::empty({n, 3}, o), torch::empty({n, 3, c}, o), torch::empty({n, 1}, o),
                                                        torch::empty({n, 3}, o), torch::empty({n, 4}, o)}; };
    auto pr = mk();
    std::array<torch::Tensor, 5> mm, vv;
    float *pp[5], *pm[5], *pv[5];
    if (state) { mm = mk(); vv = mk(); }
    for (int i = 0; i < 5; ++i) { pp[i] = ptr<float>(pr[i]); if (state) { pm[i] = ptr<float>(mm[i]); pv[i] = ptr<float>(vv[i]); } }
    check(cugs_ply_unpack(n, c, num_props, ptr<float>(data), cols.data_ptr<int32_t>(), pp, state ? pm : nullptr,
                          state ? pv : nullptr, stream_of(data)), "cugs_ply_unpack");
    ModelTensors m{pr[0].to(device), pr[1].to(device), pr[2].to(device), pr[4].to(device), pr[3].to(device)};
    if (state) {
        for (int i = 0; i < 5; ++i) {
            TORCH_CHECK(optimizer->m_[i].sizes() == mm[i].sizes(), "optimizer state shape mismatch");
            optimizer->m_[i] = mm[i].to(optimizer->m_[i].device());
            optimizer->v_[i] = vv[i].to(optimizer->v_[i].device());
        }
        optimizer->step_count_ = static_cast<int>(step);
    }
    return m;
}

}  // namespace cugs_hip
