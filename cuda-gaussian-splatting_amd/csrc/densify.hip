// densify.hip — adaptive density control on the device.
//
// Replaces, per call: DensificationController::accumulate_gradients (optimizer/densification.cpp:59-88:
// norm + three boolean-mask index/index_put pairs, each with a host sync for the mask's nonzero count),
// compute_clone_mask / compute_split_mask / compute_keep_mask (:351-442, ~25 elementwise libtorch
// kernels) and the tensor surgery of densify() (:116-325: per parameter tensor two or three masked
// index() gathers, up to four cat()s and a final masked index().clone()).
//
// Here: one kernel per iteration for the statistics; per densification one classify kernel, a
// three-channel count/scan ("plan", with the one read-back that sizes the new model - the reference
// has sum().item() twice plus the syncs inside every boolean index), one kernel that builds the
// destination->source map, and one row-gather per array that writes the new model in its final order
//      [ kept originals | clones | first children | second children ]      (all in ascending parent index)
// which is exactly what the reference's append-then-prune sequence produces.  Optimizer moments can
// ride along (mode CUGS_DENSIFY_STATE: survivors keep theirs, new Gaussians start at zero) instead
// of being lost to an optimizer rebuild (trainer.cpp:267-304).  All HBM-bound byte movement.
#include "cugs_common.h"

namespace {

constexpr int ITEMS = 4;                                  // Gaussians per thread in the count/scan kernels
constexpr int SCAN_CHUNK = CUGS_BLOCK * ITEMS;            // 1024 per workgroup
constexpr float LOG_SPLIT = 0.47000366f;                  // std::log(1.6f), the split's scale shrink (densification.cpp:253-254)

struct DensifyWs {
    unsigned long long* totals;      // [4] kept originals, clones, splits, n_out
    uint32_t* blocksum;              // [3][nb] then scanned in place
    int32_t* src_of;                 // [2n] destination row -> source row
    size_t bytes;
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline uint32_t nblocks(int64_t n) { return (uint32_t)((n + SCAN_CHUNK - 1) / SCAN_CHUNK); }

DensifyWs carve(void* base, int64_t n) {
    char* p = static_cast<char*>(base);
    size_t off = 0;
    DensifyWs w;
    w.totals = reinterpret_cast<unsigned long long*>(p + off); off = align_up(off + 8 * sizeof(unsigned long long), 256);
    w.blocksum = reinterpret_cast<uint32_t*>(p + off); off = align_up(off + sizeof(uint32_t) * 3 * ((size_t)nblocks(n) + 1), 256);
    w.src_of = reinterpret_cast<int32_t*>(p + off); off = align_up(off + sizeof(int32_t) * 2 * (size_t)n, 256);
    w.bytes = off;
    return w;
}

// flags: bit 0 clone, bit 1 split, bit 2 keep (compute_keep_mask).  An original survives the prune iff it
// is kept and not split (densification.cpp:296-303); clones and children always do (:308-311).
__device__ __forceinline__ void decode(uint8_t f, uint32_t& keep, uint32_t& clone, uint32_t& split) {
    clone = f & 1u;
    split = (f >> 1) & 1u;
    keep = ((f >> 2) & 1u) & (split ^ 1u);
}

__device__ __forceinline__ uint32_t wave_incl(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}

// Exclusive scan of three packed-independent counters over a 256-thread workgroup.
__device__ __forceinline__ void block_excl3(uint32_t a, uint32_t b, uint32_t c, uint32_t (*s_tmp)[4], uint32_t& ea,
                                            uint32_t& eb, uint32_t& ec, uint32_t* tot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t ia = wave_incl(a), ib = wave_incl(b), ic = wave_incl(c);
    if (lane == 63) { s_tmp[0][wave] = ia; s_tmp[1][wave] = ib; s_tmp[2][wave] = ic; }
    __syncthreads();
    uint32_t ba = 0, bb = 0, bc = 0, ta = 0, tb = 0, tc = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t xa = s_tmp[0][w], xb = s_tmp[1][w], xc = s_tmp[2][w];
        if (w < wave) { ba += xa; bb += xb; bc += xc; }
        ta += xa; tb += xb; tc += xc;
    }
    ea = ba + ia - a; eb = bb + ib - b; ec = bc + ic - c;
    if (tot) { tot[0] = ta; tot[1] = tb; tot[2] = tc; }
    __syncthreads();
}

// accumulate_gradients (densification.cpp:59-88)
__global__ __launch_bounds__(CUGS_BLOCK) void k_accumulate(int64_t n, const float* __restrict__ dmeans,
                                                           const int32_t* __restrict__ radii,
                                                           float* __restrict__ grad_accum,
                                                           float* __restrict__ grad_count,
                                                           float* __restrict__ max_radii) {
    const int64_t i = (int64_t)blockIdx.x * CUGS_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int r = radii[i];
    if (r > 0) {                                                             // visible in this view (:70)
        const float2 g = reinterpret_cast<const float2*>(dmeans)[i];
        grad_accum[i] += sqrtf(g.x * g.x + g.y * g.y);                        // ||dL/d(screen xy)||_2 (:76)
        grad_count[i] += 1.0f;
    }
    max_radii[i] = fmaxf(max_radii[i], (float)r);                            // :86-87 (every Gaussian)
}

// compute_clone_mask / compute_split_mask / compute_keep_mask (densification.cpp:351-442)
__global__ __launch_bounds__(CUGS_BLOCK) void k_classify(int64_t n, const float* __restrict__ grad_accum,
                                                         const float* __restrict__ grad_count,
                                                         const float* __restrict__ max_radii,
                                                         const float* __restrict__ scales,
                                                         const float* __restrict__ opacities, float grad_thr,
                                                         float size_thr, float opa_thr, int size_pruning,
                                                         float max_screen, float ws_thr,
                                                         uint8_t* __restrict__ flags, float* __restrict__ avg_out) {
    const int64_t i = (int64_t)blockIdx.x * CUGS_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float avg = grad_accum[i] / fmaxf(grad_count[i], 1.0f);            // clamp_min(1), :359-360
    const bool high = avg >= grad_thr;
    const float max_scale = fmaxf(fmaxf(cugs_expf(scales[i * 3 + 0]), cugs_expf(scales[i * 3 + 1])),
                                  cugs_expf(scales[i * 3 + 2]));
    bool keep = cugs_sigmoidf(opacities[i]) >= opa_thr;                      // :408-409
    if (size_pruning) {                                                      // :416-439
        if (max_screen > 0.0f) keep = keep && (max_radii[i] <= max_screen);
        keep = keep && (max_scale <= ws_thr);
    }
    const uint32_t clone = high && (max_scale < size_thr);                   // :366-370
    const uint32_t split = high && (max_scale >= size_thr);                  // :393-398
    flags[i] = (uint8_t)(clone | (split << 1) | ((uint32_t)keep << 2));
    if (avg_out) avg_out[i] = avg;
}

__global__ __launch_bounds__(CUGS_BLOCK) void k_count(int64_t n, const uint8_t* __restrict__ flags,
                                                      uint32_t* __restrict__ blocksum, uint32_t nb) {
    __shared__ uint32_t s_tmp[3][4];
    uint32_t k = 0, c = 0, s = 0;
    const int64_t i0 = (int64_t)blockIdx.x * SCAN_CHUNK + threadIdx.x * ITEMS;
#pragma unroll
    for (int e = 0; e < ITEMS; ++e) {
        if (i0 + e < n) {
            uint32_t kk, cc, ss;
            decode(flags[i0 + e], kk, cc, ss);
            k += kk; c += cc; s += ss;
        }
    }
    uint32_t ek, ec, es, tot[3];
    block_excl3(k, c, s, s_tmp, ek, ec, es, tot);
    if (threadIdx.x == 0) {
        blocksum[0 * (size_t)nb + blockIdx.x] = tot[0];
        blocksum[1 * (size_t)nb + blockIdx.x] = tot[1];
        blocksum[2 * (size_t)nb + blockIdx.x] = tot[2];
    }
}

// One workgroup: exclusive scan of the three rows of block sums; totals[] = {K, clones, splits, n_out}.
__global__ __launch_bounds__(CUGS_BLOCK) void k_scan_counts(uint32_t* __restrict__ blocksum, uint32_t nb,
                                                            unsigned long long* __restrict__ totals) {
    __shared__ uint32_t s_tmp[3][4];
    unsigned long long carry[3] = {0, 0, 0};
    uint32_t* row[3] = {blocksum, blocksum + nb, blocksum + 2 * (size_t)nb};
    for (uint32_t base = 0; base < nb; base += CUGS_BLOCK) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t a = i < nb ? row[0][i] : 0u, b = i < nb ? row[1][i] : 0u, c = i < nb ? row[2][i] : 0u;
        uint32_t ea, eb, ec, tot[3];
        block_excl3(a, b, c, s_tmp, ea, eb, ec, tot);
        if (i < nb) {
            row[0][i] = (uint32_t)carry[0] + ea;
            row[1][i] = (uint32_t)carry[1] + eb;
            row[2][i] = (uint32_t)carry[2] + ec;
        }
        carry[0] += tot[0]; carry[1] += tot[1]; carry[2] += tot[2];
    }
    if (threadIdx.x == 0) {
        totals[0] = carry[0]; totals[1] = carry[1]; totals[2] = carry[2];
        totals[3] = carry[0] + carry[1] + 2ull * carry[2];
    }
}

// destination row -> source row, in the final order [kept | clones | children 1 | children 2]
__global__ __launch_bounds__(CUGS_BLOCK) void k_build_map(int64_t n, const uint8_t* __restrict__ flags,
                                                          const uint32_t* __restrict__ blocksum, uint32_t nb,
                                                          const unsigned long long* __restrict__ totals,
                                                          int32_t* __restrict__ src_of) {
    __shared__ uint32_t s_tmp[3][4];
    uint32_t kf[ITEMS], cf[ITEMS], sf[ITEMS], k = 0, c = 0, s = 0;
    const int64_t i0 = (int64_t)blockIdx.x * SCAN_CHUNK + threadIdx.x * ITEMS;
#pragma unroll
    for (int e = 0; e < ITEMS; ++e) {
        kf[e] = cf[e] = sf[e] = 0;
        if (i0 + e < n) decode(flags[i0 + e], kf[e], cf[e], sf[e]);
        k += kf[e]; c += cf[e]; s += sf[e];
    }
    uint32_t ek, ec, es;
    block_excl3(k, c, s, s_tmp, ek, ec, es, nullptr);
    const uint32_t K = (uint32_t)totals[0], C = (uint32_t)totals[1], M = (uint32_t)totals[2];
    uint32_t rk = blocksum[blockIdx.x] + ek;
    uint32_t rc = K + blocksum[(size_t)nb + blockIdx.x] + ec;
    uint32_t rs = K + C + blocksum[2 * (size_t)nb + blockIdx.x] + es;
#pragma unroll
    for (int e = 0; e < ITEMS; ++e) {
        const int32_t src = (int32_t)(i0 + e);
        if (kf[e]) src_of[rk++] = src;
        if (cf[e]) src_of[rc++] = src;
        if (sf[e]) { src_of[rs] = src; src_of[rs + M] = src; ++rs; }
    }
}

// One output float per thread.  Rows below K are survivors, [K, K+C) clones, then the two children sets.
__global__ __launch_bounds__(CUGS_BLOCK) void k_gather_rows(int64_t total, int row_floats, int mode, int64_t n,
                                                            const int32_t* __restrict__ src_of,
                                                            const unsigned long long* __restrict__ totals,
                                                            const float* __restrict__ src, float* __restrict__ dst,
                                                            const float* __restrict__ noise,
                                                            const float* __restrict__ scales) {
    const int64_t e = (int64_t)blockIdx.x * CUGS_BLOCK + threadIdx.x;
    if (e >= total) return;
    const int64_t row = e / row_floats;
    const int col = (int)(e - row * row_floats);
    const int64_t K = (int64_t)totals[0], C = (int64_t)totals[1], M = (int64_t)totals[2];
    const int64_t parent = src_of[row];
    float v = src[parent * row_floats + col];
    if (mode == CUGS_DENSIFY_STATE) {
        if (row >= K) v = 0.0f;                                              // new Gaussians: fresh moments
    } else if (row >= K + C) {                                               // a split child (densification.cpp:245-262)
        const int which = row >= K + C + M ? 1 : 0;
        if (mode == CUGS_DENSIFY_SCALES) {
            v = v - LOG_SPLIT;
        } else if (mode == CUGS_DENSIFY_POSITIONS) {
            const float actual = cugs_expf(scales[parent * 3 + col] - LOG_SPLIT);
            v = v + noise[((int64_t)which * n + parent) * 3 + col] * actual;
        }
    }
    dst[e] = v;
}

// The same for rows that are whole 16-byte groups and need no per-element arithmetic (COPY / STATE: the SH
// coefficients and their moments, 48 floats per row = 81 % of the bytes): one float4 per thread.
__global__ __launch_bounds__(CUGS_BLOCK) void k_gather_rows4(int64_t total4, int row_f4, int mode,
                                                             const int32_t* __restrict__ src_of,
                                                             const unsigned long long* __restrict__ totals,
                                                             const float4* __restrict__ src, float4* __restrict__ dst) {
    const int64_t e = (int64_t)blockIdx.x * CUGS_BLOCK + threadIdx.x;
    if (e >= total4) return;
    const int64_t row = e / row_f4;
    const int col = (int)(e - row * row_f4);
    float4 v = src[(int64_t)src_of[row] * row_f4 + col];
    if (mode == CUGS_DENSIFY_STATE && row >= (int64_t)totals[0]) v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    dst[e] = v;
}

inline int grid_for(int64_t n) { return (int)((n + CUGS_BLOCK - 1) / CUGS_BLOCK); }

}  // namespace

extern "C" int cugs_densify_accumulate(int64_t n, const float* dL_dmeans_2d, const int32_t* radii,
                                       float* grad_accum, float* grad_count, float* max_radii_2d, void* stream) {
    if (n < 0) return CUGS_EINVAL;
    if (n == 0) return 0;
    if (!dL_dmeans_2d || !radii || !grad_accum || !grad_count || !max_radii_2d) return CUGS_EINVAL;
    if ((reinterpret_cast<uintptr_t>(dL_dmeans_2d) & 7u) != 0) return CUGS_EALIGN;
    hipLaunchKernelGGL(k_accumulate, dim3(grid_for(n)), dim3(CUGS_BLOCK), 0, static_cast<hipStream_t>(stream), n,
                       dL_dmeans_2d, radii, grad_accum, grad_count, max_radii_2d);
    CUGS_LAUNCH_CHECK();
    return 0;
}

extern "C" int cugs_densify_classify(int64_t n, const float* grad_accum, const float* grad_count,
                                     const float* max_radii_2d, const float* scales, const float* opacities,
                                     float grad_threshold, float size_threshold, float opacity_threshold,
                                     int apply_size_pruning, float max_screen_size, float ws_threshold,
                                     uint8_t* flags, float* avg_grad, void* stream) {
    if (n < 0) return CUGS_EINVAL;
    if (n == 0) return 0;
    if (!grad_accum || !grad_count || !max_radii_2d || !scales || !opacities || !flags) return CUGS_EINVAL;
    hipLaunchKernelGGL(k_classify, dim3(grid_for(n)), dim3(CUGS_BLOCK), 0, static_cast<hipStream_t>(stream), n,
                       grad_accum, grad_count, max_radii_2d, scales, opacities, grad_threshold, size_threshold,
                       opacity_threshold, apply_size_pruning, max_screen_size, ws_threshold, flags, avg_grad);
    CUGS_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t cugs_densify_workspace_bytes(int64_t n) {
    if (n < 0) return 0;
    return carve(nullptr, n).bytes;
}

extern "C" int cugs_densify_plan(int64_t n, const uint8_t* flags, void* workspace, size_t workspace_bytes,
                                 int64_t counts_host[4], void* stream) {
    if (n < 0 || !counts_host) return CUGS_EINVAL;
    counts_host[0] = counts_host[1] = counts_host[2] = counts_host[3] = 0;
    if (n == 0) return 0;
    if (n > 1073741823ll) return CUGS_EOVERFLOW;                 // n_out <= 2n must fit the int32 row map
    if (!flags || !workspace) return CUGS_EINVAL;
    DensifyWs ws = carve(workspace, n);
    if (workspace_bytes < ws.bytes) return CUGS_EWORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const uint32_t nb = nblocks(n);
    hipLaunchKernelGGL(k_count, dim3(nb), dim3(CUGS_BLOCK), 0, st, n, flags, ws.blocksum, nb);
    CUGS_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(CUGS_BLOCK), 0, st, ws.blocksum, nb, ws.totals);
    CUGS_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_build_map, dim3(nb), dim3(CUGS_BLOCK), 0, st, n, flags, ws.blocksum, nb, ws.totals, ws.src_of);
    CUGS_LAUNCH_CHECK();
    unsigned long long t[4] = {0, 0, 0, 0};
    CUGS_RETURN_IF_HIP(hipMemcpyAsync(t, ws.totals, sizeof(t), hipMemcpyDeviceToHost, st));
    CUGS_RETURN_IF_HIP(hipStreamSynchronize(st));
    for (int i = 0; i < 4; ++i) counts_host[i] = (int64_t)t[i];
    return 0;
}

extern "C" int cugs_densify_apply(int64_t n, int64_t n_out, const void* workspace, size_t workspace_bytes,
                                  const float* noise, const float* scales, const cugs_densify_array* arrays_host,
                                  int num_arrays, void* stream) {
    if (n < 0 || n_out < 0 || num_arrays < 0 || (num_arrays > 0 && !arrays_host)) return CUGS_EINVAL;
    if (n == 0 || n_out == 0 || num_arrays == 0) return 0;
    if (!workspace) return CUGS_EINVAL;
    DensifyWs ws = carve(const_cast<void*>(workspace), n);
    if (workspace_bytes < ws.bytes) return CUGS_EWORKSPACE;
    if (n_out > 2 * n) return CUGS_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int a = 0; a < num_arrays; ++a) {
        const cugs_densify_array& d = arrays_host[a];
        if (!d.src || !d.dst || d.row_floats <= 0) return CUGS_EINVAL;
        if (d.mode < CUGS_DENSIFY_COPY || d.mode > CUGS_DENSIFY_STATE) return CUGS_EINVAL;
        if (d.mode == CUGS_DENSIFY_POSITIONS && (!noise || !scales || d.row_floats != 3)) return CUGS_EINVAL;
        if (d.mode == CUGS_DENSIFY_SCALES && d.row_floats != 3) return CUGS_EINVAL;
        const int64_t total = n_out * d.row_floats;
        const bool plain = d.mode == CUGS_DENSIFY_COPY || d.mode == CUGS_DENSIFY_STATE;
        if (plain && d.row_floats % 4 == 0 && cugs_aligned16(d.src) && cugs_aligned16(d.dst)) {
            hipLaunchKernelGGL(k_gather_rows4, dim3(grid_for(total / 4)), dim3(CUGS_BLOCK), 0, st, total / 4, d.row_floats / 4,
                               d.mode, ws.src_of, ws.totals, reinterpret_cast<const float4*>(d.src),
                               reinterpret_cast<float4*>(d.dst));
            CUGS_LAUNCH_CHECK();
            continue;
        }
        hipLaunchKernelGGL(k_gather_rows, dim3(grid_for(total)), dim3(CUGS_BLOCK), 0, st, total, d.row_floats, d.mode, n,
                           ws.src_of, ws.totals, d.src, d.dst, noise, scales);
        CUGS_LAUNCH_CHECK();
    }
    return 0;
}
