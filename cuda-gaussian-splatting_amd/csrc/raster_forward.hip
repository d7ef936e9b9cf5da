// raster_forward.hip — per-16x16-tile front-to-back alpha blend.
//
// Replaces rasterize_forward / k_rasterize_forward (rasterizer/forward.cu:180-240, :48-174) and its
// three output fills (forward.cu:197-199).  Semantics kept exactly (they decide n_contrib and the
// image): pixel centre px+0.5 (:72-73); skip if power > 0 (:135); alpha = min(0.99, o e^power)
// (:137-140); skip if alpha < 1/255 (:141); C += alpha T rgb, T *= 1-alpha, count++ and THEN stop
// when T < 1/255 (:143-156); out = C + T bg (:165-167).
//
// gfx950 design: see cugs_raster_common.h (4 waves x 8x8 quads, LDS-staged 256-record batches,
// wave64 ballot compaction of the records each quad can see).  The inner loop is VALU-bound
// (~35 instructions per surviving (pixel, Gaussian) evaluation); HBM traffic is the 4 B index +
// 48 B record per (tile, Gaussian) pair plus 20 B per pixel written.
#include "cugs_raster_common.h"

namespace {

template <bool PACKED>
__global__ __launch_bounds__(CUGS_BLOCK) void k_raster_forward(RasterGeom geo, RasterSrc src,
                                                               float* __restrict__ out_color,
                                                               float* __restrict__ out_final_T,
                                                               int32_t* __restrict__ out_n_contrib,
                                                               float4* __restrict__ zero_buf, uint32_t zero_vec4) {
    __shared__ float4 s_rec[CUGS_BLOCK * CUGS_REC_F4];
    __shared__ int s_wave_done[4];

    // Housekeeping for the backward (cugs_rasterize_forward_zero): this kernel is bound by instruction issue and
    // leaves HBM nearly idle, so each workgroup clears its slice of the gradient accumulator here - the 64 B/Gaussian
    // fill that otherwise runs by itself in front of the backward blend.
    if (zero_buf) {
        const uint32_t per = (zero_vec4 + gridDim.x - 1) / gridDim.x;
        const uint32_t lo = blockIdx.x * per, hi = min(zero_vec4, lo + per);
        for (uint32_t e = lo + threadIdx.x; e < hi; e += CUGS_BLOCK) cugs_stnt(zero_buf + e, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    }

    // heaviest tile first when the caller brings an order (cugs_tile_order: kernel-uniform), else the spatial order
    // heaviest tile first when the caller brings an order (cugs_tile_order: kernel-uniform), else the spatial order
    unsigned tile;
    int range_start, range_end;
    if (src.tile_order) {
        const uint4 rec = src.tile_order[blockIdx.x];
        tile = rec.x; range_start = (int)rec.y; range_end = (int)rec.z;
    } else {
        tile = cugs_blend_tile(blockIdx.x, (unsigned)geo.ntx, (unsigned)(geo.ntiles / geo.ntx));
        range_start = src.tile_ranges[tile * 2 + 0];
        range_end = src.tile_ranges[tile * 2 + 1];
    }
    const int tile_x = (int)(tile % (unsigned)geo.ntx), tile_y = (int)(tile / (unsigned)geo.ntx);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int quad_x = tile_x * CUGS_TILE + (wave & 1) * 8, quad_y = tile_y * CUGS_TILE + (wave >> 1) * 8;
    const int px = quad_x + (lane & 7), py = quad_y + (lane >> 3);
    const bool inside = (px < geo.width) && (py < geo.height);
    const float pxf = (float)px + 0.5f, pyf = (float)py + 0.5f;
    const float qx0 = (float)quad_x + 0.5f, qy0 = (float)quad_y + 0.5f;

    const int num_in_range = range_end - range_start;
    const int num_batches = (num_in_range + CUGS_BLOCK - 1) / CUGS_BLOCK;

    typedef float v2f __attribute__((ext_vector_type(2)));
    float T = 1.0f, C0 = 0.0f;
    v2f C12 = {0.0f, 0.0f};                             // green/blue as one packed-fp32 accumulator (v_pk_fma_f32)
    float count = 0.0f;                                 // contributors: a float counter (exact below 2^24 per tile list)
    float open = inside ? 1.0f : 0.0f;                  // 1 while the pixel still blends, 0 once T < 1/255
    bool wave_done = (__ballot(open != 0.0f) == 0ull);

    for (int batch = 0; batch < num_batches; ++batch) {
        // whole-tile early exit (forward.cu:97-101), one flag per wave instead of an atomicMin
        if (lane == 0) s_wave_done[wave] = wave_done ? 1 : 0;
        __syncthreads();
        if (s_wave_done[0] & s_wave_done[1] & s_wave_done[2] & s_wave_done[3]) break;

        stage_record<PACKED>(src, range_start + batch * CUGS_BLOCK + tid, range_end, s_rec);
        __syncthreads();

        if (!wave_done) {
            const int batch_count = min(CUGS_BLOCK, num_in_range - batch * CUGS_BLOCK);
            for (int sub = 0; sub * CUGS_WAVE < batch_count && !wave_done; ++sub) {
                const int j = sub * CUGS_WAVE + lane;
                const ActiveRect ar = active_rect(__ballot(open != 0.0f), qx0, qy0);   // !wave_done => non-empty
                bool hit = false;
                if (j < batch_count)
                    hit = may_touch_quad(s_rec[j * CUGS_REC_F4 + 0], s_rec[j * CUGS_REC_F4 + 1], ar.x0, ar.y0, ar.wx,
                                         ar.wy);
                unsigned long long mask = __ballot(hit);
                while (mask != 0ull) {                                      // front to back
                    const float4* rp = s_rec + (sub * CUGS_WAVE + __builtin_ctzll(mask)) * CUGS_REC_F4;
                    mask &= mask - 1ull;
                    const float4 g0 = rp[0], g1 = rp[1], col = rp[2];       // wave-uniform address: broadcast
                    const float o = g1.y;
                    PixelEval e;
                    // decisions as 0/1 floats (v_fma ... clamp, cugs_raster_common.h): no v_cmp / v_cndmask pairs
                    const float alpha = pixel_alpha_raw(pxf, pyf, g0.x, g0.y, g0.z, g0.w, g1.x, o, open, e);
                    const float passf = passes_alpha_min(alpha);           // alpha >= 1/255 (forward.cu:141)
                    const float al = alpha * passf;
                    // al == 0 (skipped or finished pixel) leaves C, T and count untouched exactly
                    const float weight = al * T;
                    C0 = fmaf(weight, col.x, C0);
                    C12 = __builtin_elementwise_fma((v2f){weight, weight}, (v2f){col.z, col.w}, C12);   // record words 10,11: an aligned pair
                    T *= (1.0f - al);
                    count += passf;
                    open *= passes_alpha_min(T);                            // T < 1/255 -> done (forward.cu:150-156): the same
                                                                            // threshold; only a passing Gaussian can lower T
                    if (__ballot(open != 0.0f) == 0ull) { wave_done = true; break; }
                }
            }
        }
    }

    if (inside) {
        const int pix = py * geo.width + px;
        out_color[pix * 3 + 0] = fmaf(T, geo.bg0, C0);
        out_color[pix * 3 + 1] = fmaf(T, geo.bg1, C12.x);
        out_color[pix * 3 + 2] = fmaf(T, geo.bg2, C12.y);
        out_final_T[pix] = T;
        out_n_contrib[pix] = (int)count;
    }
}

}  // namespace

namespace {
int rasterize_forward_impl(int width, int height, const float background_host[3], const int32_t* tile_ranges,
                           const int32_t* gaussian_indices, const float* means_2d, const float* cov_2d_inv,
                           const float* rgb, const float* opacities_act, const float* packed, float* out_color,
                           float* out_final_T, int32_t* out_n_contrib, void* zero_buf, size_t zero_bytes,
                           const uint32_t* tile_order, void* stream) {
    if (width < 0 || height < 0 || !background_host) return CUGS_EINVAL;
    if (zero_bytes && (!zero_buf || (reinterpret_cast<uintptr_t>(zero_buf) & 15u) || (zero_bytes & 15u) ||
                       zero_bytes / 16 > 0xFFFFFFFFull))
        return zero_buf ? CUGS_EALIGN : CUGS_EINVAL;
    const int ntx = (width + CUGS_TILE - 1) / CUGS_TILE, nty = (height + CUGS_TILE - 1) / CUGS_TILE;
    if (ntx == 0 || nty == 0) {                               // forward.cu:204-210: nothing to draw
        if (zero_bytes) CUGS_RETURN_IF_HIP(hipMemsetAsync(zero_buf, 0, zero_bytes, static_cast<hipStream_t>(stream)));
        return 0;
    }
    if (!tile_ranges || !out_color || !out_final_T || !out_n_contrib) return CUGS_EINVAL;
    // gaussian_indices and the per-Gaussian sources may be NULL for an empty pair list (P == 0: every
    // tile range is {0,0} and nothing is dereferenced).  With indices present a source is required.
    if (gaussian_indices && !packed && (!means_2d || !cov_2d_inv || !rgb || !opacities_act)) return CUGS_EINVAL;
    if (packed && !cugs_aligned16(packed)) return CUGS_EALIGN;
    if (tile_order && !cugs_aligned16(tile_order)) return CUGS_EALIGN;
    if ((int64_t)width * height > 2147483647ll / 3) return CUGS_EOVERFLOW;
    RasterGeom geo{width, height, ntx, ntx * nty, background_host[0], background_host[1], background_host[2]};
    RasterSrc src{tile_ranges, gaussian_indices, packed, means_2d, cov_2d_inv, rgb, opacities_act, reinterpret_cast<const uint4*>(tile_order)};
    hipStream_t st = static_cast<hipStream_t>(stream);
    float4* zb = zero_bytes ? static_cast<float4*>(zero_buf) : nullptr;
    const uint32_t zv = (uint32_t)(zero_bytes / 16);
    if (packed)
        hipLaunchKernelGGL((k_raster_forward<true>), dim3(geo.ntiles), dim3(CUGS_BLOCK), 0, st, geo, src,
                           out_color, out_final_T, out_n_contrib, zb, zv);
    else
        hipLaunchKernelGGL((k_raster_forward<false>), dim3(geo.ntiles), dim3(CUGS_BLOCK), 0, st, geo, src,
                           out_color, out_final_T, out_n_contrib, zb, zv);
    CUGS_LAUNCH_CHECK();
    return 0;
}
}  // namespace

extern "C" int cugs_rasterize_forward(int width, int height, const float background_host[3],
                                      const int32_t* tile_ranges, const int32_t* gaussian_indices,
                                      const float* means_2d, const float* cov_2d_inv, const float* rgb,
                                      const float* opacities_act, const float* packed, float* out_color,
                                      float* out_final_T, int32_t* out_n_contrib, void* stream) {
    return rasterize_forward_impl(width, height, background_host, tile_ranges, gaussian_indices, means_2d, cov_2d_inv, rgb,
                                  opacities_act, packed, out_color, out_final_T, out_n_contrib, nullptr, 0, nullptr, stream);
}

extern "C" int cugs_rasterize_forward_zero(int width, int height, const float background_host[3],
                                           const int32_t* tile_ranges, const int32_t* gaussian_indices,
                                           const float* means_2d, const float* cov_2d_inv, const float* rgb,
                                           const float* opacities_act, const float* packed, float* out_color,
                                           float* out_final_T, int32_t* out_n_contrib, void* zero_buf,
                                           size_t zero_bytes, void* stream) {
    return rasterize_forward_impl(width, height, background_host, tile_ranges, gaussian_indices, means_2d, cov_2d_inv, rgb,
                                  opacities_act, packed, out_color, out_final_T, out_n_contrib, zero_buf, zero_bytes, nullptr, stream);
}

extern "C" int cugs_rasterize_forward_ordered(int width, int height, const float background_host[3],
                                              const int32_t* tile_ranges, const int32_t* gaussian_indices,
                                              const float* means_2d, const float* cov_2d_inv, const float* rgb,
                                              const float* opacities_act, const float* packed, float* out_color,
                                              float* out_final_T, int32_t* out_n_contrib, void* zero_buf,
                                              size_t zero_bytes, const uint32_t* tile_order, void* stream) {
    return rasterize_forward_impl(width, height, background_host, tile_ranges, gaussian_indices, means_2d, cov_2d_inv, rgb,
                                  opacities_act, packed, out_color, out_final_T, out_n_contrib, zero_buf, zero_bytes, tile_order,
                                  stream);
}
