// raster_backward.hip — back-to-front replay of the blend and the 2-D gradient scatter.
//
// Replaces rasterize_backward / k_rasterize_backward (rasterizer/backward.cu:239-306, :31-233) and
// its four zero-fills (backward.cu:259-262).  Reference quirks kept:
//   Q1 contributors are COUNTED from the end of the tile list; the walk stops once the count
//      exceeds the forward's n_contrib (backward.cu:140-145);
//   Q2 T /= max(1-alpha, 1e-5) (:150-151); a clamped alpha (o e^power >= 0.99) zeroes dL/do and
//      dL/dpower but dL/drgb still flows (:181-191);
//   Q3 dL/db is the combined off-diagonal derivative -dx dy (:211).
//
// What is different from the reference is the scatter.  The reference issues nine float atomics
// per (pixel, Gaussian) contribution (backward.cu:217-228).  Here all 64 pixels of a wave look at
// the same Gaussian in the same step, so the nine partials are summed across the wave with DPP row
// operations, across the tile's four waves with ds_add_f32 into a per-batch LDS table, and leave
// the workgroup once per (tile, Gaussian) as ONE atomic wave-instruction segment: 9 consecutive
// floats of a 64-byte-aligned accumulator row (16 lanes per Gaussian, 4 Gaussians per instruction),
// the shape that MI355X's memory-side float atomics serve at full rate (MI355X_MICROARCH.md §Global
// float atomics).  Bytes added per frame: 36 B x P instead of 36 B x (contributions).
// The summation order differs from any sequential order; the oracle accumulates in fp64.
#include "cugs_raster_common.h"

namespace {

constexpr int ACC_STRIDE = 9;
constexpr int ACC_LD = CUGS_BLOCK + 1;   // value k of Gaussian g at k*ACC_LD + g: banks (k+g)%32, conflict-free


template <bool PACKED>
__global__ __launch_bounds__(CUGS_BLOCK) void k_raster_backward(RasterGeom geo, RasterSrc src,
                                                                const float* __restrict__ dL_dcolor,
                                                                const float* __restrict__ final_T,
                                                                const int32_t* __restrict__ n_contrib,
                                                                float* __restrict__ grad_accum) {
    __shared__ float4 s_rec[CUGS_BLOCK * CUGS_REC_F4];
    __shared__ float s_acc[ACC_LD * ACC_STRIDE];
    __shared__ int s_gidx[CUGS_BLOCK];
    __shared__ int s_touched[CUGS_BLOCK];
    __shared__ int s_wave_done[4];

    const unsigned tile = cugs_xcd_remap(blockIdx.x, (unsigned)geo.ntiles);
    const int tile_x = (int)(tile % (unsigned)geo.ntx), tile_y = (int)(tile / (unsigned)geo.ntx);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int quad_x = tile_x * CUGS_TILE + (wave & 1) * 8, quad_y = tile_y * CUGS_TILE + (wave >> 1) * 8;
    const int px = quad_x + (lane & 7), py = quad_y + (lane >> 3);
    const bool inside = (px < geo.width) && (py < geo.height);
    const float pxf = (float)px + 0.5f, pyf = (float)py + 0.5f;
    const float qx0 = (float)quad_x + 0.5f, qy0 = (float)quad_y + 0.5f;

    const int range_start = src.tile_ranges[tile * 2 + 0];
    const int range_end = src.tile_ranges[tile * 2 + 1];
    const int num_in_range = range_end - range_start;
    const int num_batches = (num_in_range + CUGS_BLOCK - 1) / CUGS_BLOCK;

    const int pix = py * geo.width + px;
    float T = inside ? final_T[pix] : 0.0f;
    const int max_contrib = inside ? n_contrib[pix] : 0;
    float dC0 = 0.0f, dC1 = 0.0f, dC2 = 0.0f;
    if (inside) {
        dC0 = dL_dcolor[pix * 3 + 0];
        dC1 = dL_dcolor[pix * 3 + 1];
        dC2 = dL_dcolor[pix * 3 + 2];
    }
    float S0 = T * geo.bg0, S1 = T * geo.bg1, S2 = T * geo.bg2;      // backward.cu:83-87
    int found = 0;
    // A pixel with n_contrib == 0 stops at its first passing Gaussian without contributing
    // (backward.cu:141-145), so it can start out finished.
    bool done = !inside || max_contrib <= 0;
    bool wave_done = (__ballot(!done) == 0ull);

    for (int batch = num_batches - 1; batch >= 0; --batch) {
        if (lane == 0) s_wave_done[wave] = wave_done ? 1 : 0;
        __syncthreads();
        if (s_wave_done[0] & s_wave_done[1] & s_wave_done[2] & s_wave_done[3]) break;

        s_gidx[tid] = stage_record<PACKED>(src, range_start + batch * CUGS_BLOCK + tid, range_end, s_rec);
        s_touched[tid] = 0;
#pragma unroll
        for (int k = 0; k < ACC_STRIDE; ++k) s_acc[k * ACC_LD + tid] = 0.0f;
        __syncthreads();

        if (!wave_done) {
            const int batch_count = min(CUGS_BLOCK, num_in_range - batch * CUGS_BLOCK);
            const int nsub = (batch_count + CUGS_WAVE - 1) / CUGS_WAVE;
            for (int sub = nsub - 1; sub >= 0 && !wave_done; --sub) {
                const int j = sub * CUGS_WAVE + lane;
                bool hit = false;
                if (j < batch_count)
                    hit = may_touch_quad(s_rec[j * CUGS_REC_F4 + 0], s_rec[j * CUGS_REC_F4 + 1],
                                         s_rec[j * CUGS_REC_F4 + 2], qx0, qy0);
                unsigned long long mask = __ballot(hit);
                while (mask) {
                    const int bit = 63 - __builtin_clzll(mask);             // back to front
                    mask &= ~(1ull << bit);
                    const int jj = sub * CUGS_WAVE + bit;
                    const float4 g0 = s_rec[jj * CUGS_REC_F4 + 0];
                    const float4 g1 = s_rec[jj * CUGS_REC_F4 + 1];
                    const float o = s_rec[jj * CUGS_REC_F4 + 2].x;
                    const float a = g0.z, b = g0.w, c = g1.x;

                    float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, v3 = 0.0f, v4 = 0.0f, v5 = 0.0f, v6 = 0.0f,
                          v7 = 0.0f, v8 = 0.0f;
                    bool contrib = false;
                    if (!done) {
                        PixelEval e;
                        if (pixel_alpha(pxf, pyf, g0.x, g0.y, a, b, c, o, e)) {
                            ++found;
                            if (found > max_contrib) {
                                done = true;
                            } else {
                                contrib = true;
                                const float oma = fmaxf(1.0f - e.alpha, 1e-5f);
                                T /= oma;
                                const float weight = e.alpha * T;
                                v0 = dC0 * weight;
                                v1 = dC1 * weight;
                                v2 = dC2 * weight;
                                float dL_dalpha = 0.0f;
                                dL_dalpha += dC0 * (T * g1.y - S0 / oma);
                                dL_dalpha += dC1 * (T * g1.z - S1 / oma);
                                dL_dalpha += dC2 * (T * g1.w - S2 / oma);
                                S0 += weight * g1.y;
                                S1 += weight * g1.z;
                                S2 += weight * g1.w;
                                float dL_dopa = dL_dalpha * e.e;
                                float dL_dpower = dL_dalpha * e.alpha;
                                if (o * e.e >= 0.99f) { dL_dopa = 0.0f; dL_dpower = 0.0f; }
                                v3 = dL_dopa;
                                v4 = dL_dpower * (a * e.dx + b * e.dy);
                                v5 = dL_dpower * (b * e.dx + c * e.dy);
                                v6 = dL_dpower * (-0.5f * e.dx * e.dx);
                                v7 = dL_dpower * (-e.dx * e.dy);
                                v8 = dL_dpower * (-0.5f * e.dy * e.dy);
                            }
                        }
                    }
                    // wave-uniform from here: all 64 lanes take part in the DPP sums
                    if (__ballot(contrib) != 0ull) {
                        v0 = wave_sum_to_row3(v0); v1 = wave_sum_to_row3(v1); v2 = wave_sum_to_row3(v2);
                        v3 = wave_sum_to_row3(v3); v4 = wave_sum_to_row3(v4); v5 = wave_sum_to_row3(v5);
                        v6 = wave_sum_to_row3(v6); v7 = wave_sum_to_row3(v7); v8 = wave_sum_to_row3(v8);
                        const int k = lane - 48;                          // lanes 48..56 carry value k
                        if (k >= 0 && k < ACC_STRIDE) {
                            float mine = v0;
                            mine = (k == 1) ? v1 : mine; mine = (k == 2) ? v2 : mine;
                            mine = (k == 3) ? v3 : mine; mine = (k == 4) ? v4 : mine;
                            mine = (k == 5) ? v5 : mine; mine = (k == 6) ? v6 : mine;
                            mine = (k == 7) ? v7 : mine; mine = (k == 8) ? v8 : mine;
                            atomicAdd(&s_acc[k * ACC_LD + jj], mine);  // ds_add_f32, other waves too
                            if (k == 0) s_touched[jj] = 1;
                        }
                    }
                    if (__ballot(!done) == 0ull) { wave_done = true; break; }
                }
            }
        }
        __syncthreads();

        // One 64-byte accumulator row per touched Gaussian: 16 lanes per row, 9 of them adding.
        {
            const int batch_count = min(CUGS_BLOCK, num_in_range - batch * CUGS_BLOCK);
            for (int e = tid; e < batch_count * 16; e += CUGS_BLOCK) {
                const int g = e >> 4, k = e & 15;
                if (k < ACC_STRIDE && s_touched[g])
                    atomicAdd(&grad_accum[(int64_t)s_gidx[g] * CUGS_GRAD_STRIDE + k], s_acc[k * ACC_LD + g]);
            }
        }
        // the next iteration's first barrier orders these LDS reads before the re-zeroing
    }
}

// grad_accum rows -> the four reference-layout tensors of RasterizeBackwardOutput (backward.hpp).
__global__ __launch_bounds__(CUGS_BLOCK) void k_unpack_grads(int64_t n, const float* __restrict__ acc,
                                                             float* __restrict__ dL_drgb,
                                                             float* __restrict__ dL_dopa,
                                                             float* __restrict__ dL_dmeans,
                                                             float* __restrict__ dL_dcov) {
    const int64_t i = (int64_t)blockIdx.x * CUGS_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float4* row = reinterpret_cast<const float4*>(acc + i * CUGS_GRAD_STRIDE);
    const float4 r0 = row[0], r1 = row[1];
    const float r2 = acc[i * CUGS_GRAD_STRIDE + 8];
    dL_drgb[i * 3 + 0] = r0.x; dL_drgb[i * 3 + 1] = r0.y; dL_drgb[i * 3 + 2] = r0.z;
    dL_dopa[i] = r0.w;
    dL_dmeans[i * 2 + 0] = r1.x; dL_dmeans[i * 2 + 1] = r1.y;
    dL_dcov[i * 3 + 0] = r1.z; dL_dcov[i * 3 + 1] = r1.w; dL_dcov[i * 3 + 2] = r2;
}

}  // namespace

extern "C" int cugs_rasterize_backward(int width, int height, const float background_host[3],
                                       const int32_t* tile_ranges, const int32_t* gaussian_indices,
                                       const float* means_2d, const float* cov_2d_inv, const float* rgb,
                                       const float* opacities_act, const float* packed,
                                       const float* dL_dcolor, const float* final_T,
                                       const int32_t* n_contrib, int64_t n, float* grad_accum,
                                       float* dL_drgb, float* dL_dopacity_act, float* dL_dmeans_2d,
                                       float* dL_dcov_2d_inv, void* stream) {
    if (width < 0 || height < 0 || n < 0 || !background_host) return CUGS_EINVAL;
    if (n == 0) return 0;
    if (!grad_accum) return CUGS_EINVAL;
    if ((reinterpret_cast<uintptr_t>(grad_accum) & 63u) != 0) return CUGS_EALIGN;
    const int n_soa = (dL_drgb != nullptr) + (dL_dopacity_act != nullptr) + (dL_dmeans_2d != nullptr) +
                      (dL_dcov_2d_inv != nullptr);
    if (n_soa != 0 && n_soa != 4) return CUGS_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    CUGS_RETURN_IF_HIP(hipMemsetAsync(grad_accum, 0, sizeof(float) * CUGS_GRAD_STRIDE * (size_t)n, st));

    const int ntx = (width + CUGS_TILE - 1) / CUGS_TILE, nty = (height + CUGS_TILE - 1) / CUGS_TILE;
    if (ntx > 0 && nty > 0 && gaussian_indices) {               // backward.cu:267-269; NULL indices = no pairs
        if (!tile_ranges || !dL_dcolor || !final_T || !n_contrib) return CUGS_EINVAL;
        if (!packed && (!means_2d || !cov_2d_inv || !rgb || !opacities_act)) return CUGS_EINVAL;
        if (packed && !cugs_aligned16(packed)) return CUGS_EALIGN;
        if ((int64_t)width * height > 2147483647ll / 3) return CUGS_EOVERFLOW;
        RasterGeom geo{width, height, ntx, ntx * nty, background_host[0], background_host[1], background_host[2]};
        RasterSrc src{tile_ranges, gaussian_indices, packed, means_2d, cov_2d_inv, rgb, opacities_act};
        if (packed)
            hipLaunchKernelGGL((k_raster_backward<true>), dim3(geo.ntiles), dim3(CUGS_BLOCK), 0, st, geo, src,
                               dL_dcolor, final_T, n_contrib, grad_accum);
        else
            hipLaunchKernelGGL((k_raster_backward<false>), dim3(geo.ntiles), dim3(CUGS_BLOCK), 0, st, geo, src,
                               dL_dcolor, final_T, n_contrib, grad_accum);
        CUGS_LAUNCH_CHECK();
    }
    if (n_soa == 4) {
        hipLaunchKernelGGL(k_unpack_grads, dim3((unsigned)((n + CUGS_BLOCK - 1) / CUGS_BLOCK)), dim3(CUGS_BLOCK),
                           0, st, n, grad_accum, dL_drgb, dL_dopacity_act, dL_dmeans_2d, dL_dcov_2d_inv);
        CUGS_LAUNCH_CHECK();
    }
    return 0;
}
