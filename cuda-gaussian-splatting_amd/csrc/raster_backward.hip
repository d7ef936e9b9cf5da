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
// the same Gaussian in the same step, so the nine partials are first summed across the wave
// (reduce9, cugs_raster_common.h: 27 operations, each total landing in a different lane) and then
// leave the wave as ONE atomic wave-instruction whose nine active lanes hit nine consecutive floats
// of the Gaussian's 64-byte-aligned accumulator row - a single memory-side request
// (MI355X_MICROARCH.md, Global float atomics).  Measured (profiles/r01 ablation): the atomics are
// free next to the evaluation; an earlier LDS stage that merged the four waves of a tile first cost
// more in zero/flush/barrier overhead than it saved in requests.
// The summation order differs from any sequential order; the oracle accumulates in fp64.
// Per-contribution VALUES (not decisions) use v_rcp_f32 and fused multiply-adds: 1 ulp-level
// differences from the oracle's divisions, far inside the 1e-4 bar.
#include "cugs_raster_common.h"

#include <stdlib.h>

namespace {



// ABL: 0 = product; 1..3 = timing-only ablations selected by CUGS_BWD_ABLATE (tools/ablate_backward.py):
// 1 no wave reduction, 2 no global atomics, 3 no per-pixel evaluation, 4 = product + step counters written to
// accumulator row n (the tool allocates n+1 rows).  Outputs are wrong for ABL 1..3.
template <bool PACKED, int ABL>
__global__ __launch_bounds__(CUGS_BLOCK) void k_raster_backward(RasterGeom geo, RasterSrc src,
                                                                const float* __restrict__ dL_dcolor,
                                                                const float* __restrict__ final_T,
                                                                const int32_t* __restrict__ n_contrib,
                                                                float* __restrict__ grad_accum, int64_t stats_row) {
    __shared__ float4 s_rec[CUGS_BLOCK * CUGS_REC_F4];
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
    // (backward.cu:141-145), so it can start out finished.  `open`: 1 while the pixel is still walking.
    float open = (inside && max_contrib > 0) ? 1.0f : 0.0f;
    bool wave_done = (__ballot(open != 0.0f) == 0ull);
    const int my_slot = reduce9_slot(lane);
    unsigned st_steps = 0, st_contrib = 0, st_lanes = 0, st_batches = 0, st_tested = 0, st_open = 0;   // ABL == 4 only

    for (int batch = num_batches - 1; batch >= 0; --batch) {
        if (lane == 0) s_wave_done[wave] = wave_done ? 1 : 0;
        __syncthreads();
        if (s_wave_done[0] & s_wave_done[1] & s_wave_done[2] & s_wave_done[3]) break;

        stage_record<PACKED>(src, range_start + batch * CUGS_BLOCK + tid, range_end, s_rec);
        __syncthreads();

        if (ABL == 4 && !wave_done) ++st_batches;
        if (!wave_done) {
            const int batch_count = min(CUGS_BLOCK, num_in_range - batch * CUGS_BLOCK);
            const int nsub = (batch_count + CUGS_WAVE - 1) / CUGS_WAVE;
            for (int sub = nsub - 1; sub >= 0 && !wave_done; --sub) {
                const int j = sub * CUGS_WAVE + lane;
                const ActiveRect ar = active_rect(__ballot(open != 0.0f), qx0, qy0);   // !wave_done => non-empty
                bool hit = false;
                if (j < batch_count)
                    hit = may_touch_quad(s_rec[j * CUGS_REC_F4 + 0], s_rec[j * CUGS_REC_F4 + 1],
                                         s_rec[j * CUGS_REC_F4 + 2], ar.x0, ar.y0, ar.wx, ar.wy);
                unsigned long long mask = __ballot(hit);
                if (ABL == 4) st_tested += min(CUGS_WAVE, batch_count - sub * CUGS_WAVE);
                while (mask != 0ull) {                                     // back to front: highest bit first
                    if (ABL == 4) ++st_steps;
                    const int bit = 63 - __builtin_clzll(mask);
                    mask &= ~(1ull << bit);
                    const float4* rp = s_rec + (sub * CUGS_WAVE + bit) * CUGS_REC_F4;
                    const float4 g0 = rp[0], g1 = rp[1], g2 = rp[2];
                    const float a = g0.z, b = g0.w, c = g1.x, o = g2.x;

                    // ---- decisions (backward.cu:123-145), all in vector registers: see pixel_alpha()
                    PixelEval e;
                    const float alpha_p = (ABL == 3) ? 0.0f : pixel_alpha(pxf, pyf, g0.x, g0.y, a, b, c, o, open, e);
                    found += (alpha_p != 0.0f) ? 1 : 0;                    // contributors counted from the END (Q1)
                    const float al = (found <= max_contrib) ? alpha_p : 0.0f;
                    open = (alpha_p != al) ? 0.0f : open;                  // passed, but beyond n_contrib: finished

                    // ---- values (v_rcp_f32 + FMAs); al == 0 makes every sum below exactly zero
                    const bool live = (al != 0.0f);
                    const float rcp = __builtin_amdgcn_rcpf(fmaxf(1.0f - al, 1e-5f));
                    T = live ? T * rcp : T;                                // T_before = T_after / (1 - alpha)
                    const float weight = al * T;
                    float v0 = dC0 * weight, v1 = dC1 * weight, v2 = dC2 * weight;
                    float dL_dalpha = dC0 * fmaf(T, g1.y, -S0 * rcp);
                    dL_dalpha = fmaf(dC1, fmaf(T, g1.z, -S1 * rcp), dL_dalpha);
                    dL_dalpha = fmaf(dC2, fmaf(T, g1.w, -S2 * rcp), dL_dalpha);
                    S0 = fmaf(weight, g1.y, S0);
                    S1 = fmaf(weight, g1.z, S1);
                    S2 = fmaf(weight, g1.w, S2);
                    // clamp gate (backward.cu:181-191): o e >= 0.99 zeroes dL/do and dL/dpower, dL/drgb still flows
                    const float gate = (o * e.e >= 0.99f) ? 0.0f : dL_dalpha;
                    float v3 = live ? gate * e.e : 0.0f;
                    const float dpw = gate * al;                           // dL/dpower
                    float v4 = dpw * e.gx, v5 = dpw * e.gy;
                    const float hdp = -0.5f * dpw;
                    float v6 = hdp * e.dx * e.dx, v7 = -dpw * e.dx * e.dy, v8 = hdp * e.dy * e.dy;
                    if (ABL == 4) {
                        const unsigned long long lm = __ballot(live);
                        if (lm) { ++st_contrib; st_lanes += __popcll(lm); st_open += __popcll(__ballot(open != 0.0f)); }
                    }
                    // Every step reduces and adds (93% of steps contribute; an all-zero add is harmless and
                    // cheaper than the scalar test-and-branch that would skip it).
                    if (ABL == 1) {
                        asm volatile("" ::"v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v7), "v"(v8));
                    } else {
                        const float total = reduce9(v0, v1, v2, v3, v4, v5, v6, v7, v8, lane);
                        if (ABL == 2) {
                            asm volatile("" ::"v"(total));
                        } else if (my_slot >= 0) {
                            atomicAdd(&grad_accum[(int64_t)__float_as_int(g2.z) * CUGS_GRAD_STRIDE + my_slot], total);
                        }
                    }
                    if (__ballot(open != 0.0f) == 0ull) { wave_done = true; break; }
                }
            }
        }
        // the next iteration's first barrier orders this batch's LDS reads before the re-staging
    }
    if (ABL == 4 && lane == 0) {
        float* st = grad_accum + (int64_t)stats_row * CUGS_GRAD_STRIDE;
        atomicAdd(&st[0], (float)st_steps); atomicAdd(&st[1], (float)st_contrib); atomicAdd(&st[2], (float)st_lanes);
        atomicAdd(&st[3], (float)st_batches); atomicAdd(&st[4], (float)st_tested); atomicAdd(&st[5], (float)num_batches);
        atomicAdd(&st[7], (float)st_open);
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
        const char* abl_env = getenv("CUGS_BWD_ABLATE");          // timing experiments only
        const int abl = abl_env ? atoi(abl_env) : 0;
#define CUGS_LAUNCH_BWD(P, A)                                                                              \
    hipLaunchKernelGGL((k_raster_backward<P, A>), dim3(geo.ntiles), dim3(CUGS_BLOCK), 0, st, geo, src, \
                       dL_dcolor, final_T, n_contrib, grad_accum, n)
        if (packed) {
            if (abl == 1) CUGS_LAUNCH_BWD(true, 1);
            else if (abl == 2) CUGS_LAUNCH_BWD(true, 2);
            else if (abl == 3) CUGS_LAUNCH_BWD(true, 3);
            else if (abl == 4) CUGS_LAUNCH_BWD(true, 4);
            else CUGS_LAUNCH_BWD(true, 0);
        } else {
            CUGS_LAUNCH_BWD(false, 0);
        }
#undef CUGS_LAUNCH_BWD
        CUGS_LAUNCH_CHECK();
    }
    if (n_soa == 4) {
        hipLaunchKernelGGL(k_unpack_grads, dim3((unsigned)((n + CUGS_BLOCK - 1) / CUGS_BLOCK)), dim3(CUGS_BLOCK),
                           0, st, n, grad_accum, dL_drgb, dL_dopacity_act, dL_dmeans_2d, dL_dcov_2d_inv);
        CUGS_LAUNCH_CHECK();
    }
    return 0;
}

// ---- test hook (not part of include/cugs_hip.h): one wave runs reduce9 on caller data ----------
// in: [9][64] floats (value k of lane l at k*64+l); out: [64] floats (each lane's result), slots: [64] ints.
namespace {
__global__ void k_dbg_reduce9(const float* __restrict__ in, float* __restrict__ out, int* __restrict__ slots) {
    const int l = threadIdx.x;
    out[l] = reduce9(in[0 * 64 + l], in[1 * 64 + l], in[2 * 64 + l], in[3 * 64 + l], in[4 * 64 + l], in[5 * 64 + l],
                     in[6 * 64 + l], in[7 * 64 + l], in[8 * 64 + l], l);
    slots[l] = reduce9_slot(l);
}
}  // namespace
extern "C" int cugsdbg_reduce9(const float* in, float* out, int* slots, void* stream) {
    hipLaunchKernelGGL(k_dbg_reduce9, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), in, out, slots);
    CUGS_LAUNCH_CHECK();
    return 0;
}
