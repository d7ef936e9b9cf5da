// raster_backward.hip — back-to-front replay of the blend and the 2-D gradient scatter.
//
// Replaces rasterize_backward / k_rasterize_backward (rasterizer/backward.cu:239-306, :31-233) and
// its four zero-fills (backward.cu:259-262).  Reference quirks kept:
//   Q1 contributors are COUNTED from the end of the tile list; the walk stops once the count
//      exceeds the forward's n_contrib (backward.cu:140-145);
//   Q2 T /= max(1-alpha, 1e-5) (:150-151; alpha <= 0.99 makes the max a no-op); a clamped alpha
//      (o e^power >= 0.99) zeroes dL/do and dL/dpower but dL/drgb still flows (:181-191);
//   Q3 dL/db is the combined off-diagonal derivative -dx dy (:211).
//
// What is different from the reference is the scatter.  The reference issues nine float atomics
// per (pixel, Gaussian) contribution (backward.cu:217-228).  Here all 64 pixels of a wave look at
// the same Gaussian in the same step, so the partials are summed across the wave first (through LDS and a
// 16-lane DPP reduction, see the kernel's comment) and leave it as ONE atomic wave-instruction per four
// Gaussians, nine active lanes per Gaussian hitting nine consecutive floats of its 64-byte-aligned
// accumulator row - a single memory-side request each (MI355X_MICROARCH.md, Global float atomics).
//
// The kernel is bound by VALU issue (profiles/README.md), so the step is written for instruction
// count, priced with the measured costs in cugs_raster_common.h:
//   * every per-lane decision is a 0/1 float made by `v_fma ... clamp` and multiplied in: no v_cmp /
//     v_cndmask pairs.  The Q1 counter is `rem` = n_contrib - passers so far, `open` = sat(rem + 1);
//   * the colour accumulators S_c (backward.cu:83-87, 196-198) only ever enter as sum_c dL/dC_c S_c:
//     that ONE dot product D is carried instead (D += weight * G, G = sum_c dL/dC_c c_c);
//   * the five geometric gradients are accumulated as the MOMENTS of dL/dpower over the pixel offsets,
//     M1 = sum dpw (dx, dy), M2 = sum dpw (dx^2, dx dy, dy^2) - two products fewer per step than
//     dpw * (a dx + b dy) etc.; the per-Gaussian linear map to dL/dmean2d = Sigma'^-1 M1 and
//     dL/dSigma'^-1 = (-M2xx/2, -M2xy, -M2yy/2) is applied once per Gaussian by the consumer
//     (k_project_backward / k_unpack_grads).  Accumulator row: {drgb[3], dopa, M1x, M1y, M2xx, M2xy, M2yy};
//   * T *= rcp(1 - al) needs no "did it contribute" select: v_rcp_f32(1.0f) is exactly 1.0f
//     (tests/test_gpu_reduce9.py checks it on the device).
// The summation order differs from any sequential order; the oracle accumulates in fp64.
// Per-contribution VALUES (not decisions) use v_rcp_f32 and fused multiply-adds: 1 ulp-level
// differences from the oracle's divisions, far inside the 1e-4 bar.
#include "cugs_raster_common.h"

#include <cstdlib>

#ifdef CUGS_DEV
bool cugs_dev_backward_stats();
#endif

namespace {

// Two phases per wave:
//   phase 1, once per (wave, Gaussian) step, lane = pixel of the wave's 8x8 quad: the decisions, the T and D
//     recurrences and the two per-pixel scalars every gradient of this contribution is made of -
//     weight = alpha T (for dL/dcolour) and v3 = gated dL/dalpha * e (dL/dopacity; dL/dpower = v3 * opacity) -
//     written to LDS as one 8-byte store per lane;
//   phase 2, once per CUGS_BWD_HITS steps, lane = (Gaussian h = lane / 16, pixel group g = lane % 16 of four
//     horizontally adjacent pixels): reads its Gaussian's contributions back, forms the nine sums over its four
//     pixels (the per-lane constants are the pixels' dL/dcolour and coordinates; dy is common to the four, so
//     sum dpw dy, dx dy, dy^2 follow from the others), reduces them across the 16 lanes of its DPP row
//     (reduce9r16) and issues one atomic instruction: nine lanes per Gaussian, one 64-byte request each.
// Against reducing nine values over 64 lanes in every step (38 units) this costs ~19 units per step; the pending
// Gaussians are flushed before the record batch they point into is re-staged.
#define CUGS_BWD_HITS 4
#ifndef CUGS_BWD_HSTRIDE
#define CUGS_BWD_HSTRIDE 64        /* float2 per Gaussian block: the 64 pixels, no padding - see the LDS budget below */
#endif

// WIDE: accumulator larger than 4 GiB (n > 2^26 rows): 64-bit scatter addresses.
// STATS (dev builds only): step counters written to accumulator row `stats_row` (tools/ablate_backward.py).
template <bool PACKED, bool WIDE, bool STATS>
__global__ __launch_bounds__(CUGS_BLOCK) void k_raster_backward(RasterGeom geo, RasterSrc src,
                                                                const float* __restrict__ dL_dcolor,
                                                                const float* __restrict__ final_T,
                                                                const int32_t* __restrict__ n_contrib,
                                                                float* __restrict__ grad_accum, int64_t stats_row) {
    // LDS budget: 12288 B of records + 8192 B of contributions = 20480 B = 1/8 of a CU's 160 KB, so EIGHT workgroups
    // (32 waves, the CU's limit) are resident instead of the seven that 20.8 KB allowed (round 2: 16 bytes of padding
    // per contribution block, 64 + 16 bytes of hit-record and vote words).  The record index of each pending Gaussian
    // travels in a wave-uniform 64-bit scalar (16 bits each), and a wave's "all my pixels are done" vote sits in the
    // first word of its own - at that point idle - contribution block (see the loop head).
    __shared__ float4 s_rec[CUGS_BLOCK * CUGS_REC_F4];
    __shared__ float2 s_contrib[4][CUGS_BWD_HITS][CUGS_BWD_HSTRIDE];

#ifdef CUGS_DEV
    const bool no_atomics = stats_row == -2;       // development build: the kernel without its scatter (timing only; tools/ablate_backward.py)
#else
    constexpr bool no_atomics = false;
#endif
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

    const int pix = py * geo.width + px;
    float T = inside ? final_T[pix] : 0.0f;
    const int max_contrib = inside ? n_contrib[pix] : 0;
    float dC0 = 0.0f, dC1 = 0.0f, dC2 = 0.0f;
    if (inside) {
        dC0 = dL_dcolor[pix * 3 + 0];
        dC1 = dL_dcolor[pix * 3 + 1];
        dC2 = dL_dcolor[pix * 3 + 2];
    }
    // ---- phase-2 constants: this lane's Gaussian slot h and its four pixels (row gy, columns gx0 .. gx0+3)
    const int h2 = lane >> 4, g2i = lane & 15;
    const int gx0 = quad_x + (g2i & 1) * 4, gy = quad_y + (g2i >> 1);
    const float px2 = (float)gx0 + 0.5f, py2 = (float)gy + 0.5f;
    // reduce9r16 takes its first pair swizzled: red in lanes with bit 3 clear, green in the others
    const bool side = (lane & 8) != 0;
    float eA[4], eB[4], e2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float r = 0.0f, gg = 0.0f, bb = 0.0f;
        if (gx0 + i < geo.width && gy < geo.height) {
            const int q = (gy * geo.width + gx0 + i) * 3;
            r = dL_dcolor[q + 0]; gg = dL_dcolor[q + 1]; bb = dL_dcolor[q + 2];
        }
        eA[i] = side ? gg : r; eB[i] = side ? r : gg; e2[i] = bb;
    }
    const int slot2 = reduce9r16_slot(lane);
    const unsigned slot_off = (unsigned)(slot2 < 0 ? 0 : slot2) * 4u;
    unsigned long long hitrecs = 0ull;                       // record (float4) index of pending Gaussian h in bits [16h, 16h+16);
                                                             // stale slots of a partial flush stay valid indices

    // D = sum_c dL/dC_c * (colour accumulated behind the current Gaussian), starting from the background
    float D = fmaf(dC2, T * geo.bg2, fmaf(dC1, T * geo.bg1, dC0 * (T * geo.bg0)));      // backward.cu:83-87
    // rem = n_contrib - (passing Gaussians seen so far, counted from the END: Q1).  A pixel stops - without
    // contributing - at the passer that takes rem below zero (backward.cu:141-145); a pixel with
    // n_contrib == 0, or outside the image, starts out finished.  open == sat(rem + 1) throughout.
    float rem = (inside && max_contrib > 0) ? (float)max_contrib : -1.0f;
    float open = (inside && max_contrib > 0) ? 1.0f : 0.0f;
    bool wave_done = (__ballot(open != 0.0f) == 0ull);
    int pending = 0;                                            // Gaussians waiting in s_contrib (wave-uniform)
    unsigned st_steps = 0, st_contrib = 0, st_lanes = 0, st_batches = 0, st_tested = 0, st_open = 0;   // STATS only

    // phase 2 for the `cnt` pending Gaussians of this wave
    auto flush = [&](int cnt) {
        const int rec = (int)((hitrecs >> (16 * h2)) & 0xFFFFull);                     // float4 index of the record
        const float4* c4 = reinterpret_cast<const float4*>(&s_contrib[wave][h2][g2i * 4]);
        const float4 p01 = c4[0], p23 = c4[1];                                         // (weight, v3) x 4 pixels
        const float2 mean = *reinterpret_cast<const float2*>(&s_rec[rec]);
        const float4 tail = s_rec[rec + 1];                                            // c, opacity, tau, index
        const float dy = py2 - mean.y;
        const float dx0 = px2 - mean.x, dx1 = dx0 + 1.0f, dx2 = dx0 + 2.0f, dx3 = dx0 + 3.0f;
        float t = p01.y * dx0;
        float A = p01.y, M1 = t, Mxx = t * dx0;
        float RA = p01.x * eA[0], RB = p01.x * eB[0], R2 = p01.x * e2[0];
        t = p01.w * dx1; A += p01.w; M1 += t; Mxx = fmaf(t, dx1, Mxx);
        RA = fmaf(p01.z, eA[1], RA); RB = fmaf(p01.z, eB[1], RB); R2 = fmaf(p01.z, e2[1], R2);
        t = p23.y * dx2; A += p23.y; M1 += t; Mxx = fmaf(t, dx2, Mxx);
        RA = fmaf(p23.x, eA[2], RA); RB = fmaf(p23.x, eB[2], RB); R2 = fmaf(p23.x, e2[2], R2);
        t = p23.w * dx3; A += p23.w; M1 += t; Mxx = fmaf(t, dx3, Mxx);
        RA = fmaf(p23.z, eA[3], RA); RB = fmaf(p23.z, eB[3], RB); R2 = fmaf(p23.z, e2[3], R2);
        // dL/dpower = v3 * opacity (alpha = opacity * e where the gate is open); dy is common to the four pixels
        const float oA = tail.y * A;
        M1 *= tail.y; Mxx *= tail.y;
        const float M1y = dy * oA, Myy = dy * M1y, Mxy = dy * M1;
        const float total = reduce9r16(RA, RB, R2, A, M1, M1y, Mxx, Myy, Mxy, lane);
        if (slot2 >= 0 && h2 < cnt && !no_atomics) {
            const int g = __float_as_int(tail.w);
            if (WIDE) {
                atomicAdd(grad_accum + (int64_t)g * CUGS_GRAD_STRIDE + slot2, total);
            } else {
                const unsigned off = ((unsigned)g << 6) + slot_off;               // row bytes = 4 * CUGS_GRAD_STRIDE = 64
                atomicAdd(reinterpret_cast<float*>(reinterpret_cast<char*>(grad_accum) + off), total);
            }
        }
    };

    for (int batch = num_batches - 1; batch >= 0; --batch) {
        // The vote word is the first word of the wave's own contribution block: every pending Gaussian has been flushed
        // when a wave arrives here, nobody else touches the block, all four waves read the words between this barrier
        // and the next, and the block is written again (phase 1) only after that next barrier.
        if (lane == 0) s_contrib[wave][0][0].x = wave_done ? 1.0f : 0.0f;
        __syncthreads();
        if (s_contrib[0][0][0].x * s_contrib[1][0][0].x * s_contrib[2][0][0].x * s_contrib[3][0][0].x != 0.0f) break;

        stage_record<PACKED>(src, range_start + batch * CUGS_BLOCK + tid, range_end, s_rec);
        __syncthreads();

        if (STATS && !wave_done) ++st_batches;
        if (!wave_done) {
            const int batch_count = min(CUGS_BLOCK, num_in_range - batch * CUGS_BLOCK);
            const int nsub = (batch_count + CUGS_WAVE - 1) / CUGS_WAVE;
            for (int sub = nsub - 1; sub >= 0 && !wave_done; --sub) {
                const int j = sub * CUGS_WAVE + lane;
                const ActiveRect ar = active_rect(__ballot(open != 0.0f), qx0, qy0);   // !wave_done => non-empty
                bool hit = false;
                if (j < batch_count)
                    hit = may_touch_quad(s_rec[j * CUGS_REC_F4 + 0], s_rec[j * CUGS_REC_F4 + 1], ar.x0, ar.y0, ar.wx,
                                         ar.wy);
                unsigned long long mask = __ballot(hit);
                if (STATS) st_tested += min(CUGS_WAVE, batch_count - sub * CUGS_WAVE);
                while (mask != 0ull) {                                     // back to front: highest bit first
                    if (STATS) ++st_steps;
                    const int bit = 63 - __builtin_clzll(mask);
                    mask &= ~(1ull << bit);
                    const int rec = (sub * CUGS_WAVE + bit) * CUGS_REC_F4;
                    const float4 g0 = s_rec[rec], g1 = s_rec[rec + 1], col = s_rec[rec + 2];
                    const float o = g1.y;

                    // ---- decisions (backward.cu:123-145) as 0/1 floats
                    PixelEval e;
                    const float alpha = pixel_alpha_raw(pxf, pyf, g0.x, g0.y, g0.z, g0.w, g1.x, o, open, e);
                    const float passf = passes_alpha_min(alpha);           // alpha >= 1/255 (0 for a finished pixel)
                    rem -= passf;                                          // contributors counted from the END (Q1)
                    open = sat_add(rem, 1.0f);                             // 0 once a passer fell beyond n_contrib
                    const float k = passf * open;                          // 1: this Gaussian contributes here
                    const float al = alpha * k;
                    // clamp gate (backward.cu:181-191): o e >= 0.99 <=> alpha == 0.99f zeroes dL/do and dL/dpower,
                    // dL/drgb still flows
                    const float gk = k * below_alpha_cap(alpha);

                    // ---- values (v_rcp_f32 + FMAs); k == 0 makes both outputs exactly zero
                    const float rcp = __builtin_amdgcn_rcpf(1.0f - al);    // al <= 0.99; rcp(1) == 1 exactly
                    T *= rcp;                                              // T_before = T_after / (1 - alpha)
                    const float weight = al * T;
                    const float G = fmaf(dC2, col.w, fmaf(dC1, col.z, dC0 * col.x));
                    const float gate = fmaf(T, G, -(rcp * D)) * gk;        // dL/dalpha, gated
                    D = fmaf(weight, G, D);
                    if (STATS) {
                        const unsigned long long lm = __ballot(k != 0.0f);
                        if (lm) { ++st_contrib; st_lanes += __popcll(lm); st_open += __popcll(__ballot(open != 0.0f)); }
                    }
                    s_contrib[wave][pending][lane] = make_float2(weight, gate * e.e);
                    hitrecs = (hitrecs & ~(0xFFFFull << (16 * pending))) | ((unsigned long long)(unsigned)rec << (16 * pending));
                    if (++pending == CUGS_BWD_HITS) {
                        flush(CUGS_BWD_HITS);
                        pending = 0;
                    }
                    if (__ballot(open != 0.0f) == 0ull) { wave_done = true; break; }
                }
            }
            if (pending) {                 // before s_rec is re-staged (and at the end of the wave's walk)
                flush(pending);
                pending = 0;
            }
        }
        // the next iteration's first barrier orders this batch's LDS reads before the re-staging
    }
    if (STATS && lane == 0) {
        float* st = grad_accum + (int64_t)stats_row * CUGS_GRAD_STRIDE;
        atomicAdd(&st[0], (float)st_steps); atomicAdd(&st[1], (float)st_contrib); atomicAdd(&st[2], (float)st_lanes);
        atomicAdd(&st[3], (float)st_batches); atomicAdd(&st[4], (float)st_tested); atomicAdd(&st[5], (float)num_batches);
        atomicAdd(&st[7], (float)st_open);
    }
}

// grad_accum rows -> the four reference-layout tensors of RasterizeBackwardOutput (backward.hpp); applies the
// per-Gaussian map from the accumulated moments (see the header comment) with Sigma'^-1 = (a, b, c).
template <bool PACKED>
__global__ __launch_bounds__(CUGS_BLOCK) void k_unpack_grads(int64_t n, const float* __restrict__ acc,
                                                             const float* __restrict__ packed,
                                                             const float* __restrict__ cov_2d_inv,
                                                             float* __restrict__ dL_drgb,
                                                             float* __restrict__ dL_dopa,
                                                             float* __restrict__ dL_dmeans,
                                                             float* __restrict__ dL_dcov) {
    const int64_t i = (int64_t)blockIdx.x * CUGS_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float4* row = reinterpret_cast<const float4*>(acc + i * CUGS_GRAD_STRIDE);
    const float4 r0 = row[0], r1 = row[1];
    const float r2 = acc[i * CUGS_GRAD_STRIDE + 8];
    float a, b, c;
    if (PACKED) {
        const float4* rec = reinterpret_cast<const float4*>(packed + i * CUGS_PACKED_STRIDE);
        const float4 p0 = rec[0];
        a = p0.z; b = p0.w; c = rec[1].x;
    } else {
        a = cov_2d_inv[i * 3 + 0]; b = cov_2d_inv[i * 3 + 1]; c = cov_2d_inv[i * 3 + 2];
    }
    dL_drgb[i * 3 + 0] = r0.x; dL_drgb[i * 3 + 1] = r0.y; dL_drgb[i * 3 + 2] = r0.z;
    dL_dopa[i] = r0.w;
    const GradMoments m{r1.x, r1.y, r1.z, r1.w, r2};
    const Grad2D g = grads_from_moments(m, a, b, c);
    dL_dmeans[i * 2 + 0] = g.mx; dL_dmeans[i * 2 + 1] = g.my;
    dL_dcov[i * 3 + 0] = g.a; dL_dcov[i * 3 + 1] = g.b; dL_dcov[i * 3 + 2] = g.c;
}

}  // namespace

namespace {
int rasterize_backward_impl(int width, int height, const float background_host[3],
                                       const int32_t* tile_ranges, const int32_t* gaussian_indices,
                                       const float* means_2d, const float* cov_2d_inv, const float* rgb,
                                       const float* opacities_act, const float* packed,
                                       const float* dL_dcolor, const float* final_T,
                                       const int32_t* n_contrib, int64_t n, float* grad_accum,
                                       float* dL_drgb, float* dL_dopacity_act, float* dL_dmeans_2d,
                                       float* dL_dcov_2d_inv, bool prezeroed, const uint32_t* tile_order, void* stream) {
    if (width < 0 || height < 0 || n < 0 || !background_host) return CUGS_EINVAL;
    if (n == 0) return 0;
    if (!grad_accum) return CUGS_EINVAL;
    if ((reinterpret_cast<uintptr_t>(grad_accum) & 63u) != 0) return CUGS_EALIGN;
    const int n_soa = (dL_drgb != nullptr) + (dL_dopacity_act != nullptr) + (dL_dmeans_2d != nullptr) +
                      (dL_dcov_2d_inv != nullptr);
    if (n_soa != 0 && n_soa != 4) return CUGS_EINVAL;
    if (n_soa == 4 && !packed && !cov_2d_inv) return CUGS_EINVAL;
    if (packed && !cugs_aligned16(packed)) return CUGS_EALIGN;
    if (tile_order && !cugs_aligned16(tile_order)) return CUGS_EALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int64_t rows = n;
#ifdef CUGS_DEV
    const bool stats = cugs_dev_backward_stats();          // the caller allocated n + 1 rows (tools/ablate_backward.py)
    if (stats) rows = n + 1;
#endif
    if (!prezeroed || rows != n)
        CUGS_RETURN_IF_HIP(hipMemsetAsync(grad_accum, 0, sizeof(float) * CUGS_GRAD_STRIDE * (size_t)rows, st));

    const int ntx = (width + CUGS_TILE - 1) / CUGS_TILE, nty = (height + CUGS_TILE - 1) / CUGS_TILE;
    if (ntx > 0 && nty > 0 && gaussian_indices) {               // backward.cu:267-269; NULL indices = no pairs
        if (!tile_ranges || !dL_dcolor || !final_T || !n_contrib) return CUGS_EINVAL;
        if (!packed && (!means_2d || !cov_2d_inv || !rgb || !opacities_act)) return CUGS_EINVAL;
        if ((int64_t)width * height > 2147483647ll / 3) return CUGS_EOVERFLOW;
        RasterGeom geo{width, height, ntx, ntx * nty, background_host[0], background_host[1], background_host[2]};
        RasterSrc src{tile_ranges, gaussian_indices, packed, means_2d, cov_2d_inv, rgb, opacities_act, reinterpret_cast<const uint4*>(tile_order)};
        const bool wide = rows > (int64_t(1) << 26);            // 64-byte rows beyond a 32-bit byte offset
        int64_t stats_arg = n;                                  // STATS builds: the row that takes the counters
#ifdef CUGS_DEV
        if (const char* e = std::getenv("CUGS_BWD_NO_ATOMICS")) if (e[0] == '1') stats_arg = -2;
#endif
#define CUGS_LAUNCH_BWD(P, W, S)                                                                              \
    hipLaunchKernelGGL((k_raster_backward<P, W, S>), dim3(geo.ntiles), dim3(CUGS_BLOCK), 0, st, geo, src, \
                       dL_dcolor, final_T, n_contrib, grad_accum, stats_arg)
#ifdef CUGS_DEV
        if (stats) {
            if (packed) CUGS_LAUNCH_BWD(true, true, true); else CUGS_LAUNCH_BWD(false, true, true);
        } else
#endif
        if (packed) {
            if (wide) CUGS_LAUNCH_BWD(true, true, false); else CUGS_LAUNCH_BWD(true, false, false);
        } else {
            if (wide) CUGS_LAUNCH_BWD(false, true, false); else CUGS_LAUNCH_BWD(false, false, false);
        }
#undef CUGS_LAUNCH_BWD
        CUGS_LAUNCH_CHECK();
    }
    if (n_soa == 4) {
        const dim3 grid((unsigned)((n + CUGS_BLOCK - 1) / CUGS_BLOCK));
        if (packed)
            hipLaunchKernelGGL(k_unpack_grads<true>, grid, dim3(CUGS_BLOCK), 0, st, n, grad_accum, packed, cov_2d_inv,
                               dL_drgb, dL_dopacity_act, dL_dmeans_2d, dL_dcov_2d_inv);
        else
            hipLaunchKernelGGL(k_unpack_grads<false>, grid, dim3(CUGS_BLOCK), 0, st, n, grad_accum, packed, cov_2d_inv,
                               dL_drgb, dL_dopacity_act, dL_dmeans_2d, dL_dcov_2d_inv);
        CUGS_LAUNCH_CHECK();
    }
    return 0;
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
    return rasterize_backward_impl(width, height, background_host, tile_ranges, gaussian_indices, means_2d, cov_2d_inv, rgb,
                                   opacities_act, packed, dL_dcolor, final_T, n_contrib, n, grad_accum, dL_drgb,
                                   dL_dopacity_act, dL_dmeans_2d, dL_dcov_2d_inv, false, nullptr, stream);
}

extern "C" int cugs_rasterize_backward_prezeroed(int width, int height, const float background_host[3],
                                                 const int32_t* tile_ranges, const int32_t* gaussian_indices,
                                                 const float* means_2d, const float* cov_2d_inv, const float* rgb,
                                                 const float* opacities_act, const float* packed,
                                                 const float* dL_dcolor, const float* final_T,
                                                 const int32_t* n_contrib, int64_t n, float* grad_accum,
                                                 float* dL_drgb, float* dL_dopacity_act, float* dL_dmeans_2d,
                                                 float* dL_dcov_2d_inv, void* stream) {
    return rasterize_backward_impl(width, height, background_host, tile_ranges, gaussian_indices, means_2d, cov_2d_inv, rgb,
                                   opacities_act, packed, dL_dcolor, final_T, n_contrib, n, grad_accum, dL_drgb,
                                   dL_dopacity_act, dL_dmeans_2d, dL_dcov_2d_inv, true, nullptr, stream);
}

extern "C" int cugs_rasterize_backward_ordered(int width, int height, const float background_host[3],
                                               const int32_t* tile_ranges, const int32_t* gaussian_indices,
                                               const float* means_2d, const float* cov_2d_inv, const float* rgb,
                                               const float* opacities_act, const float* packed,
                                               const float* dL_dcolor, const float* final_T,
                                               const int32_t* n_contrib, int64_t n, float* grad_accum,
                                               float* dL_drgb, float* dL_dopacity_act, float* dL_dmeans_2d,
                                               float* dL_dcov_2d_inv, int prezeroed, const uint32_t* tile_order,
                                               void* stream) {
    return rasterize_backward_impl(width, height, background_host, tile_ranges, gaussian_indices, means_2d, cov_2d_inv, rgb,
                                   opacities_act, packed, dL_dcolor, final_T, n_contrib, n, grad_accum, dL_drgb,
                                   dL_dopacity_act, dL_dmeans_2d, dL_dcov_2d_inv, prezeroed != 0, tile_order, stream);
}

#ifdef CUGS_DEV
// ---- development hooks (libcugs_hip_dev.so only; not part of include/cugs_hip.h) ------------------
// cugsdbg_reduce9: one wave runs reduce9t on caller data.  in: [9][64] floats (value k of lane l at k*64+l,
// k = slot); out: [64] floats (each lane's result), slots: [64] ints.
// cugsdbg_rcp: out[i] = v_rcp_f32(in[i]).
namespace {
__global__ void k_dbg_reduce9(const float* __restrict__ in, float* __restrict__ out, int* __restrict__ slots) {
    const int l = threadIdx.x;
    const bool side = (l & 8) != 0;
    const float v0 = in[0 * 64 + l], v1 = in[1 * 64 + l];
    out[l] = reduce9t(side ? v1 : v0, side ? v0 : v1, in[2 * 64 + l], in[3 * 64 + l], in[4 * 64 + l], in[5 * 64 + l],
                      in[6 * 64 + l], in[8 * 64 + l], in[7 * 64 + l]);
    slots[l] = reduce9t_slot(l);
}
// four independent rows: in [9][64], out [64] (each lane's row total), slots [64]
__global__ void k_dbg_reduce9r16(const float* __restrict__ in, float* __restrict__ out, int* __restrict__ slots) {
    const int l = threadIdx.x;
    const bool side = (l & 8) != 0;
    const float v0 = in[0 * 64 + l], v1 = in[1 * 64 + l];
    out[l] = reduce9r16(side ? v1 : v0, side ? v0 : v1, in[2 * 64 + l], in[3 * 64 + l], in[4 * 64 + l], in[5 * 64 + l],
                        in[6 * 64 + l], in[8 * 64 + l], in[7 * 64 + l], l);
    slots[l] = reduce9r16_slot(l);
}
__global__ void k_dbg_rcp(const float* __restrict__ in, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_rcpf(in[i]);
}
__global__ void k_dbg_blend_exp(const float* __restrict__ q, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = cugs_blend_exp_q(q[i]);
}
bool g_dev_backward_stats = false;
}  // namespace
bool cugs_dev_backward_stats() { return g_dev_backward_stats; }
extern "C" int cugsdbg_backward_stats(int on) { g_dev_backward_stats = (on != 0); return 0; }
extern "C" int cugsdbg_reduce9(const float* in, float* out, int* slots, void* stream) {
    hipLaunchKernelGGL(k_dbg_reduce9, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), in, out, slots);
    CUGS_LAUNCH_CHECK();
    return 0;
}
extern "C" int cugsdbg_reduce9r16(const float* in, float* out, int* slots, void* stream) {
    hipLaunchKernelGGL(k_dbg_reduce9r16, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), in, out, slots);
    CUGS_LAUNCH_CHECK();
    return 0;
}
extern "C" int cugsdbg_blend_exp_q(const float* q, float* out, int n, void* stream) {
    hipLaunchKernelGGL(k_dbg_blend_exp, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), q, out, n);
    CUGS_LAUNCH_CHECK();
    return 0;
}
extern "C" int cugsdbg_rcp(const float* in, float* out, int n, void* stream) {
    hipLaunchKernelGGL(k_dbg_rcp, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), in, out, n);
    CUGS_LAUNCH_CHECK();
    return 0;
}
#endif
