// loss.hip — combined_loss = (1-lambda) L1 + lambda (1 - mean SSIM) and its analytic dL/dcolor, fused
// (the step between render() and render_backward() in every training iteration).
//
// Replaces, per iteration: l1_loss + ssim + ssim_loss + combined_loss (training/loss.cpp:88-140: five
// grouped 11x11 conv2d's, ~15 elementwise kernels), the libtorch autograd pass that produces dL_dcolor
// and the two clones around it (training/trainer.cpp:214-217).  Same definition: sigma-1.5 Gaussian window,
// zero padding, C1 = 0.01^2, C2 = 0.03^2, SSIM averaged over pixels and channels.
//
// Two launches over 16x16 pixel tiles (halo = window/2 in LDS):
//   k_ssim_stats: x, y tiles -> separable window sums of x, y, x^2, y^2, xy -> SSIM per pixel/channel,
//                 its partial derivatives w.r.t. (mu_x, E[x^2], E[xy]) as three maps, and the two loss sums
//                 (one fp64 partial pair per tile, reduced in a fixed order: the scalar is deterministic);
//   k_ssim_grad:  separable window sums of the three maps (the window is symmetric, so the adjoint of the
//                 convolution is the convolution) -> dL/dx = (1-lambda) sign(x-y)/n - lambda/n (G_m + 2x G_p + y G_r).
// With S = A1 A2 / (B1 B2), A1 = 2 m n + C1, A2 = 2 (r - m n) + C2, B1 = m^2 + n^2 + C1, B2 = (p - m^2) + (q - n^2) + C2
// (m = mu_x, n = mu_y, p = E[x^2], q = E[y^2], r = E[xy]):
//   dS/dm = 2 n (A2 - A1) / (B1 B2) - 2 m S (1/B1 - 1/B2),  dS/dp = -S / B2,  dS/dr = 2 A1 / (B1 B2).
// HBM-bound: reads 2 images, writes/reads 3 maps, writes 1 image (11 x 12 B per pixel).
#include "cugs_common.h"

namespace {

constexpr int LT = 16;            // tile edge
constexpr int MAX_R = 7;          // window sizes 3..15
constexpr int MAX_E = LT + 2 * MAX_R;

struct Window { float w[2 * MAX_R + 1]; int r; };

// Tiles keep the image's interleaved [.,.,3] layout in LDS: a tile row of E pixels is E*3 consecutive floats,
// loaded once for all three channels; the horizontal pass treats it as 48 output "columns" with a tap
// stride of 3 floats.
// With a compile-time radius every thread first issues all its loads (NIMG images x <= 8 floats) and only then
// writes LDS, so one HBM latency is paid per tile instead of one per element.
// PAIR: images 0 and 1 are stored interleaved, element e at s_t[0][2e], s_t[0][2e+1] (one ds_read_b64 feeds a
// packed-fp32 FMA with both); a third image, if any, goes to s_t[2] as usual.
template <int RT, int NIMG, bool PAIR = false>
__device__ __forceinline__ void load_tiles3(const float* const (&img)[NIMG], float* const (&s_t)[NIMG], int tx0, int ty0,
                                            int R, int E, int w, int h) {
    const int row_f = E * 3, total = E * row_f, w3 = w * 3;
    if constexpr (RT > 0) {
        constexpr int ET = LT + 2 * RT, PER = (ET * ET * 3 + CUGS_BLOCK - 1) / CUGS_BLOCK;
        float v[NIMG][PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int e = (int)threadIdx.x + i * CUGS_BLOCK;
            const int ey = e / row_f, ef = e - ey * row_f;
            const int gy = ty0 + ey - R, gxf = (tx0 - R) * 3 + ef;      // float index within the image row
            const bool ok = e < total && gy >= 0 && gy < h && gxf >= 0 && gxf < w3;
            const int64_t off = ok ? (int64_t)gy * w3 + gxf : 0;
#pragma unroll
            for (int m = 0; m < NIMG; ++m) { const float t = img[m][off]; v[m][i] = ok ? t : 0.0f; }   // zero padding
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int e = (int)threadIdx.x + i * CUGS_BLOCK;
            if (e < total) {
                if constexpr (PAIR) {
                    reinterpret_cast<float2*>(s_t[0])[e] = make_float2(v[0][i], v[1][i]);
#pragma unroll
                    for (int m = 2; m < NIMG; ++m) s_t[m][e] = v[m][i];
                } else {
#pragma unroll
                    for (int m = 0; m < NIMG; ++m) s_t[m][e] = v[m][i];
                }
            }
        }
    } else {
        for (int e = threadIdx.x; e < total; e += CUGS_BLOCK) {
            const int ey = e / row_f, ef = e - ey * row_f;
            const int gy = ty0 + ey - R, gxf = (tx0 - R) * 3 + ef;
            const bool ok = gy >= 0 && gy < h && gxf >= 0 && gxf < w3;
#pragma unroll
            for (int m = 0; m < NIMG; ++m) s_t[m][e] = ok ? img[m][(int64_t)gy * w3 + gxf] : 0.0f;
        }
    }
}

template <int RT>
__global__ __launch_bounds__(CUGS_BLOCK) void k_ssim_stats(int w, int h, Window win, const float* __restrict__ xr,
                                                           const float* __restrict__ yt, float* __restrict__ d_m,
                                                           float* __restrict__ d_p, float* __restrict__ d_r,
                                                           float* __restrict__ ssim_map, double* __restrict__ sums) {
    constexpr int CW = LT * 3;                                           // 48 float columns per tile row
    // With a compile-time radius the horizontal sums pass through registers and overwrite the input tiles,
    // so a block holds max(inputs, sums) instead of both (25 KB for the 11-tap window: 6 blocks per CU).
    constexpr bool ALIAS = RT > 0;
    constexpr int ET = ALIAS ? LT + 2 * RT : MAX_E;
    constexpr int IN_F = ET * ET * 3, H_F = ET * CW;
    constexpr int POOL = ALIAS ? (2 * IN_F > 5 * H_F ? 2 * IN_F : 5 * H_F) : 2 * IN_F + 5 * H_F;
    __shared__ float s_pool[POOL];
    __shared__ double s_red[2][4];
    float* const s_x = s_pool;
    float* const s_y = s_pool + IN_F;
    float* const s_hp = ALIAS ? s_pool : s_pool + 2 * IN_F;             // plane k at s_hp + k * H_F
    const int R = RT > 0 ? RT : win.r, E = LT + 2 * R;      // RT > 0: compile-time radius, loops fully unrolled
    const int tx0 = blockIdx.x * LT, ty0 = blockIdx.y * LT;
    const int tid = threadIdx.x, lx = tid & 15, ly = tid >> 4;
    const int px = tx0 + lx, py = ty0 + ly;
    const bool inside = px < w && py < h;
    typedef float v2f __attribute__((ext_vector_type(2)));
    {
        const float* const imgs[2] = {xr, yt};
        float* const dst[2] = {s_x, s_y};
        load_tiles3<RT, 2, ALIAS>(imgs, dst, tx0, ty0, R, E, w, h);     // compile-time radius: (x, y) interleaved
    }
    __syncthreads();
    // |x - y| at this thread's pixel, read before the tiles can be overwritten
    float l1c[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const int c = ((ly + R) * E + lx + R) * 3 + ch;
        if constexpr (ALIAS) {
            const float2 xy = reinterpret_cast<const float2*>(s_pool)[c];
            l1c[ch] = fabsf(xy.x - xy.y);
        } else {
            l1c[ch] = fabsf(s_x[c] - s_y[c]);
        }
    }
    // horizontal pass: E rows x 48 float columns (16 pixels x 3 channels), tap stride 3
    auto hsum = [&](int e, float (&o)[5]) {
        const int ey = e / CW, cf = e - ey * CW;
        const float* rx = s_x + ey * E * 3 + cf;
        const float* ry = s_y + ey * E * 3 + cf;
        float a = 0.0f, b = 0.0f, aa = 0.0f, bb = 0.0f, ab = 0.0f;
#pragma unroll
        for (int k = 0; k <= 2 * R; ++k) {
            const float wk = win.w[k], xv = rx[3 * k], yv = ry[3 * k];
            a = fmaf(wk, xv, a); b = fmaf(wk, yv, b);
            aa = fmaf(wk, xv * xv, aa); bb = fmaf(wk, yv * yv, bb); ab = fmaf(wk, xv * yv, ab);
        }
        o[0] = a; o[1] = b; o[2] = aa; o[3] = bb; o[4] = ab;
    };
    if constexpr (ALIAS) {
        // Packed fp32: (x, y) and (x^2, y^2) each advance with ONE v_pk_fma_f32 per tap - 5 VALU per tap instead
        // of 8; the sums are stored as the pairs (mu_x, mu_y), (E[x^2], E[y^2]) and the single E[xy].
        constexpr int PER_H = (H_F + CUGS_BLOCK - 1) / CUGS_BLOCK;
        const v2f* s_xy = reinterpret_cast<const v2f*>(s_pool);
        v2f acc_ab[PER_H], acc_sq[PER_H];
        float acc_xy[PER_H];
#pragma unroll
        for (int i = 0; i < PER_H; ++i) {
            const int e = tid + i * CUGS_BLOCK;
            acc_ab[i] = (v2f){0.0f, 0.0f}; acc_sq[i] = (v2f){0.0f, 0.0f}; acc_xy[i] = 0.0f;
            if (e < H_F) {
                const int ey = e / CW, cf = e - ey * CW;
                const v2f* row = s_xy + ey * E * 3 + cf;
#pragma unroll
                for (int k = 0; k <= 2 * R; ++k) {
                    const float wk = win.w[k];
                    const v2f xy = row[3 * k], wk2 = {wk, wk};
                    acc_ab[i] = __builtin_elementwise_fma(wk2, xy, acc_ab[i]);
                    acc_sq[i] = __builtin_elementwise_fma(wk2, xy * xy, acc_sq[i]);
                    acc_xy[i] = fmaf(wk, xy.x * xy.y, acc_xy[i]);
                }
            }
        }
        __syncthreads();                                                 // every read of the input tiles is done
        v2f* h_ab = reinterpret_cast<v2f*>(s_pool);
        v2f* h_sq = reinterpret_cast<v2f*>(s_pool + 2 * H_F);
        float* h_xy = s_pool + 4 * H_F;
#pragma unroll
        for (int i = 0; i < PER_H; ++i) {
            const int e = tid + i * CUGS_BLOCK;
            if (e < H_F) { h_ab[e] = acc_ab[i]; h_sq[e] = acc_sq[i]; h_xy[e] = acc_xy[i]; }
        }
    } else {
        for (int e = tid; e < E * CW; e += CUGS_BLOCK) {
            float o[5];
            hsum(e, o);
#pragma unroll
            for (int k = 0; k < 5; ++k) s_hp[k * H_F + e] = o[k];
        }
    }
    __syncthreads();
    float ssim_sum = 0.0f;
    double l1_acc = 0.0, ss_acc = 0.0;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float m = 0.0f, n = 0.0f, p = 0.0f, q = 0.0f, r = 0.0f;
        if constexpr (ALIAS) {
            const v2f* h_ab = reinterpret_cast<const v2f*>(s_pool);
            const v2f* h_sq = reinterpret_cast<const v2f*>(s_pool + 2 * H_F);
            const float* h_xy = s_pool + 4 * H_F;
            v2f mn = {0.0f, 0.0f}, pq = {0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k <= 2 * R; ++k) {
                const float wk = win.w[k];
                const v2f wk2 = {wk, wk};
                const int e = (ly + k) * CW + lx * 3 + ch;
                mn = __builtin_elementwise_fma(wk2, h_ab[e], mn);
                pq = __builtin_elementwise_fma(wk2, h_sq[e], pq);
                r = fmaf(wk, h_xy[e], r);
            }
            m = mn.x; n = mn.y; p = pq.x; q = pq.y;
        } else {
#pragma unroll
            for (int k = 0; k <= 2 * R; ++k) {
                const float wk = win.w[k];
                const int e = (ly + k) * CW + lx * 3 + ch;
                m = fmaf(wk, s_hp[e], m); n = fmaf(wk, s_hp[H_F + e], n); p = fmaf(wk, s_hp[2 * H_F + e], p);
                q = fmaf(wk, s_hp[3 * H_F + e], q); r = fmaf(wk, s_hp[4 * H_F + e], r);
            }
        }
        if (inside) {
            const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
            const float mn = m * n, mm = m * m, nn = n * n;
            const float A1 = 2.0f * mn + C1, A2 = 2.0f * (r - mn) + C2;
            const float B1 = mm + nn + C1, B2 = (p - mm) + (q - nn) + C2;
            const float inv = 1.0f / (B1 * B2);
            const float S = A1 * A2 * inv;
            const int64_t o = ((int64_t)py * w + px) * 3 + ch;
            d_m[o] = 2.0f * n * (A2 - A1) * inv - 2.0f * m * S * (1.0f / B1 - 1.0f / B2);
            d_p[o] = -S / B2;
            d_r[o] = 2.0f * A1 * inv;
            ssim_sum += S;
            ss_acc += (double)S;
            l1_acc += (double)l1c[ch];
        }
    }
    if (inside && ssim_map) ssim_map[(int64_t)py * w + px] = ssim_sum / 3.0f;   // mean(dim=2), loss.cpp:128

    for (int d = 32; d >= 1; d >>= 1) { l1_acc += __shfl_xor(l1_acc, d); ss_acc += __shfl_xor(ss_acc, d); }
    if ((tid & 63) == 0) { s_red[0][tid >> 6] = l1_acc; s_red[1][tid >> 6] = ss_acc; }
    __syncthreads();
    if (tid == 0) {
        // one partial pair per tile, summed (in fp64, in tile order: deterministic) by k_loss_finalize; thousands
        // of fp64 atomics on one cache line serialise at the memory side
        const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        sums[2 * blk + 0] = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
        sums[2 * blk + 1] = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
    }
}

// The per-tile partial sums -> loss_out[0] = combined loss, [1] = L1 mean, [2] = mean SSIM, [3] = 1 - mean SSIM
// (ssim_loss); one workgroup, fp64, fixed order (thread t takes tiles t, t + 256, ...; then a tree): deterministic.
// s_acc: 2 x CUGS_BLOCK doubles of LDS.  Contains barriers: call with the whole workgroup.
__device__ __forceinline__ void finalize_loss(const double* __restrict__ partials, int nblk, double count, float lambda,
                                              float* __restrict__ out, double (*s_acc)[CUGS_BLOCK]) {
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nblk; i += CUGS_BLOCK) { a += partials[2 * i]; b += partials[2 * i + 1]; }
    s_acc[0][threadIdx.x] = a; s_acc[1][threadIdx.x] = b;
    __syncthreads();
    for (int d = CUGS_BLOCK / 2; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) {
            s_acc[0][threadIdx.x] += s_acc[0][threadIdx.x + d];
            s_acc[1][threadIdx.x] += s_acc[1][threadIdx.x + d];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double l1 = s_acc[0][0] / count, ss = s_acc[1][0] / count;
        out[0] = (float)((1.0 - (double)lambda) * l1 + (double)lambda * (1.0 - ss));
        out[1] = (float)l1;
        out[2] = (float)ss;
        out[3] = (float)(1.0 - ss);
    }
    __syncthreads();
}

// `partials` != nullptr: workgroup (0, 0) first reduces the loss sums of k_ssim_stats (finalize_loss, in the LDS the
// tiles go to afterwards) - it starts first and is done long before the grid is, and the separate one-workgroup
// launch (~10 us of kernel boundary and latency for 130 KB) disappears from the iteration.
template <int RT>
__global__ __launch_bounds__(CUGS_BLOCK) void k_ssim_grad(int w, int h, Window win, float lambda,
                                                          const float* __restrict__ xr, const float* __restrict__ yt,
                                                          const float* __restrict__ d_m, const float* __restrict__ d_p,
                                                          const float* __restrict__ d_r, float* __restrict__ dL_dx,
                                                          const double* __restrict__ partials, int nblk, double count,
                                                          float* __restrict__ loss_out) {
    constexpr int CW = LT * 3;
    constexpr bool ALIAS = RT > 0;                                       // as in k_ssim_stats
    constexpr int ET = ALIAS ? LT + 2 * RT : MAX_E;
    constexpr int IN_F = ET * ET * 3, H_F = ET * CW;
    __shared__ float s_pool[ALIAS ? 3 * IN_F : 3 * IN_F + 3 * H_F];
    float* const s_a[3] = {s_pool, s_pool + IN_F, s_pool + 2 * IN_F};
    float* const s_hp = ALIAS ? s_pool : s_pool + 3 * IN_F;
    static_assert(sizeof(s_pool) >= 2 * CUGS_BLOCK * sizeof(double), "finalize_loss borrows the tile pool");
    if (partials && blockIdx.x == 0 && blockIdx.y == 0)
        finalize_loss(partials, nblk, count, lambda, loss_out, reinterpret_cast<double (*)[CUGS_BLOCK]>(s_pool));
    const int R = RT > 0 ? RT : win.r, E = LT + 2 * R;
    const int tx0 = blockIdx.x * LT, ty0 = blockIdx.y * LT;
    const int tid = threadIdx.x, lx = tid & 15, ly = tid >> 4;
    const int px = tx0 + lx, py = ty0 + ly;
    const bool inside = px < w && py < h;
    const float inv_n = 1.0f / ((float)w * (float)h * 3.0f);
    float xv[3] = {0.0f, 0.0f, 0.0f}, yv[3] = {0.0f, 0.0f, 0.0f};      // issued with the tile loads, used at the end
    if (inside) {
        const int64_t o = ((int64_t)py * w + px) * 3;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) { xv[ch] = xr[o + ch]; yv[ch] = yt[o + ch]; }
    }
    typedef float v2f __attribute__((ext_vector_type(2)));
    {
        const float* const imgs[3] = {d_m, d_p, d_r};
        float* const dst[3] = {s_a[0], s_a[1], s_a[2]};
        load_tiles3<RT, 3, ALIAS>(imgs, dst, tx0, ty0, R, E, w, h);     // compile-time radius: (d_m, d_p) interleaved
    }
    __syncthreads();
    auto hsum = [&](int e, float (&o)[3]) {
        const int ey = e / CW, cf = e - ey * CW;
        const int base = ey * E * 3 + cf;
        float a = 0.0f, b = 0.0f, c = 0.0f;
#pragma unroll
        for (int k = 0; k <= 2 * R; ++k) {
            const float wk = win.w[k];
            a = fmaf(wk, s_a[0][base + 3 * k], a);
            b = fmaf(wk, s_a[1][base + 3 * k], b);
            c = fmaf(wk, s_a[2][base + 3 * k], c);
        }
        o[0] = a; o[1] = b; o[2] = c;
    };
    if constexpr (ALIAS) {
        constexpr int PER_H = (H_F + CUGS_BLOCK - 1) / CUGS_BLOCK;
        const v2f* s_mp = reinterpret_cast<const v2f*>(s_pool);          // (d_m, d_p) pairs
        v2f acc_mp[PER_H];
        float acc_r[PER_H];
#pragma unroll
        for (int i = 0; i < PER_H; ++i) {
            const int e = tid + i * CUGS_BLOCK;
            acc_mp[i] = (v2f){0.0f, 0.0f}; acc_r[i] = 0.0f;
            if (e < H_F) {
                const int ey = e / CW, cf = e - ey * CW;
                const int base = ey * E * 3 + cf;
#pragma unroll
                for (int k = 0; k <= 2 * R; ++k) {
                    const float wk = win.w[k];
                    acc_mp[i] = __builtin_elementwise_fma((v2f){wk, wk}, s_mp[base + 3 * k], acc_mp[i]);
                    acc_r[i] = fmaf(wk, s_a[2][base + 3 * k], acc_r[i]);
                }
            }
        }
        __syncthreads();
        v2f* h_mp = reinterpret_cast<v2f*>(s_pool);
        float* h_r = s_pool + 2 * H_F;
#pragma unroll
        for (int i = 0; i < PER_H; ++i) {
            const int e = tid + i * CUGS_BLOCK;
            if (e < H_F) { h_mp[e] = acc_mp[i]; h_r[e] = acc_r[i]; }
        }
    } else {
        for (int e = tid; e < E * CW; e += CUGS_BLOCK) {
            float o[3];
            hsum(e, o);
#pragma unroll
            for (int k = 0; k < 3; ++k) s_hp[k * H_F + e] = o[k];
        }
    }
    __syncthreads();
    if (!inside) return;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float gm = 0.0f, gp = 0.0f, gr = 0.0f;
        if constexpr (ALIAS) {
            const v2f* h_mp = reinterpret_cast<const v2f*>(s_pool);
            const float* h_r = s_pool + 2 * H_F;
            v2f mp = {0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k <= 2 * R; ++k) {
                const float wk = win.w[k];
                const int e = (ly + k) * CW + lx * 3 + ch;
                mp = __builtin_elementwise_fma((v2f){wk, wk}, h_mp[e], mp);
                gr = fmaf(wk, h_r[e], gr);
            }
            gm = mp.x; gp = mp.y;
        } else {
#pragma unroll
            for (int k = 0; k <= 2 * R; ++k) {
                const float wk = win.w[k];
                const int e = (ly + k) * CW + lx * 3 + ch;
                gm = fmaf(wk, s_hp[e], gm); gp = fmaf(wk, s_hp[H_F + e], gp); gr = fmaf(wk, s_hp[2 * H_F + e], gr);
            }
        }
        const int64_t o = ((int64_t)py * w + px) * 3 + ch;
        const float x = xv[ch], y = yv[ch], d = x - y;
        const float sgn = (d > 0.0f) ? 1.0f : ((d < 0.0f) ? -1.0f : 0.0f);          // d|x|/dx, 0 at 0 like libtorch
        dL_dx[o] = (1.0f - lambda) * sgn * inv_n - lambda * inv_n * (gm + 2.0f * x * gp + y * gr);
    }
}

// The same reduction as its own launch, for callers that want the loss without the gradient.
__global__ __launch_bounds__(CUGS_BLOCK) void k_loss_finalize(const double* __restrict__ partials, int nblk, double count,
                                                               float lambda, float* __restrict__ out) {
    __shared__ double s_acc[2][CUGS_BLOCK];
    finalize_loss(partials, nblk, count, lambda, out, s_acc);
}

}  // namespace

extern "C" size_t cugs_loss_workspace_bytes(int width, int height) {
    if (width < 0 || height < 0) return 0;
    const size_t tiles = (size_t)((width + LT - 1) / LT) * (size_t)((height + LT - 1) / LT);
    return 256 + (16 * tiles + 255) / 256 * 256 + sizeof(float) * 3 * 3 * (size_t)width * (size_t)height;
}

extern "C" int cugs_combined_loss(int width, int height, const float* rendered, const float* target, float lambda,
                                  int window_size, void* workspace, size_t workspace_bytes, float* loss_out,
                                  float* ssim_map, float* dL_dcolor, void* stream) {
    if (width <= 0 || height <= 0 || !rendered || !target || !loss_out || !workspace) return CUGS_EINVAL;
    if (window_size % 2 != 1 || window_size < 3 || window_size > 2 * MAX_R + 1) return CUGS_EINVAL;   // loss.cpp:96-97
    if (workspace_bytes < cugs_loss_workspace_bytes(width, height)) return CUGS_EWORKSPACE;
    if ((int64_t)width * height > 2147483647ll / 3) return CUGS_EOVERFLOW;
    hipStream_t st = static_cast<hipStream_t>(stream);

    // get_gaussian_kernel (loss.cpp:47-83): k1 = exp(-x^2 / (2 sigma^2)) / sum; k2 = k1 (x) k1 / sum(k1 (x) k1).
    // k2 is rank one, so the separable factor is u = k1 / sqrt(sum(k1 (x) k1)).
    Window win;
    win.r = window_size / 2;
    float k1[2 * MAX_R + 1];
    float s1 = 0.0f;
    for (int i = 0; i < window_size; ++i) {
        const float x = (float)(i - win.r);
        k1[i] = expf(-x * x / (2.0f * 1.5f * 1.5f));
        s1 += k1[i];
    }
    double s2 = 0.0;
    for (int i = 0; i < window_size; ++i) k1[i] /= s1;
    for (int i = 0; i < window_size; ++i)
        for (int j = 0; j < window_size; ++j) s2 += (double)(k1[i] * k1[j]);
    for (int i = 0; i < 2 * MAX_R + 1; ++i) win.w[i] = i < window_size ? (float)((double)k1[i] / sqrt(s2)) : 0.0f;

    dim3 grid((width + LT - 1) / LT, (height + LT - 1) / LT), block(CUGS_BLOCK);
    const size_t tiles = (size_t)grid.x * grid.y;
    double* sums = reinterpret_cast<double*>(static_cast<char*>(workspace) + 256);     // [tiles][2] partial sums
    const size_t plane = 3 * (size_t)width * (size_t)height;
    float* d_m = reinterpret_cast<float*>(static_cast<char*>(workspace) + 256 + (16 * tiles + 255) / 256 * 256);
    float* d_p = d_m + plane;
    float* d_r = d_p + plane;
    if (win.r == 5)      // the reference's default window (11): compile-time radius
        hipLaunchKernelGGL(k_ssim_stats<5>, grid, block, 0, st, width, height, win, rendered, target, d_m, d_p, d_r, ssim_map, sums);
    else
        hipLaunchKernelGGL(k_ssim_stats<0>, grid, block, 0, st, width, height, win, rendered, target, d_m, d_p, d_r, ssim_map, sums);
    CUGS_LAUNCH_CHECK();
    const double count = (double)width * height * 3.0;
    if (!dL_dcolor) {
        hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(CUGS_BLOCK), 0, st, sums, (int)tiles, count, lambda, loss_out);
        CUGS_LAUNCH_CHECK();
        return 0;
    }
    if (win.r == 5)
        hipLaunchKernelGGL(k_ssim_grad<5>, grid, block, 0, st, width, height, win, lambda, rendered, target, d_m, d_p, d_r,
                           dL_dcolor, sums, (int)tiles, count, loss_out);
    else
        hipLaunchKernelGGL(k_ssim_grad<0>, grid, block, 0, st, width, height, win, lambda, rendered, target, d_m, d_p, d_r,
                           dL_dcolor, sums, (int)tiles, count, loss_out);
    CUGS_LAUNCH_CHECK();
    return 0;
}
