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
//                 (fp64 atomics: the scalar does not depend on block order beyond fp64 rounding);
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

__device__ __forceinline__ float fetch(const float* __restrict__ img, int x, int y, int ch, int w, int h) {
    return (x >= 0 && x < w && y >= 0 && y < h) ? img[((int64_t)y * w + x) * 3 + ch] : 0.0f;   // zero padding
}

template <int RT>
__global__ __launch_bounds__(CUGS_BLOCK) void k_ssim_stats(int w, int h, Window win, const float* __restrict__ xr,
                                                           const float* __restrict__ yt, float* __restrict__ d_m,
                                                           float* __restrict__ d_p, float* __restrict__ d_r,
                                                           float* __restrict__ ssim_map, double* __restrict__ sums) {
    __shared__ float s_x[MAX_E * MAX_E], s_y[MAX_E * MAX_E];
    __shared__ float s_h[5][MAX_E * LT];
    __shared__ double s_red[2][4];
    const int R = RT > 0 ? RT : win.r, E = LT + 2 * R;      // RT > 0: compile-time radius, loops fully unrolled
    const int tx0 = blockIdx.x * LT, ty0 = blockIdx.y * LT;
    const int tid = threadIdx.x, lx = tid & 15, ly = tid >> 4;
    const int px = tx0 + lx, py = ty0 + ly;
    const bool inside = px < w && py < h;
    float ssim_sum = 0.0f;          // over channels, for the map
    double l1_acc = 0.0, ss_acc = 0.0;

    for (int ch = 0; ch < 3; ++ch) {
        for (int e = tid; e < E * E; e += CUGS_BLOCK) {
            const int ey = e / E, ex = e - ey * E;
            s_x[e] = fetch(xr, tx0 + ex - R, ty0 + ey - R, ch, w, h);
            s_y[e] = fetch(yt, tx0 + ex - R, ty0 + ey - R, ch, w, h);
        }
        __syncthreads();
        // horizontal pass: E rows x 16 columns
        for (int e = tid; e < E * LT; e += CUGS_BLOCK) {
            const int ey = e / LT, cx = e - ey * LT;
            float a = 0.0f, b = 0.0f, aa = 0.0f, bb = 0.0f, ab = 0.0f;
#pragma unroll
            for (int k = 0; k <= 2 * R; ++k) {
                const float wk = win.w[k], xv = s_x[ey * E + cx + k], yv = s_y[ey * E + cx + k];
                a = fmaf(wk, xv, a); b = fmaf(wk, yv, b);
                aa = fmaf(wk, xv * xv, aa); bb = fmaf(wk, yv * yv, bb); ab = fmaf(wk, xv * yv, ab);
            }
            s_h[0][e] = a; s_h[1][e] = b; s_h[2][e] = aa; s_h[3][e] = bb; s_h[4][e] = ab;
        }
        __syncthreads();
        // vertical pass at this thread's pixel
        float m = 0.0f, n = 0.0f, p = 0.0f, q = 0.0f, r = 0.0f;
#pragma unroll
        for (int k = 0; k <= 2 * R; ++k) {
            const float wk = win.w[k];
            const int e = (ly + k) * LT + lx;
            m = fmaf(wk, s_h[0][e], m); n = fmaf(wk, s_h[1][e], n); p = fmaf(wk, s_h[2][e], p);
            q = fmaf(wk, s_h[3][e], q); r = fmaf(wk, s_h[4][e], r);
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
            l1_acc += (double)fabsf(s_x[(ly + R) * E + lx + R] - s_y[(ly + R) * E + lx + R]);
        }
        __syncthreads();            // tiles are reloaded for the next channel
    }
    if (inside && ssim_map) ssim_map[(int64_t)py * w + px] = ssim_sum / 3.0f;   // mean(dim=2), loss.cpp:128

    for (int d = 32; d >= 1; d >>= 1) { l1_acc += __shfl_xor(l1_acc, d); ss_acc += __shfl_xor(ss_acc, d); }
    if ((tid & 63) == 0) { s_red[0][tid >> 6] = l1_acc; s_red[1][tid >> 6] = ss_acc; }
    __syncthreads();
    if (tid == 0) {
        atomicAdd(&sums[0], s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3]);
        atomicAdd(&sums[1], s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3]);
    }
}

template <int RT>
__global__ __launch_bounds__(CUGS_BLOCK) void k_ssim_grad(int w, int h, Window win, float lambda,
                                                          const float* __restrict__ xr, const float* __restrict__ yt,
                                                          const float* __restrict__ d_m, const float* __restrict__ d_p,
                                                          const float* __restrict__ d_r, float* __restrict__ dL_dx) {
    __shared__ float s_a[3][MAX_E * MAX_E];
    __shared__ float s_h[3][MAX_E * LT];
    const int R = RT > 0 ? RT : win.r, E = LT + 2 * R;      // RT > 0: compile-time radius, loops fully unrolled
    const int tx0 = blockIdx.x * LT, ty0 = blockIdx.y * LT;
    const int tid = threadIdx.x, lx = tid & 15, ly = tid >> 4;
    const int px = tx0 + lx, py = ty0 + ly;
    const bool inside = px < w && py < h;
    const float inv_n = 1.0f / ((float)w * (float)h * 3.0f);

    for (int ch = 0; ch < 3; ++ch) {
        for (int e = tid; e < E * E; e += CUGS_BLOCK) {
            const int ey = e / E, ex = e - ey * E;
            const int gx = tx0 + ex - R, gy = ty0 + ey - R;
            s_a[0][e] = fetch(d_m, gx, gy, ch, w, h);
            s_a[1][e] = fetch(d_p, gx, gy, ch, w, h);
            s_a[2][e] = fetch(d_r, gx, gy, ch, w, h);
        }
        __syncthreads();
        for (int e = tid; e < E * LT; e += CUGS_BLOCK) {
            const int ey = e / LT, cx = e - ey * LT;
            float a = 0.0f, b = 0.0f, c = 0.0f;
#pragma unroll
            for (int k = 0; k <= 2 * R; ++k) {
                const float wk = win.w[k];
                a = fmaf(wk, s_a[0][ey * E + cx + k], a);
                b = fmaf(wk, s_a[1][ey * E + cx + k], b);
                c = fmaf(wk, s_a[2][ey * E + cx + k], c);
            }
            s_h[0][e] = a; s_h[1][e] = b; s_h[2][e] = c;
        }
        __syncthreads();
        float gm = 0.0f, gp = 0.0f, gr = 0.0f;
#pragma unroll
        for (int k = 0; k <= 2 * R; ++k) {
            const float wk = win.w[k];
            const int e = (ly + k) * LT + lx;
            gm = fmaf(wk, s_h[0][e], gm); gp = fmaf(wk, s_h[1][e], gp); gr = fmaf(wk, s_h[2][e], gr);
        }
        if (inside) {
            const int64_t o = ((int64_t)py * w + px) * 3 + ch;
            const float x = xr[o], y = yt[o], d = x - y;
            const float sgn = (d > 0.0f) ? 1.0f : ((d < 0.0f) ? -1.0f : 0.0f);          // d|x|/dx, 0 at 0 like libtorch
            dL_dx[o] = (1.0f - lambda) * sgn * inv_n - lambda * inv_n * (gm + 2.0f * x * gp + y * gr);
        }
        __syncthreads();
    }
}

// loss_out[0] = combined loss, [1] = L1 mean, [2] = mean SSIM, [3] = 1 - mean SSIM (ssim_loss)
__global__ void k_loss_finalize(const double* __restrict__ sums, double count, float lambda, float* __restrict__ out) {
    const double l1 = sums[0] / count, ss = sums[1] / count;
    out[0] = (float)((1.0 - (double)lambda) * l1 + (double)lambda * (1.0 - ss));
    out[1] = (float)l1;
    out[2] = (float)ss;
    out[3] = (float)(1.0 - ss);
}

}  // namespace

extern "C" size_t cugs_loss_workspace_bytes(int width, int height) {
    if (width < 0 || height < 0) return 0;
    return 256 + sizeof(float) * 3 * 3 * (size_t)width * (size_t)height;
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

    double* sums = static_cast<double*>(workspace);
    const size_t plane = 3 * (size_t)width * (size_t)height;
    float* d_m = reinterpret_cast<float*>(static_cast<char*>(workspace) + 256);
    float* d_p = d_m + plane;
    float* d_r = d_p + plane;
    CUGS_RETURN_IF_HIP(hipMemsetAsync(sums, 0, 2 * sizeof(double), st));
    dim3 grid((width + LT - 1) / LT, (height + LT - 1) / LT), block(CUGS_BLOCK);
    if (win.r == 5)      // the reference's default window (11): compile-time radius
        hipLaunchKernelGGL(k_ssim_stats<5>, grid, block, 0, st, width, height, win, rendered, target, d_m, d_p, d_r, ssim_map, sums);
    else
        hipLaunchKernelGGL(k_ssim_stats<0>, grid, block, 0, st, width, height, win, rendered, target, d_m, d_p, d_r, ssim_map, sums);
    CUGS_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(1), 0, st, sums, (double)width * height * 3.0, lambda, loss_out);
    CUGS_LAUNCH_CHECK();
    if (dL_dcolor) {
        if (win.r == 5)
            hipLaunchKernelGGL(k_ssim_grad<5>, grid, block, 0, st, width, height, win, lambda, rendered, target, d_m, d_p, d_r, dL_dcolor);
        else
            hipLaunchKernelGGL(k_ssim_grad<0>, grid, block, 0, st, width, height, win, lambda, rendered, target, d_m, d_p, d_r, dL_dcolor);
        CUGS_LAUNCH_CHECK();
    }
    return 0;
}
