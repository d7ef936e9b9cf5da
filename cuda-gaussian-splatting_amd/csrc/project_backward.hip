// project_backward.hip — per-Gaussian chain rule from 2-D gradients to (p, q, log s, logit o) fused
// with the SH colour backward (SURVEY §8 a8+a9).
//
// Replaces, in ONE launch: k_project_backward (rasterizer/projection_backward.cu:26-247), the
// recomputed view directions (projection_backward.cu:332-338), k_evaluate_sh_backward
// (core/sh_backward.cu:29-112) and five zero-fills (projection_backward.cu:275-278, sh_backward.cu:138).
// Also provides the standalone evaluate_sh_backward_cuda surface (sh_backward.cu:114-156).
//
// gfx950 mapping: one thread per Gaussian.  The [n,3,C] gradient rows (the largest write of the
// whole backward, 12C B/Gaussian) are assembled per thread in LDS (odd dword row stride) and
// leave the workgroup as contiguous 16-byte stores.  The ReLU gate (sh_backward.cu:92-100 recomputes the
// raw colour from the coefficients) comes as three bits per Gaussian from cugs_project_forward, which has the
// coefficients in LDS anyway and makes the backward's own test there (colour_gate): no re-read of the
// 12C B/Gaussian coefficients.  (The forward's clamped rgb is NOT a substitute: its rounding differs from the
// backward's recomputation, and within an ulp of zero the two disagree.)  Bound: HBM; algorithmic bytes 44+4+64(+1) read,
// 44+12C(+8) written per Gaussian.
#include "cugs_gaussian_math.h"

namespace {

template <int C>
struct ShTile {
    static constexpr int ROW = 3 * C;
    static constexpr int LROW = (ROW % 2 == 0) ? ROW + 1 : ROW;
};

template <int C, bool ALIGNED>
__device__ __forceinline__ void load_sh_rows(const float* __restrict__ sh, int64_t base, int count, float* s_sh) {
    constexpr int ROW = ShTile<C>::ROW, LROW = ShTile<C>::LROW;
    const float* src = sh + base * ROW;
    const int total = count * ROW;
    const int tid = threadIdx.x;
    if (ALIGNED && count == CUGS_BLOCK) {
        // full workgroup: every thread issues ALL its 16-byte loads before the first LDS write, so the tile
        // costs one HBM latency instead of one per loop iteration
        constexpr int TOTAL4 = CUGS_BLOCK * ROW / 4, PER = (TOTAL4 + CUGS_BLOCK - 1) / CUGS_BLOCK;
        const float4* src4 = reinterpret_cast<const float4*>(src);
        float4 v[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int e4 = tid + i * CUGS_BLOCK;
            v[i] = (e4 < TOTAL4) ? src4[e4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int e4 = tid + i * CUGS_BLOCK;
            if (e4 < TOTAL4) {
                int e = e4 * 4;
                int row = e / ROW, col = e - row * ROW;
                const float vals[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    s_sh[row * LROW + col] = vals[k];
                    if (++col == ROW) { col = 0; ++row; }
                }
            }
        }
    } else if (ALIGNED) {
        const int total4 = total >> 2;
        const float4* src4 = reinterpret_cast<const float4*>(src);
        for (int e4 = threadIdx.x; e4 < total4; e4 += CUGS_BLOCK) {
            const float4 v = src4[e4];
            int e = e4 * 4, row = e / ROW, col = e - row * ROW;
            const float vals[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s_sh[row * LROW + col] = vals[k];
                if (++col == ROW) { col = 0; ++row; }
            }
        }
        for (int e = (total4 << 2) + threadIdx.x; e < total; e += CUGS_BLOCK) {
            int row = e / ROW, col = e - row * ROW;
            s_sh[row * LROW + col] = src[e];
        }
    } else {
        for (int e = threadIdx.x; e < total; e += CUGS_BLOCK) {
            int row = e / ROW, col = e - row * ROW;
            s_sh[row * LROW + col] = src[e];
        }
    }
}

// The gradient rows of a workgroup from their FACTORS: row r = gated[r][ch] * Y[r][k] (sh_backward.cu:99-108) - the tile
// holds 20 floats per Gaussian (the 16 basis values, the three gated colour gradients, one pad) instead of the 3C
// products, 20 KB instead of 50 KB per workgroup, so FIVE workgroups share a CU where three did (the kernel's 94 VGPRs
// bound it now), and the products are formed by the thread that stores them (the same single fp32 multiplication, so bit
// for bit the same rows).  DESIGN.md 4.6.
constexpr int SH_FACTOR_ROW = 20;
__device__ __forceinline__ void store_sh_rows_from_factors(float* __restrict__ dst_base, int64_t base, int count, int num_active,
                                                           const float* s_fac) {
    constexpr int C = 16, ROW4 = 3 * C / 4;                          // 12 float4 per Gaussian
    float4* dst4 = reinterpret_cast<float4*>(dst_base + base * (3 * C));
    const int total4 = count * ROW4;
#pragma unroll
    for (int i = 0; i < ROW4; ++i) {
        const int e4 = (int)threadIdx.x + i * CUGS_BLOCK;
        if (e4 < total4) {
            const int r = e4 / ROW4, j = e4 - r * ROW4, ch = j >> 2, k = (j & 3) * 4;
            const float4 y = *reinterpret_cast<const float4*>(s_fac + r * SH_FACTOR_ROW + k);
            const float g = s_fac[r * SH_FACTOR_ROW + 16 + ch];
            cugs_stnt(dst4 + e4, make_float4(k + 0 < num_active ? g * y.x : 0.0f, k + 1 < num_active ? g * y.y : 0.0f,
                                             k + 2 < num_active ? g * y.z : 0.0f, k + 3 < num_active ? g * y.w : 0.0f));
        }
    }
}

template <int C, bool ALIGNED>
__device__ __forceinline__ void store_sh_rows(float* __restrict__ dst_base, int64_t base, int count,
                                              const float* s_sh) {
    constexpr int ROW = ShTile<C>::ROW, LROW = ShTile<C>::LROW;
    float* dst = dst_base + base * ROW;
    const int total = count * ROW;
    if (ALIGNED) {
        const int total4 = total >> 2;
        float4* dst4 = reinterpret_cast<float4*>(dst);
        for (int e4 = threadIdx.x; e4 < total4; e4 += CUGS_BLOCK) {
            int e = e4 * 4, row = e / ROW, col = e - row * ROW;
            float vals[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                vals[k] = s_sh[row * LROW + col];
                if (++col == ROW) { col = 0; ++row; }
            }
            cugs_stnt(dst4 + e4, make_float4(vals[0], vals[1], vals[2], vals[3]));
        }
        for (int e = (total4 << 2) + threadIdx.x; e < total; e += CUGS_BLOCK) {
            int row = e / ROW, col = e - row * ROW;
            dst[e] = s_sh[row * LROW + col];
        }
    } else {
        for (int e = threadIdx.x; e < total; e += CUGS_BLOCK) {
            int row = e / ROW, col = e - row * ROW;
            dst[e] = s_sh[row * LROW + col];
        }
    }
}

// ---- fused Adam (single-GPU training: cugs_project_backward_adam) -------------------------------------------
// k_fused_adam's per-element update (optimizer/fused_adam.cu:44-76) in the reference's operation order - the same
// function adam.hip runs, so the fused and the two-kernel paths give identical bits.
struct AdamFusedArgs {
    float* m[5]; float* v[5];          // ParamGroup order: positions, sh_coeffs, opacities, scales, rotations
    float lr[5];
    float beta1, beta2, eps, bc1, bc2;
};
__device__ __forceinline__ void adam_update(float& p, float g, float& m, float& v, float lr, const AdamFusedArgs& h) {
    const float mi = h.beta1 * m + (1.0f - h.beta1) * g;
    m = mi;
    const float vi = h.beta2 * v + (1.0f - h.beta2) * g * g;
    v = vi;
    const float m_hat = mi * h.bc1;
    const float v_hat = vi * h.bc2;
    p -= lr * m_hat / (sqrtf(v_hat) + h.eps);
}
// The workgroup's SH gradient tile (LDS, padded rows) applied to its contiguous chunk of coefficients, moments
// read and written as 16-byte streams: the 12C B/Gaussian gradient never goes to memory.
template <int C, bool ALIGNED>
__device__ __forceinline__ void adam_sh_rows(float* __restrict__ param, float* __restrict__ mom, float* __restrict__ var,
                                             int64_t base, int count, const float* s_sh, float lr, const AdamFusedArgs& h) {
    constexpr int ROW = ShTile<C>::ROW, LROW = ShTile<C>::LROW;
    float* P = param + base * ROW; float* M = mom + base * ROW; float* V = var + base * ROW;
    const int total = count * ROW;
    int done = 0;
    if (ALIGNED) {
        const int total4 = total >> 2;
        for (int e4 = threadIdx.x; e4 < total4; e4 += CUGS_BLOCK) {
            float4 pp = cugs_ldnt(reinterpret_cast<float4*>(P) + e4), mm = cugs_ldnt(reinterpret_cast<float4*>(M) + e4),
                   vv = cugs_ldnt(reinterpret_cast<float4*>(V) + e4);             // streamed: read and written once
            int e = e4 * 4, row = e / ROW, col = e - row * ROW;
            float g[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                g[k] = s_sh[row * LROW + col];
                if (++col == ROW) { col = 0; ++row; }
            }
            adam_update(pp.x, g[0], mm.x, vv.x, lr, h); adam_update(pp.y, g[1], mm.y, vv.y, lr, h);
            adam_update(pp.z, g[2], mm.z, vv.z, lr, h); adam_update(pp.w, g[3], mm.w, vv.w, lr, h);
            cugs_stnt(reinterpret_cast<float4*>(P) + e4, pp); cugs_stnt(reinterpret_cast<float4*>(M) + e4, mm);
            cugs_stnt(reinterpret_cast<float4*>(V) + e4, vv);
        }
        done = total4 << 2;
    }
    for (int e = done + threadIdx.x; e < total; e += CUGS_BLOCK) {
        int row = e / ROW, col = e - row * ROW;
        adam_update(P[e], s_sh[row * LROW + col], M[e], V[e], lr, h);
    }
}

// adam_sh_rows with the gradient formed from the factor tile (store_sh_rows_from_factors): the same products, the same
// update, in batches of four 16-byte pieces per thread so that a workgroup's 36 loads per thread are not twelve
// dependent round trips.
__device__ __forceinline__ void adam_sh_rows_from_factors(float* __restrict__ param, float* __restrict__ mom, float* __restrict__ var,
                                                          int64_t base, int count, int num_active, const float* s_fac,
                                                          float lr, const AdamFusedArgs& h) {
    constexpr int C = 16, ROW4 = 3 * C / 4, BATCH = 4;
    float4* P = reinterpret_cast<float4*>(param + base * (3 * C));
    float4* M = reinterpret_cast<float4*>(mom + base * (3 * C));
    float4* V = reinterpret_cast<float4*>(var + base * (3 * C));
    const int total4 = count * ROW4;
#pragma unroll
    for (int b = 0; b < ROW4; b += BATCH) {
        float4 pp[BATCH], mm[BATCH], vv[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int e4 = min((int)threadIdx.x + (b + u) * CUGS_BLOCK, total4 - 1);      // clamped: the loads stay together
            pp[u] = cugs_ldnt(P + e4); mm[u] = cugs_ldnt(M + e4); vv[u] = cugs_ldnt(V + e4);   // streamed: read and written once
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int e4 = (int)threadIdx.x + (b + u) * CUGS_BLOCK;
            if (e4 < total4) {
                const int r = e4 / ROW4, j = e4 - r * ROW4, ch = j >> 2, k = (j & 3) * 4;
                const float4 y = *reinterpret_cast<const float4*>(s_fac + r * SH_FACTOR_ROW + k);
                const float g = s_fac[r * SH_FACTOR_ROW + 16 + ch];
                adam_update(pp[u].x, k + 0 < num_active ? g * y.x : 0.0f, mm[u].x, vv[u].x, lr, h);
                adam_update(pp[u].y, k + 1 < num_active ? g * y.y : 0.0f, mm[u].y, vv[u].y, lr, h);
                adam_update(pp[u].z, k + 2 < num_active ? g * y.z : 0.0f, mm[u].z, vv[u].z, lr, h);
                adam_update(pp[u].w, k + 3 < num_active ? g * y.w : 0.0f, mm[u].w, vv[u].w, lr, h);
                cugs_stnt(P + e4, pp[u]); cugs_stnt(M + e4, mm[u]); cugs_stnt(V + e4, vv[u]);
            }
        }
    }
}

__device__ __forceinline__ int active_count(int degree) { return (degree + 1) * (degree + 1); }


// dL/dSigma' from dL/dSigma'^-1 (backward.cuh:37-64): -S^-1 G S^-1 with the incoming
// off-diagonal halved (Q3).
__device__ __forceinline__ Sym2 grad_cov_from_inv(const Sym2& inv, const Sym2& g_inv) {
    const float a = inv.a, b = inv.b, c = inv.c;
    const float da = g_inv.a, db = g_inv.b * 0.5f, dc = g_inv.c;
    const float t00 = a * da + b * db, t01 = a * db + b * dc;
    const float t10 = b * da + c * db, t11 = b * db + c * dc;
    return Sym2{-(t00 * a + t01 * b), -(t00 * b + t01 * c), -(t10 * b + t11 * c)};
}

// dL/dSigma = T^T G T, upper triangle (backward.cuh:82-107)
__device__ __forceinline__ Sym3 grad_cov3d(const M23& T, const Sym2& g) {
    const float e0 = T.r0x * g.a + T.r1x * g.b, e1 = T.r0x * g.b + T.r1x * g.c;
    const float e2 = T.r0y * g.a + T.r1y * g.b, e3 = T.r0y * g.b + T.r1y * g.c;
    const float e4 = T.r0z * g.a + T.r1z * g.b, e5 = T.r0z * g.b + T.r1z * g.c;
    Sym3 d;
    d.xx = e0 * T.r0x + e1 * T.r1x;
    d.xy = e0 * T.r0y + e1 * T.r1y;
    d.xz = e0 * T.r0z + e1 * T.r1z;
    d.yy = e2 * T.r0y + e3 * T.r1y;
    d.yz = e2 * T.r0z + e3 * T.r1z;
    d.zz = e4 * T.r0z + e5 * T.r1z;
    return d;
}

// dL/dM = 2 G_full M (backward.cuh:123-153)
__device__ __forceinline__ M3 grad_M(const Sym3& d, const M3& M) {
    M3 o;
    o.m00 = 2.0f * (d.xx * M.m00 + d.xy * M.m10 + d.xz * M.m20);
    o.m01 = 2.0f * (d.xx * M.m01 + d.xy * M.m11 + d.xz * M.m21);
    o.m02 = 2.0f * (d.xx * M.m02 + d.xy * M.m12 + d.xz * M.m22);
    o.m10 = 2.0f * (d.xy * M.m00 + d.yy * M.m10 + d.yz * M.m20);
    o.m11 = 2.0f * (d.xy * M.m01 + d.yy * M.m11 + d.yz * M.m21);
    o.m12 = 2.0f * (d.xy * M.m02 + d.yy * M.m12 + d.yz * M.m22);
    o.m20 = 2.0f * (d.xz * M.m00 + d.yz * M.m10 + d.zz * M.m20);
    o.m21 = 2.0f * (d.xz * M.m01 + d.yz * M.m11 + d.zz * M.m21);
    o.m22 = 2.0f * (d.xz * M.m02 + d.yz * M.m12 + d.zz * M.m22);
    return o;
}

// dL/dq (raw, unnormalised) from dL/dR (backward.cuh:168-227)
__device__ __forceinline__ float4 grad_quat(const QuatRot& q, const M3& g) {
    const float w = q.w, x = q.x, y = q.y, z = q.z;
    const float dw = 2.0f * (-z * g.m01 + y * g.m02 + z * g.m10 - x * g.m12 + -y * g.m20 + x * g.m21);
    const float dx = 2.0f * (y * g.m01 + z * g.m02 + y * g.m10 - 2.0f * x * g.m11 - w * g.m12 +
                             z * g.m20 + w * g.m21 - 2.0f * x * g.m22);
    const float dy = 2.0f * (-2.0f * y * g.m00 + x * g.m01 + w * g.m02 + x * g.m10 + z * g.m12 +
                             -w * g.m20 + z * g.m21 - 2.0f * y * g.m22);
    const float dz = 2.0f * (-2.0f * z * g.m00 - w * g.m01 + x * g.m02 + w * g.m10 -
                             2.0f * z * g.m11 + y * g.m12 + x * g.m20 + y * g.m21);
    const float dot = dw * w + dx * x + dy * y + dz * z;
    return make_float4(q.inv_norm * (dw - w * dot), q.inv_norm * (dx - x * dot),
                       q.inv_norm * (dy - y * dot), q.inv_norm * (dz - z * dot));
}

// Contribution of Sigma' to dL/dt through J(t) (backward.cuh:248-346); adds into dt.
__device__ __forceinline__ void add_grad_t_from_cov(const Sym2& g, const Sym3& S, const M3& W, V3 t,
                                                    float fx, float fy, const Jac& J, const M23& T, V3& dt) {
    const M23 TS = times_sym3(T, S);
    const float k0 = 2.0f * (g.a * TS.r0x + g.b * TS.r1x), k1 = 2.0f * (g.a * TS.r0y + g.b * TS.r1y);
    const float k2 = 2.0f * (g.a * TS.r0z + g.b * TS.r1z), k3 = 2.0f * (g.b * TS.r0x + g.c * TS.r1x);
    const float k4 = 2.0f * (g.b * TS.r0y + g.c * TS.r1y), k5 = 2.0f * (g.b * TS.r0z + g.c * TS.r1z);
    const float j0 = k0 * W.m00 + k1 * W.m01 + k2 * W.m02;     // dL/dJ[0][0]
    const float j2 = k0 * W.m20 + k1 * W.m21 + k2 * W.m22;     // dL/dJ[0][2]
    const float j4 = k3 * W.m10 + k4 * W.m11 + k5 * W.m12;     // dL/dJ[1][1]
    const float j5 = k3 * W.m20 + k4 * W.m21 + k5 * W.m22;     // dL/dJ[1][2]
    const float tz_inv3 = J.tz_inv2 * J.tz_inv;
    dt.x += j2 * (-fx * J.tz_inv2);
    dt.y += j5 * (-fy * J.tz_inv2);
    dt.z += j0 * (-fx * J.tz_inv2) + j2 * (2.0f * fx * t.x * tz_inv3) + j4 * (-fy * J.tz_inv2) +
            j5 * (2.0f * fy * t.y * tz_inv3);
}

struct PBPtrs {
    const float* positions; const float* rotations; const float* scales; const float* opacities;
    const float* sh; const int32_t* radii; const uint8_t* colour_gate;
    const float* grad_accum;
    const float* g_means; const float* g_cov; const float* g_rgb; const float* g_opa;
    float* d_pos; float* d_rot; float* d_scl; float* d_opa; float* d_sh; float* d_means_out;
    float* d_rgb_gated_out;      // [n,3] gated colour gradient (for the data-parallel exchange); may be NULL
    // ADAM variant only: the parameters themselves, updated in place (d_pos .. d_sh are then unused)
    float* w_pos; float* w_rot; float* w_scl; float* w_opa; float* w_sh;
};

// ADAM: instead of writing the five parameter gradients, apply the Adam update to this Gaussian's parameters in
// the same pass (every thread touches only its own Gaussian; the SH block goes through the LDS tile): 236 B/Gaussian
// of gradient writes and as many reads by a separate optimizer launch disappear (472 of 2020 B at degree 3).
// FACTORS (C == 16, aligned rows, the gate bits given, no fused optimizer step): the LDS tile holds the factors of the
// gradient rows instead of the rows (store_sh_rows_from_factors); also the route without SH gradient rows at all.
template <int C, bool ALIGNED, bool ADAM, bool FACTORS = false>
__global__ __launch_bounds__(CUGS_BLOCK) void k_project_backward(int64_t n, int degree, CamArgs cam, PBPtrs p,
                                                                AdamFusedArgs adam) {
    static_assert(!FACTORS || (C == 16 && ALIGNED), "factor tile: degree-3 storage, 16-byte rows");
    constexpr int LROW = FACTORS ? SH_FACTOR_ROW : ShTile<C>::LROW;
    __shared__ __attribute__((aligned(16))) float s_sh[CUGS_BLOCK * LROW];
    const int64_t base = (int64_t)blockIdx.x * CUGS_BLOCK;
    const int count = (int)min((int64_t)CUGS_BLOCK, n - base);
    const int64_t idx = base + threadIdx.x;
    const bool live = idx < n;
    const int num_active = active_count(degree);
    // parameters are streamed (non-temporal) unless the fused optimizer step reads them again further down
    auto ldp = [](const float* q_) { return ADAM ? *q_ : cugs_ldnt(q_); };

    const bool gate_from_sh = !FACTORS && (p.colour_gate == nullptr);      // kernel-uniform
    if (gate_from_sh) {
        load_sh_rows<C, ALIGNED>(p.sh, base, count, s_sh);
        __syncthreads();
    }

    float Y[16] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    float gated[3] = {0.0f, 0.0f, 0.0f};
    V3 pos{0.0f, 0.0f, 0.0f};
    float g_mx = 0.0f, g_my = 0.0f, g_opa = 0.0f;
    Sym2 g_inv{0.0f, 0.0f, 0.0f};
    GradMoments mom{0.0f, 0.0f, 0.0f, 0.0f, 0.0f};            // grad_accum rows carry moments (raster_backward.hip)
    const bool from_rows = (p.grad_accum != nullptr);         // kernel-uniform
    // FACTORS: the geometry inputs are requested here, with everything else the thread reads, so that the kernel pays one
    // memory round trip and not a second one behind the gradient rows (same values, same arithmetic further down)
    float in_scl[3] = {0.0f, 0.0f, 0.0f}, in_opa = 0.0f;
    float4 in_q = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    int in_radius = 0;
    if (FACTORS && live) {
        in_radius = p.radii[idx];
        in_scl[0] = ldp(p.scales + idx * 3 + 0); in_scl[1] = ldp(p.scales + idx * 3 + 1); in_scl[2] = ldp(p.scales + idx * 3 + 2);
        in_q = ADAM ? reinterpret_cast<const float4*>(p.rotations)[idx] : cugs_ldnt(reinterpret_cast<const float4*>(p.rotations) + idx);
        in_opa = ldp(p.opacities + idx);
    }
    if (live) {
        pos = V3{ldp(p.positions + idx * 3 + 0), ldp(p.positions + idx * 3 + 1), ldp(p.positions + idx * 3 + 2)};
        float g_rgb[3];
        if (from_rows) {
            const float4* row = reinterpret_cast<const float4*>(p.grad_accum + idx * CUGS_GRAD_STRIDE);
            const float4 r0 = row[0], r1 = row[1];
            g_rgb[0] = r0.x; g_rgb[1] = r0.y; g_rgb[2] = r0.z; g_opa = r0.w;
            mom = GradMoments{r1.x, r1.y, r1.z, r1.w, p.grad_accum[idx * CUGS_GRAD_STRIDE + 8]};
        } else {
            g_rgb[0] = p.g_rgb[idx * 3 + 0]; g_rgb[1] = p.g_rgb[idx * 3 + 1]; g_rgb[2] = p.g_rgb[idx * 3 + 2];
            g_opa = p.g_opa[idx];
            g_mx = p.g_means[idx * 2 + 0]; g_my = p.g_means[idx * 2 + 1];
            g_inv = Sym2{p.g_cov[idx * 3 + 0], p.g_cov[idx * 3 + 1], p.g_cov[idx * 3 + 2]};
        }
        sh_basis(degree, view_direction(pos, cam), Y);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            bool open;
            if (gate_from_sh) open = raw_colour_backward(s_sh + threadIdx.x * LROW + ch * C, Y, num_active) > 0.0f;
            else open = ((p.colour_gate[idx] >> ch) & 1u) != 0u;      // the same test, made by cugs_project_forward
            gated[ch] = g_rgb[ch] * (open ? 1.0f : 0.0f);          // sh_backward.cu:99-100
            if (p.d_rgb_gated_out) p.d_rgb_gated_out[idx * 3 + ch] = gated[ch];
        }
    }
    if (gate_from_sh) __syncthreads();                         // coefficients consumed; reuse the tile

    if (FACTORS) {
        // kernel-uniform: without dL/dsh (the data-parallel exchange builds it from the gathered colour gradients,
        // cugs_sh_backward_views) nothing goes through the tile
        if (ADAM || p.d_sh) {
            if (live) {
                float4* row = reinterpret_cast<float4*>(s_sh + threadIdx.x * SH_FACTOR_ROW);
                row[0] = make_float4(Y[0], Y[1], Y[2], Y[3]);
                row[1] = make_float4(Y[4], Y[5], Y[6], Y[7]);
                row[2] = make_float4(Y[8], Y[9], Y[10], Y[11]);
                row[3] = make_float4(Y[12], Y[13], Y[14], Y[15]);
                row[4] = make_float4(gated[0], gated[1], gated[2], 0.0f);
            }
            __syncthreads();
            if (ADAM) adam_sh_rows_from_factors(p.w_sh, adam.m[1], adam.v[1], base, count, num_active, s_sh, adam.lr[1], adam);
            else store_sh_rows_from_factors(p.d_sh, base, count, num_active, s_sh);
        }
    } else if (ADAM || p.d_sh) {                               // kernel-uniform
        if (live) {
            float* row = s_sh + threadIdx.x * LROW;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch)
#pragma unroll
                for (int k = 0; k < C; ++k) row[ch * C + k] = (k < num_active) ? gated[ch] * Y[k] : 0.0f;
        }
        __syncthreads();
        if (ADAM) adam_sh_rows<C, ALIGNED>(p.w_sh, adam.m[1], adam.v[1], base, count, s_sh, adam.lr[1], adam);
        else store_sh_rows<C, ALIGNED>(p.d_sh, base, count, s_sh);
    }
    if (!live) return;

    // ---- geometry ----
    V3 d_pos{0.0f, 0.0f, 0.0f}, d_log{0.0f, 0.0f, 0.0f};
    float4 d_q = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float d_logit = 0.0f;
    if ((FACTORS ? in_radius : p.radii[idx]) > 0) {            // projection_backward.cu:48
        const M3 W = view_rotation(cam);
        const V3 t = to_camera(cam, W, pos);
        if (!FACTORS) {
            in_scl[0] = ldp(p.scales + idx * 3 + 0); in_scl[1] = ldp(p.scales + idx * 3 + 1); in_scl[2] = ldp(p.scales + idx * 3 + 2);
        }
        const V3 s{cugs_expf(in_scl[0] + cam.log_mod), cugs_expf(in_scl[1] + cam.log_mod), cugs_expf(in_scl[2] + cam.log_mod)};
        const float4 q = FACTORS ? in_q
                       : ALIGNED ? (ADAM ? reinterpret_cast<const float4*>(p.rotations)[idx]
                                         : cugs_ldnt(reinterpret_cast<const float4*>(p.rotations) + idx))
                                 : make_float4(p.rotations[idx * 4 + 0], p.rotations[idx * 4 + 1],
                                               p.rotations[idx * 4 + 2], p.rotations[idx * 4 + 3]);
        const QuatRot qr = rotation_of(q.x, q.y, q.z, q.w);
        const M3 M = scale_columns(qr.R, s);
        const Sym3 S = gram(M);
        const Jac J = jacobian(t, cam.fx, cam.fy);
        const Sym2 cov = screen_covariance(project_matrix_full(J, W), S);
        Sym2 inv;
        if (invert_sym2(cov, inv) > 0.0f) {                    // projection_backward.cu:91
            if (from_rows) {                                   // a Gaussian with a non-zero row passed this test in the forward
                const Grad2D g2 = grads_from_moments(mom, inv.a, inv.b, inv.c);
                g_mx = g2.mx; g_my = g2.my;
                g_inv = Sym2{g2.a, g2.b, g2.c};
            }
            const M23 T = project_matrix_sparse(J, W);
            const Sym2 g_cov = grad_cov_from_inv(inv, g_inv);
            const Sym3 g_S = grad_cov3d(T, g_cov);
            const M3 g_M = grad_M(g_S, M);
            // M = R diag(s): dL/dR_ij = dL/dM_ij s_j; dL/ds_j = sum_i dL/dM_ij R_ij; x s_j for log-space
            const M3 g_R{g_M.m00 * s.x, g_M.m01 * s.y, g_M.m02 * s.z, g_M.m10 * s.x, g_M.m11 * s.y,
                         g_M.m12 * s.z, g_M.m20 * s.x, g_M.m21 * s.y, g_M.m22 * s.z};
            const M3& R = qr.R;
            d_log.x = (g_M.m00 * R.m00 + g_M.m10 * R.m10 + g_M.m20 * R.m20) * s.x;
            d_log.y = (g_M.m01 * R.m01 + g_M.m11 * R.m11 + g_M.m21 * R.m21) * s.y;
            d_log.z = (g_M.m02 * R.m02 + g_M.m12 * R.m12 + g_M.m22 * R.m22) * s.z;
            d_q = grad_quat(qr, g_R);

            V3 dt{0.0f, 0.0f, 0.0f};                            // projection_backward.cu:194-199
            dt.x += g_mx * cam.fx * J.tz_inv;
            dt.y += g_my * cam.fy * J.tz_inv;
            dt.z += g_mx * (-cam.fx * t.x * J.tz_inv2) + g_my * (-cam.fy * t.y * J.tz_inv2);
            add_grad_t_from_cov(g_cov, S, W, t, cam.fx, cam.fy, J, T, dt);
            d_pos.x = W.m00 * dt.x + W.m10 * dt.y + W.m20 * dt.z;
            d_pos.y = W.m01 * dt.x + W.m11 * dt.y + W.m21 * dt.z;
            d_pos.z = W.m02 * dt.x + W.m12 * dt.y + W.m22 * dt.z;

            const float sig = cugs_sigmoidf(FACTORS ? in_opa : ldp(p.opacities + idx));
            d_logit = g_opa * sig * (1.0f - sig);
        }
    }
    if (ADAM) {
        const float gp[3] = {d_pos.x, d_pos.y, d_pos.z}, gs[3] = {d_log.x, d_log.y, d_log.z};
        const float gq[4] = {d_q.x, d_q.y, d_q.z, d_q.w};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float w = p.w_pos[idx * 3 + k], m = adam.m[0][idx * 3 + k], v = adam.v[0][idx * 3 + k];
            adam_update(w, gp[k], m, v, adam.lr[0], adam);
            p.w_pos[idx * 3 + k] = w; adam.m[0][idx * 3 + k] = m; adam.v[0][idx * 3 + k] = v;
        }
        {
            float w = p.w_opa[idx], m = adam.m[2][idx], v = adam.v[2][idx];
            adam_update(w, d_logit, m, v, adam.lr[2], adam);
            p.w_opa[idx] = w; adam.m[2][idx] = m; adam.v[2][idx] = v;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float w = p.w_scl[idx * 3 + k], m = adam.m[3][idx * 3 + k], v = adam.v[3][idx * 3 + k];
            adam_update(w, gs[k], m, v, adam.lr[3], adam);
            p.w_scl[idx * 3 + k] = w; adam.m[3][idx * 3 + k] = m; adam.v[3][idx * 3 + k] = v;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float w = p.w_rot[idx * 4 + k], m = adam.m[4][idx * 4 + k], v = adam.v[4][idx * 4 + k];
            adam_update(w, gq[k], m, v, adam.lr[4], adam);
            p.w_rot[idx * 4 + k] = w; adam.m[4][idx * 4 + k] = m; adam.v[4][idx * 4 + k] = v;
        }
    } else {
        // the gradients are next read by the optimizer (or the exchange), a frame's worth of traffic later
        cugs_stnt(p.d_pos + idx * 3 + 0, d_pos.x); cugs_stnt(p.d_pos + idx * 3 + 1, d_pos.y); cugs_stnt(p.d_pos + idx * 3 + 2, d_pos.z);
        if (ALIGNED) cugs_stnt(reinterpret_cast<float4*>(p.d_rot) + idx, d_q);
        else { p.d_rot[idx * 4 + 0] = d_q.x; p.d_rot[idx * 4 + 1] = d_q.y; p.d_rot[idx * 4 + 2] = d_q.z; p.d_rot[idx * 4 + 3] = d_q.w; }
        cugs_stnt(p.d_scl + idx * 3 + 0, d_log.x); cugs_stnt(p.d_scl + idx * 3 + 1, d_log.y); cugs_stnt(p.d_scl + idx * 3 + 2, d_log.z);
        cugs_stnt(p.d_opa + idx, d_logit);
    }
    if (p.d_means_out) { p.d_means_out[idx * 2 + 0] = g_mx; p.d_means_out[idx * 2 + 1] = g_my; }
}

// ---- the gated colour gradient on its own (data-parallel exchange, early gather) ----
// out[i][ch] = grad_accum[i][ch] * gate bit ch of colour_gate[i]: exactly what k_project_backward writes to
// dL_drgb_gated_out, available as soon as the backward blend has finished - the all-gather of these 12 B/Gaussian
// can then run UNDER the projection backward instead of after it.  Reads one 16-byte chunk of each 64-byte row.
__global__ __launch_bounds__(CUGS_BLOCK) void k_gated_colour_grad(int64_t n, const float* __restrict__ grad_accum,
                                                                  const uint8_t* __restrict__ colour_gate,
                                                                  float* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * CUGS_BLOCK + threadIdx.x;
    if (idx >= n) return;
    const float4 r0 = *reinterpret_cast<const float4*>(grad_accum + idx * CUGS_GRAD_STRIDE);
    const unsigned gate = colour_gate[idx];
    out[idx * 3 + 0] = r0.x * ((gate & 1u) ? 1.0f : 0.0f);             // sh_backward.cu:99-100
    out[idx * 3 + 1] = r0.y * ((gate & 2u) ? 1.0f : 0.0f);
    out[idx * 3 + 2] = r0.z * ((gate & 4u) ? 1.0f : 0.0f);
}

// ---- standalone SH backward (evaluate_sh_backward_cuda, core/sh_backward.cu:114-156) ----
template <int C, bool ALIGNED>
__global__ __launch_bounds__(CUGS_BLOCK) void k_sh_backward(int64_t n, int degree, const float* __restrict__ sh,
                                                            const float* __restrict__ dirs,
                                                            const float* __restrict__ dL_dcolor,
                                                            float* __restrict__ dL_dsh) {
    constexpr int LROW = ShTile<C>::LROW;
    __shared__ float s_sh[CUGS_BLOCK * LROW];
    const int64_t base = (int64_t)blockIdx.x * CUGS_BLOCK;
    const int count = (int)min((int64_t)CUGS_BLOCK, n - base);
    const int64_t idx = base + threadIdx.x;
    const bool live = idx < n;
    const int num_active = active_count(degree);
    load_sh_rows<C, ALIGNED>(sh, base, count, s_sh);
    __syncthreads();
    float Y[16] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    float gated[3] = {0.0f, 0.0f, 0.0f};
    if (live) {
        sh_basis(degree, V3{dirs[idx * 3 + 0], dirs[idx * 3 + 1], dirs[idx * 3 + 2]}, Y);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const bool open = raw_colour_backward(s_sh + threadIdx.x * LROW + ch * C, Y, num_active) > 0.0f;
            gated[ch] = dL_dcolor[idx * 3 + ch] * (open ? 1.0f : 0.0f);
        }
    }
    __syncthreads();
    if (live) {
        float* row = s_sh + threadIdx.x * LROW;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
#pragma unroll
            for (int k = 0; k < C; ++k) row[ch * C + k] = (k < num_active) ? gated[ch] * Y[k] : 0.0f;
    }
    __syncthreads();
    store_sh_rows<C, ALIGNED>(dL_dsh, base, count, s_sh);
}

// Any other coefficient count: straight from/to global.
__global__ __launch_bounds__(CUGS_BLOCK) void k_sh_backward_generic(int64_t n, int degree, int C,
                                                                    const float* __restrict__ sh,
                                                                    const float* __restrict__ dirs,
                                                                    const float* __restrict__ dL_dcolor,
                                                                    float* __restrict__ dL_dsh) {
    const int64_t idx = (int64_t)blockIdx.x * CUGS_BLOCK + threadIdx.x;
    if (idx >= n) return;
    const int num_active = active_count(degree);
    float Y[16] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    sh_basis(degree, V3{dirs[idx * 3 + 0], dirs[idx * 3 + 1], dirs[idx * 3 + 2]}, Y);
    for (int ch = 0; ch < 3; ++ch) {
        const float* c = sh + idx * 3 * C + (int64_t)ch * C;
        float* d = dL_dsh + idx * 3 * C + (int64_t)ch * C;
        float raw = 0.0f;
        for (int k = 0; k < 16; ++k)
            if (k < num_active) raw += c[k] * Y[k];
        raw += 0.5f;
        const float g = dL_dcolor[idx * 3 + ch] * ((raw > 0.0f) ? 1.0f : 0.0f);
        for (int k = 0; k < 16; ++k)
            if (k < num_active) d[k] = g * Y[k];
        for (int k = num_active; k < C; ++k) d[k] = 0.0f;
    }
}

// ---- data-parallel SH gradient: sum over V views of gated_v (x) Y(dir_v), in view order ----------
// dL_dsh = gated_rgb_grad (x) Y(direction) is an outer product, so V ranks exchange the 12 B/Gaussian
// gated colour gradients (all-gather) instead of all-reducing the 12C B/Gaussian SH gradients; every
// rank then rebuilds the identical sum with this kernel (fixed order: bit-reproducible across ranks).
constexpr int MAX_VIEWS = 16;
struct ViewCenters { float c[MAX_VIEWS][3]; int count; };

template <int C, bool ALIGNED>
__global__ __launch_bounds__(CUGS_BLOCK) void k_sh_backward_views(int64_t n, int degree,
                                                                  const float* __restrict__ positions,
                                                                  const float* __restrict__ gated,   // [V][n][3]
                                                                  ViewCenters vc, float* __restrict__ dL_dsh) {
    constexpr int LROW = ShTile<C>::LROW;
    __shared__ float s_sh[CUGS_BLOCK * LROW];
    const int64_t base = (int64_t)blockIdx.x * CUGS_BLOCK;
    const int count = (int)min((int64_t)CUGS_BLOCK, n - base);
    const int64_t idx = base + threadIdx.x;
    const int num_active = active_count(degree);
    if (idx < n) {
        const V3 pos{positions[idx * 3 + 0], positions[idx * 3 + 1], positions[idx * 3 + 2]};
        float acc[3][C];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
#pragma unroll
            for (int k = 0; k < C; ++k) acc[ch][k] = 0.0f;
        for (int v = 0; v < vc.count; ++v) {
            CamArgs cam;                                        // only the centre is used by view_direction()
            cam.cc[0] = vc.c[v][0]; cam.cc[1] = vc.c[v][1]; cam.cc[2] = vc.c[v][2];
            float Y[16] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            sh_basis(degree, view_direction(pos, cam), Y);
            const float* g = gated + ((int64_t)v * n + idx) * 3;
            const float g0 = g[0], g1 = g[1], g2 = g[2];
#pragma unroll
            for (int k = 0; k < C; ++k) {
                if (k < num_active) {
                    acc[0][k] += g0 * Y[k];
                    acc[1][k] += g1 * Y[k];
                    acc[2][k] += g2 * Y[k];
                }
            }
        }
        float* row = s_sh + threadIdx.x * LROW;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
#pragma unroll
            for (int k = 0; k < C; ++k) row[ch * C + k] = acc[ch][k];
    }
    __syncthreads();
    store_sh_rows<C, ALIGNED>(dL_dsh, base, count, s_sh);
}

inline unsigned grid_for(int64_t n) { return (unsigned)((n + CUGS_BLOCK - 1) / CUGS_BLOCK); }

template <int C>
int launch_shv(int64_t n, int degree, const float* pos, const float* gated, const ViewCenters& vc, float* out,
               bool aligned, hipStream_t st) {
    if (aligned)
        hipLaunchKernelGGL((k_sh_backward_views<C, true>), dim3(grid_for(n)), dim3(CUGS_BLOCK), 0, st, n, degree, pos, gated, vc, out);
    else
        hipLaunchKernelGGL((k_sh_backward_views<C, false>), dim3(grid_for(n)), dim3(CUGS_BLOCK), 0, st, n, degree, pos, gated, vc, out);
    CUGS_LAUNCH_CHECK();
    return 0;
}

template <int C>
int launch_pb(int64_t n, int degree, const CamArgs& cam, const PBPtrs& p, bool aligned, hipStream_t st,
              const AdamFusedArgs* adam = nullptr) {
    const AdamFusedArgs none{};
    if (adam) {
        if constexpr (C == 16) {
            if (aligned && p.colour_gate) {
                hipLaunchKernelGGL((k_project_backward<C, true, true, true>), dim3(grid_for(n)), dim3(CUGS_BLOCK), 0, st, n, degree, cam, p, *adam);
                CUGS_LAUNCH_CHECK();
                return 0;
            }
        }
        if (aligned)
            hipLaunchKernelGGL((k_project_backward<C, true, true>), dim3(grid_for(n)), dim3(CUGS_BLOCK), 0, st, n, degree, cam, p, *adam);
        else
            hipLaunchKernelGGL((k_project_backward<C, false, true>), dim3(grid_for(n)), dim3(CUGS_BLOCK), 0, st, n, degree, cam, p, *adam);
    } else if (aligned) {
        if constexpr (C == 16) {
            if (p.colour_gate) {
                hipLaunchKernelGGL((k_project_backward<C, true, false, true>), dim3(grid_for(n)), dim3(CUGS_BLOCK), 0, st, n, degree, cam, p, none);
                CUGS_LAUNCH_CHECK();
                return 0;
            }
        }
        hipLaunchKernelGGL((k_project_backward<C, true, false>), dim3(grid_for(n)), dim3(CUGS_BLOCK), 0, st, n, degree, cam, p, none);
    } else {
        hipLaunchKernelGGL((k_project_backward<C, false, false>), dim3(grid_for(n)), dim3(CUGS_BLOCK), 0, st, n, degree, cam, p, none);
    }
    CUGS_LAUNCH_CHECK();
    return 0;
}

template <int C>
int launch_shb(int64_t n, int degree, const float* sh, const float* dirs, const float* g, float* out,
               bool aligned, hipStream_t st) {
    if (aligned)
        hipLaunchKernelGGL((k_sh_backward<C, true>), dim3(grid_for(n)), dim3(CUGS_BLOCK), 0, st, n, degree, sh, dirs, g, out);
    else
        hipLaunchKernelGGL((k_sh_backward<C, false>), dim3(grid_for(n)), dim3(CUGS_BLOCK), 0, st, n, degree, sh, dirs, g, out);
    CUGS_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int cugs_project_backward(int64_t n, int num_coeffs, int active_degree, const float* positions,
                                     const float* rotations, const float* scales, const float* opacities,
                                     const float* sh_coeffs, const int32_t* radii, const uint8_t* colour_gate,
                                     const cugs_camera* camera_host, float scale_modifier,
                                     const float* grad_accum, const float* dL_dmeans_2d,
                                     const float* dL_dcov_2d_inv, const float* dL_drgb,
                                     const float* dL_dopacity_act, float* dL_dpositions, float* dL_drotations,
                                     float* dL_dscales, float* dL_dopacities, float* dL_dsh_coeffs,
                                     float* dL_dmeans_2d_out, float* dL_drgb_gated_out, void* stream) {
    if (n < 0 || !camera_host) return CUGS_EINVAL;
    if (active_degree < 0 || active_degree > 3) return CUGS_EINVAL;
    if ((active_degree + 1) * (active_degree + 1) > num_coeffs) return CUGS_EINVAL;
    if (num_coeffs != 1 && num_coeffs != 4 && num_coeffs != 9 && num_coeffs != 16) return CUGS_EINVAL;
    if (n == 0) return 0;
    if (!positions || !rotations || !scales || !opacities || !radii || !dL_dpositions || !dL_drotations ||
        !dL_dscales || !dL_dopacities)
        return CUGS_EINVAL;
    // dL_dsh_coeffs and dL_drgb_gated_out both NULL: geometry gradients only (cugs_gated_colour_grad took the colour half)
    if (!colour_gate && !sh_coeffs) return CUGS_EINVAL;
    if (!grad_accum && (!dL_dmeans_2d || !dL_dcov_2d_inv || !dL_drgb || !dL_dopacity_act)) return CUGS_EINVAL;
    if (grad_accum && !cugs_aligned16(grad_accum)) return CUGS_EALIGN;
    const CamArgs cam = cugs_make_cam_args(camera_host, scale_modifier);
    PBPtrs p{positions, rotations, scales, opacities, sh_coeffs, radii, colour_gate, grad_accum,
             dL_dmeans_2d, dL_dcov_2d_inv, dL_drgb, dL_dopacity_act, dL_dpositions, dL_drotations,
             dL_dscales, dL_dopacities, dL_dsh_coeffs, dL_dmeans_2d_out, dL_drgb_gated_out,
             nullptr, nullptr, nullptr, nullptr, nullptr};
    const bool aligned = (!dL_dsh_coeffs || cugs_aligned16(dL_dsh_coeffs)) && cugs_aligned16(rotations) && cugs_aligned16(dL_drotations) &&
                         (colour_gate || cugs_aligned16(sh_coeffs));
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (num_coeffs) {
        case 1: return launch_pb<1>(n, active_degree, cam, p, aligned, st);
        case 4: return launch_pb<4>(n, active_degree, cam, p, aligned, st);
        case 9: return launch_pb<9>(n, active_degree, cam, p, aligned, st);
        default: return launch_pb<16>(n, active_degree, cam, p, aligned, st);
    }
}

extern "C" int cugs_project_backward_adam(int64_t n, int num_coeffs, int active_degree, float* positions,
                                          float* rotations, float* scales, float* opacities, float* sh_coeffs,
                                          const int32_t* radii, const uint8_t* colour_gate,
                                          const cugs_camera* camera_host, float scale_modifier,
                                          const float* grad_accum, const cugs_adam_fused* adam_host,
                                          float* dL_dmeans_2d_out, void* stream) {
    if (n < 0 || !camera_host || !adam_host) return CUGS_EINVAL;
    if (active_degree < 0 || active_degree > 3) return CUGS_EINVAL;
    if ((active_degree + 1) * (active_degree + 1) > num_coeffs) return CUGS_EINVAL;
    if (num_coeffs != 1 && num_coeffs != 4 && num_coeffs != 9 && num_coeffs != 16) return CUGS_EINVAL;
    if (n == 0) return 0;
    if (!positions || !rotations || !scales || !opacities || !sh_coeffs || !radii || !colour_gate || !grad_accum)
        return CUGS_EINVAL;
    if (!cugs_aligned16(grad_accum)) return CUGS_EALIGN;
    AdamFusedArgs a;
    for (int g = 0; g < 5; ++g) {
        if (!adam_host->m[g] || !adam_host->v[g]) return CUGS_EINVAL;
        a.m[g] = adam_host->m[g]; a.v[g] = adam_host->v[g]; a.lr[g] = adam_host->lr[g];
    }
    a.beta1 = adam_host->beta1; a.beta2 = adam_host->beta2; a.eps = adam_host->eps;
    a.bc1 = adam_host->bc1; a.bc2 = adam_host->bc2;
    const CamArgs cam = cugs_make_cam_args(camera_host, scale_modifier);
    PBPtrs p{positions, rotations, scales, opacities, sh_coeffs, radii, colour_gate, grad_accum,
             nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, dL_dmeans_2d_out, nullptr,
             positions, rotations, scales, opacities, sh_coeffs};
    const bool aligned = cugs_aligned16(rotations) && cugs_aligned16(sh_coeffs) && cugs_aligned16(a.m[1]) &&
                         cugs_aligned16(a.v[1]);
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (num_coeffs) {
        case 1: return launch_pb<1>(n, active_degree, cam, p, aligned, st, &a);
        case 4: return launch_pb<4>(n, active_degree, cam, p, aligned, st, &a);
        case 9: return launch_pb<9>(n, active_degree, cam, p, aligned, st, &a);
        default: return launch_pb<16>(n, active_degree, cam, p, aligned, st, &a);
    }
}

extern "C" int cugs_evaluate_sh_backward(int degree, int64_t n, int num_coeffs, const float* sh_coeffs,
                                         const float* directions, const float* dL_dcolor, float* dL_dsh,
                                         void* stream) {
    if (degree < 0 || degree > 3 || n < 0) return CUGS_EINVAL;           // sh_backward.cu:120
    if ((degree + 1) * (degree + 1) > num_coeffs) return CUGS_EINVAL;
    if (n == 0) return 0;
    if (!sh_coeffs || !directions || !dL_dcolor || !dL_dsh) return CUGS_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool aligned = cugs_aligned16(sh_coeffs) && cugs_aligned16(dL_dsh);
    switch (num_coeffs) {
        case 1: return launch_shb<1>(n, degree, sh_coeffs, directions, dL_dcolor, dL_dsh, aligned, st);
        case 4: return launch_shb<4>(n, degree, sh_coeffs, directions, dL_dcolor, dL_dsh, aligned, st);
        case 9: return launch_shb<9>(n, degree, sh_coeffs, directions, dL_dcolor, dL_dsh, aligned, st);
        case 16: return launch_shb<16>(n, degree, sh_coeffs, directions, dL_dcolor, dL_dsh, aligned, st);
        default:
            hipLaunchKernelGGL(k_sh_backward_generic, dim3(grid_for(n)), dim3(CUGS_BLOCK), 0, st, n, degree,
                               num_coeffs, sh_coeffs, directions, dL_dcolor, dL_dsh);
            CUGS_LAUNCH_CHECK();
            return 0;
    }
}

extern "C" int cugs_sh_backward_views(int degree, int64_t n, int num_coeffs, const float* positions,
                                      int num_views, const float* gated_rgb_views,
                                      const float* cam_centers_host, float* dL_dsh, void* stream) {
    if (degree < 0 || degree > 3 || n < 0 || num_views < 1 || num_views > MAX_VIEWS) return CUGS_EINVAL;
    if ((degree + 1) * (degree + 1) > num_coeffs) return CUGS_EINVAL;
    if (num_coeffs != 1 && num_coeffs != 4 && num_coeffs != 9 && num_coeffs != 16) return CUGS_EINVAL;
    if (n == 0) return 0;
    if (!positions || !gated_rgb_views || !cam_centers_host || !dL_dsh) return CUGS_EINVAL;
    ViewCenters vc;
    vc.count = num_views;
    for (int v = 0; v < MAX_VIEWS; ++v)
        for (int k = 0; k < 3; ++k) vc.c[v][k] = v < num_views ? cam_centers_host[v * 3 + k] : 0.0f;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool aligned = cugs_aligned16(dL_dsh);
    switch (num_coeffs) {
        case 1: return launch_shv<1>(n, degree, positions, gated_rgb_views, vc, dL_dsh, aligned, st);
        case 4: return launch_shv<4>(n, degree, positions, gated_rgb_views, vc, dL_dsh, aligned, st);
        case 9: return launch_shv<9>(n, degree, positions, gated_rgb_views, vc, dL_dsh, aligned, st);
        default: return launch_shv<16>(n, degree, positions, gated_rgb_views, vc, dL_dsh, aligned, st);
    }
}

extern "C" int cugs_gated_colour_grad(int64_t n, const float* grad_accum, const uint8_t* colour_gate,
                                      float* dL_drgb_gated_out, void* stream) {
    if (n < 0) return CUGS_EINVAL;
    if (n == 0) return 0;
    if (!grad_accum || !colour_gate || !dL_drgb_gated_out) return CUGS_EINVAL;
    if (!cugs_aligned16(grad_accum)) return CUGS_EALIGN;
    hipLaunchKernelGGL(k_gated_colour_grad, dim3(grid_for(n)), dim3(CUGS_BLOCK), 0, static_cast<hipStream_t>(stream), n,
                       grad_accum, colour_gate, dL_drgb_gated_out);
    CUGS_LAUNCH_CHECK();
    return 0;
}
