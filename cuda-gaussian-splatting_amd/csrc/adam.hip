// adam.hip — fused Adam over the five Gaussian parameter tensors.
//
// Replaces k_fused_adam (optimizer/fused_adam.cu:44-76) and FusedAdam::step's five launches
// (fused_adam.cu:150-163) with ONE launch over all groups.  Per element, in the reference's
// operation order (no contraction):
//   m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;  p -= lr*(m*bc1) / (sqrt(v*bc2) + eps)
// bc1/bc2 arrive as floats computed in double on the host (fused_adam.cu:145-148).
//
// gfx950 mapping: a pure HBM stream, 16 B read for p,g,m,v and 12 B written per float (28 B).
// Every lane moves 16 bytes per access (float4) when the four pointers of a group are 16-byte
// aligned; a grid-stride loop over a flat "vector index" space spanning all groups keeps the
// launch at <= 2048 workgroups.
#include "cugs_common.h"

namespace {

constexpr int MAX_GROUPS = 8;

struct AdamGroups {
    float* param[MAX_GROUPS];
    const float* grad[MAX_GROUPS];
    float* m[MAX_GROUPS];
    float* v[MAX_GROUPS];
    int64_t n[MAX_GROUPS];
    int64_t vec_begin[MAX_GROUPS + 1];   // prefix of per-group work items (float4s or floats)
    float lr[MAX_GROUPS];
    int vec4[MAX_GROUPS];                // 1: items are float4 (+ scalar tail), 0: items are floats
    int ngroups;
};

struct AdamHyper { float beta1, beta2, eps, bc1, bc2; };

__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, float lr, const AdamHyper& h) {
    const float mi = h.beta1 * m + (1.0f - h.beta1) * g;
    m = mi;
    const float vi = h.beta2 * v + (1.0f - h.beta2) * g * g;
    v = vi;
    const float m_hat = mi * h.bc1;
    const float v_hat = vi * h.bc2;
    p -= lr * m_hat / (sqrtf(v_hat) + h.eps);
}

__global__ __launch_bounds__(CUGS_BLOCK) void k_fused_adam_groups(AdamGroups G, AdamHyper h) {
    const int64_t total = G.vec_begin[G.ngroups];
    for (int64_t w = (int64_t)blockIdx.x * CUGS_BLOCK + threadIdx.x; w < total;
         w += (int64_t)gridDim.x * CUGS_BLOCK) {
        int gi = 0;
#pragma unroll
        for (int k = 1; k < MAX_GROUPS; ++k)
            if (k < G.ngroups && w >= G.vec_begin[k]) gi = k;
        const int64_t i = w - G.vec_begin[gi];
        float* P = G.param[gi]; const float* Gr = G.grad[gi]; float* M = G.m[gi]; float* V = G.v[gi];
        const float lr = G.lr[gi];
        if (G.vec4[gi]) {
            const int64_t n4 = G.n[gi] >> 2;
            if (i < n4) {
                float4 p = cugs_ldnt(reinterpret_cast<float4*>(P) + i);             // every stream here is touched once
                const float4 g = cugs_ldnt(reinterpret_cast<const float4*>(Gr) + i);
                float4 m = cugs_ldnt(reinterpret_cast<float4*>(M) + i);
                float4 v = cugs_ldnt(reinterpret_cast<float4*>(V) + i);
                adam_elem(p.x, g.x, m.x, v.x, lr, h);
                adam_elem(p.y, g.y, m.y, v.y, lr, h);
                adam_elem(p.z, g.z, m.z, v.z, lr, h);
                adam_elem(p.w, g.w, m.w, v.w, lr, h);
                cugs_stnt(reinterpret_cast<float4*>(P) + i, p);
                cugs_stnt(reinterpret_cast<float4*>(M) + i, m);
                cugs_stnt(reinterpret_cast<float4*>(V) + i, v);
            } else {                                   // scalar tail: items n4 .. n4 + (n & 3)
                const int64_t e = (n4 << 2) + (i - n4);
                adam_elem(P[e], Gr[e], M[e], V[e], lr, h);
            }
        } else {
            adam_elem(P[i], Gr[i], M[i], V[i], lr, h);
        }
    }
}

}  // namespace

extern "C" void cugs_adam_bias_correction(float beta1, float beta2, int step, float* bc1_host, float* bc2_host) {
    const double b1 = (double)beta1, b2 = (double)beta2;
    *bc1_host = (float)(1.0 / (1.0 - pow(b1, step)));
    *bc2_host = (float)(1.0 / (1.0 - pow(b2, step)));
}

extern "C" int cugs_fused_adam_groups(const cugs_adam_group* groups_host, int ngroups, float beta1, float beta2,
                                      float eps, float bc1, float bc2, void* stream) {
    if (ngroups < 0 || ngroups > MAX_GROUPS || (ngroups > 0 && !groups_host)) return CUGS_EINVAL;
    AdamGroups G;
    G.ngroups = 0;
    G.vec_begin[0] = 0;
    for (int k = 0; k < ngroups; ++k) {
        const cugs_adam_group& g = groups_host[k];
        if (!g.grad || g.n == 0) continue;                        // fused_adam.cu:156, :193
        if (g.n < 0 || !g.param || !g.m || !g.v) return CUGS_EINVAL;
        const int j = G.ngroups++;
        G.param[j] = g.param; G.grad[j] = g.grad; G.m[j] = g.m; G.v[j] = g.v; G.n[j] = g.n; G.lr[j] = g.lr;
        const bool al = cugs_aligned16(g.param) && cugs_aligned16(g.grad) && cugs_aligned16(g.m) && cugs_aligned16(g.v);
        G.vec4[j] = al ? 1 : 0;
        const int64_t items = al ? ((g.n >> 2) + (g.n & 3)) : g.n;
        G.vec_begin[j + 1] = G.vec_begin[j] + items;
    }
    for (int j = G.ngroups; j < MAX_GROUPS; ++j) {
        G.param[j] = nullptr; G.grad[j] = nullptr; G.m[j] = nullptr; G.v[j] = nullptr;
        G.n[j] = 0; G.lr[j] = 0.0f; G.vec4[j] = 0; G.vec_begin[j + 1] = G.vec_begin[G.ngroups];
    }
    const int64_t total = G.vec_begin[G.ngroups];
    if (total == 0) return 0;
    const int64_t want = (total + CUGS_BLOCK - 1) / CUGS_BLOCK;
    const unsigned grid = (unsigned)(want < 2048 ? want : 2048);
    AdamHyper h{beta1, beta2, eps, bc1, bc2};
    hipLaunchKernelGGL(k_fused_adam_groups, dim3(grid), dim3(CUGS_BLOCK), 0, static_cast<hipStream_t>(stream), G, h);
    CUGS_LAUNCH_CHECK();
    return 0;
}

extern "C" int cugs_fused_adam(float* param, const float* grad, float* m, float* v, int64_t n, float lr,
                               float beta1, float beta2, float eps, float bc1, float bc2, void* stream) {
    if (n < 0) return CUGS_EINVAL;
    if (n == 0) return 0;
    if (!param || !grad || !m || !v) return CUGS_EINVAL;
    cugs_adam_group g{param, grad, m, v, n, lr, 0.0f};
    return cugs_fused_adam_groups(&g, 1, beta1, beta2, eps, bc1, bc2, stream);
}

// ---- misc entry points ------------------------------------------------------------------
extern "C" const char* cugs_version(void) { return "cugs-hip 0.1.0 (gfx950)"; }

extern "C" const char* cugs_error_string(int code) {
    switch (code) {
        case 0: return "success";
        case CUGS_EINVAL: return "cugs: invalid argument (size, degree or null pointer)";
        case CUGS_EALIGN: return "cugs: buffer is not aligned as required";
        case CUGS_EOVERFLOW: return "cugs: count does not fit int32";
        case CUGS_EWORKSPACE: return "cugs: workspace too small";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "cugs: unknown error";
    }
}

extern "C" int cugs_device_count(int* count_host) {
    if (!count_host) return CUGS_EINVAL;
    *count_host = 0;
    hipError_t e = hipGetDeviceCount(count_host);
    if (e != hipSuccess) { *count_host = 0; (void)hipGetLastError(); }
    return 0;
}
