// ply.hip — Gaussian-model checkpoint records.
//
// Replaces the per-float loops of write_gaussian_ply / read_gaussian_ply (utils/ply_io.cpp:98-196, 258-351):
// the reference moves the five SoA tensors to the host and interleaves them one ofstream::write per float
// (62 writes per Gaussian at SH degree 3), and reads by looking every value up through a string-keyed map.
// Here the PLY vertex record is assembled on the device - [x y z | nx ny nz = 0 | f_dc_0..2 | f_rest_* with
// the reference's (coefficient, channel) interleave | opacity | scale_0..2 | rot_0..3], optionally followed by
// the Adam moments in the same order twice (m_*, v_*: true resume, which the reference cannot do) - so the
// host side is one device-to-host copy and one write, and one read + one copy + one scatter on the way back.
// HBM-bound byte shuffling: (59 + 62) x 4 B per Gaussian, 3x that with the optimizer state.
#include "cugs_common.h"

namespace {

struct PlySrc { const float* pos; const float* sh; const float* opa; const float* scl; const float* rot; };
struct PlyDst { float* pos; float* sh; float* opa; float* scl; float* rot; };

__host__ __device__ inline int model_floats(int C) { return 3 + 3 * C + 1 + 3 + 4; }
__host__ __device__ inline int block_floats(int C) { return model_floats(C) + 3; }        // + the three normals

// Column `col` (0 .. block_floats-1) of one [x..rot_3] block -> which tensor and which element of Gaussian i.
// kind: 0 positions, 1 zero (normal), 2 sh, 3 opacity, 4 scales, 5 rotations.
__device__ __forceinline__ void locate(int col, int C, int& kind, int& elem) {
    if (col < 3) { kind = 0; elem = col; return; }
    if (col < 6) { kind = 1; elem = 0; return; }
    if (col < 9) { kind = 2; elem = (col - 6) * C; return; }                   // f_dc_ch = sh[ch][0] (ply_io.cpp:164-166)
    const int rest = 3 * (C - 1);
    if (col < 9 + rest) {                                                       // f_rest_{(k-1)*3+ch} = sh[ch][k] (:170-174)
        const int r = col - 9, k = 1 + r / 3, ch = r - (k - 1) * 3;
        kind = 2; elem = ch * C + k; return;
    }
    col -= 9 + rest;
    if (col < 1) { kind = 3; elem = 0; return; }
    if (col < 4) { kind = 4; elem = col - 1; return; }
    kind = 5; elem = col - 4;
}

__device__ __forceinline__ float fetch(const PlySrc& s, int kind, int elem, int64_t i, int C) {
    switch (kind) {
        case 0: return s.pos[i * 3 + elem];
        case 2: return s.sh[i * 3 * C + elem];
        case 3: return s.opa[i];
        case 4: return s.scl[i * 3 + elem];
        case 5: return s.rot[i * 4 + elem];
        default: return 0.0f;
    }
}

__global__ __launch_bounds__(CUGS_BLOCK) void k_ply_pack(int64_t n, int C, int nblocks_per_vertex, PlySrc p, PlySrc m,
                                                         PlySrc v, float* __restrict__ out) {
    const int bf = block_floats(C), row = bf + (nblocks_per_vertex - 1) * model_floats(C);
    const int64_t e = (int64_t)blockIdx.x * CUGS_BLOCK + threadIdx.x;
    if (e >= n * row) return;
    const int64_t i = e / row;
    int col = (int)(e - i * row);
    int kind, elem;
    if (col < bf) {
        locate(col, C, kind, elem);
        out[e] = fetch(p, kind, elem, i, C);
        return;
    }
    // moments: the same order without the normals
    col -= bf;
    const int mf = model_floats(C);
    const PlySrc& s = (col < mf) ? m : v;
    if (col >= mf) col -= mf;
    locate(col < 3 ? col : col + 3, C, kind, elem);
    out[e] = fetch(s, kind, elem, i, C);
}

// One model float per thread: canonical column c (0 .. 3*model_floats or model_floats) of Gaussian i comes from
// file column col_of[c] of a vertex record with `num_props` floats (properties are looked up by NAME on the host,
// ply_io.cpp:268-296, so files with extra or reordered properties load).
__global__ __launch_bounds__(CUGS_BLOCK) void k_ply_unpack(int64_t n, int C, int num_props, int nsets,
                                                           const float* __restrict__ data,
                                                           const int32_t* __restrict__ col_of, PlyDst p, PlyDst m,
                                                           PlyDst v) {
    const int mf = model_floats(C);
    const int64_t e = (int64_t)blockIdx.x * CUGS_BLOCK + threadIdx.x;
    if (e >= n * (int64_t)mf * nsets) return;
    const int64_t i = e / ((int64_t)mf * nsets);
    int c = (int)(e - i * (int64_t)mf * nsets);
    const int set = c / mf;
    c -= set * mf;
    const float val = data[i * num_props + col_of[set * mf + c]];
    const PlyDst& d = set == 0 ? p : (set == 1 ? m : v);
    int kind, elem;
    locate(c < 3 ? c : c + 3, C, kind, elem);
    switch (kind) {
        case 0: d.pos[i * 3 + elem] = val; break;
        case 2: d.sh[i * 3 * C + elem] = val; break;
        case 3: d.opa[i] = val; break;
        case 4: d.scl[i * 3 + elem] = val; break;
        case 5: d.rot[i * 4 + elem] = val; break;
        default: break;
    }
}

inline bool complete(const float* const a[5]) { return a && a[0] && a[1] && a[2] && a[3] && a[4]; }

}  // namespace

extern "C" int cugs_ply_vertex_floats(int num_coeffs, int with_state) {
    if (num_coeffs < 1) return CUGS_EINVAL;
    return block_floats(num_coeffs) + (with_state ? 2 * model_floats(num_coeffs) : 0);
}

extern "C" int cugs_ply_pack(int64_t n, int num_coeffs, const float* const params[5], const float* const m[5],
                             const float* const v[5], float* vertices, void* stream) {
    if (n < 0 || num_coeffs < 1) return CUGS_EINVAL;
    if (n == 0) return 0;
    if (!complete(params) || !vertices) return CUGS_EINVAL;
    const bool state = (m != nullptr) || (v != nullptr);
    if (state && (!complete(m) || !complete(v))) return CUGS_EINVAL;
    const int row = cugs_ply_vertex_floats(num_coeffs, state ? 1 : 0);
    const int64_t total = n * row;
    if (total > 2147483647ll * CUGS_BLOCK) return CUGS_EOVERFLOW;
    // group order of `params`, `m`, `v`: positions, sh_coeffs, opacities, scales, rotations (ParamGroup order)
    PlySrc P{params[0], params[1], params[2], params[3], params[4]};
    PlySrc M = state ? PlySrc{m[0], m[1], m[2], m[3], m[4]} : P;
    PlySrc V = state ? PlySrc{v[0], v[1], v[2], v[3], v[4]} : P;
    hipLaunchKernelGGL(k_ply_pack, dim3((unsigned)((total + CUGS_BLOCK - 1) / CUGS_BLOCK)), dim3(CUGS_BLOCK), 0,
                       static_cast<hipStream_t>(stream), n, num_coeffs, state ? 3 : 1, P, M, V, vertices);
    CUGS_LAUNCH_CHECK();
    return 0;
}

extern "C" int cugs_ply_unpack(int64_t n, int num_coeffs, int num_props, const float* vertices, const int32_t* col_of,
                               float* const params[5], float* const m[5], float* const v[5], void* stream) {
    if (n < 0 || num_coeffs < 1 || num_props < 1) return CUGS_EINVAL;
    if (n == 0) return 0;
    if (!vertices || !col_of || !params || !params[0] || !params[1] || !params[2] || !params[3] || !params[4])
        return CUGS_EINVAL;
    const bool state = (m != nullptr) || (v != nullptr);
    if (state && (!m || !v || !m[0] || !m[1] || !m[2] || !m[3] || !m[4] || !v[0] || !v[1] || !v[2] || !v[3] || !v[4]))
        return CUGS_EINVAL;
    PlyDst P{params[0], params[1], params[2], params[3], params[4]};
    PlyDst M = state ? PlyDst{m[0], m[1], m[2], m[3], m[4]} : P;
    PlyDst V = state ? PlyDst{v[0], v[1], v[2], v[3], v[4]} : P;
    const int nsets = state ? 3 : 1;
    const int64_t total = n * (int64_t)model_floats(num_coeffs) * nsets;
    if (total > 2147483647ll * CUGS_BLOCK) return CUGS_EOVERFLOW;
    hipLaunchKernelGGL(k_ply_unpack, dim3((unsigned)((total + CUGS_BLOCK - 1) / CUGS_BLOCK)), dim3(CUGS_BLOCK), 0,
                       static_cast<hipStream_t>(stream), n, num_coeffs, num_props, nsets, vertices, col_of, P, M, V);
    CUGS_LAUNCH_CHECK();
    return 0;
}
