// cugs_gaussian_math.h — per-Gaussian fp32 math for the projection kernels (device side).
//
// What is computed follows the reference (rasterizer/projection.cuh, rasterizer/backward.cuh);
// how it is organised is ours: small value types, no arrays indexed at run time (they would go
// to scratch on gfx950), everything forced inline into the one-thread-per-Gaussian kernels.
// The floating-point ASSOCIATION of every expression on the path that decides an integer
// (camera-space point, pixel mean, Sigma, Sigma', its inverse, the radius) is the reference's
// left-to-right order, so that radii / tile counts / depth keys are bit-identical to the oracle.
#pragma once

#include "cugs_common.h"

struct V3 { float x, y, z; };
struct Sym2 { float a, b, c; };                        // [[a,b],[b,c]]
struct Sym3 { float xx, xy, xz, yy, yz, zz; };         // upper triangle
struct M3 { float m00, m01, m02, m10, m11, m12, m20, m21, m22; };   // row-major
struct M23 { float r0x, r0y, r0z, r1x, r1y, r1z; };    // 2x3, row-major

// The blend backward (raster_backward.hip) accumulates, per Gaussian, the moments of dL/dpower over the pixel
// offsets d = pixel centre - mean:  M1 = sum dpw d,  M2 = sum dpw (dx^2, dx dy, dy^2).  With power = -q/2,
// q = d^T Sigma'^-1 d (backward.cu:200-213):
//   dL/dmean2d = sum dpw * Sigma'^-1 d = Sigma'^-1 M1                      (backward.cu:203-204)
//   dL/dSigma'^-1 = (-M2xx/2, -M2xy [combined off-diagonal, Q3], -M2yy/2)   (backward.cu:209-213)
struct GradMoments { float m1x, m1y, m2xx, m2xy, m2yy; };
struct Grad2D { float mx, my, a, b, c; };
__device__ __forceinline__ Grad2D grads_from_moments(const GradMoments& m, float a, float b, float c) {
    Grad2D g;
    g.mx = a * m.m1x + b * m.m1y;
    g.my = b * m.m1x + c * m.m1y;
    g.a = -0.5f * m.m2xx;
    g.b = -m.m2xy;
    g.c = -0.5f * m.m2yy;
    return g;
}

__device__ __forceinline__ M3 view_rotation(const CamArgs& c) {
    return M3{c.view[0], c.view[1], c.view[2], c.view[4], c.view[5], c.view[6],
              c.view[8], c.view[9], c.view[10]};
}

// t = W p + tau  (projection.cu:95-99)
__device__ __forceinline__ V3 to_camera(const CamArgs& c, const M3& W, V3 p) {
    V3 t;
    t.x = W.m00 * p.x + W.m01 * p.y + W.m02 * p.z + c.view[3];
    t.y = W.m10 * p.x + W.m11 * p.y + W.m12 * p.z + c.view[7];
    t.z = W.m20 * p.x + W.m21 * p.y + W.m22 * p.z + c.view[11];
    return t;
}

// Normalised-quaternion rotation (projection.cuh:28-49); also returns 1/|q| and the unit q.
struct QuatRot { M3 R; float inv_norm, w, x, y, z; };
__device__ __forceinline__ QuatRot rotation_of(float w, float x, float y, float z) {
    QuatRot o;
    o.inv_norm = cugs_rsqrtf(w * w + x * x + y * y + z * z + 1e-12f);
    w *= o.inv_norm; x *= o.inv_norm; y *= o.inv_norm; z *= o.inv_norm;
    o.w = w; o.x = x; o.y = y; o.z = z;
    o.R.m00 = 1.0f - 2.0f * (y * y + z * z);
    o.R.m01 = 2.0f * (x * y - w * z);
    o.R.m02 = 2.0f * (x * z + w * y);
    o.R.m10 = 2.0f * (x * y + w * z);
    o.R.m11 = 1.0f - 2.0f * (x * x + z * z);
    o.R.m12 = 2.0f * (y * z - w * x);
    o.R.m20 = 2.0f * (x * z - w * y);
    o.R.m21 = 2.0f * (y * z + w * x);
    o.R.m22 = 1.0f - 2.0f * (x * x + y * y);
    return o;
}

// M = R diag(s) (projection.cuh:79-83)
__device__ __forceinline__ M3 scale_columns(const M3& R, V3 s) {
    return M3{R.m00 * s.x, R.m01 * s.y, R.m02 * s.z, R.m10 * s.x, R.m11 * s.y, R.m12 * s.z,
              R.m20 * s.x, R.m21 * s.y, R.m22 * s.z};
}

// Sigma = M M^T, upper triangle (projection.cuh:85-90)
__device__ __forceinline__ Sym3 gram(const M3& M) {
    Sym3 S;
    S.xx = M.m00 * M.m00 + M.m01 * M.m01 + M.m02 * M.m02;
    S.xy = M.m00 * M.m10 + M.m01 * M.m11 + M.m02 * M.m12;
    S.xz = M.m00 * M.m20 + M.m01 * M.m21 + M.m02 * M.m22;
    S.yy = M.m10 * M.m10 + M.m11 * M.m11 + M.m12 * M.m12;
    S.yz = M.m10 * M.m20 + M.m11 * M.m21 + M.m12 * M.m22;
    S.zz = M.m20 * M.m20 + M.m21 * M.m21 + M.m22 * M.m22;
    return S;
}

// Perspective Jacobian entries (projection.cuh:117-131).  No tan-fov clamp.
struct Jac { float j00, j02, j11, j12, tz_inv, tz_inv2; };
__device__ __forceinline__ Jac jacobian(V3 t, float fx, float fy) {
    Jac J;
    J.tz_inv = 1.0f / (t.z + 1e-6f);
    J.tz_inv2 = J.tz_inv * J.tz_inv;
    J.j00 = fx * J.tz_inv;
    J.j02 = -fx * t.x * J.tz_inv2;
    J.j11 = fy * J.tz_inv;
    J.j12 = -fy * t.y * J.tz_inv2;
    return J;
}

// T = J W with J's structural zeros multiplied out explicitly, as projection.cuh:135-140 does
// (J[1] = J[3] = 0 still enter the sums there: x*0 + ... keeps the rounding identical).
__device__ __forceinline__ M23 project_matrix_full(const Jac& J, const M3& W) {
    const float z = 0.0f;
    M23 T;
    T.r0x = J.j00 * W.m00 + z * W.m10 + J.j02 * W.m20;
    T.r0y = J.j00 * W.m01 + z * W.m11 + J.j02 * W.m21;
    T.r0z = J.j00 * W.m02 + z * W.m12 + J.j02 * W.m22;
    T.r1x = z * W.m00 + J.j11 * W.m10 + J.j12 * W.m20;
    T.r1y = z * W.m01 + J.j11 * W.m11 + J.j12 * W.m21;
    T.r1z = z * W.m02 + J.j11 * W.m12 + J.j12 * W.m22;
    return T;
}

// The backward's T omits the zero terms (projection_backward.cu:103-109, backward.cuh:271-277).
__device__ __forceinline__ M23 project_matrix_sparse(const Jac& J, const M3& W) {
    M23 T;
    T.r0x = J.j00 * W.m00 + J.j02 * W.m20;
    T.r0y = J.j00 * W.m01 + J.j02 * W.m21;
    T.r0z = J.j00 * W.m02 + J.j02 * W.m22;
    T.r1x = J.j11 * W.m10 + J.j12 * W.m20;
    T.r1y = J.j11 * W.m11 + J.j12 * W.m21;
    T.r1z = J.j11 * W.m12 + J.j12 * W.m22;
    return T;
}

// T Sigma (2x3) (projection.cuh:148-154)
__device__ __forceinline__ M23 times_sym3(const M23& T, const Sym3& S) {
    M23 P;
    P.r0x = T.r0x * S.xx + T.r0y * S.xy + T.r0z * S.xz;
    P.r0y = T.r0x * S.xy + T.r0y * S.yy + T.r0z * S.yz;
    P.r0z = T.r0x * S.xz + T.r0y * S.yz + T.r0z * S.zz;
    P.r1x = T.r1x * S.xx + T.r1y * S.xy + T.r1z * S.xz;
    P.r1y = T.r1x * S.xy + T.r1y * S.yy + T.r1z * S.yz;
    P.r1z = T.r1x * S.xz + T.r1y * S.yz + T.r1z * S.zz;
    return P;
}

// Sigma' = T Sigma T^T + 0.3 I (projection.cuh:156-164)
__device__ __forceinline__ Sym2 screen_covariance(const M23& T, const Sym3& S) {
    M23 P = times_sym3(T, S);
    Sym2 c;
    c.a = P.r0x * T.r0x + P.r0y * T.r0y + P.r0z * T.r0z;
    c.b = P.r0x * T.r1x + P.r0y * T.r1y + P.r0z * T.r1z;
    c.c = P.r1x * T.r1x + P.r1y * T.r1y + P.r1z * T.r1z;
    c.a += 0.3f;
    c.c += 0.3f;
    return c;
}

// ceil(3 sqrt(lambda_max)) (projection.cuh:178-195)
__device__ __forceinline__ int splat_radius(const Sym2& c) {
    float det = c.a * c.c - c.b * c.b;
    float trace = c.a + c.c;
    float disc = fmaxf(trace * trace - 4.0f * det, 0.0f);
    float lambda_max = 0.5f * (trace + sqrtf(disc));
    if (lambda_max <= 0.0f) return 0;
    return cugs_f2i(ceilf(3.0f * sqrtf(lambda_max)));
}

// Inverse of Sigma'; returns det (<= 0: invalid, zeros) (projection.cuh:208-226)
__device__ __forceinline__ float invert_sym2(const Sym2& c, Sym2& inv) {
    float det = c.a * c.c - c.b * c.b;
    if (det <= 0.0f) { inv = Sym2{0.0f, 0.0f, 0.0f}; return 0.0f; }
    float inv_det = 1.0f / det;
    inv.a = c.c * inv_det;
    inv.b = -c.b * inv_det;
    inv.c = c.a * inv_det;
    return det;
}

// Tile rectangle [x0,x1) x [y0,y1) of a splat (projection.cu:172-188 == sorting.cu:52-57).
struct TileRect { int x0, y0, x1, y1; };
__device__ __forceinline__ TileRect tile_rect_of(float x, float y, int radius, int w, int h,
                                                 int ntx, int nty) {
    float rf = (float)radius;
    int min_x = max(0, cugs_f2i(x - rf));
    int min_y = max(0, cugs_f2i(y - rf));
    int max_x = min(w, cugs_f2i(x + rf + 1.0f));
    int max_y = min(h, cugs_f2i(y + rf + 1.0f));
    TileRect r;
    r.x0 = min_x / CUGS_TILE;
    r.y0 = min_y / CUGS_TILE;
    r.x1 = min(ntx, (max_x + CUGS_TILE - 1) / CUGS_TILE);
    r.y1 = min(nty, (max_y + CUGS_TILE - 1) / CUGS_TILE);
    return r;
}

// The sort's per-Gaussian record (sort.hip): the depth key and the 16-byte tile rectangle {x0, y0, w | h << 16,
// tiles_touched}.  One definition for the two producers - k_depth_keys_rect, which reads the projection's outputs back
// (the stage boundary of sort_gaussians, sorting.cu:115), and k_project_forward, which has them in registers
// (cugs_project_forward_keyed: render() saves a launch and 40 MB) - so the two routes cannot differ.
// `tr` is only read when tiles > 0 && radius > 0.  three_pass: the key is the depth's offset from the near plane
// (sort.hip, RADIX_DEPTH); *out_of_range reports a Gaussian that emits pairs outside that route's range.
constexpr int CUGS_DEPTH_BITS = 9;
constexpr uint32_t CUGS_DEPTH_KEY_BASE = 0x3E4CCCCCu;  // float bits of 0.2f, minus one: offsets of visible splats start at 1, 0 = "sorts first"
constexpr uint32_t CUGS_DEPTH_KEY_SPAN = 1u << (3 * CUGS_DEPTH_BITS);
struct SortRecord { uint32_t key; int4 rect; };
__device__ __forceinline__ SortRecord sort_record_of(float depth, int tiles, int radius, TileRect tr, bool three_pass,
                                                     bool* out_of_range) {
    uint32_t key = __float_as_uint(depth);
    int x0 = 0, y0 = 0, w = 0, h = 0;             // w x h = pairs the reference's loops would write
    if (tiles > 0) {
        if (radius > 0 && tr.x1 > tr.x0 && tr.y1 > tr.y0) {              // sorting.cu:44-45
            x0 = tr.x0; y0 = tr.y0; w = tr.x1 - tr.x0; h = tr.y1 - tr.y0;
        }
        if (w == 0) key = 0u;                                            // fills nothing: Q12
    }
    *out_of_range = false;
    if (three_pass) {
        const uint32_t off = key - CUGS_DEPTH_KEY_BASE;                  // wraps for keys below the base
        const bool emits = tiles > 0 && w > 0;
        *out_of_range = emits && !(off >= 1u && off < CUGS_DEPTH_KEY_SPAN);
        key = emits ? min(off, CUGS_DEPTH_KEY_SPAN - 1u) : 0u;           // a Gaussian without pairs may stand anywhere
    }
    return SortRecord{key, make_int4(x0, y0, w | (h << 16), tiles > 0 ? tiles : 0)};
}

// The tile-rectangle record of sort_record_of in 32 bits, for images of up to 127 x 127 tiles (2032 px a side) whose tile
// counts are the rectangles' own (the projection's; sort_gaussians takes tiles_touched as an INPUT and cannot assume it):
//   emits pairs:       x0 | y0 << 7 | w << 14 | h << 21      (tiles = w * h)
//   quirk Q12 / none:  1 << 31 | tiles                       (w = h = 0; tiles = 0: no pairs at all)
// Such a record rides through the depth sort's passes beside the Gaussian index (4 more bytes per pass and Gaussian),
// and the depth-ordered rectangles are then READ IN ORDER instead of gathered: the gather of one 16-byte record per
// Gaussian into depth order was 19 us of the 1 M-Gaussian frame and 127 us of the 6 M one (a random 16-byte gather
// costs a 64-byte fetch and a memory-system request each).
constexpr int CUGS_PRECT_MAX_TILES = 127;
__host__ __device__ inline bool cugs_prect_packable(int ntx, int nty) { return ntx <= CUGS_PRECT_MAX_TILES && nty <= CUGS_PRECT_MAX_TILES; }
__device__ __forceinline__ uint32_t pack_rect(int4 r) {
    const int w = r.z & 0xFFFF, h = r.z >> 16;
    if (w == 0) return 0x80000000u | (uint32_t)r.w;
    return (uint32_t)r.x | ((uint32_t)r.y << 7) | ((uint32_t)w << 14) | ((uint32_t)h << 21);
}
__device__ __forceinline__ int4 unpack_rect(uint32_t v) {
    if (v & 0x80000000u) return make_int4(0, 0, 0, (int)(v & 0x7FFFFFFFu));
    const int x0 = (int)(v & 127u), y0 = (int)((v >> 7) & 127u), w = (int)((v >> 14) & 127u), h = (int)((v >> 21) & 127u);
    return make_int4(x0, y0, w | (h << 16), w * h);
}

// normalize(p - c) with the norm clamped at 1e-8 (projection.cu:278-280)
__device__ __forceinline__ V3 view_direction(V3 p, const CamArgs& c) {
    float dx = p.x - c.cc[0], dy = p.y - c.cc[1], dz = p.z - c.cc[2];
    float nrm = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-8f);
    return V3{dx / nrm, dy / nrm, dz / nrm};
}

// ---------------------------------------------------------------------------------------
// Real SH basis, degrees 0..3, in the reference's association (core/sh_backward.cu:45-83).
// Y has 16 slots; entries >= (degree+1)^2 are left untouched.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void sh_basis(int degree, V3 d, float (&Y)[16]) {
    const float x = d.x, y = d.y, z = d.z;
    Y[0] = 0.28209479177387814f;
    if (degree >= 1) {
        Y[1] = -0.4886025119029199f * y;
        Y[2] = 0.4886025119029199f * z;
        Y[3] = -0.4886025119029199f * x;
    }
    if (degree >= 2) {
        const float xx = x * x, yy = y * y, zz = z * z;
        Y[4] = 1.0925484305920792f * (x * y);
        Y[5] = 1.0925484305920792f * (y * z);
        Y[6] = 0.31539156525252005f * (2 * zz - xx - yy);
        Y[7] = 1.0925484305920792f * (x * z);
        Y[8] = 0.5462742152960396f * (xx - yy);
    }
    if (degree >= 3) {
        const float xx = x * x, yy = y * y, zz = z * z;
        Y[9] = 0.5900435899266435f * y * (3 * xx - yy);
        Y[10] = 2.890611442640554f * x * y * z;
        Y[11] = 0.4570457994644658f * y * (4 * zz - xx - yy);
        Y[12] = 0.3731763325901154f * z * (2 * zz - 3 * xx - 3 * yy);
        Y[13] = 0.4570457994644658f * x * (4 * zz - xx - yy);
        Y[14] = 1.4453057213202769f * z * (xx - yy);
        Y[15] = 0.5900435899266435f * x * (xx - 3 * yy);
    }
}

// colour = sum_k c_k Y_k + 0.5 in the forward kernel's association (core/sh.cu:42-76:
// "K * c[k] * poly", i.e. (K*c)*poly, NOT c*(K*poly) as the backward's basis has it).
// `c` points at one channel's coefficients; `stride` is the element stride between them.
// Raw colour as the BACKWARD recomputes it for its ReLU gate (sh_backward.cu:92-99: sum of c_k * Y_k with the
// constants folded into Y_k, then + 0.5).  Not the forward's value: there the constant multiplies the
// coefficient first (sh.cu:44-77), so the two can differ in the last bit - and within an ulp of zero the gate the
// reference applies is THIS one's sign, whatever the forward wrote (a forward 6e-8 with a closed gate exists).
__device__ __forceinline__ float raw_colour_backward(const float* c, const float (&Y)[16], int num_active) {
    float raw = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if (k < num_active) raw += c[k] * Y[k];
    return raw + 0.5f;
}

template <typename Ptr>
__device__ __forceinline__ float sh_colour(int degree, Ptr c, int stride, V3 d) {
    const float x = d.x, y = d.y, z = d.z;
    float color = 0.0f;
    color += 0.28209479177387814f * c[0];
    if (degree >= 1) {
        color += 0.4886025119029199f * (-c[1 * stride] * y + c[2 * stride] * z + -c[3 * stride] * x);
    }
    if (degree >= 2) {
        const float xx = x * x, yy = y * y, zz = z * z;
        const float xy = x * y, xz = x * z, yz = y * z;
        color += 1.0925484305920792f * c[4 * stride] * xy;
        color += 1.0925484305920792f * c[5 * stride] * yz;
        color += 0.31539156525252005f * c[6 * stride] * (2 * zz - xx - yy);
        color += 1.0925484305920792f * c[7 * stride] * xz;
        color += 0.5462742152960396f * c[8 * stride] * (xx - yy);
    }
    if (degree >= 3) {
        const float xx = x * x, yy = y * y, zz = z * z;
        color += 0.5900435899266435f * c[9 * stride] * y * (3 * xx - yy);
        color += 2.890611442640554f * c[10 * stride] * x * y * z;
        color += 0.4570457994644658f * c[11 * stride] * y * (4 * zz - xx - yy);
        color += 0.3731763325901154f * c[12 * stride] * z * (2 * zz - 3 * xx - 3 * yy);
        color += 0.4570457994644658f * c[13 * stride] * x * (4 * zz - xx - yy);
        color += 1.4453057213202769f * c[14 * stride] * z * (xx - yy);
        color += 0.5900435899266435f * c[15 * stride] * x * (xx - 3 * yy);
    }
    return color + 0.5f;
}

// ---------------------------------------------------------------------------------------
// Packed projected-Gaussian record consumed by the blend kernels (CUGS_PACKED_STRIDE floats):
//   [0] mx  [1] my  [2] a  [3] b | [4] c  [5] opacity  [6] tau  [7] 0 | [8] r  [9] 0  [10] g  [11] bl
// Everything the per-wave cull reads (mean, Sigma'^-1, tau) sits in the first two 16-byte chunks; the colour chunk
// keeps green/blue as an aligned pair (v_pk_fma_f32 in the forward blend).  Word 7 is spare in HBM; the blend
// kernels' LDS copy keeps the Gaussian index there.
// tau = ln(255 * opacity) is the largest -power at which alpha can still reach 1/255; it only
// feeds the CONSERVATIVE per-wave cull (see cugs_raster_common.h), never a result, so the
// ocml logf here does not affect parity.  tau < 0 marks "can never contribute" (opacity < 1/255,
// or a culled Gaussian whose record is all zeros).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void write_packed(float* packed, int64_t idx, float mx, float my,
                                             Sym2 inv, float r, float g, float b, float opacity) {
    float tau = (opacity >= (1.0f / 255.0f)) ? logf(255.0f * opacity) : -1.0f;
    float4* dst = reinterpret_cast<float4*>(packed + idx * CUGS_PACKED_STRIDE);
    dst[0] = make_float4(mx, my, inv.a, inv.b);
    dst[1] = make_float4(inv.c, opacity, tau, 0.0f);
    dst[2] = make_float4(r, 0.0f, g, b);
}
