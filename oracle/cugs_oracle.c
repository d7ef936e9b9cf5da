/* cugs_oracle.c — CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library, and only as the checker.  Nothing under cuda-gaussian-splatting_amd/
 * imports, links or calls it; the product path fails loudly without its HIP library.
 *
 * Each function restates one kernel of Artemarius/cuda-gaussian-splatting as plain
 * loops over the reference's array layouts, citing the file:line it follows
 * (paths relative to the reference's src/).  Arithmetic is fp32 in the
 * reference's operation order, compiled with -ffp-contract=off; the few fused
 * multiply-adds are explicit fmaf calls at the places listed in DESIGN.md
 * ("FMA placement contract"), which the HIP kernels mirror, so that every
 * integer output and every alpha / transmittance DECISION is bit-identical
 * between this file and the GPU.  expf / rsqrtf / sigmoid - and the blend's
 * exp(power) as cugs_blend_exp_q(q), <= 1e-6 relative from exp(-q/2) - come
 * from include/cugs_detmath.h for the same reason (see that header).
 *
 * PARITY STATUS (also in DESIGN.md): the reference's kernels are CUDA and
 * cannot run here; its tests hold no golden numbers for this path.  Pinned by:
 *   - oracle/_ref (the reference's own src/core/sh.cpp compiled against
 *     libtorch) for the SH forward;
 *   - the reference tests' known answers and properties (tests/test_reference_kats.py);
 *   - torch.optim.Adam, the reference's own comparator for FusedAdam;
 *   - an independent fp64 autograd model of the render equation for all gradients.
 * Sort order, tile counts, RGB and gradient VALUES are pinned by no reference
 * artefact: for those, parity is with this restatement ("parity unpinned" by
 * the reference), not with bits from an nvcc build.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include "../include/cugs_detmath.h"

#define TILE 16
#define RAST_BLOCK 256

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

/* ------------------------------------------------------------------------- */
/* Per-Gaussian helpers: rasterizer/projection.cuh                           */
/* ------------------------------------------------------------------------- */

/* projection.cuh:28-49 */
static void quat_to_rotation(float w, float x, float y, float z, float R[9]) {
    float inv_norm = cugs_rsqrtf(w * w + x * x + y * y + z * z + 1e-12f);
    w *= inv_norm; x *= inv_norm; y *= inv_norm; z *= inv_norm;
    R[0] = 1.0f - 2.0f * (y * y + z * z);
    R[1] = 2.0f * (x * y - w * z);
    R[2] = 2.0f * (x * z + w * y);
    R[3] = 2.0f * (x * y + w * z);
    R[4] = 1.0f - 2.0f * (x * x + z * z);
    R[5] = 2.0f * (y * z - w * x);
    R[6] = 2.0f * (x * z - w * y);
    R[7] = 2.0f * (y * z + w * x);
    R[8] = 1.0f - 2.0f * (x * x + y * y);
}

/* projection.cuh:65-90; M is also returned for the backward (projection_backward.cu:114-124) */
static void compute_cov_3d(const float log_scale[3], const float rot[4], float cov[6],
                           float M_out[9], float R_out[9], float s_out[3]) {
    float sx = cugs_expf(log_scale[0]);
    float sy = cugs_expf(log_scale[1]);
    float sz = cugs_expf(log_scale[2]);
    float R[9];
    quat_to_rotation(rot[0], rot[1], rot[2], rot[3], R);
    float M[9];
    M[0] = R[0] * sx; M[1] = R[1] * sy; M[2] = R[2] * sz;
    M[3] = R[3] * sx; M[4] = R[4] * sy; M[5] = R[5] * sz;
    M[6] = R[6] * sx; M[7] = R[7] * sy; M[8] = R[8] * sz;
    cov[0] = M[0] * M[0] + M[1] * M[1] + M[2] * M[2];
    cov[1] = M[0] * M[3] + M[1] * M[4] + M[2] * M[5];
    cov[2] = M[0] * M[6] + M[1] * M[7] + M[2] * M[8];
    cov[3] = M[3] * M[3] + M[4] * M[4] + M[5] * M[5];
    cov[4] = M[3] * M[6] + M[4] * M[7] + M[5] * M[8];
    cov[5] = M[6] * M[6] + M[7] * M[7] + M[8] * M[8];
    if (M_out) memcpy(M_out, M, sizeof(M));
    if (R_out) memcpy(R_out, R, sizeof(R));
    if (s_out) { s_out[0] = sx; s_out[1] = sy; s_out[2] = sz; }
}

/* projection.cuh:113-165 */
static void compute_cov_2d(const float cov_3d[6], const float W[9], const float t[3],
                           float fx, float fy, float cov_2d[3]) {
    float tx = t[0], ty = t[1], tz = t[2];
    float tz_inv = 1.0f / (tz + 1e-6f);
    float tz_inv2 = tz_inv * tz_inv;
    float J[6];
    J[0] = fx * tz_inv; J[1] = 0.0f; J[2] = -fx * tx * tz_inv2;
    J[3] = 0.0f; J[4] = fy * tz_inv; J[5] = -fy * ty * tz_inv2;
    float T[6];
    T[0] = J[0] * W[0] + J[1] * W[3] + J[2] * W[6];
    T[1] = J[0] * W[1] + J[1] * W[4] + J[2] * W[7];
    T[2] = J[0] * W[2] + J[1] * W[5] + J[2] * W[8];
    T[3] = J[3] * W[0] + J[4] * W[3] + J[5] * W[6];
    T[4] = J[3] * W[1] + J[4] * W[4] + J[5] * W[7];
    T[5] = J[3] * W[2] + J[4] * W[5] + J[5] * W[8];
    float S00 = cov_3d[0], S01 = cov_3d[1], S02 = cov_3d[2];
    float S11 = cov_3d[3], S12 = cov_3d[4], S22 = cov_3d[5];
    float TS[6];
    TS[0] = T[0] * S00 + T[1] * S01 + T[2] * S02;
    TS[1] = T[0] * S01 + T[1] * S11 + T[2] * S12;
    TS[2] = T[0] * S02 + T[1] * S12 + T[2] * S22;
    TS[3] = T[3] * S00 + T[4] * S01 + T[5] * S02;
    TS[4] = T[3] * S01 + T[4] * S11 + T[5] * S12;
    TS[5] = T[3] * S02 + T[4] * S12 + T[5] * S22;
    cov_2d[0] = TS[0] * T[0] + TS[1] * T[1] + TS[2] * T[2];
    cov_2d[1] = TS[0] * T[3] + TS[1] * T[4] + TS[2] * T[5];
    cov_2d[2] = TS[3] * T[3] + TS[4] * T[4] + TS[5] * T[5];
    cov_2d[0] += 0.3f;
    cov_2d[2] += 0.3f;
}

/* projection.cuh:178-195 */
static int compute_radius(const float cov_2d[3]) {
    float a = cov_2d[0], b = cov_2d[1], c = cov_2d[2];
    float det = a * c - b * b;
    float trace = a + c;
    float disc = fmaxf(trace * trace - 4.0f * det, 0.0f);
    float sqrt_disc = sqrtf(disc);
    float lambda_max = 0.5f * (trace + sqrt_disc);
    if (lambda_max <= 0.0f) return 0;
    float radius = ceilf(3.0f * sqrtf(lambda_max));
    return cugs_f2i(radius);
}

/* projection.cuh:208-226 */
static float compute_cov_2d_inverse(const float cov_2d[3], float inv[3]) {
    float a = cov_2d[0], b = cov_2d[1], c = cov_2d[2];
    float det = a * c - b * b;
    if (det <= 0.0f) { inv[0] = inv[1] = inv[2] = 0.0f; return 0.0f; }
    float inv_det = 1.0f / det;
    inv[0] = c * inv_det;
    inv[1] = -b * inv_det;
    inv[2] = a * inv_det;
    return det;
}

/* Tile rectangle of a projected Gaussian: projection.cu:172-188, repeated in
 * sorting.cu:52-57 (C truncation toward zero, then clamp, then /16). */
static void tile_rect(float x, float y, int radius, int img_w, int img_h,
                      int ntx, int nty, int* x0, int* y0, int* x1, int* y1) {
    float rf = (float)radius;
    int rect_min_x = imax(0, cugs_f2i(x - rf));
    int rect_min_y = imax(0, cugs_f2i(y - rf));
    int rect_max_x = imin(img_w, cugs_f2i(x + rf + 1.0f));
    int rect_max_y = imin(img_h, cugs_f2i(y + rf + 1.0f));
    *x0 = rect_min_x / TILE;
    *y0 = rect_min_y / TILE;
    *x1 = imin(ntx, (rect_max_x + TILE - 1) / TILE);
    *y1 = imin(nty, (rect_max_y + TILE - 1) / TILE);
}

/* ------------------------------------------------------------------------- */
/* k_project_gaussians: rasterizer/projection.cu:55-189                      */
/* Outputs are zero-filled first, as the launcher allocates them with zeros  */
/* (projection.cu:214-219).  view is the row-major 4x4 (projection.cu:228).  */
/* ------------------------------------------------------------------------- */
void orc_project_forward(int n, const float* positions, const float* rotations,
                         const float* scales, const float* opacities,
                         const float* view, float fx, float fy, float cx, float cy,
                         int img_w, int img_h, float scale_mod,
                         float* means_2d, float* depths, float* cov_2d_inv,
                         int32_t* radii, int32_t* tiles_touched, float* opacities_act) {
    memset(means_2d, 0, sizeof(float) * 2 * (size_t)n);
    memset(depths, 0, sizeof(float) * (size_t)n);
    memset(cov_2d_inv, 0, sizeof(float) * 3 * (size_t)n);
    memset(radii, 0, sizeof(int32_t) * (size_t)n);
    memset(tiles_touched, 0, sizeof(int32_t) * (size_t)n);
    memset(opacities_act, 0, sizeof(float) * (size_t)n);

    float W[9] = {view[0], view[1], view[2], view[4], view[5], view[6], view[8], view[9], view[10]};
    const float log_mod = logf(scale_mod + 1e-8f);   /* projection.cu:127-129 */
    int ntx = (img_w + TILE - 1) / TILE;
    int nty = (img_h + TILE - 1) / TILE;

    for (int idx = 0; idx < n; ++idx) {
        float px = positions[idx * 3 + 0], py = positions[idx * 3 + 1], pz = positions[idx * 3 + 2];
        float t_cam[3];
        t_cam[0] = W[0] * px + W[1] * py + W[2] * pz + view[3];
        t_cam[1] = W[3] * px + W[4] * py + W[5] * pz + view[7];
        t_cam[2] = W[6] * px + W[7] * py + W[8] * pz + view[11];
        if (t_cam[2] <= 0.2f) continue;                              /* :104 */

        float x_screen = fx * t_cam[0] / t_cam[2] + cx;               /* :109 */
        float y_screen = fy * t_cam[1] / t_cam[2] + cy;
        depths[idx] = t_cam[2];
        means_2d[idx * 2 + 0] = x_screen;
        means_2d[idx * 2 + 1] = y_screen;

        opacities_act[idx] = cugs_sigmoidf(opacities[idx]);           /* :119-121 */

        float log_scale[3] = {scales[idx * 3 + 0] + log_mod, scales[idx * 3 + 1] + log_mod,
                              scales[idx * 3 + 2] + log_mod};
        float rot[4] = {rotations[idx * 4 + 0], rotations[idx * 4 + 1], rotations[idx * 4 + 2],
                        rotations[idx * 4 + 3]};
        float cov_3d[6];
        compute_cov_3d(log_scale, rot, cov_3d, NULL, NULL, NULL);
        float cov2d[3];
        compute_cov_2d(cov_3d, W, t_cam, fx, fy, cov2d);
        float inv[3];
        float det = compute_cov_2d_inverse(cov2d, inv);
        if (det <= 0.0f) continue;                                    /* :152 */
        cov_2d_inv[idx * 3 + 0] = inv[0];
        cov_2d_inv[idx * 3 + 1] = inv[1];
        cov_2d_inv[idx * 3 + 2] = inv[2];

        int radius = compute_radius(cov2d);
        if (radius <= 0) continue;                                    /* :162 */
        radius = imin(radius, imax(img_w, img_h));                    /* :165-166 */
        radii[idx] = radius;

        int x0, y0, x1, y1;
        tile_rect(x_screen, y_screen, radius, img_w, img_h, ntx, nty, &x0, &y0, &x1, &y1);
        int n_tiles = (x1 - x0) * (y1 - y0);
        tiles_touched[idx] = imax(n_tiles, 0);                        /* :187-188 */
    }
}

/* View directions: projection.cu:273-280 (libtorch sub / norm / clamp_min / div). */
void orc_directions(int n, const float* positions, const float* cam_center, float* dirs) {
    for (int i = 0; i < n; ++i) {
        float dx = positions[i * 3 + 0] - cam_center[0];
        float dy = positions[i * 3 + 1] - cam_center[1];
        float dz = positions[i * 3 + 2] - cam_center[2];
        float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
        nrm = fmaxf(nrm, 1e-8f);
        dirs[i * 3 + 0] = dx / nrm;
        dirs[i * 3 + 1] = dy / nrm;
        dirs[i * 3 + 2] = dz / nrm;
    }
}

/* SH basis Y_k(dir) in the reference's association order: core/sh_backward.cu:45-83. */
static int sh_basis(int degree, float x, float y, float z, float Y[16]) {
    int num_active = 1;
    Y[0] = 0.28209479177387814f;
    if (degree >= 1) {
        Y[1] = -0.4886025119029199f * y;
        Y[2] = 0.4886025119029199f * z;
        Y[3] = -0.4886025119029199f * x;
        num_active = 4;
    }
    if (degree >= 2) {
        float xx = x * x, yy = y * y, zz = z * z;
        float xy = x * y, xz = x * z, yz = y * z;
        Y[4] = 1.0925484305920792f * xy;
        Y[5] = 1.0925484305920792f * yz;
        Y[6] = 0.31539156525252005f * (2 * zz - xx - yy);
        Y[7] = 1.0925484305920792f * xz;
        Y[8] = 0.5462742152960396f * (xx - yy);
        num_active = 9;
    }
    if (degree >= 3) {
        float xx = x * x, yy = y * y, zz = z * z;
        Y[9] = 0.5900435899266435f * y * (3 * xx - yy);
        Y[10] = 2.890611442640554f * x * y * z;
        Y[11] = 0.4570457994644658f * y * (4 * zz - xx - yy);
        Y[12] = 0.3731763325901154f * z * (2 * zz - 3 * xx - 3 * yy);
        Y[13] = 0.4570457994644658f * x * (4 * zz - xx - yy);
        Y[14] = 1.4453057213202769f * z * (xx - yy);
        Y[15] = 0.5900435899266435f * x * (xx - 3 * yy);
        num_active = 16;
    }
    return num_active;
}

/* k_evaluate_sh: core/sh.cu:19-79 (same arithmetic as evaluate_sh_cpu, core/sh.cpp:36-84).
 * Output is the UNclamped colour + 0.5; the clamp is projection.cu:284. */
void orc_sh_forward(int degree, int n, int num_coeffs, const float* sh, const float* dirs,
                    float* out) {
    for (int idx = 0; idx < n; ++idx) {
        float x = dirs[idx * 3 + 0], y = dirs[idx * 3 + 1], z = dirs[idx * 3 + 2];
        for (int ch = 0; ch < 3; ++ch) {
            const float* c = sh + (size_t)idx * 3 * num_coeffs + (size_t)ch * num_coeffs;
            float color = 0.0f;
            color += 0.28209479177387814f * c[0];
            if (degree >= 1) {
                color += 0.4886025119029199f * (-c[1] * y + c[2] * z + -c[3] * x);
            }
            if (degree >= 2) {
                float xx = x * x, yy = y * y, zz = z * z;
                float xy = x * y, xz = x * z, yz = y * z;
                color += 1.0925484305920792f * c[4] * xy;
                color += 1.0925484305920792f * c[5] * yz;
                color += 0.31539156525252005f * c[6] * (2 * zz - xx - yy);
                color += 1.0925484305920792f * c[7] * xz;
                color += 0.5462742152960396f * c[8] * (xx - yy);
            }
            if (degree >= 3) {
                float xx = x * x, yy = y * y, zz = z * z;
                color += 0.5900435899266435f * c[9] * y * (3 * xx - yy);
                color += 2.890611442640554f * c[10] * x * y * z;
                color += 0.4570457994644658f * c[11] * y * (4 * zz - xx - yy);
                color += 0.3731763325901154f * c[12] * z * (2 * zz - 3 * xx - 3 * yy);
                color += 0.4570457994644658f * c[13] * x * (4 * zz - xx - yy);
                color += 1.4453057213202769f * c[14] * z * (xx - yy);
                color += 0.5900435899266435f * c[15] * x * (xx - 3 * yy);
            }
            out[idx * 3 + ch] = color + 0.5f;
        }
    }
}

/* projection.cu:284 */
void orc_clamp_min0(int n, float* v) {
    for (int i = 0; i < n; ++i) v[i] = v[i] < 0.0f ? 0.0f : v[i];   /* NaN stays NaN like clamp_min */
}

/* k_evaluate_sh_backward: core/sh_backward.cu:29-112 */
void orc_sh_backward(int degree, int n, int num_coeffs, const float* sh, const float* dirs,
                     const float* dL_dcolor, float* dL_dsh) {
    for (int idx = 0; idx < n; ++idx) {
        float x = dirs[idx * 3 + 0], y = dirs[idx * 3 + 1], z = dirs[idx * 3 + 2];
        float Y[16];
        int num_active = sh_basis(degree, x, y, z, Y);
        for (int ch = 0; ch < 3; ++ch) {
            const float* c = sh + (size_t)idx * 3 * num_coeffs + (size_t)ch * num_coeffs;
            float* dsh = dL_dsh + (size_t)idx * 3 * num_coeffs + (size_t)ch * num_coeffs;
            float dL_dc = dL_dcolor[idx * 3 + ch];
            float raw = 0.0f;
            for (int k = 0; k < num_active; ++k) raw += c[k] * Y[k];
            raw += 0.5f;
            float gate = (raw > 0.0f) ? 1.0f : 0.0f;
            float g = dL_dc * gate;
            for (int k = 0; k < num_active; ++k) dsh[k] = g * Y[k];
            for (int k = num_active; k < num_coeffs; ++k) dsh[k] = 0.0f;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* sort_gaussians: rasterizer/sorting.cu:115-227                             */
/* ------------------------------------------------------------------------- */

/* sorting.cu:145-146: total pairs = last element of the int32 inclusive scan. */
int64_t orc_count_pairs(int n, const int32_t* tiles_touched) {
    int32_t acc = 0;
    for (int i = 0; i < n; ++i) acc = (int32_t)((uint32_t)acc + (uint32_t)tiles_touched[i]);
    return (int64_t)acc;
}

/* Stable LSD radix sort on the full 64-bit key: the contract of
 * cub::DeviceRadixSort::SortPairs the reference relies on (sorting.cu:191-210). */
static void stable_sort_pairs_u64(int64_t p, uint64_t* keys, int32_t* vals) {
    uint64_t* k2 = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(p > 0 ? p : 1));
    int32_t* v2 = (int32_t*)malloc(sizeof(int32_t) * (size_t)(p > 0 ? p : 1));
    uint64_t* ka = keys; int32_t* va = vals; uint64_t* kb = k2; int32_t* vb = v2;
    for (int pass = 0; pass < 8; ++pass) {
        int64_t count[257];
        memset(count, 0, sizeof(count));
        int sh = pass * 8;
        for (int64_t i = 0; i < p; ++i) count[((ka[i] >> sh) & 0xFF) + 1]++;
        for (int d = 0; d < 256; ++d) count[d + 1] += count[d];
        for (int64_t i = 0; i < p; ++i) {
            int64_t dst = count[(ka[i] >> sh) & 0xFF]++;
            kb[dst] = ka[i]; vb[dst] = va[i];
        }
        uint64_t* tk = ka; ka = kb; kb = tk;
        int32_t* tv = va; va = vb; vb = tv;
    }
    /* 8 passes: result is back in (keys, vals). */
    free(k2); free(v2);
}

/* k_fill_sort_pairs (sorting.cu:30-72) in Gaussian-index order at the exclusive-scan
 * offsets (sorting.cu:149-152), SortPairs, then k_compute_tile_ranges (sorting.cu:82-109)
 * into a zero-filled [tiles,2] buffer (sorting.cu:216).  total_pairs must equal
 * orc_count_pairs().
 *
 * Reference quirk kept (not in SURVEY's list): the unsorted key/value buffers
 * are torch::zeros (sorting.cu:166-167) and a Gaussian whose tile rectangle is empty in BOTH
 * axes has tiles_touched = (negative)*(negative) > 0 (projection.cu:187-188) while its fill
 * loops write nothing.  Its reserved slots therefore keep key 0 / value 0: pairs
 * (tile 0, depth bits 0, Gaussian 0) that sort to the very front of tile 0's list.
 * Returns 0, or -1 if a fill would run past total_pairs (inconsistent caller input). */
int orc_sort(int n, const float* means_2d, const float* depths, const int32_t* radii,
             const int32_t* tiles_touched, int img_w, int img_h, int64_t total_pairs,
             uint64_t* keys_sorted, int32_t* values_sorted, int32_t* tile_ranges) {
    int ntx = (img_w + TILE - 1) / TILE;
    int nty = (img_h + TILE - 1) / TILE;
    memset(tile_ranges, 0, sizeof(int32_t) * 2 * (size_t)ntx * (size_t)nty);
    if (n == 0 || total_pairs == 0) return 0;
    memset(keys_sorted, 0, sizeof(uint64_t) * (size_t)total_pairs);      /* sorting.cu:166 */
    memset(values_sorted, 0, sizeof(int32_t) * (size_t)total_pairs);     /* sorting.cu:167 */

    int64_t offset = 0;
    for (int idx = 0; idx < n; ++idx) {
        int radius = radii[idx];
        int64_t write_pos = offset;
        offset += tiles_touched[idx];
        if (radius <= 0) continue;
        float x = means_2d[idx * 2 + 0], y = means_2d[idx * 2 + 1];
        int x0, y0, x1, y1;
        tile_rect(x, y, radius, img_w, img_h, ntx, nty, &x0, &y0, &x1, &y1);
        uint32_t depth_bits = cugs_float_to_bits(depths[idx]);
        for (int ty = y0; ty < y1; ++ty)
            for (int tx = x0; tx < x1; ++tx) {
                if (write_pos >= total_pairs) return -1;
                uint64_t tile_id = (uint64_t)(ty * ntx + tx);
                keys_sorted[write_pos] = (tile_id << 32) | (uint64_t)depth_bits;
                values_sorted[write_pos] = idx;
                write_pos++;
            }
    }

    stable_sort_pairs_u64(total_pairs, keys_sorted, values_sorted);

    for (int64_t i = 0; i < total_pairs; ++i) {
        uint32_t cur = (uint32_t)(keys_sorted[i] >> 32);
        if (i == 0) {
            tile_ranges[cur * 2 + 0] = 0;
        } else {
            uint32_t prev = (uint32_t)(keys_sorted[i - 1] >> 32);
            if (cur != prev) {
                tile_ranges[prev * 2 + 1] = (int32_t)i;
                tile_ranges[cur * 2 + 0] = (int32_t)i;
            }
        }
        if (i == total_pairs - 1) tile_ranges[cur * 2 + 1] = (int32_t)total_pairs;
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Per-(pixel, Gaussian) evaluation shared by forward and backward:          */
/* forward.cu:124-141, backward.cu:123-137.                                  */
/* FMA placement contract: u = fma(a,dx,b*dy); v = fma(b,dx,c*dy);           */
/* q = fma(dx,u,dy*v); power = -0.5f*q.                                      */
/* Returns 0 when the Gaussian is skipped at this pixel.                     */
/* ------------------------------------------------------------------------- */
/* exp(power) is cugs_blend_exp_q(q) (include/cugs_detmath.h): exp(-q/2), evaluated in base 2 and clamped below */
/* at exp(-6), where opacity * it < 1/255 and the pair is skipped here as in the reference.                     */
/* Which exponential the blend restatement uses (test infrastructure: tests/test_exp_sensitivity.py measures how far the
 * image and the decisions move with the choice, i.e. what "parity with the reference's CUDA expf" can mean at all):
 *   0  cugs_blend_exp_q(q)            - the contract: shared bit for bit with the HIP kernels (default)
 *   1  cugs_expf(-0.5f * q)           - Cody-Waite form of round 1, the reference's own expression exp(power)
 *   2  libm expf(-0.5f * q)           - a correctly-rounded-to-1-ulp host exp, the closest stand-in for CUDA's 2-ulp expf
 * Modes 1 and 2 evaluate exp(power) for every power <= 0, without the clamp at exp(-6). */
static int g_blend_exp_mode = 0;
void orc_set_blend_exp_mode(int mode) { g_blend_exp_mode = mode; }
int orc_get_blend_exp_mode(void) { return g_blend_exp_mode; }

static inline int eval_alpha(float pxf, float pyf, float mx, float my, float a, float b, float c,
                             float opacity, float* dx_o, float* dy_o, float* exp_power_o,
                             float* alpha_o) {
    float dx = pxf - mx;
    float dy = pyf - my;
    float u = fmaf(a, dx, b * dy);
    float v = fmaf(b, dx, c * dy);
    float q = fmaf(dx, u, dy * v);
    float power = -0.5f * q;
    if (power > 0.0f) return 0;
    float exp_power = g_blend_exp_mode == 0 ? cugs_blend_exp_q(q)
                      : g_blend_exp_mode == 1 ? cugs_expf(power) : expf(power);
    float alpha = opacity * exp_power;
    alpha = fminf(alpha, 0.99f);
    if (alpha < 1.0f / 255.0f) return 0;
    *dx_o = dx; *dy_o = dy; *exp_power_o = exp_power; *alpha_o = alpha;
    return 1;
}

/* k_rasterize_forward: rasterizer/forward.cu:48-174.  The shared-memory batching and the
 * block-wide early exit do not change any pixel's result (a pixel that is `done`
 * ignores later batches), so the restatement walks each pixel's tile list directly.
 * FMA placement contract: C[ch] = fma(weight, rgb[ch], C[ch]); out = fma(T, bg, C). */
static void rasterize_forward_rows_impl(int nthreads, int img_w, int img_h, int row0, int row1, const float bg[3],
                                const int32_t* tile_ranges, const int32_t* gaussian_idx,
                                const float* means_2d, const float* cov_2d_inv, const float* rgb,
                                const float* opacities, float* out_color, float* out_final_T,
                                int32_t* out_n_contrib) {
    int ntx = (img_w + TILE - 1) / TILE;
    const float thr = 1.0f / 255.0f;                       /* forward.cuh:29 */
    (void)img_h;
    /* every pixel is independent: the OpenMP split over rows (nthreads > 1; full-frame checks at 1080p) cannot change
     * a bit of the result */
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 4) if (nthreads > 1)
    for (int py = row0; py < row1; ++py)
        for (int px = 0; px < img_w; ++px) {
            int tile_id = (py / TILE) * ntx + (px / TILE);
            int start = tile_ranges[tile_id * 2 + 0], end = tile_ranges[tile_id * 2 + 1];
            float pxf = (float)px + 0.5f, pyf = (float)py + 0.5f;
            float T = 1.0f, C0 = 0.0f, C1 = 0.0f, C2 = 0.0f;
            int count = 0;
            for (int s = start; s < end; ++s) {
                int g = gaussian_idx[s];
                float dx, dy, power, alpha;
                if (!eval_alpha(pxf, pyf, means_2d[g * 2], means_2d[g * 2 + 1], cov_2d_inv[g * 3],
                                cov_2d_inv[g * 3 + 1], cov_2d_inv[g * 3 + 2], opacities[g], &dx,
                                &dy, &power, &alpha))
                    continue;
                float weight = alpha * T;
                C0 = fmaf(weight, rgb[g * 3 + 0], C0);
                C1 = fmaf(weight, rgb[g * 3 + 1], C1);
                C2 = fmaf(weight, rgb[g * 3 + 2], C2);
                T *= (1.0f - alpha);
                count++;
                if (T < thr) break;                        /* forward.cu:153-156 */
            }
            int pix = py * img_w + px;
            out_color[pix * 3 + 0] = fmaf(T, bg[0], C0);
            out_color[pix * 3 + 1] = fmaf(T, bg[1], C1);
            out_color[pix * 3 + 2] = fmaf(T, bg[2], C2);
            out_final_T[pix] = T;
            out_n_contrib[pix] = count;
        }
}

void orc_rasterize_forward_rows(int img_w, int img_h, int row0, int row1, const float bg[3],
                                const int32_t* tile_ranges, const int32_t* gaussian_idx,
                                const float* means_2d, const float* cov_2d_inv, const float* rgb,
                                const float* opacities, float* out_color, float* out_final_T,
                                int32_t* out_n_contrib) {
    rasterize_forward_rows_impl(1, img_w, img_h, row0, row1, bg, tile_ranges, gaussian_idx, means_2d, cov_2d_inv, rgb,
                                opacities, out_color, out_final_T, out_n_contrib);
}

/* the same rows on `nthreads` host threads (identical bits) */
void orc_rasterize_forward_rows_mt(int nthreads, int img_w, int img_h, int row0, int row1, const float bg[3],
                                   const int32_t* tile_ranges, const int32_t* gaussian_idx,
                                   const float* means_2d, const float* cov_2d_inv, const float* rgb,
                                   const float* opacities, float* out_color, float* out_final_T,
                                   int32_t* out_n_contrib) {
    rasterize_forward_rows_impl(nthreads < 1 ? 1 : nthreads, img_w, img_h, row0, row1, bg, tile_ranges, gaussian_idx,
                                means_2d, cov_2d_inv, rgb, opacities, out_color, out_final_T, out_n_contrib);
}

void orc_rasterize_forward(int img_w, int img_h, const float bg[3], const int32_t* tile_ranges,
                           const int32_t* gaussian_idx, const float* means_2d,
                           const float* cov_2d_inv, const float* rgb, const float* opacities,
                           float* out_color, float* out_final_T, int32_t* out_n_contrib) {
    orc_rasterize_forward_rows(img_w, img_h, 0, img_h, bg, tile_ranges, gaussian_idx, means_2d,
                               cov_2d_inv, rgb, opacities, out_color, out_final_T, out_n_contrib);
}

/* k_rasterize_backward: rasterizer/backward.cu:31-233.  Per-contribution values are fp32
 * exactly as the reference computes them; the nine atomicAdd targets (backward.cu:217-228)
 * are accumulated in fp64 and rounded once, so that the oracle is a fair referee for any
 * summation order (the reference's own order is whatever the atomics happen to be).
 * Quirks kept: contributors counted from the END of the list (backward.cu:140-145),
 * T /= max(1-alpha, 1e-5) (:150-151), clamp gate on o*exp(power) >= 0.99 (:181-191),
 * combined off-diagonal dL/db = -dx*dy (:211). */
/* mag (optional, [n,9] doubles, zeroed by the caller): the MAGNITUDES the sums are made of - what bounds the error of
 * any fp32 summation of them (tools/fuzz_parity.py uses it to tell summation noise on a cancelling sum from a wrong
 * gradient): sum |drgb_c| (3), sum |dL_dopa|, sum |dpw dx|, sum |dpw dy|, sum |dpw| dx^2, sum |dpw dx dy|,
 * sum |dpw| dy^2 (dpw = dL_dpower), where dL_dopa and dpw enter with the magnitude of the OPERANDS of dL_dalpha. */
static void rasterize_backward_accumulate(int img_w, int row0, int row1, const float bg[3],
                                 const int32_t* tile_ranges, const int32_t* gaussian_idx,
                                 const float* means_2d, const float* cov_2d_inv, const float* rgb,
                                 const float* opacities, const float* dL_dcolor,
                                 const float* final_T, const int32_t* n_contrib, double* acc, double* mag) {
    int ntx = (img_w + TILE - 1) / TILE;
    for (int py = row0; py < row1; ++py)
        for (int px = 0; px < img_w; ++px) {
            int tile_id = (py / TILE) * ntx + (px / TILE);
            int start = tile_ranges[tile_id * 2 + 0], end = tile_ranges[tile_id * 2 + 1];
            float pxf = (float)px + 0.5f, pyf = (float)py + 0.5f;
            int pix = py * img_w + px;
            float T = final_T[pix];
            int max_contrib = n_contrib[pix];
            float dC0 = dL_dcolor[pix * 3 + 0], dC1 = dL_dcolor[pix * 3 + 1],
                  dC2 = dL_dcolor[pix * 3 + 2];
            float S0 = T * bg[0], S1 = T * bg[1], S2 = T * bg[2];   /* :83-87 */
            int found = 0;
            for (int s = end - 1; s >= start; --s) {
                int g = gaussian_idx[s];
                float a = cov_2d_inv[g * 3 + 0], b = cov_2d_inv[g * 3 + 1],
                      c = cov_2d_inv[g * 3 + 2];
                float o = opacities[g];
                float dx, dy, exp_power, alpha;
                if (!eval_alpha(pxf, pyf, means_2d[g * 2], means_2d[g * 2 + 1], a, b, c, o, &dx,
                                &dy, &exp_power, &alpha))
                    continue;
                found++;
                if (found > max_contrib) break;              /* :141-145 */
                float one_minus_alpha = fmaxf(1.0f - alpha, 1e-5f);
                T /= one_minus_alpha;
                float weight = alpha * T;
                float r0 = rgb[g * 3 + 0], r1 = rgb[g * 3 + 1], r2 = rgb[g * 3 + 2];
                float dr0 = dC0 * weight, dr1 = dC1 * weight, dr2 = dC2 * weight;
                float dL_dalpha = 0.0f;
                dL_dalpha += dC0 * (T * r0 - S0 / one_minus_alpha);
                dL_dalpha += dC1 * (T * r1 - S1 / one_minus_alpha);
                dL_dalpha += dC2 * (T * r2 - S2 / one_minus_alpha);
                const float S0m = S0, S1m = S1, S2m = S2;               /* the S this contribution was differenced against */
                S0 += weight * r0; S1 += weight * r1; S2 += weight * r2;
                float dL_dopa = dL_dalpha * exp_power;
                float dL_dpower = dL_dalpha * alpha;
                if (o * exp_power >= 0.99f) { dL_dopa = 0.0f; dL_dpower = 0.0f; }
                float dmx = dL_dpower * (a * dx + b * dy);
                float dmy = dL_dpower * (b * dx + c * dy);
                float da = dL_dpower * (-0.5f * dx * dx);
                float db = dL_dpower * (-dx * dy);
                float dc = dL_dpower * (-0.5f * dy * dy);
                double* A = acc + (size_t)g * 9;
                A[0] += dr0; A[1] += dr1; A[2] += dr2; A[3] += dL_dopa;
                A[4] += dmx; A[5] += dmy; A[6] += da; A[7] += db; A[8] += dc;
                if (mag) {
                    /* dL/dalpha is itself a difference - per channel T r_c against S_c / (1 - alpha), then over the
                     * channels (and, in the product, T G against D / (1 - alpha)): its magnitude is that of the
                     * operands, not of the result */
                    double* M = mag + (size_t)g * 9;
                    double da_mag = fabs((double)dC0) * (fabs((double)T * r0) + fabs((double)S0m) / one_minus_alpha)
                                  + fabs((double)dC1) * (fabs((double)T * r1) + fabs((double)S1m) / one_minus_alpha)
                                  + fabs((double)dC2) * (fabs((double)T * r2) + fabs((double)S2m) / one_minus_alpha);
                    int clamped = (o * exp_power >= 0.99f);
                    double pw = clamped ? 0.0 : da_mag * alpha, ax = fabs((double)dx), ay = fabs((double)dy);
                    M[0] += fabs((double)dr0); M[1] += fabs((double)dr1); M[2] += fabs((double)dr2);
                    M[3] += clamped ? 0.0 : da_mag * exp_power;
                    M[4] += pw * ax; M[5] += pw * ay; M[6] += pw * ax * ax; M[7] += pw * ax * ay; M[8] += pw * ay * ay;
                }
            }
        }
}

/* the fp64 sums rounded once (see above) */
static void rasterize_backward_finish(int n_gaussians, const double* acc, float* dL_drgb, float* dL_dopacity_act,
                                      float* dL_dmeans_2d, float* dL_dcov_2d_inv) {
    for (int g = 0; g < n_gaussians; ++g) {
        const double* A = acc + (size_t)g * 9;
        dL_drgb[g * 3 + 0] = (float)A[0]; dL_drgb[g * 3 + 1] = (float)A[1];
        dL_drgb[g * 3 + 2] = (float)A[2];
        dL_dopacity_act[g] = (float)A[3];
        dL_dmeans_2d[g * 2 + 0] = (float)A[4]; dL_dmeans_2d[g * 2 + 1] = (float)A[5];
        dL_dcov_2d_inv[g * 3 + 0] = (float)A[6]; dL_dcov_2d_inv[g * 3 + 1] = (float)A[7];
        dL_dcov_2d_inv[g * 3 + 2] = (float)A[8];
    }
}

static void rasterize_backward_rows_impl(int img_w, int img_h, int row0, int row1, const float bg[3],
                                 const int32_t* tile_ranges, const int32_t* gaussian_idx,
                                 const float* means_2d, const float* cov_2d_inv, const float* rgb,
                                 const float* opacities, const float* dL_dcolor,
                                 const float* final_T, const int32_t* n_contrib, int n_gaussians,
                                 float* dL_drgb, float* dL_dopacity_act, float* dL_dmeans_2d,
                                 float* dL_dcov_2d_inv, double* mag) {
    (void)img_h;
    double* acc = (double*)calloc((size_t)(n_gaussians > 0 ? n_gaussians : 1) * 9, sizeof(double));
    rasterize_backward_accumulate(img_w, row0, row1, bg, tile_ranges, gaussian_idx, means_2d, cov_2d_inv, rgb, opacities,
                                  dL_dcolor, final_T, n_contrib, acc, mag);
    rasterize_backward_finish(n_gaussians, acc, dL_drgb, dL_dopacity_act, dL_dmeans_2d, dL_dcov_2d_inv);
    free(acc);
}

/* The same rows on `nthreads` host threads (full-frame checks at 1080p).  The rows are cut into BANDS of 64 (a fixed
 * cut: it does not depend on the thread count); a band is summed in fp64 by one thread in the serial pixel order, and
 * the bands' tables are added to the total in band order (`omp ordered`) - so the result is the same for every
 * thread count, and differs from the one-table serial sum only by fp64 association (~1e-16 relative, before the one
 * rounding to fp32).  Memory: (nthreads + 1) tables of 72 bytes per Gaussian.  Returns 0, or -1 if out of memory. */
int orc_rasterize_backward_rows_mt(int nthreads, int img_w, int img_h, int row0, int row1, const float bg[3],
                                   const int32_t* tile_ranges, const int32_t* gaussian_idx,
                                   const float* means_2d, const float* cov_2d_inv, const float* rgb,
                                   const float* opacities, const float* dL_dcolor,
                                   const float* final_T, const int32_t* n_contrib, int n_gaussians,
                                   float* dL_drgb, float* dL_dopacity_act, float* dL_dmeans_2d,
                                   float* dL_dcov_2d_inv, double* mag /* optional [n,9], zeroed by the caller */) {
    (void)img_h;
    const int BAND = 64;
    const size_t cells = (size_t)(n_gaussians > 0 ? n_gaussians : 1) * 9;
    const int nb = row1 > row0 ? (row1 - row0 + BAND - 1) / BAND : 0;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > nb) nthreads = nb > 0 ? nb : 1;
    double* total = (double*)calloc(cells, sizeof(double));
    if (!total) return -1;
    int failed = 0;
#pragma omp parallel num_threads(nthreads)
    {
        double* acc = (double*)calloc(cells, sizeof(double));
        double* tmag = mag ? (double*)calloc(cells, sizeof(double)) : NULL;
        if (!acc || (mag && !tmag)) {
#pragma omp atomic write
            failed = 1;
            free(acc);
            acc = NULL;
        }
#pragma omp for ordered schedule(dynamic, 1)
        for (int b = 0; b < nb; ++b) {
            const int r0 = row0 + b * BAND, r1 = (r0 + BAND < row1) ? r0 + BAND : row1;
            if (acc)
                rasterize_backward_accumulate(img_w, r0, r1, bg, tile_ranges, gaussian_idx, means_2d, cov_2d_inv, rgb,
                                              opacities, dL_dcolor, final_T, n_contrib, acc, tmag);
#pragma omp ordered
            if (acc)
                for (size_t i = 0; i < cells; ++i) {
                    if (acc[i] != 0.0) { total[i] += acc[i]; acc[i] = 0.0; }
                    if (tmag && tmag[i] != 0.0) { mag[i] += tmag[i]; tmag[i] = 0.0; }
                }
        }
        free(acc);
        free(tmag);
    }
    if (!failed) rasterize_backward_finish(n_gaussians, total, dL_drgb, dL_dopacity_act, dL_dmeans_2d, dL_dcov_2d_inv);
    free(total);
    return failed ? -1 : 0;
}

void orc_rasterize_backward_rows(int img_w, int img_h, int row0, int row1, const float bg[3],
                                 const int32_t* tile_ranges, const int32_t* gaussian_idx,
                                 const float* means_2d, const float* cov_2d_inv, const float* rgb,
                                 const float* opacities, const float* dL_dcolor,
                                 const float* final_T, const int32_t* n_contrib, int n_gaussians,
                                 float* dL_drgb, float* dL_dopacity_act, float* dL_dmeans_2d,
                                 float* dL_dcov_2d_inv) {
    rasterize_backward_rows_impl(img_w, img_h, row0, row1, bg, tile_ranges, gaussian_idx, means_2d, cov_2d_inv, rgb,
                                 opacities, dL_dcolor, final_T, n_contrib, n_gaussians, dL_drgb, dL_dopacity_act,
                                 dL_dmeans_2d, dL_dcov_2d_inv, NULL);
}

/* the same sums plus their magnitudes (see rasterize_backward_rows_impl) */
void orc_rasterize_backward_magnitudes(int img_w, int img_h, const float bg[3], const int32_t* tile_ranges,
                                       const int32_t* gaussian_idx, const float* means_2d, const float* cov_2d_inv,
                                       const float* rgb, const float* opacities, const float* dL_dcolor,
                                       const float* final_T, const int32_t* n_contrib, int n_gaussians,
                                       float* dL_drgb, float* dL_dopacity_act, float* dL_dmeans_2d,
                                       float* dL_dcov_2d_inv, double* mag) {
    rasterize_backward_rows_impl(img_w, img_h, 0, img_h, bg, tile_ranges, gaussian_idx, means_2d, cov_2d_inv, rgb,
                                 opacities, dL_dcolor, final_T, n_contrib, n_gaussians, dL_drgb, dL_dopacity_act,
                                 dL_dmeans_2d, dL_dcov_2d_inv, mag);
}

void orc_rasterize_backward(int img_w, int img_h, const float bg[3], const int32_t* tile_ranges,
                            const int32_t* gaussian_idx, const float* means_2d,
                            const float* cov_2d_inv, const float* rgb, const float* opacities,
                            const float* dL_dcolor, const float* final_T, const int32_t* n_contrib,
                            int n_gaussians, float* dL_drgb, float* dL_dopacity_act,
                            float* dL_dmeans_2d, float* dL_dcov_2d_inv) {
    orc_rasterize_backward_rows(img_w, img_h, 0, img_h, bg, tile_ranges, gaussian_idx, means_2d,
                                cov_2d_inv, rgb, opacities, dL_dcolor, final_T, n_contrib,
                                n_gaussians, dL_drgb, dL_dopacity_act, dL_dmeans_2d,
                                dL_dcov_2d_inv);
}

/* ------------------------------------------------------------------------- */
/* Backward helpers: rasterizer/backward.cuh                                 */
/* ------------------------------------------------------------------------- */

/* backward.cuh:37-64 */
static void dcov2d_from_dcov2d_inv(const float inv[3], const float d_inv[3], float d_cov[3]) {
    float a = inv[0], b = inv[1], c = inv[2];
    float da = d_inv[0], db = d_inv[1] * 0.5f, dc = d_inv[2];
    float tmp00 = a * da + b * db;
    float tmp01 = a * db + b * dc;
    float tmp10 = b * da + c * db;
    float tmp11 = b * db + c * dc;
    d_cov[0] = -(tmp00 * a + tmp01 * b);
    d_cov[1] = -(tmp00 * b + tmp01 * c);
    d_cov[2] = -(tmp10 * b + tmp11 * c);
}

/* backward.cuh:82-107 */
static void dcov3d_from_dcov2d(const float T[6], const float d2[3], float d3[6]) {
    float da = d2[0], db = d2[1], dc = d2[2];
    float TtD[6];
    TtD[0] = T[0] * da + T[3] * db;
    TtD[1] = T[0] * db + T[3] * dc;
    TtD[2] = T[1] * da + T[4] * db;
    TtD[3] = T[1] * db + T[4] * dc;
    TtD[4] = T[2] * da + T[5] * db;
    TtD[5] = T[2] * db + T[5] * dc;
    d3[0] = TtD[0] * T[0] + TtD[1] * T[3];
    d3[1] = TtD[0] * T[1] + TtD[1] * T[4];
    d3[2] = TtD[0] * T[2] + TtD[1] * T[5];
    d3[3] = TtD[2] * T[1] + TtD[3] * T[4];
    d3[4] = TtD[2] * T[2] + TtD[3] * T[5];
    d3[5] = TtD[4] * T[2] + TtD[5] * T[5];
}

/* backward.cuh:123-153 */
static void dM_from_dcov3d(const float d3[6], const float M[9], float dM[9]) {
    float rows[3][3] = {{d3[0], d3[1], d3[2]}, {d3[1], d3[3], d3[4]}, {d3[2], d3[4], d3[5]}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            dM[i * 3 + j] =
                2.0f * (rows[i][0] * M[0 * 3 + j] + rows[i][1] * M[1 * 3 + j] + rows[i][2] * M[2 * 3 + j]);
}

/* backward.cuh:168-227 */
static void dquat_from_dR(const float rot[4], const float dR[9], float dq[4]) {
    float w = rot[0], x = rot[1], y = rot[2], z = rot[3];
    float inv_norm = cugs_rsqrtf(w * w + x * x + y * y + z * z + 1e-12f);
    w *= inv_norm; x *= inv_norm; y *= inv_norm; z *= inv_norm;
    float dw = 2.0f * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] + -y * dR[6] + x * dR[7]);
    float dx = 2.0f * (y * dR[1] + z * dR[2] + y * dR[3] - 2.0f * x * dR[4] - w * dR[5] +
                       z * dR[6] + w * dR[7] - 2.0f * x * dR[8]);
    float dy = 2.0f * (-2.0f * y * dR[0] + x * dR[1] + w * dR[2] + x * dR[3] + z * dR[5] +
                       -w * dR[6] + z * dR[7] - 2.0f * y * dR[8]);
    float dz = 2.0f * (-2.0f * z * dR[0] - w * dR[1] + x * dR[2] + w * dR[3] -
                       2.0f * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
    float dot = dw * w + dx * x + dy * y + dz * z;
    dq[0] = inv_norm * (dw - w * dot);
    dq[1] = inv_norm * (dx - x * dot);
    dq[2] = inv_norm * (dy - y * dot);
    dq[3] = inv_norm * (dz - z * dot);
}

/* backward.cuh:248-346 (accumulates into dt) */
static void dt_cam_from_cov(const float d2[3], const float cov_3d[6], const float W[9],
                            const float t[3], float fx, float fy, float dt[3]) {
    float tx = t[0], ty = t[1], tz = t[2];
    float tz_inv = 1.0f / (tz + 1e-6f);
    float tz_inv2 = tz_inv * tz_inv;
    float J0 = fx * tz_inv, J2 = -fx * tx * tz_inv2, J4 = fy * tz_inv, J5 = -fy * ty * tz_inv2;
    float T[6];
    T[0] = J0 * W[0] + J2 * W[6];
    T[1] = J0 * W[1] + J2 * W[7];
    T[2] = J0 * W[2] + J2 * W[8];
    T[3] = J4 * W[3] + J5 * W[6];
    T[4] = J4 * W[4] + J5 * W[7];
    T[5] = J4 * W[5] + J5 * W[8];
    float S00 = cov_3d[0], S01 = cov_3d[1], S02 = cov_3d[2];
    float S11 = cov_3d[3], S12 = cov_3d[4], S22 = cov_3d[5];
    float TS[6];
    TS[0] = T[0] * S00 + T[1] * S01 + T[2] * S02;
    TS[1] = T[0] * S01 + T[1] * S11 + T[2] * S12;
    TS[2] = T[0] * S02 + T[1] * S12 + T[2] * S22;
    TS[3] = T[3] * S00 + T[4] * S01 + T[5] * S02;
    TS[4] = T[3] * S01 + T[4] * S11 + T[5] * S12;
    TS[5] = T[3] * S02 + T[4] * S12 + T[5] * S22;
    float da = d2[0], db = d2[1], dc = d2[2];
    float dT[6];
    dT[0] = 2.0f * (da * TS[0] + db * TS[3]);
    dT[1] = 2.0f * (da * TS[1] + db * TS[4]);
    dT[2] = 2.0f * (da * TS[2] + db * TS[5]);
    dT[3] = 2.0f * (db * TS[0] + dc * TS[3]);
    dT[4] = 2.0f * (db * TS[1] + dc * TS[4]);
    dT[5] = 2.0f * (db * TS[2] + dc * TS[5]);
    float dJ[6];
    dJ[0] = dT[0] * W[0] + dT[1] * W[1] + dT[2] * W[2];
    dJ[1] = dT[0] * W[3] + dT[1] * W[4] + dT[2] * W[5];
    dJ[2] = dT[0] * W[6] + dT[1] * W[7] + dT[2] * W[8];
    dJ[3] = dT[3] * W[0] + dT[4] * W[1] + dT[5] * W[2];
    dJ[4] = dT[3] * W[3] + dT[4] * W[4] + dT[5] * W[5];
    dJ[5] = dT[3] * W[6] + dT[4] * W[7] + dT[5] * W[8];
    (void)dJ[1]; (void)dJ[3];
    float tz_inv3 = tz_inv2 * tz_inv;
    dt[0] += dJ[2] * (-fx * tz_inv2);
    dt[1] += dJ[5] * (-fy * tz_inv2);
    dt[2] += dJ[0] * (-fx * tz_inv2) + dJ[2] * (2.0f * fx * tx * tz_inv3) +
             dJ[4] * (-fy * tz_inv2) + dJ[5] * (2.0f * fy * ty * tz_inv3);
}

/* k_project_backward: rasterizer/projection_backward.cu:26-247.
 * Outputs zero-filled first (projection_backward.cu:275-278). */
void orc_project_backward(int n, const float* positions, const float* rotations,
                          const float* scales, const float* opacities, const float* view,
                          float fx, float fy, float cx, float cy, float scale_mod,
                          const int32_t* radii, const float* dL_dmeans_2d,
                          const float* dL_dcov_2d_inv, const float* dL_dopacity_act,
                          float* dL_dpositions, float* dL_drotations, float* dL_dscales,
                          float* dL_dopacities) {
    (void)cx; (void)cy;
    memset(dL_dpositions, 0, sizeof(float) * 3 * (size_t)n);
    memset(dL_drotations, 0, sizeof(float) * 4 * (size_t)n);
    memset(dL_dscales, 0, sizeof(float) * 3 * (size_t)n);
    memset(dL_dopacities, 0, sizeof(float) * (size_t)n);
    float W[9] = {view[0], view[1], view[2], view[4], view[5], view[6], view[8], view[9], view[10]};
    const float log_mod = logf(scale_mod + 1e-8f);
    for (int idx = 0; idx < n; ++idx) {
        if (radii[idx] <= 0) continue;                                   /* :48 */
        float px = positions[idx * 3 + 0], py = positions[idx * 3 + 1], pz = positions[idx * 3 + 2];
        float t_cam[3];
        t_cam[0] = W[0] * px + W[1] * py + W[2] * pz + view[3];
        t_cam[1] = W[3] * px + W[4] * py + W[5] * pz + view[7];
        t_cam[2] = W[6] * px + W[7] * py + W[8] * pz + view[11];
        float log_scale[3] = {scales[idx * 3 + 0] + log_mod, scales[idx * 3 + 1] + log_mod,
                              scales[idx * 3 + 2] + log_mod};
        float rot[4] = {rotations[idx * 4 + 0], rotations[idx * 4 + 1], rotations[idx * 4 + 2],
                        rotations[idx * 4 + 3]};
        float cov_3d[6], M[9], R[9], s[3];
        compute_cov_3d(log_scale, rot, cov_3d, M, R, s);
        float cov2d[3];
        compute_cov_2d(cov_3d, W, t_cam, fx, fy, cov2d);
        float inv[3];
        float det = compute_cov_2d_inverse(cov2d, inv);
        if (det <= 0.0f) continue;                                       /* :91 */

        float tz_inv = 1.0f / (t_cam[2] + 1e-6f);
        float tz_inv2 = tz_inv * tz_inv;
        float J0 = fx * tz_inv, J2 = -fx * t_cam[0] * tz_inv2;
        float J4 = fy * tz_inv, J5 = -fy * t_cam[1] * tz_inv2;
        float T_mat[6];
        T_mat[0] = J0 * W[0] + J2 * W[6];
        T_mat[1] = J0 * W[1] + J2 * W[7];
        T_mat[2] = J0 * W[2] + J2 * W[8];
        T_mat[3] = J4 * W[3] + J5 * W[6];
        T_mat[4] = J4 * W[4] + J5 * W[7];
        T_mat[5] = J4 * W[5] + J5 * W[8];
        float sx = s[0], sy = s[1], sz = s[2];

        float d_inv[3] = {dL_dcov_2d_inv[idx * 3 + 0], dL_dcov_2d_inv[idx * 3 + 1],
                          dL_dcov_2d_inv[idx * 3 + 2]};
        float dmean[2] = {dL_dmeans_2d[idx * 2 + 0], dL_dmeans_2d[idx * 2 + 1]};
        float dopa = dL_dopacity_act[idx];

        float d2[3]; dcov2d_from_dcov2d_inv(inv, d_inv, d2);
        float d3[6]; dcov3d_from_dcov2d(T_mat, d2, d3);
        float dM[9]; dM_from_dcov3d(d3, M, dM);

        float dR[9];
        dR[0] = dM[0] * sx; dR[1] = dM[1] * sy; dR[2] = dM[2] * sz;
        dR[3] = dM[3] * sx; dR[4] = dM[4] * sy; dR[5] = dM[5] * sz;
        dR[6] = dM[6] * sx; dR[7] = dM[7] * sy; dR[8] = dM[8] * sz;
        float ds[3];
        ds[0] = dM[0] * R[0] + dM[3] * R[3] + dM[6] * R[6];
        ds[1] = dM[1] * R[1] + dM[4] * R[4] + dM[7] * R[7];
        ds[2] = dM[2] * R[2] + dM[5] * R[5] + dM[8] * R[8];
        float dlog[3] = {ds[0] * sx, ds[1] * sy, ds[2] * sz};

        float dq[4]; dquat_from_dR(rot, dR, dq);

        float dt[3] = {0.0f, 0.0f, 0.0f};
        dt[0] += dmean[0] * fx * tz_inv;
        dt[1] += dmean[1] * fy * tz_inv;
        dt[2] += dmean[0] * (-fx * t_cam[0] * tz_inv2) + dmean[1] * (-fy * t_cam[1] * tz_inv2);
        dt_cam_from_cov(d2, cov_3d, W, t_cam, fx, fy, dt);

        float dpos[3];
        dpos[0] = W[0] * dt[0] + W[3] * dt[1] + W[6] * dt[2];
        dpos[1] = W[1] * dt[0] + W[4] * dt[1] + W[7] * dt[2];
        dpos[2] = W[2] * dt[0] + W[5] * dt[1] + W[8] * dt[2];

        float sig = cugs_sigmoidf(opacities[idx]);
        float dlogit = dopa * sig * (1.0f - sig);

        dL_dpositions[idx * 3 + 0] = dpos[0]; dL_dpositions[idx * 3 + 1] = dpos[1];
        dL_dpositions[idx * 3 + 2] = dpos[2];
        dL_drotations[idx * 4 + 0] = dq[0]; dL_drotations[idx * 4 + 1] = dq[1];
        dL_drotations[idx * 4 + 2] = dq[2]; dL_drotations[idx * 4 + 3] = dq[3];
        dL_dscales[idx * 3 + 0] = dlog[0]; dL_dscales[idx * 3 + 1] = dlog[1];
        dL_dscales[idx * 3 + 2] = dlog[2];
        dL_dopacities[idx] = dlogit;
    }
}

/* ------------------------------------------------------------------------- */
/* k_fused_adam: optimizer/fused_adam.cu:44-76; bias correction :145-148     */
/* ------------------------------------------------------------------------- */
void orc_adam_bias_correction(float beta1, float beta2, int step, float* bc1, float* bc2) {
    double b1 = (double)beta1, b2 = (double)beta2;
    *bc1 = (float)(1.0 / (1.0 - pow(b1, step)));
    *bc2 = (float)(1.0 / (1.0 - pow(b2, step)));
}

void orc_fused_adam(int64_t n, float* param, const float* grad, float* m, float* v, float lr,
                    float beta1, float beta2, float eps, float bc1, float bc2) {
    for (int64_t i = 0; i < n; ++i) {
        float g = grad[i];
        float mi = beta1 * m[i] + (1.0f - beta1) * g;
        m[i] = mi;
        float vi = beta2 * v[i] + (1.0f - beta2) * g * g;
        v[i] = vi;
        float m_hat = mi * bc1;
        float v_hat = vi * bc2;
        param[i] -= lr * m_hat / (sqrtf(v_hat) + eps);
    }
}

/* training/lr_schedule.hpp:49-57 */
float orc_position_lr(int step, float lr_init, float lr_final, int max_steps) {
    if (step >= max_steps) return lr_final;
    if (step <= 0) return lr_init;
    float t = (float)step / (float)max_steps;
    float log_ratio = logf(lr_final / lr_init);
    return lr_init * expf(t * log_ratio);
}

/* Exposed for tests/test_detmath.py. */
float orc_expf(float x) { return cugs_expf(x); }
void orc_expf_array(int64_t n, const float* x, float* y) {
    for (int64_t i = 0; i < n; ++i) y[i] = cugs_expf(x[i]);
}
void orc_blend_exp_q_array(int64_t n, const float* q, float* y) {
    for (int64_t i = 0; i < n; ++i) y[i] = cugs_blend_exp_q(q[i]);
}

/* ------------------------------------------------------------------------- */
/* cpu_baseline legs for bench.py: projection + SH over N   */
/* Gaussians, single thread or OpenMP over Gaussians.                        */
/* ------------------------------------------------------------------------- */
void orc_project_sh_forward(int n, int degree, int num_coeffs, const float* positions,
                            const float* rotations, const float* scales, const float* opacities,
                            const float* sh, const float* view, const float* cam_center, float fx,
                            float fy, float cx, float cy, int img_w, int img_h, float scale_mod,
                            float* means_2d, float* depths, float* cov_2d_inv, int32_t* radii,
                            int32_t* tiles_touched, float* opacities_act, float* dirs, float* rgb) {
    orc_project_forward(n, positions, rotations, scales, opacities, view, fx, fy, cx, cy, img_w,
                        img_h, scale_mod, means_2d, depths, cov_2d_inv, radii, tiles_touched,
                        opacities_act);
    orc_directions(n, positions, cam_center, dirs);
    orc_sh_forward(degree, n, num_coeffs, sh, dirs, rgb);
    orc_clamp_min0(n * 3, rgb);
}

/* The same, Gaussians split over `nthreads` OpenMP threads (each thread runs the scalar
 * loops above on its own contiguous slice).  Returns the thread count actually used. */
int orc_project_sh_forward_mt(int nthreads, int n, int degree, int num_coeffs,
                              const float* positions, const float* rotations, const float* scales,
                              const float* opacities, const float* sh, const float* view,
                              const float* cam_center, float fx, float fy, float cx, float cy,
                              int img_w, int img_h, float scale_mod, float* means_2d, float* depths,
                              float* cov_2d_inv, int32_t* radii, int32_t* tiles_touched,
                              float* opacities_act, float* dirs, float* rgb) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 1 && n / nthreads < 2048) nthreads = n / 2048 > 1 ? n / 2048 : 1;   /* a thread's share must outweigh its wake-up */
    int chunk = (n + nthreads - 1) / nthreads;
#pragma omp parallel for num_threads(nthreads) schedule(static, 1)
    for (int t = 0; t < nthreads; ++t) {
        int lo = t * chunk;
        int hi = lo + chunk < n ? lo + chunk : n;
        if (lo >= hi) continue;
        int m = hi - lo;
        size_t o = (size_t)lo;
        orc_project_sh_forward(m, degree, num_coeffs, positions + o * 3, rotations + o * 4,
                               scales + o * 3, opacities + o, sh + o * 3 * num_coeffs, view,
                               cam_center, fx, fy, cx, cy, img_w, img_h, scale_mod,
                               means_2d + o * 2, depths + o, cov_2d_inv + o * 3, radii + o,
                               tiles_touched + o, opacities_act + o, dirs + o * 3, rgb + o * 3);
    }
    return nthreads;
}
