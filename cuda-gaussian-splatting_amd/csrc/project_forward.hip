// project_forward.hip — per-Gaussian 3D->2D projection fused with SH colour (SURVEY §8 a3+a4).
//
// Replaces, in ONE launch: k_project_gaussians (rasterizer/projection.cu:55-189), the four libtorch
// passes that build view directions (projection.cu:278-280), k_evaluate_sh (core/sh.cu:19-79),
// clamp_min(0) (projection.cu:284) and the six zero-fills (projection.cu:214-219).
//
// gfx950 mapping: one thread per Gaussian, 256-thread workgroups.  The [n,3,C] SH rows (192 of
// the 236 B/Gaussian at degree 3) would be a 192-byte-strided, fully uncoalesced per-thread read;
// instead the workgroup's contiguous 256*3C-float chunk is streamed with 16-byte coalesced loads
// into LDS (row stride padded to an odd number of dwords: conflict-free ds_read_b32 across the 32
// banks) and each thread then reads its own row from LDS.  Bound: HBM, algorithmic bytes
// 44 + 12C read + 48 (+48 packed) written per Gaussian.
#include "cugs_gaussian_math.h"

namespace {

template <int C>
struct ShTile {
    static constexpr int ROW = 3 * C;
    static constexpr int LROW = (ROW % 2 == 0) ? ROW + 1 : ROW;   // odd dword stride
};

// Stream the workgroup's SH chunk (count rows of ROW floats starting at row `base`) into LDS.
template <int C, bool ALIGNED>
__device__ __forceinline__ void stage_sh_rows(const float* __restrict__ sh, int64_t base, int count,
                                              float* s_sh) {
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
            v[i] = (e4 < TOTAL4) ? cugs_ldnt(src4 + e4) : make_float4(0.f, 0.f, 0.f, 0.f);
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
        for (int e4 = tid; e4 < total4; e4 += CUGS_BLOCK) {
            float4 v = cugs_ldnt(src4 + e4);
            int e = e4 * 4;
            int row = e / ROW, col = e - row * ROW;
            float vals[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s_sh[row * LROW + col] = vals[k];
                if (++col == ROW) { col = 0; ++row; }
            }
        }
        for (int e = (total4 << 2) + tid; e < total; e += CUGS_BLOCK) {
            int row = e / ROW, col = e - row * ROW;
            s_sh[row * LROW + col] = src[e];
        }
    } else {
        for (int e = tid; e < total; e += CUGS_BLOCK) {
            int row = e / ROW, col = e - row * ROW;
            s_sh[row * LROW + col] = src[e];
        }
    }
}

struct ProjPtrs {
    const float* positions; const float* rotations; const float* scales; const float* opacities;
    const float* sh;
    float* means_2d; float* depths; float* cov_2d_inv; int32_t* radii; int32_t* tiles_touched;
    float* opacities_act; float* rgb; float* packed; uint8_t* colour_gate;
    // cugs_project_forward_keyed: the sort's depth keys / tile rectangles / range flag (its N-level workspace), or null
    uint32_t* sort_keys; int4* sort_rect; uint32_t* sort_prect; uint32_t* sort_range_flag;
    uint32_t* sort_zero; uint32_t sort_nzero;          // dwords to clear for the sort (its depth passes' super tables)
};

// PART: 0 = the whole projection (cugs_project_forward[_keyed]); 1 = the GEOMETRY half - everything that does not need
// the SH coefficients: 44 B/Gaussian in, the sort's inputs out (cugs_project_forward_geometry).  The COLOUR half is
// k_project_colour below (cugs_project_forward_colour).  The halves call the same device functions as the whole:
// identical bits.  render() queues the colour half on a side stream underneath the sort, which only needs the
// geometry half's outputs.
template <int C, bool ALIGNED, int PART>
__global__ __launch_bounds__(CUGS_BLOCK) void k_project_forward(int64_t n, int degree, CamArgs cam,
                                                                ProjPtrs p) {
    constexpr int LROW = ShTile<C>::LROW;
    constexpr int OUT_F = 6 * CUGS_BLOCK + CUGS_PACKED_STRIDE * CUGS_BLOCK;      // staged outputs: rgb, cov, packed
    constexpr int SH_F = (PART == 1) ? 0 : CUGS_BLOCK * LROW;                    // the geometry half stages no SH rows
    __shared__ __attribute__((aligned(16))) float s_sh[SH_F > OUT_F ? SH_F : OUT_F];

    const int64_t base = (int64_t)blockIdx.x * CUGS_BLOCK;
    const int count = (int)min((int64_t)CUGS_BLOCK, n - base);
    // housekeeping for the sort this launch keys (cugs_project_forward_keyed): every thread of the grid, live or not
    for (uint32_t z = blockIdx.x * CUGS_BLOCK + threadIdx.x; z < p.sort_nzero; z += gridDim.x * CUGS_BLOCK) p.sort_zero[z] = 0u;
    // this thread's geometry inputs are requested before the SH tile, so that all of a workgroup's reads
    // are in flight together (one HBM latency per workgroup, not one per phase)
    const int64_t idx = base + threadIdx.x;
    const bool live = idx < n;
    const int64_t ld = live ? idx : (n - 1);
    const V3 pos{cugs_ldnt(p.positions + ld * 3 + 0), cugs_ldnt(p.positions + ld * 3 + 1), cugs_ldnt(p.positions + ld * 3 + 2)};
    float in_opa = 0.0f, in_s0 = 0.0f, in_s1 = 0.0f, in_s2 = 0.0f;
    float4 q = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
    {
        in_opa = cugs_ldnt(p.opacities + ld);
        in_s0 = cugs_ldnt(p.scales + ld * 3 + 0); in_s1 = cugs_ldnt(p.scales + ld * 3 + 1); in_s2 = cugs_ldnt(p.scales + ld * 3 + 2);
        q = ALIGNED ? cugs_ldnt(reinterpret_cast<const float4*>(p.rotations) + ld)
                    : make_float4(p.rotations[ld * 4 + 0], p.rotations[ld * 4 + 1], p.rotations[ld * 4 + 2],
                                  p.rotations[ld * 4 + 3]);
    }
    float col[3] = {0.0f, 0.0f, 0.0f};
    if constexpr (PART != 1) {
        stage_sh_rows<C, ALIGNED>(p.sh, base, count, s_sh);
        __syncthreads();
        if (!live) return;

        // --- colour: evaluated for every Gaussian, culled ones included ---
        const V3 dir = view_direction(pos, cam);
        const float* row = s_sh + threadIdx.x * LROW;
        // (raw < 0 ? 0 : raw) rather than fmaxf: clamp_min keeps a NaN, fmaxf would drop it.
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            float raw = sh_colour(degree, row + ch * C, 1, dir);
            col[ch] = (raw < 0.0f) ? 0.0f : raw;
        }
        // the ReLU gate of the SH backward, made here while the coefficients are in LDS: bit ch = the backward's own
        // recomputation of channel ch is > 0 (sh_backward.cu:92-99) - not `col[ch] > 0`, see raw_colour_backward
        if (p.colour_gate) {                                               // kernel-uniform
            float Y[16] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            sh_basis(degree, dir, Y);
            const int num_active = (degree + 1) * (degree + 1);
            unsigned bits = 0u;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch)
                bits |= (raw_colour_backward(row + ch * C, Y, num_active) > 0.0f) ? (1u << ch) : 0u;
            p.colour_gate[idx] = (uint8_t)bits;
        }
    } else {
        if (!live) return;
    }

    // --- geometry ---
    float mx = 0.0f, my = 0.0f, depth = 0.0f, opa = 0.0f;
    Sym2 inv{0.0f, 0.0f, 0.0f};
    int radius = 0, tiles = 0;
    TileRect tr{0, 0, 0, 0};

    const M3 W = view_rotation(cam);
    const V3 t = to_camera(cam, W, pos);
    if (t.z > 0.2f) {                                       // near-plane cull, projection.cu:104
        mx = cam.fx * t.x / t.z + cam.cx;
        my = cam.fy * t.y / t.z + cam.cy;
        depth = t.z;
        opa = cugs_sigmoidf(in_opa);

        const V3 s{cugs_expf(in_s0 + cam.log_mod), cugs_expf(in_s1 + cam.log_mod), cugs_expf(in_s2 + cam.log_mod)};
        const QuatRot qr = rotation_of(q.x, q.y, q.z, q.w);
        const Sym3 S = gram(scale_columns(qr.R, s));
        const Jac J = jacobian(t, cam.fx, cam.fy);
        const Sym2 cov = screen_covariance(project_matrix_full(J, W), S);
        Sym2 inv_c;
        const float det = invert_sym2(cov, inv_c);
        if (det > 0.0f) {                                   // projection.cu:152
            inv = inv_c;
            int r = splat_radius(cov);
            if (r > 0) {                                    // projection.cu:162
                r = min(r, max(cam.width, cam.height));
                radius = r;
                const int ntx = (cam.width + CUGS_TILE - 1) / CUGS_TILE;
                const int nty = (cam.height + CUGS_TILE - 1) / CUGS_TILE;
                tr = tile_rect_of(mx, my, r, cam.width, cam.height, ntx, nty);
                tiles = max((tr.x1 - tr.x0) * (tr.y1 - tr.y0), 0);
            }
        }
    }

    p.depths[idx] = depth;
    p.radii[idx] = radius;
    p.tiles_touched[idx] = tiles;
    p.opacities_act[idx] = opa;
    if (p.sort_keys) {                                                 // kernel-uniform
        // what k_depth_keys_rect (sort.hip) would make of the four stores above, while they are still in registers
        bool bad;
        const SortRecord rec = sort_record_of(depth, tiles, radius, tr, true, &bad);
        if (bad) atomicOr(p.sort_range_flag, 1u);
        p.sort_keys[idx] = rec.key;
        if (p.sort_prect) p.sort_prect[idx] = pack_rect(rec.rect);   // kernel-uniform: images of up to 127 x 127 tiles
        else p.sort_rect[idx] = rec.rect;
    }
    if (ALIGNED && count == CUGS_BLOCK && p.packed) {
        // Full workgroup, 16-byte aligned outputs: the 12-byte-strided rgb / cov rows and the 48-byte packed
        // records are transposed through LDS (the SH tile is dead by now) and leave as contiguous 16-byte
        // stores - a third of the write requests of per-thread strided stores.
        reinterpret_cast<float2*>(p.means_2d)[idx] = make_float2(mx, my);
        if constexpr (PART == 0) __syncthreads();              // every thread has read its SH row
        float* s_rgb = s_sh;
        float* s_cov = s_sh + 3 * CUGS_BLOCK;
        float4* s_pk = reinterpret_cast<float4*>(s_sh + 6 * CUGS_BLOCK);          // 1536 floats in: 16-byte aligned
        const int t = threadIdx.x;
        s_rgb[t * 3 + 0] = col[0]; s_rgb[t * 3 + 1] = col[1]; s_rgb[t * 3 + 2] = col[2];
        s_cov[t * 3 + 0] = inv.a; s_cov[t * 3 + 1] = inv.b; s_cov[t * 3 + 2] = inv.c;
        const float tau = (opa >= (1.0f / 255.0f)) ? logf(255.0f * opa) : -1.0f;   // as write_packed
        s_pk[t * 3 + 0] = make_float4(mx, my, inv.a, inv.b);
        s_pk[t * 3 + 1] = make_float4(inv.c, opa, tau, 0.0f);
        s_pk[t * 3 + 2] = make_float4(col[0], 0.0f, col[1], col[2]);
        __syncthreads();
        if (t < 3 * CUGS_BLOCK / 4) {
            if constexpr (PART == 0) reinterpret_cast<float4*>(p.rgb + base * 3)[t] = reinterpret_cast<const float4*>(s_rgb)[t];
            reinterpret_cast<float4*>(p.cov_2d_inv + base * 3)[t] = reinterpret_cast<const float4*>(s_cov)[t];
        }
        float4* g_pk = reinterpret_cast<float4*>(p.packed + base * CUGS_PACKED_STRIDE);
        if constexpr (PART == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) g_pk[t + k * CUGS_BLOCK] = s_pk[t + k * CUGS_BLOCK];
        } else {
            // geometry half: words 0..7 of every record (the colour half owns words 8..11): of the tile's 768 16-byte
            // chunks the two of three that are geometry, still as contiguous lane-strided stores
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int e = t + k * CUGS_BLOCK;
                if (e % 3 != 2) g_pk[e] = s_pk[e];
            }
        }
        return;
    }
    p.means_2d[idx * 2 + 0] = mx;
    p.means_2d[idx * 2 + 1] = my;
    if constexpr (PART == 0) { p.rgb[idx * 3 + 0] = col[0]; p.rgb[idx * 3 + 1] = col[1]; p.rgb[idx * 3 + 2] = col[2]; }
    p.cov_2d_inv[idx * 3 + 0] = inv.a;
    p.cov_2d_inv[idx * 3 + 1] = inv.b;
    p.cov_2d_inv[idx * 3 + 2] = inv.c;
    if (p.packed) {
        if constexpr (PART == 0) {
            write_packed(p.packed, idx, mx, my, inv, col[0], col[1], col[2], opa);
        } else {
            const float tau = (opa >= (1.0f / 255.0f)) ? logf(255.0f * opa) : -1.0f;   // as write_packed
            float4* rec = reinterpret_cast<float4*>(p.packed + idx * CUGS_PACKED_STRIDE);
            rec[0] = make_float4(mx, my, inv.a, inv.b);
            rec[1] = make_float4(inv.c, opa, tau, 0.0f);
        }
    }
}

// ---- the COLOUR half of the projection (cugs_project_forward_colour): directions + SH + clamp + gate bits ----------
// One workgroup walks tiles of 256 Gaussians, blockIdx.x, blockIdx.x + gridDim.x, ...: the grid is CAPPED
// (colour_grid below) because this kernel is meant to run on a side stream underneath the sort - at three workgroups
// per CU (its 50 KB SH tile) an uncapped grid takes every CU's LDS and the sort's workgroups queue behind it.
template <int C, bool ALIGNED>
__global__ __launch_bounds__(CUGS_BLOCK) void k_project_colour(int64_t n, int degree, CamArgs cam, ProjPtrs p) {
    constexpr int LROW = ShTile<C>::LROW;
    __shared__ __attribute__((aligned(16))) float s_sh[CUGS_BLOCK * LROW];
    const int64_t ntiles = (n + CUGS_BLOCK - 1) / CUGS_BLOCK;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t base = tile * CUGS_BLOCK;
        const int count = (int)min((int64_t)CUGS_BLOCK, n - base);
        const int64_t idx = base + threadIdx.x;
        const bool live = idx < n;
        const int64_t ld = live ? idx : (n - 1);
        const V3 pos{cugs_ldnt(p.positions + ld * 3 + 0), cugs_ldnt(p.positions + ld * 3 + 1), cugs_ldnt(p.positions + ld * 3 + 2)};
        stage_sh_rows<C, ALIGNED>(p.sh, base, count, s_sh);
        __syncthreads();
        float col[3] = {0.0f, 0.0f, 0.0f};
        if (live) {
            // --- colour: evaluated for every Gaussian, culled ones included ---
            const V3 dir = view_direction(pos, cam);
            const float* row = s_sh + threadIdx.x * LROW;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                float raw = sh_colour(degree, row + ch * C, 1, dir);
                col[ch] = (raw < 0.0f) ? 0.0f : raw;                 // clamp_min keeps a NaN, fmaxf would drop it
            }
            if (p.colour_gate) {                                     // kernel-uniform; see k_project_forward
                float Y[16] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                sh_basis(degree, dir, Y);
                const int num_active = (degree + 1) * (degree + 1);
                unsigned bits = 0u;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch)
                    bits |= (raw_colour_backward(row + ch * C, Y, num_active) > 0.0f) ? (1u << ch) : 0u;
                p.colour_gate[idx] = (uint8_t)bits;
            }
            // words 8..11 of this Gaussian's packed record: one 16-byte store
            if (p.packed)
                reinterpret_cast<float4*>(p.packed + idx * CUGS_PACKED_STRIDE)[2] = make_float4(col[0], 0.0f, col[1], col[2]);
        }
        __syncthreads();                                             // every thread has read its SH row
        if (ALIGNED && count == CUGS_BLOCK) {                        // rgb rows through LDS to 16-byte stores
            const int t = threadIdx.x;
            s_sh[t * 3 + 0] = col[0]; s_sh[t * 3 + 1] = col[1]; s_sh[t * 3 + 2] = col[2];
            __syncthreads();
            if (t < 3 * CUGS_BLOCK / 4)
                reinterpret_cast<float4*>(p.rgb + base * 3)[t] = reinterpret_cast<const float4*>(s_sh)[t];
            __syncthreads();                                         // before the next tile's rows land in s_sh
        } else if (live) {
            p.rgb[idx * 3 + 0] = col[0]; p.rgb[idx * 3 + 1] = col[1]; p.rgb[idx * 3 + 2] = col[2];
        }
    }
}

#ifdef CUGS_DEV
int g_dev_colour_cap = 0;                                            // cugsdbg_colour_grid_cap: 0 = the default
#endif
// Workgroups of the colour half: one per CU (of the 256: 110 KB of every CU's LDS and three quarters of its wave slots
// stay free for the sort's kernels it runs beside).
inline int colour_grid(int64_t n) {
    int cap = 256;
#ifdef CUGS_DEV
    if (g_dev_colour_cap > 0) cap = g_dev_colour_cap;
#endif
    const int64_t tiles = (n + CUGS_BLOCK - 1) / CUGS_BLOCK;
    return (int)(tiles < cap ? tiles : cap);
}

template <int C>
int launch_colour(int64_t n, int degree, const CamArgs& cam, const ProjPtrs& p, bool aligned, hipStream_t st) {
    if (aligned)
        hipLaunchKernelGGL((k_project_colour<C, true>), dim3(colour_grid(n)), dim3(CUGS_BLOCK), 0, st, n, degree, cam, p);
    else
        hipLaunchKernelGGL((k_project_colour<C, false>), dim3(colour_grid(n)), dim3(CUGS_BLOCK), 0, st, n, degree, cam, p);
    CUGS_LAUNCH_CHECK();
    return 0;
}

// ---- standalone SH forward (evaluate_sh_cuda, core/sh.cu:81-123): unclamped ----
template <int C, bool ALIGNED>
__global__ __launch_bounds__(CUGS_BLOCK) void k_sh_forward(int64_t n, int degree,
                                                           const float* __restrict__ sh,
                                                           const float* __restrict__ dirs,
                                                           float* __restrict__ out) {
    constexpr int LROW = ShTile<C>::LROW;
    __shared__ float s_sh[CUGS_BLOCK * LROW];
    const int64_t base = (int64_t)blockIdx.x * CUGS_BLOCK;
    const int count = (int)min((int64_t)CUGS_BLOCK, n - base);
    stage_sh_rows<C, ALIGNED>(sh, base, count, s_sh);
    __syncthreads();
    const int64_t idx = base + threadIdx.x;
    if (idx >= n) return;
    const V3 d{dirs[idx * 3 + 0], dirs[idx * 3 + 1], dirs[idx * 3 + 2]};
    const float* row = s_sh + threadIdx.x * LROW;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) out[idx * 3 + ch] = sh_colour(degree, row + ch * C, 1, d);
}

// Generic-C fallbacks (C not in {1,4,9,16}): one thread per Gaussian straight from global.
__global__ __launch_bounds__(CUGS_BLOCK) void k_sh_forward_generic(int64_t n, int degree, int C,
                                                                   const float* __restrict__ sh,
                                                                   const float* __restrict__ dirs,
                                                                   float* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * CUGS_BLOCK + threadIdx.x;
    if (idx >= n) return;
    const V3 d{dirs[idx * 3 + 0], dirs[idx * 3 + 1], dirs[idx * 3 + 2]};
    for (int ch = 0; ch < 3; ++ch)
        out[idx * 3 + ch] = sh_colour(degree, sh + idx * 3 * C + (int64_t)ch * C, 1, d);
}

__global__ __launch_bounds__(CUGS_BLOCK) void k_pack_projected(int64_t n,
                                                               const float* __restrict__ means_2d,
                                                               const float* __restrict__ cov_2d_inv,
                                                               const float* __restrict__ rgb,
                                                               const float* __restrict__ opa,
                                                               float* __restrict__ packed) {
    const int64_t idx = (int64_t)blockIdx.x * CUGS_BLOCK + threadIdx.x;
    if (idx >= n) return;
    Sym2 inv{cov_2d_inv[idx * 3 + 0], cov_2d_inv[idx * 3 + 1], cov_2d_inv[idx * 3 + 2]};
    write_packed(packed, idx, means_2d[idx * 2 + 0], means_2d[idx * 2 + 1], inv, rgb[idx * 3 + 0],
                 rgb[idx * 3 + 1], rgb[idx * 3 + 2], opa[idx]);
}

inline int grid_for(int64_t n) { return (int)((n + CUGS_BLOCK - 1) / CUGS_BLOCK); }

template <int C, int PART = 0>
int launch_project(int64_t n, int degree, const CamArgs& cam, const ProjPtrs& p, bool aligned,
                   hipStream_t st) {
    if (aligned)
        hipLaunchKernelGGL((k_project_forward<C, true, PART>), dim3(grid_for(n)), dim3(CUGS_BLOCK), 0, st, n,
                           degree, cam, p);
    else
        hipLaunchKernelGGL((k_project_forward<C, false, PART>), dim3(grid_for(n)), dim3(CUGS_BLOCK), 0, st, n,
                           degree, cam, p);
    CUGS_LAUNCH_CHECK();
    return 0;
}

template <int C>
int launch_sh_forward(int64_t n, int degree, const float* sh, const float* dirs, float* out,
                      bool aligned, hipStream_t st) {
    if (aligned)
        hipLaunchKernelGGL((k_sh_forward<C, true>), dim3(grid_for(n)), dim3(CUGS_BLOCK), 0, st, n,
                           degree, sh, dirs, out);
    else
        hipLaunchKernelGGL((k_sh_forward<C, false>), dim3(grid_for(n)), dim3(CUGS_BLOCK), 0, st, n,
                           degree, sh, dirs, out);
    CUGS_LAUNCH_CHECK();
    return 0;
}

}  // namespace

namespace {
int project_forward_impl(int64_t n, int num_coeffs, int active_degree,
                                    const float* positions, const float* rotations,
                                    const float* scales, const float* opacities,
                                    const float* sh_coeffs, const cugs_camera* camera_host,
                                    float scale_modifier, float* means_2d, float* depths,
                                    float* cov_2d_inv, int32_t* radii, int32_t* tiles_touched,
                                    float* opacities_act, float* rgb, float* packed, uint8_t* colour_gate,
                                    void* sort_workspace, size_t sort_workspace_bytes, void* stream) {
    if (n < 0 || !camera_host) return CUGS_EINVAL;
    if (active_degree < 0 || active_degree > 3) return CUGS_EINVAL;
    if ((active_degree + 1) * (active_degree + 1) > num_coeffs) return CUGS_EINVAL;
    if (num_coeffs != 1 && num_coeffs != 4 && num_coeffs != 9 && num_coeffs != 16) return CUGS_EINVAL;
    if (n == 0) return 0;
    if (!positions || !rotations || !scales || !opacities || !sh_coeffs || !means_2d || !depths ||
        !cov_2d_inv || !radii || !tiles_touched || !opacities_act || !rgb)
        return CUGS_EINVAL;
    if (packed && !cugs_aligned16(packed)) return CUGS_EALIGN;
    if (n > (int64_t)2147483647) return CUGS_EOVERFLOW;

    const CamArgs cam = cugs_make_cam_args(camera_host, scale_modifier);
    ProjPtrs p{positions, rotations, scales, opacities, sh_coeffs, means_2d, depths, cov_2d_inv,
               radii, tiles_touched, opacities_act, rgb, packed, colour_gate, nullptr, nullptr, nullptr, nullptr, nullptr, 0u};
    if (sort_workspace) {
        int rc = cugs_sort_key_slots(sort_workspace, sort_workspace_bytes, n, camera_host->width, camera_host->height,
                                     &p.sort_keys, &p.sort_rect, &p.sort_prect, &p.sort_range_flag, &p.sort_zero, &p.sort_nzero);
        if (rc) return rc;
    }
    const bool aligned = cugs_aligned16(sh_coeffs) && cugs_aligned16(rotations) && cugs_aligned16(rgb) &&
                         cugs_aligned16(cov_2d_inv) && (reinterpret_cast<uintptr_t>(means_2d) & 7u) == 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (num_coeffs) {
        case 1: return launch_project<1>(n, active_degree, cam, p, aligned, st);
        case 4: return launch_project<4>(n, active_degree, cam, p, aligned, st);
        case 9: return launch_project<9>(n, active_degree, cam, p, aligned, st);
        default: return launch_project<16>(n, active_degree, cam, p, aligned, st);
    }
}
}  // namespace

extern "C" int cugs_project_forward(int64_t n, int num_coeffs, int active_degree,
                                    const float* positions, const float* rotations,
                                    const float* scales, const float* opacities,
                                    const float* sh_coeffs, const cugs_camera* camera_host,
                                    float scale_modifier, float* means_2d, float* depths,
                                    float* cov_2d_inv, int32_t* radii, int32_t* tiles_touched,
                                    float* opacities_act, float* rgb, float* packed, uint8_t* colour_gate,
                                    void* stream) {
    return project_forward_impl(n, num_coeffs, active_degree, positions, rotations, scales, opacities, sh_coeffs,
                                camera_host, scale_modifier, means_2d, depths, cov_2d_inv, radii, tiles_touched,
                                opacities_act, rgb, packed, colour_gate, nullptr, 0, stream);
}

extern "C" int cugs_project_forward_keyed(int64_t n, int num_coeffs, int active_degree,
                                          const float* positions, const float* rotations,
                                          const float* scales, const float* opacities,
                                          const float* sh_coeffs, const cugs_camera* camera_host,
                                          float scale_modifier, float* means_2d, float* depths,
                                          float* cov_2d_inv, int32_t* radii, int32_t* tiles_touched,
                                          float* opacities_act, float* rgb, float* packed, uint8_t* colour_gate,
                                          void* sort_workspace, size_t sort_workspace_bytes, void* stream) {
    if (!sort_workspace) return CUGS_EINVAL;
    return project_forward_impl(n, num_coeffs, active_degree, positions, rotations, scales, opacities, sh_coeffs,
                                camera_host, scale_modifier, means_2d, depths, cov_2d_inv, radii, tiles_touched,
                                opacities_act, rgb, packed, colour_gate, sort_workspace, sort_workspace_bytes, stream);
}

// ---- the projection in two launches (render(): the colour half travels on a side stream underneath the sort) -------
extern "C" int cugs_project_forward_geometry(int64_t n, const float* positions, const float* rotations,
                                             const float* scales, const float* opacities,
                                             const cugs_camera* camera_host, float scale_modifier, float* means_2d,
                                             float* depths, float* cov_2d_inv, int32_t* radii, int32_t* tiles_touched,
                                             float* opacities_act, float* packed, void* sort_workspace,
                                             size_t sort_workspace_bytes, void* stream) {
    if (n < 0 || !camera_host) return CUGS_EINVAL;
    if (n == 0) return 0;
    if (!positions || !rotations || !scales || !opacities || !means_2d || !depths || !cov_2d_inv || !radii ||
        !tiles_touched || !opacities_act)
        return CUGS_EINVAL;
    if (packed && !cugs_aligned16(packed)) return CUGS_EALIGN;
    if (n > (int64_t)2147483647) return CUGS_EOVERFLOW;
    const CamArgs cam = cugs_make_cam_args(camera_host, scale_modifier);
    ProjPtrs p{positions, rotations, scales, opacities, nullptr, means_2d, depths, cov_2d_inv, radii, tiles_touched,
               opacities_act, nullptr, packed, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0u};
    if (sort_workspace) {
        int rc = cugs_sort_key_slots(sort_workspace, sort_workspace_bytes, n, camera_host->width, camera_host->height,
                                     &p.sort_keys, &p.sort_rect, &p.sort_prect, &p.sort_range_flag, &p.sort_zero, &p.sort_nzero);
        if (rc) return rc;
    }
    const bool aligned = cugs_aligned16(rotations) && cugs_aligned16(cov_2d_inv) &&
                         (reinterpret_cast<uintptr_t>(means_2d) & 7u) == 0;
    return launch_project<1, 1>(n, 0, cam, p, aligned, static_cast<hipStream_t>(stream));     // C is unused by this half
}

extern "C" int cugs_project_forward_colour(int64_t n, int num_coeffs, int active_degree, const float* positions,
                                           const float* sh_coeffs, const cugs_camera* camera_host, float* rgb,
                                           float* packed, uint8_t* colour_gate, void* stream) {
    if (n < 0 || !camera_host) return CUGS_EINVAL;
    if (active_degree < 0 || active_degree > 3) return CUGS_EINVAL;
    if ((active_degree + 1) * (active_degree + 1) > num_coeffs) return CUGS_EINVAL;
    if (num_coeffs != 1 && num_coeffs != 4 && num_coeffs != 9 && num_coeffs != 16) return CUGS_EINVAL;
    if (n == 0) return 0;
    if (!positions || !sh_coeffs || !rgb) return CUGS_EINVAL;
    if (packed && !cugs_aligned16(packed)) return CUGS_EALIGN;
    if (n > (int64_t)2147483647) return CUGS_EOVERFLOW;
    const CamArgs cam = cugs_make_cam_args(camera_host, 1.0f);                                 // the colour needs no scale
    ProjPtrs p{positions, nullptr, nullptr, nullptr, sh_coeffs, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, rgb,
               packed, colour_gate, nullptr, nullptr, nullptr, nullptr, nullptr, 0u};
    const bool aligned = cugs_aligned16(sh_coeffs) && cugs_aligned16(rgb);
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (num_coeffs) {
        case 1: return launch_colour<1>(n, active_degree, cam, p, aligned, st);
        case 4: return launch_colour<4>(n, active_degree, cam, p, aligned, st);
        case 9: return launch_colour<9>(n, active_degree, cam, p, aligned, st);
        default: return launch_colour<16>(n, active_degree, cam, p, aligned, st);
    }
}

#ifdef CUGS_DEV
extern "C" int cugsdbg_colour_grid_cap(int cap) { g_dev_colour_cap = cap; return 0; }
#endif

extern "C" int cugs_evaluate_sh(int degree, int64_t n, int num_coeffs, const float* sh_coeffs,
                                const float* directions, float* out_rgb, void* stream) {
    // Input validation of evaluate_sh_cuda (core/sh.cu:84-97)
    if (degree < 0 || degree > 3 || n < 0) return CUGS_EINVAL;
    if ((degree + 1) * (degree + 1) > num_coeffs) return CUGS_EINVAL;
    if (n == 0) return 0;
    if (!sh_coeffs || !directions || !out_rgb) return CUGS_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool aligned = cugs_aligned16(sh_coeffs);
    switch (num_coeffs) {
        case 1: return launch_sh_forward<1>(n, degree, sh_coeffs, directions, out_rgb, aligned, st);
        case 4: return launch_sh_forward<4>(n, degree, sh_coeffs, directions, out_rgb, aligned, st);
        case 9: return launch_sh_forward<9>(n, degree, sh_coeffs, directions, out_rgb, aligned, st);
        case 16: return launch_sh_forward<16>(n, degree, sh_coeffs, directions, out_rgb, aligned, st);
        default:
            hipLaunchKernelGGL(k_sh_forward_generic, dim3(grid_for(n)), dim3(CUGS_BLOCK), 0, st, n,
                               degree, num_coeffs, sh_coeffs, directions, out_rgb);
            CUGS_LAUNCH_CHECK();
            return 0;
    }
}

extern "C" int cugs_pack_projected(int64_t n, const float* means_2d, const float* cov_2d_inv,
                                   const float* rgb, const float* opacities_act, float* packed,
                                   void* stream) {
    if (n < 0) return CUGS_EINVAL;
    if (n == 0) return 0;
    if (!means_2d || !cov_2d_inv || !rgb || !opacities_act || !packed) return CUGS_EINVAL;
    if (!cugs_aligned16(packed)) return CUGS_EALIGN;
    hipLaunchKernelGGL(k_pack_projected, dim3(grid_for(n)), dim3(CUGS_BLOCK), 0,
                       static_cast<hipStream_t>(stream), n, means_2d, cov_2d_inv, rgb, opacities_act,
                       packed);
    CUGS_LAUNCH_CHECK();
    return 0;
}
