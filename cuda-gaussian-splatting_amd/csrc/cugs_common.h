// cugs_common.h — shared declarations for the gfx950 kernels behind include/cugs_hip.h.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see csrc/Makefile).  Contraction
// is OFF for the whole library: the only fused multiply-adds are explicit fmaf calls, placed
// where DESIGN.md's "FMA placement contract" says, so that integer outputs and blend decisions
// are bit-identical to the CPU oracle.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cugs_hip.h"
#include "../../include/cugs_detmath.h"

#ifndef CUGS_BLOCK
#define CUGS_BLOCK 256     /* threads per workgroup; a translation unit may be built with another value (A/B builds) */
#endif
#define CUGS_WAVE 64

// Launch-error convention of the reference (CUDA_CHECK(cudaGetLastError()), projection.cu:267):
// report launch errors only, never synchronise.
#define CUGS_RETURN_IF_HIP(expr)                    \
    do {                                            \
        hipError_t _e = (expr);                     \
        if (_e != hipSuccess) return (int)_e;       \
    } while (0)

#define CUGS_LAUNCH_CHECK() CUGS_RETURN_IF_HIP(hipGetLastError())

static inline bool cugs_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Streaming accesses: data that is read or written once per launch and not touched again before a few hundred MB
// of other traffic have passed (the 192 B/Gaussian SH rows above all) go through the non-temporal path
// (global_load/store ... nt): they neither evict the lines the blend kernels are about to gather nor wait behind
// them - k_project_forward 0.083 -> 0.061 ms with nothing but its SH tile loads marked.
typedef float cugs_v4f __attribute__((ext_vector_type(4)));
typedef float cugs_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 cugs_ldnt(const float4* p) {
    const cugs_v4f t = __builtin_nontemporal_load(reinterpret_cast<const cugs_v4f*>(p));
    return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ float cugs_ldnt(const float* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void cugs_stnt(float4* p, float4 v) {
    const cugs_v4f t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<cugs_v4f*>(p));
}
__device__ __forceinline__ void cugs_stnt(float* p, float v) { __builtin_nontemporal_store(v, p); }

// Camera as kernel argument (passed by value: no per-call H2D copy, unlike projection.cu:236,274).
struct CamArgs {
    float view[16];
    float fx, fy, cx, cy;
    int width, height;
    float cc[3];
    float log_mod;      // logf(scale_modifier + 1e-8f), projection.cu:127
};

static inline CamArgs cugs_make_cam_args(const cugs_camera* c, float scale_modifier) {
    CamArgs a;
    for (int i = 0; i < 16; ++i) a.view[i] = c->view[i];
    a.fx = c->fx; a.fy = c->fy; a.cx = c->cx; a.cy = c->cy;
    a.width = c->width; a.height = c->height;
    a.cc[0] = c->cam_center[0]; a.cc[1] = c->cam_center[1]; a.cc[2] = c->cam_center[2];
    a.log_mod = logf(scale_modifier + 1e-8f);
    return a;
}

// Internal (not part of the C ABI): where the sort's N-level workspace keeps the per-Gaussian depth keys, tile
// rectangles and the range flag of its three-pass depth ordering (sort.hip: SortWsN), for the projection kernel that
// fills them in passing (cugs_project_forward_keyed).  CUGS_EWORKSPACE if `bytes` is too small for n Gaussians.
// *zero / *nzero: dwords the key kernel must clear for the sort (the depth passes' super-block tables, sort.hip).
// Exactly one of *rect / *prect comes back non-null: prect (one packed dword per Gaussian, cugs_gaussian_math.h:
// pack_rect) for images of up to 127 x 127 tiles, the 16-byte records otherwise.
int cugs_sort_key_slots(void* workspace, size_t bytes, int64_t n, int width, int height, uint32_t** keys, int4** rect,
                        uint32_t** prect, uint32_t** range_flag, uint32_t** zero, uint32_t* nzero)
    __attribute__((visibility("hidden")));

// XCD-aware bijective block remap (cdna_hip_programming.md T1): blocks b and b+8 share an XCD,
// so give each XCD a contiguous run of work items (neighbouring tiles share Gaussians -> L2 hits).
__device__ __forceinline__ unsigned cugs_xcd_remap(unsigned bid, unsigned nwg) {
    unsigned xcd = bid & 7u, q = nwg >> 3, r = nwg & 7u;
    unsigned base = (xcd < r) ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return base + (bid >> 3);
}

// Tile of workgroup `bid` in the blend kernels: the XCD remap above gives each XCD a contiguous run of a LINEAR order of
// the tiles - and that order walks the tile rows in UNITS of CUGS_ROW_GROUP rows interleaved by eight (units 0, 8, 16, ...
// then 1, 9, 17, ...), so an XCD's run is every eighth unit of the image rather than a band of nty / 8 adjacent rows.
// With bands, a scene whose splats cluster in one part of the screen (every real capture) loads the XCDs that own those
// rows and idles the others.  Same-box A/B (profiles/r03_f_row_interleave_ab.log; ms per fwd+bwd step: uniform config 3 |
// 80 % of the splats on 10 % of the screen | 50 % on 2 %):  bands 0.892 | 0.976 | 1.115;  units of 1 row 0.900 | 0.944 |
// 0.975;  2 rows 0.893 | 0.940 | 1.077;  4 rows 0.895 | 0.953 | 1.048.  Single rows balance best but cost the uniform
// scene 1 % (vertical neighbours share Gaussians and no longer share an L2); pairs of rows cost it nothing.  Horizontal
// neighbours run back to back on one XCD either way; what crosses XCDs meets in the memory-side cache, which holds the
// whole 48 MB record table.  A bijection on [0, ntx * nty) for every image size (tests/test_tile_order.py).
#ifndef CUGS_ROW_GROUP
#define CUGS_ROW_GROUP 2      /* tile rows per interleave unit (0: no interleave - bands, the round-2 order) */
#endif
__device__ __forceinline__ unsigned cugs_blend_tile(unsigned bid, unsigned ntx, unsigned nty) {
    const unsigned lin = cugs_xcd_remap(bid, ntx * nty);
#if CUGS_ROW_GROUP == 0
    return lin;
#else
    constexpr unsigned G = CUGS_ROW_GROUP;
    unsigned row = lin / ntx;                       // position in the interleaved row order
    const unsigned col = lin - row * ntx;
    // units of G rows; the unit order is (0, 8, 16, ..., 1, 9, 17, ..., 7, 15, ...) restricted to units < nunits:
    // residue class c holds ceil((nunits - c) / 8) units; the last unit of the image may be short
    const unsigned nunits = (nty + G - 1u) / G;
    unsigned c = 0, urow = row;                     // urow: row position counted in rows, consumed class by class
#pragma unroll
    for (unsigned k = 0; k < 7u; ++k) {
        const unsigned units = (nunits + 7u - c) >> 3;                          // units with residue c
        // rows in class c: G per unit, minus what the image's last (short) unit lacks if it is in this class
        const unsigned last_in = (units > 0u && ((nunits - 1u) & 7u) == c) ? 1u : 0u;
        const unsigned rows_c = units * G - last_in * (nunits * G - nty);
        if (urow >= rows_c) { urow -= rows_c; ++c; }
    }
    const unsigned unit = c + 8u * (urow / G);
    return (unit * G + urow % G) * ntx + col;
#endif
}
