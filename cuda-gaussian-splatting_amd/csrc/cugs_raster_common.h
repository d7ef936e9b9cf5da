// cugs_raster_common.h — pieces shared by the forward and backward blend kernels.
//
// Work decomposition (gfx950): one 256-thread workgroup per 16x16 tile = 4 wave64s, each wave
// owning one 8x8 pixel QUAD of the tile (lane -> (lane&7, lane>>3)).  The tile's depth-sorted
// Gaussian list is consumed in batches of 256 records staged in LDS (one record per thread,
// three 16-byte gathers from the packed table).  Within a batch every wave first tests 64
// records at a time, one record per lane, against its own quad (`may_touch_quad`, below), turns
// the result into a 64-bit ballot mask, and then walks only the set bits: the per-pixel
// evaluation runs for the (wave, Gaussian) pairs that can matter instead of all P x 256.
//
// The cull is CONSERVATIVE and therefore invisible in the results: a record is dropped for a wave
// only when alpha < 1/255 is certain for all 64 pixel centres, with a margin that covers the
// fp32 rounding of the per-pixel power (see the derivation at may_touch_quad).  The per-pixel
// path itself (pixel_alpha) follows DESIGN.md's FMA placement contract and cugs_detmath.h, so
// every skip / clamp / termination decision equals the oracle's bit for bit.
#pragma once

#include "cugs_gaussian_math.h"

#define CUGS_REC_F4 3      // float4s per LDS record (CUGS_PACKED_STRIDE / 4)

struct RasterGeom {
    int width, height, ntx, ntiles;
    float bg0, bg1, bg2;
};

struct RasterSrc {
    const int32_t* tile_ranges;
    const int32_t* gidx;
    const float* packed;        // may be NULL -> gather from the four arrays below
    const float* means_2d;
    const float* cov_2d_inv;
    const float* rgb;
    const float* opa;
};

// Stage list entry `li` (if < end) into LDS slot threadIdx.x.  The LDS copy of the record carries the
// Gaussian index (as bits) in word 10 in place of 1/a, so that the backward's scatter address comes
// with the same 16-byte read as the opacity.  Returns the Gaussian index (or -1).
template <bool PACKED>
__device__ __forceinline__ int stage_record(const RasterSrc& s, int li, int end, float4* s_rec) {
    if (li >= end) return -1;
    const int g = s.gidx[li];
    float4 r0, r1, r2;
    if (PACKED) {
        const float4* src = reinterpret_cast<const float4*>(s.packed + (int64_t)g * CUGS_PACKED_STRIDE);
        r0 = src[0]; r1 = src[1]; r2 = src[2];
    } else {
        const float a = s.cov_2d_inv[g * 3 + 0], c = s.cov_2d_inv[g * 3 + 2], o = s.opa[g];
        r0 = make_float4(s.means_2d[g * 2 + 0], s.means_2d[g * 2 + 1], a, s.cov_2d_inv[g * 3 + 1]);
        r1 = make_float4(c, s.rgb[g * 3 + 0], s.rgb[g * 3 + 1], s.rgb[g * 3 + 2]);
        r2 = make_float4(o, (o >= (1.0f / 255.0f)) ? logf(255.0f * o) : -1.0f, 0.0f, 0.0f);
    }
    r2.z = __int_as_float(g);
    s_rec[threadIdx.x * CUGS_REC_F4 + 0] = r0;
    s_rec[threadIdx.x * CUGS_REC_F4 + 1] = r1;
    s_rec[threadIdx.x * CUGS_REC_F4 + 2] = r2;
    return g;
}

// Can the Gaussian in (r0, r1, r2) reach alpha >= 1/255 at ANY pixel centre of the rectangle of
// centres [qx0, qx0+wx] x [qy0, qy0+wy] (the wave's quad, or the bounding box of its pixels that are
// still active: ActiveRect below)?  `false` only when certainly not.
//
// alpha = min(0.99, o * exp(power)) >= 1/255 needs -power <= tau := ln(255 o)  (tau < 0: never;
// the record stores tau = -1 for o < 1/255, where o * exp(power <= 0) <= o < 1/255 exactly).
// -power = q/2 with q(d) = a dx^2 + 2 b dx dy + c dy^2, d = centre - mean.  Over the rectangle of
// quad offsets the minimum of q is 0 if the mean lies inside, else it is attained on an edge;
// along each edge q is a 1-D convex parabola (a, c > 0), minimised at the clamped vertex.
// Rounding: the oracle's fp32 power differs from the exact one by at most ~4 ulp of
// B = |a| X^2 + 2 |b| X Y + |c| Y^2 (X, Y the largest |offset|); q_min here carries a similar
// error; ocml logf ~2 ulp; detexp 1 ulp.  The slack 0.01 tau + 0.05 + 4e-6 B dominates all of it
// by orders of magnitude.  Any NaN makes the final comparison false -> not culled.
__device__ __forceinline__ bool may_touch_quad(float4 r0, float4 r1, float4 r2, float qx0, float qy0,
                                               float wx, float wy) {
    const float mx = r0.x, my = r0.y, a = r0.z, b = r0.w, c = r1.x;
    const float tau = r2.y;
    if (!(tau >= 0.0f)) return false;
    if (!(a > 0.0f && c > 0.0f)) return true;
    const float ia = __builtin_amdgcn_rcpf(a), ic = __builtin_amdgcn_rcpf(c);   // cull only: 1 ulp is irrelevant
    const float lx = qx0 - mx, hx = lx + wx, ly = qy0 - my, hy = ly + wy;
    const bool inside = (lx <= 0.0f) && (hx >= 0.0f) && (ly <= 0.0f) && (hy >= 0.0f);
    const float X = fmaxf(fabsf(lx), fabsf(hx)), Y = fmaxf(fabsf(ly), fabsf(hy));
    const float B = a * X * X + 2.0f * fabsf(b) * X * Y + c * Y * Y;
    const float b2 = b + b;
    // q restricted to the four edges, each at its clamped 1-D minimiser
    const float y1 = fminf(fmaxf(-b * lx * ic, ly), hy);
    const float y2 = fminf(fmaxf(-b * hx * ic, ly), hy);
    const float x3 = fminf(fmaxf(-b * ly * ia, lx), hx);
    const float x4 = fminf(fmaxf(-b * hy * ia, lx), hx);
    const float q1 = a * lx * lx + b2 * lx * y1 + c * y1 * y1;
    const float q2 = a * hx * hx + b2 * hx * y2 + c * y2 * y2;
    const float q3 = a * x3 * x3 + b2 * x3 * ly + c * ly * ly;
    const float q4 = a * x4 * x4 + b2 * x4 * hy + c * hy * hy;
    const float qmin = inside ? 0.0f : fminf(fminf(q1, q2), fminf(q3, q4));
    return !(0.5f * qmin > tau * 1.01f + 0.05f + 4e-6f * B);
}

// Bounding box, in quad coordinates 0..7, of the lanes set in `active` (lane = y*8 + x; active != 0).
// Pure scalar bit arithmetic on the wave-uniform ballot.  Pixels that are finished (or outside the
// image) can no longer be affected by any Gaussian, so culling against the box of the remaining
// ones stays conservative and prunes the tail where only a few pixels of a quad are still open.
struct ActiveRect { float x0, y0, wx, wy; };
__device__ __forceinline__ ActiveRect active_rect(unsigned long long active, float quad_cx, float quad_cy) {
    unsigned cols = (unsigned)active | (unsigned)(active >> 32);
    cols |= cols >> 16;
    cols |= cols >> 8;
    cols &= 0xFFu;
    const int x0 = __builtin_ctz(cols), x1 = 31 - __builtin_clz(cols);
    const int y0 = __builtin_ctzll(active) >> 3, y1 = (63 - __builtin_clzll(active)) >> 3;
    return ActiveRect{quad_cx + (float)x0, quad_cy + (float)y0, (float)(x1 - x0), (float)(y1 - y0)};
}

// forward.cu:124-141 / backward.cu:123-137 for one pixel.  FMA placement contract:
//   u = fma(a,dx,b*dy); v = fma(b,dx,c*dy); q = fma(dx,u,dy*v); power = -0.5f*q.
// power < -5.6 can be dropped before the exponential: exp(-5.6)(1+2^-22) = 0.0036979 and
// opacity <= 1, so alpha < 1/255 = 0.0039216 is certain (the oracle reaches the same skip
// through its alpha test).  Returns false when the Gaussian is skipped at this pixel.
struct PixelEval { float dx, dy, power, e, alpha; };
__device__ __forceinline__ bool pixel_alpha(float pxf, float pyf, float mx, float my, float a, float b,
                                            float c, float o, PixelEval& r) {
    r.dx = pxf - mx;
    r.dy = pyf - my;
    const float u = fmaf(a, r.dx, b * r.dy);
    const float v = fmaf(b, r.dx, c * r.dy);
    const float q = fmaf(r.dx, u, r.dy * v);
    r.power = -0.5f * q;
    if (r.power > 0.0f || r.power < -5.6f) return false;
    r.e = cugs_expf_core(r.power);
    r.alpha = fminf(o * r.e, 0.99f);
    return !(r.alpha < (1.0f / 255.0f));
}

// ---------------------------------------------------------------------------------------
// reduce9: sums NINE per-lane values across the 64 lanes of a wave and leaves each total in a
// different lane (27 VALU-class operations instead of 9 x 6 DPP adds + a 9-way select).
//
// Idea ("transpose-reduce"): at every halving step two partner lanes exchange the half of the
// values they will not keep, so the number of live values per lane halves as the number of lanes
// sharing a sum doubles.  gfx950 has single instructions for the two cross-row exchanges:
//   v_permlane32_swap D,S : swaps D[32..63] with S[0..31]         -> D+S = {sum32(D) | sum32(S)}
//   v_permlane16_swap D,S : swaps D.row1<->S.row0, D.row3<->S.row2 -> D+S = rows {D, S, D, S}
// and the in-row steps use DPP row_mirror / row_half_mirror / quad_perm with a select.
// All 64 lanes must be active (EXEC full): callers run it in wave-uniform control flow with
// zeros in lanes that have nothing to add.
//
// Result: after reduce9(v0..v8) the return value of lane L holds the wave total of slot
// reduce9_slot(L) (or is meaningless when that is -1):
//   row r = L >> 4, i = L & 15:  i == 0 -> {v0, v2, v1, v3}[r];  i == 8 -> {v4, v6, v5, v7}[r];
//   i == 4 and r == 0 -> v8.
// ---------------------------------------------------------------------------------------
// The swaps are issued through inline asm: with ROCm 7.2's hipcc the two results of
// __builtin_amdgcn_permlane{16,32}_swap collapse into one register when both feed the same add
// (observed: `v_permlane32_swap v2, v3 ; v_add_f32 v2, v2, v2`).  hipcc pads nothing inside an asm
// statement, so the 2 wait states it otherwise inserts between a VALU write and a swap that reads
// the register (s_nop 1) are part of the string; each stage's swaps share one block.
typedef unsigned cugs_u2 __attribute__((ext_vector_type(2)));
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}

__device__ __forceinline__ int reduce9_slot(int lane) {
    const int r = lane >> 4, i = lane & 15;
    if (i == 0) return (r == 0) ? 0 : (r == 1) ? 2 : (r == 2) ? 1 : 3;
    if (i == 8) return (r == 0) ? 4 : (r == 1) ? 6 : (r == 2) ? 5 : 7;
    if (i == 4 && r == 0) return 8;
    return -1;
}

__device__ __forceinline__ float reduce9(float v0, float v1, float v2, float v3, float v4, float v5, float v6,
                                         float v7, float v8, int lane) {
    // halves: lanes 0..31 keep the first of each pair, lanes 32..63 the second
    float z8 = 0.0f;
    asm volatile("s_nop 1\n\t"
                 "v_permlane32_swap_b32 %0, %1\n\t"
                 "v_permlane32_swap_b32 %2, %3\n\t"
                 "v_permlane32_swap_b32 %4, %5\n\t"
                 "v_permlane32_swap_b32 %6, %7\n\t"
                 "v_permlane32_swap_b32 %8, %9"
                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(v8), "+v"(z8));
    float x0 = v0 + v1, x1 = v2 + v3, x2 = v4 + v5, x3 = v6 + v7, x4 = v8 + z8;
    // rows: even rows keep the first, odd rows the second  -> rows {v0,v2,v1,v3}, {v4,v6,v5,v7}, {v8,0,0,0}
    float z4 = 0.0f;
    asm volatile("s_nop 1\n\t"
                 "v_permlane16_swap_b32 %0, %1\n\t"
                 "v_permlane16_swap_b32 %2, %3\n\t"
                 "v_permlane16_swap_b32 %4, %5"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(z4));
    const float y0 = x0 + x1, y1 = x2 + x3, y2 = x4 + z4;
    // in-row: i <-> 15-i ; lanes with bit3 = 0 keep y0, bit3 = 1 keep y1 ; y2 plain
    const bool hi8 = (lane & 8) != 0;
    const float z0 = (hi8 ? y1 : y0) + dpp_mov<0x140>(hi8 ? y0 : y1);
    const float z1 = y2 + dpp_mov<0x140>(y2);
    // i <-> i^7 ; bit2 = 0 keeps z0, bit2 = 1 keeps z1
    const bool hi4 = (lane & 4) != 0;
    float w = (hi4 ? z1 : z0) + dpp_mov<0x141>(hi4 ? z0 : z1);
    w += dpp_mov<0x1B>(w);      // quad_perm [3,2,1,0]
    w += dpp_mov<0xB1>(w);      // quad_perm [1,0,3,2]
    return w;
}
