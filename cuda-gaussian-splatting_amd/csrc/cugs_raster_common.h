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
// quad offsets the minimum of q is 0 if the mean lies inside, else it is attained on an edge facing the
// mean; along each edge q is a 1-D convex parabola (a, c > 0), minimised at the clamped vertex.
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
    // q is convex with its minimum at the mean, so over a rectangle that does not contain the mean the minimum
    // lies on an edge FACING the mean: the vertical edge on the mean's side if the mean is outside the x-range,
    // the horizontal one likewise (a point of a far edge is reached from the mean through a near edge, where q
    // is smaller).  Each at its clamped 1-D minimiser.
    const float xe = (lx > 0.0f) ? lx : hx, ye = (ly > 0.0f) ? ly : hy;          // the near edges (if outside)
    const float yv = fminf(fmaxf(-b * xe * ic, ly), hy);
    const float xv = fminf(fmaxf(-b * ye * ia, lx), hx);
    const float qx = a * xe * xe + b2 * xe * yv + c * yv * yv;
    const float qy = a * xv * xv + b2 * xv * ye + c * ye * ye;
    const bool out_x = !((lx <= 0.0f) && (hx >= 0.0f)), out_y = !((ly <= 0.0f) && (hy >= 0.0f));
    const float qmin = inside ? 0.0f : fminf(out_x ? qx : 3.0e38f, out_y ? qy : 3.0e38f);
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

// forward.cu:124-141 / backward.cu:123-137 for one pixel, written so that every per-lane decision
// stays in VECTOR registers.  (hipcc keeps a C++ `bool` that differs per lane as a 64-bit lane mask in
// scalar registers and turns `a && b`, `done |= x` into s_and/s_or_b64; a CU has ONE scalar unit for its
// four SIMDs, and the ~45 scalar instructions per step this produced cost more issue time than the
// ~110 vector ones.)  The pixel's "still open" state is a float flag (1 or 0) multiplied into the
// opacity, and skips become alpha = 0, which makes every later product exactly zero.
//
// Decisions are the oracle's, bit for bit.  FMA placement contract:
//   u = fma(a,dx,b*dy); v = fma(b,dx,c*dy); q = fma(dx,u,dy*v); power = -0.5f*q.
//   * power > 0 -> skipped: the exponent is replaced by -6;
//   * power < -6 is clamped to -6: exp(-6)(1+2^-22) = 0.00248 and opacity <= 1, so alpha < 1/255 =
//     0.00392 is certain - the oracle reaches the same skip through its alpha test (and its exp flushes
//     below -87.3); inside [-6, 0] cugs_expf_core IS cugs_expf;
//   * open == 0 -> alpha = 0 < 1/255 -> skipped.
// Returns alpha if the Gaussian passes at this pixel, else exactly 0.
struct PixelEval { float dx, dy, gx, gy, e; };
__device__ __forceinline__ float pixel_alpha(float pxf, float pyf, float mx, float my, float a, float b,
                                             float c, float o, float open, PixelEval& r) {
    r.dx = pxf - mx;
    r.dy = pyf - my;
    r.gx = fmaf(a, r.dx, b * r.dy);
    r.gy = fmaf(b, r.dx, c * r.dy);
    const float power = -0.5f * fmaf(r.dx, r.gx, r.dy * r.gy);
    float pw = fmaxf(power, -6.0f);
    pw = (power > 0.0f) ? -6.0f : pw;
    r.e = cugs_expf_small(pw);                                // pw in [-6, 0]: same bits as cugs_expf
    const float alpha = fminf((o * open) * r.e, 0.99f);       // o * 1.0f is exact
    return (alpha < (1.0f / 255.0f)) ? 0.0f : alpha;
}

// ---------------------------------------------------------------------------------------
// reduce9: sums NINE per-lane values across the 64 lanes of a wave and leaves each total in a
// different lane (33 VALU-class operations instead of 9 x 6 DPP adds + a 9-way select).
//
// "Transpose-reduce": at every halving step two partner lanes exchange the half of the values they
// will not keep (`keep = side ? b : a; give = side ? a : b; keep + dpp(give)`), so the number of live
// values per lane halves as the number of lanes sharing a sum doubles.  Inside each 16-lane row the
// partners / sides are
//   row_mirror (i <-> 15-i, side = bit3), row_half_mirror (i <-> i^7, side = bit2),
//   quad_perm[3,2,1,0] (i <-> i^3, side = bit1), quad_perm[1,0,3,2] (i <-> i^1, side = bit0)
// (before each step both partners hold the same set of values: their higher side bits agree); live
// values 9 -> 5 -> 3 -> 2 -> 1.  The four row totals of the single remaining value are then combined
// with gfx950's v_permlane16_swap / v_permlane32_swap (x = t, y = t; swap; x + y), which measured
// markedly cheaper than doing the two cross-row halvings first on nine values (8 swaps).
// All 64 lanes must be active (EXEC full): callers run it in wave-uniform control flow with zeros in
// lanes that have nothing to add.
//
// Result: lane L (any row; i = L & 15) returns the WAVE total of slot reduce9_slot(L):
//   i even: slot 4*bit1 + 2*bit2 + bit3  (i = 0,8,4,12,2,10,6,14 -> slots 0..7);  i == 1: slot 8;
//   only row 0's lanes are designated to deliver.
//
// The swaps are issued through inline asm: with ROCm 7.2's hipcc the two results of
// __builtin_amdgcn_permlane{16,32}_swap collapse into one register when both feed the same add
// (observed: `v_permlane32_swap v2, v3 ; v_add_f32 v2, v2, v2`).  hipcc pads nothing inside an asm
// statement, so the 2 wait states it otherwise inserts between a VALU write and a swap that reads
// the register (s_nop 1) are part of the string.
// ---------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ float xchg_add(bool side, float a, float b) {
    return (side ? b : a) + dpp_mov<CTRL>(side ? a : b);
}

__device__ __forceinline__ int reduce9_slot(int lane) {
    const int li = lane & 15;
    if (lane >= 16) return -1;
    if (li == 1) return 8;
    if (li & 1) return -1;
    return ((li >> 1) & 1) * 4 + ((li >> 2) & 1) * 2 + ((li >> 3) & 1);
}

__device__ __forceinline__ float reduce9(float v0, float v1, float v2, float v3, float v4, float v5, float v6,
                                         float v7, float v8, int lane) {
    const bool s3 = (lane & 8) != 0, s2 = (lane & 4) != 0, s1 = (lane & 2) != 0, s0 = (lane & 1) != 0;
    const float a0 = xchg_add<0x140>(s3, v0, v1), a1 = xchg_add<0x140>(s3, v2, v3);      // row_mirror
    const float a2 = xchg_add<0x140>(s3, v4, v5), a3 = xchg_add<0x140>(s3, v6, v7);
    const float a4 = v8 + dpp_mov<0x140>(v8);
    const float b0 = xchg_add<0x141>(s2, a0, a1), b1 = xchg_add<0x141>(s2, a2, a3);      // row_half_mirror
    const float b2 = a4 + dpp_mov<0x141>(a4);
    const float c0 = xchg_add<0x1B>(s1, b0, b1);                                         // quad_perm [3,2,1,0]
    const float c1 = b2 + dpp_mov<0x1B>(b2);
    float x = xchg_add<0xB1>(s0, c0, c1);                                                // quad_perm [1,0,3,2]
    float y = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));         // rows 0+1, 2+3
    x += y;
    y = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));         // halves
    return x + y;
}
