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
    const uint4* tile_order;    // may be NULL -> the spatial order of cugs_blend_tile; else workgroup b works on record b:
                                // {tile, first pair, one past the last pair, 0} - tile and range in ONE load (cugs_tile_order)
};

// Stage list entry `li` (if < end) into LDS slot threadIdx.x.  The LDS copy of the record carries the
// Gaussian index (as bits) in its spare word 7, so that the backward's scatter address comes
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
        r1 = make_float4(c, o, (o >= (1.0f / 255.0f)) ? logf(255.0f * o) : -1.0f, 0.0f);
        r2 = make_float4(s.rgb[g * 3 + 0], 0.0f, s.rgb[g * 3 + 1], s.rgb[g * 3 + 2]);
    }
    r1.w = __int_as_float(g);
    s_rec[threadIdx.x * CUGS_REC_F4 + 0] = r0;
    s_rec[threadIdx.x * CUGS_REC_F4 + 1] = r1;
    s_rec[threadIdx.x * CUGS_REC_F4 + 2] = r2;
    return g;
}

// Can the Gaussian in (r0, r1: the record's geometry chunks) reach alpha >= 1/255 at ANY pixel centre of the rectangle of
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
// error; ocml logf ~2 ulp; cugs_blend_exp_q <= 1e-6 relative.  The slack 0.01 tau + 0.05 + 4e-6 B dominates all of it
// by orders of magnitude.  Any NaN makes the final comparison false -> not culled.
__device__ __forceinline__ bool may_touch_quad(float4 r0, float4 r1, float qx0, float qy0,
                                               float wx, float wy) {
    const float mx = r0.x, my = r0.y, a = r0.z, b = r0.w, c = r1.x;
    const float tau = r1.z;
    if (!(tau >= 0.0f)) return false;
    if (!(a > 0.0f && c > 0.0f)) return true;
    const float ia = __builtin_amdgcn_rcpf(a), ic = __builtin_amdgcn_rcpf(c);   // cull only: 1 ulp is irrelevant
    const float lx = qx0 - mx, hx = lx + wx, ly = qy0 - my, hy = ly + wy;
    const bool inside = (lx <= 0.0f) && (hx >= 0.0f) && (ly <= 0.0f) && (hy >= 0.0f);
    // largest |offset| per axis: lx <= hx, so max(|lx|, |hx|) = max(-lx, hx) - ONE v_max_f32 (as C, fmaxf costs two
    // more instructions that canonicalise its inputs; the values only scale the slack)
    float X, Y;
    asm("v_max_f32_e64 %0, -%1, %2" : "=v"(X) : "v"(lx), "v"(hx));
    asm("v_max_f32_e64 %0, -%1, %2" : "=v"(Y) : "v"(ly), "v"(hy));
    const float B = a * X * X + 2.0f * fabsf(b) * X * Y + c * Y * Y;
    const float b2 = b + b;
    // q is convex with its minimum at the mean, so over a rectangle that does not contain the mean the minimum
    // lies on an edge FACING the mean: the vertical edge on the mean's side if the mean is outside the x-range,
    // the horizontal one likewise (a point of a far edge is reached from the mean through a near edge, where q
    // is smaller).  Each at its clamped 1-D minimiser.
    const float xe = (lx > 0.0f) ? lx : hx, ye = (ly > 0.0f) ? ly : hy;          // the near edges (if outside)
    const float yv = __builtin_amdgcn_fmed3f(-b * xe * ic, ly, hy);             // clamp (ly <= hy): one v_med3_f32
    const float xv = __builtin_amdgcn_fmed3f(-b * ye * ia, lx, hx);             // a NaN input still reaches q through a, b, c
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
//   u = fma(a,dx,b*dy); v = fma(b,dx,c*dy); q = fma(dx,u,dy*v); power = -0.5f*q (never formed here);
//   e = cugs_blend_exp_q(q) = exp(-q/2) clamped below at exp(-6) (cugs_detmath.h: the oracle calls the same
//   function, so e has the same bits there);
//   * power > 0 <=> q < 0 -> skipped: q * 2^127 + 1 (<= -1 for every normal negative q, >= 1 for q >= 0) joins
//     the minimum with 0.99, so alpha comes out negative and fails the 1/255 test;
//   * power < -6: e = exp(-6)(1 + 1e-6) = 0.00248 and opacity <= 1, so alpha < 1/255 = 0.00392 is certain - the
//     oracle reaches the same skip through its alpha test;
//   * open == 0 -> alpha = 0 < 1/255 -> skipped.
// Returns alpha if the Gaussian passes at this pixel, else exactly 0.
struct PixelEval { float dx, dy, gx, gy, e; };
// min(0.99, o e^power) with the power > 0 skip and the open flag folded in, BEFORE the alpha >= 1/255 test.
__device__ __forceinline__ float pixel_alpha_raw(float pxf, float pyf, float mx, float my, float a, float b,
                                                 float c, float o, float open, PixelEval& r) {
    r.dx = pxf - mx;
    r.dy = pyf - my;
    r.gx = fmaf(a, r.dx, b * r.dy);
    r.gy = fmaf(b, r.dx, c * r.dy);
    const float q = fmaf(r.dx, r.gx, r.dy * r.gy);
    r.e = cugs_blend_exp_q(q);
    // q * 2^127 + 1 is >= 1 for q >= 0 and <= -1 for every normal q < 0: as third operand of the minimum it leaves
    // min(alpha, 0.99) alone or turns it negative - below 1/255, skipped (one v_fma + one v_min3)
    const float gate = fmaf(q, 0x1p127f, 1.0f);
    return __builtin_fminf(__builtin_fminf((o * open) * r.e, 0.99f), gate);   // o * 1.0f is exact
}
__device__ __forceinline__ float pixel_alpha(float pxf, float pyf, float mx, float my, float a, float b,
                                             float c, float o, float open, PixelEval& r) {
    const float alpha = pixel_alpha_raw(pxf, pyf, mx, my, a, b, c, o, open, r);
    return (alpha < (1.0f / 255.0f)) ? 0.0f : alpha;
}

// ---------------------------------------------------------------------------------------
// Cost model behind the choices below (tools/microbench/valu_rate.hip on MI355X, 8 waves per SIMD,
// chip-wide wave-instructions per second, plain fma/mul/add = 1 unit):
//   v_fma/v_mul/v_add/v_mov (also with `clamp`)  950 G/s   1.0      v_max/v_min/v_med3            585 G/s  1.6
//   DPP add (any control, any bank mask)       560-590     1.6-1.7  v_cmp (vcc or sgpr pair)      560     1.7
//   v_cndmask (sgpr mask)                        518       1.85     v_rcp/v_exp, v_permlane*_swap 300     3.2
// The blend kernels are bound by this issue rate, so per-lane decisions are kept as 0/1 FLOATS produced by
// `v_fma ... clamp` and multiplied in (1 unit each) instead of v_cmp + v_cndmask (3.55 units a pair), and the
// wave reduction avoids v_cndmask altogether.
// ---------------------------------------------------------------------------------------

// saturate(a*b + c) in one instruction (v_fma_f32 ... clamp).
__device__ __forceinline__ float sat_fma(float a, float b, float c) {
    return __builtin_amdgcn_fmed3f(fmaf(a, b, c), 0.0f, 1.0f);
}
__device__ __forceinline__ float sat_add(float a, float b) { return __builtin_amdgcn_fmed3f(a + b, 0.0f, 1.0f); }

// 1.0f if x >= 1/255 (the reference's `alpha < 1/255 -> skip`, forward.cu:141 / backward.cu:137) else 0.0f, exactly,
// for every non-NaN x <= 2^31: with p = the float just below 1/255, x >= 1/255 <=> x > p <=> x - p >= ulp = 2^-31.
// The fma rounds (x - p) * 2^32 once: >= 2 when x passes, <= 0 when it does not.  NaN -> 0 (dx10 clamp).
#define CUGS_ALPHA_MIN_PRED 0x1.01010p-8f      /* 0x3B808080: pred(1.0f/255.0f), 1/255 = 0x3B808081 */
__device__ __forceinline__ float passes_alpha_min(float x) {
    return sat_fma(x, 0x1p32f, -CUGS_ALPHA_MIN_PRED * 0x1p32f);
}
// 1.0f if x < 0.99f else 0.0f (x <= 0.99f always here): 0.99f - x >= ulp(0.5..1) = 2^-24.
__device__ __forceinline__ float below_alpha_cap(float x) { return sat_fma(x, -0x1p32f, 0.99f * 0x1p32f); }

// ---------------------------------------------------------------------------------------
// reduce9t: sums NINE per-lane values across the 64 lanes of a wave and leaves each total in a different
// lane ("transpose-reduce": at every halving step two partner lanes exchange the half of the values they
// will not keep, so the number of live values halves as the number of lanes sharing a sum doubles),
// without a single v_cndmask:
//   stage 1  row_mirror (i <-> 15-i, side = lane bit 3 = banks {0,1} | {2,3}):  9 -> 5 values.
//            The "keep mine / give the other" selection is the DPP BANK MASK: `x + dpp(x)` written to banks
//            0,1 from one register and to banks 2,3 from the other (two DPP adds per pair, 3.4 units instead of
//            two selects + one DPP add, 5.3).  The first pair needs no selection at all: the caller hands it
//            over already swizzled (cA holds value 0 in side-0 lanes and value 1 in side-1 lanes, cB the
//            opposite - for the blend backward these are dL/dcolour products whose per-lane factor is
//            swizzled once per kernel), so it is ONE DPP add.
//   stage 2  row_half_mirror (i <-> i^7, side = lane bit 2 = banks {0,2} | {1,3}):  5 -> 3, same trick.
//   stage 3  v_permlane16_swap (rows 0<->1, 2<->3): swaps odd rows of x with even rows of y, so x + y is the
//            transposed pair sum with no selection: 3 -> 2 (the odd value is summed with a copy of itself).
//   stage 4  v_permlane32_swap (halves): 2 -> 1.
//   stages 5, 6  the four lanes of each bank still hold partial sums of the same value: two quad_perm adds.
// 13 DPP adds + 3 swaps + 5 plain = ~37 units (the select-based version: 57).  All 64 lanes must be active.
//
// Result: lane L returns the WAVE total of slot reduce9t_slot(L); within rows 0..2 every lane of a bank holds
// the same total, the first lane of each bank is designated to deliver:
//   row 0: banks 0..3 -> slots 0, 2, 1, 3;  row 1: banks 0..3 -> slots 4, 6, 5, 8;  row 2: slot 7.
// Slot k is the k-th value in the argument order (v0 v1 | v2 v3 | v4 v5 | v6 v8 | v7), where the caller passes
// v0/v1 swizzled as described.
//
// Hand-scheduled inline asm: hipcc pads nothing inside an asm statement, so the wait states gfx950 needs
// between a VALU write and a DPP / permlane read of the same register (2) are s_nops in the string; the
// instruction order keeps producers and DPP consumers at least two instructions apart elsewhere.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int reduce9t_slot(int lane) {
    if (lane & 3) return -1;
    const int bank = (lane >> 2) & 3, row = lane >> 4;
    const int perm[4] = {0, 2, 1, 3};
    if (row == 0) return perm[bank];
    if (row == 1) return bank == 3 ? 8 : 4 + perm[bank];
    if (row == 2 && bank == 0) return 7;
    return -1;
}

__device__ __forceinline__ float reduce9t(float cA, float cB, float v2, float v3, float v4, float v5, float v6,
                                          float v8, float v7) {
    float a0, a1, a2, a3, a4, b0, b1, b2, d;
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %10, %9 row_mirror row_mask:0xf bank_mask:0xf\n\t"       // a0 = cA + mirror(cB)
        "v_add_f32_dpp %1, %11, %11 row_mirror row_mask:0xf bank_mask:0x3\n\t"      // a1 = v2 + mirror(v2) | v3 + mirror(v3)
        "v_add_f32_dpp %1, %12, %12 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %2, %13, %13 row_mirror row_mask:0xf bank_mask:0x3\n\t"      // a2 = v4 | v5
        "v_add_f32_dpp %2, %14, %14 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %3, %15, %15 row_mirror row_mask:0xf bank_mask:0x3\n\t"      // a3 = v6 | v8
        "v_add_f32_dpp %3, %16, %16 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %4, %17, %17 row_mirror row_mask:0xf bank_mask:0xf\n\t"      // a4 = v7
        "v_add_f32_dpp %5, %0, %0 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"   // b0 = a0 | a1
        "v_add_f32_dpp %5, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %7, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"   // b2 = a4
        "v_add_f32_dpp %6, %2, %2 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"   // b1 = a2 | a3
        "v_add_f32_dpp %6, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_mov_b32 %8, %7\n\t"                                                      // d = copy of b2
        "s_nop 1\n\t"
        "v_permlane16_swap_b32 %5, %6\n\t"                                          // rows: (b0,b1) transposed
        "v_permlane16_swap_b32 %7, %8\n\t"
        "v_add_f32 %5, %5, %6\n\t"                                                  // c0
        "v_add_f32 %7, %7, %8\n\t"                                                  // c1
        "s_nop 1\n\t"
        "v_permlane32_swap_b32 %5, %7\n\t"                                          // halves: (c0,c1) transposed
        "v_add_f32 %8, %5, %7\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %8, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %8, %8, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(d)
        : "v"(cA), "v"(cB), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v8), "v"(v7));
    return d;
}

// ---------------------------------------------------------------------------------------
// reduce9r16: the same transpose-reduce inside each 16-lane DPP ROW independently (four rows = four different
// Gaussians in the blend backward's second phase).  Stages 1 and 2 as in reduce9t (bank-masked DPP, first pair
// swizzled by lane bit 3); the two in-bank stages use selects (side = lane bit 1, then bit 0): there is no mask
// finer than a bank.  9 -> 5 -> 3 -> 2 -> 1 values; ~36 units per call = 9 per row.
// Result, per row: lane r (0..15) returns the ROW total of slot reduce9r16_slot(r):
//   r = 4*bank + 0 -> slots 0, 2, 1, 3 (bank 0..3);  r = 4*bank + 2 -> slots 4, 6, 5, 8;  r = 1 -> slot 7.
// ---------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ float xchg_add(bool side, float a, float b) {
    return (side ? b : a) + dpp_mov<CTRL>(side ? a : b);
}

__device__ __forceinline__ int reduce9r16_slot(int lane) {
    const int r = lane & 15, bank = r >> 2;
    const int perm[4] = {0, 2, 1, 3};
    if ((r & 3) == 0) return perm[bank];
    if ((r & 3) == 2) return bank == 3 ? 8 : 4 + perm[bank];
    if (r == 1) return 7;
    return -1;
}

__device__ __forceinline__ float reduce9r16(float cA, float cB, float v2, float v3, float v4, float v5, float v6,
                                            float v8, float v7, int lane) {
    float a0, a1, a2, a3, a4, b0, b1, b2;
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %9, %8 row_mirror row_mask:0xf bank_mask:0xf\n\t"        // a0 = cA + mirror(cB)
        "v_add_f32_dpp %1, %10, %10 row_mirror row_mask:0xf bank_mask:0x3\n\t"      // a1 = v2 | v3
        "v_add_f32_dpp %1, %11, %11 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %2, %12, %12 row_mirror row_mask:0xf bank_mask:0x3\n\t"      // a2 = v4 | v5
        "v_add_f32_dpp %2, %13, %13 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %3, %14, %14 row_mirror row_mask:0xf bank_mask:0x3\n\t"      // a3 = v6 | v8
        "v_add_f32_dpp %3, %15, %15 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %4, %16, %16 row_mirror row_mask:0xf bank_mask:0xf\n\t"      // a4 = v7
        "v_add_f32_dpp %5, %0, %0 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"   // b0 = a0 | a1
        "v_add_f32_dpp %5, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %7, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"   // b2 = a4
        "v_add_f32_dpp %6, %2, %2 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"   // b1 = a2 | a3
        "v_add_f32_dpp %6, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "s_nop 1\n\t"                                                               // the compiler's DPP reads b1 next
        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(b0), "=&v"(b1), "=&v"(b2)
        : "v"(cA), "v"(cB), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v8), "v"(v7));
    const bool s1 = (lane & 2) != 0, s0 = (lane & 1) != 0;
    const float c0 = xchg_add<0x4E>(s1, b0, b1);                   // quad_perm [2,3,0,1]
    const float c1 = b2 + dpp_mov<0x4E>(b2);
    return xchg_add<0xB1>(s0, c0, c1);                             // quad_perm [1,0,3,2]
}
