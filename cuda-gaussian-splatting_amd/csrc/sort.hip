// sort.hip — (tile | depth) ordering of (tile, Gaussian) pairs.
//
// Replaces sort_gaussians (rasterizer/sorting.cu:115-227): the int32 cumsum + blocking .item()
// (:145-146), k_fill_sort_pairs (:30-72), cub::DeviceRadixSort::SortPairs over all 64 key bits
// (:191-210) and k_compute_tile_ranges (:82-109).
//
// Required result (bit-exact): pairs ordered by the 64-bit key (tile_id << 32 | float_bits(depth)),
// ties in ascending Gaussian index (CUB's sort is stable and the reference fills in index order).
//
// How it is produced here (not the reference's schedule): a stable LSD sort by the full key is the
// same permutation as (1) a stable sort of the N Gaussians by depth bits, (2) emitting each
// Gaussian's pairs in that order, (3) a stable sort of the P pairs by tile id alone.  (1) moves
// 8 B x N x 4 passes, (3) moves 8 B x P x ceil(log2(tiles)/8) passes (2 at 1080p) instead of
// 12 B x P x 8 passes.  Every pass is the same three kernels: per-workgroup digit histogram,
// per-digit row scan, stable scatter with wave64 ballot ranking.  All HBM-bound integer work.
#include "cugs_gaussian_math.h"

namespace {

constexpr int RADIX = 256;
constexpr int IPT = 16;                           // items per thread
constexpr int CHUNK = CUGS_BLOCK * IPT;           // 4096 items per workgroup
constexpr int WAVE_ITEMS = CUGS_WAVE * IPT;       // 1024 contiguous items per wave
constexpr int FILL_CHUNK = CUGS_BLOCK;            // Gaussians per workgroup in scan/fill (one per thread)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline uint32_t nblocks_for(int64_t count, int chunk) { return (uint32_t)((count + chunk - 1) / chunk); }

// Two caller-owned scratch buffers.  The N-level one is filled by cugs_sort_count_pairs (depth order,
// scanned block sums, pair total) and read by cugs_sort_pairs; the pair-level one can only be sized
// once the pair count is known.
struct SortWsN {
    unsigned long long* total;   // [0] pair total (u64), dword 4: zero-pair counter (quirk Q12)
    uint32_t* dkey[2];           // depth bits, ping-pong              [n]
    uint32_t* dval[2];           // Gaussian index, ping-pong          [n]
    uint32_t* tot;               // [RADIX]
    uint32_t* blocksum;          // per FILL_CHUNK block pair counts   [nfill + 2]
    uint32_t* hist;              // [RADIX][nblk_n] digit-major
    size_t bytes;
};
struct SortWsP {
    uint32_t* ptile[2];          // tile id per pair, ping-pong        [P]
    uint32_t* pidx[2];           // Gaussian index per pair            [P]
    uint32_t* hist;              // [RADIX][nblk_p]
    size_t bytes;
};

struct Carver {
    char* base; size_t off = 0;
    template <typename T> T* take(size_t count) {
        size_t o = off;
        off = align_up(off + sizeof(T) * count, 256);
        return reinterpret_cast<T*>(base + o);
    }
};

SortWsN carve_n(void* base, int64_t n) {
    Carver c{static_cast<char*>(base)};
    SortWsN w;
    w.total = c.take<unsigned long long>(32);
    for (int i = 0; i < 2; ++i) w.dkey[i] = c.take<uint32_t>((size_t)n);
    for (int i = 0; i < 2; ++i) w.dval[i] = c.take<uint32_t>((size_t)n);
    w.tot = c.take<uint32_t>(RADIX);
    w.blocksum = c.take<uint32_t>((size_t)nblocks_for(n, FILL_CHUNK) + 2);
    w.hist = c.take<uint32_t>((size_t)RADIX * (nblocks_for(n, CHUNK) + 1));
    w.bytes = c.off;
    return w;
}
SortWsP carve_p(void* base, int64_t pairs) {
    Carver c{static_cast<char*>(base)};
    SortWsP w;
    for (int i = 0; i < 2; ++i) w.ptile[i] = c.take<uint32_t>((size_t)pairs);
    for (int i = 0; i < 2; ++i) w.pidx[i] = c.take<uint32_t>((size_t)pairs);
    w.hist = c.take<uint32_t>((size_t)RADIX * (nblocks_for(pairs, CHUNK) + 1));
    w.bytes = c.off;
    return w;
}

// ------------------------------------------------------------------------------------
// wave / workgroup scan helpers (low-frequency paths; plain shuffles)
// ------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}

// Exclusive scan of one value per thread over a 256-thread workgroup; *total = workgroup sum.
// s_tmp: 4 dwords of LDS.  Contains two barriers.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* s_tmp, uint32_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = wave_inclusive_scan(v);
    if (lane == 63) s_tmp[wave] = inc;
    __syncthreads();
    uint32_t w0 = s_tmp[0], w1 = s_tmp[1], w2 = s_tmp[2], w3 = s_tmp[3];
    uint32_t base = (wave > 0 ? w0 : 0u) + (wave > 1 ? w1 : 0u) + (wave > 2 ? w2 : 0u);
    if (total) *total = w0 + w1 + w2 + w3;
    __syncthreads();
    return base + inc - v;
}

// Reference quirk Q12: a splat whose tile rectangle is empty in both axes still has
// tiles_touched = (negative) x (negative) > 0 (projection.cu:187-188); k_fill_sort_pairs writes
// nothing for it and its reserved slots keep the zero-initialised (key 0, value 0) pairs
// (sorting.cu:166-167), which sort to the front of tile 0.  Reproduced here by giving such a
// Gaussian the depth key 0 (its pairs are emitted first) and emitting (tile 0, Gaussian 0).
__device__ __forceinline__ bool fills_nothing(const float* __restrict__ means_2d, int radius, uint32_t idx,
                                              int img_w, int img_h, int ntx, int nty) {
    if (radius <= 0) return true;                                   // sorting.cu:44-45
    const TileRect tr = tile_rect_of(means_2d[idx * 2 + 0], means_2d[idx * 2 + 1], radius, img_w, img_h,
                                     ntx, nty);
    return tr.x1 <= tr.x0 || tr.y1 <= tr.y0;
}

__global__ __launch_bounds__(CUGS_BLOCK) void k_depth_keys(uint32_t n, const float* __restrict__ depths,
                                                           const float* __restrict__ means_2d,
                                                           const int32_t* __restrict__ radii,
                                                           const int32_t* __restrict__ tiles, int img_w,
                                                           int img_h, int ntx, int nty,
                                                           uint32_t* __restrict__ keys) {
    const uint32_t i = blockIdx.x * CUGS_BLOCK + threadIdx.x;
    if (i >= n) return;
    uint32_t key = __float_as_uint(depths[i]);
    if (tiles[i] > 0 && fills_nothing(means_2d, radii[i], i, img_w, img_h, ntx, nty)) key = 0u;
    keys[i] = key;
}

// ------------------------------------------------------------------------------------
// One radix pass = hist + row scan + scatter.  SRC_DEPTH: first pass of the depth sort, which
// generates the Gaussian index on the fly instead of reading a value array.
// ------------------------------------------------------------------------------------
template <bool SRC_DEPTH>
__device__ __forceinline__ uint32_t load_key(const uint32_t* __restrict__ keys, uint32_t i) {
    return keys[i];
}

template <bool SRC_DEPTH>
__global__ __launch_bounds__(CUGS_BLOCK) void k_radix_hist(const uint32_t* __restrict__ keys,
                                                           uint32_t count, int shift, uint32_t mask,
                                                           uint32_t* __restrict__ hist, uint32_t nblk) {
    __shared__ uint32_t s_cnt[RADIX];
    s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t wbase = blockIdx.x * CHUNK + wave * WAVE_ITEMS;
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        uint32_t i = wbase + r * CUGS_WAVE + lane;
        if (i < count) atomicAdd(&s_cnt[(load_key<SRC_DEPTH>(keys, i) >> shift) & mask], 1u);
    }
    __syncthreads();
    hist[threadIdx.x * nblk + blockIdx.x] = s_cnt[threadIdx.x];
}

// Block d: exclusive scan of row d of hist (in place); tot[d] = row sum.
__global__ __launch_bounds__(CUGS_BLOCK) void k_radix_scan_rows(uint32_t* __restrict__ hist,
                                                                uint32_t nblk, uint32_t* __restrict__ tot) {
    __shared__ uint32_t s_tmp[4];
    uint32_t* row = hist + (size_t)blockIdx.x * nblk;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nblk; base += CUGS_BLOCK) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = i < nblk ? row[i] : 0u;
        uint32_t total;
        uint32_t ex = block_exclusive_scan(v, s_tmp, &total);
        if (i < nblk) row[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) tot[blockIdx.x] = carry;
}

// Stable scatter.  Ranking: each wave owns a contiguous 1024-item slice and walks it in rounds of
// 64; in a round the lanes holding the same digit find each other with 8 ballots (match-any), the
// rank is the popcount below the lane, and the group's highest lane advances the wave's running
// base in LDS.  The (key, value) pairs are first placed at their position in the workgroup's LOCALLY
// sorted order in LDS and then streamed out, so that consecutive lanes write consecutive global
// addresses inside each digit's run (4 B items scattered straight to 128-256 buckets cost ~2x).
template <bool SRC_DEPTH>
__global__ __launch_bounds__(CUGS_BLOCK) void k_radix_scatter(
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint32_t count,
    int shift, uint32_t mask, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ tot,
    uint32_t nblk, uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {
    __shared__ uint32_t s_lbase[4][RADIX];     // per (wave, digit): count, then running LOCAL position
    __shared__ uint32_t s_lstart[RADIX];       // first local position of digit d
    __shared__ uint32_t s_gbase[RADIX];        // first global position of this workgroup's digit-d run
    __shared__ uint32_t s_key[CHUNK];
    __shared__ uint32_t s_val[CHUNK];
    __shared__ uint32_t s_tmp[4];
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t bbase = blockIdx.x * CHUNK;
    const uint32_t wbase = bbase + wave * WAVE_ITEMS;
    const uint32_t count_blk = min((uint32_t)CHUNK, count - bbase);

#pragma unroll
    for (int w = 0; w < 4; ++w) s_lbase[w][tid] = 0;
    __syncthreads();

    uint32_t k[IPT], v[IPT];
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        uint32_t i = wbase + r * CUGS_WAVE + lane;
        bool ok = i < count;
        k[r] = ok ? keys_in[i] : 0xFFFFFFFFu;
        v[r] = ok ? (SRC_DEPTH ? i : vals_in[i]) : 0u;
        if (ok) atomicAdd(&s_lbase[wave][(k[r] >> shift) & mask], 1u);
    }
    __syncthreads();

    {   // digit d = tid
        const uint32_t c0 = s_lbase[0][tid], c1 = s_lbase[1][tid], c2 = s_lbase[2][tid], c3 = s_lbase[3][tid];
        const uint32_t dig_base = block_exclusive_scan(tot[tid], s_tmp, nullptr);            // global digit start
        const uint32_t lstart = block_exclusive_scan(c0 + c1 + c2 + c3, s_tmp, nullptr);     // local digit start
        s_gbase[tid] = dig_base + hist[tid * nblk + blockIdx.x];
        s_lstart[tid] = lstart;
        s_lbase[0][tid] = lstart;
        s_lbase[1][tid] = lstart + c0;
        s_lbase[2][tid] = lstart + c0 + c1;
        s_lbase[3][tid] = lstart + c0 + c1 + c2;
    }
    __syncthreads();

    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const uint32_t i = wbase + r * CUGS_WAVE + lane;
        const bool ok = i < count;
        const uint32_t d = (k[r] >> shift) & mask;
        unsigned long long peers = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        if (ok) {
            const uint32_t base = s_lbase[wave][d];
            const uint32_t pos = base + __popcll(peers & lt_mask);
            s_key[pos] = k[r];
            s_val[pos] = v[r];
            if ((peers >> lane) == 1ull) s_lbase[wave][d] = base + __popcll(peers);
        }
    }
    __syncthreads();

#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const uint32_t j = r * CUGS_BLOCK + tid;
        if (j < count_blk) {
            const uint32_t key = s_key[j];
            const uint32_t d = (key >> shift) & mask;
            const uint32_t dst = s_gbase[d] + (j - s_lstart[d]);
            keys_out[dst] = key;
            vals_out[dst] = s_val[j];
        }
    }
}

// ------------------------------------------------------------------------------------
// Pair emission in depth order
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(CUGS_BLOCK) void k_fill_blocksums(uint32_t n,
                                                               const uint32_t* __restrict__ order,
                                                               const int32_t* __restrict__ tiles,
                                                               uint32_t* __restrict__ blocksum) {
    __shared__ uint32_t s_tmp[4];
    const uint32_t i = blockIdx.x * FILL_CHUNK + threadIdx.x;
    const uint32_t acc = i < n ? (uint32_t)tiles[order[i]] : 0u;
    uint32_t total;
    block_exclusive_scan(acc, s_tmp, &total);
    if (threadIdx.x == 0) blocksum[blockIdx.x] = total;
}

// Single workgroup: exclusive scan of blocksum[0..nb) in place; *total = the 64-bit grand total =
// sum(tiles_touched), the reference's cumsum[-1].item() (sorting.cu:145-146).
__global__ __launch_bounds__(CUGS_BLOCK) void k_scan_blocksums(uint32_t* __restrict__ blocksum, uint32_t nb,
                                                               unsigned long long* __restrict__ total) {
    __shared__ uint32_t s_tmp[4];
    unsigned long long carry = 0;
    for (uint32_t base = 0; base < nb; base += CUGS_BLOCK) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = i < nb ? blocksum[i] : 0u;
        uint32_t chunk_total;
        uint32_t ex = block_exclusive_scan(v, s_tmp, &chunk_total);
        if (i < nb) blocksum[i] = (uint32_t)carry + ex;     // valid whenever the total fits int32 (checked on the host)
        carry += chunk_total;
    }
    if (threadIdx.x == 0) *total = carry;
}

// k_fill_sort_pairs (sorting.cu:30-72), walked in depth order; only the tile id and the index are
// stored (the depth half of the key is implied by the order).  One Gaussian per thread computes
// its rectangle and its offset (workgroup scan); the workgroup then emits its pairs COOPERATIVELY:
// output position k is owned by lane k % 256, which finds the Gaussian by binary search over the
// 256 offsets in LDS - consecutive lanes write consecutive pairs (fully coalesced), and a splat
// covering thousands of tiles no longer serialises one thread.
__global__ __launch_bounds__(CUGS_BLOCK) void k_fill_pairs(
    uint32_t n, uint32_t total_pairs, const uint32_t* __restrict__ order,
    const int32_t* __restrict__ tiles, const float* __restrict__ means_2d,
    const int32_t* __restrict__ radii, int img_w, int img_h, int ntx, int nty,
    const uint32_t* __restrict__ blocksum, uint32_t* __restrict__ ptile, uint32_t* __restrict__ pidx,
    uint32_t* __restrict__ zero_pairs) {
    __shared__ uint32_t s_tmp[4];
    __shared__ uint32_t s_off[CUGS_BLOCK + 1];
    __shared__ uint32_t s_g[CUGS_BLOCK];
    __shared__ int s_x0[CUGS_BLOCK], s_y0[CUGS_BLOCK], s_w[CUGS_BLOCK], s_cnt[CUGS_BLOCK];
    const uint32_t tid = threadIdx.x;
    const uint32_t i = blockIdx.x * FILL_CHUNK + tid;
    uint32_t g = 0, t = 0;
    int x0 = 0, y0 = 0, w = 0, real = 0;          // real = pairs the reference's loops would write
    if (i < n) {
        g = order[i];
        t = (uint32_t)tiles[g];
        if (t > 0) {
            const int radius = radii[g];
            if (radius > 0) {                                        // sorting.cu:44-45
                const TileRect tr = tile_rect_of(means_2d[g * 2 + 0], means_2d[g * 2 + 1], radius, img_w, img_h,
                                                 ntx, nty);
                if (tr.x1 > tr.x0 && tr.y1 > tr.y0) {
                    x0 = tr.x0; y0 = tr.y0; w = tr.x1 - tr.x0;
                    real = w * (tr.y1 - tr.y0);
                }
            }
            if ((uint32_t)real < t) atomicAdd(zero_pairs, t - (uint32_t)real);   // quirk Q12 slots (rare)
        }
    }
    uint32_t blk_total;
    const uint32_t off = block_exclusive_scan(t, s_tmp, &blk_total);
    s_off[tid] = off;
    s_g[tid] = g; s_x0[tid] = x0; s_y0[tid] = y0; s_w[tid] = w; s_cnt[tid] = real;
    if (tid == 0) s_off[CUGS_BLOCK] = blk_total;
    __syncthreads();

    const uint32_t out_base = blocksum[blockIdx.x];
    for (uint32_t k = tid; k < blk_total; k += CUGS_BLOCK) {
        // largest j with s_off[j] <= k among entries with a non-empty span: upper_bound - 1
        uint32_t lo = 0, hi = CUGS_BLOCK;                            // invariant: s_off[lo] <= k < s_off[hi]
#pragma unroll
        for (int step = 0; step < 8; ++step) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_off[mid] <= k) lo = mid; else hi = mid;
        }
        const uint32_t local = k - s_off[lo];
        uint32_t tile = 0u, idx = 0u;                                // slots the reference leaves at zero (Q12)
        if ((int)local < s_cnt[lo]) {
            const int ww = s_w[lo];
            const int row = (int)local / ww;
            tile = (uint32_t)((s_y0[lo] + row) * ntx + s_x0[lo] + ((int)local - row * ww));
            idx = s_g[lo];
        }
        const uint32_t dst = out_base + k;
        if (dst < total_pairs) {                                     // never write past the buffers
            ptile[dst] = tile;
            pidx[dst] = idx;
        }
    }
}

// k_compute_tile_ranges (sorting.cu:82-109) on the sorted tile ids; optionally rebuilds the
// reference's sorted 64-bit keys (SortingOutput::gaussian_keys_sorted, sorting.hpp:20).
__global__ __launch_bounds__(CUGS_BLOCK) void k_tile_ranges(uint32_t total_pairs,
                                                            const uint32_t* __restrict__ ptile,
                                                            const int32_t* __restrict__ pidx,
                                                            const float* __restrict__ depths,
                                                            int32_t* __restrict__ tile_ranges,
                                                            uint64_t* __restrict__ keys_sorted,
                                                            const uint32_t* __restrict__ zero_pairs) {
    const uint32_t i = blockIdx.x * CUGS_BLOCK + threadIdx.x;
    if (i >= total_pairs) return;
    const uint32_t cur = ptile[i];
    if (i == 0) {
        tile_ranges[cur * 2 + 0] = 0;
    } else {
        const uint32_t prev = ptile[i - 1];
        if (cur != prev) {
            tile_ranges[prev * 2 + 1] = (int32_t)i;
            tile_ranges[cur * 2 + 0] = (int32_t)i;
        }
    }
    if (i == total_pairs - 1) tile_ranges[cur * 2 + 1] = (int32_t)total_pairs;
    if (keys_sorted)   // Q12 pairs are the leading entries of tile 0 and carry depth bits 0
        keys_sorted[i] = (i < *zero_pairs) ? 0ull
                                            : (((uint64_t)cur << 32) | (uint64_t)__float_as_uint(depths[pidx[i]]));
}

template <bool SRC_DEPTH>
int radix_pass(const uint32_t* kin, const uint32_t* vin, uint32_t count, int shift, int bits,
               uint32_t* hist, uint32_t* tot, uint32_t* kout, uint32_t* vout, hipStream_t st) {
    const uint32_t nblk = nblocks_for(count, CHUNK);
    const uint32_t mask = (1u << bits) - 1u;
    hipLaunchKernelGGL((k_radix_hist<SRC_DEPTH>), dim3(nblk), dim3(CUGS_BLOCK), 0, st, kin, count, shift,
                       mask, hist, nblk);
    CUGS_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_radix_scan_rows, dim3(RADIX), dim3(CUGS_BLOCK), 0, st, hist, nblk, tot);
    CUGS_LAUNCH_CHECK();
    hipLaunchKernelGGL((k_radix_scatter<SRC_DEPTH>), dim3(nblk), dim3(CUGS_BLOCK), 0, st, kin, vin, count,
                       shift, mask, hist, tot, nblk, kout, vout);
    CUGS_LAUNCH_CHECK();
    return 0;
}

int tile_bits(int tiles) {
    int b = 1;
    while ((1 << b) < tiles) ++b;
    return b;
}

}  // namespace

extern "C" size_t cugs_sort_workspace_bytes(int64_t n) {
    if (n < 0) return 0;
    return carve_n(nullptr, n).bytes;
}

extern "C" size_t cugs_sort_pair_workspace_bytes(int64_t total_pairs) {
    if (total_pairs < 0) return 0;
    return carve_p(nullptr, total_pairs).bytes;
}

// Everything that does not depend on the pair count runs BEFORE the blocking read-back, so the
// device is busy (depth keys, the 4-pass depth sort, per-block pair sums and their scan) while the
// host waits for the 8-byte total - the reference idles on cumsum[-1].item() instead (sorting.cu:146).
extern "C" int cugs_sort_count_pairs(int64_t n, const float* means_2d, const float* depths,
                                     const int32_t* radii, const int32_t* tiles_touched, int width,
                                     int height, void* workspace, size_t workspace_bytes,
                                     int64_t* total_pairs_host, void* stream) {
    if (n < 0 || width < 0 || height < 0 || !total_pairs_host) return CUGS_EINVAL;
    *total_pairs_host = 0;
    if (n == 0) return 0;
    if (n > 2147483647ll) return CUGS_EOVERFLOW;
    if (!means_2d || !depths || !radii || !tiles_touched || !workspace) return CUGS_EINVAL;
    SortWsN ws = carve_n(workspace, n);
    if (workspace_bytes < ws.bytes) return CUGS_EWORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int ntx = (width + CUGS_TILE - 1) / CUGS_TILE, nty = (height + CUGS_TILE - 1) / CUGS_TILE;
    const uint32_t un = (uint32_t)n;

    // (1) stable sort of the Gaussians by depth bits (positive floats order as unsigned ints)
    hipLaunchKernelGGL(k_depth_keys, dim3(nblocks_for(n, CUGS_BLOCK)), dim3(CUGS_BLOCK), 0, st, un, depths,
                       means_2d, radii, tiles_touched, width, height, ntx, nty, ws.dkey[1]);
    CUGS_LAUNCH_CHECK();
    int rc;
    if ((rc = radix_pass<true>(ws.dkey[1], nullptr, un, 0, 8, ws.hist, ws.tot, ws.dkey[0], ws.dval[0], st))) return rc;
    if ((rc = radix_pass<false>(ws.dkey[0], ws.dval[0], un, 8, 8, ws.hist, ws.tot, ws.dkey[1], ws.dval[1], st))) return rc;
    if ((rc = radix_pass<false>(ws.dkey[1], ws.dval[1], un, 16, 8, ws.hist, ws.tot, ws.dkey[0], ws.dval[0], st))) return rc;
    if ((rc = radix_pass<false>(ws.dkey[0], ws.dval[0], un, 24, 8, ws.hist, ws.tot, ws.dkey[1], ws.dval[1], st))) return rc;

    // (2a) pair counts per 256-Gaussian block in depth order, their scan, and the grand total
    const uint32_t nfill = nblocks_for(n, FILL_CHUNK);
    hipLaunchKernelGGL(k_fill_blocksums, dim3(nfill), dim3(CUGS_BLOCK), 0, st, un, ws.dval[1], tiles_touched,
                       ws.blocksum);
    CUGS_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_scan_blocksums, dim3(1), dim3(CUGS_BLOCK), 0, st, ws.blocksum, nfill, ws.total);
    CUGS_LAUNCH_CHECK();

    unsigned long long host_total = 0;
    CUGS_RETURN_IF_HIP(hipMemcpyAsync(&host_total, ws.total, sizeof(host_total), hipMemcpyDeviceToHost, st));
    CUGS_RETURN_IF_HIP(hipStreamSynchronize(st));
    if (host_total > 2147483647ull) return CUGS_EOVERFLOW;   // the reference indexes pairs with int
    *total_pairs_host = (int64_t)host_total;
    return 0;
}

extern "C" int cugs_sort_pairs(int64_t n, int64_t total_pairs, const float* means_2d,
                               const float* depths, const int32_t* radii,
                               const int32_t* tiles_touched, int width, int height, void* workspace,
                               size_t workspace_bytes, void* pair_workspace, size_t pair_workspace_bytes,
                               uint64_t* keys_sorted, int32_t* values_sorted, int32_t* tile_ranges,
                               void* stream) {
    if (n < 0 || total_pairs < 0 || width < 0 || height < 0 || !tile_ranges) return CUGS_EINVAL;
    if (n > 2147483647ll || total_pairs > 2147483647ll) return CUGS_EOVERFLOW;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int ntx = (width + CUGS_TILE - 1) / CUGS_TILE, nty = (height + CUGS_TILE - 1) / CUGS_TILE;
    const int tiles = ntx * nty;
    if (tiles > 0)   // untouched tiles stay {0,0} (sorting.cu:216)
        CUGS_RETURN_IF_HIP(hipMemsetAsync(tile_ranges, 0, sizeof(int32_t) * 2 * (size_t)tiles, st));
    if (n == 0 || total_pairs == 0 || tiles == 0) return 0;      // sorting.cu:133-139,154-160
    if (!means_2d || !depths || !radii || !tiles_touched || !values_sorted || !workspace || !pair_workspace)
        return CUGS_EINVAL;
    SortWsN ws = carve_n(workspace, n);
    SortWsP wp = carve_p(pair_workspace, total_pairs);
    if (workspace_bytes < ws.bytes || pair_workspace_bytes < wp.bytes) return CUGS_EWORKSPACE;

    const uint32_t un = (uint32_t)n, up = (uint32_t)total_pairs;
    uint32_t* zero_pairs = reinterpret_cast<uint32_t*>(ws.total) + 4;
    CUGS_RETURN_IF_HIP(hipMemsetAsync(zero_pairs, 0, sizeof(uint32_t), st));
    const uint32_t* order = ws.dval[1];                 // left there by cugs_sort_count_pairs
    const uint32_t nfill = nblocks_for(n, FILL_CHUNK);

    // (2b) pairs in depth order; (3) stable sort by tile id, last pass landing in values_sorted
    const int bits = tile_bits(tiles);
    const int npass = (bits + 7) / 8;
    const int per = (bits + npass - 1) / npass;
    uint32_t* vals_final = reinterpret_cast<uint32_t*>(values_sorted);
    uint32_t* tk[2] = {wp.ptile[0], wp.ptile[1]};
    uint32_t* tv[2] = {wp.pidx[0], wp.pidx[1]};
    hipLaunchKernelGGL(k_fill_pairs, dim3(nfill), dim3(CUGS_BLOCK), 0, st, un, up, order, tiles_touched,
                       means_2d, radii, width, height, ntx, nty, ws.blocksum, tk[0], tv[0], zero_pairs);
    CUGS_LAUNCH_CHECK();
    int cur = 0, rc;
    for (int p = 0; p < npass; ++p) {
        const int shift = p * per;
        const int b = (bits - shift) < per ? (bits - shift) : per;
        uint32_t* vout = (p == npass - 1) ? vals_final : tv[cur ^ 1];
        if ((rc = radix_pass<false>(tk[cur], tv[cur], up, shift, b, wp.hist, ws.tot, tk[cur ^ 1], vout, st))) return rc;
        cur ^= 1;
    }
    hipLaunchKernelGGL(k_tile_ranges, dim3(nblocks_for(total_pairs, CUGS_BLOCK)), dim3(CUGS_BLOCK), 0, st,
                       up, tk[cur], values_sorted, depths, tile_ranges, keys_sorted, zero_pairs);
    CUGS_LAUNCH_CHECK();
    return 0;
}
