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
// 8 B x N x 3 passes (9-bit digits on the 27-bit offset of the depth bits from the near plane; 4 passes of 8 bits on
// the raw bits for views outside that range), (3) moves 8 B x P x ceil(log2(tiles)/8) passes (2 at 1080p) instead of
// 12 B x P x 8 passes.  Every pass is the same three kernels: per-workgroup digit histogram,
// per-digit row scan, stable scatter with wave64 ballot ranking.  All HBM-bound integer work.
//
// Views with many pairs per Gaussian (>= 13: dense scenes, close-ups), on images of up to 256 x 256 tiles, save
// the first of the pair-level passes: the pairs are EMITTED already ordered by tile column (k_col_emit) - which
// needs a histogram and a ranking per (Gaussian, column) instead of per pair - and one stable pass by tile row
// finishes (3).  Both routes give the same permutation (tests/test_gpu_parity.py runs each against the oracle).
#include "cugs_gaussian_math.h"

#include <atomic>
#include <cstdlib>

namespace {

constexpr int RADIX = 256;
// The depth sort of views whose depths lie in [near plane, ~13 000) - every view the reference's projection can
// produce in practice: it culls z <= 0.2 - runs THREE passes of 9 bits on the key's offset from the near plane's bit
// pattern instead of four passes of 8 bits on the raw float bits: positive floats order like their bit patterns, and
// [0.2, 13 107) spans 2^27 patterns.  A kernel boundary costs ~5 us on this part and a pass is three kernels.  The key
// kernel checks the range of every Gaussian that emits pairs; a view outside it is reported through the pair count
// (-1: "redo") and takes the four-pass route on the raw bits.
constexpr int RADIX_DEPTH = 512, DEPTH_BITS = CUGS_DEPTH_BITS;      // key range and base: cugs_gaussian_math.h (sort_record_of)
constexpr int IPT = 16;                           // items per thread
constexpr int CHUNK_MIN = CUGS_BLOCK * IPT;       // 4096 items per workgroup: depth sort; sizes the histogram buffers
#ifndef CUGS_PAIR_CHUNK_MULT
#define CUGS_PAIR_CHUNK_MULT 1
#endif
#ifndef CUGS_DEPTH_CHUNK
#define CUGS_DEPTH_CHUNK 4096
#endif
constexpr int CHUNK_DEPTH = CUGS_DEPTH_CHUNK;     // items per workgroup in the three 9-bit depth passes
static_assert(CHUNK_DEPTH <= CHUNK_MIN && CHUNK_MIN % CHUNK_DEPTH == 0, "the histogram buffers are carved for CHUNK_DEPTH blocks");
constexpr int CHUNK_PAIR = CUGS_PAIR_CHUNK_MULT * CHUNK_MIN;   // pair-level passes (8192 / 16384: 4 % / 30 % slower, profiles/README.md)
constexpr int FILL_CHUNK = CUGS_BLOCK;            // Gaussians per workgroup in scan/fill (one per thread)
#ifndef CUGS_COL_WAVES
#define CUGS_COL_WAVES 16
#endif
constexpr int COL_WAVES = CUGS_COL_WAVES;         // waves per workgroup of the column-ordered emission: the longer a
constexpr int COL_CHUNK = COL_WAVES * CUGS_WAVE;  // workgroup's run in each tile column, the fewer partial lines it writes
static_assert(COL_CHUNK % FILL_CHUNK == 0, "whole FILL_CHUNK blocks per column workgroup");

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline uint32_t nblocks_for(int64_t count, int chunk) { return (uint32_t)((count + chunk - 1) / chunk); }

// Two caller-owned scratch buffers.  The N-level one is filled by cugs_sort_count_pairs (depth order,
// scanned block sums, pair total) and read by cugs_sort_pairs; the pair-level one can only be sized
// once the pair count is known.
struct SortWsN {
    unsigned long long* total;   // [0] pair total (u64); dwords 4-5: zero-pair counter (quirk Q12) and its snapshot
    uint32_t* dkey[2];           // depth bits, ping-pong              [n]
    uint32_t* dval[2];           // Gaussian index, ping-pong          [n]
    int4* rect[2];               // {x0, y0, w | h << 16, tiles_touched} per Gaussian: [0] input order, [1] depth order
    uint32_t* prect[2];          // the same record packed into a dword (pack_rect), riding through the depth passes  [n]
    uint32_t* tot;               // [RADIX_DEPTH]
    uint32_t* blocksum;          // per FILL_CHUNK block pair counts   [nfill + 2]
    uint32_t* hist;              // [nblk_n][RADIX_DEPTH] block-major digit counts of the current depth pass
    uint32_t* sup;               // SUP_TABLES tables [nsb][RADIX_DEPTH]: the same counts summed per super-block, one table per pass
    uint32_t sup_entries;        // dwords in one table
    uint32_t* colhist;           // [RADIX][ncol] pairs per (tile column, COL_CHUNK block), as counted
    uint32_t* colscan;           // ... and scanned along each column's row
    uint32_t* bin_table;         // direct binning: [rows][tiles] pairs per (workgroup of the depth order, tile), then their prefix
    uint32_t* bin_ttot;          // [tiles] pairs per tile
    uint32_t* bin_tpre;          // [tiles] pairs of the earlier tiles of the tile's 64-tile chunk
    uint32_t* bin_csum;          // [chunks] pairs per 64-tile chunk
    uint32_t* bin_win;           // [BIN_WINDOWS_MAX] pairs per window of the scatter's workgroups (k_bin_scan; cleared by k_bin_count)
    uint32_t* bin_tbase;         // [tiles + 1] first real pair of each tile; [tiles] = pair total
    size_t bytes;
};
struct SortWsP {
    void* ptile[2];              // tile id per pair (u16 when the tile count allows, else u32), ping-pong [P]
    uint32_t* pidx[2];           // Gaussian index per pair            [P]
    uint32_t* hist;              // [nblk_p][RADIX] block-major
    uint32_t* sup;               // SUP_TABLES tables [nsb][RADIX], one per pair-level pass
    uint32_t sup_entries;
    size_t bytes;
};

// Scatter offsets without a scan kernel (round 3), for passes of up to SCANFREE_MAX_BLOCKS workgroups.  A radix pass
// needs, per workgroup b and digit d, the number of items with digit d in the workgroups before b, and the digit totals.
// Round 2 got them from a third kernel per pass (k_radix_scan_rows over a digit-major table): ~5 us of kernel boundary
// for a few microseconds of work, five times a frame.  Now the histogram kernel writes its counts BLOCK-major (one
// contiguous row per workgroup) and adds them into a per-super-block table (one row per SB workgroups; atomics on a
// table zeroed by an earlier kernel of the stream: SB adds per address), and every scatter workgroup sums, with coalesced
// row loads issued at its very start and shared out over all its threads: the super rows before its own (<= nblk / SB),
// the block rows of its own super-block before it (< SB), and all super rows for the digit totals.
// Same box, config 3: sort 0.2295 -> 0.2099 ms (with the packed rectangles riding along).  What it costs is the row
// sums at the head of every scatter workgroup: (nblk / SB + SB) / GRP loads per thread, GRP = threads per digit.  Beyond
// SCANFREE_MAX_LOADS of them the third kernel stays (6 M Gaussians: 43 per thread in the depth passes, sort +17 us; 40 M
// pairs: +70-100 us, or - with a third table level - thousands of atomics per address at ~15 ns each;
// profiles/r03_m_scanfree_ab.log).
constexpr int SUP_TABLES = 4;                     // passes that may follow one zeroing: 4 depth passes (general route) / 4 pair passes (32-bit tile ids)
constexpr uint32_t SCANFREE_MAX_BLOCKS = 4096u, SCANFREE_MAX_LOADS = 26u;
inline uint32_t sup_block(uint32_t nblk) { return nblk <= 512u ? 16u : 64u; }
// grp: threads per digit of the pass's scatter workgroups (NT >> digit bits)
inline bool scan_free(uint32_t nblk, uint32_t grp) {
    const uint32_t sb = sup_block(nblk);
    return nblk <= SCANFREE_MAX_BLOCKS && ((nblk + sb - 1u) / sb + sb) <= SCANFREE_MAX_LOADS * grp;
}
inline uint32_t sup_rows(uint32_t nblk, uint32_t grp) { return scan_free(nblk, grp) ? (nblk + sup_block(nblk) - 1u) / sup_block(nblk) : 0u; }
// most rows any nblk' <= nblk can need: the workspace is carved for a capacity
inline uint32_t sup_rows_bound(uint32_t nblk) { return (nblk < SCANFREE_MAX_BLOCKS ? nblk : SCANFREE_MAX_BLOCKS) / 16u + 4u; }
// dwords of ONE table for a pass over nblk workgroups with rows of rdx digits: the tables of a sort lie back to back at
// this stride, and that much (x the number of passes) is what the zeroing kernel clears
inline uint32_t sup_used(uint32_t nblk, int rdx, uint32_t grp) { return (uint32_t)rdx * sup_rows(nblk, grp); }

// Direct binning (k_bin_count / k_bin_scan / k_bin_scatter further down): table geometry, needed by the workspace carving.
constexpr int BIN_TILES_CAP = 10240;              // tiles of the image (k_bin_count's LDS row)
constexpr uint32_t BIN_GROUP = 4096u, BIN_ROWS_MAX = 512u;
constexpr uint32_t BIN_WINDOWS_MAX = 64u;         // windows (8 tile rows x <= 64 tile columns) whose weights order the scatter's workgroups
// Gaussians per table row (and per workgroup of k_bin_count): measured best of 2048 ... 16384 at 1 M Gaussians.  The route is
// taken for up to BIN_ROWS_MAX rows = 2 M Gaussians: at 6 M (40 M pairs) it ties with the radix passes (0.83 ms both,
// profiles/r03_s_direct_binning.log), which stay in charge there.
inline uint32_t bin_group() {
#ifdef CUGS_DEV
    if (const char* e = std::getenv("CUGS_BIN_GROUP")) return (uint32_t)std::atoi(e);   // development build: sweeps (tools/ablate_bin.py)
#endif
    return BIN_GROUP;
}
inline uint32_t bin_rows_max() {
#ifdef CUGS_DEV
    if (const char* e = std::getenv("CUGS_BIN_ROWS_MAX")) return (uint32_t)std::atoi(e);   // development build: sweeps
#endif
    return BIN_ROWS_MAX;
}
inline bool bin_route_n(int64_t n) { return n <= (int64_t)bin_group() * bin_rows_max(); }
inline uint32_t bin_rows(uint32_t n) { return (n + bin_group() - 1u) / bin_group(); }
inline uint32_t bin_table_rows(int64_t n) { return n > 0 && bin_route_n(n) ? bin_rows((uint32_t)n) : 1u; }

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
    for (int i = 0; i < 2; ++i) w.rect[i] = c.take<int4>((size_t)n);
    for (int i = 0; i < 2; ++i) w.prect[i] = c.take<uint32_t>((size_t)n);
    w.tot = c.take<uint32_t>(RADIX_DEPTH);
    w.blocksum = c.take<uint32_t>((size_t)nblocks_for(n, FILL_CHUNK) + 2);
    w.hist = c.take<uint32_t>((size_t)RADIX_DEPTH * (nblocks_for(n, CHUNK_DEPTH) + 1));
    w.sup_entries = (uint32_t)RADIX_DEPTH * sup_rows_bound(nblocks_for(n, CHUNK_DEPTH));
    w.sup = c.take<uint32_t>((size_t)SUP_TABLES * w.sup_entries);
    w.colhist = c.take<uint32_t>((size_t)RADIX * (nblocks_for(n, COL_CHUNK) + 1));
    w.colscan = c.take<uint32_t>((size_t)RADIX * (nblocks_for(n, COL_CHUNK) + 1));
    w.bin_table = c.take<uint32_t>((size_t)bin_table_rows(n) * BIN_TILES_CAP);
    w.bin_ttot = c.take<uint32_t>(BIN_TILES_CAP);
    w.bin_tpre = c.take<uint32_t>(BIN_TILES_CAP);
    w.bin_csum = c.take<uint32_t>(BIN_TILES_CAP / 64 + 4);
    w.bin_win = c.take<uint32_t>(BIN_WINDOWS_MAX);
    w.bin_tbase = c.take<uint32_t>(BIN_TILES_CAP + 1);
    w.bytes = c.off;
    return w;
}
SortWsP carve_p(void* base, int64_t pairs) {
    Carver c{static_cast<char*>(base)};
    SortWsP w;
    for (int i = 0; i < 2; ++i) w.ptile[i] = c.take<uint32_t>((size_t)pairs);     // sized for the u32 case
    for (int i = 0; i < 2; ++i) w.pidx[i] = c.take<uint32_t>((size_t)pairs);
    w.hist = c.take<uint32_t>((size_t)RADIX * (nblocks_for(pairs, CHUNK_MIN) + 1));
    w.sup_entries = (uint32_t)RADIX * sup_rows_bound(nblocks_for(pairs, CHUNK_MIN));
    w.sup = c.take<uint32_t>((size_t)SUP_TABLES * w.sup_entries);
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

// Exclusive scan of one value per thread over a workgroup of NW waves; *total = workgroup sum.
// s_tmp: NW dwords of LDS.  Contains two barriers.
template <int NW = 4>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* s_tmp, uint32_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = wave_inclusive_scan(v);
    if (lane == 63) s_tmp[wave] = inc;
    __syncthreads();
    uint32_t base = 0, sum = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const uint32_t t = s_tmp[w];
        base += (w < wave) ? t : 0u;
        sum += t;
    }
    if (total) *total = sum;
    __syncthreads();
    return base + inc - v;
}

// Reference quirk Q12: a splat whose tile rectangle is empty in both axes still has
// tiles_touched = (negative) x (negative) > 0 (projection.cu:187-188); k_fill_sort_pairs writes
// nothing for it and its reserved slots keep the zero-initialised (key 0, value 0) pairs
// (sorting.cu:166-167), which sort to the front of tile 0.  Reproduced here by giving such a
// Gaussian the depth key 0 (its pairs are emitted first) and emitting (tile 0, Gaussian 0).
//
// One sequential read of the projection outputs produces the depth key and a 16-byte tile-rectangle
// record per Gaussian, so that the depth-ordered stages gather ONE record per Gaussian instead of
// tiles_touched, radius and mean separately (random 4-8 byte gathers were what bounded k_fill_pairs).
__global__ __launch_bounds__(CUGS_BLOCK) void k_depth_keys_rect(uint32_t n, const float* __restrict__ depths,
                                                                const float* __restrict__ means_2d,
                                                                const int32_t* __restrict__ radii,
                                                                const int32_t* __restrict__ tiles, int img_w,
                                                                int img_h, int ntx, int nty,
                                                                uint32_t* __restrict__ keys,
                                                                int4* __restrict__ rect,
                                                                uint32_t* __restrict__ range_flag,
                                                                uint32_t* __restrict__ zero, uint32_t nzero) {
    const uint32_t i = blockIdx.x * CUGS_BLOCK + threadIdx.x;
    for (uint32_t z = i; z < nzero; z += gridDim.x * CUGS_BLOCK) zero[z] = 0u;     // the depth passes' super tables
    if (i >= n) return;
    const int t = tiles[i];
    const int radius = t > 0 ? radii[i] : 0;
    TileRect tr{0, 0, 0, 0};
    if (t > 0 && radius > 0) tr = tile_rect_of(means_2d[i * 2 + 0], means_2d[i * 2 + 1], radius, img_w, img_h, ntx, nty);
    bool bad;
    const SortRecord r = sort_record_of(depths[i], t, radius, tr, range_flag != nullptr, &bad);
    if (bad) atomicOr(range_flag, 1u);
    keys[i] = r.key;
    rect[i] = r.rect;
}

// Pair-level kernels take their item count either exactly (dev_count == nullptr: `count`) or, for the
// predicted-capacity path (cugs_sort_pairs_predicted), as min(*dev_count, count) with `count` the capacity
// of the buffers - the host has not read the total yet.
__device__ __forceinline__ uint32_t live_count(uint32_t count, const unsigned long long* __restrict__ dev_count) {
    if (!dev_count) return count;
    const unsigned long long t = *dev_count;
    return t < (unsigned long long)count ? (uint32_t)t : count;
}

// ctl: when given, block 0 hands the Q12 counter k_fill_pairs has finished adding to (ctl[0]) over to
// k_tile_ranges (ctl[1]) and re-arms it, so that cugs_sort_pairs may be repeated on one count.
template <typename K, int NT, int CHUNK, int RDX = RADIX>
__global__ __launch_bounds__(NT) void k_radix_hist(const K* __restrict__ keys, uint32_t count_or_cap,
                                                   const unsigned long long* __restrict__ dev_count, int shift,
                                                   uint32_t mask, uint32_t* __restrict__ hist, uint32_t nblk,
                                                   uint32_t* __restrict__ ctl, uint32_t* __restrict__ sup, uint32_t sb) {
    const uint32_t count = live_count(count_or_cap, dev_count);
    constexpr int PER = CHUNK / NT;                           // consecutive keys per thread (order is irrelevant here)
    constexpr int NWORDS = PER * (int)sizeof(K) / 4;          // ... fetched as dwords in 16- or 8-byte loads
    static_assert(NWORDS >= 2 && NWORDS * 4 == PER * (int)sizeof(K), "whole 8-byte loads per thread");
    static_assert(NT >= RDX, "one thread per digit");
    __shared__ uint32_t s_cnt[RDX];
    if (threadIdx.x < RDX) s_cnt[threadIdx.x] = 0;
    if (ctl && blockIdx.x == 0 && threadIdx.x == 0) { ctl[1] = ctl[0]; ctl[0] = 0u; }
    __syncthreads();
    const uint32_t bbase = blockIdx.x * CHUNK;
    if (bbase + CHUNK <= count) {
        uint32_t wds[NWORDS];
        if constexpr (NWORDS % 4 == 0) {
            const uint4* src = reinterpret_cast<const uint4*>(keys + bbase + threadIdx.x * PER);
#pragma unroll
            for (int v = 0; v < NWORDS / 4; ++v) {
                const uint4 q = src[v];
                wds[4 * v] = q.x; wds[4 * v + 1] = q.y; wds[4 * v + 2] = q.z; wds[4 * v + 3] = q.w;
            }
        } else {
            const uint2* src = reinterpret_cast<const uint2*>(keys + bbase + threadIdx.x * PER);
#pragma unroll
            for (int v = 0; v < NWORDS / 2; ++v) {
                const uint2 q = src[v];
                wds[2 * v] = q.x; wds[2 * v + 1] = q.y;
            }
        }
#pragma unroll
        for (int c = 0; c < NWORDS; ++c) {
            if (sizeof(K) == 4) {
                atomicAdd(&s_cnt[(wds[c] >> shift) & mask], 1u);
            } else {
                atomicAdd(&s_cnt[((wds[c] & 0xFFFFu) >> shift) & mask], 1u);
                atomicAdd(&s_cnt[((wds[c] >> 16) >> shift) & mask], 1u);
            }
        }
    } else {
        for (uint32_t i = bbase + threadIdx.x; i < count; i += NT)
            atomicAdd(&s_cnt[((uint32_t)keys[i] >> shift) & mask], 1u);
    }
    __syncthreads();
    if (threadIdx.x < RDX) {
        const uint32_t c = s_cnt[threadIdx.x];
        if (sup) {                                                              // scan-free pass (kernel-uniform)
            hist[(size_t)blockIdx.x * RDX + threadIdx.x] = c;                   // block-major: one contiguous row
            if (c) atomicAdd(&sup[(size_t)(blockIdx.x / sb) * RDX + threadIdx.x], c);
        } else {
            hist[(size_t)threadIdx.x * nblk + blockIdx.x] = c;                  // digit-major, for k_radix_scan_rows
        }
    }
}

// Block d: exclusive scan of row d of hist into `out` (may be hist itself); tot[d] = row sum.
__global__ __launch_bounds__(CUGS_BLOCK) void k_radix_scan_rows(const uint32_t* hist, uint32_t* out,
                                                                uint32_t nblk, uint32_t* __restrict__ tot) {
    __shared__ uint32_t s_tmp[4];
    constexpr int PER = 8;                                   // consecutive entries per thread: 2048 per iteration
    const uint32_t* row = hist + (size_t)blockIdx.x * nblk;
    uint32_t* orow = out + (size_t)blockIdx.x * nblk;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nblk; base += CUGS_BLOCK * PER) {
        const uint32_t i0 = base + threadIdx.x * PER;
        uint32_t v[PER], sum = 0;
#pragma unroll
        for (int e = 0; e < PER; ++e) { v[e] = (i0 + e < nblk) ? row[i0 + e] : 0u; sum += v[e]; }
        uint32_t total;
        uint32_t run = carry + block_exclusive_scan(sum, s_tmp, &total);
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            if (i0 + e < nblk) orow[i0 + e] = run;
            run += v[e];
        }
        carry += total;
    }
    if (threadIdx.x == 0) tot[blockIdx.x] = carry;
}

// Stable scatter.  Ranking: each wave owns a contiguous 1024-item slice and walks it in rounds of
// 64; in a round the lanes holding the same digit find each other with one ballot per digit bit (match-any), the
// rank is the popcount below the lane, and the group's highest lane advances the wave's running
// base in LDS.  The (key, value) pairs are first placed at their position in the workgroup's LOCALLY
// sorted order in LDS and then streamed out, so that consecutive lanes write consecutive global
// addresses inside each digit's run (4 B items scattered straight to 128-256 buckets cost ~2x).
// IOTA: first pass of the depth sort, which generates the Gaussian index instead of reading a value
// array.  NB: digit width (the match-any needs one ballot per digit bit).  NT: threads per workgroup -
// 256 for the pair-level passes (thousands of workgroups), 1024 for the depth sort, whose 4096-item
// chunks are too few to fill the chip with 4 waves each.
// V2: a second dword per item travels with the first (the depth sort's packed tile rectangle, pack_rect).
template <typename K, bool IOTA, int NB, int NT, bool ARANK, int CHUNK, int RDX = RADIX, bool V2 = false>
__global__ __launch_bounds__(NT) void k_radix_scatter(
    const K* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint32_t count_or_cap,
    const unsigned long long* __restrict__ dev_count, int shift, uint32_t mask_rt, const uint32_t* __restrict__ hist,
    const uint32_t* __restrict__ sup, uint32_t sb, const uint32_t* __restrict__ tot, uint32_t nblk,
    K* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
    const uint32_t* __restrict__ vals2_in = nullptr, uint32_t* __restrict__ vals2_out = nullptr) {
    const uint32_t mask = ARANK ? mask_rt : ((1u << NB) - 1u);       // the ballot ranking needs the width at compile time
    const uint32_t count = live_count(count_or_cap, dev_count);
    if (blockIdx.x * CHUNK >= count) return;              // chunks beyond the live items (capacity path): nothing to move
    constexpr int NW = NT / CUGS_WAVE;                    // waves
    constexpr int PER = CHUNK / NT;                       // items per thread
    constexpr int SLICE = CUGS_WAVE * PER;                // contiguous items per wave
    static_assert(NT >= RDX && (1 << NB) <= RDX, "one thread per digit");
    __shared__ uint32_t s_lbase[NW][RDX];      // per (wave, digit): count, then running LOCAL position
    __shared__ uint32_t s_lstart[RDX];         // first local position of digit d
    __shared__ uint32_t s_gbase[RDX];          // first global position of this workgroup's digit-d run
    __shared__ K s_key[CHUNK];
    __shared__ uint32_t s_val[CHUNK];
    __shared__ uint32_t s_val2[V2 ? CHUNK : 1];
    __shared__ uint32_t s_tmp[NW];
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t bbase = blockIdx.x * CHUNK;
    const uint32_t wbase = bbase + wave * SLICE;
    const uint32_t count_blk = min((uint32_t)CHUNK, count - bbase);

    // Global offsets of this workgroup's digit runs (see the note at sup_block): thread t takes digit t % RDX and every
    // (NT / RDX)-th row, partial sums meet in LDS further down.  The loads go out first thing and are consumed after the
    // local ranking.  Rows of blocks beyond the live count hold zeros (their histogram workgroups wrote them).
    constexpr int ND = ARANK ? RDX : (1 << NB);              // digits that can be non-zero (the rows are RDX wide)
    constexpr int GRP = NT / ND;                             // threads per digit
    uint32_t pre = 0u, totd = 0u;
    if (sup) {                                               // scan-free pass (kernel-uniform)
        const uint32_t d = tid % ND, part = tid / ND;
        const uint32_t mysb = blockIdx.x / sb, nsb = (nblk + sb - 1u) / sb;
#pragma unroll 4
        for (uint32_t r = part; r < nsb; r += GRP) {             // super rows: totals, and the rows before mine
            const uint32_t v = sup[(size_t)r * RDX + d];
            totd += v;
            pre += (r < mysb) ? v : 0u;
        }
#pragma unroll 4
        for (uint32_t b2 = mysb * sb + part; b2 < blockIdx.x; b2 += GRP) pre += hist[(size_t)b2 * RDX + d];   // my super-block
    } else if (tid < RDX) {                                  // k_radix_scan_rows has scanned the digit-major table
        pre = hist[(size_t)tid * nblk + blockIdx.x];
        totd = tot[tid];
    }

    for (uint32_t e = tid; e < NW * RDX; e += NT) (&s_lbase[0][0])[e] = 0;
    if (tid < RDX) { s_gbase[tid] = 0u; s_lstart[tid] = 0u; }     // accumulators of the partial sums (pre, totals)
    __syncthreads();
    if (sup && GRP > 1) {
        if (pre) atomicAdd(&s_gbase[tid % ND], pre);
        if (totd) atomicAdd(&s_lstart[tid % ND], totd);
    } else if (tid < RDX) {
        s_gbase[tid] = pre; s_lstart[tid] = totd;
    }

    uint32_t k[PER], v[PER], v2[V2 ? PER : 1];
#pragma unroll
    for (int r = 0; r < PER; ++r) {
        uint32_t i = wbase + r * CUGS_WAVE + lane;
        bool ok = i < count;
        // last use of this pass's input: streamed, so that it does not evict the output being written for the next pass
        k[r] = ok ? (uint32_t)__builtin_nontemporal_load(keys_in + i) : 0xFFFFFFFFu;
        v[r] = ok ? (IOTA ? i : __builtin_nontemporal_load(vals_in + i)) : 0u;
        if constexpr (V2) v2[r] = ok ? __builtin_nontemporal_load(vals2_in + i) : 0u;
        if (ok) atomicAdd(&s_lbase[wave][(k[r] >> shift) & mask], 1u);
    }
    __syncthreads();

    {   // digit d = tid (threads beyond the radix only take part in the barriers)
        const bool dig = tid < RDX;
        uint32_t cnt = 0;
        if (dig) {
#pragma unroll
            for (int w = 0; w < NW; ++w) cnt += s_lbase[w][tid];
        }
        const uint32_t before = dig ? s_gbase[tid] : 0u;                                           // digit d in the workgroups before this one
        const uint32_t dig_base = block_exclusive_scan<NW>(dig ? s_lstart[tid] : 0u, s_tmp, nullptr);   // global digit start
        const uint32_t lstart = block_exclusive_scan<NW>(cnt, s_tmp, nullptr);                     // local digit start
        if (dig) {
            s_gbase[tid] = dig_base + before;
            s_lstart[tid] = lstart;
            uint32_t run = lstart;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const uint32_t c = s_lbase[w][tid];
                s_lbase[w][tid] = run;
                run += c;
            }
        }
    }
    __syncthreads();

    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < PER; ++r) {
        const uint32_t i = wbase + r * CUGS_WAVE + lane;
        const bool ok = i < count;
        const uint32_t d = (k[r] >> shift) & mask;
        if constexpr (ARANK) {
            // One LDS atomic with return per item: the hardware serves the lanes of a wave instruction that hit the
            // same counter in ascending lane order and a wave's LDS instructions in issue order, so the returned
            // values ARE the stable positions.  That ordering is not in the ISA manual: it is verified on the device
            // before this path is ever selected (k_probe_lds_order), and the ballot path below stays as the fallback.
            if (ok) {
                const uint32_t pos = atomicAdd(&s_lbase[wave][d], 1u);
                s_key[pos] = (K)k[r];
                s_val[pos] = v[r];
                if constexpr (V2) s_val2[pos] = v2[r];
            }
            continue;
        }
        unsigned long long peers = __ballot(ok);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        if (ok) {
            const uint32_t base = s_lbase[wave][d];
            const uint32_t pos = base + __popcll(peers & lt_mask);
            s_key[pos] = (K)k[r];
            s_val[pos] = v[r];
            if constexpr (V2) s_val2[pos] = v2[r];
            if ((peers >> lane) == 1ull) s_lbase[wave][d] = base + __popcll(peers);
        }
    }
    __syncthreads();

#pragma unroll
    for (int r = 0; r < PER; ++r) {
        const uint32_t j = r * NT + tid;
        if (j < count_blk) {
            const K key = s_key[j];
            const uint32_t d = ((uint32_t)key >> shift) & mask;
            const uint32_t dst = s_gbase[d] + (j - s_lstart[d]);
            keys_out[dst] = key;
            vals_out[dst] = s_val[j];
            if constexpr (V2) vals2_out[dst] = s_val2[j];
        }
    }
}

// ------------------------------------------------------------------------------------
// Pair emission in depth order
// ------------------------------------------------------------------------------------
// Column-ordered emission, step 1: the pairs of each COL_CHUNK block of the depth order per tile column
// (colhist[column][block], digit-major like the radix histograms): a Gaussian adds its rectangle height to each
// column it covers; slots the reference leaves at zero (quirk Q12) count for column 0, where their
// (tile 0, Gaussian 0) pairs go.  One sequential read of the depth-ordered rectangle records.
__global__ __launch_bounds__(COL_CHUNK) void k_col_hist(uint32_t n, const int4* __restrict__ rect_sorted,
                                                        uint32_t* __restrict__ colhist, uint32_t ncol) {
    __shared__ uint32_t s_col[RADIX];
    const uint32_t tid = threadIdx.x;
    if (tid < RADIX) s_col[tid] = 0u;
    __syncthreads();
    const uint32_t i = blockIdx.x * COL_CHUNK + tid;
    if (i < n) {
        const int4 r = rect_sorted[i];
        const uint32_t t = (uint32_t)r.w;
        const int w = r.z & 0xFFFF, h = r.z >> 16;
        if ((uint32_t)(w * h) < t) atomicAdd(&s_col[0], t - (uint32_t)(w * h));
        for (int c = 0; c < w; ++c) atomicAdd(&s_col[r.x + c], (uint32_t)h);
    }
    __syncthreads();
    if (tid < RADIX) colhist[(size_t)tid * ncol + blockIdx.x] = s_col[tid];
}

// prect_sorted (when given): the packed rectangles arrive IN depth order (they rode through the passes): a sequential
// read instead of the gather.
__global__ __launch_bounds__(CUGS_BLOCK) void k_fill_blocksums(uint32_t n,
                                                               const uint32_t* __restrict__ order,
                                                               const int4* __restrict__ rect,
                                                               int4* __restrict__ rect_sorted,
                                                               uint32_t* __restrict__ blocksum,
                                                               const uint32_t* __restrict__ prect_sorted) {
    __shared__ uint32_t s_tmp[4];
    const uint32_t i = blockIdx.x * FILL_CHUNK + threadIdx.x;
    uint32_t acc = 0u;
    if (i < n) {
        const int4 r = prect_sorted ? unpack_rect(__builtin_nontemporal_load(prect_sorted + i))
                                    : rect[order[i]];             // the one gather per Gaussian
        rect_sorted[i] = r;
        acc = (uint32_t)r.w;
    }
    uint32_t total;
    block_exclusive_scan(acc, s_tmp, &total);
    if (threadIdx.x == 0) blocksum[blockIdx.x] = total;
}

// Single workgroup: exclusive scan of blocksum[0..nb) in place; *total = the 64-bit grand total =
// sum(tiles_touched), the reference's cumsum[-1].item() (sorting.cu:145-146).  16 consecutive entries per
// thread, so 16384 entries cost two barriers.  Also arms the Q12 counter (ctl[0] = 0).
// range_flag (three-pass depth sort): non-zero = some depth key lay outside the range that route covers, the order
// is not valid: the device-side total becomes 0 (every pair-level kernel then does nothing), the host-visible one
// (total[1], total_mapped) -1, and the flag is re-armed.
constexpr int SCAN_NT = 1024;                     // one workgroup, on the critical path of the pair count: as wide as it gets
__global__ __launch_bounds__(SCAN_NT) void k_scan_blocksums(uint32_t* __restrict__ blocksum, uint32_t nb,
                                                            unsigned long long* __restrict__ total,
                                                            uint32_t* __restrict__ ctl,
                                                            unsigned long long* __restrict__ total_mapped,
                                                            uint32_t* __restrict__ range_flag) {
    __shared__ uint32_t s_tmp[SCAN_NT / CUGS_WAVE];
    constexpr int PER = 16;
    unsigned long long carry = 0;
    for (uint32_t base = 0; base < nb; base += SCAN_NT * PER) {
        const uint32_t i0 = base + threadIdx.x * PER;
        uint32_t v[PER], sum = 0;
#pragma unroll
        for (int e = 0; e < PER; ++e) { v[e] = (i0 + e < nb) ? blocksum[i0 + e] : 0u; sum += v[e]; }
        uint32_t chunk_total;
        uint32_t run = (uint32_t)carry + block_exclusive_scan<SCAN_NT / CUGS_WAVE>(sum, s_tmp, &chunk_total);
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            if (i0 + e < nb) blocksum[i0 + e] = run;      // valid whenever the total fits int32 (checked on the host)
            run += v[e];
        }
        carry += chunk_total;
    }
    if (threadIdx.x == 0) {
        bool bad = false;
        if (range_flag) { bad = *range_flag != 0u; *range_flag = 0u; }
        const unsigned long long host_total = bad ? ~0ull : carry;
        total[0] = bad ? 0ull : carry;
        total[1] = host_total;
        ctl[0] = 0u;
        // the caller's pinned host variable, when it is mapped into the device's address space: one store here
        // instead of a copy kernel on the critical path of the predicted-capacity sort (~5 us)
        if (total_mapped) __hip_atomic_store(total_mapped, host_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// k_fill_sort_pairs (sorting.cu:30-72), walked in depth order; only the tile id and the index are
// stored (the depth half of the key is implied by the order).  One Gaussian per thread computes its
// rectangle and its offset (workgroup scan); each wave then emits the pairs of its own 64 Gaussians
// COOPERATIVELY, 256 output slots at a time: the Gaussians whose span starts inside the window stamp
// their lane number at that slot, a running maximum over the window (4 consecutive slots per lane + a
// wave scan) turns the stamps into an owner per slot, and the slots are then written lane-strided -
// consecutive lanes write consecutive pairs, and a splat covering thousands of tiles does not
// serialise one thread.  No workgroup barrier inside the loop.
// Housekeeping shared out over the grid: the {0,0} ranges of untouched tiles (sorting.cu:216).
template <typename K>
__global__ __launch_bounds__(CUGS_BLOCK) void k_fill_pairs(
    uint32_t n, uint32_t pairs_or_cap, const unsigned long long* __restrict__ dev_count,
    const uint32_t* __restrict__ order, const int4* __restrict__ rect_sorted, int ntx,
    const uint32_t* __restrict__ blocksum, K* __restrict__ ptile, uint32_t* __restrict__ pidx,
    uint32_t* __restrict__ zero_pairs, int32_t* __restrict__ tile_ranges, uint32_t range_dwords,
    uint32_t* __restrict__ sup_zero, uint32_t nsup) {
    constexpr int WIN = 4 * CUGS_WAVE;                               // output slots per wave iteration
    __shared__ uint32_t s_tmp[4];
    __shared__ uint32_t s_off[CUGS_BLOCK + 1];
    __shared__ int4 s_info[CUGS_BLOCK];                              // {Gaussian, x0, y0, rect width}
    __shared__ int s_cnt[CUGS_BLOCK];
    __shared__ uint4 s_own[4][CUGS_WAVE];                            // per wave: owner lane of each window slot
    const uint32_t total_pairs = live_count(pairs_or_cap, dev_count);
    for (uint32_t z = blockIdx.x * CUGS_BLOCK + threadIdx.x; z < range_dwords; z += gridDim.x * CUGS_BLOCK)
        tile_ranges[z] = 0;
    for (uint32_t z = blockIdx.x * CUGS_BLOCK + threadIdx.x; z < nsup; z += gridDim.x * CUGS_BLOCK)
        sup_zero[z] = 0u;                                            // the pair passes' super tables (radix_pass)
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t i = blockIdx.x * FILL_CHUNK + tid;
    uint32_t g = 0, t = 0;
    int x0 = 0, y0 = 0, w = 0, real = 0;          // real = pairs the reference's loops would write
    if (i < n) {
        g = __builtin_nontemporal_load(order + i);                     // last use of both streams
        typedef int v4i_ __attribute__((ext_vector_type(4)));
        const v4i_ r = __builtin_nontemporal_load(reinterpret_cast<const v4i_*>(rect_sorted) + i);
        t = (uint32_t)r.w;
        x0 = r.x; y0 = r.y; w = r.z & 0xFFFF;
        real = w * (r.z >> 16);
        if ((uint32_t)real < t) atomicAdd(zero_pairs, t - (uint32_t)real);       // quirk Q12 slots (rare)
    }
    uint32_t blk_total;
    const uint32_t off = block_exclusive_scan(t, s_tmp, &blk_total);
    s_off[tid] = off;
    s_info[tid] = make_int4((int)g, x0, y0, w);
    s_cnt[tid] = real;
    if (tid == 0) s_off[CUGS_BLOCK] = blk_total;
    __syncthreads();

    const uint32_t out_base = blocksum[blockIdx.x];
    const uint32_t wstart = s_off[wave * CUGS_WAVE], wend = s_off[wave * CUGS_WAVE + CUGS_WAVE];
    uint4* own4 = s_own[wave];
    const uint32_t* own = reinterpret_cast<const uint32_t*>(own4);
    uint32_t carry = 0;                                              // owner of the slot before the window
    for (uint32_t base = wstart; base < wend; base += WIN) {
        own4[lane] = make_uint4(0u, 0u, 0u, 0u);
        __builtin_amdgcn_wave_barrier();
        if (t > 0 && off >= base && off - base < (uint32_t)WIN) reinterpret_cast<uint32_t*>(own4)[off - base] = lane;
        __builtin_amdgcn_wave_barrier();
        uint4 a = own4[lane];
        a.y = max(a.x, a.y); a.z = max(a.y, a.z); a.w = max(a.z, a.w);
        uint32_t inc = a.w;                                          // inclusive running maximum over lanes
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(inc, d);
            if ((int)lane >= d) inc = max(inc, o);
        }
        uint32_t pre = __shfl_up(inc, 1);
        pre = max(lane == 0 ? 0u : pre, carry);
        a.x = max(a.x, pre); a.y = max(a.y, pre); a.z = max(a.z, pre); a.w = max(a.w, pre);
        carry = __shfl(a.w, 63);
        own4[lane] = a;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t k = base + e * CUGS_WAVE + lane;
            if (k < wend) {
                const uint32_t j = wave * CUGS_WAVE + own[e * CUGS_WAVE + lane];
                const uint32_t local = k - s_off[j];
                uint32_t tile = 0u, idx = 0u;                        // slots the reference leaves at zero (Q12)
                if ((int)local < s_cnt[j]) {
                    const int4 info = s_info[j];
                    const int row = (int)local / info.w;
                    tile = (uint32_t)((info.z + row) * ntx + info.y + ((int)local - row * info.w));
                    idx = (uint32_t)info.x;
                }
                const uint32_t dst = out_base + k;
                if (dst < total_pairs) {                             // never write past the buffers
                    ptile[dst] = (K)tile;
                    pidx[dst] = idx;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Pair emission ordered by TILE COLUMN (then depth, then row): the first pass of the stable sort by tile id,
// done while the pairs are generated.  Key written per pair: (row << 8 | column), images of <= 256 x 256 tiles.
// The unit is an ITEM = (Gaussian, column it covers), worth `height` consecutive pairs - ~3x fewer items than
// pairs at 1080p.  A workgroup takes COL_CHUNK Gaussians of the depth order and
//   (b) sets, per column, one bit per Gaussian that covers it (LDS bit matrix; OR commutes, so no ordering issue);
//   (c) counts each column's bits (prefix per 32-Gaussian word) and scans the counts over the columns;
//   (d) drops each item's record at  column start + set bits below its Gaussian  - the items are now sorted by
//       (column, depth) - with its height beside it;
//   (e) scans the heights: the local slot of every item's first pair, and per column the offset between local
//       slots and the column's run in the output (column start + pairs of the workgroups before this one, from
//       the scanned column histogram of k_col_hist);
//   (g) streams the pairs out, 64 sorted items per wave round (rounds handed out by an LDS counter), lane = item:
//       neighbouring lanes own neighbouring runs of the output, so the `height` store instructions of a round
//       complete each other's cache lines (writing from UNSORTED items cost 2x the HBM write requests; a slot-
//       parallel loop with a 6-step owner search per slot was bound by its ~136 instructions per 64 pairs).
// LDS holds ICAP items; a workgroup with more (dense views) works in batches of whole Gaussians.
// Slots the reference's loops leave at zero (tiles_touched beyond the w x h pairs of the rectangle: the Q12
// Gaussians, whose rectangle is empty) are ONE more item of the Gaussian worth that many (tile 0, Gaussian 0)
// pairs, in a pseudo column ordered before column 0 and sharing its run.  Q12 Gaussians are first in depth
// order, hence first in column 0, hence (row pass) first in tile 0, where the reference's zero pairs sort to.
#ifdef CUGS_DEV
__device__ unsigned long long g_emit_prof[16];     // development build: 100 MHz ticks per phase of k_col_emit, summed over workgroups
#define EMIT_TICK(slot) do { if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); \
        atomicAdd(&g_emit_prof[slot], now_ - tick_); tick_ = now_; } } while (0)
#else
#define EMIT_TICK(slot) do { } while (0)
#endif
__global__ __launch_bounds__(COL_CHUNK) void k_col_emit(
    uint32_t n, uint32_t pairs_or_cap, const unsigned long long* __restrict__ dev_count,
    const uint32_t* __restrict__ order, const int4* __restrict__ rect_sorted,
    const uint32_t* __restrict__ colscan, const uint32_t* __restrict__ coltot, uint32_t nblk,
    uint16_t* __restrict__ ptile, uint32_t* __restrict__ pidx,
    uint32_t* __restrict__ zero_pairs, int32_t* __restrict__ tile_ranges, uint32_t range_dwords,
    uint32_t* __restrict__ sup_zero, uint32_t nsup) {
    constexpr int NT = COL_CHUNK, NW = COL_WAVES;
    for (uint32_t z = blockIdx.x * COL_CHUNK + threadIdx.x; z < nsup; z += gridDim.x * COL_CHUNK)
        sup_zero[z] = 0u;                                            // the row pass's super table (radix_pass)
    constexpr int NWORD = NT / 32;                                   // bit-matrix words per column
    constexpr int NC = RADIX + 1, ZCOL = RADIX;                      // tile columns + the zero-slot pseudo column
    constexpr int NCP = RADIX + 4;
    constexpr int PER = 6;                                           // staged items per thread
    constexpr int ICAP = PER * NT;
    constexpr int IWIN = ICAP - (RADIX + 1);                         // a batch: the Gaussians whose first item is in one window
    static_assert(NT >= NC, "one thread per column");
    __shared__ uint32_t s_cover[NWORD][NCP];
    __shared__ uint16_t s_wpre[NWORD][NCP];
    __shared__ uint32_t s_istart[NCP];                               // first sorted item of each column (this batch)
    __shared__ uint32_t s_gpos[NCP];                                 // where the workgroup's next pair of each column goes
    __shared__ uint32_t s_delta[NCP];                                // output position - local slot, per column (this batch)
    __shared__ uint2 s_item[ICAP];                                   // {row0 << 8 | column (bit 31: zero item), Gaussian}
    __shared__ uint32_t s_poff[ICAP + 1];                            // height, then local slot of the item's first pair
    __shared__ uint32_t s_tmp[NW];
    __shared__ uint32_t s_next;
#ifdef CUGS_DEV
    unsigned long long tick_ = __builtin_readcyclecounter();
#endif
    const uint32_t total_pairs = live_count(pairs_or_cap, dev_count);
    for (uint32_t z = blockIdx.x * NT + threadIdx.x; z < range_dwords; z += gridDim.x * NT) tile_ranges[z] = 0;
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t i = blockIdx.x * NT + tid;
    uint32_t g = 0, t = 0;
    int x0 = 0, y0 = 0, w = 0, h = 0;
    if (i < n) {
        g = order[i];
        const int4 r = rect_sorted[i];
        t = (uint32_t)r.w;
        x0 = r.x; y0 = r.y; w = r.z & 0xFFFF; h = r.z >> 16;
    }
    const uint32_t nzero = (uint32_t)(w * h) < t ? t - (uint32_t)(w * h) : 0u;
    if (nzero) atomicAdd(zero_pairs, nzero);                         // quirk Q12 slots (rare)
    const uint32_t ni = t == 0u ? 0u : (uint32_t)w + (nzero ? 1u : 0u);   // items of this Gaussian
    {   // where this workgroup's run of each column starts
        const bool col = tid < RADIX;
        const uint32_t col_start = block_exclusive_scan<NW>(col ? coltot[tid] : 0u, s_tmp, nullptr);
        if (col) s_gpos[tid] = col_start + colscan[(size_t)tid * nblk + blockIdx.x];
    }
    uint32_t itot;
    const uint32_t ioff = block_exclusive_scan<NW>(ni, s_tmp, &itot);
    const uint32_t my_batch = ioff / IWIN;
    const uint32_t nbatch = (itot + IWIN - 1) / IWIN;
    const uint32_t word = tid >> 5, bit = 1u << (tid & 31u);
    const uint32_t cu = tid == 0 ? (uint32_t)ZCOL : tid - 1u;        // column of thread `tid` in scan order: pseudo column first
    EMIT_TICK(0);

    for (uint32_t k = 0; k < nbatch; ++k) {
        const bool mine = t > 0u && my_batch == k;
        for (uint32_t e = tid; e < NWORD * NCP; e += NT) (&s_cover[0][0])[e] = 0u;
        __syncthreads();
        if (mine) {                                                  // (b)
            for (int c = 0; c < w; ++c) atomicOr(&s_cover[word][x0 + c], bit);
            if (nzero) atomicOr(&s_cover[word][ZCOL], bit);
        }
        __syncthreads();
        EMIT_TICK(1);
        uint32_t cnt = 0u;                                           // (c)
        if (tid < NC) {
            uint32_t bits[NWORD];
#pragma unroll
            for (int q = 0; q < NWORD; ++q) bits[q] = s_cover[q][cu];
#pragma unroll
            for (int q = 0; q < NWORD; ++q) {
                s_wpre[q][cu] = (uint16_t)cnt;
                cnt += __popc(bits[q]);
            }
        }
        uint32_t icount;
        const uint32_t ist = block_exclusive_scan<NW>(cnt, s_tmp, &icount);
        if (tid < NC) s_istart[cu] = ist;
        __syncthreads();
        EMIT_TICK(2);
        if (mine) {                                                  // (d)
            for (int c = 0; c < w; ++c) {
                const uint32_t col = (uint32_t)(x0 + c);
                const uint32_t at = s_istart[col] + s_wpre[word][col] + __popc(s_cover[word][col] & (bit - 1u));
                s_item[at] = make_uint2(((uint32_t)y0 << 8) | col, g);
                s_poff[at] = (uint32_t)h;
            }
            if (nzero) {
                const uint32_t at = s_istart[ZCOL] + s_wpre[word][ZCOL] + __popc(s_cover[word][ZCOL] & (bit - 1u));
                s_item[at] = make_uint2(0x80000000u, 0u);
                s_poff[at] = nzero;
            }
        }
        __syncthreads();
        EMIT_TICK(3);
        {                                                            // (e) heights -> local slots; s_poff[icount] = pairs of the batch
            const uint32_t e0 = tid * PER;
            uint32_t v[PER], sum = 0u;
#pragma unroll
            for (int e = 0; e < PER; ++e) { v[e] = (e0 + e < icount) ? s_poff[e0 + e] : 0u; sum += v[e]; }
            uint32_t run = block_exclusive_scan<NW>(sum, s_tmp, nullptr);
#pragma unroll
            for (int e = 0; e < PER; ++e) {
                if (e0 + e <= icount) s_poff[e0 + e] = run;
                run += v[e];
            }
        }
        __syncthreads();
        if (tid >= 1u && tid < NC) {                                 // column cu = tid - 1; column 0 also serves the pseudo column
            const uint32_t first = (cu == 0u) ? s_istart[ZCOL] : ist;          // the pseudo column's items sit right before column 0's
            const uint32_t lo_slot = s_poff[first], hi_slot = s_poff[ist + cnt];
            const uint32_t d = s_gpos[cu] - lo_slot;
            s_delta[cu] = d;
            if (cu == 0u) s_delta[ZCOL] = d;
            s_gpos[cu] += hi_slot - lo_slot;
        }
        if (tid == 0) s_next = 0u;
        __syncthreads();
        EMIT_TICK(4);
        const uint32_t nround = (icount + CUGS_WAVE - 1) / CUGS_WAVE;  // (g)
        while (true) {
            uint32_t r = 0u;
            if (lane == 0) r = atomicAdd(&s_next, 1u);
            r = __builtin_amdgcn_readfirstlane(r);
            if (r >= nround) break;
            const uint32_t it = r * CUGS_WAVE + lane;
            const bool valid = it < icount;
            uint32_t slot = 0u, cnt_it = 0u;
            uint2 rec = make_uint2(0u, 0u);
            if (valid) {
                slot = s_poff[it];
                cnt_it = s_poff[it + 1] - slot;
                rec = s_item[it];
            }
            const bool zero = rec.x >> 31;
            const uint32_t dst0 = slot + s_delta[zero ? (uint32_t)ZCOL : (rec.x & 255u)];
            // lane = item, walking down its rows: neighbouring lanes hold neighbouring runs of the output, so the
            // `height` store instructions of a round fill the same cache lines between them
            if (!zero) {
                const uint32_t room = dst0 < total_pairs ? total_pairs - dst0 : 0u;     // never write past the buffers
                const uint32_t rows = min(cnt_it, room);
                uint16_t* kp = ptile + dst0;
                uint32_t* ip = pidx + dst0;
                uint32_t key = rec.x;
                for (uint32_t y = 0; y < rows; ++y) {
                    kp[y] = (uint16_t)key;
                    ip[y] = rec.y;
                    key += 256u;
                }
            }
            for (unsigned long long m = __ballot(valid && zero); m; m &= m - 1ull) {   // zero slots: by the whole wave
                const int l = __builtin_ctzll(m);
                const uint32_t p = __shfl(dst0, l), c = __shfl(cnt_it, l);
                for (uint32_t sl = lane; sl < c; sl += CUGS_WAVE)
                    if (p + sl < total_pairs) { ptile[p + sl] = 0; pidx[p + sl] = 0u; }
            }
        }
        EMIT_TICK(5);
        __syncthreads();
        EMIT_TICK(6);
    }
}

// k_compute_tile_ranges (sorting.cu:82-109) on the sorted tile ids; optionally rebuilds the
// reference's sorted 64-bit keys (SortingOutput::gaussian_keys_sorted, sorting.hpp:20).
// COLKEY: the keys are (row << 8 | column) as written by k_col_emit; ntx turns them back into tile ids.
template <typename K, bool COLKEY>
__global__ __launch_bounds__(CUGS_BLOCK) void k_tile_ranges(uint32_t pairs_or_cap,
                                                            const unsigned long long* __restrict__ dev_count,
                                                            const K* __restrict__ ptile, uint32_t ntx,
                                                            const int32_t* __restrict__ pidx,
                                                            const float* __restrict__ depths,
                                                            int32_t* __restrict__ tile_ranges,
                                                            uint64_t* __restrict__ keys_sorted,
                                                            const uint32_t* __restrict__ zero_pairs) {
    constexpr int PER = 8;                                   // consecutive pairs per thread (one or two 16-byte loads)
    const uint32_t total_pairs = live_count(pairs_or_cap, dev_count);
    const uint32_t i0 = (blockIdx.x * CUGS_BLOCK + threadIdx.x) * PER;
    if (i0 >= total_pairs) return;
    uint32_t t[PER];
    if (i0 + PER <= total_pairs) {
        if (sizeof(K) == 2) {
            const uint4 q = *reinterpret_cast<const uint4*>(ptile + i0);
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { t[2 * e] = w[e] & 0xFFFFu; t[2 * e + 1] = w[e] >> 16; }
        } else {
            const uint4 q0 = reinterpret_cast<const uint4*>(ptile + i0)[0], q1 = reinterpret_cast<const uint4*>(ptile + i0)[1];
            t[0] = q0.x; t[1] = q0.y; t[2] = q0.z; t[3] = q0.w; t[4] = q1.x; t[5] = q1.y; t[6] = q1.z; t[7] = q1.w;
        }
    } else {
#pragma unroll
        for (int e = 0; e < PER; ++e) t[e] = (i0 + e < total_pairs) ? (uint32_t)ptile[i0 + e] : 0u;
    }
    uint32_t prev = (i0 == 0) ? 0u : (uint32_t)ptile[i0 - 1];
    if constexpr (COLKEY) {
        prev = (prev >> 8) * ntx + (prev & 255u);
#pragma unroll
        for (int e = 0; e < PER; ++e) t[e] = (t[e] >> 8) * ntx + (t[e] & 255u);
    }
    const uint32_t nzero = keys_sorted ? *zero_pairs : 0u;
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        const uint32_t i = i0 + e;
        if (i >= total_pairs) break;
        const uint32_t cur = t[e];
        if (i == 0) {
            tile_ranges[cur * 2 + 0] = 0;
        } else if (cur != prev) {
            tile_ranges[prev * 2 + 1] = (int32_t)i;
            tile_ranges[cur * 2 + 0] = (int32_t)i;
        }
        if (i == total_pairs - 1) tile_ranges[cur * 2 + 1] = (int32_t)total_pairs;
        if (keys_sorted)   // Q12 pairs are the leading entries of tile 0 and carry depth bits 0
            keys_sorted[i] = (i < nzero) ? 0ull : (((uint64_t)cur << 32) | (uint64_t)__float_as_uint(depths[pidx[i]]));
        prev = cur;
    }
}

// ------------------------------------------------------------------------------------
// Tile order for the blend kernels (round 3): the tiles sorted by the length of their lists, longest first - the order
// in which the blend kernels' workgroups should be handed out (they take tile_order[blockIdx.x]): a view whose splats
// cluster (every real capture) has a few hundred tiles with lists many times the mean, and in the spatial order those
// workgroups start whenever their position comes up, the last of them long after the rest of the chip has drained.
// Heaviest first, same kernels (tools/lpt_order.py, same box): 80 % of the splats on 10 % of the screen - forward blend
// 176 -> 133 us, backward 409 -> 306; 50 % on 2 % - 182 -> 150, 543 -> 400; the uniform scene unchanged (134 / 432).
// A counting sort over 513 buckets (lengths with 4 bits below the leading one, i.e. to 6 %; empty tiles last) by ONE
// workgroup; the order inside a bucket is whatever the LDS atomics make it - any permutation is a correct order.
// ------------------------------------------------------------------------------------
constexpr uint32_t ORDER_BUCKETS = 513u;          // 32 x 16 length classes + the empty tiles
constexpr uint32_t ORDER_LDS = 576u;              // dwords of LDS the procedure needs (9 buckets per lane of one wave)
__device__ __forceinline__ uint32_t order_bucket(uint32_t len) {
    if (len == 0u) return ORDER_BUCKETS - 1u;
    const uint32_t e = 31u - (uint32_t)__clz((int)len);
    const uint32_t m = e >= 4u ? (len >> (e - 4u)) & 15u : (len << (4u - e)) & 15u;
    return 511u - (e * 16u + m);
}
// By every thread of ONE workgroup (any size that is a multiple of 64).  s_hist: ORDER_LDS dwords of LDS.
// A record of the order is {tile, first pair, one past the last pair, 0}: the blend workgroup that takes it has its tile
// and its range in one 16-byte load (a bare tile id puts a second, dependent memory round trip in front of every
// workgroup: +6 % on the 100 k-Gaussian forward-only frame, whose workgroups are a few microseconds long).
template <typename LenFn, typename StartFn>
__device__ __forceinline__ void write_tile_order(uint32_t tiles, LenFn len_of, StartFn start_of, uint4* __restrict__ order,
                                                 uint32_t* s_hist) {
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    for (uint32_t b = tid; b < ORDER_LDS; b += nt) s_hist[b] = 0u;
    __syncthreads();
    for (uint32_t t = tid; t < tiles; t += nt) atomicAdd(&s_hist[order_bucket(len_of(t))], 1u);
    __syncthreads();
    if (tid < (uint32_t)CUGS_WAVE) {                                  // exclusive scan of the bucket counts by one wave
        uint32_t v[9], sum = 0u;
#pragma unroll
        for (int k = 0; k < 9; ++k) { v[k] = s_hist[tid * 9u + k]; sum += v[k]; }
        uint32_t run = wave_inclusive_scan(sum) - sum;
#pragma unroll
        for (int k = 0; k < 9; ++k) { s_hist[tid * 9u + k] = run; run += v[k]; }
    }
    __syncthreads();
    for (uint32_t t = tid; t < tiles; t += nt) {
        const uint32_t len = len_of(t), first = len ? start_of(t) : 0u;
        order[atomicAdd(&s_hist[order_bucket(len)], 1u)] = make_uint4(t, first, first + len, 0u);
    }
    __syncthreads();
}
__global__ __launch_bounds__(1024) void k_tile_order(uint32_t tiles, const int32_t* __restrict__ tile_ranges,
                                                      uint4* __restrict__ order) {
    __shared__ uint32_t s_hist[ORDER_LDS];
    write_tile_order(tiles, [&](uint32_t t) { return (uint32_t)(tile_ranges[2 * t + 1] - tile_ranges[2 * t]); },
                     [&](uint32_t t) { return (uint32_t)tile_ranges[2 * t]; }, order, s_hist);
}

// ------------------------------------------------------------------------------------
// Direct binning (round 3): steps (2)-(4) as ONE counting sort by tile id, for views of up to 2 M Gaussians on images of up
// to BIN_T_MAX tiles whose per-Gaussian records are the projection's packed rectangles (render()'s route).  The two
// radix passes over the pairs (emit tile id + index, histogram, scatter, histogram, scatter, range detection: eight
// launches, every pair written three times and read four) become three launches that write every pair ONCE:
//   k_bin_count    a workgroup takes a GROUP of 4096 consecutive Gaussians of the depth order and counts, in LDS, how
//                  many of them cover each tile: row b of the table [groups][tiles];
//   k_bin_scan     per tile, the exclusive prefix of its column of the table (= where group b's first pair of the tile
//                  goes inside the tile's list), the tile's total, and the totals of 64-tile chunks;
//   k_bin_scatter  a wave per (group, block of 8 x 8 tiles) scans the chunk totals into its tiles' starts, walks the
//                  group's records in depth order and writes each pair's Gaussian index straight to
//                  tile start + prefix + pairs of this group written so far; group 0's waves publish the ranges.
// What makes the last kernel a STABLE sort (ties in depth order, bit for bit what the radix passes give): every
// (group, tile) has exactly one writer, a lane that visits the records in order.
// Measured, whole sort, same box (tools/ablate_bin.py, profiles/r03_s_direct_binning.log): 1 M Gaussians / 8.4 M pairs
// 0.200 -> 0.178 ms, 45 M pairs 0.46-0.48 -> 0.355, 100 k Gaussians 0.130 -> 0.119; 6 M Gaussians / 40 M pairs 0.825 vs 0.83
// (not taken there).  What was tried on the way is in profiles/README.md (a workgroup per group with tile ownership by
// wave and 64-bit cover words for the ranking: 0.25 ms, latency-bound at one workgroup per CU; a wave per tile row or
// per band of four rows: 0.23-0.25, instruction-bound on the per-record scalar loop).
// ------------------------------------------------------------------------------------
constexpr int BIN_WAVES = 16, BIN_NT = BIN_WAVES * CUGS_WAVE;
constexpr int BIN_T_MAX = BIN_TILES_CAP;
// The scatter's workgroups: up to 8 horizontally adjacent blocks of 8 x 8 tiles, evenly filled (15 block columns: 8 + 7)
inline uint32_t bin_window_groups(int ntx) { const uint32_t nbx = ((uint32_t)ntx + 7u) / 8u; return (nbx + 7u) / 8u; }
inline uint32_t bin_window_cols(int ntx) {       // tile columns per window
    const uint32_t nbx = ((uint32_t)ntx + 7u) / 8u, gxs = bin_window_groups(ntx);
    return ((nbx + gxs - 1u) / gxs) * 8u;
}
inline uint32_t bin_windows(int ntx, int nty) { return (((uint32_t)nty + 7u) / 8u) * bin_window_groups(ntx); }
inline bool bin_route(int ntx, int nty) {
    return cugs_prect_packable(ntx, nty) && ntx * nty <= BIN_T_MAX;
}

// rect: the records in INPUT order (gathered through `order` and packed here, prect_out keeps them for the scatter) unless
// prect_in holds them in depth order already (they rode through the depth passes).
// Counting costs FOUR LDS atomics per Gaussian, whatever its size: +1 / -1 at the corners of its rectangle in a grid of
// differences, then a prefix sum along the rows and one along the columns (a loop over the w x h tiles of each lane's own
// rectangle keeps a quarter of the lanes busy, and an LDS atomic instruction costs the same ~9 clocks of the CU's LDS
// pipeline with 15 active lanes as with 64: 33 us at 8.4 pairs per Gaussian, 200 at 45).
constexpr int BIN_D_MAX = BIN_T_MAX + 2 * CUGS_PRECT_MAX_TILES + 2;   // (ntx + 1) x (nty + 1) differences
__global__ __launch_bounds__(BIN_NT) void k_bin_count(uint32_t n, uint32_t group, const uint32_t* __restrict__ order,
                                                      const int4* __restrict__ rect, const uint32_t* __restrict__ prect_in,
                                                      uint32_t* __restrict__ prect_out, uint32_t ntx, uint32_t nty,
                                                      uint32_t* __restrict__ table, uint32_t* __restrict__ zero_pairs,
                                                      uint32_t* __restrict__ win) {
    __shared__ int32_t s_d[BIN_D_MAX];
    if (blockIdx.x == 0 && threadIdx.x < BIN_WINDOWS_MAX) win[threadIdx.x] = 0u;      // k_bin_scan adds the windows' pairs up
    __shared__ int32_t s_part[8][128];
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const uint32_t w2 = ntx + 1u, cells = w2 * (nty + 1u);
    for (uint32_t e = tid; e < cells; e += BIN_NT) s_d[e] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * group;
    const uint32_t end = min(n, base + group);
    for (uint32_t i = base + tid; i < end; i += BIN_NT) {
        uint32_t pr;
        if (prect_in) {
            pr = prect_in[i];
        } else {
            pr = pack_rect(rect[order[i]]);                           // the one gather per Gaussian
            prect_out[i] = pr;
        }
        if (pr & 0x80000000u) {                                       // no rectangle: nothing, or quirk Q12's zero slots
            const uint32_t z = pr & 0x7FFFFFFFu;
            if (z) atomicAdd(zero_pairs, z);
            continue;
        }
        const uint32_t x0 = pr & 127u, y0 = (pr >> 7) & 127u, w = (pr >> 14) & 127u, h = (pr >> 21) & 127u;
        atomicAdd(&s_d[y0 * w2 + x0], 1);
        atomicAdd(&s_d[y0 * w2 + x0 + w], -1);
        atomicAdd(&s_d[(y0 + h) * w2 + x0], -1);
        atomicAdd(&s_d[(y0 + h) * w2 + x0 + w], 1);
    }
    __syncthreads();
    for (uint32_t row = wave; row < nty; row += BIN_WAVES) {           // prefix along x: a wave per row
        int32_t carry = 0;
        for (uint32_t c0 = 0; c0 < ntx; c0 += CUGS_WAVE) {
            const uint32_t x = c0 + lane;
            const int32_t v = x < ntx ? s_d[row * w2 + x] : 0;
            const int32_t inc = (int32_t)wave_inclusive_scan((uint32_t)v) + carry;
            if (x < ntx) s_d[row * w2 + x] = inc;
            carry = __shfl(inc, 63);
        }
    }
    __syncthreads();
    {   // prefix along y: thread = (column, one of eight runs of rows)
        const uint32_t x = tid & 127u, seg = tid >> 7;
        const uint32_t rps = (nty + 7u) / 8u;
        const uint32_t ya = min(nty, seg * rps), yb = min(nty, ya + rps);
        int32_t sum = 0;
        if (x < ntx)
            for (uint32_t y = ya; y < yb; ++y) sum += s_d[y * w2 + x];
        s_part[seg][x] = sum;
        __syncthreads();
        int32_t run = 0;
        for (uint32_t s2 = 0; s2 < seg; ++s2) run += s_part[s2][x];
        if (x < ntx)
            for (uint32_t y = ya; y < yb; ++y) {
                run += s_d[y * w2 + x];
                s_d[y * w2 + x] = run;
            }
    }
    __syncthreads();
    uint32_t* out = table + (size_t)blockIdx.x * (ntx * nty);
    for (uint32_t row = wave; row < nty; row += BIN_WAVES)
        for (uint32_t x = lane; x < ntx; x += CUGS_WAVE) out[row * ntx + x] = (uint32_t)s_d[row * w2 + x];
}

// Workgroup = 64 tiles x 16 runs of table rows.  In place: table[b][t] becomes the number of pairs of tile t in the
// workgroups before b; ttot[t] = pairs of tile t.  Block 0 also takes the snapshots the scatter works from: the Q12 count
// k_bin_count has finished adding to (snap[0]; the counter is re-armed) and the depth range flag (snap[1]; re-armed).
// No grid-wide step here: a workgroup that waits for the others' totals must first make its own visible across the
// XCDs' L2s (a release fence = an L2 write-back with 8 MB of freshly written table in it: this kernel took 52 us that
// way).  Each workgroup leaves the prefix of its 64 tiles' totals and their sum (a chunk) instead; the scatter's waves
// finish the scan over the <= 160 chunk sums themselves.
__global__ __launch_bounds__(BIN_NT) void k_bin_scan(uint32_t rows, uint32_t tiles, uint32_t* __restrict__ table,
                                                     uint32_t* __restrict__ ttot, uint32_t* __restrict__ tpre,
                                                     uint32_t* __restrict__ csum, uint32_t* __restrict__ q12,
                                                     uint32_t* __restrict__ range_flag, uint32_t* __restrict__ snap,
                                                     uint32_t* __restrict__ win, uint32_t ntx, uint32_t win_cols,
                                                     uint32_t gxs) {
    __shared__ uint32_t s_seg[BIN_WAVES][CUGS_WAVE];
    __shared__ uint32_t s_win[BIN_WINDOWS_MAX];                       // touched by wave 0 only
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        snap[0] = q12[0]; q12[1] = q12[0]; q12[0] = 0u;
        snap[1] = *range_flag; *range_flag = 0u;
    }
    const uint32_t tl = threadIdx.x & 63u, seg = threadIdx.x >> 6;
    const uint32_t t = blockIdx.x * CUGS_WAVE + tl;
    const uint32_t rps = (rows + BIN_WAVES - 1u) / BIN_WAVES;
    const uint32_t r0 = min(rows, seg * rps), r1 = min(rows, r0 + rps);
    uint32_t sum = 0u;
    if (t < tiles) {
#pragma unroll 8
        for (uint32_t r = r0; r < r1; ++r) sum += table[(size_t)r * tiles + t];
    }
    s_seg[seg][tl] = sum;
    __syncthreads();
    uint32_t pre = 0u, tot = 0u;
#pragma unroll
    for (uint32_t s2 = 0; s2 < (uint32_t)BIN_WAVES; ++s2) {
        const uint32_t v = s_seg[s2][tl];
        pre += s2 < seg ? v : 0u;
        tot += v;
    }
    if (t < tiles) {
        uint32_t run = pre;
#pragma unroll 8
        for (uint32_t r = r0; r < r1; ++r) {
            const uint32_t v = table[(size_t)r * tiles + t];
            table[(size_t)r * tiles + t] = run;
            run += v;
        }
    }
    if (seg == 0u) {                                                  // this workgroup's 64 tiles (a CHUNK): totals, their prefix, their sum
        const uint32_t v = t < tiles ? tot : 0u;
        const uint32_t inc = wave_inclusive_scan(v);
        if (t < tiles) { ttot[t] = v; tpre[t] = inc - v; }
        if (tl == 63u) csum[blockIdx.x] = inc;
        // pairs per WINDOW of the scatter (8 tile rows x win_cols tile columns: one workgroup per group of the depth order):
        // what the scatter orders its workgroups by (win == NULL: more windows than BIN_WINDOWS_MAX, no ordering)
        if (win) {                                                    // (kernel-uniform; all of it inside this one wave)
            s_win[tl] = 0u;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (v) {
                const uint32_t ty = t / ntx, tx = t - ty * ntx;
                atomicAdd(&s_win[(ty >> 3) * gxs + tx / win_cols], v);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t mine = s_win[tl];
            if (mine) atomicAdd(&win[tl], mine);
        }
    }
}

// A WAVE per (group of the depth order, block of 8 x 8 tiles): lane = tile, the next free slot of the lane's tile lives in a
// register, and the wave walks - in depth order - the records of the group whose rectangle touches its block; each one
// is ONE masked store: the lanes inside the rectangle write the Gaussian's index to their tile's slot and advance.
// Stable by construction (one wave per tile, records in order): no atomics, no ranking, no LDS round trip in the loop, and
// tens of thousands of independent waves that hide each other's latencies.
// A workgroup is up to eight horizontally adjacent blocks.  Its waves first share out a pre-filter - each takes a slice of the
// group's records straight from memory and lists in LDS, in order, the ones that touch the workgroup's 8-row, <= 64-column
// window (one in twelve at 1080p), with their Gaussian index - and every wave then tests only the listed ones against its
// own block.
constexpr int BIN_BLK = 8;                        // tile block edge: 64 tiles, one per lane
constexpr int BIN_WG_WAVES = 8;
constexpr int BIN_SLICE = 512;                    // records per wave and stage in the pre-filter
constexpr int BIN_SLICE_STEPS = BIN_SLICE / CUGS_WAVE;
// STAGE (dense views: from BIN_STAGE_RATIO pairs per Gaussian on): a lane collects the indices for its tile in LDS and
// writes them sixteen at a time - one 64-byte run of its tile's list - instead of one scattered 4-byte store per pair
// (45 M of those are two thirds of the kernel on the dense 1080p view).  The sparse views keep the direct stores: their
// (group, tile) runs are ~4 entries long, and the buffer's LDS would halve the resident workgroups.
#ifndef CUGS_BIN_RUN
#define CUGS_BIN_RUN 16
#endif
constexpr int BIN_RUN = CUGS_BIN_RUN;             // entries a lane collects before the wave writes (16: two workgroups per CU, sort 0.270-0.279 ms
                                                  // on the dense 1080p view; 8: three per CU, 0.277-0.285; without the buffer 0.345)
constexpr int BIN_RUN_STRIDE = BIN_RUN + 3;       // LDS row stride in dwords: odd (the lanes' rows start in different banks), and the
                                                  // flush reads up to three entries beyond `held`
constexpr uint32_t BIN_STAGE_RATIO = 13u;
template <bool STAGE>
__global__ __launch_bounds__(BIN_WG_WAVES * CUGS_WAVE) void k_bin_scatter(
    uint32_t n, uint32_t group, uint32_t nbx, uint32_t nby, uint32_t gxs, uint32_t pairs_or_cap, bool predicted,
    const uint32_t* __restrict__ order, const uint32_t* __restrict__ prect, uint32_t ntx, uint32_t nty,
    const uint32_t* __restrict__ table, const uint32_t* __restrict__ ttot, const uint32_t* __restrict__ tpre,
    const uint32_t* __restrict__ csum, const uint32_t* __restrict__ snap, uint32_t* __restrict__ tbase,
    unsigned long long* __restrict__ total, unsigned long long* __restrict__ total_mapped, uint32_t* __restrict__ out,
    int32_t* __restrict__ tile_ranges, uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ win,
    uint32_t ablate) {
    __shared__ uint2 s_cand[BIN_WG_WAVES][BIN_SLICE];                 // {packed rectangle, Gaussian} of the listed records
    __shared__ uint32_t s_cnt[BIN_WG_WAVES];
    __shared__ uint32_t s_run[STAGE ? BIN_WG_WAVES * CUGS_WAVE * BIN_RUN_STRIDE + 4 : 1];   // (+ the read-ahead of the last row's flush)
#ifdef CUGS_DEV
    const uint32_t abl = ablate;                                      // tools/ablate_bin.py
#else
    constexpr uint32_t abl = 0u;
#endif
    const uint32_t nt = blockDim.x, nw = nt >> 6, tid = threadIdx.x, wid = tid >> 6, lane = tid & 63u;
    const uint32_t per_group = nby * gxs;
    // Which (group, window) this workgroup takes.  Balanced views: group-major (the windows of one group side by side:
    // they read the same records).  When one window holds over twice the mean (a view whose splats cluster: half of every
    // group's records can fall into ONE 8 x 8 tile block, whose wave then walks them one by one for tens of microseconds):
    // window-major with the heaviest window first, so that those long workgroups all start at once instead of one per
    // group all the way to the end of the grid (k_bin_scatter 155 -> 96 us with half of the splats on 2 % of the screen).
    uint32_t blk = blockIdx.x / per_group, rem = blockIdx.x - blk * per_group;
    const uint32_t win_mine = (win && lane < per_group) ? win[lane] : 0u;     // pairs of window `lane` (k_bin_scan)
    const uint32_t zero = snap[0];                                    // quirk Q12's (tile 0, Gaussian 0) pairs: the head of tile 0's list
    const bool bad = snap[1] != 0u;                                   // a depth key outside the three-pass range: nothing is valid
    const uint32_t tiles = ntx * nty;

    // Tile starts, by every wave for itself (a kernel of its own for this scan cost 10 us of the frame): the totals of the
    // 64-tile chunks (k_bin_scan's workgroups: <= 160 of them) scanned across the lanes, + the tile's prefix inside its chunk.
    const uint32_t nch = (tiles + CUGS_WAVE - 1u) / CUGS_WAVE;
    uint32_t cpre[3] = {0u, 0u, 0u};                                  // exclusive prefix of chunk (lane + 64 i)
    uint32_t pairs = zero;                                            // the pair total, SATURATING at 2^32 - 1 (such a total never fits)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if ((uint32_t)i * CUGS_WAVE < nch) {                          // kernel-uniform: 128 chunks at 1080p = two scans
            const uint32_t c = lane + (uint32_t)i * CUGS_WAVE;
            const uint32_t v = c < nch ? csum[c] : 0u;
            uint32_t inc = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(inc, d);
                const uint32_t sum = inc + o;
                if ((int)lane >= d) inc = sum < o ? 0xFFFFFFFFu : sum;
            }
            const uint32_t with = inc + pairs;                        // + the chunks of the earlier scans (and the Q12 pairs)
            cpre[i] = (with < inc ? 0xFFFFFFFFu : with) - v;          // (meaningless once saturated: nothing is written then)
            const uint32_t last = __shfl(inc, 63), tot = last + pairs;
            pairs = tot < last ? 0xFFFFFFFFu : tot;
        }
    }
    const bool fits = !bad && (!predicted || pairs <= pairs_or_cap);
    cpre[0] -= zero; cpre[1] -= zero; cpre[2] -= zero;                // (the Q12 pairs are added to `start` below)
    // does one window hold over twice the mean?  (one compare and a ballot per wave; `pairs` is the total)
    if (__ballot((unsigned long long)win_mine * per_group > 2ull * pairs) != 0ull) {     // kernel-uniform
        uint32_t rank = 0u;                                           // (v_readlane with a scalar lane: no LDS round trips)
        for (uint32_t w2 = 0; w2 < per_group; ++w2) {
            const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)win_mine, (int)w2);
            rank += (o > win_mine || (o == win_mine && w2 < lane)) ? 1u : 0u;
        }
        const uint32_t groups = gridDim.x / per_group;
        const uint32_t slot = blockIdx.x / groups;
        blk = blockIdx.x - slot * groups;
        rem = (uint32_t)__builtin_ctzll(__ballot(lane < per_group && rank == slot));
    }
    const uint32_t by = rem / gxs, gx = rem - by * gxs;

    const uint32_t bx = gx * nw + wid;
    const bool active = bx < nbx;
    const uint32_t tx = bx * BIN_BLK + (lane & 7u), ty = by * BIN_BLK + (lane >> 3);      // this lane's tile
    const bool tile_ok = active && tx < ntx && ty < nty;
    const uint32_t t = tile_ok ? ty * ntx + tx : 0u;
    uint32_t start;                                                   // where the first REAL pair of the tile goes
    {
        const uint32_t ch = t >> 6;
        const uint32_t p0 = __shfl(cpre[0], ch & 63u), p1 = __shfl(cpre[1], ch & 63u), p2 = __shfl(cpre[2], ch & 63u);
        start = zero + (ch < 64u ? p0 : ch < 128u ? p1 : p2) + (tile_ok ? tpre[t] : 0u);
    }
    uint32_t pos = tile_ok ? (start + table[(size_t)blk * tiles + t]) * 4u : 0u;           // BYTE offset of the tile's next slot
    if (blk == 0u) {
        // group 0's waves cover every tile once: they publish what k_scan_blocksums / k_tile_ranges publish on the radix
        // route.  When the pairs do not fit the buffer (or the depth order is invalid) the result is declared invalid
        // through the total, nothing is written, and EVERY range is {0,0}: the blend queued behind this kernel then does
        // nothing instead of walking an unwritten index buffer.
        if (tile_ok) {
            const uint32_t c = fits ? ttot[t] + (t == 0u ? zero : 0u) : 0u;                 // {0,0} for untouched tiles (sorting.cu:216)
            tile_ranges[2 * t + 0] = c ? (int32_t)(t == 0u ? 0u : start) : 0;
            tile_ranges[2 * t + 1] = c ? (int32_t)(start + ttot[t]) : 0;
            tbase[t] = start;
        }
        if (rem == 0u && tid == 0u) {
            unsigned long long exact = zero;                          // in 64 bits: int32 overflow is the host's check
            for (uint32_t c = 0; c < nch; ++c) exact += csum[c];
            tbase[tiles] = pairs;
            const unsigned long long host_total = bad ? ~0ull : exact;
            total[0] = bad ? 0ull : exact;
            total[1] = host_total;
            if (total_mapped) __hip_atomic_store(total_mapped, host_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (tile_order && blockIdx.x == 0u) {                             // the blend kernels' workgroup order, by this one workgroup
        uint32_t* const s_lds = reinterpret_cast<uint32_t*>(&s_cand[0][0]);   // (the candidate lists are not in use yet)
        uint32_t* const s_chunk = s_lds + ORDER_LDS;                  // exclusive prefix of every chunk, for the tile starts
        if (wid == 0u) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (lane + (uint32_t)i * CUGS_WAVE < nch) s_chunk[lane + (uint32_t)i * CUGS_WAVE] = cpre[i];
        }
        write_tile_order(tiles, [&](uint32_t t2) { return fits ? ttot[t2] + (t2 == 0u ? zero : 0u) : 0u; },
                         [&](uint32_t t2) { return t2 == 0u ? 0u : zero + s_chunk[t2 >> 6] + tpre[t2]; },
                         reinterpret_cast<uint4*>(tile_order), s_lds);
    }
    if (!fits) return;
    if (blk == 0u)                                                    // the Q12 slots: (tile 0, Gaussian 0) pairs
        for (uint32_t k = rem * nt + tid; k < zero; k += per_group * nt) out[k] = 0u;
    // the workgroup's window, in tiles
    const uint32_t win_y0 = by * BIN_BLK, win_y1 = win_y0 + BIN_BLK;
    const uint32_t win_x0 = gx * nw * BIN_BLK, win_x1 = win_x0 + nw * BIN_BLK;
    const uint32_t blk_x0 = bx * BIN_BLK, blk_x1 = blk_x0 + BIN_BLK;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    char* const outb = reinterpret_cast<char*>(out);
    uint32_t* const run = s_run + (STAGE ? tid * BIN_RUN_STRIDE : 0u);      // STAGE: this lane's collected indices
    uint32_t held = 0u;
    const uint32_t base = blk * group;
    const uint32_t end = min(n, base + group);                        // base < end: the grid covers ceil(n / group) groups
    const uint32_t stage = nw * BIN_SLICE;
    for (uint32_t c0 = base; c0 < end; c0 += stage) {
        __syncthreads();                                              // every wave is done with the last stage's lists
        {   // pre-filter: this wave's slice against the workgroup's window (a record without a rectangle has bit 31 set)
            const uint32_t s_begin = c0 + wid * BIN_SLICE;
            uint32_t pr[BIN_SLICE_STEPS], gq[BIN_SLICE_STEPS];
            bool ov[BIN_SLICE_STEPS];
#pragma unroll
            for (int u = 0; u < BIN_SLICE_STEPS; ++u) {               // all loads first (clamped addresses, no branches):
                const uint32_t i = s_begin + (uint32_t)u * CUGS_WAVE + lane;      // ONE memory round trip per stage
                pr[u] = prect[min(i, end - 1u)];
                gq[u] = order[min(i, end - 1u)];
                if (i >= end) pr[u] = 0x80000000u;
            }
#pragma unroll
            for (int u = 0; u < BIN_SLICE_STEPS; ++u) {
                const uint32_t x0 = pr[u] & 127u, y0 = (pr[u] >> 7) & 127u, w = (pr[u] >> 14) & 127u, h = (pr[u] >> 21) & 127u;
                ov[u] = (int32_t)pr[u] >= 0 && y0 < win_y1 && y0 + h > win_y0 && x0 < win_x1 && x0 + w > win_x0;
            }
            uint32_t found = 0u;
#pragma unroll
            for (int u = 0; u < BIN_SLICE_STEPS; ++u) {
                const unsigned long long m = __ballot(ov[u]);
                if (ov[u]) s_cand[wid][found + (uint32_t)__popcll(m & lt_mask)] = make_uint2(pr[u], gq[u]);
                found += (uint32_t)__popcll(m);
            }
            if (lane == 0u) s_cnt[wid] = found;
        }
        __syncthreads();
        if (!active) continue;
        for (uint32_t w2 = 0; w2 < nw; ++w2) {                        // the slices' lists one after the other: depth order
            const uint32_t c = s_cnt[w2];
            for (uint32_t k0 = 0; k0 < c; k0 += CUGS_WAVE) {
                uint2 rec = make_uint2(0x80000000u, 0u);
                if (k0 + lane < c) rec = s_cand[w2][k0 + lane];
                const uint32_t x0v = rec.x & 127u, wv = (rec.x >> 14) & 127u;
                unsigned long long m = __ballot((int32_t)rec.x >= 0 && x0v < blk_x1 && x0v + wv > blk_x0);
                if (abl & 8u) m = 0ull;
                while (m != 0ull) {
                    const int l = __builtin_ctzll(m);
                    m &= m - 1ull;
                    const uint32_t prl = (uint32_t)__builtin_amdgcn_readlane((int)rec.x, l);
                    const uint32_t g = (uint32_t)__builtin_amdgcn_readlane((int)rec.y, l);
                    const uint32_t x0 = prl & 127u, y0 = (prl >> 7) & 127u, w = (prl >> 14) & 127u, h = (prl >> 21) & 127u;
                    const bool in = ((tx - x0) < w) & ((ty - y0) < h);     // unsigned: inside the rectangle
                    if constexpr (STAGE) {
                        if (in) run[held++] = g;
                        if (__ballot(held == (uint32_t)BIN_RUN) != 0ull) {
                            // one lane's row is full: EVERY lane writes the whole 16-byte pieces it holds (all lanes at
                            // once - a flush by the one or two full lanes alone is a string of nearly empty instructions)
                            // and keeps the up to three entries left over
                            const uint32_t whole = held & ~3u;
#pragma unroll
                            for (int q = 0; q < BIN_RUN; q += 4)
                                if ((uint32_t)q < whole && !(abl & 1u))
                                    *reinterpret_cast<uint4*>(outb + pos + 4 * q) = make_uint4(run[q], run[q + 1], run[q + 2], run[q + 3]);
                            const uint32_t rest = held - whole;
                            const uint32_t r0 = run[whole], r1 = run[whole + 1u], r2 = run[whole + 2u];   // (reads ahead of `held`: values unused)
                            if (rest > 0u) run[0] = r0;
                            if (rest > 1u) run[1] = r1;
                            if (rest > 2u) run[2] = r2;
                            pos += 4u * whole;
                            held = rest;
                        }
                    } else if (in) {
                        if (!(abl & 1u)) *reinterpret_cast<uint32_t*>(outb + pos) = g;
                        pos += 4u;
                    }
                }
            }
        }
    }
    if constexpr (STAGE) {                                            // what the lanes still hold
        for (uint32_t q = 0; __ballot(q < held) != 0ull; ++q)
            if (q < held && !(abl & 1u)) *reinterpret_cast<uint32_t*>(outb + pos + 4u * q) = run[q];
    }
}

// SortingOutput::gaussian_keys_sorted for the direct route (only when the caller asks for the keys): the tile of pair i
// is the last tile whose list starts at or before i.
__global__ __launch_bounds__(CUGS_BLOCK) void k_bin_keys(uint32_t pairs_or_cap, const unsigned long long* __restrict__ dev_count,
                                                         uint32_t tiles, const uint32_t* __restrict__ tbase,
                                                         const uint32_t* __restrict__ zero_snap,
                                                         const int32_t* __restrict__ pidx, const float* __restrict__ depths,
                                                         uint64_t* __restrict__ keys_sorted) {
    if (dev_count && *dev_count > (unsigned long long)pairs_or_cap) return;   // the pairs did not fit: no indices were written
    const uint32_t total = live_count(pairs_or_cap, dev_count);
    const uint32_t i = blockIdx.x * CUGS_BLOCK + threadIdx.x;
    if (i >= total) return;
    if (i < *zero_snap) { keys_sorted[i] = 0ull; return; }            // Q12 pairs: key 0
    uint32_t lo = 0u, hi = tiles;                                     // largest t in [0, tiles) with tbase[t] <= i (tbase[0] = Z <= i)
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (tbase[mid] <= i) lo = mid; else hi = mid;
    }
    keys_sorted[i] = ((uint64_t)lo << 32) | (uint64_t)__float_as_uint(depths[pidx[i]]);
}

#ifdef CUGS_DEV
// ---- development build only (libcugs_hip_dev.so): ranking by one LDS atomic-with-return per item -------------
// Measured 7 % faster than the ballot ranking (sort 0.250 -> 0.232 ms at config 3), but it relies on an ordering
// of same-address LDS lanes that the ISA manual does not state.  The shipped library therefore always ranks with
// wave ballots, keeps no mode variable and reads no environment; this path, its on-device probe and the
// cugsdbg_sort_rank_mode hook exist only for experiments (tests/test_gpu_parity.py runs both modes against the
// oracle in a child process that loads the development library).
// Does an LDS atomic with return serve the lanes of one wave instruction that hit the SAME address in ascending
// lane order, and successive instructions of a wave in issue order?  Each lane checks that the value it got back
// equals the number of earlier (round, lane) items with its digit, for random, clustered, constant, same-bank and
// strided digit patterns; *violations counts the mismatches.
__device__ __forceinline__ uint32_t probe_digit(uint32_t set, uint32_t wave, uint32_t r, uint32_t lane) {
    uint32_t h = (set * 4u + wave) * 8u + r;
    h = (h ^ 61u) ^ (h >> 16); h *= 9u; h ^= h >> 4; h *= 0x27d4eb2du; h ^= h >> 15;
    uint32_t x = h + lane * 0x9E3779B9u;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    switch (set % 6u) {
        case 0: return x & 255u;
        case 1: return x & 127u;
        case 2: return x & 3u;
        case 3: return 5u;
        case 4: return (x & 1u) ? 7u : 39u;                       // same LDS bank, 32 dwords apart
        default: return (lane * 37u + (x & 1u)) & 63u;
    }
}
__global__ __launch_bounds__(CUGS_BLOCK) void k_probe_lds_order(uint32_t* __restrict__ violations) {
    __shared__ uint32_t cnt[4][RADIX];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63, set = blockIdx.x;
    for (uint32_t i = threadIdx.x; i < 4 * RADIX; i += CUGS_BLOCK) (&cnt[0][0])[i] = 0;
    __syncthreads();
    uint32_t d[8], got[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) d[r] = probe_digit(set, wave, r, lane);
#pragma unroll
    for (int r = 0; r < 8; ++r) got[r] = atomicAdd(&cnt[wave][d[r]], 1u);
    uint32_t bad = 0;
    for (int r = 0; r < 8; ++r) {
        uint32_t expect = 0;
        for (int rr = 0; rr <= r; ++rr)
            for (uint32_t l = 0; l < 64u && !(rr == r && l >= lane); ++l) expect += probe_digit(set, wave, rr, l) == d[r];
        bad += expect != got[r];
    }
    if (bad) atomicAdd(violations, bad);
}

// -1: not probed yet (ballot ranking is used), 0: ballot ranking, 1: atomic ranking.
std::atomic<int> g_rank_mode{-1};
int rank_mode() { return g_rank_mode.load(std::memory_order_relaxed); }
#else
constexpr int rank_mode() { return 0; }            // ballot ranking: defined by the ISA, no state
#endif


template <typename K, bool IOTA, int NT, int CHUNK, int RDX = RADIX>
int radix_pass(const K* kin, const uint32_t* vin, uint32_t count, const unsigned long long* dev_count, int shift, int bits,
               uint32_t* hist, uint32_t* sup, uint32_t* tot, K* kout, uint32_t* vout, uint32_t* ctl, hipStream_t st,
               const uint32_t* v2in = nullptr, uint32_t* v2out = nullptr) {
    // `sup`: this pass's super-block table, ZEROED by an earlier kernel of the stream (the key kernel / the projection for
    // the depth passes, the pair emission for the pair passes); used for passes of up to SCANFREE_MAX_BLOCKS workgroups
    const uint32_t nblk = nblocks_for(count, CHUNK);
    const uint32_t sb = sup_block(nblk);
    static_assert(NT % RDX == 0, "whole thread groups per digit");
    if (!scan_free(nblk, (uint32_t)NT >> bits)) sup = nullptr;
    hipLaunchKernelGGL((k_radix_hist<K, NT, CHUNK, RDX>), dim3(nblk), dim3(NT), 0, st, kin, count, dev_count, shift,
                       (1u << bits) - 1u, hist, nblk, ctl, sup, sb);
    CUGS_LAUNCH_CHECK();
    if (!sup) {
        hipLaunchKernelGGL(k_radix_scan_rows, dim3(RDX), dim3(CUGS_BLOCK), 0, st, hist, hist, nblk, tot);
        CUGS_LAUNCH_CHECK();
    }
    if constexpr (RDX == RADIX_DEPTH) {                    // the 9-bit passes of the depth sort: ballot ranking only
        if (v2in)                                          // the packed tile rectangle rides along
            hipLaunchKernelGGL((k_radix_scatter<K, IOTA, DEPTH_BITS, NT, false, CHUNK, RDX, true>), dim3(nblk), dim3(NT), 0, st, kin,
                               vin, count, dev_count, shift, 0u, hist, sup, sb, tot, nblk, kout, vout, v2in, v2out);
        else
            hipLaunchKernelGGL((k_radix_scatter<K, IOTA, DEPTH_BITS, NT, false, CHUNK, RDX>), dim3(nblk), dim3(NT), 0, st, kin, vin, count,
                               dev_count, shift, 0u, hist, sup, sb, tot, nblk, kout, vout);
        CUGS_LAUNCH_CHECK();
        return 0;
    }
    if (v2in) return CUGS_EINVAL;
#ifdef CUGS_DEV
    if (rank_mode() == 1) {               // digit width only matters to the ballot ranking: one instantiation
        hipLaunchKernelGGL((k_radix_scatter<K, IOTA, 8, NT, true, CHUNK>), dim3(nblk), dim3(NT), 0, st, kin, vin, count, dev_count,
                           shift, (1u << bits) - 1u, hist, sup, sb, tot, nblk, kout, vout);
        CUGS_LAUNCH_CHECK();
        return 0;
    }
#endif
#define CUGS_SCATTER(NB)                                                                                          \
    hipLaunchKernelGGL((k_radix_scatter<K, IOTA, NB, NT, false, CHUNK>), dim3(nblk), dim3(NT), 0, st, kin, vin, count, dev_count, \
                       shift, 0u, hist, sup, sb, tot, nblk, kout, vout)
    switch (bits) {
        case 1: CUGS_SCATTER(1); break;
        case 2: CUGS_SCATTER(2); break;
        case 3: CUGS_SCATTER(3); break;
        case 4: CUGS_SCATTER(4); break;
        case 5: CUGS_SCATTER(5); break;
        case 6: CUGS_SCATTER(6); break;
        case 7: CUGS_SCATTER(7); break;
        default: CUGS_SCATTER(8); break;
    }
#undef CUGS_SCATTER
    CUGS_LAUNCH_CHECK();
    return 0;
}

#ifdef CUGS_DEV
inline bool bin_stage_enabled() { const char* e = std::getenv("CUGS_BIN_NO_STAGE"); return !(e && e[0] == '1'); }   // A/B
#else
constexpr bool bin_stage_enabled() { return true; }
#endif
#ifdef CUGS_DEV
std::atomic<void*> g_mark_event{nullptr};         // development build: an event recorded right before k_bin_scatter
#endif
#ifdef CUGS_DEV
inline uint32_t bin_ablate() { const char* e = std::getenv("CUGS_BIN_ABLATE"); return e ? (uint32_t)std::atoi(e) : 0u; }
#else
constexpr uint32_t bin_ablate() { return 0u; }
#endif

// Column-ordered pair emission: (row << 8 | column) must fit the 16-bit key.  Measured on MI355X, 1 M Gaussians at
// 1080p, whole sort (tools/sort_routes.py): 8.4 pairs per Gaussian 0.249 ms against 0.218 ms for emission in depth
// order + two radix passes; 14.8: 0.273 / 0.294; 23.5: 0.331 / 0.423; 45.2: 0.513 / 0.706 - it costs more per
// Gaussian and 7 instead of 13 us per million pairs, and pays from ~13 pairs per Gaussian (dense views, close-ups).
inline bool column_path(int ntx, int nty) { return ntx <= 256 && nty <= 256; }
#ifdef CUGS_DEV
std::atomic<int> g_col_min_ratio{13};             // development build: movable, to measure both routes on one view
inline bool column_path_pays(uint32_t n, uint32_t pairs) {
    return (unsigned long long)pairs >= (unsigned long long)g_col_min_ratio.load(std::memory_order_relaxed) * n;
}
#else
inline bool column_path_pays(uint32_t n, uint32_t pairs) { return (unsigned long long)pairs >= 13ull * n; }
#endif

int tile_bits(int tiles) {
    int b = 1;
    while ((1 << b) < tiles) ++b;
    return b;
}

// Steps (2b)-(4) of cugs_sort_pairs for one tile-id width.
template <typename K>
int sort_pairs_typed(const SortWsN& ws, const SortWsP& wp, uint32_t un, uint32_t up, const float* means_2d,
                     const float* depths, const int32_t* radii, const int32_t* tiles_touched, int width, int height,
                     int ntx, int nty, uint64_t* keys_sorted, int32_t* values_sorted, int32_t* tile_ranges,
                     const unsigned long long* dev_count, hipStream_t st, bool direct,
                     unsigned long long* total_mapped = nullptr, uint32_t* tile_order = nullptr) {
    const int tiles = ntx * nty;
    const uint32_t* order = ws.dval[1];                 // left there by cugs_sort_count_pairs
    if (direct) {
        // (3b) every pair straight to its place (queue_count(direct) left the prefixes, the tile starts and the depth-ordered
        // packed rectangles in the N-level workspace); the pair-level workspace is not used
        const uint32_t* zsnap = reinterpret_cast<const uint32_t*>(ws.total) + 10;       // [0] Q12 pairs, [1] depth range flag
        // blocks of 8 x 8 tiles, one per wave; workgroups of up to 8 horizontally adjacent blocks, evenly filled
        // (15 block columns: 8 + 7).  Launched for capacity 0 too: the kernel publishes the totals and the ranges.
        const uint32_t nbx = ((uint32_t)ntx + BIN_BLK - 1u) / BIN_BLK, nby = ((uint32_t)nty + BIN_BLK - 1u) / BIN_BLK;
        const uint32_t gxs = bin_window_groups(ntx);
        const uint32_t waves = bin_window_cols(ntx) / BIN_BLK;
        const uint32_t* const win = bin_windows(ntx, nty) <= BIN_WINDOWS_MAX ? ws.bin_win : nullptr;
#ifdef CUGS_DEV
        if (hipEvent_t mark = static_cast<hipEvent_t>(g_mark_event.load(std::memory_order_relaxed)))   // tools/late_colour.py
            CUGS_RETURN_IF_HIP(hipEventRecord(mark, st));
#endif
        // (the capacity stands for the pair count in the choice of the variant: it follows the previous frames' counts)
        const bool staged = (unsigned long long)up >= (unsigned long long)BIN_STAGE_RATIO * un && bin_stage_enabled();
#define CUGS_LAUNCH_SCATTER(S)                                                                                                        \
        hipLaunchKernelGGL(k_bin_scatter<S>, dim3(bin_rows(un) * nby * gxs), dim3(waves * CUGS_WAVE), 0, st, un, bin_group(), nbx, nby, gxs, \
                           up, dev_count != nullptr, order, static_cast<const uint32_t*>(ws.prect[1]), (uint32_t)ntx, (uint32_t)nty,      \
                           ws.bin_table, ws.bin_ttot, ws.bin_tpre, ws.bin_csum, zsnap, ws.bin_tbase, ws.total, total_mapped,             \
                           reinterpret_cast<uint32_t*>(values_sorted), tile_ranges, tile_order, win, bin_ablate())
        if (staged) CUGS_LAUNCH_SCATTER(true); else CUGS_LAUNCH_SCATTER(false);
#undef CUGS_LAUNCH_SCATTER
        CUGS_LAUNCH_CHECK();
        if (keys_sorted) {
            hipLaunchKernelGGL(k_bin_keys, dim3(nblocks_for(up, CUGS_BLOCK)), dim3(CUGS_BLOCK), 0, st, up, dev_count, (uint32_t)tiles,
                               ws.bin_tbase, zsnap, values_sorted, depths, keys_sorted);
            CUGS_LAUNCH_CHECK();
        }
        return 0;
    }
    const uint32_t nfill = nblocks_for(un, FILL_CHUNK);
    uint32_t* ctl = reinterpret_cast<uint32_t*>(ws.total) + 4;        // [0] Q12 counter, [1] its snapshot
    const int bits = tile_bits(tiles);
    const int npass = (bits + 7) / 8;
    const int per = (bits + npass - 1) / npass;
    K* tk[2] = {static_cast<K*>(wp.ptile[0]), static_cast<K*>(wp.ptile[1])};
    uint32_t* tv[2] = {wp.pidx[0], wp.pidx[1]};
    uint32_t* vals_final = reinterpret_cast<uint32_t*>(values_sorted);
    if constexpr (sizeof(K) == 2) {
        if (column_path(ntx, nty) && column_path_pays(un, up)) {
            // pairs emitted in tile-column order (row << 8 | column keys), then ONE stable pass by row
            const uint32_t ncol = nblocks_for(un, COL_CHUNK);
            hipLaunchKernelGGL(k_col_hist, dim3(ncol), dim3(COL_CHUNK), 0, st, un, ws.rect[1], ws.colhist, ncol);
            CUGS_LAUNCH_CHECK();
            hipLaunchKernelGGL(k_radix_scan_rows, dim3(RADIX), dim3(CUGS_BLOCK), 0, st, ws.colhist, ws.colscan, ncol, ws.tot);
            CUGS_LAUNCH_CHECK();
            hipLaunchKernelGGL(k_col_emit, dim3(ncol), dim3(COL_CHUNK), 0, st, un, up, dev_count, order, ws.rect[1],
                               ws.colscan, ws.tot, ncol, tk[0], tv[0], ctl, tile_ranges, (uint32_t)(2 * tiles), wp.sup,
                               sup_used(nblocks_for(up, CHUNK_PAIR), RADIX, 512u >> tile_bits(nty)));
            CUGS_LAUNCH_CHECK();
            int rc = radix_pass<K, false, 512, CHUNK_PAIR>(tk[0], tv[0], up, dev_count, 8, tile_bits(nty), wp.hist, wp.sup,
                                                           ws.tot, tk[1], vals_final, ctl, st);
            if (rc) return rc;
            hipLaunchKernelGGL((k_tile_ranges<K, true>), dim3(nblocks_for(up, CUGS_BLOCK * 8)), dim3(CUGS_BLOCK), 0, st, up,
                               dev_count, tk[1], (uint32_t)ntx, values_sorted, depths, tile_ranges, keys_sorted, ctl + 1);
            CUGS_LAUNCH_CHECK();
            return 0;
        }
    }
    if (npass > SUP_TABLES) return CUGS_EINVAL;
    // each pass's super table (if that pass runs scan-free: radix_pass applies the same rule), back to back
    const uint32_t nblk_p = nblocks_for(up, CHUNK_PAIR);
    uint32_t sup_off[SUP_TABLES + 1] = {0u};
    for (int p = 0; p < npass; ++p) {
        const int shift = p * per;
        const int b = (bits - shift) < per ? (bits - shift) : per;
        sup_off[p + 1] = sup_off[p] + sup_used(nblk_p, RADIX, 512u >> b);
    }
    hipLaunchKernelGGL((k_fill_pairs<K>), dim3(nfill), dim3(CUGS_BLOCK), 0, st, un, up, dev_count, order, ws.rect[1], ntx,
                       ws.blocksum, tk[0], tv[0], ctl, tile_ranges, (uint32_t)(2 * tiles), wp.sup, sup_off[npass]);
    CUGS_LAUNCH_CHECK();
    int cur = 0, rc;
    for (int p = 0; p < npass; ++p) {
        const int shift = p * per;
        const int b = (bits - shift) < per ? (bits - shift) : per;
        uint32_t* vout = (p == npass - 1) ? vals_final : tv[cur ^ 1];
        rc = radix_pass<K, false, 512, CHUNK_PAIR>(tk[cur], tv[cur], up, dev_count, shift, b, wp.hist, wp.sup + sup_off[p], ws.tot,
                                                   tk[cur ^ 1], vout, p == 0 ? ctl : nullptr, st);   // 512 threads: measured best of 256/512/1024
        if (rc) return rc;
        cur ^= 1;
    }
    hipLaunchKernelGGL((k_tile_ranges<K, false>), dim3(nblocks_for(up, CUGS_BLOCK * 8)), dim3(CUGS_BLOCK), 0, st, up, dev_count, tk[cur],
                       (uint32_t)ntx, values_sorted, depths, tile_ranges, keys_sorted, ctl + 1);
    CUGS_LAUNCH_CHECK();
    return 0;
}

// Steps (1)-(2a): everything that does not depend on the pair count.  Queued, never blocks.
// three_pass: the depth sort on 27-bit offsets from the near plane (see RADIX_DEPTH); if a depth key turns out to lie
// outside that range the totals say so (k_scan_blocksums) and the caller runs this again with three_pass = false.
int queue_count(const SortWsN& ws, uint32_t un, const float* means_2d, const float* depths, const int32_t* radii,
                const int32_t* tiles_touched, int width, int height, int ntx, int nty, hipStream_t st,
                unsigned long long* total_mapped = nullptr, bool three_pass = true, bool prekeyed = false,
                bool direct = false) {
    uint32_t* range_flag = reinterpret_cast<uint32_t*>(ws.total) + 6;
    uint32_t* const q12 = reinterpret_cast<uint32_t*>(ws.total) + 4;   // [0] Q12 counter, [1] its snapshot
    int rc;
    bool riding = false;
    if (direct && !(three_pass && prekeyed && bin_route(ntx, nty))) return CUGS_EINVAL;
    if (three_pass) {
        // (1) stable sort of the Gaussians by depth: keys -> dkey[0], three passes [0] -> [1] -> [0] -> [1]
        // prekeyed: cugs_project_forward_keyed has left dkey[0], rect[0] and the range flag in this workspace already
        if (!prekeyed) {
            hipLaunchKernelGGL(k_depth_keys_rect, dim3(nblocks_for(un, CUGS_BLOCK)), dim3(CUGS_BLOCK), 0, st, un, depths,
                               means_2d, radii, tiles_touched, width, height, ntx, nty, ws.dkey[0], ws.rect[0], range_flag,
                               ws.sup, 3u * sup_used(nblocks_for(un, CHUNK_DEPTH), RADIX_DEPTH, 1024u >> DEPTH_BITS));
            CUGS_LAUNCH_CHECK();
        }
        // prekeyed on an image of up to 127 x 127 tiles: the projection left PACKED rectangles (prect[0]) and they ride
        // through the passes beside the index, [0] -> [1] -> [0] -> [1]
        // ... when the passes' workgroups are at most one per CU anyway: the second value stream takes the scatter's LDS
        // from 68 to 84 KB, i.e. from two resident workgroups per CU to one - free at 1 M Gaussians (245 workgroups on
        // 256 CUs: sort 0.2285 -> 0.2239 ms, projection -1.5 us, same box), a loss at 6 M (1465 workgroups), where the
        // gather stays (profiles/r03_j_packed_rect_ride_ab.log)
        riding = prekeyed && cugs_prect_packable(ntx, nty) && nblocks_for(un, CHUNK_MIN) <= 256u;
        uint32_t* const* pr = ws.prect;
        const uint32_t used = sup_used(nblocks_for(un, CHUNK_DEPTH), RADIX_DEPTH, 1024u >> DEPTH_BITS);
        uint32_t* const sup0 = ws.sup, *const sup1 = ws.sup + used, *const sup2 = ws.sup + 2 * (size_t)used;
        if ((rc = radix_pass<uint32_t, true, 1024, CHUNK_DEPTH, RADIX_DEPTH>(ws.dkey[0], nullptr, un, nullptr, 0, DEPTH_BITS, ws.hist, sup0, ws.tot, ws.dkey[1], ws.dval[1], nullptr, st, riding ? pr[0] : nullptr, pr[1]))) return rc;
        if ((rc = radix_pass<uint32_t, false, 1024, CHUNK_DEPTH, RADIX_DEPTH>(ws.dkey[1], ws.dval[1], un, nullptr, DEPTH_BITS, DEPTH_BITS, ws.hist, sup1, ws.tot, ws.dkey[0], ws.dval[0], nullptr, st, riding ? pr[1] : nullptr, pr[0]))) return rc;
        // direct: the last pass's histogram kernel also arms the Q12 counter k_bin_count adds to
        if ((rc = radix_pass<uint32_t, false, 1024, CHUNK_DEPTH, RADIX_DEPTH>(ws.dkey[0], ws.dval[0], un, nullptr, 2 * DEPTH_BITS, DEPTH_BITS, ws.hist, sup2, ws.tot, ws.dkey[1], ws.dval[1], direct ? q12 : nullptr, st, riding ? pr[0] : nullptr, pr[1]))) return rc;
    } else {
        // (1) the general route: four passes of 8 bits on the raw depth bits (positive floats order as unsigned ints)
        hipLaunchKernelGGL(k_depth_keys_rect, dim3(nblocks_for(un, CUGS_BLOCK)), dim3(CUGS_BLOCK), 0, st, un, depths,
                           means_2d, radii, tiles_touched, width, height, ntx, nty, ws.dkey[1], ws.rect[0],
                           static_cast<uint32_t*>(nullptr), ws.sup, 4u * sup_used(nblocks_for(un, CHUNK_MIN), RADIX, 1024u >> 8));
        CUGS_LAUNCH_CHECK();
        uint32_t* sp[SUP_TABLES];
        for (int t = 0; t < SUP_TABLES; ++t) sp[t] = ws.sup + (size_t)t * sup_used(nblocks_for(un, CHUNK_MIN), RADIX, 1024u >> 8);
        if ((rc = radix_pass<uint32_t, true, 1024, CHUNK_MIN>(ws.dkey[1], nullptr, un, nullptr, 0, 8, ws.hist, sp[0], ws.tot, ws.dkey[0], ws.dval[0], nullptr, st))) return rc;
        if ((rc = radix_pass<uint32_t, false, 1024, CHUNK_MIN>(ws.dkey[0], ws.dval[0], un, nullptr, 8, 8, ws.hist, sp[1], ws.tot, ws.dkey[1], ws.dval[1], nullptr, st))) return rc;
        if ((rc = radix_pass<uint32_t, false, 1024, CHUNK_MIN>(ws.dkey[1], ws.dval[1], un, nullptr, 16, 8, ws.hist, sp[2], ws.tot, ws.dkey[0], ws.dval[0], nullptr, st))) return rc;
        if ((rc = radix_pass<uint32_t, false, 1024, CHUNK_MIN>(ws.dkey[0], ws.dval[0], un, nullptr, 24, 8, ws.hist, sp[3], ws.tot, ws.dkey[1], ws.dval[1], nullptr, st))) return rc;
    }
    if (direct) {
        // (2)-(3a) direct binning: pairs per (workgroup of the depth order, tile), their prefixes, tile starts, the total
        // (the totals are published by the scatter: sort_pairs_typed(direct), which the caller launches in any case)
        const uint32_t tiles = (uint32_t)(ntx * nty), rows = bin_rows(un);
        uint32_t* const snap = reinterpret_cast<uint32_t*>(ws.total) + 10;   // [0] Q12 pairs, [1] depth range flag
        hipLaunchKernelGGL(k_bin_count, dim3(rows), dim3(BIN_NT), 0, st, un, bin_group(), ws.dval[1], ws.rect[0],
                           riding ? static_cast<const uint32_t*>(ws.prect[1]) : static_cast<const uint32_t*>(nullptr),
                           riding ? static_cast<uint32_t*>(nullptr) : ws.prect[1], (uint32_t)ntx, (uint32_t)nty, ws.bin_table, q12,
                           ws.bin_win);
        CUGS_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_bin_scan, dim3((tiles + CUGS_WAVE - 1) / CUGS_WAVE), dim3(BIN_NT), 0, st, rows, tiles, ws.bin_table,
                           ws.bin_ttot, ws.bin_tpre, ws.bin_csum, q12, range_flag, snap, bin_windows(ntx, nty) <= BIN_WINDOWS_MAX ? ws.bin_win : nullptr,
                           (uint32_t)ntx, bin_window_cols(ntx), bin_window_groups(ntx));
        CUGS_LAUNCH_CHECK();
        return 0;
    }
    // (2a) pair counts per 256-Gaussian block in depth order, their scan, and the grand total
    const uint32_t nfill = nblocks_for(un, FILL_CHUNK);
    hipLaunchKernelGGL(k_fill_blocksums, dim3(nfill), dim3(CUGS_BLOCK), 0, st, un, ws.dval[1], ws.rect[0], ws.rect[1],
                       ws.blocksum, riding ? static_cast<const uint32_t*>(ws.prect[1]) : static_cast<const uint32_t*>(nullptr));
    CUGS_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_scan_blocksums, dim3(1), dim3(SCAN_NT), 0, st, ws.blocksum, nfill, ws.total,
                       reinterpret_cast<uint32_t*>(ws.total) + 4, total_mapped, three_pass ? range_flag : static_cast<uint32_t*>(nullptr));
    CUGS_LAUNCH_CHECK();
    return 0;
}

#ifdef CUGS_DEV
std::atomic<int> g_direct_route{1};               // development build: 0 = the radix route on every view (A/B measurements)
inline bool direct_route_enabled() { return g_direct_route.load(std::memory_order_relaxed) != 0; }
#else
constexpr bool direct_route_enabled() { return true; }
#endif

template <typename... A>
int sort_pairs_dispatch(int tiles, A... args) {
    if (tile_bits(tiles) <= 16) return sort_pairs_typed<uint16_t>(args...);   // always when column_path() holds
    return sort_pairs_typed<uint32_t>(args...);
}

}  // namespace

int cugs_sort_key_slots(void* workspace, size_t bytes, int64_t n, int width, int height, uint32_t** keys, int4** rect,
                        uint32_t** prect, uint32_t** range_flag, uint32_t** zero, uint32_t* nzero) {
    if (!workspace || n < 0 || n > 2147483647ll || width < 0 || height < 0) return CUGS_EINVAL;
    if ((width + CUGS_TILE - 1) / CUGS_TILE > 32767 || (height + CUGS_TILE - 1) / CUGS_TILE > 32767) return CUGS_EOVERFLOW;
    SortWsN ws = carve_n(workspace, n);
    if (bytes < ws.bytes) return CUGS_EWORKSPACE;
    *keys = ws.dkey[0];
    // the same rule queue_count applies when it picks up what the projection left (prekeyed)
    const bool packed = cugs_prect_packable((width + CUGS_TILE - 1) / CUGS_TILE, (height + CUGS_TILE - 1) / CUGS_TILE) &&
                        nblocks_for(n, CHUNK_MIN) <= 256u;
    *rect = packed ? nullptr : ws.rect[0];
    *prect = packed ? ws.prect[0] : nullptr;
    *range_flag = reinterpret_cast<uint32_t*>(ws.total) + 6;
    *zero = ws.sup;                                // the key kernel also clears the depth passes' super tables
    *nzero = 3u * sup_used(nblocks_for(n, CHUNK_DEPTH), RADIX_DEPTH, 1024u >> DEPTH_BITS);       // the three passes of the fast depth route
    return 0;
}

extern "C" size_t cugs_sort_workspace_bytes(int64_t n) {
    if (n < 0) return 0;
    return carve_n(nullptr, n).bytes;
}

extern "C" size_t cugs_sort_pair_workspace_bytes(int64_t total_pairs) {
    if (total_pairs < 0) return 0;
    return carve_p(nullptr, total_pairs).bytes;
}

// Everything that does not depend on the pair count runs BEFORE the blocking read-back, so the
// device is busy (depth keys, the depth sort, per-block pair sums and their scan) while the
// host waits for the 8-byte total - the reference idles on cumsum[-1].item() instead (sorting.cu:146).
namespace {
int sort_count_pairs_impl(int64_t n, const float* means_2d, const float* depths,
                                     const int32_t* radii, const int32_t* tiles_touched, int width,
                                     int height, void* workspace, size_t workspace_bytes,
                                     int64_t* total_pairs_host, void* stream, bool wide) {
    if (n < 0 || width < 0 || height < 0 || !total_pairs_host) return CUGS_EINVAL;
    *total_pairs_host = 0;
    if (n == 0) return 0;
    if (n > 2147483647ll) return CUGS_EOVERFLOW;
    if (!means_2d || !depths || !radii || !tiles_touched || !workspace) return CUGS_EINVAL;
    SortWsN ws = carve_n(workspace, n);
    if (workspace_bytes < ws.bytes) return CUGS_EWORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int ntx = (width + CUGS_TILE - 1) / CUGS_TILE, nty = (height + CUGS_TILE - 1) / CUGS_TILE;
    if (ntx > 32767 || nty > 32767) return CUGS_EOVERFLOW;          // rectangle extents travel as 16-bit halves
    // wide: the caller knows this view's depths leave the range of the three-pass depth sort (an earlier sort of it
    // said so): the general four-pass route at once instead of a wasted three-pass attempt
    int rc = queue_count(ws, (uint32_t)n, means_2d, depths, radii, tiles_touched, width, height, ntx, nty, st, nullptr, !wide);
    if (rc) return rc;
#ifdef CUGS_DEV
    // development build: first blocking sort of the process verifies the LDS ordering the atomic ranking relies on
    const bool probing = rank_mode() < 0;
    uint32_t* probe_word = reinterpret_cast<uint32_t*>(ws.total) + 8;
    uint32_t violations = 1;
    if (probing) {
        CUGS_RETURN_IF_HIP(hipMemsetAsync(probe_word, 0, sizeof(uint32_t), st));
        hipLaunchKernelGGL(k_probe_lds_order, dim3(96), dim3(CUGS_BLOCK), 0, st, probe_word);
        CUGS_LAUNCH_CHECK();
        CUGS_RETURN_IF_HIP(hipMemcpyAsync(&violations, probe_word, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    }
#endif
    // straight into the caller's variable: if that is pinned host memory the copy is one DMA, no staging
    CUGS_RETURN_IF_HIP(hipMemcpyAsync(total_pairs_host, ws.total + 1, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    CUGS_RETURN_IF_HIP(hipStreamSynchronize(st));
    if (*total_pairs_host == -1) {
        // a depth key outside the range of the three-pass depth sort (a view with splats nearer than the near plane
        // or farther than ~13 000 units): the general four-pass route, once more
        rc = queue_count(ws, (uint32_t)n, means_2d, depths, radii, tiles_touched, width, height, ntx, nty, st, nullptr, false);
        if (rc) return rc;
        CUGS_RETURN_IF_HIP(hipMemcpyAsync(total_pairs_host, ws.total + 1, sizeof(int64_t), hipMemcpyDeviceToHost, st));
        CUGS_RETURN_IF_HIP(hipStreamSynchronize(st));
    }
#ifdef CUGS_DEV
    if (probing) g_rank_mode.store(violations == 0 ? 1 : 0, std::memory_order_relaxed);
#endif
    if ((unsigned long long)*total_pairs_host > 2147483647ull) {   // the reference indexes pairs with int
        *total_pairs_host = 0;
        return CUGS_EOVERFLOW;
    }
    return 0;
}
}  // namespace

extern "C" int cugs_sort_count_pairs(int64_t n, const float* means_2d, const float* depths,
                                     const int32_t* radii, const int32_t* tiles_touched, int width,
                                     int height, void* workspace, size_t workspace_bytes,
                                     int64_t* total_pairs_host, void* stream) {
    return sort_count_pairs_impl(n, means_2d, depths, radii, tiles_touched, width, height, workspace, workspace_bytes,
                                 total_pairs_host, stream, false);
}

extern "C" int cugs_sort_count_pairs_wide(int64_t n, const float* means_2d, const float* depths,
                                          const int32_t* radii, const int32_t* tiles_touched, int width,
                                          int height, void* workspace, size_t workspace_bytes,
                                          int64_t* total_pairs_host, void* stream) {
    return sort_count_pairs_impl(n, means_2d, depths, radii, tiles_touched, width, height, workspace, workspace_bytes,
                                 total_pairs_host, stream, true);
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
    if (n == 0 || total_pairs == 0 || tiles == 0) {              // sorting.cu:133-139,154-160
        if (tiles > 0)   // every tile stays {0,0} (sorting.cu:216)
            CUGS_RETURN_IF_HIP(hipMemsetAsync(tile_ranges, 0, sizeof(int32_t) * 2 * (size_t)tiles, st));
        return 0;
    }
    if (!means_2d || !depths || !radii || !tiles_touched || !values_sorted || !workspace || !pair_workspace)
        return CUGS_EINVAL;
    SortWsN ws = carve_n(workspace, n);
    SortWsP wp = carve_p(pair_workspace, total_pairs);
    if (workspace_bytes < ws.bytes || pair_workspace_bytes < wp.bytes) return CUGS_EWORKSPACE;

    // (2b) pairs in depth order; (3) stable sort by tile id, last pass landing in values_sorted; (4) ranges
    return sort_pairs_dispatch(tiles, ws, wp, (uint32_t)n, (uint32_t)total_pairs, means_2d, depths, radii, tiles_touched,
                               width, height, ntx, nty, keys_sorted, values_sorted, tile_ranges,
                               static_cast<const unsigned long long*>(nullptr), st, false);
}

// The whole sort without a host round trip: the caller PREDICTS the pair count (`capacity`, e.g. the last
// frame's count plus a margin), sizes pair_workspace / keys_sorted / values_sorted for it, and every
// pair-level kernel takes the live count from device memory.  The 8-byte total is copied to
// *total_pairs_host asynchronously (use pinned memory); once the stream (or an event recorded after this
// call) has completed, the results are valid iff 0 <= *total_pairs_host <= capacity - otherwise call
// cugs_sort_pairs with the now known count (the N-level workspace still holds the depth order).  The
// ~45 us the device idles in cugs_sort_count_pairs + cugs_sort_pairs while the host reads the total
// and launches the rest (3 % of a 1 M-Gaussian 1080p frame) disappear.
namespace {
int sort_pairs_predicted_impl(int64_t n, int64_t capacity, const float* means_2d, const float* depths,
                              const int32_t* radii, const int32_t* tiles_touched, int width, int height,
                              void* workspace, size_t workspace_bytes, void* pair_workspace,
                              size_t pair_workspace_bytes, uint64_t* keys_sorted, int32_t* values_sorted,
                              int32_t* tile_ranges, int64_t* total_pairs_host, void* stream, bool prekeyed,
                              bool wide = false, uint32_t* tile_order = nullptr) {
    if (n < 0 || capacity < 0 || width < 0 || height < 0 || !tile_ranges || !total_pairs_host) return CUGS_EINVAL;
    if (n > 2147483647ll || capacity > 2147483647ll) return CUGS_EOVERFLOW;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int ntx = (width + CUGS_TILE - 1) / CUGS_TILE, nty = (height + CUGS_TILE - 1) / CUGS_TILE;
    const int tiles = ntx * nty;
    if (ntx > 32767 || nty > 32767) return CUGS_EOVERFLOW;
    *total_pairs_host = 0;
    // tile_order (optional): every exit that leaves valid ranges also leaves a valid order of the tiles
    auto order_from_ranges = [&]() -> int {
        if (!tile_order) return 0;
        hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, st, (uint32_t)tiles, tile_ranges, reinterpret_cast<uint4*>(tile_order));
        CUGS_LAUNCH_CHECK();
        return 0;
    };
    if (n == 0 || tiles == 0) {
        if (tiles > 0) {
            CUGS_RETURN_IF_HIP(hipMemsetAsync(tile_ranges, 0, sizeof(int32_t) * 2 * (size_t)tiles, st));
            return order_from_ranges();
        }
        return 0;
    }
    if (!means_2d || !depths || !radii || !tiles_touched || !workspace) return CUGS_EINVAL;
    SortWsN ws = carve_n(workspace, n);
    if (workspace_bytes < ws.bytes) return CUGS_EWORKSPACE;
    // pinned host memory is mapped into the device's address space: the scan kernel then stores the total there
    // itself; anything else (pageable memory) gets the asynchronous copy
    unsigned long long* mapped = nullptr;
    {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, total_pairs_host) == hipSuccess && attr.type == hipMemoryTypeHost &&
            attr.devicePointer)
            mapped = static_cast<unsigned long long*>(attr.devicePointer);
        else
            (void)hipGetLastError();                              // an unregistered pointer is not an error here
    }
    // the projection's own records on an image of up to ~10 000 tiles: every pair is written once, by a counting sort
    // over the tiles (k_bin_*), instead of emitted and carried through two radix passes
    // (capacity below 2^30: the scatter keeps 32-bit byte offsets into the index buffer)
    const bool direct = prekeyed && !wide && bin_route(ntx, nty) && bin_route_n(n) && capacity < (int64_t(1) << 30) &&
                        direct_route_enabled();
    int rc = queue_count(ws, (uint32_t)n, means_2d, depths, radii, tiles_touched, width, height, ntx, nty, st, mapped, !wide,
                         prekeyed && !wide, direct);
    if (rc) return rc;
    if (direct) {
        // the scatter publishes the totals (and, with capacity 0, only does that and clears the ranges)
        if (capacity > 0 && !values_sorted) return CUGS_EINVAL;
        SortWsP none{};
        rc = sort_pairs_dispatch(tiles, ws, none, (uint32_t)n, (uint32_t)capacity, means_2d, depths, radii, tiles_touched,
                                 width, height, ntx, nty, capacity > 0 ? keys_sorted : static_cast<uint64_t*>(nullptr),
                                 values_sorted, tile_ranges, static_cast<const unsigned long long*>(ws.total), st, true, mapped,
                                 tile_order);
        if (rc) return rc;
        if (!mapped)
            CUGS_RETURN_IF_HIP(hipMemcpyAsync(total_pairs_host, ws.total + 1, sizeof(int64_t), hipMemcpyDeviceToHost, st));
        return 0;
    }
    if (!mapped)
        CUGS_RETURN_IF_HIP(hipMemcpyAsync(total_pairs_host, ws.total + 1, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    if (capacity == 0) {                                          // valid iff the total turns out to be 0
        CUGS_RETURN_IF_HIP(hipMemsetAsync(tile_ranges, 0, sizeof(int32_t) * 2 * (size_t)tiles, st));
        return order_from_ranges();
    }
    if (!values_sorted || !pair_workspace) return CUGS_EINVAL;
    SortWsP wp = carve_p(pair_workspace, capacity);
    if (pair_workspace_bytes < wp.bytes) return CUGS_EWORKSPACE;
    rc = sort_pairs_dispatch(tiles, ws, wp, (uint32_t)n, (uint32_t)capacity, means_2d, depths, radii, tiles_touched,
                             width, height, ntx, nty, keys_sorted, values_sorted, tile_ranges,
                             static_cast<const unsigned long long*>(ws.total), st, false);
    if (rc) return rc;
    return order_from_ranges();      // the radix route's ranges come from the (clamped) sorted pairs: always a valid partition
}
}  // namespace

extern "C" int cugs_sort_pairs_predicted(int64_t n, int64_t capacity, const float* means_2d, const float* depths,
                                         const int32_t* radii, const int32_t* tiles_touched, int width, int height,
                                         void* workspace, size_t workspace_bytes, void* pair_workspace,
                                         size_t pair_workspace_bytes, uint64_t* keys_sorted, int32_t* values_sorted,
                                         int32_t* tile_ranges, int64_t* total_pairs_host, void* stream) {
    return sort_pairs_predicted_impl(n, capacity, means_2d, depths, radii, tiles_touched, width, height, workspace,
                                     workspace_bytes, pair_workspace, pair_workspace_bytes, keys_sorted, values_sorted,
                                     tile_ranges, total_pairs_host, stream, false);
}

// cugs_sort_pairs_predicted for a `workspace` that cugs_project_forward_keyed has filled on this stream, for these very
// arrays, since the last sort that used it: the per-Gaussian key / rectangle kernel (one launch, 40 MB per million
// Gaussians) is skipped.  Everything else, the validity rule and the fallback (cugs_sort_count_pairs + cugs_sort_pairs,
// which rebuild the keys from the arrays) are those of cugs_sort_pairs_predicted.
extern "C" int cugs_sort_pairs_predicted_keyed(int64_t n, int64_t capacity, const float* means_2d, const float* depths,
                                               const int32_t* radii, const int32_t* tiles_touched, int width, int height,
                                               void* workspace, size_t workspace_bytes, void* pair_workspace,
                                               size_t pair_workspace_bytes, uint64_t* keys_sorted, int32_t* values_sorted,
                                               int32_t* tile_ranges, int64_t* total_pairs_host, void* stream) {
    return sort_pairs_predicted_impl(n, capacity, means_2d, depths, radii, tiles_touched, width, height, workspace,
                                     workspace_bytes, pair_workspace, pair_workspace_bytes, keys_sorted, values_sorted,
                                     tile_ranges, total_pairs_host, stream, true);
}

// cugs_sort_pairs_predicted_keyed that also leaves, in tile_order[tiles][4], the tiles ordered by the length of their lists,
// longest first ({tile, first pair, one past the last pair, 0} each): what cugs_rasterize_forward_ordered / cugs_rasterize_backward_ordered hand their workgroups out by.
extern "C" int cugs_sort_pairs_predicted_keyed_ordered(int64_t n, int64_t capacity, const float* means_2d, const float* depths,
                                                       const int32_t* radii, const int32_t* tiles_touched, int width,
                                                       int height, void* workspace, size_t workspace_bytes,
                                                       void* pair_workspace, size_t pair_workspace_bytes,
                                                       uint64_t* keys_sorted, int32_t* values_sorted, int32_t* tile_ranges,
                                                       int64_t* total_pairs_host, uint32_t* tile_order, void* stream) {
    if (!tile_order) return CUGS_EINVAL;
    if (reinterpret_cast<uintptr_t>(tile_order) & 15u) return CUGS_EALIGN;
    return sort_pairs_predicted_impl(n, capacity, means_2d, depths, radii, tiles_touched, width, height, workspace,
                                     workspace_bytes, pair_workspace, pair_workspace_bytes, keys_sorted, values_sorted,
                                     tile_ranges, total_pairs_host, stream, true, false, tile_order);
}

// The same order from any valid tile_ranges (e.g. after cugs_sort_pairs): one small launch.
extern "C" int cugs_tile_order(int width, int height, const int32_t* tile_ranges, uint32_t* tile_order, void* stream) {
    if (width < 0 || height < 0) return CUGS_EINVAL;
    const int64_t tiles = (int64_t)((width + CUGS_TILE - 1) / CUGS_TILE) * ((height + CUGS_TILE - 1) / CUGS_TILE);
    if (tiles == 0) return 0;
    if (!tile_ranges || !tile_order || tiles > 2147483647ll) return CUGS_EINVAL;
    if (reinterpret_cast<uintptr_t>(tile_order) & 15u) return CUGS_EALIGN;
    hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), (uint32_t)tiles, tile_ranges,
                       reinterpret_cast<uint4*>(tile_order));
    CUGS_LAUNCH_CHECK();
    return 0;
}

// cugs_sort_pairs_predicted on the GENERAL depth route (four 8-bit passes over the raw depth bits): for views whose
// depths leave the range of the three-pass sort - an earlier sort of the view reported -1.  Never reports -1 itself.
extern "C" int cugs_sort_pairs_predicted_wide(int64_t n, int64_t capacity, const float* means_2d, const float* depths,
                                              const int32_t* radii, const int32_t* tiles_touched, int width, int height,
                                              void* workspace, size_t workspace_bytes, void* pair_workspace,
                                              size_t pair_workspace_bytes, uint64_t* keys_sorted, int32_t* values_sorted,
                                              int32_t* tile_ranges, int64_t* total_pairs_host, void* stream) {
    return sort_pairs_predicted_impl(n, capacity, means_2d, depths, radii, tiles_touched, width, height, workspace,
                                     workspace_bytes, pair_workspace, pair_workspace_bytes, keys_sorted, values_sorted,
                                     tile_ranges, total_pairs_host, stream, false, true);
}

#ifdef CUGS_DEV
// Development build only: a hipEvent_t the keyed predicted sort records on its stream right before k_bin_scatter (NULL: none).
extern "C" int cugsdbg_sort_mark_event(void* event) { g_mark_event.store(event, std::memory_order_relaxed); return 0; }
// Development build only: 0 = never take the direct-binning route, 1 = take it where it applies; returns the setting.
extern "C" int cugsdbg_sort_direct_route(int on) {
    if (on == 0 || on == 1) g_direct_route.store(on, std::memory_order_relaxed);
    return g_direct_route.load(std::memory_order_relaxed);
}
// Development build only: pairs per Gaussian from which the column-ordered emission is used (0: always).
extern "C" int cugsdbg_sort_column_ratio(int ratio) {
    if (ratio >= 0) g_col_min_ratio.store(ratio, std::memory_order_relaxed);
    return g_col_min_ratio.load(std::memory_order_relaxed);
}
// Development build only: read (and clear) the per-phase tick sums of k_col_emit.
extern "C" int cugsdbg_emit_profile(unsigned long long out[16]) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_emit_prof), sizeof(unsigned long long) * 16) != hipSuccess) return -100;
    unsigned long long zero[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_emit_prof), zero, sizeof(zero)) != hipSuccess) return -100;
    return 0;
}
// Debug hook (development build only, not part of the ABI header): force (0 = ballot, 1 = atomic) or query (-2) the ranking mode of the
// radix scatter; returns the mode in effect (-1 = not probed yet).
extern "C" int cugsdbg_sort_rank_mode(int mode) {
    if (mode == 0 || mode == 1 || mode == -1) g_rank_mode.store(mode, std::memory_order_relaxed);
    return g_rank_mode.load(std::memory_order_relaxed);
}
#endif
