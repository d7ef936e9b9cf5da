// lds_atomic_lanes.hip — what does an LDS atomic instruction cost on gfx950 when few of its lanes are active, and what do
// scattered 4-byte global stores cost per active lane?  (dev tool, not product; the numbers behind csrc/sort.hip's
// k_bin_count - four atomics per Gaussian through a grid of differences instead of one per covered tile from divergent
// loops - and behind the store cost of k_bin_scatter.)
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_atomic_lanes.bin lds_atomic_lanes.hip ; run on the GPU box.
// Every workgroup: 16 waves, ITERS rounds; in a round each ACTIVE lane does one ds_add_u32 (returnless) on a pseudo-random
// dword of a 32 KB LDS table (or one global_store_dword to a pseudo-random dword of a 64 MB buffer).  Active lanes per
// wave: 64, 32, 16, 8, 4, 1 (the low lanes).  Reported per CU: clocks per wave instruction and per active lane.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITERS = 1024, NT = 1024, TABLE = 8192;

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__global__ __launch_bounds__(NT) void k_lds(uint32_t* out, int active, int same_row) {
    __shared__ uint32_t s_tab[TABLE];
    for (int e = threadIdx.x; e < TABLE; e += NT) s_tab[e] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t h = mix(blockIdx.x * NT + threadIdx.x + 1u);
    if ((int)lane < active) {
        for (int it = 0; it < ITERS; ++it) {
            h = h * 1664525u + 1013904223u;
            // same_row: consecutive lanes hit consecutive dwords of a random row (a tile row segment); else fully random
            const uint32_t idx = same_row ? (((h >> 8) & ~63u) + lane) & (TABLE - 1) : (h >> 8) & (TABLE - 1);
            atomicAdd(&s_tab[idx], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = s_tab[blockIdx.x & (TABLE - 1)];
}

__global__ __launch_bounds__(NT) void k_store(uint32_t* buf, uint32_t mask_words, int active) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t h = mix(blockIdx.x * NT + threadIdx.x + 1u);
    if ((int)lane < active) {
        for (int it = 0; it < ITERS / 4; ++it) {
            h = h * 1664525u + 1013904223u;
            buf[(h >> 4) & mask_words] = h;
        }
    }
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;                       // Hz
    uint32_t* out; CHECK(hipMalloc(&out, sizeof(uint32_t) * cus * 4));
    const uint32_t words = 16u << 20;                              // 64 MB
    uint32_t* buf; CHECK(hipMalloc(&buf, sizeof(uint32_t) * words));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int lanes[] = {64, 32, 16, 8, 4, 1};
    printf("%d CUs, %.0f MHz (nominal): one 1024-thread workgroup per CU, %d rounds\n", cus, clk / 1e6, ITERS);
    for (int pass = 0; pass < 2; ++pass)                           // the first pass warms the clocks
        for (int same = 0; same < 2; ++same)
            for (int a : lanes) {
                for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_lds, dim3(cus), dim3(NT), 0, 0, out, a, same);
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0));
                const int reps = 10;
                for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_lds, dim3(cus), dim3(NT), 0, 0, out, a, same);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
                const double instr_per_cu = 16.0 * ITERS;          // wave instructions per CU
                const double clocks = ms * 1e-3 * clk;
                if (pass) printf("ds_add_u32 %-10s %2d active lanes: %.3f ms  %.1f clocks / wave instruction  %.2f clocks / active lane\n",
                                 same ? "row" : "random", a, ms, clocks / instr_per_cu, clocks / (instr_per_cu * a));
            }
    for (int pass = 0; pass < 2; ++pass)
        for (int a : lanes) {
            for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_store, dim3(cus), dim3(NT), 0, 0, buf, words - 1, a);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            const int reps = 5;
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_store, dim3(cus), dim3(NT), 0, 0, buf, words - 1, a);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
            const double instr_per_cu = 16.0 * ITERS / 4;
            const double clocks = ms * 1e-3 * clk;
            if (pass) printf("global_store_dword random %2d active lanes: %.3f ms  %.1f clocks / wave instruction  %.2f clocks / active lane  %.1f G stores/s chip-wide\n",
                             a, ms, clocks / instr_per_cu, clocks / (instr_per_cu * a), cus * instr_per_cu * a / (ms * 1e-3) / 1e9);
        }
    return 0;
}
