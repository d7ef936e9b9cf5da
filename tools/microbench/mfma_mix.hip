// mfma_mix.hip — how much vector issue does v_mfma_f32_16x16x4_f32 leave on its SIMD? (dev tool, not product)
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_mix.bin mfma_mix.hip ; run on the GPU box.
// Case A: every wave runs trips of [1 MFMA + NF independent v_fma_f32], NF = 0..12: cycles per trip.
// Case B: W waves per SIMD, half of them MFMA-only, the other half v_fma-only: each kind's rate beside the other.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ITERS = 1024;

template <int NF>
__global__ void k_mix(float* out, unsigned long long* cyc, float seed) {
    float a[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) a[i] = seed + i + threadIdx.x;
    const float m = seed * 0.5f + 1.0f, c = seed * 0.25f;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    const float av = seed + threadIdx.x, bv = seed * 2.0f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NF; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av, acc1, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NF; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
#pragma unroll
    for (int i = 0; i < 12; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// waves with (wave index / 4) even: MFMA only; odd: VALU only (waves w and w + 4 share a SIMD)
template <bool MF_ON, bool VA_ON>
__global__ void k_split(float* out, unsigned long long* cyc, float seed) {
    const int wave = threadIdx.x >> 6;
    const bool mf = ((wave >> 2) & 1) == 0;
    float a[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) a[i] = seed + i + threadIdx.x;
    const float m = seed * 0.5f + 1.0f, c = seed * 0.25f;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    const float av = seed + threadIdx.x, bv = seed * 2.0f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mf) {
        if (MF_ON)
            for (int it = 0; it < ITERS; ++it) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av, acc1, 0, 0, 0);
            }
    } else {
        if (VA_ON)
            for (int it = 0; it < ITERS; ++it) {
#pragma unroll
                for (int i = 0; i < 12; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
#pragma unroll
    for (int i = 0; i < 12; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

static double avg(const std::vector<unsigned long long>& v, int stride, int off, int period) {
    double s = 0; int n = 0;
    for (size_t i = 0; i < v.size(); ++i) if ((int)((i / stride) % period) == off) { s += (double)v[i]; ++n; }
    return n ? s / n : 0.0;
}

int main() {
    const int grid = 256;
    float* out; unsigned long long* cyc;
    CHECK(hipMalloc(&out, sizeof(float) * grid * 1024));
    CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * grid * 16));
    std::vector<unsigned long long> h(grid * 16);
#define RUN_MIX(NF, THREADS)                                                                          \
    {                                                                                                 \
        hipLaunchKernelGGL(k_mix<NF>, dim3(grid), dim3(THREADS), 0, 0, out, cyc, 1.0f);              \
        CHECK(hipDeviceSynchronize());                                                                \
        hipLaunchKernelGGL(k_mix<NF>, dim3(grid), dim3(THREADS), 0, 0, out, cyc, 1.0f);              \
        CHECK(hipDeviceSynchronize());                                                                \
        const int waves = grid * THREADS / 64;                                                        \
        h.resize(waves);                                                                              \
        CHECK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost));   \
        double s = 0; for (auto v : h) s += (double)v;                                                \
        const double per_wave_trip = s / waves / (2.0 * ITERS);                                       \
        printf("mix  waves/SIMD %d  fillers/MFMA %2d : %.1f cycles per (MFMA + fillers) per wave, %.1f per SIMD\n", \
               THREADS / 256, NF, per_wave_trip, per_wave_trip / (THREADS / 256));                    \
    }
    RUN_MIX(0, 256) RUN_MIX(2, 256) RUN_MIX(4, 256) RUN_MIX(6, 256) RUN_MIX(8, 256) RUN_MIX(12, 256)
    RUN_MIX(0, 1024) RUN_MIX(2, 1024) RUN_MIX(4, 1024) RUN_MIX(6, 1024) RUN_MIX(8, 1024) RUN_MIX(12, 1024)
#define RUN_SPLIT(MF, VA, THREADS)                                                                    \
    {                                                                                                 \
        hipLaunchKernelGGL((k_split<MF, VA>), dim3(grid), dim3(THREADS), 0, 0, out, cyc, 1.0f);      \
        CHECK(hipDeviceSynchronize());                                                                \
        hipLaunchKernelGGL((k_split<MF, VA>), dim3(grid), dim3(THREADS), 0, 0, out, cyc, 1.0f);      \
        CHECK(hipDeviceSynchronize());                                                                \
        const int waves = grid * THREADS / 64;                                                        \
        h.resize(waves);                                                                              \
        CHECK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost));   \
        const double m = avg(h, 4, 0, 2), v = avg(h, 4, 1, 2);                                        \
        printf("split threads %4d  mfma %d valu %d : MFMA waves %.1f cycles/MFMA, VALU waves %.2f cycles/v_fma\n", \
               THREADS, (int)MF, (int)VA, m / (2.0 * ITERS), v / (12.0 * ITERS));                     \
    }
    RUN_SPLIT(true, false, 512) RUN_SPLIT(false, true, 512) RUN_SPLIT(true, true, 512)
    RUN_SPLIT(true, false, 1024) RUN_SPLIT(false, true, 1024) RUN_SPLIT(true, true, 1024)
    return 0;
}
