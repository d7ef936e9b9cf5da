// valu_rate.hip — issue-rate microbenchmark for gfx950 VALU instruction classes (dev tool, not product).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate.bin valu_rate.hip ; run on the GPU box.
// For each instruction class and each waves/SIMD setting it prints wave-instructions per SIMD-cycle
// (s_memtime shader cycles) and wall-clock G wave-instr/s chip-wide.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ITERS = 2048;   // loop trips
constexpr int UNROLL = 16;    // instructions per trip (independent chains)

template <int KIND>
__global__ __launch_bounds__(256) void k_rate(float* out, unsigned long long* cyc, float seed) {
    float a[UNROLL];
    float b2[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) { a[i] = seed + i + threadIdx.x; b2[i] = seed * i; }
    const float m = seed * 0.5f + 1.0f, c = seed * 0.25f;
    const bool side = (threadIdx.x & 8) != 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (KIND == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 2) asm volatile("v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]));
            if (KIND == 3) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "s"(__builtin_amdgcn_read_exec()));
            if (KIND == 4) {   // packed fma on register pairs (a[i], b2[i]) built as 64-bit
                if (i % 2 == 0) {
                    double d;  // placeholder type for a 64-bit VGPR pair
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<double*>(&a[i])) : "v"(*reinterpret_cast<const double*>(&b2[i])));
                    (void)d;
                }
            }
            if (KIND == 5) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (KIND == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (KIND == 7) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]));
            if (KIND == 8) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 9) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a[i]), "v"(m) : "vcc");
            if (KIND == 11) { if (i % 2 == 0) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[i + 1])); }
            if (KIND == 12) { if (i % 2 == 0) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[i + 1])); }
            if (KIND == 13) asm volatile("v_add_f32_dpp %0, %1, %0 row_mirror row_mask:0xf bank_mask:0x3" : "+v"(a[i]) : "v"(b2[i]));
            if (KIND == 14) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m));
            if (KIND == 15) asm volatile("v_mov_b32_dpp %0, %1 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b2[i]));
            if (KIND == 16) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (KIND == 17) asm volatile("v_fma_f32 %0, %0, %1, %2 clamp" : "+v"(a[i]) : "v"(m), "v"(c));
            if (KIND == 18) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            if (KIND == 19) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b2[i]));
            if (KIND == 20) asm volatile("v_add_f32 %0, %0, %1 row_bcast:15 row_mask:0xa" : "+v"(a[i]) : "v"(b2[i]));
            if (KIND == 21) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" :: "v"(a[i]), "v"(m) : "s20", "s21");
            if (KIND == 22) asm volatile("v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(a[i]) :: "vcc");
            if (KIND == 23) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 10) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&a[i & ~1])) : "v"(*reinterpret_cast<const double*>(&b2[i & ~1])));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, int n_instr_per_trip) {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    for (int wps : {2, 8}) {
        const int blocks = cus * wps;   // 256 threads = 4 waves = one per SIMD; wps blocks per CU
        float* out; unsigned long long* cyc;
        CHECK(hipMalloc(&out, sizeof(float) * blocks * 256));
        CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * blocks));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        const int reps = 10;
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0f);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        std::vector<unsigned long long> h(blocks);
        CHECK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        const double med = (double)h[blocks / 2];
        const double instr_per_wave = (double)ITERS * n_instr_per_trip;
        // per SIMD: wps waves each issuing instr_per_wave instructions within ~med cycles
        printf("%-28s waves/SIMD %d: %.2f cycles per wave-instr per SIMD (memtime), wall %.3f ms -> %.1f G wave-instr/s chip, eff clock*rate\n",
               name, wps, med / (instr_per_wave * wps), ms, (double)blocks * 4 * instr_per_wave / (ms * 1e-3) / 1e9);
        CHECK(hipFree(out)); CHECK(hipFree(cyc));
    }
}

int main() {
    run<0>("v_fma_f32", UNROLL);
    run<1>("v_mul_f32", UNROLL);
    run<2>("v_add_f32_dpp row_mirror", UNROLL);
    run<7>("v_add_f32_dpp quad_perm", UNROLL);
    run<3>("v_cndmask_b32 (sgpr mask)", UNROLL);
    run<4>("v_pk_fma_f32", UNROLL / 2);
    run<10>("v_pk_mul_f32", UNROLL);
    run<5>("v_exp_f32", UNROLL);
    run<6>("v_rcp_f32", UNROLL);
    run<8>("v_add_u32", UNROLL);
    run<9>("v_cmp_lt_f32 vcc", UNROLL);
    run<11>("v_permlane16_swap", UNROLL / 2);
    run<12>("v_permlane32_swap", UNROLL / 2);
    run<13>("v_add_f32_dpp bank_mask:0x3", UNROLL);
    run<14>("v_cndmask_b32 vcc (e32)", UNROLL);
    run<15>("v_mov_b32_dpp", UNROLL);
    run<16>("v_med3_f32", UNROLL);
    run<17>("v_fma_f32 clamp", UNROLL);
    run<18>("v_add_f32_dpp row_ror:8", UNROLL);
    run<19>("v_mov_b32", UNROLL);
    run<20>("v_add_f32 row_bcast:15", UNROLL);
    run<21>("v_cmp_lt_f32 sgpr pair", UNROLL);
    run<22>("v_addc_co_u32 vcc", UNROLL);
    run<23>("v_max_f32", UNROLL);
    return 0;
}
