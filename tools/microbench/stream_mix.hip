// stream_mix.hip — what does the memory side of an MI355X sustain for a per-Gaussian stream that WRITES more than it
// reads?  (dev tool, not product; the yardstick for k_project_backward - 132 B read, 244 B written per Gaussian - and
// for the fused optimizer step: DESIGN.md 4.6 / 4.7 quote "2.9 TB/s of stores" from those two kernels alone.)
// Build: hipcc --offload-arch=gfx950 -O3 -o stream_mix.bin stream_mix.hip ; run on the GPU box.
// Every thread moves 16-byte pieces: R loads from R separate input streams, W stores to W separate output streams, all
// perfectly coalesced (lane i takes piece base + i), one piece per stream per thread, 256-thread workgroups, no
// arithmetic.  Stores and loads plain or non-temporal.  Reported: total bytes / time, and the store share.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float v4f __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int R, int W, bool NT, int PER>
__global__ __launch_bounds__(256) void k_mix(const float4* __restrict__ in, float4* __restrict__ out, size_t pieces) {
    // PER consecutive workgroup-sized chunks per workgroup: 256 * PER pieces of every stream
    const size_t base = (size_t)blockIdx.x * 256u * PER + threadIdx.x;
    float4 v[PER];
#pragma unroll
    for (int p = 0; p < PER; ++p) v[p] = make_float4(1.0f, 2.0f, 3.0f, (float)p);
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int p = 0; p < PER; ++p) {
            const size_t i = base + (size_t)p * 256u;
            if (i < pieces) {
                const float4* a = in + (size_t)r * pieces + i;
                float4 t;
                if (NT) {
                    const v4f q = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(a));
                    t = make_float4(q.x, q.y, q.z, q.w);
                } else t = *a;
                v[p].x += t.x; v[p].y += t.y; v[p].z += t.z; v[p].w += t.w;
            }
        }
    if (W == 0) {                                                   // keep the loads alive: never true (inputs are zero)
#pragma unroll
        for (int p = 0; p < PER; ++p)
            if (v[p].x == 12345.5f) out[base + (size_t)p * 256u] = v[p];
    }
#pragma unroll
    for (int w = 0; w < W; ++w)
#pragma unroll
        for (int p = 0; p < PER; ++p) {
            const size_t i = base + (size_t)p * 256u;
            if (i < pieces) {
                float4* a = out + (size_t)w * pieces + i;
                const float f = (float)(w + 1);
                if (NT) {
                    const v4f q = {v[p].x * f, v[p].y + f, v[p].z - f, v[p].w * f};
                    __builtin_nontemporal_store(q, reinterpret_cast<v4f*>(a));
                } else {
                    const v4f q = {v[p].x * f, v[p].y + f, v[p].z - f, v[p].w * f};
                    __builtin_amdgcn_sched_barrier(0);                     // (keeps the four dwords one store)
                    *reinterpret_cast<v4f*>(a) = q;
                }
            }
        }
}

template <int R, int W, bool NT, int PER>
static void run(const char* name, const float4* in, float4* out, size_t pieces, hipEvent_t e0, hipEvent_t e1) {
    const unsigned grid = (unsigned)((pieces + 256u * PER - 1) / (256u * PER));
    for (int i = 0; i < 3; ++i) k_mix<R, W, NT, PER><<<grid, 256>>>(in, out, pieces);
    const int reps = 20;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) k_mix<R, W, NT, PER><<<grid, 256>>>(in, out, pieces);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, bytes = (double)pieces * 16.0 * (R + W);
    printf("  %-34s %7.1f us  %5.2f TB/s in total, %5.2f TB/s of stores\n", name, us, bytes / us * 1e-6,
           (double)pieces * 16.0 * W / us * 1e-6);
}

int main() {
    // 1 M "Gaussians" of 16 B per stream and piece: 8 streams = 128 B per Gaussian ... sized like the projection backward:
    // pieces = 2 M -> one stream is 32 MB; R + W = 12 streams = 384 MB per launch (the kernel moves 376 MB)
    const size_t pieces = 2u << 20;
    float4 *in, *out;
    CHECK(hipMalloc(&in, pieces * 16 * 12)); CHECK(hipMalloc(&out, pieces * 16 * 12));
    CHECK(hipMemset(in, 0, pieces * 16 * 12)); CHECK(hipMemset(out, 0, pieces * 16 * 12));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int pass = 0; pass < 2; ++pass) {                          // the first pass warms the clocks
        printf("pass %d: 32 MB per stream, 256-thread workgroups\n", pass);
        run<12, 0, false, 1>("12 loads", in, out, pieces, e0, e1);
        run<12, 0, true, 1>("12 loads, non-temporal", in, out, pieces, e0, e1);
        run<0, 12, false, 1>("12 stores", in, out, pieces, e0, e1);
        run<0, 12, true, 1>("12 stores, non-temporal", in, out, pieces, e0, e1);
        run<0, 12, true, 4>("12 stores, nt, 4 chunks per wg", in, out, pieces, e0, e1);
        run<6, 6, false, 1>("6 loads + 6 stores", in, out, pieces, e0, e1);
        run<6, 6, true, 1>("6 loads + 6 stores, non-temporal", in, out, pieces, e0, e1);
        run<4, 8, false, 1>("4 loads + 8 stores", in, out, pieces, e0, e1);
        run<4, 8, true, 1>("4 loads + 8 stores, non-temporal", in, out, pieces, e0, e1);
        run<4, 8, false, 2>("4 + 8, 2 chunks per workgroup", in, out, pieces, e0, e1);
        run<4, 8, true, 2>("4 + 8, nt, 2 chunks per workgroup", in, out, pieces, e0, e1);
        run<4, 8, false, 4>("4 + 8, 4 chunks per workgroup", in, out, pieces, e0, e1);
        run<4, 8, true, 4>("4 + 8, nt, 4 chunks per workgroup", in, out, pieces, e0, e1);
        run<4, 8, true, 8>("4 + 8, nt, 8 chunks per workgroup", in, out, pieces, e0, e1);
        run<8, 4, false, 1>("8 loads + 4 stores", in, out, pieces, e0, e1);          // the forward projection's mix
        run<8, 4, true, 1>("8 loads + 4 stores, non-temporal", in, out, pieces, e0, e1);
        run<8, 4, true, 2>("8 + 4, nt, 2 chunks per workgroup", in, out, pieces, e0, e1);
        run<8, 4, true, 4>("8 + 4, nt, 4 chunks per workgroup", in, out, pieces, e0, e1);
    }
    return 0;
}
