// adapter_bench.cpp — the C++ host (cugs_hip_torch: what a maintainer links under the reference's render() /
// render_backward() / FusedAdam) timed on raw binary inputs written by tools/bench_cpp_host.py:
//   adapter_bench <dir> <n> <C> <width> <height> <steps> <warmup> [adam]
// One step = render + render_backward (+ the fused optimizer step with `adam`), as bench.py times the Python mirror.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "cugs_hip_torch.hpp"

static torch::Tensor load(const std::string& p, std::vector<int64_t> shape) {
    auto t = torch::empty(shape, torch::kFloat32);
    FILE* f = fopen(p.c_str(), "rb");
    if (!f || fread(t.data_ptr<float>(), sizeof(float), t.numel(), f) != static_cast<size_t>(t.numel())) {
        fprintf(stderr, "cannot read %s\n", p.c_str());
        exit(2);
    }
    fclose(f);
    return t.to(torch::kCUDA);
}

int main(int argc, char** argv) {
    if (argc < 8) return 1;
    const std::string d = argv[1];
    const int64_t n = atoll(argv[2]), C = atoll(argv[3]);
    const int w = atoi(argv[4]), h = atoi(argv[5]), steps = atoi(argv[6]), warmup = atoi(argv[7]);
    const bool adam = argc > 8 && std::string(argv[8]) == "adam";
    try {
        cugs_hip::ModelTensors m{load(d + "/positions.bin", {n, 3}), load(d + "/sh_coeffs.bin", {n, 3, C}),
                                 load(d + "/opacities.bin", {n, 1}), load(d + "/rotations.bin", {n, 4}),
                                 load(d + "/scales.bin", {n, 3})};
        auto camt = load(d + "/camera.bin", {26}).to(torch::kCPU);
        const float* cf = camt.data_ptr<float>();
        cugs_camera cam{};
        for (int i = 0; i < 16; ++i) cam.view[i] = cf[i];
        cam.fx = cf[16]; cam.fy = cf[17]; cam.cx = cf[18]; cam.cy = cf[19];
        cam.width = w; cam.height = h;
        cam.cam_center[0] = cf[20]; cam.cam_center[1] = cf[21]; cam.cam_center[2] = cf[22];
        cugs_hip::RenderSettings st;
        st.background[0] = cf[23]; st.background[1] = cf[24]; st.background[2] = cf[25];
        st.active_sh_degree = C == 16 ? 3 : C == 9 ? 2 : C == 4 ? 1 : 0;
        auto g = load(d + "/dl_dcolor.bin", {h, w, 3});
        // learning rates at zero: every launch of the optimizer step, the model (hence the workload) staying put
        cugs_hip::FusedAdam opt({m.positions, m.sh_coeffs, m.opacities, m.scales, m.rotations}, {0.f, 0.f, 0.f, 0.f, 0.f});
        int64_t pairs = 0;
        auto step = [&]() {
            auto out = cugs_hip::render(m, cam, st);
            pairs = out.gaussian_indices.numel();
            auto grads = cugs_hip::render_backward(g, out, m, cam, st, adam ? &opt : nullptr);
        };
        for (int i = 0; i < warmup; ++i) step();
        TORCH_CHECK(hipDeviceSynchronize() == hipSuccess, "sync failed");
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < steps; ++i) step();
        TORCH_CHECK(hipDeviceSynchronize() == hipSuccess, "sync failed");
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / steps;
        printf("{\"host\": \"c++ (cugs_hip_torch)\", \"ms_per_step\": %.4f, \"mpixels_per_s\": %.1f, \"pairs\": %lld, \"steps\": %d, "
               "\"warmup\": %d, \"adam_fused\": %s}\n", ms, (double)w * h / ms / 1e3, (long long)pairs, steps, warmup,
               adam ? "true" : "false");
        return 0;
    } catch (const std::exception& e) {
        fprintf(stderr, "adapter_bench failed: %s\n", e.what());
        return 4;
    }
}
