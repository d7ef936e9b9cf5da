// ref_sh_shim.cpp — extern "C" entry around the REFERENCE's evaluate_sh_cpu
// (src/core/sh.cpp:8-87, compiled in place from /root/reference by oracle/Makefile
// into oracle/_ref/libref_sh.so).  Test infrastructure: used only to check the
// oracle's SH restatement against reference-authored code.
#include "core/sh.hpp"

#include <cstdint>
#include <cstring>

extern "C" int ref_evaluate_sh_cpu(int degree, int64_t n, int num_coeffs, const float* sh,
                                   const float* dirs, float* out) {
    try {
        auto opts = torch::TensorOptions().dtype(torch::kFloat32);
        auto sh_t = torch::from_blob(const_cast<float*>(sh), {n, 3, num_coeffs}, opts);
        auto d_t = torch::from_blob(const_cast<float*>(dirs), {n, 3}, opts);
        auto r = cugs::evaluate_sh_cpu(degree, sh_t, d_t).contiguous();
        std::memcpy(out, r.data_ptr<float>(), sizeof(float) * 3 * static_cast<size_t>(n));
        return 0;
    } catch (const c10::Error&) {
        return 1;   // the reference's TORCH_CHECK input validation (test_sh.cpp:127-143)
    } catch (...) {
        return 2;
    }
}
