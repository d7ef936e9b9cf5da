// views.hip — training targets from a device-resident view cache.
//
// The reference decodes the sampled training image on the CPU EVERY iteration (stb_image), converts it to
// float (data/image_io.cpp:35-39), resizes it on the CPU when the camera and file resolutions differ
// (resize_image, :47-100, called from training/trainer.cpp:186-196) and copies 12 B/pixel to the device.
// With 288 GB of HBM the decoded 8-bit images of a whole dataset stay on the device (3 B/pixel: 6 MB per
// 1080p view); one launch turns a cached image into the float [H, W, 3] target - the x 1/255 conversion and,
// if needed, the reference's bilinear resize with its exact operation order (fp32, no contraction), so the
// target is bit-identical to what the reference's CPU path would upload.
#include "cugs_common.h"

namespace {

__device__ __forceinline__ float texel(const uint8_t* __restrict__ src, int w, int x, int y, int c) {
    return (float)src[((size_t)y * w + x) * 3 + c] * (1.0f / 255.0f);          // image_io.cpp:35-39
}

__global__ __launch_bounds__(CUGS_BLOCK) void k_image_to_float(int sw, int sh, const uint8_t* __restrict__ src, int dw,
                                                               int dh, float* __restrict__ dst) {
    const int64_t e = (int64_t)blockIdx.x * CUGS_BLOCK + threadIdx.x;
    if (e >= (int64_t)dw * dh * 3) return;
    const int c = (int)(e % 3);
    const int64_t pix = e / 3;
    const int x = (int)(pix % dw), y = (int)(pix / dw);
    if (sw == dw && sh == dh) {                                                 // trainer.cpp:192: no resize
        dst[e] = texel(src, sw, x, y, c);
        return;
    }
    // resize_image (image_io.cpp:60-96): pixel-centre bilinear, edges clamped
    const float x_scale = (float)sw / (float)dw, y_scale = (float)sh / (float)dh;
    const float src_y = ((float)y + 0.5f) * y_scale - 0.5f;
    const int y0 = max(0, (int)floorf(src_y));
    const int y1 = min(sh - 1, y0 + 1);
    const float fy = src_y - (float)y0;
    const float src_x = ((float)x + 0.5f) * x_scale - 0.5f;
    const int x0 = max(0, (int)floorf(src_x));
    const int x1 = min(sw - 1, x0 + 1);
    const float fx = src_x - (float)x0;
    const float v00 = texel(src, sw, x0, y0, c), v10 = texel(src, sw, x1, y0, c);
    const float v01 = texel(src, sw, x0, y1, c), v11 = texel(src, sw, x1, y1, c);
    const float top = v00 + (v10 - v00) * fx;
    const float bot = v01 + (v11 - v01) * fx;
    dst[e] = top + (bot - top) * fy;
}

}  // namespace

extern "C" int cugs_image_to_float(int src_width, int src_height, const uint8_t* src_rgb8, int dst_width, int dst_height,
                                   float* dst, void* stream) {
    if (src_width <= 0 || src_height <= 0) return CUGS_EINVAL;
    if (dst_width <= 0 || dst_height <= 0) return CUGS_EINVAL;                  // image_io.cpp:48-50
    if (!src_rgb8 || !dst) return CUGS_EINVAL;
    const int64_t total = (int64_t)dst_width * dst_height * 3;
    if (total > 2147483647ll * CUGS_BLOCK) return CUGS_EOVERFLOW;
    hipLaunchKernelGGL(k_image_to_float, dim3((unsigned)((total + CUGS_BLOCK - 1) / CUGS_BLOCK)), dim3(CUGS_BLOCK), 0,
                       static_cast<hipStream_t>(stream), src_width, src_height, src_rgb8, dst_width, dst_height, dst);
    CUGS_LAUNCH_CHECK();
    return 0;
}
