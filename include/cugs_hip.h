/* cugs_hip.h — C ABI of libcugs_hip.so: the MI355X-native differentiable Gaussian-splat
 * rasterizer + fused Adam, as a drop-in for the hot path of
 * Artemarius/cuda-gaussian-splatting (namespace cugs).
 *
 * Conventions:
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless its name
 *     ends in _host; arrays use the reference's layouts and dtypes (fp32 / int32, row-major);
 *   - the callee allocates nothing: outputs and scratch are caller-owned; outputs the
 *     reference zero-fills (torch::zeros) are fully written by the kernels;
 *   - every function is stream-ordered on `stream` (a hipStream_t passed as void*; NULL =
 *     the null stream), re-entrant, and keeps no global state;
 *   - return value: 0 on success, a positive hipError_t from the HIP runtime, or a negative
 *     CUGS_E* argument error.  Nothing throws or aborts across this boundary; the adapter
 *     turns non-zero into std::runtime_error the way CUDA_CHECK does (utils/cuda_utils.cuh:12-20);
 *   - only launch errors are reported (no device sync), as in the reference
 *     (projection.cu:267); cugs_sort_count_pairs is the one blocking call (sorting.cu:146).
 *
 * Citations are file:line in the reference's src/ tree.
 */
#ifndef CUGS_HIP_H
#define CUGS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CUGS_TILE 16                 /* rasterizer/sorting.hpp:16 kTileSize */
#define CUGS_PACKED_STRIDE 12        /* floats per packed projected-Gaussian record */
#define CUGS_GRAD_STRIDE 16          /* floats per packed 2-D gradient accumulator row */

#define CUGS_EINVAL (-1)             /* bad size / degree / null pointer */
#define CUGS_EALIGN (-2)             /* a buffer that must be 16-byte aligned is not */
#define CUGS_EOVERFLOW (-3)          /* pair count does not fit int32 (the reference's index type) */
#define CUGS_EWORKSPACE (-4)         /* workspace too small for (n, total_pairs) */

/* POD camera: what the adapter derives from cugs::CameraInfo (core/types.hpp:78-109).
 * view = row-major 4x4 world-to-camera, the float[16] built at projection.cu:228-233;
 * cam_center = -R^T t (types.hpp:98-100), the 3 floats uploaded at projection.cu:273-275. */
typedef struct cugs_camera {
    float view[16];
    float fx, fy, cx, cy;
    int32_t width, height;
    float cam_center[3];
    float reserved;
} cugs_camera;

const char* cugs_version(void);
/* Message for a return code of this library (hipGetErrorString for positive codes). */
const char* cugs_error_string(int code);

/* ---- a3+a4: project_gaussians (projection.cu:195-289) -------------------------------
 * One launch: k_project_gaussians (projection.cu:55-189) + view directions
 * (projection.cu:273-280) + k_evaluate_sh (core/sh.cu:19-79) + clamp_min(0) (projection.cu:284).
 * positions [n,3], rotations [n,4] (wxyz), scales [n,3] (log), opacities [n] (logit),
 * sh_coeffs [n,3,num_coeffs]; active_degree in 0..3 with (active_degree+1)^2 <= num_coeffs.
 * Outputs: means_2d [n,2], depths [n], cov_2d_inv [n,3], radii [n] i32, tiles_touched [n] i32,
 * opacities_act [n], rgb [n,3] (clamped).  `packed` ([n,CUGS_PACKED_STRIDE] floats, 16-byte
 * aligned) is optional scratch consumed by cugs_rasterize_*; pass NULL to skip it.
 * `colour_gate` ([n] bytes, optional): bit ch = the SH backward's ReLU gate of channel ch, i.e. the sign test of
 * the colour AS THE BACKWARD RECOMPUTES IT (sh_backward.cu:92-99), made here while the coefficients are on
 * chip; cugs_project_backward takes it instead of re-reading them.  rgb > 0 is not that test: forward (sh.cu:44-77)
 * and backward round differently, and within an ulp of zero they disagree. */
int cugs_project_forward(int64_t n, int num_coeffs, int active_degree,
                         const float* positions, const float* rotations, const float* scales,
                         const float* opacities, const float* sh_coeffs,
                         const cugs_camera* camera_host, float scale_modifier,
                         float* means_2d, float* depths, float* cov_2d_inv, int32_t* radii,
                         int32_t* tiles_touched, float* opacities_act, float* rgb,
                         float* packed, uint8_t* colour_gate, void* stream);

/* cugs_project_forward that ALSO leaves the sort's per-Gaussian inputs - the depth key and the tile rectangle
 * sort_gaussians derives from depths / means_2d / radii / tiles_touched first thing (sorting.cu:37-60, :115-131) -
 * in `sort_workspace` (the N-level buffer, cugs_sort_workspace_bytes(n)), while they are in registers.  Not in the
 * reference, whose render() (rasterizer.cpp:58-75) hands the arrays from one stage to the next; this is what a
 * render() built on this library calls, followed by cugs_sort_pairs_predicted_keyed on the same stream with the same
 * workspace and camera size.  All outputs of cugs_project_forward are written as well, bit for bit. */
int cugs_project_forward_keyed(int64_t n, int num_coeffs, int active_degree,
                               const float* positions, const float* rotations, const float* scales,
                               const float* opacities, const float* sh_coeffs,
                               const cugs_camera* camera_host, float scale_modifier,
                               float* means_2d, float* depths, float* cov_2d_inv, int32_t* radii,
                               int32_t* tiles_touched, float* opacities_act, float* rgb,
                               float* packed, uint8_t* colour_gate, void* sort_workspace,
                               size_t sort_workspace_bytes, void* stream);

/* ---- a4: evaluate_sh_cuda (core/sh.cu:81-123), output NOT clamped ------------------- */
int cugs_evaluate_sh(int degree, int64_t n, int num_coeffs, const float* sh_coeffs,
                     const float* directions, float* out_rgb, void* stream);

/* ---- a9: evaluate_sh_backward_cuda (core/sh_backward.cu:114-156) -------------------- */
int cugs_evaluate_sh_backward(int degree, int64_t n, int num_coeffs, const float* sh_coeffs,
                              const float* directions, const float* dL_dcolor,
                              float* dL_dsh, void* stream);

/* The projection in TWO launches, for a caller that overlaps the colour half with the sort (render() does: the sort
 * needs depths / means / radii / tile counts only, while 81 % of the projection's reads - the SH rows - feed nothing
 * before the forward blend).  Same device functions as cugs_project_forward: every output bit-identical.
 *   cugs_project_forward_geometry  k_project_gaussians' half (projection.cu:55-189): 44 B/Gaussian in; means_2d, depths,
 *                                  cov_2d_inv, radii, tiles_touched, opacities_act, words 0..7 of each packed record
 *                                  and - when sort_workspace is given - the sort's keys, exactly as
 *                                  cugs_project_forward_keyed leaves them.  sort_workspace may be NULL.
 *   cugs_project_forward_colour    directions + k_evaluate_sh + clamp (projection.cu:273-284): rgb, colour_gate (may be
 *                                  NULL) and words 8..11 of each packed record (packed may be NULL).
 * The two may run concurrently on different streams (they write disjoint bytes); the blend kernels need both. */
int cugs_project_forward_geometry(int64_t n, const float* positions, const float* rotations, const float* scales,
                                  const float* opacities, const cugs_camera* camera_host, float scale_modifier,
                                  float* means_2d, float* depths, float* cov_2d_inv, int32_t* radii,
                                  int32_t* tiles_touched, float* opacities_act, float* packed, void* sort_workspace,
                                  size_t sort_workspace_bytes, void* stream);
int cugs_project_forward_colour(int64_t n, int num_coeffs, int active_degree, const float* positions,
                                const float* sh_coeffs, const cugs_camera* camera_host, float* rgb, float* packed,
                                uint8_t* colour_gate, void* stream);

/* Fills `packed` from the reference-layout projection outputs, for callers that did not get
 * it from cugs_project_forward (e.g. tests that drive rasterize_forward directly). */
int cugs_pack_projected(int64_t n, const float* means_2d, const float* cov_2d_inv,
                        const float* rgb, const float* opacities_act, float* packed,
                        void* stream);

/* ---- a5: sort_gaussians (sorting.cu:115-227) ----------------------------------------
 * Replaces the cumsum + .item() (sorting.cu:145-146), k_fill_sort_pairs (:30-72),
 * cub::DeviceRadixSort::SortPairs (:191-210; contract: ascending, stable, full 64-bit key)
 * and k_compute_tile_ranges (:82-109).  The two workspace queries replace CUB's two-call
 * temp-storage idiom (:191-198): `workspace` (N-level, cugs_sort_workspace_bytes(n)) carries
 * state from cugs_sort_count_pairs to cugs_sort_pairs and must be the same buffer in both calls,
 * with the same per-Gaussian inputs; `pair_workspace` (cugs_sort_pair_workspace_bytes(P)) can
 * only be sized after the count.  Size of the N-level buffer: ~59 bytes per Gaussian + 0.2 MB, and - up to 2 M
 * Gaussians - 40 KB per 4096 Gaussians more for the table of the keyed route's pair binning (69 MB in all at 1 M). */
size_t cugs_sort_workspace_bytes(int64_t n);
size_t cugs_sort_pair_workspace_bytes(int64_t total_pairs);

/* total_pairs = sum(tiles_touched).  BLOCKS until the value is on the host (the reference's one
 * forced sync) - after having queued all the work that does not depend on it (depth ordering of
 * the Gaussians), so the device is busy while the host waits. */
int cugs_sort_count_pairs(int64_t n, const float* means_2d, const float* depths,
                          const int32_t* radii, const int32_t* tiles_touched, int width, int height,
                          void* workspace, size_t workspace_bytes, int64_t* total_pairs_host,
                          void* stream);

/* keys_sorted [P] u64 (tile_id<<32 | depth bits; may be NULL), values_sorted [P] i32,
 * tile_ranges [tiles,2] i32 ({0,0} for untouched tiles, sorting.cu:216). */
int cugs_sort_pairs(int64_t n, int64_t total_pairs, const float* means_2d, const float* depths,
                    const int32_t* radii, const int32_t* tiles_touched, int width, int height,
                    void* workspace, size_t workspace_bytes, void* pair_workspace,
                    size_t pair_workspace_bytes, uint64_t* keys_sorted, int32_t* values_sorted,
                    int32_t* tile_ranges, void* stream);

/* The whole sort (cugs_sort_count_pairs + cugs_sort_pairs) WITHOUT the host round trip: the caller predicts
 * the pair count (`capacity`: e.g. the previous frame's count plus a margin) and sizes pair_workspace
 * (cugs_sort_pair_workspace_bytes(capacity)), keys_sorted and values_sorted for it; the pair-level kernels take
 * the live count from device memory.  The total is copied to *total_pairs_host asynchronously (pinned host
 * memory recommended); once the stream has completed, the outputs are valid iff
 * 0 <= *total_pairs_host <= capacity.  Otherwise: a count above the capacity -> call cugs_sort_pairs with the now
 * known count (the N-level workspace still holds the depth order); -1 -> some splat's depth lies outside
 * [0.2, ~13 000), the range the three-pass depth ordering of this entry point covers: call cugs_sort_count_pairs
 * (which then takes the general route) and cugs_sort_pairs.  Never blocks. */
int cugs_sort_pairs_predicted(int64_t n, int64_t capacity, const float* means_2d, const float* depths,
                              const int32_t* radii, const int32_t* tiles_touched, int width, int height,
                              void* workspace, size_t workspace_bytes, void* pair_workspace,
                              size_t pair_workspace_bytes, uint64_t* keys_sorted, int32_t* values_sorted,
                              int32_t* tile_ranges, int64_t* total_pairs_host, void* stream);

/* cugs_sort_pairs_predicted on a `workspace` that cugs_project_forward_keyed filled for these arrays (same n, same
 * width x height) on this stream since the last sort that used it: the per-Gaussian key / rectangle launch is
 * skipped (one kernel and 40 MB per million Gaussians).  Outputs and validity rule are those of
 * cugs_sort_pairs_predicted.  BOTH fallbacks (a count above the capacity, or -1) are cugs_sort_count_pairs followed by
 * cugs_sort_pairs, which rebuild everything from the arrays: on images of up to 10 240 tiles and up to 2 M Gaussians this
 * entry point bins the pairs straight into their tiles' lists (no pair-level radix passes, csrc/sort.hip k_bin_*) and
 * does not leave in `workspace` what cugs_sort_pairs alone continues from.  On either miss every tile range is {0,0}:
 * a blend queued behind the sort before the host has looked at the count then does nothing (the index buffer is
 * unwritten).  pair_workspace is not touched on that route (it may still be sized as for cugs_sort_pairs_predicted). */
int cugs_sort_pairs_predicted_keyed(int64_t n, int64_t capacity, const float* means_2d, const float* depths,
                                    const int32_t* radii, const int32_t* tiles_touched, int width, int height,
                                    void* workspace, size_t workspace_bytes, void* pair_workspace,
                                    size_t pair_workspace_bytes, uint64_t* keys_sorted, int32_t* values_sorted,
                                    int32_t* tile_ranges, int64_t* total_pairs_host, void* stream);

/* cugs_sort_pairs_predicted_keyed that also leaves in tile_order[tiles][4] (16-byte aligned) the tiles ordered by the
 * length of their lists, longest first (to 6 %; empty tiles last), as records {tile, first pair, one past the last pair,
 * 0} - what cugs_rasterize_forward_ordered / cugs_rasterize_backward_ordered hand their workgroups out by (tile and
 * range in one load).  Not in the reference; what a render() built on this library calls: on views whose splats
 * cluster (every real capture) the blend kernels run a quarter shorter, on uniform ones the same.
 * Whenever the call leaves valid tile ranges it leaves a valid order (every tile once, with its range), misses included. */
int cugs_sort_pairs_predicted_keyed_ordered(int64_t n, int64_t capacity, const float* means_2d, const float* depths,
                                            const int32_t* radii, const int32_t* tiles_touched, int width, int height,
                                            void* workspace, size_t workspace_bytes, void* pair_workspace,
                                            size_t pair_workspace_bytes, uint64_t* keys_sorted, int32_t* values_sorted,
                                            int32_t* tile_ranges, int64_t* total_pairs_host, uint32_t* tile_order,
                                            void* stream);

/* The same order from any valid tile_ranges (e.g. after cugs_sort_pairs): one small launch. */
int cugs_tile_order(int width, int height, const int32_t* tile_ranges, uint32_t* tile_order, void* stream);

/* The same two entry points on the GENERAL depth route (four 8-bit passes over the raw depth bits: any positive
 * depth), for a caller that already knows the view leaves the three-pass range - an earlier sort of it reported -1.
 * cugs_sort_count_pairs tries the three-pass route first and repeats on the general one; a host that renders such a
 * view every frame (a scene in millimetres, a far backdrop) would pay that twice per frame, and
 * cugs_sort_pairs_predicted would report -1 every time.  Outputs and every other rule are unchanged; the predicted
 * variant never reports -1.  The host keeps the "this view is wide" bit (render() does: sticky per stream, re-probed
 * every 256 sorts). */
int cugs_sort_count_pairs_wide(int64_t n, const float* means_2d, const float* depths,
                               const int32_t* radii, const int32_t* tiles_touched, int width, int height,
                               void* workspace, size_t workspace_bytes, int64_t* total_pairs_host,
                               void* stream);
int cugs_sort_pairs_predicted_wide(int64_t n, int64_t capacity, const float* means_2d, const float* depths,
                                   const int32_t* radii, const int32_t* tiles_touched, int width, int height,
                                   void* workspace, size_t workspace_bytes, void* pair_workspace,
                                   size_t pair_workspace_bytes, uint64_t* keys_sorted, int32_t* values_sorted,
                                   int32_t* tile_ranges, int64_t* total_pairs_host, void* stream);

/* ---- a6: rasterize_forward (forward.cu:180-240, kernel :48-174) ---------------------
 * out_color [H,W,3], out_final_T [H,W], out_n_contrib [H,W] i32.  `packed` may be NULL
 * (records are then gathered from the four reference-layout arrays). */
int cugs_rasterize_forward(int width, int height, const float background_host[3],
                           const int32_t* tile_ranges, const int32_t* gaussian_indices,
                           const float* means_2d, const float* cov_2d_inv, const float* rgb,
                           const float* opacities_act, const float* packed,
                           float* out_color, float* out_final_T, int32_t* out_n_contrib,
                           void* stream);

/* The same blend, and in passing a zero-fill of `zero_buf` (zero_bytes: multiple of 16, buffer 16-byte aligned): the
 * blend kernel is bound by instruction issue and leaves HBM idle, so the [n, CUGS_GRAD_STRIDE] accumulator the
 * backward needs is cleared here for free instead of by a fill in front of cugs_rasterize_backward
 * (then call cugs_rasterize_backward_prezeroed). */
int cugs_rasterize_forward_zero(int width, int height, const float background_host[3],
                                const int32_t* tile_ranges, const int32_t* gaussian_indices,
                                const float* means_2d, const float* cov_2d_inv, const float* rgb,
                                const float* opacities_act, const float* packed,
                                float* out_color, float* out_final_T, int32_t* out_n_contrib,
                                void* zero_buf, size_t zero_bytes, void* stream);

/* cugs_rasterize_forward_zero whose workgroups take their tile AND its range from the records tile_order[0 .. tiles)[4]
 * (cugs_tile_order / cugs_sort_pairs_predicted_keyed_ordered; NULL: the spatial order and tile_ranges).  Any order of the
 * tiles is a correct one: the outputs do not depend on it, bit for bit; the records must agree with tile_ranges.  zero_buf / zero_bytes may be NULL / 0. */
int cugs_rasterize_forward_ordered(int width, int height, const float background_host[3],
                                   const int32_t* tile_ranges, const int32_t* gaussian_indices,
                                   const float* means_2d, const float* cov_2d_inv, const float* rgb,
                                   const float* opacities_act, const float* packed, float* out_color,
                                   float* out_final_T, int32_t* out_n_contrib, void* zero_buf, size_t zero_bytes,
                                   const uint32_t* tile_order, void* stream);

/* ---- a7: rasterize_backward (backward.cu:239-306, kernel :31-233) -------------------
 * grad_accum: [n,CUGS_GRAD_STRIDE] floats, 64-byte aligned scratch (zeroed by the callee).  A row is
 *   {dL_drgb[3], dL_dopacity_act, M1x, M1y, M2xx, M2xy, M2yy, 0...}
 * Words 4..8 are NOT gradients: they are the MOMENTS of dL/dpower over the pixel offsets d = pixel centre - mean,
 *   M1 = sum dL/dpower * (dx, dy),   M2 = sum dL/dpower * (dx^2, dx dy, dy^2),
 * from which the reference's two tensors follow by a per-Gaussian linear map with Sigma'^-1 = (a, b, c)
 * (backward.cu:200-213):   dL_dmeans_2d   = (a M1x + b M1y,  b M1x + c M1y)
 *                          dL_dcov_2d_inv = (-M2xx / 2,  -M2xy,  -M2yy / 2)      [Q3: [1] is the combined off-diagonal]
 * A C caller that wants gradients must either pass the four reference-layout outputs below (the callee applies the
 * map: dL_drgb [n,3], dL_dopacity_act [n], dL_dmeans_2d [n,2], dL_dcov_2d_inv [n,3]; all four or none), or hand the
 * rows to cugs_project_backward / cugs_project_backward_adam, which apply it themselves.  Words 0..3 are final. */
int cugs_rasterize_backward(int width, int height, const float background_host[3],
                            const int32_t* tile_ranges, const int32_t* gaussian_indices,
                            const float* means_2d, const float* cov_2d_inv, const float* rgb,
                            const float* opacities_act, const float* packed,
                            const float* dL_dcolor, const float* final_T,
                            const int32_t* n_contrib, int64_t n, float* grad_accum,
                            float* dL_drgb, float* dL_dopacity_act, float* dL_dmeans_2d,
                            float* dL_dcov_2d_inv, void* stream);

/* cugs_rasterize_backward for a grad_accum that is ALREADY all zeros (cleared by cugs_rasterize_forward_zero since
 * its last use): skips the fill.  Everything else as above. */
int cugs_rasterize_backward_prezeroed(int width, int height, const float background_host[3],
                                      const int32_t* tile_ranges, const int32_t* gaussian_indices,
                                      const float* means_2d, const float* cov_2d_inv, const float* rgb,
                                      const float* opacities_act, const float* packed,
                                      const float* dL_dcolor, const float* final_T,
                                      const int32_t* n_contrib, int64_t n, float* grad_accum,
                                      float* dL_drgb, float* dL_dopacity_act, float* dL_dmeans_2d,
                                      float* dL_dcov_2d_inv, void* stream);

/* cugs_rasterize_backward (prezeroed == 0) or cugs_rasterize_backward_prezeroed (!= 0) with the workgroups handed out in
 * the order of the records tile_order[0 .. tiles)[4] (NULL: the spatial order).  The sums are the same up to the order of
 * the atomic adds. */
int cugs_rasterize_backward_ordered(int width, int height, const float background_host[3],
                                    const int32_t* tile_ranges, const int32_t* gaussian_indices,
                                    const float* means_2d, const float* cov_2d_inv, const float* rgb,
                                    const float* opacities_act, const float* packed,
                                    const float* dL_dcolor, const float* final_T,
                                    const int32_t* n_contrib, int64_t n, float* grad_accum,
                                    float* dL_drgb, float* dL_dopacity_act, float* dL_dmeans_2d,
                                    float* dL_dcov_2d_inv, int prezeroed, const uint32_t* tile_order, void* stream);

/* ---- a8+a9: project_backward (projection_backward.cu:253-344, kernel :26-247) -------
 * One launch: k_project_backward + directions + k_evaluate_sh_backward.  The incoming 2-D
 * gradients come either from grad_accum (the MOMENT rows cugs_rasterize_backward leaves - layout there; the
 * kernel turns words 4..8 into dL_dmeans_2d / dL_dcov_2d_inv with the Sigma'^-1 it recomputes; preferred) or,
 * when grad_accum is NULL, from the four reference-layout arrays (which hold gradients, not moments).
 * colour_gate ([n] bytes from cugs_project_forward of the same model, camera and degree) supplies the ReLU gate (sh_backward.cu:92-100); when NULL the gate is
 * recomputed from sh_coeffs as the reference does - the same bits, 12 num_coeffs more bytes read per
 * Gaussian.  dL_dmeans_2d_out ([n,2], may be NULL)
 * receives BackwardOutput::dL_dmeans_2d (rasterizer.cpp:184) when grad_accum is used.
 * dL_dsh_coeffs may be NULL when dL_drgb_gated_out ([n,3]: dL_drgb with the ReLU gate applied) is
 * given instead: the data-parallel exchange (cugs_sh_backward_views) rebuilds the SH gradient from it.
 * Both NULL: geometry gradients only (the caller took the colour half with cugs_gated_colour_grad). */
int cugs_project_backward(int64_t n, int num_coeffs, int active_degree,
                          const float* positions, const float* rotations, const float* scales,
                          const float* opacities, const float* sh_coeffs, const int32_t* radii,
                          const uint8_t* colour_gate, const cugs_camera* camera_host,
                          float scale_modifier, const float* grad_accum,
                          const float* dL_dmeans_2d, const float* dL_dcov_2d_inv,
                          const float* dL_drgb, const float* dL_dopacity_act,
                          float* dL_dpositions, float* dL_drotations, float* dL_dscales,
                          float* dL_dopacities, float* dL_dsh_coeffs, float* dL_dmeans_2d_out,
                          float* dL_drgb_gated_out, void* stream);

/* ---- a8+a9+a11 in one launch (single-GPU training; trainer.cpp:228-242 calls render_backward, then
 * FusedAdam::apply_gradients + step): the projection/SH backward applies k_fused_adam's update
 * (fused_adam.cu:44-76) to each Gaussian's own parameters as soon as its gradients exist, so the five gradient
 * tensors (236 B/Gaussian at degree 3) are neither written nor read back by an optimizer launch.  Same
 * arithmetic in the same order as cugs_project_backward followed by cugs_fused_adam_groups: identical bits.
 * The parameters are updated IN PLACE; the 2-D gradients come from grad_accum, the ReLU gate from colour_gate
 * (both required).  adam_host: moments and learning rates in ParamGroup order {positions, sh_coeffs,
 * opacities, scales, rotations} (lr_schedule.hpp:23-29), bc1/bc2 from cugs_adam_bias_correction.  Not for
 * data-parallel training: there the gradients must be exchanged between backward and optimizer. */
typedef struct cugs_adam_fused {
    float* m[5];
    float* v[5];
    float lr[5];
    float beta1, beta2, eps, bc1, bc2;
} cugs_adam_fused;
int cugs_project_backward_adam(int64_t n, int num_coeffs, int active_degree, float* positions,
                               float* rotations, float* scales, float* opacities, float* sh_coeffs,
                               const int32_t* radii, const uint8_t* colour_gate,
                               const cugs_camera* camera_host, float scale_modifier,
                               const float* grad_accum, const cugs_adam_fused* adam_host,
                               float* dL_dmeans_2d_out, void* stream);

/* ---- data-parallel extension (no counterpart in the single-GPU reference) ----------
 * dL_dsh[i] = sum over views v of gated_rgb_views[v][i] (x) Y(normalize(positions[i] - centre_v)),
 * accumulated in view order.  gated_rgb_views: [num_views][n][3] (the all-gather of every rank's
 * dL_drgb_gated_out); cam_centers_host: [num_views][3]; num_views <= 16.  Equals the sum of the
 * per-view evaluate_sh_backward_cuda results (sh_backward.cu:29-112) without moving them. */
int cugs_sh_backward_views(int degree, int64_t n, int num_coeffs, const float* positions,
                           int num_views, const float* gated_rgb_views,
                           const float* cam_centers_host, float* dL_dsh, void* stream);

/* dL_drgb_gated_out[i][ch] = grad_accum[i][ch] * (bit ch of colour_gate[i]) - the [n,3] tensor
 * cugs_project_backward writes to its dL_drgb_gated_out, made from the backward blend's accumulator alone
 * (sh_backward.cu:92-100 applied to the blend's dL_drgb): the data-parallel exchange starts the all-gather of
 * these 12 B/Gaussian before the projection backward runs and lets it travel underneath. */
int cugs_gated_colour_grad(int64_t n, const float* grad_accum, const uint8_t* colour_gate,
                           float* dL_drgb_gated_out, void* stream);

/* ---- a11: FusedAdam (optimizer/fused_adam.cu:44-76,140-219) -------------------------
 * bc1 = 1/(1-beta1^t), bc2 = 1/(1-beta2^t) computed in double on the host, then float
 * (fused_adam.cu:145-148,161-162). */
void cugs_adam_bias_correction(float beta1, float beta2, int step, float* bc1_host,
                               float* bc2_host);

/* One parameter tensor (k_fused_adam, fused_adam.cu:44-76): in-place on param, m, v. */
int cugs_fused_adam(float* param, const float* grad, float* m, float* v, int64_t n, float lr,
                    float beta1, float beta2, float eps, float bc1, float bc2, void* stream);

typedef struct cugs_adam_group {
    float* param;
    const float* grad;     /* NULL: group skipped (fused_adam.cu:156 "if (!grads_[i].defined())") */
    float* m;
    float* v;
    int64_t n;
    float lr;
    float reserved;
} cugs_adam_group;

/* All parameter groups in ONE launch (the reference launches one kernel per group,
 * fused_adam.cu:155-163).  ngroups <= 8. */
int cugs_fused_adam_groups(const cugs_adam_group* groups_host, int ngroups, float beta1,
                           float beta2, float eps, float bc1, float bc2, void* stream);

/* ---- N1: combined_loss + dL/dcolor (training/loss.cpp:88-140, trainer.cpp:214-217) ----
 * L = (1-lambda) mean|x-y| + lambda (1 - mean SSIM), SSIM with a window_size x window_size sigma-1.5
 * Gaussian window (odd, 3..15; the reference default is 11), zero padding, C1 = 1e-4, C2 = 9e-4.
 * rendered, target: [H,W,3].  loss_out: DEVICE float[4] = {loss, L1, mean SSIM, 1 - mean SSIM} (no host
 * sync).  ssim_map ([H,W], mean over channels = ssim(), loss.cpp:93) and dL_dcolor ([H,W,3] = d loss /
 * d rendered, what autograd returns at trainer.cpp:217) are optional.  workspace: cugs_loss_workspace_bytes. */
size_t cugs_loss_workspace_bytes(int width, int height);
int cugs_combined_loss(int width, int height, const float* rendered, const float* target, float lambda,
                       int window_size, void* workspace, size_t workspace_bytes, float* loss_out,
                       float* ssim_map, float* dL_dcolor, void* stream);

/* ---- N2: adaptive density control (optimizer/densification.cpp) ------------------------------
 * cugs_densify_accumulate: DensificationController::accumulate_gradients (densification.cpp:59-88), one
 *   launch, no host sync: for radii > 0, grad_accum += ||dL_dmeans_2d||_2 and grad_count += 1; for every
 *   Gaussian max_radii_2d = max(max_radii_2d, radii).  All three accumulators are float [n].
 * cugs_densify_classify: compute_clone_mask / compute_split_mask / compute_keep_mask (:351-442) as one
 *   byte per Gaussian: bit 0 clone candidate, bit 1 split candidate, bit 2 keep.  The thresholds are the
 *   reference's float products (size = percent_dense * scene_extent, ws = 0.1f * scene_extent);
 *   apply_size_pruning = (opacity_reset_every > 0 && step > opacity_reset_every); max_screen_size <= 0
 *   disables the screen-size test.  avg_grad (nullable) receives grad_accum / max(grad_count, 1), the key
 *   of the reference's max_gaussians top-k (:127-134).
 * cugs_densify_plan: counts and orders the result of densify() (:116-325) for FINAL flags (bit 0 clone,
 *   bit 1 split, bit 2 keep; the caller has applied any budget).  counts_host = {kept originals, clones,
 *   splits, n_out = kept + clones + 2 * splits}; an original is kept iff bit 2 is set and bit 1 is not.
 *   Blocks on a 32-byte read-back (the reference's sum().item() calls, :118,179).
 * cugs_densify_apply: writes each array of the new model, n_out rows in the reference's final order
 *   [kept originals | clones | first children | second children], each group in ascending parent index.
 *   Modes: COPY (new rows copy the parent: sh, opacities, rotations), POSITIONS (children:
 *   parent + noise[which][parent] * exp(scale - log 1.6), :253-262; noise is [2, n, 3] standard normal
 *   supplied by the caller in place of the reference's torch::randn_like), SCALES (children: scale -
 *   log 1.6), STATE (optimizer moments: survivors keep theirs, new rows are zero - the reference rebuilds
 *   its optimizer instead, trainer.cpp:267-304).  `scales` is the OLD scales array [n, 3].
 */
enum { CUGS_DENSIFY_COPY = 0, CUGS_DENSIFY_POSITIONS = 1, CUGS_DENSIFY_SCALES = 2, CUGS_DENSIFY_STATE = 3 };
typedef struct cugs_densify_array {
    const float* src;     /* [n, row_floats] device */
    float* dst;           /* [n_out, row_floats] device */
    int32_t row_floats;
    int32_t mode;         /* CUGS_DENSIFY_* */
} cugs_densify_array;
int cugs_densify_accumulate(int64_t n, const float* dL_dmeans_2d, const int32_t* radii, float* grad_accum,
                            float* grad_count, float* max_radii_2d, void* stream);
int cugs_densify_classify(int64_t n, const float* grad_accum, const float* grad_count, const float* max_radii_2d,
                          const float* scales, const float* opacities, float grad_threshold, float size_threshold,
                          float opacity_threshold, int apply_size_pruning, float max_screen_size,
                          float ws_threshold, uint8_t* flags, float* avg_grad, void* stream);
size_t cugs_densify_workspace_bytes(int64_t n);
int cugs_densify_plan(int64_t n, const uint8_t* flags, void* workspace, size_t workspace_bytes,
                      int64_t counts_host[4], void* stream);
int cugs_densify_apply(int64_t n, int64_t n_out, const void* workspace, size_t workspace_bytes, const float* noise,
                       const float* scales, const cugs_densify_array* arrays_host, int num_arrays, void* stream);

/* ---- N3: Gaussian-model PLY records (utils/ply_io.cpp:98-196, 258-351) -------------------------
 * The vertex record of the reference's checkpoint format, assembled / taken apart on the device:
 *   x y z | nx ny nz (zero) | f_dc_0..2 | f_rest_0..3(C-1)-1 (f_rest_{(k-1)*3+ch} = sh[ch][k]) | opacity |
 *   scale_0..2 | rot_0..3                                         = 14 + 3C floats (62 at SH degree 3),
 * optionally followed by the Adam first and second moments in the same order without the normals (2 x (11 + 3C)
 * floats): what a resumable checkpoint needs and the reference does not store.
 * params / m / v: five device pointers each in ParamGroup order {positions, sh_coeffs, opacities, scales,
 * rotations}; m and v both NULL = no optimizer state.
 * cugs_ply_pack:   vertices [n, cugs_ply_vertex_floats(C, state)] <- the tensors.
 * cugs_ply_unpack: the tensors <- vertices [n, num_props] as read from a file whose properties may be in any
 *   order or include others: col_of[c] (device, 11 + 3C entries, x3 with state) is the file column of canonical
 *   model float c (the record order above without the normals; then m, then v), looked up by NAME on the host.
 */
int cugs_ply_vertex_floats(int num_coeffs, int with_state);
int cugs_ply_pack(int64_t n, int num_coeffs, const float* const params[5], const float* const m[5],
                  const float* const v[5], float* vertices, void* stream);
int cugs_ply_unpack(int64_t n, int num_coeffs, int num_props, const float* vertices, const int32_t* col_of,
                    float* const params[5], float* const m[5], float* const v[5], void* stream);

/* ---- N4: training target from a cached 8-bit view (data/image_io.cpp:35-39, 47-100) -------------
 * dst [dst_height, dst_width, 3] float <- src [src_height, src_width, 3] uint8 (device, decoded once by the
 * caller): value * (1/255), and - when the sizes differ - the reference's resize_image (pixel-centre bilinear,
 * clamped edges, same fp32 operation order), i.e. the tensor trainer.cpp:186-198 builds on the CPU and uploads
 * every iteration, bit for bit. */
int cugs_image_to_float(int src_width, int src_height, const uint8_t* src_rgb8, int dst_width, int dst_height,
                        float* dst, void* stream);

/* Device properties the host side needs without linking the HIP runtime itself. */
int cugs_device_count(int* count_host);

#ifdef __cplusplus
}
#endif
#endif /* CUGS_HIP_H */
