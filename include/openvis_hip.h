/* openvis_hip.h — C ABI of libopenvis_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the per-frame dense inference path of clownrat6/OpenVIS.
 * Every pointer is a DEVICE pointer unless its name ends in _host. Every entry
 * point enqueues on `stream` (a hipStream_t passed as void*; NULL = the null
 * stream), never synchronises, allocates nothing, keeps no global state and is
 * re-entrant. Return value: 0 on success, otherwise an OVIS_E* code; the message
 * is available from ovis_last_error() (thread-local). A failed call has launched
 * nothing. (The reference only printf()s launch errors — ms_deform_im2col_cuda.cuh:953-957.)
 *
 * Reference file:line citations are into /root/reference/.
 */
#ifndef OPENVIS_HIP_H
#define OPENVIS_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OVIS_OK 0
#define OVIS_EINVAL 1   /* bad argument (null pointer, non-positive size, unsupported shape) */
#define OVIS_ELAUNCH 2  /* hipLaunchKernel / runtime error */

typedef void* ovis_stream_t; /* hipStream_t */

/* Library ABI version (bumped when a signature changes). */
int ovis_abi_version(void);
/* Last error message of the calling thread ("" if none). */
const char* ovis_last_error(void);

/* ---- B1: multi-scale deformable attention forward ("K1") -------------------------------
 * Replaces pybind `MultiScaleDeformableAttention.ms_deform_attn_forward`
 *   openvis/modeling/pixel_decoder/ops/src/vision.cpp:19, ms_deform_attn.h:26-44,
 *   cuda/ms_deform_attn_cuda.cu:25-85, kernel cuda/ms_deform_im2col_cuda.cuh:242-304.
 *   value            [batch, spatial_size, num_heads, channels]      contiguous
 *   spatial_shapes   [num_levels, 2] int64 (H_l, W_l)                 (device, as in the reference)
 *   level_start_index[num_levels]    int64
 *   sampling_loc     [batch, num_query, num_heads, num_levels, num_point, 2]  (x, y) in [0,1]
 *   attn_weight      [batch, num_query, num_heads, num_levels, num_point]
 *   out              [batch, num_query, num_heads*channels]           fully overwritten
 * The reference's im2col_step chunking (cuda.cu:55-80) is a launch detail with no
 * numerical effect; one launch covers the whole batch here. */
int ovis_msda_forward_f32(const float* value, const int64_t* spatial_shapes,
                          const int64_t* level_start_index, const float* sampling_loc,
                          const float* attn_weight, float* out, int batch, int spatial_size,
                          int num_heads, int channels, int num_levels, int num_query,
                          int num_point, ovis_stream_t stream);
int ovis_msda_forward_f64(const double* value, const int64_t* spatial_shapes,
                          const int64_t* level_start_index, const double* sampling_loc,
                          const double* attn_weight, double* out, int batch, int spatial_size,
                          int num_heads, int channels, int num_levels, int num_query,
                          int num_point, ovis_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
