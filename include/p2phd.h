/*
 * p2phd.h -- C ABI of libp2phd_hip.so: the MI355X (gfx950) hot path of pix2pixHD audio
 * super-resolution (batched MDCT4/IMDCT4 + generator/discriminator conv stack).
 *
 * The reference (ishine/pix2pixHDAudioSR) has no FFI on this path: its boundary is the Python
 * module API (models/mdct.py, models/networks.py, models/pix2pixHD_model.py).  Every entry
 * point below names the reference code it replaces (file:line relative to the reference
 * checkout); the Python mirror of those modules in pix2pixhdaudiosr_amd/ binds these symbols
 * with ctypes (see INTEGRATION.md for the stub a reference maintainer would add).
 *
 * Conventions
 *   - all data pointers are DEVICE pointers, caller-allocated (torch tensors own them);
 *   - `stream` is a hipStream_t passed as void* (the caller's current stream); no entry point
 *     allocates, frees or synchronises, so calls are graph-capturable;
 *   - return 0 on success, a negative P2PHD_E* code otherwise; p2phd_last_error() gives text
 *     (thread-local);
 *   - thread-compatible, not thread-safe per output buffer.
 */
#ifndef P2PHD_H
#define P2PHD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P2PHD_OK            0
#define P2PHD_EINVAL       -1   /* bad argument / unsupported geometry */
#define P2PHD_ELAUNCH      -2   /* HIP launch error */
#define P2PHD_EUNSUPPORTED -3   /* valid request this build does not implement */

/* element types of activation / weight buffers */
#define P2PHD_F32  0
#define P2PHD_BF16 1

const char* p2phd_last_error(void);
int p2phd_abi_version(void);
/* fills name (<= cap bytes) with the device's gcnArchName; returns CU count or <0 */
int p2phd_device_info(char* name, int cap);

/* ------------------------------------------------------------------------------------------
 * MDCT4 / IMDCT4 (models/mdct.py:461-566).  n_fft a power of two in [16, 4096].
 *
 * Tables: p2phd_mdct4_tables_floats(n_fft) floats, filled on the HOST by
 * p2phd_mdct4_tables_fill (fp64 trigonometry rounded once to fp32), uploaded by the caller and
 * passed back as the device pointer `tables`.  They replace exp1/exp2 of mdct.py:483-484,539-540.
 * ---------------------------------------------------------------------------------------- */
size_t p2phd_mdct4_tables_floats(int n_fft);
int p2phd_mdct4_tables_fill(int n_fft, float* host_out);

/* Frame geometry exactly as MDCT4.forward computes it (mdct.py:488-500), including the
 * len(signal) quirk: dim0 is the size of the first dimension of the input (the batch size for a
 * [B,T] signal, T for a 1-D one).  Pure host integer arithmetic. */
int p2phd_mdct4_frame_layout(int64_t dim0, int64_t T, int hop, int win, int center,
                             int64_t* start_pad, int64_t* end_pad, int64_t* n_frames);

/* Framed transform: out[b,t,k] = scale * sum_n w[n] xpad[b, t*hop+n] cos(2pi/N (n+1/2+N/4)(k+1/2)),
 * xpad = x shifted right by start_pad with zeros outside [0,T).  x [B,T] f32, window [win] f32,
 * out [B,F,N/2] f32.  MDCT4.forward (mdct.py:486-513) = this with scale 1; the backward of
 * IMDCT4 = this with x = grad, start_pad = crop, scale 4/N. */
int p2phd_mdct4_fwd(const float* x, int64_t B, int64_t T, int n_fft, int hop, int win,
                    const float* window, const float* tables, int64_t start_pad, int64_t n_frames,
                    float scale, float* out, void* stream);

/* Inverse framed transform with windowed overlap-add:
 * out[b,m] = scale * sum_t w[q] y_t[q], q = m + crop_start - t*hop in [0,win),
 * y_t[n] = sum_k spec[b,t,k] cos(2pi/N (n+1/2+N/4)(k+1/2)).   spec [B,F,N/2] f32, out [B,out_len] f32.
 * IMDCT4.forward (mdct.py:542-566) = this with scale 4/N, crop_start = win/2 (center) and
 * out_len = min(out_length, (F-1)*hop [+win if !center]); the backward of MDCT4 = this with scale 1,
 * crop_start = start_pad, out_len = T. */
int p2phd_imdct4_fwd(const float* spec, int64_t B, int64_t n_frames, int n_fft, int hop, int win,
                     const float* window, const float* tables, int64_t crop_start, int64_t out_len,
                     float scale, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* P2PHD_H */
