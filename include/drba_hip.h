/*
 * drba_hip.h — C ABI of libdrba_hip.so, the MI355X (gfx950) kernel library behind the
 * DRBA per-frame hot path.
 *
 * The reference (routineLife1/DRBA) has no FFI/plugin layer: its boundary is the Python
 * operator surface.  Each entry point below replaces one reference
 * operator (or a fused group of them) and cites it.  Conventions for every function:
 *   - all pointers are DEVICE pointers to contiguous fp32 NCHW data unless noted;
 *   - caller owns all memory; the library never allocates, frees or synchronises;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*), re-entrant across streams;
 *   - returns 0 on success or a negative DRBA_E* code (see drba_error_string);
 *   - `ws` arguments are caller-provided scratch of at least the documented size.
 * The reference-side binding a maintainer would add is a ctypes stub; see INTEGRATION.md.
 */
#ifndef DRBA_HIP_H
#define DRBA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DRBA_OK 0
#define DRBA_EINVAL (-1)   /* bad argument (null pointer, non-positive size, unknown mode) */
#define DRBA_EUNSUPPORTED (-2) /* shape/config outside what the kernels were built for */
#define DRBA_ELAUNCH (-3)  /* hipGetLastError() reported a launch failure */

int drba_abi_version(void);
const char *drba_error_string(int code);

/* ---- forward splat ----------------------------------------------------------------------
 * replaces: models/softsplat/softsplat.py:248-293 (softsplat) + :306-367 (kernel softsplat_out)
 *           == models/softsplat/softsplat_torch.py:19-179.
 * mode: 0 sum, 1 avg, 2 linear, 3 soft.  eps: 0 addeps(+1e-7, default), 1 zeroeps, 2 clipeps.
 * metric: [N,1,H,W] or NULL (required for linear/soft).  ws: N*H*W*(C+1) floats. */
int drba_softsplat(const float *in, const float *flow, const float *metric, float *out, float *ws,
                   int N, int C, int H, int W, int mode, int eps, void *stream);
size_t drba_softsplat_ws_floats(int N, int C, int H, int W);

/* ---- backward warp -----------------------------------------------------------------------
 * replaces: models/rife_426_heavy/warplayer.py:8-22 (padding 0 = border) and
 *           models/model_gmfss_union/MetricNet.py:10-20 (padding 1 = zeros).
 * grid_sample(bilinear, align_corners=True) on base-grid + flow/((size-1)/2). */
int drba_backwarp(const float *in, const float *flow, float *out, int N, int C, int H, int W,
                  int padding, void *stream);

/* ---- flow magnitude: models/utils/tools.py:77-80 (distance_calculator) */
int drba_flow_distance(const float *flow, float *out, int N, int H, int W, void *stream);

/* ---- fused flow reversal: models/rife.py:59-73 (calc_flow tail)
 * out = 2 * where(splat_avg(1, f) < 0.999, max(H,W), -splat_avg(f, f)).  ws: N*H*W*3 floats. */
int drba_flow_reverse(const float *flow, float *out, float *ws, int N, int H, int W, void *stream);

/* ---- fused linear DRM, one direction: models/drm.py:65-107 with linear=True
 * u = d_other/(d_self+d_other) * t * 2 with d = |flow| + eps; out = splat_avg(u, self*u) with
 * uncovered pixels (ones-splat < 0.999) keeping u.  drm_t1_t01 = (self=flow10, other=flow12),
 * drm_t1_t12 = (self=flow12, other=flow10).  ws: N*H*W*2 floats.  If t_dev != NULL the timestep
 * is read from that device float instead of `t` (so a captured HIP graph can be replayed for any t). */
int drba_drm_rife_linear(const float *flow_self, const float *flow_other, float t, const float *t_dev,
                         float eps, float *out, float *ws, int N, int H, int W, void *stream);

/* ---- DRM building blocks for the non-fused variants (drm.py:110-195, :10-62) */
/* ratio maps: a = d10/(d10+d12), b = d12/(d10+d12), d = |flow| + eps; either output may be NULL */
int drba_drm_ratio(const float *flow10, const float *flow12, float eps, float *drm10, float *drm12,
                   int N, int H, int W, void *stream);
/* out = a * mul + add  (n elements) */
int drba_affine(const float *a, float mul, float add, float *out, size_t n, void *stream);
/* out[n,c,:,:] = flow[n,c,:,:] * map[n,0,:,:]  (C channels) */
int drba_mul_map(const float *x, const float *map, float *out, int N, int C, int H, int W, void *stream);
/* out = cover < 0.999 ? value : aligned   (n elements; NaN cover keeps aligned) */
int drba_fill_holes(const float *aligned, const float *cover, const float *value, float *out,
                    size_t n, void *stream);
/* non-linear retiming get_drm_t (drm.py:10-62): scalar bisection replayed per element */
int drba_drm_retime(const float *drm, float *out, double t, double precision, size_t n, void *stream);

/* ---- resize / frame conversion: models/utils/tools.py:33-38, :59-72 ------------------------
 * bilinear, align_corners=False; src coordinate = scale*(dst+0.5)-0.5 (clamped at 0) with
 * scale_y/scale_x given explicitly (in/out for size-based calls, 1/scale_factor otherwise). */
int drba_resize_bilinear(const float *in, float *out, int NC, int Hin, int Win, int Hout, int Wout,
                         float scale_y, float scale_x, void *stream);
int drba_u8hwc_to_f32nchw(const uint8_t *in, float *out, int H, int W, void *stream); /* /255. */
int drba_f32nchw_to_u8hwc(const float *in, uint8_t *out, int H, int W, void *stream); /* trunc(x*255.) */

/* ---- scene-cut metric: tools.py:27-30 + pytorch_msssim/__init__.py:83-136 (ssim_matlab)
 * x1, x2: [1,3,32,32] thumbnails (already resized); out: 1 float on device. */
int drba_ssim3d_32(const float *x1, const float *x2, float *out, void *stream);

/* ---- convolutions (models/rife_426_heavy/IFNet_HDv3.py:11-16, :28-47, :50-59, :65-82) ------
 * fp32 implicit GEMM on v_mfma_f32_16x16x4_f32.  Weights must be pre-packed by the matching
 * drba_pack_* call for the same `cfg`; cfg is chosen by drba_conv3x3_pick_cfg.
 * epilogue: y = acc + bias; if (beta) y = y*beta[c] + residual; if (act) y = lrelu_0.2(y). */
int drba_conv3x3_pick_cfg(int Cin, int Cout, int Ho, int Wo, int stride); /* cost-model default */
int drba_conv3x3_num_cfgs(void);         /* configs are 0..num-1; a host may time them and keep the fastest */
int drba_conv3x3_cfg_stride(int cfg);    /* the stride (1 or 2) a config was built for */
size_t drba_conv3x3_packed_floats(int Cin, int Cout, int cfg);
int drba_conv3x3_pack(const float *w /*[Cout,Cin,3,3] host or device-visible*/, float *packed,
                      int Cin, int Cout, int cfg);  /* HOST function: both pointers are host memory */
int drba_conv3x3(const float *in, const float *packed_w, const float *bias, const float *beta,
                 const float *residual, float *out, int N, int Cin, int H, int W, int Cout,
                 int stride, int act, int cfg, void *stream);

/* ConvTranspose2d(k=4, s=2, p=1) as four 2x2 phase convolutions; pixel_shuffle=1 writes
 * PixelShuffle(2) of the result directly (IFNet_HDv3.py:79-82), else plain [Cout,2H,2W]. */
int drba_deconv4x4_pick_cfg(int Cin, int Cout, int H, int W);
int drba_deconv4x4_num_cfgs(void);
size_t drba_deconv4x4_packed_floats(int Cin, int Cout, int cfg);
int drba_deconv4x4_pack(const float *w /*[Cin,Cout,4,4] host*/, float *packed, int Cin, int Cout, int cfg);
int drba_deconv4x4s2(const float *in, const float *packed_w, const float *bias, float *out,
                     int N, int Cin, int H, int W, int Cout, int pixel_shuffle, int cfg, void *stream);

/* ---- IFNet glue (IFNet_HDv3.py:84-96, :126-177) -------------------------------------------
 * Build one IFBlock's input at 1/scale resolution without materialising the full-resolution
 * concat: channels [warp(img0,flow[:2]) 3, warp(img1,flow[2:4]) 3, warp(f0) 16, warp(f1) 16,
 * timestep 1, (mask 1, feat 8, flow/scale 4 when flow != NULL)], bilinear-downsampled
 * (align_corners=False, src = scale*(dst+0.5)-0.5).  mask/feat are evaluated on the fly as the
 * x prev_scale bilinear upsample of the previous stage's 13-channel head output tmp_prev [13,hp,wp]
 * (channels 4 and 5..12), so they never exist at full resolution.
 * flow == NULL: first stage, no warp, 39 ch.  timestep_map may be NULL -> timestep_scalar. */
int drba_ifblock_input(const float *img0, const float *img1, const float *f0, const float *f1,
                       const float *timestep_map, float timestep_scalar, const float *flow,
                       const float *tmp_prev, int hp, int wp, float prev_scale, float *out,
                       int H, int W, int h, int w, float scale, void *stream);
/* Upsample the 13-channel head output by `scale` and fold it into the running flow:
 * flow_out = (flow_in ? flow_in : 0) + up(tmp[0:4])*scale.  mask / feat (full resolution,
 * = up(tmp[4]), up(tmp[5:13])) are written only when non-NULL. */
int drba_ifblock_update(const float *tmp, const float *flow_in, float *flow_out, float *mask,
                        float *feat, int h, int w, int H, int W, float scale, void *stream);
/* Final synthesis: out = warp(img0,flow[:2])*sigmoid(m) + warp(img1,flow[2:4])*(1-sigmoid(m)),
 * m = x scale bilinear upsample of mask_lo [h, w] (channel 4 of the last head output). */
int drba_warp_blend(const float *img0, const float *img1, const float *flow, const float *mask_lo,
                    int h, int w, float scale, float *out, int H, int W, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DRBA_HIP_H */
