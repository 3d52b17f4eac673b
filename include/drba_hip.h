/*
 * drba_hip.h — C ABI of libdrba_hip.so, the MI355X (gfx950) kernel library behind the
 * DRBA per-frame hot path.
 *
 * The reference (routineLife1/DRBA) has no FFI/plugin layer: its boundary is the Python
 * operator surface.  Each entry point below replaces one reference
 * operator (or a fused group of them) and cites it.  Conventions for every function:
 *   - all pointers are DEVICE pointers to contiguous fp32 NCHW data unless noted;
 *   - caller owns all memory; the library never allocates, frees or synchronises -- one exception (ABI version 2):
 *     drba_conv3x3 with a family-2 configuration keeps 64 bytes of work counters per (device, stream), allocated (hipMalloc +
 *     hipMemset + one device synchronisation) on that stream's first such launch on the current device and never freed --
 *     that first launch cannot be stream-captured; the table is mutex-protected;
 *   - kernels that need more than 64 KB of LDS raise their limit once per (kernel, device) (hipFuncSetAttribute applies to
 *     the current device): the library may be driven on several GPUs of one process;
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

/* ABI version.  9: drba_conv3x3_shuffle (a 3x3 convolution storing through PixelShuffle(2): GridNet's tail); drba_quad_interleave / drba_softsplat_index / drba_softsplat_gather_quad (drba_softsplat in pieces: the interleaved copy of a feature tensor kept by the caller).  8: configuration ids appended behind every earlier id of drba_conv3x3 (three, family 4: the waves of a workgroup split rows and
 * cout tiles) and drba_deconv4x4s2 (four: rows and couts split across the waves, both row phases per work item); drba_status_word / drba_status_clear (the always-on, synchronisation-free overflow report of kernel family 4); the
 * *_pack entry points of family 4 refuse (DRBA_EUNSUPPORTED) a weight the two-term fp16 form cannot hold (|w| >= 65504 or non-finite).
 * 7: drba_rife_splat_ws_floats -- the workspace of drba_flow_reverse / drba_drm_rife_linear(_batch) grew by a reach map in
 * front (scratch) and one flag per 32 x 16 output tile behind (same zero-on-entry, zero-on-return contract for everything but the map).  6: drba_stage_conv16_* (the fused stage input + first convolution in the two-term fp16 form, scale 1 and 2), drba_stage_item_t grew by
 * img0_x4 / img1_x4 ([H][W][4] frames), drba_to_inp_x4, drba_rgbx, drba_drm_rife_linear_batch, drba_set_range_check (debug).  5: kernel family 4 (two-term fp16 split; configuration ids appended behind every earlier id of drba_conv3x3 /
 * drba_deconv4x4s2, so no earlier id changes meaning), drba_deconv4x4_cfg_family, and a `terms` argument (3 or 2) on the
 * drba_linear_split_* entry points and on drba_window_attention; drba_head_fused16_*.  4: drba_softsplat_again; the encoder features may be given in the pair-interleaved layout ONLY: drba_head_fused accepts f_out ==
 * NULL (nothing but f_pair_out is written), and every stage-input entry point accepts items with f0 == f1 == NULL when
 * f0_pair / f1_pair are set (the first, unwarped stage reads the pair layout too).  3: drba_stage_item_t grew by term[DRBA_MAX_FLOW_TERMS]; drba_flow_terms_t and the entry points that take the
 * running flow as terms (drba_ifblock_input_lazy_batch, drba_warp_blend_lazy_batch); drba_stage_conv0_*.
 * 2: drba_timing_* and drba_softmax_expect2 removed; drba_flow_reverse and drba_drm_rife_linear take a
 * workspace that must be ZERO on entry (they leave it zero on return: self-cleaning accumulator) instead of clearing it
 * themselves; batched stage entry points added; drba_conv3x3_cfg_family added and configuration ids 19 (LDS-DMA, 32
 * channels) / 20 (K split across waves) behind drba_conv3x3; the allocation exception above.  1: the first release. */
#define DRBA_ABI_VERSION 9  /* the ONE place the number lives: api_misc.hip returns it, drba_amd/_lib.py parses it */
int drba_abi_version(void);
const char *drba_error_string(int code);
/* ABI 6, debug: with the range check on, every entry point that ran a kernel of family 4 (two fp16 terms per operand:
 * drba_conv3x3 / drba_deconv4x4s2 with a family-4 configuration, drba_linear_split_* and drba_window_attention with terms = 2,
 * drba_head_fused16, drba_stage_conv16_batch) scans the output it has just enqueued for inf / NaN -- one extra kernel, one
 * 4-byte copy and ONE STREAM SYNCHRONISATION per call -- and returns DRBA_EUNSUPPORTED instead of handing them on.  Family 4
 * holds an activation as fp16(x / 16) + ..., i.e. |x| >= 65504 * 16 ~ 1.05e6 overflows (weights, attention Q / V: 65504),
 * where families 0 - 3 keep fp32's range; a non-finite INPUT reads the same.  Returns the previous setting.  The Python
 * layer switches it on when DRBA_CHECK_RANGE=1 is set (drba_amd/_lib.py). */
int drba_set_range_check(int on);
/* ABI 8, always on once requested, no synchronisation: every kernel of family 4 (the list above) tests the values it STORES and,
 * when one is inf / NaN, writes 1 into its byte of the current device's status word -- 8 bytes of host-mapped memory, one byte
 * per kernel group (DRBA_STATUS_*), sticky until drba_status_clear().  The family's operands overflow fp16 where families 0 - 3
 * keep fp32's range (an activation of |x| >= 65504 * 16, an attention Q / V of 65504; an overflowed or non-finite INPUT reads the
 * same: it reaches the stored value), so a caller reads the word whenever it likes -- the model wrappers once per step -- and
 * raises instead of handing inf on; it sees every kernel that has FINISHED by then, a later one at the next read.
 * drba_status_word: allocates the word for the CURRENT device on the first call (hipHostMalloc: not inside a stream capture;
 * never freed) and returns its HOST address; until it has been called for a device the kernels launched there report nothing.
 * The cost on the kernels is one multiply-add per stored value in their epilogues. */
#define DRBA_STATUS_CONV_SPLIT 0 /* drba_conv3x3 / drba_deconv4x4s2, family-4 ids of conv_split.hip */
#define DRBA_STATUS_CONV_DMA 1   /* ... of conv_dma.hip (32 channels) */
#define DRBA_STATUS_CONV_KS 2    /* ... of conv_ks.hip (K split) */
#define DRBA_STATUS_LINEAR 3     /* drba_linear_split*, terms = 2 */
#define DRBA_STATUS_ATTENTION 4  /* drba_window_attention, terms = 2 */
#define DRBA_STATUS_STAGE 5      /* drba_stage_conv16_batch */
#define DRBA_STATUS_HEAD 6       /* drba_head_fused16 */
int drba_status_word(volatile unsigned long long **host_word);
int drba_status_clear(void);

/* ABI 8, optional: a HIP stream restricted to a subset of the CUs (hipExtStreamCreateWithCUMask) for callers that partition
 * the chip between the latency-bound low-resolution chain of the NEXT steps and the full-resolution kernels of the current
 * ones (drba_amd/models/lookahead.py; not in the reference, which has one stream).  mask: `words` x 32 bits, bit i = CU i in
 * the runtime's numbering (measured on MI355X, tools/exp/cu_mask/census.hip: consecutive bits go round the 8 XCDs, so a
 * contiguous range of 8 k bits is k CUs on every XCD).  The stream belongs to the caller (drba_stream_destroy); every other
 * entry point takes it like any stream. */
/* ABI 8: the per-(device, stream) work counters of the family-2 convolution (the allocation exception at the top) brought to
 * their initial state ON `stream`: the entry is created if the stream has none yet (hipMalloc + a device synchronisation: call it
 * once outside any capture), then one 64-byte memset is enqueued and the launch parity starts over.  A stream capture that
 * begins with this call replays correctly whatever its number of family-2 launches (two counter sets alternate per launch and
 * each launch clears the other set: a captured sequence with an odd count would otherwise start its second replay on a dirty
 * set); call it again on the stream before eager launches follow a replayed graph. */
int drba_conv_state_reset(void *stream);
int drba_stream_create_cu_mask(const uint32_t *mask, int words, void **stream);
int drba_stream_destroy(void *stream);

/* ---- kernel trace (measurement only; bench.py's roofline object) ---------------------------
 * Between drba_trace_begin() and drba_trace_end() every kernel the library launches (on any stream, from the one host
 * thread that drives it) carries an event pair on its own dispatch packet, i.e. each record is the kernel's own
 * execution time, as rocprofv3's kernel trace reports it.  drba_trace_begin() clears the records, drba_trace_end() stops
 * recording and drba_trace_resume() continues without clearing.  drba_trace_count() = records so far (launch order);
 * drba_trace_get(i, ...) waits for launch i and returns its demangled kernel name (without the argument list), its
 * grid in workgroups and its duration.  Not part of the reference's surface. */
int drba_trace_begin(void);
int drba_trace_end(void);
int drba_trace_resume(void);
int drba_trace_count(void);
int drba_trace_get(int i, const char **name, unsigned *grid3, float *ms);
/* start of launch i relative to the start of launch 0 (ms, from the same event pairs) and the stream it was issued on */
int drba_trace_get_start(int i, float *ms_since_first, unsigned long long *stream);

/* ---- forward splat ----------------------------------------------------------------------
 * replaces: models/softsplat/softsplat.py:248-293 (softsplat) + :306-367 (kernel softsplat_out)
 *           == models/softsplat/softsplat_torch.py:19-179.
 * mode: 0 sum, 1 avg, 2 linear, 3 soft.  eps: 0 addeps(+1e-7, default), 1 zeroeps, 2 clipeps.
 * metric: [N,1,H,W] or NULL (required for linear/soft).  ws: drba_softsplat_ws_floats(N,C,H,W) floats. */
int drba_softsplat(const float *in, const float *flow, const float *metric, float *out, float *ws,
                   int N, int C, int H, int W, int mode, int eps, void *stream);
size_t drba_softsplat_ws_floats(int N, int C, int H, int W);
/* ABI 4: ANOTHER input [N,C,H,W] splatted along the same flow with the same metric and mode: the sorted index the last
 * drba_softsplat call left in `ws` (same N, H, W; ws sized for the largest C of the calls that share it) is reused -- the
 * count / scan / fill launches are not repeated.  GMFSS_UNION splats a frame, its 64-channel features, the timestep map and
 * the ones-mask along one (flow, metric) pair (model_gmfss_union/GMFSS.py:92-117). */
int drba_softsplat_again(const float *in, float *out, float *ws, int N, int C, int H, int W, int mode, int eps, void *stream);
/* ABI 9: drba_softsplat in pieces, for a caller that keeps the quad-interleaved copy of a feature tensor across calls (the
 * gathers of inputs with C >= 16, C % 4 == 0 read [N][C/4][H*W][4]; drba_softsplat / _again rewrite that copy into `ws` on every
 * call, and GMFSS splats each pyramid level of a frame once per output frame of two consecutive steps):
 *   drba_quad_interleave         [N,C,H,W] -> [N][C/4][H*W][4] (C % 4 == 0);
 *   drba_softsplat_index         the sorted index of (flow, metric, mode) into `ws`, no gather (ws: drba_softsplat_ws_floats);
 *   drba_softsplat_gather_quad   drba_softsplat_again for a source already in that layout (DRBA_EUNSUPPORTED for other C). */
int drba_quad_interleave(const float *in, float *out, int N, int C, int H, int W, void *stream);
int drba_softsplat_index(const float *flow, const float *metric, float *ws, int N, int H, int W, int mode, void *stream);
int drba_softsplat_gather_quad(const float *in_quad, float *out, float *ws, int N, int C, int H, int W, int mode, int eps,
                               void *stream);

/* ---- backward warp -----------------------------------------------------------------------
 * replaces: models/rife_426_heavy/warplayer.py:8-22 (padding 0 = border) and
 *           models/model_gmfss_union/MetricNet.py:10-20 (padding 1 = zeros).
 * grid_sample(bilinear, align_corners=True) on base-grid + flow/((size-1)/2). */
int drba_backwarp(const float *in, const float *flow, float *out, int N, int C, int H, int W,
                  int padding, void *stream);

/* ---- flow magnitude: models/utils/tools.py:77-80 (distance_calculator) */
int drba_flow_distance(const float *flow, float *out, int N, int H, int W, void *stream);

/* ---- fused flow reversal: models/rife.py:59-73 (calc_flow tail)
 * out = 2 * where(splat_avg(1, f) < 0.999, max(H,W), -splat_avg(f, f)).  ws: drba_rife_splat_ws_floats(N, H, W, 2) floats.
 * ABI 7 layout: [1 MB of scratch: the reach map -- per 32 x 16 tile of sources the halo their splat flows need, written by the
 * pre-pass, so that an output tile scans a 4- or 8-pixel halo where nothing reaches it from further; any content on entry and
 * on return] [the accumulator of the rare long / converging sources, N*H*W*3 floats] [one flag per tile, set where the pre-pass
 * scattered into it: only those tiles read their accumulator entries].  Everything behind the first MB must be ZERO on entry; the
 * kernels write every entry they found non-zero back to zero, so it is zero again on return: allocate the buffer zeroed once
 * and keep it for these entry points (no memset per call). */
size_t drba_rife_splat_ws_floats(int N, int H, int W, int values); /* values: 2 = flow reversal, 1 = linear DRM */
int drba_flow_reverse(const float *flow, float *out, float *ws, int N, int H, int W, void *stream);

/* ---- fused linear DRM, one direction: models/drm.py:65-107 with linear=True
 * u = d_other/(d_self+d_other) * t * 2 with d = |flow| + eps; out = splat_avg(u, self*u) with
 * uncovered pixels (ones-splat < 0.999) keeping u.  drm_t1_t01 = (self=flow10, other=flow12),
 * drm_t1_t12 = (self=flow12, other=flow10).  ws: drba_rife_splat_ws_floats(N, H, W, 1) floats, zero on entry and on return (as above).  If t_dev != NULL the timestep
 * is read from that device float instead of `t` (so a captured HIP graph can be replayed for any t). */
int drba_drm_rife_linear(const float *flow_self, const float *flow_other, float t, const float *t_dev,
                         float eps, float *out, float *ws, int N, int H, int W, void *stream);
/* ABI 6: n_jobs (<= DRBA_MAX_STAGE_ITEMS) maps of one geometry in one launch pair: job k = drba_drm_rife_linear(flow_self,
 * flow_other, t, NULL, eps, out, ., 1, H, W) on tensors of its own (the maps of a group of DRBA steps: 2 launches instead of
 * 2 per map).  ws: drba_rife_splat_ws_floats(n_jobs, H, W, 1) floats, zero on entry and on return. */
typedef struct drba_drm_job {
  const float *flow_self, *flow_other; /* [2,H,W] each */
  float t;
  float *out; /* [1,H,W] */
} drba_drm_job_t;
int drba_drm_rife_linear_batch(const drba_drm_job_t *jobs, int n_jobs, float eps, float *ws, int H, int W, void *stream);

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
/* to_inp (tools.py:59-60 = resize(to_tensor(img), dst_size)) and to_out (tools.py:63-64 = to_cv2(resize(x, src_size))) as
 * ONE kernel each: uint8 HWC -> /255. -> bilinear -> fp32 [1,3,Hout,Wout], and fp32 [1,3,Hin,Win] -> bilinear -> *255.
 * truncated -> uint8 HWC; reverse_channels != 0 also performs the BGR -> RGB flip of the encoder pipe (tools.py:202).
 * Arithmetic pinned to ATen's CPU kernel (fma(scale, dst+0.5, -0.5); fma(w0, a, w1*b) per axis): bit-exact. */
int drba_to_inp(const uint8_t *img_hwc, float *out, int Hin, int Win, int Hout, int Wout, float scale_y, float scale_x,
                void *stream);
/* ABI 6: drba_to_inp that also writes the frame pixel-major, [Hout][Wout][4] = (c0, c1, c2, 0) (out_x4, 16-byte aligned, may be
 * NULL): the copy the gathers read their image taps from (drba_stage_item_t.img0_x4), at no extra pass over the frame. */
int drba_to_inp_x4(const uint8_t *img_hwc, float *out, float *out_x4, int Hin, int Win, int Hout, int Wout, float scale_y,
                   float scale_x, void *stream);
int drba_to_out(const float *in, uint8_t *out_hwc, int Hin, int Win, int Hout, int Wout, float scale_y, float scale_x,
                int reverse_channels, void *stream);

/* ---- scene-cut metric: tools.py:27-30 + pytorch_msssim/__init__.py:83-136 (ssim_matlab)
 * x1, x2: [1,3,32,32] thumbnails (already resized); out: 1 float on device. */
int drba_ssim3d_32(const float *x1, const float *x2, float *out, void *stream);

/* ---- convolutions (models/rife_426_heavy/IFNet_HDv3.py:11-16, :28-47, :50-59, :65-82) ------
 * fp32 implicit GEMM on v_mfma_f32_16x16x4_f32.  Weights must be pre-packed by the matching
 * drba_pack_* call for the same `cfg`; cfg is chosen by drba_conv3x3_pick_cfg.
 * loader:   pre_act != 0 applies PReLU with one shared slope (nn.PReLU()) to the input as it is staged
 *           (FeatureNet.py:9-27, FusionNet.py:6-31: PReLU precedes every conv).
 * epilogue: y = acc + bias; if (beta) y = y*beta[c] + residual (ResConv) else y += residual (+ residual2);
 *           act: 0 none, 1 LeakyReLU(0.2), 2 PReLU(post_slope), 3 ReLU, 4 tanh(y)*10 (MetricNet.py:41-42,63). */
int drba_conv3x3_pick_cfg(int Cin, int Cout, int Ho, int Wo, int stride); /* cost-model default */
int drba_conv3x3_num_cfgs(void);         /* configs are 0..num-1; a host may time them and keep the fastest */
int drba_conv3x3_cfg_stride(int cfg);    /* the stride (1 or 2) a config was built for */
/* kernel family of a config: 0 = fp32 MFMA (conv.hip), 1 = split-bf16 with register staging (conv_split.hip: stride 1,
 * Cin % 32 == 0), 2 = split-bf16 with every operand streamed by LDS-DMA (conv_dma.hip: additionally W % 4 == 0; the
 * launch returns DRBA_EUNSUPPORTED otherwise: Cin == 32, Cout <= 32), 3 = split-bf16 with K split across the waves of a
 * workgroup (conv_ks.hip: Cin = 64 / 96 / 128 / 192, small maps), 4 = the two-term fp16 split (the tiles of families
 * 1 - 3 with each fp32 operand taken as h + 2^-11 l, two fp16 terms = 22 significand bits, three matrix-core products
 * instead of six, fp32 accumulation; activations must stay below 65504 * 16 in magnitude).  Families 1 - 3 reproduce the fp32 product to the
 * last bit of the operands; family 4 drops operand bits 23 - 24, which the fp32 accumulation over 9 Cin terms hides
 * (measured against fp64: tests/gpu_checks.py check_conv_layers, the same bound for all families).  A host that wants
 * 24-bit operands everywhere leaves family 4 out of the ids it offers to its tuner. */
int drba_conv3x3_cfg_family(int cfg);
size_t drba_conv3x3_packed_floats(int Cin, int Cout, int cfg);
int drba_conv3x3_pack(const float *w /*[Cout,Cin,3,3] host or device-visible*/, float *packed,
                      int Cin, int Cout, int cfg);  /* HOST function: both pointers are host memory */
int drba_conv3x3(const float *in, const float *packed_w, const float *bias, const float *beta,
                 const float *residual, const float *residual2, float *out, int N, int Cin, int H, int W,
                 int Cout, int stride, int act, float post_slope, int pre_act, float pre_slope, int cfg,
                 void *stream);

/* ABI 9.  drba_conv3x3 (stride 1, bias, activation; no residual operands, no pre-activation) storing through
 * PixelShuffle(2): out is [N, Cout / 4, 2H, 2W] (reference: models/model_gmfss_union/FusionNet.py:100-103, GridNet's
 * upsample conv + nn.PixelShuffle(2)).  packed_w is drba_conv3x3_pack's output for the same cfg.  Only configurations whose
 * tile carries that store form accept (today: the two-term 4 x 32 x 64 tiles); every other id returns DRBA_EUNSUPPORTED and
 * the caller runs drba_conv3x3 + drba_pixel_shuffle2.  W % 4 == 0, Cout % 4 == 0. */
int drba_conv3x3_shuffle(const float *in, const float *packed_w, const float *bias, float *out, int N, int Cin, int H, int W,
                         int Cout, int act, float post_slope, int cfg, void *stream);

/* ConvTranspose2d(k=4, s=2, p=1) as four 2x2 phase convolutions; pixel_shuffle=1 writes
 * PixelShuffle(2) of the result directly (IFNet_HDv3.py:79-82), else plain [Cout,2H,2W]. */
int drba_deconv4x4_pick_cfg(int Cin, int Cout, int H, int W);
int drba_deconv4x4_num_cfgs(void);
int drba_deconv4x4_cfg_family(int cfg);  /* 0 fp32 MFMA, 1 three-term bf16 split, 4 two-term fp16 split (as above) */
size_t drba_deconv4x4_packed_floats(int Cin, int Cout, int cfg);
int drba_deconv4x4_pack(const float *w /*[Cin,Cout,4,4] host*/, float *packed, int Cin, int Cout, int cfg);
int drba_deconv4x4s2(const float *in, const float *packed_w, const float *bias, float *out,
                     int N, int Cin, int H, int W, int Cout, int pixel_shuffle, int pre_act, float pre_slope,
                     int cfg, void *stream);

/* ---- a chain of convolution layers in one call ------------------------------------------------
 * replaces: IFBlock.conv0 + convblock + lastconv (IFNet_HDv3.py:64-83, :90-96) or Head (:28-47) issued layer by
 * layer.  Layer i reads layer i-1's output (layer 0: `in`); the last layer writes `out`, the others alternate
 * between scratch0 / scratch1 (each sized for the largest intermediate).  conv layers: drba_conv3x3 semantics with
 * `residual` = add the layer's own input in the epilogue (needs beta for the ResConv form); deconv layers:
 * drba_deconv4x4s2 semantics.  packed_w / cfg as for the single-layer entry points. */
typedef struct drba_conv_layer {
  const float *packed_w, *bias, *beta;
  int cin, cout, stride, act, cfg, residual, deconv, pixel_shuffle;
} drba_conv_layer_t;
int drba_conv_chain(const float *in, float *out, float *scratch0, float *scratch1,
                    const drba_conv_layer_t *layers, int n_layers, int N, int H, int W, void *stream);

/* ---- IFNet glue (IFNet_HDv3.py:84-96, :126-177) -------------------------------------------
 * Build one IFBlock's input at 1/scale resolution without materialising the full-resolution
 * concat: channels [warp(img0,flow[:2]) 3, warp(img1,flow[2:4]) 3, warp(f0) 16, warp(f1) 16,
 * timestep 1, (mask 1, feat 8, flow/scale 4 when flow != NULL)], bilinear-downsampled
 * (align_corners=False, src = scale*(dst+0.5)-0.5).  mask/feat are evaluated on the fly as the
 * x prev_scale bilinear upsample of the previous stage's 13-channel head output tmp_prev [13,hp,wp]
 * (channels 4 and 5..12), so they never exist at full resolution.
 * flow == NULL: first stage, no warp, 39 ch.  timestep_map may be NULL -> timestep_scalar.
 * f0_pair / f1_pair: optional copies of f0 / f1 in the pair-interleaved layout [8][H][W][2] written by
 * drba_pair_interleave (both or neither); the warped stages then fetch two channels per 16-byte load. */
int drba_ifblock_input(const float *img0, const float *img1, const float *f0, const float *f1,
                       const float *f0_pair, const float *f1_pair,
                       const float *timestep_map, float timestep_scalar, const float *flow,
                       const float *tmp_prev, int hp, int wp, float prev_scale, float *out,
                       int H, int W, int h, int w, float scale, void *stream);
/* The warped stage input (flow != NULL form of drba_ifblock_input) with tmp_prev's footprint staged through LDS, and --
 * when flow_out != NULL -- with the PREVIOUS stage's flow update (IFNet_HDv3.py:92-95,160) folded in:
 *   flow_new = (flow ? flow : 0) + up(tmp_prev[0:4]) * prev_scale   is formed per sample point, written to flow_out
 *   [4,H,W] and used for the warps (flow = the running flow BEFORE that update; drba_ifblock_update is then not called).
 * The fold requires scale <= 2 (every full-resolution pixel is a sample point exactly once).  flow_out == NULL: `flow` is
 * the finished running flow, as in drba_ifblock_input.  scale in {1,2,4,...,32}, prev_scale == 2*scale (IFNet's pyramid). */
int drba_ifblock_input_lds(const float *img0, const float *img1, const float *f0, const float *f1,
                           const float *f0_pair, const float *f1_pair, const float *timestep_map,
                           float timestep_scalar, const float *flow, const float *tmp_prev, int hp, int wp,
                           float prev_scale, float *flow_out, float *out, int H, int W, int h, int w, float scale,
                           void *stream);
/* Batched forms: the items of one stage (the interpolations of one step: `-t 2` -> 2) in ONE launch per kernel.  Each item
 * names its own frames / features / timestep / flow / previous head output / outputs (the fields have the meaning of the
 * arguments above; flow_out != NULL requests the fold); the items agree on which optional pointers are given.
 * n_items <= DRBA_MAX_STAGE_ITEMS. */
#define DRBA_MAX_STAGE_ITEMS 8
#define DRBA_MAX_FLOW_TERMS 4
typedef struct drba_stage_item {
  const float *img0, *img1, *f0, *f1, *f0_pair, *f1_pair, *timestep_map;
  float timestep_scalar;
  const float *flow, *tmp_prev;
  float *flow_out, *out;
  const float *term[DRBA_MAX_FLOW_TERMS]; /* the "lazy" entry points: head outputs [13,h_i,w_i] of the stages BEFORE tmp_prev, oldest first */
  /* ABI 6, optional (both or neither; 16-byte aligned): the two frames additionally as [H][W][4] = (c0, c1, c2, 0) -- drba_rgbx, or
   * drba_to_inp's second output.  The gathers are bound by the NUMBER of vector-memory instructions a wave issues: with a pixel's
   * three channels in one 16-byte unit the two taps of a row are two 16-byte loads instead of three 8-byte ones (one per plane). */
  const float *img0_x4, *img1_x4;
} drba_stage_item_t;
/* The running flow as a list of terms instead of a full-resolution tensor (IFNet_HDv3.py:146-160: flow = flow + up(tmp_i[:, :4]) * s_i
 * after every stage): term i is the head output of an earlier stage, [13, h[i], w[i]], upsampled x scale[i].  The entry points
 * that take one evaluate flow = sum_i up(term_i[0:4]) * scale[i] + up(tmp_prev[0:4]) * prev_scale at their sample points
 * (every product and sum rounded as the stored updates would be) -- no drba_ifblock_update pass, no flow tensor.
 * item.flow and item.flow_out must be NULL there. */
typedef struct drba_flow_terms {
  int n;
  int h[DRBA_MAX_FLOW_TERMS], w[DRBA_MAX_FLOW_TERMS];
  float scale[DRBA_MAX_FLOW_TERMS];
} drba_flow_terms_t;
int drba_ifblock_input_batch(const drba_stage_item_t *items, int n_items, int hp, int wp, float prev_scale, int H, int W,
                             int h, int w, float scale, void *stream);
int drba_ifblock_input_lds_batch(const drba_stage_item_t *items, int n_items, int hp, int wp, float prev_scale, int H, int W,
                                 int h, int w, float scale, void *stream);
/* The scale-1 stage input fused with the IFBlock's first convolution (IFNet_HDv3.py:85-88 followed by conv0[0] of :64-66,
 * 52 -> 16 channels, stride 2, pad 1, LeakyReLU(0.2)): drba_ifblock_input_lds_batch(scale = 1) + drba_conv3x3(stride 2)
 * in one kernel, the 52-channel stage input never reaches HBM (stage_conv.hip).  items[k].out is the CONVOLUTION's output
 * [16, (H-1)/2+1, (W-1)/2+1]; f0_pair / f1_pair and tmp_prev are required; flow_out non-NULL folds the previous stage's
 * flow update in exactly as drba_ifblock_input_lds_batch does; with `terms` the flow is the lazy sum.  prev_scale must be 2.  `packed_w`: device copy of
 * drba_stage_conv0_pack's output (HOST function: w [16,52,3,3] and packed are host memory). */
/* IFNet's context encoder (IFNet_HDv3.py:23-47 `Head`: conv 3->16 stride 2, two convs 16->16, each + LeakyReLU(0.2), then
 * ConvTranspose2d(16,16,4,2,1)) in one kernel, intermediates in LDS (head_fused.hip).  img [N,3,H,W] -> f_out [N,16,H,W] and
 * f_pair_out, the same values pair-interleaved per sample ([8,H,W,2]: what drba_pair_interleave(f_out) would write).
 * f_out may be NULL (ABI 4): only the pair layout is written -- every kernel of the pipeline reads that one.
 * H even, W % 4 == 0, 16-byte aligned outputs.  drba_head_fused_pack is a HOST function (weights in the reference's
 * layouts: w0 [16,3,3,3], w1 / w2 [16,16,3,3], w3 [16 in,16 out,4,4], biases [16]); packed_w is its output copied to the device. */
size_t drba_head_fused_packed_floats(void);
int drba_head_fused_pack(const float *w0, const float *b0, const float *w1, const float *b1, const float *w2, const float *b2,
                         const float *w3, const float *b3, float *packed);
int drba_head_fused(const float *img, const float *packed_w, float *f_out, float *f_pair_out, int N, int H, int W, void *stream);
/* the same fusion in the two-term fp16 form (kernel family 4 of drba_conv3x3_cfg_family; head_fused16.hip): same arguments,
 * same outputs to the tolerance of the family, its own packing */
size_t drba_head_fused16_packed_floats(void);
int drba_head_fused16_pack(const float *w0, const float *b0, const float *w1, const float *b1, const float *w2, const float *b2,
                           const float *w3, const float *b3, float *packed);
int drba_head_fused16(const float *img, const float *packed_w, float *f_out, float *f_pair_out, int N, int H, int W, void *stream);
/* drba_ifblock_input_lds_batch with the flow given as terms (any scale of the pyramid; nothing but `out` is written). */
int drba_ifblock_input_lazy_batch(const drba_stage_item_t *items, int n_items, const drba_flow_terms_t *terms, int hp, int wp,
                                  float prev_scale, int H, int W, int h, int w, float scale, void *stream);
/* drba_warp_blend_fold for the items of a stage in one launch, the flow before the last stage given as terms: the items'
 * img0, img1, tmp_prev (the LAST head output [13,h,w], stage scale `scale` >= 1), term[] and out ([3,H,W]) are read. */
int drba_warp_blend_lazy_batch(const drba_stage_item_t *items, int n_items, const drba_flow_terms_t *terms, int h, int w,
                               float scale, int H, int W, void *stream);
size_t drba_stage_conv0_packed_floats(void);
int drba_stage_conv0_pack(const float *w, float *packed);
int drba_stage_conv0_supported(int H, int W, float scale, float prev_scale, int Cout);
int drba_stage_conv0_batch(const drba_stage_item_t *items, int n_items, const drba_flow_terms_t *terms /* NULL: flow / fold as above */,
                           int hp, int wp, float prev_scale, int H, int W, const float *packed_w, const float *bias, void *stream);
/* ABI 6: the same fusion in the two-term fp16 form (kernel family 4 of drba_conv3x3_cfg_family; stage_conv16.hip): the gathered
 * values are split into two fp16 terms on their way into LDS and the convolution runs on the 16-bit matrix pipe (3 products per
 * multiply, fp32 accumulation), the channels in 4 groups of 16 instead of 13 of 4.  Same items / terms / hp / wp / prev_scale
 * semantics, same flows (bit for bit), convolution output to the tolerance of the family; w [Cout,52,3,3].
 * `scale` = 1: Cout = 16, prev_scale = 2, any of the three flow forms.  `scale` = 2 (stage_conv16_s2): the stage input at half
 * resolution (each pixel the mean of its 2 x 2 warped sample points), out [Cout, ((H/2)-1)/2+1, ((W/2)-1)/2+1], Cout = 16 or 32,
 * prev_scale = 4, the flow as `terms` (scales >= 8) and items with img0_x4 / img1_x4 only -- DRBA_EUNSUPPORTED otherwise (the
 * caller takes drba_ifblock_input_lazy_batch + drba_conv3x3).  packed_w 16-byte aligned.  drba_stage_conv16_pack is a HOST function. */
size_t drba_stage_conv16_packed_floats(int Cout);
int drba_stage_conv16_pack(const float *w, int Cout, float *packed);
int drba_stage_conv16_supported(int H, int W, float scale, float prev_scale, int Cout);
int drba_stage_conv16_batch(const drba_stage_item_t *items, int n_items, const drba_flow_terms_t *terms, int hp, int wp, float prev_scale,
                            int H, int W, float scale, int Cout, const float *packed_w, const float *bias, void *stream);
/* drba_ifblock_update for several items (arrays of n_items device pointers; flow_in may be NULL or hold NULLs). */
int drba_ifblock_update_batch(const float *const *tmp, const float *const *flow_in, float *const *flow_out, int n_items,
                              int h, int w, int H, int W, float scale, void *stream);
/* [C,H,W] -> [C/2,H,W,2] (C even): channel pairs interleaved per pixel. */
int drba_pair_interleave(const float *in, float *out, int C, int H, int W, void *stream);
/* ABI 6: a frame [3,H,W] -> [H,W,4] = (c0, c1, c2, 0), 16-byte aligned (drba_stage_item_t.img0_x4 / img1_x4). */
int drba_rgbx(const float *img, float *out, int H, int W, void *stream);
/* Upsample the 13-channel head output by `scale` and fold it into the running flow:
 * flow_out = (flow_in ? flow_in : 0) + up(tmp[0:4])*scale.  mask / feat (full resolution,
 * = up(tmp[4]), up(tmp[5:13])) are written only when non-NULL. */
int drba_ifblock_update(const float *tmp, const float *flow_in, float *flow_out, float *mask,
                        float *feat, int h, int w, int H, int W, float scale, void *stream);
/* Final synthesis: out = warp(img0,flow[:2])*sigmoid(m) + warp(img1,flow[2:4])*(1-sigmoid(m)),
 * m = x scale bilinear upsample of mask_lo [h, w] (channel 4 of the last head output). */
int drba_warp_blend(const float *img0, const float *img1, const float *flow, const float *mask_lo,
                    int h, int w, float scale, float *out, int H, int W, void *stream);
/* drba_warp_blend with the LAST stage's flow update folded in: flow = (flow_prev ? flow_prev : 0) + up(tmp_last[0:4]) * scale
 * per pixel (not stored), mask = channel 4 of tmp_last [13,h,w]. */
int drba_warp_blend_fold(const float *img0, const float *img1, const float *flow_prev, const float *tmp_last, int h, int w,
                         float scale, float *out, int H, int W, void *stream);

/* ---- GMFSS / GMFSS_UNION glue (models/model_gmfss_union: MetricNet.py, FusionNet.py, GMFSS.py) ---
 * MetricNet.forward input (MetricNet.py:45-60, geometry.py:87-108), 14 channels at the half-res size:
 * [img0 3, img1 3, -mean_c|img0 - backwarp0(img1,f01)|, -mean_c|img1 - backwarp0(img0,f10)|,
 *  f01/((W-1)/2,(H-1)/2), f10/(..), fwd_occ, bwd_occ] with occ = |f + flow_warp(b,f)| > 0.01(|f|+|b|)+0.5 */
int drba_metric_input(const float *img0, const float *img1, const float *flow01, const float *flow10,
                      float *out, int H, int W, void *stream);
/* nn.PixelShuffle(2): in [4C,H,W] -> out [C,2H,2W] (FusionNet.py:41-44) */
int drba_pixel_shuffle2(const float *in, float *out, int C, int H, int W, void *stream);
/* GMFSS.py:116-122: out_k = (cover0 < 0.999 || cover1 < 0.999) ? 1 : t_k */
int drba_timestep_fix(const float *t0, const float *t1, const float *cover0, const float *cover1,
                      float *out0, float *out1, size_t n, void *stream);
/* GMFSS.py:125-150: out_x = (t0/t1 > thr) ? y : x; out_y = (t1/t0 > thr) ? x : y; t maps [H,W] broadcast over C */
int drba_swap_select(const float *x, const float *y, const float *t0, const float *t1, float *out_x,
                     float *out_y, int C, int H, int W, float thr, void *stream);
/* torch.clamp(x, lo, hi) (GMFSS.py:155) */
int drba_clamp(const float *in, float *out, float lo, float hi, size_t n, void *stream);

/* ---- GMFlow operators around its matrix products (models/gmflow) ---------------------------------
 * The q/k/v/merge/MLP projections are drba_linear_split_*, the windowed QK^T / softmax / PV is drba_window_attention, the global
 * and local correlations drba_global_expect2 / drba_local_corr_flow (all hand-written MFMA kernels of this library: nothing is
 * issued through a vendor BLAS); these entry points are everything around them. */
/* generic direct convolution, zero padding (backbone.py:46 7x7 s2 stem, :63 / :24 1x1 projections) */
int drba_conv_direct(const float *in, const float *w /*[Cout,Cin,K,K]*/, const float *bias, float *out,
                     int N, int Cin, int H, int W, int Cout, int K, int stride, int pad, void *stream);
/* nn.InstanceNorm2d(eps, no affine) over `planes` = N*C planes of HW elements, optional ReLU (backbone.py:27-36) */
int drba_instance_norm(const float *in, float *out, float *ws, int planes, size_t HW, float eps, int relu,
                       void *stream); /* ws: drba_instance_norm_ws_floats(planes) floats */
size_t drba_instance_norm_ws_floats(int planes);
int drba_add_act(const float *a, const float *b, float *out, size_t n, int relu, void *stream); /* a + b [, ReLU] */
/* utils.py:57-69 normalize_img: (x - mean[c]) / std[c]; mean3/std3 are HOST arrays of 3 floats */
int drba_channel_normalize3(const float *in, float *out, int N, size_t HW, const float *mean3, const float *std3,
                            void *stream);
/* nn.LayerNorm(cols): out = (residual ? residual : 0) + LN(x)*w + b  (transformer.py:178-185) */
int drba_layernorm(const float *x, const float *w, const float *b, const float *residual, float *out,
                   size_t rows, int cols, float eps, void *stream);
int drba_gelu(const float *x, float *out, size_t n, void *stream); /* nn.GELU(), erf form */
/* single_head_split_window_attention (transformer.py:46-113) in one kernel: q, k, v, out are [B, H*W, C] (C = 128);
 * the roll by half a window (shift != 0), the splits x splits window partition, the -100 region mask of
 * generate_shift_window_attn_mask (transformer.py:19-43), softmax(q k^T / scale) v and the inverse partition / roll
 * are applied through index maps; the score matrix is never stored.  splits = 1, shift = 0 is full attention. */
int drba_window_attention(const float *q, const float *k, const float *v, float *out, int B, int H, int W, int C,
                          int splits, int shift, float scale, int ldq, int ldk, int ldv, float *ws, int terms, void *stream);
/* terms: 3 = both GEMMs on the fp32 matrix cores; 2 = their operands as two fp16 terms (kernel family 4: 22 bits, fp32
 * accumulation; scores, masks and the softmax stay fp32) */
/* ws: drba_window_attention_ws_floats(B, H, W, splits) floats (may be 0 / NULL): shapes with few, long windows split each
 * window's keys over several workgroups and merge the partial softmax states through it */
size_t drba_window_attention_ws_floats(int B, int H, int W, int splits);
/* ldq / ldk / ldv: row strides in floats (>= C, multiples of 4; 16-byte aligned bases): q, k, v may be column slices
 * of one fused projection output [B*H*W, 3C]; out rows are C apart */
/* nn.Linear on token-major activations (transformer.py:142-208: q/k/v/merge projections, MLP 256 -> 1024 -> GELU -> 128):
 * out[M, N] = x[M, K] (row stride ldx floats) . w[N, K]^T (+ bias) (, exact GELU).  fp32 operands evaluated on the 16-bit
 * matrix cores with fp32 accumulation, `terms` = 3: three bf16 terms each (24 bits, six products), `terms` = 2: two fp16
 * terms (22 bits, three products; kernel family 4 of drba_conv3x3_cfg_family) -- the value a weight was PACKED with must
 * be the one it is used with; K % 32 == 0.  The weight is packed once with drba_linear_split_pack (host buffers;
 * drba_linear_split_packed_floats floats). */
size_t drba_linear_split_packed_floats(int K, int N, int terms);
int drba_linear_split_pack(const float *w /*[N,K] host*/, float *packed, int K, int N, int terms);
int drba_linear_split(const float *x, const float *packed_w, const float *bias, float *out, int M, int K, int N, int ldx,
                      int gelu, int terms, void *stream);
/* the same on cat(x1, x2) along the features without materialising it (transformer.py:201: mlp(cat(source, message))):
 * x1 [M, K1], x2 [M, K2], w [N, K1 + K2]; K1, K2 multiples of 32 */
int drba_linear_split_cat(const float *x1, const float *x2, const float *packed_w, const float *bias, float *out, int M,
                          int K1, int K2, int N, int ldx1, int ldx2, int gelu, int terms, void *stream);
/* the same for N = 128 with the layer's norm fused (transformer.py:178-185, :203-207):
 * out[M,128] = (residual ? residual : 0) + LayerNorm_128(x . w^T + bias) * ln_w + ln_b */
int drba_linear_split_layernorm(const float *x, const float *packed_w, const float *bias, const float *ln_w,
                                const float *ln_b, const float *residual, float *out, int M, int K, int ldx, float eps,
                                int terms, void *stream);
/* in-place row softmax of x/scale + mask[(row/rows_per_mat) % n_masks][row % rows_per_mat] (transformer.py:91-96) */
int drba_softmax_rows(float *x, const float *mask, size_t rows, int cols, int rows_per_mat, int n_masks,
                      float scale, void *stream);
/* Global correlation / global flow propagation WITHOUT the L x L score matrix (matching.py:7-38 global_correlation_softmax;
 * transformer.py:355-372, FeatureFlowAttention's global branch): out[2][L] = softmax_j(<q_i, k_j> / scale) . val_j, flash
 * style on the fp32 matrix cores.  q, k: token-major [L][ldq / ldk] rows of C = 128 features.  vals == NULL: val_j = pixel
 * coordinate (j % w, j / w) and the query's own coordinate is subtracted (the correlation flow); else vals = [2][L] planes
 * (the flow to propagate).  ws: drba_global_expect2_ws_floats(L) floats (NULL: no key split). */
int drba_global_expect2(const float *q, const float *k, const float *vals, float *out, float *ws, int L, int C, int w,
                        float scale, int ldq, int ldk, void *stream);
size_t drba_global_expect2_ws_floats(int L);
/* plain batched fp32 matrix product C[b] = A[b] B[b]^T (trans_b) or A[b] B[b]: only the degenerate shifted-window case of
 * transformer.py:46-113 (windows one pixel wide, frames below 128 px) uses it; everything else is fused */
int drba_bmm(const float *a, const float *b, float *c, int batch, int M, int N, int K, int trans_b, void *stream);
/* matching.py:41-89: local correlation softmax flow, radius r, features NCHW */
int drba_local_corr_flow(const float *f0, const float *f1, float *out, int C, int H, int W, int radius, void *stream);
/* transformer.py:374-409: local-window flow propagation; q_tok / k_tok token-major [H*W, C], flow [2,H,W] */
int drba_local_attn_flow(const float *q_tok, const float *k_tok, const float *flow, float *out, int C, int H,
                         int W, int radius, void *stream);
/* gmflow.py:76-89: convex upsampling, mask [9*factor^2, h, w], flow [2,h,w] -> out [2, factor*h, factor*w] */
int drba_convex_upsample(const float *mask, const float *flow, float *out, int h, int w, int factor, void *stream);
/* geometry.py:53-84 flow_warp: bilinear, zeros padding */
int drba_flow_warp(const float *in, const float *flow, float *out, int C, int H, int W, void *stream);
/* F.interpolate(bilinear, align_corners=True) * mul (gmflow.py:118) */
int drba_resize_bilinear_ac(const float *in, float *out, int NC, int Hin, int Win, int Hout, int Wout, float mul,
                            void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DRBA_HIP_H */
