// The running flow of IFNet, evaluated where it is needed instead of kept as a full-resolution tensor.
// IFNet_HDv3.py:146-160: after stage i, flow = flow + up(tmp_i[:, :4], x s_i) * s_i (bilinear, align_corners=False), starting
// from None.  The unfused form materialises that sum at full resolution after every stage (8 P bytes of read-modify-write
// per sample and stage: 0.9 of the 9.4 ms of kernel time of a 4K scale-0.5 step) only for the next stage's gather to read
// it back at its sample points.  A "term" is one earlier head output [13, h_i, w_i] with its stage scale: the kernels that
// need the flow at a point (stage-input gathers, stage_conv0, the final warp_blend) stage the 4 flow channels of every
// term under their tile in LDS (a few hundred floats: the maps are 1/4 .. 1/32 resolution) and form
//   flow(X, Y) = ((up_0 * s_0 + up_1 * s_1) + ...)       oldest term first, every product and sum rounded to fp32
// exactly as the sequence of stored updates would: same values as the materialised flow up to the contraction of the
// upsample itself (lerp2_fma).  The newest head output (tmp_prev, also the source of mask / feat) is added by the caller.
#pragma once
#include <string.h>

#include "common.hpp"

namespace drba {

constexpr int kMaxTerms = DRBA_MAX_FLOW_TERMS;
typedef float f32x4t __attribute__((ext_vector_type(4)));

struct FlowTermsArg {  // kernel-argument form of drba_flow_terms_t
  int n;
  int h[kMaxTerms], w[kMaxTerms];
  float scale[kMaxTerms], inv[kMaxTerms];
};

static inline bool flow_terms_arg(const drba_flow_terms_t *t, FlowTermsArg &a) {
  memset(&a, 0, sizeof(a));
  if (!t) return true;
  if (t->n < 0 || t->n > kMaxTerms) return false;
  a.n = t->n;
  for (int i = 0; i < t->n; ++i) {
    if (t->h[i] <= 0 || t->w[i] <= 0 || !(t->scale[i] > 0.f)) return false;
    a.h[i] = t->h[i], a.w[i] = t->w[i], a.scale[i] = t->scale[i], a.inv[i] = (float)(1.0 / (double)t->scale[i]);
  }
  return true;
}

// Footprints of the terms under a tile whose sample points span [Xa, Xb] x [Ya, Yb] (full resolution, in-image).
// lds (16-byte aligned): [kMaxTerms][CAP_R * CAP_C][4 channels] -- the four flow channels of a footprint pixel are one
// 16-byte LDS word, so a sample point reads a term's 2 x 2 taps with 4 ds_read_b128 instead of 16 ds_read_b32 (round 4: the
// final warp_blend evaluates 4 terms per pixel, 64 of its 84 LDS reads); org[i] = (rx0, ry0) of term i.  All threads of the workgroup call this (before a
// barrier of their own).  CAP must cover (extent - 1) / s_i + 3 per axis; footprints are clipped to it defensively.
template <int CAP_R, int CAP_C, int NTHREADS>
__device__ __forceinline__ void terms_stage(float *lds, const FlowTermsArg &T, const float *const *ptr, int Xa, int Ya, int Xb, int Yb,
                                            int tid, int (&rx0)[kMaxTerms], int (&ry0)[kMaxTerms]) {
  constexpr int CAP = CAP_R * CAP_C;
  constexpr int PER = (CAP + NTHREADS - 1) / NTHREADS;  // footprint pixels per lane and term
  // all loads of all terms are issued before the first LDS write: one memory round trip for the workgroup's prologue instead
  // of one per term (round 5: the per-term "load, wait, write" loop was 4 dependent L2 / HBM latencies on the critical path of
  // every gather workgroup, 2-8k clocks)
  f32x4t v[kMaxTerms][PER];
  bool on[kMaxTerms][PER];
#pragma unroll
  for (int i = 0; i < kMaxTerms; ++i) {
    rx0[i] = ry0[i] = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) on[i][k] = false, v[i][k] = (f32x4t){0.f, 0.f, 0.f, 0.f};
    if (i < T.n) {
      rx0[i] = lerp_src(Xa, T.inv[i], T.w[i]).i0, ry0[i] = lerp_src(Ya, T.inv[i], T.h[i]).i0;
      const int rw = min(lerp_src(Xb, T.inv[i], T.w[i]).i1 - rx0[i] + 1, CAP_C);
      const int rh = min(lerp_src(Yb, T.inv[i], T.h[i]).i1 - ry0[i] + 1, CAP_R);
      const size_t plane = (size_t)T.h[i] * T.w[i];
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const int e = tid + k * NTHREADS;
        const int r = e / CAP_C, col = e - r * CAP_C;
        on[i][k] = e < CAP && r < rh && col < rw;
        if (on[i][k]) {
          const float *src = ptr[i] + (size_t)(ry0[i] + r) * T.w[i] + rx0[i] + col;
#pragma unroll
          for (int c = 0; c < 4; ++c) v[i][k][c] = src[(size_t)c * plane];
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < kMaxTerms; ++i)
#pragma unroll
    for (int k = 0; k < PER; ++k)
      if (on[i][k]) *reinterpret_cast<f32x4t *>(lds + (size_t)(i * CAP + tid + k * NTHREADS) * 4) = v[i][k];
}

// Sum of the terms at full-resolution pixel (X, Y); returns false (fl untouched) when there are none.
template <int CAP_R, int CAP_C>
__device__ __forceinline__ bool terms_flow(const float *lds, const FlowTermsArg &T, const int (&rx0)[kMaxTerms], const int (&ry0)[kMaxTerms],
                                           int X, int Y, float (&fl)[4]) {
  constexpr int CAP = CAP_R * CAP_C;
#pragma unroll
  for (int i = 0; i < kMaxTerms; ++i) {
    if (i < T.n) {
      const Lerp a = lerp_src(Y, T.inv[i], T.h[i]), b = lerp_src(X, T.inv[i], T.w[i]);
      const int o00 = min(a.i0 - ry0[i], CAP_R - 1) * CAP_C + min(b.i0 - rx0[i], CAP_C - 1);
      const int o01 = min(a.i0 - ry0[i], CAP_R - 1) * CAP_C + min(b.i1 - rx0[i], CAP_C - 1);
      const int o10 = min(a.i1 - ry0[i], CAP_R - 1) * CAP_C + min(b.i0 - rx0[i], CAP_C - 1);
      const int o11 = min(a.i1 - ry0[i], CAP_R - 1) * CAP_C + min(b.i1 - rx0[i], CAP_C - 1);
      const float *p = lds + (size_t)i * CAP * 4;
      const f32x4t q00 = *reinterpret_cast<const f32x4t *>(p + o00 * 4), q01 = *reinterpret_cast<const f32x4t *>(p + o01 * 4);
      const f32x4t q10 = *reinterpret_cast<const f32x4t *>(p + o10 * 4), q11 = *reinterpret_cast<const f32x4t *>(p + o11 * 4);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float term = __fmul_rn(lerp2_fma(a.w0, a.w1, b.w0, b.w1, q00[c], q01[c], q10[c], q11[c]), T.scale[i]);
        fl[c] = i == 0 ? term : __fadd_rn(fl[c], term);
      }
    }
  }
  return T.n > 0;
}

}  // namespace drba
