// IFNet stage input fused into the first convolution of the block (scale-1 stage of the 1080p path):
//   conv0.0( cat(warp(img0), warp(img1), warp(f0), warp(f1), timestep, mask, feat, flow) )      IFNet_HDv3.py:146-156, :85-91
// The 52-channel full-resolution stage input (435 MB at 1088x1920) is never written: a workgroup builds the window
// its output tile needs, four virtual channels at a time, straight into the LDS tile the MFMAs read.  The gather
// arithmetic is the same as ifblock_input's (same helpers, same order), the implicit GEMM the same as conv_mfma's
// stride-2 configuration (16x16x4 fp32 MFMA, chunk of 4 channels, taps in row-major order).
#include "common.hpp"

using namespace drba;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_ptr;

namespace drba_stage {

constexpr int TH = 4, TW = 32;                  // output tile (stride-2 conv): one row per wave, 32 columns
constexpr int TR = (TH - 1) * 2 + 3, TC = (TW - 1) * 2 + 3;   // 9 x 65 full-resolution window
constexpr int WPX = TR * TC;                    // 585 window pixels
[[maybe_unused]] constexpr int PPT = (WPX + 255) / 256;  // window pixels per thread (3)
constexpr int CHS = WPX | 1;                    // odd channel stride: conflict-free stride-2 A reads
constexpr int CK = 4, NCHUNK = 13;   // 52 virtual channels: 3+3 images, 16+16 features, timestep, mask + feat 9, flow 4

template <int NT>
__global__ void __launch_bounds__(256)
stage_conv_s1(const float *__restrict__ img0, const float *__restrict__ img1, const float *__restrict__ f0,
              const float *__restrict__ f1, const float *__restrict__ tmap, float tscalar,
              const float *__restrict__ flow, const float *__restrict__ tmp_prev, int hp, int wp, float inv_prev_scale,
              const float *__restrict__ wfrag, const float *__restrict__ bias, float *__restrict__ out, int H, int W,
              int Cout, int Ho, int Wo) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int FRAG = 9 * NT * 64;  // packed weight floats per (cout tile, chunk): [tap][nt][64 lanes]
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NW4 = NCHUNK * FRAG / 4;         // 16-byte units of this cout tile's fragments (all 13 chunks)
  constexpr int WL = ((NW4 + 255) / 256) * 1024; // LDS floats reserved for them: whole 256-lane loads, the tail lanes write zeros
  float *wl = smem;
  float *buf0 = smem + WL, *buf1 = buf0 + CK * CHS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kq = lane >> 4;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, cz = blockIdx.z;
  const size_t P = (size_t)H * W, p_prev = (size_t)hp * wp;
  const int gy0 = y0 * 2 - 1, gx0 = x0 * 2 - 1;

  // weights of this cout tile -> LDS once (16-byte LDS-direct loads; lanes past the end read zeros / padding)
  {
    const float *wsrc = wfrag + (size_t)cz * NCHUNK * FRAG;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)wsrc, 0, NCHUNK * FRAG * 4, 0x00020000);
#pragma unroll
    for (int i = 0; i < (NW4 + 255) / 256; ++i)
      if (i * 256 + wave * 64 < NW4)  // wave-uniform: skip loads that lie wholly past the block
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr)(wl + (i * 256 + wave * 64) * 4), 16,
                                                 (unsigned)(i * 256 + tid) * 16u, 0, 0, 0);
  }

  // per-thread window pixels: warp taps of both directions, computed once and reused by all 38 warped channels
  bool ok[PPT];
  size_t qpix[PPT];
  Taps t0[PPT], t1[PPT];
  int lofs[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int e = tid + i * 256;
    const int r = e / TC, col = e - r * TC;
    const int gy = gy0 + r, gx = gx0 + col;
    ok[i] = e < WPX && gy >= 0 && gy < H && gx >= 0 && gx < W;
    lofs[i] = e < WPX ? e : -1;
    const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
    qpix[i] = (size_t)cy * W + cx;
    const float fl0 = flow[qpix[i]], fl1 = flow[P + qpix[i]], fl2 = flow[2 * P + qpix[i]], fl3 = flow[3 * P + qpix[i]];
    t0[i] = taps_border(warp_coord(cx, W, fl0), warp_coord(cy, H, fl1), W, H);
    t1[i] = taps_border(warp_coord(cx, W, fl2), warp_coord(cy, H, fl3), W, H);
  }
  // F.interpolate at scale 1 (align_corners=False) is the identity with weights (1, 0): kept in the reference's form
  auto ident = [](float v) -> float { return 1.f * (1.f * v + 0.f * 0.f) + 0.f * 0.f; };

  // value of virtual channel c (compile-time) at this thread's i-th window pixel
  auto vch = [&](auto cc, int i) -> float {
    constexpr int c = decltype(cc)::value;
    if constexpr (c < 3) return ident(sample(img0 + (size_t)c * P, W, t0[i]));
    else if constexpr (c < 6) return ident(sample(img1 + (size_t)(c - 3) * P, W, t1[i]));
    else if constexpr (c < 22) return ident(sample(f0 + (size_t)(c - 6) * P, W, t0[i]));
    else if constexpr (c < 38) return ident(sample(f1 + (size_t)(c - 22) * P, W, t1[i]));
    else if constexpr (c == 38) return ident(tmap ? tmap[qpix[i]] : tscalar);
    else if constexpr (c < 48) {  // mask (tmp[4]) and feat (tmp[5:13]): x prev_scale upsample of the previous head output
      const int gy = (int)(qpix[i] / W), gx = (int)(qpix[i] - (size_t)gy * W);
      const Lerp a = lerp_src(gy, inv_prev_scale, hp), b = lerp_src(gx, inv_prev_scale, wp);
      const float *tp = tmp_prev + (size_t)(4 + c - 39) * p_prev;
      const float top = b.w0 * tp[(size_t)a.i0 * wp + b.i0] + b.w1 * tp[(size_t)a.i0 * wp + b.i1];
      const float bot = b.w0 * tp[(size_t)a.i1 * wp + b.i0] + b.w1 * tp[(size_t)a.i1 * wp + b.i1];
      return ident(a.w0 * top + a.w1 * bot);
    } else {
      return (ident(flow[(size_t)(c - 48) * P + qpix[i]]) * 1.f) / 1.f;  // interpolate(flow) * 1. / scale, scale == 1
    }
  };
  auto build = [&](auto qq, float *buf) {  // chunk q: 4 virtual channels of the window -> LDS
    constexpr int q = decltype(qq)::value;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      if (lofs[i] < 0) continue;
      float v[CK];
      v[0] = ok[i] ? vch(std::integral_constant<int, q * CK + 0>{}, i) : 0.f;
      v[1] = ok[i] ? vch(std::integral_constant<int, q * CK + 1>{}, i) : 0.f;
      v[2] = ok[i] ? vch(std::integral_constant<int, q * CK + 2>{}, i) : 0.f;
      v[3] = ok[i] ? vch(std::integral_constant<int, q * CK + 3>{}, i) : 0.f;
#pragma unroll
      for (int c = 0; c < CK; ++c) buf[c * CHS + lofs[i]] = v[c];
    }
  };

  f32x4 acc[2][NT];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int c = 0; c < NT; ++c) acc[b][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int a_off = kq * CHS + (wave * 2) * TC + m * 2;

  auto mfma_chunk = [&](int q, const float *cur) {
    const float *ab = cur + a_off;
    const float *wb = wl + q * FRAG + lane;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const float *at = ab + (tap / 3) * TC + (tap % 3);
      float bv[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bv[nt] = wb[(tap * NT + nt) * 64];
#pragma unroll
      for (int mw = 0; mw < 2; ++mw) {
        const float av = at[mw * 32];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mw][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[nt], acc[mw][nt], 0, 0, 0);
      }
    }
  };

  build(std::integral_constant<int, 0>{}, buf0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // weight fragments have landed
  __syncthreads();
  auto step = [&](auto qq) {  // MFMAs of chunk q on one buffer, chunk q+1 built into the other, one barrier
    constexpr int q = decltype(qq)::value;
    float *cur = (q & 1) ? buf1 : buf0, *nxt = (q & 1) ? buf0 : buf1;
    if constexpr (q + 1 < NCHUNK) build(std::integral_constant<int, q + 1>{}, nxt);
    mfma_chunk(q, cur);
    __syncthreads();
  };
  step(std::integral_constant<int, 0>{});
  step(std::integral_constant<int, 1>{});
  step(std::integral_constant<int, 2>{});
  step(std::integral_constant<int, 3>{});
  step(std::integral_constant<int, 4>{});
  step(std::integral_constant<int, 5>{});
  step(std::integral_constant<int, 6>{});
  step(std::integral_constant<int, 7>{});
  step(std::integral_constant<int, 8>{});
  step(std::integral_constant<int, 9>{});
  step(std::integral_constant<int, 10>{});
  step(std::integral_constant<int, 11>{});
  step(std::integral_constant<int, 12>{});

  // epilogue: bias + LeakyReLU(0.2); D gives each lane 4 consecutive x for one cout
  const int y = y0 + wave;
  if (y < Ho) {
#pragma unroll
    for (int mw = 0; mw < 2; ++mw)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int co = cz * (16 * NT) + nt * 16 + m;
        const int xb = x0 + mw * 16 + kq * 4;
        if (co >= Cout || xb >= Wo) continue;
        const float bs = bias ? bias[co] : 0.f;
        const size_t idx = ((size_t)co * Ho + y) * Wo + xb;
        f32x4 v = acc[mw][nt];
        if ((Wo & 3) == 0) {
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = lrelu02(v[k] + bs);
          *reinterpret_cast<f32x4 *>(out + idx) = v;
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (xb + k < Wo) out[idx + k] = lrelu02(v[k] + bs);
        }
      }
  }
#endif
}

}  // namespace drba_stage

extern "C" {

// conv0.0 (3x3, stride 2, pad 1, LeakyReLU 0.2) of an IFBlock applied to the scale-1 stage input built on the fly.
// packed_w: drba_conv3x3_pack output of configuration `pack_cfg` = 8 (stride 2, 4-channel chunks, 16 cout per tile).
// out: [Cout, (H+1)/2, (W+1)/2].  52 input channels (the stage has a running flow); scale-1 stage only.
int drba_stage_conv0_s1(const float *img0, const float *img1, const float *f0, const float *f1, const float *timestep_map,
                        float timestep_scalar, const float *flow, const float *tmp_prev, int hp, int wp, float prev_scale,
                        const float *packed_w, const float *bias, float *out, int H, int W, int Cout, int pack_cfg,
                        void *stream) {
  if (!img0 || !img1 || !f0 || !f1 || !flow || !tmp_prev || !packed_w || !out) return DRBA_EINVAL;
  if (H <= 1 || W <= 1 || hp <= 0 || wp <= 0 || Cout <= 0 || !(prev_scale > 0.f)) return DRBA_EINVAL;
  if (pack_cfg != 8) return DRBA_EUNSUPPORTED;
  using namespace drba_stage;
  constexpr int NT = 1;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int n_ct = (Cout + 16 * NT - 1) / (16 * NT);
  dim3 g((Wo + TW - 1) / TW, (Ho + TH - 1) / TH, n_ct);
  constexpr int NW4 = NCHUNK * 9 * NT * 64 / 4;
  const size_t lds = ((size_t)((NW4 + 255) / 256) * 1024 + 2 * CK * CHS) * sizeof(float);
  const float ips = (float)(1.0 / (double)prev_scale);
  hipLaunchKernelGGL(stage_conv_s1<NT>, g, dim3(256), lds, (hipStream_t)stream, img0, img1, f0, f1, timestep_map,
                     timestep_scalar, flow, tmp_prev, hp, wp, ips, packed_w, bias, out, H, W, Cout, Ho, Wo);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

}  // extern "C"
