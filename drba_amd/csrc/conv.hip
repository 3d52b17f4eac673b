// 3x3 convolution (stride 1/2, pad 1) and ConvTranspose2d(4, 2, 1) as fp32 implicit GEMM on
// v_mfma_f32_16x16x4_f32 (exact fp32: bit-equal to a k-ordered fmaf chain; gfx950 has no
// TF32/xf32 path and bf16 inputs break the 1e-3 parity bar, SURVEY.md 'Hard parts').
//
// GEMM view per workgroup (4 waves = one per SIMD):
//   M = output pixels : a TH x TW tile, split into 16-pixel row segments (MFMA rows)
//   N = output channels: NT tiles of 16 (MFMA cols)
//   K = (tap, cin)    : cin consumed 4 at a time (MFMA k = 4), CK channels staged per LDS chunk
// MFMA operand fetch: lane l supplies A[pixel l&15][cin l>>4] and B[cin l>>4][cout l&15] with one
// ds_read_b32 each.  LDS layouts are padded so both reads are bank-conflict free:
//   input tile  [CK][rows][cols], channel stride == 16 (mod 32) for stride 1, odd for stride 2
//   weight slab [tap][CK][NTP],  NTP == 16 (mod 32)
// Accumulator D: lane holds cout l&15 for pixels 4*(l>>4)..+3 -> one float4 store along x.
#include "common.hpp"

#include <string.h>

using namespace drba;

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int round_up_mod32_16(int v) {  // smallest r >= v with r % 32 == 16
  int r = v - (v % 32) + 16;
  return r >= v ? r : r + 32;
}

template <int S_, int RW_, int MW_, int NT_, int CK_>
struct ConvCfg {
  static constexpr int S = S_, RW = RW_, MW = MW_, NT = NT_, CK = CK_;
  static constexpr int TH = 4 * RW, TW = 16 * MW, NTC = 16 * NT;
  static constexpr int TR = (TH - 1) * S + 3, TC = (TW - 1) * S + 3;
  static constexpr int CHS = (S == 1) ? round_up_mod32_16(TR * TC) : ((TR * TC) | 1);
  static constexpr int NTP = (NT % 2) ? 16 * NT : 16 * NT + 16;
  static constexpr int SLAB = 9 * CK * NTP;  // floats per (cout-tile, chunk)
  static constexpr int LDS_FLOATS = ((CK * CHS + 3) / 4) * 4 + SLAB;
  static constexpr int W_OFF = ((CK * CHS + 3) / 4) * 4;
};

template <class Cfg>
__global__ void __launch_bounds__(256)
conv3x3_mfma(const float *__restrict__ in, const float *__restrict__ wpk, const float *__restrict__ bias,
             const float *__restrict__ beta, const float *__restrict__ res, float *__restrict__ out, int Cin,
             int H, int W, int Cout, int Ho, int Wo, int act, int n_ctiles) {
  constexpr int S = Cfg::S, RW = Cfg::RW, MW = Cfg::MW, NT = Cfg::NT, CK = Cfg::CK;
  constexpr int TH = Cfg::TH, TW = Cfg::TW, TR = Cfg::TR, TC = Cfg::TC, CHS = Cfg::CHS, NTP = Cfg::NTP;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *s_in = smem;
  float *s_w = smem + Cfg::W_OFF;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, kq = lane >> 4;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
  const int cz = blockIdx.z % n_ctiles, n = blockIdx.z / n_ctiles;
  in += (size_t)n * Cin * H * W;
  out += (size_t)n * Cout * Ho * Wo;
  if (res) res += (size_t)n * Cout * Ho * Wo;

  f32x4 acc[RW][MW][NT];
#pragma unroll
  for (int a = 0; a < RW; ++a)
#pragma unroll
    for (int b = 0; b < MW; ++b)
#pragma unroll
      for (int c = 0; c < NT; ++c) acc[a][b][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nchunks = (Cin + CK - 1) / CK;
  const float *a_base = s_in + kq * CHS + (wave * RW * S) * TC + m * S;
  const float *b_base = s_w + kq * NTP + m;
  const int gy0 = y0 * S - 1, gx0 = x0 * S - 1;

  for (int q = 0; q < nchunks; ++q) {
    __syncthreads();
    // ---- stage the input tile (zero padding outside the image / beyond Cin)
    for (int e = tid; e < CK * TR * TC; e += 256) {
      const int c = e / (TR * TC), rem = e - c * (TR * TC);
      const int r = rem / TC, col = rem - r * TC;
      const int gy = gy0 + r, gx = gx0 + col, ci = q * CK + c;
      float v = 0.f;
      if (ci < Cin && gy >= 0 && gy < H && gx >= 0 && gx < W) v = in[((size_t)ci * H + gy) * W + gx];
      s_in[c * CHS + r * TC + col] = v;
    }
    // ---- stage this (cout tile, chunk)'s weight slab: contiguous, pre-padded on the host
    {
      const float4 *src = reinterpret_cast<const float4 *>(wpk + ((size_t)cz * nchunks + q) * Cfg::SLAB);
      float4 *dst = reinterpret_cast<float4 *>(s_w);
      for (int e = tid; e < Cfg::SLAB / 4; e += 256) dst[e] = src[e];
    }
    __syncthreads();
    // ---- MFMA over (tap, 4-channel group)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap % 3;
#pragma unroll
      for (int cg = 0; cg < CK / 4; ++cg) {
        float bv[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv[nt] = b_base[(tap * CK + cg * 4) * NTP + nt * 16];
#pragma unroll
        for (int rw = 0; rw < RW; ++rw)
#pragma unroll
          for (int mw = 0; mw < MW; ++mw) {
            const float av = a_base[(cg * 4) * CHS + (rw * S + ky) * TC + mw * 16 * S + kx];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[rw][mw][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[nt], acc[rw][mw][nt], 0, 0, 0);
          }
      }
    }
  }

  // ---- epilogue: bias, optional beta*y + residual, optional LeakyReLU(0.2)
  const bool vec = (Wo & 3) == 0;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co = cz * Cfg::NTC + nt * 16 + m;
    if (co >= Cout) continue;
    const float bs = bias ? bias[co] : 0.f;
    const float bt = beta ? beta[co] : 0.f;
#pragma unroll
    for (int rw = 0; rw < RW; ++rw) {
      const int y = y0 + wave * RW + rw;
      if (y >= Ho) continue;
#pragma unroll
      for (int mw = 0; mw < MW; ++mw) {
        const int xb = x0 + mw * 16 + kq * 4;
        if (xb >= Wo) continue;
        const size_t idx = ((size_t)co * Ho + y) * Wo + xb;
        f32x4 v = acc[rw][mw][nt];
        if (vec) {
          f32x4 r = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (beta) r = *reinterpret_cast<const f32x4 *>(res + idx);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float t = v[k] + bs;
            if (beta) t = t * bt + r[k];
            v[k] = act ? lrelu02(t) : t;
          }
          *reinterpret_cast<f32x4 *>(out + idx) = v;
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (xb + k >= Wo) continue;
            float t = v[k] + bs;
            if (beta) t = t * bt + res[idx + k];
            out[idx + k] = act ? lrelu02(t) : t;
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// ConvTranspose2d(k=4, s=2, p=1): out[o, 2j+py, 2i+px] = b[o] + sum_{c, a, b in {0,1}}
//   in[c, j+dy(py,a), i+dx(px,b)] * W[c, o, ky(py,a), kx(px,b)]
// with (py=0: (ky,dy) = (1,0),(3,-1); py=1: (0,+1),(2,0)), same along x.  Each of the 4 output
// phases is a 2x2-tap convolution over the same haloed input tile; blockIdx.z carries the phase.
template <int RW_, int MW_, int NT_, int CK_>
struct DeconvCfg {
  static constexpr int RW = RW_, MW = MW_, NT = NT_, CK = CK_;
  static constexpr int TH = 4 * RW, TW = 16 * MW, NTC = 16 * NT;
  static constexpr int TR = TH + 2, TC = TW + 2;
  static constexpr int CHS = round_up_mod32_16(TR * TC);
  static constexpr int NTP = (NT % 2) ? 16 * NT : 16 * NT + 16;
  static constexpr int SLAB = 4 * CK * NTP;  // floats per (cout-tile, phase, chunk)
  static constexpr int W_OFF = ((CK * CHS + 3) / 4) * 4;
  static constexpr int LDS_FLOATS = W_OFF + SLAB;
};

template <class Cfg>
__global__ void __launch_bounds__(256)
deconv4x4_mfma(const float *__restrict__ in, const float *__restrict__ wpk, const float *__restrict__ bias,
               float *__restrict__ out, int Cin, int H, int W, int Cout, int pixel_shuffle, int n_ctiles) {
  constexpr int RW = Cfg::RW, MW = Cfg::MW, NT = Cfg::NT, CK = Cfg::CK;
  constexpr int TH = Cfg::TH, TW = Cfg::TW, TR = Cfg::TR, TC = Cfg::TC, CHS = Cfg::CHS, NTP = Cfg::NTP;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *s_in = smem;
  float *s_w = smem + Cfg::W_OFF;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, kq = lane >> 4;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
  const int phase = blockIdx.z & 3, cz = (blockIdx.z >> 2) % n_ctiles, n = (blockIdx.z >> 2) / n_ctiles;
  const int py = phase >> 1, px = phase & 1;
  in += (size_t)n * Cin * H * W;
  out += (size_t)n * Cout * (2 * H) * (2 * W);

  f32x4 acc[RW][MW][NT];
#pragma unroll
  for (int a = 0; a < RW; ++a)
#pragma unroll
    for (int b = 0; b < MW; ++b)
#pragma unroll
      for (int c = 0; c < NT; ++c) acc[a][b][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nchunks = (Cin + CK - 1) / CK;
  // tile row of tap a: (row + 1 + dy); py=0 -> dy = {0,-1}; py=1 -> dy = {+1, 0}
  const int ry0 = py ? 2 : 1, ry1 = py ? 1 : 0;
  const int rx0 = px ? 2 : 1, rx1 = px ? 1 : 0;
  const float *a_base = s_in + kq * CHS + (wave * RW) * TC + m;
  const float *b_base = s_w + kq * NTP + m;

  for (int q = 0; q < nchunks; ++q) {
    __syncthreads();
    for (int e = tid; e < CK * TR * TC; e += 256) {
      const int c = e / (TR * TC), rem = e - c * (TR * TC);
      const int r = rem / TC, col = rem - r * TC;
      const int gy = y0 - 1 + r, gx = x0 - 1 + col, ci = q * CK + c;
      float v = 0.f;
      if (ci < Cin && gy >= 0 && gy < H && gx >= 0 && gx < W) v = in[((size_t)ci * H + gy) * W + gx];
      s_in[c * CHS + r * TC + col] = v;
    }
    {
      const float4 *src =
          reinterpret_cast<const float4 *>(wpk + (((size_t)cz * 4 + phase) * nchunks + q) * Cfg::SLAB);
      float4 *dst = reinterpret_cast<float4 *>(s_w);
      for (int e = tid; e < Cfg::SLAB / 4; e += 256) dst[e] = src[e];
    }
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 4; ++tap) {
      const int ro = (tap >> 1) ? ry1 : ry0, cof = (tap & 1) ? rx1 : rx0;
      const float *a_tap = a_base + ro * TC + cof;
#pragma unroll
      for (int cg = 0; cg < CK / 4; ++cg) {
        float bv[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv[nt] = b_base[(tap * CK + cg * 4) * NTP + nt * 16];
#pragma unroll
        for (int rw = 0; rw < RW; ++rw)
#pragma unroll
          for (int mw = 0; mw < MW; ++mw) {
            const float av = a_tap[(cg * 4) * CHS + rw * TC + mw * 16];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[rw][mw][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[nt], acc[rw][mw][nt], 0, 0, 0);
          }
      }
    }
  }

  const int Ho = 2 * H, Wo = 2 * W;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co = cz * Cfg::NTC + nt * 16 + m;
    if (co >= Cout) continue;
    const float bs = bias ? bias[co] : 0.f;
#pragma unroll
    for (int rw = 0; rw < RW; ++rw) {
      const int j = y0 + wave * RW + rw;
      if (j >= H) continue;
#pragma unroll
      for (int mw = 0; mw < MW; ++mw)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = x0 + mw * 16 + kq * 4 + k;
          if (i >= W) continue;
          const float v = acc[rw][mw][nt][k] + bs;
          const int oy = 2 * j + py, ox = 2 * i + px;
          if (pixel_shuffle) {  // PixelShuffle(2): [Cout/4, 2*Ho, 2*Wo]
            const int c13 = co >> 2, si = (co >> 1) & 1, sj = co & 1;
            out[((size_t)c13 * (2 * Ho) + (2 * oy + si)) * (2 * Wo) + (2 * ox + sj)] = v;
          } else {
            out[((size_t)co * Ho + oy) * Wo + ox] = v;
          }
        }
    }
  }
}

// ------------------------------------------------------------------------------------------ cfg tables
//                 S  RW MW NT CK
using C0 = ConvCfg<1, 2, 4, 2, 8>;  // 8x64 px, 32 cout   (block4 ResConv)
using C1 = ConvCfg<1, 2, 2, 4, 8>;  // 8x32 px, 64 cout   (block3 ResConv)
using C2 = ConvCfg<1, 1, 2, 6, 8>;  // 4x32 px, 96 cout   (block2 ResConv)
using C3 = ConvCfg<1, 1, 2, 4, 8>;  // 4x32 px, 64 cout   (block0/1 ResConv, cout split over z)
using C4 = ConvCfg<1, 2, 4, 1, 8>;  // 8x64 px, 16 cout   (encode.cnn1/2)
using C5 = ConvCfg<2, 2, 2, 1, 4>;  // 8x32 px, 16 cout   (block4.conv0.0, encode.cnn0)
using C6 = ConvCfg<2, 2, 2, 2, 4>;  // 8x32 px, 32 cout
using C7 = ConvCfg<2, 1, 2, 4, 4>;  // 4x32 px, 64 cout
using C8 = ConvCfg<2, 1, 2, 3, 4>;  // 4x32 px, 48 cout
constexpr int kNumConvCfg = 9;

//                   RW MW NT CK
using D0 = DeconvCfg<2, 2, 4, 8>;  // 8x32 px, 64 cout (52 used)
using D1 = DeconvCfg<2, 4, 1, 8>;  // 8x64 px, 16 cout (encode.cnn3)
using D2 = DeconvCfg<1, 2, 4, 8>;  // 4x32 px, 64 cout (small maps)
constexpr int kNumDeconvCfg = 3;

struct CfgInfo {
  int S, TH, TW, NTC, NTP, CK, SLAB, lds_bytes;
};
template <class C>
constexpr CfgInfo conv_info() {
  return {C::S, C::TH, C::TW, C::NTC, C::NTP, C::CK, C::SLAB, C::LDS_FLOATS * 4};
}
template <class C>
constexpr CfgInfo deconv_info() {
  return {1, C::TH, C::TW, C::NTC, C::NTP, C::CK, C::SLAB, C::LDS_FLOATS * 4};
}
const CfgInfo kConv[kNumConvCfg] = {conv_info<C0>(), conv_info<C1>(), conv_info<C2>(), conv_info<C3>(), conv_info<C4>(),
                                    conv_info<C5>(), conv_info<C6>(), conv_info<C7>(), conv_info<C8>()};
const CfgInfo kDeconv[kNumDeconvCfg] = {deconv_info<D0>(), deconv_info<D1>(), deconv_info<D2>()};

template <class Cfg>
int launch_conv(const float *in, const float *wpk, const float *bias, const float *beta, const float *res, float *out,
                int N, int Cin, int H, int W, int Cout, int Ho, int Wo, int act, hipStream_t s) {
  const int n_ct = (Cout + Cfg::NTC - 1) / Cfg::NTC;
  dim3 g((Wo + Cfg::TW - 1) / Cfg::TW, (Ho + Cfg::TH - 1) / Cfg::TH, N * n_ct);
  hipLaunchKernelGGL(conv3x3_mfma<Cfg>, g, dim3(256), Cfg::LDS_FLOATS * sizeof(float), s, in, wpk, bias, beta, res, out,
                     Cin, H, W, Cout, Ho, Wo, act, n_ct);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

template <class Cfg>
int launch_deconv(const float *in, const float *wpk, const float *bias, float *out, int N, int Cin, int H, int W,
                  int Cout, int ps, hipStream_t s) {
  const int n_ct = (Cout + Cfg::NTC - 1) / Cfg::NTC;
  dim3 g((W + Cfg::TW - 1) / Cfg::TW, (H + Cfg::TH - 1) / Cfg::TH, N * n_ct * 4);
  hipLaunchKernelGGL(deconv4x4_mfma<Cfg>, g, dim3(256), Cfg::LDS_FLOATS * sizeof(float), s, in, wpk, bias, out, Cin, H,
                     W, Cout, ps, n_ct);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

}  // namespace

extern "C" {

int drba_conv3x3_pick_cfg(int Cin, int Cout, int Ho, int Wo, int stride) {
  (void)Cin;
  if (stride == 1) {
    if (Cout <= 16) return 4;
    if (Cout <= 32) return 0;
    if (Cout <= 64) return ((size_t)Ho * Wo >= 8192) ? 1 : 3;
    if (Cout == 96) return 2;
    return 3;
  }
  if (stride == 2) {
    if (Cout <= 16) return 5;
    if (Cout <= 32) return 6;
    if (Cout == 48 || Cout == 96) return 8;
    return 7;
  }
  return DRBA_EUNSUPPORTED;
}

size_t drba_conv3x3_packed_floats(int Cin, int Cout, int cfg) {
  if (cfg < 0 || cfg >= kNumConvCfg || Cin <= 0 || Cout <= 0) return 0;
  const CfgInfo &c = kConv[cfg];
  const size_t n_ct = (Cout + c.NTC - 1) / c.NTC, nch = (Cin + c.CK - 1) / c.CK;
  return n_ct * nch * c.SLAB;
}

// packed[(cz*nchunks + q)][tap][c][NTP] = w[cz*NTC + j][q*CK + c][tap], zero outside
int drba_conv3x3_pack(const float *w, float *packed, int Cin, int Cout, int cfg) {
  if (!w || !packed || cfg < 0 || cfg >= kNumConvCfg || Cin <= 0 || Cout <= 0) return DRBA_EINVAL;
  const CfgInfo &c = kConv[cfg];
  const int n_ct = (Cout + c.NTC - 1) / c.NTC, nch = (Cin + c.CK - 1) / c.CK;
  memset(packed, 0, sizeof(float) * (size_t)n_ct * nch * c.SLAB);
  for (int cz = 0; cz < n_ct; ++cz)
    for (int q = 0; q < nch; ++q) {
      float *slab = packed + ((size_t)cz * nch + q) * c.SLAB;
      for (int tap = 0; tap < 9; ++tap)
        for (int cc = 0; cc < c.CK; ++cc) {
          const int ci = q * c.CK + cc;
          if (ci >= Cin) continue;
          for (int j = 0; j < c.NTC; ++j) {
            const int co = cz * c.NTC + j;
            if (co >= Cout) break;
            slab[(tap * c.CK + cc) * c.NTP + j] = w[((size_t)co * Cin + ci) * 9 + tap];
          }
        }
    }
  return DRBA_OK;
}

int drba_conv3x3(const float *in, const float *packed_w, const float *bias, const float *beta, const float *residual,
                 float *out, int N, int Cin, int H, int W, int Cout, int stride, int act, int cfg, void *stream) {
  if (!in || !packed_w || !out || N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return DRBA_EINVAL;
  if (cfg < 0 || cfg >= kNumConvCfg || kConv[cfg].S != stride) return DRBA_EINVAL;
  if (beta && !residual) return DRBA_EINVAL;
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  hipStream_t s = (hipStream_t)stream;
#define DRBA_CONV_CASE(ID, T) \
  case ID:                    \
    return launch_conv<T>(in, packed_w, bias, beta, residual, out, N, Cin, H, W, Cout, Ho, Wo, act, s);
  switch (cfg) {
    DRBA_CONV_CASE(0, C0)
    DRBA_CONV_CASE(1, C1)
    DRBA_CONV_CASE(2, C2)
    DRBA_CONV_CASE(3, C3)
    DRBA_CONV_CASE(4, C4)
    DRBA_CONV_CASE(5, C5)
    DRBA_CONV_CASE(6, C6)
    DRBA_CONV_CASE(7, C7)
    DRBA_CONV_CASE(8, C8)
  }
#undef DRBA_CONV_CASE
  return DRBA_EUNSUPPORTED;
}

int drba_deconv4x4_pick_cfg(int Cin, int Cout, int H, int W) {
  (void)Cin;
  if (Cout <= 16) return 1;
  return ((size_t)H * W >= 4096) ? 0 : 2;
}

size_t drba_deconv4x4_packed_floats(int Cin, int Cout, int cfg) {
  if (cfg < 0 || cfg >= kNumDeconvCfg || Cin <= 0 || Cout <= 0) return 0;
  const CfgInfo &c = kDeconv[cfg];
  const size_t n_ct = (Cout + c.NTC - 1) / c.NTC, nch = (Cin + c.CK - 1) / c.CK;
  return n_ct * 4 * nch * c.SLAB;
}

// w: [Cin, Cout, 4, 4].  packed[((cz*4 + phase)*nchunks + q)][tap = 2a+b][c][NTP]
int drba_deconv4x4_pack(const float *w, float *packed, int Cin, int Cout, int cfg) {
  if (!w || !packed || cfg < 0 || cfg >= kNumDeconvCfg || Cin <= 0 || Cout <= 0) return DRBA_EINVAL;
  const CfgInfo &c = kDeconv[cfg];
  const int n_ct = (Cout + c.NTC - 1) / c.NTC, nch = (Cin + c.CK - 1) / c.CK;
  memset(packed, 0, sizeof(float) * (size_t)n_ct * 4 * nch * c.SLAB);
  for (int cz = 0; cz < n_ct; ++cz)
    for (int phase = 0; phase < 4; ++phase) {
      const int py = phase >> 1, px = phase & 1;
      for (int q = 0; q < nch; ++q) {
        float *slab = packed + (((size_t)cz * 4 + phase) * nch + q) * c.SLAB;
        for (int tap = 0; tap < 4; ++tap) {
          const int a = tap >> 1, b = tap & 1;
          const int ky = py ? (a ? 2 : 0) : (a ? 3 : 1);
          const int kx = px ? (b ? 2 : 0) : (b ? 3 : 1);
          for (int cc = 0; cc < c.CK; ++cc) {
            const int ci = q * c.CK + cc;
            if (ci >= Cin) continue;
            for (int j = 0; j < c.NTC; ++j) {
              const int co = cz * c.NTC + j;
              if (co >= Cout) break;
              slab[(tap * c.CK + cc) * c.NTP + j] = w[(((size_t)ci * Cout + co) * 4 + ky) * 4 + kx];
            }
          }
        }
      }
    }
  return DRBA_OK;
}

int drba_deconv4x4s2(const float *in, const float *packed_w, const float *bias, float *out, int N, int Cin, int H,
                     int W, int Cout, int pixel_shuffle, int cfg, void *stream) {
  if (!in || !packed_w || !out || N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return DRBA_EINVAL;
  if (cfg < 0 || cfg >= kNumDeconvCfg) return DRBA_EINVAL;
  if (pixel_shuffle && (Cout & 3)) return DRBA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  switch (cfg) {
    case 0: return launch_deconv<D0>(in, packed_w, bias, out, N, Cin, H, W, Cout, pixel_shuffle, s);
    case 1: return launch_deconv<D1>(in, packed_w, bias, out, N, Cin, H, W, Cout, pixel_shuffle, s);
    case 2: return launch_deconv<D2>(in, packed_w, bias, out, N, Cin, H, W, Cout, pixel_shuffle, s);
  }
  return DRBA_EUNSUPPORTED;
}

}  // extern "C"
