// nn.Linear (no transposition of the activations: x [M, K] token-major, w [N, K], out [M, N]) for GMFlow's transformer
// (transformer.py:142-208: q/k/v/merge projections and the 256 -> 1024 -> 128 MLP with GELU) on the bf16 matrix
// cores with fp32-level accuracy: both operands are split into three bf16 terms and the six leading partial products
// are accumulated in fp32, exactly as in conv_split.hip (see there for the error argument and the measured MFMA rates).
//
// GEMM view per workgroup (4 waves): 64 tokens x 128 output features, K in chunks of 32.
//   MFMA A operand = weights (row = output feature), B operand = activations (column = token): the accumulator then
//   holds 4 consecutive output features of ONE token per lane -> a float4 store into the token-major output.
//   The waves split the TOKENS (16 each) and each computes all 8 feature tiles for its own, so an activation fragment
//   never crosses waves: lane (token, channel group) loads 32 contiguous bytes of its row, splits them and holds the
//   three bf16 operands in registers -- no LDS round trip for the activations.
//   The weight fragments of the chunk (8 x 3 KB, split on the host and packed in fragment order) are copied
//   global -> LDS by LDS-direct loads (no registers, no VALU), double-buffered, one s_waitcnt vmcnt(0) + barrier per
//   chunk; each is read once per wave and feeds 6 MFMAs.  108 registers: 4 workgroups per CU -- measured, occupancy
//   matters more here than blocking: 64 tokens per wave (372 registers) 88 TFLOP/s, 32 per wave 132, 16 per wave 146
//   on the 1024 -> 128 layer (fp32-equivalent; rocBLAS fp32: 110).  An earlier form that shared the split activations
//   through LDS and kept the weights in registers reached 116.
//   Epilogue: + bias, then optionally the exact GELU (erf form, as nn.GELU()) or -- for the 128-feature layers, where
//   a wave holds whole output rows -- LayerNorm(128) with its affine and the residual add.
#include "common.hpp"
#include "conv_split.hpp"

#include <string.h>

#include <type_traits>

using namespace drba;

namespace drba_linear {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int CK = 32;    // K per chunk = the K of one bf16 MFMA
__device__ __forceinline__ void split2(float a, float b, unsigned &h, unsigned &m, unsigned &l) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  auto pk = [](float x, float y) -> unsigned {
    const bf16x2 p = __builtin_convertvector(f32x2{x, y}, bf16x2);
    return __builtin_bit_cast(unsigned, p);
  };
  h = pk(a, b);
  const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
  m = pk(ra, rb);
  l = pk(ra - __uint_as_float(m << 16), rb - __uint_as_float(m & 0xffff0000u));
}

// two-term fp16 form (conv_split.hip "Two-term form"): x * 2^-shift = h + 2^-11 l
__device__ __forceinline__ void split2_f16(float a, float b, unsigned &h, unsigned &l) {
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 v = (f32x2){a, b} * (1.f / (float)(1 << drba::kSplitActShift));
  const f16x2 hh = __builtin_convertvector(v, f16x2);
  const f32x2 r = (v - __builtin_convertvector(hh, f32x2)) * 2048.f;
  const f16x2 ll = __builtin_convertvector(r, f16x2);
  h = __builtin_bit_cast(unsigned, hh);
  l = __builtin_bit_cast(unsigned, ll);
}

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f)); }

typedef __attribute__((address_space(3))) void *lds_ptr;

template <int MTW_, int NT_, int PL_ = 3>  // PL: 16-bit terms per operand (3 bf16 / 2 fp16)
struct LinCfg {
  static constexpr int MTW = MTW_, NT = NT_, PL = PL_;
  static constexpr int TM = 64 * MTW, TN = 16 * NT;
  static constexpr int WBUF = NT * PL * 64;  // 16-byte units of weight fragments per chunk
};

// EPI 0: + bias.  EPI 1: + bias, GELU.  EPI 2 (N == the workgroup's 128 features): + bias, LayerNorm over the row
// (nn.LayerNorm(128): biased variance, eps) * ln_w + ln_b, + residual row -- transformer.py:178-185, :203-207.
template <class Cfg, int EPI>
__global__ void __launch_bounds__(256, Cfg::MTW >= 2 ? 2 : 1)  // the 128-token tile: two workgroups per CU (<= 256 registers)
linear_split_kernel(const float *__restrict__ x, const u32x4 *__restrict__ wfrag, const float *__restrict__ bias,
                     float *__restrict__ out, int M, int K, int N, int ldx, int n_ntiles, const float *__restrict__ ln_w,
                     const float *__restrict__ ln_b, const float *__restrict__ residual, float eps,
                     const float *__restrict__ x2, int ldx2, int q_split, unsigned char *status) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int MTW = Cfg::MTW, NT = Cfg::NT, TM = Cfg::TM, WBUF = Cfg::WBUF, PL = Cfg::PL;
  __shared__ __attribute__((aligned(16))) u32x4 wl[2 * WBUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kq = lane >> 4;
  const int t = xcd_band((int)blockIdx.x, (int)gridDim.x);
  const int bn = t % n_ntiles, bm = t / n_ntiles;
  const int m0 = bm * TM + wave * (16 * MTW), ft0 = bn * NT;
  const int nchunks = K / CK, n_ftiles = (N + 15) >> 4;

  // weights: fragment (nt, plane) of chunk q lives at 16-byte unit ((ft0 + nt) * nchunks + q) * PL * 64 + plane * 64 + lane
  const __amdgpu_buffer_rsrc_t wrs =
      __builtin_amdgcn_make_buffer_rsrc((void *)wfrag, 0, n_ftiles * nchunks * PL * 1024, 0x00020000);
  auto wissue = [&](int q, int buf) {  // NT * PL one-KB fragments, round-robin over the 4 waves
#pragma unroll
    for (int i = 0; i < (NT * PL + 3) / 4; ++i) {
      const int f = wave + 4 * i;  // fragment index nt * PL + plane (wave-uniform)
      if (f < NT * PL) {
        const int nt = f / PL, pl = f - nt * PL;
        const int ft = min(ft0 + nt, n_ftiles - 1);  // tiles past N: any valid fragment, their results are not stored
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr)(wl + buf * WBUF + f * 64), 16, (unsigned)lane * 16u,
                                                 (unsigned)(((ft * nchunks + q) * PL + pl) * 1024), 0, 0);
      }
    }
  };

  // activations: lane (token n16 of tile mt, channel group kq) <- 32 bytes of its row per chunk
  // chunks [0, q_split) come from x, the rest from x2 (the MLP's input cat(source, message) without the copy)
  const float *xrow[MTW], *xrow2[MTW];
  bool xok[MTW];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) {
    const int tok = m0 + mt * 16 + n16;
    xok[mt] = tok < M;
    xrow[mt] = x + (size_t)(xok[mt] ? tok : 0) * ldx + kq * 8;
    xrow2[mt] = x2 ? x2 + (size_t)(xok[mt] ? tok : 0) * ldx2 + kq * 8 - (size_t)q_split * CK : xrow[mt];
  }
  f32x4 ra[MTW], rb[MTW];
  auto xfetch = [&](int q) {
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
      const float *src = (q < q_split ? xrow[mt] : xrow2[mt]) + q * CK;
      ra[mt] = xok[mt] ? *reinterpret_cast<const f32x4 *>(src) : (f32x4){0.f, 0.f, 0.f, 0.f};
      rb[mt] = xok[mt] ? *reinterpret_cast<const f32x4 *>(src + 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };

  f32x4 acc[MTW][NT], acl[MTW][PL == 2 ? NT : 1];  // acl (PL = 2): the h*l + l*h products, weight 2^-11
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if constexpr (PL == 2) acl[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

  xfetch(0);
  wissue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  auto chunk = [&](int q, auto curc) {
    constexpr int cur = decltype(curc)::value;
    // split this chunk's activations, then put the next chunk's loads in flight under the MFMAs
    u32x4 xh[MTW], xm[MTW], xl[MTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
      u32x4 h, mm = (u32x4){0u, 0u, 0u, 0u}, l;
      const float v[8] = {ra[mt][0], ra[mt][1], ra[mt][2], ra[mt][3], rb[mt][0], rb[mt][1], rb[mt][2], rb[mt][3]};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        unsigned hh, hm = 0u, hl;
        if constexpr (PL == 3) split2(v[2 * i], v[2 * i + 1], hh, hm, hl);
        else split2_f16(v[2 * i], v[2 * i + 1], hh, hl);
        h[i] = hh, mm[i] = hm, l[i] = hl;
      }
      xh[mt] = h, xm[mt] = mm, xl[mt] = l;
    }
    if (q + 1 < nchunks) {
      xfetch(q + 1);
      wissue(q + 1, cur ^ 1);  // that buffer was last read in the previous chunk, before the previous barrier
    }
    const u32x4 *wb = wl + cur * WBUF + lane;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if constexpr (PL == 3) {
        const bf16x8 wh = __builtin_bit_cast(bf16x8, wb[(nt * 3 + 0) * 64]);
        const bf16x8 wm = __builtin_bit_cast(bf16x8, wb[(nt * 3 + 1) * 64]);
        const bf16x8 wlo = __builtin_bit_cast(bf16x8, wb[(nt * 3 + 2) * 64]);
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
          const bf16x8 ah = __builtin_bit_cast(bf16x8, xh[mt]), am = __builtin_bit_cast(bf16x8, xm[mt]), al = __builtin_bit_cast(bf16x8, xl[mt]);
          f32x4 c = acc[mt][nt];
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, ah, c, 0, 0, 0);  // smallest terms first
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, al, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, am, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, ah, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, am, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ah, c, 0, 0, 0);
          acc[mt][nt] = c;
        }
      } else {
        const f16x8 wh = __builtin_bit_cast(f16x8, wb[(nt * 2 + 0) * 64]), wlo = __builtin_bit_cast(f16x8, wb[(nt * 2 + 1) * 64]);
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
          const f16x8 ah = __builtin_bit_cast(f16x8, xh[mt]), al = __builtin_bit_cast(f16x8, xl[mt]);
          f32x4 c = acl[mt][nt];
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo, ah, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, al, c, 0, 0, 0);
          acl[mt][nt] = c;
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, ah, acc[mt][nt], 0, 0, 0);
        }
      }
    }
    if (q + 1 < nchunks) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next chunk's weights are in LDS (and its activations in registers)
      __builtin_amdgcn_s_barrier();
    }
  };
  for (int q = 0; q < nchunks; q += 2) {
    chunk(q, std::integral_constant<int, 0>{});
    if (q + 1 < nchunks) chunk(q + 1, std::integral_constant<int, 1>{});
  }

  if constexpr (PL == 2) {  // join the two sums, undo the activation pre-scale (exact powers of two)
    float nf = 0.f;
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        acc[mt][nt] = (acc[mt][nt] + acl[mt][nt] * (1.f / 2048.f)) * (float)(1 << drba::kSplitActShift);
#pragma unroll
        for (int i = 0; i < 4; ++i) nf = drba::nf_fold(nf, acc[mt][nt][i]);  // the family's overflow report (common.hpp)
      }
    drba::nf_report(status, DRBA_STATUS_LINEAR, nf);
  }
  if constexpr (EPI == 2) {
    // a lane holds 32 of its token's 128 outputs (features 16*nt + 4*kq + i); the other 96 sit in the lanes with the
    // same token and the other three kq -> two xor-shuffles complete a row sum
    auto row_sum = [](float v) {
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      return v;
    };
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
      const int tok = m0 + mt * 16 + n16;
      float sum = 0.f;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int f = nt * 16 + kq * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (bias) acc[mt][nt][i] += bias[f + i];
          sum += acc[mt][nt][i];
        }
      }
      const float mean = row_sum(sum) * (1.f / 128.f);
      float var = 0.f;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float d = acc[mt][nt][i] - mean;
          var += d * d;
        }
      const float inv = 1.f / sqrtf(row_sum(var) * (1.f / 128.f) + eps);
      if (tok >= M) continue;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int f = nt * 16 + kq * 4;
        const f32x4 gw = *reinterpret_cast<const f32x4 *>(ln_w + f), gb = *reinterpret_cast<const f32x4 *>(ln_b + f);
        f32x4 v = (acc[mt][nt] - mean) * inv * gw + gb;
        if (residual) v = *reinterpret_cast<const f32x4 *>(residual + (size_t)tok * 128 + f) + v;
        *reinterpret_cast<f32x4 *>(out + (size_t)tok * 128 + f) = v;
      }
    }
    return;
  }
  const bool vec = (N & 3) == 0;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int f = (ft0 + nt) * 16 + kq * 4;
    if (f >= N) continue;
    f32x4 bv;
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[i] = (bias && f + i < N) ? bias[f + i] : 0.f;
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
      const int tok = m0 + mt * 16 + n16;
      if (tok >= M) continue;
      f32x4 v = acc[mt][nt] + bv;
      if (EPI == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = gelu_erf(v[i]);
      }
      float *dst = out + (size_t)tok * N + f;
      if (vec && f + 3 < N) *reinterpret_cast<f32x4 *>(dst) = v;
      else
        for (int i = 0; i < 4; ++i)
          if (f + i < N) dst[i] = v[i];
    }
  }
#endif
}

}  // namespace drba_linear

extern "C" {

size_t drba_linear_split_packed_floats(int K, int N, int terms) {
  if (K <= 0 || N <= 0 || K % drba_linear::CK || (terms != 2 && terms != 3)) return 0;
  return (size_t)((N + 15) / 16) * (K / drba_linear::CK) * terms * 64 * 4;
}

// packed (16-byte units): [feature tile][chunk][plane h/m/l or h/l][lane] = 8 x 16 bit (split_weight_terms) of
// w[16*tile + (lane & 15)][32*chunk + 8*(lane >> 4) + i]
int drba_linear_split_pack(const float *w, float *packed, int K, int N, int terms) {
  using namespace drba_linear;
  if (!w || !packed || drba_linear_split_packed_floats(K, N, terms) == 0) return DRBA_EINVAL;
  if (terms == 2 && !drba::two_term_weights_ok(w, (size_t)N * K)) return DRBA_EUNSUPPORTED;
  memset(packed, 0, sizeof(float) * drba_linear_split_packed_floats(K, N, terms));
  unsigned short *dst = reinterpret_cast<unsigned short *>(packed);
  const int nft = (N + 15) / 16, nch = K / CK;
  for (int ft = 0; ft < nft; ++ft)
    for (int q = 0; q < nch; ++q)
      for (int lane = 0; lane < 64; ++lane) {
        const int n = ft * 16 + (lane & 15);
        if (n >= N) continue;
        for (int i = 0; i < 8; ++i) {
          unsigned short term[3];
          drba::split_weight_terms(w[(size_t)n * K + q * CK + 8 * (lane >> 4) + i], terms, term);
          for (int pl = 0; pl < terms; ++pl) dst[(((((size_t)ft * nch + q) * terms + pl) * 64) + lane) * 8 + i] = term[pl];
        }
      }
  return DRBA_OK;
}

static int linear_launch(const float *x, const float *packed_w, const float *bias, float *out, int M, int K, int N, int ldx,
                         int epi, const float *ln_w, const float *ln_b, const float *residual, float eps, int terms, void *stream,
                         const float *x2 = nullptr, int ldx2 = 0, int K1 = 0) {
  using namespace drba_linear;
  if (!x || !packed_w || !out || M <= 0 || K <= 0 || N <= 0 || (terms != 2 && terms != 3)) return DRBA_EINVAL;
  if (K % CK) return DRBA_EUNSUPPORTED;
  if ((ldx & 3) || ldx < (x2 ? K1 : K)) return DRBA_EINVAL;  // rows are read as 16-byte vectors
  if (x2 && (K1 <= 0 || K1 >= K || K1 % CK || (ldx2 & 3) || ldx2 < K - K1)) return DRBA_EINVAL;
  const int q_split = x2 ? K1 / CK : K / CK;
  const u32x4 *wf = reinterpret_cast<const u32x4 *>(packed_w);
  // 64 tokens x 128 features per workgroup; 128 tokens (two 16-token tiles per wave) for the two-term form on large token counts:
  // a chunk's 16 KB of weight fragments and its barrier + global-load wait are then paid once per 48 MFMAs of a wave instead of 24
  // (same-box A/B, 69 120 tokens: 256 -> 1024 + GELU 312 -> 289 us; with one feature tile -- the 128-feature LayerNorm epilogue --
  // the wider tile halves the workgroups and loses, 27.7 -> 34 us: taken only where >= 1536 workgroups remain)
  static const int wide_min = env_int("DRBA_LIN_WIDE_MIN", 1536);  // (TUNING builds only)
  const bool wide = terms == 2 && (long long)((M + 127) / 128) * ((N + 127) / 128) >= wide_min;
  const int TM = wide ? 128 : 64, TN = 128;
  const int n_ntiles = (N + TN - 1) / TN, n_mtiles = (M + TM - 1) / TM;
  const dim3 grid((unsigned)(n_ntiles * n_mtiles));
  unsigned char *status = terms == 2 ? drba::status_bytes() : nullptr;
#define DRBA_LIN(E, P, MT)                                                                                                   \
  DRBA_LAUNCH((linear_split_kernel<LinCfg<MT, 8, P>, E>), grid, dim3(kBlock), 0, (hipStream_t)stream, x, wf, bias, out, M, K, N, \
                    ldx, n_ntiles, ln_w, ln_b, residual, eps, x2, ldx2, q_split, status)
  if (terms == 3) {
    if (epi == 2) DRBA_LIN(2, 3, 1);
    else if (epi == 1) DRBA_LIN(1, 3, 1);
    else DRBA_LIN(0, 3, 1);
  } else if (wide) {
    if (epi == 2) DRBA_LIN(2, 2, 2);
    else if (epi == 1) DRBA_LIN(1, 2, 2);
    else DRBA_LIN(0, 2, 2);
  } else {
    if (epi == 2) DRBA_LIN(2, 2, 1);
    else if (epi == 1) DRBA_LIN(1, 2, 1);
    else DRBA_LIN(0, 2, 1);
  }
#undef DRBA_LIN
  DRBA_CHECK_LAUNCH();
  return terms == 2 ? range_checked(DRBA_OK, out, (size_t)M * N, stream) : DRBA_OK;
}

int drba_linear_split(const float *x, const float *packed_w, const float *bias, float *out, int M, int K, int N, int ldx,
                      int gelu, int terms, void *stream) {
  return linear_launch(x, packed_w, bias, out, M, K, N, ldx, gelu ? 1 : 0, nullptr, nullptr, nullptr, 0.f, terms, stream);
}

int drba_linear_split_cat(const float *x1, const float *x2, const float *packed_w, const float *bias, float *out, int M,
                          int K1, int K2, int N, int ldx1, int ldx2, int gelu, int terms, void *stream) {
  if (!x2) return DRBA_EINVAL;
  return linear_launch(x1, packed_w, bias, out, M, K1 + K2, N, ldx1, gelu ? 1 : 0, nullptr, nullptr, nullptr, 0.f, terms, stream, x2,
                       ldx2, K1);
}

int drba_linear_split_layernorm(const float *x, const float *packed_w, const float *bias, const float *ln_w,
                                const float *ln_b, const float *residual, float *out, int M, int K, int ldx, float eps,
                                int terms, void *stream) {
  if (!ln_w || !ln_b || !(eps > 0.f)) return DRBA_EINVAL;
  return linear_launch(x, packed_w, bias, out, M, K, 128, ldx, 2, ln_w, ln_b, residual, eps, terms, stream);
}

}  // extern "C"
