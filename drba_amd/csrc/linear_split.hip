// nn.Linear (no transposition of the activations: x [M, K] token-major, w [N, K], out [M, N]) for GMFlow's transformer
// (transformer.py:142-208: q/k/v/merge projections and the 256 -> 1024 -> 128 MLP with GELU) on the bf16 matrix
// cores with fp32-level accuracy: both operands are split into three bf16 terms and the six leading partial products
// are accumulated in fp32, exactly as in conv_split.hip (see there for the error argument and the measured MFMA rates).
//
// GEMM view per workgroup (4 waves): MT x 16 tokens by 4 x NTW x 16 output features (64 x 256 for wide layers, 128 x 128
// for the 128-feature ones), K in chunks of 32; 96 MFMAs per wave per chunk and barrier.
//   MFMA A operand = weights (row = output feature), B operand = activations (column = token): the accumulator then
//   holds 4 consecutive output features of ONE token per lane -> a float4 store into the token-major output.
//   Activations: thread (token, channel group of 8) loads 32 contiguous bytes of its row, splits them and writes the
//     three planes to LDS ([plane][group][token][8 x bf16], double-buffered: one barrier per chunk); every wave reads all
//     token tiles of the chunk (3 16-byte reads each).
//   Weights: split and packed in fragment order on the host; wave w owns NTW consecutive output-feature tiles and
//     fetches their 3 x NTW fragments per chunk straight from L2, one chunk ahead.
//   Epilogue: + bias, optional exact GELU (erf form, as nn.GELU()).
#include "common.hpp"

#include <string.h>

#include <type_traits>

using namespace drba;

namespace drba_linear {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int CK = 32;    // K per chunk = the K of one bf16 MFMA
template <int MT_, int NTW_>
struct LinCfg {
  static constexpr int MT = MT_, NTW = NTW_;   // token tiles per workgroup (shared by the waves), feature tiles per wave
  static constexpr int TM = 16 * MT, TN = 64 * NTW;
  static constexpr int PLANE = 4 * TM + 4;     // 16-byte units per plane: [group 4][token] (+4: planes on different banks)
  static constexpr int BUF = 3 * PLANE;
  static constexpr int SIT = TM * 4 / 256;     // staging items (token, channel group) per thread per chunk
};

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

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f)); }

template <class Cfg, bool GELU>
__global__ void __launch_bounds__(256)
linear_split_kernel(const float *__restrict__ x, const u32x4 *__restrict__ wfrag, const float *__restrict__ bias,
                    float *__restrict__ out, int M, int K, int N, int ldx, int n_ntiles) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int MT = Cfg::MT, NTW = Cfg::NTW, TM = Cfg::TM, TN = Cfg::TN, PLANE = Cfg::PLANE, BUF = Cfg::BUF, SIT = Cfg::SIT;
  __shared__ __attribute__((aligned(16))) u32x4 lds[2 * BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kq = lane >> 4;
  // feature tile innermost: the workgroups sharing a token tile run back to back (its rows stay in L2)
  const int t = xcd_band((int)blockIdx.x, (int)gridDim.x);
  const int bn = t % n_ntiles, bm = t / n_ntiles;
  const int m0 = bm * TM, n0 = bn * TN;
  const int nchunks = K / CK;

  // staging: item = (token, channel group of 8): 32 contiguous bytes of the token's row per chunk; lanes run along tokens
  f32x4 pa[SIT], pb[SIT];
  const float *srow[SIT];
  bool sok[SIT];
#pragma unroll
  for (int it = 0; it < SIT; ++it) {
    const int item = tid + 256 * it, stok = item % TM, sgrp = item / TM;
    sok[it] = m0 + stok < M;
    srow[it] = x + (size_t)(sok[it] ? m0 + stok : 0) * ldx + sgrp * 8;
  }
  auto fetch = [&](int q) {
#pragma unroll
    for (int it = 0; it < SIT; ++it) {
      pa[it] = sok[it] ? *reinterpret_cast<const f32x4 *>(srow[it] + q * CK) : (f32x4){0.f, 0.f, 0.f, 0.f};
      pb[it] = sok[it] ? *reinterpret_cast<const f32x4 *>(srow[it] + q * CK + 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int it = 0; it < SIT; ++it) {
      u32x4 h, mm, l;
      const float v[8] = {pa[it][0], pa[it][1], pa[it][2], pa[it][3], pb[it][0], pb[it][1], pb[it][2], pb[it][3]};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        unsigned hh, hm, hl;
        split2(v[2 * i], v[2 * i + 1], hh, hm, hl);
        h[i] = hh, mm[i] = hm, l[i] = hl;
      }
      const int item = tid + 256 * it;
      u32x4 *b = lds + buf * BUF + (item / TM) * TM + item % TM;
      b[0] = h, b[PLANE] = mm, b[2 * PLANE] = l;
    }
  };

  // weight fragments of this wave: [feature tile][chunk][plane][lane]
  const int ftile0 = (n0 >> 4) + wave * NTW;
  const int n_ftiles = (N + 15) >> 4;
  u32x4 wf[2][NTW][3];
  auto wfetch = [&](int q, int slot) {
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int ft = min(ftile0 + nt, n_ftiles - 1);  // tiles past N: any valid fragment, their results are not stored
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) wf[slot][nt][pl] = wfrag[(((size_t)ft * nchunks + q) * 3 + pl) * 64 + lane];
    }
  };

  f32x4 acc[MT][NTW];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  fetch(0);
  wfetch(0, 0);
  stage(0);
  if (nchunks > 1) fetch(1);
  lds_barrier();
  // one chunk; `cur` (LDS buffer and weight-register slot) is a compile-time constant -- indexed with q & 1 the
  // fragment registers become a scratch array
  auto chunk = [&](int q, auto curc) {
    constexpr int cur = decltype(curc)::value;
    if (q + 1 < nchunks) wfetch(q + 1, cur ^ 1);
    const u32x4 *tb = lds + cur * BUF + kq * TM + n16;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const bf16x8 xh = __builtin_bit_cast(bf16x8, tb[mt * 16]);
      const bf16x8 xm = __builtin_bit_cast(bf16x8, tb[PLANE + mt * 16]);
      const bf16x8 xl = __builtin_bit_cast(bf16x8, tb[2 * PLANE + mt * 16]);
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const bf16x8 wh = __builtin_bit_cast(bf16x8, wf[cur][nt][0]);
        const bf16x8 wm = __builtin_bit_cast(bf16x8, wf[cur][nt][1]);
        const bf16x8 wl = __builtin_bit_cast(bf16x8, wf[cur][nt][2]);
        f32x4 c = acc[mt][nt];
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, c, 0, 0, 0);  // smallest terms first
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xm, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xm, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, c, 0, 0, 0);
        acc[mt][nt] = c;
      }
    }
    if (q + 1 < nchunks) {
      stage(cur ^ 1);  // the other buffer: last read in the previous chunk, before the previous barrier
      if (q + 2 < nchunks) fetch(q + 2);
      lds_barrier();
    }
  };
  for (int q = 0; q < nchunks; q += 2) {
    chunk(q, std::integral_constant<int, 0>{});
    if (q + 1 < nchunks) chunk(q + 1, std::integral_constant<int, 1>{});
  }

  // accumulator: lane (token n16 of tile mt, feature rows 4*kq..+3 of tile nt) -> one float4 of the token's output row
  const bool vec = (N & 3) == 0;
  f32x4 bv[NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    const int f = (ftile0 + nt) * 16 + kq * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[nt][i] = (bias && f + i < N) ? bias[f + i] : 0.f;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int tok = m0 + mt * 16 + n16;
    if (tok >= M) continue;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int f = (ftile0 + nt) * 16 + kq * 4;
      if (f >= N) continue;
      f32x4 v = acc[mt][nt] + bv[nt];
      if (GELU) {
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

static inline float bf16_round(float x) {
  unsigned u;
  memcpy(&u, &x, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  u &= 0xffff0000u;
  float r;
  memcpy(&r, &u, 4);
  return r;
}
static inline unsigned short bf16_bits(float exact) {
  unsigned u;
  memcpy(&u, &exact, 4);
  return (unsigned short)(u >> 16);
}

}  // namespace drba_linear

extern "C" {

size_t drba_linear_split_packed_floats(int K, int N) {
  if (K <= 0 || N <= 0 || K % drba_linear::CK) return 0;
  return (size_t)((N + 15) / 16) * (K / drba_linear::CK) * 3 * 64 * 4;
}

// packed (16-byte units): [feature tile][chunk][plane h/m/l][lane] = 8 bf16 of w[16*tile + (lane & 15)][32*chunk + 8*(lane >> 4) + i]
int drba_linear_split_pack(const float *w, float *packed, int K, int N) {
  using namespace drba_linear;
  if (!w || !packed || drba_linear_split_packed_floats(K, N) == 0) return DRBA_EINVAL;
  memset(packed, 0, sizeof(float) * drba_linear_split_packed_floats(K, N));
  unsigned short *dst = reinterpret_cast<unsigned short *>(packed);
  const int nft = (N + 15) / 16, nch = K / CK;
  for (int ft = 0; ft < nft; ++ft)
    for (int q = 0; q < nch; ++q)
      for (int lane = 0; lane < 64; ++lane) {
        const int n = ft * 16 + (lane & 15);
        if (n >= N) continue;
        for (int i = 0; i < 8; ++i) {
          const float x = w[(size_t)n * K + q * CK + 8 * (lane >> 4) + i];
          const float h = bf16_round(x), m = bf16_round(x - h), l = bf16_round(x - h - m);
          const float term[3] = {h, m, l};
          for (int pl = 0; pl < 3; ++pl) dst[(((((size_t)ft * nch + q) * 3 + pl) * 64) + lane) * 8 + i] = bf16_bits(term[pl]);
        }
      }
  return DRBA_OK;
}

int drba_linear_split(const float *x, const float *packed_w, const float *bias, float *out, int M, int K, int N, int ldx,
                      int gelu, void *stream) {
  using namespace drba_linear;
  if (!x || !packed_w || !out || M <= 0 || K <= 0 || N <= 0) return DRBA_EINVAL;
  if (K % CK) return DRBA_EUNSUPPORTED;
  if (ldx < K || (ldx & 3)) return DRBA_EINVAL;  // rows are read as 16-byte vectors
  const u32x4 *wf = reinterpret_cast<const u32x4 *>(packed_w);
  auto go = [&](auto cfg) {
    using Cfg = decltype(cfg);
    const int n_ntiles = (N + Cfg::TN - 1) / Cfg::TN, n_mtiles = (M + Cfg::TM - 1) / Cfg::TM;
    const dim3 grid((unsigned)(n_ntiles * n_mtiles));
    if (gelu) DRBA_LAUNCH_TIMED((linear_split_kernel<Cfg, true>), grid, dim3(kBlock), 0, (hipStream_t)stream, x, wf, bias, out, M, K, N, ldx, n_ntiles);
    else DRBA_LAUNCH_TIMED((linear_split_kernel<Cfg, false>), grid, dim3(kBlock), 0, (hipStream_t)stream, x, wf, bias, out, M, K, N, ldx, n_ntiles);
  };
  if (N > 128) go(LinCfg<4, 4>{});  // 64 tokens x 256 features
  else go(LinCfg<8, 2>{});          // 128 tokens x 128 features
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

}  // extern "C"
