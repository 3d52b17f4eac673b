// Fused shifted-window single-head attention of GMFlow's transformer on the fp32 matrix cores
// (reference: models/gmflow/transformer.py:46-113 single_head_split_window_attention, :19-43 for the shift mask).
//
// The reference rolls q/k/v, splits them into K x K windows, forms the L x L score matrix of every window with a
// GEMM, adds the -100 region mask, takes a row softmax, multiplies by v, merges the windows and rolls back.  Here one
// kernel does all of it and the score matrix never exists in memory:
//
//   * roll / split / merge are an index map (token t of window (wy, wx) <-> pixel ((y+sh) % h, (x+sw) % w)), applied
//     when q/k/v rows are read and when the output row is written;
//   * a workgroup owns 64 query rows of one window (16 per wave) and streams the window's keys / values through LDS
//     in chunks of 64, with the next chunk's global loads in flight under the current chunk's MFMAs;
//   * scores are produced TRANSPOSED (S^T = K Q^T) so that the accumulator layout of mfma_f32_16x16x4f32 -- lane
//     (n = lane % 16, rows 4*(lane/16)+i) -- leaves each lane with 4 keys of ONE query row: the row statistics of the
//     online softmax are per-lane scalars (two xor-shuffles to combine the 4 lane groups), and the probabilities are
//     already in B-operand position for O^T += V^T P^T, whose accumulator again has the query row in lane % 16, so
//     the running rescale is a per-lane multiply;
//   * the region mask is computed from the rolled coordinates (ids 0..8), not read from an [nwin, L, L] table.
//
// Per call at GMFSS_UNION 1080p (fine scale: 128 windows x 540 tokens x 128 channels) the reference formulation moves
// 2 x 149 MB of scores through HBM three times; this kernel reads q, k, v once and writes the output once.
#include "common.hpp"

using namespace drba;

namespace drba_attn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kC = 128;         // channels (GMFlow feature_channels; one head)
constexpr int kRows = 64;       // query rows per workgroup (16 per wave)
constexpr int kKeys = 64;       // keys per chunk
constexpr int kStride = kC + 4; // LDS row stride in floats: 16-byte aligned, bank = 4*key + ... conflict-free for both reads
constexpr int kLdsBytes = (2 * kKeys * kStride + kKeys) * 4;  // 67.8 KB: two workgroups per CU

struct Geometry {
  int h, w, splits, wh, ww, sh, sw, L, shift;
};

// rolled-image position of token t of window win -> source row in the [b, h*w, C] arrays, and its mask region
__device__ __forceinline__ size_t token_row(const Geometry &g, int win, int t, int &region) {
  const int per = g.splits * g.splits;
  const int bi = win / per, wi = win - bi * per;
  const int wy = wi / g.splits, wx = wi - wy * g.splits;
  const int ly = t / g.ww, lx = t - ly * g.ww;
  const int y = wy * g.wh + ly, x = wx * g.ww + lx;
  region = 0;
  int sy = y, sx = x;
  if (g.shift) {
    region = 3 * (y < g.h - g.wh ? 0 : (y < g.h - g.sh ? 1 : 2)) + (x < g.w - g.ww ? 0 : (x < g.w - g.sw ? 1 : 2));
    sy = y + g.sh;
    if (sy >= g.h) sy -= g.h;
    sx = x + g.sw;
    if (sx >= g.w) sx -= g.w;
  }
  return ((size_t)bi * g.h + sy) * g.w + sx;
}

__global__ void __launch_bounds__(256)
window_attention_kernel(const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
                        float *__restrict__ out, Geometry g, int nwin, int qtiles, float scale) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) float lds[];  // kLdsBytes: keys, values, key regions
  float *Ks = lds, *Vs = lds + kKeys * kStride;
  int *Kreg = reinterpret_cast<int *>(lds + 2 * kKeys * kStride);

  // all query tiles of a window on one XCD (workgroups are dealt round-robin over the 8 XCDs): the window's k / v
  // are then fetched into one L2 instead of eight
  const int lin = blockIdx.x;
  const int xcd = lin & 7, slot = lin >> 3;
  const int win = (slot / qtiles) * 8 + xcd, qt = slot - (slot / qtiles) * qtiles;
  if (win >= nwin) return;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, grp = lane >> 4;

  // ---- this lane's query row: B operand of S^T = K Q^T, channel 16j + 4*grp + i for step (j, i)
  const int qtok = qt * kRows + wave * 16 + n16;
  const bool qlive = qtok < g.L;
  int qreg;
  const size_t qrow = token_row(g, win, qlive ? qtok : g.L - 1, qreg);
  f32x4 qf[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) qf[j] = *reinterpret_cast<const f32x4 *>(q + qrow * kC + 16 * j + 4 * grp);

  // ---- chunk loader: thread -> (key = tid/32 + 8*it, 4 channels at 4*(tid%32))
  const int lkey = tid >> 5, lc4 = (tid & 31) * 4;
  f32x4 pk[8], pv[8];
  int preg[8];
  auto fetch = [&](int chunk) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int t = chunk * kKeys + lkey + 8 * it;
      const size_t row = token_row(g, win, t < g.L ? t : g.L - 1, preg[it]);
      pk[it] = *reinterpret_cast<const f32x4 *>(k + row * kC + lc4);
      pv[it] = *reinterpret_cast<const f32x4 *>(v + row * kC + lc4);
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int key = lkey + 8 * it;
      *reinterpret_cast<f32x4 *>(&Ks[key * kStride + lc4]) = pk[it];
      *reinterpret_cast<f32x4 *>(&Vs[key * kStride + lc4]) = pv[it];
      if ((tid & 31) == 0) Kreg[key] = preg[it];
    }
  };

  f32x4 o[8];  // O^T tiles: o[dt][i] = O[q = n16][channel 16*dt + 4*grp + i]
#pragma unroll
  for (int dt = 0; dt < 8; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  const int chunks = (g.L + kKeys - 1) / kKeys;
  fetch(0);
  for (int ch = 0; ch < chunks; ++ch) {
    __syncthreads();  // every wave is done reading the previous chunk
    stage();
    __syncthreads();
    if (ch + 1 < chunks) fetch(ch + 1);

    // ---- S^T tiles: s[t][i] = <K[key = 16t + 4*grp + i], Q[q = n16]>
    f32x4 s[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4 kf = *reinterpret_cast<const f32x4 *>(&Ks[(16 * t + n16) * kStride + 16 * j + 4 * grp]);
#pragma unroll
        for (int i = 0; i < 4; ++i) s[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[i], qf[j][i], s[t], 0, 0, 0);
      }
    }

    // ---- scale, mask, online softmax
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int4 kr = *reinterpret_cast<const int4 *>(&Kreg[16 * t + 4 * grp]);
      const int krs[4] = {kr.x, kr.y, kr.z, kr.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float x = s[t][i] / scale;  // the reference divides (transformer.py:91)
        if (g.shift && krs[i] != qreg) x += -100.f;
        if (ch * kKeys + 16 * t + 4 * grp + i >= g.L) x = -INFINITY;
        s[t][i] = x;
        mx = fmaxf(mx, x);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __expf(m_run - m_new);  // 0 on the first chunk
    float ls = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s[t][i] = __expf(s[t][i] - m_new);
        ls += s[t][i];
      }
    ls += __shfl_xor(ls, 16, 64);
    ls += __shfl_xor(ls, 32, 64);
    l_run = l_run * alpha + ls;
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) o[dt] *= alpha;

    // ---- O^T += V^T P^T: A = V^T (row = channel 16*dt + n16, k = key 16t + 4*grp + i), B = P^T (this lane's s[t][i])
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float *vrow = &Vs[(16 * t + 4 * grp + i) * kStride + n16];
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vrow[16 * dt], s[t][i], o[dt], 0, 0, 0);
      }
    }
  }

  if (qlive) {
    const float inv = 1.f / l_run;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) *reinterpret_cast<f32x4 *>(out + qrow * kC + 16 * dt + 4 * grp) = o[dt] * inv;
  }
#endif
}

}  // namespace drba_attn

extern "C" int drba_window_attention(const float *q, const float *k, const float *v, float *out, int B, int H, int W, int C,
                                     int splits, int shift, float scale, void *stream) {
  if (!q || !k || !v || !out || B <= 0 || H <= 0 || W <= 0 || splits <= 0 || !(scale > 0.f)) return DRBA_EINVAL;
  if (C != drba_attn::kC) return DRBA_EUNSUPPORTED;  // GMFlow's feature_channels
  if (H % splits || W % splits) return DRBA_EINVAL;  // the reference's window split needs whole windows
  drba_attn::Geometry g;
  g.h = H, g.w = W, g.splits = splits, g.wh = H / splits, g.ww = W / splits;
  g.sh = g.wh / 2, g.sw = g.ww / 2, g.L = g.wh * g.ww, g.shift = shift ? 1 : 0;
  const int nwin = B * splits * splits;
  const int qtiles = (g.L + drba_attn::kRows - 1) / drba_attn::kRows;
  const int groups = (nwin + 7) / 8;
  static const hipError_t lds_ok =  // beyond the default 64 KB dynamic-LDS limit
      hipFuncSetAttribute(reinterpret_cast<const void *>(drba_attn::window_attention_kernel),
                          hipFuncAttributeMaxDynamicSharedMemorySize, drba_attn::kLdsBytes);
  if (lds_ok != hipSuccess) return DRBA_ELAUNCH;
  DRBA_LAUNCH_TIMED(drba_attn::window_attention_kernel, dim3((unsigned)(groups * 8 * qtiles)), dim3(kBlock), drba_attn::kLdsBytes,
                    (hipStream_t)stream, q, k, v, out, g, nwin, qtiles, scale);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}
