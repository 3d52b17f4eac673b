// Global correlation softmax -> expected value, without the L x L score matrix
// (reference: models/gmflow/matching.py:7-38 global_correlation_softmax and models/gmflow/transformer.py:355-372, the
// global branch of FeatureFlowAttention).
//
// Both are out[i] = sum_j softmax_j(<Q_i, K_j> / sqrt(C)) * val_j with a 2-component value per key: the key's pixel
// coordinate (minus the query's own coordinate afterwards: the flow), or the flow to propagate.  The reference forms the
// L x L matrix with a GEMM (8640^2 floats = 298 MB per direction at 1080p, written, read by the softmax, read again by
// the second GEMM); round 1 did the same through the vendor BLAS plus a softmax kernel.  Here a workgroup owns 64 query
// rows (16 per wave), streams the keys through LDS in chunks of 64 with the next chunk's loads in flight, forms the
// scores TRANSPOSED on the fp32 matrix cores (S^T = K Q^T: a lane ends up with 16 keys of ONE query row, so the online
// softmax statistics are per-lane scalars) and accumulates the two weighted sums with plain FMAs: the scores never
// leave registers.  HBM traffic is Q + K once per query tile from L2 (4.4 MB each at 1080p).
//
// The keys are split into `ksplit` runs (separate workgroups) so that a launch has >= 2 workgroups per CU
// (8640 queries are only 135 tiles); global_expect2_merge combines the runs' (max, sum, weighted sums).
#include "common.hpp"

using namespace drba;

namespace drba_gcorr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kC = 128;          // GMFlow feature_channels
constexpr int kRows = 64;        // query rows per workgroup
constexpr int kKeys = 64;        // keys per chunk
constexpr int kStride = kC + 4;  // LDS row stride (floats): conflict-free 16-byte fragment reads
constexpr int kLdsBytes = (kKeys * kStride + 2 * kKeys) * 4;

__global__ void __launch_bounds__(256)
global_expect2_kernel(const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ vals,
                      float *__restrict__ out, float *__restrict__ part, int L, int w, float inv_scale, int ldq, int ldk,
                      int ksplit) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *Ks = lds, *Vx = lds + kKeys * kStride, *Vy = Vx + kKeys;
  constexpr int TPW = 4, LIT = 8;
  const int qt = blockIdx.x / ksplit, ks = blockIdx.x - qt * ksplit;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, grp = lane >> 4;

  // this lane's query row: B operand of S^T = K Q^T, channel 16j + 4*grp + i for step (j, i)
  const int qtok = qt * kRows + wave * 16 + n16;
  const bool qlive = qtok < L;
  const size_t qrow = (size_t)min(qtok, L - 1);
  f32x4 qf[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) qf[j] = *reinterpret_cast<const f32x4 *>(q + qrow * ldq + 16 * j + 4 * grp);

  // chunk loader: thread -> (key = tid/32 + 8*it, 4 channels at 4*(tid%32)); value pair of key tid for tid < 64
  const int lkey = tid >> 5, lc4 = (tid & 31) * 4;
  f32x4 pk[LIT];
  float pvx = 0.f, pvy = 0.f;
  auto fetch = [&](int chunk) {
#pragma unroll
    for (int it = 0; it < LIT; ++it) {
      const size_t row = (size_t)min(chunk * kKeys + lkey + 8 * it, L - 1);
      pk[it] = *reinterpret_cast<const f32x4 *>(k + row * ldk + lc4);
    }
    if (tid < kKeys) {
      const int key = min(chunk * kKeys + tid, L - 1);
      if (vals) {
        pvx = vals[key];
        pvy = vals[L + key];
      } else {  // the key's pixel coordinate (x, y)
        const int y = key / w;
        pvx = (float)(key - y * w);
        pvy = (float)y;
      }
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int it = 0; it < LIT; ++it) *reinterpret_cast<f32x4 *>(&Ks[(lkey + 8 * it) * kStride + lc4]) = pk[it];
    if (tid < kKeys) {
      Vx[tid] = pvx;
      Vy[tid] = pvy;
    }
  };

  float m_run = -INFINITY, l_run = 0.f, ax = 0.f, ay = 0.f;  // l / ax / ay: this lane's 16 keys per chunk only
  const int all_chunks = (L + kKeys - 1) / kKeys, per = (all_chunks + ksplit - 1) / ksplit;
  const int ch0 = ks * per, chunks = min(all_chunks, ch0 + per);
  if (ch0 < chunks) fetch(ch0);
  for (int ch = ch0; ch < chunks; ++ch) {
    __syncthreads();
    stage();
    __syncthreads();
    if (ch + 1 < chunks) fetch(ch + 1);

    // S^T tiles: s[t][i] = <K[key = 16t + 4*grp + i], Q[q = n16]>; K fragments read two steps ahead of their MFMAs
    f32x4 s[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *kbase = &Ks[n16 * kStride + 4 * grp];
    constexpr int QK_STEPS = 8 * TPW;
    auto kfrag = [&](int step) {  // step = j * TPW + t
      return *reinterpret_cast<const f32x4 *>(kbase + 16 * (step % TPW) * kStride + 16 * (step / TPW));
    };
    f32x4 kring[3];
    kring[0] = kfrag(0), kring[1] = kfrag(1);
#pragma unroll
    for (int step = 0; step < QK_STEPS; ++step) {
      if (step + 2 < QK_STEPS) kring[(step + 2) % 3] = kfrag(step + 2);
      const f32x4 kf = kring[step % 3];
      const int j = step / TPW, t = step % TPW;
#pragma unroll
      for (int i = 0; i < 4; ++i) s[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[i], qf[j][i], s[t], 0, 0, 0);
    }

    // scale, online softmax over the chunk (row maximum shared by the 4 lane groups of a query), weighted sums
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float x = s[t][i] * inv_scale;
        if (ch * kKeys + 16 * t + 4 * grp + i >= L) x = -INFINITY;
        s[t][i] = x;
        mx = fmaxf(mx, x);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);  // finite: every chunk holds at least one real key
    const float alpha = __expf(m_run - m_new);
    float ls = 0.f, sx = 0.f, sy = 0.f;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const f32x4 vx = *reinterpret_cast<const f32x4 *>(&Vx[16 * t + 4 * grp]);
      const f32x4 vy = *reinterpret_cast<const f32x4 *>(&Vy[16 * t + 4 * grp]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float e = __expf(s[t][i] - m_new);
        ls += e;
        sx = fmaf(e, vx[i], sx);
        sy = fmaf(e, vy[i], sy);
      }
    }
    l_run = l_run * alpha + ls;
    ax = ax * alpha + sx;
    ay = ay * alpha + sy;
    m_run = m_new;
  }
  // combine the 4 lane groups of a query row (same running maximum)
  l_run += __shfl_xor(l_run, 16, 64), ax += __shfl_xor(ax, 16, 64), ay += __shfl_xor(ay, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64), ax += __shfl_xor(ax, 32, 64), ay += __shfl_xor(ay, 32, 64);
  if (!qlive || grp != 0) return;
  if (ksplit > 1) {
    *reinterpret_cast<f32x4 *>(part + ((size_t)qtok * ksplit + ks) * 4) = f32x4{m_run, l_run, ax, ay};
    return;
  }
  float ox = ax / l_run, oy = ay / l_run;
  if (!vals) {
    const int y = qtok / w;
    ox -= (float)(qtok - y * w);
    oy -= (float)y;
  }
  out[qtok] = ox;
  out[L + qtok] = oy;
#endif
}

__global__ void __launch_bounds__(256)
global_expect2_merge(const float *__restrict__ part, float *__restrict__ out, int L, int w, int ksplit, int coords) {
  const int qtok = blockIdx.x * 256 + threadIdx.x;
  if (qtok >= L) return;
  const f32x4 *p = reinterpret_cast<const f32x4 *>(part) + (size_t)qtok * ksplit;
  float m = -INFINITY;
  for (int s = 0; s < ksplit; ++s) m = fmaxf(m, p[s][0]);
  float l = 0.f, ax = 0.f, ay = 0.f;
  for (int s = 0; s < ksplit; ++s) {
    const f32x4 v = p[s];
    const float wgt = v[0] == -INFINITY ? 0.f : __expf(v[0] - m);  // a run without keys: m = -inf, sums 0
    l += wgt * v[1];
    ax += wgt * v[2];
    ay += wgt * v[3];
  }
  float ox = ax / l, oy = ay / l;
  if (coords) {
    const int y = qtok / w;
    ox -= (float)(qtok - y * w);
    oy -= (float)y;
  }
  out[qtok] = ox;
  out[L + qtok] = oy;
}

// Plain batched C[b] = A[b] * B[b]^T (trans_b) or A[b] * B[b], fp32 FMAs, one output per thread.  Only the degenerate
// shifted-window case of the transformer (a window one pixel wide or high, i.e. frames below 128 pixels, where the
// reference's mask table is ill-formed and reproduced from the host) takes this path; everything else is fused.
__global__ void __launch_bounds__(256)
bmm_naive_kernel(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ c, int M, int N, int K,
                 int trans_b) {
  const size_t per = (size_t)M * N, total = per * gridDim.y;
  (void)total;
  const int bi = blockIdx.y;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= per) return;
  const int m = (int)(i / N), n = (int)(i - (size_t)m * N);
  const float *ar = a + ((size_t)bi * M + m) * K;
  const float *bb = b + (size_t)bi * N * K;
  float acc = 0.f;
  if (trans_b) {
    const float *br = bb + (size_t)n * K;
    for (int kk = 0; kk < K; ++kk) acc = fmaf(ar[kk], br[kk], acc);
  } else {
    for (int kk = 0; kk < K; ++kk) acc = fmaf(ar[kk], bb[(size_t)kk * N + n], acc);
  }
  c[(size_t)bi * per + i] = acc;
}

static int pick_ksplit(int L) {
  const int qtiles = (L + kRows - 1) / kRows, chunks = (L + kKeys - 1) / kKeys;
  static const int force = env_int("DRBA_GCORR_KSPLIT", 0);  // (TUNING builds only)
  if (force > 0) return force;
  // 125 registers and 34 KB of LDS: three workgroups per CU.  The launch ends when the fullest CU does -- (workgroups on it) x
  // (chunks per workgroup) -- so the split is the one that minimises that product among those that stay resident, not the first
  // power of two past 512 workgroups: 8640 tokens = 135 query tiles, 4 runs put 3 workgroups of 34 chunks on 28 CUs (267 us),
  // 5 runs 3 x 27 on 163 (221 us; same-box sweep of 1..12 runs, tools/exp/gcorr_target.py).  + 2: a workgroup's fixed cost
  // (query fragments, first fetch, partial write) in chunk times.
  int best = 1, best_cost = 0x7fffffff;
  for (int ks = 1; ks <= 12 && chunks / ks >= 4; ++ks) {
    const int per_cu = (qtiles * ks + 255) / 256;
    if (per_cu > 3 && ks > 1) break;
    const int cost = per_cu * ((chunks + ks - 1) / ks + 2);
    if (cost < best_cost) best = ks, best_cost = cost;
  }
  return best;
}

}  // namespace drba_gcorr

extern "C" size_t drba_global_expect2_ws_floats(int L) {
  if (L <= 0) return 0;
  const int ks = drba_gcorr::pick_ksplit(L);
  return ks > 1 ? (size_t)L * ks * 4 : 0;
}

extern "C" int drba_global_expect2(const float *q, const float *k, const float *vals, float *out, float *ws, int L, int C,
                                   int w, float scale, int ldq, int ldk, void *stream) {
  if (!q || !k || !out || L <= 0 || w <= 0 || !(scale > 0.f)) return DRBA_EINVAL;
  if (C != drba_gcorr::kC) return DRBA_EUNSUPPORTED;
  if (ldq < C || ldk < C || ((ldq | ldk) & 3)) return DRBA_EINVAL;
  const int ksplit = ws ? drba_gcorr::pick_ksplit(L) : 1;
  const int qtiles = (L + drba_gcorr::kRows - 1) / drba_gcorr::kRows;
  DRBA_LAUNCH(drba_gcorr::global_expect2_kernel, dim3((unsigned)(qtiles * ksplit)), dim3(kBlock), drba_gcorr::kLdsBytes,
              (hipStream_t)stream, q, k, vals, out, ws, L, w, 1.f / scale, ldq, ldk, ksplit);
  if (ksplit > 1)
    DRBA_LAUNCH(drba_gcorr::global_expect2_merge, dim3((unsigned)((L + 255) / 256)), dim3(kBlock), 0, (hipStream_t)stream, ws,
                out, L, w, ksplit, vals ? 0 : 1);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

extern "C" int drba_bmm(const float *a, const float *b, float *c, int batch, int M, int N, int K, int trans_b, void *stream) {
  if (!a || !b || !c || batch <= 0 || M <= 0 || N <= 0 || K <= 0) return DRBA_EINVAL;
  DRBA_LAUNCH(drba_gcorr::bmm_naive_kernel, dim3((unsigned)(((size_t)M * N + 255) / 256), (unsigned)batch), dim3(kBlock), 0,
              (hipStream_t)stream, a, b, c, M, N, K, trans_b);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}
