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
//
// Measured (tools/attn_bench.py, MI355X): fine scale 230 us (83 TFLOP/s fp32) against 560-780 us for BLAS QK^T +
// softmax kernel + BLAS PV + the roll / split copies; coarse scale (8 windows x 2160 tokens) 348 us against 335-440 us.
// In-kernel cycle counters put QK^T at 34 cycles per MFMA (32 is the issue rate) and PV at 43; the coarse launch is
// limited by its shape -- 272 workgroups on 256 CUs, so 16 CUs carry two -- not by the inner loops: such launches split
// each window's keys into 4 runs (separate workgroups, merged by window_attention_merge): 260 us.  Splitting each
// chunk's keys over two waves (8-wave workgroups) was tried to raise the waves per SIMD there and was slower (365 us).
#include <stdlib.h>

#include <algorithm>

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
  unsigned ww_magic;  // ceil(2^32 / ww): t / ww == umulhi(t, ww_magic) for t, ww < 2^16
};

// rolled-image position of token t of window (bi, wy, wx) -> source row in the [b, h*w, C] arrays, and its mask region
struct Window {
  int bi, y0, x0;
};
__device__ __forceinline__ Window window_of(const Geometry &g, int win) {
  const int per = g.splits * g.splits;
  const int bi = win / per, wi = win - bi * per;
  const int wy = wi / g.splits, wx = wi - wy * g.splits;
  return Window{bi, wy * g.wh, wx * g.ww};
}
__device__ __forceinline__ unsigned token_row(const Geometry &g, const Window &wd, int t, int &region) {
  const int ly = g.ww == 1 ? t : (int)__umulhi((unsigned)t, g.ww_magic), lx = t - ly * g.ww;  // ceil(2^32/1) overflows
  const int y = wd.y0 + ly, x = wd.x0 + lx;
  region = 0;
  int sy = y, sx = x;
  if (g.shift) {
    region = 3 * ((y >= g.h - g.wh) + (y >= g.h - g.sh)) + (x >= g.w - g.ww) + (x >= g.w - g.sw);
    sy = y + g.sh;
    sy -= sy >= g.h ? g.h : 0;
    sx = x + g.sw;
    sx -= sx >= g.w ? g.w : 0;
  }
  return (unsigned)((wd.bi * g.h + sy) * g.w + sx);
}

__global__ void __launch_bounds__(256)
window_attention_kernel(const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
                        float *__restrict__ out, Geometry g, int nwin, int qtiles, float scale, int ldq, int ldk, int ldv, int ksplit,
                        float *__restrict__ part) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) float lds[];  // kLdsBytes: keys, values, key regions
  float *Ks = lds, *Vs = lds + kKeys * kStride;
  int *Kreg = reinterpret_cast<int *>(lds + 2 * kKeys * kStride);
  constexpr int TPW = 4;  // key tiles of a chunk
  constexpr int LIT = 8;  // loader iterations per thread

  // all query tiles of a window on one XCD (workgroups are dealt round-robin over the 8 XCDs): the window's k / v
  // are then fetched into one L2 instead of eight
  const int lin = blockIdx.x;
  const int xcd = lin & 7;
  int slot = lin >> 3;
  // key split (launches with fewer than two workgroups per CU, e.g. GMFlow's coarse scale: 8 windows x 34 query tiles):
  // workgroup (window, query tile, ks) walks one contiguous run of the window's key chunks and leaves
  // its running (max, sum, unnormalised O) in `part`; window_attention_merge combines the runs
  const int ks = slot % ksplit;
  slot /= ksplit;
  const int win = (slot / qtiles) * 8 + xcd, qt = slot - (slot / qtiles) * qtiles;
  if (win >= nwin) return;
  const Window wd = window_of(g, win);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, grp = lane >> 4;

  // ---- this lane's query row: B operand of S^T = K Q^T, channel 16j + 4*grp + i for step (j, i)
  const int qtok = qt * kRows + wave * 16 + n16;
  const bool qlive = qtok < g.L;
  int qreg;
  const size_t qrow = token_row(g, wd, min(qtok, g.L - 1), qreg);
  f32x4 qf[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) qf[j] = *reinterpret_cast<const f32x4 *>(q + qrow * ldq + 16 * j + 4 * grp);

  // ---- chunk loader: thread -> (key = tid/32 + 8*it, 4 channels at 4*(tid%32))
  const int lkey = tid >> 5, lc4 = (tid & 31) * 4;
  f32x4 pk[LIT], pv[LIT];
  auto fetch = [&](int chunk) {
#pragma unroll
    for (int it = 0; it < LIT; ++it) {
      int unused;
      const size_t row = token_row(g, wd, min(chunk * kKeys + lkey + 8 * it, g.L - 1), unused);
      pk[it] = *reinterpret_cast<const f32x4 *>(k + row * ldk + lc4);
      pv[it] = *reinterpret_cast<const f32x4 *>(v + row * ldv + lc4);
    }
  };
  auto stage = [&](int chunk) {
#pragma unroll
    for (int it = 0; it < LIT; ++it) {
      const int key = lkey + 8 * it;
      *reinterpret_cast<f32x4 *>(&Ks[key * kStride + lc4]) = pk[it];
      *reinterpret_cast<f32x4 *>(&Vs[key * kStride + lc4]) = pv[it];
    }
    if (g.shift && tid < kKeys) {
      int region;
      token_row(g, wd, min(chunk * kKeys + tid, g.L - 1), region);
      Kreg[tid] = region;
    }
  };

  // O^T tiles.  Tile dt < 4 holds channels 4*m + dt in its row m, tile dt >= 4 channels 64 + 4*m + (dt - 4): a lane's
  // A operands for the 8 tiles are then two 16-byte LDS reads of one value row, and o[dt][i] = O[q = n16][channel
  // 16*grp + 4*i + dt (+64)], i.e. 2 x 16 contiguous output channels per lane
  f32x4 o[8];
#pragma unroll
  for (int dt = 0; dt < 8; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  const int all_chunks = (g.L + kKeys - 1) / kKeys, per = (all_chunks + ksplit - 1) / ksplit;
  const int ch0 = ks * per, chunks = min(all_chunks, ch0 + per);  // this workgroup's chunks: [ch0, chunks)
  const float inv_scale = 1.f / scale;
  if (ch0 < chunks) fetch(ch0);
  for (int ch = ch0; ch < chunks; ++ch) {
    __syncthreads();  // every wave is done reading the previous chunk
    stage(ch);
    __syncthreads();
    if (ch + 1 < chunks) fetch(ch + 1);

    // ---- S^T tiles: s[t][i] = <K[key = 16*t + 4*grp + i], Q[q = n16]>.  One 16-byte K fragment feeds 4 MFMAs;
    // fragments are read two steps ahead of their use so the LDS latency sits under the MFMAs of the steps before
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

    // first value fragments of the PV product: in flight under the softmax arithmetic
    const float *vbase = &Vs[4 * grp * kStride + 4 * n16];
    auto vfrag = [&](int step, int half) {  // step = t * 4 + i
      return *reinterpret_cast<const f32x4 *>(vbase + (16 * (step / 4) + (step % 4)) * kStride + 64 * half);
    };
    f32x4 vring[2][2];
    vring[0][0] = vfrag(0, 0), vring[0][1] = vfrag(0, 1);

    // ---- scale, mask, online softmax
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      int krs[4] = {0, 0, 0, 0};
      if (g.shift) {
        const int4 kr = *reinterpret_cast<const int4 *>(&Kreg[16 * t + 4 * grp]);
        krs[0] = kr.x, krs[1] = kr.y, krs[2] = kr.z, krs[3] = kr.w;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float x = s[t][i] * inv_scale;  // scores / sqrt(C) (transformer.py:91)
        if (g.shift && krs[i] != qreg) x += -100.f;
        if (ch * kKeys + 16 * t + 4 * grp + i >= g.L) x = -INFINITY;
        s[t][i] = x;
        mx = fmaxf(mx, x);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);       // finite: every chunk holds at least one real key
    const float alpha = __expf(m_run - m_new);  // 0 on the first chunk
    float ls = 0.f;
#pragma unroll
    for (int t = 0; t < TPW; ++t)
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

    // ---- O^T += V^T P^T: A = V^T (row m = n16 of tile dt, k = key 16t + 4*grp + i), B = P^T (this lane's s[t][i])
    constexpr int PV_STEPS = 4 * TPW;
#pragma unroll
    for (int step = 0; step < PV_STEPS; ++step) {
      if (step + 1 < PV_STEPS) vring[(step + 1) & 1][0] = vfrag(step + 1, 0), vring[(step + 1) & 1][1] = vfrag(step + 1, 1);
      const float p = s[step / 4][step % 4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vring[step & 1][0][dt], p, o[dt], 0, 0, 0);
        o[4 + dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vring[step & 1][1][dt], p, o[4 + dt], 0, 0, 0);
      }
    }
  }

  if (ksplit > 1) {
    // partial state of query row (window, token) for key run ks: [m, l, pad, pad, O[128]] (132 floats)
    if (qlive) {
      float *prow = part + (((size_t)win * g.L + qtok) * ksplit + ks) * 132;
      if (grp == 0) {
        prow[0] = m_run;
        prow[1] = l_run;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<f32x4 *>(prow + 4 + 16 * grp + 4 * i) = f32x4{o[0][i], o[1][i], o[2][i], o[3][i]};
        *reinterpret_cast<f32x4 *>(prow + 4 + 64 + 16 * grp + 4 * i) = f32x4{o[4][i], o[5][i], o[6][i], o[7][i]};
      }
    }
    return;
  }
  if (qlive) {
    const float inv = 1.f / l_run;
    float *orow = out + qrow * kC + 16 * grp;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<f32x4 *>(orow + 4 * i) = f32x4{o[0][i], o[1][i], o[2][i], o[3][i]} * inv;
      *reinterpret_cast<f32x4 *>(orow + 64 + 4 * i) = f32x4{o[4][i], o[5][i], o[6][i], o[7][i]} * inv;
    }
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// The same kernel in the two-term fp16 form (kernel family 4; conv_split.hip "Two-term form"): every fp32 operand of the two
// GEMMs is taken as h + 2^-11 l with two fp16 terms and contracted with v_mfma_f32_16x16x32_f16 (three products per fragment
// pair, the cross products in a second accumulator), 48 + 48 MFMAs of 16 clocks per 64-key chunk and wave instead of
// 128 + 128 fp32 ones of 32.  Scores, masks, the online softmax and the running rescale stay fp32 as above.
//   * K of a chunk lives in LDS as two fp16 planes [key][16 segments of 16 bytes]; a lane's A operand for the 32-channel step j
//     of key tile t is one 16-byte read.  K is pre-scaled by 2^-4 (undone with the 1 / sqrt(C) factor), Q is split once into
//     registers;
//   * the MFMA's K index is free as long as both operands agree, and so is the key a score row stands for: score row n of
//     tile t is key 16 t + n, lane group g leaves the softmax with keys 16 t + 4 g + i of tiles 2 ss, 2 ss + 1 as its 8 k-values
//     of PV step ss -- so V goes to LDS TRANSPOSED as [channel][unit U = 4 ss + g][8 x fp16] with element e of unit U standing
//     for key 32 (U >> 2) + 16 (e >> 2) + 4 (U & 3) + (e & 3), and a loader thread holds exactly those keys of 4 channels: 16-byte
//     writes, no 2-byte scatter;
//   * O^T tiles hold channels 16 dt + 4 g + i: 16-byte stores of consecutive channels.
// LDS banks (tools/exp/attn/lds_banks.py evaluates the guide's lane-group rules): segment s of key k sits at s ^ (k & 15), unit u
// of channel c at u ^ ((c >> 1) & 7) ^ ((c >> 4) & 1): every read and write below is conflict-free.  (Rows padded by 16 bytes --
// the form up to round 5 -- cost 2 x on both 16-byte reads, whose lane groups are not 16 consecutive lanes, and 4 x on the V^T
// writes: SQ_LDS_BANK_CONFLICT was half of SQ_LDS_IDX_ACTIVE, the LDS busy 42 % of the launch.)
// WAVES = 8: 128 query rows share a staged chunk -- a loader thread converts 4 keys instead of 8, half the LDS writes and L2 reads per
// query row -- as ONE workgroup per CU at the same two waves per SIMD (256 registers; three or four waves per SIMD do not fit: the
// O^T accumulators (64), the split Q (32) and the chunk in flight leave the S / PV working set nothing under 168 / 128 registers --
// 138 / 131 spilled, 2 x slower).  8 windows x 2160 tokens 150 -> 126 us, 128 windows x 540 140 -> 134 (same box).  With the CU's LDS to
// itself that workgroup keeps two chunk images (128.5 KB): chunk c + 1 is converted and written while chunk c is worked on, one
// barrier per chunk instead of two -- 128 -> 123 us, 135 -> 129.
constexpr int kKU = 16;                    // 16-byte units per K row (128 halves)
constexpr int kVU = 8;                     // 16-byte units per V^T row (64 keys)
constexpr int kLds16Bytes = (2 * kKeys * kKU + 2 * kC * kVU) * 16 + kKeys * 4;  // 32768 + 32768 + 256 = 64.25 KB: two workgroups per CU
// Token table: the source row and mask region of every key a workgroup walks, worked out once (token_row: a multiply-high
// division, the roll and the region compares -- 8 per loader thread and chunk otherwise, ~15 % of the kernel's VALU work) and
// kept behind the chunk image as row | region << 28.  Two workgroups per CU leave room for kTabCap keys; longer walks (or
// arrays of 2^24 rows / 4 GB and more) evaluate token_row per chunk as before.
constexpr int kTabCap = 3968;  // 62 chunks: 2 x (65792 + 15872) = 163328 of 163840 bytes
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int SHIFT>  // (a, b) * 2^-SHIFT -> packed fp16 h and packed (remainder * 2^11)
__device__ __forceinline__ void split2h(float a, float b, unsigned &h, unsigned &l) {
  const f32x2 v = (f32x2){a, b} * (1.f / (float)(1 << SHIFT));
  const f16x2 hh = __builtin_convertvector(v, f16x2);
  // (v - h) * 2048 as fma(h, -2048, 2048 v): v - h is exact in fp32 (h is v rounded to 11 bits) and so is the scaling, so the single
  // rounding of the fma returns the same bits -- and the fp16 operand goes into v_fma_mix_f32 without a conversion of its own
  const f32x2 v2k = (f32x2){a, b} * (2048.f / (float)(1 << SHIFT));
  const f32x2 r = {__builtin_fmaf((float)hh[0], -2048.f, v2k[0]), __builtin_fmaf((float)hh[1], -2048.f, v2k[1])};
  const f16x2 ll = __builtin_convertvector(r, f16x2);
  h = __builtin_bit_cast(unsigned, hh);
  l = __builtin_bit_cast(unsigned, ll);
}
[[maybe_unused]] constexpr int kKShift = 4;
__device__ __forceinline__ int vt_swz(int ch) { return ((ch >> 1) & 7) ^ ((ch >> 4) & 1); }

template <int WAVES>  // 4: two workgroups per CU; 8: one (128 query rows share a staged chunk, a loader thread stages 4 keys)
__global__ void __launch_bounds__(64 * WAVES, 2)  // two waves per SIMD either way: <= 256 registers
window_attention16_kernel(const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
                          float *__restrict__ out, Geometry g, int nwin, int qtiles, float scale, int ldq, int ldk, int ldv, int ksplit,
                          float *__restrict__ part, unsigned char *status, int tab_keys) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) u32x4 lds16[];
  // chunk image: K planes [key][kKU], V^T planes [channel][kVU], key regions; WAVES = 8 (one workgroup per CU) keeps TWO images: chunk
  // c + 1 is converted and written while chunk c is worked on, one barrier per chunk instead of two
  constexpr int NBUF = WAVES == 8 ? 2 : 1, IMG = kLds16Bytes / 16;  // 16-byte units per image
  int *tab = reinterpret_cast<int *>(lds16 + NBUF * IMG);  // [tab_keys] (0: none), behind the image(s)
  constexpr int TPW = 4, LIT = 32 / WAVES, ROWS = 16 * WAVES, NTH = 64 * WAVES;  // LIT: keys a loader thread holds

  const int lin = blockIdx.x;
  const int xcd = lin & 7;
  int slot = lin >> 3;
  const int ks = slot % ksplit;
  slot /= ksplit;
  const int win = (slot / qtiles) * 8 + xcd, qt = slot - (slot / qtiles) * qtiles;
  if (win >= nwin) return;
  const Window wd = window_of(g, win);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, grp = lane >> 4;
  const bool wlive = qt * ROWS + wave * 16 < g.L;  // a wave past the window's last query row only helps staging

  // ---- this lane's query row: B operand of S^T = K Q^T, channels 32 j + 8 grp .. + 7 for step j, split once
  const int qtok = qt * ROWS + wave * 16 + n16;
  const bool qlive = qtok < g.L;
  int qreg;
  const size_t qrow = token_row(g, wd, min(qtok, g.L - 1), qreg);
  u32x4 qh[4], ql[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x4 a = *reinterpret_cast<const f32x4 *>(q + qrow * ldq + 32 * j + 8 * grp);
    const f32x4 b = *reinterpret_cast<const f32x4 *>(q + qrow * ldq + 32 * j + 8 * grp + 4);
    unsigned h, l;
    split2h<0>(a[0], a[1], h, l), qh[j][0] = h, ql[j][0] = l;
    split2h<0>(a[2], a[3], h, l), qh[j][1] = h, ql[j][1] = l;
    split2h<0>(b[0], b[1], h, l), qh[j][2] = h, ql[j][2] = l;
    split2h<0>(b[2], b[3], h, l), qh[j][3] = h, ql[j][3] = l;
  }

  // ---- chunk loader: thread -> V^T unit lU = tid / 32 (mod 8), 4 channels at 4 * lq, the unit's 8 keys -- or, with 8 waves, 4 of
  // them: elements e0 .. e0 + 3, the halves crossed over at lq bit 3 (16 distinct bank pairs per 8-byte write group)
  const int lq = tid & 31, lc4 = lq * 4, lU = (tid >> 5) & 7;
  const int e0 = WAVES == 8 ? 4 * ((tid >> 8) ^ ((lq >> 3) & 1)) : 0;
  auto key_of = [&](int it) { return 32 * (lU >> 2) + 16 * ((e0 + it) >> 2) + 4 * (lU & 3) + (it & 3); };
  f32x4 pk[LIT], pv[LIT];
  const int all_chunks = (g.L + kKeys - 1) / kKeys, per = (all_chunks + ksplit - 1) / ksplit;
  const int ch0 = ks * per, chunks = min(all_chunks, ch0 + per);
  const bool use_tab = tab_keys > 0;  // (the host sizes it for `per` chunks or passes 0)
  if (use_tab) {
    for (int i = tid; i < (chunks - ch0) * kKeys; i += NTH) {
      int region;
      const unsigned row = token_row(g, wd, min(ch0 * kKeys + i, g.L - 1), region);
      tab[i] = (int)(row | (unsigned)region << 28);
    }
    __syncthreads();
  }
  auto fetch = [&](int chunk) {
    if (use_tab) {
#pragma unroll
      for (int half = 0; half < LIT / 4; ++half) {
        const int4 r = *reinterpret_cast<const int4 *>(&tab[(chunk - ch0) * kKeys + key_of(4 * half)]);
        const int rows[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // 32-bit byte offsets from a full-rate 24-bit multiply (the host offers the table only when they fit): the loads
          // take the scalar base + 32-bit offset form instead of 64-bit multiply-adds per key
          const unsigned row = (unsigned)rows[i] & 0x00ffffffu;
          pk[4 * half + i] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(k) + 4u * (__umul24(row, (unsigned)ldk) + (unsigned)lc4));
          pv[4 * half + i] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(v) + 4u * (__umul24(row, (unsigned)ldv) + (unsigned)lc4));
        }
      }
      return;
    }
#pragma unroll
    for (int it = 0; it < LIT; ++it) {
      int unused;
      const size_t row = token_row(g, wd, min(chunk * kKeys + key_of(it), g.L - 1), unused);
      pk[it] = *reinterpret_cast<const f32x4 *>(k + row * ldk + lc4);
      pv[it] = *reinterpret_cast<const f32x4 *>(v + row * ldv + lc4);
    }
  };
  auto stage = [&](int chunk, int b) {  // into image b
    u32x4 *KsH = lds16 + b * IMG, *VtH = KsH + 2 * kKeys * kKU, *VtL = VtH + kC * kVU;
    int *Kreg = reinterpret_cast<int *>(VtL + kC * kVU);
    // K: [key][segment ^ (key & 15)] planes, 8 bytes (4 channels) per plane and key
#pragma unroll
    for (int it = 0; it < LIT; ++it) {
      const int key = key_of(it);
      u32x2 h, l;
      unsigned a, b;
      split2h<kKShift>(pk[it][0], pk[it][1], a, b), h[0] = a, l[0] = b;
      split2h<kKShift>(pk[it][2], pk[it][3], a, b), h[1] = a, l[1] = b;
      unsigned char *dst = reinterpret_cast<unsigned char *>(KsH + key * kKU + ((lq >> 1) ^ (key & 15))) + 8 * (lq & 1);
      *reinterpret_cast<u32x2 *>(dst) = h;
      *reinterpret_cast<u32x2 *>(dst + kKeys * kKU * 16) = l;
    }
    // V^T: this thread's 8 keys of channel lc4 + c are one 16-byte unit [channel][unit lU ^ swizzle]
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int ch = lc4 + c, u = ch * kVU + (lU ^ vt_swz(ch));
      if constexpr (WAVES == 8) {
        u32x2 h, l;
        unsigned a, b;
        split2h<0>(pv[0][c], pv[1][c], a, b), h[0] = a, l[0] = b;
        split2h<0>(pv[2][c], pv[3][c], a, b), h[1] = a, l[1] = b;
        unsigned char *dst = reinterpret_cast<unsigned char *>(VtH + u) + 2 * e0;
        *reinterpret_cast<u32x2 *>(dst) = h;
        *reinterpret_cast<u32x2 *>(dst + kC * kVU * 16) = l;
      } else {
        u32x4 h, l;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          unsigned a, b;
          split2h<0>(pv[(2 * p) % LIT][c], pv[(2 * p + 1) % LIT][c], a, b);
          h[p] = a, l[p] = b;
        }
        VtH[u] = h;
        VtL[u] = l;
      }
    }
    if (g.shift && !use_tab && tid < kKeys) {
      int region;
      token_row(g, wd, min(chunk * kKeys + tid, g.L - 1), region);
      Kreg[tid] = region;
    }
  };

  f32x4 oh[8], ol[8];  // O^T tile dt: channels 16 dt + 4 grp + i of query n16; hi: h*h products, lo: the cross products (2^-11)
#pragma unroll
  for (int dt = 0; dt < 8; ++dt) oh[dt] = ol[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  const float s_scale = (float)(1 << kKShift) / scale;  // undoes K's pre-scale, applies 1 / sqrt(C)
  const int vsw = (n16 >> 1) & 7;  // vt_swz(16 dt + n16) = vsw ^ (dt & 1)
  if (ch0 < chunks) fetch(ch0);
  if (NBUF == 2 && ch0 < chunks) {
    stage(ch0, 0);
    if (ch0 + 1 < chunks) fetch(ch0 + 1);
  }
  for (int ch = ch0; ch < chunks; ++ch) {
    const int cb = NBUF == 2 ? (ch - ch0) & 1 : 0;
    if (NBUF == 2) {
      __syncthreads();  // image cb is written, every wave is done reading image cb ^ 1 (chunk ch - 1)
      if (ch + 1 < chunks) {
        stage(ch + 1, cb ^ 1);
        if (ch + 2 < chunks) fetch(ch + 2);
      }
    } else {
      __syncthreads();  // every wave is done reading the previous chunk
      stage(ch, 0);
      __syncthreads();
      if (ch + 1 < chunks) fetch(ch + 1);
    }
    if (!wlive) continue;
    const u32x4 *KsH = lds16 + cb * IMG, *KsL = KsH + kKeys * kKU, *VtH = KsH + 2 * kKeys * kKU, *VtL = VtH + kC * kVU;
    const int *Kreg = reinterpret_cast<const int *>(VtL + kC * kVU);

    // ---- S^T tiles: A = K rows (keys 16 t + n16, channels 32 j + 8 grp ..), B = this lane's Q fragment of step j
    f32x4 sh[TPW], sl[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) sh[t] = sl[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f16x8 bh = __builtin_bit_cast(f16x8, qh[j]), bl = __builtin_bit_cast(f16x8, ql[j]);
      const int seg = (4 * j + grp) ^ n16;
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const int u = (16 * t + n16) * kKU + seg;
        const f16x8 ah = __builtin_bit_cast(f16x8, KsH[u]), al = __builtin_bit_cast(f16x8, KsL[u]);
        sl[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, sl[t], 0, 0, 0);
        sl[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, sl[t], 0, 0, 0);
        sh[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, sh[t], 0, 0, 0);
      }
    }

    // ---- scale, mask, online softmax (fp32)
    float s[TPW][4];
    float mx = -INFINITY;
    const bool tail = (ch + 1) * kKeys > g.L;  // only the window's last chunk can hold keys past its end
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int key0 = 16 * t + 4 * grp;  // this lane's 4 keys of tile t
      int krs[4] = {0, 0, 0, 0};
      if (g.shift) {
        const int4 kr = *reinterpret_cast<const int4 *>(use_tab ? &tab[(ch - ch0) * kKeys + key0] : &Kreg[key0]);
        krs[0] = kr.x, krs[1] = kr.y, krs[2] = kr.z, krs[3] = kr.w;
        if (use_tab)
#pragma unroll
          for (int i = 0; i < 4; ++i) krs[i] = (int)((unsigned)krs[i] >> 28);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float x = (sh[t][i] + sl[t][i] * (1.f / 2048.f)) * s_scale;  // scores / sqrt(C) (transformer.py:91)
        if (g.shift && krs[i] != qreg) x += -100.f;
        if (tail && ch * kKeys + key0 + i >= g.L) x = -INFINITY;
        s[t][i] = x;
        mx = fmaxf(mx, x);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);       // finite: every chunk holds at least one real key
    const float alpha = __expf(m_run - m_new);  // 0 on the first chunk
    float ls = 0.f;
#pragma unroll
    for (int t = 0; t < TPW; ++t)
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
    for (int dt = 0; dt < 8; ++dt) oh[dt] *= alpha, ol[dt] *= alpha;

    // ---- O^T += V^T P^T: step ss contracts unit 4 ss + grp; B = this lane's probabilities of tiles 2 ss, 2 ss + 1
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      u32x4 ph, pl;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        unsigned a, b;
        split2h<0>(s[2 * ss + (p >> 1)][2 * (p & 1)], s[2 * ss + (p >> 1)][2 * (p & 1) + 1], a, b);
        ph[p] = a, pl[p] = b;
      }
      const f16x8 bh = __builtin_bit_cast(f16x8, ph), bl = __builtin_bit_cast(f16x8, pl);
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        const int u = (16 * dt + n16) * kVU + ((4 * ss + grp) ^ vsw ^ (dt & 1));
        const f16x8 ah = __builtin_bit_cast(f16x8, VtH[u]), al = __builtin_bit_cast(f16x8, VtL[u]);
        ol[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, ol[dt], 0, 0, 0);
        ol[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, ol[dt], 0, 0, 0);
        oh[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, oh[dt], 0, 0, 0);
      }
    }
  }

  // lane (query n16, group grp) holds channels 16 dt + 4 grp + i of its row
  if (ksplit > 1) {
    if (qlive) {
      float *prow = part + (((size_t)win * g.L + qtok) * ksplit + ks) * 132;
      if (grp == 0) {
        prow[0] = m_run;
        prow[1] = l_run;
      }
      float nf = nf_fold(0.f, l_run);  // the family's overflow report (common.hpp): Q / K / V past fp16's range end up here
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        const f32x4 o = oh[dt] + ol[dt] * (1.f / 2048.f);
#pragma unroll
        for (int i = 0; i < 4; ++i) nf = nf_fold(nf, o[i]);
        *reinterpret_cast<f32x4 *>(prow + 4 + 16 * dt + 4 * grp) = o;
      }
      nf_report(status, DRBA_STATUS_ATTENTION, nf);
    }
    return;
  }
  if (qlive) {
    const float inv = 1.f / l_run;
    float *orow = out + qrow * kC + 4 * grp;
    float nf = 0.f;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
      const f32x4 o = (oh[dt] + ol[dt] * (1.f / 2048.f)) * inv;
#pragma unroll
      for (int i = 0; i < 4; ++i) nf = nf_fold(nf, o[i]);
      *reinterpret_cast<f32x4 *>(orow + 16 * dt) = o;
    }
    nf_report(status, DRBA_STATUS_ATTENTION, nf);
  }
#endif
}

// out row = sum_ks exp(m_ks - m) O_ks / sum_ks exp(m_ks - m) l_ks, m = max_ks m_ks; one wave per query row, 2 channels per lane
__global__ void __launch_bounds__(256)
window_attention_merge(const float *__restrict__ part, float *__restrict__ out, Geometry g, int nwin, int ksplit) {
  const int lane = threadIdx.x & 63;
  const size_t rowi = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (rowi >= (size_t)nwin * g.L) return;
  const int win = (int)(rowi / g.L), tok = (int)(rowi - (size_t)win * g.L);
  const float *p = part + rowi * ksplit * 132;
  float m = -INFINITY;
  for (int s = 0; s < ksplit; ++s) m = fmaxf(m, p[s * 132]);
  float l = 0.f, a = 0.f, b = 0.f;
  for (int s = 0; s < ksplit; ++s) {
    const float ms = p[s * 132];
    const float w = ms == -INFINITY ? 0.f : __expf(ms - m);  // a run without keys (short windows) has m = -inf, l = 0
    l += w * p[s * 132 + 1];
    a += w * p[s * 132 + 4 + 2 * lane];
    b += w * p[s * 132 + 4 + 2 * lane + 1];
  }
  int region;
  const Window wd = window_of(g, win);
  const size_t orow = token_row(g, wd, tok, region);
  const float inv = 1.f / l;
  out[orow * kC + 2 * lane] = a * inv;
  out[orow * kC + 2 * lane + 1] = b * inv;
}

}  // namespace drba_attn

// Waves per workgroup of the two-term kernel: 8 (128 query rows per staged chunk, one workgroup per CU) from 192 tokens per window
// on -- below that half of an 8-wave workgroup would only help staging.  Same box, 8 windows x 2160 tokens: 150 -> 126 us, 128
// windows x 540: 140 -> 134 (tools/exp/attn/waves_sweep.sh).
static int attn16_waves(int L) {
  static const int w = env_int("DRBA_ATTN_WAVES", 0);  // (TUNING builds only)
  if (w == 4 || w == 8) return w;
  return L >= 192 ? 8 : 4;
}
// key runs per window (separate workgroups, merged by window_attention_merge), fp32 kernel: the rule it was measured with
static int attn_ksplit(int nwin, int L) {
  const int rows = drba_attn::kRows, qtiles = (L + rows - 1) / rows, chunks = (L + drba_attn::kKeys - 1) / drba_attn::kKeys;
  const long long wgs = (long long)((nwin + 7) / 8) * 8 * qtiles;
  if (wgs >= 2 * 256 || chunks < 8) return 1;  // two workgroups per CU already, or too few chunks to share out
  return chunks >= 16 ? 4 : 2;  // measured on 8 windows x 2160 tokens: 349 us unsplit, 287 us in two runs, 260 us in four
}
// two-term kernel: the launch ends with the fullest CU, i.e. after (rounds of resident workgroups) x (a workgroup's chunks + its
// fixed cost of ~2.5 chunk times); a split pays the partial rows and the merge launch once more.  8 windows x 2160 tokens, 8 waves
// (136 workgroups per run on 256 slots), runs 1..10: 143 / 159 / 126 / 143 / 127 / 142 / 133 / 155 / 144 / 148 us -- the model's
// order; 128 windows x 540: every split slower than none (tools/exp/attn/waves8_ksplit.sh).
static int attn16_ksplit(int nwin, int L, int waves) {
  static const int force = env_int("DRBA_ATTN_KSPLIT", 0);  // (TUNING builds only)
  if (force > 0) return force;
  const int rows = 16 * waves, qtiles = (L + rows - 1) / rows, chunks = (L + drba_attn::kKeys - 1) / drba_attn::kKeys;
  const long long per_run = (long long)((nwin + 7) / 8) * 8 * qtiles, slots = 256 * (waves == 8 ? 1 : 2);
  int best = 1;
  long long best_cost = -1;
  for (int ks = 1; ks <= 10 && (ks == 1 || chunks / ks >= 4); ++ks) {
    const long long rounds = (per_run * ks + slots - 1) / slots;
    const long long cost = rounds * (2 * ((chunks + ks - 1) / ks) + 5) + (ks > 1 ? 3 : 0);
    if (best_cost < 0 || cost < best_cost) best = ks, best_cost = cost;
  }
  return best;
}

// floats of workspace drba_window_attention needs for this shape (0: none); `terms` is not known here: the larger of the two forms'
extern "C" size_t drba_window_attention_ws_floats(int B, int H, int W, int splits) {
  if (B <= 0 || H <= 0 || W <= 0 || splits <= 0 || H % splits || W % splits) return 0;
  const int nwin = B * splits * splits, L = (H / splits) * (W / splits);
  const int ks = std::max(attn_ksplit(nwin, L), attn16_ksplit(nwin, L, attn16_waves(L)));
  return ks > 1 ? (size_t)nwin * L * ks * 132 : 0;
}

extern "C" int drba_window_attention(const float *q, const float *k, const float *v, float *out, int B, int H, int W, int C,
                                     int splits, int shift, float scale, int ldq, int ldk, int ldv, float *ws, int terms,
                                     void *stream) {
  if (!q || !k || !v || !out || B <= 0 || H <= 0 || W <= 0 || splits <= 0 || !(scale > 0.f)) return DRBA_EINVAL;
  if (terms != 2 && terms != 3) return DRBA_EINVAL;  // 3: fp32 MFMA (the name counts operand bits: 24); 2: two fp16 terms
  if (C != drba_attn::kC) return DRBA_EUNSUPPORTED;  // GMFlow's feature_channels
  if (ldq < C || ldk < C || ldv < C || ((ldq | ldk | ldv) & 3)) return DRBA_EINVAL;  // rows are read as 16-byte vectors
  if (H % splits || W % splits) return DRBA_EINVAL;  // the reference's window split needs whole windows
  drba_attn::Geometry g;
  g.h = H, g.w = W, g.splits = splits, g.wh = H / splits, g.ww = W / splits;
  g.sh = g.wh / 2, g.sw = g.ww / 2, g.L = g.wh * g.ww, g.shift = shift ? 1 : 0;
  if (g.ww >= 65536 || g.L >= 65536) return DRBA_EUNSUPPORTED;  // range of the multiply-shift division
  // a window one pixel wide or high has shift 0 in that direction; the reference builds its mask with slice(-0, None),
  // which is the whole axis -- a degenerate table this kernel's coordinate rule does not reproduce
  if (g.shift && (g.sh == 0 || g.sw == 0)) return DRBA_EUNSUPPORTED;
  g.ww_magic = g.ww == 1 ? 0u : (unsigned)(((1ull << 32) + g.ww - 1) / g.ww);
  const int nwin = B * splits * splits;
  const int waves = terms == 2 ? attn16_waves(g.L) : 4, rows = 16 * waves;
  const int qtiles = (g.L + rows - 1) / rows;
  const int groups = (nwin + 7) / 8;
  // without a workspace every workgroup walks all keys
  const int ksplit = !ws ? 1 : terms == 2 ? attn16_ksplit(nwin, g.L, waves) : attn_ksplit(nwin, g.L);
  const dim3 grid((unsigned)(groups * 8 * qtiles * ksplit));
  // beyond the default 64 KB dynamic-LDS limit
  if (terms == 2) {
    const int all_chunks = (g.L + drba_attn::kKeys - 1) / drba_attn::kKeys, walk = (all_chunks + ksplit - 1) / ksplit * drba_attn::kKeys;
    const long long rows_all = (long long)B * H * W;
    const bool fits = rows_all < (1ll << 24) && rows_all * std::max(ldk, ldv) < (1ll << 30) && ldk < (1 << 24) && ldv < (1 << 24);
    const int tab_keys = walk <= drba_attn::kTabCap && fits ? walk : 0;
    const int lds_bytes = drba_attn::kLds16Bytes * (waves == 8 ? 2 : 1) + tab_keys * 4;
    auto go = [&](auto kern) -> int {
      if (max_dynamic_lds(reinterpret_cast<const void *>(kern), lds_bytes) != hipSuccess) return DRBA_ELAUNCH;
      DRBA_LAUNCH(kern, grid, dim3(64 * waves), lds_bytes, (hipStream_t)stream, q, k, v, out, g, nwin, qtiles, scale, ldq, ldk, ldv,
                  ksplit, ws, status_bytes(), tab_keys);
      return DRBA_OK;
    };
    const int rc = waves == 8 ? go(drba_attn::window_attention16_kernel<8>) : go(drba_attn::window_attention16_kernel<4>);
    if (rc != DRBA_OK) return rc;
  } else {
    if (max_dynamic_lds(reinterpret_cast<const void *>(drba_attn::window_attention_kernel), drba_attn::kLdsBytes) != hipSuccess)
      return DRBA_ELAUNCH;
    DRBA_LAUNCH(drba_attn::window_attention_kernel, grid, dim3(kBlock), drba_attn::kLdsBytes, (hipStream_t)stream, q, k, v,
                      out, g, nwin, qtiles, scale, ldq, ldk, ldv, ksplit, ws);
  }
  if (ksplit > 1)
    DRBA_LAUNCH(drba_attn::window_attention_merge, dim3((unsigned)(((size_t)nwin * g.L + 3) / 4)), dim3(kBlock), 0,
                       (hipStream_t)stream, ws, out, g, nwin, ksplit);
  DRBA_CHECK_LAUNCH();
  return terms == 2 ? range_checked(DRBA_OK, out, (size_t)B * H * W * C, stream) : DRBA_OK;  // (out is [B, H*W, C], dense)
}
