// 3x3 convolution (stride 1/2, pad 1) and ConvTranspose2d(4, 2, 1) as fp32 implicit GEMM on
// v_mfma_f32_16x16x4_f32 (exact fp32: bit-equal to a k-ordered fmaf chain; gfx950 has no
// TF32/xf32 path and plain bf16 inputs break the 1e-3 parity bar, SURVEY.md 'Hard parts').
// Stride-1 layers with Cin % 32 == 0 also have the split-bf16 kernels of conv_split.hip (three bf16 terms per
// fp32 operand, fp32-level error) behind the same entry points; this file stays the kernel for stride 2,
// transposed convolutions, ragged channel counts and small maps.
//
// GEMM view per workgroup (4 waves = one per SIMD):
//   M = output pixels : a TH x TW tile, split into 16-pixel row segments (MFMA rows)
//   N = output channels: NT tiles of 16 (MFMA cols)
//   K = (tap, cin)    : cin consumed 4 at a time (MFMA k = 4), CK channels staged per LDS chunk
// MFMA operand fetch: lane l supplies A[pixel l&15][cin l>>4] and B[cin l>>4][cout l&15].
//   A: one ds_read_b32 from the LDS input tile [CK][rows][cols]; channel stride == 16 (mod 32)
//      for stride 1 and odd for stride 2, which makes the read bank-conflict free.
//   B: weights are pre-packed on the host in MFMA *fragment order*
//      ([cout tile][chunk][tap][cg][nt][64 lanes]): a chunk's block is contiguous and a B operand is one
//      conflict-free ds_read_b32 (lane-linear).  (Feeding B straight from L2 was measured first: hipcc sinks the
//      loads next to their MFMAs and every k-step eats an L1/L2 round trip -- 35 % MFMA utilisation.)
// Staging is LDS-direct (buffer_load ... lds, no registers, no address math in the chunk loop): the chunk's input
// windows and its weight block land in one of two LDS buffers while the MFMAs run on the other; one
// s_waitcnt vmcnt(0) + workgroup barrier per chunk.  The MFMA loop itself is software-pipelined by hand (the LDS
// reads of step j+1 issue before the MFMAs of step j).  Workgroups are remapped so that an XCD owns a contiguous run
// of tiles (private L2s).
// Accumulator D: lane holds cout l&15 for pixels 4*(l>>4)..+3 -> one float4 store along x.
//
// KS = 1: the 4 waves split the tile's rows (TH = 4*RW) and share the LDS input tile.
// KS = 4: the 4 waves share the whole tile (TH = RW) and split K: wave w owns chunks q == w
//         (mod 4), stages them into wave-private LDS (no workgroup barrier in the K loop) and the
//         partial accumulators are reduced through LDS at the end.  For small maps, where a
//         serial K loop over all input channels is the latency floor.
//
// MODE 0: conv3x3.  MODE 1: one row phase py (both column phases px) of ConvTranspose2d(k=4,s=2,p=1):
//   out[o, 2j+py, 2i+px] = b[o] + sum_{c, a, b in {0,1}} in[c, j+dy(py,a), i+dx(px,b)] * W[c, o, ky(py,a), kx(px,b)]
//   with (py=0: (ky,dy) = (1,0),(3,-1); py=1: (0,+1),(2,0)), same along x: a 2x2-tap convolution
//   over the same haloed input tile; the workgroup index carries the row phase, both column phases are computed
//   by the same workgroup so that its stores are runs of consecutive output columns.
#include "common.hpp"
#include "conv_split.hpp"

#include <stdlib.h>
#include <string.h>

using namespace drba;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_ptr;

namespace drba_conv {

constexpr int round_up_mod32_16(int v) {  // smallest r >= v with r % 32 == 16
  int r = v - (v % 32) + 16;
  return r >= v ? r : r + 32;
}

template <int MODE_, int S_, int RW_, int MW_, int NT_, int CK_, int KS_>
struct ConvCfg {
  static constexpr int MODE = MODE_, S = S_, RW = RW_, MW = MW_, NT = NT_, CK = CK_, KS = KS_;
  static constexpr int NTAP = (MODE == 0) ? 9 : 4;
  static constexpr int TH = (KS == 1 ? 4 : 1) * RW, TW = 16 * MW, NTC = 16 * NT;
  static constexpr int TR = (TH - 1) * S + 3, TC = (TW - 1) * S + 3;
  static constexpr int NL = (KS == 1) ? 256 : 64;       // lanes filling one buffer (the workgroup, or one wave)
  // Staging is LDS-direct (buffer_load ... lds): lane l of a load writes LDS dword base + l.  The chunk's CK channel
  // windows are laid out back to back ([CK][CHS], CHS = window + bank padding) and fetched as one linear run of
  // SPB loads of NL lanes: a lane's source (channel-in-chunk, row, col) is fixed for the whole kernel.
  static constexpr int CHS = (S == 1) ? round_up_mod32_16(TR * TC) : ((TR * TC) | 1);
  static constexpr int BUF = ((CK * CHS + 63) / 64) * 64;  // floats per staging buffer: whole 64-lane loads, 16-byte aligned end
  static constexpr int SPB = (CK * CHS + NL - 1) / NL;
  static constexpr int CG = CK / 4;                      // MFMA k-groups per tap per chunk
  static constexpr int FRAG = NTAP * CG * NT * 64;       // packed weight floats per (cout tile[, phase], chunk)
  // MODE 1: a workgroup computes BOTH column phases (px = 0, 1) of its row phase, so that its lanes own runs of
  // consecutive output columns (coalesced stores; 64 scattered dwords per store instruction otherwise)
  static constexpr int NPX = (MODE == 1) ? 2 : 1;
  static constexpr int NTILES = NPX * RW * MW * NT;
  // Weight path: KS=1 stages the chunk's fragment block through LDS (shared by the 4 waves, 16-byte LDS-direct
  // loads); KS=4 (small maps, wave-private chunks) reads B fragments straight from L2 -- staging them per
  // wave would quadruple the LDS writes (measured: 40 -> 56 us on the 64-ch 136x240 ResConv).
  static constexpr bool WLDS = (KS == 1);
  static constexpr int SPW = WLDS ? (FRAG / 4 + NL - 1) / NL : 0;   // 16-byte weight loads per lane per chunk
  static constexpr int FPAD = WLDS ? ((FRAG / 4 + 63) / 64) * 256 : 0;  // fragment block rounded up to whole wave loads
  static constexpr int BUFALL = BUF + NPX * FPAD;        // input tile (+ weight fragments of every column phase) of one chunk
  static constexpr int LDS_STAGE = 2 * BUFALL * (KS == 1 ? 1 : 4);
  static constexpr int LDS_RED = (KS == 1) ? 0 : 4 * NTILES * 256;
  static constexpr int LDS_FLOATS = LDS_STAGE > LDS_RED ? LDS_STAGE : LDS_RED;
  static_assert(MODE == 0 || S == 1, "deconv phases read the input at stride 1");
};

template <class Cfg, bool PRE>
__global__ void __launch_bounds__(256)
conv_mfma(const float *__restrict__ in, const float *__restrict__ wfrag, const float *__restrict__ bias,
          const float *__restrict__ beta, const float *__restrict__ res, const float *__restrict__ res2,
          float *__restrict__ out, int Cin, int H, int W, int Cout, int Ho, int Wo, int act, float post_slope,
          float pre_slope, int n_ctiles, int pixel_shuffle) {
#if defined(__HIP_DEVICE_COMPILE__)  // the body uses device-only types/builtins (buffer resources, LDS-direct loads); the host pass only needs the launch stub
  constexpr int MODE = Cfg::MODE, S = Cfg::S, RW = Cfg::RW, MW = Cfg::MW, NT = Cfg::NT, CK = Cfg::CK, KS = Cfg::KS;
  constexpr int TH = Cfg::TH, TW = Cfg::TW, TR = Cfg::TR, TC = Cfg::TC, CHS = Cfg::CHS;
  constexpr int NL = Cfg::NL, SPB = Cfg::SPB, SPW = Cfg::SPW, CG = Cfg::CG, NTAP = Cfg::NTAP;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: keeps chunk indices and LDS bases in SGPRs
  const int m = lane & 15, kq = lane >> 4;
  // XCD-aware tile order: workgroups are dispatched round-robin over the 8 XCDs (linear id % 8), each with a private
  // L2.  Remap so that an XCD owns a contiguous run of tiles in (image, tile row, tile column, cout tile[, phase])
  // order with the innermost indices sharing their input window: the halo rows/columns neighbouring tiles share and
  // the window every cout tile / deconv phase re-reads then hit that XCD's L2 instead of being fetched by up to 8.
  const int nbx = gridDim.x, nby = gridDim.y;
  int t = xcd_band((int)(blockIdx.x + nbx * (blockIdx.y + nby * blockIdx.z)), (int)(nbx * nby * gridDim.z));
  int py = 0;
  if (MODE == 1) {
    py = t & 1;
    t >>= 1;
  }
  const int cz = t % n_ctiles;
  t /= n_ctiles;
  const int bx = t % nbx;
  t /= nbx;
  const int by = t % nby, n = t / nby;
  const int x0 = bx * TW, y0 = by * TH;
  in += (size_t)n * Cin * H * W;
  if (MODE == 0) {
    out += (size_t)n * Cout * Ho * Wo;
    if (res) res += (size_t)n * Cout * Ho * Wo;
    if (res2) res2 += (size_t)n * Cout * Ho * Wo;
  } else {
    out += (size_t)n * Cout * (2 * H) * (2 * W);
  }

  constexpr int NPX = Cfg::NPX;
  f32x4 acc[NPX][RW][MW][NT];
#pragma unroll
  for (int p = 0; p < NPX; ++p)
#pragma unroll
    for (int a = 0; a < RW; ++a)
#pragma unroll
      for (int b = 0; b < MW; ++b)
#pragma unroll
        for (int c = 0; c < NT; ++c) acc[p][a][b][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nchunks = (Cin + CK - 1) / CK;
  float *buf0 = smem + (KS == 1 ? 0 : wave * 2 * Cfg::BUFALL);
  float *buf1 = buf0 + Cfg::BUFALL;
  const int ltid = (KS == 1) ? tid : lane;
  const int row0 = (KS == 1) ? wave * RW : 0;
  const int a_off = kq * CHS + (row0 * S) * TC + m * S;
  const int gy0 = y0 * S - 1, gx0 = x0 * S - 1;
  const int q_first = (KS == 1) ? 0 : wave, q_step = (KS == 1) ? 1 : 4;
  // packed weights: [cout tile][phase = 2*py + px][chunk][FRAG]; the two column phases of this row phase are adjacent
  const float *wf_base = wfrag + ((size_t)(MODE == 0 ? cz : cz * 4 + 2 * py) * nchunks) * Cfg::FRAG;
  const size_t px_stride = (size_t)nchunks * Cfg::FRAG;  // floats between the px = 0 and px = 1 fragment blocks
  // deconv tap -> tile offsets: tap = 2a+b; row = rw + 1 + dy, py=0: dy={0,-1}; py=1: dy={+1,0}
  const int dro[2] = {py ? 2 : 1, py ? 1 : 0};
  const int dco[2][2] = {{1, 0}, {2, 1}};  // [px][b]

  // ---- staging: global -> LDS without passing through registers.  Each lane owns the same (row, col) of the
  // window for every channel, so its byte offset inside a channel plane is computed once; the channel / chunk
  // offset rides in the scalar soffset.  Out-of-image taps, window padding lanes and channels >= Cin read as
  // zero through the buffer range check (an out-of-range lane still writes its 0 to LDS).
  constexpr unsigned kOOB = 0x7FFFFFF0u;
  const unsigned plane_bytes = (unsigned)H * (unsigned)W * 4u;
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, Cin * (int)plane_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc((void *)wf_base, 0, NPX * nchunks * Cfg::FRAG * 4, 0x00020000);
  unsigned voff[SPB];
#pragma unroll
  for (int i = 0; i < SPB; ++i) {
    const int L = ltid + i * NL;             // LDS dword within the buffer
    const int c = L / CHS, e = L - c * CHS;  // channel within the chunk, element of its window (or bank padding)
    const int r = e / TC, col = e - r * TC;
    const int gy = gy0 + r, gx = gx0 + col;
    const bool ok = c < CK && e < TR * TC && gy >= 0 && gy < H && gx >= 0 && gx < W;
    voff[i] = ok ? (unsigned)c * plane_bytes + (unsigned)(gy * W + gx) * 4u : kOOB;
  }
  const int wslot = (KS == 1) ? wave * 64 : 0;  // first LDS dword this wave's lanes write within a load
  auto issue = [&](int q, float *buf) {
    const unsigned soff = (unsigned)(q * CK) * plane_bytes;  // channels >= Cin fall past num_records: zeros
#pragma unroll
    for (int i = 0; i < SPB; ++i)
      if (i * NL + wslot < CK * CHS)  // wave-uniform: waves wholly past the buffer skip the load
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr)(buf + i * NL + wslot), 4, voff[i], soff, 0, 0);
    if (Cfg::WLDS) {
#pragma unroll
      for (int p = 0; p < NPX; ++p)
#pragma unroll
        for (int i = 0; i < SPW; ++i)  // lanes past the fragment block fetch the next chunk's head into LDS padding
          if (i * NL + wslot < Cfg::FRAG / 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rs_w, (lds_ptr)(buf + Cfg::BUF + p * Cfg::FPAD + (i * NL + wslot) * 4), 16, (unsigned)(i * NL + ltid) * 16u,
                (unsigned)(((size_t)p * px_stride + (size_t)q * Cfg::FRAG) * 4), 0, 0);
    }
  };
  auto sync = [&]() {  // the chunk just requested has landed and everybody is done reading the other buffer
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (KS == 1) __syncthreads();
    else __builtin_amdgcn_wave_barrier();  // wave-private buffers
  };

  if (q_first < nchunks) issue(q_first, buf0);
  sync();
  int it = 0;
  for (int q = q_first; q < nchunks; q += q_step, ++it) {
    const float *cur = (it & 1) ? buf1 : buf0;
    float *nxt = (it & 1) ? buf0 : buf1;
    const int qn = q + q_step;
    if (qn < nchunks) issue(qn, nxt);  // next chunk's loads fly under this chunk's MFMAs
    const float *ab = cur + a_off;
    // this chunk's weight fragments [tap][cg][nt][64 lanes]: LDS copy (KS=1) or L2 (KS=4)
    const float *wb = Cfg::WLDS ? cur + Cfg::BUF + lane : wf_base + (size_t)q * Cfg::FRAG + lane;
    if constexpr (Cfg::WLDS) {
      // Software-pipelined over the (tap, k-group) steps: the A/B fragments of step j+1 are read from LDS while the
      // MFMAs of step j issue, so a wave waits for LDS once per chunk instead of once per step (left to the
      // compiler, the reads sit directly in front of the MFMAs that consume them: MFMA pipe 55-65 % busy).
      constexpr int NSTEP = NPX * NTAP * CG;  // (column phase,) tap, k-group
      float av[2][RW][MW], bv[2][NT];
      auto fetch = [&](int j, int slot) {
        const int p = j / (NTAP * CG), jj = j - p * (NTAP * CG);
        const int tap = jj / CG, cg = jj - tap * CG;
        const float *at = (MODE == 0) ? ab + (tap / 3) * TC + (tap % 3) : ab + dro[tap >> 1] * TC + dco[p][tap & 1];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv[slot][nt] = wb[p * Cfg::FPAD + ((tap * CG + cg) * NT + nt) * 64];
#pragma unroll
        for (int rw = 0; rw < RW; ++rw)
#pragma unroll
          for (int mw = 0; mw < MW; ++mw) av[slot][rw][mw] = at[(cg * 4) * CHS + (rw * S) * TC + mw * 16 * S];
      };
      fetch(0, 0);
#pragma unroll
      for (int j = 0; j < NSTEP; ++j) {
        const int slot = j & 1, p = j / (NTAP * CG);
        if (j + 1 < NSTEP) fetch(j + 1, slot ^ 1);
#pragma unroll
        for (int rw = 0; rw < RW; ++rw)
#pragma unroll
          for (int mw = 0; mw < MW; ++mw) {
            float a = av[slot][rw][mw];
            // PReLU (one shared slope) of pre-activated convolutions, applied on the way into the MFMA
            // (the staged data never passes through registers); prelu(0) = 0 keeps the zero padding
            if (PRE) a = a > 0.f ? a : pre_slope * a;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[p][rw][mw][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[slot][nt], acc[p][rw][mw][nt], 0, 0, 0);
          }
        // keep the issue order: the LDS reads of step j+1, then the MFMAs of step j
        __builtin_amdgcn_sched_group_barrier(0x100, RW * MW + NT, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, RW * MW * NT, 0);
      }
    } else {
      // split-K: the B fragments come from L2 (global loads); the compiler hoists them across the taps on its own,
      // and pinning an LDS-style pipeline here serialises one L2 round trip per step (measured 21 -> 48 us)
#pragma unroll
      for (int p = 0; p < NPX; ++p)
#pragma unroll
        for (int tap = 0; tap < NTAP; ++tap) {
          const float *at = (MODE == 0) ? ab + (tap / 3) * TC + (tap % 3) : ab + dro[tap >> 1] * TC + dco[p][tap & 1];
#pragma unroll
          for (int cg = 0; cg < CG; ++cg) {
            float bv[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[nt] = wb[p * px_stride + ((tap * CG + cg) * NT + nt) * 64];
#pragma unroll
            for (int rw = 0; rw < RW; ++rw)
#pragma unroll
              for (int mw = 0; mw < MW; ++mw) {
                float a = at[(cg * 4) * CHS + (rw * S) * TC + mw * 16 * S];
                if (PRE) a = a > 0.f ? a : pre_slope * a;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                  acc[p][rw][mw][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[nt], acc[p][rw][mw][nt], 0, 0, 0);
              }
          }
        }
    }
    sync();
  }

  // ---- epilogue
  const bool vec = (Wo & 3) == 0;
  auto store_tile = [&](auto post, int rw, int mw, int nt, f32x4 v, f32x4 v1) {  // v1: the px = 1 phase of a transposed conv
    (void)v1;
    (void)post;
    const int co = cz * Cfg::NTC + nt * 16 + m;
    const int y = y0 + row0 + rw;
    const int xb = x0 + mw * 16 + kq * 4;
    if (co >= Cout && !(MODE == 1 && pixel_shuffle)) return;  // (PixelShuffle: Cout % 4 == 0, lane pairs stay together)
    const float bs = (bias && co < Cout) ? bias[co] : 0.f;
    if (MODE == 0) {
      // y = acc + bias; ResConv: y = y*beta + res; otherwise y += res (+ res2); then the post activation:
      // act 0 none, 1 LeakyReLU(0.2), 2 PReLU(post_slope), 3 ReLU, 4 tanh(y)*10
      if (y >= Ho || xb >= Wo) return;
      const float bt = beta ? beta[co] : 0.f;
      const size_t idx = ((size_t)co * Ho + y) * Wo + xb;
      if (vec) {
        f32x4 r = (f32x4){0.f, 0.f, 0.f, 0.f}, r2 = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (res) r = *reinterpret_cast<const f32x4 *>(res + idx);
        if (res2) r2 = *reinterpret_cast<const f32x4 *>(res2 + idx);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float t = v[k] + bs;
          if (beta) t = t * bt + r[k];
          else {
            if (res) t = t + r[k];
            if (res2) t = t + r2[k];
          }
          v[k] = post(t);
        }
        *reinterpret_cast<f32x4 *>(out + idx) = v;
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (xb + k >= Wo) continue;
          float t = v[k] + bs;
          if (beta) t = t * bt + res[idx + k];
          else {
            if (res) t = t + res[idx + k];
            if (res2) t = t + res2[idx + k];
          }
          out[idx + k] = post(t);
        }
      }
    } else {
      // transposed conv, row phase py, both column phases: lane (cout m, column group kq) holds the input columns
      // xb..xb+3 for px = 0 (v) and px = 1 (v1), i.e. the 8 consecutive output columns 2*xb .. 2*xb+7 of row 2y+py.
      const int Hd = 2 * H, Wd = 2 * W;
      const int oy = 2 * y + py;
      if (!pixel_shuffle) {
        if (y >= H || xb >= W) return;
        float *dst = out + ((size_t)co * Hd + oy) * Wd + 2 * xb;
        if (xb + 3 < W && (Wd & 3) == 0) {
          *reinterpret_cast<f32x4 *>(dst) = (f32x4){v[0] + bs, v1[0] + bs, v[1] + bs, v1[1] + bs};
          *reinterpret_cast<f32x4 *>(dst + 4) = (f32x4){v[2] + bs, v1[2] + bs, v[3] + bs, v1[3] + bs};
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (xb + k < W) {
              dst[2 * k] = v[k] + bs;
              dst[2 * k + 1] = v1[k] + bs;
            }
        }
      } else {
        // + PixelShuffle(2): cout = 4*c13 + 2*si + sj lands at row 2*oy+si, column 2*(2i+px)+sj = 4i + 2px + sj of plane
        // c13.  Lanes m and m^1 (sj = 0 / 1, same c13 and si) exchange their values, after which each holds the 4
        // consecutive columns 4i..4i+3 for every i; the even lane stores i = xb, xb+1, the odd lane i = xb+2, xb+3.
        const int c13 = co >> 2, si = (co >> 1) & 1, sj = co & 1;
        f32x4 o0, o1;  // partner's px=0 / px=1 values (bias of ITS cout added before the exchange)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          v[k] += bs;
          v1[k] += bs;
          o0[k] = quad_xor1(v[k]);
          o1[k] = quad_xor1(v1[k]);
        }
        if (co >= Cout || y >= H || xb >= W) return;
        float *dst = out + ((size_t)c13 * (2 * Hd) + (2 * oy + si)) * (size_t)(2 * Wd) + 4 * xb;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int k = sj * 2 + kk;  // even lane: input columns xb, xb+1; odd lane: xb+2, xb+3
          if (xb + k >= W) continue;
          const f32x4 q = sj ? (f32x4){o0[k], v[k], o1[k], v1[k]} : (f32x4){v[k], o0[k], v1[k], o1[k]};
          *reinterpret_cast<f32x4 *>(dst + 4 * k) = q;
        }
      }
    }
  };

  // The post activation is selected ONCE around the tile loops: selected per element, every inlined copy of the switch
  // (with a tanhf expansion to jump over) ends up in the epilogue and its instruction fetch costs more than the stores.
  auto finish = [&](auto post) {
    if (KS == 1) {
  #pragma unroll
      for (int rw = 0; rw < RW; ++rw)
  #pragma unroll
        for (int mw = 0; mw < MW; ++mw)
  #pragma unroll
          for (int nt = 0; nt < NT; ++nt) store_tile(post, rw, mw, nt, acc[0][rw][mw][nt], acc[NPX - 1][rw][mw][nt]);
    } else {
      // cross-wave K reduction: partials -> LDS, then wave w finishes tiles t == w (mod 4)
      constexpr int NTILES = Cfg::NTILES, NT1 = RW * MW * NT;  // NTILES = NPX * NT1, phase-major
      __syncthreads();  // every wave is done with its staging buffers
      f32x4 *red = reinterpret_cast<f32x4 *>(smem);
  #pragma unroll
      for (int p = 0; p < NPX; ++p)
  #pragma unroll
        for (int rw = 0; rw < RW; ++rw)
  #pragma unroll
          for (int mw = 0; mw < MW; ++mw)
  #pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              red[(wave * NTILES + p * NT1 + (rw * MW + mw) * NT + nt) * 64 + lane] = acc[p][rw][mw][nt];
      __syncthreads();
      auto total = [&](int t) -> f32x4 {
        f32x4 v = red[(0 * NTILES + t) * 64 + lane];
        v += red[(1 * NTILES + t) * 64 + lane];
        v += red[(2 * NTILES + t) * 64 + lane];
        v += red[(3 * NTILES + t) * 64 + lane];
        return v;
      };
  #pragma unroll
      for (int rw = 0; rw < RW; ++rw)
  #pragma unroll
        for (int mw = 0; mw < MW; ++mw)
  #pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int t = (rw * MW + mw) * NT + nt;
            if ((t & 3) != wave) continue;
            store_tile(post, rw, mw, nt, total(t), total((NPX - 1) * NT1 + t));
          }
    }
  };
  if (MODE == 1) finish([](float t) { return t; });
  else switch (act) {
      case 1: finish([](float t) { return lrelu02(t); }); break;
      case 2: finish([post_slope](float t) { return t > 0.f ? t : post_slope * t; }); break;
      case 3: finish([](float t) { return fmaxf(t, 0.f); }); break;
      case 4: finish([](float t) { return tanhf(t) * 10.f; }); break;
      default: finish([](float t) { return t; }); break;
    }
#endif
}

}  // namespace drba_conv

using namespace drba_conv;

namespace {

// ------------------------------------------------------------------------------------------ cfg tables
//                 M  S  RW MW NT CK KS
using C0 = ConvCfg<0, 1, 2, 4, 2, 8, 1>;   // 8x64 px x 32 cout
using C1 = ConvCfg<0, 1, 1, 4, 2, 8, 1>;   // 4x64 px x 32 cout
using C2 = ConvCfg<0, 1, 1, 2, 2, 8, 1>;   // 4x32 px x 32 cout
using C3 = ConvCfg<0, 1, 1, 2, 4, 8, 1>;   // 4x32 px x 64 cout
using C4 = ConvCfg<0, 1, 2, 4, 1, 8, 1>;   // 8x64 px x 16 cout
using C5 = ConvCfg<0, 1, 1, 2, 2, 8, 4>;   // 1x32 px x 32 cout, split-K
using C6 = ConvCfg<0, 1, 2, 2, 2, 8, 4>;   // 2x32 px x 32 cout, split-K
using C7 = ConvCfg<0, 1, 1, 2, 1, 8, 4>;   // 1x32 px x 16 cout, split-K
using C8 = ConvCfg<0, 2, 2, 2, 1, 4, 1>;   // 8x32 px x 16 cout, stride 2
using C9 = ConvCfg<0, 2, 2, 2, 2, 4, 1>;   // 8x32 px x 32 cout, stride 2
using C10 = ConvCfg<0, 2, 1, 2, 2, 4, 1>;  // 4x32 px x 32 cout, stride 2
using C11 = ConvCfg<0, 2, 1, 2, 2, 4, 4>;  // 1x32 px x 32 cout, stride 2, split-K
using C12 = ConvCfg<0, 2, 1, 2, 1, 4, 4>;  // 1x32 px x 16 cout, stride 2, split-K
using C13 = ConvCfg<0, 2, 2, 2, 2, 4, 4>;  // 2x32 px x 32 cout, stride 2, split-K
constexpr int kNumConvCfg = 14;

using D0 = ConvCfg<1, 1, 2, 2, 4, 8, 1>;  // 8x32 px x 64 cout (52 used)
using D1 = ConvCfg<1, 1, 2, 4, 1, 8, 1>;  // 8x64 px x 16 cout (encode.cnn3)
using D2 = ConvCfg<1, 1, 1, 2, 4, 8, 1>;  // 4x32 px x 64 cout
using D3 = ConvCfg<1, 1, 1, 2, 2, 8, 4>;  // 1x32 px x 32 cout, split-K (small maps)
using D4 = ConvCfg<1, 1, 2, 2, 2, 8, 4>;  // 2x32 px x 32 cout, split-K
using D5 = ConvCfg<1, 1, 1, 4, 2, 8, 1>;  // 4x64 px x 32 cout
constexpr int kNumDeconvCfg = 6;

struct CfgInfo {
  int S, TH, TW, NTC, NT, CK, KS, RW, MW, NTAP, NPX, per_chunk, lds_bytes;  // per_chunk: packed floats per (cout tile[, phase], chunk)
};
template <class C>
constexpr CfgInfo cfg_info() {
  return {C::S, C::TH, C::TW, C::NTC, C::NT, C::CK, C::KS, C::RW, C::MW, C::NTAP, C::NPX, C::FRAG, C::LDS_FLOATS * 4};
}
const CfgInfo kConv[kNumConvCfg] = {cfg_info<C0>(), cfg_info<C1>(), cfg_info<C2>(), cfg_info<C3>(), cfg_info<C4>(),
                                    cfg_info<C5>(), cfg_info<C6>(), cfg_info<C7>(), cfg_info<C8>(), cfg_info<C9>(),
                                    cfg_info<C10>(), cfg_info<C11>(), cfg_info<C12>(), cfg_info<C13>()};
const CfgInfo kDeconv[kNumDeconvCfg] = {cfg_info<D0>(), cfg_info<D1>(), cfg_info<D2>(),
                                        cfg_info<D3>(), cfg_info<D4>(), cfg_info<D5>()};

// Raises the kernel's dynamic-LDS limit past the default 64 KB (gfx950: 160 KB per CU); once per kernel instantiation and device.
template <class Cfg, bool PRE>
hipError_t lds_limit() {
  if (Cfg::LDS_FLOATS * sizeof(float) <= 64 * 1024) return hipSuccess;
  return max_dynamic_lds(reinterpret_cast<const void *>(conv_mfma<Cfg, PRE>), Cfg::LDS_FLOATS * (int)sizeof(float));
}

template <class Cfg>
int launch(const float *in, const float *wpk, const float *bias, const float *beta, const float *res, const float *res2,
           float *out, int N, int Cin, int H, int W, int Cout, int Ho, int Wo, int act, float post_slope, int pre_act,
           float pre_slope, int ps, hipStream_t s) {
  const int n_ct = (Cout + Cfg::NTC - 1) / Cfg::NTC;
  const int gh = Cfg::MODE == 0 ? Ho : H, gw = Cfg::MODE == 0 ? Wo : W;
  dim3 g((gw + Cfg::TW - 1) / Cfg::TW, (gh + Cfg::TH - 1) / Cfg::TH, N * n_ct * (Cfg::MODE == 0 ? 1 : 2));  // x2: row phases
  // PRE: PReLU pre-activation compiled into the MFMA loop (FeatureNet / MetricNet / GridNet convolutions)
  auto go = [&](auto kernel, hipError_t lds_ok) -> int {
    if (lds_ok != hipSuccess) return DRBA_ELAUNCH;
    DRBA_LAUNCH(kernel, g, dim3(256), Cfg::LDS_FLOATS * sizeof(float), s, in, wpk, bias, beta, res, res2, out, Cin, H,
                      W, Cout, Ho, Wo, act, post_slope, pre_slope, n_ct, ps);
    return DRBA_OK;
  };
  const int rc = pre_act ? go(conv_mfma<Cfg, true>, lds_limit<Cfg, true>()) : go(conv_mfma<Cfg, false>, lds_limit<Cfg, false>());
  if (rc != DRBA_OK) return rc;
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

// Config choice by a small cost model (cycles): a workgroup's serial time is its MFMA issue time
// (32 cycles per 16x16x4 fp32 MFMA per SIMD) plus a fixed cost per staged chunk; workgroups run
// `bpc` per CU over 256 CUs and co-resident workgroups share the CU's MFMA pipes.  Small maps
// therefore pick the split-K configs, large maps the big-tile ones.
int pick(const CfgInfo *tab, int ntab, int stride, int Cin, int Cout, int gh, int gw, int zmul) {
  int best = -1;
  double best_t = 0;
  for (int id = 0; id < ntab; ++id) {
    const CfgInfo &c = tab[id];
    if (c.S != stride) continue;
    const int nch = (Cin + c.CK - 1) / c.CK;
    const double blocks =
        (double)((gw + c.TW - 1) / c.TW) * ((gh + c.TH - 1) / c.TH) * ((Cout + c.NTC - 1) / c.NTC) * zmul;
    const double cpw = (c.KS == 1) ? nch : (nch + 3) / 4;  // chunks per wave
    const double mfma_cyc = c.NPX * c.RW * c.MW * c.NT * (double)c.NTAP * (c.CK / 4) * cpw * 32.0;
    const double fixed = cpw * 900.0 + (c.KS == 4 ? 1500.0 : 0.0) + 2500.0;
    int bpc = 160 * 1024 / (c.lds_bytes > 0 ? c.lds_bytes : 1);
    if (bpc > 3) bpc = 3;
    if (bpc < 1) bpc = 1;
    const double slots = 256.0 * bpc;
    const double rounds = blocks <= slots ? 1.0 : blocks / slots;
    const double share = blocks <= 256.0 ? 1.0 : (blocks >= slots ? (double)bpc : blocks / 256.0);
    const double t = rounds * (mfma_cyc * share + fixed);
    if (best < 0 || t < best_t) {
      best = id;
      best_t = t;
    }
  }
  return best < 0 ? DRBA_EUNSUPPORTED : best;
}

size_t packed_floats(const CfgInfo &c, int Cin, int Cout, int phases) {
  const size_t n_ct = (Cout + c.NTC - 1) / c.NTC, nch = (Cin + c.CK - 1) / c.CK;
  return n_ct * phases * nch * c.per_chunk;
}

int env_override(const char *name, int ntab) {  // (compiled out of the release build: common.hpp env_int)
  const int id = env_int(name, -1);
  return (id >= 0 && id < ntab) ? id : -1;
}

}  // namespace

extern "C" {

// cfg ids: 0..kNumConvCfg-1 the fp32 MFMA table above, then the split-bf16 family of conv_split.hip (stride 1, Cin a
// multiple of 32; drba_conv3x3_packed_floats returns 0 for a layer a config cannot run).  DRBA_CONV_SPLIT=0 hides it (TUNING builds only).
static bool split_enabled() {
  static const bool on = env_int("DRBA_CONV_SPLIT", 1) != 0;
  return on;
}
// ... then the LDS-DMA family of conv_dma.hip (same arithmetic, same constraints plus W % 4 == 0)
static int first_dma_cfg() { return kNumConvCfg + (split_enabled() ? conv_split_num_cfgs() : 0); }
// ... then the K-split family of conv_ks.hip (Cin = 64 / 96 / 128 / 192)
static int first_ks_cfg() { return first_dma_cfg() + (split_enabled() ? conv_dma_num_cfgs() : 0); }
// ... then the two-term fp16 family (family 4; conv_split.hip "Two-term form"): 22-bit operands instead of 24, half the
// matrix-core work.  A caller opts in by offering these ids to its tuner; none of the ids before them changes meaning.
static int first_f16_cfg() { return first_ks_cfg() + (split_enabled() ? conv_ks_num_cfgs() : 0); }
static int first_f16_dma_cfg() { return first_f16_cfg() + (split_enabled() ? conv_split_num_cfgs() : 0); }
static int first_f16_ks_cfg() { return first_f16_dma_cfg() + (split_enabled() ? conv_dma_num_cfgs() : 0); }
// ... and its stride-2 tiles (family 4, stride 2: conv_split.hip MODE 2)
static int first_f16_s2_cfg() { return first_f16_ks_cfg() + (split_enabled() ? conv_ks_num_cfgs() : 0); }
// ... and (round 6) its stride-1 tiles whose waves split rows and couts (conv_split.hip CS = 2): appended, family 4, stride 1
static int first_f16_cs_cfg() { return first_f16_s2_cfg() + (split_enabled() ? conv_split_s2_num_cfgs() : 0); }
int drba_conv3x3_num_cfgs(void) { return first_f16_cs_cfg() + (split_enabled() ? conv_split_cs_num_cfgs() : 0); }
int drba_conv3x3_cfg_stride(int cfg) {
  if (cfg >= first_f16_cs_cfg() && cfg < drba_conv3x3_num_cfgs()) return conv_split_cfg_stride(cfg - first_f16_cs_cfg() + conv_split_cs_first());
  if (cfg >= first_f16_s2_cfg() && cfg < first_f16_cs_cfg()) return 2;
  if (cfg >= kNumConvCfg && cfg < drba_conv3x3_num_cfgs()) return 1;
  return (cfg < 0 || cfg >= kNumConvCfg) ? DRBA_EINVAL : kConv[cfg].S;
}
int drba_conv3x3_cfg_family(int cfg) {
  if (cfg < 0 || cfg >= drba_conv3x3_num_cfgs()) return DRBA_EINVAL;
  return cfg < kNumConvCfg ? 0 : (cfg < first_dma_cfg() ? 1 : (cfg < first_ks_cfg() ? 2 : (cfg < first_f16_cfg() ? 3 : 4)));
}
int drba_deconv4x4_num_cfgs(void) { return kNumDeconvCfg + (split_enabled() ? deconv_split_total_cfgs() : 0); }
int drba_deconv4x4_cfg_family(int cfg) {
  if (cfg < 0 || cfg >= drba_deconv4x4_num_cfgs()) return DRBA_EINVAL;
  return cfg < kNumDeconvCfg ? 0 : (cfg - kNumDeconvCfg < deconv_split_f16_first() ? 1 : 4);
}

// DRBA_CONV_CFG=<id> / DRBA_DECONV_CFG=<id> in the environment override the choice (experiments only).
int drba_conv3x3_pick_cfg(int Cin, int Cout, int Ho, int Wo, int stride) {
  if (stride != 1 && stride != 2) return DRBA_EUNSUPPORTED;
  if (Cin <= 0 || Cout <= 0 || Ho <= 0 || Wo <= 0) return DRBA_EINVAL;
  const int ov = env_override("DRBA_CONV_CFG", kNumConvCfg);
  if (ov >= 0 && kConv[ov].S == stride) return ov;
  return pick(kConv, kNumConvCfg, stride, Cin, Cout, Ho, Wo, 1);
}

size_t drba_conv3x3_packed_floats(int Cin, int Cout, int cfg) {
  if (cfg >= first_f16_cs_cfg())
    return cfg < drba_conv3x3_num_cfgs() ? conv_split_packed_floats(Cin, Cout, cfg - first_f16_cs_cfg() + conv_split_cs_first()) : 0;
  if (cfg >= first_f16_s2_cfg())
    return conv_split_packed_floats(Cin, Cout, cfg - first_f16_s2_cfg() + conv_split_s2_first());
  if (cfg >= first_f16_ks_cfg()) return conv_ks_packed_floats(Cin, Cout, cfg - first_f16_ks_cfg() + conv_ks_f16_first());
  if (cfg >= first_f16_dma_cfg()) return conv_dma_packed_floats(Cin, Cout, cfg - first_f16_dma_cfg() + conv_dma_f16_first());
  if (cfg >= first_f16_cfg()) return conv_split_packed_floats(Cin, Cout, cfg - first_f16_cfg() + conv_split_f16_first());
  if (cfg >= first_ks_cfg()) return conv_ks_packed_floats(Cin, Cout, cfg - first_ks_cfg());
  if (cfg >= first_dma_cfg()) return conv_dma_packed_floats(Cin, Cout, cfg - first_dma_cfg());
  if (cfg >= kNumConvCfg) return conv_split_packed_floats(Cin, Cout, cfg - kNumConvCfg);
  if (cfg < 0 || cfg >= kNumConvCfg || Cin <= 0 || Cout <= 0) return 0;
  return packed_floats(kConv[cfg], Cin, Cout, 1);
}

// fragment order: packed[(((cz*nchunks + q)*9 + tap)*CG + cg)*NT + nt][lane] =
//   w[cz*NTC + nt*16 + (lane&15)][q*CK + cg*4 + (lane>>4)][tap], zero outside Cout/Cin
int drba_conv3x3_pack(const float *w, float *packed, int Cin, int Cout, int cfg) {
  if (cfg >= first_f16_cs_cfg() && cfg < drba_conv3x3_num_cfgs())
    return conv_split_pack(w, packed, Cin, Cout, cfg - first_f16_cs_cfg() + conv_split_cs_first());
  if (cfg >= first_f16_s2_cfg() && cfg < first_f16_cs_cfg())
    return conv_split_pack(w, packed, Cin, Cout, cfg - first_f16_s2_cfg() + conv_split_s2_first());
  if (cfg >= first_f16_ks_cfg() && cfg < first_f16_s2_cfg())
    return conv_ks_pack(w, packed, Cin, Cout, cfg - first_f16_ks_cfg() + conv_ks_f16_first());
  if (cfg >= first_f16_dma_cfg() && cfg < first_f16_ks_cfg())
    return conv_dma_pack(w, packed, Cin, Cout, cfg - first_f16_dma_cfg() + conv_dma_f16_first());
  if (cfg >= first_f16_cfg() && cfg < first_f16_dma_cfg())
    return conv_split_pack(w, packed, Cin, Cout, cfg - first_f16_cfg() + conv_split_f16_first());
  if (cfg >= first_ks_cfg() && cfg < first_f16_cfg()) return conv_ks_pack(w, packed, Cin, Cout, cfg - first_ks_cfg());
  if (cfg >= first_dma_cfg() && cfg < first_ks_cfg()) return conv_dma_pack(w, packed, Cin, Cout, cfg - first_dma_cfg());
  if (cfg >= kNumConvCfg && cfg < first_dma_cfg()) return conv_split_pack(w, packed, Cin, Cout, cfg - kNumConvCfg);
  if (!w || !packed || cfg < 0 || cfg >= kNumConvCfg || Cin <= 0 || Cout <= 0) return DRBA_EINVAL;
  const CfgInfo &c = kConv[cfg];
  const int n_ct = (Cout + c.NTC - 1) / c.NTC, nch = (Cin + c.CK - 1) / c.CK, CG = c.CK / 4;
  memset(packed, 0, sizeof(float) * packed_floats(c, Cin, Cout, 1));
  for (int cz = 0; cz < n_ct; ++cz)
    for (int q = 0; q < nch; ++q)
      for (int tap = 0; tap < 9; ++tap)
        for (int cg = 0; cg < CG; ++cg)
          for (int nt = 0; nt < c.NT; ++nt) {
            float *frag = packed + (((((size_t)cz * nch + q) * 9 + tap) * CG + cg) * c.NT + nt) * 64;
            for (int lane = 0; lane < 64; ++lane) {
              const int co = cz * c.NTC + nt * 16 + (lane & 15), ci = q * c.CK + cg * 4 + (lane >> 4);
              if (co < Cout && ci < Cin) frag[lane] = w[((size_t)co * Cin + ci) * 9 + tap];
            }
          }
  return DRBA_OK;
}

int drba_conv3x3(const float *in, const float *packed_w, const float *bias, const float *beta, const float *residual,
                 const float *residual2, float *out, int N, int Cin, int H, int W, int Cout, int stride, int act,
                 float post_slope, int pre_act, float pre_slope, int cfg, void *stream) {
  if (!in || !packed_w || !out || N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return DRBA_EINVAL;
  if (beta && !residual) return DRBA_EINVAL;
  if (residual2 && !residual) return DRBA_EINVAL;
  if (act < 0 || act > 4) return DRBA_EINVAL;
  // (family 4: with drba_set_range_check on, the output is scanned for the inf / NaN an fp16 overflow of an operand leaves)
  const size_t n_out = (size_t)N * Cout * ((H - 1) / (stride > 0 ? stride : 1) + 1) * ((W - 1) / (stride > 0 ? stride : 1) + 1);
  if (cfg >= first_f16_cs_cfg() && cfg < drba_conv3x3_num_cfgs()) {
    if (stride != conv_split_cfg_stride(cfg - first_f16_cs_cfg() + conv_split_cs_first())) return DRBA_EINVAL;
    return range_checked(conv_split_launch(cfg - first_f16_cs_cfg() + conv_split_cs_first(), in, packed_w, bias, beta, residual, residual2, out, N,
                                           Cin, H, W, Cout, act, post_slope, pre_act, pre_slope, stream), out, n_out, stream);
  }
  if (cfg >= first_f16_s2_cfg() && cfg < first_f16_cs_cfg()) {
    if (stride != 2) return DRBA_EINVAL;
    return range_checked(conv_split_launch(cfg - first_f16_s2_cfg() + conv_split_s2_first(), in, packed_w, bias, beta, residual, residual2, out, N,
                                           Cin, H, W, Cout, act, post_slope, pre_act, pre_slope, stream), out, n_out, stream);
  }
  if (cfg >= first_f16_ks_cfg() && cfg < first_f16_s2_cfg()) {
    if (stride != 1) return DRBA_EINVAL;
    return range_checked(conv_ks_launch(cfg - first_f16_ks_cfg() + conv_ks_f16_first(), in, packed_w, bias, beta, residual, residual2, out, N, Cin,
                                        H, W, Cout, act, post_slope, pre_act, pre_slope, stream), out, n_out, stream);
  }
  if (cfg >= first_f16_dma_cfg() && cfg < first_f16_ks_cfg()) {
    if (stride != 1) return DRBA_EINVAL;
    return range_checked(conv_dma_launch(cfg - first_f16_dma_cfg() + conv_dma_f16_first(), in, packed_w, bias, beta, residual, residual2, out, N,
                                         Cin, H, W, Cout, act, post_slope, pre_act, pre_slope, stream), out, n_out, stream);
  }
  if (cfg >= first_f16_cfg() && cfg < first_f16_dma_cfg()) {
    if (stride != 1) return DRBA_EINVAL;
    return range_checked(conv_split_launch(cfg - first_f16_cfg() + conv_split_f16_first(), in, packed_w, bias, beta, residual, residual2, out,
                                           N, Cin, H, W, Cout, act, post_slope, pre_act, pre_slope, stream), out, n_out, stream);
  }
  if (cfg >= first_ks_cfg() && cfg < first_f16_cfg()) {
    if (stride != 1) return DRBA_EINVAL;
    return conv_ks_launch(cfg - first_ks_cfg(), in, packed_w, bias, beta, residual, residual2, out, N, Cin, H, W, Cout, act,
                          post_slope, pre_act, pre_slope, stream);
  }
  if (cfg >= first_dma_cfg() && cfg < first_ks_cfg()) {
    if (stride != 1) return DRBA_EINVAL;
    return conv_dma_launch(cfg - first_dma_cfg(), in, packed_w, bias, beta, residual, residual2, out, N, Cin, H, W, Cout, act,
                           post_slope, pre_act, pre_slope, stream);
  }
  if (cfg >= kNumConvCfg && cfg < first_dma_cfg()) {
    if (stride != 1) return DRBA_EINVAL;
    return conv_split_launch(cfg - kNumConvCfg, in, packed_w, bias, beta, residual, residual2, out, N, Cin, H, W, Cout, act,
                             post_slope, pre_act, pre_slope, stream);
  }
  if (cfg < 0 || cfg >= kNumConvCfg || kConv[cfg].S != stride) return DRBA_EINVAL;
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  hipStream_t s = (hipStream_t)stream;
#define DRBA_CASE(ID, T) \
  case ID:               \
    return launch<T>(in, packed_w, bias, beta, residual, residual2, out, N, Cin, H, W, Cout, Ho, Wo, act, post_slope, \
                     pre_act, pre_slope, 0, s);
  switch (cfg) {
    DRBA_CASE(0, C0)
    DRBA_CASE(1, C1)
    DRBA_CASE(2, C2)
    DRBA_CASE(3, C3)
    DRBA_CASE(4, C4)
    DRBA_CASE(5, C5)
    DRBA_CASE(6, C6)
    DRBA_CASE(7, C7)
    DRBA_CASE(8, C8)
    DRBA_CASE(9, C9)
    DRBA_CASE(10, C10)
    DRBA_CASE(11, C11)
    DRBA_CASE(12, C12)
    DRBA_CASE(13, C13)
  }
#undef DRBA_CASE
  return DRBA_EUNSUPPORTED;
}

int drba_deconv4x4_pick_cfg(int Cin, int Cout, int H, int W) {
  if (Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return DRBA_EINVAL;
  const int ov = env_override("DRBA_DECONV_CFG", kNumDeconvCfg);
  if (ov >= 0) return ov;
  return pick(kDeconv, kNumDeconvCfg, 1, Cin, Cout, H, W, 2);  // two row-phase workgroups per tile
}

size_t drba_deconv4x4_packed_floats(int Cin, int Cout, int cfg) {
  if (cfg >= kNumDeconvCfg)
    return cfg < drba_deconv4x4_num_cfgs() ? deconv_split_packed_floats(Cin, Cout, cfg - kNumDeconvCfg) : 0;
  if (cfg < 0 || cfg >= kNumDeconvCfg || Cin <= 0 || Cout <= 0) return 0;
  return packed_floats(kDeconv[cfg], Cin, Cout, 4);
}

// w: [Cin, Cout, 4, 4].  packed[((((cz*4 + phase)*nchunks + q)*4 + tap)*CG + cg)*NT + nt][lane], tap = 2a+b,
// ky = py ? (a ? 2 : 0) : (a ? 3 : 1), kx likewise from (px, b).
int drba_deconv4x4_pack(const float *w, float *packed, int Cin, int Cout, int cfg) {
  if (cfg >= kNumDeconvCfg && cfg < drba_deconv4x4_num_cfgs())
    return deconv_split_pack(w, packed, Cin, Cout, cfg - kNumDeconvCfg);
  if (!w || !packed || cfg < 0 || cfg >= kNumDeconvCfg || Cin <= 0 || Cout <= 0) return DRBA_EINVAL;
  const CfgInfo &c = kDeconv[cfg];
  const int n_ct = (Cout + c.NTC - 1) / c.NTC, nch = (Cin + c.CK - 1) / c.CK, CG = c.CK / 4;
  memset(packed, 0, sizeof(float) * packed_floats(c, Cin, Cout, 4));
  for (int cz = 0; cz < n_ct; ++cz)
    for (int phase = 0; phase < 4; ++phase) {
      const int py = phase >> 1, px = phase & 1;
      for (int q = 0; q < nch; ++q)
        for (int tap = 0; tap < 4; ++tap) {
          const int a = tap >> 1, b = tap & 1;
          const int ky = py ? (a ? 2 : 0) : (a ? 3 : 1);
          const int kx = px ? (b ? 2 : 0) : (b ? 3 : 1);
          for (int cg = 0; cg < CG; ++cg)
            for (int nt = 0; nt < c.NT; ++nt) {
              float *frag = packed + ((((((size_t)cz * 4 + phase) * nch + q) * 4 + tap) * CG + cg) * c.NT + nt) * 64;
              for (int lane = 0; lane < 64; ++lane) {
                const int co = cz * c.NTC + nt * 16 + (lane & 15), ci = q * c.CK + cg * 4 + (lane >> 4);
                if (co < Cout && ci < Cin) frag[lane] = w[(((size_t)ci * Cout + co) * 4 + ky) * 4 + kx];
              }
            }
        }
    }
  return DRBA_OK;
}

int drba_deconv4x4s2(const float *in, const float *packed_w, const float *bias, float *out, int N, int Cin, int H,
                     int W, int Cout, int pixel_shuffle, int pre_act, float pre_slope, int cfg, void *stream) {
  if (!in || !packed_w || !out || N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return DRBA_EINVAL;
  if (pixel_shuffle && (Cout & 3)) return DRBA_EINVAL;
  if (cfg >= kNumDeconvCfg && cfg < drba_deconv4x4_num_cfgs()) {
    const int rc = deconv_split_launch(cfg - kNumDeconvCfg, in, packed_w, bias, out, N, Cin, H, W, Cout, pixel_shuffle, pre_act,
                                       pre_slope, stream);
    return drba_deconv4x4_cfg_family(cfg) == 4 ? range_checked(rc, out, (size_t)N * Cout * 4 * H * W, stream) : rc;
  }
  if (cfg < 0 || cfg >= kNumDeconvCfg) return DRBA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
#define DRBA_CASE(ID, T) \
  case ID:               \
    return launch<T>(in, packed_w, bias, nullptr, nullptr, nullptr, out, N, Cin, H, W, Cout, 2 * H, 2 * W, 0, 0.f,     \
                     pre_act, pre_slope, pixel_shuffle, s);
  switch (cfg) {
    DRBA_CASE(0, D0)
    DRBA_CASE(1, D1)
    DRBA_CASE(2, D2)
    DRBA_CASE(3, D3)
    DRBA_CASE(4, D4)
    DRBA_CASE(5, D5)
  }
#undef DRBA_CASE
  return DRBA_EUNSUPPORTED;
}

// drba_conv3x3 (stride 1, bias, activation; no residual operands, no pre-activation) storing through PixelShuffle(2):
// out is [N, Cout / 4, 2H, 2W].  Only configurations whose tile carries that store form accept (two-term 4 x 32 x 64 tiles of
// conv_split.hip); every other id returns DRBA_EUNSUPPORTED, and the caller runs drba_conv3x3 + drba_pixel_shuffle2 instead.
int drba_conv3x3_shuffle(const float *in, const float *packed_w, const float *bias, float *out, int N, int Cin, int H, int W,
                         int Cout, int act, float post_slope, int cfg, void *stream) {
  if (!in || !packed_w || !out || N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return DRBA_EINVAL;
  if (act < 0 || act > 4) return DRBA_EINVAL;
  if ((Cout & 3) || (W & 3)) return DRBA_EUNSUPPORTED;
  const size_t n_out = (size_t)N * Cout * H * W;
  int id = -1;
  if (cfg >= first_f16_cs_cfg() && cfg < drba_conv3x3_num_cfgs()) id = cfg - first_f16_cs_cfg() + conv_split_cs_first();
  else if (cfg >= first_f16_cfg() && cfg < first_f16_dma_cfg()) id = cfg - first_f16_cfg() + conv_split_f16_first();
  if (id < 0 || conv_split_cfg_stride(id) != 1) return DRBA_EUNSUPPORTED;
  return range_checked(conv_split_launch(id, in, packed_w, bias, nullptr, nullptr, nullptr, out, N, Cin, H, W, Cout, act, post_slope, 0,
                                         0.f, stream, 1), out, n_out, stream);
}

// A chain of convolutions launched from one call (the IFBlock cores and the context encoder are 4..11 dependent
// layers; issuing them from C++ costs a few microseconds per launch instead of a Python/ctypes round trip each, which
// is what bounds the step once the GPU side is below 4 ms).  Layer i reads the previous layer's output (layer 0 reads
// `in`), writes `out` if it is the last one and otherwise alternates between scratch0 / scratch1; `residual` adds the
// layer's own input in the epilogue (ResConv).  Each scratch buffer must hold the largest intermediate tensor.
int drba_conv_chain(const float *in, float *out, float *scratch0, float *scratch1, const drba_conv_layer_t *layers,
                    int n_layers, int N, int H, int W, void *stream) {
  if (!in || !out || !layers || n_layers <= 0 || N <= 0 || H <= 0 || W <= 0) return DRBA_EINVAL;
  if (n_layers > 1 && (!scratch0 || !scratch1)) return DRBA_EINVAL;
  float *buf[2] = {scratch0, scratch1};
  const float *src = in;
  int h = H, w = W;
  for (int i = 0; i < n_layers; ++i) {
    const drba_conv_layer_t &L = layers[i];
    float *dst = (i == n_layers - 1) ? out : buf[i & 1];
    int rc;
    if (L.deconv) {
      rc = drba_deconv4x4s2(src, L.packed_w, L.bias, dst, N, L.cin, h, w, L.cout, L.pixel_shuffle, 0, 0.f, L.cfg, stream);
      h *= 2;
      w *= 2;
    } else {
      rc = drba_conv3x3(src, L.packed_w, L.bias, L.beta, L.residual ? src : nullptr, nullptr, dst, N, L.cin, h, w, L.cout,
                        L.stride, L.act, 0.f, 0, 0.f, L.cfg, stream);
      h = (h - 1) / L.stride + 1;
      w = (w - 1) / L.stride + 1;
    }
    if (rc != DRBA_OK) return rc;
    src = dst;
  }
  return DRBA_OK;
}

}  // extern "C"
