// GMFlow operators that are not plain GEMMs (models/gmflow/*): direct k x k convolution for the 7x7 / 1x1
// layers, InstanceNorm, LayerNorm, GELU, masked row softmax, the fused correlation-softmax -> flow
// expectations (global, local), local-window flow propagation, convex upsampling, flow_warp and the
// align_corners=True resize.  The q/k/v/merge/MLP projections and the QK^T / PV products are plain GEMMs and go
// through the vendor BLAS from the host (allowed for plain library GEMMs); everything around them is here.
#include "common.hpp"

using namespace drba;

namespace {

__device__ __forceinline__ float wave_sum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
  return v;
}
// block-wide reductions over 256 threads (4 waves); result broadcast to all threads
__device__ __forceinline__ float block_sum256(float v, float *red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max256(float v, float *red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// ------------------------------------------------------------------------------------------ direct conv
// out[n,co,y,x] = b[co] + sum_{ci,ky,kx} in[n,ci,y*s-p+ky,x*s-p+kx] * w[co,ci,ky,kx]   (zero padding)
// One lane per output pixel and kCob output channels: an input tap is loaded once and feeds kCob FMAs whose weights
// are wave-uniform (scalar loads).  Used for the 7x7 stem and the 1x1 projections (small share of GMFlow's FLOPs).
constexpr int kCob = 16;
__global__ void __launch_bounds__(256)
conv_direct_kernel(const float *__restrict__ in, const float *__restrict__ w, const float *__restrict__ bias,
                   float *__restrict__ out, int N, int Cin, int H, int W, int Cout, int Ho, int Wo, int K, int S, int P,
                   int cblocks) {
  const int n = blockIdx.y / cblocks, co0 = (blockIdx.y - n * cblocks) * kCob;
  const Tile2D tp = tile_pixel(Wo, Ho);
  if (!tp.valid) return;
  const int x = tp.x, y = tp.y;
  float acc[kCob];
#pragma unroll
  for (int j = 0; j < kCob; ++j) acc[j] = (bias && co0 + j < Cout) ? bias[co0 + j] : 0.f;
  const float *ip = in + (size_t)n * Cin * H * W;
  const size_t wstride = (size_t)Cin * K * K;
  const float *wp = w + (size_t)co0 * wstride;
  for (int ci = 0; ci < Cin; ++ci)
    for (int ky = 0; ky < K; ++ky) {
      const int gy = y * S - P + ky;
      const bool oky = gy >= 0 && gy < H;
      for (int kx = 0; kx < K; ++kx) {
        const int gx = x * S - P + kx;
        const float v = (oky && gx >= 0 && gx < W) ? ip[((size_t)ci * H + gy) * W + gx] : 0.f;
        const float *wt = wp + (ci * K + ky) * K + kx;
#pragma unroll
        for (int j = 0; j < kCob; ++j)
          if (co0 + j < Cout) acc[j] += v * wt[(size_t)j * wstride];
      }
    }
#pragma unroll
  for (int j = 0; j < kCob; ++j)
    if (co0 + j < Cout) out[(((size_t)n * Cout + co0 + j) * Ho + y) * Wo + x] = acc[j];
}

// ------------------------------------------------------------------------------------------ norms / pointwise
// nn.InstanceNorm2d (eps 1e-5, no affine).  A plane is split into kInChunks chunks so that the whole chip works on
// it: pass 1 reduces every chunk to (mean, M2) with the two-pass formula inside the chunk, pass 2 merges the chunk
// statistics of its plane (Chan's parallel update, exact mean/variance algebra) and normalises its chunk.
// Two reads + one write of the tensor instead of three reads by one workgroup per plane.
constexpr int kInChunks = 32;
__global__ void __launch_bounds__(256)
instance_norm_stats_kernel(const float *__restrict__ in, float *__restrict__ part, size_t HW) {
  __shared__ float red[4];
  const size_t clen = (HW + kInChunks - 1) / kInChunks;
  const size_t lo = (size_t)blockIdx.x * clen, hi = min(lo + clen, HW);
  const float *p = in + (size_t)blockIdx.y * HW;
  float s = 0.f;
  for (size_t i = lo + threadIdx.x; i < hi; i += 256) s += p[i];
  const float cntf = hi > lo ? (float)(hi - lo) : 0.f;
  const float mean = cntf > 0.f ? block_sum256(s, red) / cntf : 0.f;
  float v = 0.f;
  for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
    const float d = p[i] - mean;
    v += d * d;
  }
  const float m2 = block_sum256(v, red);
  if (threadIdx.x == 0) {
    float *o = part + ((size_t)blockIdx.y * kInChunks + blockIdx.x) * 2;
    o[0] = mean;
    o[1] = m2;
  }
}

__global__ void __launch_bounds__(256)
instance_norm_apply_kernel(const float *__restrict__ in, const float *__restrict__ part, float *__restrict__ out, size_t HW,
                           float eps, int relu) {
  const size_t clen = (HW + kInChunks - 1) / kInChunks;
  // merge the plane's chunk statistics (every thread does the same 32-step merge: cheaper than a reduction + barrier)
  const float *pp = part + (size_t)blockIdx.y * kInChunks * 2;
  float n = 0.f, mean = 0.f, m2 = 0.f;
  for (int c = 0; c < kInChunks; ++c) {
    const size_t lo_c = (size_t)c * clen, hi_c = min(lo_c + clen, HW);
    if (hi_c <= lo_c) break;
    const float nb = (float)(hi_c - lo_c), mb = pp[2 * c], m2b = pp[2 * c + 1];
    const float d = mb - mean, nn = n + nb;
    mean += d * (nb / nn);
    m2 += m2b + d * d * (n * nb / nn);
    n = nn;
  }
  const float inv = 1.f / sqrtf(m2 / (float)HW + eps);
  const size_t lo = (size_t)blockIdx.x * clen, hi = min(lo + clen, HW);
  const float *p = in + (size_t)blockIdx.y * HW;
  float *o = out + (size_t)blockIdx.y * HW;
  for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
    const float y = (p[i] - mean) * inv;
    o[i] = relu ? fmaxf(y, 0.f) : y;
  }
}

__global__ void __launch_bounds__(256)
add_act_kernel(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ out, size_t n, int relu) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = a[i] + b[i];
    out[i] = relu ? fmaxf(v, 0.f) : v;
  }
}

// (x - mean[c]) / std[c] over [N, C, HW]
__global__ void __launch_bounds__(256)
channel_affine_kernel(const float *__restrict__ in, float *__restrict__ out, int C, size_t HW, float m0, float m1,
                      float m2, float s0, float s1, float s2, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)((i / HW) % C);
    const float m = c == 0 ? m0 : c == 1 ? m1 : m2, s = c == 0 ? s0 : c == 1 ? s1 : s2;
    out[i] = (in[i] - m) / s;
  }
}

// nn.LayerNorm(cols) per row, optional residual: out = (res ? res : 0) + LN(x)*w + b.  One wave per row.
__global__ void __launch_bounds__(256)
layernorm_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ b,
                 const float *__restrict__ res, float *__restrict__ out, size_t rows, int cols, float eps) {
  const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float *p = x + row * cols;
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) s += p[c];
  s = wave_sum(s);
  const float mean = __shfl(s, 0, 64) / (float)cols;
  float v = 0.f;
  for (int c = lane; c < cols; c += 64) {
    const float d = p[c] - mean;
    v += d * d;
  }
  v = wave_sum(v);
  const float inv = 1.f / sqrtf(__shfl(v, 0, 64) / (float)cols + eps);
  for (int c = lane; c < cols; c += 64) {
    const float y = (p[c] - mean) * inv * w[c] + b[c];
    out[row * cols + c] = res ? res[row * cols + c] + y : y;
  }
}

__global__ void __launch_bounds__(256) gelu_kernel(const float *__restrict__ x, float *__restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    out[i] = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));  // nn.GELU() (erf form)
  }
}

// in-place softmax over each row of x/scale (+ mask[(row / rows_per_mat) % n_masks][row % rows_per_mat][:]).
// The row lives in registers between the single read and the single write: ROWS_PER_BLOCK = 4 -> one wave per row
// (window attention, 540 columns: no barriers, shuffles only), ROWS_PER_BLOCK = 1 -> one workgroup per row (up to
// 256 * EPT columns: the 2160- and 8640-column rows of the 1/8-resolution layers).
template <int EPT, int ROWS_PER_BLOCK>
__global__ void __launch_bounds__(256)
softmax_rows_kernel(float *__restrict__ x, const float *__restrict__ mask, size_t rows, int cols, int rows_per_mat,
                    int n_masks, float scale) {
  __shared__ float red[4];
  constexpr int LANES = 256 / ROWS_PER_BLOCK;  // threads cooperating on one row
  const int sub = threadIdx.x / LANES, t = threadIdx.x - sub * LANES;
  const size_t row = (size_t)blockIdx.x * ROWS_PER_BLOCK + sub;
  const bool live = row < rows;
  float *p = x + (live ? row : 0) * cols;
  const float *m = mask ? mask + (((row / rows_per_mat) % n_masks) * (size_t)rows_per_mat + (row % rows_per_mat)) * cols : nullptr;
  float v[EPT];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < EPT; ++i) {
    const int c = t + i * LANES;
    v[i] = -INFINITY;
    if (live && c < cols) {
      v[i] = p[c] / scale;  // scores / sqrt(c): the reference divides
      if (m) v[i] += m[c];
    }
    mx = fmaxf(mx, v[i]);
  }
  mx = ROWS_PER_BLOCK == 1 ? block_max256(mx, red) : wave_max(mx);
  if (ROWS_PER_BLOCK != 1) mx = __shfl(mx, 0, 64);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < EPT; ++i) {
    v[i] = expf(v[i] - mx);  // exp(-inf) = 0 for the padding slots
    s += v[i];
  }
  s = ROWS_PER_BLOCK == 1 ? block_sum256(s, red) : wave_sum(s);
  if (ROWS_PER_BLOCK != 1) s = __shfl(s, 0, 64);
#pragma unroll
  for (int i = 0; i < EPT; ++i) {
    const int c = t + i * LANES;
    if (live && c < cols) p[c] = v[i] / s;
  }
}

// rows too long for registers: three passes over the row in global memory, one workgroup per row
__global__ void __launch_bounds__(256)
softmax_rows_long_kernel(float *__restrict__ x, const float *__restrict__ mask, int cols, int rows_per_mat, int n_masks,
                         float scale) {
  __shared__ float red[4];
  const size_t row = blockIdx.x;
  float *p = x + row * cols;
  const float *m = mask ? mask + (((row / rows_per_mat) % n_masks) * (size_t)rows_per_mat + (row % rows_per_mat)) * cols : nullptr;
  float mx = -INFINITY;
  for (int c = threadIdx.x; c < cols; c += 256) {
    float v = p[c] / scale;
    if (m) v += m[c];
    p[c] = v;
    mx = fmaxf(mx, v);
  }
  mx = block_max256(mx, red);
  float s = 0.f;
  for (int c = threadIdx.x; c < cols; c += 256) {
    const float e = expf(p[c] - mx);
    p[c] = e;
    s += e;
  }
  s = block_sum256(s, red);
  for (int c = threadIdx.x; c < cols; c += 256) p[c] = p[c] / s;
}

// matching.py:41-89 with local_radius r: for every pixel, correlation of feature0 with feature1 at the (2r+1)^2
// integer offsets (zeros outside the image, those taps get -1e4), softmax, expected offset.
// One lane per pixel with the (2r+1)^2 running dot products in registers: the channel loop reads feature0 once and
// feature1's neighbourhood with loads that coalesce across the lanes of a row (NCHW planes), instead of one wave per
// pixel with its lanes strided over 64 channel planes (2.8 ms -> 0.2 ms at 144x240, 128 channels, r = 4).
// A workgroup owns 64 pixels (a 32 x 2 tile, one per lane) and its 4 waves split the channels: at 144 x 240 one lane
// per pixel alone is 540 waves for 1024 SIMDs, each with a 128-deep dependent loop; four partial dot products per tap,
// summed through LDS by wave 0, put a wave on every SIMD twice over and quarter the loop.
template <int R>
__global__ void __launch_bounds__(256)
local_corr_flow_kernel(const float *__restrict__ f0, const float *__restrict__ f1, float *__restrict__ out, int C, int H,
                       int W, float scale) {
  constexpr int N = 2 * R + 1;
  __shared__ float red[3][N * N][64];
  const size_t P = (size_t)H * W;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tiles_x = (W + 31) / 32;
  const int t = xcd_band(blockIdx.x, gridDim.x);
  const int ty = t / tiles_x, tx = t - ty * tiles_x;
  const int x = tx * 32 + (lane & 31), y = ty * 2 + (lane >> 5);
  const bool valid = x < W && y < H;
  const int xc = min(x, W - 1), yc = min(y, H - 1);  // lanes past the image compute on a valid pixel and do not store
  const size_t p = (size_t)yc * W + xc;
  float acc[N * N];
#pragma unroll
  for (int k = 0; k < N * N; ++k) acc[k] = 0.f;
  // clamped tap addresses: out-of-image taps read a valid pixel and are overridden below
  int offy[N], offx[N];
#pragma unroll
  for (int d = 0; d < N; ++d) {
    offy[d] = min(max(yc + d - R, 0), H - 1) * W;
    offx[d] = min(max(xc + d - R, 0), W - 1);
  }
  const int c0 = (C * wave) / 4, c1 = (C * (wave + 1)) / 4;
  for (int c = c0; c < c1; ++c) {
    const float a = f0[(size_t)c * P + p];
    const float *pl = f1 + (size_t)c * P;
#pragma unroll
    for (int dy = 0; dy < N; ++dy)
#pragma unroll
      for (int dx = 0; dx < N; ++dx) acc[dy * N + dx] += a * pl[offy[dy] + offx[dx]];
  }
  if (wave > 0) {
#pragma unroll
    for (int k = 0; k < N * N; ++k) red[wave - 1][k][lane] = acc[k];
  }
  __syncthreads();
  if (wave > 0 || !valid) return;
#pragma unroll
  for (int k = 0; k < N * N; ++k) acc[k] += red[0][k][lane] + red[1][k][lane] + red[2][k][lane];
  float mx = -INFINITY;
#pragma unroll
  for (int dy = 0; dy < N; ++dy)
#pragma unroll
    for (int dx = 0; dx < N; ++dx) {
      const int yy = y + dy - R, xx = x + dx - R;
      const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
      acc[dy * N + dx] = in ? acc[dy * N + dx] / scale : -1e4f;
      mx = fmaxf(mx, acc[dy * N + dx]);
    }
  float s = 0.f, ax = 0.f, ay = 0.f;
#pragma unroll
  for (int dy = 0; dy < N; ++dy)
#pragma unroll
    for (int dx = 0; dx < N; ++dx) {
      const float e = expf(acc[dy * N + dx] - mx);
      s += e;
      ax += e * (float)(x + dx - R);
      ay += e * (float)(y + dy - R);
    }
  out[p] = ax / s - (float)x;
  out[P + p] = ay / s - (float)y;
}

// The same on the fp32 matrix cores (C == 128; the kernel above stays for other channel counts).  The one-lane-per-pixel form
// issues (2r+1)^2 scattered 4-byte loads per channel and lane -- 5.6 M wave-level loads at 144 x 240 x 128, 431 us per direction
// in the GMFSS_UNION step -- for 1.4 GFLOP.  Here a wave owns 16 pixels of ROWS rows: their feature0 columns are the A operand
// (pixels x channels, held in registers for the whole kernel), a feature1 row segment of 32 columns (x0 - 4 .. x0 + 27) is the
// B operand, and v_mfma_f32_16x16x4_f32 forms all 16 x 32 pixel pairs of (target row, feature1 row) over the 128 channels; the
// 9 in-window pairs of each row are parked in LDS as [pixel][dy][dx].  28 % of the products are used -- the matrix pipe is idle
// otherwise.  Masking, softmax and the expected offset are the epilogue above with the 81 taps of a pixel dealt to 4 / ROWS
// lanes (partial maxima and sums combined by xor-shuffles: the sums associate differently, nothing else changes).
typedef float f32x4m __attribute__((ext_vector_type(4)));
template <int R, int ROWS>
__global__ void __launch_bounds__(256)
local_corr_flow_mfma_kernel(const float *__restrict__ f0, const float *__restrict__ f1, float *__restrict__ out, int H, int W,
                            float scale) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int N = 2 * R + 1, KS = 32;  // KS: k-steps of 4 channels (C == 128: straight-line loads)
  constexpr int ITEMS = 2 * (N + ROWS - 1), LPP = 4 / ROWS;  // (feature1 row, column block) pairs; lanes per pixel in the epilogue
  static_assert(R == 4 && (ROWS == 1 || ROWS == 2), "the 32-column feature1 segment covers x0 - 4 .. x0 + 15 + 4");
  __shared__ float sc[4][ROWS][16][N * N + 1];
  const size_t P = (size_t)H * W;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n16 = lane & 15, grp = lane >> 4;
  const int tiles_x = (W + 63) / 64;
  const int t = xcd_band(blockIdx.x, gridDim.x);
  const int ty = t / tiles_x, tx = t - ty * tiles_x;
  const int x0 = tx * 64 + wave * 16, y0 = ty * ROWS;
  // A: feature0 of target pixel (x0 + n16, y0 + tr), channels 4 s + grp
  float a[ROWS][KS];
  {
    const int xa = min(x0 + n16, W - 1);
#pragma unroll
    for (int tr = 0; tr < ROWS; ++tr) {
      const float *src = f0 + (size_t)grp * P + (size_t)min(y0 + tr, H - 1) * W + xa;
#pragma unroll
      for (int s = 0; s < KS; ++s) a[tr][s] = src[(size_t)(4 * s) * P];
    }
  }
  int qx[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) qx[nt] = min(max(x0 - R + 16 * nt + n16, 0), W - 1);
  // (feature1 row, column block) pairs in order, the next pair's 32 fragments in flight under this pair's MFMAs
  auto fetch = [&](int it, float (&b)[KS]) {
    const int r = it >> 1, nt = it & 1;
    const int yy = min(max(y0 - R + r, 0), H - 1);
    const float *src = f1 + (size_t)grp * P + (size_t)yy * W + qx[nt];
#pragma unroll
    for (int s = 0; s < KS; ++s) b[s] = src[(size_t)(4 * s) * P];
  };
  auto work = [&](int it, const float (&b)[KS]) {  // feature1 row y0 - R + r: dy = r - tr for target row tr
    const int r = it >> 1, nt = it & 1;
    f32x4m acc[ROWS];
#pragma unroll
    for (int tr = 0; tr < ROWS; ++tr) acc[tr] = (f32x4m){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int tr = 0; tr < ROWS; ++tr)  // (ROWS 2: the first / last row's unused product costs less than a branch around it)
        acc[tr] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tr][s], b[s], acc[tr], 0, 0, 0);
    // lane (n16, grp) holds pairs (target m = 4 grp + i, feature1 column x0 - R + 16 nt + n16): dx index = 16 nt + n16 - m
#pragma unroll
    for (int tr = 0; tr < ROWS; ++tr) {
      const int dy = r - tr;
      if (dy < 0 || dy >= N) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = 4 * grp + i, dxi = 16 * nt + n16 - m;
        if (dxi >= 0 && dxi < N) sc[wave][tr][m][dy * N + dxi] = acc[tr][i];
      }
    }
  };
  float b0[KS], b1[KS];
  fetch(0, b0);
  for (int it = 0; it < ITEMS; it += 2) {
    fetch(it + 1, b1);
    work(it, b0);
    if (it + 2 < ITEMS) fetch(it + 2, b0);
    work(it + 1, b1);
  }
  __syncthreads();
  const int px = lane / LPP, part = lane - px * LPP;
  const int tr = px >> 4, m = px & 15;
  const int x = x0 + m, y = y0 + tr;
  const float *sv = sc[wave][tr][m];
  constexpr int TPL = (N * N + LPP - 1) / LPP;  // taps per lane: k = part + LPP j
  float v[TPL];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < TPL; ++j) {
    const int k = part + LPP * j, dy = k / N, dx = k - dy * N;
    const int yy = y + dy - R, xx = x + dx - R;
    const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
    v[j] = k >= N * N ? -INFINITY : in ? sv[min(k, N * N - 1)] / scale : -1e4f;
    mx = fmaxf(mx, v[j]);
  }
#pragma unroll
  for (int d = 1; d < LPP; d <<= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
  float s = 0.f, ax = 0.f, ay = 0.f;
#pragma unroll
  for (int j = 0; j < TPL; ++j) {
    const int k = part + LPP * j, dy = k / N, dx = k - dy * N;
    const float e = expf(v[j] - mx);  // (a tap past the 81st: exp(-inf) = 0)
    s += e;
    ax += e * (float)(x + dx - R);
    ay += e * (float)(y + dy - R);
  }
#pragma unroll
  for (int d = 1; d < LPP; d <<= 1) {
    s += __shfl_xor(s, d, 64);
    ax += __shfl_xor(ax, d, 64);
    ay += __shfl_xor(ay, d, 64);
  }
  if (part != 0 || x >= W || y >= H) return;
  const size_t p = (size_t)y * W + x;
  out[p] = ax / s - (float)x;
  out[P + p] = ay / s - (float)y;
#endif
}

// transformer.py:374-409 (local_window_radius r): q . k over the (2r+1)^2 zero-padded window (out-of-image keys are
// zero vectors: score 0, NOT masked), softmax, weighted sum of the zero-padded flow window.
__global__ void __launch_bounds__(256)
local_attn_flow_kernel(const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ flow,
                       float *__restrict__ out, int C, int H, int W, int r, float scale) {
  const size_t P = (size_t)H * W;
  const size_t p = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= P) return;
  const int lane = threadIdx.x & 63, y = (int)(p / W), x = (int)(p % W);
  float mx = -INFINITY, s = 0.f, ax = 0.f, ay = 0.f;
  for (int dy = -r; dy <= r; ++dy)
    for (int dx = -r; dx <= r; ++dx) {
      const int yy = y + dy, xx = x + dx;
      const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
      float sc = 0.f, fx = 0.f, fy = 0.f;
      if (in) {
        const size_t pq = (size_t)yy * W + xx;
        float d = 0.f;
        for (int c = lane; c < C; c += 64) d += q[p * C + c] * k[pq * C + c];  // token-major [P, C]
        d = wave_sum(d);
        sc = __shfl(d, 0, 64) / scale;
        fx = flow[pq];
        fy = flow[P + pq];
      }
      const float nm = fmaxf(mx, sc);
      const float f = expf(mx - nm), e = expf(sc - nm);
      s = s * f + e;
      ax = ax * f + e * fx;
      ay = ay * f + e * fy;
      mx = nm;
    }
  if (lane == 0) {
    out[p] = ax / s;
    out[P + p] = ay / s;
  }
}

// gmflow.py:76-89 (upsample_factor K): mask [9*K*K, h, w] -> softmax over the 9 taps; up[c, K*y+i, K*x+j] =
// sum_t softmax_t * (K * flow[c, y+ty-1, x+tx-1]) with zero padding.
__global__ void __launch_bounds__(256)
convex_upsample_kernel(const float *__restrict__ mask, const float *__restrict__ flow, float *__restrict__ out, int h,
                       int w, int K) {
  const size_t P = (size_t)h * w;
  const int HW_o = K * w;
  const size_t total = P * K * K;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int X = (int)(i % HW_o), Y = (int)(i / HW_o);
    const int x = X / K, jj = X % K, y = Y / K, ii = Y % K;
    const size_t p = (size_t)y * w + x;
    float m[9], mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      m[t] = mask[((size_t)(t * K + ii) * K + jj) * P + p];
      mx = fmaxf(mx, m[t]);
    }
    float s = 0.f, ax = 0.f, ay = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float e = expf(m[t] - mx);
      s += e;
      const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
      if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
        ax += e * ((float)K * flow[(size_t)yy * w + xx]);
        ay += e * ((float)K * flow[P + (size_t)yy * w + xx]);
      }
    }
    out[i] = ax / s;
    out[total + i] = ay / s;
  }
}

// geometry.py:53-84 flow_warp: bilinear, zeros padding, coords normalised 2c/(size-1)-1
__device__ __forceinline__ float fw_coord(int c, float f, int n) {
  const float g = 2.f * ((float)c + f) / (float)(n - 1) - 1.f;
  return (g + 1.f) * (((float)n - 1.f) / 2.f);
}
__global__ void __launch_bounds__(256)
flow_warp_kernel(const float *__restrict__ in, const float *__restrict__ flow, float *__restrict__ out, int C, int H,
                 int W) {
  const size_t P = (size_t)H * W;
  const Tile2D tp = tile_pixel(W, H);
  if (!tp.valid) return;
  const size_t p = (size_t)tp.y * W + tp.x;
  const float sx = fw_coord(tp.x, flow[p], W), sy = fw_coord(tp.y, flow[P + p], H);
  const bool fin = isfinite(sx) && isfinite(sy);
  const float fx = floorf(sx), fy = floorf(sy);
  const int x0 = (int)fx, y0 = (int)fy;
  const float wx1 = sx - fx, wy1 = sy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
  const bool okx0 = x0 >= 0 && x0 < W, okx1 = x0 + 1 >= 0 && x0 + 1 < W;
  const bool oky0 = y0 >= 0 && y0 < H, oky1 = y0 + 1 >= 0 && y0 + 1 < H;
  for (int c = 0; c < C; ++c) {
    const float *pl = in + (size_t)c * P;
    float v = 0.f;
    if (fin) {
      if (okx0 && oky0) v += pl[(size_t)y0 * W + x0] * (wx0 * wy0);
      if (okx1 && oky0) v += pl[(size_t)y0 * W + x0 + 1] * (wx1 * wy0);
      if (okx0 && oky1) v += pl[(size_t)(y0 + 1) * W + x0] * (wx0 * wy1);
      if (okx1 && oky1) v += pl[(size_t)(y0 + 1) * W + x0 + 1] * (wx1 * wy1);
    }
    out[(size_t)c * P + p] = v;
  }
}

// F.interpolate(bilinear, align_corners=True): src = dst * (in-1)/(out-1)
__global__ void __launch_bounds__(256)
resize_ac_kernel(const float *__restrict__ in, float *__restrict__ out, int NC, int Hin, int Win, int Hout, int Wout,
                 float mul) {
  const size_t total = (size_t)NC * Hout * Wout;
  const float ry = Hout > 1 ? (float)(Hin - 1) / (float)(Hout - 1) : 0.f;
  const float rx = Wout > 1 ? (float)(Win - 1) / (float)(Wout - 1) : 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % Wout), oy = (int)((i / Wout) % Hout), c = (int)(i / ((size_t)Wout * Hout));
    const float sy = ry * (float)oy, sx = rx * (float)ox;
    const int y0 = min((int)sy, Hin - 1), x0 = min((int)sx, Win - 1);
    const int y1 = y0 + (y0 < Hin - 1 ? 1 : 0), x1 = x0 + (x0 < Win - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float *p = in + (size_t)c * Hin * Win;
    const float top = (1.f - lx) * p[(size_t)y0 * Win + x0] + lx * p[(size_t)y0 * Win + x1];
    const float bot = (1.f - lx) * p[(size_t)y1 * Win + x0] + lx * p[(size_t)y1 * Win + x1];
    out[i] = ((1.f - ly) * top + ly * bot) * mul;
  }
}

}  // namespace

extern "C" {

// 1x1 convolutions (pad 0, stride S; GMFlow's 96 -> 128 projection, the strided 1x1 shortcuts, the 256 -> 144 convex-upsampling
// head: backbone.py:24-30,96, gmflow.py:60-63) on the fp32 matrix cores instead of scalar-weight FMAs: a wave owns 16 output
// pixels and 64 output channels, v_mfma_f32_16x16x4_f32 with A = the pixels' input channels (lane l: channel 4k + (l >> 4),
// pixel l & 15 -- a 64-byte run per channel, loaded once per k-step and shared by the four cout tiles), B = the weights
// (lane l: w[co0 + (l & 15)][4k + (l >> 4)], L1 / L2 resident), D = 4 consecutive pixels of one cout per lane.  Exact fp32
// products; the sum over the input channels is formed in another order than the direct kernel's.  Round 4: the direct kernel
// took 350 us per call on the 1080p GMFSS_UNION shapes (2.1 ms of a 47 ms step).
typedef float f32x4c __attribute__((ext_vector_type(4)));
constexpr int kC1NT = 4;  // cout tiles of 16 per wave
__global__ void __launch_bounds__(256)
conv1x1_mfma_kernel(const float *__restrict__ in, const float *__restrict__ w, const float *__restrict__ bias, float *__restrict__ out,
                    int Cin, int H, int W, int Cout, int Ho, int Wo, int S, int ptiles, int cgroups) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int task = blockIdx.x * 4 + wave;            // (pixel tile, cout group)
  const int n = blockIdx.y;
  if (task >= ptiles * cgroups) return;
  const int cg = task / ptiles, pt = task - cg * ptiles;
  const int Po = Ho * Wo;
  const int m = lane & 15, kq = lane >> 4;
  const int p = min(pt * 16 + m, Po - 1);            // this lane's A pixel (clamped: loadable)
  const int py = p / Wo, px = p - py * Wo;
  const float *a_ptr = in + ((size_t)n * Cin + kq) * H * W + (size_t)(py * S) * W + px * S;
  const size_t a_step = (size_t)4 * H * W;
  const int co0 = cg * 16 * kC1NT;
  const float *b_ptr[kC1NT];
#pragma unroll
  for (int t = 0; t < kC1NT; ++t) b_ptr[t] = w + (size_t)min(co0 + t * 16 + m, Cout - 1) * Cin + kq;
  f32x4c acc[kC1NT];
#pragma unroll
  for (int t = 0; t < kC1NT; ++t) acc[t] = (f32x4c){0.f, 0.f, 0.f, 0.f};
  const int ksteps = Cin >> 2;                        // Cin % 4 == 0 (checked by the launcher)
  int k = 0;
  for (; k + 4 <= ksteps; k += 4) {  // four k-steps of loads in flight in front of their 16 MFMAs
    float a[4], b[4][kC1NT];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a[j] = a_ptr[(size_t)(k + j) * a_step];
#pragma unroll
      for (int t = 0; t < kC1NT; ++t) b[j][t] = b_ptr[t][4 * (k + j)];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < kC1NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j][t], acc[t], 0, 0, 0);
  }
  for (; k < ksteps; ++k) {
    const float a = a_ptr[(size_t)k * a_step];
#pragma unroll
    for (int t = 0; t < kC1NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b_ptr[t][4 * k], acc[t], 0, 0, 0);
  }
  // lane: pixels pt * 16 + 4 kq .. + 3 of cout co0 + 16 t + m
  const int q0 = pt * 16 + 4 * kq;
#pragma unroll
  for (int t = 0; t < kC1NT; ++t) {
    const int co = co0 + t * 16 + m;
    if (co >= Cout) continue;
    const float bs = bias ? bias[co] : 0.f;
    float *dst = out + ((size_t)n * Cout + co) * Po + q0;
    if (q0 + 3 < Po && (Po & 3) == 0) {
      *reinterpret_cast<f32x4c *>(dst) = (f32x4c){acc[t][0] + bs, acc[t][1] + bs, acc[t][2] + bs, acc[t][3] + bs};
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (q0 + i < Po) dst[i] = acc[t][i] + bs;
    }
  }
}

int drba_conv_direct(const float *in, const float *w, const float *bias, float *out, int N, int Cin, int H, int W,
                     int Cout, int K, int stride, int pad, void *stream) {
  if (!in || !w || !out || N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || K <= 0 || stride <= 0 || pad < 0)
    return DRBA_EINVAL;
  const int Ho = (H + 2 * pad - K) / stride + 1, Wo = (W + 2 * pad - K) / stride + 1;
  if (K == 1 && pad == 0 && (Cin & 3) == 0 && N <= 65535 && (((uintptr_t)out) & 15) == 0) {
    const int ptiles = (Ho * Wo + 15) / 16, cgroups = (Cout + 16 * kC1NT - 1) / (16 * kC1NT);
    DRBA_LAUNCH(conv1x1_mfma_kernel, dim3((ptiles * cgroups + 3) / 4, N), dim3(kBlock), 0, (hipStream_t)stream, in, w, bias, out, Cin, H,
                W, Cout, Ho, Wo, stride, ptiles, cgroups);
    DRBA_CHECK_LAUNCH();
    return DRBA_OK;
  }
  const int cblocks = (Cout + kCob - 1) / kCob;
  DRBA_LAUNCH(conv_direct_kernel, dim3(tiles_for(Wo, Ho), N * cblocks), dim3(kBlock), 0, (hipStream_t)stream, in, w,
                     bias, out, N, Cin, H, W, Cout, Ho, Wo, K, stride, pad, cblocks);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

size_t drba_instance_norm_ws_floats(int planes) { return (size_t)(planes > 0 ? planes : 0) * kInChunks * 2; }

int drba_instance_norm(const float *in, float *out, float *ws, int planes, size_t HW, float eps, int relu, void *stream) {
  if (!in || !out || !ws || planes <= 0 || HW == 0) return DRBA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  DRBA_LAUNCH(instance_norm_stats_kernel, dim3(kInChunks, planes), dim3(kBlock), 0, s, in, ws, HW);
  DRBA_LAUNCH(instance_norm_apply_kernel, dim3(kInChunks, planes), dim3(kBlock), 0, s, in, ws, out, HW, eps, relu);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_add_act(const float *a, const float *b, float *out, size_t n, int relu, void *stream) {
  if (!a || !b || !out || n == 0) return DRBA_EINVAL;
  DRBA_LAUNCH(add_act_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, a, b, out, n, relu);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_channel_normalize3(const float *in, float *out, int N, size_t HW, const float *mean3, const float *std3,
                            void *stream) {
  if (!in || !out || !mean3 || !std3 || N <= 0 || HW == 0) return DRBA_EINVAL;
  const size_t n = (size_t)N * 3 * HW;
  DRBA_LAUNCH(channel_affine_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, in, out, 3, HW,
                     mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], n);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_layernorm(const float *x, const float *w, const float *b, const float *residual, float *out, size_t rows,
                   int cols, float eps, void *stream) {
  if (!x || !w || !b || !out || rows == 0 || cols <= 0) return DRBA_EINVAL;
  DRBA_LAUNCH(layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(kBlock), 0, (hipStream_t)stream, x, w, b,
                     residual, out, rows, cols, eps);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_gelu(const float *x, float *out, size_t n, void *stream) {
  if (!x || !out || n == 0) return DRBA_EINVAL;
  DRBA_LAUNCH(gelu_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, x, out, n);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_softmax_rows(float *x, const float *mask, size_t rows, int cols, int rows_per_mat, int n_masks, float scale,
                      void *stream) {
  if (!x || rows == 0 || cols <= 0 || rows_per_mat <= 0 || !(scale > 0.f) || (mask && n_masks <= 0)) return DRBA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int nm = mask ? n_masks : 1;
#define DRBA_SOFTMAX(EPT, RPB)                                                                                     \
  DRBA_LAUNCH((softmax_rows_kernel<EPT, RPB>), dim3((unsigned)((rows + RPB - 1) / RPB)), dim3(kBlock), 0, s, x, \
                     mask, rows, cols, rows_per_mat, nm, scale)
  if (cols <= 64 * 9) DRBA_SOFTMAX(9, 4);
  else if (cols <= 64 * 16) DRBA_SOFTMAX(16, 4);
  else if (cols <= 256 * 9) DRBA_SOFTMAX(9, 1);
  else if (cols <= 256 * 34) DRBA_SOFTMAX(34, 1);
  else
    DRBA_LAUNCH(softmax_rows_long_kernel, dim3((unsigned)rows), dim3(kBlock), 0, s, x, mask, cols, rows_per_mat, nm,
                       scale);
#undef DRBA_SOFTMAX
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_local_corr_flow(const float *f0, const float *f1, float *out, int C, int H, int W, int radius, void *stream) {
  if (!f0 || !f1 || !out || C <= 0 || H <= 0 || W <= 0 || radius <= 0) return DRBA_EINVAL;
  if (radius != 4) return DRBA_EUNSUPPORTED;  // the radius GMFlow's refinement stage uses (gmflow.py corr_radius_list)
  static const int rows = env_int("DRBA_LCORR_ROWS", 1);  // (TUNING builds only)
  if (C == 128 && rows == 2)
    DRBA_LAUNCH((local_corr_flow_mfma_kernel<4, 2>), dim3((unsigned)(((W + 63) / 64) * ((H + 1) / 2))), dim3(kBlock), 0, (hipStream_t)stream,
                f0, f1, out, H, W, sqrtf((float)C));
  else if (C == 128)  // (GMFlow's feature channels)
    DRBA_LAUNCH((local_corr_flow_mfma_kernel<4, 1>), dim3((unsigned)(((W + 63) / 64) * H)), dim3(kBlock), 0, (hipStream_t)stream, f0, f1,
                out, H, W, sqrtf((float)C));
  else
    DRBA_LAUNCH(local_corr_flow_kernel<4>, dim3((unsigned)(((W + 31) / 32) * ((H + 1) / 2))), dim3(kBlock), 0,
                       (hipStream_t)stream, f0, f1, out, C, H, W, sqrtf((float)C));
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_local_attn_flow(const float *q_tok, const float *k_tok, const float *flow, float *out, int C, int H, int W,
                         int radius, void *stream) {
  if (!q_tok || !k_tok || !flow || !out || C <= 0 || H <= 0 || W <= 0 || radius <= 0) return DRBA_EINVAL;
  const size_t P = (size_t)H * W;
  DRBA_LAUNCH(local_attn_flow_kernel, dim3((unsigned)((P + 3) / 4)), dim3(kBlock), 0, (hipStream_t)stream, q_tok,
                     k_tok, flow, out, C, H, W, radius, sqrtf((float)C));
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_convex_upsample(const float *mask, const float *flow, float *out, int h, int w, int factor, void *stream) {
  if (!mask || !flow || !out || h <= 0 || w <= 0 || factor <= 0) return DRBA_EINVAL;
  DRBA_LAUNCH(convex_upsample_kernel, dim3(grid_for((size_t)h * w * factor * factor)), dim3(kBlock), 0,
                     (hipStream_t)stream, mask, flow, out, h, w, factor);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_flow_warp(const float *in, const float *flow, float *out, int C, int H, int W, void *stream) {
  if (!in || !flow || !out || C <= 0 || H <= 1 || W <= 1) return DRBA_EINVAL;
  DRBA_LAUNCH(flow_warp_kernel, dim3(tiles_for(W, H)), dim3(kBlock), 0, (hipStream_t)stream, in, flow, out, C, H,
                     W);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_resize_bilinear_ac(const float *in, float *out, int NC, int Hin, int Win, int Hout, int Wout, float mul,
                            void *stream) {
  if (!in || !out || NC <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0) return DRBA_EINVAL;
  DRBA_LAUNCH(resize_ac_kernel, dim3(grid_for((size_t)NC * Hout * Wout)), dim3(kBlock), 0, (hipStream_t)stream, in,
                     out, NC, Hin, Win, Hout, Wout, mul);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

}  // extern "C"
