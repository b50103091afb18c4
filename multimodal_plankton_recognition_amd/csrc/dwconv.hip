// EfficientNet building blocks that the ResNet / transformer paths do not have (timm `efficientnet_b0` behind
// src/image_encoder.py:16,24 -- the backbone the reference's cards name, model_cards/example_multi.yaml:9):
//   * depthwise k x k convolution (3x3 / 5x5, stride 1 / 2, pad k/2), channels-last bf16, forward, data gradient and
//     weight gradient.  One multiply-add per loaded element: HBM / L2-bound, so no MFMA -- a thread owns one 16-byte
//     channel group of one pixel and walks the taps (neighbouring pixels' reads of the same input rows hit in L2);
//   * squeeze-excite gating y = x * gate[b][c] and its backward (dx = dy * gate, dgate = sum over pixels of dy * x).
// The 1x1 expansions / projections and the head run on the implicit-GEMM conv kernels, BatchNorm on batchnorm.hip, SiLU on
// the bf16 elementwise passes of transformer_bf16.hip, the SE bottleneck on the fp32 GEMM.
#include "common.h"

struct DwGeom {
  int B, H, W, C, P, Q, R, S, sh, sw, ph, pw;
};

// wt[tap][c] = w[c][tap]    (torch [C][1][R][S] -> tap-major: a thread's 8 channels of one tap are two float4)
// (flip: tap RS - 1 - tap -- a stride-1 data gradient is the forward convolution with the flipped filter)
__global__ __launch_bounds__(256) void dw_pack_kernel(const float* __restrict__ w, float* __restrict__ wt, int C, int RS,
                                                      int flip = 0) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= C * RS) return;
  const int tap = i / C, c = i - tap * C;
  wt[i] = w[(size_t)c * RS + (flip ? RS - 1 - tap : tap)];
}

// y[b][p][q][c] = sum_{r,s} x[b][p*sh-ph+r][q*sw-pw+s][c] * wt[r*S+s][c]
__global__ __launch_bounds__(256) void dw_fwd_kernel(const uint4* __restrict__ x, const float* __restrict__ w,
                                                     uint4* __restrict__ y, const DwGeom g) {
  const uint32_t G = g.C / 8;
  const uint32_t total = (uint32_t)g.B * g.P * g.Q * G;            // (< 2^31: checked by the host)
  for (uint32_t idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int cg = (int)(idx % G);
    uint32_t pix = idx / G;
    const int q = (int)(pix % (uint32_t)g.Q);
    pix /= (uint32_t)g.Q;
    const int p = (int)(pix % (uint32_t)g.P), b = (int)(pix / (uint32_t)g.P);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int r = 0; r < g.R; ++r) {
      const int h = p * g.sh - g.ph + r;
      if (h < 0 || h >= g.H) continue;
      for (int s = 0; s < g.S; ++s) {
        const int ww = q * g.sw - g.pw + s;
        if (ww < 0 || ww >= g.W) continue;
        float f[8];
        unpack8(x[((size_t)(b * g.H + h) * g.W + ww) * G + cg], f);
        const float4* wp = reinterpret_cast<const float4*>(w + (size_t)(r * g.S + s) * g.C + cg * 8);
        const float4 w0 = wp[0], w1 = wp[1];
        acc[0] = fmaf(f[0], w0.x, acc[0]); acc[1] = fmaf(f[1], w0.y, acc[1]); acc[2] = fmaf(f[2], w0.z, acc[2]);
        acc[3] = fmaf(f[3], w0.w, acc[3]); acc[4] = fmaf(f[4], w1.x, acc[4]); acc[5] = fmaf(f[5], w1.y, acc[5]);
        acc[6] = fmaf(f[6], w1.z, acc[6]); acc[7] = fmaf(f[7], w1.w, acc[7]);
      }
    }
    y[idx] = pack8(acc);
  }
}

// dx[b][h][w][c] = sum_{r,s : (h+ph-r) % sh == 0, ...} dy[b][(h+ph-r)/sh][(w+pw-s)/sw][c] * w[c][r][s]
__global__ __launch_bounds__(256) void dw_dgrad_kernel(const uint4* __restrict__ dy, const float* __restrict__ w,
                                                       uint4* __restrict__ dx, const DwGeom g) {
  const uint32_t G = g.C / 8;
  const uint32_t total = (uint32_t)g.B * g.H * g.W * G;
  for (uint32_t idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int cg = (int)(idx % G);
    uint32_t pix = idx / G;
    const int ww = (int)(pix % (uint32_t)g.W);
    pix /= (uint32_t)g.W;
    const int h = (int)(pix % (uint32_t)g.H), b = (int)(pix / (uint32_t)g.H);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int r = 0; r < g.R; ++r) {
      const int hp = h + g.ph - r;
      if (hp < 0 || hp % g.sh) continue;
      const int p = hp / g.sh;
      if (p >= g.P) continue;
      for (int s = 0; s < g.S; ++s) {
        const int wq = ww + g.pw - s;
        if (wq < 0 || wq % g.sw) continue;
        const int q = wq / g.sw;
        if (q >= g.Q) continue;
        float f[8];
        unpack8(dy[((size_t)(b * g.P + p) * g.Q + q) * G + cg], f);
        const float4* wp = reinterpret_cast<const float4*>(w + (size_t)(r * g.S + s) * g.C + cg * 8);
        const float4 w0 = wp[0], w1 = wp[1];
        acc[0] = fmaf(f[0], w0.x, acc[0]); acc[1] = fmaf(f[1], w0.y, acc[1]); acc[2] = fmaf(f[2], w0.z, acc[2]);
        acc[3] = fmaf(f[3], w0.w, acc[3]); acc[4] = fmaf(f[4], w1.x, acc[4]); acc[5] = fmaf(f[5], w1.y, acc[5]);
        acc[6] = fmaf(f[6], w1.z, acc[6]); acc[7] = fmaf(f[7], w1.w, acc[7]);
      }
    }
    dx[idx] = pack8(acc);
  }
}

// ---- register-tiled forms (round 3) for the shapes EfficientNet uses: square filters of 3 or 5 taps, stride 1 or 2, pad k / 2.
// The per-pixel kernels above issue one conditional 16-byte load per tap and pixel (9 / 25 per output group, each behind a
// branch: the loop waits for every load before the next) and ran at 1.8 TB/s on the 112 x 112 x 32 map.  Here a thread owns T
// CONSECUTIVE outputs of one image row for its 8 channels: a filter row's (T - 1) * stride + S input groups are loaded once,
// unconditionally (out-of-image columns masked to zero), and slide past the T outputs in registers -- 12 loads instead of 40 for
// a 5-tap row at T = 8, all independent and in flight together.
template <int S, int SW, int T>
__global__ __launch_bounds__(256) void dw_fwd_tiled_kernel(const uint4* __restrict__ x, const float* __restrict__ w,
                                                           uint4* __restrict__ y, const DwGeom g) {
  constexpr int NX = (T - 1) * SW + S;
  const uint32_t G = g.C / 8;
  const uint32_t QT = (g.Q + T - 1) / T;
  const uint32_t total = (uint32_t)g.B * g.P * QT * G;
  for (uint32_t idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int cg = (int)(idx % G);
    uint32_t st = idx / G;
    const int q0 = (int)(st % QT) * T;
    st /= QT;
    const int p = (int)(st % (uint32_t)g.P), b = (int)(st / (uint32_t)g.P);
    float acc[T][8];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[i][e] = 0.f;
    const int wl = q0 * SW - g.pw;
#pragma unroll
    for (int r = 0; r < S; ++r) {
      const int h = p * SW - g.ph + r;
      if (h < 0 || h >= g.H) continue;
      const uint4* row = x + ((size_t)(b * g.H + h) * g.W) * G + cg;
      uint4 v[NX];
#pragma unroll
      for (int j = 0; j < NX; ++j) {
        const int ww = wl + j;
        const bool ok = ww >= 0 && ww < g.W;
        v[j] = ok ? row[(size_t)ww * G] : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const float4* wp = reinterpret_cast<const float4*>(w + (size_t)(r * S + s) * g.C + cg * 8);
        const float4 w0 = wp[0], w1 = wp[1];
#pragma unroll
        for (int i = 0; i < T; ++i) {
          float f[8];
          unpack8(v[i * SW + s], f);
          acc[i][0] = fmaf(f[0], w0.x, acc[i][0]); acc[i][1] = fmaf(f[1], w0.y, acc[i][1]);
          acc[i][2] = fmaf(f[2], w0.z, acc[i][2]); acc[i][3] = fmaf(f[3], w0.w, acc[i][3]);
          acc[i][4] = fmaf(f[4], w1.x, acc[i][4]); acc[i][5] = fmaf(f[5], w1.y, acc[i][5]);
          acc[i][6] = fmaf(f[6], w1.z, acc[i][6]); acc[i][7] = fmaf(f[7], w1.w, acc[i][7]);
        }
      }
    }
    uint4* out = y + ((size_t)(b * g.P + p) * g.Q + q0) * G + cg;
#pragma unroll
    for (int i = 0; i < T; ++i)
      if (q0 + i < g.Q) out[(size_t)i * G] = pack8(acc[i]);
  }
}

// Stride-2 data gradient, T = 8 consecutive columns w0 .. w0 + 7 (w0 a multiple of 8, so the column parity pattern of the taps
// is a compile-time fact): a gradient row p = (h + ph - r) / 2 contributes through the taps with r = (h + ph) mod 2, and column
// w0 + i takes tap s from gradient column (w0 + i + pw - s) / 2 when that is whole -- six gradient groups per row, loaded once.
template <int S>
__global__ __launch_bounds__(256) void dw_dgrad_s2_tiled_kernel(const uint4* __restrict__ dy, const float* __restrict__ w,
                                                                uint4* __restrict__ dx, const DwGeom g) {
  constexpr int T = 8, PW = S / 2, NQ = 6;
  const uint32_t G = g.C / 8;
  const uint32_t WT = (g.W + T - 1) / T;
  const uint32_t total = (uint32_t)g.B * g.H * WT * G;
  for (uint32_t idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int cg = (int)(idx % G);
    uint32_t st = idx / G;
    const int w0 = (int)(st % WT) * T;
    st /= WT;
    const int h = (int)(st % (uint32_t)g.H), b = (int)(st / (uint32_t)g.H);
    float acc[T][8];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[i][e] = 0.f;
    const int q0 = w0 / 2 - 1;
#pragma unroll
    for (int r = 0; r < S; ++r) {
      const int hp = h + PW - r;
      if (hp < 0 || (hp & 1)) continue;
      const int p = hp >> 1;
      if (p >= g.P) continue;
      const uint4* row = dy + ((size_t)(b * g.P + p) * g.Q) * G + cg;
      uint4 v[NQ];
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const int q = q0 + j;
        const bool ok = q >= 0 && q < g.Q;
        v[j] = ok ? row[(size_t)q * G] : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const float4* wp = reinterpret_cast<const float4*>(w + (size_t)(r * S + s) * g.C + cg * 8);
        const float4 w0f = wp[0], w1f = wp[1];
#pragma unroll
        for (int i = 0; i < T; ++i) {
          const int d = i + PW - s;              // (compile-time after unrolling)
          if (d % 2 == 0) {
            float f[8];
            unpack8(v[d / 2 + 1], f);
            acc[i][0] = fmaf(f[0], w0f.x, acc[i][0]); acc[i][1] = fmaf(f[1], w0f.y, acc[i][1]);
            acc[i][2] = fmaf(f[2], w0f.z, acc[i][2]); acc[i][3] = fmaf(f[3], w0f.w, acc[i][3]);
            acc[i][4] = fmaf(f[4], w1f.x, acc[i][4]); acc[i][5] = fmaf(f[5], w1f.y, acc[i][5]);
            acc[i][6] = fmaf(f[6], w1f.z, acc[i][6]); acc[i][7] = fmaf(f[7], w1f.w, acc[i][7]);
          }
        }
      }
    }
    uint4* out = dx + ((size_t)(b * g.H + h) * g.W + w0) * G + cg;
#pragma unroll
    for (int i = 0; i < T; ++i)
      if (w0 + i < g.W) out[(size_t)i * G] = pack8(acc[i]);
  }
}

// Weight gradient, tiled: block = Gp channel groups (any Gp <= 32: the host picks G / ceil(G / 32), so that no block is mostly
// idle) x 256 / Gp strip lanes, ONE filter row r per block (the R blocks of a slab are neighbours in dispatch order and share
// the slab's lines in L2); a thread walks strips of T = 4 consecutive outputs: 4 gradient groups and 3 * stride + S input
// groups per strip, all independent loads, S x 8 accumulators.  The per-pixel kernel above gave a thread ONE pixel lane at
// C >= 1024 (Gp = 256) and a serial chain of 1 + S loads per pixel: 290-350 us for the 29 MB maps of the 7 x 7 stages.
template <int S, int SW>
__global__ __launch_bounds__(256) void dw_wgrad_tiled_kernel(const uint4* __restrict__ x, const uint4* __restrict__ dy,
                                                             float* __restrict__ part, const DwGeom g, int strips_per_slab,
                                                             int Gp) {
  constexpr int T = 4, NX = (T - 1) * SW + S;
  __shared__ float red[256][S * 8 + 1];
  const int G = g.C / 8, nl = 256 / Gp;
  const int cl = threadIdx.x % Gp, rl = threadIdx.x / Gp;
  const int r = blockIdx.x % S;
  const int cg = (blockIdx.x / S) * Gp + cl;
  const uint32_t QT = (g.Q + T - 1) / T;
  const uint32_t nstrips = (uint32_t)g.B * g.P * QT;
  const uint32_t s0 = blockIdx.y * (uint32_t)strips_per_slab;
  uint32_t s1 = s0 + strips_per_slab;
  if (s1 > nstrips) s1 = nstrips;
  float acc[S][8];
#pragma unroll
  for (int s = 0; s < S; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[s][e] = 0.f;
  if (cg < G && rl < nl) {
    for (uint32_t st = s0 + rl; st < s1; st += nl) {
      const int q0 = (int)(st % QT) * T;
      const uint32_t bp = st / QT;
      const int p = (int)(bp % (uint32_t)g.P), b = (int)(bp / (uint32_t)g.P);
      const int h = p * SW - g.ph + r;
      if (h < 0 || h >= g.H) continue;
      const uint4* drow = dy + ((size_t)bp * g.Q + q0) * G + cg;
      const uint4* xrow = x + ((size_t)(b * g.H + h) * g.W) * G + cg;
      const int wl = q0 * SW - g.pw;
      uint4 dv[T], xv[NX];
#pragma unroll
      for (int i = 0; i < T; ++i) dv[i] = q0 + i < g.Q ? drow[(size_t)i * G] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
      for (int j = 0; j < NX; ++j) {
        const int ww = wl + j;
        xv[j] = (ww >= 0 && ww < g.W) ? xrow[(size_t)ww * G] : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int i = 0; i < T; ++i) {
        float d[8];
        unpack8(dv[i], d);
#pragma unroll
        for (int s = 0; s < S; ++s) {
          float a[8];
          unpack8(xv[i * SW + s], a);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[s][e] = fmaf(a[e], d[e], acc[s][e]);
        }
      }
    }
  }
#pragma unroll
  for (int s = 0; s < S; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e) red[threadIdx.x][s * 8 + e] = acc[s][e];
  __syncthreads();
  // (S * 8 sums of nl lanes per channel group: one thread per (channel group, tap s), 8 channels each)
  for (int o = threadIdx.x; o < Gp * S; o += 256) {
    const int c2 = o % Gp, s = o / Gp;
    const int cg2 = (blockIdx.x / S) * Gp + c2;
    if (cg2 >= G) continue;
    float out[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = 0.f;
      for (int l = 0; l < nl; ++l) a += red[l * Gp + c2][s * 8 + e];
      out[e] = a;
    }
    float* dst = part + ((size_t)blockIdx.y * (S * S) + r * S + s) * g.C + cg2 * 8;
    *reinterpret_cast<float4*>(dst) = make_float4(out[0], out[1], out[2], out[3]);
    *reinterpret_cast<float4*>(dst + 4) = make_float4(out[4], out[5], out[6], out[7]);
  }
}

// partial weight gradients of one slab of output pixels and ONE filter row r: part[slab][tap][C].
// Block = Gp channel groups (power of two >= C/8, <= 256) x 256/Gp pixel lanes; a thread keeps the S <= 5 taps of the row for
// its 8 channels in registers (dy loaded once per pixel, x once per tap), lanes reduced through LDS.
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const uint4* __restrict__ x, const uint4* __restrict__ dy,
                                                       float* __restrict__ part, const DwGeom g, int pix_per_slab, int Gp) {
  __shared__ float red[256][41];
  const int G = g.C / 8, RS = g.R * g.S, nl = 256 / Gp;
  const int cl = threadIdx.x % Gp, rl = threadIdx.x / Gp;
  // the R filter rows of one slab are neighbours in dispatch order (x = column block * R + r): they run at the same time and
  // the slab's x / dy lines are fetched from HBM once and found in the Infinity Cache / L2 by the others, instead of the
  // whole activation being re-read by R separate sweeps
  const int r = blockIdx.x % g.R;
  const int cg = (blockIdx.x / g.R) * Gp + cl;
  const uint32_t npix = (uint32_t)g.B * g.P * g.Q;
  const uint32_t p0 = blockIdx.y * (uint32_t)pix_per_slab;
  uint32_t p1 = p0 + pix_per_slab;
  if (p1 > npix) p1 = npix;
  float acc[5][8];
#pragma unroll
  for (int s = 0; s < 5; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[s][e] = 0.f;
  if (cg < G) {
    for (uint32_t pix = p0 + rl; pix < p1; pix += nl) {
      const int q = (int)(pix % (uint32_t)g.Q);
      const uint32_t bp = pix / (uint32_t)g.Q;
      const int p = (int)(bp % (uint32_t)g.P), b = (int)(bp / (uint32_t)g.P);
      const int h = p * g.sh - g.ph + r;
      if (h < 0 || h >= g.H) continue;
      float d[8];
      unpack8(dy[(size_t)pix * G + cg], d);
      const size_t rowbase = (size_t)(b * g.H + h) * g.W;
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        const int ww = q * g.sw - g.pw + s;
        if (s < g.S && ww >= 0 && ww < g.W) {
          float a[8];
          unpack8(x[(rowbase + ww) * G + cg], a);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[s][e] = fmaf(a[e], d[e], acc[s][e]);
        }
      }
    }
  }
#pragma unroll
  for (int s = 0; s < 5; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e) red[threadIdx.x][s * 8 + e] = acc[s][e];
  __syncthreads();
  if (rl == 0 && cg < G) {
    for (int s = 0; s < g.S; ++s) {
      float out[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float a = 0.f;
        for (int l = 0; l < nl; ++l) a += red[l * Gp + cl][s * 8 + e];
        out[e] = a;
      }
      float* dst = part + ((size_t)blockIdx.y * RS + r * g.S + s) * g.C + cg * 8;
      *reinterpret_cast<float4*>(dst) = make_float4(out[0], out[1], out[2], out[3]);
      *reinterpret_cast<float4*>(dst + 4) = make_float4(out[4], out[5], out[6], out[7]);
    }
  }
}

// dw[c][r][s] += sum_slab part[slab][tap][c]   (slabs split over gridDim.y, a few-way atomic per element)
__global__ __launch_bounds__(256) void dw_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int nslabs,
                                                              int RS, int C) {
  // (threads in dw's own order, c * RS + tap: consecutive lanes add to consecutive addresses -- one cache-line operation per 32
  //  lanes at the memory side instead of one per lane; the partial rows are read with stride C instead)
  const int o = blockIdx.x * 256 + threadIdx.x;       // = c * RS + tap
  if (o >= RS * C) return;
  const int c = o / RS, tap = o - c * RS;
  const int i = tap * C + c;
  const int per = (nslabs + gridDim.y - 1) / gridDim.y;
  int s0 = blockIdx.y * per, s1 = s0 + per;
  if (s1 > nslabs) s1 = nslabs;
  float a = 0.f;
  for (int sl = s0; sl < s1; ++sl) a += part[(size_t)sl * RS * C + i];
  atomicAdd(dw + o, a);
}

// y = x * gate[b][c]     x, y: [B][L][C] bf16, gate: [B][C] fp32
__global__ __launch_bounds__(256) void se_scale_kernel(const uint4* __restrict__ x, const float* __restrict__ gate,
                                                       uint4* __restrict__ y, int B, int L, int C) {
  const int G = C / 8;
  const long long total = (long long)B * L * G;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int cg = (int)(idx % G), b = (int)(idx / ((long long)L * G));
    float f[8];
    unpack8(x[idx], f);
    const float* gp = gate + (size_t)b * C + cg * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] *= gp[e];
    y[idx] = pack8(f);
  }
}

// Per-image channel sums over the L pixels: out[b][c] = scale * sum_l x[b][l][c] (* dy[b][l][c] when MUL): the SE squeeze
// (mean) and the gate gradient.  Block = Gp channel groups (power of two >= C/8, at most 256) x 256/Gp row lanes of ONE
// image, so that narrow-and-large maps (C = 32 at 112 x 112) still fill the block; lanes reduced through LDS.
template <bool MUL>
__global__ __launch_bounds__(256) void se_sum_kernel(const uint4* __restrict__ x, const uint4* __restrict__ dy,
                                                     float* __restrict__ out, int L, int C, int Gp, float scale) {
  __shared__ float red[256][9];
  const int G = C / 8, nl = 256 / Gp;
  const int cl = threadIdx.x % Gp, rl = threadIdx.x / Gp;
  const int cg = blockIdx.x * Gp + cl, b = blockIdx.y;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (cg < G && rl < nl) {
    // (U pixels per iteration, their loads first: one block per image leaves 4 waves per CU, each needs loads in flight)
    constexpr int U = MUL ? 4 : 8;
    for (int l = rl; l < L; l += nl * U) {
      uint4 xv[U], dv[MUL ? U : 1];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int lu = l + u * nl;
        const size_t o = ((size_t)b * L + (lu < L ? lu : l)) * G + cg;
        xv[u] = lu < L ? x[o] : make_uint4(0u, 0u, 0u, 0u);
        if (MUL) dv[u] = dy[o];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float a[8];
        unpack8(xv[u], a);
        if (MUL) {
          float d[8];
          unpack8(dv[u], d);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] = fmaf(a[e], d[e], acc[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += a[e];
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = acc[e];
  __syncthreads();
  if (rl == 0 && cg < G) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = 0.f;
      for (int l = 0; l < nl; ++l) a += red[l * Gp + cl][e];
      out[(size_t)b * C + cg * 8 + e] = a * scale;
    }
  }
}

static inline int se_gp(int C) {
  int gp = 1;
  while (gp < C / 8 && gp < 256) gp <<= 1;
  return gp;
}
// channel groups per block of the per-image sums: G / ceil(G / 64) (no power of two needed: a block of 256 at G = 144 left
// 44 % of its threads without a channel group and ONE row lane for the rest)
static inline int se_gp_sum(int C) {
  const int G = C / 8;
  return ceil_div(G, ceil_div(G, 64));
}

static inline unsigned dw_grid(long long n) {
  long long g = (n + 255) / 256;
  return (unsigned)(g < 16384 ? (g < 1 ? 1 : g) : 16384);
}

static inline bool dw_geom(DwGeom* g, int B, int H, int W, int C, int R, int S, int sh, int sw, int ph, int pw) {
  g->B = B; g->H = H; g->W = W; g->C = C; g->R = R; g->S = S; g->sh = sh; g->sw = sw; g->ph = ph; g->pw = pw;
  g->P = (H + 2 * ph - R) / sh + 1;
  g->Q = (W + 2 * pw - S) / sw + 1;
  return B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && R > 0 && S > 0 && sh > 0 && sw > 0 && g->P > 0 && g->Q > 0 &&
         (long long)B * H * W * (C / 8) < (1ll << 31) && (long long)B * g->P * g->Q * (C / 8) < (1ll << 31);   // 32-bit indices
}

// the shapes the register-tiled kernels are written for
static inline bool dw_tiled_shape(const DwGeom& g) {
  return g.R == g.S && (g.S == 3 || g.S == 5) && g.sh == g.sw && (g.sw == 1 || g.sw == 2) && g.ph == g.S / 2 && g.pw == g.S / 2 &&
         (long long)g.B * g.H * g.W * (g.C / 8) < (1ll << 30);
}
static inline void dw_fwd_tiled_launch(const DwGeom& g, const uint4* x, const float* wt, uint4* y, hipStream_t st) {
  const int G = g.C / 8;
  if (g.sw == 1) {
    const unsigned grid = dw_grid((long long)g.B * g.P * ((g.Q + 7) / 8) * G);
    if (g.S == 3) dw_fwd_tiled_kernel<3, 1, 8><<<grid, 256, 0, st>>>(x, wt, y, g);
    else dw_fwd_tiled_kernel<5, 1, 8><<<grid, 256, 0, st>>>(x, wt, y, g);
  } else {
    const unsigned grid = dw_grid((long long)g.B * g.P * ((g.Q + 3) / 4) * G);
    if (g.S == 3) dw_fwd_tiled_kernel<3, 2, 4><<<grid, 256, 0, st>>>(x, wt, y, g);
    else dw_fwd_tiled_kernel<5, 2, 4><<<grid, 256, 0, st>>>(x, wt, y, g);
  }
}

static inline int dw_pix_per_slab(long long npix) {      // at most 1024 slabs of at least 256 pixels
  long long p = (npix + 1023) / 1024;
  return (int)(p < 256 ? 256 : p);
}

extern "C" {

int mpr_dwconv_fwd(const void* x, const float* w, float* wt, void* y, int B, int H, int W, int C, int R, int S, int sh, int sw,
                   int ph, int pw, void* stream) {
  DwGeom g;
  MPR_REQUIRE(x && w && wt && y && dw_geom(&g, B, H, W, C, R, S, sh, sw, ph, pw), "mpr_dwconv_fwd: bad arguments (C %% 8 == 0 needed, C=%d)", C);
  hipStream_t st = (hipStream_t)stream;
  dw_pack_kernel<<<ceil_div(C * R * S, 256), 256, 0, st>>>(w, wt, C, R * S);
  if (dw_tiled_shape(g)) dw_fwd_tiled_launch(g, (const uint4*)x, wt, (uint4*)y, st);
  else dw_fwd_kernel<<<dw_grid((long long)B * g.P * g.Q * (C / 8)), 256, 0, st>>>((const uint4*)x, wt, (uint4*)y, g);
  MPR_LAUNCH_CHECK("dw_fwd_kernel");
  return MPR_OK;
}

int mpr_dwconv_dgrad(const void* dy, const float* w, float* wt, void* dx, int B, int H, int W, int C, int R, int S, int sh, int sw,
                     int ph, int pw, void* stream) {
  DwGeom g;
  MPR_REQUIRE(dy && w && wt && dx && dw_geom(&g, B, H, W, C, R, S, sh, sw, ph, pw), "mpr_dwconv_dgrad: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (dw_tiled_shape(g) && sh == 1 && g.P == H && g.Q == W) {
    // stride 1, "same" padding: the forward convolution of dy with the flipped filter
    dw_pack_kernel<<<ceil_div(C * R * S, 256), 256, 0, st>>>(w, wt, C, R * S, 1);
    dw_fwd_tiled_launch(g, (const uint4*)dy, wt, (uint4*)dx, st);
  } else if (dw_tiled_shape(g) && sh == 2) {
    dw_pack_kernel<<<ceil_div(C * R * S, 256), 256, 0, st>>>(w, wt, C, R * S);
    const unsigned grid = dw_grid((long long)B * H * ((W + 7) / 8) * (C / 8));
    if (S == 3) dw_dgrad_s2_tiled_kernel<3><<<grid, 256, 0, st>>>((const uint4*)dy, wt, (uint4*)dx, g);
    else dw_dgrad_s2_tiled_kernel<5><<<grid, 256, 0, st>>>((const uint4*)dy, wt, (uint4*)dx, g);
  } else {
    dw_pack_kernel<<<ceil_div(C * R * S, 256), 256, 0, st>>>(w, wt, C, R * S);
    dw_dgrad_kernel<<<dw_grid((long long)B * H * W * (C / 8)), 256, 0, st>>>((const uint4*)dy, wt, (uint4*)dx, g);
  }
  MPR_LAUNCH_CHECK("dw_dgrad_kernel");
  return MPR_OK;
}

long long mpr_dwconv_wgrad_workspace_floats(int B, int P, int Q, int C, int R, int S) {
  const long long npix = (long long)B * P * Q;
  const int pps = dw_pix_per_slab(npix);
  return (npix + pps - 1) / pps * R * S * C;
}

int mpr_dwconv_wgrad(const void* x, const void* dy, float* dw, float* workspace, int accumulate, int B, int H, int W, int C,
                     int R, int S, int sh, int sw, int ph, int pw, void* stream) {
  DwGeom g;
  MPR_REQUIRE(x && dy && dw && workspace && dw_geom(&g, B, H, W, C, R, S, sh, sw, ph, pw), "mpr_dwconv_wgrad: bad arguments");
  const long long npix = (long long)B * g.P * g.Q;
  const int pps = dw_pix_per_slab(npix);
  const int slabs = (int)((npix + pps - 1) / pps);
  hipStream_t st = (hipStream_t)stream;
  MPR_REQUIRE(S <= 5 && R <= 65535, "mpr_dwconv_wgrad: filter rows of at most 5 taps (S=%d)", S);
  if (dw_tiled_shape(g)) {
    const int G = C / 8, gp = ceil_div(G, ceil_div(G, 32));
    const long long nstrips = (long long)B * g.P * ((g.Q + 3) / 4);
    const int sps = (int)((nstrips + slabs - 1) / slabs);
    const dim3 grid(ceil_div(G, gp) * S, slabs);
#define MPR_DWW(S_, SW_) dw_wgrad_tiled_kernel<S_, SW_><<<grid, 256, 0, st>>>((const uint4*)x, (const uint4*)dy, workspace, g, sps, gp)
    if (S == 3 && sw == 1) MPR_DWW(3, 1); else if (S == 3) MPR_DWW(3, 2); else if (sw == 1) MPR_DWW(5, 1); else MPR_DWW(5, 2);
#undef MPR_DWW
  } else {
    const int gp = se_gp(C);
    dw_wgrad_kernel<<<dim3(ceil_div(C / 8, gp) * R, slabs), 256, 0, st>>>((const uint4*)x, (const uint4*)dy, workspace, g, pps, gp);
  }
  MPR_LAUNCH_CHECK("dw_wgrad_kernel");
  if (!accumulate) MPR_HIP(hipMemsetAsync(dw, 0, sizeof(float) * R * S * C, st));
  dw_wgrad_reduce_kernel<<<dim3(ceil_div(R * S * C, 256), slabs >= 32 ? 16 : 1), 256, 0, st>>>(workspace, dw, slabs, R * S, C);
  MPR_LAUNCH_CHECK("dw_wgrad_reduce_kernel");
  return MPR_OK;
}

int mpr_se_scale(const void* x, const float* gate, void* y, int B, int L, int C, void* stream) {
  MPR_REQUIRE(x && gate && y && B > 0 && L > 0 && C > 0 && C % 8 == 0, "mpr_se_scale: bad arguments");
  se_scale_kernel<<<dw_grid((long long)B * L * (C / 8)), 256, 0, (hipStream_t)stream>>>((const uint4*)x, gate, (uint4*)y, B, L, C);
  MPR_LAUNCH_CHECK("se_scale_kernel");
  return MPR_OK;
}

int mpr_se_dgate(const void* x, const void* dy, float* dgate, int B, int L, int C, void* stream) {
  MPR_REQUIRE(x && dy && dgate && B > 0 && B <= 65535 && L > 0 && C > 0 && C % 8 == 0, "mpr_se_dgate: bad arguments");
  const int gp = se_gp_sum(C);
  se_sum_kernel<true><<<dim3(ceil_div(C / 8, gp), B), 256, 0, (hipStream_t)stream>>>((const uint4*)x, (const uint4*)dy, dgate, L, C,
                                                                                     gp, 1.f);
  MPR_LAUNCH_CHECK("se_sum_kernel");
  return MPR_OK;
}

int mpr_se_pool(const void* x, float* pooled, int B, int L, int C, void* stream) {
  MPR_REQUIRE(x && pooled && B > 0 && B <= 65535 && L > 0 && C > 0 && C % 8 == 0, "mpr_se_pool: bad arguments");
  const int gp = se_gp_sum(C);
  se_sum_kernel<false><<<dim3(ceil_div(C / 8, gp), B), 256, 0, (hipStream_t)stream>>>((const uint4*)x, nullptr, pooled, L, C, gp,
                                                                                      1.f / (float)L);
  MPR_LAUNCH_CHECK("se_sum_kernel");
  return MPR_OK;
}

}  // extern "C"
