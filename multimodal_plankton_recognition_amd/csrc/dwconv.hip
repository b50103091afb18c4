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
__global__ __launch_bounds__(256) void dw_pack_kernel(const float* __restrict__ w, float* __restrict__ wt, int C, int RS) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= C * RS) return;
  const int tap = i / C, c = i - tap * C;
  wt[i] = w[(size_t)c * RS + tap];
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
  const int i = blockIdx.x * 256 + threadIdx.x;       // = tap * C + c
  if (i >= RS * C) return;
  const int per = (nslabs + gridDim.y - 1) / gridDim.y;
  int s0 = blockIdx.y * per, s1 = s0 + per;
  if (s1 > nslabs) s1 = nslabs;
  float a = 0.f;
  for (int sl = s0; sl < s1; ++sl) a += part[(size_t)sl * RS * C + i];
  const int tap = i / C, c = i - tap * C;
  atomicAdd(dw + (size_t)c * RS + tap, a);
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
  if (cg < G) {
    for (int l = rl; l < L; l += nl) {
      float a[8];
      const size_t o = ((size_t)b * L + l) * G + cg;
      unpack8(x[o], a);
      if (MUL) {
        float d[8];
        unpack8(dy[o], d);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(a[e], d[e], acc[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += a[e];
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

static inline int dw_pix_per_slab(long long npix) {      // at most 1024 slabs of at least 256 pixels
  long long p = (npix + 1023) / 1024;
  return (int)(p < 256 ? 256 : p);
}

extern "C" {

int mpr_dwconv_fwd(const void* x, const float* w, float* wt, void* y, int B, int H, int W, int C, int R, int S, int sh, int sw,
                   int ph, int pw, void* stream) {
  DwGeom g;
  MPR_REQUIRE(x && w && wt && y && dw_geom(&g, B, H, W, C, R, S, sh, sw, ph, pw), "mpr_dwconv_fwd: bad arguments (C %% 8 == 0 needed, C=%d)", C);
  dw_pack_kernel<<<ceil_div(C * R * S, 256), 256, 0, (hipStream_t)stream>>>(w, wt, C, R * S);
  dw_fwd_kernel<<<dw_grid((long long)B * g.P * g.Q * (C / 8)), 256, 0, (hipStream_t)stream>>>((const uint4*)x, wt, (uint4*)y, g);
  MPR_LAUNCH_CHECK("dw_fwd_kernel");
  return MPR_OK;
}

int mpr_dwconv_dgrad(const void* dy, const float* w, float* wt, void* dx, int B, int H, int W, int C, int R, int S, int sh, int sw,
                     int ph, int pw, void* stream) {
  DwGeom g;
  MPR_REQUIRE(dy && w && wt && dx && dw_geom(&g, B, H, W, C, R, S, sh, sw, ph, pw), "mpr_dwconv_dgrad: bad arguments");
  dw_pack_kernel<<<ceil_div(C * R * S, 256), 256, 0, (hipStream_t)stream>>>(w, wt, C, R * S);
  dw_dgrad_kernel<<<dw_grid((long long)B * H * W * (C / 8)), 256, 0, (hipStream_t)stream>>>((const uint4*)dy, wt, (uint4*)dx, g);
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
  const int gp = se_gp(C);
  dw_wgrad_kernel<<<dim3(ceil_div(C / 8, gp) * R, slabs), 256, 0, st>>>((const uint4*)x, (const uint4*)dy, workspace, g, pps, gp);
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
  const int gp = se_gp(C);
  se_sum_kernel<true><<<dim3(ceil_div(C / 8, gp), B), 256, 0, (hipStream_t)stream>>>((const uint4*)x, (const uint4*)dy, dgate, L, C,
                                                                                     gp, 1.f);
  MPR_LAUNCH_CHECK("se_sum_kernel");
  return MPR_OK;
}

int mpr_se_pool(const void* x, float* pooled, int B, int L, int C, void* stream) {
  MPR_REQUIRE(x && pooled && B > 0 && B <= 65535 && L > 0 && C > 0 && C % 8 == 0, "mpr_se_pool: bad arguments");
  const int gp = se_gp(C);
  se_sum_kernel<false><<<dim3(ceil_div(C / 8, gp), B), 256, 0, (hipStream_t)stream>>>((const uint4*)x, nullptr, pooled, L, C, gp,
                                                                                      1.f / (float)L);
  MPR_LAUNCH_CHECK("se_sum_kernel");
  return MPR_OK;
}

}  // extern "C"
