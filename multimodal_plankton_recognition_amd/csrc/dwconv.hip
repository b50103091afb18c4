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

// y[b][p][q][c] = sum_{r,s} x[b][p*sh-ph+r][q*sw-pw+s][c] * w[c][r][s]       (w: torch [C][1][R][S] fp32)
__global__ __launch_bounds__(256) void dw_fwd_kernel(const uint4* __restrict__ x, const float* __restrict__ w,
                                                     uint4* __restrict__ y, const DwGeom g) {
  const int G = g.C / 8, RS = g.R * g.S;
  const long long total = (long long)g.B * g.P * g.Q * G;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int cg = (int)(idx % G);
    long long pix = idx / G;
    const int q = (int)(pix % g.Q);
    pix /= g.Q;
    const int p = (int)(pix % g.P), b = (int)(pix / g.P);
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
        const float* wp = w + (size_t)cg * 8 * RS + r * g.S + s;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(f[e], wp[e * RS], acc[e]);
      }
    }
    y[idx] = pack8(acc);
  }
}

// dx[b][h][w][c] = sum_{r,s : (h+ph-r) % sh == 0, ...} dy[b][(h+ph-r)/sh][(w+pw-s)/sw][c] * w[c][r][s]
__global__ __launch_bounds__(256) void dw_dgrad_kernel(const uint4* __restrict__ dy, const float* __restrict__ w,
                                                       uint4* __restrict__ dx, const DwGeom g) {
  const int G = g.C / 8, RS = g.R * g.S;
  const long long total = (long long)g.B * g.H * g.W * G;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int cg = (int)(idx % G);
    long long pix = idx / G;
    const int ww = (int)(pix % g.W);
    pix /= g.W;
    const int h = (int)(pix % g.H), b = (int)(pix / g.H);
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
        const float* wp = w + (size_t)cg * 8 * RS + r * g.S + s;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(f[e], wp[e * RS], acc[e]);
      }
    }
    dx[idx] = pack8(acc);
  }
}

// partial weight gradients of one slab of output pixels: part[slab][tap][C]; thread = (tap, channel group)
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const uint4* __restrict__ x, const uint4* __restrict__ dy,
                                                       float* __restrict__ part, const DwGeom g, int pix_per_slab) {
  const int G = g.C / 8, RS = g.R * g.S;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= RS * G) return;
  const int tap = t / G, cg = t - tap * G;
  const int r = tap / g.S, s = tap - r * g.S;
  const long long npix = (long long)g.B * g.P * g.Q;
  const long long p0 = (long long)blockIdx.y * pix_per_slab;
  long long p1 = p0 + pix_per_slab;
  if (p1 > npix) p1 = npix;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  for (long long pix = p0; pix < p1; ++pix) {
    const int q = (int)(pix % g.Q);
    const long long bp = pix / g.Q;
    const int p = (int)(bp % g.P), b = (int)(bp / g.P);
    const int h = p * g.sh - g.ph + r, ww = q * g.sw - g.pw + s;
    if (h < 0 || h >= g.H || ww < 0 || ww >= g.W) continue;
    float a[8], d[8];
    unpack8(x[((size_t)(b * g.H + h) * g.W + ww) * G + cg], a);
    unpack8(dy[(size_t)pix * G + cg], d);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = fmaf(a[e], d[e], acc[e]);
  }
  float* out = part + ((size_t)blockIdx.y * RS + tap) * g.C + cg * 8;
  *reinterpret_cast<float4*>(out) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  *reinterpret_cast<float4*>(out + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
}

// dw[c][r][s] (+)= sum_slab part[slab][tap][c]
__global__ __launch_bounds__(256) void dw_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int nslabs,
                                                              int RS, int C, int accumulate) {
  const int i = blockIdx.x * 256 + threadIdx.x;       // = tap * C + c
  if (i >= RS * C) return;
  double a = 0.0;
  for (int sl = 0; sl < nslabs; ++sl) a += (double)part[(size_t)sl * RS * C + i];
  const int tap = i / C, c = i - tap * C;
  float* o = dw + (size_t)c * RS + tap;
  *o = accumulate ? *o + (float)a : (float)a;
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

// dgate[b][c] = sum_l dy[b][l][c] * x[b][l][c];  block = 64 channel groups x 4 row lanes of one image
__global__ __launch_bounds__(256) void se_dgate_kernel(const uint4* __restrict__ x, const uint4* __restrict__ dy,
                                                       float* __restrict__ dgate, int L, int C) {
  __shared__ float red[3][8][64];
  const int G = C / 8, cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int cg = blockIdx.x * 64 + cl, b = blockIdx.y;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (cg < G) {
    for (int l = rl; l < L; l += 4) {
      float a[8], d[8];
      const size_t o = ((size_t)b * L + l) * G + cg;
      unpack8(x[o], a);
      unpack8(dy[o], d);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaf(a[e], d[e], acc[e]);
    }
  }
  if (rl > 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl - 1][e][cl] = acc[e];
  }
  __syncthreads();
  if (rl == 0 && cg < G) {
#pragma unroll
    for (int e = 0; e < 8; ++e) dgate[(size_t)b * C + cg * 8 + e] = acc[e] + red[0][e][cl] + red[1][e][cl] + red[2][e][cl];
  }
}

static inline unsigned dw_grid(long long n) {
  long long g = (n + 255) / 256;
  return (unsigned)(g < 16384 ? (g < 1 ? 1 : g) : 16384);
}

static inline bool dw_geom(DwGeom* g, int B, int H, int W, int C, int R, int S, int sh, int sw, int ph, int pw) {
  g->B = B; g->H = H; g->W = W; g->C = C; g->R = R; g->S = S; g->sh = sh; g->sw = sw; g->ph = ph; g->pw = pw;
  g->P = (H + 2 * ph - R) / sh + 1;
  g->Q = (W + 2 * pw - S) / sw + 1;
  return B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && R > 0 && S > 0 && sh > 0 && sw > 0 && g->P > 0 && g->Q > 0;
}

#define DW_PIX_PER_SLAB 256

extern "C" {

int mpr_dwconv_fwd(const void* x, const float* w, void* y, int B, int H, int W, int C, int R, int S, int sh, int sw, int ph,
                   int pw, void* stream) {
  DwGeom g;
  MPR_REQUIRE(x && w && y && dw_geom(&g, B, H, W, C, R, S, sh, sw, ph, pw), "mpr_dwconv_fwd: bad arguments (C %% 8 == 0 needed, C=%d)", C);
  dw_fwd_kernel<<<dw_grid((long long)B * g.P * g.Q * (C / 8)), 256, 0, (hipStream_t)stream>>>((const uint4*)x, w, (uint4*)y, g);
  MPR_LAUNCH_CHECK("dw_fwd_kernel");
  return MPR_OK;
}

int mpr_dwconv_dgrad(const void* dy, const float* w, void* dx, int B, int H, int W, int C, int R, int S, int sh, int sw, int ph,
                     int pw, void* stream) {
  DwGeom g;
  MPR_REQUIRE(dy && w && dx && dw_geom(&g, B, H, W, C, R, S, sh, sw, ph, pw), "mpr_dwconv_dgrad: bad arguments");
  dw_dgrad_kernel<<<dw_grid((long long)B * H * W * (C / 8)), 256, 0, (hipStream_t)stream>>>((const uint4*)dy, w, (uint4*)dx, g);
  MPR_LAUNCH_CHECK("dw_dgrad_kernel");
  return MPR_OK;
}

long long mpr_dwconv_wgrad_workspace_floats(int B, int P, int Q, int C, int R, int S) {
  const long long slabs = ((long long)B * P * Q + DW_PIX_PER_SLAB - 1) / DW_PIX_PER_SLAB;
  return slabs * R * S * C;
}

int mpr_dwconv_wgrad(const void* x, const void* dy, float* dw, float* workspace, int accumulate, int B, int H, int W, int C,
                     int R, int S, int sh, int sw, int ph, int pw, void* stream) {
  DwGeom g;
  MPR_REQUIRE(x && dy && dw && workspace && dw_geom(&g, B, H, W, C, R, S, sh, sw, ph, pw), "mpr_dwconv_wgrad: bad arguments");
  const long long npix = (long long)B * g.P * g.Q;
  const int slabs = (int)((npix + DW_PIX_PER_SLAB - 1) / DW_PIX_PER_SLAB);
  MPR_REQUIRE(slabs <= 65535, "mpr_dwconv_wgrad: too many pixel slabs (%d)", slabs);
  hipStream_t st = (hipStream_t)stream;
  dw_wgrad_kernel<<<dim3(ceil_div(R * S * (C / 8), 256), slabs), 256, 0, st>>>((const uint4*)x, (const uint4*)dy, workspace, g,
                                                                               DW_PIX_PER_SLAB);
  MPR_LAUNCH_CHECK("dw_wgrad_kernel");
  dw_wgrad_reduce_kernel<<<ceil_div(R * S * C, 256), 256, 0, st>>>(workspace, dw, slabs, R * S, C, accumulate);
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
  se_dgate_kernel<<<dim3(ceil_div(C / 8, 64), B), 256, 0, (hipStream_t)stream>>>((const uint4*)x, (const uint4*)dy, dgate, L, C);
  MPR_LAUNCH_CHECK("se_dgate_kernel");
  return MPR_OK;
}

}  // extern "C"
