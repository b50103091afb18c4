// fp32 PARITY path of the conv stacks (timm ResNet behind src/image_encoder.py:16,24; ProfileCNN / _BasicBlock,
// src/profile_encoder.py:111-240): fp32 feature maps, fp32 filters, every product on the exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32: bitwise a k-ordered fmaf chain), statistics in double, NO atomics -- every sum has a fixed order,
// so two runs of one build are bit-identical.  Selected by `precision: 32` (trainer / card), as Lightning's flag selects
// fp32 in the reference; the bf16 kernels (conv_win / conv_igemm / conv_wgrad*) stay the throughput path.
// This is the mode the reference's fp32 CPU fixtures are compared against at 1e-4 / 1e-3 (tests/test_f32_path_gpu.py).
//
// One implicit-GEMM kernel, three operand mappings (64x64 tile, 4 waves x one 32x32 MFMA tile, reduction in chunks of 32
// staged k-major in LDS, next chunk prefetched into registers underneath the MFMAs):
//   forward        y [m=(b,p,q)][n=k]      = sum_(r,s,c)  x[b, p*sh-ph+r, q*sw-pw+s, c] * w[k,c,r,s]
//   data gradient  dx[m=(b,h,w)][n=c]      = sum_(r,s,k)  dy[b, (h+ph-r)/sh, (w+pw-s)/sw, k] * w[k,c,r,s]   (+ add)
//   weight grad.   dw[m=k][n=(r,s,c)]      = sum_(b,p,q)  dy[b,p,q,k] * x[b, p*sh-ph+r, q*sw-pw+s, c]
// Filters are addressed through element strides (sk, sc, sr, ss), so torch's OIHW memory and the channels-last [K][R][S][C]
// memory of the bf16 path's masters are both served without a copy.  The weight gradient splits its (long) pixel reduction
// over blockIdx.z into partial tiles that a second kernel sums in split order.
#include "common.h"

#define CF_BK 32
#define CF_LD 68

struct ConvF32 {
  const float* a;
  const float* b;
  float* out;
  const float* add;
  int B, H, W, C, K, R, S, sh, sw, ph, pw, P, Q;
  long long wk, wc, wr, ws;
  int M, N, Kred;
  int chunks_per_split, direct, accumulate;
  FastDiv d_in;    // innermost reduction / column extent: C (forward, weight gradient columns) or K (data gradient)
  FastDiv d_s;     // S
  FastDiv d_row;   // pixels per image row of the M (or pixel) index: Q (forward, weight gradient) or W (data gradient)
  FastDiv d_img;   // pixels per image: P*Q or H*W
};

template <int MODE>
__global__ __launch_bounds__(256) void conv_f32_kernel(const ConvF32 p) {
  __shared__ float As[CF_BK][CF_LD];
  __shared__ float Bs[CF_BK][CF_LD];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  int kbeg = 0, kend = p.Kred;
  if (MODE == 2) {
    kbeg = blockIdx.z * p.chunks_per_split * CF_BK;
    kend = min(p.Kred, kbeg + p.chunks_per_split * CF_BK);
  }

  // ---- per-thread staging coordinates (8 elements of each operand per chunk)
  // forward / data gradient, A: reduction index fastest (one (r,s,c|k) decode per chunk, 8 output pixels decoded once)
  // weight gradient, A and every n-fastest operand: column fastest, 8 reduction indices per chunk
  int pb[8], ph0[8], pw0[8];       // decoded pixels of the 8 A rows (MODE 0/1)
  bool pv[8];
  if (MODE != 2) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = m0 + (tid >> 5) + 8 * i;
      pv[i] = m < p.M;
      const uint32_t b = fdiv(m, p.d_img), rem = m - b * p.d_img.d, r = fdiv(rem, p.d_row), c = rem - r * p.d_row.d;
      pb[i] = b;
      if (MODE == 0) { ph0[i] = (int)r * p.sh - p.ph; pw0[i] = (int)c * p.sw - p.pw; }
      else           { ph0[i] = (int)r + p.ph;        pw0[i] = (int)c + p.pw; }
    }
  }
  // MODE 2: this thread's column n = (r, s, c) of the weight gradient
  int nr = 0, ns = 0, nc = 0;
  bool nvalid = false;
  if (MODE == 2) {
    const int n = n0 + (tid & 63);
    nvalid = n < p.N;
    const uint32_t rs = fdiv(n, p.d_in);
    nc = n - rs * p.d_in.d;
    nr = fdiv(rs, p.d_s);
    ns = rs - nr * p.d_s.d;
  }

  float ra[8], rb[8];
  auto load = [&](int k0) {
    if (MODE == 0) {
      const int kk = k0 + (tid & 31);
      const bool kv = kk < kend;
      const uint32_t rs = fdiv(kk, p.d_in), c = kk - rs * p.d_in.d, r = fdiv(rs, p.d_s), s = rs - r * p.d_s.d;
      const long long woff = (long long)c * p.wc + (long long)r * p.wr + (long long)s * p.ws;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int ih = ph0[i] + (int)r, iw = pw0[i] + (int)s;
        const bool ok = kv && pv[i] && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
        ra[i] = ok ? p.a[(((long long)pb[i] * p.H + ih) * p.W + iw) * p.C + c] : 0.f;
        const int n = n0 + (tid >> 5) + 8 * i;
        rb[i] = (kv && n < p.N) ? p.b[(long long)n * p.wk + woff] : 0.f;
      }
    } else if (MODE == 1) {
      {
        const int kk = k0 + (tid & 31);
        const bool kv = kk < kend;
        const uint32_t rs = fdiv(kk, p.d_in), k = kk - rs * p.d_in.d, r = fdiv(rs, p.d_s), s = rs - r * p.d_s.d;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int th = ph0[i] - (int)r, tw = pw0[i] - (int)s;
          bool ok = kv && pv[i] && th >= 0 && tw >= 0;
          int oh = th, ow = tw;
          if (p.sh != 1) { oh = th / p.sh; ok = ok && (oh * p.sh == th); }
          if (p.sw != 1) { ow = tw / p.sw; ok = ok && (ow * p.sw == tw); }
          ok = ok && oh < p.P && ow < p.Q;
          ra[i] = ok ? p.a[(((long long)pb[i] * p.P + oh) * p.Q + ow) * p.K + k] : 0.f;
        }
      }
      const int c = n0 + (tid & 63);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int kk = k0 + (tid >> 6) + 4 * i;
        const uint32_t rs = fdiv(kk, p.d_in), k = kk - rs * p.d_in.d, r = fdiv(rs, p.d_s), s = rs - r * p.d_s.d;
        rb[i] = (kk < kend && c < p.N)
                    ? p.b[(long long)k * p.wk + (long long)c * p.wc + (long long)r * p.wr + (long long)s * p.ws] : 0.f;
      }
    } else {
      const int k = m0 + (tid & 63);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int pix = k0 + (tid >> 6) + 4 * i;
        const bool kv = pix < kend;
        ra[i] = (kv && k < p.M) ? p.a[(long long)pix * p.K + k] : 0.f;
        const uint32_t b = fdiv(pix, p.d_img), rem = pix - b * p.d_img.d, op = fdiv(rem, p.d_row), oq = rem - op * p.d_row.d;
        const int ih = (int)op * p.sh - p.ph + nr, iw = (int)oq * p.sw - p.pw + ns;
        const bool ok = kv && nvalid && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
        rb[i] = ok ? p.b[(((long long)b * p.H + ih) * p.W + iw) * p.C + nc] : 0.f;
      }
    }
  };
  auto store = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) {
        As[tid & 31][(tid >> 5) + 8 * i] = ra[i];
        Bs[tid & 31][(tid >> 5) + 8 * i] = rb[i];
      } else if (MODE == 1) {
        As[tid & 31][(tid >> 5) + 8 * i] = ra[i];
        Bs[(tid >> 6) + 4 * i][tid & 63] = rb[i];
      } else {
        As[(tid >> 6) + 4 * i][tid & 63] = ra[i];
        Bs[(tid >> 6) + 4 * i][tid & 63] = rb[i];
      }
    }
  };

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;

  if (kbeg < kend) load(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += CF_BK) {
    store();
    __syncthreads();
    if (k0 + CF_BK < kend) load(k0 + CF_BK);
#pragma unroll
    for (int s2 = 0; s2 < CF_BK / 2; ++s2) {
      const float a = As[2 * s2 + (lane >> 5)][wm * 32 + (lane & 31)];
      const float b = Bs[2 * s2 + (lane >> 5)][wn * 32 + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }

  const int n = n0 + wn * 32 + (lane & 31);
  if (n >= p.N) return;
  long long wcol = 0;
  if (MODE == 2 && p.direct) {
    const uint32_t rs = fdiv(n, p.d_in), c = n - rs * p.d_in.d, r = fdiv(rs, p.d_s), s = rs - r * p.d_s.d;
    wcol = (long long)c * p.wc + (long long)r * p.wr + (long long)s * p.ws;
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int m = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
    if (m >= p.M) continue;
    if (MODE == 0) {
      p.out[(long long)m * p.N + n] = acc[e];
    } else if (MODE == 1) {
      const long long o = (long long)m * p.N + n;
      p.out[o] = p.add ? acc[e] + p.add[o] : acc[e];
    } else if (p.direct) {
      float* d = p.out + (long long)m * p.wk + wcol;
      *d = p.accumulate ? *d + acc[e] : acc[e];
    } else {
      p.out[((long long)blockIdx.z * p.M + m) * p.N + n] = acc[e];
    }
  }
}

// partial tiles [splits][K][R*S*C] -> dw through the filter's strides, summed in split order
__global__ __launch_bounds__(256) void conv_f32_wgrad_reduce_kernel(const float* partial, float* dw, int splits, int M, int N,
                                                                    long long wk, long long wc, long long wr, long long ws,
                                                                    FastDiv d_c, FastDiv d_s, int accumulate) {
  const long long total = (long long)M * N;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    float v = 0.f;
    for (int z = 0; z < splits; ++z) v += partial[z * total + i];
    const int m = (int)(i / N), n = (int)(i - (long long)m * N);
    const uint32_t rs = fdiv(n, d_c), c = n - rs * d_c.d, r = fdiv(rs, d_s), s = rs - r * d_s.d;
    float* d = dw + (long long)m * wk + (long long)c * wc + (long long)r * wr + (long long)s * ws;
    *d = accumulate ? *d + v : v;
  }
}

// ---------------------------------------------------------------------------------------------- BatchNorm (fp32 maps)
// statistics: per-channel sum and sum of squares in DOUBLE; partial[nparts][2][C]
__global__ __launch_bounds__(256) void f32_bn_stats_kernel(const float* x, double* partial, long long rows, int C) {
  __shared__ double red[2][4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  double s = 0.0, q = 0.0;
  if (c < C)
    for (long long r = (long long)blockIdx.y * 4 + rl; r < rows; r += (long long)gridDim.y * 4) {
      const double v = x[r * C + c];
      s += v;
      q += v * v;
    }
  red[0][rl][cl] = s;
  red[1][rl][cl] = q;
  __syncthreads();
  if (rl == 0 && c < C) {
    partial[((long long)blockIdx.y * 2 + 0) * C + c] = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
    partial[((long long)blockIdx.y * 2 + 1) * C + c] = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
  }
}

__global__ void f32_bn_finalize_kernel(const double* partial, int nparts, long long count, const float* gamma,
                                       const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                       float* scale, float* shift, float* mean, float* invstd, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0, q = 0.0;
  for (int i = 0; i < nparts; ++i) {
    s += partial[((long long)i * 2 + 0) * C + c];
    q += partial[((long long)i * 2 + 1) * C + c];
  }
  const double mu = s / (double)count;
  double var = q / (double)count - mu * mu;
  if (var < 0.0) var = 0.0;
  const float istd = (float)(1.0 / sqrt(var + (double)eps));
  mean[c] = (float)mu;
  invstd[c] = istd;
  const float sc = gamma[c] * istd;
  scale[c] = sc;
  shift[c] = beta[c] - (float)mu * sc;
  if (running_mean) {
    const double unb = count > 1 ? var * (double)count / (double)(count - 1) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
  }
}

__global__ __launch_bounds__(256) void f32_bn_apply_kernel(const float* x, const float* scale, const float* shift,
                                                            const float* residual, int relu, float* y, long long total,
                                                            int C) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    float v = x[i] * scale[c] + shift[c];
    if (residual) v += residual[i];
    if (relu) v = fmaxf(v, 0.f);
    y[i] = v;
  }
}

// backward sums: dz = dy * (y > 0) when mask_y != NULL;  partial[nparts][2][C] = (sum dz, sum dz * xhat), in double
__global__ __launch_bounds__(256) void f32_bn_bwd_reduce_kernel(const float* dy, const float* mask_y, const float* x,
                                                                 const float* mean, const float* invstd, double* partial,
                                                                 long long rows, int C) {
  __shared__ double red[2][4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  double s = 0.0, q = 0.0;
  if (c < C) {
    const float mu = mean[c], is = invstd[c];
    for (long long r = (long long)blockIdx.y * 4 + rl; r < rows; r += (long long)gridDim.y * 4) {
      float dz = dy[r * C + c];
      if (mask_y && !(mask_y[r * C + c] > 0.f)) dz = 0.f;
      s += (double)dz;
      q += (double)dz * (double)((x[r * C + c] - mu) * is);
    }
  }
  red[0][rl][cl] = s;
  red[1][rl][cl] = q;
  __syncthreads();
  if (rl == 0 && c < C) {
    partial[((long long)blockIdx.y * 2 + 0) * C + c] = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
    partial[((long long)blockIdx.y * 2 + 1) * C + c] = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
  }
}

// dgamma, dbeta (assigned or accumulated) and the coefficients of dx = k1*dz + k2*x + k3 (coef [3][C]); count == 0: eval
// mode (running statistics: dx = gamma * invstd * dz)
__global__ void f32_bn_bwd_finalize_kernel(const double* partial, int nparts, long long count, const float* gamma,
                                           const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                           int accumulate, float* coef, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0, q = 0.0;
  for (int i = 0; i < nparts; ++i) {
    s += partial[((long long)i * 2 + 0) * C + c];
    q += partial[((long long)i * 2 + 1) * C + c];
  }
  if (accumulate) { dgamma[c] += (float)q; dbeta[c] += (float)s; }
  else            { dgamma[c] = (float)q;  dbeta[c] = (float)s; }
  const double g = gamma[c], is = invstd[c], mu = mean[c];
  if (count > 0) {
    // dx = g*is*(dz - s/n - xhat*q/n),  xhat = (x - mu)*is
    const double k2 = -g * is * is * q / (double)count;
    coef[c] = (float)(g * is);
    coef[C + c] = (float)k2;
    coef[2 * C + c] = (float)(-g * is * s / (double)count - k2 * mu);
  } else {
    coef[c] = (float)(g * is);
    coef[C + c] = 0.f;
    coef[2 * C + c] = 0.f;
  }
}

__global__ __launch_bounds__(256) void f32_bn_bwd_apply_kernel(const float* dy, const float* mask_y, const float* x,
                                                                const float* coef, float* dx, float* dz_out, long long total,
                                                                int C) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    float dz = dy[i];
    if (mask_y && !(mask_y[i] > 0.f)) dz = 0.f;
    if (dz_out) dz_out[i] = dz;
    dx[i] = coef[c] * dz + coef[C + c] * x[i] + coef[2 * C + c];
  }
}

// ---------------------------------------------------------------------------------------------- pooling (fp32 maps)
// idx: linear input position ih*W + iw of the first maximum in torch's scan order (int32), -1 for an empty window
__global__ __launch_bounds__(256) void f32_maxpool_fwd_kernel(const float* x, float* y, int* idx, int B, int H, int W, int C,
                                                               int P, int Q, int RH, int RW, int SH, int SW, int PH, int PW) {
  const long long total = (long long)B * P * Q * C;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    long long t = i / C;
    const int q = (int)(t % Q);
    t /= Q;
    const int pp = (int)(t % P), b = (int)(t / P);
    float best = -INFINITY;
    int bi = -1;
    for (int r = 0; r < RH; ++r) {
      const int ih = pp * SH - PH + r;
      if (ih < 0 || ih >= H) continue;
      for (int s = 0; s < RW; ++s) {
        const int iw = q * SW - PW + s;
        if (iw < 0 || iw >= W) continue;
        const float v = x[(((long long)b * H + ih) * W + iw) * C + c];
        if (v > best || bi < 0 || v != v) {
          if (!(best != best)) { best = v; bi = ih * W + iw; }      // (a NaN, once taken, stays: torch propagates it)
        }
      }
    }
    y[i] = best;
    idx[i] = bi;
  }
}

// gather form: every input position sums the outputs whose arg-max it is (fixed order: no atomics)
__global__ __launch_bounds__(256) void f32_maxpool_bwd_kernel(const float* dy, const int* idx, float* dx, int B, int H, int W,
                                                               int C, int P, int Q, int RH, int RW, int SH, int SW, int PH,
                                                               int PW) {
  const long long total = (long long)B * H * W * C;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    long long t = i / C;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H), b = (int)(t / H);
    const int me = h * W + w;
    float g = 0.f;
    // outputs p with p*SH - PH <= h <= p*SH - PH + RH - 1
    const int p_lo = max(0, (h + PH - RH + SH) / SH), p_hi = min(P - 1, (h + PH) / SH);
    const int q_lo = max(0, (w + PW - RW + SW) / SW), q_hi = min(Q - 1, (w + PW) / SW);
    for (int pp = p_lo; pp <= p_hi; ++pp)
      for (int q = q_lo; q <= q_hi; ++q) {
        const long long o = (((long long)b * P + pp) * Q + q) * C + c;
        if (idx[o] == me) g += dy[o];
      }
    dx[i] = g;
  }
}

// [B][L][C] -> [B][C]: mean (mode 0, summed in position order in fp32 like torch's CPU mean over a small extent -- in
// double here, rounded once) or max + first arg-max (mode 1)
__global__ __launch_bounds__(256) void f32_global_pool_fwd_kernel(const float* x, float* y, int* idx, int B, int L, int C,
                                                                   int mode) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i - b * C;
  const float* px = x + (long long)b * L * C + c;
  if (mode == 0) {
    double s = 0.0;
    for (int l = 0; l < L; ++l) s += (double)px[(long long)l * C];
    y[i] = (float)(s / (double)L);
  } else {
    float best = px[0];
    int bi = 0;
    for (int l = 1; l < L; ++l) {
      const float v = px[(long long)l * C];
      if (v > best || (v != v && !(best != best))) { best = v; bi = l; }
    }
    y[i] = best;
    idx[i] = bi;
  }
}

__global__ __launch_bounds__(256) void f32_global_pool_bwd_kernel(const float* dy, const int* idx, float* dx, int B, int L,
                                                                   int C, int mode) {
  const long long total = (long long)B * L * C;
  const float inv = 1.f / (float)L;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long long t = i / C;
    const int l = (int)(t % L), b = (int)(t / L);
    const float g = dy[(long long)b * C + c];
    dx[i] = mode == 0 ? g * inv : (idx[(long long)b * C + c] == l ? g : 0.f);
  }
}

// ================================================================================================ C ABI
static inline int ew_grid(long long total) {
  long long g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

static int fill_geom(ConvF32& p, int B, int H, int W, int C, int K, int R, int S, int sh, int sw, int ph, int pw,
                     long long wk, long long wc, long long wr, long long ws, const char* who) {
  MPR_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && K > 0 && R > 0 && S > 0 && sh > 0 && sw > 0 && ph >= 0 && pw >= 0,
              "%s: bad geometry", who);
  p.B = B; p.H = H; p.W = W; p.C = C; p.K = K; p.R = R; p.S = S; p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw;
  p.P = (H + 2 * ph - R) / sh + 1;
  p.Q = (W + 2 * pw - S) / sw + 1;
  MPR_REQUIRE(p.P > 0 && p.Q > 0, "%s: empty output (%d x %d)", who, p.P, p.Q);
  MPR_REQUIRE((long long)B * p.P * p.Q < (1ll << 31) && (long long)B * H * W < (1ll << 31) &&
                  (long long)R * S * C < (1ll << 31) && (long long)R * S * K < (1ll << 31),
              "%s: index range", who);
  p.wk = wk; p.wc = wc; p.wr = wr; p.ws = ws;
  p.d_s = make_fastdiv(S);
  p.chunks_per_split = 0; p.direct = 1; p.accumulate = 0; p.add = nullptr;
  return MPR_OK;
}

extern "C" {

int mpr_f32_conv_fwd(const float* x, const float* w, long long sk, long long sc, long long sr, long long ss, float* y, int B,
                     int H, int W, int C, int K, int R, int S, int sh, int sw, int ph, int pw, void* stream) {
  MPR_REQUIRE(x && w && y, "mpr_f32_conv_fwd: null pointer");
  ConvF32 p;
  if (int e = fill_geom(p, B, H, W, C, K, R, S, sh, sw, ph, pw, sk, sc, sr, ss, "mpr_f32_conv_fwd")) return e;
  p.a = x; p.b = w; p.out = y;
  p.M = B * p.P * p.Q; p.N = K; p.Kred = R * S * C;
  p.d_in = make_fastdiv(C); p.d_row = make_fastdiv(p.Q); p.d_img = make_fastdiv(p.P * p.Q);
  dim3 grid(ceil_div(p.N, 64), ceil_div(p.M, 64), 1);
  MPR_REQUIRE(grid.y <= 65535u * 1024u, "mpr_f32_conv_fwd: too many rows");
  conv_f32_kernel<0><<<grid, 256, 0, (hipStream_t)stream>>>(p);
  MPR_LAUNCH_CHECK("conv_f32_kernel<fwd>");
  return MPR_OK;
}

int mpr_f32_conv_dgrad(const float* dy, const float* w, long long sk, long long sc, long long sr, long long ss, float* dx,
                       const float* add, int B, int H, int W, int C, int K, int R, int S, int sh, int sw, int ph, int pw,
                       void* stream) {
  MPR_REQUIRE(dy && w && dx, "mpr_f32_conv_dgrad: null pointer");
  ConvF32 p;
  if (int e = fill_geom(p, B, H, W, C, K, R, S, sh, sw, ph, pw, sk, sc, sr, ss, "mpr_f32_conv_dgrad")) return e;
  p.a = dy; p.b = w; p.out = dx; p.add = add;
  p.M = B * H * W; p.N = C; p.Kred = R * S * K;
  p.d_in = make_fastdiv(K); p.d_row = make_fastdiv(W); p.d_img = make_fastdiv(H * W);
  dim3 grid(ceil_div(p.N, 64), ceil_div(p.M, 64), 1);
  conv_f32_kernel<1><<<grid, 256, 0, (hipStream_t)stream>>>(p);
  MPR_LAUNCH_CHECK("conv_f32_kernel<dgrad>");
  return MPR_OK;
}

static int wgrad_splits(int B, int P, int Q, int K, int R, int S, int C) {
  const long long pix = (long long)B * P * Q;
  const int chunks = (int)((pix + CF_BK - 1) / CF_BK);
  const int tiles = ceil_div(K, 64) * ceil_div(R * S * C, 64);
  int splits = 1024 / tiles;
  if (splits < 1) splits = 1;
  if (splits > chunks) splits = chunks;
  if (splits > 256) splits = 256;
  return splits;
}

/* floats of scratch mpr_f32_conv_wgrad needs for this geometry (0: it writes the gradient directly) */
long long mpr_f32_conv_wgrad_scratch_floats(int B, int H, int W, int C, int K, int R, int S, int sh, int sw, int ph, int pw) {
  const int P = (H + 2 * ph - R) / sh + 1, Q = (W + 2 * pw - S) / sw + 1;
  if (B <= 0 || P <= 0 || Q <= 0) return 0;
  const int splits = wgrad_splits(B, P, Q, K, R, S, C);
  return splits > 1 ? (long long)splits * K * R * S * C : 0;
}

int mpr_f32_conv_wgrad(const float* x, const float* dy, float* dw, long long sk, long long sc, long long sr, long long ss,
                       int accumulate, float* scratch, long long scratch_floats, int B, int H, int W, int C, int K, int R,
                       int S, int sh, int sw, int ph, int pw, void* stream) {
  MPR_REQUIRE(x && dy && dw, "mpr_f32_conv_wgrad: null pointer");
  ConvF32 p;
  if (int e = fill_geom(p, B, H, W, C, K, R, S, sh, sw, ph, pw, sk, sc, sr, ss, "mpr_f32_conv_wgrad")) return e;
  p.a = dy; p.b = x;
  p.M = K; p.N = R * S * C; p.Kred = B * p.P * p.Q;
  p.d_in = make_fastdiv(C); p.d_row = make_fastdiv(p.Q); p.d_img = make_fastdiv(p.P * p.Q);
  const int splits = wgrad_splits(B, p.P, p.Q, K, R, S, C);
  const int chunks = ceil_div(p.Kred, CF_BK);
  p.chunks_per_split = ceil_div(chunks, splits);
  const int nz = ceil_div(chunks, p.chunks_per_split);
  p.accumulate = accumulate;
  if (nz > 1) {
    MPR_REQUIRE(scratch && scratch_floats >= (long long)nz * p.M * p.N,
                "mpr_f32_conv_wgrad: %lld floats of scratch needed (mpr_f32_conv_wgrad_scratch_floats), got %lld",
                (long long)nz * p.M * p.N, scratch ? scratch_floats : 0ll);
    p.direct = 0;
    p.out = scratch;
  } else {
    p.direct = 1;
    p.out = dw;
  }
  dim3 grid(ceil_div(p.N, 64), ceil_div(p.M, 64), nz);
  conv_f32_kernel<2><<<grid, 256, 0, (hipStream_t)stream>>>(p);
  MPR_LAUNCH_CHECK("conv_f32_kernel<wgrad>");
  if (nz > 1) {
    conv_f32_wgrad_reduce_kernel<<<ew_grid((long long)p.M * p.N), 256, 0, (hipStream_t)stream>>>(
        scratch, dw, nz, p.M, p.N, sk, sc, sr, ss, p.d_in, p.d_s, accumulate);
    MPR_LAUNCH_CHECK("conv_f32_wgrad_reduce_kernel");
  }
  return MPR_OK;
}

/* partial rows (of 2*C doubles) mpr_f32_bn_stats / mpr_f32_bn_bwd_reduce write */
int mpr_f32_bn_parts(long long rows, int C) {
  long long n = (rows + 63) / 64;
  return (int)(n < 1 ? 1 : (n > 128 ? 128 : n));
}

int mpr_f32_bn_stats(const float* x, void* partial, long long rows, int C, void* stream) {
  MPR_REQUIRE(x && partial && rows > 0 && C > 0, "mpr_f32_bn_stats: bad arguments");
  dim3 grid(ceil_div(C, 64), mpr_f32_bn_parts(rows, C));
  f32_bn_stats_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, (double*)partial, rows, C);
  MPR_LAUNCH_CHECK("f32_bn_stats_kernel");
  return MPR_OK;
}

int mpr_f32_bn_finalize(const void* partial, int nparts, long long count, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, float momentum, float eps, float* scale, float* shift,
                        float* mean, float* invstd, int C, void* stream) {
  MPR_REQUIRE(partial && gamma && beta && scale && shift && mean && invstd && nparts > 0 && count > 0 && C > 0,
              "mpr_f32_bn_finalize: bad arguments");
  f32_bn_finalize_kernel<<<ceil_div(C, 64), 64, 0, (hipStream_t)stream>>>((const double*)partial, nparts, count, gamma, beta,
                                                                          running_mean, running_var, momentum, eps, scale,
                                                                          shift, mean, invstd, C);
  MPR_LAUNCH_CHECK("f32_bn_finalize_kernel");
  return MPR_OK;
}

int mpr_f32_bn_apply(const float* x, const float* scale, const float* shift, const float* residual, int relu, float* y,
                     long long rows, int C, void* stream) {
  MPR_REQUIRE(x && scale && shift && y && rows > 0 && C > 0, "mpr_f32_bn_apply: bad arguments");
  f32_bn_apply_kernel<<<ew_grid(rows * C), 256, 0, (hipStream_t)stream>>>(x, scale, shift, residual, relu, y, rows * C, C);
  MPR_LAUNCH_CHECK("f32_bn_apply_kernel");
  return MPR_OK;
}

int mpr_f32_bn_bwd_reduce(const float* dy, const float* mask_y, const float* x, const float* mean, const float* invstd,
                          void* partial, long long rows, int C, void* stream) {
  MPR_REQUIRE(dy && x && mean && invstd && partial && rows > 0 && C > 0, "mpr_f32_bn_bwd_reduce: bad arguments");
  dim3 grid(ceil_div(C, 64), mpr_f32_bn_parts(rows, C));
  f32_bn_bwd_reduce_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(dy, mask_y, x, mean, invstd, (double*)partial, rows, C);
  MPR_LAUNCH_CHECK("f32_bn_bwd_reduce_kernel");
  return MPR_OK;
}

int mpr_f32_bn_bwd_finalize(const void* partial, int nparts, long long count, const float* gamma, const float* mean,
                            const float* invstd, float* dgamma, float* dbeta, int accumulate, float* coef, int C,
                            void* stream) {
  MPR_REQUIRE(partial && gamma && mean && invstd && dgamma && dbeta && coef && nparts > 0 && count >= 0 && C > 0,
              "mpr_f32_bn_bwd_finalize: bad arguments");
  f32_bn_bwd_finalize_kernel<<<ceil_div(C, 64), 64, 0, (hipStream_t)stream>>>((const double*)partial, nparts, count, gamma,
                                                                              mean, invstd, dgamma, dbeta, accumulate, coef,
                                                                              C);
  MPR_LAUNCH_CHECK("f32_bn_bwd_finalize_kernel");
  return MPR_OK;
}

int mpr_f32_bn_bwd_apply(const float* dy, const float* mask_y, const float* x, const float* coef, float* dx, float* dz_out,
                         long long rows, int C, void* stream) {
  MPR_REQUIRE(dy && x && coef && dx && rows > 0 && C > 0, "mpr_f32_bn_bwd_apply: bad arguments");
  f32_bn_bwd_apply_kernel<<<ew_grid(rows * C), 256, 0, (hipStream_t)stream>>>(dy, mask_y, x, coef, dx, dz_out, rows * C, C);
  MPR_LAUNCH_CHECK("f32_bn_bwd_apply_kernel");
  return MPR_OK;
}

int mpr_f32_maxpool_fwd(const float* x, float* y, int* idx, int B, int H, int W, int C, int RH, int RW, int SH, int SW, int PH,
                        int PW, void* stream) {
  MPR_REQUIRE(x && y && idx && B > 0 && H > 0 && W > 0 && C > 0 && RH > 0 && RW > 0 && SH > 0 && SW > 0,
              "mpr_f32_maxpool_fwd: bad arguments");
  const int P = (H + 2 * PH - RH) / SH + 1, Q = (W + 2 * PW - RW) / SW + 1;
  MPR_REQUIRE(P > 0 && Q > 0 && (long long)H * W < (1ll << 31), "mpr_f32_maxpool_fwd: bad geometry");
  f32_maxpool_fwd_kernel<<<ew_grid((long long)B * P * Q * C), 256, 0, (hipStream_t)stream>>>(x, y, idx, B, H, W, C, P, Q, RH,
                                                                                             RW, SH, SW, PH, PW);
  MPR_LAUNCH_CHECK("f32_maxpool_fwd_kernel");
  return MPR_OK;
}

int mpr_f32_maxpool_bwd(const float* dy, const int* idx, float* dx, int B, int H, int W, int C, int RH, int RW, int SH, int SW,
                        int PH, int PW, void* stream) {
  MPR_REQUIRE(dy && idx && dx && B > 0 && H > 0 && W > 0 && C > 0, "mpr_f32_maxpool_bwd: bad arguments");
  const int P = (H + 2 * PH - RH) / SH + 1, Q = (W + 2 * PW - RW) / SW + 1;
  MPR_REQUIRE(P > 0 && Q > 0, "mpr_f32_maxpool_bwd: bad geometry");
  f32_maxpool_bwd_kernel<<<ew_grid((long long)B * H * W * C), 256, 0, (hipStream_t)stream>>>(dy, idx, dx, B, H, W, C, P, Q, RH,
                                                                                             RW, SH, SW, PH, PW);
  MPR_LAUNCH_CHECK("f32_maxpool_bwd_kernel");
  return MPR_OK;
}

int mpr_f32_global_pool_fwd(const float* x, float* y, int* idx, int B, int L, int C, int mode, void* stream) {
  MPR_REQUIRE(x && y && (mode == 0 || idx) && B > 0 && L > 0 && C > 0 && (mode == 0 || mode == 1),
              "mpr_f32_global_pool_fwd: bad arguments");
  f32_global_pool_fwd_kernel<<<ceil_div(B * C, 256), 256, 0, (hipStream_t)stream>>>(x, y, idx, B, L, C, mode);
  MPR_LAUNCH_CHECK("f32_global_pool_fwd_kernel");
  return MPR_OK;
}

int mpr_f32_global_pool_bwd(const float* dy, const int* idx, float* dx, int B, int L, int C, int mode, void* stream) {
  MPR_REQUIRE(dy && dx && (mode == 0 || idx) && B > 0 && L > 0 && C > 0 && (mode == 0 || mode == 1),
              "mpr_f32_global_pool_bwd: bad arguments");
  f32_global_pool_bwd_kernel<<<ew_grid((long long)B * L * C), 256, 0, (hipStream_t)stream>>>(dy, idx, dx, B, L, C, mode);
  MPR_LAUNCH_CHECK("f32_global_pool_bwd_kernel");
  return MPR_OK;
}

}  // extern "C"
