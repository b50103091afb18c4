// Mixed-precision ("bf16-mixed") transformer path: timm's pre-norm ViT blocks behind src/image_encoder.py:16,24 and torch's
// post-norm nn.TransformerEncoderLayer of ProfileTransformer (src/profile_encoder.py:22-30,57-68).
//
// Layout: the residual stream is fp32 [rows = B*T][D]; everything that is a GEMM operand -- LayerNorm outputs, qkv,
// attention outputs, MLP activations and all their gradients -- is bf16, so the four linears of a block (96 % of a
// ViT's FLOPs) run as 1x1 implicit-GEMM convolutions on the bf16 MFMA kernels of conv_igemm.hip / conv_wgrad.hip
// (mpr_conv_fwd / mpr_conv_dgrad / mpr_conv_wgrad with R = S = 1).  This file holds what sits between those GEMMs:
//   * residual add (+ bias + dropout) fused with LayerNorm, one wave per row, the row cached in registers;
//   * LayerNorm backward fused with the residual-gradient add and the dgamma/dbeta partial sums;
//   * bias + GELU/ReLU (+ dropout) on bf16, and the backward elementwise passes fused with the bias-gradient column sums;
//   * fused attention for T <= 256 tokens and head size 32 / 64: one workgroup per (batch, head) keeps K and V (forward)
//     in LDS, S^T = K Q^T and O^T = V^T P^T on v_mfma_f32_32x32x16_bf16 with the softmax in registers between them
//     (the accumulator of the first product is the B operand of the second: no LDS round trip); backward recomputes P
//     from the saved log-sum-exp in two kernels -- query-on-lane for dQ, key-on-lane for dK / dV -- so no product needs a
//     cross-lane transpose or an atomic.  The n x n score matrix never reaches HBM.
// Dropout masks are never stored: element i of a tensor is kept iff hash(seed, i) >= p, regenerated in backward.
#include "common.h"

__device__ __forceinline__ uint32_t tb_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ bool tb_keep(uint32_t seed, unsigned long long i, float p) {
  const uint32_t h = tb_mix32(tb_mix32((uint32_t)i ^ seed) + 0x9e3779b9U * (seed | 1u) + (uint32_t)(i >> 32));
  return (float)(h >> 8) * (1.f / 16777216.f) >= p;
}
__device__ __forceinline__ void unpack4(const uint2& v, float* f) {
  f[0] = bf16lo(v.x); f[1] = bf16hi(v.x); f[2] = bf16lo(v.y); f[3] = bf16hi(v.y);
}
__device__ __forceinline__ uint2 pack4(const float* f) {
  uint2 v;
  v.x = pack_bf16x2(f[0], f[1]);
  v.y = pack_bf16x2(f[2], f[3]);
  return v;
}
// erf by Abramowitz-Stegun 7.1.26 (|error| < 1.5e-7, far below the bf16 rounding of the results): the exact-erf GELU of
// torch costs ~40 VALU instructions per element through erff, which made the MLP activation passes compute-bound
// (125 us for 310 MB); this form shares its one exponential with the Gaussian term of the derivative.
__device__ __forceinline__ void gelu_terms(float v, float* cdf, float* gauss) {
  const float z = fabsf(v) * 0.70710678118654752f;
  const float e = __expf(-z * z);                       // exp(-v^2 / 2)
  const float t = __frcp_rn(fmaf(0.3275911f, z, 1.f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float erf_abs = fmaf(-poly, e, 1.f);
  *cdf = 0.5f * (1.f + copysignf(erf_abs, v));
  *gauss = e;
}
__device__ __forceinline__ float act_fwd(float v, int act) {
  if (act == 1) {
    float cdf, g;
    gelu_terms(v, &cdf, &g);
    return v * cdf;
  }
  if (act == 2) return fmaxf(v, 0.f);
  if (act == 3) return v / (1.f + __expf(-v));            // SiLU
  if (act == 4) return 1.f / (1.f + __expf(-v));          // sigmoid
  return v;
}
__device__ __forceinline__ float act_grad(float v, int act) {
  if (act == 1) {
    float cdf, g;
    gelu_terms(v, &cdf, &g);
    return fmaf(v * 0.3989422804014327f, g, cdf);
  }
  if (act == 2) return v > 0.f ? 1.f : 0.f;
  if (act == 3 || act == 4) {
    const float sg = 1.f / (1.f + __expf(-v));
    return act == 3 ? sg * fmaf(v, 1.f - sg, 1.f) : sg * (1.f - sg);
  }
  return 1.f;
}

// ------------------------------------------------------------------------------------------ add + LayerNorm forward
struct AddLnParams {
  const float* x;          // [rows][D] residual stream (NULL: zeros, i.e. s = drop(r + rbias) as fp32)
  const uint16_t* r;       // bf16 branch output added to it (or NULL)
  const float* rbias;      // bias of the branch's last linear (or NULL)
  float p_drop;
  uint32_t seed;
  const float* gamma;      // NULL: no LayerNorm (only s_out = x + drop(r + rbias))
  const float* beta;
  float eps;
  float* s_out;            // x + drop(r + rbias)   (or NULL)
  float* y32;              // LN output, fp32 (or NULL)
  uint16_t* y16;           // LN output, bf16 (or NULL)
  float* mean;
  float* rstd;
  int rows, D;
};

// one wave per row; lane owns columns 4*(lane + 64 i) .. +3, i < NV
template <int NV>
__global__ __launch_bounds__(256) void tf_add_ln_fwd_kernel(const AddLnParams p) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= p.rows) return;
  const size_t o = (size_t)row * p.D;
  float v[NV][4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = 4 * (lane + 64 * i);
    if (j < p.D) {
      const float4 xv = p.x ? *reinterpret_cast<const float4*>(p.x + o + j) : make_float4(0.f, 0.f, 0.f, 0.f);
      v[i][0] = xv.x; v[i][1] = xv.y; v[i][2] = xv.z; v[i][3] = xv.w;
      if (p.r) {
        float a[4];
        unpack4(*reinterpret_cast<const uint2*>(p.r + o + j), a);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float t = a[c] + (p.rbias ? p.rbias[j + c] : 0.f);
          if (p.p_drop > 0.f) t = tb_keep(p.seed, o + j + c, p.p_drop) ? t / (1.f - p.p_drop) : 0.f;
          v[i][c] += t;
        }
      }
      if (p.s_out) *reinterpret_cast<float4*>(p.s_out + o + j) = make_float4(v[i][0], v[i][1], v[i][2], v[i][3]);
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    } else {
      v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f;
    }
  }
  if (!p.gamma) return;
  const float mu = wave_sum(s) / p.D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (4 * (lane + 64 * i) < p.D) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float d = v[i][c] - mu;
        q = fmaf(d, d, q);
      }
    }
  const float rs = 1.f / sqrtf(wave_sum(q) / p.D + p.eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = 4 * (lane + 64 * i);
    if (j < p.D) {
      const float4 g = *reinterpret_cast<const float4*>(p.gamma + j);
      const float4 b = *reinterpret_cast<const float4*>(p.beta + j);
      float y[4];
      y[0] = (v[i][0] - mu) * rs * g.x + b.x;
      y[1] = (v[i][1] - mu) * rs * g.y + b.y;
      y[2] = (v[i][2] - mu) * rs * g.z + b.z;
      y[3] = (v[i][3] - mu) * rs * g.w + b.w;
      if (p.y32) *reinterpret_cast<float4*>(p.y32 + o + j) = make_float4(y[0], y[1], y[2], y[3]);
      if (p.y16) *reinterpret_cast<uint2*>(p.y16 + o + j) = pack4(y);
    }
  }
  if (lane == 0) { p.mean[row] = mu; p.rstd[row] = rs; }
}

// ------------------------------------------------------------------------------------------ LayerNorm backward
// ds = rstd * (g - mean(g) - xhat * mean(g * xhat)) (+ dskip),  g = (dy16 [+ dy32]) * gamma;  the workgroup's partial
// dgamma / dbeta (its waves walk rows_per_block rows, the lane's columns are fixed) go to part[block][2][D]
struct LnBwdParams {
  const uint16_t* dy16;    // bf16 gradient of the LN output (from a dgrad GEMM), or NULL
  const float* dy32;       // fp32 gradient of the LN output (post-norm: the LN output is also the residual stream), or NULL
  const float* s;          // LN input
  const float* gamma;
  const float* mean;
  const float* rstd;
  const float* dskip;      // added to ds (or NULL)
  float* ds;
  float* part;
  int rows, D, rows_per_block;
};

template <int NV>
__global__ __launch_bounds__(256) void tf_ln_bwd_kernel(const LnBwdParams p) {
  __shared__ float red[3][2][NV * 256];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r0 = blockIdx.x * p.rows_per_block;
  int r1 = r0 + p.rows_per_block;
  if (r1 > p.rows) r1 = p.rows;
  float dg[NV][4], db[NV][4], gm[NV][4];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = 4 * (lane + 64 * i);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      dg[i][c] = 0.f;
      db[i][c] = 0.f;
      gm[i][c] = j < p.D ? p.gamma[j + c] : 0.f;
    }
  }
  for (int row = r0 + w; row < r1; row += 4) {
    const size_t o = (size_t)row * p.D;
    const float mu = p.mean[row], rs = p.rstd[row];
    float dy[NV][4], xh[NV][4];
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int j = 4 * (lane + 64 * i);
      if (j < p.D) {
        if (p.dy16) unpack4(*reinterpret_cast<const uint2*>(p.dy16 + o + j), dy[i]);
        else dy[i][0] = dy[i][1] = dy[i][2] = dy[i][3] = 0.f;
        if (p.dy32) {
          const float4 t = *reinterpret_cast<const float4*>(p.dy32 + o + j);
          dy[i][0] += t.x; dy[i][1] += t.y; dy[i][2] += t.z; dy[i][3] += t.w;
        }
        const float4 sv = *reinterpret_cast<const float4*>(p.s + o + j);
        xh[i][0] = (sv.x - mu) * rs; xh[i][1] = (sv.y - mu) * rs; xh[i][2] = (sv.z - mu) * rs; xh[i][3] = (sv.w - mu) * rs;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float g = dy[i][c] * gm[i][c];
          a += g;
          b = fmaf(g, xh[i][c], b);
          dg[i][c] = fmaf(dy[i][c], xh[i][c], dg[i][c]);
          db[i][c] += dy[i][c];
        }
      }
    }
    a = wave_sum(a) / p.D;
    b = wave_sum(b) / p.D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int j = 4 * (lane + 64 * i);
      if (j < p.D) {
        float out[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) out[c] = rs * (dy[i][c] * gm[i][c] - a - xh[i][c] * b);
        if (p.dskip) {
          const float4 t = *reinterpret_cast<const float4*>(p.dskip + o + j);
          out[0] += t.x; out[1] += t.y; out[2] += t.z; out[3] += t.w;
        }
        *reinterpret_cast<float4*>(p.ds + o + j) = make_float4(out[0], out[1], out[2], out[3]);
      }
    }
  }
  // waves 1..3 hand their column partials to wave 0
  if (w > 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        red[w - 1][0][(i * 4 + c) * 64 + lane] = dg[i][c];
        red[w - 1][1][(i * 4 + c) * 64 + lane] = db[i][c];
      }
  }
  __syncthreads();
  if (w == 0) {
    float* out = p.part + (size_t)blockIdx.x * 2 * p.D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int j = 4 * (lane + 64 * i);
      if (j < p.D) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float a = dg[i][c], b = db[i][c];
          for (int u = 0; u < 3; ++u) {
            a += red[u][0][(i * 4 + c) * 64 + lane];
            b += red[u][1][(i * 4 + c) * 64 + lane];
          }
          out[j + c] = a;
          out[p.D + j + c] = b;
        }
      }
    }
  }
}

// out0[c] += sum_p part[p][c] (c < n0), out1[c - n0] += sum_p part[p][c] (n0 <= c < n): the partial rows are split over
// gridDim.y workgroups (a few-way fp32 atomic per column); the caller zeroes the outputs when it does not accumulate
__global__ __launch_bounds__(256) void tf_colsum_kernel(const float* __restrict__ part, float* __restrict__ out0,
                                                        float* __restrict__ out1, int nparts, int n, int n0) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= n) return;
  const int per = (nparts + gridDim.y - 1) / gridDim.y;
  int q = blockIdx.y * per, q1 = q + per;
  if (q1 > nparts) q1 = nparts;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (; q + 3 < q1; q += 4) {
    a0 += part[(size_t)q * n + c];
    a1 += part[(size_t)(q + 1) * n + c];
    a2 += part[(size_t)(q + 2) * n + c];
    a3 += part[(size_t)(q + 3) * n + c];
  }
  for (; q < q1; ++q) a0 += part[(size_t)q * n + c];
  atomicAdd(c < n0 ? out0 + c : out1 + (c - n0), (a0 + a1) + (a2 + a3));
}

// ------------------------------------------------------------------------------------------ elementwise, bf16
// y = drop(act(x + bias[col]))   (16-byte groups; 32-bit index arithmetic, the bias as two float4 per group)
__global__ __launch_bounds__(256) void tf_bias_act_fwd_kernel(const uint4* __restrict__ x, const float* __restrict__ bias,
                                                              int act, float p_drop, uint32_t seed, uint4* __restrict__ y,
                                                              long long ngroups, int D) {
  const uint32_t G = (uint32_t)D / 8, n = (uint32_t)ngroups, stride = gridDim.x * 256;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    float f[8];
    unpack8(x[i], f);
    if (bias) {
      const uint32_t col = (i % G) * 8;
      const float4 b0 = *reinterpret_cast<const float4*>(bias + col), b1 = *reinterpret_cast<const float4*>(bias + col + 4);
      f[0] += b0.x; f[1] += b0.y; f[2] += b0.z; f[3] += b0.w; f[4] += b1.x; f[5] += b1.y; f[6] += b1.z; f[7] += b1.w;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = act_fwd(f[e], act);
      if (p_drop > 0.f) v = tb_keep(seed, (unsigned long long)i * 8 + e, p_drop) ? v / (1.f - p_drop) : 0.f;
      f[e] = v;
    }
    y[i] = pack8(f);
  }
}

// Backward elementwise passes fused with the column sums a bias gradient needs.
//   MODE 0: column sums of a bf16 tensor only                                   (qkv bias)
//   MODE 1: dx = dy * dropmask/(1-p) * act'(x + bias)   (bf16 -> bf16)           (MLP activation; dbias = colsum(dx))
//   MODE 2: dx = bf16(dy32 * dropmask/(1-p))            (fp32 -> bf16)           (branch output added to the residual)
// Block = 64 column groups (8 columns each) x 4 row lanes, walks a slab of rows; 8 column sums per thread in registers,
// reduced over the row lanes in LDS and written as the slab's partial row part[slab][D] (summed by tf_colsum_kernel: a
// thousand workgroups adding into the same D addresses ran 10x slower than the pass itself).
struct EwBwdParams {
  const void* dy;
  const uint16_t* x;
  const float* bias;
  int act;
  float p_drop;
  uint32_t seed;
  uint16_t* dx;
  float* part;       // [gridDim.y][D] partial column sums (or NULL)
  int rows, D, rows_per_slab;
};

// the same passes without column sums (no bias behind the activation: SiLU of the conv stacks) -- flat over 16-byte groups, so
// that narrow tensors (C = 32) keep every lane busy
template <int MODE>
__global__ __launch_bounds__(256) void tf_ew_bwd_flat_kernel(const EwBwdParams p) {
  const long long ngroups = (long long)p.rows * p.D / 8;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < ngroups; i += (long long)gridDim.x * 256) {
    const size_t o = (size_t)i * 8;
    float g[8];
    if (MODE == 2) {
      const float4 a = *reinterpret_cast<const float4*>((const float*)p.dy + o);
      const float4 b = *reinterpret_cast<const float4*>((const float*)p.dy + o + 4);
      g[0] = a.x; g[1] = a.y; g[2] = a.z; g[3] = a.w; g[4] = b.x; g[5] = b.y; g[6] = b.z; g[7] = b.w;
    } else {
      unpack8(*reinterpret_cast<const uint4*>((const uint16_t*)p.dy + o), g);
    }
    if (p.p_drop > 0.f) {
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = tb_keep(p.seed, o + e, p.p_drop) ? g[e] / (1.f - p.p_drop) : 0.f;
    }
    if (MODE == 1 && p.act != 0) {
      float xv[8];
      unpack8(*reinterpret_cast<const uint4*>(p.x + o), xv);
      if (p.bias) {
        const int col = (int)(o % p.D);
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[e] += p.bias[col + e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] *= act_grad(xv[e], p.act);
    }
    *reinterpret_cast<uint4*>(p.dx + o) = pack8(g);
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void tf_ew_bwd_kernel(const EwBwdParams p) {
  __shared__ float red[3][8][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int col = (blockIdx.x * 64 + cl) * 8;
  const bool active = col < p.D;
  const int r0 = blockIdx.y * p.rows_per_slab;
  int r1 = r0 + p.rows_per_slab;
  if (r1 > p.rows) r1 = p.rows;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  float bv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bv[e] = (MODE == 1 && p.bias && active) ? p.bias[col + e] : 0.f;
  if (active) {
    for (int row = r0 + rl; row < r1; row += 4) {
      const size_t o = (size_t)row * p.D + col;
      float g[8];
      if (MODE == 2) {
        const float4 a = *reinterpret_cast<const float4*>((const float*)p.dy + o);
        const float4 b = *reinterpret_cast<const float4*>((const float*)p.dy + o + 4);
        g[0] = a.x; g[1] = a.y; g[2] = a.z; g[3] = a.w; g[4] = b.x; g[5] = b.y; g[6] = b.z; g[7] = b.w;
      } else {
        unpack8(*reinterpret_cast<const uint4*>((const uint16_t*)p.dy + o), g);
      }
      if (MODE != 0) {
        if (p.p_drop > 0.f) {
#pragma unroll
          for (int e = 0; e < 8; ++e) g[e] = tb_keep(p.seed, o + e, p.p_drop) ? g[e] / (1.f - p.p_drop) : 0.f;
        }
        if (MODE == 1 && p.act != 0) {
          float xv[8];
          unpack8(*reinterpret_cast<const uint4*>(p.x + o), xv);
#pragma unroll
          for (int e = 0; e < 8; ++e) g[e] *= act_grad(xv[e] + bv[e], p.act);
        }
        const uint4 out = pack8(g);
        *reinterpret_cast<uint4*>(p.dx + o) = out;
        unpack8(out, g);          // the bias gradient sums what the GEMMs will see
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += g[e];
    }
  }
  if (!p.part) return;
  if (rl > 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl - 1][e][cl] = acc[e];
  }
  __syncthreads();
  if (rl == 0 && active) {
    float out[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) out[e] = acc[e] + red[0][e][cl] + red[1][e][cl] + red[2][e][cl];
    float* const dst = p.part + (size_t)blockIdx.y * p.D + col;
    *reinterpret_cast<float4*>(dst) = make_float4(out[0], out[1], out[2], out[3]);
    *reinterpret_cast<float4*>(dst + 4) = make_float4(out[4], out[5], out[6], out[7]);
  }
}

// ------------------------------------------------------------------------------------------ fused attention
struct AttnParams {
  const uint16_t* qkv;     // [B][T][3*heads*HD] bf16 (q | k | v, head-major inside each), WITHOUT the in-projection bias
  const float* bias;       // [3*heads*HD] or NULL -- added while the operands are loaded
  const unsigned char* mask;   // [B][T], 1 = padding key (or NULL)
  uint16_t* out;           // [B][T][heads*HD] bf16
  float* lse;              // [B*heads][T] log-sum-exp of the scaled scores
  const uint16_t* dout;    // backward: gradient of out
  float* delta;            // backward: [B*heads][T] rowsum(dO * O)
  uint16_t* dqkv;          // backward: [B][T][3*heads*HD] bf16
  int B, T, heads;
  float scale, p_drop;
  uint32_t seed;
};

__device__ __forceinline__ bf16x8 as_frag(const uint4& v) { return __builtin_bit_cast(bf16x8, v); }

// 8 bf16 of `src` (+ 8 floats of bias) -> rounded bf16
__device__ __forceinline__ uint4 add_bias8(const uint4& v, const float* bias) {
  if (!bias) return v;
  float f[8];
  unpack8(v, f);
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] += bias[e];
  return pack8(f);
}

// Stage rows [0, Tp) x HD of one (token-major, row stride `ld` elements) operand into LDS: row image (row stride RS bytes)
// and / or transposed image (row stride TS bytes, [HD][Tp]); rows >= T are zero.
// Staging of one (token-major, row stride `ld` elements) operand of a (batch, head) into LDS, in two phases so that the
// loads of SEVERAL operands are in flight together before the first LDS write (one memory latency per kernel, not one per
// operand): stage_load issues the 16-byte loads of rows [0, Tp) (rows >= T: zeros), stage_store adds the bias and writes the
// row image (row stride RS bytes) and / or the transposed image ([HD][Tp], row stride TS bytes).
template <int HD>
struct StageRegs {
  static constexpr int G = HD / 8, IT = G / 2;      // 64 NB threads cover 32 NB rows x G chunks in G / 2 rounds
  uint4 v[IT];
};

template <int HD, int NT>
__device__ __forceinline__ void stage_load(StageRegs<HD>& r, const uint16_t* __restrict__ src, size_t ld, int T) {
  constexpr int G = HD / 8;
#pragma unroll
  for (int it = 0; it < StageRegs<HD>::IT; ++it) {
    const int idx = threadIdx.x + NT * it;
    const int t = idx / G, g = idx - t * G;
    r.v[it] = make_uint4(0, 0, 0, 0);
    if (t < T) r.v[it] = *reinterpret_cast<const uint4*>(src + (size_t)t * ld + 8 * g);
  }
}

template <int HD, int NT>
__device__ __forceinline__ void stage_store(const StageRegs<HD>& r, const float* __restrict__ bias, int T, int Tp,
                                            unsigned char* rows, int RS, unsigned char* trans, int TS) {
  constexpr int G = HD / 8;
#pragma unroll
  for (int it = 0; it < StageRegs<HD>::IT; ++it) {
    const int idx = threadIdx.x + NT * it;
    const int t = idx / G, g = idx - t * G;
    if (t >= Tp) continue;
    const uint4 w = t < T ? add_bias8(r.v[it], bias ? bias + 8 * g : nullptr) : r.v[it];
    if (rows) *reinterpret_cast<uint4*>(rows + t * RS + g * 16) = w;
    if (trans) {
      const uint32_t wds[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int e = 0; e < 8; ++e)
        *reinterpret_cast<uint16_t*>(trans + (8 * g + e) * TS + t * 2) = (uint16_t)(wds[e >> 1] >> (16 * (e & 1)));
    }
  }
}

// operand of k-step s (keys / queries 16 s .. 16 s + 15 of block j, permuted as an accumulator delivers them) read from a
// transposed image: element i of lane half hh is column 32 j + 16 s + 8 (i >> 2) + 4 hh + (i & 3) of row `r`
__device__ __forceinline__ bf16x8 trans_frag(const unsigned char* img, int TS, int row, int j, int s, int hh) {
  const unsigned char* a = img + row * TS + (32 * j + 16 * s + 4 * hh) * 2;
  const uint2 lo = *reinterpret_cast<const uint2*>(a);
  const uint2 hi = *reinterpret_cast<const uint2*>(a + 16);
  return as_frag(make_uint4(lo.x, lo.y, hi.x, hi.y));
}

// The same operand read from the ROW image ([Tp][HD], row stride RS bytes) with the transposing LDS read: a 16-lane group
// reads a 4 (rows) x 16 (columns) block -- lane (lq = (lane & 15) >> 2, lp = lane & 3) supplies the address of 4 columns
// of row lq, lane l16 of the group receives rows 0..3 of column l16 -- so two reads (rows + 0 and + 8) deliver the two
// groups of four consecutive keys / queries a lane needs at its own column 32 d + (lane & 31).  No transposed copy in LDS
// (29 KB per operand at 224 tokens, and its staging: eight 2-byte LDS writes per 16 bytes).
typedef __attribute__((ext_vector_type(4))) short tf_s16x4;
typedef __attribute__((ext_vector_type(8))) short tf_s16x8;
__device__ __forceinline__ bf16x8 trans_frag_rows(uint32_t img_lds, int RS, int d, int j, int s, int lane) {
  const int g16 = lane >> 4, lq = (lane & 15) >> 2, lp = lane & 3, hh = lane >> 5;
  const uint32_t a = img_lds + (uint32_t)((32 * j + 16 * s + 4 * hh + lq) * RS + (32 * d + 16 * (g16 & 1) + 4 * lp) * 2);
  tf_s16x4 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(a + (uint32_t)(8 * RS)) : "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi)::"memory");
  return __builtin_bit_cast(bf16x8, (tf_s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

__device__ __forceinline__ bf16x8 acc_frag(const float* v) {   // 8 accumulator registers -> bf16 operand
  return as_frag(pack8(v));
}

template <int HD, int NB>
__global__ __launch_bounds__(64 * NB) void attn_fwd_kernel(const AttnParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int RS = HD * 2 + 16, NC = HD / 16, ND = HD / 32;
  const int nblk = (p.T + 31) >> 5, Tp = nblk * 32, TS = Tp * 2 + 8;
  unsigned char* const Ks = smem;
  unsigned char* const Vs = smem + Tp * RS;
  const uint32_t vs_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)Vs;
  const int b = blockIdx.x / p.heads, h = blockIdx.x - b * p.heads;
  const int dm = p.heads * HD;
  const size_t d3 = 3 * (size_t)dm;
  const uint16_t* const base = p.qkv + (size_t)b * p.T * d3 + h * HD;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
  const int q = 32 * w + r, qc = q < p.T ? q : p.T - 1;
  StageRegs<HD> rk, rv;
  stage_load<HD, 64 * NB>(rk, base + dm, d3, p.T);
  stage_load<HD, 64 * NB>(rv, base + 2 * dm, d3, p.T);
  uint4 qraw[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) qraw[c] = *reinterpret_cast<const uint4*>(base + (size_t)qc * d3 + 16 * c + 8 * hh);
  stage_store<HD, 64 * NB>(rk, p.bias ? p.bias + dm + h * HD : nullptr, p.T, Tp, Ks, RS, nullptr, 0);
  stage_store<HD, 64 * NB>(rv, p.bias ? p.bias + 2 * dm + h * HD : nullptr, p.T, Tp, Vs, RS, nullptr, 0);
  __syncthreads();
  if (w >= nblk) return;
  bf16x8 qf[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) qf[c] = as_frag(add_bias8(qraw[c], p.bias ? p.bias + h * HD + 16 * c + 8 * hh : nullptr));
  // The scores of one 32-key block are 16 accumulator registers; keeping all of them (8-9 blocks: 128-144 VGPRs) through
  // the softmax put the kernel at 203 VGPRs -- one 8-wave workgroup per CU, nothing running beside its staging and softmax
  // phases.  They are formed TWICE instead (28 more MFMAs per wave, identical values): once for the row maximum, once for
  // exp / sum / P V, one block at a time.
  auto scores = [&](int j) -> f32x16 {
    f32x16 sj;
#pragma unroll
    for (int e = 0; e < 16; ++e) sj[e] = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const bf16x8 kf = as_frag(*reinterpret_cast<const uint4*>(Ks + (32 * j + r) * RS + (16 * c + 8 * hh) * 2));
      sj = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[c], sj, 0, 0, 0);
    }
    return sj;
  };
  // softmax over the keys of this lane's query, in as few VALU instructions per score as possible (they, not the
  // MFMAs, are this kernel's arithmetic): invalid keys (tail of the last block / padding mask) are excluded through a
  // per-block bit mask that is only built where it can matter; max over the RAW scores (scale > 0), p = exp2(s * c - m * c)
  // with c = scale * log2(e) as one FMA + v_exp_f32; the 1 / sum normalisation moves behind the P V product.
  const unsigned char* const mrow = p.mask ? p.mask + (size_t)b * p.T : nullptr;
  auto bad_bits = [&](int j) -> uint32_t {      // bit e = accumulator register e of block j holds an invalid key
    uint32_t bad = 0;
    if (mrow || j == nblk - 1) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = 32 * j + (e & 3) + 8 * (e >> 2) + 4 * hh;
        if (!(key < p.T && !(mrow && mrow[key]))) bad |= 1u << e;
      }
    }
    return bad;
  };
  float m = -INFINITY;
  for (int j = 0; j < nblk; ++j) {
    const f32x16 sj = scores(j);
    const uint32_t bad = bad_bits(j);
#pragma unroll
    for (int e = 0; e < 16; ++e)
      if (!((bad >> e) & 1u)) m = fmaxf(m, sj[e]);
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  if (m == -INFINITY) m = 0.f;
  const float c2 = p.scale * 1.4426950408889634f, mc = -m * c2;
  float sum = 0.f;
  const unsigned long long rowi = ((unsigned long long)blockIdx.x * p.T + q) * p.T;
  f32x16 o[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
  for (int j = 0; j < nblk; ++j) {
    f32x16 sj = scores(j);
    const uint32_t bad = bad_bits(j);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float pv = __builtin_amdgcn_exp2f(fmaf(sj[e], c2, mc));
      if ((bad >> e) & 1u) pv = 0.f;
      sj[e] = pv;
      sum += pv;
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      if (p.p_drop > 0.f) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int e = 8 * s2 + i;
          const int key = 32 * j + (e & 3) + 8 * (e >> 2) + 4 * hh;
          if (!tb_keep(p.seed, rowi + key, p.p_drop)) sj[e] = 0.f;
        }
      }
      float pv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) pv[i] = sj[8 * s2 + i];
      const bf16x8 pf = acc_frag(pv);
#pragma unroll
      for (int d = 0; d < ND; ++d)
        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trans_frag_rows(vs_lds, RS, d, j, s2, lane), pf, o[d], 0, 0, 0);
    }
  }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.f / sum;
  const float keep_scale = p.p_drop > 0.f ? inv / (1.f - p.p_drop) : inv;
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[d][e] *= keep_scale;
  if (q < p.T) {
    uint16_t* const orow = p.out + ((size_t)b * p.T + q) * dm + h * HD;
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float f[4] = {o[d][4 * g], o[d][4 * g + 1], o[d][4 * g + 2], o[d][4 * g + 3]};
        *reinterpret_cast<uint2*>(orow + 32 * d + 8 * g + 4 * hh) = pack4(f);
      }
    if (hh == 0) p.lse[(size_t)blockIdx.x * p.T + q] = m * p.scale + __logf(sum);
  }
#endif
}

// backward, part 1 (query on the lane): delta = rowsum(dO * O), dQ = dS K
template <int HD, int NB>
__global__ __launch_bounds__(64 * NB) void attn_bwd_dq_kernel(const AttnParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int RS = HD * 2 + 16, NC = HD / 16, ND = HD / 32;
  const int nblk = (p.T + 31) >> 5, Tp = nblk * 32, TS = Tp * 2 + 8;
  unsigned char* const Ks = smem;
  unsigned char* const Vs = smem + Tp * RS;
  const uint32_t ks_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)Ks;
  const int b = blockIdx.x / p.heads, h = blockIdx.x - b * p.heads;
  const int dm = p.heads * HD;
  const size_t d3 = 3 * (size_t)dm;
  const uint16_t* const base = p.qkv + (size_t)b * p.T * d3 + h * HD;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
  const int q = 32 * w + r, qc = q < p.T ? q : p.T - 1;
  const uint16_t* const dorow = p.dout + ((size_t)b * p.T + qc) * dm + h * HD;
  const uint16_t* const orow = p.out + ((size_t)b * p.T + qc) * dm + h * HD;
  StageRegs<HD> rk, rv;
  stage_load<HD, 64 * NB>(rk, base + dm, d3, p.T);
  stage_load<HD, 64 * NB>(rv, base + 2 * dm, d3, p.T);
  uint4 qraw[NC], dvraw[NC], ovraw[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    qraw[c] = *reinterpret_cast<const uint4*>(base + (size_t)qc * d3 + 16 * c + 8 * hh);
    dvraw[c] = *reinterpret_cast<const uint4*>(dorow + 16 * c + 8 * hh);
    ovraw[c] = *reinterpret_cast<const uint4*>(orow + 16 * c + 8 * hh);
  }
  const float lse = p.lse[(size_t)blockIdx.x * p.T + qc];
  stage_store<HD, 64 * NB>(rk, p.bias ? p.bias + dm + h * HD : nullptr, p.T, Tp, Ks, RS, nullptr, 0);
  stage_store<HD, 64 * NB>(rv, p.bias ? p.bias + 2 * dm + h * HD : nullptr, p.T, Tp, Vs, RS, nullptr, 0);
  __syncthreads();
  if (w >= nblk) return;
  bf16x8 qf[NC], dof[NC];
  float delta = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    qf[c] = as_frag(add_bias8(qraw[c], p.bias ? p.bias + h * HD + 16 * c + 8 * hh : nullptr));
    const uint4 dv = dvraw[c];
    const uint4 ov = ovraw[c];
    dof[c] = as_frag(dv);
    float a[8], bq[8];
    unpack8(dv, a);
    unpack8(ov, bq);
#pragma unroll
    for (int e = 0; e < 8; ++e) delta = fmaf(a[e], bq[e], delta);
  }
  delta += __shfl_xor(delta, 32, 64);
  if (q < p.T && hh == 0) p.delta[(size_t)blockIdx.x * p.T + q] = delta;
  const unsigned char* const mrow = p.mask ? p.mask + (size_t)b * p.T : nullptr;
  const unsigned long long rowi = ((unsigned long long)blockIdx.x * p.T + q) * p.T;
  const float dscale = p.p_drop > 0.f ? 1.f / (1.f - p.p_drop) : 1.f;
  const float c2 = p.scale * 1.4426950408889634f, nlse2 = -lse * 1.4426950408889634f;
  f32x16 dq[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) dq[d][e] = 0.f;
  for (int j = 0; j < nblk; ++j) {
    f32x16 s, dp;
#pragma unroll
    for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const bf16x8 kf = as_frag(*reinterpret_cast<const uint4*>(Ks + (32 * j + r) * RS + (16 * c + 8 * hh) * 2));
      const bf16x8 vf = as_frag(*reinterpret_cast<const uint4*>(Vs + (32 * j + r) * RS + (16 * c + 8 * hh) * 2));
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[c], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[c], dp, 0, 0, 0);
    }
    float ds[16];
    const bool check = mrow || j == nblk - 1;        // (wave-uniform) only the last key block has a tail, only masks mask
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = 32 * j + (e & 3) + 8 * (e >> 2) + 4 * hh;
      float pr = __builtin_amdgcn_exp2f(fmaf(s[e], c2, nlse2));
      if (check && !(key < p.T && !(mrow && mrow[key]))) pr = 0.f;
      float dpe = dp[e] * dscale;
      if (p.p_drop > 0.f && !tb_keep(p.seed, rowi + key, p.p_drop)) dpe = 0.f;
      ds[e] = pr * (dpe - delta) * p.scale;
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const bf16x8 dsf = acc_frag(ds + 8 * s2);
#pragma unroll
      for (int d = 0; d < ND; ++d)
        dq[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trans_frag_rows(ks_lds, RS, d, j, s2, lane), dsf, dq[d], 0, 0, 0);
    }
  }
  if (q < p.T) {
    uint16_t* const drow = p.dqkv + ((size_t)b * p.T + q) * d3 + h * HD;
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float f[4] = {dq[d][4 * g], dq[d][4 * g + 1], dq[d][4 * g + 2], dq[d][4 * g + 3]};
        *reinterpret_cast<uint2*>(drow + 32 * d + 8 * g + 4 * hh) = pack4(f);
      }
  }
#endif
}

// backward, part 2 (key on the lane): dV^T = dO^T P, dK^T = Q^T dS
template <int HD, int NB>
__global__ __launch_bounds__(64 * NB) void attn_bwd_dkv_kernel(const AttnParams p) {      // (176 VGPRs: one 8-wave workgroup per CU; a 128-VGPR cap spills 82)
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int RS = HD * 2 + 16, NC = HD / 16, ND = HD / 32;
  const int nblk = (p.T + 31) >> 5, Tp = nblk * 32, TS = Tp * 2 + 8;
  unsigned char* const Qs = smem;
  unsigned char* const Os = smem + Tp * RS;
  float* const lse_s = reinterpret_cast<float*>(smem + 2 * Tp * RS);
  const uint32_t qs_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)Qs;
  const uint32_t os_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)Os;
  float* const del_s = lse_s + Tp;
  const int b = blockIdx.x / p.heads, h = blockIdx.x - b * p.heads;
  const int dm = p.heads * HD;
  const size_t d3 = 3 * (size_t)dm;
  const uint16_t* const base = p.qkv + (size_t)b * p.T * d3 + h * HD;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
  const int key = 32 * w + r, kc = key < p.T ? key : p.T - 1;
  StageRegs<HD> rq, ro;
  stage_load<HD, 64 * NB>(rq, base, d3, p.T);
  stage_load<HD, 64 * NB>(ro, p.dout + (size_t)b * p.T * dm + h * HD, dm, p.T);
  uint4 kraw[NC], vraw[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    kraw[c] = *reinterpret_cast<const uint4*>(base + dm + (size_t)kc * d3 + 16 * c + 8 * hh);
    vraw[c] = *reinterpret_cast<const uint4*>(base + 2 * dm + (size_t)kc * d3 + 16 * c + 8 * hh);
  }
  for (int t = threadIdx.x; t < Tp; t += 64 * NB) {
    lse_s[t] = t < p.T ? -1.4426950408889634f * p.lse[(size_t)blockIdx.x * p.T + t] : 0.f;     // (-lse * log2 e)
    del_s[t] = t < p.T ? p.delta[(size_t)blockIdx.x * p.T + t] : 0.f;
  }
  stage_store<HD, 64 * NB>(rq, p.bias ? p.bias + h * HD : nullptr, p.T, Tp, Qs, RS, nullptr, 0);
  stage_store<HD, 64 * NB>(ro, nullptr, p.T, Tp, Os, RS, nullptr, 0);
  __syncthreads();
  if (w >= nblk) return;
  const bool key_valid = key < p.T && !(p.mask && p.mask[(size_t)b * p.T + kc]);
  bf16x8 kf[NC], vf[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    kf[c] = as_frag(add_bias8(kraw[c], p.bias ? p.bias + dm + h * HD + 16 * c + 8 * hh : nullptr));
    vf[c] = as_frag(add_bias8(vraw[c], p.bias ? p.bias + 2 * dm + h * HD + 16 * c + 8 * hh : nullptr));
  }
  const float dscale = p.p_drop > 0.f ? 1.f / (1.f - p.p_drop) : 1.f;
  const float c2 = p.scale * 1.4426950408889634f;
  f32x16 dk[ND], dv[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) { dk[d][e] = 0.f; dv[d][e] = 0.f; }
  for (int i = 0; i < nblk; ++i) {
    f32x16 s, dp;
#pragma unroll
    for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const bf16x8 qa = as_frag(*reinterpret_cast<const uint4*>(Qs + (32 * i + r) * RS + (16 * c + 8 * hh) * 2));
      const bf16x8 oa = as_frag(*reinterpret_cast<const uint4*>(Os + (32 * i + r) * RS + (16 * c + 8 * hh) * 2));
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[c], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oa, vf[c], dp, 0, 0, 0);
    }
    float pd[16], ds[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int q = 32 * i + (e & 3) + 8 * (e >> 2) + 4 * hh;
      const bool valid = key_valid && q < p.T;
      const float pr = valid ? __builtin_amdgcn_exp2f(fmaf(s[e], c2, lse_s[q])) : 0.f;
      float keepf = dscale;
      if (p.p_drop > 0.f && !tb_keep(p.seed, ((unsigned long long)blockIdx.x * p.T + q) * p.T + key, p.p_drop)) keepf = 0.f;
      pd[e] = pr * keepf;
      ds[e] = pr * (dp[e] * keepf - del_s[q]) * p.scale;
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const bf16x8 pf = acc_frag(pd + 8 * s2);
      const bf16x8 dsf = acc_frag(ds + 8 * s2);
#pragma unroll
      for (int d = 0; d < ND; ++d) {
        dv[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trans_frag_rows(os_lds, RS, d, i, s2, lane), pf, dv[d], 0, 0, 0);
        dk[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trans_frag_rows(qs_lds, RS, d, i, s2, lane), dsf, dk[d], 0, 0, 0);
      }
    }
  }
  if (key < p.T) {
    uint16_t* const drow = p.dqkv + ((size_t)b * p.T + key) * d3 + h * HD;
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float fk[4] = {dk[d][4 * g], dk[d][4 * g + 1], dk[d][4 * g + 2], dk[d][4 * g + 3]};
        const float fv[4] = {dv[d][4 * g], dv[d][4 * g + 1], dv[d][4 * g + 2], dv[d][4 * g + 3]};
        *reinterpret_cast<uint2*>(drow + dm + 32 * d + 8 * g + 4 * hh) = pack4(fk);
        *reinterpret_cast<uint2*>(drow + 2 * dm + 32 * d + 8 * g + 4 * hh) = pack4(fv);
      }
  }
#endif
}

// ------------------------------------------------------------------------------------------ host side
// partial rows per workgroup of tf_colsum_kernel: ~16 (the kernel is a latency chain of dependent row reads), at most 128
// workgroups per column block adding into one address
static inline int colsum_splits(int nparts) {
  int s = nparts / 16;
  return s < 1 ? 1 : (s > 128 ? 128 : s);
}

static inline unsigned tb_grid(long long n) {
  long long g = (n + 255) / 256;
  return (unsigned)(g < 8192 ? (g < 1 ? 1 : g) : 8192);
}

template <typename K>
static int set_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
      mpr_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize, %zu): %s", bytes, hipGetErrorString(e));
      return MPR_EHIP;
    }
  }
  return MPR_OK;
}

extern "C" {

int mpr_tf_add_ln_fwd(const float* x, const void* r, const float* rbias, float p_drop, unsigned seed, const float* gamma,
                      const float* beta, float eps, float* s_out, float* y32, void* y16, float* mean, float* rstd,
                      int rows, int D, void* stream) {
  MPR_REQUIRE((x || r) && rows > 0 && D > 0 && D % 4 == 0 && D <= 2048, "mpr_tf_add_ln_fwd: needs D %% 4 == 0 and D <= 2048 (D=%d)", D);
  MPR_REQUIRE(!gamma || (beta && mean && rstd && (y32 || y16)), "mpr_tf_add_ln_fwd: LayerNorm needs beta, mean, rstd and an output");
  MPR_REQUIRE(gamma || s_out, "mpr_tf_add_ln_fwd: nothing to write");
  AddLnParams p = {x, (const uint16_t*)r, rbias, p_drop, seed, gamma, beta, eps, s_out, y32, (uint16_t*)y16, mean, rstd, rows, D};
  const dim3 grid(ceil_div(rows, 4));
  hipStream_t st = (hipStream_t)stream;
  const int nv = ceil_div(D, 256);
  if (nv <= 1) tf_add_ln_fwd_kernel<1><<<grid, 256, 0, st>>>(p);
  else if (nv <= 2) tf_add_ln_fwd_kernel<2><<<grid, 256, 0, st>>>(p);
  else if (nv <= 3) tf_add_ln_fwd_kernel<3><<<grid, 256, 0, st>>>(p);
  else if (nv <= 4) tf_add_ln_fwd_kernel<4><<<grid, 256, 0, st>>>(p);
  else tf_add_ln_fwd_kernel<8><<<grid, 256, 0, st>>>(p);
  MPR_LAUNCH_CHECK("tf_add_ln_fwd_kernel");
  return MPR_OK;
}

// rows per workgroup of the LayerNorm backward (4 waves walk them): fewer rows = more workgroups in flight for a pass that is
// bound by the latency of its row-by-row reductions, more partial rows for tf_colsum_kernel (tuning knob)
static int g_ln_bwd_rows = 32;
int mpr_tf_set_ln_bwd_rows(int rows) {
  const int old = g_ln_bwd_rows;
  if (rows >= 4) g_ln_bwd_rows = rows;
  return old;
}
#define TF_LN_BWD_ROWS g_ln_bwd_rows
int mpr_tf_ln_bwd_workspace_floats(int rows, int D) { return 2 * D * ceil_div(rows, TF_LN_BWD_ROWS); }

int mpr_tf_ln_bwd(const void* dy16, const float* dy32, const float* s, const float* gamma, const float* mean,
                  const float* rstd, const float* dskip, float* ds, float* dgamma, float* dbeta, float* workspace,
                  int accumulate, int rows, int D, void* stream) {
  MPR_REQUIRE((dy16 || dy32) && s && gamma && mean && rstd && ds && dgamma && dbeta && workspace, "mpr_tf_ln_bwd: null pointer");
  MPR_REQUIRE(rows > 0 && D > 0 && D % 4 == 0 && D <= 1024, "mpr_tf_ln_bwd: needs D %% 4 == 0 and D <= 1024 (D=%d)", D);
  LnBwdParams p = {(const uint16_t*)dy16, dy32, s, gamma, mean, rstd, dskip, ds, workspace, rows, D, TF_LN_BWD_ROWS};
  const int nb = ceil_div(rows, TF_LN_BWD_ROWS);
  hipStream_t st = (hipStream_t)stream;
  const int nv = ceil_div(D, 256);
  if (nv <= 1) tf_ln_bwd_kernel<1><<<nb, 256, 0, st>>>(p);
  else if (nv <= 2) tf_ln_bwd_kernel<2><<<nb, 256, 0, st>>>(p);
  else if (nv <= 3) tf_ln_bwd_kernel<3><<<nb, 256, 0, st>>>(p);
  else tf_ln_bwd_kernel<4><<<nb, 256, 0, st>>>(p);
  MPR_LAUNCH_CHECK("tf_ln_bwd_kernel");
  if (!accumulate) {
    MPR_HIP(hipMemsetAsync(dgamma, 0, sizeof(float) * D, st));
    MPR_HIP(hipMemsetAsync(dbeta, 0, sizeof(float) * D, st));
  }
  tf_colsum_kernel<<<dim3(ceil_div(2 * D, 256), colsum_splits(nb)), 256, 0, st>>>(workspace, dgamma, dbeta, nb, 2 * D, D);
  MPR_LAUNCH_CHECK("tf_colsum_kernel");
  return MPR_OK;
}

int mpr_tf_bias_act_fwd(const void* x, const float* bias, int act, float p_drop, unsigned seed, void* y, long long rows, int D,
                        void* stream) {
  MPR_REQUIRE(x && y && rows > 0 && D > 0 && D % 8 == 0 && act >= 0 && act <= 4, "mpr_tf_bias_act_fwd: bad arguments (D=%d)", D);
  const long long ng = rows * D / 8;
  MPR_REQUIRE(ng < (1ll << 31), "mpr_tf_bias_act_fwd: tensor too large (%lld groups)", ng);
  tf_bias_act_fwd_kernel<<<tb_grid(ng), 256, 0, (hipStream_t)stream>>>((const uint4*)x, bias, act, p_drop, seed, (uint4*)y, ng, D);
  MPR_LAUNCH_CHECK("tf_bias_act_fwd_kernel");
  return MPR_OK;
}

static inline void ew_bwd_grid(int rows, int D, int* gx, int* slabs, int* rps) {
  *gx = ceil_div(D, 512);
  int n = 4096 / *gx;          // ~16 workgroups per CU: enough loads in flight to cover the memory latency
  if (n > ceil_div(rows, 8)) n = ceil_div(rows, 8);
  if (n < 1) n = 1;
  *rps = ceil_div(rows, n);
  *slabs = ceil_div(rows, *rps);
}

int mpr_tf_ew_bwd_workspace_floats(int rows, int D) {
  int gx, slabs, rps;
  ew_bwd_grid(rows, D, &gx, &slabs, &rps);
  return slabs * D;
}

// mode 0: dbias += colsum(dy bf16); 1: dx = dy * drop * act'(x + bias) (bf16), dbias += colsum(dx);
// 2: dx = bf16(dy fp32 * drop), dbias += colsum(dx).  dbias may be NULL (modes 1, 2); workspace: see above.
int mpr_tf_ew_bwd(int mode, const void* dy, const void* x, const float* bias, int act, float p_drop, unsigned seed, void* dx,
                  float* dbias, float* workspace, int rows, int D, void* stream) {
  MPR_REQUIRE(dy && rows > 0 && D > 0 && D % 8 == 0 && mode >= 0 && mode <= 2, "mpr_tf_ew_bwd: bad arguments (mode=%d D=%d)", mode, D);
  MPR_REQUIRE(mode == 0 ? dbias != nullptr : dx != nullptr, "mpr_tf_ew_bwd: missing output");
  MPR_REQUIRE(!dbias || workspace, "mpr_tf_ew_bwd: the bias gradient needs the workspace");
  MPR_REQUIRE(!(mode == 1 && act != 0) || x, "mpr_tf_ew_bwd: the activation gradient needs x");
  int gx, slabs, rps;
  ew_bwd_grid(rows, D, &gx, &slabs, &rps);
  EwBwdParams p = {dy, (const uint16_t*)x, bias, act, p_drop, seed, (uint16_t*)dx, dbias ? workspace : nullptr, rows, D, rps};
  const dim3 grid(gx, slabs);
  hipStream_t st = (hipStream_t)stream;
  if (!dbias && mode != 0) {
    const unsigned fg = tb_grid((long long)rows * D / 8);
    if (mode == 1) tf_ew_bwd_flat_kernel<1><<<fg, 256, 0, st>>>(p);
    else tf_ew_bwd_flat_kernel<2><<<fg, 256, 0, st>>>(p);
    MPR_LAUNCH_CHECK("tf_ew_bwd_flat_kernel");
    return MPR_OK;
  }
  if (mode == 0) tf_ew_bwd_kernel<0><<<grid, 256, 0, st>>>(p);
  else if (mode == 1) tf_ew_bwd_kernel<1><<<grid, 256, 0, st>>>(p);
  else tf_ew_bwd_kernel<2><<<grid, 256, 0, st>>>(p);
  MPR_LAUNCH_CHECK("tf_ew_bwd_kernel");
  if (dbias) {
    tf_colsum_kernel<<<dim3(ceil_div(D, 256), colsum_splits(slabs)), 256, 0, st>>>(workspace, dbias, dbias, slabs, D, D);
    MPR_LAUNCH_CHECK("tf_colsum_kernel");
  }
  return MPR_OK;
}

// T <= 256 (8 waves, one 32-token block each); head size 32 also T <= 288 on 9 waves (LDS of the dK/dV kernel is the limit)
int mpr_attn_supported(int T, int head_dim) {
  return T >= 1 && ((head_dim == 64 && T <= 256) || (head_dim == 32 && T <= 288));
}

int mpr_attn_fwd(const void* qkv, const float* bias, const void* key_padding_mask, void* out, float* lse, int B, int T,
                 int heads, int head_dim, float scale, float p_drop, unsigned seed, void* stream) {
  MPR_REQUIRE(qkv && out && lse && B > 0 && heads > 0, "mpr_attn_fwd: bad arguments");
  MPR_REQUIRE(mpr_attn_supported(T, head_dim), "mpr_attn_fwd: head size 64 with T <= 256 or head size 32 with T <= 288 only (T=%d, head=%d)", T, head_dim);
  AttnParams p = {(const uint16_t*)qkv, bias, (const unsigned char*)key_padding_mask, (uint16_t*)out, lse, nullptr, nullptr,
                  nullptr, B, T, heads, scale, p_drop, seed};
  const int Tp = (T + 31) / 32 * 32;
  const size_t lds = 2 * (size_t)Tp * (head_dim * 2 + 16);      // K and V row images
  hipStream_t st = (hipStream_t)stream;
  if (head_dim == 64) {
    if (int rc = set_lds(attn_fwd_kernel<64, 8>, lds)) return rc;
    attn_fwd_kernel<64, 8><<<B * heads, 512, lds, st>>>(p);
  } else if (T <= 256) {
    if (int rc = set_lds(attn_fwd_kernel<32, 8>, lds)) return rc;
    attn_fwd_kernel<32, 8><<<B * heads, 512, lds, st>>>(p);
  } else {
    if (int rc = set_lds(attn_fwd_kernel<32, 9>, lds)) return rc;
    attn_fwd_kernel<32, 9><<<B * heads, 576, lds, st>>>(p);
  }
  MPR_LAUNCH_CHECK("attn_fwd_kernel");
  return MPR_OK;
}

int mpr_attn_bwd(const void* qkv, const float* bias, const void* key_padding_mask, const void* out, const void* dout,
                 const float* lse, float* delta, void* dqkv, int B, int T, int heads, int head_dim, float scale, float p_drop,
                 unsigned seed, void* stream) {
  MPR_REQUIRE(qkv && out && dout && lse && delta && dqkv && B > 0 && heads > 0, "mpr_attn_bwd: bad arguments");
  MPR_REQUIRE(mpr_attn_supported(T, head_dim), "mpr_attn_bwd: head size 64 with T <= 256 or head size 32 with T <= 288 only (T=%d, head=%d)", T, head_dim);
  AttnParams p = {(const uint16_t*)qkv, bias, (const unsigned char*)key_padding_mask, (uint16_t*)out, (float*)lse,
                  (const uint16_t*)dout, delta, (uint16_t*)dqkv, B, T, heads, scale, p_drop, seed};
  const int Tp = (T + 31) / 32 * 32;
  const size_t rs = head_dim * 2 + 16, ts = Tp * 2 + 8;
  const size_t lds_q = 2 * Tp * rs;                              // K and V row images
  const size_t lds_kv = 2 * Tp * rs + 2 * Tp * sizeof(float);   // Q and dO row images + lse, delta
  hipStream_t st = (hipStream_t)stream;
  if (head_dim == 64) {
    if (int rc = set_lds(attn_bwd_dq_kernel<64, 8>, lds_q)) return rc;
    if (int rc = set_lds(attn_bwd_dkv_kernel<64, 8>, lds_kv)) return rc;
    attn_bwd_dq_kernel<64, 8><<<B * heads, 512, lds_q, st>>>(p);
    attn_bwd_dkv_kernel<64, 8><<<B * heads, 512, lds_kv, st>>>(p);
  } else if (T <= 256) {
    if (int rc = set_lds(attn_bwd_dq_kernel<32, 8>, lds_q)) return rc;
    if (int rc = set_lds(attn_bwd_dkv_kernel<32, 8>, lds_kv)) return rc;
    attn_bwd_dq_kernel<32, 8><<<B * heads, 512, lds_q, st>>>(p);
    attn_bwd_dkv_kernel<32, 8><<<B * heads, 512, lds_kv, st>>>(p);
  } else {
    if (int rc = set_lds(attn_bwd_dq_kernel<32, 9>, lds_q)) return rc;
    if (int rc = set_lds(attn_bwd_dkv_kernel<32, 9>, lds_kv)) return rc;
    attn_bwd_dq_kernel<32, 9><<<B * heads, 576, lds_q, st>>>(p);
    attn_bwd_dkv_kernel<32, 9><<<B * heads, 576, lds_kv, st>>>(p);
  }
  MPR_LAUNCH_CHECK("attn_bwd kernels");
  return MPR_OK;
}

// bf16 <-> fp32 casts (the boundaries of the mixed-precision path)
__global__ __launch_bounds__(256) void tf_cast_to_bf16_kernel(const float4* __restrict__ x, uint2* __restrict__ y, long long n4) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = x[i];
    const float f[4] = {v.x, v.y, v.z, v.w};
    y[i] = pack4(f);
  }
}
__global__ __launch_bounds__(256) void tf_cast_to_f32_kernel(const uint2* __restrict__ x, float4* __restrict__ y, long long n4) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float f[4];
    unpack4(x[i], f);
    y[i] = make_float4(f[0], f[1], f[2], f[3]);
  }
}

int mpr_tf_cast(const void* x, void* y, long long n, int to_bf16, void* stream) {
  MPR_REQUIRE(x && y && n > 0 && n % 4 == 0, "mpr_tf_cast: needs n %% 4 == 0 (n=%lld)", n);
  if (to_bf16) tf_cast_to_bf16_kernel<<<tb_grid(n / 4), 256, 0, (hipStream_t)stream>>>((const float4*)x, (uint2*)y, n / 4);
  else tf_cast_to_f32_kernel<<<tb_grid(n / 4), 256, 0, (hipStream_t)stream>>>((const uint2*)x, (float4*)y, n / 4);
  MPR_LAUNCH_CHECK("tf_cast kernel");
  return MPR_OK;
}

}  // extern "C"
