// fp32 building blocks of the transformer encoders: ProfileTransformer (src/profile_encoder.py:9-68 -- torch's
// post-norm nn.TransformerEncoderLayer, LN eps 1e-5, key-padding mask) and timm's pre-norm ViT (LN eps 1e-6).
// All GEMMs (QKV / out-proj / MLP / QK^T / PV and their gradients) run on mpr_gemm_f32 (two-level batch for
// the per-head products); this file holds the row-wise and elementwise pieces.  One wave per row, wave
// shuffles for the reductions, no LDS.
#include "common.h"

// ---------------------------------------------------------------------------------------------- LayerNorm
// y = LN(x [+ r]) * gamma + beta;  optionally writes s = x + r (the residual stream); saves mean / rstd per row
__global__ __launch_bounds__(256) void add_layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float eps,
                                                                float* __restrict__ y, float* __restrict__ sum_out,
                                                                float* __restrict__ mean, float* __restrict__ rstd,
                                                                int rows, int D) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const size_t o = (size_t)row * D;
  float s = 0.f;
  for (int j = lane; j < D; j += 64) s += x[o + j] + (r ? r[o + j] : 0.f);
  const float mu = wave_sum(s) / D;
  float v = 0.f;
  for (int j = lane; j < D; j += 64) {
    const float d = x[o + j] + (r ? r[o + j] : 0.f) - mu;
    v = fmaf(d, d, v);
  }
  const float rs = 1.f / sqrtf(wave_sum(v) / D + eps);
  for (int j = lane; j < D; j += 64) {
    const float t = x[o + j] + (r ? r[o + j] : 0.f);
    if (sum_out) sum_out[o + j] = t;
    y[o + j] = (t - mu) * rs * gamma[j] + beta[j];
  }
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// ds = d(x + r) = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy * gamma  [+ dskip];  per-block partials of
// dgamma / dbeta go to part[grid][2][D] (summed by ln_param_grad_kernel)
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ s,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ dskip, float* __restrict__ ds,
                                                            int rows, int D) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const size_t o = (size_t)row * D;
  const float mu = mean[row], rs = rstd[row];
  float a = 0.f, b = 0.f;
  for (int j = lane; j < D; j += 64) {
    const float g = dy[o + j] * gamma[j], xh = (s[o + j] - mu) * rs;
    a += g;
    b = fmaf(g, xh, b);
  }
  a = wave_sum(a) / D;
  b = wave_sum(b) / D;
  for (int j = lane; j < D; j += 64) {
    const float g = dy[o + j] * gamma[j], xh = (s[o + j] - mu) * rs;
    ds[o + j] = rs * (g - a - xh * b) + (dskip ? dskip[o + j] : 0.f);
  }
}

// dgamma[j] = sum_rows dy*xhat, dbeta[j] = sum_rows dy    (one thread per column, coalesced along j)
__global__ __launch_bounds__(256) void ln_param_grad_kernel(const float* __restrict__ dy, const float* __restrict__ s,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            float* __restrict__ part, int rows, int D, int rows_per_block) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= D) return;
  const int r0 = blockIdx.y * rows_per_block;
  int r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  float a = 0.f, b = 0.f;
  for (int r = r0; r < r1; ++r) {
    const float g = dy[(size_t)r * D + j];
    a = fmaf(g, (s[(size_t)r * D + j] - mean[r]) * rstd[r], a);
    b += g;
  }
  part[((size_t)blockIdx.y * 2) * D + j] = a;
  part[((size_t)blockIdx.y * 2 + 1) * D + j] = b;
}

// out[c] = sum_p part[p][c]   (c < n)
__global__ __launch_bounds__(256) void column_sum_kernel(const float* __restrict__ part, float* __restrict__ out, int nparts,
                                                         int n, int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= n) return;
  double a = 0.0;
  for (int p = 0; p < nparts; ++p) a += (double)part[(size_t)p * n + c];
  out[c] = accumulate ? out[c] + (float)a : (float)a;
}

// ---------------------------------------------------------------------------------------------- softmax
// P[z][i][:] = softmax(scale * S[z][i][:] masked by key_mask[b][:]),  z = b*H + h; in place
__global__ __launch_bounds__(256) void masked_softmax_fwd_kernel(float* __restrict__ S, const unsigned char* __restrict__ mask,
                                                                 float scale, int rows, int T, int rows_per_batch) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float* s = S + (size_t)row * T;
  const unsigned char* m = mask ? mask + (size_t)(row / rows_per_batch) * T : nullptr;
  float mx = -INFINITY;
  for (int j = lane; j < T; j += 64)
    if (!m || !m[j]) mx = fmaxf(mx, s[j] * scale);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < T; j += 64) {
    const float e = (!m || !m[j]) ? expf(s[j] * scale - mx) : 0.f;
    s[j] = e;
    sum += e;
  }
  const float inv = 1.f / wave_sum(sum);
  for (int j = lane; j < T; j += 64) s[j] *= inv;
}

// dS = scale * P * (dP - sum_j dP*P); in place over dP
__global__ __launch_bounds__(256) void softmax_bwd_kernel(float* __restrict__ dP, const float* __restrict__ P, float scale,
                                                          int rows, int T) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float* d = dP + (size_t)row * T;
  const float* p = P + (size_t)row * T;
  float dot = 0.f;
  for (int j = lane; j < T; j += 64) dot = fmaf(d[j], p[j], dot);
  dot = wave_sum(dot);
  for (int j = lane; j < T; j += 64) d[j] = scale * p[j] * (d[j] - dot);
}

// ---------------------------------------------------------------------------------------------- elementwise
__device__ __forceinline__ uint32_t tf_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

// y = act(x + bias[col]) with optional inverted dropout; act: 0 none, 1 exact-erf GELU, 2 ReLU
__global__ __launch_bounds__(256) void bias_act_fwd_kernel(const float* __restrict__ x, const float* __restrict__ bias,
                                                           int act, float p_drop, uint32_t seed, float* __restrict__ y,
                                                           unsigned char* __restrict__ mask, long long n, int D) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float v = x[i] + (bias ? bias[i % D] : 0.f);
    if (act == 1) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    else if (act == 2) v = fmaxf(v, 0.f);
    else if (act == 3) v = v / (1.f + expf(-v));          // SiLU
    else if (act == 4) v = 1.f / (1.f + expf(-v));        // sigmoid
    if (p_drop > 0.f) {
      const uint32_t h = tf_mix32(tf_mix32((uint32_t)i ^ seed) + 0x9e3779b9U * (seed | 1u) + (uint32_t)(i >> 32));
      const bool keep = (float)(h >> 8) * (1.f / 16777216.f) >= p_drop;
      v = keep ? v / (1.f - p_drop) : 0.f;
      mask[i] = keep;
    }
    y[i] = v;
  }
}

// dx = dy * dropmask/(1-p) * act'(x + bias)
__global__ __launch_bounds__(256) void bias_act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ bias, int act, float p_drop,
                                                           const unsigned char* __restrict__ mask, float* __restrict__ dx,
                                                           long long n, int D) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float g = dy[i];
    if (p_drop > 0.f) g = mask[i] ? g / (1.f - p_drop) : 0.f;
    if (act != 0) {
      const float v = x[i] + (bias ? bias[i % D] : 0.f);
      if (act == 1) {
        const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752f));
        const float pdf = 0.3989422804014327f * expf(-0.5f * v * v);
        g *= cdf + v * pdf;
      } else if (act == 2) {
        g = v > 0.f ? g : 0.f;
      } else {
        const float sg = 1.f / (1.f + expf(-v));
        g *= act == 3 ? sg * (1.f + v * (1.f - sg)) : sg * (1.f - sg);
      }
    }
    dx[i] = g;
  }
}

// y[row][:] = x[row][:] + table[index[row]][:]
__global__ __launch_bounds__(256) void embedding_add_fwd_kernel(const float* __restrict__ x, const float* __restrict__ table,
                                                                const long long* __restrict__ index, float* __restrict__ y,
                                                                int rows, int D) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* t = table + (size_t)index[row] * D;
  for (int j = lane; j < D; j += 64) y[(size_t)row * D + j] = x[(size_t)row * D + j] + t[j];
}

// dtable[index[row]][:] += dy[row][:]  (dtable zeroed by the host wrapper; the padding row is skipped)
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const float* __restrict__ dy, const long long* __restrict__ index,
                                                            float* __restrict__ dtable, int rows, int D, long long padding_idx) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const long long id = index[row];
  if (id == padding_idx) return;
  for (int j = lane; j < D; j += 64) atomicAdd(dtable + (size_t)id * D + j, dy[(size_t)row * D + j]);
}

// y = a + b
__global__ __launch_bounds__(256) void add_f32_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      float* __restrict__ y, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = a[i] + b[i];
}

static inline unsigned tf_grid(long long n) {
  long long g = (n + 255) / 256;
  return (unsigned)(g < 4096 ? (g < 1 ? 1 : g) : 4096);
}

extern "C" {

int mpr_add_layernorm_fwd(const float* x, const float* residual, const float* gamma, const float* beta, float eps, float* y,
                          float* sum_out, float* mean, float* rstd, int rows, int D, void* stream) {
  MPR_REQUIRE(x && gamma && beta && y && mean && rstd && rows > 0 && D > 0, "mpr_add_layernorm_fwd: bad arguments");
  add_layernorm_fwd_kernel<<<ceil_div(rows, 4), 256, 0, (hipStream_t)stream>>>(x, residual, gamma, beta, eps, y, sum_out,
                                                                               mean, rstd, rows, D);
  MPR_LAUNCH_CHECK("add_layernorm_fwd_kernel");
  return MPR_OK;
}

// ds = dLN/d(x+r) (+ dskip); dgamma/dbeta (accumulate flag) ; workspace: 2*D*ceil(rows/256) floats
int mpr_layernorm_bwd(const float* dy, const float* s, const float* gamma, const float* mean, const float* rstd,
                      const float* dskip, float* ds, float* dgamma, float* dbeta, float* workspace, int accumulate,
                      int rows, int D, void* stream) {
  MPR_REQUIRE(dy && s && gamma && mean && rstd && ds && dgamma && dbeta && workspace, "mpr_layernorm_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  layernorm_bwd_kernel<<<ceil_div(rows, 4), 256, 0, st>>>(dy, s, gamma, mean, rstd, dskip, ds, rows, D);
  MPR_LAUNCH_CHECK("layernorm_bwd_kernel");
  const int rpb = 256, nparts = ceil_div(rows, rpb);
  ln_param_grad_kernel<<<dim3(ceil_div(D, 256), nparts), 256, 0, st>>>(dy, s, mean, rstd, workspace, rows, D, rpb);
  MPR_LAUNCH_CHECK("ln_param_grad_kernel");
  // workspace rows alternate (dgamma partial, dbeta partial): view as [nparts][2*D]
  column_sum_kernel<<<ceil_div(2 * D, 256), 256, 0, st>>>(workspace, workspace + (size_t)nparts * 2 * D, nparts, 2 * D, 0);
  MPR_LAUNCH_CHECK("column_sum_kernel");
  // split the [2*D] result
  const float* tot = workspace + (size_t)nparts * 2 * D;
  if (accumulate) {
    add_f32_kernel<<<tf_grid(D), 256, 0, st>>>(dgamma, tot, dgamma, D);
    add_f32_kernel<<<tf_grid(D), 256, 0, st>>>(dbeta, tot + D, dbeta, D);
  } else {
    MPR_HIP(hipMemcpyAsync(dgamma, tot, sizeof(float) * D, hipMemcpyDeviceToDevice, st));
    MPR_HIP(hipMemcpyAsync(dbeta, tot + D, sizeof(float) * D, hipMemcpyDeviceToDevice, st));
  }
  MPR_LAUNCH_CHECK("layernorm_bwd tail");
  return MPR_OK;
}

int mpr_layernorm_bwd_workspace_floats(int rows, int D) { return 2 * D * (ceil_div(rows, 256) + 1); }

int mpr_masked_softmax_fwd(float* S, const void* key_padding_mask, float scale, int batch, int heads, int Tq, int T,
                           void* stream) {
  MPR_REQUIRE(S && batch > 0 && heads > 0 && Tq > 0 && T > 0, "mpr_masked_softmax_fwd: bad arguments");
  const int rows = batch * heads * Tq;
  masked_softmax_fwd_kernel<<<ceil_div(rows, 4), 256, 0, (hipStream_t)stream>>>(S, (const unsigned char*)key_padding_mask,
                                                                                scale, rows, T, heads * Tq);
  MPR_LAUNCH_CHECK("masked_softmax_fwd_kernel");
  return MPR_OK;
}

int mpr_softmax_bwd(float* dP, const float* P, float scale, int rows, int T, void* stream) {
  MPR_REQUIRE(dP && P, "mpr_softmax_bwd: null pointer");
  softmax_bwd_kernel<<<ceil_div(rows, 4), 256, 0, (hipStream_t)stream>>>(dP, P, scale, rows, T);
  MPR_LAUNCH_CHECK("softmax_bwd_kernel");
  return MPR_OK;
}

int mpr_bias_act_fwd(const float* x, const float* bias, int act, float p_drop, unsigned seed, float* y, void* mask,
                     long long n, int D, void* stream) {
  MPR_REQUIRE(x && y && (p_drop == 0.f || mask) && act >= 0 && act <= 4, "mpr_bias_act_fwd: bad arguments");
  bias_act_fwd_kernel<<<tf_grid(n), 256, 0, (hipStream_t)stream>>>(x, bias, act, p_drop, seed, y, (unsigned char*)mask, n, D);
  MPR_LAUNCH_CHECK("bias_act_fwd_kernel");
  return MPR_OK;
}

int mpr_bias_act_bwd(const float* dy, const float* x, const float* bias, int act, float p_drop, const void* mask, float* dx,
                     long long n, int D, void* stream) {
  MPR_REQUIRE(dy && dx && (act == 0 || x), "mpr_bias_act_bwd: bad arguments");
  bias_act_bwd_kernel<<<tf_grid(n), 256, 0, (hipStream_t)stream>>>(dy, x, bias, act, p_drop, (const unsigned char*)mask, dx, n,
                                                                   D);
  MPR_LAUNCH_CHECK("bias_act_bwd_kernel");
  return MPR_OK;
}

int mpr_embedding_add_fwd(const float* x, const float* table, const long long* index, float* y, int rows, int D,
                          void* stream) {
  MPR_REQUIRE(x && table && index && y, "mpr_embedding_add_fwd: null pointer");
  embedding_add_fwd_kernel<<<ceil_div(rows, 4), 256, 0, (hipStream_t)stream>>>(x, table, index, y, rows, D);
  MPR_LAUNCH_CHECK("embedding_add_fwd_kernel");
  return MPR_OK;
}

int mpr_embedding_bwd(const float* dy, const long long* index, float* dtable, int table_rows, int rows, int D,
                      long long padding_idx, void* stream) {
  MPR_REQUIRE(dy && index && dtable, "mpr_embedding_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  MPR_HIP(hipMemsetAsync(dtable, 0, sizeof(float) * (size_t)table_rows * D, st));
  embedding_bwd_kernel<<<ceil_div(rows, 4), 256, 0, st>>>(dy, index, dtable, rows, D, padding_idx);
  MPR_LAUNCH_CHECK("embedding_bwd_kernel");
  return MPR_OK;
}

int mpr_add_f32(const float* a, const float* b, float* y, long long n, void* stream) {
  MPR_REQUIRE(a && b && y, "mpr_add_f32: null pointer");
  add_f32_kernel<<<tf_grid(n), 256, 0, (hipStream_t)stream>>>(a, b, y, n);
  MPR_LAUNCH_CHECK("add_f32_kernel");
  return MPR_OK;
}

}  // extern "C"
