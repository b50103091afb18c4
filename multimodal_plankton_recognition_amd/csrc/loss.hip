// Cross-modal coordination losses on the projected embeddings, fp32 end to end
// (reference: src/coordination.py -- CLIPLoss :17-47, SigLIPLoss :67-95, +beta*MSE variants :50-64,98-112).
// Pipeline (host side orchestrates; the B x B products run on the exact-fp32 MFMA GEMM):
//   l2norm_fwd  : u = x / max(|x|, 1e-12)                                  (F.normalize, :33-34)
//   gemm        : S[b] = U[b] V[b]^T per bucket (raw cosines)               (:38, buckets :29-37)
//   clip        : row / column log-sum-exp of S*exp(logit_scale) -> loss    (:40-45)
//                 then G = dLoss/dlogits in place, d(logit_scale)
//   siglip      : one pass for the loss, one pass for G, d(logit_scale), d(bias)   (:89-95)
//   gemm        : dU = G V, dV = G^T U
//   l2norm_bwd  : dx = (du - u (u.du)) / |x|  [+ beta * dMSE/dx]
// S (n x n fp32 per bucket) is materialised: 1 MB at n = 512, 64 MB at the DP=8 global batch of 4096 --
// it stays resident in the 256 MiB Infinity Cache between the passes.
#include "common.h"

// ------------------------------------------------------------------------------------------------ normalise
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ u,
                                                         float* __restrict__ inv, int rows, int D) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (size_t)row * D;
  float ss = 0.f;
  for (int j = lane; j < D; j += 64) ss = fmaf(xr[j], xr[j], ss);
  ss = wave_sum(ss);
  const float r = 1.f / fmaxf(sqrtf(ss), 1e-12f);
  for (int j = lane; j < D; j += 64) u[(size_t)row * D + j] = xr[j] * r;
  if (lane == 0) inv[row] = r;
}

// dx = inv * (du - u * (u . du)) * gout   [+ mse_coef * gout * (x - other)]
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ du, const float* __restrict__ u,
                                                         const float* __restrict__ inv, const float* __restrict__ x,
                                                         const float* __restrict__ other, float mse_coef,
                                                         const float* __restrict__ gout, float* __restrict__ dx,
                                                         int rows, int D) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const size_t o = (size_t)row * D;
  float dot = 0.f;
  for (int j = lane; j < D; j += 64) dot = fmaf(u[o + j], du[o + j], dot);
  dot = wave_sum(dot);
  const float r = inv[row], g = gout ? gout[0] : 1.f;
  for (int j = lane; j < D; j += 64) {
    float v = r * (du[o + j] - u[o + j] * dot);
    if (x) v += mse_coef * (x[o + j] - other[o + j]);
    dx[o + j] = v * g;
  }
}

// ------------------------------------------------------------------------------------------------ CLIP
// one wave per row: lse over the row of S*scale; also the diagonal logit
__global__ __launch_bounds__(256) void clip_row_lse_kernel(const float* __restrict__ S, const float* __restrict__ ls,
                                                           float* __restrict__ row_lse, float* __restrict__ diag,
                                                           int n, int rows) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float scale = expf(ls[0]);
  const float* s = S + (size_t)row * n;
  float m = -INFINITY, acc = 0.f;
  for (int j = lane; j < n; j += 64) {
    const float l = s[j] * scale;
    if (l > m) { acc = acc * expf(m - l) + 1.f; m = l; } else acc += expf(l - m);
  }
  const float gm = wave_max(m);
  acc = wave_sum(acc * expf(m - gm));
  if (lane == 0) {
    row_lse[row] = gm + logf(acc);
    diag[row] = s[row % n] * scale;
  }
}

// block = 64 columns x 4 row-lanes
__global__ __launch_bounds__(256) void clip_col_lse_kernel(const float* __restrict__ S, const float* __restrict__ ls,
                                                           float* __restrict__ col_lse, int n) {
  __shared__ float sm[4][64], sa[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  const float scale = expf(ls[0]);
  const float* s = S + (size_t)blockIdx.y * n * n;
  float m = -INFINITY, acc = 0.f;
  if (col < n)
    for (int i = part; i < n; i += 4) {
      const float l = s[(size_t)i * n + col] * scale;
      if (l > m) { acc = acc * expf(m - l) + 1.f; m = l; } else acc += expf(l - m);
    }
  sm[part][threadIdx.x & 63] = m;
  sa[part][threadIdx.x & 63] = acc;
  __syncthreads();
  if (part == 0 && col < n) {
    float gm = m;
    for (int i = 1; i < 4; ++i) gm = fmaxf(gm, sm[i][threadIdx.x]);
    float t = 0.f;
    for (int i = 0; i < 4; ++i) t += sa[i][threadIdx.x] * expf(sm[i][threadIdx.x] - gm);
    col_lse[(size_t)blockIdx.y * n + col] = gm + logf(t);
  }
}

// loss = sum_rows (row_lse + col_lse - 2 diag) / (2 * rows)      (rows = buckets * n)
__global__ __launch_bounds__(1024) void clip_loss_kernel(const float* __restrict__ row_lse,
                                                         const float* __restrict__ col_lse,
                                                         const float* __restrict__ diag, float* __restrict__ loss,
                                                         int rows) {
  __shared__ double red[1024];
  double a = 0.0;
  for (int i = threadIdx.x; i < rows; i += 1024) a += (double)row_lse[i] + (double)col_lse[i] - 2.0 * (double)diag[i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = (float)(red[0] / (2.0 * rows));
}

// S <- dLoss/dS (raw cosine) in place; dls partial sums -> part[gridDim.x]
__global__ __launch_bounds__(256) void clip_grad_kernel(float* __restrict__ S, const float* __restrict__ ls,
                                                        const float* __restrict__ row_lse,
                                                        const float* __restrict__ col_lse, float* __restrict__ part,
                                                        int n, long long total, float coef) {
  __shared__ float red[4];
  const float scale = expf(ls[0]);
  float dls = 0.f;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const unsigned u = (unsigned)idx;
    const unsigned row = u / (unsigned)n, j = u - row * n;       // row = bucket*n + i
    const unsigned i = row % (unsigned)n, bcol = row - i + j;     // column's lse index = bucket*n + j
    const float l = S[idx] * scale;
    float g = expf(l - row_lse[row]) + expf(l - col_lse[bcol]);
    if (i == j) g -= 2.f;
    g *= coef;
    dls = fmaf(g, l, dls);
    S[idx] = g * scale;
  }
  dls = wave_sum(dls);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dls;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// ------------------------------------------------------------------------------------------------ CLIP, row block
// Data-parallel form: a rank owns `rows` samples of one modality and holds S_blk = X_loc * Y_all^T
// [rows][ncols]; its positives sit at column diag_off + i.  Row LSEs are local; the softmax over the
// other axis needs the other ranks' LSE vector (all-gathered by the caller).
__global__ __launch_bounds__(256) void clip_block_row_lse_kernel(const float* __restrict__ S,
                                                                 const float* __restrict__ ls,
                                                                 float* __restrict__ row_lse, float* __restrict__ diag,
                                                                 int rows, int ncols, int diag_off) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float scale = expf(ls[0]);
  const float* s = S + (size_t)row * ncols;
  float m = -INFINITY, acc = 0.f;
  for (int j = lane; j < ncols; j += 64) {
    const float l = s[j] * scale;
    if (l > m) { acc = acc * expf(m - l) + 1.f; m = l; } else acc += expf(l - m);
  }
  const float gm = wave_max(m);
  acc = wave_sum(acc * expf(m - gm));
  if (lane == 0) {
    row_lse[row] = gm + logf(acc);
    diag[row] = s[diag_off + row] * scale;
  }
}

// S_blk <- coef * (exp(l - lse_own[i]) + exp(l - lse_other[j]) - 2 [j == diag_off + i]) * scale, in place;
// part[grid] = partial sums of G * l (d logit_scale)
__global__ __launch_bounds__(256) void clip_block_grad_kernel(float* __restrict__ S, const float* __restrict__ ls,
                                                              const float* __restrict__ lse_own,
                                                              const float* __restrict__ lse_other,
                                                              float* __restrict__ part, int rows, int ncols,
                                                              int diag_off, float coef) {
  __shared__ float red[4];
  const float scale = expf(ls[0]);
  const long long total = (long long)rows * ncols;
  float dls = 0.f;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const unsigned u = (unsigned)idx;
    const unsigned i = u / (unsigned)ncols, j = u - i * ncols;
    const float l = S[idx] * scale;
    float g = expf(l - lse_own[i]) + expf(l - lse_other[j]);
    if ((int)j == diag_off + (int)i) g -= 2.f;
    g *= coef;
    dls = fmaf(g, l, dls);
    S[idx] = g * scale;
  }
  dls = wave_sum(dls);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dls;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// ------------------------------------------------------------------------------------------------ SigLIP
__device__ __forceinline__ float log_sigmoid(float x) { return fminf(x, 0.f) - log1pf(expf(-fabsf(x))); }

// part[grid] = sum of -logsigmoid(sign * z);  z = S*scale + bias
__global__ __launch_bounds__(256) void siglip_fwd_kernel(const float* __restrict__ S, const float* __restrict__ ls,
                                                         const float* __restrict__ bias, float* __restrict__ part,
                                                         int n, long long total) {
  __shared__ float red[4];
  const float scale = expf(ls[0]), b = bias[0];
  float a = 0.f;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const unsigned u = (unsigned)idx;
    const unsigned row = u / (unsigned)n, j = u - row * n, i = row % (unsigned)n;
    const float z = fmaf(S[idx], scale, b);
    a -= log_sigmoid(i == j ? z : -z);
  }
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// S <- dLoss/dS in place; part[0][grid] = d(logit_scale) partials, part[1][grid] = d(bias) partials
__global__ __launch_bounds__(256) void siglip_grad_kernel(float* __restrict__ S, const float* __restrict__ ls,
                                                          const float* __restrict__ bias, float* __restrict__ part,
                                                          int n, long long total, float coef) {
  __shared__ float red[2][4];
  const float scale = expf(ls[0]), b = bias[0];
  float dls = 0.f, db = 0.f;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const unsigned u = (unsigned)idx;
    const unsigned row = u / (unsigned)n, j = u - row * n, i = row % (unsigned)n;
    const float l = S[idx] * scale, z = l + b;
    const float sg = i == j ? 1.f : -1.f;
    // d/dz [-logsigmoid(sg*z)] = -sg * sigmoid(-sg*z)
    const float g = -sg * coef / (1.f + expf(sg * z));
    dls = fmaf(g, l, dls);
    db += g;
    S[idx] = g * scale;
  }
  dls = wave_sum(dls);
  db = wave_sum(db);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = dls; red[1][threadIdx.x >> 6] = db; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    part[gridDim.x + blockIdx.x] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

// ------------------------------------------------------------------------------------------------ helpers
// part[grid] = sum (a-b)^2
__global__ __launch_bounds__(256) void sqdiff_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             float* __restrict__ part, long long total) {
  __shared__ float red[4];
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const float d = a[i] - b[i];
    s = fmaf(d, d, s);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// out[0] = (accumulate ? out[0] : 0) + mul * gmul * sum(part[0..n))
__global__ __launch_bounds__(256) void finish_sum_kernel(const float* __restrict__ part, int n, float mul,
                                                         const float* __restrict__ gmul, float* __restrict__ out,
                                                         int accumulate) {
  __shared__ double red[256];
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) a += (double)part[i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float v = (float)(red[0] * (double)mul) * (gmul ? gmul[0] : 1.f);
    out[0] = accumulate ? out[0] + v : v;
  }
}

#define LOSS_GRID 512

extern "C" {

int mpr_loss_workspace_floats(void) { return 2 * LOSS_GRID; }

int mpr_l2norm_fwd(const float* x, float* u, float* inv, int rows, int D, void* stream) {
  MPR_REQUIRE(x && u && inv && rows > 0 && D > 0, "mpr_l2norm_fwd: bad arguments");
  l2norm_fwd_kernel<<<ceil_div(rows, 4), 256, 0, (hipStream_t)stream>>>(x, u, inv, rows, D);
  MPR_LAUNCH_CHECK("l2norm_fwd_kernel");
  return MPR_OK;
}

// dx = gout * ( inv*(du - u (u.du)) + mse_coef*(x - other) )   (x/other may be NULL)
int mpr_l2norm_bwd(const float* du, const float* u, const float* inv, const float* x, const float* other,
                   float mse_coef, const float* gout, float* dx, int rows, int D, void* stream) {
  MPR_REQUIRE(du && u && inv && dx, "mpr_l2norm_bwd: null pointer");
  MPR_REQUIRE((x == nullptr) == (other == nullptr), "mpr_l2norm_bwd: x and other go together");
  l2norm_bwd_kernel<<<ceil_div(rows, 4), 256, 0, (hipStream_t)stream>>>(du, u, inv, x, other, mse_coef, gout, dx, rows, D);
  MPR_LAUNCH_CHECK("l2norm_bwd_kernel");
  return MPR_OK;
}

// S: [buckets][n][n] raw cosines.  Outputs: row_lse, col_lse, diag: [buckets*n]; loss[1].
int mpr_clip_fwd(const float* S, const float* logit_scale, float* row_lse, float* col_lse, float* diag, float* loss,
                 int buckets, int n, void* stream) {
  MPR_REQUIRE(S && logit_scale && row_lse && col_lse && diag && loss, "mpr_clip_fwd: null pointer");
  MPR_REQUIRE((long long)buckets * n * n < (1ll << 32), "mpr_clip_fwd: similarity matrix too large");
  hipStream_t st = (hipStream_t)stream;
  const int rows = buckets * n;
  clip_row_lse_kernel<<<ceil_div(rows, 4), 256, 0, st>>>(S, logit_scale, row_lse, diag, n, rows);
  MPR_LAUNCH_CHECK("clip_row_lse_kernel");
  clip_col_lse_kernel<<<dim3(ceil_div(n, 64), buckets), 256, 0, st>>>(S, logit_scale, col_lse, n);
  MPR_LAUNCH_CHECK("clip_col_lse_kernel");
  clip_loss_kernel<<<1, 1024, 0, st>>>(row_lse, col_lse, diag, loss, rows);
  MPR_LAUNCH_CHECK("clip_loss_kernel");
  return MPR_OK;
}

// S <- dLoss/dS_raw (in place, NOT yet multiplied by the upstream gradient); d_logit_scale[1] (x gout).
int mpr_clip_bwd(float* S, const float* logit_scale, const float* row_lse, const float* col_lse, const float* gout,
                 float* d_logit_scale, float* workspace, int buckets, int n, void* stream) {
  MPR_REQUIRE(S && logit_scale && row_lse && col_lse && d_logit_scale && workspace, "mpr_clip_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const long long total = (long long)buckets * n * n;
  const int grid = (int)((total + 255) / 256 < LOSS_GRID ? (total + 255) / 256 : LOSS_GRID);
  clip_grad_kernel<<<grid, 256, 0, st>>>(S, logit_scale, row_lse, col_lse, workspace, n, total,
                                         1.f / (2.f * (float)n * (float)buckets));
  MPR_LAUNCH_CHECK("clip_grad_kernel");
  finish_sum_kernel<<<1, 256, 0, st>>>(workspace, grid, 1.f, gout, d_logit_scale, 0);
  MPR_LAUNCH_CHECK("finish_sum_kernel");
  return MPR_OK;
}

// Row-block (data-parallel) CLIP: S_blk [rows][ncols] = X_loc Y_all^T raw cosines, positives at column
// diag_off + i.  sum_out[0] = sum_i (row_lse[i] - diag[i])  (un-normalised local part of the loss).
int mpr_clip_block_fwd(const float* S, const float* logit_scale, float* row_lse, float* diag, float* sum_out,
                       float* workspace, int rows, int ncols, int diag_off, void* stream) {
  MPR_REQUIRE(S && logit_scale && row_lse && diag && sum_out && workspace, "mpr_clip_block_fwd: null pointer");
  MPR_REQUIRE(diag_off >= 0 && diag_off + rows <= ncols, "mpr_clip_block_fwd: diagonal outside the block");
  MPR_REQUIRE((long long)rows * ncols < (1ll << 32), "mpr_clip_block_fwd: block too large");
  hipStream_t st = (hipStream_t)stream;
  clip_block_row_lse_kernel<<<ceil_div(rows, 4), 256, 0, st>>>(S, logit_scale, row_lse, diag, rows, ncols, diag_off);
  MPR_LAUNCH_CHECK("clip_block_row_lse_kernel");
  // sum (row_lse - diag): reuse the squared-difference-free helper via two finishing sums
  finish_sum_kernel<<<1, 256, 0, st>>>(row_lse, rows, 1.f, nullptr, sum_out, 0);
  finish_sum_kernel<<<1, 256, 0, st>>>(diag, rows, -1.f, nullptr, sum_out, 1);
  MPR_LAUNCH_CHECK("finish_sum_kernel");
  return MPR_OK;
}

// S_blk <- dLoss/dS_raw (in place); d_logit_scale_part[0] = sum G*l over this block (x gout if given).
int mpr_clip_block_bwd(float* S, const float* logit_scale, const float* lse_own, const float* lse_other,
                       const float* gout, float coef, float* d_logit_scale_part, float* workspace, int rows,
                       int ncols, int diag_off, void* stream) {
  MPR_REQUIRE(S && logit_scale && lse_own && lse_other && d_logit_scale_part && workspace,
              "mpr_clip_block_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const long long total = (long long)rows * ncols;
  const int grid = (int)((total + 255) / 256 < LOSS_GRID ? (total + 255) / 256 : LOSS_GRID);
  clip_block_grad_kernel<<<grid, 256, 0, st>>>(S, logit_scale, lse_own, lse_other, workspace, rows, ncols, diag_off,
                                               coef);
  MPR_LAUNCH_CHECK("clip_block_grad_kernel");
  finish_sum_kernel<<<1, 256, 0, st>>>(workspace, grid, 1.f, gout, d_logit_scale_part, 0);
  MPR_LAUNCH_CHECK("finish_sum_kernel");
  return MPR_OK;
}

int mpr_siglip_fwd(const float* S, const float* logit_scale, const float* bias, float* loss, float* workspace,
                   int buckets, int n, void* stream) {
  MPR_REQUIRE(S && logit_scale && bias && loss && workspace, "mpr_siglip_fwd: null pointer");
  MPR_REQUIRE((long long)buckets * n * n < (1ll << 32), "mpr_siglip_fwd: similarity matrix too large");
  hipStream_t st = (hipStream_t)stream;
  const long long total = (long long)buckets * n * n;
  const int grid = (int)((total + 255) / 256 < LOSS_GRID ? (total + 255) / 256 : LOSS_GRID);
  siglip_fwd_kernel<<<grid, 256, 0, st>>>(S, logit_scale, bias, workspace, n, total);
  MPR_LAUNCH_CHECK("siglip_fwd_kernel");
  finish_sum_kernel<<<1, 256, 0, st>>>(workspace, grid, 1.f / ((float)n * (float)buckets), nullptr, loss, 0);
  MPR_LAUNCH_CHECK("finish_sum_kernel");
  return MPR_OK;
}

int mpr_siglip_bwd(float* S, const float* logit_scale, const float* bias, const float* gout, float* d_logit_scale,
                   float* d_bias, float* workspace, int buckets, int n, void* stream) {
  MPR_REQUIRE(S && logit_scale && bias && d_logit_scale && d_bias && workspace, "mpr_siglip_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const long long total = (long long)buckets * n * n;
  const int grid = (int)((total + 255) / 256 < LOSS_GRID ? (total + 255) / 256 : LOSS_GRID);
  siglip_grad_kernel<<<grid, 256, 0, st>>>(S, logit_scale, bias, workspace, n, total, 1.f / ((float)n * (float)buckets));
  MPR_LAUNCH_CHECK("siglip_grad_kernel");
  finish_sum_kernel<<<1, 256, 0, st>>>(workspace, grid, 1.f, gout, d_logit_scale, 0);
  finish_sum_kernel<<<1, 256, 0, st>>>(workspace + grid, grid, 1.f, gout, d_bias, 0);
  MPR_LAUNCH_CHECK("finish_sum_kernel");
  return MPR_OK;
}

// loss[0] += beta * mean((a-b)^2)
int mpr_mse_add(const float* a, const float* b, float beta, float* loss, float* workspace, long long total,
                void* stream) {
  MPR_REQUIRE(a && b && loss && workspace && total > 0, "mpr_mse_add: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int grid = (int)((total + 255) / 256 < LOSS_GRID ? (total + 255) / 256 : LOSS_GRID);
  sqdiff_partial_kernel<<<grid, 256, 0, st>>>(a, b, workspace, total);
  MPR_LAUNCH_CHECK("sqdiff_partial_kernel");
  finish_sum_kernel<<<1, 256, 0, st>>>(workspace, grid, beta / (float)total, nullptr, loss, 1);
  MPR_LAUNCH_CHECK("finish_sum_kernel");
  return MPR_OK;
}
}  // extern "C"

// ---------------------------------------------------------------------------------------------- RankLoss
// src/coordination.py:115-135: S = u v^T with the diagonal negated; loss = ( mean relu(margin + column sums)
// + mean relu(margin + row sums) ) / 2.   S: [n][n] raw cosines (left untouched: the sign flip is applied on the fly).
__global__ __launch_bounds__(256) void rank_row_sums_kernel(const float* __restrict__ S, float* __restrict__ row_sum, int n) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n) return;
  const float* s = S + (size_t)row * n;
  float a = 0.f;
  for (int j = lane; j < n; j += 64) a += j == row ? -s[j] : s[j];
  a = wave_sum(a);
  if (lane == 0) row_sum[row] = a;
}

__global__ __launch_bounds__(256) void rank_col_sums_kernel(const float* __restrict__ S, float* __restrict__ col_sum, int n) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= n) return;
  float a = 0.f;
  for (int i = 0; i < n; ++i) {
    const float v = S[(size_t)i * n + col];
    a += i == col ? -v : v;
  }
  col_sum[col] = a;
}

__global__ __launch_bounds__(256) void rank_loss_kernel(const float* __restrict__ row_sum, const float* __restrict__ col_sum,
                                                        float margin, float* __restrict__ loss, int n) {
  __shared__ double red[256];
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) a += (double)fmaxf(margin + row_sum[i], 0.f) + (double)fmaxf(margin + col_sum[i], 0.f);
  red[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = (float)(red[0] / (2.0 * n));
}

// G[i][j] = gout / (2 n) * (1[margin + col_sum[j] > 0] + 1[margin + row_sum[i] > 0]) * (i == j ? -1 : 1)
__global__ __launch_bounds__(256) void rank_grad_kernel(float* __restrict__ G, const float* __restrict__ row_sum,
                                                        const float* __restrict__ col_sum, float margin,
                                                        const float* __restrict__ gout, int n) {
  const long long total = (long long)n * n;
  const float g = (gout ? gout[0] : 1.f) / (2.f * n);
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int i = (int)(idx / n), j = (int)(idx - (long long)i * n);
    const float v = g * ((margin + col_sum[j] > 0.f ? 1.f : 0.f) + (margin + row_sum[i] > 0.f ? 1.f : 0.f));
    G[idx] = i == j ? -v : v;
  }
}

extern "C" {

int mpr_rank_fwd(const float* S, float margin, float* row_sum, float* col_sum, float* loss, int n, void* stream) {
  MPR_REQUIRE(S && row_sum && col_sum && loss && n > 0, "mpr_rank_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  rank_row_sums_kernel<<<(n + 3) / 4, 256, 0, st>>>(S, row_sum, n);
  MPR_LAUNCH_CHECK("rank_row_sums_kernel");
  rank_col_sums_kernel<<<(n + 255) / 256, 256, 0, st>>>(S, col_sum, n);
  MPR_LAUNCH_CHECK("rank_col_sums_kernel");
  rank_loss_kernel<<<1, 256, 0, st>>>(row_sum, col_sum, margin, loss, n);
  MPR_LAUNCH_CHECK("rank_loss_kernel");
  return MPR_OK;
}

int mpr_rank_bwd(float* G, const float* row_sum, const float* col_sum, float margin, const float* gout, int n, void* stream) {
  MPR_REQUIRE(G && row_sum && col_sum && n > 0, "mpr_rank_bwd: bad arguments");
  const long long total = (long long)n * n;
  const int grid = (int)((total + 255) / 256 < LOSS_GRID ? (total + 255) / 256 : LOSS_GRID);
  rank_grad_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(G, row_sum, col_sum, margin, gout, n);
  MPR_LAUNCH_CHECK("rank_grad_kernel");
  return MPR_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------- SigLIP, row block
// Data-parallel SigLIP (extension of src/coordination.py:76-95 to a batch sharded over ranks): S_blk [rows][ncols] holds the
// raw cosines of this rank's rows against ALL columns, the positive of row i sits at column diag_off + i.  Every (i, j)
// pair belongs to exactly one rank's row block, so loss and parameter gradients are plain sums over ranks.
__global__ __launch_bounds__(256) void siglip_block_fwd_kernel(const float* __restrict__ S, const float* __restrict__ ls,
                                                               const float* __restrict__ bias, float* __restrict__ part,
                                                               int ncols, int diag_off, long long total) {
  __shared__ float red[4];
  const float scale = expf(ls[0]), b = bias[0];
  float a = 0.f;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int i = (int)(idx / ncols), j = (int)(idx - (long long)i * ncols);
    const float z = fmaf(S[idx], scale, b);
    a -= log_sigmoid(j == diag_off + i ? z : -z);
  }
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// S <- coef * dLoss/dS_raw in place; part[0][grid] / part[1][grid]: d(logit_scale) / d(bias) partials of this block
__global__ __launch_bounds__(256) void siglip_block_grad_kernel(float* __restrict__ S, const float* __restrict__ ls,
                                                                const float* __restrict__ bias, float* __restrict__ part,
                                                                int ncols, int diag_off, long long total, float coef) {
  __shared__ float red[2][4];
  const float scale = expf(ls[0]), b = bias[0];
  float dls = 0.f, db = 0.f;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int i = (int)(idx / ncols), j = (int)(idx - (long long)i * ncols);
    const float l = S[idx] * scale, z = l + b;
    const float sg = j == diag_off + i ? 1.f : -1.f;
    const float g = -sg * coef / (1.f + expf(sg * z));
    dls = fmaf(g, l, dls);
    db += g;
    S[idx] = g * scale;
  }
  dls = wave_sum(dls);
  db = wave_sum(db);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = dls; red[1][threadIdx.x >> 6] = db; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    part[gridDim.x + blockIdx.x] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

extern "C" {

int mpr_siglip_block_fwd(const float* S, const float* logit_scale, const float* bias, float* sum_out, float* workspace,
                         int rows, int ncols, int diag_off, void* stream) {
  MPR_REQUIRE(S && logit_scale && bias && sum_out && workspace && rows > 0 && ncols > 0, "mpr_siglip_block_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const long long total = (long long)rows * ncols;
  const int grid = (int)((total + 255) / 256 < LOSS_GRID ? (total + 255) / 256 : LOSS_GRID);
  siglip_block_fwd_kernel<<<grid, 256, 0, st>>>(S, logit_scale, bias, workspace, ncols, diag_off, total);
  MPR_LAUNCH_CHECK("siglip_block_fwd_kernel");
  finish_sum_kernel<<<1, 256, 0, st>>>(workspace, grid, 1.f, nullptr, sum_out, 0);
  MPR_LAUNCH_CHECK("finish_sum_kernel");
  return MPR_OK;
}

int mpr_siglip_block_bwd(float* S, const float* logit_scale, const float* bias, float coef, float* d_logit_scale_part,
                         float* d_bias_part, float* workspace, int rows, int ncols, int diag_off, void* stream) {
  MPR_REQUIRE(S && logit_scale && bias && workspace && rows > 0 && ncols > 0, "mpr_siglip_block_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const long long total = (long long)rows * ncols;
  const int grid = (int)((total + 255) / 256 < LOSS_GRID ? (total + 255) / 256 : LOSS_GRID);
  siglip_block_grad_kernel<<<grid, 256, 0, st>>>(S, logit_scale, bias, workspace, ncols, diag_off, total, coef);
  MPR_LAUNCH_CHECK("siglip_block_grad_kernel");
  if (d_logit_scale_part) finish_sum_kernel<<<1, 256, 0, st>>>(workspace, grid, 1.f, nullptr, d_logit_scale_part, 0);
  if (d_bias_part) finish_sum_kernel<<<1, 256, 0, st>>>(workspace + grid, grid, 1.f, nullptr, d_bias_part, 0);
  MPR_LAUNCH_CHECK("finish_sum_kernel");
  return MPR_OK;
}

// out[0] = sum (a - b)^2   (the local share of the MSE term of CLIPPlus / SigLIPPlus, src/coordination.py:60-64,108-112)
int mpr_sqdiff_sum(const float* a, const float* b, float* out, float* workspace, long long total, void* stream) {
  MPR_REQUIRE(a && b && out && workspace && total > 0, "mpr_sqdiff_sum: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int grid = (int)((total + 255) / 256 < LOSS_GRID ? (total + 255) / 256 : LOSS_GRID);
  sqdiff_partial_kernel<<<grid, 256, 0, st>>>(a, b, workspace, total);
  MPR_LAUNCH_CHECK("sqdiff_partial_kernel");
  finish_sum_kernel<<<1, 256, 0, st>>>(workspace, grid, 1.f, nullptr, out, 0);
  MPR_LAUNCH_CHECK("finish_sum_kernel");
  return MPR_OK;
}

}  // extern "C"
