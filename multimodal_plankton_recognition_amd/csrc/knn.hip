// Few-shot evaluation: exact brute-force k-nearest-neighbour search + inverse-distance weighted vote,
// standing in for src/ann.py:6-34 (pynndescent.NNDescent configured "to mimic deterministic NN-search" + sklearn's
// weighted_mode) as driven by scripts/benchmark_cross.py:24-96 / benchmark_raw.py:24-49.
//   1. mpr_gemm_f32 (caller) forms the inner products X G^T of a block of queries against the gallery;
//   2. knn_select: one wave per query turns them into distances (euclidean: ||x||^2 + ||g||^2 - 2 x.g, cosine:
//      1 - x.g / (|x||g|)) and extracts the k smallest in (distance, index) order by k min-scans of the row
//      (it stays in L1/L2: the gallery of a few-shot run is a few thousand points);
//   3. knn_refine recomputes the k reported distances directly, sqrt(sum (x - g)^2) in fp32, so that a query that IS a
//      gallery point reports exactly 0 (the reference's weights switch to an indicator on zero distances);
//   4. knn_vote: one thread per query, weights 1/d (rows holding a zero distance: indicator of the zeros), class with the
//      largest summed weight, ties to the smallest class id (weighted_mode's rule).
#include "common.h"

__global__ __launch_bounds__(256) void knn_sqnorm_kernel(const float* __restrict__ X, float* __restrict__ out, int rows, int D) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float a = 0.f;
  for (int j = lane; j < D; j += 64) {
    const float v = X[(size_t)row * D + j];
    a = fmaf(v, v, a);
  }
  a = wave_sum(a);
  if (lane == 0) out[row] = a;
}

// dots: [nq][ng] inner products (overwritten with the selection distances); idx / dist: [nq][k]
__global__ __launch_bounds__(256) void knn_select_kernel(float* __restrict__ dots, const float* __restrict__ qn,
                                                         const float* __restrict__ gn, int metric, int nq, int ng, int k,
                                                         long long* __restrict__ idx, float* __restrict__ dist) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= nq) return;
  float* d = dots + (size_t)row * ng;
  const float xn = qn[row];
  for (int j = lane; j < ng; j += 64) {
    float v;
    if (metric == 0) v = fmaxf(xn + gn[j] - 2.f * d[j], 0.f);
    else v = 1.f - d[j] * rsqrtf(fmaxf(xn * gn[j], 1e-30f));
    d[j] = v;
  }
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  float pd = -INFINITY;
  int pj = -1;
  for (int t = 0; t < k; ++t) {
    float bd = INFINITY;
    int bj = 0x7fffffff;
    for (int j = lane; j < ng; j += 64) {
      const float v = d[j];
      const bool after = v > pd || (v == pd && j > pj);
      if (after && (v < bd || (v == bd && j < bj))) { bd = v; bj = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float od = __shfl_xor(bd, o, 64);
      const int oj = __shfl_xor(bj, o, 64);
      if (od < bd || (od == bd && oj < bj)) { bd = od; bj = oj; }
    }
    if (lane == 0) {
      idx[(size_t)row * k + t] = bj == 0x7fffffff ? -1 : bj;
      dist[(size_t)row * k + t] = bd;
    }
    pd = bd;
    pj = bj;
  }
}

// exact reported distances: euclidean sqrt(sum (x - g)^2); cosine 1 - x.g/(|x||g|) from a direct dot product
__global__ __launch_bounds__(256) void knn_refine_kernel(const float* __restrict__ X, const float* __restrict__ G,
                                                         const long long* __restrict__ idx, float* __restrict__ dist,
                                                         int metric, int nq, int k, int D) {
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (item >= nq * k) return;
  const long long g = idx[item];
  if (g < 0) return;
  const float* x = X + (size_t)(item / k) * D;
  const float* y = G + (size_t)g * D;
  float a = 0.f, b = 0.f, c = 0.f;
  for (int j = lane; j < D; j += 64) {
    const float u = x[j], v = y[j];
    if (metric == 0) { const float e = u - v; a = fmaf(e, e, a); }
    else { a = fmaf(u, v, a); b = fmaf(u, u, b); c = fmaf(v, v, c); }
  }
  a = wave_sum(a);
  if (metric != 0) { b = wave_sum(b); c = wave_sum(c); }
  if (lane == 0) dist[item] = metric == 0 ? sqrtf(a) : fmaxf(1.f - a * rsqrtf(fmaxf(b * c, 1e-30f)), 0.f);
}

// idx / dist: [nq][m] neighbours (m = k x number of query modalities, hstacked as src/ann.py:20-21); labels: [ng]
__global__ __launch_bounds__(256) void knn_vote_kernel(const long long* __restrict__ idx, const float* __restrict__ dist,
                                                       const long long* __restrict__ labels, int nq, int m,
                                                       long long* __restrict__ pred) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= nq) return;
  const long long* id = idx + (size_t)q * m;
  const float* ds = dist + (size_t)q * m;
  bool any_zero = false;
  for (int j = 0; j < m; ++j) any_zero |= (id[j] >= 0 && ds[j] == 0.f);
  long long best_c = -1;
  float best_w = -1.f;
  for (int j = 0; j < m; ++j) {
    if (id[j] < 0) continue;
    const long long c = labels[id[j]];
    float w = 0.f;
    for (int l = 0; l < m; ++l) {                       // total weight of class c, summed in neighbour order
      if (id[l] < 0 || labels[id[l]] != c) continue;
      w += any_zero ? (ds[l] == 0.f ? 1.f : 0.f) : 1.f / ds[l];
    }
    if (w > best_w || (w == best_w && c < best_c)) { best_w = w; best_c = c; }
  }
  pred[q] = best_c;
}

extern "C" {

int mpr_knn_sqnorm(const float* X, float* out, int rows, int D, void* stream) {
  MPR_REQUIRE(X && out && rows > 0 && D > 0, "mpr_knn_sqnorm: bad arguments");
  knn_sqnorm_kernel<<<ceil_div(rows, 4), 256, 0, (hipStream_t)stream>>>(X, out, rows, D);
  MPR_LAUNCH_CHECK("knn_sqnorm_kernel");
  return MPR_OK;
}

int mpr_knn_select(float* dots, const float* q_sqnorm, const float* g_sqnorm, const float* X, const float* G, int metric,
                   int nq, int ng, int k, int D, long long* idx, float* dist, void* stream) {
  MPR_REQUIRE(dots && q_sqnorm && g_sqnorm && X && G && idx && dist, "mpr_knn_select: null pointer");
  MPR_REQUIRE(nq > 0 && ng > 0 && k > 0 && D > 0 && (metric == 0 || metric == 1), "mpr_knn_select: bad sizes / metric");
  hipStream_t st = (hipStream_t)stream;
  knn_select_kernel<<<ceil_div(nq, 4), 256, 0, st>>>(dots, q_sqnorm, g_sqnorm, metric, nq, ng, k, idx, dist);
  MPR_LAUNCH_CHECK("knn_select_kernel");
  knn_refine_kernel<<<ceil_div(nq * k, 4), 256, 0, st>>>(X, G, idx, dist, metric, nq, k, D);
  MPR_LAUNCH_CHECK("knn_refine_kernel");
  return MPR_OK;
}

int mpr_knn_vote(const long long* idx, const float* dist, const long long* labels, int nq, int m, long long* pred,
                 void* stream) {
  MPR_REQUIRE(idx && dist && labels && pred && nq > 0 && m > 0, "mpr_knn_vote: bad arguments");
  knn_vote_kernel<<<ceil_div(nq, 256), 256, 0, (hipStream_t)stream>>>(idx, dist, labels, nq, m, pred);
  MPR_LAUNCH_CHECK("knn_vote_kernel");
  return MPR_OK;
}

}  // extern "C"
