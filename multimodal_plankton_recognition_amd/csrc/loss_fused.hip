// CLIP / SigLIP coordination losses WITHOUT the similarity matrix in memory (reference: src/coordination.py:26-47, 76-95;
// + beta * MSE of :60-64 / :108-112 in the normalisation backward).  fp32 end to end: every 64 x 64 tile of
// S = X Y^T is formed on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, the k-ordered fmaf chain of gemm_f32.hip), consumed
// in registers / LDS and dropped; the backward pass forms it again instead of reading it back.
//
// Row-block form, the same for one process and for data parallel (distributed.py): a "role" z owns `b` rows of one
// modality (z = 0: my images against ALL profiles, z = 1: my profiles against ALL images) and `n = world * b` columns;
// the positive of row i sits at column off + i.  Per role
//   forward : lse_own[i] = log sum_j exp(scale * S_ij)          (clipf_lse_kernel: per-tile (max, sum) pairs -> clipf_finish)
//             share      = sum_i (lse_own[i] - scale * S_i,off+i)
//   backward: G_ij = coef * (exp(l_ij - lse_own[i]) + exp(l_ij - lse_other[j]) - 2 [j == off + i]) * scale,
//             dX = G Y   (clipf_grad_kernel: a 64-row strip walks its share of the column tiles, S tile -> G tile in LDS ->
//             MFMA into a 64 x 512 accumulator held in registers; partial strips are summed by clipf_norm_bwd_kernel,
//             which also applies the F.normalize backward, the MSE term and the upstream gradient)
// lse_other is the OTHER role's lse_own of all ranks (all-gathered by the caller; one process: its own array).
// SigLIP uses the same two tile kernels with G_ij = -sg * coef * sigmoid(-sg * z_ij) * scale, z = l + bias, sg = +1 on
// the positives: its forward needs no second axis (role 0 only), its backward runs both roles for dX.
// `buckets` (src/coordination.py:29-37) = independent problems c = 0..nprob-1 over consecutive row blocks (one process only).
// Five launches per step (normalise, tiles, finish, gradient tiles, normalise backward) against thirteen for the
// materialised form; HBM traffic is the embeddings plus nchunk partial strips.
#include "common.h"

#define CF_BK 32
#define CF_MAXCHUNK 16

struct ClipfParams {
  const float* x;        // rows of this rank:  x + z * zstride + (c * b + i) * D
  const float* g;        // rows of all ranks:  g + (j / b) * rstride + (1 - z) * zstride + (c * b + j % b) * D
  long long zstride, rstride;
  const float* ls;       // logit_scale (the multiplier is exp)
  const float* bias;     // SigLIP bias; NULL: CLIP
  int b, n, D, off, nprob, nct;
  // forward
  float* part_m;         // [2][nprob][nct][b]   per-tile row maximum       (SigLIP: unused)
  float* part_s;         // [2][nprob][nct][b]   per-tile sum exp(l - max)  (SigLIP: [nprob][nrt][nct] partial sums)
  float* diag;           // [2][nprob * b]
  // backward
  const float* lse_own;  // [2][nprob * b]
  const float* lse_other;   // index (j / b) * lrs + (1 - z) * lzs + c * b + j % b
  int lzs, lrs;
  float coef;
  float* dpart;          // [2][nprob][nchunk][b][D]
  float* spart;          // [2][nprob * nrt * nchunk]: d logit_scale, d bias partial sums (role 0)
  int nchunk, tiles_per_chunk;
  int ngroup, tiles_per_group;      // forward: column tiles per workgroup
};

struct CfStage {          // row-major operand chunks, 32 depth values (+ 4 of padding: conflict-free 16-byte reads) per row
  float a[2][64][CF_BK + 4];
  float b[2][64][CF_BK + 4];
};

__device__ __forceinline__ float cf_log_sigmoid(float x) { return fminf(x, 0.f) - log1pf(expf(-fabsf(x))); }

// 4 consecutive depth values of one row (V4: D % 4 == 0, one 16-byte load; otherwise element-wise with the tail masked)
template <bool V4>
__device__ __forceinline__ float4 cf_ld4(const float* __restrict__ base, int off, int gk, int D) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (off < 0) return v;
  if (V4) {
    if (gk < D) v = *reinterpret_cast<const float4*>(base + off + gk);
  } else {
    if (gk < D) v.x = base[off + gk];
    if (gk + 1 < D) v.y = base[off + gk + 1];
    if (gk + 2 < D) v.z = base[off + gk + 2];
    if (gk + 3 < D) v.w = base[off + gk + 3];
  }
  return v;
}

// 64 x 64 tile of X Y^T over the full depth D: each of the 4 waves returns its 32 x 32 quadrant (wm = wid >> 1 rows,
// wn = wid & 1 columns); element e of the result is row (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), column lane & 31.
// Operand chunks of 32 depth values are staged row-major in double-buffered LDS (16-byte global loads, 16-byte LDS
// writes, two register sets of look-ahead as in gemm_f32.hip); an MFMA lane reads 4 consecutive depth values of its row
// with one ds_read_b128 and feeds them to 4 MFMAs -- the depth index of MFMA step c of group t is 8 t + 4 (lane >> 5) + c
// for BOTH operands, i.e. a permutation of the summation order, not of the product.  Ends with a barrier: the staging
// buffers are free on return.
template <bool V4>
__device__ __forceinline__ f32x16 clipf_s_tile(const ClipfParams& p, CfStage& sm, const float* __restrict__ xb,
                                               const float* __restrict__ yb, int m0, int n0) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid >> 1, wn = wid & 1;
  const int kq = tid & 7, r0 = tid >> 3;
  int xo[2], yo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = r0 + 32 * i, gi = m0 + r, gj = n0 + r;
    xo[i] = gi < p.b ? gi * p.D : -1;
    yo[i] = gj < p.n ? (int)((gj / p.b) * p.rstride) + (gj % p.b) * p.D : -1;
  }
  float4 ra[2][2], rb[2][2];
  auto load = [&](int k0, float4* xa, float4* xv) {
    const int gk = k0 + 4 * kq;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      xa[i] = cf_ld4<V4>(xb, xo[i], gk, p.D);
      xv[i] = cf_ld4<V4>(yb, yo[i], gk, p.D);
    }
  };
  auto store = [&](int buf, const float4* xa, const float4* xv) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<float4*>(&sm.a[buf][r0 + 32 * i][4 * kq]) = xa[i];
      *reinterpret_cast<float4*>(&sm.b[buf][r0 + 32 * i][4 * kq]) = xv[i];
    }
  };
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int nk = (p.D + CF_BK - 1) / CF_BK;
  load(0, ra[0], rb[0]);
  if (nk > 1) load(CF_BK, ra[1], rb[1]);
  store(0, ra[0], rb[0]);
  __syncthreads();
  auto step = [&](int t, float4* free_a, float4* free_b, const float4* next_a, const float4* next_b) {
    const int cur = t & 1;
    if (t + 2 < nk) load((t + 2) * CF_BK, free_a, free_b);
#pragma unroll
    for (int t4 = 0; t4 < CF_BK / 8; ++t4) {
      const float4 a = *reinterpret_cast<const float4*>(&sm.a[cur][wm * 32 + (lane & 31)][8 * t4 + 4 * (lane >> 5)]);
      const float4 b = *reinterpret_cast<const float4*>(&sm.b[cur][wn * 32 + (lane & 31)][8 * t4 + 4 * (lane >> 5)]);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    }
    if (t + 1 < nk) store(cur ^ 1, next_a, next_b);
    __syncthreads();
  };
  for (int t = 0; t < nk; t += 2) {
    step(t, ra[0], rb[0], ra[1], rb[1]);
    if (t + 1 < nk) step(t + 1, ra[1], rb[1], ra[0], rb[0]);
  }
  return acc;
}

// ------------------------------------------------------------------------------------------------ normalise
// u = x / max(|x|, 1e-12) for both modalities in one launch (F.normalize, src/coordination.py:33-34): uv [2][rows][D]
__global__ __launch_bounds__(256) void clipf_norm_kernel(const float* __restrict__ a, const float* __restrict__ p,
                                                         float* __restrict__ uv, float* __restrict__ inv, int rows, int D) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, z = blockIdx.y;
  if (row >= rows) return;
  const float* xr = (z ? p : a) + (size_t)row * D;
  float* ur = uv + ((size_t)z * rows + row) * D;
  float ss = 0.f;
  for (int j = lane; j < D; j += 64) ss = fmaf(xr[j], xr[j], ss);
  ss = wave_sum(ss);
  const float r = 1.f / fmaxf(sqrtf(ss), 1e-12f);
  for (int j = lane; j < D; j += 64) ur[j] = xr[j] * r;
  if (lane == 0) inv[z * rows + row] = r;
}

// ------------------------------------------------------------------------------------------------ forward tiles
// grid (ngroup, nrt, roles * nprob): a workgroup walks tiles_per_group column tiles of one 64-row strip, keeping the
// running (max, sum) pair of its rows in registers
template <bool V4>
__global__ __launch_bounds__(256) void clipf_lse_kernel(const ClipfParams p) {
  __shared__ CfStage sm;
  __shared__ float tile[64][65];
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid >> 1, wn = wid & 1;
  const int grp = blockIdx.x, rt = blockIdx.y, z = blockIdx.z / p.nprob, c = blockIdx.z - z * p.nprob;
  const int m0 = rt * 64;
  const float* xb = p.x + z * p.zstride + (long long)c * p.b * p.D;
  const float* yb = p.g + (1 - z) * p.zstride + (long long)c * p.b * p.D;
  const float scale = expf(p.ls[0]);
  const bool sig = p.bias != nullptr;
  const float bb = sig ? p.bias[0] : 0.f;
  const int r = tid >> 2, q = tid & 3, gi = m0 + r;
  float m = -INFINITY, s = 0.f, dg = 0.f;     // CLIP: running pair over this thread's columns; SigLIP: s = running sum
  bool has_dg = false;
  const int ct0 = grp * p.tiles_per_group, ct1 = min(ct0 + p.tiles_per_group, p.nct);
  for (int ct = ct0; ct < ct1; ++ct) {
    const int n0 = ct * 64;
    const f32x16 acc = clipf_s_tile<V4>(p, sm, xb, yb, m0, n0);     // (its barriers also fence the previous tile's readers)
#pragma unroll
    for (int e = 0; e < 16; ++e)
      tile[wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)][wn * 32 + (lane & 31)] = acc[e] * scale;
    __syncthreads();
    if (sig) {
      if (gi < p.b)
        for (int jj = 0; jj < 16; ++jj) {
          const int j = n0 + q * 16 + jj;
          if (j < p.n) {
            const float zv = tile[r][q * 16 + jj] + bb;
            s -= cf_log_sigmoid(j == p.off + gi ? zv : -zv);
          }
        }
      continue;
    }
    float mt = -INFINITY;
    for (int jj = 0; jj < 16; ++jj)
      if (n0 + q * 16 + jj < p.n) mt = fmaxf(mt, tile[r][q * 16 + jj]);
    if (mt > -INFINITY) {
      const float mn = fmaxf(m, mt);
      float st = 0.f;
      for (int jj = 0; jj < 16; ++jj)
        if (n0 + q * 16 + jj < p.n) st += expf(tile[r][q * 16 + jj] - mn);
      s = s * expf(m - mn) + st;               // (m == -inf: s == 0 and exp(-inf) == 0)
      m = mn;
    }
    const int dj = p.off + gi - n0;
    if (q == 0 && dj >= 0 && dj < 64) { dg = tile[r][dj]; has_dg = true; }
  }
  if (sig) {
    s = wave_sum(s);
    if (lane == 0) red[wid] = s;
    __syncthreads();
    if (tid == 0) p.part_s[((size_t)c * gridDim.y + rt) * gridDim.x + grp] = red[0] + red[1] + red[2] + red[3];
    return;
  }
  // the 4 threads of a row hold disjoint column sets: merge the pairs (every group has a valid column: m is finite
  // in at least one of them)
#pragma unroll
  for (int o = 1; o <= 2; o <<= 1) {
    const float mo = __shfl_xor(m, o), so = __shfl_xor(s, o);
    const float mn = fmaxf(m, mo);
    s = (m > -INFINITY ? s * expf(m - mn) : 0.f) + (mo > -INFINITY ? so * expf(mo - mn) : 0.f);
    m = mn;
  }
  if (q == 0 && gi < p.b) {
    const size_t o = (((size_t)z * p.nprob + c) * p.ngroup + grp) * p.b + gi;
    p.part_m[o] = m;
    p.part_s[o] = s;
    if (has_dg) p.diag[((size_t)z * p.nprob + c) * p.b + gi] = dg;
  }
}

// lse[idx] over the tile pairs; out[0] = mul * sum (lse - diag)          (one workgroup)
__global__ __launch_bounds__(1024) void clipf_finish_kernel(const float* __restrict__ part_m, const float* __restrict__ part_s,
                                                            const float* __restrict__ diag, float* __restrict__ lse,
                                                            float* __restrict__ out, int total, int b, int nct, float mul) {
  __shared__ double red[1024];
  double a = 0.0;
  for (int idx = threadIdx.x; idx < total; idx += 1024) {
    const int zc = idx / b, i = idx - zc * b;
    const float* pm = part_m + (size_t)zc * nct * b + i;
    const float* ps = part_s + (size_t)zc * nct * b + i;
    float m = pm[0];
#pragma unroll 8
    for (int t = 1; t < nct; ++t) m = fmaxf(m, pm[(size_t)t * b]);
    float s = 0.f;
#pragma unroll 8
    for (int t = 0; t < nct; ++t) s += ps[(size_t)t * b] * expf(pm[(size_t)t * b] - m);
    const float l = m + logf(s);
    lse[idx] = l;
    a += (double)l - (double)diag[idx];
  }
  red[threadIdx.x] = a;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(red[0] * (double)mul);
}

// out[0] = mul * sum part[0..n)                                          (one workgroup)
__global__ __launch_bounds__(256) void clipf_sum_kernel(const float* __restrict__ part, int n, float mul, float* __restrict__ out) {
  __shared__ double red[256];
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) a += (double)part[i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(red[0] * (double)mul);
}

// ------------------------------------------------------------------------------------------------ gradient tiles
// grid (nchunk * ndc, nrt, 2 * nprob): strip of 64 rows x this chunk's column tiles x 512 embedding columns
template <bool V4>
__global__ __launch_bounds__(256) void clipf_grad_kernel(const ClipfParams p) {
  __shared__ CfStage sm;
  __shared__ float gs[64][64 + 4];        // G tile, column(j)-major: the A operand of dX += G Y
  __shared__ float lo[64];
  __shared__ float red[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid >> 1, wn = wid & 1;
  const int chunk = blockIdx.x % p.nchunk, dc = blockIdx.x / p.nchunk;
  const int rt = blockIdx.y, z = blockIdx.z / p.nprob, c = blockIdx.z - z * p.nprob;
  const int m0 = rt * 64, d0 = dc * 512 + wid * 128;
  const float* xb = p.x + z * p.zstride + (long long)c * p.b * p.D;
  const float* yb = p.g + (1 - z) * p.zstride + (long long)c * p.b * p.D;
  const float scale = expf(p.ls[0]);
  const bool sig = p.bias != nullptr;
  const float bb = sig ? p.bias[0] : 0.f;
  if (!sig && tid < 64) lo[tid] = m0 + tid < p.b ? p.lse_own[((size_t)z * p.nprob + c) * p.b + m0 + tid] : 0.f;
  f32x16 dx[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) dx[a][t][e] = 0.f;
  float dls = 0.f, db = 0.f;
  int dld[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) dld[t] = V4 ? min(d0 + 4 * (lane & 31), p.D - 4) : min(d0 + 4 * (lane & 31) + t, p.D - 1);
  const int ct0 = chunk * p.tiles_per_chunk, ct1 = min(ct0 + p.tiles_per_chunk, p.nct);
  for (int ct = ct0; ct < ct1; ++ct) {
    const int n0 = ct * 64;
    const f32x16 acc = clipf_s_tile<V4>(p, sm, xb, yb, m0, n0);        // (its first barrier also publishes lo[])
    const int j = n0 + wn * 32 + (lane & 31);
    float lt = 0.f;
    if (!sig && j < p.n) lt = p.lse_other[(size_t)(j / p.b) * p.lrs + (size_t)(1 - z) * p.lzs + (size_t)c * p.b + j % p.b];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int il = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), gi = m0 + il;
      const float l = acc[e] * scale;
      float g = 0.f;
      if (gi < p.b && j < p.n) {
        const bool pos = j == p.off + gi;
        if (sig) {
          const float sg = pos ? 1.f : -1.f;
          g = -sg * p.coef / (1.f + expf(sg * (l + bb)));        // d/dz [-logsigmoid(sg z)] = -sg sigmoid(-sg z)
        } else {
          g = (expf(l - lo[il]) + expf(l - lt) - (pos ? 2.f : 0.f)) * p.coef;
        }
        dls = fmaf(g, l, dls);
        db += g;
      }
      gs[wn * 32 + (lane & 31)][il] = g * scale;
    }
    __syncthreads();
    // dX[64][128 of this wave] += G[64 x 64] Y[64 x 128]: Y rows straight from global memory (L2-resident embeddings),
    // one 16-byte load per lane and depth step -- accumulator tile t of a lane holds column d0 + 4 (lane & 31) + t.
    // Rows past n / columns past D are clamped, not masked: their G entries are zero / their results are not stored.
    {
      int gj = min(n0 + (lane >> 5), p.n - 1);
      int q = gj / p.b, r = gj - q * p.b;
#pragma unroll 4
      for (int s2 = 0; s2 < 32; ++s2) {
        const int jl = 2 * s2 + (lane >> 5);
        const float a0 = gs[jl][lane & 31], a1 = gs[jl][32 + (lane & 31)];
        const float* yr = yb + (long long)q * p.rstride + (long long)r * p.D;
        float bv[4];
        if (V4) {
          const float4 v = *reinterpret_cast<const float4*>(yr + dld[0]);
          bv[0] = v.x; bv[1] = v.y; bv[2] = v.z; bv[3] = v.w;
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) bv[t] = yr[dld[t]];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          dx[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv[t], dx[0][t], 0, 0, 0);
          dx[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv[t], dx[1][t], 0, 0, 0);
        }
        if (gj + 2 < p.n) {
          gj += 2;
          r += 2;
          while (r >= p.b) { r -= p.b; ++q; }
        }
      }
    }
    __syncthreads();
  }
  float* out = p.dpart + ((((size_t)z * p.nprob + c) * p.nchunk + chunk) * p.b) * p.D;
  const int dcol = d0 + 4 * (lane & 31);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int gi = m0 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
      if (gi >= p.b) continue;
      if (V4) {
        if (dcol < p.D)
          *reinterpret_cast<float4*>(out + (size_t)gi * p.D + dcol) = make_float4(dx[a][0][e], dx[a][1][e], dx[a][2][e], dx[a][3][e]);
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if (dcol + t < p.D) out[(size_t)gi * p.D + dcol + t] = dx[a][t][e];
      }
    }
  if (z == 0 && dc == 0) {
    dls = wave_sum(dls);
    db = wave_sum(db);
    if (lane == 0) { red[0][wid] = dls; red[1][wid] = db; }
    __syncthreads();
    if (tid == 0) {
      const int nparts = p.nprob * gridDim.y * p.nchunk;
      const int slot = (c * gridDim.y + rt) * p.nchunk + chunk;
      p.spart[slot] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
      p.spart[nparts + slot] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
  }
}

// dx = gout * ( inv * (du - u (u . du)) + mse_coef * (x - other) ),  du = sum of the nchunk partial strips; both
// modalities in one launch (grid.y = 2); the last workgroup of role 0 reduces d logit_scale / d bias.
__global__ __launch_bounds__(256) void clipf_norm_bwd_kernel(const float* __restrict__ dpart, int nchunk, int nprob, int b,
                                                             const float* __restrict__ uv, const float* __restrict__ inv,
                                                             const float* __restrict__ a, const float* __restrict__ pr,
                                                             float mse_coef, const float* __restrict__ gout,
                                                             float* __restrict__ da, float* __restrict__ dp,
                                                             const float* __restrict__ spart, int nparts,
                                                             float* __restrict__ d_ls, float* __restrict__ d_bias, int D) {
  const int rows = nprob * b, z = blockIdx.y, lane = threadIdx.x & 63;
  const float g = gout ? gout[0] : 1.f;
  if (blockIdx.x == gridDim.x - 1) {                 // the extra workgroup: scalar gradients
    if (z == 0 && (d_ls || d_bias)) {
      __shared__ double red[2][256];
      double s0 = 0.0, s1 = 0.0;
      for (int i = threadIdx.x; i < nparts; i += 256) { s0 += (double)spart[i]; s1 += (double)spart[nparts + i]; }
      red[0][threadIdx.x] = s0;
      red[1][threadIdx.x] = s1;
      __syncthreads();
      for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
        __syncthreads();
      }
      if (threadIdx.x == 0) {
        if (d_ls) d_ls[0] = (float)red[0][0] * g;
        if (d_bias) d_bias[0] = (float)red[1][0] * g;
      }
    }
    return;
  }
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int c = row / b, i = row - c * b;
  const float* part = dpart + ((((size_t)z * nprob + c) * nchunk) * b + i) * D;
  const float* ur = uv + ((size_t)z * rows + row) * D;
  const float* xr = a ? (z ? pr : a) + (size_t)row * D : nullptr;
  const float* orow = a ? (z ? a : pr) + (size_t)row * D : nullptr;
  float* out = (z ? dp : da) + (size_t)row * D;
  // D <= 64 * 16 is kept in registers (one pass over the partial strips); longer rows take the two-pass route
  float du[16];
  float dot = 0.f;
  const bool inreg = D <= 1024;
  if (inreg) {
    // strip-major: the (up to 16) loads of one strip are independent and go out together -- with the strips innermost every
    // element waited for its nchunk loads one after the other (this kernel runs alone on the chip, in the junction of the
    // step).  Same order of additions per element.
#pragma unroll
    for (int k = 0; k < 16; ++k) du[k] = 0.f;
    for (int t0 = 0; t0 < nchunk; t0 += 4) {             // four strips of loads in flight (the kernel is latency-bound)
      float tmp[4][16];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* pt = part + (size_t)(t0 + u) * b * D;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const int d = lane + 64 * k;
          tmp[u][k] = (t0 + u < nchunk && d < D) ? pt[d] : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < 16; ++k) du[k] += tmp[u][k];      // (+ 0.f for the strips / columns that do not exist)
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int d = lane + 64 * k;
      if (d < D) dot = fmaf(ur[d], du[k], dot);
    }
  } else {
    for (int d = lane; d < D; d += 64) {
      float s = 0.f;
      for (int t = 0; t < nchunk; ++t) s += part[(size_t)t * b * D + d];
      dot = fmaf(ur[d], s, dot);
    }
  }
  dot = wave_sum(dot);
  const float r = inv[z * rows + row];
  if (inreg) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int d = lane + 64 * k;
      if (d < D) {
        float v = r * (du[k] - ur[d] * dot);
        if (xr) v += mse_coef * (xr[d] - orow[d]);
        out[d] = v * g;
      }
    }
  } else {
    for (int d = lane; d < D; d += 64) {
      float s = 0.f;
      for (int t = 0; t < nchunk; ++t) s += part[(size_t)t * b * D + d];
      float v = r * (s - ur[d] * dot);
      if (xr) v += mse_coef * (xr[d] - orow[d]);
      out[d] = v * g;
    }
  }
}

static int clipf_nchunk(int nct) { return nct < CF_MAXCHUNK ? nct : CF_MAXCHUNK; }

extern "C" {

// floats of workspace for mpr_clipf_fwd / mpr_clipf_bwd of one problem size (the larger of the two phases)
long long mpr_clipf_workspace_floats(int world, int b, int D, int nprob) {
  const long long n = (long long)world * b, nct = (n + 63) / 64, nrt = (b + 63) / 64;
  const long long fwd = 2 * 2 * (long long)nprob * nct * b + 2ll * nprob * b + (long long)nprob * nrt * nct;
  const long long bwd = 2ll * nprob * clipf_nchunk((int)nct) * b * D + 2ll * nprob * nrt * CF_MAXCHUNK;
  return fwd > bwd ? fwd : bwd;
}

// uv [2][rows][D] = F.normalize(image_emb), F.normalize(profile_emb); inv [2][rows] = 1 / max(|x|, 1e-12)
int mpr_clipf_norm(const float* image_emb, const float* profile_emb, float* uv, float* inv, int rows, int D, void* stream) {
  MPR_REQUIRE(image_emb && profile_emb && uv && inv && rows > 0 && D > 0, "mpr_clipf_norm: bad arguments");
  clipf_norm_kernel<<<dim3(ceil_div(rows, 4), 2), 256, 0, (hipStream_t)stream>>>(image_emb, profile_emb, uv, inv, rows, D);
  MPR_LAUNCH_CHECK("clipf_norm_kernel");
  return MPR_OK;
}

static int clipf_fill(ClipfParams& p, const float* gathered, const float* logit_scale, const float* bias, int world, int rank,
                      int b, int D, int nprob, const char* who) {
  MPR_REQUIRE(gathered && logit_scale, "%s: null pointer", who);
  MPR_REQUIRE(world > 0 && rank >= 0 && rank < world && b > 0 && D > 0 && nprob > 0, "%s: bad sizes", who);
  MPR_REQUIRE(world == 1 || nprob == 1, "%s: buckets need world == 1", who);
  MPR_REQUIRE(2ll * world * nprob * b * D < (1ll << 31), "%s: embeddings too large for 32-bit offsets", who);
  // one process: uv [2][nprob * b][D];   data parallel: gathered [world][2][b][D]
  p.zstride = (long long)nprob * b * D;
  p.rstride = 2ll * b * D;
  p.g = gathered;
  p.x = gathered + (long long)rank * p.rstride;
  p.ls = logit_scale;
  p.bias = bias;
  p.b = b; p.n = world * b; p.D = D; p.off = rank * b; p.nprob = nprob;
  p.nct = ceil_div(p.n, 64);
  return MPR_OK;
}

// Forward.  gathered: [world][2][b][D] normalised embeddings of every rank (world == 1: [2][nprob * b][D], the output of
// mpr_clipf_norm).  CLIP (bias == NULL): lse [2][nprob * b] = this rank's row log-sum-exps of both roles,
// out[0] = mul * sum over both roles (lse - positive logit).  SigLIP: lse unused (may be NULL),
// out[0] = mul * sum over this rank's image rows of -logsigmoid(+-z).
int mpr_clipf_fwd(const float* gathered, const float* logit_scale, const float* bias, float* lse, float* out, float mul,
                  float* workspace, int world, int rank, int b, int D, int nprob, void* stream) {
  ClipfParams p = {};
  const int rc = clipf_fill(p, gathered, logit_scale, bias, world, rank, b, D, nprob, "mpr_clipf_fwd");
  if (rc != MPR_OK) return rc;
  MPR_REQUIRE(out && workspace && (bias || lse), "mpr_clipf_fwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int nrt = ceil_div(b, 64);
  const long long np = 2ll * nprob * p.nct * b;
  p.part_m = workspace;
  p.part_s = workspace + np;
  p.diag = workspace + 2 * np;
  if (bias) p.part_s = workspace;
  const int roles = bias ? 1 : 2;
  MPR_REQUIRE(roles * nprob <= 65535, "mpr_clipf_fwd: too many buckets");
  // ~3 workgroups per CU (they hide each other's load latency), as few (max, sum) pairs per row as that allows
  const int strips = nrt * roles * nprob;
  int want = (768 + strips - 1) / strips;
  if (want > p.nct) want = p.nct;
  p.tiles_per_group = ceil_div(p.nct, want);
  p.ngroup = ceil_div(p.nct, p.tiles_per_group);
  if (D % 4 == 0) clipf_lse_kernel<true><<<dim3(p.ngroup, nrt, roles * nprob), 256, 0, st>>>(p);
  else clipf_lse_kernel<false><<<dim3(p.ngroup, nrt, roles * nprob), 256, 0, st>>>(p);
  MPR_LAUNCH_CHECK("clipf_lse_kernel");
  if (bias) {
    clipf_sum_kernel<<<1, 256, 0, st>>>(p.part_s, nprob * nrt * p.ngroup, mul, out);
    MPR_LAUNCH_CHECK("clipf_sum_kernel");
  } else {
    clipf_finish_kernel<<<1, 1024, 0, st>>>(p.part_m, p.part_s, p.diag, lse, out, 2 * nprob * b, b, p.ngroup, mul);
    MPR_LAUNCH_CHECK("clipf_finish_kernel");
  }
  return MPR_OK;
}

// Backward: d_image [nprob * b][D], d_profile [nprob * b][D] (gradients of the UN-normalised embeddings, x gout),
// d_logit_scale / d_bias [1] (this rank's share, x gout; may be NULL).  CLIP: lse_own [2][nprob * b] of mpr_clipf_fwd,
// lse_other = the all-gathered lse of every rank [world][2][b] (world == 1: lse_own itself).  uv / inv: this rank's
// mpr_clipf_norm outputs.  image_emb / profile_emb + mse_coef add mse_coef * (x - other) (the *Plus losses; NULL: none).
int mpr_clipf_bwd(const float* gathered, const float* logit_scale, const float* bias, const float* lse_own,
                  const float* lse_other, float coef, const float* uv, const float* inv, const float* image_emb,
                  const float* profile_emb, float mse_coef, const float* gout, float* d_image, float* d_profile,
                  float* d_logit_scale, float* d_bias, float* workspace, int world, int rank, int b, int D, int nprob,
                  void* stream) {
  ClipfParams p = {};
  const int rc = clipf_fill(p, gathered, logit_scale, bias, world, rank, b, D, nprob, "mpr_clipf_bwd");
  if (rc != MPR_OK) return rc;
  MPR_REQUIRE(uv && inv && d_image && d_profile && workspace && (bias || (lse_own && lse_other)), "mpr_clipf_bwd: null pointer");
  MPR_REQUIRE((image_emb == nullptr) == (profile_emb == nullptr), "mpr_clipf_bwd: image_emb and profile_emb go together");
  hipStream_t st = (hipStream_t)stream;
  const int nrt = ceil_div(b, 64), ndc = ceil_div(D, 512);
  p.lse_own = lse_own;
  p.lse_other = lse_other;
  p.lzs = world == 1 ? nprob * b : b;
  p.lrs = world == 1 ? 0 : 2 * b;
  p.coef = coef;
  p.nchunk = clipf_nchunk(p.nct);
  p.tiles_per_chunk = ceil_div(p.nct, p.nchunk);
  p.dpart = workspace;
  p.spart = workspace + 2ll * nprob * p.nchunk * b * D;
  MPR_REQUIRE(2 * nprob <= 65535, "mpr_clipf_bwd: too many buckets");
  if (D % 4 == 0) clipf_grad_kernel<true><<<dim3(p.nchunk * ndc, nrt, 2 * nprob), 256, 0, st>>>(p);
  else clipf_grad_kernel<false><<<dim3(p.nchunk * ndc, nrt, 2 * nprob), 256, 0, st>>>(p);
  MPR_LAUNCH_CHECK("clipf_grad_kernel");
  const int rows = nprob * b;
  clipf_norm_bwd_kernel<<<dim3(ceil_div(rows, 4) + 1, 2), 256, 0, st>>>(p.dpart, p.nchunk, nprob, b, uv, inv, image_emb,
                                                                       profile_emb, mse_coef, gout, d_image, d_profile,
                                                                       p.spart, nprob * nrt * p.nchunk, d_logit_scale,
                                                                       d_bias, D);
  MPR_LAUNCH_CHECK("clipf_norm_bwd_kernel");
  return MPR_OK;
}

}  // extern "C"
