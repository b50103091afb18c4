// Train-mode BatchNorm (1-D and 2-D are the same thing on channels-last [rows][C] bf16 tensors),
// forward and backward, HBM-bound: every pass moves 16 B per lane, channel group fixed per thread.
//   forward : statistics come fused from the producing conv's epilogue (partial sums) ->
//             bn_finalize_stats (mean/var -> scale/shift, running-stat update) -> bn_apply
//             (y = [relu](x*scale + shift [+ residual])).
//   backward: bn_bwd_reduce (sum dz, sum dz*xhat, dz = dy * relu-mask) -> bn_bwd_finalize
//             (dgamma, dbeta, per-channel coefficients) -> bn_bwd_apply (dx = k1*dz + k2*x + k3).
// Semantics follow torch.nn.BatchNorm{1,2}d defaults used by the reference
// (src/profile_encoder.py:126,129,168; timm ResNet): biased variance for normalisation, unbiased for
// running_var, momentum 0.1, eps 1e-5.
#include "common.h"

// ---------------------------------------------------------------------------------------------
// partials [nparts][2][C] -> per-channel scale/shift (+ saved mean / invstd, running-stat update)
__global__ __launch_bounds__(1024) void bn_finalize_stats_kernel(
    const float* __restrict__ partials, int nparts, float count, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* running_mean, float* running_var, float momentum, float eps,
    float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean_out,
    float* __restrict__ invstd_out, int C) {
  // (256-thread blocks -- 32 channels x 8 partial-row lanes, 4 waves: small enough to start beside the long-running
  //  conv workgroups of a concurrent stream instead of queueing for a whole CU's worth of wave slots and registers)
  __shared__ double red[2][32][33];
  const int PL = blockDim.x >> 5;
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), pl = threadIdx.x >> 5;
  double s1 = 0.0, s2 = 0.0;
  if (c < C)
    for (int i0 = pl; i0 < nparts; i0 += PL * 8) {     // 16 independent loads in flight
      float a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + PL * u;
        a[u] = i < nparts ? partials[((size_t)i * 2) * C + c] : 0.f;
        b[u] = i < nparts ? partials[((size_t)i * 2 + 1) * C + c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s1 += (double)a[u]; s2 += (double)b[u]; }
    }
  red[0][pl][threadIdx.x & 31] = s1;
  red[1][pl][threadIdx.x & 31] = s2;
  __syncthreads();
  if (pl == 0 && c < C) {
    for (int i = 1; i < PL; ++i) { s1 += red[0][i][threadIdx.x]; s2 += red[1][i][threadIdx.x]; }
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    scale[c] = g * invstd;
    shift[c] = b - (float)mean * g * invstd;
    mean_out[c] = (float)mean;
    invstd_out[c] = invstd;
    if (running_mean) {
      const double unbiased = count > 1.f ? var * (double)count / ((double)count - 1.0) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
  }
}

// Pre-reduction of long partial lists (the stem conv at batch 512 leaves 25088 rows): [nparts][2][C] ->
// [nsplit][2][C], slice s sums rows s, s+nsplit, ...; spreads the read over many CUs instead of C/32 blocks.
__global__ __launch_bounds__(256) void bn_reduce_partials_kernel(const float* __restrict__ partials, int nparts,
                                                                 float* __restrict__ out, int nsplit, int C2) {
  __shared__ double red[8][33];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), pl = threadIdx.x >> 5, s = blockIdx.y;
  double a = 0.0;
  if (c < C2)
    for (int i = s + pl * nsplit; i < nparts; i += 8 * nsplit) a += (double)partials[(size_t)i * C2 + c];
  red[pl][threadIdx.x & 31] = a;
  __syncthreads();
  if (pl == 0 && c < C2) {
    for (int i = 1; i < 8; ++i) a += red[i][threadIdx.x];
    out[(size_t)s * C2 + c] = (float)a;
  }
}

// The same pre-reduction with the finalize step folded in ("last workgroup finishes the job"): every workgroup writes
// its slice row, releases it (agent-scope fence) and takes a ticket; the workgroup that draws the last ticket acquires,
// sums the nsplit slice rows per channel (double) and computes the per-channel coefficients -- one launch where there
// were two.  These 5-us launches sit in the dependent chain of every BatchNorm, forward and backward (80 per step), and
// under the concurrent weight-gradient stream each one queues for 10-30 us.
//   MODE 0 (forward):  scale / shift / mean / invstd (+ running statistics)      MODE 1 (backward): dgamma / dbeta / coef
// `ticket` is one int, zero on entry and reset to zero by the last workgroup.
struct BnFinalizeArgs {
  float count, momentum, eps;
  const float *gamma, *beta, *mean_in, *invstd_in;
  float *running_mean, *running_var, *scale, *shift, *mean_out, *invstd_out;   // MODE 0
  float *dgamma, *dbeta, *coef;                                                // MODE 1
  int accumulate;
};

template <int MODE>
__global__ __launch_bounds__(256) void bn_reduce_finalize_kernel(const float* __restrict__ partials, int nparts,
                                                                 float* __restrict__ out, int nsplit, int C,
                                                                 int* __restrict__ ticket, const BnFinalizeArgs a) {
  const int C2 = 2 * C;
  __shared__ double red[8][33];
  __shared__ int last;
  {
    const int c = blockIdx.x * 32 + (threadIdx.x & 31), pl = threadIdx.x >> 5, s = blockIdx.y;
    double v = 0.0;
    if (c < C2)
      for (int i = s + pl * nsplit; i < nparts; i += 8 * nsplit) v += (double)partials[(size_t)i * C2 + c];
    red[pl][threadIdx.x & 31] = v;
    __syncthreads();
    if (pl == 0 && c < C2) {
      for (int i = 1; i < 8; ++i) v += red[i][threadIdx.x];
      out[(size_t)s * C2 + c] = (float)v;
    }
  }
  __threadfence();                       // release this workgroup's slice row (agent scope)
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(ticket, 1) == (int)(gridDim.x * gridDim.y) - 1;
  __syncthreads();
  if (!last) return;
  __threadfence();                       // acquire the other workgroups' rows
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    double s1 = 0.0, s2 = 0.0;
    for (int s = 0; s < nsplit; ++s) {
      s1 += (double)__builtin_nontemporal_load(out + (size_t)s * C2 + c);
      s2 += (double)__builtin_nontemporal_load(out + (size_t)s * C2 + C + c);
    }
    if (MODE == 0) {
      const double mean = s1 / a.count;
      double var = s2 / a.count - mean * mean;
      if (var < 0.0) var = 0.0;
      const float invstd = (float)(1.0 / sqrt(var + (double)a.eps));
      const float g = a.gamma ? a.gamma[c] : 1.f, b = a.beta ? a.beta[c] : 0.f;
      a.scale[c] = g * invstd;
      a.shift[c] = b - (float)mean * g * invstd;
      a.mean_out[c] = (float)mean;
      a.invstd_out[c] = invstd;
      if (a.running_mean) {
        const double unbiased = a.count > 1.f ? var * (double)a.count / ((double)a.count - 1.0) : var;
        a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * (float)mean;
        a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * (float)unbiased;
      }
    } else {
      const float sum_dz = (float)s1, sum_dzx = (float)s2;
      const float g = a.gamma ? a.gamma[c] : 1.f, is = a.invstd_in[c], mu = a.mean_in[c];
      if (a.dgamma) a.dgamma[c] = a.accumulate ? a.dgamma[c] + sum_dzx : sum_dzx;
      if (a.dbeta) a.dbeta[c] = a.accumulate ? a.dbeta[c] + sum_dz : sum_dz;
      const float k1 = g * is;
      const float k2 = -g * is * is * sum_dzx / a.count;
      const float k3 = -g * is * sum_dz / a.count - k2 * mu;
      a.coef[c] = k1;
      a.coef[C + c] = k2;
      a.coef[2 * C + c] = k3;
    }
  }
  if (threadIdx.x == 0) *ticket = 0;     // ready for the next use of this slot
}

// eval-mode coefficients from the running statistics
__global__ void bn_eval_coefs_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                     const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                                     float* __restrict__ scale, float* __restrict__ shift, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    const float s = gamma[c] / sqrtf(rv[c] + eps);   // exact 1/sqrt: stays close to the fp32 oracle
    scale[c] = s;
    shift[c] = beta[c] - rm[c] * s;
  }
}

__global__ void bn_eval_invstd_kernel(const float* __restrict__ rv, float eps, float* __restrict__ invstd, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) invstd[c] = 1.f / sqrtf(rv[c] + eps);
}

// standalone statistics pass (for producers without a fused epilogue): partials [grid][2][C]
__global__ __launch_bounds__(256) void bn_stats_kernel(const bf16_t* __restrict__ x, float* __restrict__ partials,
                                                       long long nvec, int C) {
  const int cg = C >> 3;
  __shared__ float red[256][17];
  const int nthr = blockDim.x;
  const int g = threadIdx.x % cg;
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  for (long long v = (long long)blockIdx.x * nthr + threadIdx.x; v < nvec; v += (long long)gridDim.x * nthr) {
    float f[8];
    unpack8(reinterpret_cast<const uint4*>(x)[v], f);
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] += f[e]; s2[e] += f[e] * f[e]; }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[threadIdx.x][e] = s1[e]; red[threadIdx.x][8 + e] = s2[e]; }
  __syncthreads();
  for (int i = threadIdx.x; i < cg * 16; i += nthr) {
    const int gg = i >> 4, e = i & 15;
    float a = 0.f;
    for (int t = gg; t < nthr; t += cg) a += red[t][e];
    partials[((size_t)blockIdx.x * 2 + (e >> 3)) * C + gg * 8 + (e & 7)] = a;
  }
  (void)g;
}

// y = [relu](x*scale + shift [+ res])
// Finalize folded into the CONSUMER: every workgroup of the apply kernels sums the (few) pre-reduced slice rows itself
// and derives the per-channel coefficients into LDS -- redundant arithmetic on a few KB of L2-resident data instead of a
// 5-us launch in the dependent chain of every BatchNorm (which queues for 10-30 us behind the concurrent streams).
// No cross-workgroup synchronisation: workgroup 0 alone writes the results anyone else needs later.
//   MODE 0: lds[0] = scale, lds[1] = shift            MODE 1: lds[0..2] = k1, k2, k3 (dx = k1*dz + k2*x + k3)
constexpr int FIN_MAX_C = 512;
template <int MODE>
__device__ __forceinline__ void bn_block_finalize(const float* __restrict__ slices, int nsl, int C, const BnFinalizeArgs& a,
                                                  float (*lds)[FIN_MAX_C]) {
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    double s1 = 0.0, s2 = 0.0;
    for (int r = 0; r < nsl; ++r) {
      s1 += (double)slices[((size_t)r * 2) * C + c];
      s2 += (double)slices[((size_t)r * 2 + 1) * C + c];
    }
    if (MODE == 0) {
      const double mean = s1 / a.count;
      double var = s2 / a.count - mean * mean;
      if (var < 0.0) var = 0.0;
      const float invstd = (float)(1.0 / sqrt(var + (double)a.eps));
      const float g = a.gamma ? a.gamma[c] : 1.f, b = a.beta ? a.beta[c] : 0.f;
      const float sc = g * invstd, sh = b - (float)mean * g * invstd;
      lds[0][c] = sc;
      lds[1][c] = sh;
      if (blockIdx.x == 0) {
        a.scale[c] = sc;
        a.shift[c] = sh;
        a.mean_out[c] = (float)mean;
        a.invstd_out[c] = invstd;
        if (a.running_mean) {
          const double unbiased = a.count > 1.f ? var * (double)a.count / ((double)a.count - 1.0) : var;
          a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * (float)mean;
          a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * (float)unbiased;
        }
      }
    } else {
      const float sum_dz = (float)s1, sum_dzx = (float)s2;
      const float g = a.gamma ? a.gamma[c] : 1.f, is = a.invstd_in[c], mu = a.mean_in[c];
      // count <= 0: BatchNorm ran on its RUNNING statistics (eval mode; mean_in / invstd_in are those): an affine
      // map, dx = gamma * invstd * dz -- the batch-statistics terms vanish
      const bool ev = a.count <= 0.f;
      const float k1 = g * is;
      const float k2 = ev ? 0.f : -g * is * is * sum_dzx / a.count;
      lds[0][c] = k1;
      lds[1][c] = k2;
      lds[2][c] = ev ? 0.f : -g * is * sum_dz / a.count - k2 * mu;
      if (blockIdx.x == 0) {
        if (a.dgamma) a.dgamma[c] = a.accumulate ? a.dgamma[c] + sum_dzx : sum_dzx;
        if (a.dbeta) a.dbeta[c] = a.accumulate ? a.dbeta[c] + sum_dz : sum_dz;
      }
    }
  }
  __syncthreads();
}

// ACT: 0 none, 1 ReLU, 2 SiLU (EfficientNet's BatchNormAct2d)
template <int ACT, bool RES>
__global__ __launch_bounds__(256) void bn_apply_kernel(const bf16_t* __restrict__ x, const float* __restrict__ scale,
                                                       const float* __restrict__ shift,
                                                       const bf16_t* __restrict__ res, bf16_t* __restrict__ y,
                                                       long long nvec, int cg, const float* __restrict__ slices, int nsl,
                                                       const BnFinalizeArgs fin) {
  // blockDim is a multiple of cg (host guarantees): the channel group is thread-invariant
  const int g = threadIdx.x % cg;
  float sc[8], sh[8];
  __shared__ float fl[2][FIN_MAX_C];
  if (slices) {          // statistics not finalized yet: do it here (see bn_block_finalize)
    bn_block_finalize<0>(slices, nsl, cg * 8, fin, fl);
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = fl[0][g * 8 + e]; sh[e] = fl[1][g * 8 + e]; }
  } else {
    *reinterpret_cast<float4*>(sc) = reinterpret_cast<const float4*>(scale)[g * 2];
    *reinterpret_cast<float4*>(sc + 4) = reinterpret_cast<const float4*>(scale)[g * 2 + 1];
    *reinterpret_cast<float4*>(sh) = reinterpret_cast<const float4*>(shift)[g * 2];
    *reinterpret_cast<float4*>(sh + 4) = reinterpret_cast<const float4*>(shift)[g * 2 + 1];
  }
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec;
       v += (long long)gridDim.x * blockDim.x) {
    float f[8];
    unpack8(reinterpret_cast<const uint4*>(x)[v], f);
    float r[8];
    if (RES) unpack8(reinterpret_cast<const uint4*>(res)[v], r);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = fmaf(f[e], sc[e], sh[e]);
      if (RES) t += r[e];
      if (ACT == 1) t = fmaxf(t, 0.f);
      if (ACT == 2) t = t / (1.f + __expf(-t));
      f[e] = t;
    }
    reinterpret_cast<uint4*>(y)[v] = pack8(f);
  }
}

// A downsampling block's output in ONE pass: y = act(BN(x) + bf16(BN_r(xr))), both BatchNorms on raw conv outputs.  The
// shortcut's normalised map ("identity") is never written and read back; it is rounded to bf16 in the register exactly
// as the stored map was, so the result is bit-identical to bn_apply(xr) followed by bn_apply(x, residual).
template <int ACT>
__global__ __launch_bounds__(256) void bn_apply_dual_kernel(const bf16_t* __restrict__ x, const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ slices, int nsl,
                                                            const BnFinalizeArgs fin, const bf16_t* __restrict__ xr,
                                                            const float* __restrict__ scale_r,
                                                            const float* __restrict__ shift_r,
                                                            const float* __restrict__ slices_r, int nsl_r,
                                                            const BnFinalizeArgs fin_r, bf16_t* __restrict__ y,
                                                            long long nvec, int cg) {
  const int g = threadIdx.x % cg;
  float sc[8], sh[8], scr[8], shr[8];
  __shared__ float fl[2][FIN_MAX_C];
  auto coefs = [&](const float* slc, int n, const BnFinalizeArgs& a, const float* scp, const float* shp, float* c, float* h) {
    if (slc) {
      bn_block_finalize<0>(slc, n, cg * 8, a, fl);
#pragma unroll
      for (int e = 0; e < 8; ++e) { c[e] = fl[0][g * 8 + e]; h[e] = fl[1][g * 8 + e]; }
      __syncthreads();          // fl is reused by the second BatchNorm
    } else {
      *reinterpret_cast<float4*>(c) = reinterpret_cast<const float4*>(scp)[g * 2];
      *reinterpret_cast<float4*>(c + 4) = reinterpret_cast<const float4*>(scp)[g * 2 + 1];
      *reinterpret_cast<float4*>(h) = reinterpret_cast<const float4*>(shp)[g * 2];
      *reinterpret_cast<float4*>(h + 4) = reinterpret_cast<const float4*>(shp)[g * 2 + 1];
    }
  };
  coefs(slices, nsl, fin, scale, shift, sc, sh);
  coefs(slices_r, nsl_r, fin_r, scale_r, shift_r, scr, shr);
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec;
       v += (long long)gridDim.x * blockDim.x) {
    float f[8], r[8];
    unpack8(reinterpret_cast<const uint4*>(x)[v], f);
    unpack8(reinterpret_cast<const uint4*>(xr)[v], r);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = fmaf(f[e], sc[e], sh[e]) + round_bf16(fmaf(r[e], scr[e], shr[e]));
      if (ACT == 1) t = fmaxf(t, 0.f);
      f[e] = t;
    }
    reinterpret_cast<uint4*>(y)[v] = pack8(f);
  }
}

// mask modes for the backward passes
enum { MASK_NONE = 0, MASK_Y = 1, MASK_RECOMPUTE = 2, MASK_SILU = 3 };   // MASK_SILU: dz = dy * silu'(x*scale + shift)

template <int MODE>
__device__ __forceinline__ void masked_dz(const uint4* dy, const uint4* ymask, const float* xf, const float* sc,
                                          const float* sh, long long v, float* dz) {
  unpack8(dy[v], dz);
  if (MODE == MASK_Y) {
    float yv[8];
    unpack8(ymask[v], yv);
#pragma unroll
    for (int e = 0; e < 8; ++e) dz[e] = yv[e] > 0.f ? dz[e] : 0.f;
  } else if (MODE == MASK_RECOMPUTE) {
#pragma unroll
    for (int e = 0; e < 8; ++e) dz[e] = round_bf16(fmaf(xf[e], sc[e], sh[e])) > 0.f ? dz[e] : 0.f;
  } else if (MODE == MASK_SILU) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float t = fmaf(xf[e], sc[e], sh[e]), sg = 1.f / (1.f + __expf(-t));
      dz[e] *= sg * fmaf(t, 1.f - sg, 1.f);
    }
  }
}

// partials [grid][2][C]: sum dz, sum dz*xhat
template <int MODE>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(
    const bf16_t* __restrict__ dy, const bf16_t* __restrict__ ymask, const bf16_t* __restrict__ x,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ scale,
    const float* __restrict__ shift, float* __restrict__ partials, long long nvec, int C, int nslices) {
  const int cg = C >> 3;
  __shared__ float red[256][17];
  const int nthr = blockDim.x;   // a multiple of cg (host guarantees), so the channel group is thread-invariant
  const int g = threadIdx.x % cg;
  float mu[8], is[8], sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    mu[e] = mean[g * 8 + e];
    is[e] = invstd[g * 8 + e];
    sc[e] = MODE >= MASK_RECOMPUTE ? scale[g * 8 + e] : 0.f;
    sh[e] = MODE >= MASK_RECOMPUTE ? shift[g * 8 + e] : 0.f;
  }
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  for (long long v = (long long)blockIdx.x * nthr + threadIdx.x; v < nvec; v += (long long)gridDim.x * nthr) {
    float xf[8], dz[8];
    unpack8(reinterpret_cast<const uint4*>(x)[v], xf);
    masked_dz<MODE>(reinterpret_cast<const uint4*>(dy), reinterpret_cast<const uint4*>(ymask), xf, sc, sh, v, dz);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s1[e] += dz[e];
      s2[e] += dz[e] * (xf[e] - mu[e]) * is[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[threadIdx.x][e] = s1[e]; red[threadIdx.x][8 + e] = s2[e]; }
  __syncthreads();
  for (int i = threadIdx.x; i < cg * 16; i += nthr) {
    const int gg = i >> 4, e = i & 15;
    float a = 0.f;
    for (int t = gg; t < nthr; t += cg) a += red[t][e];
    // nslices > 0: the workgroups add into a few zeroed slice rows (one fp32 atomic per workgroup and sum) -- the
    // consumer then finalizes from them directly and the pre-reduction launch between the two passes disappears
    if (nslices > 0) atomicAdd(&partials[((size_t)(blockIdx.x % nslices) * 2 + (e >> 3)) * C + gg * 8 + (e & 7)], a);
    else partials[((size_t)blockIdx.x * 2 + (e >> 3)) * C + gg * 8 + (e & 7)] = a;
  }
}

// partials -> dgamma, dbeta, coef[3][C] with dx = coef0*dz + coef1*x + coef2
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(
    const float* __restrict__ partials, int nparts, float count, const float* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ invstd, float* __restrict__ dgamma,
    float* __restrict__ dbeta, int accumulate, float* __restrict__ coef, int C) {
  __shared__ double red[2][32][33];
  const int PL = blockDim.x >> 5;                    // partial-row lanes (8 at the 256-thread launch, see above)
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), pl = threadIdx.x >> 5;
  double s1 = 0.0, s2 = 0.0;
  if (c < C)
    for (int i0 = pl; i0 < nparts; i0 += PL * 8) {   // 16 independent loads in flight (the chain is latency-bound)
      float a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + PL * u;
        a[u] = i < nparts ? partials[((size_t)i * 2) * C + c] : 0.f;
        b[u] = i < nparts ? partials[((size_t)i * 2 + 1) * C + c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s1 += (double)a[u]; s2 += (double)b[u]; }
    }
  red[0][pl][threadIdx.x & 31] = s1;
  red[1][pl][threadIdx.x & 31] = s2;
  __syncthreads();
  if (pl == 0 && c < C) {
    for (int i = 1; i < PL; ++i) { s1 += red[0][i][threadIdx.x]; s2 += red[1][i][threadIdx.x]; }
    const float sum_dz = (float)s1, sum_dzx = (float)s2;
    const float g = gamma ? gamma[c] : 1.f, is = invstd[c], mu = mean[c];
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + sum_dzx : sum_dzx;
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + sum_dz : sum_dz;
    const float k1 = g * is;
    const float k2 = -g * is * is * sum_dzx / count;
    const float k3 = -g * is * sum_dz / count - k2 * mu;
    coef[c] = k1;
    coef[C + c] = k2;
    coef[2 * C + c] = k3;
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const bf16_t* __restrict__ dy, const bf16_t* __restrict__ ymask, const bf16_t* __restrict__ x,
    const float* __restrict__ coef, const float* __restrict__ scale, const float* __restrict__ shift,
    bf16_t* __restrict__ dx, bf16_t* __restrict__ dz_out, long long nvec, int C, const float* __restrict__ slices,
    int nsl, const BnFinalizeArgs fin) {
  const int cg = C >> 3;
  const int g = threadIdx.x % cg;   // blockDim is a multiple of cg
  float k1[8], k2[8], k3[8], sc[8], sh[8];
  __shared__ float fl[3][FIN_MAX_C];
  if (slices) bn_block_finalize<1>(slices, nsl, C, fin, fl);   // backward sums not finalized yet: do it here
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    k1[e] = slices ? fl[0][g * 8 + e] : coef[g * 8 + e];
    k2[e] = slices ? fl[1][g * 8 + e] : coef[C + g * 8 + e];
    k3[e] = slices ? fl[2][g * 8 + e] : coef[2 * C + g * 8 + e];
    sc[e] = MODE >= MASK_RECOMPUTE ? scale[g * 8 + e] : 0.f;
    sh[e] = MODE >= MASK_RECOMPUTE ? shift[g * 8 + e] : 0.f;
  }
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec;
       v += (long long)gridDim.x * blockDim.x) {
    float xf[8], dz[8], o[8];
    unpack8(reinterpret_cast<const uint4*>(x)[v], xf);
    masked_dz<MODE>(reinterpret_cast<const uint4*>(dy), reinterpret_cast<const uint4*>(ymask), xf, sc, sh, v, dz);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = fmaf(k1[e], dz[e], fmaf(k2[e], xf[e], k3[e]));
    reinterpret_cast<uint4*>(dx)[v] = pack8(o);
    if (dz_out) reinterpret_cast<uint4*>(dz_out)[v] = pack8(dz);
  }
}

static inline int ew_grid(long long nvec, int block) {
  long long g = (nvec + block - 1) / block;
  return (int)(g < 2048 ? (g < 1 ? 1 : g) : 2048);
}
static inline int cg_block(int cg) { return (256 / cg) * cg; }

extern "C" {

int mpr_bn_reduce_rows(long long rows, int C) {   // rows of the partial buffers used by bn_stats / bn_bwd_reduce
  const long long nvec = rows * C / 8;
  return ew_grid(nvec, cg_block(C / 8));
}

int mpr_bn_stats(const void* x, float* partials, long long rows, int C, void* stream) {
  MPR_REQUIRE(C % 8 == 0 && C / 8 <= 256, "mpr_bn_stats: C must be a multiple of 8, <= 2048 (got %d)", C);
  const long long nvec = rows * C / 8;
  const int block = cg_block(C / 8);
  bn_stats_kernel<<<ew_grid(nvec, block), block, 0, (hipStream_t)stream>>>((const bf16_t*)x, partials, nvec, C);
  MPR_LAUNCH_CHECK("bn_stats_kernel");
  return MPR_OK;
}

// out [nsplit][2][C] <- partials [nparts][2][C]   (use when nparts is in the thousands)
int mpr_bn_reduce_partials(const float* partials, int nparts, float* out, int nsplit, int C, void* stream) {
  MPR_REQUIRE(partials && out && nsplit > 0 && nsplit <= nparts, "mpr_bn_reduce_partials: bad arguments");
  bn_reduce_partials_kernel<<<dim3(ceil_div(2 * C, 32), nsplit), 256, 0, (hipStream_t)stream>>>(partials, nparts, out,
                                                                                               nsplit, 2 * C);
  MPR_LAUNCH_CHECK("bn_reduce_partials_kernel");
  return MPR_OK;
}

int mpr_bn_finalize_stats(const float* partials, int nparts, long long count, const float* gamma,
                          const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                          float* scale, float* shift, float* mean, float* invstd, int C, void* stream) {
  MPR_REQUIRE(partials && scale && shift && mean && invstd, "mpr_bn_finalize_stats: null pointer");
  bn_finalize_stats_kernel<<<ceil_div(C, 32), 256, 0, (hipStream_t)stream>>>(
      partials, nparts, (float)count, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, mean,
      invstd, C);
  MPR_LAUNCH_CHECK("bn_finalize_stats_kernel");
  return MPR_OK;
}

// mpr_bn_reduce_partials + mpr_bn_finalize_stats in ONE launch (the last workgroup finalizes).
// slices: scratch [nsplit][2][C] floats; ticket: one int, zero on entry (the kernel leaves it zero again).
int mpr_bn_reduce_finalize_stats(const float* partials, int nparts, float* slices, int nsplit, int* ticket,
                                 long long count, const float* gamma, const float* beta, float* running_mean,
                                 float* running_var, float momentum, float eps, float* scale, float* shift, float* mean,
                                 float* invstd, int C, void* stream) {
  MPR_REQUIRE(partials && slices && ticket && scale && shift && mean && invstd && nsplit > 0 && nsplit <= nparts,
              "mpr_bn_reduce_finalize_stats: bad arguments");
  BnFinalizeArgs a = {};
  a.count = (float)count; a.momentum = momentum; a.eps = eps;
  a.gamma = gamma; a.beta = beta; a.running_mean = running_mean; a.running_var = running_var;
  a.scale = scale; a.shift = shift; a.mean_out = mean; a.invstd_out = invstd;
  bn_reduce_finalize_kernel<0><<<dim3(ceil_div(2 * C, 32), nsplit), 256, 0, (hipStream_t)stream>>>(
      partials, nparts, slices, nsplit, C, ticket, a);
  MPR_LAUNCH_CHECK("bn_reduce_finalize_kernel<0>");
  return MPR_OK;
}

// mpr_bn_reduce_partials + mpr_bn_bwd_finalize in ONE launch.
int mpr_bn_reduce_bwd_finalize(const float* partials, int nparts, float* slices, int nsplit, int* ticket,
                               long long count, const float* gamma, const float* mean, const float* invstd,
                               float* dgamma, float* dbeta, int accumulate, float* coef, int C, void* stream) {
  MPR_REQUIRE(partials && slices && ticket && mean && invstd && coef && nsplit > 0 && nsplit <= nparts,
              "mpr_bn_reduce_bwd_finalize: bad arguments");
  BnFinalizeArgs a = {};
  a.count = (float)count;
  a.gamma = gamma; a.mean_in = mean; a.invstd_in = invstd;
  a.dgamma = dgamma; a.dbeta = dbeta; a.coef = coef; a.accumulate = accumulate;
  bn_reduce_finalize_kernel<1><<<dim3(ceil_div(2 * C, 32), nsplit), 256, 0, (hipStream_t)stream>>>(
      partials, nparts, slices, nsplit, C, ticket, a);
  MPR_LAUNCH_CHECK("bn_reduce_finalize_kernel<1>");
  return MPR_OK;
}

int mpr_bn_eval_coefs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                      float eps, float* scale, float* shift, int C, void* stream) {
  bn_eval_coefs_kernel<<<ceil_div(C, 256), 256, 0, (hipStream_t)stream>>>(gamma, beta, running_mean, running_var,
                                                                          eps, scale, shift, C);
  MPR_LAUNCH_CHECK("bn_eval_coefs_kernel");
  return MPR_OK;
}

// invstd = 1 / sqrt(running_var + eps): what an eval-mode BatchNorm's BACKWARD needs beside running_mean
int mpr_bn_eval_invstd(const float* running_var, float eps, float* invstd, int C, void* stream) {
  MPR_REQUIRE(running_var && invstd, "mpr_bn_eval_invstd: null pointer");
  bn_eval_invstd_kernel<<<ceil_div(C, 256), 256, 0, (hipStream_t)stream>>>(running_var, eps, invstd, C);
  MPR_LAUNCH_CHECK("bn_eval_invstd_kernel");
  return MPR_OK;
}

static int launch_bn_apply(const bf16_t* xp, const float* scale, const float* shift, const bf16_t* rp, int relu,
                           bf16_t* yp, long long nvec, int C, int grid, int BLK, const float* slices, int nsl,
                           const BnFinalizeArgs& a, hipStream_t st) {
  if (relu == 2 && rp) bn_apply_kernel<2, true><<<grid, BLK, 0, st>>>(xp, scale, shift, rp, yp, nvec, C / 8, slices, nsl, a);
  else if (relu == 2) bn_apply_kernel<2, false><<<grid, BLK, 0, st>>>(xp, scale, shift, rp, yp, nvec, C / 8, slices, nsl, a);
  else if (relu && rp) bn_apply_kernel<1, true><<<grid, BLK, 0, st>>>(xp, scale, shift, rp, yp, nvec, C / 8, slices, nsl, a);
  else if (relu) bn_apply_kernel<1, false><<<grid, BLK, 0, st>>>(xp, scale, shift, rp, yp, nvec, C / 8, slices, nsl, a);
  else if (rp) bn_apply_kernel<0, true><<<grid, BLK, 0, st>>>(xp, scale, shift, rp, yp, nvec, C / 8, slices, nsl, a);
  else bn_apply_kernel<0, false><<<grid, BLK, 0, st>>>(xp, scale, shift, rp, yp, nvec, C / 8, slices, nsl, a);
  MPR_LAUNCH_CHECK("bn_apply_kernel");
  return MPR_OK;
}

int mpr_bn_apply(const void* x, const float* scale, const float* shift, const void* residual, int relu, void* y,
                 long long rows, int C, void* stream) {
  MPR_REQUIRE(C % 8 == 0, "mpr_bn_apply: C must be a multiple of 8 (got %d)", C);
  MPR_REQUIRE(C / 8 <= 256, "mpr_bn_apply: C must be <= 2048");
  const long long nvec = rows * C / 8;
  const int BLK = cg_block(C / 8), grid = ew_grid(nvec, BLK);
  hipStream_t st = (hipStream_t)stream;
  const bf16_t *xp = (const bf16_t*)x, *rp = (const bf16_t*)residual;
  bf16_t* yp = (bf16_t*)y;
  return launch_bn_apply(xp, scale, shift, rp, relu, yp, nvec, C, grid, BLK, nullptr, 0, BnFinalizeArgs{}, st);
}

// mpr_bn_finalize_stats folded into mpr_bn_apply: `slices` [nsl][2][C] are the (pre-reduced) partial sums; every
// workgroup derives scale / shift itself, workgroup 0 also writes scale, shift, mean, invstd (for the backward pass)
// and updates the running statistics.  C <= 512.
int mpr_bn_apply_fin(const void* x, const float* slices, int nsl, long long count, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float momentum, float eps, float* scale, float* shift,
                     float* mean, float* invstd, const void* residual, int relu, void* y, long long rows, int C,
                     void* stream) {
  MPR_REQUIRE(C % 8 == 0 && C <= FIN_MAX_C, "mpr_bn_apply_fin: C must be a multiple of 8, <= 512 (got %d)", C);
  MPR_REQUIRE(slices && nsl > 0 && scale && shift && mean && invstd, "mpr_bn_apply_fin: null pointer");
  const long long nvec = rows * C / 8;
  const int BLK = cg_block(C / 8), grid = ew_grid(nvec, BLK);
  BnFinalizeArgs a = {};
  a.count = (float)count; a.momentum = momentum; a.eps = eps;
  a.gamma = gamma; a.beta = beta; a.running_mean = running_mean; a.running_var = running_var;
  a.scale = scale; a.shift = shift; a.mean_out = mean; a.invstd_out = invstd;
  return launch_bn_apply((const bf16_t*)x, scale, shift, (const bf16_t*)residual, relu, (bf16_t*)y, nvec, C, grid, BLK,
                         slices, nsl, a, (hipStream_t)stream);
}

// y = act(BN(x) + bf16(BN_r(xr))) in one pass (bn_apply_dual_kernel).  Each BatchNorm is either finalized (slices == NULL:
// scale / shift are read) or pending as in mpr_bn_apply_fin (slices [nsl][2][C]: every workgroup derives the coefficients,
// workgroup 0 writes scale, shift, mean, invstd and updates the running statistics).  C <= 512.
int mpr_bn_apply_dual(const void* x, const float* slices, int nsl, long long count, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, float momentum, float eps, float* scale, float* shift,
                      float* mean, float* invstd, const void* xr, const float* slices_r, int nsl_r, long long count_r,
                      const float* gamma_r, const float* beta_r, float* running_mean_r, float* running_var_r,
                      float momentum_r, float eps_r, float* scale_r, float* shift_r, float* mean_r, float* invstd_r,
                      int relu, void* y, long long rows, int C, void* stream) {
  MPR_REQUIRE(C % 8 == 0 && C <= FIN_MAX_C, "mpr_bn_apply_dual: C must be a multiple of 8, <= 512 (got %d)", C);
  MPR_REQUIRE(x && xr && y && scale && shift && scale_r && shift_r, "mpr_bn_apply_dual: null pointer");
  MPR_REQUIRE(!slices || (nsl > 0 && mean && invstd), "mpr_bn_apply_dual: pending statistics need nsl > 0, mean, invstd");
  MPR_REQUIRE(!slices_r || (nsl_r > 0 && mean_r && invstd_r),
              "mpr_bn_apply_dual: pending shortcut statistics need nsl > 0, mean, invstd");
  MPR_REQUIRE(relu == 0 || relu == 1, "mpr_bn_apply_dual: relu must be 0 or 1 (got %d)", relu);
  const long long nvec = rows * C / 8;
  const int BLK = cg_block(C / 8), grid = ew_grid(nvec, BLK);
  BnFinalizeArgs a = {}, b = {};
  a.count = (float)count; a.momentum = momentum; a.eps = eps;
  a.gamma = gamma; a.beta = beta; a.running_mean = running_mean; a.running_var = running_var;
  a.scale = scale; a.shift = shift; a.mean_out = mean; a.invstd_out = invstd;
  b.count = (float)count_r; b.momentum = momentum_r; b.eps = eps_r;
  b.gamma = gamma_r; b.beta = beta_r; b.running_mean = running_mean_r; b.running_var = running_var_r;
  b.scale = scale_r; b.shift = shift_r; b.mean_out = mean_r; b.invstd_out = invstd_r;
  hipStream_t st = (hipStream_t)stream;
#define ARGS (const bf16_t*)x, scale, shift, slices, nsl, a, (const bf16_t*)xr, scale_r, shift_r, slices_r, nsl_r, b, (bf16_t*)y, nvec, C / 8
  if (relu) bn_apply_dual_kernel<1><<<grid, BLK, 0, st>>>(ARGS);
  else bn_apply_dual_kernel<0><<<grid, BLK, 0, st>>>(ARGS);
#undef ARGS
  MPR_LAUNCH_CHECK("bn_apply_dual_kernel");
  return MPR_OK;
}

// mask_mode: 0 none, 1 relu mask from y (y > 0), 2 recompute relu mask from x*scale+shift, 3 SiLU derivative at x*scale+shift
int mpr_bn_bwd_reduce(const void* dy, const void* y, const void* x, const float* mean, const float* invstd,
                      const float* scale, const float* shift, int mask_mode, float* partials, long long rows,
                      int C, void* stream) {
  MPR_REQUIRE(C % 8 == 0 && C / 8 <= 256, "mpr_bn_bwd_reduce: C must be a multiple of 8, <= 2048 (got %d)", C);
  MPR_REQUIRE(mask_mode != MASK_Y || y, "mpr_bn_bwd_reduce: mask_mode 1 needs y");
  MPR_REQUIRE(dy && x && mean && invstd, "mpr_bn_bwd_reduce: null pointer (dy, x, mean, invstd)");
  const long long nvec = rows * C / 8;
  const int block = cg_block(C / 8), grid = ew_grid(nvec, block);
  hipStream_t st = (hipStream_t)stream;
#define ARGS (const bf16_t*)dy, (const bf16_t*)y, (const bf16_t*)x, mean, invstd, scale, shift, partials, nvec, C, 0
  if (mask_mode == MASK_NONE) bn_bwd_reduce_kernel<MASK_NONE><<<grid, block, 0, st>>>(ARGS);
  else if (mask_mode == MASK_Y) bn_bwd_reduce_kernel<MASK_Y><<<grid, block, 0, st>>>(ARGS);
  else if (mask_mode == MASK_SILU) bn_bwd_reduce_kernel<MASK_SILU><<<grid, block, 0, st>>>(ARGS);
  else bn_bwd_reduce_kernel<MASK_RECOMPUTE><<<grid, block, 0, st>>>(ARGS);
#undef ARGS
  MPR_LAUNCH_CHECK("bn_bwd_reduce_kernel");
  return MPR_OK;
}

// the same pass with the sums added into `nslices` rows of slices[nslices][2][C] (zeroed here) instead of one row per workgroup
int mpr_bn_bwd_reduce_slices(const void* dy, const void* y, const void* x, const float* mean, const float* invstd,
                             const float* scale, const float* shift, int mask_mode, float* slices, int nslices,
                             int prezeroed, long long rows, int C, void* stream) {
  MPR_REQUIRE(C % 8 == 0 && C / 8 <= 256 && nslices > 0 && slices, "mpr_bn_bwd_reduce_slices: bad arguments (C=%d)", C);
  MPR_REQUIRE(mask_mode != MASK_Y || y, "mpr_bn_bwd_reduce_slices: mask_mode 1 needs y");
  MPR_REQUIRE(dy && x && mean && invstd, "mpr_bn_bwd_reduce_slices: null pointer (dy, x, mean, invstd)");
  const long long nvec = rows * C / 8;
  const int block = cg_block(C / 8), grid = ew_grid(nvec, block);
  hipStream_t st = (hipStream_t)stream;
  if (!prezeroed) MPR_HIP(hipMemsetAsync(slices, 0, sizeof(float) * 2 * (size_t)nslices * C, st));
#define ARGS (const bf16_t*)dy, (const bf16_t*)y, (const bf16_t*)x, mean, invstd, scale, shift, slices, nvec, C, nslices
  if (mask_mode == MASK_NONE) bn_bwd_reduce_kernel<MASK_NONE><<<grid, block, 0, st>>>(ARGS);
  else if (mask_mode == MASK_Y) bn_bwd_reduce_kernel<MASK_Y><<<grid, block, 0, st>>>(ARGS);
  else if (mask_mode == MASK_SILU) bn_bwd_reduce_kernel<MASK_SILU><<<grid, block, 0, st>>>(ARGS);
  else bn_bwd_reduce_kernel<MASK_RECOMPUTE><<<grid, block, 0, st>>>(ARGS);
#undef ARGS
  MPR_LAUNCH_CHECK("bn_bwd_reduce_kernel");
  return MPR_OK;
}

int mpr_bn_bwd_finalize(const float* partials, int nparts, long long count, const float* gamma, const float* mean,
                        const float* invstd, float* dgamma, float* dbeta, int accumulate, float* coef, int C,
                        void* stream) {
  bn_bwd_finalize_kernel<<<ceil_div(C, 32), 256, 0, (hipStream_t)stream>>>(
      partials, nparts, (float)count, gamma, mean, invstd, dgamma, dbeta, accumulate, coef, C);
  MPR_LAUNCH_CHECK("bn_bwd_finalize_kernel");
  return MPR_OK;
}

int mpr_bn_bwd_apply(const void* dy, const void* y, const void* x, const float* coef, const float* scale,
                     const float* shift, int mask_mode, void* dx, void* dz_out, long long rows, int C,
                     void* stream) {
  MPR_REQUIRE(C % 8 == 0, "mpr_bn_bwd_apply: C must be a multiple of 8 (got %d)", C);
  MPR_REQUIRE(C / 8 <= 256, "mpr_bn_bwd_apply: C must be <= 2048");
  const long long nvec = rows * C / 8;
  const int BLK = cg_block(C / 8), grid = ew_grid(nvec, BLK);
  hipStream_t st = (hipStream_t)stream;
#define ARGS (const bf16_t*)dy, (const bf16_t*)y, (const bf16_t*)x, coef, scale, shift, (bf16_t*)dx, (bf16_t*)dz_out, nvec, C, \
             nullptr, 0, BnFinalizeArgs{}
  if (mask_mode == MASK_NONE) bn_bwd_apply_kernel<MASK_NONE><<<grid, BLK, 0, st>>>(ARGS);
  else if (mask_mode == MASK_Y) bn_bwd_apply_kernel<MASK_Y><<<grid, BLK, 0, st>>>(ARGS);
  else if (mask_mode == MASK_SILU) bn_bwd_apply_kernel<MASK_SILU><<<grid, BLK, 0, st>>>(ARGS);
  else bn_bwd_apply_kernel<MASK_RECOMPUTE><<<grid, BLK, 0, st>>>(ARGS);
#undef ARGS
  MPR_LAUNCH_CHECK("bn_bwd_apply_kernel");
  return MPR_OK;
}

// mpr_bn_bwd_finalize folded into mpr_bn_bwd_apply: `slices` [nsl][2][C] are the (pre-reduced) sums of
// mpr_bn_bwd_reduce; every workgroup derives the dx coefficients itself, workgroup 0 writes dgamma / dbeta.  C <= 512.
int mpr_bn_bwd_apply_fin(const void* dy, const void* y, const void* x, const float* slices, int nsl, long long count,
                         const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                         int accumulate, const float* scale, const float* shift, int mask_mode, void* dx, void* dz_out,
                         long long rows, int C, void* stream) {
  MPR_REQUIRE(C % 8 == 0 && C <= FIN_MAX_C, "mpr_bn_bwd_apply_fin: C must be a multiple of 8, <= 512 (got %d)", C);
  MPR_REQUIRE(slices && nsl > 0 && mean && invstd, "mpr_bn_bwd_apply_fin: null pointer");
  const long long nvec = rows * C / 8;
  const int BLK = cg_block(C / 8), grid = ew_grid(nvec, BLK);
  hipStream_t st = (hipStream_t)stream;
  BnFinalizeArgs a = {};
  a.count = (float)count;
  a.gamma = gamma; a.mean_in = mean; a.invstd_in = invstd;
  a.dgamma = dgamma; a.dbeta = dbeta; a.accumulate = accumulate;
#define ARGS (const bf16_t*)dy, (const bf16_t*)y, (const bf16_t*)x, nullptr, scale, shift, (bf16_t*)dx, (bf16_t*)dz_out, nvec, C, \
             slices, nsl, a
  if (mask_mode == MASK_NONE) bn_bwd_apply_kernel<MASK_NONE><<<grid, BLK, 0, st>>>(ARGS);
  else if (mask_mode == MASK_Y) bn_bwd_apply_kernel<MASK_Y><<<grid, BLK, 0, st>>>(ARGS);
  else if (mask_mode == MASK_SILU) bn_bwd_apply_kernel<MASK_SILU><<<grid, BLK, 0, st>>>(ARGS);
  else bn_bwd_apply_kernel<MASK_RECOMPUTE><<<grid, BLK, 0, st>>>(ARGS);
#undef ARGS
  MPR_LAUNCH_CHECK("bn_bwd_apply_kernel");
  return MPR_OK;
}

}  // extern "C"
