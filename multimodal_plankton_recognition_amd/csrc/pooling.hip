// Pooling layers on channels-last bf16 tensors (HBM-bound, 16 B / lane):
//   * fused BatchNorm-apply + ReLU + MaxPool (the stem of both ResNets: timm ResNet-18
//     conv1->bn1->act1->maxpool, and ProfileCNN src/profile_encoder.py:167-170,217-220) -- the
//     post-ReLU full-resolution activation is never written to HBM; the winning tap is kept as 1 byte.
//   * its backward (gather form: every input position sums the windows that selected it)
//   * global average pool (ResNet head) and global max pool (ProfileCNN 'avgpool' is an
//     AdaptiveMaxPool1d, src/profile_encoder.py:177,232) with backward.
// Tie-breaking follows torch's CPU kernel: first maximum in (kh, kw) scan order wins.
#include "common.h"

struct PoolGeom {
  int B, H, W, C, P, Q, RH, RW, SH, SW, PH, PW;
};

__global__ __launch_bounds__(256) void bn_relu_maxpool_fwd_kernel(const bf16_t* __restrict__ x,
                                                                  const float* __restrict__ scale,
                                                                  const float* __restrict__ shift,
                                                                  bf16_t* __restrict__ y,
                                                                  unsigned char* __restrict__ idx, PoolGeom g,
                                                                  int apply_bn) {
  const int cg = g.C >> 3;
  const long long total = (long long)g.B * g.P * g.Q * cg;
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < total;
       v += (long long)gridDim.x * blockDim.x) {
    const int c8 = (int)((unsigned)v % (unsigned)cg);
    unsigned t = (unsigned)v / (unsigned)cg;   // < 2^31 elements (host-checked)
    const int q = (int)(t % g.Q); t /= g.Q;
    const int p = (int)(t % g.P);
    const int b = (int)(t / g.P);
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sc[e] = apply_bn ? scale[c8 * 8 + e] : 1.f;
      sh[e] = apply_bn ? shift[c8 * 8 + e] : 0.f;
    }
    float best[8];
    int bi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = 0; }
    for (int kh = 0; kh < g.RH; ++kh) {
      const int ih = p * g.SH - g.PH + kh;
      if ((unsigned)ih >= (unsigned)g.H) continue;
      for (int kw = 0; kw < g.RW; ++kw) {
        const int iw = q * g.SW - g.PW + kw;
        if ((unsigned)iw >= (unsigned)g.W) continue;
        float f[8];
        unpack8(reinterpret_cast<const uint4*>(x)[(((long long)b * g.H + ih) * g.W + iw) * cg + c8], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float a = f[e];
          if (apply_bn) a = round_bf16(fmaxf(fmaf(a, sc[e], sh[e]), 0.f));
          if (a > best[e]) { best[e] = a; bi[e] = kh * g.RW + kw; }
        }
      }
    }
    reinterpret_cast<uint4*>(y)[v] = pack8(best);
    uint2 pk;
    pk.x = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
    pk.y = bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24);
    reinterpret_cast<uint2*>(idx)[v] = pk;
  }
}

// dA[b,ih,iw,c] = sum over windows (p,q) that contain (ih,iw) and whose winning tap is this position
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const bf16_t* __restrict__ dy,
                                                          const unsigned char* __restrict__ idx,
                                                          bf16_t* __restrict__ dx, PoolGeom g) {
  const int cg = g.C >> 3;
  const long long total = (long long)g.B * g.H * g.W * cg;
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < total;
       v += (long long)gridDim.x * blockDim.x) {
    const int c8 = (int)((unsigned)v % (unsigned)cg);
    unsigned t = (unsigned)v / (unsigned)cg;   // < 2^31 elements (host-checked)
    const int iw = (int)(t % g.W); t /= g.W;
    const int ih = (int)(t % g.H);
    const int b = (int)(t / g.H);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int kh = 0; kh < g.RH; ++kh) {
      const int ph = ih + g.PH - kh;
      if (ph < 0 || ph % g.SH) continue;
      const int p = ph / g.SH;
      if (p >= g.P) continue;
      for (int kw = 0; kw < g.RW; ++kw) {
        const int pw = iw + g.PW - kw;
        if (pw < 0 || pw % g.SW) continue;
        const int q = pw / g.SW;
        if (q >= g.Q) continue;
        const long long o = (((long long)b * g.P + p) * g.Q + q) * cg + c8;
        const uint2 pk = reinterpret_cast<const uint2*>(idx)[o];
        float d[8];
        unpack8(reinterpret_cast<const uint4*>(dy)[o], d);
        const int tap = kh * g.RW + kw;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int w = (e < 4 ? (pk.x >> (8 * e)) : (pk.y >> (8 * (e - 4)))) & 0xff;
          if (w == tap) acc[e] += d[e];
        }
      }
    }
    reinterpret_cast<uint4*>(dx)[v] = pack8(acc);
  }
}

// x [B][L][C] bf16 -> y [B][C] fp32 (mean over L).  Four neighbouring lanes share one 8-channel group and take every fourth
// position (this kernel sits in the forward / backward junction of the step, alone on the chip: with one thread per group,
// 49 dependent-address loads in a row and 128 workgroups, it ran at 1.5 TB/s); partial sums are joined lane 0 <- (0+1)+(2+3).
__global__ __launch_bounds__(256) void global_avgpool_fwd_kernel(const bf16_t* __restrict__ x, float* __restrict__ y,
                                                                 int B, int L, int C) {
  const int cg = C >> 3;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int v = gid >> 2, part = gid & 3;
  const bool live = v < B * cg;                 // (whole quads are live or dead together: the shuffles below stay in-quad)
  const int b = live ? v / cg : 0, c8 = live ? v - b * cg : 0;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (live)
    for (int l = part; l < L; l += 4) {
      float f[8];
      unpack8(reinterpret_cast<const uint4*>(x)[((long long)b * L + l) * cg + c8], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += f[e];
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    acc[e] += __shfl_xor(acc[e], 1);
    acc[e] += __shfl_xor(acc[e], 2);
  }
  if (live && part == 0) {
    const float inv = 1.f / (float)L;
#pragma unroll
    for (int e = 0; e < 8; ++e) y[(long long)b * C + c8 * 8 + e] = acc[e] * inv;
  }
}

__global__ __launch_bounds__(256) void global_avgpool_bwd_kernel(const float* __restrict__ dy, bf16_t* __restrict__ dx,
                                                                 int B, int L, int C) {
  const int cg = C >> 3;
  const long long total = (long long)B * L * cg;
  const float inv = 1.f / (float)L;
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < total;
       v += (long long)gridDim.x * blockDim.x) {
    const int c8 = (int)((unsigned)v % (unsigned)cg);
    const int b = (int)((unsigned)v / (unsigned)(L * cg));
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = dy[(long long)b * C + c8 * 8 + e] * inv;
    reinterpret_cast<uint4*>(dx)[v] = pack8(f);
  }
}

// x [B][L][C] bf16 -> y [B][C] fp32 (max over L, first maximum wins), idx [B][C] int32
__global__ __launch_bounds__(256) void global_maxpool_fwd_kernel(const bf16_t* __restrict__ x, float* __restrict__ y,
                                                                 int* __restrict__ idx, int B, int L, int C) {
  const int cg = C >> 3;
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= B * cg) return;
  const int b = v / cg, c8 = v - b * cg;
  float best[8];
  int bi[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = 0; }
  for (int l = 0; l < L; ++l) {
    float f[8];
    unpack8(reinterpret_cast<const uint4*>(x)[((long long)b * L + l) * cg + c8], f);
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (f[e] > best[e]) { best[e] = f[e]; bi[e] = l; }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    y[(long long)b * C + c8 * 8 + e] = best[e];
    idx[(long long)b * C + c8 * 8 + e] = bi[e];
  }
}

__global__ __launch_bounds__(256) void global_maxpool_bwd_kernel(const float* __restrict__ dy,
                                                                 const int* __restrict__ idx, bf16_t* __restrict__ dx,
                                                                 int B, int L, int C) {
  const int cg = C >> 3;
  const long long total = (long long)B * L * cg;
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < total;
       v += (long long)gridDim.x * blockDim.x) {
    const int c8 = (int)((unsigned)v % (unsigned)cg);
    unsigned t = (unsigned)v / (unsigned)cg;   // < 2^31 elements (host-checked)
    const int l = (int)(t % L);
    const int b = (int)(t / L);
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const long long o = (long long)b * C + c8 * 8 + e;
      f[e] = idx[o] == l ? dy[o] : 0.f;
    }
    reinterpret_cast<uint4*>(dx)[v] = pack8(f);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Stem backward with the max-pool gradient gathered on the fly: dz[b,ih,iw,c] = relu'(x*scale+shift) * (sum of the
// pooled gradients of the windows that selected this position).  Used by both BatchNorm-backward passes, so the
// full-resolution gradient (822 MB at batch 512) is never written and re-read.
__device__ __forceinline__ void pool_grad_gather(const bf16_t* __restrict__ dy, const unsigned char* __restrict__ idx,
                                                 const PoolGeom& g, int b, int ih, int iw, int c8, int cg, float* acc) {
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  // strides are 1 or 2 (host-checked): no integer division.  Candidate windows per axis are listed first
  // (<= 2 per axis for the 3/2/1 pooling of both stems), then only the valid (row, column) pairs are visited.
  // (window <= 2*stride per axis, host-checked => at most two candidates per axis, kept in named registers)
  int q0 = -1, q1 = -1, kw0 = 0, kw1 = 0;
  for (int kw = 0; kw < g.RW; ++kw) {
    int q = iw + g.PW - kw;
    if (q < 0) continue;
    if (g.SW == 2) { if (q & 1) continue; q >>= 1; }
    if (q >= g.Q) continue;
    if (q0 < 0) { q0 = q; kw0 = kw; } else { q1 = q; kw1 = kw; }
  }
  for (int kh = 0; kh < g.RH; ++kh) {
    int p = ih + g.PH - kh;
    if (p < 0) continue;
    if (g.SH == 2) { if (p & 1) continue; p >>= 1; }
    if (p >= g.P) continue;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int q = a ? q1 : q0, kw = a ? kw1 : kw0;
      if (q < 0) continue;
      const unsigned o = ((unsigned)(b * g.P + p) * g.Q + q) * cg + c8;
      const uint2 pk = reinterpret_cast<const uint2*>(idx)[o];
      float d[8];
      unpack8(reinterpret_cast<const uint4*>(dy)[o], d);
      const unsigned tap = kh * g.RW + kw;
      const unsigned tap4 = tap * 0x01010101u;
      const unsigned m0 = pk.x ^ tap4, m1 = pk.y ^ tap4;     // a zero byte marks a lane whose arg-max is this tap
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (((m0 >> (8 * e)) & 0xff) == 0) acc[e] += d[e];
        if (((m1 >> (8 * e)) & 0xff) == 0) acc[4 + e] += d[4 + e];
      }
    }
  }
}

// PASS 0: partial sums (sum dz, sum dz*xhat) -> partials [grid][2][C];  PASS 1: dx = k1*dz + k2*x + k3
template <int PASS>
__global__ __launch_bounds__(256) void pool_bn_bwd_kernel(const bf16_t* __restrict__ dy,
                                                          const unsigned char* __restrict__ idx,
                                                          const bf16_t* __restrict__ x, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, const float* __restrict__ coef,
                                                          float* __restrict__ partials, bf16_t* __restrict__ dx, PoolGeom g,
                                                          FastDiv div_cg, FastDiv div_hw, FastDiv div_w) {
  const int cg = g.C >> 3;
  const int nthr = blockDim.x;                    // a multiple of cg: the channel group is thread-invariant
  const int c8 = threadIdx.x % cg;
  __shared__ float red[PASS == 0 ? 256 : 1][17];
  float sc[8], sh[8], p1[8], p2[8], p3[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = scale[c8 * 8 + e];
    sh[e] = shift[c8 * 8 + e];
    if (PASS == 0) { p1[e] = mean[c8 * 8 + e]; p2[e] = invstd[c8 * 8 + e]; p3[e] = 0.f; }
    else { p1[e] = coef[c8 * 8 + e]; p2[e] = coef[g.C + c8 * 8 + e]; p3[e] = coef[2 * g.C + c8 * 8 + e]; }
  }
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  const unsigned total = (unsigned)g.B * g.H * g.W * cg;
  for (unsigned v = blockIdx.x * nthr + threadIdx.x; v < total; v += gridDim.x * nthr) {
    const unsigned pix = fdiv(v, div_cg);                    // v = pix*cg + c8
    const unsigned b = fdiv(pix, div_hw);
    const unsigned rem = pix - b * (unsigned)(g.H * g.W);
    const int ih = fdiv(rem, div_w);
    const int iw = rem - ih * g.W;
    float xf[8], dz[8];
    unpack8(reinterpret_cast<const uint4*>(x)[v], xf);
    pool_grad_gather(dy, idx, g, b, ih, iw, c8, cg, dz);
#pragma unroll
    for (int e = 0; e < 8; ++e) dz[e] = round_bf16(fmaf(xf[e], sc[e], sh[e])) > 0.f ? dz[e] : 0.f;
    if (PASS == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { s1[e] += dz[e]; s2[e] += dz[e] * (xf[e] - p1[e]) * p2[e]; }
    } else {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = fmaf(p1[e], dz[e], fmaf(p2[e], xf[e], p3[e]));
      reinterpret_cast<uint4*>(dx)[v] = pack8(o);
    }
  }
  if (PASS == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[threadIdx.x][e] = s1[e]; red[threadIdx.x][8 + e] = s2[e]; }
    __syncthreads();
    for (int i = threadIdx.x; i < cg * 16; i += nthr) {
      const int gg = i >> 4, e = i & 15;
      float a = 0.f;
      for (int t = gg; t < nthr; t += cg) a += red[t][e];
      partials[((size_t)blockIdx.x * 2 + (e >> 3)) * g.C + gg * 8 + (e & 7)] = a;
    }
  }
}

// The ResNet stem's geometry (3x3 window, stride 2, pad 1, even H and W) as 2x2 source blocks: the block
// {2i, 2i+1} x {2j, 2j+1} is touched by exactly the four windows (i,j), (i,j+1), (i+1,j), (i+1,j+1), so one thread loads
// those four (arg-max bytes, pooled gradient) pairs ONCE and finishes all four pixels of the block -- 9 compare-selects
// per channel and 96 B of window data per 64 B of result, against up to four window look-ups PER PIXEL (and a
// runtime-geometry loop) in the generic kernel above.  Tap codes are the forward kernel's kh*3 + kw.
template <int PASS>
__global__ __launch_bounds__(256) void pool_bn_bwd_s2_kernel(const bf16_t* __restrict__ dy,
                                                             const unsigned char* __restrict__ idx,
                                                             const bf16_t* __restrict__ x, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, const float* __restrict__ coef,
                                                             float* __restrict__ partials, bf16_t* __restrict__ dx, PoolGeom g,
                                                             FastDiv div_cg, FastDiv div_pq, FastDiv div_q) {
  const int cg = g.C >> 3;
  const int nthr = blockDim.x;                    // a multiple of cg: the channel group is thread-invariant
  const int c8 = threadIdx.x % cg;
  __shared__ float red[PASS == 0 ? 256 : 1][17];
  float sc[8], sh[8], p1[8], p2[8], p3[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = scale[c8 * 8 + e];
    sh[e] = shift[c8 * 8 + e];
    if (PASS == 0) { p1[e] = mean[c8 * 8 + e]; p2[e] = invstd[c8 * 8 + e]; p3[e] = 0.f; }
    else { p1[e] = coef[c8 * 8 + e]; p2[e] = coef[g.C + c8 * 8 + e]; p3[e] = coef[2 * g.C + c8 * 8 + e]; }
  }
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  const unsigned total = (unsigned)g.B * g.P * g.Q * cg;       // 2x2 blocks x channel groups
  for (unsigned v = blockIdx.x * nthr + threadIdx.x; v < total; v += gridDim.x * nthr) {
    const unsigned blk = fdiv(v, div_cg);                      // v = blk*cg + c8
    const unsigned b = fdiv(blk, div_pq);
    const unsigned rem = blk - b * (unsigned)(g.P * g.Q);
    const unsigned i = fdiv(rem, div_q);
    const unsigned j = rem - i * g.Q;
    // the four windows (clamped addresses; a window beyond the edge contributes nothing)
    const bool hasj = j + 1 < (unsigned)g.Q, hasi = i + 1 < (unsigned)g.P;
    const unsigned w00 = ((b * g.P + i) * g.Q + j) * cg + c8;
    const unsigned w01 = hasj ? w00 + cg : w00, w10 = hasi ? w00 + g.Q * cg : w00;
    const unsigned w11 = (hasi && hasj) ? w00 + g.Q * cg + cg : w00;
    const uint2 k00 = reinterpret_cast<const uint2*>(idx)[w00], k01 = reinterpret_cast<const uint2*>(idx)[w01];
    const uint2 k10 = reinterpret_cast<const uint2*>(idx)[w10], k11 = reinterpret_cast<const uint2*>(idx)[w11];
    float d00[8], d01[8], d10[8], d11[8];
    unpack8(reinterpret_cast<const uint4*>(dy)[w00], d00);
    unpack8(reinterpret_cast<const uint4*>(dy)[w01], d01);
    unpack8(reinterpret_cast<const uint4*>(dy)[w10], d10);
    unpack8(reinterpret_cast<const uint4*>(dy)[w11], d11);
    if (!hasj) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { d01[e] = 0.f; d11[e] = 0.f; }
    }
    if (!hasi) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { d10[e] = 0.f; d11[e] = 0.f; }
    }
    // source pixels of the block: vec index of (b, 2i + a, 2j + c)
    const unsigned x00 = ((b * g.H + 2 * i) * g.W + 2 * j) * cg + c8;
    const unsigned xo[4] = {x00, x00 + cg, x00 + g.W * cg, x00 + g.W * cg + cg};
    float xf[4][8];
#pragma unroll
    for (int t = 0; t < 4; ++t) unpack8(reinterpret_cast<const uint4*>(x)[xo[t]], xf[t]);
    float dz[4][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const unsigned sft = 8 * (e & 3);
      const unsigned a00 = ((e < 4 ? k00.x : k00.y) >> sft) & 0xff, a01 = ((e < 4 ? k01.x : k01.y) >> sft) & 0xff;
      const unsigned a10 = ((e < 4 ? k10.x : k10.y) >> sft) & 0xff, a11 = ((e < 4 ? k11.x : k11.y) >> sft) & 0xff;
      dz[0][e] = a00 == 4 ? d00[e] : 0.f;
      dz[1][e] = (a00 == 5 ? d00[e] : 0.f) + (a01 == 3 ? d01[e] : 0.f);
      dz[2][e] = (a00 == 7 ? d00[e] : 0.f) + (a10 == 1 ? d10[e] : 0.f);
      dz[3][e] = ((a00 == 8 ? d00[e] : 0.f) + (a01 == 6 ? d01[e] : 0.f)) + ((a10 == 2 ? d10[e] : 0.f) + (a11 == 0 ? d11[e] : 0.f));
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int e = 0; e < 8; ++e) dz[t][e] = round_bf16(fmaf(xf[t][e], sc[e], sh[e])) > 0.f ? dz[t][e] : 0.f;
      if (PASS == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] += dz[t][e]; s2[e] += dz[t][e] * (xf[t][e] - p1[e]) * p2[e]; }
      } else {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = fmaf(p1[e], dz[t][e], fmaf(p2[e], xf[t][e], p3[e]));
        reinterpret_cast<uint4*>(dx)[xo[t]] = pack8(o);
      }
    }
  }
  if (PASS == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[threadIdx.x][e] = s1[e]; red[threadIdx.x][8 + e] = s2[e]; }
    __syncthreads();
    for (int i = threadIdx.x; i < cg * 16; i += nthr) {
      const int gg = i >> 4, e = i & 15;
      float a = 0.f;
      for (int t = gg; t < nthr; t += cg) a += red[t][e];
      partials[((size_t)blockIdx.x * 2 + (e >> 3)) * g.C + gg * 8 + (e & 7)] = a;
    }
  }
}

static inline int ew_grid(long long n, int block) {
  long long g = (n + block - 1) / block;
  return (int)(g < 4096 ? (g < 1 ? 1 : g) : 4096);
}

extern "C" {

// y[B,P,Q,C] = maxpool(relu(x*scale+shift)) (scale == NULL: plain max pool); idx: 1 byte per output element
int mpr_bn_relu_maxpool_fwd(const void* x, const float* scale, const float* shift, void* y, void* idx, int B,
                            int H, int W, int C, int RH, int RW, int SH, int SW, int PH, int PW, void* stream) {
  MPR_REQUIRE(x && y && idx, "mpr_bn_relu_maxpool_fwd: null pointer");
  MPR_REQUIRE(C % 8 == 0, "mpr_bn_relu_maxpool_fwd: C must be a multiple of 8 (got %d)", C);
  MPR_REQUIRE(RH * RW <= 255 && PH < RH && PW < RW, "mpr_bn_relu_maxpool_fwd: bad window");
  PoolGeom g = {B, H, W, C, (H + 2 * PH - RH) / SH + 1, (W + 2 * PW - RW) / SW + 1, RH, RW, SH, SW, PH, PW};
  MPR_REQUIRE(g.P > 0 && g.Q > 0, "mpr_bn_relu_maxpool_fwd: empty output");
  const long long total = (long long)B * g.P * g.Q * (C / 8);
  bn_relu_maxpool_fwd_kernel<<<ew_grid(total, 256), 256, 0, (hipStream_t)stream>>>(
      (const bf16_t*)x, scale, shift, (bf16_t*)y, (unsigned char*)idx, g, scale != nullptr);
  MPR_LAUNCH_CHECK("bn_relu_maxpool_fwd_kernel");
  return MPR_OK;
}

int mpr_maxpool_bwd(const void* dy, const void* idx, void* dx, int B, int H, int W, int C, int RH, int RW, int SH,
                    int SW, int PH, int PW, void* stream) {
  MPR_REQUIRE(dy && idx && dx, "mpr_maxpool_bwd: null pointer");
  MPR_REQUIRE(C % 8 == 0, "mpr_maxpool_bwd: C must be a multiple of 8 (got %d)", C);
  PoolGeom g = {B, H, W, C, (H + 2 * PH - RH) / SH + 1, (W + 2 * PW - RW) / SW + 1, RH, RW, SH, SW, PH, PW};
  const long long total = (long long)B * H * W * (C / 8);
  maxpool_bwd_kernel<<<ew_grid(total, 256), 256, 0, (hipStream_t)stream>>>((const bf16_t*)dy,
                                                                           (const unsigned char*)idx, (bf16_t*)dx, g);
  MPR_LAUNCH_CHECK("maxpool_bwd_kernel");
  return MPR_OK;
}

static inline int pool_bn_block(int C) { return (256 / (C / 8)) * (C / 8); }

int mpr_pool_bn_bwd_rows(int B, int H, int W, int C) {
  const long long nvec = (long long)B * H * W * (C / 8);
  const int block = pool_bn_block(C);
  long long g = (nvec + block - 1) / block;
  return (int)(g < 2048 ? (g < 1 ? 1 : g) : 2048);
}

// Fused (max-pool backward -> ReLU mask -> BatchNorm backward) for the stem: pass 0 writes partial sums
// [mpr_pool_bn_bwd_rows][2][C]; pass 1 (after mpr_bn_bwd_finalize gave coef) writes dx [B,H,W,C].
int mpr_pool_bn_bwd(int pass, const void* dy_pooled, const void* idx, const void* x, const float* scale,
                    const float* shift, const float* mean, const float* invstd, const float* coef, float* partials,
                    void* dx, int B, int H, int W, int C, int RH, int RW, int SH, int SW, int PH, int PW, void* stream) {
  MPR_REQUIRE(dy_pooled && idx && x && scale && shift, "mpr_pool_bn_bwd: null pointer");
  MPR_REQUIRE(C % 8 == 0 && C / 8 <= 256, "mpr_pool_bn_bwd: C must be a multiple of 8, <= 2048 (got %d)", C);
  MPR_REQUIRE((long long)B * H * W * C < (1ll << 31), "mpr_pool_bn_bwd: tensor exceeds 2^31 elements");
  MPR_REQUIRE(pass == 0 ? (mean && invstd && partials) : (coef && dx), "mpr_pool_bn_bwd: missing pass operands");
  PoolGeom g = {B, H, W, C, (H + 2 * PH - RH) / SH + 1, (W + 2 * PW - RW) / SW + 1, RH, RW, SH, SW, PH, PW};
  const int block = pool_bn_block(C), grid = mpr_pool_bn_bwd_rows(B, H, W, C);
  hipStream_t st = (hipStream_t)stream;
  MPR_REQUIRE((SH == 1 || SH == 2) && (SW == 1 || SW == 2), "mpr_pool_bn_bwd: strides must be 1 or 2");
  MPR_REQUIRE(RW <= 2 * SW, "mpr_pool_bn_bwd: window width %d > 2 * stride %d is not supported", RW, SW);
  const FastDiv dcg = make_fastdiv(C / 8), dhw = make_fastdiv(H * W), dw = make_fastdiv(W);
  if (RH == 3 && RW == 3 && SH == 2 && SW == 2 && PH == 1 && PW == 1 && H % 2 == 0 && W % 2 == 0) {
    // the image stem: 2x2 source blocks, every window loaded once
    const FastDiv dpq = make_fastdiv(g.P * g.Q), dq = make_fastdiv(g.Q);
    if (pass == 0)
      pool_bn_bwd_s2_kernel<0><<<grid, block, 0, st>>>((const bf16_t*)dy_pooled, (const unsigned char*)idx,
                                                       (const bf16_t*)x, scale, shift, mean, invstd, nullptr, partials,
                                                       nullptr, g, dcg, dpq, dq);
    else
      pool_bn_bwd_s2_kernel<1><<<grid, block, 0, st>>>((const bf16_t*)dy_pooled, (const unsigned char*)idx,
                                                       (const bf16_t*)x, scale, shift, nullptr, nullptr, coef, nullptr,
                                                       (bf16_t*)dx, g, dcg, dpq, dq);
    MPR_LAUNCH_CHECK("pool_bn_bwd_s2_kernel");
    return MPR_OK;
  }
  if (pass == 0)
    pool_bn_bwd_kernel<0><<<grid, block, 0, st>>>((const bf16_t*)dy_pooled, (const unsigned char*)idx, (const bf16_t*)x,
                                                  scale, shift, mean, invstd, nullptr, partials, nullptr, g, dcg, dhw, dw);
  else
    pool_bn_bwd_kernel<1><<<grid, block, 0, st>>>((const bf16_t*)dy_pooled, (const unsigned char*)idx, (const bf16_t*)x,
                                                  scale, shift, nullptr, nullptr, coef, nullptr, (bf16_t*)dx, g, dcg, dhw, dw);
  MPR_LAUNCH_CHECK("pool_bn_bwd_kernel");
  return MPR_OK;
}

int mpr_global_avgpool_fwd(const void* x, float* y, int B, int L, int C, void* stream) {
  MPR_REQUIRE(C % 8 == 0, "mpr_global_avgpool_fwd: C must be a multiple of 8");
  global_avgpool_fwd_kernel<<<ceil_div(4 * B * (C / 8), 256), 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, y, B, L, C);
  MPR_LAUNCH_CHECK("global_avgpool_fwd_kernel");
  return MPR_OK;
}

int mpr_global_avgpool_bwd(const float* dy, void* dx, int B, int L, int C, void* stream) {
  MPR_REQUIRE(C % 8 == 0, "mpr_global_avgpool_bwd: C must be a multiple of 8");
  global_avgpool_bwd_kernel<<<ew_grid((long long)B * L * (C / 8), 256), 256, 0, (hipStream_t)stream>>>(
      dy, (bf16_t*)dx, B, L, C);
  MPR_LAUNCH_CHECK("global_avgpool_bwd_kernel");
  return MPR_OK;
}

int mpr_global_maxpool_fwd(const void* x, float* y, int* idx, int B, int L, int C, void* stream) {
  MPR_REQUIRE(C % 8 == 0, "mpr_global_maxpool_fwd: C must be a multiple of 8");
  global_maxpool_fwd_kernel<<<ceil_div(B * (C / 8), 256), 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, y, idx, B,
                                                                                        L, C);
  MPR_LAUNCH_CHECK("global_maxpool_fwd_kernel");
  return MPR_OK;
}

int mpr_global_maxpool_bwd(const float* dy, const int* idx, void* dx, int B, int L, int C, void* stream) {
  MPR_REQUIRE(C % 8 == 0, "mpr_global_maxpool_bwd: C must be a multiple of 8");
  global_maxpool_bwd_kernel<<<ew_grid((long long)B * L * (C / 8), 256), 256, 0, (hipStream_t)stream>>>(
      dy, idx, (bf16_t*)dx, B, L, C);
  MPR_LAUNCH_CHECK("global_maxpool_bwd_kernel");
  return MPR_OK;
}

}  // extern "C"
