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

// x [B][L][C] bf16 -> y [B][C] fp32 (mean over L)
__global__ __launch_bounds__(256) void global_avgpool_fwd_kernel(const bf16_t* __restrict__ x, float* __restrict__ y,
                                                                 int B, int L, int C) {
  const int cg = C >> 3;
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= B * cg) return;
  const int b = v / cg, c8 = v - b * cg;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  for (int l = 0; l < L; ++l) {
    float f[8];
    unpack8(reinterpret_cast<const uint4*>(x)[((long long)b * L + l) * cg + c8], f);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += f[e];
  }
  const float inv = 1.f / (float)L;
#pragma unroll
  for (int e = 0; e < 8; ++e) y[(long long)b * C + c8 * 8 + e] = acc[e] * inv;
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

int mpr_global_avgpool_fwd(const void* x, float* y, int B, int L, int C, void* stream) {
  MPR_REQUIRE(C % 8 == 0, "mpr_global_avgpool_fwd: C must be a multiple of 8");
  global_avgpool_fwd_kernel<<<ceil_div(B * (C / 8), 256), 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, y, B, L, C);
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
