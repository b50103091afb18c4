// Stem convolutions with a tiny input-channel count (ResNet-18 conv1: 1 -> 64, 7x7/2;
// ProfileCNN conv1: 6 -> base, k3/2, src/profile_encoder.py:167).  K = Cin*R*S is 49 / 18 taps,
// so these layers are bound by the bytes of their OUTPUT (822 MB of bf16 at batch 512), not by math:
// a direct fp32 convolution from the fp32 input, 8 output channels per lane so that each pixel's
// channel vector leaves as 16-B stores, with the BatchNorm partial sums fused in.  No dgrad is
// needed (the input is data).  The weight gradient is a long skinny reduction over pixels.
#include "common.h"

#define STEM_MAX_TAPS 64

struct StemGeom {
  int B, H, W, Cin, K, R, S, sh, sw, ph, pw, P, Q, taps;
};

// x [B,H,W,Cin] fp32 (Cin == 1: identical to NCHW), w [K][Cin][R][S] fp32 (torch OIHW)
// y [B,P,Q,K] bf16, partials [grid][2][K]
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       bf16_t* __restrict__ y, float* __restrict__ partials,
                                                       StemGeom g) {
  extern __shared__ __attribute__((aligned(16))) float wl[];   // [taps][K]  (tap = (c, r, s))
  __shared__ float red[256][17];
  const int cgn = g.K >> 3;
  const int nthr = blockDim.x;
  for (int i = threadIdx.x; i < g.taps * g.K; i += nthr) {
    const int t = i / g.K, k = i - t * g.K;
    wl[i] = w[(size_t)k * g.taps + t];
  }
  __syncthreads();
  const int cg = threadIdx.x % cgn;
  const int ppb = nthr / cgn;                       // pixels per block pass
  const int pl = threadIdx.x / cgn;
  const int npix = g.B * g.P * g.Q;
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  for (int pix = blockIdx.x * ppb + pl; pix < npix; pix += gridDim.x * ppb) {
    const int q = pix % g.Q;
    int t = pix / g.Q;
    const int p = t % g.P;
    const int b = t / g.P;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int r = 0; r < g.R; ++r) {
      const int ih = p * g.sh - g.ph + r;
      if ((unsigned)ih >= (unsigned)g.H) continue;
      for (int s = 0; s < g.S; ++s) {
        const int iw = q * g.sw - g.pw + s;
        if ((unsigned)iw >= (unsigned)g.W) continue;
        const float* xp = x + (((size_t)b * g.H + ih) * g.W + iw) * g.Cin;
        for (int c = 0; c < g.Cin; ++c) {
          const float xv = xp[c];
          const float* wp = wl + ((c * g.R + r) * g.S + s) * g.K + cg * 8;
          const float4 w0 = *reinterpret_cast<const float4*>(wp);
          const float4 w1 = *reinterpret_cast<const float4*>(wp + 4);
          acc[0] = fmaf(xv, w0.x, acc[0]); acc[1] = fmaf(xv, w0.y, acc[1]);
          acc[2] = fmaf(xv, w0.z, acc[2]); acc[3] = fmaf(xv, w0.w, acc[3]);
          acc[4] = fmaf(xv, w1.x, acc[4]); acc[5] = fmaf(xv, w1.y, acc[5]);
          acc[6] = fmaf(xv, w1.z, acc[6]); acc[7] = fmaf(xv, w1.w, acc[7]);
        }
      }
    }
    const uint4 pk = pack8(acc);
    reinterpret_cast<uint4*>(y)[(size_t)pix * cgn + cg] = pk;
    if (partials) {
      float f[8];
      unpack8(pk, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) { s1[e] += f[e]; s2[e] += f[e] * f[e]; }
    }
  }
  if (partials) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[threadIdx.x][e] = s1[e]; red[threadIdx.x][8 + e] = s2[e]; }
    __syncthreads();
    for (int i = threadIdx.x; i < cgn * 16; i += nthr) {
      const int gg = i >> 4, e = i & 15;
      float a = 0.f;
      for (int t = gg; t < nthr; t += cgn) a += red[t][e];
      partials[((size_t)blockIdx.x * 2 + (e >> 3)) * g.K + gg * 8 + (e & 7)] = a;
    }
  }
}

// dw[k][tap] += sum_pix dy[pix][k] * x[pix @ tap]   (dw zeroed by the host wrapper; fp32 atomics, one per
// (block, element)).  Thread = (8 output channels) x (a strided subset of the taps).
template <int MAXT>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ x, const bf16_t* __restrict__ dy,
                                                         float* __restrict__ dw, StemGeom g, int pix_per_block, int dense_ok) {
  const int cgn = g.K >> 3;
  const int nthr = blockDim.x;
  const int cg = threadIdx.x % cgn;
  const int tl = threadIdx.x / cgn, ntl = nthr / cgn;      // tap lane
  // MAXT = taps per thread (host guarantees taps <= ntl*MAXT)
  int tc[MAXT], trr[MAXT], tss[MAXT];
  bool tv[MAXT];
#pragma unroll
  for (int j = 0; j < MAXT; ++j) {
    const int t = tl + j * ntl;
    tv[j] = t < g.taps;
    const int tt = tv[j] ? t : 0;
    tc[j] = tt / (g.R * g.S);
    const int rs = tt - tc[j] * g.R * g.S;
    trr[j] = rs / g.S;
    tss[j] = rs - trr[j] * g.S;
  }
  float acc[MAXT][8];
#pragma unroll
  for (int j = 0; j < MAXT; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[j][e] = 0.f;
  const int npix = g.B * g.P * g.Q;
  const int p0 = blockIdx.x * pix_per_block;
  int p1 = p0 + pix_per_block;
  if (p1 > npix) p1 = npix;
  int q = p0 % g.Q, t0 = p0 / g.Q;
  int p = t0 % g.P, b = t0 / g.P;
  // four pixels per iteration, their loads first (round 3: one pixel per iteration was a serial chain of two loads and eight
  // multiply-adds, 640 ns per pixel -- 1 ms for EfficientNet's 3x3 stem at batch 256)
  constexpr int U = 4;
  for (int pix = p0; pix < p1; pix += U) {
    uint4 dv[U];
    float xv[U][MAXT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool in = pix + u < p1;
      dv[u] = in ? reinterpret_cast<const uint4*>(dy)[(size_t)(pix + u) * cgn + cg] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
      for (int j = 0; j < MAXT; ++j) {
        const int ih = p * g.sh - g.ph + trr[j], iw = q * g.sw - g.pw + tss[j];
        const bool ok = in && tv[j] && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
        xv[u][j] = ok ? x[(((size_t)b * g.H + ih) * g.W + iw) * g.Cin + tc[j]] : 0.f;
      }
      if (++q == g.Q) { q = 0; if (++p == g.P) { p = 0; ++b; } }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float d[8];
      unpack8(dv[u], d);
#pragma unroll
      for (int j = 0; j < MAXT; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[j][e] = fmaf(xv[u][j], d[e], acc[j][e]);
    }
  }
  // The block's K x taps sums leave through LDS as ONE dense run of atomics (consecutive lanes -> consecutive addresses: a wave
  // instruction is two cache-line operations at the memory side).  Straight from the registers a lane's address is
  // (8 cg + e) * taps + t -- every lane another cache line, K x taps line operations per block: that, not the arithmetic, was
  // this kernel's time (round 3).
  extern __shared__ float dense[];      // [K * taps] (host: <= 48 KB, else the scattered form)
  if (dense_ok) {
#pragma unroll
    for (int j = 0; j < MAXT; ++j)
      if (tv[j]) {
        const int t = tl + j * ntl;
#pragma unroll
        for (int e = 0; e < 8; ++e) dense[(cg * 8 + e) * g.taps + t] = acc[j][e];
      }
    __syncthreads();
    for (int i = threadIdx.x; i < g.K * g.taps; i += nthr) atomicAdd(dw + i, dense[i]);
    return;
  }
#pragma unroll
  for (int j = 0; j < MAXT; ++j)
    if (tv[j]) {
      const int t = tl + j * ntl;
#pragma unroll
      for (int e = 0; e < 8; ++e) atomicAdd(dw + (size_t)(cg * 8 + e) * g.taps + t, acc[j][e]);
    }
}

// One input channel, 3 x 3 filter (EfficientNet's stem on grayscale plankton images; round 3): the generic kernel above gives a
// thread ONE tap -- nine of its 64 tap lanes work, every lane re-loads the same gradient group -- and took 1 ms at batch 256.
// Here a thread owns 8 output channels x ALL nine taps (72 accumulators) and walks pixels: lanes of a wave = cgn channel groups x
// 64 / cgn consecutive pixels (the gradient rows of a wave are one contiguous run, the nine x values of a pixel three short
// runs); lanes of equal channel group meet by wave shuffles, the block's waves in LDS, blocks by fp32 atomics.
// (Workgroups of 16 waves and at most one per CU: every workgroup ends in K x 9 fp32 atomics on the SAME addresses, and those
//  serialise at the memory side at ~0.4 us each -- 2048 workgroups of four waves spent 800 of their 840 us there and starved the
//  profile branch's kernels on the other stream meanwhile.)
__global__ __launch_bounds__(1024) void stem_wgrad_c1k3_kernel(const float* __restrict__ x, const bf16_t* __restrict__ dy,
                                                               float* __restrict__ dw, StemGeom g, int pix_per_block) {
  __shared__ float red[16][64][9];
  const int cgn = g.K >> 3;                    // power of two, <= 64 (host)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwv = blockDim.x >> 6;
  const int cg = lane % cgn, pl = tid / cgn, npl = (int)blockDim.x / cgn;
  float acc[9][8];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[t][e] = 0.f;
  const int npix = g.B * g.P * g.Q;
  const int p0 = blockIdx.x * pix_per_block;
  int p1 = p0 + pix_per_block;
  if (p1 > npix) p1 = npix;
  for (int pix = p0 + pl; pix < p1; pix += npl) {
    const int q = pix % g.Q, t0 = pix / g.Q;
    const int p = t0 % g.P, b = t0 / g.P;
    const uint4 dv = reinterpret_cast<const uint4*>(dy)[(size_t)pix * cgn + cg];
    float xv[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s2 = 0; s2 < 3; ++s2) {
        const int ih = p * g.sh - g.ph + r, iw = q * g.sw - g.pw + s2;
        const bool ok = (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
        xv[r * 3 + s2] = ok ? x[((size_t)b * g.H + ih) * g.W + iw] : 0.f;
      }
    float d[8];
    unpack8(dv, d);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[t][e] = fmaf(xv[t], d[e], acc[t][e]);
  }
  // lanes of equal channel group within the wave
  for (int off = cgn; off < 64; off <<= 1)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[t][e] += __shfl_xor(acc[t][e], off);
  // waves -> LDS [wave][K * 9] in dw's own order, then ONE dense run of atomics (consecutive lanes -> consecutive addresses)
  float* const dense = &red[0][0][0];      // 16 x 576 floats: K <= 64 (host)
  __syncthreads();
  if (lane < cgn) {
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int t = 0; t < 9; ++t) dense[wave * 576 + (cg * 8 + e) * 9 + t] = acc[t][e];
  }
  __syncthreads();
  for (int i = tid; i < g.K * 9; i += (int)blockDim.x) {
    float a = 0.f;
    for (int w2 = 0; w2 < nwv; ++w2) a += dense[w2 * 576 + i];
    atomicAdd(dw + i, a);
  }
}

// ---------------------------------------------------------------------------------------------------------
// ResNet stem (1 input channel, 7x7, stride 2, pad 3) as a space-to-depth problem: the stride-2 7x7 filter on
// 1 channel equals a stride-1 4x4 filter on the 4 phase channels of the 2x2-decimated image (filter padded to
// 8x8 with a zero first row/column).  Padding the phases to 8 channels makes it a plain MFMA implicit GEMM
// (K = 4*4*8 = 128) for the generic conv / wgrad kernels, and the halo is materialised as zeros so that the
// conv runs with pad 0:  xs[b][i][j][2*dy+dx] = x[b][2*(i-2)+dy][2*(j-2)+dx],  i in [0, H/2+3).
__global__ __launch_bounds__(256) void stem_s2d_kernel(const float* __restrict__ x, bf16_t* __restrict__ xs, int B,
                                                       int H, int W) {
  const int H2 = H / 2 + 3, W2 = W / 2 + 3;
  const long long total = (long long)B * H2 * W2;
  for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < total; v += (long long)gridDim.x * 256) {
    const unsigned u = (unsigned)v;
    const int j = u % W2;
    const unsigned t = u / W2;
    const int i = t % H2;
    const int b = t / H2;
    float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int r0 = 2 * (i - 2), c0 = 2 * (j - 2);
    if (r0 >= 0 && r0 < H && c0 >= 0 && c0 < W) {
      const float2 a = *reinterpret_cast<const float2*>(x + ((size_t)b * H + r0) * W + c0);
      const float2 c = *reinterpret_cast<const float2*>(x + ((size_t)b * H + r0 + 1) * W + c0);
      f[0] = a.x; f[1] = a.y; f[2] = c.x; f[3] = c.y;
    }
    reinterpret_cast<uint4*>(xs)[v] = pack8(f);
  }
}

// w [K][1][7][7] -> w2 [K][8][4][4]:  w2[k][2*dy+dx][r2][s2] = w[k][2*r2+dy-1][2*s2+dx-1]  (0 outside / ch >= 4)
__global__ void stem_w_s2d_kernel(const float* __restrict__ w, float* __restrict__ w2, int K) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K * 128) return;
  const int s2 = i & 3, r2 = (i >> 2) & 3, c = (i >> 4) & 7, k = i >> 7;
  float v = 0.f;
  if (c < 4) {
    const int r = 2 * r2 + (c >> 1) - 1, s = 2 * s2 + (c & 1) - 1;
    if (r >= 0 && s >= 0) v = w[(k * 7 + r) * 7 + s];
  }
  w2[i] = v;
}

// dw [K][1][7][7] (+)= gather of dw2 [K][8][4][4]
__global__ void stem_dw_gather_kernel(const float* __restrict__ dw2, float* __restrict__ dw, int K, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K * 49) return;
  const int s = i % 7, r = (i / 7) % 7, k = i / 49;
  const int r2 = (r + 1) >> 1, dy = (r + 1) & 1, s2 = (s + 1) >> 1, dx = (s + 1) & 1;
  const float v = dw2[((k * 8 + 2 * dy + dx) * 4 + r2) * 4 + s2];
  dw[i] = accumulate ? dw[i] + v : v;
}

static inline int stem_block(int K) { return (256 / (K / 8)) * (K / 8); }

extern "C" {

// xs [B][H/2+3][W/2+3][8] bf16 <- x [B][H][W] fp32 (1 channel; H, W even)
int mpr_stem_s2d(const float* x, void* xs, int B, int H, int W, void* stream) {
  MPR_REQUIRE(x && xs, "mpr_stem_s2d: null pointer");
  MPR_REQUIRE(H % 2 == 0 && W % 2 == 0 && H > 0 && W > 0, "mpr_stem_s2d: H and W must be even (got %d x %d)", H, W);
  const long long total = (long long)B * (H / 2 + 3) * (W / 2 + 3);
  MPR_REQUIRE(total * 8 < (1ll << 31), "mpr_stem_s2d: tensor exceeds 2^31 elements");
  long long g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  stem_s2d_kernel<<<(unsigned)g, 256, 0, (hipStream_t)stream>>>(x, (bf16_t*)xs, B, H, W);
  MPR_LAUNCH_CHECK("stem_s2d_kernel");
  return MPR_OK;
}

// w2 [K][8][4][4] fp32 <- w [K][1][7][7] fp32
int mpr_stem_w_s2d(const float* w, float* w2, int K, void* stream) {
  MPR_REQUIRE(w && w2 && K > 0, "mpr_stem_w_s2d: bad arguments");
  stem_w_s2d_kernel<<<ceil_div(K * 128, 256), 256, 0, (hipStream_t)stream>>>(w, w2, K);
  MPR_LAUNCH_CHECK("stem_w_s2d_kernel");
  return MPR_OK;
}

// dw [K][1][7][7] fp32 (+)= dw2 [K][8][4][4] fp32
int mpr_stem_dw_gather(const float* dw2, float* dw, int K, int accumulate, void* stream) {
  MPR_REQUIRE(dw2 && dw && K > 0, "mpr_stem_dw_gather: bad arguments");
  stem_dw_gather_kernel<<<ceil_div(K * 49, 256), 256, 0, (hipStream_t)stream>>>(dw2, dw, K, accumulate);
  MPR_LAUNCH_CHECK("stem_dw_gather_kernel");
  return MPR_OK;
}

int mpr_stem_fwd_stat_rows(int B, int P, int Q, int K) {
  const int ppb = stem_block(K) / (K / 8);
  int grid = ceil_div(B * P * Q, ppb * 8);          // >= 8 pixels per thread
  if (grid > 2048) grid = 2048;
  if (grid < 1) grid = 1;
  return grid;
}

int mpr_stem_fwd(const float* x, const float* w, void* y, float* partials, int B, int H, int W, int Cin, int K, int R,
                 int S, int sh, int sw, int ph, int pw, void* stream) {
  MPR_REQUIRE(x && w && y, "mpr_stem_fwd: null pointer");
  MPR_REQUIRE(K % 8 == 0 && K / 8 <= 256, "mpr_stem_fwd: K must be a multiple of 8, <= 2048 (got %d)", K);
  StemGeom g = {B, H, W, Cin, K, R, S, sh, sw, ph, pw, (H + 2 * ph - R) / sh + 1, (W + 2 * pw - S) / sw + 1,
                Cin * R * S};
  MPR_REQUIRE(g.P > 0 && g.Q > 0, "mpr_stem_fwd: empty output");
  MPR_REQUIRE((long long)B * g.P * g.Q * K < (1ll << 31), "mpr_stem_fwd: tensor exceeds 2^31 elements");
  const size_t smem = sizeof(float) * g.taps * K;
  MPR_REQUIRE(smem <= 48 * 1024, "mpr_stem_fwd: filter too large for the direct kernel (%zu B)", smem);
  const int grid = mpr_stem_fwd_stat_rows(B, g.P, g.Q, K);
  stem_fwd_kernel<<<grid, stem_block(K), smem, (hipStream_t)stream>>>(x, w, (bf16_t*)y, partials, g);
  MPR_LAUNCH_CHECK("stem_fwd_kernel");
  return MPR_OK;
}

int mpr_stem_wgrad(const float* x, const void* dy, float* dw, int accumulate, int B, int H, int W, int Cin, int K,
                   int R, int S, int sh, int sw, int ph, int pw, void* stream) {
  MPR_REQUIRE(x && dy && dw, "mpr_stem_wgrad: null pointer");
  MPR_REQUIRE(K % 8 == 0 && K / 8 <= 256, "mpr_stem_wgrad: K must be a multiple of 8, <= 2048 (got %d)", K);
  StemGeom g = {B, H, W, Cin, K, R, S, sh, sw, ph, pw, (H + 2 * ph - R) / sh + 1, (W + 2 * pw - S) / sw + 1,
                Cin * R * S};
  const int block = stem_block(K), ntl = block / (K / 8);
  MPR_REQUIRE(g.taps <= ntl * 4, "mpr_stem_wgrad: %d taps do not fit %d tap lanes x 4", g.taps, ntl);
  hipStream_t st = (hipStream_t)stream;
  if (!accumulate) MPR_HIP(hipMemsetAsync(dw, 0, sizeof(float) * (size_t)K * g.taps, st));
  const int npix = B * g.P * g.Q;
  const int cgn = K / 8;
  if (Cin == 1 && R == 3 && S == 3 && cgn <= 8 && (cgn & (cgn - 1)) == 0) {
    int ppb1 = ceil_div(npix, 256);
    if (ppb1 < 1024) ppb1 = 1024;
    stem_wgrad_c1k3_kernel<<<ceil_div(npix, ppb1), 1024, 0, st>>>(x, (const bf16_t*)dy, dw, g, ppb1);
    MPR_LAUNCH_CHECK("stem_wgrad_c1k3_kernel");
    return MPR_OK;
  }
  // (every block ends in K x taps fp32 atomics on the SAME addresses: 2048 blocks of 16 pixels on the profile stem's 32 K
  //  positions spent most of their 161 us queueing there)
  int grid = 8192;
  int ppb = ceil_div(npix, grid);
  if (ppb < 64) ppb = 64;
  grid = ceil_div(npix, ppb);
  const int maxt = ceil_div(g.taps, ntl);
  const size_t dense_b = sizeof(float) * (size_t)K * g.taps;
  const int dense_ok = dense_b <= 48 * 1024;
  const size_t lds = dense_ok ? dense_b : 0;
  if (maxt == 1) stem_wgrad_kernel<1><<<grid, block, lds, st>>>(x, (const bf16_t*)dy, dw, g, ppb, dense_ok);
  else if (maxt == 2) stem_wgrad_kernel<2><<<grid, block, lds, st>>>(x, (const bf16_t*)dy, dw, g, ppb, dense_ok);
  else if (maxt == 3) stem_wgrad_kernel<3><<<grid, block, lds, st>>>(x, (const bf16_t*)dy, dw, g, ppb, dense_ok);
  else stem_wgrad_kernel<4><<<grid, block, lds, st>>>(x, (const bf16_t*)dy, dw, g, ppb, dense_ok);
  MPR_LAUNCH_CHECK("stem_wgrad_kernel");
  return MPR_OK;
}

}  // extern "C"
