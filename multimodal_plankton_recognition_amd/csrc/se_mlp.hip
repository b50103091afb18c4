// Squeeze-excite bottleneck of EfficientNet (timm SqueezeExcite behind src/image_encoder.py:16,24; the backbone of
// model_cards/example_multi.yaml:9):  gate = sigmoid(W2 silu(W1 pooled + b1) + b2)  on the pooled [B][C] map, fp32.
//
// Round 3: as two fp32 GEMMs + two bias/activation passes forward and six GEMMs + two activation passes backward this was 14
// launches of 10-20 us per block for ~30 MFLOP -- 320 of the 1050 launches of an EfficientNet-B0 step.  Here: one kernel
// forward (a workgroup per image, pooled row and hidden vector in LDS), two backward (per image: the chain back to the pooled
// map + bias gradients; per 64 channels x batch slab: both weight gradients).  rd <= 64 (EfficientNet-B0: 4 .. 48); wider
// bottlenecks stay on the GEMM path (the host decides).
#include "common.h"

#define SE_RD_MAX 64

__device__ __forceinline__ float se_sigmoid(float z) { return 1.f / (1.f + expf(-z)); }

// z1[b][j] = W1[j] . pooled[b] + b1[j];  r = silu(z1);  gate[b][c] = sigmoid(W2[c] . r + b2[c])
// One workgroup per image.  First product: a thread owns channels c, c + 256, ... and ALL rd bottleneck rows at once (rd
// accumulators), so every W1 load is coalesced across the workgroup and rd x C / 256 of them are in flight together; the partial
// sums meet by wave shuffles + LDS.  (A wave per bottleneck row walked its row in 18 dependent-latency steps, twelve rows in
// sequence: 89 us for the 1152-wide blocks; four images per workgroup to share the weight reads made it 116 -- the weights come
// from L2 either way, the parallelism was the point.)  Second product: a thread per channel on its own W2 row.
template <int RDP>
__global__ __launch_bounds__(256) void se_mlp_fwd_kernel(const float* __restrict__ pooled, const float* __restrict__ w1,
                                                         const float* __restrict__ b1, const float* __restrict__ w2,
                                                         const float* __restrict__ b2, float* __restrict__ z1,
                                                         float* __restrict__ r, float* __restrict__ gate, int C, int rd) {
  __shared__ float part[4][RDP];
  __shared__ float sh[RDP];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float acc[RDP];
#pragma unroll
  for (int j = 0; j < RDP; ++j) acc[j] = 0.f;
  for (int c = tid; c < C; c += 256) {
    const float pv = pooled[(size_t)b * C + c];
#pragma unroll
    // (rows past rd: the last row again, times zero -- NO branch per row: behind a branch every load waits for its own use
    //  before the next is issued, 240 serial round trips per thread: 74 us per workgroup for 110 K multiply-adds)
    for (int j = 0; j < RDP; ++j) {
      const float wv = w1[(size_t)(j < rd ? j : rd - 1) * C + c];
      acc[j] = fmaf(j < rd ? wv : 0.f, pv, acc[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < RDP; ++j) {
    float a = acc[j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
    if (lane == 0) part[wave][j] = a;
  }
  __syncthreads();
  if (tid < rd) {
    const float z = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]) + b1[tid];
    const float h = z * se_sigmoid(z);
    z1[(size_t)b * rd + tid] = z;
    r[(size_t)b * rd + tid] = h;
    sh[tid] = h;
  }
  __syncthreads();
  // (W2 through LDS tiles of 256 channels x rd: the global reads are contiguous; a thread walking its own row strides the lanes
  //  by rd floats -- 64 cache lines per load instruction)
  extern __shared__ float sw[];      // [256][rd + 1]
  for (int c0 = 0; c0 < C; c0 += 256) {
    if (c0) __syncthreads();
    const int nc = C - c0 < 256 ? C - c0 : 256;
    for (int i = tid; i < nc * rd; i += 256) sw[(i / rd) * (rd + 1) + i % rd] = w2[(size_t)c0 * rd + i];
    __syncthreads();
    if (tid < nc) {
      const int c = c0 + tid;
      float a = b2[c];
      const float* wr = sw + tid * (rd + 1);
#pragma unroll
      for (int j = 0; j < RDP; ++j) {
        const int jc = j < rd ? j : rd - 1;
        a = fmaf(j < rd ? wr[jc] : 0.f, sh[jc], a);
      }
      gate[(size_t)b * C + c] = se_sigmoid(a);
    }
  }
}


// per image: dz2 = dgate * gate (1 - gate);  dr = W2^T dz2;  dz1 = dr * silu'(z1);  dpooled = scale * W1^T dz1
// (+ the bias gradients db2 += dz2, db1 += dz1 by fp32 atomics: one per image and element)
template <int RDP>
__global__ __launch_bounds__(256) void se_mlp_bwd_kernel(const float* __restrict__ dgate, const float* __restrict__ gate,
                                                         const float* __restrict__ z1, const float* __restrict__ w1,
                                                         const float* __restrict__ w2, float* __restrict__ dz2,
                                                         float* __restrict__ dz1, float* __restrict__ dpooled,
                                                         float* __restrict__ db1, float* __restrict__ db2, int C, int rd,
                                                         float scale) {
  extern __shared__ float sm[];
  float* const sd = sm;                  // dz2 row [C]
  float* const sh = sm + C;              // dz1 [RDP]
  float* const red = sm + C + RDP;       // [256][RDP + 1]
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int c = tid; c < C; c += 256) {
    const float g = gate[(size_t)b * C + c];
    const float d = dgate[(size_t)b * C + c] * g * (1.f - g);
    dz2[(size_t)b * C + c] = d;
    sd[c] = d;
    atomicAdd(db2 + c, d);
  }
  __syncthreads();
  float pr[RDP];
#pragma unroll
  for (int j = 0; j < RDP; ++j) pr[j] = 0.f;
  for (int c = tid; c < C; c += 256) {
    const float d = sd[c];
    const float* wr = w2 + (size_t)c * rd;
#pragma unroll
    for (int j = 0; j < RDP; ++j) {
      const float wv = wr[j < rd ? j : rd - 1];
      pr[j] = fmaf(j < rd ? wv : 0.f, d, pr[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < RDP; ++j) red[tid * (RDP + 1) + j] = pr[j];
  __syncthreads();
  if (tid < rd) {
    float a = 0.f;
    for (int t = 0; t < 256; ++t) a += red[t * (RDP + 1) + tid];
    const float z = z1[(size_t)b * rd + tid];
    const float s = se_sigmoid(z);
    const float d = a * s * (1.f + z * (1.f - s));
    dz1[(size_t)b * rd + tid] = d;
    sh[tid] = d;
    atomicAdd(db1 + tid, d);
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float a = 0.f;
    for (int j = 0; j < rd; ++j) a = fmaf(w1[(size_t)j * C + c], sh[j], a);
    dpooled[(size_t)b * C + c] = a * scale;
  }
}

// dW2[c][j] += sum_b dz2[b][c] r[b][j];  dW1[j][c] += sum_b dz1[b][j] pooled[b][c]
// Block = 64 channels x 4 batch lanes (a wave = one batch lane: r / dz1 of its image are wave-uniform), grid.y batch slabs; the
// lanes meet in LDS sixteen bottleneck columns at a time, the slabs by fp32 atomics.
template <int RDP>
__global__ __launch_bounds__(256) void se_mlp_wgrad_kernel(const float* __restrict__ dz2, const float* __restrict__ dz1,
                                                           const float* __restrict__ r, const float* __restrict__ pooled,
                                                           float* __restrict__ dw1, float* __restrict__ dw2, int B, int C,
                                                           int rd) {
  constexpr int JC = RDP < 16 ? RDP : 16;
  __shared__ float red[4][64][2 * JC + 1];
  const int tid = threadIdx.x, lane = tid & 63, bl = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = blockIdx.x * 64 + lane;
  const bool cok = c < C;
  float a1[RDP], a2[RDP];
#pragma unroll
  for (int j = 0; j < RDP; ++j) { a1[j] = 0.f; a2[j] = 0.f; }
  for (int b = blockIdx.y * 4 + bl; b < B; b += 4 * gridDim.y) {
    const float d2 = cok ? dz2[(size_t)b * C + c] : 0.f;
    const float pl = cok ? pooled[(size_t)b * C + c] : 0.f;
    const float* rr = r + (size_t)b * rd;
    const float* dd = dz1 + (size_t)b * rd;
#pragma unroll
    for (int j = 0; j < RDP; ++j) {
      const int jc = j < rd ? j : rd - 1;
      const float rv = rr[jc], dv = dd[jc];
      a2[j] = fmaf(d2, j < rd ? rv : 0.f, a2[j]);
      a1[j] = fmaf(j < rd ? dv : 0.f, pl, a1[j]);
    }
  }
#pragma unroll
  for (int j0 = 0; j0 < RDP; j0 += JC) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < JC; ++j) {
      red[bl][lane][j] = a1[j0 + j];
      red[bl][lane][JC + j] = a2[j0 + j];
    }
    __syncthreads();
    if (bl == 0) {
      if (cok) {
#pragma unroll
        for (int j = 0; j < JC; ++j)
          if (j0 + j < rd) {
            const float s1 = (red[0][lane][j] + red[1][lane][j]) + (red[2][lane][j] + red[3][lane][j]);
            atomicAdd(dw1 + (size_t)(j0 + j) * C + c, s1);
          }
      }
      // dW2[c][j]: lanes take (channel, bottleneck column) pairs in memory order -- 64 / JC channels x JC consecutive columns
      // per instruction -- instead of one channel per lane at a stride of rd floats (a cache-line operation per lane)
      constexpr int CPI = 64 / JC;
      const int jl = lane % JC, cl = lane / JC;
      for (int cc = 0; cc < 64; cc += CPI) {
        const int c2 = blockIdx.x * 64 + cc + cl;
        if (c2 < C && j0 + jl < rd) {
          const float s2 = (red[0][cc + cl][JC + jl] + red[1][cc + cl][JC + jl]) + (red[2][cc + cl][JC + jl] + red[3][cc + cl][JC + jl]);
          atomicAdd(dw2 + (size_t)c2 * rd + j0 + jl, s2);
        }
      }
    }
  }
}

// y = x * gate[b][c] + add[b][c]     (the data gradient of the gated map: dy * gate + the pooled path's per-image constant)
__global__ __launch_bounds__(256) void se_scale_add_kernel(const uint4* __restrict__ x, const float* __restrict__ gate,
                                                           const float* __restrict__ add, uint4* __restrict__ y, int B, int L,
                                                           int C) {
  const int G = C / 8;
  const long long total = (long long)B * L * G;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int cg = (int)(idx % G), b = (int)(idx / ((long long)L * G));
    float f[8];
    unpack8(x[idx], f);
    const float4* gp = reinterpret_cast<const float4*>(gate + (size_t)b * C + cg * 8);
    const float4* ap = reinterpret_cast<const float4*>(add + (size_t)b * C + cg * 8);
    const float4 g0 = gp[0], g1 = gp[1], a0 = ap[0], a1 = ap[1];
    f[0] = fmaf(f[0], g0.x, a0.x); f[1] = fmaf(f[1], g0.y, a0.y); f[2] = fmaf(f[2], g0.z, a0.z); f[3] = fmaf(f[3], g0.w, a0.w);
    f[4] = fmaf(f[4], g1.x, a1.x); f[5] = fmaf(f[5], g1.y, a1.y); f[6] = fmaf(f[6], g1.z, a1.z); f[7] = fmaf(f[7], g1.w, a1.w);
    y[idx] = pack8(f);
  }
}

extern "C" {

int mpr_se_mlp_max_rd(void) { return SE_RD_MAX; }

int mpr_se_mlp_fwd(const float* pooled, const float* w1, const float* b1, const float* w2, const float* b2, float* z1, float* r,
                   float* gate, int B, int C, int rd, void* stream) {
  MPR_REQUIRE(pooled && w1 && b1 && w2 && b2 && z1 && r && gate && B > 0 && C > 0 && rd > 0 && rd <= SE_RD_MAX && C <= 8192,
              "mpr_se_mlp_fwd: bad arguments (rd=%d, at most %d)", rd, SE_RD_MAX);
  const int rdp = rd <= 8 ? 8 : rd <= 16 ? 16 : rd <= 32 ? 32 : 64;
  hipStream_t st = (hipStream_t)stream;
  const size_t ldsf = sizeof(float) * 256 * (size_t)(rd + 1);      // <= 66.5 KB
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)se_mlp_fwd_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    attr_set = true;
  }
  if (rdp == 8) se_mlp_fwd_kernel<8><<<B, 256, ldsf, st>>>(pooled, w1, b1, w2, b2, z1, r, gate, C, rd);
  else if (rdp == 16) se_mlp_fwd_kernel<16><<<B, 256, ldsf, st>>>(pooled, w1, b1, w2, b2, z1, r, gate, C, rd);
  else if (rdp == 32) se_mlp_fwd_kernel<32><<<B, 256, ldsf, st>>>(pooled, w1, b1, w2, b2, z1, r, gate, C, rd);
  else se_mlp_fwd_kernel<64><<<B, 256, ldsf, st>>>(pooled, w1, b1, w2, b2, z1, r, gate, C, rd);
  MPR_LAUNCH_CHECK("se_mlp_fwd_kernel");
  return MPR_OK;
}

int mpr_se_mlp_bwd(const float* dgate, const float* gate, const float* z1, const float* r, const float* pooled, const float* w1,
                   const float* w2, float* dz2, float* dz1, float* dpooled, float* dw1, float* db1, float* dw2, float* db2,
                   float dpooled_scale, int B, int C, int rd, void* stream) {
  MPR_REQUIRE(dgate && gate && z1 && r && pooled && w1 && w2 && dz2 && dz1 && dpooled && dw1 && db1 && dw2 && db2 && B > 0 &&
                  C > 0 && rd > 0 && rd <= SE_RD_MAX && C <= 8192,
              "mpr_se_mlp_bwd: bad arguments (rd=%d, at most %d)", rd, SE_RD_MAX);
  hipStream_t st = (hipStream_t)stream;
  const int rdp = rd <= 8 ? 8 : rd <= 16 ? 16 : rd <= 32 ? 32 : 64;
  const size_t lds = sizeof(float) * ((size_t)C + rdp + 256 * (rdp + 1));
  const dim3 wg(ceil_div(C, 64), B >= 64 ? 8 : 1);
#define MPR_SE(RDP_)                                                                                                           \
  do {                                                                                                                         \
    static bool attr_set = false;                                                                                              \
    if (!attr_set) {                                                                                                           \
      hipFuncSetAttribute((const void*)se_mlp_bwd_kernel<RDP_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);       \
      attr_set = true;                                                                                                         \
    }                                                                                                                          \
    se_mlp_bwd_kernel<RDP_><<<B, 256, lds, st>>>(dgate, gate, z1, w1, w2, dz2, dz1, dpooled, db1, db2, C, rd, dpooled_scale);  \
    se_mlp_wgrad_kernel<RDP_><<<wg, 256, 0, st>>>(dz2, dz1, r, pooled, dw1, dw2, B, C, rd);                                    \
  } while (0)
  if (rdp == 8) MPR_SE(8); else if (rdp == 16) MPR_SE(16); else if (rdp == 32) MPR_SE(32); else MPR_SE(64);
#undef MPR_SE
  MPR_LAUNCH_CHECK("se_mlp_bwd_kernel");
  return MPR_OK;
}

int mpr_se_scale_add(const void* x, const float* gate, const float* add, void* y, int B, int L, int C, void* stream) {
  MPR_REQUIRE(x && gate && add && y && B > 0 && L > 0 && C > 0 && C % 8 == 0, "mpr_se_scale_add: bad arguments");
  long long g = ((long long)B * L * (C / 8) + 255) / 256;
  se_scale_add_kernel<<<(unsigned)(g < 16384 ? g : 16384), 256, 0, (hipStream_t)stream>>>((const uint4*)x, gate, add, (uint4*)y,
                                                                                         B, L, C);
  MPR_LAUNCH_CHECK("se_scale_add_kernel");
  return MPR_OK;
}

}  // extern "C"
