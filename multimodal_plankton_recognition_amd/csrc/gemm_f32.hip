// Exact-fp32 batched GEMM on the f32-input MFMA (v_mfma_f32_32x32x2_f32: bitwise a k-ordered fmaf
// chain, runs at the fp32 vector peak).  Used where the reference's arithmetic is small and parity
// with its fp32 CPU path should be tight: the bias-free projection heads (src/model.py:31-32,40-41,80-82),
// the classifier heads of ImageModel / ProfileModel (src/model.py:164,316) and the all-pairs similarity
// matrix + its two gradient products in the coordination losses (src/coordination.py:38,89).
//   C[b] = alpha * op(A[b]) * op(B[b]) (+ bias) + beta * C[b],   op = identity or transpose, row-major, any ld.
// One workgroup = 4 waves = 64x64 of C, one 32x32 MFMA tile per wave.  K advances in chunks of 32 staged
// k-major in double-buffered LDS (so the MFMA operands -- lane -> row, lane>>5 -> k -- are conflict-free
// ds_read_b32); the global loads of chunk t+2 are issued before the 16 MFMAs of chunk t (two register sets), chunk t+1
// is written to the other LDS buffer after them: one barrier per chunk.  These problems are ~0.3 GFLOP each
// (latency-bound), so the point is to keep loads in flight, not tile reuse.
#include "common.h"

struct GemmF32Params {
  const float* A;
  const float* B;
  float* C;
  const float* bias;   // optional [N], added to every row
  int M, N, K, lda, ldb, ldc, transA, transB;
  long long sA, sB, sC;      // outer batch strides
  int inner;                 // two-level batch: z = outer_index * inner + inner_index
  long long iA, iB, iC;      // inner batch strides
  float alpha, beta;
};

#define GF_BK 32

__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmF32Params p) {
  __shared__ float As[2][GF_BK][64 + 4];
  __shared__ float Bs[2][GF_BK][64 + 4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int zo = blockIdx.z / p.inner, zi = blockIdx.z - zo * p.inner;
  const float* A = p.A + (long long)zo * p.sA + (long long)zi * p.iA;
  const float* B = p.B + (long long)zo * p.sB + (long long)zi * p.iB;
  float* C = p.C + (long long)zo * p.sC + (long long)zi * p.iC;

  // per-thread staging coordinates: 8 elements of each operand per chunk, coalesced along the operand's
  // contiguous axis
  int am[8], ak[8], bn[8], bk[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = tid + 256 * i;
    if (!p.transA) { ak[i] = idx & (GF_BK - 1); am[i] = idx / GF_BK; } else { am[i] = idx & 63; ak[i] = idx >> 6; }
    if (!p.transB) { bn[i] = idx & 63; bk[i] = idx >> 6; } else { bk[i] = idx & (GF_BK - 1); bn[i] = idx / GF_BK; }
  }
  // two register sets: chunk t+2 is requested while chunk t is multiplied and chunk t+1 waits in the other set -- these
  // problems run ONE workgroup per CU (64 of them for a 512 x 512 product), so nothing else hides the ~2 us of a load:
  // with one chunk of lookahead a 514-deep product took 38 us, 17 exposed latencies
  float ra[2][8], rb[2][8];
  auto load = [&](int k0, float* xa, float* xb) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int gm = m0 + am[i], gk = k0 + ak[i];
      xa[i] = (gm < p.M && gk < p.K) ? (p.transA ? A[(long long)gk * p.lda + gm] : A[(long long)gm * p.lda + gk]) : 0.f;
      const int gn = n0 + bn[i], gkb = k0 + bk[i];
      xb[i] = (gn < p.N && gkb < p.K) ? (p.transB ? B[(long long)gn * p.ldb + gkb] : B[(long long)gkb * p.ldb + gn]) : 0.f;
    }
  };
  auto store = [&](int buf, const float* xa, const float* xb) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      As[buf][ak[i]][am[i]] = xa[i];
      Bs[buf][bk[i]][bn[i]] = xb[i];
    }
  };

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;

  const int nk = (p.K + GF_BK - 1) / GF_BK;
  if (nk > 0) {
    load(0, ra[0], rb[0]);
    if (nk > 1) load(GF_BK, ra[1], rb[1]);
    store(0, ra[0], rb[0]);
  }
  __syncthreads();
  auto step = [&](int t, float* free_a, float* free_b, const float* next_a, const float* next_b) {
    const int cur = t & 1;
    if (t + 2 < nk) load((t + 2) * GF_BK, free_a, free_b);          // (the set whose chunk t is already in LDS)
#pragma unroll
    for (int s2 = 0; s2 < GF_BK / 2; ++s2) {
      const float a = As[cur][2 * s2 + (lane >> 5)][wm * 32 + (lane & 31)];
      const float b = Bs[cur][2 * s2 + (lane >> 5)][wn * 32 + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (t + 1 < nk) store(cur ^ 1, next_a, next_b);
    __syncthreads();
  };
  for (int t = 0; t < nk; t += 2) {
    step(t, ra[0], rb[0], ra[1], rb[1]);
    if (t + 1 < nk) step(t + 1, ra[1], rb[1], ra[0], rb[0]);
  }

  const int n = n0 + wn * 32 + (lane & 31);
  if (n < p.N) {
    const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
      if (m < p.M) {
        float* c = C + (long long)m * p.ldc + n;
        float v = p.alpha * acc[e] + bv;
        if (p.beta != 0.f) v += p.beta * *c;
        *c = v;
      }
    }
  }
}

extern "C" {

int mpr_gemm_f32(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int lda, int ldb,
                 int ldc, int transA, int transB, float alpha, float beta, int batch, long long strideA,
                 long long strideB, long long strideC, void* stream) {
  MPR_REQUIRE(A && B && C, "mpr_gemm_f32: null pointer");
  MPR_REQUIRE(M > 0 && N > 0 && K >= 0 && batch > 0, "mpr_gemm_f32: bad sizes M=%d N=%d K=%d batch=%d", M, N, K, batch);
  GemmF32Params p = {A, B, C, bias, M, N, K, lda, ldb, ldc, transA, transB, strideA, strideB, strideC, 1, 0, 0, 0, alpha, beta};
  dim3 grid(ceil_div(N, 64), ceil_div(M, 64), batch);
  gemm_f32_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(p);
  MPR_LAUNCH_CHECK("gemm_f32_kernel");
  return MPR_OK;
}

// Two-level batch (e.g. batch x heads with operands that are strided slices of one packed tensor):
// problem z = o * inner + i uses A + o*sAo + i*sAi, etc.
int mpr_gemm_f32_b2(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int transA,
                    int transB, float alpha, float beta, int outer, int inner, long long sAo, long long sAi, long long sBo,
                    long long sBi, long long sCo, long long sCi, void* stream) {
  MPR_REQUIRE(A && B && C, "mpr_gemm_f32_b2: null pointer");
  MPR_REQUIRE(M > 0 && N > 0 && K >= 0 && outer > 0 && inner > 0 && (long long)outer * inner <= 65535,
              "mpr_gemm_f32_b2: bad sizes M=%d N=%d K=%d batch=%dx%d", M, N, K, outer, inner);
  GemmF32Params p = {A, B, C, nullptr, M, N, K, lda, ldb, ldc, transA, transB, sAo, sBo, sCo, inner, sAi, sBi, sCi, alpha, beta};
  dim3 grid(ceil_div(N, 64), ceil_div(M, 64), outer * inner);
  gemm_f32_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(p);
  MPR_LAUNCH_CHECK("gemm_f32_kernel");
  return MPR_OK;
}

}  // extern "C"
