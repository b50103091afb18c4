// Exact-fp32 batched GEMM on the f32-input MFMA (v_mfma_f32_32x32x2_f32: bitwise a k-ordered fmaf
// chain, runs at the fp32 vector peak).  Used where the reference's arithmetic is small and parity
// with its fp32 CPU path should be tight: the bias-free projection heads (src/model.py:31-32,40-41,80-82),
// the classifier heads of ImageModel / ProfileModel (src/model.py:164,316) and the all-pairs similarity
// matrix + its two gradient products in the coordination losses (src/coordination.py:38,89).
//   C[b] = alpha * op(A[b]) * op(B[b]) + beta * C[b],   op = identity or transpose, row-major, any ld.
// One workgroup = 4 waves = 64x64 of C; K in chunks of 16 staged k-major in LDS so that the MFMA
// operands (lane -> row, lane>>5 -> k) are conflict-free ds_read_b32.
#include "common.h"

struct GemmF32Params {
  const float* A;
  const float* B;
  float* C;
  const float* bias;   // optional [N], added to every row
  int M, N, K, lda, ldb, ldc, transA, transB;
  long long sA, sB, sC;
  float alpha, beta;
};

__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmF32Params p) {
  __shared__ float As[16][64 + 4];
  __shared__ float Bs[16][64 + 4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const float* A = p.A + (long long)blockIdx.z * p.sA;
  const float* B = p.B + (long long)blockIdx.z * p.sB;
  float* C = p.C + (long long)blockIdx.z * p.sC;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;

  for (int k0 = 0; k0 < p.K; k0 += 16) {
    // stage 64x16 of op(A) and 16x64 of op(B); 4 elements per thread each, coalesced along the
    // operand's contiguous axis
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i;
      int m, k;
      if (!p.transA) { k = idx & 15; m = idx >> 4; } else { m = idx & 63; k = idx >> 6; }
      const int gm = m0 + m, gk = k0 + k;
      float v = 0.f;
      if (gm < p.M && gk < p.K) v = p.transA ? A[(long long)gk * p.lda + gm] : A[(long long)gm * p.lda + gk];
      As[k][m] = v;
      int n, kb;
      if (!p.transB) { n = idx & 63; kb = idx >> 6; } else { kb = idx & 15; n = idx >> 4; }
      const int gn = n0 + n, gkb = k0 + kb;
      float w = 0.f;
      if (gn < p.N && gkb < p.K) w = p.transB ? B[(long long)gn * p.ldb + gkb] : B[(long long)gkb * p.ldb + gn];
      Bs[kb][n] = w;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const float a = As[2 * s + (lane >> 5)][wm * 32 + (lane & 31)];
      const float b = Bs[2 * s + (lane >> 5)][wn * 32 + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
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
  GemmF32Params p = {A, B, C, bias, M, N, K, lda, ldb, ldc, transA, transB, strideA, strideB, strideC, alpha, beta};
  dim3 grid(ceil_div(N, 64), ceil_div(M, 64), batch);
  gemm_f32_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(p);
  MPR_LAUNCH_CHECK("gemm_f32_kernel");
  return MPR_OK;
}

}  // extern "C"
