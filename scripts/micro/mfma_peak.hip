// Practical bf16 MFMA ceiling of this MI355X (power-limited clock): waves that do nothing but v_mfma_f32_32x32x16_bf16.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x & 3); b[e] = (__bf16)1.0f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0];
  if (s == 12345.f) out[0] = s;
}
int main() {
  float* out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wpc : {4, 8, 16}) {        // waves per CU
    const int blocks = 256 * wpc / 4, iters = 20000;
    mfma_loop<4><<<blocks, 256>>>(out, 100);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      mfma_loop<4><<<blocks, 256>>>(out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double flops = (double)blocks * 4 * iters * 4 * 32768.0;
      printf("waves/CU %2d: %.2f ms  %.0f TFLOP/s  (=> %.2f GHz if every SIMD issues one MFMA per 32 clk)\n", wpc, ms,
             flops / ms / 1e9, flops / ms / 1e9 / 2516.0 * 2.4);
    }
  }
  return 0;
}
