// Round 3: does the MFMA SHAPE change the clock the chip holds under load?  (MI355X_MICROARCH.md, DVFS give-back (7): the
// 16x16x32 loop delivered ~1.12-1.15x the FLOP/s of the 32x32x16 loop at equal cycles.)  A conv_win-like inner loop: a wave
// owns a 64 x 64 output tile, both operands are re-read from LDS with ds_read_b128 every k-step (random bf16 data),
// 8 waves per workgroup, 2 workgroups per CU.  Build: hipcc --offload-arch=gfx950 -O3 mfma_shape.hip -o /tmp/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int SHAPE>   // 0: 32x32x16, 1: 16x16x32
__global__ __launch_bounds__(512, 4) void loop_kernel(const uint4* __restrict__ src, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [A: 256 rows x 128 B][B: 128 rows x 128 B]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid >> 1, wn = wid & 1;
  for (int i = tid; i < 384 * 8; i += 512) reinterpret_cast<uint4*>(smem)[i] = src[(blockIdx.x % 7) * 384 * 8 + i];
  __syncthreads();
  const unsigned char* A = smem + wm * 64 * 128;
  const unsigned char* B = smem + 256 * 128 + wn * 64 * 128;
  float sum = 0.f;
  if (SHAPE == 0) {
    f32x16 acc[2][2];
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 2; ++i) for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;
    const int frow = lane & 31, fh = lane >> 5;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const bf16x8*>(A + swz(i * 32 + frow, ks * 2 + fh));
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(B + swz(j * 32 + frow, ks * 2 + fh));
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j], af[i], acc[j][i], 0, 0, 0);
      }
    }
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 2; ++i) for (int e = 0; e < 16; ++e) sum += acc[j][i][e];
  } else {
    f32x4 acc[4][4];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) for (int e = 0; e < 4; ++e) acc[j][i][e] = 0.f;
    const int frow = lane & 15, fq = lane >> 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[4], bf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(A + swz(i * 16 + frow, ks * 4 + fq));
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(B + swz(j * 16 + frow, ks * 4 + fq));
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[j][i], 0, 0, 0);
      }
    }
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) for (int e = 0; e < 4; ++e) sum += acc[j][i][e];
  }
  if (sum == 12345.678f) out[0] = sum;
}

int main() {
  const size_t n16 = 7 * 384 * 8;
  std::vector<unsigned short> h(n16 * 8);
  srand(1);
  for (auto& v : h) {      // random bf16 in about [-2, 2]
    float f = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
    unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16);
  }
  uint4* src; float* out;
  hipMalloc(&src, n16 * 16); hipMalloc(&out, 4);
  hipMemcpy(src, h.data(), n16 * 16, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 512, iters = 4000;
  const size_t lds = 384 * 128;
  for (int rep = 0; rep < 4; ++rep)
    for (int shape = 0; shape < 2; ++shape) {
      auto launch = [&](int it) {
        if (shape == 0) loop_kernel<0><<<blocks, 512, lds>>>(src, out, it);
        else loop_kernel<1><<<blocks, 512, lds>>>(src, out, it);
      };
      launch(200);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int k = 0; k < 5; ++k) launch(iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flops = 5.0 * blocks * 8 * (double)iters * 64 * 64 * 64 * 2;
      printf("%s: %.2f ms  %.0f TFLOP/s\n", shape == 0 ? "32x32x16" : "16x16x32", ms, flops / ms / 1e9);
    }
  return 0;
}
