// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the train_multi hot path.
// Written for wave64 / MFMA / 160 KiB LDS only -- there is no other target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define MPR_OK 0
#define MPR_EINVAL 1
#define MPR_EHIP 2

extern "C" void mpr_set_error(const char* fmt, ...);
extern "C" void* mpr_prof_begin(int kind, double work, void* stream);   // no-ops unless mpr_prof_enable(1)
extern "C" void mpr_prof_end(void* token, void* stream);
extern "C" void mpr_prof_bytes(void* token, double algorithmic_bytes);

#define MPR_REQUIRE(cond, ...)                  \
  do {                                          \
    if (!(cond)) {                              \
      mpr_set_error(__VA_ARGS__);               \
      return MPR_EINVAL;                        \
    }                                           \
  } while (0)

#define MPR_LAUNCH_CHECK(what)                                              \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      mpr_set_error("%s: %s", what, hipGetErrorString(e__));                \
      return MPR_EHIP;                                                      \
    }                                                                       \
  } while (0)

#define MPR_HIP(call)                                                       \
  do {                                                                      \
    hipError_t e__ = (call);                                                \
    if (e__ != hipSuccess) {                                                \
      mpr_set_error("%s: %s", #call, hipGetErrorString(e__));               \
      return MPR_EHIP;                                                      \
    }                                                                       \
  } while (0)

// Exact unsigned division by a runtime constant: q = umulhi(n, mul) >> sh for n < 2^31.
struct FastDiv {
  uint32_t mul, sh, d;
};
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  if (d == 1) {
    f.mul = 0;
    f.sh = 0;
    return f;
  }
  uint32_t l = 0;
  while ((1u << l) < d) ++l;  // ceil(log2 d)
  uint64_t m = ((uint64_t(1) << (32 + l - 1)) + d - 1) / d;   // magic for n < 2^31
  f.mul = (uint32_t)m;
  f.sh = l - 1;
  return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
  return f.d == 1 ? n : (__umulhi(n, f.mul) >> f.sh);
}

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }
__device__ __forceinline__ float bf16lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  // ONE v_cvt_pk_bf16_f32 (RNE, NaN-preserving).  Two scalar conversions + (ua | ub << 16) compile to two conversions, a shift
  // and an SDWA or: four VALU instructions per pair in every epilogue of the library.
  const f32x2_t f = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_t));
}
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = bf16lo(v.x); f[1] = bf16hi(v.x); f[2] = bf16lo(v.y); f[3] = bf16hi(v.y);
  f[4] = bf16lo(v.z); f[5] = bf16hi(v.z); f[6] = bf16lo(v.w); f[7] = bf16hi(v.w);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 v;
  v.x = pack_bf16x2(f[0], f[1]); v.y = pack_bf16x2(f[2], f[3]);
  v.z = pack_bf16x2(f[4], f[5]); v.w = pack_bf16x2(f[6], f[7]);
  return v;
}
__device__ __forceinline__ float round_bf16(float x) { return (float)((bf16_t)x); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Bijective XCD-aware block remap: blocks b and b+8 share an XCD (round-robin dispatch), so give
// every XCD one contiguous run of tiles (neighbouring tiles then share halo rows / weight panels in
// that XCD's 4 MiB L2).  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  int q = nwg >> 3, r = nwg & 7, x = bid & 7, i = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
