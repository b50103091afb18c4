// GPU side of the input pipeline: the per-step, per-sample work of src/data.py:73-91 (ImageTransformTrain),
// :124-141 (ProfileTransformTrain) and :198-204 (PairAugmentation) on PRE-DECODED batches -- the JPEG decode, the scale-bar
// crop and the Lanczos resize to ceil(1.05 T) are done once when the dataset is cached (host, data.py), everything that is
// random per step runs here:
//   image   u8 [B][S][S]  -> random crop T x T, vertical flip, (paired) horizontal flip, /255*2-1      -> fp32 [B][1][T][T]
//   profile fp32 [B][Lmax][C] raw counts + length -> log1p / ceiling * 2 - 1, resize to S samples (anti-aliased bilinear:
//           torchvision's tensor Resize), random crop T, + sigma * N(0,1), (paired) time reversal         -> fp32 [B][T][C]
// The random decisions (offsets, flips) are inputs, so a batch is reproducible and testable against the host transforms.
#include "common.h"

__device__ __forceinline__ uint32_t ag_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

__global__ __launch_bounds__(256) void aug_image_kernel(const unsigned char* __restrict__ src, const int* __restrict__ top,
                                                        const int* __restrict__ left, const unsigned char* __restrict__ vflip,
                                                        const unsigned char* __restrict__ hflip, float* __restrict__ dst,
                                                        int B, int S, int T) {
  const long long total = (long long)B * T * T;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int x = (int)(idx % T);
    const long long r = idx / T;
    const int y = (int)(r % T), b = (int)(r / T);
    const int sy = top[b] + (vflip[b] ? T - 1 - y : y);
    const int sx = left[b] + (hflip[b] ? T - 1 - x : x);
    const float v = (float)src[((size_t)b * S + sy) * S + sx] / 255.f;
    dst[idx] = v * 2.f - 1.f;
  }
}

__global__ __launch_bounds__(256) void aug_profile_kernel(const float* __restrict__ raw, const int* __restrict__ length,
                                                          const int* __restrict__ left, const unsigned char* __restrict__ reverse,
                                                          const float* __restrict__ ceiling, float* __restrict__ dst, int B,
                                                          int Lmax, int C, int S, int T, float sigma, uint32_t seed) {
  const long long total = (long long)B * T * C;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % C);
    const long long r = idx / C;
    const int t = (int)(r % T), b = (int)(r / T);
    const int L = length[b];
    const int j = left[b] + (reverse[b] ? T - 1 - t : t);          // sample of the resized profile
    // torchvision's tensor Resize = interpolate(mode='bilinear', antialias=True) (aten _upsample_bilinear2d_aa): a
    // triangle filter of half-width max(scale, 1) around scale * (j + 0.5), weights normalised over the taps inside the
    // signal -- plain bilinear when upscaling, an area-weighted average when the profile is longer than S
    const float scale = (float)L / (float)S;
    const float support = scale >= 1.f ? scale : 1.f, invscale = scale >= 1.f ? 1.f / scale : 1.f;
    const float center = scale * ((float)j + 0.5f);
    int xmin = (int)(center - support + 0.5f);
    xmin = xmin < 0 ? 0 : xmin;
    int xmax = (int)(center + support + 0.5f);
    xmax = xmax > L ? L : xmax;
    const float* p = raw + (size_t)b * Lmax * C + c;
    const float inv = ceiling[c];
    float acc = 0.f, wsum = 0.f;
    for (int i = xmin; i < xmax; ++i) {
      float w = 1.f - fabsf(((float)i - center + 0.5f) * invscale);
      w = w > 0.f ? w : 0.f;
      acc += w * (logf(p[(size_t)i * C] + 1.f) / inv * 2.f - 1.f);
      wsum += w;
    }
    float v = wsum > 0.f ? acc / wsum : 0.f;
    if (sigma > 0.f) {                                              // Box-Muller on two counter hashes
      const uint32_t h1 = ag_mix32(ag_mix32((uint32_t)idx ^ seed) + 0x9e3779b9U * (seed | 1u) + (uint32_t)(idx >> 32));
      const uint32_t h2 = ag_mix32(h1 ^ 0x85ebca6bU);
      const float u1 = ((float)(h1 >> 8) + 1.f) * (1.f / 16777216.f), u2 = (float)(h2 >> 8) * (1.f / 16777216.f);
      v += sigma * sqrtf(-2.f * __logf(u1)) * __cosf(6.2831853071795865f * u2);
    }
    dst[idx] = v;
  }
}

static inline unsigned ag_grid(long long n) {
  long long g = (n + 255) / 256;
  return (unsigned)(g < 16384 ? (g < 1 ? 1 : g) : 16384);
}

extern "C" {

int mpr_aug_image(const void* src_u8, const int* top, const int* left, const void* vflip, const void* hflip, float* dst, int B,
                  int S, int T, void* stream) {
  MPR_REQUIRE(src_u8 && top && left && vflip && hflip && dst && B > 0 && T > 0 && S >= T, "mpr_aug_image: bad arguments");
  aug_image_kernel<<<ag_grid((long long)B * T * T), 256, 0, (hipStream_t)stream>>>(
      (const unsigned char*)src_u8, top, left, (const unsigned char*)vflip, (const unsigned char*)hflip, dst, B, S, T);
  MPR_LAUNCH_CHECK("aug_image_kernel");
  return MPR_OK;
}

int mpr_aug_profile(const float* raw, const int* length, const int* left, const void* reverse, const float* ceiling, float* dst,
                    int B, int Lmax, int C, int S, int T, float sigma, unsigned seed, void* stream) {
  MPR_REQUIRE(raw && length && left && reverse && ceiling && dst && B > 0 && Lmax > 0 && C > 0 && T > 0 && S >= T,
              "mpr_aug_profile: bad arguments");
  aug_profile_kernel<<<ag_grid((long long)B * T * C), 256, 0, (hipStream_t)stream>>>(raw, length, left, (const unsigned char*)reverse,
                                                                                    ceiling, dst, B, Lmax, C, S, T, sigma, seed);
  MPR_LAUNCH_CHECK("aug_profile_kernel");
  return MPR_OK;
}

}  // extern "C"
