// Small HBM-bound pieces of the optimisation step:
//   * multi-tensor SGD (momentum / nesterov / weight decay) over every parameter in ONE launch
//     (reference: optim.SGD(self.parameters(), **optim_args), src/model.py:147-148 -- it covers the
//     encoders, the projections AND the loss parameters)
//   * encoder tail: concat(features, metadata / denom) + inverted dropout
//     (src/image_encoder.py:25-29, src/profile_encoder.py:64-68,234-240)
//   * softmax cross-entropy + argmax for the single-modality classifiers (src/model.py:167,197,227)
#include "common.h"

struct SgdEntry {
  float* p;
  const float* g;
  float* m;
  long long n;
};

// torch.optim.SGD: g += wd*p; buf = first ? g : mom*buf + (1-damp)*g; g = nesterov ? g + mom*buf : buf; p -= lr*g
__global__ __launch_bounds__(256) void sgd_multi_kernel(const SgdEntry* __restrict__ table, float lr, float momentum,
                                                        float dampening, float wd, int nesterov, int first_step) {
  const SgdEntry e = table[blockIdx.y];
  const long long n4 = e.n >> 2;
  const long long stride = (long long)gridDim.x * 256;
  const bool aligned = ((((uintptr_t)e.p) | ((uintptr_t)e.g) | ((uintptr_t)e.m)) & 15) == 0;
  if (aligned) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
      float4 p = reinterpret_cast<float4*>(e.p)[i];
      const float4 g4 = reinterpret_cast<const float4*>(e.g)[i];
      float4 m = first_step || momentum == 0.f ? make_float4(0, 0, 0, 0) : reinterpret_cast<float4*>(e.m)[i];
      float pv[4] = {p.x, p.y, p.z, p.w}, gv[4] = {g4.x, g4.y, g4.z, g4.w}, mv[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float g = fmaf(wd, pv[k], gv[k]);
        if (momentum != 0.f) {
          mv[k] = first_step ? g : fmaf(momentum, mv[k], (1.f - dampening) * g);
          g = nesterov ? fmaf(momentum, mv[k], g) : mv[k];
        }
        pv[k] = fmaf(-lr, g, pv[k]);
      }
      reinterpret_cast<float4*>(e.p)[i] = make_float4(pv[0], pv[1], pv[2], pv[3]);
      if (momentum != 0.f) reinterpret_cast<float4*>(e.m)[i] = make_float4(mv[0], mv[1], mv[2], mv[3]);
    }
  }
  const long long tail0 = aligned ? (n4 << 2) : 0;
  for (long long i = tail0 + (long long)blockIdx.x * 256 + threadIdx.x; i < e.n; i += stride) {
    float g = fmaf(wd, e.p[i], e.g[i]);
    if (momentum != 0.f) {
      const float mb = first_step ? g : fmaf(momentum, e.m[i], (1.f - dampening) * g);
      e.m[i] = mb;
      g = nesterov ? fmaf(momentum, mb, g) : mb;
    }
    e.p[i] = fmaf(-lr, g, e.p[i]);
  }
}

__device__ __forceinline__ uint32_t mix32(uint32_t x) {   // lowbias32 integer hash
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

// out[b][0:F] = feat[b][:], out[b][F:F+Mm] = meta[b][:] * inv_denom; then inverted dropout with keep mask
__global__ __launch_bounds__(256) void tail_fwd_kernel(const float* __restrict__ feat, const long long* __restrict__ meta,
                                                       float inv_denom, float p_drop, uint32_t seed,
                                                       float* __restrict__ out, unsigned char* __restrict__ mask, int B,
                                                       int F, int Mm) {
  const int W = F + Mm;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * W) return;
  const int b = i / W, c = i - b * W;
  float v = c < F ? feat[(size_t)b * F + c] : (float)meta[(size_t)b * Mm + (c - F)] * inv_denom;
  if (p_drop > 0.f) {
    const uint32_t h = mix32(mix32((uint32_t)i ^ seed) + 0x9e3779b9U * (seed | 1u));
    const bool keep = (float)(h >> 8) * (1.f / 16777216.f) >= p_drop;
    v = keep ? v / (1.f - p_drop) : 0.f;
    mask[i] = keep;
  }
  out[i] = v;
}

__global__ __launch_bounds__(256) void tail_bwd_kernel(const float* __restrict__ dout, const unsigned char* __restrict__ mask,
                                                       float p_drop, float* __restrict__ dfeat, int B, int F, int Mm) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * F) return;
  const int b = i / F, c = i - b * F;
  const size_t o = (size_t)b * (F + Mm) + c;
  float v = dout[o];
  if (p_drop > 0.f) v = mask[o] ? v / (1.f - p_drop) : 0.f;
  dfeat[i] = v;
}

// one wave per row: loss_i = lse(logits_i) - logits_i[label_i]; argmax; dlogits = (softmax - onehot) / rows
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* __restrict__ logits,
                                                         const long long* __restrict__ labels,
                                                         float* __restrict__ row_loss, long long* __restrict__ argmax,
                                                         float* __restrict__ dlogits, int rows, int C) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* l = logits + (size_t)row * C;
  float m = -INFINITY;
  int am = 0;
  for (int j = lane; j < C; j += 64)
    if (l[j] > m) { m = l[j]; am = j; }
  // first maximum wins (torch.argmax on CPU returns the lowest index among equals)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(m, o, 64);
    const int oa = __shfl_xor(am, o, 64);
    if (om > m || (om == m && oa < am)) { m = om; am = oa; }
  }
  float s = 0.f;
  for (int j = lane; j < C; j += 64) s += expf(l[j] - m);
  s = wave_sum(s);
  const float lse = m + logf(s);
  const int lab = labels ? (int)labels[row] : 0;
  if (lane == 0) {
    if (row_loss) row_loss[row] = labels ? lse - l[lab] : 0.f;
    if (argmax) argmax[row] = am;
  }
  if (dlogits && labels)
    for (int j = lane; j < C; j += 64)
      dlogits[(size_t)row * C + j] = (expf(l[j] - lse) - (j == lab ? 1.f : 0.f)) / (float)rows;
}

// y[i] = x[i] * s[0]
__global__ __launch_bounds__(256) void scale_by_scalar_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                              float* __restrict__ y, long long n) {
  const float k = s[0];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = x[i] * k;
}

// mean of a float vector -> out[0]
__global__ __launch_bounds__(256) void mean_kernel(const float* __restrict__ x, float* __restrict__ out, int n) {
  __shared__ double red[256];
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) a += (double)x[i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(red[0] / n);
}

extern "C" {

// table: device array of {float* p, const float* g, float* m, int64 n} (32 bytes each)
int mpr_sgd_multi(const void* table, int ntensors, long long max_numel, float lr, float momentum, float dampening,
                  float weight_decay, int nesterov, int first_step, void* stream) {
  MPR_REQUIRE(table && ntensors > 0, "mpr_sgd_multi: empty table");
  MPR_REQUIRE(ntensors <= 65535, "mpr_sgd_multi: too many tensors (%d)", ntensors);
  long long gx = (max_numel / 4 + 255) / 256;
  if (gx > 256) gx = 256;
  if (gx < 1) gx = 1;
  sgd_multi_kernel<<<dim3((unsigned)gx, ntensors), 256, 0, (hipStream_t)stream>>>(
      (const SgdEntry*)table, lr, momentum, dampening, weight_decay, nesterov, first_step);
  MPR_LAUNCH_CHECK("sgd_multi_kernel");
  return MPR_OK;
}

int mpr_tail_fwd(const float* feat, const long long* meta, float inv_denom, float p_drop, unsigned seed, float* out,
                 void* mask, int B, int F, int Mm, void* stream) {
  MPR_REQUIRE(feat && out && (Mm == 0 || meta), "mpr_tail_fwd: null pointer");
  MPR_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || mask), "mpr_tail_fwd: bad dropout arguments");
  tail_fwd_kernel<<<ceil_div(B * (F + Mm), 256), 256, 0, (hipStream_t)stream>>>(feat, meta, inv_denom, p_drop, seed, out,
                                                                               (unsigned char*)mask, B, F, Mm);
  MPR_LAUNCH_CHECK("tail_fwd_kernel");
  return MPR_OK;
}

int mpr_tail_bwd(const float* dout, const void* mask, float p_drop, float* dfeat, int B, int F, int Mm, void* stream) {
  MPR_REQUIRE(dout && dfeat, "mpr_tail_bwd: null pointer");
  tail_bwd_kernel<<<ceil_div(B * F, 256), 256, 0, (hipStream_t)stream>>>(dout, (const unsigned char*)mask, p_drop, dfeat,
                                                                        B, F, Mm);
  MPR_LAUNCH_CHECK("tail_bwd_kernel");
  return MPR_OK;
}

// loss[0] = mean_i CE(logits_i, labels_i); argmax[rows] (int64); dlogits (optional) = dLoss/dlogits
int mpr_softmax_ce(const float* logits, const long long* labels, float* row_loss, float* loss, long long* argmax,
                   float* dlogits, int rows, int C, void* stream) {
  MPR_REQUIRE(logits && rows > 0 && C > 0, "mpr_softmax_ce: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  softmax_ce_kernel<<<ceil_div(rows, 4), 256, 0, st>>>(logits, labels, row_loss, argmax, dlogits, rows, C);
  MPR_LAUNCH_CHECK("softmax_ce_kernel");
  if (loss && row_loss && labels) {
    mean_kernel<<<1, 256, 0, st>>>(row_loss, loss, rows);
    MPR_LAUNCH_CHECK("mean_kernel");
  }
  return MPR_OK;
}

int mpr_scale_by_scalar(const float* x, const float* s, float* y, long long n, void* stream) {
  long long g = (n + 255) / 256;
  if (g > 2048) g = 2048;
  scale_by_scalar_kernel<<<(unsigned)g, 256, 0, (hipStream_t)stream>>>(x, s, y, n);
  MPR_LAUNCH_CHECK("scale_by_scalar_kernel");
  return MPR_OK;
}

}  // extern "C"
