// ProfileLSTM (src/profile_encoder.py:71-108): the pointwise cell of nn.LSTM (gate order i, f, g, o), forward and
// backward, fp32.  The input projection of a whole layer and the per-step recurrent product h_{t-1} W_hh^T run on
// mpr_gemm_f32 (time-major activations: step t is a contiguous [B][4d] slice, the recurrent product accumulates into it);
// these kernels apply the gate nonlinearities and the state update.  Sequential by nature: latency-bound, no tuning.
#include "common.h"

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// G: [B][4d] pre-activations (x W_ih^T + b_ih + h_prev W_hh^T), b_hh added here
__global__ __launch_bounds__(256) void lstm_cell_fwd_kernel(const float* __restrict__ G, const float* __restrict__ b_hh,
                                                            const float* __restrict__ c_prev, float* __restrict__ act,
                                                            float* __restrict__ c, float* __restrict__ h, int B, int d) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * d) return;
  const int b = idx / d, j = idx - b * d;
  const float* g = G + (size_t)b * 4 * d;
  const float gi = sigmoidf_(g[j] + b_hh[j]);
  const float gf = sigmoidf_(g[d + j] + b_hh[d + j]);
  const float gg = tanhf(g[2 * d + j] + b_hh[2 * d + j]);
  const float go = sigmoidf_(g[3 * d + j] + b_hh[3 * d + j]);
  const float cp = c_prev ? c_prev[idx] : 0.f;
  const float cn = fmaf(gf, cp, gi * gg);
  float* a = act + (size_t)b * 4 * d;
  a[j] = gi; a[d + j] = gf; a[2 * d + j] = gg; a[3 * d + j] = go;
  c[idx] = cn;
  h[idx] = go * tanhf(cn);
}

// dh = dh_a (+ dh_b); dc: in = gradient of c_t from step t+1, out = gradient of c_{t-1}; dG: pre-activation gradients
__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(const float* __restrict__ act, const float* __restrict__ c_prev,
                                                            const float* __restrict__ c, const float* __restrict__ dh_a,
                                                            const float* __restrict__ dh_b, float* __restrict__ dc,
                                                            float* __restrict__ dG, int B, int d) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * d) return;
  const int b = idx / d, j = idx - b * d;
  const float* a = act + (size_t)b * 4 * d;
  const float gi = a[j], gf = a[d + j], gg = a[2 * d + j], go = a[3 * d + j];
  const float tc = tanhf(c[idx]);
  const float dh = dh_a[idx] + (dh_b ? dh_b[idx] : 0.f);
  const float dct = dc[idx] + dh * go * (1.f - tc * tc);
  const float cp = c_prev ? c_prev[idx] : 0.f;
  float* g = dG + (size_t)b * 4 * d;
  g[j] = dct * gg * gi * (1.f - gi);
  g[d + j] = dct * cp * gf * (1.f - gf);
  g[2 * d + j] = dct * gi * (1.f - gg * gg);
  g[3 * d + j] = dh * tc * go * (1.f - go);
  dc[idx] = dct * gf;
}

extern "C" {

int mpr_lstm_cell_fwd(const float* G, const float* b_hh, const float* c_prev, float* act, float* c, float* h, int B, int d,
                      void* stream) {
  MPR_REQUIRE(G && b_hh && act && c && h && B > 0 && d > 0, "mpr_lstm_cell_fwd: bad arguments");
  lstm_cell_fwd_kernel<<<ceil_div(B * d, 256), 256, 0, (hipStream_t)stream>>>(G, b_hh, c_prev, act, c, h, B, d);
  MPR_LAUNCH_CHECK("lstm_cell_fwd_kernel");
  return MPR_OK;
}

int mpr_lstm_cell_bwd(const float* act, const float* c_prev, const float* c, const float* dh_a, const float* dh_b, float* dc,
                      float* dG, int B, int d, void* stream) {
  MPR_REQUIRE(act && c && dh_a && dc && dG && B > 0 && d > 0, "mpr_lstm_cell_bwd: bad arguments");
  lstm_cell_bwd_kernel<<<ceil_div(B * d, 256), 256, 0, (hipStream_t)stream>>>(act, c_prev, c, dh_a, dh_b, dc, dG, B, d);
  MPR_LAUNCH_CHECK("lstm_cell_bwd_kernel");
  return MPR_OK;
}

}  // extern "C"
