// Weight-gradient of the implicit-GEMM convolution: dW[k][(r,s,c)] = sum_pix dy[pix][k] * xg[pix][(r,s,c)]
// (xg = x gathered at tap (r,s); pix = (b,p,q)).  Both operands are "pixel-major" in memory while the
// contraction runs over pixels, so each tile is staged row-major [pixel][channel] in LDS and consumed
// through ds_read_b64_tr_b16 (the CDNA4 transposing LDS read), which hands every lane the 4 consecutive
// pixels of ONE channel that v_mfma_f32_32x32x16_bf16 wants -- no shuffles, no second LDS image.
// The pixel axis is split over workgroups (split-K); partial tiles are summed into a zeroed fp32
// [K][R*S*C] buffer with global_atomic_add_f32 (each wave instruction = two 128-B row segments), and a
// tiny kernel permutes that buffer into torch's OIHW layout.
// Replaces the weight-gradient half of cuDNN/MIOpen conv backward behind the reference's
// loss.backward() (Lightning fit loop over src/model.py:93-101).
#include "common.h"

struct WgradParams {
  const bf16_t* x;
  const bf16_t* dy;
  float* dw;
  int H, W, C, K, P, Q, R, S, sh, sw, ph, pw;
  int Mpix, Ng, cps, ntm, ntn;
  FastDiv div_pq, div_q, div_c, div_s;
};

template <int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void conv_wgrad_kernel(const WgradParams p) {
  constexpr int T = 64 * WM * WN;
  constexpr int BM = 64 * WM, BN = 64 * WN, BKP = 32;
  constexpr int ASTR = BM * 2 + 64, BSTR = BN * 2 + 64;   // row strides = 64 or 192 mod 256: tr reads conflict-free
  constexpr int A_BYTES = BKP * ASTR, STAGE = BKP * (ASTR + BSTR);
  constexpr int CA = BM / 8, CB = BN / 8;                 // 16-B chunks per tile row
  constexpr int A_IT = (BKP * CA + T - 1) / T, B_IT = (BKP * CB + T - 1) / T;
  static_assert(T % CA == 0 && T % CB == 0, "column chunk must be thread-invariant");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  int bid = blockIdx.x;
  const int nt = bid % p.ntn; bid /= p.ntn;
  const int mt = bid % p.ntm;
  const int split = bid / p.ntm;
  const int m0 = mt * BM, n0 = nt * BN;
  const int pix_begin = split * p.cps * BKP;
  int pix_end = pix_begin + p.cps * BKP;
  if (pix_end > p.Mpix) pix_end = p.Mpix;
  const int nchunks = (pix_end - pix_begin + BKP - 1) / BKP;

  // A operand (dy): thread-fixed channel chunk
  const int a_ch = tid % CA, a_r0 = tid / CA;
  const bool a_col_ok = (m0 + a_ch * 8) < p.K;
  // B operand (x gather): thread-fixed (tap, channel) chunk
  const int b_ch = tid % CB, b_r0 = tid / CB;
  const int ncol = n0 + b_ch * 8;
  const bool b_col_ok = ncol < p.Ng;
  int tr, ts, tc;
  {
    uint32_t t = fdiv(ncol, p.div_c);
    tc = ncol - t * p.C;
    tr = fdiv(t, p.div_s);
    ts = t - tr * p.S;
  }

  uint4 areg[A_IT], breg[B_IT];
  auto load_chunk = [&](int ci) {
    const int kb = pix_begin + ci * BKP;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int row = a_r0 + i * (T / CA);
      const int pix = kb + row;
      const bool ok = a_col_ok && row < BKP && pix < pix_end;
      areg[i] = ok ? *reinterpret_cast<const uint4*>(p.dy + (size_t)pix * p.K + m0 + a_ch * 8)
                   : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int row = b_r0 + i * (T / CB);
      const int pix = kb + row;
      bool ok = b_col_ok && row < BKP && pix < pix_end;
      int off = 0;
      if (ok) {
        const uint32_t b = fdiv(pix, p.div_pq);
        const uint32_t rem = pix - b * (uint32_t)(p.P * p.Q);
        const uint32_t pp = fdiv(rem, p.div_q);
        const uint32_t qq = rem - pp * p.Q;
        const int ih = (int)pp * p.sh - p.ph + tr, iw = (int)qq * p.sw - p.pw + ts;
        ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        off = ((b * p.H + ih) * p.W + iw) * p.C + tc;
      }
      breg[i] = ok ? *reinterpret_cast<const uint4*>(p.x + off) : make_uint4(0, 0, 0, 0);
    }
  };
  auto store_chunk = [&](int buf) {
    unsigned char* a = smem + buf * STAGE;
    unsigned char* b = a + A_BYTES;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int row = a_r0 + i * (T / CA);
      if (row < BKP) *reinterpret_cast<uint4*>(a + row * ASTR + a_ch * 16) = areg[i];
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int row = b_r0 + i * (T / CB);
      if (row < BKP) *reinterpret_cast<uint4*>(b + row * BSTR + b_ch * 16) = breg[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // transposing-read lane geometry: 16-lane group g16 reads a 4(k) x 16(col) block; lane 4q+pq supplies
  // the address of k-row q, columns 4pq..4pq+3 and receives column (lane&15), k-rows 0..3.
  const int g16 = lane >> 4, lq = (lane & 15) >> 2, lp = lane & 3;
  const int tr_col = 16 * (g16 & 1) + 4 * lp;      // element column inside a 32-wide operand tile
  const int tr_row = 8 * (g16 >> 1) + lq;          // k-row inside a 16-deep k-step (+4 for the 2nd read)

  if (nchunks > 0) {
    load_chunk(0);
    store_chunk(0);
  }
  __syncthreads();
  for (int ci = 0; ci < nchunks; ++ci) {
    const int cur = ci & 1;
    if (ci + 1 < nchunks) load_chunk(ci + 1);
    const unsigned char* a = smem + cur * STAGE;
    const unsigned char* b = a + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const unsigned char* pa = a + (ks * 16 + tr_row) * ASTR + (wm * 64 + t * 32 + tr_col) * 2;
        const unsigned char* pb = b + (ks * 16 + tr_row) * BSTR + (wn * 64 + t * 32 + tr_col) * 2;
        typedef s16x4 __attribute__((address_space(3))) * lds_v4;
        s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(pa));
        s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(pa + 4 * ASTR));
        s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(pb));
        s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(pb + 4 * BSTR));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        s16x8 av = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
        s16x8 bv = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
        af[t] = __builtin_bit_cast(bf16x8, av);
        bfr[t] = __builtin_bit_cast(bf16x8, bv);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (ci + 1 < nchunks) store_chunk(cur ^ 1);
    __syncthreads();
  }

  // D[i = out-channel (regs)][n = (tap, channel) (lanes)] -> fp32 atomics, 128 B contiguous per half-wave
  const int ln = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 64 + j * 32 + ln;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int k = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (k < p.K && n < p.Ng) atomicAdd(p.dw + (size_t)k * p.Ng + n, acc[i][j][e]);
      }
    }
}

// [K][(r,s,c)] fp32 -> OIHW fp32 (optionally accumulating into an existing gradient)
__global__ void wgrad_unpack_kernel(const float* __restrict__ src, float* __restrict__ dst, int K, int C,
                                    int R, int S, int accumulate) {
  const int total = K * C * R * S;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int s = i % S, t = i / S;
    int r = t % R; t /= R;
    int c = t % C;
    int k = t / C;
    const float v = src[(size_t)k * (R * S * C) + (r * S + s) * C + c];
    dst[i] = accumulate ? dst[i] + v : v;
  }
}

extern "C" {

// dw_oihw[K,C,R,S] (fp32) = sum over the batch;  workspace: K*R*S*C floats (zeroed here).
int mpr_conv_wgrad(const void* x, const void* dy, float* workspace, float* dw_oihw, int accumulate, int B,
                   int H, int W, int C, int K, int R, int S, int sh, int sw, int ph, int pw, void* stream) {
  MPR_REQUIRE(x && dy && workspace && dw_oihw, "mpr_conv_wgrad: null pointer");
  MPR_REQUIRE(C % 8 == 0 && K % 8 == 0, "mpr_conv_wgrad: C (%d) and K (%d) must be multiples of 8", C, K);
  const int P = (H + 2 * ph - R) / sh + 1, Q = (W + 2 * pw - S) / sw + 1;
  MPR_REQUIRE(P > 0 && Q > 0, "mpr_conv_wgrad: empty output");
  MPR_REQUIRE((long long)B * H * W * C < (1ll << 31) && (long long)B * P * Q * K < (1ll << 31),
              "mpr_conv_wgrad: tensor exceeds 2^31 elements");
  hipStream_t st = (hipStream_t)stream;
  WgradParams p;
  p.x = (const bf16_t*)x; p.dy = (const bf16_t*)dy; p.dw = workspace;
  p.H = H; p.W = W; p.C = C; p.K = K; p.P = P; p.Q = Q; p.R = R; p.S = S; p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw;
  p.Mpix = B * P * Q; p.Ng = R * S * C;
  p.div_pq = make_fastdiv(P * Q); p.div_q = make_fastdiv(Q); p.div_c = make_fastdiv(C); p.div_s = make_fastdiv(S);
  MPR_HIP(hipMemsetAsync(workspace, 0, sizeof(float) * (size_t)K * p.Ng, st));

  // tile shape: M = out channels (64 or 128 per WG), N = (tap, channel) columns in 64 / 128 / 192
  const int WM = K <= 64 ? 1 : 2;
  int WN;
  if (p.Ng % 192 == 0) WN = 3;
  else if (p.Ng <= 64) WN = 1;
  else WN = 2;
  const int BM = 64 * WM, BN = 64 * WN;
  p.ntm = ceil_div(K, BM); p.ntn = ceil_div(p.Ng, BN);
  const int tiles = p.ntm * p.ntn;
  const int total_chunks = ceil_div(p.Mpix, 32);
  int nsplit = ceil_div(1024, tiles);                       // aim at ~4 workgroups per CU
  if (nsplit > ceil_div(total_chunks, 8)) nsplit = ceil_div(total_chunks, 8);   // >= 8 chunks per split
  if (nsplit < 1) nsplit = 1;
  p.cps = ceil_div(total_chunks, nsplit);
  nsplit = ceil_div(total_chunks, p.cps);
  dim3 grid(tiles * nsplit);
#define MPR_WG(WM_, WN_) conv_wgrad_kernel<WM_, WN_><<<grid, 64 * WM_ * WN_, 0, st>>>(p)
  void* tok = mpr_prof_begin(2, 2.0 * (double)p.Mpix * (double)K * (double)p.Ng, st);
  if (WM == 1 && WN == 1) MPR_WG(1, 1);
  else if (WM == 1 && WN == 2) MPR_WG(1, 2);
  else if (WM == 1 && WN == 3) MPR_WG(1, 3);
  else if (WM == 2 && WN == 1) MPR_WG(2, 1);
  else if (WM == 2 && WN == 2) MPR_WG(2, 2);
  else MPR_WG(2, 3);
  mpr_prof_end(tok, st);
#undef MPR_WG
  MPR_LAUNCH_CHECK("conv_wgrad_kernel");
  const int total = K * C * R * S;
  const int g2 = ceil_div(total, 256) < 2048 ? ceil_div(total, 256) : 2048;
  wgrad_unpack_kernel<<<g2, 256, 0, st>>>(workspace, dw_oihw, K, C, R, S, accumulate);
  MPR_LAUNCH_CHECK("wgrad_unpack_kernel");
  return MPR_OK;
}

}  // extern "C"
