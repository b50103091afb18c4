// ResNet stem, fused and recomputed:  conv 7x7/2 (1 -> 64 channels) -> BatchNorm -> ReLU -> MaxPool 3x3/2
// (timm ResNet conv1/bn1/act1/maxpool behind src/image_encoder.py:16,24) without ever writing the full-resolution
// conv output.  At batch 512 that map is 822 MB of bf16; the unfused chain (stem.hip + pooling.hip) wrote it once and
// read it four times (BatchNorm+pool forward, two BatchNorm-backward passes, weight gradient): ~7.5 GB of the step's
// 40 GB.  The convolution itself is 105 GFLOP -- cheap -- so it is RECOMPUTED instead:
//
//   forward   pass A  conv tile (fp32, in registers) -> per-channel sum / sum of squares         (nothing written)
//             pass B  conv tile -> BatchNorm -> ReLU -> bf16 -> 3x3/2 max pool in registers ->  pooled map + 1-byte arg-max
//                     code (15 where the pooled activation is 0: ReLU' of the winner is 0, no gradient flows there -- so
//                     the backward needs neither a ReLU mask nor the activation)
//   backward  ONE pass over (pooled gradient, arg-max codes, input): dz = pool-scatter(dpooled), then
//                     Z[c][tap] = sum_px dz[px][c] * patch[px][tap]      (MFMA)
//                     G[t'][t]  = sum_px patch[px][t'] * patch[px][t]    (MFMA, the Gram matrix of the input patches)
//             and a finalize that uses conv LINEARITY (y[px][c] = sum_t w[c][t] patch[px][t]) for everything that
//             needed y:   sum dz         = Z[c][one]            (a padded tap slot holds the constant 1)
//                         sum dz * y     = sum_t w[c][t] Z[c][t]
//                         dW[c][t]       = k1 Z[c][t] + k2 (w G)[c][t] + k3 G[one][t]      (dx = k1 dz + k2 y + k3)
//             so the backward never recomputes the convolution at all.
//
// Geometry: the input is copied once per step into a zero-padded bf16 image xb[B][H+6][W+8] (3 rows / 3 columns of
// padding in front), so tap row r of conv row h is image row 2h + r and the 8-tap group of conv column c starts at
// column 2c: a 4-byte-aligned 16-byte load IS the MFMA operand fragment (k = 8 taps of one filter row; 7 rows x 8 = 56
// of the 64 k slots are used, the weights of the rest are zero).
//
// Forward work item = (image, 32-column block) per WAVE, streamed top to bottom: no LDS, no barriers.  Column blocks
// overlap by two conv columns (block t = columns 30t-1 .. 30t+30) so that the 15 pooled columns of a block need no
// neighbour; the vertical 3-window is a running max over the rows, the horizontal one two DPP wave shifts.  Pooling
// works on integer keys (bf16 bits << 4 | 15 - tap code): one v_max3 picks the largest value AND, among equals, the
// first tap in torch's (kh, kw) scan order.
#include "common.h"

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define STEMF_INVALID 0   // key of a position outside the map (real keys are >= 7 after the column offsets)

struct StemfFin {   // BatchNorm finalize folded into pass B (cf. bn_block_finalize in batchnorm.hip)
  float count, momentum, eps;
  const float *gamma, *beta;
  float *running_mean, *running_var, *scale, *shift, *mean_out, *invstd_out;
};

struct StemfP {
  const bf16_t* xb;     // [B][HP][WP]
  const bf16_t* wp;     // [64 channels][64 slots], slot = r*8 + s (zero for r == 7 or s == 7)
  int B, H, W, P, Q, HP, WP, NT, P2, Q2, nitems;
  float* stats;         // pass A: [stat_slices][2][64], zeroed, fp32 atomics
  int stat_slices;
  const float* slices;  // pass B: partial sums to finalize here (train) or NULL (scale / shift given: eval)
  int nsl;
  StemfFin fin;
  bf16_t* pooled;       // [B][P2][Q2][64]
  unsigned char* idx;   // [B][P2][Q2][64] tap code kh*3 + kw (15: pooled activation is 0), or NULL
};

__global__ __launch_bounds__(256) void stemf_prep_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         bf16_t* __restrict__ xb, bf16_t* __restrict__ wp, int B, int H,
                                                         int W) {
  const unsigned HP = H + 6, WP = W + 8, VPR = WP / 8;
  const unsigned nvec = (unsigned)B * HP * VPR;
  for (unsigned v = blockIdx.x * 256 + threadIdx.x; v < nvec; v += gridDim.x * 256) {
    const unsigned vc = v % VPR, t = v / VPR;
    const int hr = (int)(t % HP), b = (int)(t / HP), h = hr - 3;
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int wc = (int)vc * 8 + e - 3;
      f[e] = (h >= 0 && h < H && wc >= 0 && wc < W) ? x[((size_t)b * H + h) * W + wc] : 0.f;
    }
    reinterpret_cast<uint4*>(xb)[v] = pack8(f);
  }
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < 4096; i += 256) {
      const int c = i >> 6, r = (i >> 3) & 7, s = i & 7;
      wp[i] = (bf16_t)((r < 7 && s < 7) ? w[c * 49 + r * 7 + s] : 0.f);
    }
}

__device__ __forceinline__ int stemf_lane_up1(int v) {   // lane i <- lane i + 1 (DPP wave_shl:1); 0 (= outside the map) at the end
  return __builtin_amdgcn_mov_dpp(v, 0x130, 0xf, 0xf, true);
}
__device__ __forceinline__ int stemf_max3(int a, int b, int c) { return max(max(a, b), c); }
typedef float f32x2 __attribute__((ext_vector_type(2)));
// sum over the 16 lanes of a DPP row (every lane of the row ends up with it): quad swaps, half mirror, mirror
__device__ __forceinline__ float stemf_row_sum16(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));
  return v;
}

// MODE 0: statistics only.  MODE 1: BatchNorm + ReLU + max pool.
template <int MODE>
__global__ __launch_bounds__(256, 2) void stemf_fwd_kernel(const StemfP p) {
  __shared__ __attribute__((aligned(16))) float fl[2][64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int i = lane & 31, fh = lane >> 5;
  if (MODE == 1) {
    if (p.slices) {
      if (tid < 64) {
        const int c = tid;
        double s1 = 0.0, s2 = 0.0;
        for (int r = 0; r < p.nsl; ++r) {
          s1 += (double)p.slices[((size_t)r * 2) * 64 + c];
          s2 += (double)p.slices[((size_t)r * 2 + 1) * 64 + c];
        }
        const double mean = s1 / p.fin.count;
        double var = s2 / p.fin.count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)p.fin.eps));
        const float g = p.fin.gamma[c], bb = p.fin.beta[c];
        const float sc = g * invstd, sh = bb - (float)mean * g * invstd;
        fl[0][c] = sc;
        fl[1][c] = sh;
        if (blockIdx.x == 0) {
          p.fin.scale[c] = sc;
          p.fin.shift[c] = sh;
          p.fin.mean_out[c] = (float)mean;
          p.fin.invstd_out[c] = invstd;
          if (p.fin.running_mean) {
            const double unbiased = p.fin.count > 1.f ? var * (double)p.fin.count / ((double)p.fin.count - 1.0) : var;
            p.fin.running_mean[c] = (1.f - p.fin.momentum) * p.fin.running_mean[c] + p.fin.momentum * (float)mean;
            p.fin.running_var[c] = (1.f - p.fin.momentum) * p.fin.running_var[c] + p.fin.momentum * (float)unbiased;
          }
        }
      }
    } else if (tid < 64) {
      fl[0][tid] = p.fin.scale[tid];
      fl[1][tid] = p.fin.shift[tid];
    }
    __syncthreads();
  }
  if (MODE == 0) {
    if (tid < 128) fl[tid >> 6][tid & 63] = 0.f;
    __syncthreads();
  }
  int item = blockIdx.x * 4 + wid;
  const bool inactive = item >= p.nitems;          // (a spare wave of the last workgroup: it still joins the barriers)
  if (inactive) item = p.nitems - 1;
  const int b = item / p.NT, t = item - b * p.NT;
  const int c = 30 * t - 1 + i;                       // conv column of this lane
  const bool valid_c = c >= 0 && c < p.Q;
  const bool owned = valid_c && i >= 1 && i <= 30;    // every conv column is owned by exactly one block
  const int cc = c < 0 ? 0 : (c >= p.Q ? p.Q - 1 : c);

  // weight fragments (A operand): lane row n = channel 32j + (lane & 31), k = 8 taps of filter row 2ks + fh
  bf16x8 wf[4][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      wf[ks][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p.wp + (32 * j + i) * 64 + (2 * ks + fh) * 8));

  // Patch fragments (B operand): lane column m = pixel i, k = 8 taps of filter row 2ks + fh = image row 2h + 2ks + fh.
  // Conv row h + 1 needs the rows of conv row h shifted by one k step -- fragment ks of row h + 1 IS fragment ks + 1 of
  // row h -- so the fragments live in an 8-slot register ring (slot (h + ks) & 7, all indices static in the 8-row unrolled
  // body): ONE 16-byte load per lane and conv row, issued 4 rows ahead (the loop is latency-bound otherwise).  The
  // fh = 1 lanes' fragment of k step 3 is image row 2h + 7: a real row (rows 0 .. H+5 exist) under zero weights.
  const bf16_t* const xcol = p.xb + (size_t)b * p.HP * p.WP + 2 * cc + (size_t)fh * p.WP;
  bf16x8 ring[8];
  auto load_frag = [&](int row) -> bf16x8 {          // image row `row` + fh
    const int rr = row > p.H + 4 ? p.H + 4 : row;    // (prefetches past the last conv row: any readable row)
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_a4*>(xcol + (size_t)rr * p.WP));
  };
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) ring[ks] = load_frag(2 * ks);                 // conv row 0
#pragma unroll
  for (int k = 1; k < 4; ++k) ring[3 + k] = load_frag(2 * k + 6);              // new fragments of conv rows 1..3
  // conv row h (h & 7 == U): prefetch the new fragment of row h + 4, then 8 MFMAs on slots U .. U + 3
#define STEMF_CONV_ROW(U, h, acc)                                                                          \
  do {                                                                                                     \
    ring[((U) + 7) & 7] = load_frag(2 * ((h) + 4) + 6);                                                    \
    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                       \
      _Pragma("unroll") for (int e_ = 0; e_ < 16; ++e_) acc[j_][e_] = 0.f;                                 \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_)                                                    \
      _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                     \
        acc[j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks_][j_], ring[((U) + ks_) & 7], acc[j_], 0, 0, 0); \
  } while (0)

  if (MODE == 0 && !inactive) {
    // packed fp32 math (two accumulator registers per instruction), sums kept per lane over all rows
    f32x2 s1[2][8], s2[2][8];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) { s1[j][e] = (f32x2){0.f, 0.f}; s2[j][e] = (f32x2){0.f, 0.f}; }
    for (int h0 = 0; h0 < p.P; h0 += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (h0 + u < p.P) {                            // (wave-uniform; P is even, not necessarily a multiple of 8)
          f32x16 acc[2];
          STEMF_CONV_ROW(u, h0 + u, acc);
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const f32x2 v = {acc[j][2 * e], acc[j][2 * e + 1]};
              s1[j][e] += v;
              s2[j][e] = __builtin_elementwise_fma(v, v, s2[j][e]);
            }
        }
      }
    }
    // fold the 16 pixel lanes of every DPP row (un-owned lanes contribute 0), the rows and waves of the workgroup in LDS,
    // then ONE global atomic per (channel, sum) and workgroup: same-address global atomics serialise at the memory side
    // (512 adds per address took ~200 us here), so few adders per address and many slice rows
    const float own = owned ? 1.f : 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float a1 = stemf_row_sum16(s1[j][e >> 1][e & 1] * own), a2 = stemf_row_sum16(s2[j][e >> 1][e & 1] * own);
        if ((lane & 15) == 0) {
          const int ch = 32 * j + (e & 3) + 8 * (e >> 2) + 4 * fh;
          atomicAdd(&fl[0][ch], a1);
          atomicAdd(&fl[1][ch], a2);
        }
      }
  }
  if (MODE == 0) {
    __syncthreads();
    if (tid < 128) atomicAdd(p.stats + (size_t)(blockIdx.x % p.stat_slices) * 128 + tid, fl[tid >> 6][tid & 63]);
    return;
  }
  if (inactive) return;

  // ---- MODE 1
  // scale / shift stay in LDS (64 VGPRs otherwise): a lane's four consecutive channels are one 16-byte read, re-read per row
  const float4* const scl = reinterpret_cast<const float4*>(&fl[0][4 * fh]);   // [2 * (4j + g)] -> channels 32j + 8g + 4fh ..
  const float4* const shl = reinterpret_cast<const float4*>(&fl[1][4 * fh]);
  // Key of an activated value a >= 0: its fp32 bits with the low 4 mantissa bits replaced by (15 - tap code): integer max
  // = the largest activation (to 19 mantissa bits: the arg-max is taken on the fp32 activations, as in the reference;
  // only the pooled result is rounded to bf16, and rounding commutes with max) and, among equals, the first tap in torch's
  // (kh, kw) scan order.  The row part of the code (15 - 3 kh) goes in below, the column part (- kw) is subtracted in the
  // horizontal step.  Padding (rows / columns outside the map) is key 0: below every real key, and a window whose
  // maximum is 0 gets code 15 anyway.
  int kp[2][16];                                       // running vertical max, carried from row 2p-1 as kh = 0
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) kp[j][e] = 0;
  // two accumulator registers -> two keys with row code `rc`: packed fma, ReLU, one v_and_or each
  auto act_keys = [&](float y0, float y1, float s0, float s1_, float h0, float h1, int rc, int& k0, int& k1) {
    const f32x2 t = __builtin_elementwise_fma((f32x2){y0, y1}, (f32x2){s0, s1_}, (f32x2){h0, h1});
    k0 = (__builtin_bit_cast(int, fmaxf(t[0], 0.f)) & ~15) | rc;
    k1 = (__builtin_bit_cast(int, fmaxf(t[1], 0.f)) & ~15) | rc;
  };
  const int vmask = valid_c ? -1 : 0;
  const bool store_lane = !(i & 1) && i <= 28 && (15 * t + (i >> 1)) < p.Q2;
  const int q = 15 * t + (i >> 1);
  for (int h0 = 0; h0 < p.P; h0 += 8) {
#pragma unroll
    for (int u = 0; u < 8; u += 2) {
      if (h0 + u < p.P) {                              // (wave-uniform; P is even)
        const int h = h0 + u, pr = h >> 1;
        f32x16 acc[2];
        STEMF_CONV_ROW(u, h, acc);
        asm volatile("" ::: "memory");                // (keeps the scale / shift reads inside the loop)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 s4 = scl[2 * (4 * j + g)], h4 = shl[2 * (4 * j + g)];
            int k0, k1, k2, k3;
            act_keys(acc[j][4 * g], acc[j][4 * g + 1], s4.x, s4.y, h4.x, h4.y, 12, k0, k1);         // kh = 1
            act_keys(acc[j][4 * g + 2], acc[j][4 * g + 3], s4.z, s4.w, h4.z, h4.w, 12, k2, k3);
            kp[j][4 * g] = max(kp[j][4 * g], k0);
            kp[j][4 * g + 1] = max(kp[j][4 * g + 1], k1);
            kp[j][4 * g + 2] = max(kp[j][4 * g + 2], k2);
            kp[j][4 * g + 3] = max(kp[j][4 * g + 3], k3);
          }
        STEMF_CONV_ROW(u + 1, h + 1, acc);
        // window pr complete: rows 2pr-1 (kh 0, carried), 2pr (kh 1), 2pr+1 (kh 2); then the 3 columns via two wave shifts
        int kq[2][16];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 s4 = scl[2 * (4 * j + g)], h4 = shl[2 * (4 * j + g)];
            int kb[4];
            act_keys(acc[j][4 * g], acc[j][4 * g + 1], s4.x, s4.y, h4.x, h4.y, 9, kb[0], kb[1]);      // kh = 2
            act_keys(acc[j][4 * g + 2], acc[j][4 * g + 3], s4.z, s4.w, h4.z, h4.w, 9, kb[2], kb[3]);
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
              const int e = 4 * g + k4;
              const int k = max(kp[j][e], kb[k4]) & vmask;            // columns outside the map: 0
              kp[j][e] = kb[k4] | 15;                                 // the same row as kh = 0 of window pr + 1
              const int n1 = stemf_lane_up1(k);
              const int n2 = stemf_lane_up1(n1);
              kq[j][e] = stemf_max3(k, n1 - 1, n2 - 2);
            }
          }
        if (store_lane) {
          const size_t o = ((size_t)(b * p.P2 + pr) * p.Q2 + q) * 64;
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int ch0 = 32 * j + 8 * g + 4 * fh;
              const int k0 = kq[j][4 * g], k1 = kq[j][4 * g + 1], k2 = kq[j][4 * g + 2], k3 = kq[j][4 * g + 3];
              uint2 pv;                              // (n - kw can dip below 0 only for padding: max3 never returns it)
              pv.x = pack_bf16x2(__builtin_bit_cast(float, k0 & ~15), __builtin_bit_cast(float, k1 & ~15));
              pv.y = pack_bf16x2(__builtin_bit_cast(float, k2 & ~15), __builtin_bit_cast(float, k3 & ~15));
              *reinterpret_cast<uint2*>(p.pooled + o + ch0) = pv;
              if (p.idx) {
                // 15 - tap code sits in the low 4 bits; a key below 16 is an activation of 0 (code 15: no gradient)
                const uint32_t c0 = k0 < 16 ? 15u : (~k0 & 15u), c1 = k1 < 16 ? 15u : (~k1 & 15u);
                const uint32_t c2 = k2 < 16 ? 15u : (~k2 & 15u), c3 = k3 < 16 ? 15u : (~k3 & 15u);
                *reinterpret_cast<uint32_t*>(p.idx + o + ch0) = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
              }
            }
        }
      }
    }
  }
#undef STEMF_CONV_ROW
}

// ---------------------------------------------------------------------------------------------------------------
// backward
struct StembP {
  const bf16_t* xb;
  const bf16_t* dp;             // pooled gradient [B][P2][Q2][64]
  const unsigned char* idx;
  float* partial;               // [gridDim.x][7][16][64]: Z tiles (0,0) (0,1) (1,0) (1,1) [channel tile][tap tile], G (0,0) (0,1) (1,1)
  int B, H, W, P, Q, HP, WP, P2, Q2, NSEG, nitems;
};

__global__ __launch_bounds__(256, 2) void stemf_bwd_kernel(const StembP p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[7 * 16 * 64 * 4];    // 28 KB: 4 x 4 KB dz images, then the reduction
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  unsigned char* const img = smem + wid * 4096;        // this wave's [32 px][64 ch] bf16 image, 16-B chunks XOR-swizzled by row
  const int fh = lane >> 5, tl = lane & 31;
  f32x16 accz[2][2], accg[3];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int bq = 0; bq < 2; ++bq)
#pragma unroll
      for (int e = 0; e < 16; ++e) accz[a][bq][e] = 0.f;
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int e = 0; e < 16; ++e) accg[a][e] = 0.f;

  // elementwise geometry: lane -> (2x2 block jb of the 16-column segment, channel group g)
  const int jb = lane >> 3, g = lane & 7;
  // transposing-read geometry (A operand = dz^T: row = channel, k = 8 pixels)
  const int tr_row = 8 * fh + ((lane & 15) >> 2);                  // + 16 ks (+ 4 for the second read)
  const int tr_ch = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);       // + 32 jc
  // patch fragments (B operand / Gram operand): lane -> tap slot 32 tt + tl = r*8 + s, pixels 8 fh .. 8 fh + 7 of the k step
  int pr_[2], ps_[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) { pr_[tt] = (32 * tt + tl) >> 3; ps_[tt] = tl & 7; }
  const uint32_t sel = (tl & 1) ? 0x07060302u : 0x05040100u;       // odd / even elements of a 32-byte run (s & 1 == tl & 1)

  // the four pooling windows that touch a lane's 2 x 2 block (arg-max codes + pooled gradient), loaded ONE ITEM AHEAD:
  // the scatter below is the first thing an item does, its operands must not be a fresh global round trip
  auto load_windows = [&](int item, uint2* kk, uint4* dd) {
    const int seg = item % p.NSEG, t2 = item / p.NSEG;
    const int i2 = t2 % p.P2, b = t2 / p.P2;
    const int J = 8 * seg + jb;
    const bool hasj = J + 1 < p.Q2, hasi = i2 + 1 < p.P2;
    const unsigned w00 = ((unsigned)(b * p.P2 + i2) * p.Q2 + J) * 8 + g;
    const unsigned w[4] = {w00, hasj ? w00 + 8 : w00, hasi ? w00 + p.Q2 * 8 : w00, (hasi && hasj) ? w00 + p.Q2 * 8 + 8 : w00};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      kk[t] = reinterpret_cast<const uint2*>(p.idx)[w[t]];
      dd[t] = reinterpret_cast<const uint4*>(p.dp)[w[t]];
    }
  };
  uint2 kn[4];
  uint4 dn[4];
  const int item0 = blockIdx.x * 4 + wid, istep = gridDim.x * 4;
  if (item0 < p.nitems) load_windows(item0, kn, dn);
  for (int item = item0; item < p.nitems; item += istep) {
    const int seg = item % p.NSEG;
    const int t2 = item / p.NSEG;
    const int i2 = t2 % p.P2, b = t2 / p.P2;
    const int c0 = 16 * seg, h0 = 2 * i2;
    // ---- dz for the 2 x 2 block (rows h0, h0+1; columns c0 + 2 jb, + 1), 8 channels
    {
      const int J = 8 * seg + jb;
      const bool hasj = J + 1 < p.Q2, hasi = i2 + 1 < p.P2;
      const uint2 k00 = kn[0], k01 = kn[1], k10 = kn[2], k11 = kn[3];
      float d00[8], d01[8], d10[8], d11[8];
      unpack8(dn[0], d00);
      unpack8(dn[1], d01);
      unpack8(dn[2], d10);
      unpack8(dn[3], d11);
      if (item + istep < p.nitems) load_windows(item + istep, kn, dn);
      if (!hasj) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { d01[e] = 0.f; d11[e] = 0.f; }
      }
      if (!hasi) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { d10[e] = 0.f; d11[e] = 0.f; }
      }
      float dz[4][8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const unsigned sft = 8 * (e & 3);
        const unsigned a00 = ((e < 4 ? k00.x : k00.y) >> sft) & 0xff, a01 = ((e < 4 ? k01.x : k01.y) >> sft) & 0xff;
        const unsigned a10 = ((e < 4 ? k10.x : k10.y) >> sft) & 0xff, a11 = ((e < 4 ? k11.x : k11.y) >> sft) & 0xff;
        dz[0][e] = a00 == 4 ? d00[e] : 0.f;
        dz[1][e] = (a00 == 5 ? d00[e] : 0.f) + (a01 == 3 ? d01[e] : 0.f);
        dz[2][e] = (a00 == 7 ? d00[e] : 0.f) + (a10 == 1 ? d10[e] : 0.f);
        dz[3][e] = ((a00 == 8 ? d00[e] : 0.f) + (a01 == 6 ? d01[e] : 0.f)) + ((a10 == 2 ? d10[e] : 0.f) + (a11 == 0 ? d11[e] : 0.f));
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int row = 16 * (t >> 1) + 2 * jb + (t & 1);
        *reinterpret_cast<uint4*>(img + row * 128 + ((g ^ (row & 7)) << 4)) = pack8(dz[t]);
      }
    }
    // ---- patch fragments straight from the padded image (L1-resident: neighbouring lanes overlap)
    bf16x8 pf[2][2];   // [ks][tt]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int r = pr_[tt] > 6 ? 6 : pr_[tt];
        const bf16_t* src = p.xb + ((size_t)(b * p.HP + 2 * (h0 + ks) + r) * p.WP + 2 * (c0 + 8 * fh) + (ps_[tt] & ~1));
        const u32x4 lo = *reinterpret_cast<const u32x4_a4*>(src);
        const u32x4 hi = *reinterpret_cast<const u32x4_a4*>(src + 8);
        u32x4 o;
        o[0] = __builtin_amdgcn_perm(lo[1], lo[0], sel);
        o[1] = __builtin_amdgcn_perm(lo[3], lo[2], sel);
        o[2] = __builtin_amdgcn_perm(hi[1], hi[0], sel);
        o[3] = __builtin_amdgcn_perm(hi[3], hi[2], sel);
        if (pr_[tt] == 7) {                                  // padded slots: slot 56 (r 7, s 0) is the constant 1
          const uint32_t one = ps_[tt] == 0 ? 0x3F803F80u : 0u;
          o[0] = one; o[1] = one; o[2] = one; o[3] = one;
        }
        pf[ks][tt] = __builtin_bit_cast(bf16x8, o);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this wave's image is complete (wave-private: no barrier)
    typedef s16x4 __attribute__((address_space(3))) * lds_v4;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[2];
#pragma unroll
      for (int jc = 0; jc < 2; ++jc) {
        const int ch0 = 32 * jc + tr_ch;
        const int r0 = 16 * ks + tr_row, r1 = r0 + 4;
        const unsigned char* a0 = img + r0 * 128 + (((ch0 >> 3) ^ (r0 & 7)) << 4) + (ch0 & 7) * 2;
        const unsigned char* a1 = img + r1 * 128 + (((ch0 >> 3) ^ (r1 & 7)) << 4) + (ch0 & 7) * 2;
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)a0);
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)a1);
        af[jc] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
      }
#pragma unroll
      for (int jc = 0; jc < 2; ++jc)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
          accz[jc][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[jc], pf[ks][tt], accz[jc][tt], 0, 0, 0);
      accg[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[ks][0], pf[ks][0], accg[0], 0, 0, 0);
      accg[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[ks][0], pf[ks][1], accg[1], 0, 0, 0);
      accg[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[ks][1], pf[ks][1], accg[2], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // reads done before the next item rewrites the image
  }
  // ---- sum the 4 waves in LDS, one partial block per workgroup (plain stores: deterministic, nothing to zero)
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);
  for (int k = tid; k < 7 * 16 * 64; k += 256) red[k] = 0.f;
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int bq = 0; bq < 2; ++bq)
#pragma unroll
      for (int e = 0; e < 16; ++e) atomicAdd(&red[((a * 2 + bq) * 16 + e) * 64 + lane], accz[a][bq][e]);
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int e = 0; e < 16; ++e) atomicAdd(&red[((4 + a) * 16 + e) * 64 + lane], accg[a][e]);
  __syncthreads();
  float* out = p.partial + (size_t)blockIdx.x * (7 * 16 * 64);
  for (int k = tid; k < 7 * 16 * 64; k += 256) out[k] = red[k];
}

// partial blocks -> totals (double): blockIdx.y sums every gridDim.y-th block, 8 loads in flight, one fp64 atomic per value
__global__ __launch_bounds__(256) void stemf_bwd_reduce_kernel(const float* __restrict__ partial, int nparts,
                                                               double* __restrict__ total) {
  const int k = blockIdx.x * 256 + threadIdx.x;       // < 7168
  double s = 0.0;
  for (int q0 = blockIdx.y; q0 < nparts; q0 += gridDim.y * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int q = q0 + u * gridDim.y;
      v[u] = q < nparts ? partial[(size_t)q * 7168 + k] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (double)v[u];
  }
  atomicAdd(total + k, s);
}

struct StembFin {
  const double* total;          // [7][16][64]
  const bf16_t* wp;             // [64][64]
  const float *gamma, *mean, *invstd;   // eval: mean = running_mean, invstd = running_var (with eps)
  float eps, count;
  int eval;
  float *dw, *dgamma, *dbeta;   // dw [64][49]
  int acc_dw, acc_bn;
};

__global__ __launch_bounds__(256) void stemf_bwd_math_kernel(const StembFin a) {
  __shared__ float Z[64][65], G[64][65], Wb[64][65];
  __shared__ float K1[64], K2[64], K3[64];
  const int tid = threadIdx.x;
  for (int k = tid; k < 7168; k += 256) {
    const int T = k >> 10, e = (k >> 6) & 15, l = k & 63;
    const int row = (e & 3) + 8 * (e >> 2) + 4 * (l >> 5), col = l & 31;
    const float v = (float)a.total[k];
    if (T < 4) Z[32 * (T >> 1) + row][32 * (T & 1) + col] = v;
    else if (T == 4) G[row][col] = v;
    else if (T == 5) { G[row][32 + col] = v; G[32 + col][row] = v; }
    else G[32 + row][32 + col] = v;
  }
  for (int k = tid; k < 4096; k += 256) Wb[k >> 6][k & 63] = (float)a.wp[k];
  __syncthreads();
  const int c_lo = blockIdx.x * 8;                    // this block's 8 channels
  if (tid < 8) {
    const int c = c_lo + tid;
    double sy = 0.0;
    for (int t = 0; t < 56; ++t) sy += (double)Wb[c][t] * (double)Z[c][t];
    const double s1 = (double)Z[c][56];
    const float g = a.gamma ? a.gamma[c] : 1.f;
    const float mu = a.mean[c];
    const float is = a.eval ? 1.f / sqrtf(a.invstd[c] + a.eps) : a.invstd[c];
    const float sum_dz = (float)s1, sum_dzx = (float)((sy - (double)mu * s1) * (double)is);
    if (a.dgamma) a.dgamma[c] = a.acc_bn ? a.dgamma[c] + sum_dzx : sum_dzx;
    if (a.dbeta) a.dbeta[c] = a.acc_bn ? a.dbeta[c] + sum_dz : sum_dz;
    const float k1 = g * is;
    const float k2 = a.eval ? 0.f : -g * is * is * sum_dzx / a.count;
    const float k3 = a.eval ? 0.f : -g * is * sum_dz / a.count - k2 * mu;
    K1[c] = k1; K2[c] = k2; K3[c] = k3;
  }
  __syncthreads();
  for (int o = c_lo * 49 + tid; o < (c_lo + 8) * 49; o += 256) {
    const int c = o / 49, rs = o - c * 49, r = rs / 7, s = rs - r * 7, t = r * 8 + s;
    float wg = 0.f;
    for (int u = 0; u < 56; ++u) wg = fmaf(Wb[c][u], G[u][t], wg);
    const float v = K1[c] * Z[c][t] + K2[c] * wg + K3[c] * G[56][t];
    a.dw[o] = a.acc_dw ? a.dw[o] + v : v;
  }
}

extern "C" {

// 1 iff the fused stem serves this geometry: 1 input channel, 7x7 / stride 2 / pad 3, 64 output channels, pooled 3x3/2/1
int mpr_stemf_supported(int H, int W, int K) {
  return K == 64 && H >= 16 && W >= 32 && H % 4 == 0 && W % 32 == 0;
}
static inline int stemf_nt(int Q) { return (Q / 2 + 14) / 15; }

// x fp32 [B,H,W] (one channel) -> xb bf16 [B][H+6][W+8] (zero padded);  w fp32 [64][1][7][7] -> wp bf16 [64][64]
int mpr_stemf_prep(const float* x, const float* w, void* xb, void* wp, int B, int H, int W, void* stream) {
  MPR_REQUIRE(x && w && xb && wp, "mpr_stemf_prep: null pointer");
  MPR_REQUIRE(mpr_stemf_supported(H, W, 64), "mpr_stemf_prep: unsupported geometry %dx%d", H, W);
  MPR_REQUIRE((long long)B * (H + 6) * (W + 8) < (1ll << 31), "mpr_stemf_prep: tensor exceeds 2^31 elements");
  const long long nvec = (long long)B * (H + 6) * ((W + 8) / 8);
  const int grid = (int)((nvec + 255) / 256 < 4096 ? (nvec + 255) / 256 : 4096);
  stemf_prep_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, w, (bf16_t*)xb, (bf16_t*)wp, B, H, W);
  MPR_LAUNCH_CHECK("stemf_prep_kernel");
  return MPR_OK;
}

static void stemf_geom(StemfP& p, int B, int H, int W) {
  p.B = B; p.H = H; p.W = W; p.P = H / 2; p.Q = W / 2; p.HP = H + 6; p.WP = W + 8;
  p.P2 = p.P / 2; p.Q2 = p.Q / 2; p.NT = stemf_nt(p.Q); p.nitems = B * p.NT;
}

// pass A: per-channel sum / sum of squares of the bf16-rounded conv output, added into stats[nslices][2][64]
int mpr_stemf_stats(const void* xb, const void* wp, float* stats, int nslices, int prezeroed, int B, int H, int W,
                    void* stream) {
  MPR_REQUIRE(xb && wp && stats && nslices > 0, "mpr_stemf_stats: bad arguments");
  MPR_REQUIRE(mpr_stemf_supported(H, W, 64), "mpr_stemf_stats: unsupported geometry %dx%d", H, W);
  StemfP p = {};
  p.xb = (const bf16_t*)xb; p.wp = (const bf16_t*)wp; p.stats = stats; p.stat_slices = nslices;
  stemf_geom(p, B, H, W);
  hipStream_t st = (hipStream_t)stream;
  if (!prezeroed) MPR_HIP(hipMemsetAsync(stats, 0, sizeof(float) * 2 * 64 * (size_t)nslices, st));
  stemf_fwd_kernel<0><<<ceil_div(p.nitems, 4), 256, 0, st>>>(p);
  MPR_LAUNCH_CHECK("stemf_fwd_kernel<0>");
  return MPR_OK;
}

// pass B: pooled = maxpool3x3/2(relu(bn(conv))).  slices != NULL (train): finalize the statistics here (scale / shift /
// mean / invstd are OUTPUTS, running statistics updated); slices == NULL (eval): scale / shift are inputs.
// idx may be NULL when no backward pass will follow.
int mpr_stemf_pool(const void* xb, const void* wp, const float* slices, int nsl, long long count, const float* gamma,
                   const float* beta, float* running_mean, float* running_var, float momentum, float eps, float* scale,
                   float* shift, float* mean, float* invstd, void* pooled, void* idx, int B, int H, int W, void* stream) {
  MPR_REQUIRE(xb && wp && pooled && scale && shift, "mpr_stemf_pool: null pointer");
  MPR_REQUIRE(!slices || (nsl > 0 && gamma && beta && mean && invstd), "mpr_stemf_pool: train mode needs gamma, beta, mean, invstd");
  MPR_REQUIRE(mpr_stemf_supported(H, W, 64), "mpr_stemf_pool: unsupported geometry %dx%d", H, W);
  MPR_REQUIRE((long long)B * (H / 2) * (W / 2) * 64 < (1ll << 31), "mpr_stemf_pool: tensor exceeds 2^31 elements");
  StemfP p = {};
  p.xb = (const bf16_t*)xb; p.wp = (const bf16_t*)wp; p.slices = slices; p.nsl = nsl;
  p.fin.count = (float)count; p.fin.momentum = momentum; p.fin.eps = eps;
  p.fin.gamma = gamma; p.fin.beta = beta; p.fin.running_mean = running_mean; p.fin.running_var = running_var;
  p.fin.scale = scale; p.fin.shift = shift; p.fin.mean_out = mean; p.fin.invstd_out = invstd;
  p.pooled = (bf16_t*)pooled; p.idx = (unsigned char*)idx;
  stemf_geom(p, B, H, W);
  stemf_fwd_kernel<1><<<ceil_div(p.nitems, 4), 256, 0, (hipStream_t)stream>>>(p);
  MPR_LAUNCH_CHECK("stemf_fwd_kernel<1>");
  return MPR_OK;
}

// workgroups of mpr_stemf_bwd == partial blocks (7168 floats each) it writes
int mpr_stemf_bwd_parts(int B, int H, int W) {
  const int items = B * (H / 4) * (W / 32);
  const int g = ceil_div(items, 4);
  return g < 512 ? g : 512;
}

// backward, one pass: partial[parts][7168] (see StembP) from the pooled gradient, the arg-max codes and xb
int mpr_stemf_bwd(const void* xb, const void* dpooled, const void* idx, float* partial, int B, int H, int W,
                  void* stream) {
  MPR_REQUIRE(xb && dpooled && idx && partial, "mpr_stemf_bwd: null pointer");
  MPR_REQUIRE(mpr_stemf_supported(H, W, 64), "mpr_stemf_bwd: unsupported geometry %dx%d", H, W);
  StembP p = {};
  p.xb = (const bf16_t*)xb; p.dp = (const bf16_t*)dpooled; p.idx = (const unsigned char*)idx;
  p.partial = partial;
  p.B = B; p.H = H; p.W = W; p.P = H / 2; p.Q = W / 2; p.HP = H + 6; p.WP = W + 8; p.P2 = p.P / 2; p.Q2 = p.Q / 2;
  p.NSEG = p.Q / 16; p.nitems = B * p.P2 * p.NSEG;
  stemf_bwd_kernel<<<mpr_stemf_bwd_parts(B, H, W), 256, 0, (hipStream_t)stream>>>(p);
  MPR_LAUNCH_CHECK("stemf_bwd_kernel");
  return MPR_OK;
}

// partial blocks -> dW [64][1][7][7], dgamma, dbeta.  scratch: 7168 doubles.  eval != 0: BatchNorm ran on the running
// statistics (mean = running_mean, invstd_or_var = running_var): dx = scale * dz.
int mpr_stemf_bwd_finalize(const float* partial, int nparts, void* scratch, const void* wp, long long count,
                           const float* gamma, const float* mean, const float* invstd_or_var, float eps, int eval,
                           float* dw, int accumulate_dw, float* dgamma, float* dbeta, int accumulate_bn, void* stream) {
  MPR_REQUIRE(partial && nparts > 0 && scratch && wp && mean && invstd_or_var && dw, "mpr_stemf_bwd_finalize: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  MPR_HIP(hipMemsetAsync(scratch, 0, 7168 * sizeof(double), st));
  stemf_bwd_reduce_kernel<<<dim3(28, 16), 256, 0, st>>>(partial, nparts, (double*)scratch);
  MPR_LAUNCH_CHECK("stemf_bwd_reduce_kernel");
  StembFin a = {};
  a.total = (const double*)scratch; a.wp = (const bf16_t*)wp; a.gamma = gamma; a.mean = mean; a.invstd = invstd_or_var;
  a.eps = eps; a.count = (float)count; a.eval = eval; a.dw = dw; a.dgamma = dgamma; a.dbeta = dbeta;
  a.acc_dw = accumulate_dw; a.acc_bn = accumulate_bn;
  stemf_bwd_math_kernel<<<8, 256, 0, st>>>(a);
  MPR_LAUNCH_CHECK("stemf_bwd_math_kernel");
  return MPR_OK;
}

}  // extern "C"
