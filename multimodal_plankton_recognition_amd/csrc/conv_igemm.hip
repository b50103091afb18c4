// Implicit-GEMM convolution, forward and data-gradient, bf16 NHWC in / fp32 MFMA accumulate.
//
// Replaces the cuDNN/MIOpen calls behind timm's ResNet convs (reference call site
// src/image_encoder.py:24) and nn.Conv1d in ProfileCNN (src/profile_encoder.py:125-128,167,187)
// -- a 1-D conv is the H == 1 case.
//
// GEMM view (rows = pixels of the tensor being produced, cols = its channels):
//   forward : y[m = (b,p,q)][n = k]  = sum_{r,s,c} x [b, p*st-pad+r, q*st-pad+s, c] * Wf[k][(r,s,c)]
//   dgrad   : dx[m = (b,h,w)][n = c] = sum_{r,s,k} dy[b, (h+pad-r)/st, (w+pad-s)/st, k] * Wd[c][(r,s,k)]
// A rows are gathered on the fly (zero for padding taps / stride holes), B is a pre-packed
// [N][Kg] bf16 panel (K contiguous, zero padded to 64).  One workgroup = 4 waves, each wave owns a
// 64x64 output block as 2x2 v_mfma_f32_32x32x16_bf16 tiles; WG tile = (64*WM) x (64*WN).
// K advances in 64-element (128 B) chunks: global -> VGPR (issued before the MFMAs of the current
// chunk) -> XOR-swizzled LDS (written after them), double buffered, one barrier per chunk.
// The MFMA is issued "swapped" (weights as the A operand) so that a lane ends up with 4 consecutive
// channels of one pixel; the epilogue stages the fp32 tile in LDS, then writes whole 16-B channel
// groups, fusing (a) per-channel sum / sum-of-squares partials for train-mode BatchNorm and (b) an
// optional residual add (the skip-connection gradient in dgrad).
#include "common.h"

struct ConvGemmParams {
  const bf16_t* src;   // fwd: x [B,sH,sW,sC];  dgrad: dy [B,sH,sW,sC]
  const bf16_t* wpk;   // [Npad][Kgpad]
  bf16_t* dst;         // [M][Nout]
  const bf16_t* add;   // optional [M][Nout]
  float* stats;        // optional [gridM][2][Nout]
  int stat_slices;     // > 0: the tiles ADD their sums into stats[tile % stat_slices] (zeroed by the host) instead of a row each
  int sH, sW, sC;
  unsigned src_bytes, wpk_bytes;   // extents for the buffer descriptors of the LDS-DMA kernel
  unsigned rowpat;                 // sum_r 1 << (r*S): one bit per filter row (tap masks of the LDS-DMA kernel)
  int M, Nout, Kg, Kgpad, nk, ntn;
  int R, S, sh, sw, ph, pw;
  FastDiv div_pq, div_q;   // row m -> (b, p, q)
  int Pm, Qm;
  unsigned long long* stamps; // timing experiments: per-workgroup phase time stamps (100 MHz), or NULL
  int dbg;                   // timing experiments: bit 2 / 3 = do not ISSUE the activation / weight DMA at all
  int par_rows, par_valid;   // stride-2 dgrad parity classes (LDS-DMA kernel): padded / real rows per class, 0 = off
  const bf16_t* add_even;    // parity classes only: [B, Pm, Qm, Nout] added at the pixels (2 h2, 2 w2) -- the data gradient
                             // of a 1x1 / stride-2 shortcut convolution, which is zero everywhere else
  // BatchNorm-backward fusion (template BNB, data gradient of the LDS-DMA kernel): the result is the gradient w.r.t. a block
  // output relu(bn2(x2) + identity): the epilogue applies the ReLU mask (mask_y > 0) and adds sum dz, sum dz * xhat of bn2
  // into `stats` (slice rows) -- as conv_win.hip does for the stride-1 data gradients
  const bf16_t *bnb_mask_y, *bnb_x;   // [M][Nout]
  const float *bnb_mean, *bnb_invstd; // [Nout]
};

__device__ __forceinline__ int swz_off(int row, int chunk) {   // byte offset in a [rows][128 B] tile
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

template <int WM, int WN, bool DGRAD>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvGemmParams p) {
  constexpr int BM = WM * 64, BN = WN * 64;
  constexpr int A_IT = BM / 32, B_IT = BN / 32;
  constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
  constexpr int CS_STRIDE = BN * 4 + 16;   // fp32 epilogue tile row stride (bytes)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = bid / p.ntn, nt = bid - mt * p.ntn;
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- per-thread gather coordinates (fixed over K)
  const int c8 = tid & 7, r0 = tid >> 3;
  int rbase[A_IT], rh[A_IT], rw[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    int m = m0 + r0 + 32 * i;
    if (m < p.M) {
      uint32_t b = fdiv(m, p.div_pq);
      uint32_t rem = m - b * (uint32_t)(p.Pm * p.Qm);
      uint32_t pp = fdiv(rem, p.div_q);
      uint32_t qq = rem - pp * p.Qm;
      rbase[i] = b * p.sH * p.sW;
      if (!DGRAD) { rh[i] = (int)pp * p.sh - p.ph; rw[i] = (int)qq * p.sw - p.pw; }
      else        { rh[i] = (int)pp + p.ph;        rw[i] = (int)qq + p.pw; }
    } else {
      rbase[i] = 0; rh[i] = -(1 << 20); rw[i] = -(1 << 20);
    }
  }
  // tap tracking for this thread's 8-channel segment: k = kc*64 + c8*8 -> (r, s, c)
  int tc, tr, ts;
  {
    int k = c8 * 8;
    int t = k / p.sC;
    tc = k - t * p.sC;
    tr = t / p.S;
    ts = t - tr * p.S;
  }
  const bf16_t* wrow = p.wpk + (size_t)(n0 + r0) * p.Kgpad + c8 * 8;

  uint4 areg[A_IT], breg[B_IT];
  auto load_chunk = [&](int kc) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      bool ok;
      int off;
      if (!DGRAD) {
        int ih = rh[i] + tr, iw = rw[i] + ts;
        ok = (unsigned)ih < (unsigned)p.sH && (unsigned)iw < (unsigned)p.sW && tr < p.R;
        off = (rbase[i] + ih * p.sW + iw) * p.sC + tc;
      } else {
        int oh = rh[i] - tr, ow = rw[i] - ts;
        ok = oh >= 0 && ow >= 0 && tr < p.R;
        if (p.sh == 2) { ok = ok && !(oh & 1); oh >>= 1; }
        if (p.sw == 2) { ok = ok && !(ow & 1); ow >>= 1; }
        ok = ok && oh < p.sH && ow < p.sW;
        off = (rbase[i] + oh * p.sW + ow) * p.sC + tc;
      }
      areg[i] = ok ? *reinterpret_cast<const uint4*>(p.src + off) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i)
      breg[i] = *reinterpret_cast<const uint4*>(wrow + (size_t)(32 * i) * p.Kgpad + kc * 64);
    // advance the tap by one 64-element chunk
    tc += 64;
    while (tc >= p.sC) { tc -= p.sC; if (++ts == p.S) { ts = 0; ++tr; } }
  };
  auto store_chunk = [&](int buf) {
    unsigned char* a = smem + buf * STAGE;
    unsigned char* b = a + A_BYTES;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) *reinterpret_cast<uint4*>(a + swz_off(r0 + 32 * i, c8)) = areg[i];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) *reinterpret_cast<uint4*>(b + swz_off(r0 + 32 * i, c8)) = breg[i];
  };

  f32x16 acc[2][2];   // [ntile][mtile], D'[n][m]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  load_chunk(0);
  store_chunk(0);
  __syncthreads();

  const int frow = lane & 31, fh = lane >> 5;
  for (int kc = 0; kc < p.nk; ++kc) {
    const int cur = kc & 1;
    if (kc + 1 < p.nk) load_chunk(kc + 1);
    const unsigned char* a = smem + cur * STAGE;
    const unsigned char* b = a + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        af[t] = *reinterpret_cast<const bf16x8*>(a + swz_off(wm * 64 + t * 32 + frow, ks * 2 + fh));
        bfr[t] = *reinterpret_cast<const bf16x8*>(b + swz_off(wn * 64 + t * 32 + frow, ks * 2 + fh));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[i], af[j], acc[i][j], 0, 0, 0);
    }
    if (kc + 1 < p.nk) store_chunk(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: fp32 tile -> LDS -> 16-B channel groups (+ residual, + BN partial sums)
  // lane holds, for pixel m = lane&31, channels n = 8g + 4*(lane>>5) + (0..3) in acc regs 4g..4g+3
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = wm * 64 + j * 32 + frow;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = wn * 64 + i * 32 + 8 * g + 4 * fh;
        float4 v = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
        *reinterpret_cast<float4*>(smem + row * CS_STRIDE + col * 4) = v;
      }
    }
  __syncthreads();

  constexpr int CPR = BN / 8;          // 16-B output chunks per tile row
  constexpr int RPP = 256 / CPR;       // rows per pass
  const int ch = tid % CPR, rr = tid / CPR;
  const int ncol = n0 + ch * 8;
  const bool col_ok = ncol < p.Nout;
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
#pragma unroll 4
  for (int r = rr; r < BM; r += RPP) {
    const int m = m0 + r;
    if (m < p.M && col_ok) {
      const float4 lo = *reinterpret_cast<const float4*>(smem + r * CS_STRIDE + ch * 32);
      const float4 hi = *reinterpret_cast<const float4*>(smem + r * CS_STRIDE + ch * 32 + 16);
      float f[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      const size_t o = (size_t)m * p.Nout + ncol;
      if (p.add) {
        float g[8];
        unpack8(*reinterpret_cast<const uint4*>(p.add + o), g);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] += g[e];
      }
      const uint4 pk = pack8(f);
      *reinterpret_cast<uint4*>(p.dst + o) = pk;
      if (p.stats) {
        float q[8];
        unpack8(pk, q);   // statistics of the values BatchNorm will actually read
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] += q[e]; s2[e] += q[e] * q[e]; }
      }
    }
  }
  if (p.stats) {
    // threads with equal `ch` are CPR lanes apart inside a wave; fold them, then fold the 4 waves
#pragma unroll
    for (int o = CPR; o < 64; o <<= 1)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s1[e] += __shfl_xor(s1[e], o, 64);
        s2[e] += __shfl_xor(s2[e], o, 64);
      }
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);   // [4 waves][CPR][16]
    if (lane < CPR) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(wid * CPR + lane) * 16 + e] = s1[e];
        red[(wid * CPR + lane) * 16 + 8 + e] = s2[e];
      }
    }
    __syncthreads();
    if (tid < CPR * 16) {
      const int c = tid >> 4, e = tid & 15;
      const float v = red[(0 * CPR + c) * 16 + e] + red[(1 * CPR + c) * 16 + e] +
                      red[(2 * CPR + c) * 16 + e] + red[(3 * CPR + c) * 16 + e];
      const int n = n0 + c * 8 + (e & 7);
      if (n < p.Nout) {
        if (p.stat_slices > 0) atomicAdd(&p.stats[((size_t)(mt % p.stat_slices) * 2 + (e >> 3)) * p.Nout + n], v);
        else p.stats[((size_t)mt * 2 + (e >> 3)) * p.Nout + n] = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA variant (source channels a multiple of 64, i.e. every K chunk is one tap x 64 channels).
// Operand tiles go global -> LDS directly (buffer_load_dwordx4 ... lds: no VGPR staging, no ds_write), into a
// ring of STAGES buffers; a chunk is issued STAGES-1 chunks ahead and retired by a COUNTED s_waitcnt vmcnt
// followed by one raw s_barrier per chunk (a __syncthreads() would drain the DMA queue).  The LDS image
// of a DMA instruction is lane-linear (64 lanes x 16 B = 8 rows x 128 B), so the bank swizzle is applied
// on the SOURCE side: lane (row, physical chunk) fetches logical chunk = physical ^ ((row >> 1) & 7), and
// the fragment reads use the same XOR.  Padding taps / stride holes / the M tail are given an out-of-range
// buffer offset: the descriptor's range check turns those lanes into zero writes.

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, and on gfx9 global STORES count
// in vmcnt until acknowledged: an epilogue that syncs after its stores pays a full store round trip per barrier.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int V>
struct IgInt { static constexpr int value = V; };
template <int I, int N, class F>
__device__ __forceinline__ void ig_unroll(F&& f) {      // f(IgInt<I>) ... f(IgInt<N-1>)
  if constexpr (I < N) {
    f(IgInt<I>{});
    ig_unroll<I + 1, N>(f);
  }
}

// BK = K depth of a ring stage: 64 (128-B tile rows) or 32 (64-B rows: half the bytes per stage, so a deeper ring /
// more workgroups per CU fit the 160 KB -- the loop is bound by the LATENCY of the operand stream, i.e. by the bytes
// in flight per CU).  Swizzle keys: 128-B rows (row >> 1) & 7 over 8 chunks, 64-B rows (row >> 2) & 3
// over 4 chunks; both make every ds_read_b128 lane group hit 16 distinct 16-B bank slots.
// M16 (round 3, the default at BK = 64): the wave's 64 x 64 block as 4 x 4 v_mfma_f32_16x16x32_bf16 sub-tiles, fragment pairs
// in four slots walked as a snake over the quadrants -- conv_win.hip's loop (same fragments, same LDS traffic, bit-identical
// sums); the chip holds a higher clock on that shape.
template <int WM, int WN, int STAGES, bool DGRAD, int BK = 64, bool BNB = false, bool M16 = false>
__global__ __launch_bounds__(64 * WM * WN) void conv_igemm_dma_kernel(const ConvGemmParams p) {
  static_assert(!M16 || BK == 64, "the 16x16x32 form is written for 128-byte rows");
#if defined(__HIP_DEVICE_COMPILE__)   // the buffer-descriptor builtins do not exist in the host pass of hipcc
  constexpr int NW = WM * WN, T = 64 * NW;
  constexpr int BM = WM * 64, BN = WN * 64;
  constexpr int ROWB = BK * 2, CPRW = ROWB / 16, RPI = 64 / CPRW;   // row bytes, 16-B chunks per row, rows per DMA instr
  constexpr int KS = BK / 16;                                       // MFMA k-steps per stage
  constexpr int A_IT = BM / RPI / NW, B_IT = BN / RPI / NW, IPC = A_IT + B_IT;
  static_assert(BK == 32 || (A_IT % 2 == 0 && B_IT % 2 == 0), "instruction parity must follow the local index");
  static_assert(A_IT >= 1 && B_IT >= 1, "tile too small for the DMA instruction shape");
  constexpr int A_BYTES = BM * ROWB, STAGE = (BM + BN) * ROWB;
  constexpr int CS_STRIDE = BN * 4 + 16;
  // (the host sizes the dynamic LDS as max(ring, one 64-row slab of the epilogue tile))
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  int mt = bid / p.ntn;
  const int nt = bid - mt * p.ntn;
  // parity classes: the four classes of one pixel range are NEIGHBOURS in launch order (they read the same dy rows: one
  // HBM fetch per XCD instead of four) -- which also spreads the heavy class (4 taps of a 3x3 filter) over the whole launch
  // instead of leaving it for the tail
  if (DGRAD && p.par_rows > 0) mt = (mt & 3) * (p.par_rows / BM) + (mt >> 2);
  const int m0 = mt * BM, n0 = nt * BN;
#define MPR_STAMP(k) do { if (p.stamps && tid == 0) p.stamps[(size_t)blockIdx.x * 4 + (k)] = wall_clock64(); } while (0)
  MPR_STAMP(0);

  // DMA geometry: instruction I covers tile rows 8I..8I+7; lane -> (row 8I + lane/8, physical chunk lane%8)
  // Source addressing is linear in the tap: byte offset = row part (VGPR, fixed) + tap part (SGPR, per chunk);
  // validity of each of the <= 32 taps for a row is one bit of a per-row mask, and invalid lanes get an
  // out-of-range buffer offset, for which the hardware range check makes the DMA write zeros.
  //   BK = 32: instruction I covers rows 16I..16I+15; lane -> (row 16I + lane/4, physical chunk lane%4), key lane>>4
  const int lrow = BK == 64 ? lane >> 3 : lane >> 2;
  const int lc_even = BK == 64 ? (lane & 7) ^ (lane >> 4)         // logical chunk for even I; odd I: ^ 4
                               : (lane & 3) ^ (lane >> 4);        // (no instruction parity with 64-B rows)
  constexpr int PARITY_XOR = BK == 64 ? 4 : 0;
  //   fwd  : pix = rbase + (rh + r)*sW + (rw + s)                      -> ch =  2*sC*sW,      cw =  2*sC
  //   dgrad: pix = rbase + ((rh - r)/sh)*sW + (rw - s)/sw  (when valid) -> ch = -2*sC*sW/sh,  cw = -2*sC/sw
  const int tstep_h = DGRAD ? -(2 * p.sC * p.sW) / p.sh : 2 * p.sC * p.sW;
  const int tstep_w = DGRAD ? -(2 * p.sC) / p.sw : 2 * p.sC;
  // Stride-2 data gradient, parity classes (p.par_rows > 0): only taps with r == (h + ph) mod 2 and
  // s == (w + pw) mod 2 reach a gradient pixel, so the rows are regrouped by (h mod 2, w mod 2) -- a tile
  // then belongs to ONE class and simply skips the other taps (1/4 of the chunks on average instead of
  // issuing all of them with 3/4 of the rows zero-filled).  Row m -> class m / par_rows, and
  // (b, h/2, w/2) = decode(m mod par_rows) with Pm = H/2, Qm = W/2.
  const bool PAR = DGRAD && p.par_rows > 0;
  const int cls = PAR ? m0 / p.par_rows : 0;
  const int par_h = cls >> 1, par_w = cls & 1;
  const int mbase = cls * p.par_rows;
  uint32_t rowpat_even = 0, rowpat_odd = 0;      // bit r * S of the filter rows with r even / odd (wave-uniform)
  for (int r = 0; r < p.R && r * p.S < 32; ++r) {
    if (r & 1) rowpat_odd |= 1u << (r * p.S);
    else rowpat_even |= 1u << (r * p.S);
  }
  uint32_t voff[A_IT], vmask[A_IT];
#pragma unroll
  for (int j = 0; j < A_IT; ++j) {
    const int m = m0 + RPI * (wid * A_IT + j) + lrow - mbase;
    voff[j] = 0;
    vmask[j] = 0;
    if (m < (PAR ? p.par_valid : p.M)) {
      const uint32_t b = fdiv(m, p.div_pq);
      const uint32_t rem = m - b * (uint32_t)(p.Pm * p.Qm);
      uint32_t pp = fdiv(rem, p.div_q);
      uint32_t qq = rem - pp * p.Qm;
      if (PAR) { pp = 2 * pp + par_h; qq = 2 * qq + par_w; }
      int rh, rw;
      if (!DGRAD) { rh = (int)pp * p.sh - p.ph; rw = (int)qq * p.sw - p.pw; }
      else        { rh = (int)pp + p.ph;        rw = (int)qq + p.pw; }
      const int lc = lc_even ^ ((j & 1) ? PARITY_XOR : 0);
      voff[j] = (uint32_t)(b * p.sH * p.sW) * (uint32_t)(2 * p.sC) + (uint32_t)(rh * (DGRAD ? -tstep_h : tstep_h)) +
                (uint32_t)(rw * (DGRAD ? -tstep_w : tstep_w)) + (uint32_t)(lc * 16);
      // tap (r, s) is valid iff row-tap r and column-tap s are both valid (the conditions are per axis):
      // S column bits are built once and replicated for every valid row tap
      uint32_t mk = 0;
      if (!DGRAD) {
        // valid taps are an index range per axis -> closed-form masks (no loops in the prologue)
        const int slo = max(0, -rw), shi = min(p.S, p.sW - rw);
        const int rlo = max(0, -rh), rhi = min(p.R, p.sH - rh);
        if (shi > slo && rhi > rlo) {
          const uint32_t cm = ((1u << shi) - 1u) & ~((1u << slo) - 1u);
          mk = cm * (p.rowpat & ((1u << (rhi * p.S)) - 1u) & ~((1u << (rlo * p.S)) - 1u));
        }
      } else if (p.sh == 1 && p.sw == 1) {
        // 0 <= rh - r < sH  <=>  r in [rh - sH + 1, rh]
        const int slo = max(0, rw - p.sW + 1), shi = min(p.S, rw + 1);
        const int rlo = max(0, rh - p.sH + 1), rhi = min(p.R, rh + 1);
        if (shi > slo && rhi > rlo) {
          const uint32_t cm = ((1u << shi) - 1u) & ~((1u << slo) - 1u);
          mk = cm * (p.rowpat & ((1u << (rhi * p.S)) - 1u) & ~((1u << (rlo * p.S)) - 1u));
        }
      } else if (p.sh == 2 && p.sw == 2) {
        // stride 2, closed form as well (round 3: the loops below cost ~50 VALU instructions per row, 8 rows per lane in the
        // prologue of a workgroup that lives 12 us): tap s reaches gradient column (rw - s) / 2 iff rw - s is even, >= 0 and
        // < 2 sW -- a parity pattern AND an index range per axis
        const int slo = max(0, rw - 2 * (p.sW - 1)), shi = min(p.S, rw + 1);
        const int rlo = max(0, rh - 2 * (p.sH - 1)), rhi = min(p.R, rh + 1);
        if (shi > slo && rhi > rlo) {
          const uint32_t cm = ((1u << shi) - 1u) & ~((1u << slo) - 1u) & ((rw & 1) ? 0xAAAAAAAAu : 0x55555555u);
          const uint32_t rows = ((rh & 1) ? rowpat_odd : rowpat_even) & ((1u << (rhi * p.S)) - 1u) & ~((1u << (rlo * p.S)) - 1u);
          mk = cm * rows;
        }
      } else {
        uint32_t cmask = 0;
        for (int s2 = 0; s2 < p.S; ++s2) {
          int ow = rw - s2;
          bool ok = ow >= 0;
          if (p.sw == 2) { ok = ok && !(ow & 1); ow >>= 1; }
          ok = ok && ow < p.sW;
          cmask |= (ok ? 1u : 0u) << s2;
        }
        for (int r = 0; r < p.R; ++r) {
          int oh = rh - r;
          bool ok = oh >= 0;
          if (p.sh == 2) { ok = ok && !(oh & 1); oh >>= 1; }
          ok = ok && oh < p.sH;
          mk |= (ok ? cmask : 0u) << (r * p.S);
        }
      }
      vmask[j] = mk;
    }
  }
  uint32_t boff[B_IT];
#pragma unroll
  for (int j = 0; j < B_IT; ++j)
    boff[j] = ((uint32_t)(n0 + RPI * (wid * B_IT + j) + lrow) * (uint32_t)p.Kgpad +
               (lc_even ^ ((j & 1) ? PARITY_XOR : 0)) * 8) * 2;
  const __amdgpu_buffer_rsrc_t rs_a =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, p.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.wpk, 0, p.wpk_bytes, 0x00020000);

  const int ncb = p.sC / BK;     // BK-channel blocks per tap
  const int ntaps = p.R * p.S;
  // parity classes walk only the taps r = tr0, tr0+2, ... / s = ts0, ts0+2, ... of their class
  const int tr0 = PAR ? ((par_h + p.ph) & 1) : 0, ts0 = PAR ? ((par_w + p.pw) & 1) : 0;
  const int tstep = PAR ? 2 : 1;
  const int nk = PAR ? ((p.R - tr0 + 1) >> 1) * ((p.S - ts0 + 1) >> 1) * ncb : p.nk * (64 / BK);
  int tr = tr0, ts = ts0, cb = 0, tap = tr0 * p.S + ts0;    // wave-uniform tap state of the NEXT chunk to issue
  auto issue_chunk = [&](int kc, int buf) {
    unsigned char* sa = smem + buf * STAGE;
    unsigned char* sb = sa + A_BYTES;
    const uint32_t toff = (uint32_t)(tr * tstep_h + ts * tstep_w + cb * ROWB);
    const uint32_t bit = tap < ntaps ? (1u << tap) : 0u;      // K padding chunks: every lane reads zeros
    if (PAR) kc = tap * ncb + cb;                             // weight chunk of this (tap, channel block)
    if (!(p.dbg & 4))
#pragma unroll
    for (int j = 0; j < A_IT; ++j) {
      const uint32_t v = (vmask[j] & bit) ? voff[j] + toff : 0xFFFFFFF0u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(sa + (wid * A_IT + j) * 1024),
                                               16, v, 0, 0, 0);
    }
    if (!(p.dbg & 8))
#pragma unroll
    for (int j = 0; j < B_IT; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (__attribute__((address_space(3))) void*)(sb + (wid * B_IT + j) * 1024),
                                               16, boff[j], kc * ROWB, 0, 0);
    if (++cb == ncb) {
      cb = 0;
      ts += tstep;
      if (ts >= p.S) { ts = ts0; tr += tstep; }
      tap = tr * p.S + ts;
    }
  };

  f32x16 acc[M16 ? 1 : 2][M16 ? 1 : 2];
  f32x4 acc16[M16 ? 4 : 1][M16 ? 4 : 1];      // [channel sub-tile of 16][pixel sub-tile of 16]
#pragma unroll
  for (int i = 0; i < (M16 ? 1 : 2); ++i)
#pragma unroll
    for (int j = 0; j < (M16 ? 1 : 2); ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll
  for (int i = 0; i < (M16 ? 4 : 1); ++i)
#pragma unroll
    for (int j = 0; j < (M16 ? 4 : 1); ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc16[i][j][e] = 0.f;

  int issued = 0;
  for (; issued < STAGES - 1 && issued < nk; ++issued) issue_chunk(issued, issued % STAGES);

  MPR_STAMP(1);
  const int frow = lane & 31, fh = lane >> 5;
  // fragment read offsets inside a stage: row*128 + (((2*ks + fh) ^ key) << 4), key = (row >> 1) & 7 = (frow >> 1) & 7
  //                                     (64-B rows: row*64 + (((2*ks + fh) ^ ((row >> 2) & 3)) << 4))
  auto frag_off = [](int row, int chunk) {
    return BK == 64 ? swz_off(row, chunk) : row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4);
  };
  uint32_t a_rd[2][KS], b_rd[2][KS];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      a_rd[t][ks] = frag_off(wm * 64 + t * 32 + frow, ks * 2 + fh);
      b_rd[t][ks] = A_BYTES + frag_off(wn * 64 + t * 32 + frow, ks * 2 + fh);
    }
  // 16x16x32: lane -> row lane % 16 of a 16-row sub-tile, 8-channel quarter lane / 16 of a 32-deep step; sub-tile t sits 16 rows =
  // 2048 B further with the same swizzle key
  const int fr16 = lane & 15, fq = lane >> 4;
  uint32_t a16[2], b16[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    a16[ks] = frag_off(wm * 64 + fr16, ks * 4 + fq);
    b16[ks] = A_BYTES + frag_off(wn * 64 + fr16, ks * 4 + fq);
  }
  for (int kc = 0; kc < nk; ++kc) {
    // retire chunk kc: everything but the (issued - kc - 1) younger chunks of THIS wave must have landed
    const int younger = issued - kc - 1;
    if (STAGES >= 3 && younger >= STAGES - 2) wait_vmcnt<(STAGES - 2) * IPC>();
    else if (STAGES >= 4 && younger == STAGES - 3) wait_vmcnt<(STAGES >= 4 ? (STAGES - 3) : 0) * IPC>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const unsigned char* a = smem + (kc % STAGES) * STAGE;
    // software-pipelined k-steps: the fragments of step ks+1 are in flight (their own registers) while the four
    // MFMAs of step ks run -- left to itself hipcc reuses one register set and exposes the LDS latency 4x per chunk.
    // The first reads go out BEFORE the next chunk's DMA is issued, so its address arithmetic hides their latency.
    if constexpr (M16) {
      bf16x8 S0[2], S1[2], S2[2], S3[2];
      auto ldA = [&](bf16x8* sl, int ia, int ks) {      // pixel sub-tiles 2 ia, 2 ia + 1
        sl[0] = *reinterpret_cast<const bf16x8*>(a + a16[ks] + (2 * ia) * 2048);
        sl[1] = *reinterpret_cast<const bf16x8*>(a + a16[ks] + (2 * ia + 1) * 2048);
      };
      auto ldB = [&](bf16x8* sl, int jb, int ks) {      // channel sub-tiles 2 jb, 2 jb + 1
        sl[0] = *reinterpret_cast<const bf16x8*>(a + b16[ks] + (2 * jb) * 2048);
        sl[1] = *reinterpret_cast<const bf16x8*>(a + b16[ks] + (2 * jb + 1) * 2048);
      };
      auto quad = [&](const bf16x8* sb, const bf16x8* sa, auto JB_, auto IA_) {
        constexpr int jb = decltype(JB_)::value, ia = decltype(IA_)::value;
#pragma unroll
        for (int dj = 0; dj < 2; ++dj)
#pragma unroll
          for (int di = 0; di < 2; ++di)
            acc16[2 * jb + dj][2 * ia + di] =
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(sb[dj], sa[di], acc16[2 * jb + dj][2 * ia + di], 0, 0, 0);
      };
      ldA(S0, 0, 0); ldB(S1, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (issued < nk) { issue_chunk(issued, issued % STAGES); ++issued; }
      ldB(S2, 1, 0); ldA(S3, 1, 0);
      __builtin_amdgcn_sched_barrier(0);
      quad(S1, S0, IgInt<0>{}, IgInt<0>{});                                   // B0.0 A0.0
      __builtin_amdgcn_sched_barrier(0);
      quad(S2, S0, IgInt<1>{}, IgInt<0>{});                                   // B1.0 A0.0
      __builtin_amdgcn_sched_barrier(0);
      ldA(S0, 1, 1);                                                          //     A1.1 -> S0
      __builtin_amdgcn_sched_barrier(0);
      quad(S2, S3, IgInt<1>{}, IgInt<1>{});                                   // B1.0 A1.0
      __builtin_amdgcn_sched_barrier(0);
      ldB(S2, 0, 1);                                                          //     B0.1 -> S2
      __builtin_amdgcn_sched_barrier(0);
      quad(S1, S3, IgInt<0>{}, IgInt<1>{});                                   // B0.0 A1.0
      __builtin_amdgcn_sched_barrier(0);
      ldA(S3, 0, 1); ldB(S1, 1, 1);                                           //     A0.1 -> S3, B1.1 -> S1
      __builtin_amdgcn_sched_barrier(0);
      quad(S2, S0, IgInt<0>{}, IgInt<1>{});                                   // B0.1 A1.1
      __builtin_amdgcn_sched_barrier(0);
      quad(S1, S0, IgInt<1>{}, IgInt<1>{});                                   // B1.1 A1.1
      __builtin_amdgcn_sched_barrier(0);
      quad(S1, S3, IgInt<1>{}, IgInt<0>{});                                   // B1.1 A0.1
      __builtin_amdgcn_sched_barrier(0);
      quad(S2, S3, IgInt<0>{}, IgInt<0>{});                                   // B0.1 A0.1
      __builtin_amdgcn_sched_barrier(0);
    } else {
    bf16x8 af[2][2], bfr[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      af[0][t] = *reinterpret_cast<const bf16x8*>(a + a_rd[t][0]);
      bfr[0][t] = *reinterpret_cast<const bf16x8*>(a + b_rd[t][0]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (issued < nk) { issue_chunk(issued, issued % STAGES); ++issued; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (ks < KS - 1) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          af[nxt][t] = *reinterpret_cast<const bf16x8*>(a + a_rd[t][ks + 1]);
          bfr[nxt][t] = *reinterpret_cast<const bf16x8*>(a + b_rd[t][ks + 1]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[cur][i], af[cur][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    }
  }

  // Every DMA has landed (the loop's own waits are inline asm the compiler cannot see): say so with a wait it CAN see,
  // or it guards each LDS access of the epilogue with vmcnt(0) -- which, once the first global stores are out, waits
  // for their acknowledgement (a store round trip per slab).
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched
  MPR_STAMP(2);
  // ---- epilogue: one 64-row slab of the fp32 tile at a time -> LDS -> 16-B channel groups (+ residual, BN partial sums).
  // Slab by slab keeps the LDS need at 64 rows (34 KB at BN = 128), below the ring, so the ring alone sets occupancy.
  constexpr int CPR = BN / 8;
  constexpr int RPP = T / CPR;
  static_assert(64 % RPP == 0, "a slab must be whole row passes");
  const int ch = tid % CPR, rr = tid / CPR;
  const int ncol = n0 + ch * 8;
  const bool col_ok = ncol < p.Nout;
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  float bmu[8], bis[8];      // BatchNorm-backward fusion: this thread's 8 channels are fixed, so are their coefficients
  if (BNB) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int n = col_ok ? ncol + e : 0;
      bmu[e] = p.bnb_mean[n];
      bis[e] = p.bnb_invstd[n];
    }
  }
  if (p.dbg & 16) {      // timing experiment: no epilogue (one store keeps the accumulators alive)
    if constexpr (M16) { if (acc16[0][0][0] + acc16[1][2][1] + acc16[2][1][2] + acc16[3][3][3] == 12345.f) p.dst[0] = (bf16_t)1.f; }
    else { if (acc[0][0][0] + acc[0][M16 ? 0 : 1][1] + acc[M16 ? 0 : 1][0][2] + acc[M16 ? 0 : 1][M16 ? 0 : 1][3] == 12345.f) p.dst[0] = (bf16_t)1.f; }
    return;
  }
  // The epilogue's global operands -- residual, half-resolution shortcut gradient, ReLU-mask source, BatchNorm input: up to
  // four maps -- are loaded ONE SLAB AHEAD of their use (round 3).  Loaded inside the row loop they exposed a full memory
  // latency per 64-row slab: the stride-2 data gradients, whose reductions are 2-8 chunks long, spent more time waiting for
  // these loads than multiplying (layer2.0: 308 us for 59 GFLOP, 0.08 matrix-pipe utilisation).
  constexpr int NIT = 64 / RPP;                 // rows of a slab per thread
  const bool has_add = p.add != nullptr;
  const bool has_even = PAR && cls == 0 && p.add_even != nullptr;
  // (the 16-wave tile has 128 VGPRs per wave: one buffer, loaded at the top of its own slab -- under the staging and its two
  //  barriers -- instead of a whole slab ahead)
  constexpr int NBUF = NW >= 16 ? 1 : 2;
  uint4 pre_add[NBUF][NIT], pre_even[NBUF][NIT], pre_my[NBUF][NIT], pre_bx[NBUF][NIT];
  int pre_m[NBUF][NIT];                         // output pixel of the row, -1: no such row / column group
  auto prefetch = [&](auto BUF_, int slab) {
    constexpr int buf = decltype(BUF_)::value;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      int m = m0 + slab * 64 + rr + it * RPP - mbase;
      const bool row_ok = m < (PAR ? p.par_valid : p.M);
      const int mc = row_ok ? m : 0;            // class-local row == pixel index of the half-resolution grid
      if (PAR && row_ok) {     // class-local row -> pixel (b, 2*h2 + par_h, 2*w2 + par_w) of the [B, 2*Pm, 2*Qm] gradient
        const uint32_t b = fdiv(m, p.div_pq);
        const uint32_t rem = m - b * (uint32_t)(p.Pm * p.Qm);
        const uint32_t h2 = fdiv(rem, p.div_q);
        const uint32_t w2 = rem - h2 * p.Qm;
        m = ((b * 2 * p.Pm + 2 * h2 + par_h) * 2 * p.Qm) + 2 * w2 + par_w;
      }
      const bool ok = row_ok && col_ok;
      pre_m[buf][it] = ok ? m : -1;
      const size_t o = (size_t)(ok ? m : 0) * p.Nout + (col_ok ? ncol : 0);     // (a safe address for rows that do not exist)
      if (has_add) pre_add[buf][it] = *reinterpret_cast<const uint4*>(p.add + o);
      if (has_even) pre_even[buf][it] = *reinterpret_cast<const uint4*>(p.add_even + (size_t)mc * p.Nout + (col_ok ? ncol : 0));
      if (BNB) {
        pre_my[buf][it] = *reinterpret_cast<const uint4*>(p.bnb_mask_y + o);
        pre_bx[buf][it] = *reinterpret_cast<const uint4*>(p.bnb_x + o);
      }
    }
  };
  if constexpr (NBUF == 2) prefetch(IgInt<0>{}, 0);
  ig_unroll<0, WM>([&](auto SLAB_) {
    constexpr int slab = decltype(SLAB_)::value, buf = NBUF == 2 ? (slab & 1) : 0;
    if constexpr (NBUF == 1) prefetch(IgInt<0>{}, slab);
    else if constexpr (slab + 1 < WM) prefetch(IgInt<(slab + 1) & 1>{}, slab + 1);
    lds_barrier();      // ring (first pass) / previous slab fully consumed
    if (wm == slab) {
      if constexpr (M16) {
#pragma unroll
        for (int jn = 0; jn < 4; ++jn)
#pragma unroll
          for (int im = 0; im < 4; ++im) {
            const int row = im * 16 + fr16;
            const int col = wn * 64 + jn * 16 + 4 * fq;
            float4 v = make_float4(acc16[jn][im][0], acc16[jn][im][1], acc16[jn][im][2], acc16[jn][im][3]);
            *reinterpret_cast<float4*>(smem + row * CS_STRIDE + col * 4) = v;
          }
      } else {
#pragma unroll
      for (int i = 0; i < (M16 ? 1 : 2); ++i)
#pragma unroll
        for (int j = 0; j < (M16 ? 1 : 2); ++j) {
          const int row = j * 32 + frow;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int col = wn * 64 + i * 32 + 8 * g + 4 * fh;
            float4 v = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
            *reinterpret_cast<float4*>(smem + row * CS_STRIDE + col * 4) = v;
          }
        }
      }
    }
    lds_barrier();
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int r = rr + it * RPP;
      const int m = pre_m[buf][it];
      if (m >= 0) {
        const float4 lo = *reinterpret_cast<const float4*>(smem + r * CS_STRIDE + ch * 32);
        const float4 hi = *reinterpret_cast<const float4*>(smem + r * CS_STRIDE + ch * 32 + 16);
        float f[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        const size_t o = (size_t)m * p.Nout + ncol;
        if (has_add) {
          float g[8];
          unpack8(pre_add[buf][it], g);
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] += g[e];
        }
        if (has_even) {
          float g[8];
          unpack8(pre_even[buf][it], g);
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] += g[e];
        }
        if (BNB) {
          // dz = (mask_y > 0) * q with q the bf16-rounded gradient; sums of dz and dz * xhat
          float q[8], ym[8], xf[8];
          unpack8(pack8(f), q);
          unpack8(pre_my[buf][it], ym);
          unpack8(pre_bx[buf][it], xf);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            q[e] = ym[e] > 0.f ? q[e] : 0.f;
            s1[e] += q[e];
            s2[e] = fmaf(q[e], xf[e], s2[e]);      // (sum dz * x; xhat is formed once per tile below: conv_win.hip, bnb_finish)
          }
          *reinterpret_cast<uint4*>(p.dst + o) = pack8(q);
        } else {
          const uint4 pk = pack8(f);
          *reinterpret_cast<uint4*>(p.dst + o) = pk;
          if (p.stats) {
            float q[8];
            unpack8(pk, q);
#pragma unroll
            for (int e = 0; e < 8; ++e) { s1[e] += q[e]; s2[e] += q[e] * q[e]; }
          }
        }
      }
    }
  });
  if (p.stats) {
    if (BNB) {
#pragma unroll
      for (int e = 0; e < 8; ++e) s2[e] = (s2[e] - bmu[e] * s1[e]) * bis[e];
    }
#pragma unroll
    for (int o = CPR; o < 64; o <<= 1)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s1[e] += __shfl_xor(s1[e], o, 64);
        s2[e] += __shfl_xor(s2[e], o, 64);
      }
    lds_barrier();
    float* red = reinterpret_cast<float*>(smem);   // [NW waves][CPR][16]
    if (lane < CPR) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(wid * CPR + lane) * 16 + e] = s1[e];
        red[(wid * CPR + lane) * 16 + 8 + e] = s2[e];
      }
    }
    lds_barrier();
    if (tid < CPR * 16) {
      const int c = tid >> 4, e = tid & 15;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += red[(w * CPR + c) * 16 + e];
      const int n = n0 + c * 8 + (e & 7);
      if (n < p.Nout) {
        if (p.stat_slices > 0) atomicAdd(&p.stats[((size_t)(mt % p.stat_slices) * 2 + (e >> 3)) * p.Nout + n], v);
        else p.stats[((size_t)mt * 2 + (e >> 3)) * p.Nout + n] = v;
      }
    }
  }
  MPR_STAMP(3);
#undef MPR_STAMP
#endif   // __HIP_DEVICE_COMPILE__
}

// ------------------------------------------------------------------------------------------------
// weight packing:  fp32 filter (any dense layout, given by its element strides: OIHW or the [K][R][S][C] memory of
// a channels-last weight)  ->  Wf [Kpad128][ceil64(R*S*C)]  and  Wd [Cpad128][ceil64(R*S*K)]  bf16
struct PackEntry {            // 12 x int64: one row of the table of mpr_conv_pack_weights_multi
  const float* w;
  bf16_t* wf;
  bf16_t* wd;                 // may be NULL
  long long K, C, R, S, sk, sc, sr, ss, pad_;
};

template <bool TILED_WD>
__device__ __forceinline__ void pack_one(const PackEntry& e, int first, int step) {
  const int K = (int)e.K, C = (int)e.C, R = (int)e.R, S = (int)e.S;
  const int KgF = R * S * C, KgD = R * S * K;
  const int KgFpad = (KgF + 63) / 64 * 64, KgDpad = (KgD + 63) / 64 * 64;
  const int totalF = ((K + 127) / 128 * 128) * KgFpad, totalD = e.wd ? ((C + 127) / 128 * 128) * KgDpad : 0;
  for (int i = first; i < totalF + totalD; i += step) {
    if (i < totalF) {
      const int k = i / KgFpad, g = i - k * KgFpad;
      float v = 0.f;
      if (k < K && g < KgF) {
        const int t = g / C, c = g - t * C, r = t / S, s = t - r * S;
        v = e.w[k * e.sk + c * e.sc + r * e.sr + s * e.ss];
      }
      e.wf[i] = (bf16_t)v;
    } else if (!TILED_WD) {
      const int j = i - totalF;
      const int c = j / KgDpad, g = j - c * KgDpad;
      float v = 0.f;
      if (c < C && g < KgD) {
        const int t = g / K, k = g - t * K, r = t / S, s = t - r * S;
        v = e.w[k * e.sk + c * e.sc + r * e.sr + s * e.ss];
      }
      e.wd[j] = (bf16_t)v;
    } else {
      // tiled variant: only the zero padding of the data-gradient panel here, its body below
      const int j = i - totalF;
      const int c = j / KgDpad, g = j - c * KgDpad;
      if (c >= C || g >= KgD) e.wd[j] = (bf16_t)0.f;
    }
  }
}

// Data-gradient panel wd[c][(tap, k)] = w[k][tap][c]: a K x C transpose per tap.  Read straight through, one side of it
// is a 4-byte access every K (or R*S*C) elements -- 577 MB of traffic for 45 MB of filters; through a 32 x 33 LDS tile
// both sides are contiguous.  Whole 256-thread workgroup per tile, tiles strided over gridDim.x.
__device__ __forceinline__ void pack_wd_tiled(const PackEntry& e) {
  __shared__ float tile[32][33];
  const int K = (int)e.K, C = (int)e.C, taps = (int)(e.R * e.S), S = (int)e.S;
  const int KgDpad = (taps * K + 63) / 64 * 64;
  const int tk = (K + 31) / 32, tc = (C + 31) / 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
  for (int t = blockIdx.x; t < taps * tk * tc; t += gridDim.x) {
    const int tap = t / (tk * tc), rem = t - tap * (tk * tc);
    const int k0 = (rem / tc) * 32, c0 = (rem % tc) * 32;
    const int r = tap / S, s = tap - r * S;
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int k = k0 + ty + 8 * p, c = c0 + tx;
      tile[ty + 8 * p][tx] = (k < K && c < C) ? e.w[k * e.sk + c * e.sc + r * e.sr + s * e.ss] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int c = c0 + ty + 8 * p, k = k0 + tx;
      if (c < C && k < K) e.wd[(size_t)c * KgDpad + tap * K + k] = (bf16_t)tile[tx][ty + 8 * p];
    }
  }
}

__global__ void conv_pack_weights_kernel(const PackEntry e) {
  pack_one<false>(e, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// Refresh of an EXISTING forward panel whose filter rows are contiguous in memory ([K][R][S][C] channels-last filters,
// nn.Linear weights): row k of the panel is a straight cast of R*S*C contiguous floats -- 8 elements per thread, one
// division per 8 elements, 16-byte stores.  The zero padding (rows >= K, columns >= R*S*C) was written by the first pack
// (mpr_conv_pack_weights_strided) and is left alone, as is the data-gradient panel's.
__device__ __forceinline__ void pack_fwd_rows_vec(const PackEntry& e, int first, int step) {
  const int K = (int)e.K, KgF = (int)(e.R * e.S * e.C), KgFpad = (KgF + 63) / 64 * 64, G = KgF / 8;
  for (int i = first; i < K * G; i += step) {
    const int k = i / G, g = i - k * G;
    const float4* src = reinterpret_cast<const float4*>(e.w + (size_t)k * e.sk + g * 8);
    const float4 a = src[0], b = src[1];
    const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    *reinterpret_cast<uint4*>(e.wf + (size_t)k * KgFpad + g * 8) = pack8(f);
  }
}

__global__ __launch_bounds__(256) void conv_pack_weights_multi_kernel(const PackEntry* __restrict__ table) {
  const PackEntry e = table[blockIdx.y];
  const bool rows_contiguous = e.sc == 1 && e.C % 8 == 0 && (e.sk * 4) % 16 == 0 &&
                               ((e.R * e.S == 1) || (e.ss == e.C && e.sr == e.S * e.C));
  if (rows_contiguous) {
    pack_fwd_rows_vec(e, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
  } else {
    pack_one<true>(e, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
  }
  if (e.wd) pack_wd_tiled(e);
}

static inline int pad_to(int x, int a) { return (x + a - 1) / a * a; }

extern "C" {

// Sizes (in bf16 elements) of the packed panels for a [K,C,R,S] filter.
int mpr_conv_packed_sizes(int K, int C, int R, int S, long long* fwd_elems, long long* dgrad_elems) {
  if (fwd_elems) *fwd_elems = (long long)pad_to(K, 128) * pad_to(R * S * C, 64);
  if (dgrad_elems) *dgrad_elems = (long long)pad_to(C, 128) * pad_to(R * S * K, 64);
  return MPR_OK;
}

// w: logical [K,C,R,S] with element strides (sk, sc, sr, ss)
int mpr_conv_pack_weights_strided(const float* w, long long sk, long long sc, long long sr, long long ss, void* w_fwd,
                                  void* w_dgrad, int K, int C, int R, int S, void* stream) {
  MPR_REQUIRE(w && w_fwd, "mpr_conv_pack_weights: null pointer");
  const int KgF = R * S * C, KgD = R * S * K;
  const int total = pad_to(K, 128) * pad_to(KgF, 64) + (w_dgrad ? pad_to(C, 128) * pad_to(KgD, 64) : 0);
  const int grid = ceil_div(total, 256) < 2048 ? ceil_div(total, 256) : 2048;
  PackEntry e{w, (bf16_t*)w_fwd, (bf16_t*)w_dgrad, K, C, R, S, sk, sc, sr, ss, 0};
  conv_pack_weights_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(e);
  MPR_LAUNCH_CHECK("conv_pack_weights_kernel");
  return MPR_OK;
}

int mpr_conv_pack_weights(const float* w_oihw, void* w_fwd, void* w_dgrad, int K, int C, int R, int S,
                          void* stream) {
  return mpr_conv_pack_weights_strided(w_oihw, (long long)C * R * S, (long long)R * S, S, 1, w_fwd, w_dgrad, K, C, R,
                                       S, stream);
}

// table: n rows of 12 int64 on the device: {w, w_fwd, w_dgrad (0 = none), K, C, R, S, sk, sc, sr, ss, 0}.
// One launch repacks every filter of a model after the optimizer step.
int mpr_conv_pack_weights_multi(const void* table, int n, void* stream) {
  MPR_REQUIRE(table && n > 0, "mpr_conv_pack_weights_multi: empty table");
  static_assert(sizeof(PackEntry) == 12 * 8, "PackEntry must be 12 x int64");
  conv_pack_weights_multi_kernel<<<dim3(128, n), 256, 0, (hipStream_t)stream>>>((const PackEntry*)table);
  MPR_LAUNCH_CHECK("conv_pack_weights_multi_kernel");
  return MPR_OK;
}

// Kernel / tile selection, shared by the launcher and the stat-row query.
//   mode 1: LDS-DMA ring  (source channels % 64 == 0 and enough rows to fill the chip)
//   mode 0: register-staged kernel (any channel count that is a multiple of 8; small problems)
// shifted-window kernel for 3x3 / stride 1 / pad 1 (conv_win.hip)
bool mpr_win_eligible(long long M, int H, int W, int srcC, int Nout, int R, int S, int sh, int sw, int ph, int pw,
                      long long min_rows);
int mpr_win_stat_rows(int B, int H, int W, int Nout);
struct WinBnb {   // BatchNorm-backward fusion of a data gradient (conv_win.hip)
  int mask_mode;
  const void *mask_y, *bn_x;
  const float *mean, *invstd, *scale, *shift;
  float* slices;
  int nslices, prezeroed;
};
int mpr_win_launch(bool dgrad, const void* src, const void* wpk, void* dst, const void* add, float* stats, int B, int H,
                   int W, int srcC, int Nout, hipStream_t st, const WinBnb* bnb);

// BatchNorm partial sums of the forward convolutions: n > 0 = every tile adds (fp32 atomics) into one of n slice rows that
// the launcher zeroes -- the consumer finalizes from the n rows directly and the pre-reduction launch between a convolution and
// its BatchNorm disappears; 0 = one row per tile (bitwise reproducible sums).  Returns the previous setting.
static int g_stat_slices = 4;      // (= ops.FIN_SLICES)
int mpr_conv_stat_slices() { return g_stat_slices; }
// one-shot promise of the caller that the slice rows handed to the NEXT mpr_conv_fwd are already zero (a per-step arena
// zeroed in one go): the launcher then skips its own memset -- 40 fill launches of ~5 us per C3 step otherwise
static int g_stats_prezeroed = 0;
extern "C" int mpr_conv_stats_prezeroed(int on) {
  g_stats_prezeroed = on;
  return 0;
}
bool mpr_conv_take_prezeroed() {
  const bool v = g_stats_prezeroed != 0;
  g_stats_prezeroed = 0;
  return v;
}
extern "C" int mpr_conv_set_stat_slices(int n) {
  const int old = g_stat_slices;
  if (n >= 0) g_stat_slices = n;          // (n < 0: query only)
  return old;
}

static int g_dma_min_rows = 16384;
extern "C" int mpr_conv_set_dma_min_rows(int rows) {   // tuning / test knob; returns the previous value
  const int old = g_dma_min_rows;
  g_dma_min_rows = rows;
  return old;
}

// tile / ring-depth variants of the LDS-DMA kernel (tuning knob; defaults are the measured best)
//   narrow (N <= 64): 0 = 256x64, 2 stages (2 WG/CU)   1 = 256x64, 3 stages (1 WG/CU)   2 = 128x64, 2 stages (3 WG/CU)
//                     3 = 256x64, BK 32, 4 stages        4 = 256x64, BK 32, 3 stages
//   wide   (N > 64) : 0 = 256x128, 3 stages, 8 waves    1 = 128x128, 2 stages (2 WG/CU)   2 = 128x128, 3 stages
//                     3 = 128x128, BK 32, 3 stages (3 WG/CU)   4 = 128x128, BK 32, 4 stages (2 WG/CU)
//                     5 = 256x128, BK 32, 3 stages (2 WG/CU)   6 = 256x128, BK 32, 4 stages (1 WG/CU)
// (wide 7 since round 2: 256 x 256 tiles on 16 waves for every output width >= 256 -- inside the step, beside the weight-
//  gradient stream, fewer and fatter workgroups win: C5 share 59.0 -> 57.1 ms, C3 10.21 -> 10.10 ms in the in-process A/B)
static int g_variant_narrow = 0, g_variant_wide = 7;
extern "C" int mpr_conv_set_variant(int narrow, int wide) {
  g_variant_narrow = narrow;
  g_variant_wide = wide;
  return 0;
}

static unsigned long long* g_debug_stamps = nullptr;
extern "C" int mpr_conv_debug_stamps(void* buf) {   // 4 x uint64 per workgroup of the next LDS-DMA conv launches
  g_debug_stamps = (unsigned long long*)buf;
  return 0;
}
static int g_debug_drop = 0;   // bit 0: drop the activation operand's loads, bit 1: the weight operand's (LDS-DMA kernel)
extern "C" int mpr_conv_debug_drop_operand(int mask) {
  const int old = g_debug_drop;
  g_debug_drop = mask;
  return old;
}
static int g_dgrad_parity = 1;
extern "C" int mpr_conv_set_dgrad_parity(int on) {
  const int old = g_dgrad_parity;
  g_dgrad_parity = on;
  return old;
}

static inline void igemm_config(long long M, int Nout, int srcC, int taps, int* mode, int* BM, int* BN) {
  const bool narrow = Nout <= 64;
  if (srcC % 64 == 0 && taps <= 32 && M >= g_dma_min_rows) {
    const int vw = g_variant_wide;
    // wide outputs of big GEMMs (the transformer's qkv / MLP linears: N >= 1536, one tap): 256 x 256 tile on 16 waves --
    // 128 FLOP per LDS-DMA byte instead of 64; measured 858 / 931 / 906 vs 780 / 761 / 719 TFLOP/s on ViT-B/16's qkv
    // forward / fc1 forward / fc2 data gradient (scripts/bench_linear_variants.py), slower below that width (too few tiles)
    const bool big = vw == 1 && Nout >= 1536 && M >= 8192 && taps == 1;
    *mode = 1; *BN = narrow ? 64 : ((big || (vw >= 7 && vw <= 9 && Nout >= 256)) ? 256 : 128);
    *BM = narrow ? (g_variant_narrow == 2 ? 128 : 256)
                 : ((big || vw == 0 || vw == 5 || vw == 6 || ((vw == 7 || vw == 8) && Nout >= 256)) ? 256 : 128);
  } else if (srcC % 32 == 0 && taps <= 32 && M >= g_dma_min_rows) {
    // source channels a multiple of 32 only (EfficientNet-B0's 96 / 480 / 672-channel maps): the LDS-DMA ring on 64-byte rows
    // (32-deep chunks) instead of the register-staged kernel
    *mode = 1; *BM = narrow ? 256 : 128; *BN = narrow ? 64 : 128;
  } else {
    *mode = 0; *BM = narrow ? 256 : 128; *BN = narrow ? 64 : 128;
  }
}

static int launch_igemm(bool dgrad, ConvGemmParams& p, hipStream_t st) {
  p.rowpat = 0;
  for (int r = 0; r < p.R && r * p.S < 32; ++r) p.rowpat |= 1u << (r * p.S);
  int mode, BM, BN;
  igemm_config(p.M, p.Nout, p.sC, p.R * p.S, &mode, &BM, &BN);
  const bool bnb = dgrad && p.bnb_x != nullptr;       // (its slice rows and their zeroing: the caller's, see conv_dgrad_impl)
  if (bnb && BN != 64) { BM = 128; BN = 128; }        // the fused BatchNorm-backward epilogue exists on the default tiles only
  const bool narrow = BN == 64;
  // algorithmic flops: a strided data gradient touches each (pixel, tap) pair of the forward conv once
  const double flops = 2.0 * (double)p.M * (double)p.Nout * (double)p.Kg / (dgrad ? p.sh * p.sw : 1);
  // algorithmic HBM bytes: source and weights read once, result written once (+ the fused residual read once)
  const double algo_bytes = (double)p.src_bytes + 2.0 * p.Nout * p.Kg + 2.0 * (double)p.M * p.Nout * (p.add ? 2 : 1);
  // stride-2 data gradient on even extents: rows regrouped into the 4 (h mod 2, w mod 2) classes, each padded to
  // whole tiles, and every tile walks only its class's taps (see the kernel)
  p.par_rows = 0; p.par_valid = 0;
  // (a 1x1 filter has a single tap to walk either way; g_dgrad_parity == 2 forces the regrouping for the tests)
  if (dgrad && mode == 1 && p.sh == 2 && p.sw == 2 && p.Pm % 2 == 0 && p.Qm % 2 == 0 &&
      (g_dgrad_parity == 2 || (g_dgrad_parity == 1 && p.R * p.S > 1))) {
    const int B = p.M / (p.Pm * p.Qm);
    p.Pm /= 2; p.Qm /= 2;
    p.div_pq = make_fastdiv(p.Pm * p.Qm); p.div_q = make_fastdiv(p.Qm);
    p.par_valid = B * p.Pm * p.Qm;
    p.par_rows = ceil_div(p.par_valid, BM) * BM;
    p.M = 4 * p.par_rows;
  }
  // other strided data gradients (3/4 of the taps are holes) run better on the deep 256x128 ring
  const bool s2dgrad = dgrad && mode == 1 && !narrow && (p.sh == 2 || p.sw == 2) && p.par_rows == 0;
  if (s2dgrad) { BM = 256; BN = 128; }
  p.ntn = ceil_div(p.Nout, BN);
  const int gm = ceil_div(p.M, BM);
  dim3 grid(gm * p.ntn);
  if (!bnb) {
    p.stat_slices = (p.stats && !dgrad) ? g_stat_slices : 0;
    const bool prezeroed = mpr_conv_take_prezeroed();
    if (p.stat_slices > 0 && !prezeroed)
      MPR_HIP(hipMemsetAsync(p.stats, 0, sizeof(float) * 2 * (size_t)p.stat_slices * p.Nout, st));
  }
  // profiler kinds: 0/1 = LDS-DMA kernel fwd/dgrad (the dominant kernel), 3/4 = register-staged kernel fwd/dgrad
  // timing-only experiment (MI355X guide, traffic pricing): a zero-record descriptor drops every load through it
  p.dbg = g_debug_drop;
  p.stamps = g_debug_stamps;
  if (g_debug_drop & 1) p.src_bytes = 0;
  if (g_debug_drop & 2) p.wpk_bytes = 0;
  void* tok = mpr_prof_begin((mode == 1 ? 0 : 3) + (dgrad ? 1 : 0), flops, st);
  mpr_prof_bytes(tok, algo_bytes);
  const bool m16 = !(g_debug_drop & 32);      // (debug bit 5: the 32x32x16 form, comparisons)
  if (mode == 1) {
#define MPR_DMA5(WM_, WN_, ST_, DG_, BK_)                                                                 \
  do {                                                                                                \
    static bool attr_set = false;                                                                     \
    if (!attr_set) {                                                                                  \
      hipFuncSetAttribute((const void*)conv_igemm_dma_kernel<WM_, WN_, ST_, DG_, BK_>,                \
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                    \
      attr_set = true;                                                                                \
    }                                                                                                 \
    static bool attr16_set = false;                                                                   \
    if (!attr16_set) {                                                                                \
      hipFuncSetAttribute((const void*)conv_igemm_dma_kernel<WM_, WN_, ST_, DG_, BK_, false, (BK_ == 64)>, \
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                    \
      attr16_set = true;                                                                              \
    }                                                                                                 \
    const size_t ring_ = (size_t)ST_ * (64 * WM_ + 64 * WN_) * (2 * BK_);                             \
    const size_t epi_ = (size_t)64 * (64 * WN_ * 4 + 16);                                             \
    if (m16)                                                                                          \
      conv_igemm_dma_kernel<WM_, WN_, ST_, DG_, BK_, false, (BK_ == 64)>                              \
          <<<grid, 64 * WM_ * WN_, ring_ > epi_ ? ring_ : epi_, st>>>(p);                             \
    else                                                                                              \
      conv_igemm_dma_kernel<WM_, WN_, ST_, DG_, BK_>                                                  \
          <<<grid, 64 * WM_ * WN_, ring_ > epi_ ? ring_ : epi_, st>>>(p);                             \
  } while (0)
#define MPR_DMA(WM_, WN_, ST_, BK_) do { if (dgrad) MPR_DMA5(WM_, WN_, ST_, true, BK_); else MPR_DMA5(WM_, WN_, ST_, false, BK_); } while (0)
    if (bnb) {
      // fused BatchNorm-backward epilogue: default tiles of the parity-class data gradient only
#define MPR_DMAB(WM_, WN_, ST_)                                                                          \
  do {                                                                                                \
    static bool attr_set = false;                                                                     \
    if (!attr_set) {                                                                                  \
      hipFuncSetAttribute((const void*)conv_igemm_dma_kernel<WM_, WN_, ST_, true, 64, true>,          \
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                    \
      attr_set = true;                                                                                \
    }                                                                                                 \
    static bool attr16_set = false;                                                                   \
    if (!attr16_set) {                                                                                \
      hipFuncSetAttribute((const void*)conv_igemm_dma_kernel<WM_, WN_, ST_, true, 64, true, true>,    \
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                    \
      attr16_set = true;                                                                              \
    }                                                                                                 \
    const size_t ring_ = (size_t)ST_ * (64 * WM_ + 64 * WN_) * 128;                                   \
    const size_t epi_ = (size_t)64 * (64 * WN_ * 4 + 16);                                             \
    if (m16)                                                                                          \
      conv_igemm_dma_kernel<WM_, WN_, ST_, true, 64, true, true>                                      \
          <<<grid, 64 * WM_ * WN_, ring_ > epi_ ? ring_ : epi_, st>>>(p);                             \
    else                                                                                              \
      conv_igemm_dma_kernel<WM_, WN_, ST_, true, 64, true>                                            \
          <<<grid, 64 * WM_ * WN_, ring_ > epi_ ? ring_ : epi_, st>>>(p);                             \
  } while (0)
      if (narrow) MPR_DMAB(4, 1, 2); else MPR_DMAB(2, 2, 2);
#undef MPR_DMAB
    } else if (p.sC % 64 != 0) {      // 32-deep chunks (see igemm_config); tiles as the register-staged kernel's
      if (narrow) MPR_DMA(4, 1, 4, 32);
      else if (s2dgrad) MPR_DMA5(4, 2, 3, true, 32);
      else MPR_DMA(2, 2, 3, 32);
    } else if (narrow) {
      switch (g_variant_narrow) {
        case 1: MPR_DMA(4, 1, 3, 64); break;
        case 2: MPR_DMA(2, 1, 2, 64); break;
        case 3: MPR_DMA(4, 1, 4, 32); break;
        case 4: MPR_DMA(4, 1, 3, 32); break;
        default: MPR_DMA(4, 1, 2, 64); break;
      }
    } else if (s2dgrad) {
      MPR_DMA5(4, 2, 3, true, 64);
    } else {
      switch (g_variant_wide) {
        case 0: MPR_DMA(4, 2, 3, 64); break;
        case 2: MPR_DMA(2, 2, 3, 64); break;
        case 3: MPR_DMA(2, 2, 3, 32); break;
        case 4: MPR_DMA(2, 2, 4, 32); break;
        case 5: MPR_DMA(4, 2, 3, 32); break;
        case 6: MPR_DMA(4, 2, 4, 32); break;
        case 7: if (BN == 256) MPR_DMA(4, 4, 2, 64); else MPR_DMA(2, 2, 2, 64); break;     // 256 x 256, 16 waves
        case 8: if (BN == 256) MPR_DMA(4, 4, 3, 32); else MPR_DMA(2, 2, 2, 64); break;
        case 9: if (BN == 256) MPR_DMA(2, 4, 2, 64); else MPR_DMA(2, 2, 2, 64); break;     // 128 x 256, 8 waves
        default: if (BN == 256) MPR_DMA(4, 4, 2, 64); else MPR_DMA(2, 2, 2, 64); break;
      }
    }
#undef MPR_DMA
#undef MPR_DMA5
  } else {
    const size_t stage2 = (size_t)2 * (BM + BN) * 128, epi = (size_t)BM * (BN * 4 + 16);
    const size_t smem = stage2 > epi ? stage2 : epi;
#define MPR_IGEMM(WM_, WN_, DG_)                                                                      \
  do {                                                                                                \
    static bool attr_set = false;                                                                     \
    if (!attr_set) {                                                                                  \
      hipFuncSetAttribute((const void*)conv_igemm_kernel<WM_, WN_, DG_>,                              \
                          hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);                     \
      attr_set = true;                                                                                \
    }                                                                                                 \
    conv_igemm_kernel<WM_, WN_, DG_><<<grid, 256, smem, st>>>(p);                                     \
  } while (0)
    if (narrow) { if (dgrad) MPR_IGEMM(4, 1, true); else MPR_IGEMM(4, 1, false); }
    else        { if (dgrad) MPR_IGEMM(2, 2, true); else MPR_IGEMM(2, 2, false); }
#undef MPR_IGEMM
  }
  mpr_prof_end(tok, st);
  MPR_LAUNCH_CHECK("conv_igemm_kernel");
  return MPR_OK;
}

// Forward convolutions on maps narrower than this take the LDS-DMA implicit GEMM although the shifted-window kernel is
// eligible: the padded raster of the window kernel computes (W+1)(H+1) positions per image (+31 % at 7 x 7) and the wide
// 256 x 256 tile of the implicit GEMM holds fewer CUs for the same work
static int g_win_fwd_min_w = 8;   // layer4's 7 x 7 maps: 9.98 vs 10.02 ms per C3 step (scripts/step_ab_flags.py)
int mpr_conv_set_window_fwd_min_width(int w) {
  const int old = g_win_fwd_min_w;
  g_win_fwd_min_w = w;
  return old;
}
static inline bool win_fwd(long long M, int H, int W, int C, int K, int R, int S, int sh, int sw, int ph, int pw) {
  return W >= g_win_fwd_min_w && mpr_win_eligible(M, H, W, C, K, R, S, sh, sw, ph, pw, g_dma_min_rows);
}

// Rows of the BatchNorm partial-sum buffer mpr_conv_fwd will write (slice rows, or one per row tile).
int mpr_conv_fwd_stat_rows(int B, int P, int Q, int K, int C, int R, int S, int sh, int sw, int ph, int pw) {
  if (g_stat_slices > 0) return g_stat_slices;
  if (win_fwd((long long)B * P * Q, P, Q, C, K, R, S, sh, sw, ph, pw))   // (stride 1: H == P)
    return mpr_win_stat_rows(B, P, Q, K);
  int mode, BM, BN;
  igemm_config((long long)B * P * Q, K, C, R * S, &mode, &BM, &BN);
  return ceil_div(B * P * Q, BM);
}

// y[B,P,Q,K] = conv(x[B,H,W,C], w) ; stats (optional): [mpr_conv_fwd_stat_rows][2][K] fp32 partial sums
int mpr_conv_fwd(const void* x, const void* w_fwd, void* y, float* stats, int B, int H, int W, int C,
                 int K, int R, int S, int sh, int sw, int ph, int pw, void* stream) {
  // the one-shot "slice rows are already zero" promise (mpr_conv_stats_prezeroed) belongs to THIS call: taken before any
  // argument check can return (it would otherwise stay armed and make a later convolution skip its memset), re-armed
  // right in front of the launcher that consumes it
  const bool prezeroed = mpr_conv_take_prezeroed();
  MPR_REQUIRE(x && w_fwd && y, "mpr_conv_fwd: null pointer");
  MPR_REQUIRE(C % 8 == 0 && K % 8 == 0, "mpr_conv_fwd: C (%d) and K (%d) must be multiples of 8", C, K);
  MPR_REQUIRE(sh >= 1 && sw >= 1 && R >= 1 && S >= 1 && ph >= 0 && pw >= 0, "mpr_conv_fwd: bad geometry");
  const int P = (H + 2 * ph - R) / sh + 1, Q = (W + 2 * pw - S) / sw + 1;
  MPR_REQUIRE(P > 0 && Q > 0, "mpr_conv_fwd: empty output");
  MPR_REQUIRE((long long)B * H * W * C < (1ll << 31) && (long long)B * P * Q * K < (1ll << 31),
              "mpr_conv_fwd: tensor exceeds 2^31 elements");
  if (win_fwd((long long)B * P * Q, H, W, C, K, R, S, sh, sw, ph, pw)) {
    // 3x3 / stride 1 / pad 1: shifted-window kernel (conv_win.hip)
    void* tok = mpr_prof_begin(6, 2.0 * (double)B * P * Q * K * 9.0 * C, (hipStream_t)stream);      // kind 6: window fwd
    mpr_prof_bytes(tok, 2.0 * ((double)B * H * W * C + 9.0 * C * K + (double)B * P * Q * K));
    g_stats_prezeroed = prezeroed;
    const int rc = mpr_win_launch(false, x, w_fwd, y, nullptr, stats, B, H, W, C, K, (hipStream_t)stream, nullptr);
    mpr_prof_end(tok, (hipStream_t)stream);
    return rc;
  }
  ConvGemmParams p;
  p.src = (const bf16_t*)x; p.wpk = (const bf16_t*)w_fwd; p.dst = (bf16_t*)y; p.add = nullptr; p.add_even = nullptr; p.stats = stats;
  p.bnb_mask_y = p.bnb_x = nullptr; p.bnb_mean = p.bnb_invstd = nullptr;
  p.sH = H; p.sW = W; p.sC = C;
  p.src_bytes = (unsigned)((size_t)B * H * W * C * 2);
  p.wpk_bytes = (unsigned)((size_t)pad_to(K, 128) * pad_to(R * S * C, 64) * 2);
  p.M = B * P * Q; p.Nout = K; p.Kg = R * S * C; p.Kgpad = pad_to(p.Kg, 64); p.nk = p.Kgpad / 64;
  p.R = R; p.S = S; p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw;
  p.Pm = P; p.Qm = Q; p.div_pq = make_fastdiv(P * Q); p.div_q = make_fastdiv(Q);
  g_stats_prezeroed = prezeroed;
  return launch_igemm(false, p, (hipStream_t)stream);
}

// dx[B,H,W,C] = conv_transpose(dy[B,P,Q,K], w) (+ add[B,H,W,C] if given)
static bool dgrad_parity_path(int B, int H, int W, int C, int K, int R, int S, int sh, int sw) {
  // (mirrors launch_igemm: LDS-DMA kernel + stride 2 on even extents + more than one tap)
  int mode, BM, BN;
  igemm_config((long long)B * H * W, C, K, R * S, &mode, &BM, &BN);
  return mode == 1 && sh == 2 && sw == 2 && H % 2 == 0 && W % 2 == 0 && g_dgrad_parity >= 1 && R * S > 1;
}

// Can mpr_conv_dgrad take `add_even` for this geometry (stride-2 data gradient on the parity-class path)?
int mpr_conv_dgrad_add_even_supported(int B, int H, int W, int C, int K, int R, int S, int sh, int sw, int ph, int pw) {
  (void)ph; (void)pw;
  return C % 8 == 0 && K % 8 == 0 && dgrad_parity_path(B, H, W, C, K, R, S, sh, sw) ? 1 : 0;
}

struct DgradBnb {   // BatchNorm-backward fusion of the parity-class data gradient (mask from mask_y)
  const void *mask_y, *bn_x;
  const float *mean, *invstd;
  float* slices;
  int nslices, prezeroed;
};

static int conv_dgrad_impl(const void* dy, const void* w_dgrad, void* dx, const void* add, const void* add_even, int B, int H,
                           int W, int C, int K, int R, int S, int sh, int sw, int ph, int pw, void* stream,
                           const DgradBnb* bnb = nullptr) {
  MPR_REQUIRE(dy && w_dgrad && dx, "mpr_conv_dgrad: null pointer");
  MPR_REQUIRE(C % 8 == 0 && K % 8 == 0, "mpr_conv_dgrad: C (%d) and K (%d) must be multiples of 8", C, K);
  MPR_REQUIRE((sh == 1 || sh == 2) && (sw == 1 || sw == 2), "mpr_conv_dgrad: strides must be 1 or 2 (got %d,%d)", sh, sw);
  const int P = (H + 2 * ph - R) / sh + 1, Q = (W + 2 * pw - S) / sw + 1;
  MPR_REQUIRE(P > 0 && Q > 0, "mpr_conv_dgrad: empty output");
  MPR_REQUIRE((long long)B * H * W * C < (1ll << 31) && (long long)B * P * Q * K < (1ll << 31),
              "mpr_conv_dgrad: tensor exceeds 2^31 elements");
  MPR_REQUIRE(!add_even || (dgrad_parity_path(B, H, W, C, K, R, S, sh, sw) && P == H / 2 && Q == W / 2),
              "mpr_conv_dgrad_s2: geometry not on the parity-class path (ask mpr_conv_dgrad_add_even_supported)");
  if (mpr_win_eligible((long long)B * H * W, H, W, K, C, R, S, sh, sw, ph, pw, g_dma_min_rows)) {
    void* tok = mpr_prof_begin(7, 2.0 * (double)B * H * W * C * 9.0 * K, (hipStream_t)stream);      // kind 7: window dgrad
    mpr_prof_bytes(tok, 2.0 * ((double)B * P * Q * K + 9.0 * C * K + (double)B * H * W * C * (add ? 2 : 1)));
    const int rc = mpr_win_launch(true, dy, w_dgrad, dx, add, nullptr, B, H, W, K, C, (hipStream_t)stream, nullptr);
    mpr_prof_end(tok, (hipStream_t)stream);
    return rc;
  }
  ConvGemmParams p;
  p.src = (const bf16_t*)dy; p.wpk = (const bf16_t*)w_dgrad; p.dst = (bf16_t*)dx; p.add = (const bf16_t*)add;
  p.add_even = (const bf16_t*)add_even;
  p.stats = nullptr;
  p.bnb_mask_y = p.bnb_x = nullptr; p.bnb_mean = p.bnb_invstd = nullptr;
  if (bnb) {
    p.bnb_mask_y = (const bf16_t*)bnb->mask_y; p.bnb_x = (const bf16_t*)bnb->bn_x;
    p.bnb_mean = bnb->mean; p.bnb_invstd = bnb->invstd;
    p.stats = bnb->slices; p.stat_slices = bnb->nslices;
    if (!bnb->prezeroed)
      MPR_HIP(hipMemsetAsync(bnb->slices, 0, sizeof(float) * 2 * (size_t)bnb->nslices * C, (hipStream_t)stream));
  }
  p.sH = P; p.sW = Q; p.sC = K;
  p.src_bytes = (unsigned)((size_t)B * P * Q * K * 2);
  p.wpk_bytes = (unsigned)((size_t)pad_to(C, 128) * pad_to(R * S * K, 64) * 2);
  p.M = B * H * W; p.Nout = C; p.Kg = R * S * K; p.Kgpad = pad_to(p.Kg, 64); p.nk = p.Kgpad / 64;
  p.R = R; p.S = S; p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw;
  p.Pm = H; p.Qm = W; p.div_pq = make_fastdiv(H * W); p.div_q = make_fastdiv(W);
  return launch_igemm(true, p, (hipStream_t)stream);
}

int mpr_conv_dgrad(const void* dy, const void* w_dgrad, void* dx, const void* add, int B, int H, int W,
                   int C, int K, int R, int S, int sh, int sw, int ph, int pw, void* stream) {
  return conv_dgrad_impl(dy, w_dgrad, dx, add, nullptr, B, H, W, C, K, R, S, sh, sw, ph, pw, stream);
}

// Stride-2 data gradient + the gradient of a parallel 1x1 / stride-2 / pad-0 convolution given on the HALF-resolution
// grid (add_even [B, H/2, W/2, C]: it only reaches the even pixels) -- the block-input gradient of a ResNet
// downsampling block without materialising the 3/4-zero full-resolution shortcut gradient.
int mpr_conv_dgrad_s2(const void* dy, const void* w_dgrad, void* dx, const void* add_even, int B, int H, int W,
                      int C, int K, int R, int S, int sh, int sw, int ph, int pw, void* stream) {
  MPR_REQUIRE(add_even, "mpr_conv_dgrad_s2: null add_even");
  return conv_dgrad_impl(dy, w_dgrad, dx, nullptr, add_even, B, H, W, C, K, R, S, sh, sw, ph, pw, stream);
}

// mpr_conv_dgrad_s2 + the BatchNorm backward of the block output it differentiates, relu(bn2(x2) + identity): dz = the
// gradient masked by mask_y > 0 (mask_y: that block output), slices [nslices][2][C] += sum dz, sum dz * xhat(bn_x) -- the
// downsampling block's counterpart of mpr_conv_dgrad_bn (mask mode 1); add_even may be NULL.
int mpr_conv_dgrad_s2_bn(const void* dy, const void* w_dgrad, void* dz, const void* add_even, const void* mask_y,
                         const void* bn_x, const float* mean, const float* invstd, float* slices, int nslices, int prezeroed,
                         int B, int H, int W, int C, int K, int R, int S, int sh, int sw, int ph, int pw, void* stream) {
  MPR_REQUIRE(mask_y && bn_x && mean && invstd && slices && nslices > 0, "mpr_conv_dgrad_s2_bn: null pointer");
  const int P = (H + 2 * ph - R) / sh + 1, Q = (W + 2 * pw - S) / sw + 1;
  MPR_REQUIRE(C % 8 == 0 && K % 64 == 0 && dgrad_parity_path(B, H, W, C, K, R, S, sh, sw) && P == H / 2 && Q == W / 2,
              "mpr_conv_dgrad_s2_bn: geometry not on the parity-class path with 64-deep chunks (K %% 64 == 0; ask "
              "mpr_conv_dgrad_add_even_supported)");
  DgradBnb bnb = {mask_y, bn_x, mean, invstd, slices, nslices, prezeroed};
  return conv_dgrad_impl(dy, w_dgrad, dz, nullptr, add_even, B, H, W, C, K, R, S, sh, sw, ph, pw, stream, &bnb);
}

// Is the data gradient with fused BatchNorm-backward reduction (mpr_conv_dgrad_bn) available for this geometry?
int mpr_conv_dgrad_bn_supported(int B, int H, int W, int C, int K, int R, int S, int sh, int sw, int ph, int pw) {
  return mpr_win_eligible((long long)B * H * W, H, W, K, C, R, S, sh, sw, ph, pw, g_dma_min_rows) ? 1 : 0;
}

// dz[B,H,W,C] = relu_mask * (conv_transpose(dy, w) (+ add)), the gradient w.r.t. the OUTPUT of a BatchNorm (+ ReLU) layer
// whose input was bn_x, plus that BatchNorm's backward sums slices[nslices][2][C] += (sum dz, sum dz * xhat).
// mask_mode 1: mask = mask_y > 0 (the block output);  2: mask = bf16(bn_x * scale + shift) > 0 (recomputed).
int mpr_conv_dgrad_bn(const void* dy, const void* w_dgrad, void* dz, const void* add, int mask_mode, const void* mask_y,
                      const void* bn_x, const float* mean, const float* invstd, const float* scale, const float* shift,
                      float* slices, int nslices, int prezeroed, int B, int H, int W, int C, int K, int R, int S, int sh,
                      int sw, int ph, int pw, void* stream) {
  MPR_REQUIRE(dy && w_dgrad && dz && bn_x && mean && invstd && slices && nslices > 0, "mpr_conv_dgrad_bn: null pointer");
  MPR_REQUIRE(mask_mode == 1 ? mask_y != nullptr : (mask_mode == 2 && scale && shift), "mpr_conv_dgrad_bn: bad mask mode %d", mask_mode);
  MPR_REQUIRE(mpr_conv_dgrad_bn_supported(B, H, W, C, K, R, S, sh, sw, ph, pw),
              "mpr_conv_dgrad_bn: geometry not served (3x3 / stride 1 / pad 1, K %% 64 == 0, enough rows)");
  MPR_REQUIRE((long long)B * H * W * C < (1ll << 31), "mpr_conv_dgrad_bn: tensor exceeds 2^31 elements");
  WinBnb bnb = {mask_mode, mask_y, bn_x, mean, invstd, scale, shift, slices, nslices, prezeroed};
  void* tok = mpr_prof_begin(7, 2.0 * (double)B * H * W * C * 9.0 * K, (hipStream_t)stream);
  mpr_prof_bytes(tok, 2.0 * ((double)B * H * W * K + 9.0 * C * K + (double)B * H * W * C * (add ? 3 : 2) +
                             (mask_mode == 1 ? (double)B * H * W * C : 0.0)));
  const int rc = mpr_win_launch(true, dy, w_dgrad, dz, add, nullptr, B, H, W, K, C, (hipStream_t)stream, &bnb);
  mpr_prof_end(tok, (hipStream_t)stream);
  return rc;
}

}  // extern "C"
