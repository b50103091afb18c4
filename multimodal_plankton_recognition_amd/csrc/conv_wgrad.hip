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
  unsigned x_bytes, dy_bytes;   // extents for the buffer descriptors of the LDS-DMA kernel
  int Mpix, Ng, cps, ntm, ntn, ngroups;
  unsigned rowpat;              // sum_r 1 << (r*S): one bit per filter row (tap masks of the LDS-DMA kernel)
  FastDiv div_pq, div_q, div_c, div_s;
  unsigned long long* stamps;   // timing experiments: 4 x uint64 (100 MHz) per workgroup of the LDS-DMA kernel, or NULL
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
  // XCD-aware placement: the ntn column tiles of one (row tile, pixel split) group stream the SAME dy pixels and
  // overlapping x pixels, so they are put on one XCD (blocks b and b+8 share an XCD) in consecutive slots: the
  // group's operands then come from HBM once per group instead of once per column tile.  Speed only.
  const int slot = blockIdx.x >> 3;
  const int nt = slot % p.ntn;
  const int grp = (slot / p.ntn) * 8 + (blockIdx.x & 7);
  if (grp >= p.ngroups) return;
  const int mt = grp % p.ntm;
  const int split = grp / p.ntm;
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

// ------------------------------------------------------------------------------------------------
// LDS-DMA variant (C % 64 == 0, K % 64 == 0, <= 31 taps): 64-pixel chunks go global -> LDS with
// buffer_load_dwordx4 ... lds into a 2-deep ring (one counted wait + one raw barrier per chunk, 16 MFMAs per
// wave in between).  The DMA image is lane-linear, so rows cannot be padded; the transposing reads are kept
// conflict-free by XOR-ing the 64-B segment index with a function of the pixel row, applied on the DMA
// source side and in the read address.  Pixel decoding (b,p,q) -> source offset + per-tap validity bits is
// done once per pixel per wave (one pixel per lane, in registers); a DMA lane then needs two wave shuffles +
// 3 VALU: offset = entry.base + lane-constant tap offset, validity = entry.mask & tap bit (invalid lanes get
// an out-of-range buffer offset -> hardware zero fill).
template <int N>
__device__ __forceinline__ void wg_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ s16x4 wg_read_tr(uint32_t lds_addr) {   // ds_read_b64_tr_b16 the compiler cannot see
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(lds_addr) : "memory");
  return v;
}
template <int CH>   // 64-B segment swizzle for unpadded rows of CH 16-B chunks (row strides 128 / 256 / 384 B)
__device__ __forceinline__ int wg_seg_xor(int row) {
  return CH == 16 ? (row & 3) : ((row >> 1) & 1);
}

// 16x16x32 form (round 3): 64-B segments XORed with the low row bits as below, and the two 32-B units of a segment swapped by
// bit 2 of the row -- a half-wave of the transposing reads then covers eight consecutive pixel rows x 32 B in eight distinct
// 32-B bank groups whatever the row stride (rows of 512 B take the segment key from two row bits instead of one)
template <int CH>
__device__ __forceinline__ int wg_seg_xor16(int row) {
  return CH % 16 == 0 ? (row & 3) : ((row >> 1) & 1);
}

// M16: the products as v_mfma_f32_16x16x32_bf16 (the chip holds a higher clock on that shape: DESIGN.md section 9): a chunk is two
// 32-pixel steps, a wave's 64 x 64 block 4 x 4 sub-tiles; lane group g = lane / 16 contracts over pixels 4 g .. 4 g + 3 (first
// read) and 16 + 4 g .. (second read) of a step for BOTH operands (conv_wgrad_win.hip).
template <int WM, int WN, bool M16 = false>
__global__ __launch_bounds__(64 * WM * WN) void conv_wgrad_dma_kernel(const WgradParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NW = WM * WN;
  constexpr int BM = 64 * WM, BN = 64 * WN, BKP = 64;
  constexpr int ASTR = BM * 2, BSTR = BN * 2;
  constexpr int CA = BM / 8, CB = BN / 8;
  constexpr int A_BYTES = BKP * ASTR, STAGE = BKP * (ASTR + BSTR);
  constexpr int A_INSTR = A_BYTES / 1024, B_INSTR = BKP * BSTR / 1024;
  constexpr int A_IT = (A_INSTR + NW - 1) / NW, B_IT = (B_INSTR + NW - 1) / NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;   // LDS byte address

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  // XCD-aware placement: the ntn column tiles of one (row tile, pixel split) group stream the SAME dy pixels and
  // overlapping x pixels, so they are put on one XCD (blocks b and b+8 share an XCD) in consecutive slots: the
  // group's operands then come from HBM once per group instead of once per column tile.  Speed only.
  const int slot = blockIdx.x >> 3;
  const int nt = slot % p.ntn;
  const int grp = (slot / p.ntn) * 8 + (blockIdx.x & 7);
  if (grp >= p.ngroups) return;
  const int mt = grp % p.ntm;
  const int split = grp / p.ntm;
  const int m0 = mt * BM, n0 = nt * BN;
  const int pix_begin = split * p.cps * BKP;
  int pix_end = pix_begin + p.cps * BKP;
  if (pix_end > p.Mpix) pix_end = p.Mpix;
  const int nchunks = (pix_end - pix_begin + BKP - 1) / BKP;
#define WG_STAMP(k) do { if (p.stamps && tid == 0) p.stamps[(size_t)blockIdx.x * 4 + (k)] = wall_clock64(); } while (0)
  WG_STAMP(0);

  // ---- lane-constant DMA coordinates
  uint32_t a_off[A_IT];
  int a_row[A_IT];
#pragma unroll
  for (int j = 0; j < A_IT; ++j) {
    const int q = (wid + j * NW) * 64 + lane;
    const int row = q / CA, pc = q % CA;
    int lc = (((pc >> 2) ^ (M16 ? wg_seg_xor16<CA>(row) : wg_seg_xor<CA>(row))) << 2) | (pc & 3);
    if (M16) lc ^= ((row >> 2) & 1) << 1;
    a_row[j] = row;
    a_off[j] = (m0 + lc * 8 < p.K) ? (uint32_t)((row * p.K + m0 + lc * 8) * 2) : 0xFFFFFFF0u;
  }
  uint32_t b_off[B_IT], b_bit[B_IT];
  int b_row[B_IT];
#pragma unroll
  for (int j = 0; j < B_IT; ++j) {
    const int q = (wid + j * NW) * 64 + lane;
    const int row = q / CB, pc = q % CB;
    int lc = (((pc >> 2) ^ (M16 ? wg_seg_xor16<CB>(row) : wg_seg_xor<CB>(row))) << 2) | (pc & 3);
    if (M16) lc ^= ((row >> 2) & 1) << 1;
    const int n = n0 + lc * 8;
    b_row[j] = row;
    b_off[j] = 0;
    b_bit[j] = 0;
    if (n < p.Ng) {
      const uint32_t t = fdiv(n, p.div_c);
      const int c = n - t * p.C;
      const uint32_t tr = fdiv(t, p.div_s);
      const int ts = t - tr * p.S;
      b_off[j] = (uint32_t)((((int)tr * p.W + ts) * p.C + c) * 2);
      b_bit[j] = 1u << t;
    }
  }
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dy_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);

  // Every wave decodes the chunk's 64 pixels itself, one pixel per lane, and keeps the result in registers;
  // a DMA lane fetches the entry of ITS pixel row with a wave shuffle (no LDS table: hipcc would fence every
  // LDS read behind the in-flight LDS-DMA with s_waitcnt vmcnt(0)).
  uint32_t ex = 0, ey = 0;      // {source byte offset of tap (0,0), tap-validity bits} of pixel (chunk base + lane)
  auto decode_chunk = [&](int ci) {     // software-pipelined: runs while the previous chunk's DMA is in flight
    const int pix = pix_begin + ci * BKP + lane;
    ex = 0;
    ey = 0;
    if (pix < pix_end) {
      const uint32_t b = fdiv(pix, p.div_pq);
      const uint32_t rem = pix - b * (uint32_t)(p.P * p.Q);
      const uint32_t pp = fdiv(rem, p.div_q);
      const uint32_t qq = rem - pp * p.Q;
      const int ih0 = (int)pp * p.sh - p.ph, iw0 = (int)qq * p.sw - p.pw;
      ex = (uint32_t)((((int)b * p.H + ih0) * p.W + iw0) * p.C * 2);
      // valid taps form an index range per axis: columns [slo, shi), rows [rlo, rhi) -> closed-form bit mask
      const int slo = max(0, -iw0), shi = min(p.S, p.W - iw0);
      const int rlo = max(0, -ih0), rhi = min(p.R, p.H - ih0);
      if (shi > slo && rhi > rlo) {
        const uint32_t cm = ((1u << shi) - 1u) & ~((1u << slo) - 1u);
        const uint32_t rows = p.rowpat & ((1u << (rhi * p.S)) - 1u) & ~((1u << (rlo * p.S)) - 1u);
        ey = cm * rows;
      }
    }
  };
  auto issue_chunk = [&](int ci) {      // uses the (ex, ey) decoded for chunk ci
    unsigned char* sa = smem + (ci & 1) * STAGE;
    unsigned char* sb = sa + A_BYTES;
    const int kb = pix_begin + ci * BKP;
    const int soff_a = kb * p.K * 2;
#pragma unroll
    for (int j = 0; j < A_IT; ++j) {
      const int I = wid + j * NW;
      if (A_INSTR % NW == 0 || I < A_INSTR) {
        const uint32_t v = (kb + a_row[j] < pix_end) ? a_off[j] : 0xFFFFFFF0u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(sa + I * 1024), 16, v,
                                                 soff_a, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < B_IT; ++j) {
      const int I = wid + j * NW;
      if (B_INSTR % NW == 0 || I < B_INSTR) {
        const uint32_t rx = __shfl(ex, b_row[j], 64), ry = __shfl(ey, b_row[j], 64);
        const uint32_t v = (ry & b_bit[j]) ? rx + b_off[j] : 0xFFFFFFF0u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (__attribute__((address_space(3))) void*)(sb + I * 1024), 16, v, 0,
                                                 0, 0);
      }
    }
  };

  f32x16 acc[M16 ? 1 : 2][M16 ? 1 : 2];
  f32x4 acc16[M16 ? 4 : 1][M16 ? 4 : 1];      // [sub-tile of 16 output channels][sub-tile of 16 (tap, channel) columns]
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

  // transposing-read lane geometry (see conv_wgrad_kernel); the 64-B segment XOR is lane-constant because the
  // pixel row of a read is (multiple of 4) + lq
  const int g16 = lane >> 4, lq = (lane & 15) >> 2, lp = lane & 3;
  uint32_t a_rd[M16 ? 4 : 2], b_rd[M16 ? 4 : 2];
  if constexpr (M16) {
    // sub-tile t (16 columns = a 32-B unit): segment wm * 2 + t / 2, unit t % 2; rows 4 g16 + lq (+ 16: second read; + 32: next step)
    const int rowl = 4 * g16 + lq;
    const int inseg = (4 * lp) * 2;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      a_rd[t] = rowl * ASTR + (((wm * 2 + (t >> 1)) ^ wg_seg_xor16<CA>(lq)) << 6) + (((t & 1) ^ (g16 & 1)) << 5) + inseg;
      b_rd[t] = A_BYTES + rowl * BSTR + (((wn * 2 + (t >> 1)) ^ wg_seg_xor16<CB>(lq)) << 6) + (((t & 1) ^ (g16 & 1)) << 5) + inseg;
    }
  } else {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int rowl = 8 * (g16 >> 1) + lq;
    const int inseg = (16 * (g16 & 1) + 4 * lp) * 2;
    a_rd[t] = rowl * ASTR + (((wm * 2 + t) ^ wg_seg_xor<CA>(lq)) << 6) + inseg;
    b_rd[t] = A_BYTES + rowl * BSTR + (((wn * 2 + t) ^ wg_seg_xor<CB>(lq)) << 6) + inseg;
  }
  }

  WG_STAMP(1);
  if (nchunks > 0) {
    decode_chunk(0);
    issue_chunk(0);
    decode_chunk(1);
  }
  for (int ci = 0; ci < nchunks; ++ci) {
    wg_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (ci + 1 < nchunks) {
      issue_chunk(ci + 1);
      decode_chunk(ci + 2);
    }
    const uint32_t a = lds0 + (ci & 1) * STAGE;
    // software-pipelined k-steps: wait for the fragments of step ks, put the transposing reads of step ks+1 in flight
    // (their own registers), then the four MFMAs of step ks.  The reads are INLINE ASM: through the builtin hipcc
    // cannot tell them from the LDS-DMA writes in flight and puts s_waitcnt vmcnt(0) in front -- every chunk would wait
    // for the next chunk's DMA it has just issued.  Completion is by hand: the s_waitcnt lgkmcnt(0) below carries the
    // destination registers as in/out operands, so every consumer is ordered behind it.
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    if constexpr (M16) {
      // (the 16-wave tile has 128 VGPRs per wave: one fragment set -- its four waves per SIMD hide the reads of the next step
      //  behind each other's MFMAs)
      constexpr int NB = NW >= 16 ? 1 : 2;
      s16x4 ra[NB][4][2], rb[NB][4][2];      // [buffer][sub-tile][half]
      auto load_frags = [&](int buf, int ks) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const uint32_t pa = a + a_rd[t] + ks * 32 * ASTR;
          const uint32_t pb = a + b_rd[t] + ks * 32 * BSTR;
          ra[buf][t][0] = wg_read_tr(pa);
          ra[buf][t][1] = wg_read_tr(pa + 16 * ASTR);
          rb[buf][t][0] = wg_read_tr(pb);
          rb[buf][t][1] = wg_read_tr(pb + 16 * BSTR);
        }
      };
      load_frags(0, 0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int cur = NB == 2 ? (ks & 1) : 0;
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(ra[cur][0][0]), "+v"(ra[cur][0][1]), "+v"(ra[cur][1][0]), "+v"(ra[cur][1][1]),
                       "+v"(ra[cur][2][0]), "+v"(ra[cur][2][1]), "+v"(ra[cur][3][0]), "+v"(ra[cur][3][1]),
                       "+v"(rb[cur][0][0]), "+v"(rb[cur][0][1]), "+v"(rb[cur][1][0]), "+v"(rb[cur][1][1]),
                       "+v"(rb[cur][2][0]), "+v"(rb[cur][2][1]), "+v"(rb[cur][3][0]), "+v"(rb[cur][3][1])
                     :
                     : "memory");
        if (NB == 2 && ks < 1) load_frags(cur ^ 1, ks + 1);
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 af[4], bfr[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          af[t] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(ra[cur][t][0], ra[cur][t][1], 0, 1, 2, 3, 4, 5, 6, 7));
          bfr[t] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(rb[cur][t][0], rb[cur][t][1], 0, 1, 2, 3, 4, 5, 6, 7));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc16[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (NB == 1 && ks < 1) load_frags(0, ks + 1);
      }
    } else {
    s16x4 ra[2][2][2], rb[2][2][2];      // [buffer][tile][half]
    auto load_frags = [&](int buf, int ks) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const uint32_t pa = a + a_rd[t] + ks * 16 * ASTR;
        const uint32_t pb = a + b_rd[t] + ks * 16 * BSTR;
        ra[buf][t][0] = wg_read_tr(pa);
        ra[buf][t][1] = wg_read_tr(pa + 4 * ASTR);
        rb[buf][t][0] = wg_read_tr(pb);
        rb[buf][t][1] = wg_read_tr(pb + 4 * BSTR);
      }
    };
    load_frags(0, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int cur = ks & 1;
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(ra[cur][0][0]), "+v"(ra[cur][0][1]), "+v"(ra[cur][1][0]), "+v"(ra[cur][1][1]),
                     "+v"(rb[cur][0][0]), "+v"(rb[cur][0][1]), "+v"(rb[cur][1][0]), "+v"(rb[cur][1][1])
                   :
                   : "memory");
      if (ks < 3) load_frags(cur ^ 1, ks + 1);
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        af[t] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(ra[cur][t][0], ra[cur][t][1], 0, 1, 2, 3, 4, 5, 6, 7));
        bfr[t] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(rb[cur][t][0], rb[cur][t][1], 0, 1, 2, 3, 4, 5, 6, 7));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    }
  }

  WG_STAMP(2);
  const int ln = lane & 31, lh = lane >> 5;
  if constexpr (M16) {
    // 16 x 16 sub-tiles: lane -> column lane % 16, rows 4 (lane / 16) + e
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + (lane & 15);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = m0 + wm * 64 + i * 16 + 4 * (lane >> 4) + e;
          if (k < p.K && n < p.Ng) atomicAdd(p.dw + (size_t)k * p.Ng + n, acc16[i][j][e]);
        }
      }
  } else {
#pragma unroll
  for (int i = 0; i < (M16 ? 1 : 2); ++i)
#pragma unroll
    for (int j = 0; j < (M16 ? 1 : 2); ++j) {
      const int n = n0 + wn * 64 + j * 32 + ln;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int k = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (k < p.K && n < p.Ng) atomicAdd(p.dw + (size_t)k * p.Ng + n, acc[i][j][e]);
      }
    }
  }
  WG_STAMP(3);
#undef WG_STAMP
#endif   // __HIP_DEVICE_COMPILE__
}

static int g_wgrad_target_wgs = 512;
extern "C" int mpr_conv_set_wgrad_target_wgs(int n) {   // tuning knob; returns the previous value
  const int old = g_wgrad_target_wgs;
  g_wgrad_target_wgs = n;
  return old;
}
// (default 1 since round 2: alone the 256 x 256 tile is slower on three of four transformer shapes, but INSIDE the step --
//  beside the data-gradient chain, where fewer and fatter workgroups cost less CU-time -- ViT-B/16 + ProfileTransformer at
//  batch 128 runs 58.5 ms per step with it against 61.2 (tile 2 / 3: 60.0))
static int g_wgrad_tile = 1;
extern "C" int mpr_conv_set_wgrad_tile(int v) {   // tuning knob (see mpr_conv_wgrad); returns the previous value
  const int old = g_wgrad_tile;
  g_wgrad_tile = v;
  return old;
}
static unsigned long long* g_wgrad_stamps = nullptr;
extern "C" int mpr_conv_debug_wgrad_stamps(void* buf) {
  g_wgrad_stamps = (unsigned long long*)buf;
  return 0;
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

static int g_wgrad_dma_min_pix = 16384;
// 1: the LDS-DMA gather kernel on v_mfma_f32_16x16x32_bf16.  Built and tested (tests/test_wgrad_gather_mfma_gpu.py), NOT the default:
// C3 9.334 vs 9.329 ms per step, C5's per-GPU share 30.52 vs 30.32 (its 16-wave tile has registers for ONE fragment set in this form)
static int g_wgrad_mfma16 = 0;

extern "C" {

// sliding-window kernel for 3x3 / stride 1 / pad 1 (conv_wgrad_win.hip)
bool mpr_wgw_eligible(long long Mpix, int H, int W, int C, int K, int R, int S, int sh, int sw, int ph, int pw,
                      long long min_pix);
int mpr_wgw_launch(const void* x, const void* dy, float* dw, int B, int H, int W, int C, int K, int target_wgs,
                   float* scratch, long long scratch_floats, hipStream_t st);
void mpr_wgw_take_scratch(float** buf, long long* floats);

int mpr_conv_debug_wgrad_mfma16(int on) {   // timing / test knob; returns the previous value
  const int old = g_wgrad_mfma16;
  g_wgrad_mfma16 = on;
  return old;
}

int mpr_conv_set_wgrad_dma_min_pixels(int pixels) {   // tuning / test knob; returns the previous value
  const int old = g_wgrad_dma_min_pix;
  g_wgrad_dma_min_pix = pixels;
  return old;
}

// dw_oihw[K,C,R,S] (fp32) = sum over the batch;  workspace: K*R*S*C floats (zeroed here).
// dw_oihw may be NULL: see below.
// Every choice of a launch as an ARGUMENT (round 3; ops.py uses this form: nothing process-global is set "for the next call"):
//   scratch / scratch_floats: device floats lent to THIS call for the window kernel's partial tiles (NULL: fp32 atomics);
//   target_wgs: workgroup-count target of the split over pixels (<= 0: the library default / mpr_conv_set_wgrad_target_wgs);
//   kernel: -1 = by geometry, 0 = never the sliding-window kernel (gather / register-staged), 1 = as -1.
static int wgrad_impl(const void* x, const void* dy, float* workspace, float* dw_oihw, int accumulate, int B, int H, int W,
                      int C, int K, int R, int S, int sh, int sw, int ph, int pw, float* scratch, long long scratch_floats,
                      int target_wgs, int kernel, void* stream);

int mpr_conv_wgrad_ex(const void* x, const void* dy, float* workspace, float* dw_oihw, int accumulate, int B, int H, int W,
                      int C, int K, int R, int S, int sh, int sw, int ph, int pw, void* scratch, long long scratch_floats,
                      int target_wgs, int kernel, void* stream) {
  return wgrad_impl(x, dy, workspace, dw_oihw, accumulate, B, H, W, C, K, R, S, sh, sw, ph, pw, (float*)scratch,
                    scratch ? scratch_floats : 0, target_wgs, kernel, stream);
}

int mpr_conv_wgrad(const void* x, const void* dy, float* workspace, float* dw_oihw, int accumulate, int B,
                   int H, int W, int C, int K, int R, int S, int sh, int sw, int ph, int pw, void* stream) {
  // the one-shot scratch loan (mpr_conv_set_wgrad_scratch) belongs to THIS call whatever happens next: taken before any
  // argument check can return, or it would stay armed for an unrelated later call
  float* scratch = nullptr;
  long long scratch_floats = 0;
  mpr_wgw_take_scratch(&scratch, &scratch_floats);
  return wgrad_impl(x, dy, workspace, dw_oihw, accumulate, B, H, W, C, K, R, S, sh, sw, ph, pw, scratch, scratch_floats, 0,
                    -1, stream);
}

static int wgrad_impl(const void* x, const void* dy, float* workspace, float* dw_oihw, int accumulate, int B, int H, int W,
                      int C, int K, int R, int S, int sh, int sw, int ph, int pw, float* scratch, long long scratch_floats,
                      int target_wgs, int kernel, void* stream) {
  const int target = target_wgs > 0 ? target_wgs : g_wgrad_target_wgs;
  MPR_REQUIRE(x && dy && workspace, "mpr_conv_wgrad: null pointer");
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
  p.stamps = g_wgrad_stamps;
  p.rowpat = 0;
  for (int r = 0; r < R && r * S < 32; ++r) p.rowpat |= 1u << (r * S);
  p.x_bytes = (unsigned)((size_t)B * H * W * C * 2);
  p.dy_bytes = (unsigned)((size_t)B * P * Q * K * 2);
  // dw_oihw == NULL: the gradient stays in `workspace` as [K][R][S][C] (the physical layout of a channels-last
  // weight); `accumulate` then adds into what is there instead of zeroing it first
  if (dw_oihw || !accumulate) MPR_HIP(hipMemsetAsync(workspace, 0, sizeof(float) * (size_t)K * p.Ng, st));

  if (kernel != 0 && mpr_wgw_eligible(p.Mpix, H, W, C, K, R, S, sh, sw, ph, pw, g_wgrad_dma_min_pix)) {
    // 3x3 / stride 1 / pad 1: sliding-window kernel (conv_wgrad_win.hip); profiler kind 8
    void* tok = mpr_prof_begin(8, 2.0 * (double)p.Mpix * (double)K * (double)p.Ng, st);
    mpr_prof_bytes(tok, (double)p.x_bytes + (double)p.dy_bytes + 4.0 * K * p.Ng);
    const int rc = mpr_wgw_launch(x, dy, workspace, B, H, W, C, K, target, scratch, scratch_floats, st);
    mpr_prof_end(tok, st);
    if (rc != MPR_OK) return rc;
    if (dw_oihw) {
      const int total = K * C * R * S;
      const int g2 = ceil_div(total, 256) < 2048 ? ceil_div(total, 256) : 2048;
      wgrad_unpack_kernel<<<g2, 256, 0, st>>>(workspace, dw_oihw, K, C, R, S, accumulate);
      MPR_LAUNCH_CHECK("wgrad_unpack_kernel");
    }
    return MPR_OK;
  }
  if (C % 64 == 0 && K % 64 == 0 && R * S <= 31 && p.Mpix >= g_wgrad_dma_min_pix) {
    // LDS-DMA ring kernel: 64-pixel chunks, 2 workgroups per CU
    int WM = K <= 64 ? 1 : 2;
    int WN = p.Ng <= 64 ? 1 : ((C == 64 && p.Ng % 192 == 0) ? 3 : 2);
    // big one-tap GEMMs (transformer linears): larger output tiles = more FLOP per LDS-DMA byte (g_wgrad_tile: 0 = 128 x 128,
    // 1 = 256 x 256 on 16 waves (default), 2 = 256 x 128, 3 = 128 x 256 on 8 waves)
    if (g_wgrad_tile && R * S == 1 && K >= 512 && p.Ng >= 512 && p.Mpix >= 8192) {
      WM = g_wgrad_tile == 3 ? 2 : 4;
      WN = g_wgrad_tile == 2 ? 2 : 4;
    }
    const int BM = 64 * WM, BN = 64 * WN;
    p.ntm = ceil_div(K, BM); p.ntn = ceil_div(p.Ng, BN);
    const int tiles = p.ntm * p.ntn;
    const int total_chunks = ceil_div(p.Mpix, 64);
    // ONE round of workgroups: 2 fit a CU (64 KB of LDS each), 512 slots on the chip.  A grid of 768 (1.5 rounds) left
    // the second round half empty -- measured: 360-400 of 512 slots alive on average, 25 % of the kernel time.
    int nsplit = target / tiles;
    while (nsplit > 8 && (p.ntm * nsplit) % 8) --nsplit;                          // whole XCD groups of 8 (no padding WGs)
    if (nsplit > ceil_div(total_chunks, 4)) nsplit = ceil_div(total_chunks, 4);   // >= 4 chunks per split
    if (nsplit < 1) nsplit = 1;
    p.cps = ceil_div(total_chunks, nsplit);
    nsplit = ceil_div(total_chunks, p.cps);
    p.ngroups = p.ntm * nsplit;
  dim3 grid(((p.ngroups + 7) / 8) * 8 * p.ntn);
    void* tok = mpr_prof_begin(2, 2.0 * (double)p.Mpix * (double)K * (double)p.Ng, st);
    mpr_prof_bytes(tok, (double)p.x_bytes + (double)p.dy_bytes + 4.0 * K * p.Ng);   // x, dy read once; dw written once
#define MPR_WGD(WM_, WN_)                                                                             \
  do {                                                                                                \
    const size_t smem_ = (size_t)2 * 64 * (128 * WM_ + 128 * WN_);                             \
    static bool attr_set = false;                                                                     \
    if (!attr_set) {                                                                                  \
      hipFuncSetAttribute((const void*)conv_wgrad_dma_kernel<WM_, WN_>,                               \
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                    \
      hipFuncSetAttribute((const void*)conv_wgrad_dma_kernel<WM_, WN_, true>,                         \
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                    \
      attr_set = true;                                                                                \
    }                                                                                                 \
    if (!g_wgrad_mfma16) conv_wgrad_dma_kernel<WM_, WN_><<<grid, 64 * WM_ * WN_, smem_, st>>>(p);     \
    else conv_wgrad_dma_kernel<WM_, WN_, true><<<grid, 64 * WM_ * WN_, smem_, st>>>(p);               \
  } while (0)
    if (WM == 1 && WN == 1) MPR_WGD(1, 1);
    else if (WM == 1 && WN == 2) MPR_WGD(1, 2);
    else if (WM == 1 && WN == 3) MPR_WGD(1, 3);
    else if (WM == 2 && WN == 1) MPR_WGD(2, 1);
    else if (WM == 2 && WN == 2) MPR_WGD(2, 2);
    else if (WM == 4 && WN == 4) MPR_WGD(4, 4);
    else if (WM == 4 && WN == 2) MPR_WGD(4, 2);
    else if (WM == 2 && WN == 4) MPR_WGD(2, 4);
    else MPR_WGD(2, 3);
#undef MPR_WGD
    mpr_prof_end(tok, st);
    MPR_LAUNCH_CHECK("conv_wgrad_dma_kernel");
    const int total = K * C * R * S;
    const int g2 = ceil_div(total, 256) < 2048 ? ceil_div(total, 256) : 2048;
    if (dw_oihw) {
      wgrad_unpack_kernel<<<g2, 256, 0, st>>>(workspace, dw_oihw, K, C, R, S, accumulate);
      MPR_LAUNCH_CHECK("wgrad_unpack_kernel");
    }
    return MPR_OK;
  }

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
  p.ngroups = p.ntm * nsplit;
  dim3 grid(((p.ngroups + 7) / 8) * 8 * p.ntn);
#define MPR_WG(WM_, WN_) conv_wgrad_kernel<WM_, WN_><<<grid, 64 * WM_ * WN_, 0, st>>>(p)
  void* tok = mpr_prof_begin(5, 2.0 * (double)p.Mpix * (double)K * (double)p.Ng, st);   // kind 5: register-staged wgrad
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
  if (dw_oihw) {
    wgrad_unpack_kernel<<<g2, 256, 0, st>>>(workspace, dw_oihw, K, C, R, S, accumulate);
    MPR_LAUNCH_CHECK("wgrad_unpack_kernel");
  }
  return MPR_OK;
}

}  // extern "C"
