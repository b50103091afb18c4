// Weight gradient of 3x3 / stride 1 / pad 1 convolutions on a sliding activation window -- the weight-gradient
// counterpart of conv_win.hip (the 13 body convolutions of ResNet-18 behind src/image_encoder.py:24).
//
//   dW[k][(r,s)][c] = sum over pixels G of dy[G][k] * x[G + (r-1)*(W+1) + (s-1)][c]
//
// in the padded raster of conv_win.hip (G = (b*(H+1) + h)*(W+1) + w: one zero column after every image row, one zero
// row after every image, so the tap shift is LINEAR in G and every out-of-image neighbour is a pad position that the
// DMA zero-fills).  The plain LDS-DMA kernel (conv_wgrad.hip) gathers a fresh [64 pixels][128 (tap, channel)] tile
// per tap block -- 32 KB of operands per 2.1 MFLOP, which pins it to the CU's global->LDS fill rate.  Here a workgroup
// owns ALL NINE taps of a (64*KH output channels) x (64 input channels) block: the 64 input channels of the pixels
// live in a 256-row LDS ring that slides along the raster (64 new rows per 64-pixel chunk), the nine taps read it at
// shifted rows, and only dy streams beside it: 16-24 KB per 4.7-9.4 MFLOP chunk (295-393 FLOP/B).
//
// Workgroup: 3 (filter row r) x 2 (channel half) x KH (64-wide slab of output channels) waves; a wave accumulates
// 2 x 3 MFMA tiles: [64 output channels] x [3 taps (r, 0..2) x 32 channels].  Both operands are pixel-major in
// memory while the contraction runs over pixels, so fragments come from ds_read_b64_tr_b16 as in conv_wgrad.hip.
// The pixel axis is split over workgroups; partial tiles are added into the [K][R][S][C] gradient with fp32 atomics.
#include "common.h"

struct WgwParams {
  const bf16_t* x;     // [B,H,W,C]
  const bf16_t* dy;    // [B,H,W,K]
  float* dw;           // [K][9*C]
  float* part;         // partial slices [nsplit][K][9*C] (plain stores + reduction pass) or NULL (atomics into dw)
  int H, W, C, K;
  int Wp, img, Gtot, halo8;   // W+1, (H+1)*(W+1), B*img, (W+2) rounded up to 8
  int Ng, ncb, nkt, cps, total_chunks;
  unsigned x_bytes, dy_bytes;
  FastDiv div_img, div_wp;
  int dbg;   // timing experiments: bit 0 = skip the atomic epilogue
  unsigned long long* probe;   // timing experiments: 8 x uint64 shader-clock sums per workgroup (wave 0), or NULL
};

template <int N>
__device__ __forceinline__ void wgw_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Transposing LDS read as INLINE ASM.  Through the builtin, hipcc cannot tell the read from the LDS-DMA writes in
// flight and guards it with s_waitcnt vmcnt(0) -- i.e. every chunk would wait for the NEXT chunk's DMA it has just
// issued, and nothing would overlap.  The asm form is invisible to that pass; completion is handled by hand:
// wgw_lds_wait() is the s_waitcnt lgkmcnt(0) and carries the destination registers as in/out operands, so every
// consumer is ordered behind it.
__device__ __forceinline__ s16x4 wgw_read_tr(uint32_t lds_addr) {
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(lds_addr) : "memory");
  return v;
}

// PS = 2 (64 output channels only): a chunk is 128 pixels and a second set of six waves works on its upper 64 -- twelve
// waves per workgroup as at KH = 2, where six (1.5 per SIMD) left every wave's read -> wait -> MFMA chain exposed: the loop
// took the same ~2700 cycles per 64 pixels with half the MFMA work.  The two halves' accumulators are summed through LDS
// at the end.  The ring holds two chunks + both halos then (384 rows; 512 allocated).
// PB = 2: every wave walks two 64-pixel blocks per chunk (8 k-steps between barriers instead of 4): the per-chunk fixed
// costs -- barrier skew, DMA issue, the first fragments' LDS latency, ~1400 cycles -- are paid half as often.
template <int KH, int PS = 1, int PB = 1>
__global__ __launch_bounds__(384 * KH * PS) void conv_wgrad_win_kernel(const WgwParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(KH * PS <= 2, "twelve waves at most");
  constexpr int NW = 6 * KH * PS;
  constexpr int CHUNK = 64 * PS * PB;          // pixels per chunk
  constexpr int RING = 512;                    // x ring rows (128 B each: 64 channels): 3 chunks + both halos (<= 512), power of two
  constexpr int NST = 3;                       // dy stages: chunk ci + 2 is in flight while chunk ci is multiplied
  constexpr int XBYTES = RING * 128;
  constexpr int DROW = 128 * KH;               // dy stage row bytes (64*KH output channels)
  constexpr int DSTAGE = CHUNK * DROW;
  constexpr int D_INSTR = DSTAGE / 1024;       // 8*KH*PS DMA instructions per dy chunk
  constexpr int D_IT = (D_INSTR + NW - 1) / NW;
  constexpr int XI_IT = (CHUNK / 8 + 16 + NW - 1) / NW;    // initial window: up to CHUNK + 2*64 rows
  constexpr int XC_IT = (CHUNK / 8 + NW - 1) / NW;         // per chunk: CHUNK new rows
  auto ring = [](int G) -> int { return G & (RING - 1); };
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [x ring 64 KB][dy stage 0][dy stage 1][dy stage 2]
  unsigned char* const dyst = smem + XBYTES;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;   // LDS byte address of smem

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid % 3, wh = (wid / 3) & 1;                      // filter row, channel half
  const int wk = PS == 1 ? wid / 6 : 0, wp = PS == 1 ? 0 : wid / 6;   // output-channel slab | pixel half of the chunk
  // consecutive work items share the pixel range; the hardware deals consecutive blocks to DIFFERENT XCDs, so the items are
  // renumbered to give every XCD one contiguous run (their dy / x rows then come from the same L2 lines)
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int cb = bid % p.ncb;
  const int kt = (bid / p.ncb) % p.nkt;
  const int split = bid / (p.ncb * p.nkt);
  const int c0 = split * p.cps;
  int c1 = c0 + p.cps;
  if (c1 > p.total_chunks) c1 = p.total_chunks;
  if (c0 >= c1) return;

  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dy_bytes, 0x00020000);

  // raster position -> source pixel (or "pad": zero-filled by an out-of-range offset)
  auto pixel_of = [&](int G, uint32_t& pix) -> bool {
    if (G < 0 || G >= p.Gtot) return false;
    const uint32_t b = fdiv(G, p.div_img);
    const uint32_t pp = G - b * p.img;
    const uint32_t hh = fdiv(pp, p.div_wp);
    const uint32_t ww = pp - hh * p.Wp;
    pix = (b * p.H + hh) * p.W + ww;
    return hh < (uint32_t)p.H && ww < (uint32_t)p.W;
  };
  // x rows [G8, G8+8) (G8 a multiple of 8) -> ring rows G8 & 255 ...; lane -> (row lane/8, physical chunk lane%8),
  // 64-B segment swizzle on the source side: logical chunk = ((pc >> 2) ^ ((row >> 1) & 1)) << 2 | (pc & 3)
  auto x_off = [&](int G8) -> uint32_t {
    const int G = G8 + (lane >> 3);
    const int row = ring(G);
    const int pc = lane & 7;
    const int lc = (((pc >> 2) ^ ((row >> 1) & 1)) << 2) | (pc & 3);
    uint32_t pix;
    return pixel_of(G, pix) ? pix * (uint32_t)(2 * p.C) + (uint32_t)((cb * 64 + lc * 8) * 2) : 0xFFFFFFF0u;
  };
  auto fire_x8 = [&](int G8, uint32_t v) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(
        rs_x, (__attribute__((address_space(3))) void*)(smem + ring(G8) * 128), 16, v, 0, 0, 0);
  };
  auto issue_x8 = [&](int G8) { fire_x8(G8, x_off(G8)); };
  // dy rows of chunk ci -> stage ci % NST: instruction I covers 1024 / DROW rows
  auto dy_off = [&](int ci, int I) -> uint32_t {
    constexpr int CH = DROW / 16;                       // 16-B chunks per row (8 or 16)
    const int q = I * 64 + lane;
    const int row = q / CH, pc = q % CH;
    const int key = CH == 16 ? (row & 3) : ((row >> 1) & 1);
    const int lc = (((pc >> 2) ^ key) << 2) | (pc & 3);
    uint32_t pix;
    return pixel_of(ci * CHUNK + row, pix) ? pix * (uint32_t)(2 * p.K) + (uint32_t)((kt * 64 * KH + lc * 8) * 2)
                                        : 0xFFFFFFF0u;
  };
  // The source offsets of a chunk's DMA (two fastdivs per lane and instruction) are computed ONE CHUNK AHEAD, underneath
  // the MFMAs of the previous chunk; after the barrier only the buffer_load ... lds instructions themselves remain.
  // (Computed right there they cost every wave ~900 cycles with the matrix pipe idle: all waves sit at the same point.)
  uint32_t vx[XC_IT], vd[D_IT];
  auto prep_chunk = [&](int ci) {       // chunk ci: its 64 leading-edge ring rows and its dy rows
    const int lo = ci * CHUNK + p.halo8;                  // (first raster row behind the window of chunk ci - 1)
#pragma unroll
    for (int j = 0; j < XC_IT; ++j) {
      const int I = wid + j * NW;
      vx[j] = ((CHUNK / 8) % NW == 0 || I < CHUNK / 8) ? x_off(lo + 8 * I) : 0xFFFFFFF0u;
    }
#pragma unroll
    for (int j = 0; j < D_IT; ++j) {
      const int I = wid + j * NW;
      vd[j] = (D_INSTR % NW == 0 || I < D_INSTR) ? dy_off(ci, I) : 0xFFFFFFF0u;
    }
  };
  auto fire_chunk = [&](int ci) {
    const int lo = ci * CHUNK + p.halo8;
    unsigned char* st = dyst + (ci % NST) * DSTAGE;
#pragma unroll
    for (int j = 0; j < XC_IT; ++j) {
      const int I = wid + j * NW;
      if ((CHUNK / 8) % NW == 0 || I < CHUNK / 8) fire_x8(lo + 8 * I, vx[j]);
    }
#pragma unroll
    for (int j = 0; j < D_IT; ++j) {
      const int I = wid + j * NW;
      if (D_INSTR % NW == 0 || I < D_INSTR)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_d, (__attribute__((address_space(3))) void*)(st + I * 1024), 16, vd[j],
                                                 0, 0, 0);
    }
  };

  f32x16 acc[2][3];     // [m-tile of 32 output channels][tap s]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][s][e] = 0.f;

  // transposing-read lane geometry (conv_wgrad.hip): a 16-lane group reads a 4 (pixel) x 16 (column) block
  const int g16 = lane >> 4, lq = (lane & 15) >> 2, lp = lane & 3;
  const int rowl = 8 * (g16 >> 1) + lq;                        // pixel row inside a 16-deep k-step (+4: second read)
  const int inseg = (16 * (g16 & 1) + 4 * lp) * 2;             // byte inside a 64-B segment
  // dy fragment offsets inside a stage (pixel rows are chunk-local: key is lane-constant)
  uint32_t a_rd[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int seg = wk * 2 + i;                                // 64-B segment = 32 output channels
    const int key = KH == 2 ? (lq & 3) : ((lq >> 1) & 1);
    a_rd[i] = (wp * 64 * PB + rowl) * DROW + ((seg ^ key) << 6) + inseg;
  }

  // ---- prologue: window of the first chunk + its dy
  {
    const int lo = c0 * CHUNK - p.halo8, hi = c0 * CHUNK + CHUNK + p.halo8;
#pragma unroll
    for (int j = 0; j < XI_IT; ++j) {
      const int G8 = lo + 8 * (wid + j * NW);
      if (G8 < hi) issue_x8(G8);
    }
    prep_chunk(c0);          // (only its dy half is used: the window above already holds chunk c0's rows)
    {
      unsigned char* st = dyst + (c0 % NST) * DSTAGE;
#pragma unroll
      for (int j = 0; j < D_IT; ++j) {
        const int I = wid + j * NW;
        if (D_INSTR % NW == 0 || I < D_INSTR)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_d, (__attribute__((address_space(3))) void*)(st + I * 1024), 16, vd[j],
                                                   0, 0, 0);
      }
    }
    // two chunks of look-ahead: chunk c0 + 1 goes out here, the offsets of c0 + 2 are prepared for the first iteration
    if (c0 + 1 < c1) {
      prep_chunk(c0 + 1);
      fire_chunk(c0 + 1);
    }
    if (c0 + 2 < c1) prep_chunk(c0 + 2);
    wgw_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  unsigned long long pr_wait = 0, pr_bar = 0, pr_comp = 0, pr_iss = 0;
#define WGW_NOW() (p.probe ? __builtin_readcyclecounter() : 0ull)
  const unsigned long long pr_t0 = WGW_NOW();
  // Loop invariant at the top of iteration ci: chunk ci's operands are in LDS for EVERY wave (confirmed by the previous
  // barrier), chunk ci + 1's are in flight (issued one iteration ago).  So the first fragment reads of chunk ci go out
  // BEFORE this iteration's wait + barrier -- which only confirm chunk ci + 1 and release the stage / ring rows that chunk
  // ci + 2 overwrites (last read in iteration ci - 1) -- and the matrix pipe is not drained at every chunk boundary: with a
  // two-stage ring the barrier sat between the DMA wait and the first LDS read, ~1100 idle cycles per 2300-cycle chunk.
  for (int ci = c0; ci < c1; ++ci) {
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const uint32_t da = lds0 + XBYTES + (ci % NST) * DSTAGE;
    // software-pipelined k-steps: wait for the fragments of step ks, put the reads of step ks+1 in flight, then the
    // six MFMAs of step ks
    s16x4 ra[2][2][2], rb[2][3][2];      // [buffer][tile][half]
    auto issue_reads = [&](int buf, int ks) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const uint32_t pa = da + a_rd[i] + ks * 16 * DROW;
        ra[buf][i][0] = wgw_read_tr(pa);
        ra[buf][i][1] = wgw_read_tr(pa + 4 * DROW);
      }
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int d = (wr - 1) * p.Wp + (s - 1);
        const int r0 = ring(ci * CHUNK + wp * 64 * PB + ks * 16 + rowl + d);
        const int r1 = (r0 + 4) & (RING - 1);
        const int lowb = ((wh ^ ((r0 >> 1) & 1)) << 6) + inseg;       // (r0 + 4) >> 1 has the parity of r0 >> 1
        rb[buf][s][0] = wgw_read_tr(lds0 + r0 * 128 + lowb);
        rb[buf][s][1] = wgw_read_tr(lds0 + r1 * 128 + lowb);
      }
    };
    issue_reads(0, 0);
    const unsigned long long q0 = WGW_NOW();
    wgw_wait_vmcnt<0>();
    const unsigned long long q1 = WGW_NOW();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const unsigned long long q2 = WGW_NOW();
    pr_wait += q1 - q0;
    pr_bar += q2 - q1;
    // chunk ci + 2: its new ring rows at the leading edge + its dy (stage and rows were last read in iteration ci - 1)
    if (ci + 2 < c1) fire_chunk(ci + 2);
    const unsigned long long q3 = WGW_NOW();
    pr_iss += q3 - q2;
#pragma unroll
    for (int ks = 0; ks < 4 * PB; ++ks) {
      const int cur = ks & 1;
      // (per-wave probe, scripts/wgw_probe_waves.py: the three waves of a SIMD are served oldest first -- 2196 / 2491 / 2818
      //  cycles of "compute" per chunk, barrier waits 820 / 460 / 96 -- and the youngest runs its last k-steps alone at
      //  LDS-latency pace: ~900 of the 3200 cycles per chunk.  Rotating s_setprio per k-step made it worse: 3700.)
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(ra[cur][0][0]), "+v"(ra[cur][0][1]), "+v"(ra[cur][1][0]), "+v"(ra[cur][1][1]),
                     "+v"(rb[cur][0][0]), "+v"(rb[cur][0][1]), "+v"(rb[cur][1][0]), "+v"(rb[cur][1][1]),
                     "+v"(rb[cur][2][0]), "+v"(rb[cur][2][1])
                   :
                   : "memory");
      if (ks < 4 * PB - 1) issue_reads(cur ^ 1, ks + 1);
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 af[2], bfr[3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        af[i] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(ra[cur][i][0], ra[cur][i][1], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
      for (int s = 0; s < 3; ++s)
        bfr[s] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(rb[cur][s][0], rb[cur][s][1], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < 3; ++s)
          acc[i][s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[s], acc[i][s], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (ks == 1 && ci + 3 < c1) prep_chunk(ci + 3);      // address arithmetic underneath the MFMAs just issued
    }
    pr_comp += WGW_NOW() - q3;
  }
  if (p.probe && (p.dbg & 2) && lane == 0) {      // per-wave probe: 16 x 8 uint64 per workgroup (mpr_conv_set_wgrad_window(1 | 2 << 8))
    unsigned long long* o = p.probe + ((size_t)blockIdx.x * 16 + wid) * 8;
    o[0] = WGW_NOW() - pr_t0; o[1] = pr_wait; o[2] = pr_bar; o[3] = pr_iss; o[4] = pr_comp; o[5] = (unsigned long long)(c1 - c0);
  } else
  if (p.probe && tid == 0) {
    unsigned long long* o = p.probe + (size_t)blockIdx.x * 8;
    o[0] = WGW_NOW() - pr_t0; o[1] = pr_wait; o[2] = pr_bar; o[3] = pr_iss; o[4] = pr_comp; o[5] = (unsigned long long)(c1 - c0);
  }
#undef WGW_NOW

  if (p.dbg & 1) {      // timing experiment: no epilogue (one conditional store keeps the accumulators alive)
    if (acc[0][0][0] + acc[1][2][5] == 12345.f) p.dw[0] = 1.f;
    return;
  }
  if (PS == 2) {
    // the upper pixel half's accumulators join the lower half's through LDS: 3 tiles (72 KB) per pass
    float* const red = reinterpret_cast<float*>(smem);
    const int role = wid % 6;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __syncthreads();
      if (wp == 1) {
#pragma unroll
        for (int s2 = 0; s2 < 3; ++s2)
#pragma unroll
          for (int e = 0; e < 16; ++e) red[((role * 3 + s2) * 16 + e) * 64 + lane] = acc[i][s2][e];
      }
      __syncthreads();
      if (wp == 0) {
#pragma unroll
        for (int s2 = 0; s2 < 3; ++s2)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][s2][e] += red[((role * 3 + s2) * 16 + e) * 64 + lane];
      }
    }
    if (wp == 1) return;
  }
  // D[k (regs)][n (lanes)], 128 B contiguous per half-wave: plain stores into this pixel split's slice (summed by
  // wgw_reduce_kernel) when the caller lent scratch memory -- the chip adds ~1.3 TB/s of fp32 atomics but stores
  // ~6 TB/s, and every launch emits 256 CUs x 295 KB of accumulators -- else fp32 atomics into [K][(r,s)][C]
  const int ln = lane & 31, lh = lane >> 5;
  float* const slice = p.part ? p.part + (size_t)split * p.K * p.Ng : nullptr;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int n = (wr * 3 + s) * p.C + cb * 64 + wh * 32 + ln;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int k = kt * 64 * KH + wk * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (slice) slice[(size_t)k * p.Ng + n] = acc[i][s][e];
        else atomicAdd(p.dw + (size_t)k * p.Ng + n, acc[i][s][e]);
      }
    }
#endif   // __HIP_DEVICE_COMPILE__
}

// dw[i] += sum over the pixel splits of part[s][i]   (float4 per thread; the slices are L2 / Infinity-Cache warm)
__global__ __launch_bounds__(256) void wgw_reduce_kernel(const float4* __restrict__ part, int nsplit, float4* __restrict__ dw,
                                                          int n4) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
    float4 a = dw[i];
    int s = 0;
    for (; s + 4 <= nsplit; s += 4) {
      const float4 v0 = part[(size_t)s * n4 + i], v1 = part[(size_t)(s + 1) * n4 + i];
      const float4 v2 = part[(size_t)(s + 2) * n4 + i], v3 = part[(size_t)(s + 3) * n4 + i];
      a.x += (v0.x + v1.x) + (v2.x + v3.x); a.y += (v0.y + v1.y) + (v2.y + v3.y);
      a.z += (v0.z + v1.z) + (v2.z + v3.z); a.w += (v0.w + v1.w) + (v2.w + v3.w);
    }
    for (; s < nsplit; ++s) {
      const float4 v = part[(size_t)s * n4 + i];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    dw[i] = a;
  }
}

static int g_wgw_on = 1;
static float* g_wgw_scratch = nullptr;        // lent by the caller for the NEXT launch (mpr_conv_set_wgrad_scratch)
static long long g_wgw_scratch_floats = 0;
static int g_wgw_target = 512;
static int g_wgw_dbg = 0;
static unsigned long long* g_wgw_probe = nullptr;

extern "C" {   // (internal to the library, except the two knobs: declared in conv_wgrad.hip / mpr_hip.h)

int mpr_conv_set_wgrad_scratch(void* buf, long long floats) {
  g_wgw_scratch = (float*)buf;
  g_wgw_scratch_floats = buf ? floats : 0;
  return 0;
}

int mpr_conv_set_wgrad_window(int on) {   // bits 8.. = timing-experiment flags (bit 8: skip the atomic epilogue)
  const int old = g_wgw_on;
  g_wgw_on = on & 255;
  g_wgw_dbg = on >> 8;
  return old;
}

int mpr_conv_debug_wgrad_probe(void* buf) {   // 8 x uint64 per workgroup of the next sliding-window wgrad launches
  g_wgw_probe = (unsigned long long*)buf;
  return 0;
}

bool mpr_wgw_eligible(long long Mpix, int H, int W, int C, int K, int R, int S, int sh, int sw, int ph, int pw,
                      long long min_pix) {
  return g_wgw_on && R == 3 && S == 3 && sh == 1 && sw == 1 && ph == 1 && pw == 1 && C % 64 == 0 && K % 64 == 0 &&
         W >= 2 && W <= 56 && H >= 2 && Mpix >= min_pix && (Mpix / (H * W)) * (long long)(H + 1) * (W + 1) < (1ll << 30);
}

// dw [K][3][3][C] += (zeroed by the caller unless accumulating) the weight gradient of x [B,H,W,C], dy [B,H,W,K]
// one-shot scratch of mpr_conv_set_wgrad_scratch: read and cleared by EVERY mpr_conv_wgrad call, whichever kernel it takes
void mpr_wgw_take_scratch(float** buf, long long* floats) {
  *buf = g_wgw_scratch;
  *floats = g_wgw_scratch_floats;
  g_wgw_scratch = nullptr;
  g_wgw_scratch_floats = 0;
}

int mpr_wgw_launch(const void* x, const void* dy, float* dw, int B, int H, int W, int C, int K, int target_wgs,
                   float* scratch, long long scratch_floats, hipStream_t st) {
  WgwParams p;
  p.x = (const bf16_t*)x; p.dy = (const bf16_t*)dy; p.dw = dw;
  p.H = H; p.W = W; p.C = C; p.K = K;
  p.Wp = W + 1; p.img = (H + 1) * (W + 1); p.Gtot = B * p.img; p.halo8 = (W + 2 + 7) / 8 * 8;
  p.Ng = 9 * C; p.ncb = C / 64;
  const int KH = K % 128 == 0 ? 2 : 1;
  const int PS = (KH == 1 && g_wgw_on != 2) ? 2 : 1;      // (mpr_conv_set_wgrad_window(2): round-1 forms, comparisons)
  // (two pixel blocks per chunk at K % 128 == 0: 170 -> 158 us on layer2's shape, 149 -> 155 on layer3's, 168 -> 170 on
  //  layer4's -- the barrier is not what the loop loses; kept as mpr_conv_set_wgrad_window(3) for experiments)
  const int PB = (KH == 2 && g_wgw_on == 3) ? 2 : 1;
  p.nkt = K / (64 * KH);
  p.total_chunks = ceil_div(p.Gtot, 64 * PS * PB);
  const int tiles = p.ncb * p.nkt;
  int nsplit = (target_wgs > 0 ? target_wgs : g_wgw_target) / tiles;
  if (nsplit > ceil_div(p.total_chunks, 4)) nsplit = ceil_div(p.total_chunks, 4);
  if (nsplit < 1) nsplit = 1;
  p.cps = ceil_div(p.total_chunks, nsplit);
  nsplit = ceil_div(p.total_chunks, p.cps);
  p.x_bytes = (unsigned)((size_t)B * H * W * C * 2);
  p.dy_bytes = (unsigned)((size_t)B * H * W * K * 2);
  p.div_img = make_fastdiv(p.img); p.div_wp = make_fastdiv(p.Wp);
  p.dbg = g_wgw_dbg;
  p.probe = g_wgw_probe;
  // partial slices instead of atomics when the caller lent enough scratch for THIS launch (one-shot)
  p.part = (scratch && (long long)nsplit * K * p.Ng <= scratch_floats) ? scratch : nullptr;
  const dim3 grid(nsplit * tiles);
  const size_t lds = (size_t)512 * 128 + 3 * (size_t)(64 * PS * PB) * 128 * KH;      // x ring + three dy stages
  if (PB == 2) {
    static bool attr_set = false;
    if (!attr_set) {
      hipFuncSetAttribute((const void*)conv_wgrad_win_kernel<2, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_set = true;
    }
    conv_wgrad_win_kernel<2, 1, 2><<<grid, 768, lds, st>>>(p);
  } else if (PS == 2) {
    static bool attr_set = false;
    if (!attr_set) {
      hipFuncSetAttribute((const void*)conv_wgrad_win_kernel<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_set = true;
    }
    conv_wgrad_win_kernel<1, 2><<<grid, 768, lds, st>>>(p);
  } else if (KH == 2) {
    static bool attr_set = false;
    if (!attr_set) {
      hipFuncSetAttribute((const void*)conv_wgrad_win_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_set = true;
    }
    conv_wgrad_win_kernel<2><<<grid, 768, lds, st>>>(p);
  } else {
    static bool attr_set = false;
    if (!attr_set) {
      hipFuncSetAttribute((const void*)conv_wgrad_win_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_set = true;
    }
    conv_wgrad_win_kernel<1><<<grid, 384, lds, st>>>(p);
  }
  MPR_LAUNCH_CHECK("conv_wgrad_win_kernel");
  if (p.part) {
    const int n4 = K * p.Ng / 4;
    wgw_reduce_kernel<<<ceil_div(n4, 256) < 1024 ? ceil_div(n4, 256) : 1024, 256, 0, st>>>((const float4*)p.part, nsplit,
                                                                                          (float4*)dw, n4);
    MPR_LAUNCH_CHECK("wgw_reduce_kernel");
  }
  return MPR_OK;
}

}  // extern "C"
