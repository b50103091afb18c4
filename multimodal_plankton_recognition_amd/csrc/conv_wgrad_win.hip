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
// live in a 512-row LDS ring that slides along the raster (64 new rows per 64-pixel chunk), the nine taps read it at
// shifted rows, and only dy streams beside it: 16-24 KB per 4.7-9.4 MFLOP chunk (295-393 FLOP/B).
//
// Workgroup: 3 (filter row r) x 2 (channel half) x KH (64-wide slab of output channels) waves; a wave accumulates
// 2 x 3 MFMA tiles: [64 output channels] x [3 taps (r, 0..2) x 32 channels].  Both operands are pixel-major in
// memory while the contraction runs over pixels, so fragments come from ds_read_b64_tr_b16 as in conv_wgrad.hip.
// The pixel axis is split over workgroups; partial tiles go to lent scratch + one reduction pass (or fp32 atomics).
//
// Round 3 (what bounded the loop, measured with the per-wave probe + SQ counters -- scripts/wgw_probe_resid.py,
// scripts/prof_wgw_pmc.sh): not the LDS (SQ_LDS_BANK_CONFLICT 0, LDS array 26 % busy) and not the matrix pipe, but the SIMD's
// VECTOR ISSUE -- ~50 VALU instructions of address arithmetic per k-step and wave beside six MFMAs, plus ~25 per DMA
// instruction for the raster decode -- and the way a workgroup-wide barrier per chunk interacts with oldest-first wave
// arbitration.  Now: fragment reads with immediate offsets off one base register per (fragment, chunk) (mirrored ring: no
// wrap inside a chunk), reads interleaved with the MFMAs under counted lgkmcnt waits and pipelined across chunk boundaries,
// raster decode from a per-geometry table, each wave's DMA in one early burst placed by wave age, s_setprio by k-step.
// 3890 -> 2750 cycles per 64-pixel chunk (2304 = the MFMAs alone); launches at 160 workgroups 173-214 -> 142-164 us.
// (Two six-wave workgroups per CU instead of one of twelve were tried first: the dispatcher never co-schedules them --
//  a 6-wave group needs two waves on two of the SIMDs, and the resource check takes ceil(6/4) x 168 VGPRs on EVERY SIMD.)
#include "common.h"

struct WgwParams {
  const bf16_t* x;     // [B,H,W,C]
  const bf16_t* dy;    // [B,H,W,K]
  float* dw;           // [K][9*C]
  float* part;         // partial slices [nsplit][K][9*C] (plain stores + reduction pass) or NULL (atomics into dw)
  int H, W, C, K;
  int Wp, img, Gtot, halo8, mir;   // W+1, (H+1)*(W+1), B*img, (W+2) rounded up to 8, mirrored ring rows (chunk + 2 halo8)
  int Ng, ncb, nkt, cps, total_chunks;
  unsigned x_bytes, dy_bytes;
  FastDiv div_img, div_wp;
  int dbg;   // timing experiments: bit 0 = skip the atomic epilogue
  unsigned long long* probe;   // timing experiments: 8 x uint64 shader-clock sums per workgroup (wave 0), or NULL
  const uint32_t* rtab;        // raster table (runtime.cpp: mpr_raster_table): [margin + G] = pixel + 1, 0 = pad
  unsigned rtab_bytes;
};
#define WGW_RASTER_MARGIN 256   // = MPR_RASTER_MARGIN of runtime.cpp

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

// Immediate-offset form of the transposing read: the k-step and the second half of a fragment are OFFSETS of one base
// register per (fragment, chunk).  Round 3: with every address formed by VALU instructions (one v_add per read + the ring
// wrap and swizzle of the x rows: ~50 per k-step and wave) the loop was bound by the SIMD's VECTOR ISSUE, not by the matrix
// pipe or the LDS -- three waves x 50 x 4 cycles = 600 cycles of address arithmetic per k-step beside 576 of MFMA.
template <int OFF>
__device__ __forceinline__ s16x4 wgw_read_tr_o(uint32_t lds_addr) {
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(OFF) : "memory");
  return v;
}
// counted wait: LDS reads return in order, so "at most N outstanding" = everything older than the N youngest has landed;
// the fragments that the following MFMA consumes ride through as in/out operands (ordering for the compiler)
template <int N>
__device__ __forceinline__ void wgw_wait_lgkm(s16x4& a, s16x4& b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wgw_wait_lgkm(s16x4& a, s16x4& b, s16x4& c, s16x4& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wgw_wait_lgkm(s16x4 (&a)[4][2], s16x4& b0, s16x4& b1) {
  asm volatile("s_waitcnt lgkmcnt(%10)"
               : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[2][0]), "+v"(a[2][1]), "+v"(a[3][0]),
                 "+v"(a[3][1]), "+v"(b0), "+v"(b1)
               : "n"(N)
               : "memory");
}
template <int V>
struct WgwInt { static constexpr int value = V; };

// PS = 2 (64 output channels only): a chunk is 128 pixels and a second set of six waves works on its upper 64 -- twelve
// waves per workgroup as at KH = 2, where six (1.5 per SIMD) left every wave's read -> wait -> MFMA chain exposed.  The two
// halves' accumulators are summed through LDS at the end.
//
// LDS: [x ring: RING rows of 128 B][mirror: the ring's first CHUNK + 2 halo rows once more][3 dy stages].  The mirror makes a
// chunk's window CONTIGUOUS (rows cbase .. cbase + CHUNK + 2 halo, no wrap), which is what lets the fragment reads use
// immediate offsets; its rows are written by a second DMA instruction with the same source offsets (3 of 8 groups).
//
// Main loop, per wave and 16-pixel k-step: six MFMAs (2 tiles of output channels x 3 taps) and the ten transposing reads of
// the NEXT k-step, two behind each MFMA, in the order their consumers run (A0 B0 A1 B1 B2) with counted lgkmcnt waits --
// the pipeline runs on across chunk boundaries (the first fragments of chunk ci + 1 are read during the last k-step of
// chunk ci: the barrier at the top of iteration ci has confirmed them).  The DMA instructions of chunk ci + 2 go out one per
// k-step behind its first MFMA (an LDS-DMA instruction holds the wave's issue for 60-180 cycles).
//
// M16 (round 3, the default): the products as v_mfma_f32_16x16x32_bf16 -- the chip holds a higher clock on that shape
// (scripts/micro/mfma_shape.hip: +13.7 % in an LDS-fed loop) and conv_win_kernel gained 3-10 % per launch from it.  A wave's
// tile is 4 sub-tiles of 16 output channels x 6 sub-tiles (tap s, 16 channels); a k-step is 32 pixels: 20 transposing reads
// for 24 MFMAs of 16 cycles, the same LDS traffic per MFMA cycle as before.  The contraction index of a lane group g = lane / 16 is
// mapped to pixels 4 g .. 4 g + 3 (first read) and 16 + 4 g .. (second read) of the step for BOTH operands, so that a half-wave
// reads eight consecutive rows x 32 B; the 32-B units of a 64-B segment are swapped by bit 2 of the row on the DMA's source side
// (rows r and r + 4 would otherwise meet in the same banks).  The four dy sub-tiles of a step stay in registers (double
// buffered: the next step's arrive during this one), the six x sub-tiles pass through a four-slot ring three MFMA groups ahead.
template <int KH, int PS = 1, int PRIO = 0, bool PROBE = false, bool M16 = false>
__global__ __launch_bounds__(384 * KH * PS) void conv_wgrad_win_kernel(const WgwParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(KH * PS <= 2, "twelve waves at most");
  constexpr int NW = 6 * KH * PS;
  constexpr int CHUNK = 64 * PS;               // pixels per chunk
  constexpr int KSTEPS = 4;                    // 16-pixel k-steps per wave and chunk
  constexpr int RING = 512;                    // x ring rows (128 B each: 64 channels): 3 chunks + both halos (<= 512), power of two
  constexpr int MIRMAX = CHUNK + 128;          // mirror rows allocated (p.mir = CHUNK + 2 halo8 of them are kept)
  constexpr int NST = 3;                       // dy stages: chunk ci + 2 is in flight while chunk ci is multiplied
  constexpr int XBYTES = (RING + MIRMAX) * 128;
  constexpr int DROW = 128 * KH;               // dy stage row bytes (64*KH output channels)
  constexpr int DSTAGE = CHUNK * DROW;
  constexpr int D_INSTR = DSTAGE / 1024;       // 8*KH*PS DMA instructions per dy chunk
  constexpr int D_IT = (D_INSTR + NW - 1) / NW;
  constexpr int XI_IT = (CHUNK / 8 + 16 + NW - 1) / NW;    // initial window: up to CHUNK + 2*64 rows
  constexpr int XC_IT = (CHUNK / 8 + NW - 1) / NW;         // per chunk: CHUNK new rows
  static_assert(XC_IT + D_IT <= KSTEPS, "one DMA piece per k-step");
  auto ring = [](int G) -> int { return G & (RING - 1); };
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [x ring + mirror][dy stage 0][dy stage 1][dy stage 2]
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
  // x rows [G8, G8+8) (G8 a multiple of 8) -> ring rows ring(G8) ...; lane -> (row lane/8, physical chunk lane%8),
  // 64-B segment swizzle on the source side: logical chunk = ((pc >> 2) ^ ((row >> 1) & 1)) << 2 | (pc & 3)
  // (bit 1 of the ring row is bit 1 of the row inside its group of eight: G8 and RING are multiples of 8)
  auto x_off = [&](int G8) -> uint32_t {
    const int G = G8 + (lane >> 3);
    const int pc = lane & 7;
    int lc = (((pc >> 2) ^ ((lane >> 4) & 1)) << 2) | (pc & 3);
    if (M16) lc ^= ((lane >> 5) & 1) << 1;              // 32-B unit ^ bit 2 of the row
    uint32_t pix;
    return pixel_of(G, pix) ? pix * (uint32_t)(2 * p.C) + (uint32_t)((cb * 64 + lc * 8) * 2) : 0xFFFFFFF0u;
  };
  auto fire_x8 = [&](int G8, uint32_t v) {
    const int rr = ring(G8);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(smem + rr * 128), 16, v, 0, 0, 0);
    if (rr < p.mir)      // (wave-uniform) the same rows once more behind the ring
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(smem + (rr + RING) * 128), 16, v,
                                               0, 0, 0);
  };
  auto issue_x8 = [&](int G8) { fire_x8(G8, x_off(G8)); };
  // dy rows of chunk ci -> stage ci % NST: instruction I covers 1024 / DROW rows
  auto dy_off = [&](int ci, int I) -> uint32_t {
    constexpr int CH = DROW / 16;                       // 16-B chunks per row (8 or 16)
    const int q = I * 64 + lane;
    const int row = q / CH, pc = q % CH;
    const int key = CH == 16 ? (row & 3) : ((row >> 1) & 1);
    int lc = (((pc >> 2) ^ key) << 2) | (pc & 3);
    if (M16) lc ^= ((row >> 2) & 1) << 1;
    uint32_t pix;
    return pixel_of(ci * CHUNK + row, pix) ? pix * (uint32_t)(2 * p.K) + (uint32_t)((kt * 64 * KH + lc * 8) * 2)
                                        : 0xFFFFFFF0u;
  };
  // Steady state.  The SIMD serves its oldest wave first, so between two barriers the first wave of a SIMD (wid < 4) runs
  // ahead and then idles at the barrier while the youngest of the three finishes last, alone (round-3 per-wave probe: barrier
  // waits 1300 / 450 / 70 cycles of 3200 per chunk).  An LDS-DMA instruction holds its wave's issue for 60-180 cycles, and what
  // the LAST wave issues late in the chunk is fully exposed.  So every wave issues its whole share of chunk ci + 2's DMA in ONE
  // burst early in iteration ci: the two younger waves of a SIMD behind their first MFMA of k-step 0 -- the oldest holds the
  // matrix pipe then anyway --, the oldest behind its first MFMA of k-step 1, when the other two have theirs out and fill the
  // pipe.  One wave-uniform branch per k-step 0 / 1 and wave.
  // The raster position -> pixel map comes from the table (one dword per lane and piece, loaded a whole chunk before its use
  // with a per-lane-constant voffset and a scalar soffset: no address VALU) instead of the two divisions of pixel_of:
  // ~25 VALU instructions per piece, 400 of 3250 cycles per chunk.  t[J] holds piece J's table value, converted in place to
  // the source offset at the top of the iteration that fires it, and is reloaded for the next chunk right behind the DMA.
  constexpr int NP = XC_IT + D_IT;
  // (K = 64 on twelve waves, where every byte of x and dy streams in from HBM once: every wave in k-step 0 -- 159 -> 139 us
  //  per layer1 launch; dbg bit 5 flips the choice)
  const int dma_ks = (wid < 4 && ((PS == 1) != ((p.dbg & 32) != 0))) ? 1 : 0;
  const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc((void*)p.rtab, 0, p.rtab_bytes, 0x00020000);
  constexpr int DCH = DROW / 16;                               // 16-B chunks per dy row (8 or 16); 64 / DCH rows per instruction
  const int lane_x4 = (lane >> 3) * 4, lane_d4 = (lane / DCH) * 4;
  uint32_t xcol, dcol;                                         // column terms of the source offsets (- one row: the table holds pixel + 1)
  {
    const int pc = lane & 7;
    int lc = (((pc >> 2) ^ ((lane >> 4) & 1)) << 2) | (pc & 3);
    if (M16) lc ^= ((lane >> 5) & 1) << 1;
    xcol = (uint32_t)((cb * 64 + lc * 8) * 2 - 2 * p.C);
    const int pd = lane % DCH, rin = lane / DCH;
    const int key = DCH == 16 ? (rin & 3) : ((rin >> 1) & 1);
    int ld = (((pd >> 2) ^ key) << 2) | (pd & 3);
    // (bit 2 of the dy row: 8 rows per instruction at 128-B rows; 4 at 256-B rows, where it is the parity of the instruction
    //  index wid + j * NW -- NW is even)
    static_assert(DCH == 8 || NW % 2 == 0, "the row bit of a dy piece must not depend on the piece");
    if (M16) ld ^= (DCH == 8 ? (rin >> 2) & 1 : wid & 1) << 1;
    dcol = (uint32_t)((kt * 64 * KH + ld * 8) * 2 - 2 * p.K);
  }
  // (raw table values t[] and converted offsets tc[] in SEPARATE registers: with the conversion in place, a wave that fires its
  //  burst in the later slot found "t[J] may have a load pending" on the path through the earlier slot's reload -- the two
  //  slots exclude each other, which the compiler's wait insertion cannot know -- and got s_waitcnt vmcnt(0) in front of every
  //  DMA instruction of the burst: each waited for the previous one to LAND)
  uint32_t t[NP], tc[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) { t[j] = 0; tc[j] = 0; }
  // piece J of chunk ci (J < XC_IT: a group of eight x rows; else a kilobyte of dy rows): fire its DMA from the offsets in t[J]
  auto fire_piece = [&](auto J_, int ci) {
    constexpr int J = decltype(J_)::value;
    if constexpr (J < XC_IT) {
      const int I = wid + J * NW;
      if ((CHUNK / 8) % NW == 0 || I < CHUNK / 8) fire_x8(ci * CHUNK + p.halo8 + 8 * I, tc[J]);
    } else if constexpr (J < NP) {
      const int I = wid + (J - XC_IT) * NW;
      if (D_INSTR % NW == 0 || I < D_INSTR)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rs_d, (__attribute__((address_space(3))) void*)(dyst + (ci % NST) * DSTAGE + I * 1024), 16, tc[J], 0, 0, 0);
    }
  };
  // ... and load its table value for chunk ci
  auto load_piece = [&](auto J_, int ci) {
    constexpr int J = decltype(J_)::value;
    if constexpr (J < XC_IT) {
      const int I = wid + J * NW;
      if ((CHUNK / 8) % NW == 0 || I < CHUNK / 8)
        t[J] = __builtin_amdgcn_raw_buffer_load_b32(rs_t, lane_x4, (ci * CHUNK + p.halo8 + 8 * I + WGW_RASTER_MARGIN) * 4, 0);
    } else if constexpr (J < NP) {
      const int I = wid + (J - XC_IT) * NW;
      if (D_INSTR % NW == 0 || I < D_INSTR)
        t[J] = __builtin_amdgcn_raw_buffer_load_b32(rs_t, lane_d4, (ci * CHUNK + I * (64 / DCH) + WGW_RASTER_MARGIN) * 4, 0);
    }
  };
  auto convert_pieces = [&]() {
#pragma unroll
    for (int j = 0; j < NP; ++j)
      tc[j] = t[j] ? __umul24(t[j], (uint32_t)(2 * (j < XC_IT ? p.C : p.K))) + (j < XC_IT ? xcol : dcol) : 0xFFFFFFF0u;
  };

  f32x16 acc[M16 ? 1 : 2][M16 ? 1 : 3];     // [m-tile of 32 output channels][tap s]
  f32x4 acc16[M16 ? 4 : 1][M16 ? 6 : 1];    // [sub-tile of 16 output channels][tap s, channel sub-tile]
#pragma unroll
  for (int i = 0; i < (M16 ? 1 : 2); ++i)
#pragma unroll
    for (int s = 0; s < (M16 ? 1 : 3); ++s)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][s][e] = 0.f;
#pragma unroll
  for (int i = 0; i < (M16 ? 4 : 1); ++i)
#pragma unroll
    for (int s = 0; s < (M16 ? 6 : 1); ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc16[i][s][e] = 0.f;

  // transposing-read lane geometry (conv_wgrad.hip): a 16-lane group reads a 4 (pixel) x 16 (column) block
  const int g16 = lane >> 4, lq = (lane & 15) >> 2, lp = lane & 3;
  // 32x32x16: pixel row inside a 16-deep k-step (+4: second read); 16x16x32: inside a 32-deep step (+16: second read)
  const int rowl = M16 ? 4 * g16 + lq : 8 * (g16 >> 1) + lq;
  const int inseg = M16 ? (4 * lp) * 2 : (16 * (g16 & 1) + 4 * lp) * 2;      // byte inside a 64-B segment (16x16x32: a 32-B unit)
  // dy fragment offsets inside a stage (pixel rows are chunk-local: key is lane-constant)
  uint32_t a_rd[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int seg = wk * 2 + i;                                // 64-B segment = 32 output channels
    const int key = KH == 2 ? (lq & 3) : ((lq >> 1) & 1);
    a_rd[i] = lds0 + XBYTES + (wp * 64 + rowl) * DROW + ((seg ^ key) << 6) + inseg;
  }
  // x fragment rows relative to the chunk's lowest window row (raster position ci * CHUNK - halo8), per tap column s
  int rowoff[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) rowoff[s] = wp * 64 + rowl + (wr - 1) * p.Wp + (s - 1) + p.halo8;
  // fragment base addresses of a chunk: k-step and fragment half are immediate offsets of these
  uint32_t abase[2], bbase[3];
  auto set_bases = [&](int ci) {
    const int cbase = ring(ci * CHUNK - p.halo8);              // (scalar) ring row of the window's first row; no wrap behind it
    const uint32_t ds = (uint32_t)((ci % NST) * DSTAGE);
#pragma unroll
    for (int i = 0; i < 2; ++i) abase[i] = a_rd[i] + ds;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int r = cbase + rowoff[s];
      bbase[s] = lds0 + r * 128 + ((wh ^ ((r >> 1) & 1)) << 6) + inseg;      // (r + 16 ks + 4) >> 1 has the parity of r >> 1
    }
  };

  // 16x16x32: per-lane fragment offsets, relative to the dy stage / to the window's first ring row (a multiple of 8 rows, so the
  // swizzle bits of a row are those of its offset); the chunk's part is a scalar added at the read
  uint32_t aoff[M16 ? 4 : 1], boff[M16 ? 6 : 1];
  if constexpr (M16) {
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
      const int seg = wk * 2 + (i4 >> 1), sub = i4 & 1;
      const int key = KH == 2 ? (lq & 3) : ((lq >> 1) & 1);
      aoff[i4] = lds0 + XBYTES + (wp * 64 + rowl) * DROW + ((seg ^ key) << 6) + ((sub ^ (g16 & 1)) << 5) + inseg;
    }
#pragma unroll
    for (int n = 0; n < 6; ++n) {
      const int r = rowoff[n >> 1], sub = n & 1;
      boff[n] = lds0 + r * 128 + ((wh ^ ((r >> 1) & 1)) << 6) + ((sub ^ ((r >> 2) & 1)) << 5) + inseg;
    }
  }

  // ---- prologue: window of the first chunk + its dy, all of chunk c0 + 1 (every wave takes part, offsets by division),
  //      and the DMA waves' table values of chunk c0 + 2 (fired in the first iteration)
  {
    const int lo = c0 * CHUNK - p.halo8, hi = c0 * CHUNK + CHUNK + p.halo8;
#pragma unroll
    for (int j = 0; j < XI_IT; ++j) {
      const int G8 = lo + 8 * (wid + j * NW);
      if (G8 < hi) issue_x8(G8);
    }
    auto issue_dy = [&](int ci) {
#pragma unroll
      for (int j = 0; j < D_IT; ++j) {
        const int I = wid + j * NW;
        if (D_INSTR % NW == 0 || I < D_INSTR)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(
              rs_d, (__attribute__((address_space(3))) void*)(dyst + (ci % NST) * DSTAGE + I * 1024), 16, dy_off(ci, I), 0, 0, 0);
      }
    };
    issue_dy(c0);
    if (c0 + 1 < c1) {
#pragma unroll
      for (int j = 0; j < XC_IT; ++j) {
        const int I = wid + j * NW;
        if ((CHUNK / 8) % NW == 0 || I < CHUNK / 8) issue_x8((c0 + 1) * CHUNK + p.halo8 + 8 * I);
      }
      issue_dy(c0 + 1);
    }
    if (c0 + 2 < c1) {      // (table values: converted at the top of the first iteration)
      load_piece(WgwInt<0>{}, c0 + 2); load_piece(WgwInt<1>{}, c0 + 2); load_piece(WgwInt<2>{}, c0 + 2); load_piece(WgwInt<3>{}, c0 + 2);
    }
    wgw_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  unsigned long long pr_wait = 0, pr_bar = 0, pr_comp = 0, pr_iss = 0;
#define WGW_NOW() (PROBE ? __builtin_readcyclecounter() : 0ull)
  const unsigned long long pr_t0 = WGW_NOW();
  const unsigned long long pr_w0 = PROBE ? (wall_clock64() & 0xFFFFFFFFull) : 0ull;

  typedef __attribute__((ext_vector_type(8))) short s16x8;
#define WGW_FRAG(v) __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector((v)[0], (v)[1], 0, 1, 2, 3, 4, 5, 6, 7))
  if constexpr (M16) {
    s16x4 A16[2][4][2], B16[4][2];      // dy sub-tiles [buffer = step parity][sub-tile][half]; x sub-tile ring [slot][half]
    // (aoff / boff hold the CURRENT chunk's fragment addresses; they move to the next chunk by a scalar difference -- ten VALU
    //  instructions per chunk: at 16 cycles per MFMA the matrix instructions alone take half of the SIMD's issue slots, and an
    //  address add per read pair showed as +10 % cycles per chunk)
#define WGW_RA16(buf, i4, KN) do { A16[buf][i4][0] = wgw_read_tr_o<(KN) * 32 * DROW>(aoff[i4]); A16[buf][i4][1] = wgw_read_tr_o<(KN) * 32 * DROW + 16 * DROW>(aoff[i4]); } while (0)
#define WGW_RB16(slot, n, KN) do { B16[slot][0] = wgw_read_tr_o<(KN) * 4096>(boff[n]); B16[slot][1] = wgw_read_tr_o<(KN) * 4096 + 2048>(boff[n]); } while (0)
#define WGW_MF16(n, i4) acc16[i4][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WGW_FRAG(A16[cur][i4]), WGW_FRAG(B16[WGW_SL(n)]), acc16[i4][n], 0, 0, 0)
#define WGW_MF16_REST(n) do { WGW_MF16(n, 1); WGW_MF16(n, 2); WGW_MF16(n, 3); } while (0)
    // one 32-pixel step KS of a chunk: six groups of four MFMAs (x sub-tile n against the four dy sub-tiles).  Behind the first
    // MFMA of group n go out the reads of the NEXT step's dy sub-tile n (n < 4) and of the x sub-tile three groups ahead (ring of
    // four slots; two groups ahead left the waves waiting on the LDS three times as long as the 32x32x16 loop:
    // SQ_WAIT_INST_LDS 16.5 M against 5.2 M wave-cycles per launch).  Reads return in order: counted waits.
    // In the chunk's second step every dy read and the reads of x sub-tiles 0..2 belong to the NEXT chunk: `d_dy` / `d_x` (scalar
    // address differences) move aoff[0..3], boff[0..2] at its top and boff[3..5] behind their last read.
    auto kstep16 = [&](auto KS_, auto&& slot0, auto&& slot3, int d_dy, int d_x) {
      constexpr int KS = decltype(KS_)::value;
      constexpr int cur = KS, nxt = KS ^ 1, KN = KS ^ 1;
#define WGW_SL(n) ((6 * KS + (n)) & 3)
#define WGW_SLN(n) ((6 * KS + (n) + 3) & 3)
      if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(3 - 2 * KS);
      if constexpr (KS == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) aoff[i] += d_dy;
        boff[0] += d_x; boff[1] += d_x; boff[2] += d_x;
      }
      wgw_wait_lgkm<4>(A16[cur], B16[WGW_SL(0)][0], B16[WGW_SL(0)][1]);          // younger: x 1, x 2
      __builtin_amdgcn_sched_barrier(0);
      WGW_MF16(0, 0);
      __builtin_amdgcn_sched_barrier(0);
      WGW_RA16(nxt, 0, KN);
      WGW_RB16(WGW_SLN(0), 3, KS);
      slot0();
      __builtin_amdgcn_sched_barrier(0);
      WGW_MF16_REST(0);
      __builtin_amdgcn_sched_barrier(0);
      wgw_wait_lgkm<6>(B16[WGW_SL(1)][0], B16[WGW_SL(1)][1]);                    // younger: x 2, dy' 0, x 3
      __builtin_amdgcn_sched_barrier(0);
      WGW_MF16(1, 0);
      __builtin_amdgcn_sched_barrier(0);
      WGW_RA16(nxt, 1, KN);
      WGW_RB16(WGW_SLN(1), 4, KS);
      __builtin_amdgcn_sched_barrier(0);
      WGW_MF16_REST(1);
      __builtin_amdgcn_sched_barrier(0);
      wgw_wait_lgkm<8>(B16[WGW_SL(2)][0], B16[WGW_SL(2)][1]);                    // younger: dy' 0, x 3, dy' 1, x 4
      __builtin_amdgcn_sched_barrier(0);
      WGW_MF16(2, 0);
      __builtin_amdgcn_sched_barrier(0);
      WGW_RA16(nxt, 2, KN);
      WGW_RB16(WGW_SLN(2), 5, KS);
      if constexpr (KS == 1) { boff[3] += d_x; boff[4] += d_x; boff[5] += d_x; }
      __builtin_amdgcn_sched_barrier(0);
      WGW_MF16_REST(2);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(2 - 2 * KS);
      wgw_wait_lgkm<8>(B16[WGW_SL(3)][0], B16[WGW_SL(3)][1]);                    // younger: dy' 1, x 4, dy' 2, x 5
      __builtin_amdgcn_sched_barrier(0);
      WGW_MF16(3, 0);
      __builtin_amdgcn_sched_barrier(0);
      WGW_RA16(nxt, 3, KN);
      WGW_RB16(WGW_SLN(3), 0, KN);
      slot3();
      __builtin_amdgcn_sched_barrier(0);
      WGW_MF16_REST(3);
      __builtin_amdgcn_sched_barrier(0);
      wgw_wait_lgkm<8>(B16[WGW_SL(4)][0], B16[WGW_SL(4)][1]);                    // younger: dy' 2, x 5, dy' 3, x' 0
      __builtin_amdgcn_sched_barrier(0);
      WGW_MF16(4, 0);
      __builtin_amdgcn_sched_barrier(0);
      WGW_RB16(WGW_SLN(4), 1, KN);
      __builtin_amdgcn_sched_barrier(0);
      WGW_MF16_REST(4);
      __builtin_amdgcn_sched_barrier(0);
      wgw_wait_lgkm<6>(B16[WGW_SL(5)][0], B16[WGW_SL(5)][1]);                    // younger: dy' 3, x' 0, x' 1
      __builtin_amdgcn_sched_barrier(0);
      WGW_MF16(5, 0);
      __builtin_amdgcn_sched_barrier(0);
      WGW_RB16(WGW_SLN(5), 2, KN);
      __builtin_amdgcn_sched_barrier(0);
      WGW_MF16_REST(5);
      __builtin_amdgcn_sched_barrier(0);
#undef WGW_SLN
#undef WGW_SL
    };
    auto ds_of = [&](int ci) -> uint32_t { return (uint32_t)((ci % NST) * DSTAGE); };
    auto cb_of = [&](int ci) -> uint32_t { return (uint32_t)(ring(ci * CHUNK - p.halo8) * 128); };
    {
      const uint32_t d0 = ds_of(c0), w0 = cb_of(c0);
#pragma unroll
      for (int i = 0; i < 4; ++i) aoff[i] += d0;
#pragma unroll
      for (int n = 0; n < 6; ++n) boff[n] += w0;
      WGW_RA16(0, 0, 0); WGW_RA16(0, 1, 0); WGW_RA16(0, 2, 0); WGW_RA16(0, 3, 0);
      WGW_RB16(0, 0, 0); WGW_RB16(1, 1, 0); WGW_RB16(2, 2, 0);
    }
    // (loop invariant, DMA and table handling: as in the 32x32x16 loop below)
    for (int ci = c0; ci < c1; ++ci) {
      const unsigned long long q0 = WGW_NOW();
      wgw_wait_vmcnt<0>();
      const unsigned long long q1 = WGW_NOW();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const unsigned long long q2 = WGW_NOW();
      pr_wait += q1 - q0;
      pr_bar += q2 - q1;
      const bool more2 = ci + 2 < c1 && !(p.dbg & 8);
      const bool more3 = ci + 3 < c1 && !(p.dbg & 4);
      auto dma_burst = [&]() {
        if (more2) { fire_piece(WgwInt<0>{}, ci + 2); fire_piece(WgwInt<1>{}, ci + 2); fire_piece(WgwInt<2>{}, ci + 2); fire_piece(WgwInt<3>{}, ci + 2); }
        if (more3) { load_piece(WgwInt<0>{}, ci + 3); load_piece(WgwInt<1>{}, ci + 3); load_piece(WgwInt<2>{}, ci + 3); load_piece(WgwInt<3>{}, ci + 3); }
      };
      const int d_dy = (int)ds_of(ci + 1) - (int)ds_of(ci), d_x = (int)cb_of(ci + 1) - (int)cb_of(ci);
      kstep16(WgwInt<0>{},
              [&]() {
                convert_pieces();
#pragma unroll
                for (int j = 0; j < NP; ++j) asm volatile("" : "+v"(tc[j]));
                if (dma_ks == 0) dma_burst();
              },
              [&]() { if (dma_ks == 1) dma_burst(); }, 0, 0);
      // (beyond the last chunk the reads of the second step are dummies that keep the wait counts uniform: any address inside
      //  the allocation is fine)
      kstep16(WgwInt<1>{}, [&]() {}, [&]() {}, d_dy, d_x);
      pr_comp += WGW_NOW() - q2;
    }
#undef WGW_MF16_REST
#undef WGW_MF16
#undef WGW_RB16
#undef WGW_RA16
  } else {
  s16x4 ra[2][2][2], rb[2][3][2];      // [buffer][tile][half]: the fragments of k-step ks live in buffer ks & 1
#define WGW_RA(buf, i, KN) do { ra[buf][i][0] = wgw_read_tr_o<(KN) * 16 * DROW>(abase[i]); ra[buf][i][1] = wgw_read_tr_o<(KN) * 16 * DROW + 4 * DROW>(abase[i]); } while (0)
#define WGW_RB(buf, s, KN) do { rb[buf][s][0] = wgw_read_tr_o<(KN) * 2048>(bbase[s]); rb[buf][s][1] = wgw_read_tr_o<(KN) * 2048 + 512>(bbase[s]); } while (0)
#define WGW_MFMA(i, s) acc[i][s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WGW_FRAG(ra[cur][i]), WGW_FRAG(rb[cur][s]), acc[i][s], 0, 0, 0)
  // one k-step: the MFMAs of step KS (fragments in buffer KS & 1) with the reads of the next step behind them; `slot_a` runs
  // behind the first MFMA (the wave's share of the next-but-one chunk's DMA, in k-step 0 or 1).  When KS is the chunk's last step the
  // bases have been switched to the next chunk already.  Outstanding reads at the top: the ten of this step, in the order
  // A0 B0 A1 B1 B2 (two each).
  auto kstep = [&](auto KS_, auto&& slot_a) {
    constexpr int KS = decltype(KS_)::value;
    constexpr int cur = KS & 1, nxt = cur ^ 1, KN = (KS + 1) % KSTEPS;
    // a wave that is behind outranks one that is ahead -- the SIMD otherwise serves its oldest wave first, which then idles at
    // the barrier while the youngest finishes alone
    if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(3 - KS);
    wgw_wait_lgkm<6>(ra[cur][0][0], ra[cur][0][1], rb[cur][0][0], rb[cur][0][1]);
    __builtin_amdgcn_sched_barrier(0);
    WGW_MFMA(0, 0);
    WGW_RA(nxt, 0, KN);
    slot_a();
    __builtin_amdgcn_sched_barrier(0);
    wgw_wait_lgkm<6>(ra[cur][1][0], ra[cur][1][1]);        // younger: B1 B2 A0'
    __builtin_amdgcn_sched_barrier(0);
    WGW_MFMA(1, 0);
    WGW_RB(nxt, 0, KN);
    __builtin_amdgcn_sched_barrier(0);
    wgw_wait_lgkm<6>(rb[cur][1][0], rb[cur][1][1]);        // younger: B2 A0' B0'
    __builtin_amdgcn_sched_barrier(0);
    WGW_MFMA(0, 1);
    WGW_RA(nxt, 1, KN);
    __builtin_amdgcn_sched_barrier(0);
    WGW_MFMA(1, 1);
    WGW_RB(nxt, 1, KN);
    __builtin_amdgcn_sched_barrier(0);
    wgw_wait_lgkm<8>(rb[cur][2][0], rb[cur][2][1]);        // younger: A0' B0' A1' B1'
    __builtin_amdgcn_sched_barrier(0);
    WGW_MFMA(0, 2);
    WGW_RB(nxt, 2, KN);
    __builtin_amdgcn_sched_barrier(0);
    WGW_MFMA(1, 2);
    __builtin_amdgcn_sched_barrier(0);
  };
  // first fragments of the first chunk
  set_bases(c0);
  WGW_RA(0, 0, 0); WGW_RB(0, 0, 0); WGW_RA(0, 1, 0); WGW_RB(0, 1, 0); WGW_RB(0, 2, 0);
  // Loop invariant at the top of iteration ci: chunk ci's operands are in LDS for EVERY wave (confirmed by the previous
  // barrier), chunk ci + 1's are in flight (issued one iteration ago); the first fragment reads of chunk ci are out.  The
  // wait + barrier confirm chunk ci + 1 and release the stage / ring rows that chunk ci + 2 overwrites (last read in
  // iteration ci - 1: a wave that has passed its last k-step's waits has every one of that chunk's reads back).
  // (The table values are converted UNCONDITIONALLY at the top, so that the compiler knows every t[J] a DMA instruction takes
  //  its offsets from is ready; with the conversion under a condition it guarded EACH DMA instruction with s_waitcnt vmcnt(0),
  //  i.e. waited for the previous one to land.)
  for (int ci = c0; ci < c1; ++ci) {
    const unsigned long long q0 = WGW_NOW();
    wgw_wait_vmcnt<0>();
    const unsigned long long q1 = WGW_NOW();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const unsigned long long q2 = WGW_NOW();
    pr_wait += q1 - q0;
    pr_bar += q2 - q1;
    // chunk ci + 2's source offsets from the table values loaded one iteration ago (the wait above covers them; chunk
    // c0 + 2's were loaded in the prologue).  Slot J of the k-steps: fire piece J of chunk ci + 2, reload t[J] for chunk ci + 3.
    const bool more2 = ci + 2 < c1 && !(p.dbg & 8);      // (dbg bit 3: timing experiment, no DMA issue in the loop)
    const bool more3 = ci + 3 < c1 && !(p.dbg & 4);      // (dbg bit 2: no table loads)
    auto dma_burst = [&]() {
      if (more2) { fire_piece(WgwInt<0>{}, ci + 2); fire_piece(WgwInt<1>{}, ci + 2); fire_piece(WgwInt<2>{}, ci + 2); fire_piece(WgwInt<3>{}, ci + 2); }
      if (more3) { load_piece(WgwInt<0>{}, ci + 3); load_piece(WgwInt<1>{}, ci + 3); load_piece(WgwInt<2>{}, ci + 3); load_piece(WgwInt<3>{}, ci + 3); }
    };
    kstep(WgwInt<0>{}, [&]() {
      // (behind the first MFMA; pinned by an empty asm that "uses" every offset: the optimizer otherwise sinks each conversion
      //  into the conditional block of its DMA instruction, behind a counted vmcnt wait that by then also covers the DMA
      //  instructions just issued)
      convert_pieces();
#pragma unroll
      for (int j = 0; j < NP; ++j) asm volatile("" : "+v"(tc[j]));
      if (dma_ks == 0) dma_burst();
    });
    kstep(WgwInt<1>{}, [&]() { if (dma_ks == 1) dma_burst(); });
    kstep(WgwInt<2>{}, [&]() {});
    // the next chunk's bases (its first fragments are read during this chunk's last k-step; beyond the last chunk the
    // reads are dummies that keep the wait counts uniform: any address inside the allocation is fine)
    set_bases(ci + 1);
    __builtin_amdgcn_sched_barrier(0);
    kstep(WgwInt<3>{}, [&]() {});
    pr_comp += WGW_NOW() - q2;
  }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the dummy reads of the last k-step
#undef WGW_MFMA
#undef WGW_RB
#undef WGW_RA
#undef WGW_FRAG
  if (PROBE && p.probe && (p.dbg & 2) && lane == 0) {      // per-wave probe: 16 x 8 uint64 per workgroup (mpr_conv_set_wgrad_window(1 | 2 << 8))
    unsigned long long* o = p.probe + ((size_t)blockIdx.x * 16 + wid) * 8;
    o[0] = WGW_NOW() - pr_t0; o[1] = pr_wait; o[2] = pr_bar; o[3] = pr_iss; o[4] = pr_comp; o[5] = (unsigned long long)(c1 - c0);
    // where and when: HW_ID (wave slot 3:0, SIMD 5:4, CU 11:8, SH 12, SE 15:13) | XCC_ID << 32; start / end on the 100 MHz clock
    o[6] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
    o[7] = pr_w0 | (wall_clock64() << 32);
  } else
  if (PROBE && p.probe && tid == 0) {
    unsigned long long* o = p.probe + (size_t)blockIdx.x * 8;
    o[0] = WGW_NOW() - pr_t0; o[1] = pr_wait; o[2] = pr_bar; o[3] = pr_iss; o[4] = pr_comp; o[5] = (unsigned long long)(c1 - c0);
  }
#undef WGW_NOW

  if (p.dbg & 1) {      // timing experiment: no epilogue (one conditional store keeps the accumulators alive)
    if constexpr (M16) { if (acc16[0][0][0] + acc16[3][5][2] == 12345.f) p.dw[0] = 1.f; }
    else { if (acc[0][0][0] + acc[1][2][5] == 12345.f) p.dw[0] = 1.f; }
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
        if constexpr (M16) {
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int n = 0; n < 6; ++n)
#pragma unroll
              for (int e = 0; e < 4; ++e) red[(role * 48 + (h * 6 + n) * 4 + e) * 64 + lane] = acc16[2 * i + h][n][e];
        } else {
#pragma unroll
        for (int s2 = 0; s2 < 3; ++s2)
#pragma unroll
          for (int e = 0; e < 16; ++e) red[((role * 3 + s2) * 16 + e) * 64 + lane] = acc[i][s2][e];
        }
      }
      __syncthreads();
      if (wp == 0) {
        if constexpr (M16) {
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int n = 0; n < 6; ++n)
#pragma unroll
              for (int e = 0; e < 4; ++e) acc16[2 * i + h][n][e] += red[(role * 48 + (h * 6 + n) * 4 + e) * 64 + lane];
        } else {
#pragma unroll
        for (int s2 = 0; s2 < 3; ++s2)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][s2][e] += red[((role * 3 + s2) * 16 + e) * 64 + lane];
        }
      }
    }
    if (wp == 1) return;
  }
  // D[k (regs)][n (lanes)], 128 B contiguous per half-wave: plain stores into this pixel split's slice (summed by
  // wgw_reduce_kernel) when the caller lent scratch memory -- the chip adds ~1.3 TB/s of fp32 atomics but stores
  // ~6 TB/s, and every launch emits 256 CUs x 295 KB of accumulators -- else fp32 atomics into [K][(r,s)][C]
  const int ln = lane & 31, lh = lane >> 5;
  float* const slice = p.part ? p.part + (size_t)split * p.K * p.Ng : nullptr;
  if constexpr (M16) {
    // 16 x 16 sub-tiles: lane -> column lane % 16, rows 4 (lane / 16) + e
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
      for (int n6 = 0; n6 < 6; ++n6) {
        const int n = (wr * 3 + (n6 >> 1)) * p.C + cb * 64 + wh * 32 + (n6 & 1) * 16 + (lane & 15);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = kt * 64 * KH + wk * 64 + i4 * 16 + 4 * (lane >> 4) + e;
          if (slice) slice[(size_t)k * p.Ng + n] = acc16[i4][n6][e];
          else atomicAdd(p.dw + (size_t)k * p.Ng + n, acc16[i4][n6][e]);
        }
      }
    return;
  }
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

// dw[i] += sum over the pixel splits of part[s][i].  A workgroup owns 256 / SL float4 columns and walks the splits on SL
// lanes per column (layer1: 9216 columns x 160-256 splits -- one thread per column left 36 workgroups chasing a serial chain of
// loads, 38 us per launch), partial sums meet in LDS in a fixed order (run-to-run identical).  The slices are L2 /
// Infinity-Cache warm.
template <int SL>
__global__ __launch_bounds__(256) void wgw_reduce_kernel(const float4* __restrict__ part, int nsplit, float* __restrict__ dw,
                                                          int n4) {
  constexpr int COLS = 256 / SL;
  __shared__ float4 red[SL][COLS];
  const int tid = threadIdx.x, col = tid % COLS, sl = tid / COLS;
  const int i = blockIdx.x * COLS + col;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
    int s = sl;
    for (; s + 3 * SL < nsplit; s += 4 * SL) {
      const float4 v0 = part[(size_t)s * n4 + i], v1 = part[(size_t)(s + SL) * n4 + i];
      const float4 v2 = part[(size_t)(s + 2 * SL) * n4 + i], v3 = part[(size_t)(s + 3 * SL) * n4 + i];
      a.x += (v0.x + v1.x) + (v2.x + v3.x); a.y += (v0.y + v1.y) + (v2.y + v3.y);
      a.z += (v0.z + v1.z) + (v2.z + v3.z); a.w += (v0.w + v1.w) + (v2.w + v3.w);
    }
    for (; s < nsplit; s += SL) {
      const float4 v = part[(size_t)s * n4 + i];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  red[sl][col] = a;
  __syncthreads();
  // thread t < 4 COLS: scalar t of the workgroup's 4 COLS contiguous floats
  for (int t = tid; t < 4 * COLS; t += 256) {
    const size_t o = (size_t)blockIdx.x * (4 * COLS) + t;
    if (o < (size_t)n4 * 4) {
      const float* const r = reinterpret_cast<const float*>(&red[0][0]) + t;
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < SL; ++q) v += r[q * 4 * COLS];
      dw[o] += v;
    }
  }
}

extern "C" const uint32_t* mpr_raster_table(int B, int H, int W, long long* entries);   // runtime.cpp
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
         W >= 2 && W <= 56 && H >= 2 && Mpix >= min_pix && Mpix < (1ll << 24) /* 24-bit multiply of the table path */ &&
         (Mpix / (H * W)) * (long long)(H + 1) * (W + 1) < (1ll << 30);
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
  const int PS = (KH == 1 && g_wgw_on != 2) ? 2 : 1;      // (mpr_conv_set_wgrad_window(2): the six-wave form at K = 64, comparisons)
  p.nkt = K / (64 * KH);
  p.total_chunks = ceil_div(p.Gtot, 64 * PS);
  p.mir = 64 * PS + 2 * p.halo8;
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
  long long rt_entries = 0;
  p.rtab = mpr_raster_table(B, H, W, &rt_entries);
  MPR_REQUIRE(p.rtab != nullptr, "conv_wgrad (window): raster table allocation failed");
  p.rtab_bytes = (unsigned)(rt_entries * 4);
  // partial slices instead of atomics when the caller lent enough scratch for THIS launch (one-shot)
  p.part = (scratch && (long long)nsplit * K * p.Ng <= scratch_floats) ? scratch : nullptr;
  const dim3 grid(nsplit * tiles);
  // x ring + its mirror + three dy stages: 136 KB (K % 128 == 0), 144 KB (K = 64 on twelve waves), 112 KB (six waves)
  const size_t lds = (size_t)(512 + 64 * PS + 128) * 128 + 3 * (size_t)(64 * PS) * 128 * KH;
#define MPR_WGW(KH_, PS_, PRIO_, PROBE_, M16_)                                                                         \
  do {                                                                                                                 \
    static bool attr_set = false;                                                                                      \
    if (!attr_set) {                                                                                                   \
      hipFuncSetAttribute((const void*)conv_wgrad_win_kernel<KH_, PS_, PRIO_, PROBE_, M16_>,                           \
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                     \
      attr_set = true;                                                                                                 \
    }                                                                                                                  \
    conv_wgrad_win_kernel<KH_, PS_, PRIO_, PROBE_, M16_><<<grid, 384 * KH_ * PS_, lds, st>>>(p);                       \
  } while (0)
#define MPR_WGW3(KH_, PS_, M16_)                                                                                       \
  do {                                                                                                                 \
    if (p.probe) { if (prio) MPR_WGW(KH_, PS_, 1, true, M16_); else MPR_WGW(KH_, PS_, 0, true, M16_); }                \
    else { if (prio) MPR_WGW(KH_, PS_, 1, false, M16_); else MPR_WGW(KH_, PS_, 0, false, M16_); }                      \
  } while (0)
  // (dbg bit 6: the 32x32x16 form, comparisons)
#define MPR_WGW2(KH_, PS_) do { if (p.dbg & 64) MPR_WGW3(KH_, PS_, false); else MPR_WGW3(KH_, PS_, true); } while (0)
  // s_setprio by k-step (a wave that is behind outranks one that is ahead): 3125 -> 2750 cycles per chunk at K % 128 == 0,
  // 139 -> 134 us per layer1 launch; dbg bit 4 switches it off (comparisons)
  const bool prio = !(p.dbg & 16);
  if (PS == 2) MPR_WGW2(1, 2);
  else if (KH == 2) MPR_WGW2(2, 1);
  else MPR_WGW2(1, 1);
#undef MPR_WGW2
#undef MPR_WGW3
#undef MPR_WGW
  MPR_LAUNCH_CHECK("conv_wgrad_win_kernel");
  if (p.part) {
    const int n4 = K * p.Ng / 4;
    if (nsplit >= 32) wgw_reduce_kernel<16><<<ceil_div(n4, 16), 256, 0, st>>>((const float4*)p.part, nsplit, dw, n4);
    else if (nsplit >= 8) wgw_reduce_kernel<4><<<ceil_div(n4, 64), 256, 0, st>>>((const float4*)p.part, nsplit, dw, n4);
    else wgw_reduce_kernel<1><<<ceil_div(n4, 256), 256, 0, st>>>((const float4*)p.part, nsplit, dw, n4);
    MPR_LAUNCH_CHECK("wgw_reduce_kernel");
  }
  return MPR_OK;
}

}  // extern "C"
