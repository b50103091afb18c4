// Shifted-window implicit GEMM for 3x3 / stride 1 / pad 1 convolutions (forward and data gradient) -- the 13 "body"
// convolutions of a ResNet-18 (timm BasicBlock conv1/conv2 behind src/image_encoder.py:24), which carry 80 % of its
// FLOPs.  (On maps narrower than 8 pixels the FORWARD pass takes conv_igemm.hip's 256 x 256 tile instead: the padded raster
// below costs +31 % positions at 7 x 7, mpr_conv_set_window_fwd_min_width.)
//
// Why a second kernel: measured per-workgroup phase times (scripts/conv_phases.py) put the plain LDS-DMA implicit
// GEMM (conv_igemm.hip) exactly on the CU's global->LDS fill rate (~73 GB/s per CU, MI355X guide "gather into LDS"):
// it re-fetches every input pixel once per filter tap, 32 KB of operands per 2.1 MFLOP chunk.  Here the activation
// operand of a 256-row tile is loaded ONCE per 64-channel block as a haloed window and all nine taps read it at
// shifted LDS rows; only the weight panel still streams per (tap, channel block).  That is 185 KB per 37.7 MFLOP
// (204 FLOP/B instead of 64), which moves the loop from the fill rate to the MFMA pipe.
//
// The shift must be LINEAR in the LDS row for that, so pixels are numbered in a padded raster: one zero column after
// every image row and one zero row after every image (G = (b*(H+1) + h)*(W+1) + w).  Tap (r, s) of output G then
// reads position G + (r-1)*(W+1) + (s-1) (data gradient: the mirrored shift), and every out-of-image neighbour IS a
// pad position -- which the window DMA zero-fills through an out-of-range buffer offset.  An output tile is 256
// consecutive padded positions (3.6 % .. 31 % of them pads, dropped in the epilogue); its window is the tile plus
// W+2 rows of halo on both sides.
//
// Workgroup = 4 waves.  Wide tile 256 x 128: waves 2 x 2, each 128 x 64 as 4 x 2 v_mfma_f32_32x32x16_bf16 (a weight
// fragment feeds four MFMAs, an activation fragment two); narrow tile 256 x 64 (64 output channels): waves 4 x 1,
// each 64 x 64.  LDS: window (<= 47 KB) + 2-stage weight ring (2 x 16 KB) -> two workgroups per CU.
#include "common.h"

struct WinParams {
  const bf16_t* src;   // [B,H,W,C] (forward: x; data gradient: dy)
  const bf16_t* wpk;   // packed panel [Npad128][9*C]
  bf16_t* dst;         // [B,H,W,Nout]
  const bf16_t* add;   // optional residual [B,H,W,Nout]
  float* stats;        // optional [tiles_m][2][Nout]
  int stat_slices;     // > 0: workgroups ADD into stats[index % stat_slices] (zeroed by the host), see mpr_conv_set_stat_slices
  int H, W, C, Nout;
  int Wp, img, halo, Gtot, wrows;   // W+1, (H+1)*(W+1), W+2, B*img, 256 + 2*halo
  int ntn, Kgpad, ncb;
  unsigned src_bytes, wpk_bytes;
  FastDiv div_img, div_wp;
  unsigned long long* probe;   // timing experiments: 8 x uint64 of shader-clock sums per workgroup, or NULL
  // BatchNorm-backward fusion of a data gradient (template BNB): the result is the gradient w.r.t. a BatchNorm OUTPUT
  // that went through a ReLU, so the epilogue applies the ReLU mask and accumulates the two BatchNorm-backward sums
  // (sum dz, sum dz * xhat) into `stats` -- the separate reduce pass over (dy, y, x) disappears.
  int mask_mode;               // 1: mask = mask_y > 0;  2: mask = bf16(bn_x * scale + shift) > 0 (ReLU mask recomputed)
  int coef_off;                // byte offset in dynamic LDS of the [4][BN] fp32 table of BatchNorm coefficients (BNB)
  const bf16_t* mask_y;        // [B,H,W,Nout] (mode 1)
  const bf16_t* bn_x;          // [B,H,W,Nout] BatchNorm input: xhat = (bn_x - mean) * invstd
  const float *bn_mean, *bn_invstd, *bn_scale, *bn_shift;   // [Nout]
  const uint32_t* rtab;        // raster table (runtime.cpp: mpr_raster_table): [256 + G] = pixel + 1, 0 = pad (conv_win_l1_kernel)
  unsigned rtab_bytes;
};
#define WIN_RASTER_MARGIN 256   // = MPR_RASTER_MARGIN of runtime.cpp

__device__ __forceinline__ int win_swz(int row, int chunk) {   // byte offset in a [rows][128 B] image
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

__device__ __forceinline__ void win_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt_win() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int V>
struct WinI { static constexpr int value = V; };

// BNB: 0 = plain; 1 = BatchNorm-backward fusion with the ReLU mask read from mask_y; 2 = ... with the mask recomputed from
// the BatchNorm input (compile-time: the two modes keep different operands alive, and the 8-wave tile must stay under 128
// VGPRs for its second workgroup per CU -- at one per CU nothing runs beside a workgroup's prologue and epilogue:
// +100 us per layer2 launch, five times what the extra map read costs)
// M16: the products as v_mfma_f32_16x16x32_bf16 (wave tile 64 x 64 = 4 x 4 sub-tiles): the chip holds a higher clock on this
// shape (scripts/micro/mfma_shape.hip).  A 32-deep step needs 4 + 4 fragments of 16 x 32 -- twice the registers of the 32x32x16
// form at the same wave tile -- so the fragments travel in PAIRS through four slots in a snake over the tile's quadrants.
template <int WM, int WN, int TM, int TN, int STAGES, bool DGRAD, bool ADD = false, int BNB = 0, bool M16 = false>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 8 ? 4 : 2)) void conv_win_kernel(const WinParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NW = WM * WN, T = 64 * NW;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  static_assert(BM == 256 && TM % 2 == 0, "tile height is fixed at 256 rows");
  constexpr int WI_MAX = (BM + 2 * 58 + 7) / 8;        // window DMA instructions at the largest halo (W = 56)
  constexpr int WIW = (WI_MAX + NW - 1) / NW;          // ... per wave
  constexpr int B_IT = BN / 8 / NW;
  static_assert(B_IT >= 1, "weight tile too small for the DMA instruction shape");
  constexpr int BSTAGE = BN * 128;
  constexpr int CS_STRIDE = BN * 4 + 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = bid / p.ntn, nt = bid - mt * p.ntn;
  const int G0 = mt * BM, n0 = nt * BN;
  const int wrows8 = (p.wrows + 7) & ~7;
  unsigned char* const bring = smem + wrows8 * 128;

  // ---- window DMA geometry: instruction I covers window rows 8I..8I+7, lane -> (row 8I + lane/8, physical chunk lane%8);
  //      the bank swizzle (logical chunk = physical ^ ((row >> 1) & 7)) is applied on the SOURCE side
  uint32_t wvoff[WIW];
  uint32_t wmask = 0;
#pragma unroll
  for (int k = 0; k < WIW; ++k) {
    const int I = wid + k * NW;
    const int j = 8 * I + (lane >> 3);
    const int G = G0 - p.halo + j;
    wvoff[k] = 0;
    if (j < p.wrows && G >= 0 && G < p.Gtot) {
      const uint32_t b = fdiv(G, p.div_img);
      const uint32_t pp = G - b * p.img;
      const uint32_t hh = fdiv(pp, p.div_wp);
      const uint32_t ww = pp - hh * p.Wp;
      if (hh < (uint32_t)p.H && ww < (uint32_t)p.W) {
        const uint32_t pix = (b * p.H + hh) * p.W + ww;
        const int lc = (lane & 7) ^ ((4 * I + (lane >> 4)) & 7);
        wvoff[k] = pix * (uint32_t)(2 * p.C) + lc * 16;
        wmask |= 1u << k;
      }
    }
  }
  uint32_t boff[B_IT];
#pragma unroll
  for (int j = 0; j < B_IT; ++j) {
    const int q = wid * B_IT + j;
    const int lc = (lane & 7) ^ ((4 * q + (lane >> 4)) & 7);
    boff[j] = ((uint32_t)(n0 + 8 * q + (lane >> 3)) * (uint32_t)p.Kgpad + lc * 8) * 2;
  }
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, p.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpk, 0, p.wpk_bytes, 0x00020000);

  auto issue_window = [&](int cb) {
#pragma unroll
    for (int k = 0; k < WIW; ++k) {
      const int I = wid + k * NW;
      if (8 * I < wrows8) {     // wave-uniform
        const uint32_t v = ((wmask >> k) & 1u) ? wvoff[k] + cb * 128 : 0xFFFFFFF0u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(smem + I * 1024), 16, v,
                                                 0, 0, 0);
      }
    }
  };
  auto issue_weights = [&](int tap, int cb, int stage) {
    const int soff = (tap * p.ncb + cb) * 128;
#pragma unroll
    for (int j = 0; j < B_IT; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rs_b, (__attribute__((address_space(3))) void*)(bring + stage * BSTAGE + (wid * B_IT + j) * 1024), 16, boff[j],
          soff, 0, 0);
  };

  static_assert(!M16 || (TM == 2 && TN == 2), "the 16x16x32 form is written for 64 x 64 wave tiles");
  f32x16 acc[M16 ? 1 : TN][M16 ? 1 : TM];
  f32x4 acc16[M16 ? 4 : 1][M16 ? 4 : 1];      // [channel sub-tile of 16][pixel sub-tile of 16]
#pragma unroll
  for (int j = 0; j < (M16 ? 1 : TN); ++j)
#pragma unroll
    for (int i = 0; i < (M16 ? 1 : TM); ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;
#pragma unroll
  for (int j = 0; j < (M16 ? 4 : 1); ++j)
#pragma unroll
    for (int i = 0; i < (M16 ? 4 : 1); ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc16[j][i][e] = 0.f;
  const int fr16 = lane & 15, fq = lane >> 4;      // (16x16x32: lane -> row of a 16-row sub-tile, 8-channel quarter of a 32-deep step)

  const int frow = lane & 31, fh = lane >> 5;
  const int baseA = wm * (TM * 32) + frow + p.halo;       // window row of this lane's pixel, m-tile 0, no shift
  uint32_t b_rd[M16 ? 1 : TN][M16 ? 2 : 4];      // 32x32x16: [n-tile][k-step of 16]
#pragma unroll
  for (int j = 0; j < (M16 ? 1 : TN); ++j)
#pragma unroll
    for (int ks = 0; ks < (M16 ? 2 : 4); ++ks) {
      // 16x16x32: [k-step of 32] of channel sub-tile 0; sub-tile t sits 16 rows = 2048 B further with the same swizzle key
      if constexpr (M16) b_rd[0][ks] = win_swz(wn * 64 + fr16, ks * 4 + fq);
      else b_rd[j][ks] = win_swz(wn * (TN * 32) + j * 32 + frow, ks * 2 + fh);
    }

  // (timing probe: wave 0's shader-clock sums of the loop phases; all dead code when p.probe == NULL)
  unsigned long long pr_t0 = 0, pr_wait = 0, pr_bar = 0, pr_comp = 0, pr_a = 0, pr_b = 0, pr_loop = 0, pr_rbar = 0,
                     pr_riss = 0, pr_ew = 0, pr_er = 0;
#define PROBE_NOW() (p.probe ? __builtin_readcyclecounter() : 0ull)
  pr_t0 = PROBE_NOW();
  issue_window(0);
  const int nk = 9 * p.ncb;
  // weight ring: chunk kc lives in stage kc % STAGES and is issued STAGES-1 chunks ahead (the loop is bound by the
  // LATENCY of that stream: one chunk of look-ahead leaves every iteration waiting for its own weights)
  int itap = 0, icb = 0, issued = 0;          // (tap, channel block) of the next chunk to issue
  auto issue_next = [&]() {
    issue_weights(itap, icb, issued % STAGES);
    ++issued;
    if (++itap == 9) { itap = 0; ++icb; }
  };
  for (int i = 0; i < STAGES - 1 && i < nk; ++i) issue_next();
  int tap = 0, cb = 0, tr = 0, ts = 0;
  for (int kc = 0; kc < nk; ++kc) {
    bool drain = kc == 0;
    if (tap == 0 && cb > 0) {
      // next 64-channel block: every wave is done with the old window (it sits behind this barrier), reload it
      const unsigned long long q0 = PROBE_NOW();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const unsigned long long q1 = PROBE_NOW();
      issue_window(cb);
      pr_rbar += q1 - q0;
      pr_riss += PROBE_NOW() - q1;
      drain = true;
    }
    // retire chunk kc: at most the (issued - kc - 1) younger weight chunks of THIS wave may still be in flight
    const int younger = issued - kc - 1;
    pr_a = PROBE_NOW();
    if (!drain && STAGES >= 3 && younger == STAGES - 2) wait_vmcnt_win<(STAGES - 2) * B_IT>();
    else wait_vmcnt_win<0>();
    pr_b = PROBE_NOW();
    pr_wait += pr_b - pr_a;
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    pr_a = PROBE_NOW();
    pr_bar += pr_a - pr_b;

    // shifted fragment rows of this tap: rows of the TM m-tiles are 32 apart -> same swizzle key, immediate offsets
    const int d = DGRAD ? (1 - tr) * p.Wp + (1 - ts) : (tr - 1) * p.Wp + (ts - 1);
    const int rowA = baseA + d;
    const int key = (rowA >> 1) & 7;
    const unsigned char* const arow = smem + rowA * 128;
    const unsigned char* const bst = bring + (kc % STAGES) * BSTAGE;
    if constexpr (M16) {
      // pairs of fragments in four slots, snake over the quadrants of the 64 x 64 wave tile (Bx = channel sub-tiles 2x, 2x+1;
      // Ax = pixel sub-tiles 2x, 2x+1; ".k" = 32-deep step k of the chunk):
      //   q0 B0.0 A0.0 | q1 B1.0 A0.0 | q2 B1.0 A1.0 | q3 B0.0 A1.0 | q4 B0.1 A1.1 | q5 B1.1 A1.1 | q6 B1.1 A0.1 | q7 B0.1 A0.1
      // every pair is read once per chunk (8 pair reads = the 16 ds_read_b128 of the 32x32x16 form) into the slot its
      // predecessor has just left; 32 fragment registers as before.
      const int rowA16 = wm * 64 + fr16 + p.halo + d;
      const int key16 = (rowA16 >> 1) & 7;
      const unsigned char* const arow16 = smem + rowA16 * 128;
      bf16x8 S0[2], S1[2], S2[2], S3[2];
      auto ldA = [&](bf16x8* s, int ia, int ks) {      // pixel sub-tiles 2 ia, 2 ia + 1
        s[0] = *reinterpret_cast<const bf16x8*>(arow16 + (2 * ia) * 2048 + (((4 * ks + fq) ^ key16) << 4));
        s[1] = *reinterpret_cast<const bf16x8*>(arow16 + (2 * ia + 1) * 2048 + (((4 * ks + fq) ^ key16) << 4));
      };
      auto ldB = [&](bf16x8* s, int jb, int ks) {      // channel sub-tiles 2 jb, 2 jb + 1
        s[0] = *reinterpret_cast<const bf16x8*>(bst + b_rd[0][ks] + (2 * jb) * 2048);
        s[1] = *reinterpret_cast<const bf16x8*>(bst + b_rd[0][ks] + (2 * jb + 1) * 2048);
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
      if (issued < nk) issue_next();     // (into the stage chunk kc-1 just vacated) behind the first fragment reads
      ldB(S2, 1, 0); ldA(S3, 1, 0);
      __builtin_amdgcn_sched_barrier(0);
      quad(S1, S0, WinI<0>{}, WinI<0>{});                                   // q0: B0.0 A0.0
      __builtin_amdgcn_sched_barrier(0);
      quad(S2, S0, WinI<1>{}, WinI<0>{});                                   // q1: B1.0 A0.0
      __builtin_amdgcn_sched_barrier(0);
      ldA(S0, 1, 1);                                                        //     A1.1 -> S0
      __builtin_amdgcn_sched_barrier(0);
      quad(S2, S3, WinI<1>{}, WinI<1>{});                                   // q2: B1.0 A1.0
      __builtin_amdgcn_sched_barrier(0);
      ldB(S2, 0, 1);                                                        //     B0.1 -> S2
      __builtin_amdgcn_sched_barrier(0);
      quad(S1, S3, WinI<0>{}, WinI<1>{});                                   // q3: B0.0 A1.0
      __builtin_amdgcn_sched_barrier(0);
      ldA(S3, 0, 1); ldB(S1, 1, 1);                                         //     A0.1 -> S3, B1.1 -> S1
      __builtin_amdgcn_sched_barrier(0);
      quad(S2, S0, WinI<0>{}, WinI<1>{});                                   // q4: B0.1 A1.1
      __builtin_amdgcn_sched_barrier(0);
      quad(S1, S0, WinI<1>{}, WinI<1>{});                                   // q5: B1.1 A1.1
      __builtin_amdgcn_sched_barrier(0);
      quad(S1, S3, WinI<1>{}, WinI<0>{});                                   // q6: B1.1 A0.1
      __builtin_amdgcn_sched_barrier(0);
      quad(S2, S3, WinI<0>{}, WinI<0>{});                                   // q7: B0.1 A0.1
      __builtin_amdgcn_sched_barrier(0);
    } else {
    bf16x8 af[2][TM], bfr[2][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[0][i] = *reinterpret_cast<const bf16x8*>(arow + i * 4096 + ((fh ^ key) << 4));
#pragma unroll
      for (int j = 0; j < TN; ++j) bfr[0][j] = *reinterpret_cast<const bf16x8*>(bst + b_rd[j][0]);
      __builtin_amdgcn_sched_barrier(0);
      if (issued < nk) issue_next();     // (into the stage chunk kc-1 just vacated) behind the first fragment reads
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int cur = ks & 1, nxt = cur ^ 1;
        if (ks < 3) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
            af[nxt][i] = *reinterpret_cast<const bf16x8*>(arow + i * 4096 + (((2 * (ks + 1) + fh) ^ key) << 4));
#pragma unroll
          for (int j = 0; j < TN; ++j) bfr[nxt][j] = *reinterpret_cast<const bf16x8*>(bst + b_rd[j][M16 ? 0 : ks + 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int i = 0; i < TM; ++i)
            acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[cur][j], af[cur][i], acc[j][i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (++ts == 3) { ts = 0; ++tr; }
    if (++tap == 9) { tap = 0; tr = 0; ++cb; }
    pr_comp += PROBE_NOW() - pr_a;
  }
  pr_loop = PROBE_NOW();
  // every DMA has landed; say so with a wait the compiler can see (see conv_igemm.hip)
  __builtin_amdgcn_s_waitcnt(0x0F70);

  // ---- epilogue: 64-row slabs of the fp32 tile -> LDS -> 16-B channel groups (+ residual, BN partial sums); pad
  //      positions of the raster are dropped here
  constexpr int CPR = BN / 8;
  constexpr int RPP = T / CPR;
  static_assert(64 % RPP == 0, "a slab must be whole row passes");
  const int ch = tid % CPR, rr = tid / CPR;
  const int ncol = n0 + ch * 8;
  const bool col_ok = ncol < p.Nout;
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  // BatchNorm-backward fusion: the coefficients of this tile's BN channels live in LDS ([mean | invstd | scale | shift][BN],
  // behind everything the epilogue stages: registers are what keeps the second workgroup off the CU)
  float* const coef = reinterpret_cast<float*>(smem + p.coef_off);
  // (the table sits inside the window / ring allocation: written only AFTER the barrier that ends every wave's main loop,
  //  published by the barrier between the staging writes and reads)
  auto fill_coef = [&]() {
    if (BNB && tid < BN) {
      const int n = n0 + tid < p.Nout ? n0 + tid : 0;
      coef[tid] = p.bn_mean[n];
      coef[BN + tid] = p.bn_invstd[n];
      if (BNB == 2) {
        coef[2 * BN + tid] = p.bn_scale[n];
        coef[3 * BN + tid] = p.bn_shift[n];
      }
    }
  };
  // dz = mask * q (q: the bf16-rounded gradient values), sums of dz and dz * xhat; returns the masked, re-packed group.
  // xv / yv: this group of the BatchNorm input / of the mask source, loaded by the caller (all rows at once: the loop
  // is latency-bound otherwise)
  auto bnb_group = [&](float* q, const uint4& xv, const uint4& yv) -> uint4 {
    float xf[8];
    unpack8(xv, xf);
    if (BNB == 1) {
      float ym[8];
      unpack8(yv, ym);
#pragma unroll
      for (int e = 0; e < 8; ++e) q[e] = ym[e] > 0.f ? q[e] : 0.f;
    } else {
      // bf16(v) > 0 <=> v > 0: rounding keeps the sign, and bf16 has fp32's exponent range
      const float4 c0 = *reinterpret_cast<const float4*>(coef + 2 * BN + ch * 8), c1 = *reinterpret_cast<const float4*>(coef + 2 * BN + ch * 8 + 4);
      const float4 h0 = *reinterpret_cast<const float4*>(coef + 3 * BN + ch * 8), h1 = *reinterpret_cast<const float4*>(coef + 3 * BN + ch * 8 + 4);
      const float sc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w}, sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) q[e] = fmaf(xf[e], sc[e], sh[e]) > 0.f ? q[e] : 0.f;
    }
    // (s2 collects sum dz * x here; the thread's channels are fixed, so sum dz * xhat = invstd * (s2 - mean * s1) is formed ONCE
    //  behind the tile's rows -- bnb_finish -- instead of a subtract and a multiply per element: the fused epilogues are bound by
    //  their VALU instructions, ~76 per 8-channel group before this)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s1[e] += q[e];
      s2[e] = fmaf(q[e], xf[e], s2[e]);
    }
    return pack8(q);
  };
  auto bnb_finish = [&]() {
    const float4 m0 = *reinterpret_cast<const float4*>(coef + ch * 8), m1 = *reinterpret_cast<const float4*>(coef + ch * 8 + 4);
    const float4 i0 = *reinterpret_cast<const float4*>(coef + BN + ch * 8), i1 = *reinterpret_cast<const float4*>(coef + BN + ch * 8 + 4);
    const float mu[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w}, is[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) s2[e] = (s2[e] - mu[e] * s1[e]) * is[e];
  };
  if (!ADD) {
    // No residual: the tile is rounded to bf16 IN REGISTERS and staged as bf16 (half the LDS bytes), every wave writing
    // its own rows at once -- one pass of 256 rows at BN = 64, two of 128 rows at BN = 128 (LDS capacity) -- and the
    // read side hands finished 16-B channel groups to the global store.  (The slab path below serialises the waves and
    // moves fp32; it is kept for the fused residual add, which must see the unrounded accumulator.)
    // (the fused BatchNorm-backward epilogue of the 128-channel tile also stages ALL 256 rows at once -- 69.6 KB, inside the
    //  window + ring allocation -- so that no wave holds its accumulators while another pass is read: its rows then fit the 128
    //  VGPRs in ONE burst of global loads instead of four, each of which was a full round trip: round-3 probe, 16.4 K of the
    //  workgroup's 73.7 K cycles at layer2)
    constexpr int EP = (BN == 64 || BNB != 0) ? 1 : 2;          // passes
    constexpr int RPASS = BM / EP;                // rows per pass
    constexpr int RS = BN * 2 + 16;               // staged row stride (bytes): 16 rows of a write group hit 16 bank pairs
    static_assert(RPASS % (TM * 32) == 0 && RPASS % RPP == 0, "a pass is whole wave rows");
#pragma unroll
    for (int ep = 0; ep < EP; ++ep) {
      win_lds_barrier();
      if (ep == 0) fill_coef();
      const unsigned long long e0 = PROBE_NOW();
      if ((wm * TM * 32) / RPASS == ep) {
        const int wrow0 = wm * TM * 32 - ep * RPASS;
        if constexpr (M16) {
#pragma unroll
          for (int i2 = 0; i2 < 4; ++i2)
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) {
              const int row = wrow0 + i2 * 16 + fr16;
              const int col = wn * 64 + j2 * 16 + 4 * fq;
              uint2 v;
              v.x = pack_bf16x2(acc16[j2][i2][0], acc16[j2][i2][1]);
              v.y = pack_bf16x2(acc16[j2][i2][2], acc16[j2][i2][3]);
              *reinterpret_cast<uint2*>(smem + row * RS + col * 2) = v;
            }
        } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int row = wrow0 + i * 32 + frow;
              const int col = wn * (TN * 32) + j * 32 + 8 * g + 4 * fh;
              uint2 v;
              v.x = pack_bf16x2(acc[M16 ? 0 : j][M16 ? 0 : i][4 * g], acc[M16 ? 0 : j][M16 ? 0 : i][4 * g + 1]);
              v.y = pack_bf16x2(acc[M16 ? 0 : j][M16 ? 0 : i][4 * g + 2], acc[M16 ? 0 : j][M16 ? 0 : i][4 * g + 3]);
              *reinterpret_cast<uint2*>(smem + row * RS + col * 2) = v;
            }
        }
      }
      win_lds_barrier();
      const unsigned long long e1 = PROBE_NOW();
      pr_ew += e1 - e0;
      // all of this thread's rows are read in one burst (unconditionally: the LDS latency is paid once, not per row),
      // the raster decode runs underneath, then the stores go out
      constexpr int NSUB = (BNB == 1 && T == 512) ? 2 : 1;      // (mask from the block output: three maps per row -- two bursts)
      constexpr int NR = RPASS / RPP / NSUB;
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub) {
        uint4 pk[NR];
#pragma unroll
        for (int n = 0; n < NR; ++n) pk[n] = *reinterpret_cast<const uint4*>(smem + (rr + (sub * NR + n) * RPP) * RS + ch * 16);
        uint32_t pix[NR];
#pragma unroll
        for (int n = 0; n < NR; ++n) {
          const int G = G0 + ep * RPASS + rr + (sub * NR + n) * RPP;
          const uint32_t Gc = G < p.Gtot ? G : 0;
          const uint32_t b = fdiv(Gc, p.div_img);
          const uint32_t pp = Gc - b * p.img;
          const uint32_t hh = fdiv(pp, p.div_wp);
          const uint32_t ww = pp - hh * p.Wp;
          const bool ok = col_ok && G < p.Gtot && hh < (uint32_t)p.H && ww < (uint32_t)p.W;
          pix[n] = ok ? (b * p.H + hh) * p.W + ww : 0xFFFFFFFFu;
        }
        uint4 xv[BNB ? NR : 1], yv[BNB ? NR : 1];
        if (BNB) {
#pragma unroll
          for (int n = 0; n < NR; ++n) {
            const size_t o = (size_t)(pix[n] != 0xFFFFFFFFu ? pix[n] : 0u) * p.Nout + (col_ok ? ncol : 0);
            xv[n] = *reinterpret_cast<const uint4*>(p.bn_x + o);
            yv[n] = BNB == 1 ? *reinterpret_cast<const uint4*>(p.mask_y + o) : make_uint4(0, 0, 0, 0);
          }
        }
#pragma unroll
        for (int n = 0; n < NR; ++n) {
          if (pix[n] != 0xFFFFFFFFu) {
            const size_t o = (size_t)pix[n] * p.Nout + ncol;
            if (BNB) {
              float q[8];
              unpack8(pk[n], q);
              *reinterpret_cast<uint4*>(p.dst + o) = bnb_group(q, xv[n], yv[n]);
            } else {
              *reinterpret_cast<uint4*>(p.dst + o) = pk[n];
              if (p.stats) {
                float q[8];
                unpack8(pk[n], q);
#pragma unroll
                for (int e = 0; e < 8; ++e) { s1[e] += q[e]; s2[e] += q[e] * q[e]; }
              }
            }
          }
        }
      }
      pr_er += PROBE_NOW() - e1;
    }
  } else {
    // Fused residual (the skip-connection gradient of a data gradient) must see the UNROUNDED accumulator: the tile is
    // staged as fp32, every wave whose rows are in the pass writing at once -- the whole 256 x 64 tile in ONE pass (69 KB,
    // inside the window + ring allocation), a 256 x 128 tile in four 64-row passes -- and the read side adds the residual
    // in coalesced 16-B channel groups.  (Until round 2 this walked 64-row slabs with one writing wave each: four barrier
    // pairs on the 64-channel tile, 289 vs 194 us without the residual on the 64 -> 64 @ 56x56 data gradient.)
    constexpr int EPA = BN == 64 ? 1 : 4;
    constexpr int RPA = BM / EPA;
    static_assert(RPA % RPP == 0 && RPA % 32 == 0, "a pass is whole row passes of whole m-tiles");
#pragma unroll
    for (int ep = 0; ep < EPA; ++ep) {
      win_lds_barrier();
      if (ep == 0) fill_coef();
      if constexpr (M16) {
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) {
          const int grow = wm * 64 + i2 * 16;                // this 16-row sub-tile inside the 256-row tile (wave-uniform)
          if (grow / RPA == ep) {
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) {
              const int row = grow - ep * RPA + fr16;
              const int col = wn * 64 + j2 * 16 + 4 * fq;
              float4 v = make_float4(acc16[j2][i2][0], acc16[j2][i2][1], acc16[j2][i2][2], acc16[j2][i2][3]);
              *reinterpret_cast<float4*>(smem + row * CS_STRIDE + col * 4) = v;
            }
          }
        }
      } else {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int grow = wm * (TM * 32) + i * 32;          // this 32-row m-tile inside the 256-row tile (wave-uniform)
        if (grow / RPA == ep) {
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int row = grow - ep * RPA + frow;
              const int col = wn * (TN * 32) + j * 32 + 8 * g + 4 * fh;
              float4 v = make_float4(acc[M16 ? 0 : j][M16 ? 0 : i][4 * g], acc[M16 ? 0 : j][M16 ? 0 : i][4 * g + 1],
                                     acc[M16 ? 0 : j][M16 ? 0 : i][4 * g + 2], acc[M16 ? 0 : j][M16 ? 0 : i][4 * g + 3]);
              *reinterpret_cast<float4*>(smem + row * CS_STRIDE + col * 4) = v;
            }
        }
      }
      }
      win_lds_barrier();
      constexpr int NSUBA = (BNB && T == 512) ? 2 : 1;      // (as above: 128 VGPRs)
      constexpr int NRA = RPA / RPP / NSUBA;
      // raster decode of all of this thread's rows, then ALL their global loads (residual, BatchNorm input, mask) in one
      // burst, then the LDS reads and the arithmetic: one exposed latency per pass instead of one per row
#pragma unroll
      for (int sub = 0; sub < NSUBA; ++sub) {
        uint32_t pixa[NRA];
#pragma unroll
        for (int n = 0; n < NRA; ++n) {
          const int G = G0 + ep * RPA + rr + (sub * NRA + n) * RPP;
          const uint32_t Gc = G < p.Gtot ? G : 0;
          const uint32_t b = fdiv(Gc, p.div_img);
          const uint32_t pp = Gc - b * p.img;
          const uint32_t hh = fdiv(pp, p.div_wp);
          const uint32_t ww = pp - hh * p.Wp;
          const bool ok = col_ok && G < p.Gtot && hh < (uint32_t)p.H && ww < (uint32_t)p.W;
          pixa[n] = ok ? (b * p.H + hh) * p.W + ww : 0xFFFFFFFFu;
        }
        uint4 av[NRA], xv[BNB ? NRA : 1], yv[BNB ? NRA : 1];
#pragma unroll
        for (int n = 0; n < NRA; ++n) {
          const size_t o = (size_t)(pixa[n] != 0xFFFFFFFFu ? pixa[n] : 0u) * p.Nout + (col_ok ? ncol : 0);
          av[n] = *reinterpret_cast<const uint4*>(p.add + o);
          if (BNB) {
            xv[n] = *reinterpret_cast<const uint4*>(p.bn_x + o);
            yv[n] = BNB == 1 ? *reinterpret_cast<const uint4*>(p.mask_y + o) : make_uint4(0, 0, 0, 0);
          }
        }
#pragma unroll
        for (int n = 0; n < NRA; ++n) {
          if (pixa[n] != 0xFFFFFFFFu) {
            const int r = rr + (sub * NRA + n) * RPP;
            const float4 lo = *reinterpret_cast<const float4*>(smem + r * CS_STRIDE + ch * 32);
            const float4 hi = *reinterpret_cast<const float4*>(smem + r * CS_STRIDE + ch * 32 + 16);
            float f[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            const size_t o = (size_t)pixa[n] * p.Nout + ncol;
            float g[8];
            unpack8(av[n], g);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += g[e];
            const uint4 pk = pack8(f);
            if (BNB) {
              float q[8];
              unpack8(pk, q);
              *reinterpret_cast<uint4*>(p.dst + o) = bnb_group(q, xv[n], yv[n]);
            } else {
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
      }
    }
  }
  if (p.stats) {
    // per-thread sums -> LDS [T][16 (+1 pad)] -> CPR*16 threads each add the RPP rows of one (channel group, value):
    // one barrier, no cross-lane traffic (a shuffle tree here is 32-48 dependent ds_bpermutes)
    if constexpr (BNB != 0) bnb_finish();      // (the coefficient table is read before the reduction overwrites the staging area)
    win_lds_barrier();
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[tid * 17 + e] = s1[e];
      red[tid * 17 + 8 + e] = s2[e];
    }
    win_lds_barrier();
    if (tid < CPR * 16) {
      const int c = tid >> 4, e = tid & 15;
      float v = 0.f;
#pragma unroll 8
      for (int q = 0; q < RPP; ++q) v += red[(q * CPR + c) * 17 + e];
      const int n = n0 + c * 8 + (e & 7);
      if (n < p.Nout) {
        if (p.stat_slices > 0) atomicAdd(&p.stats[((size_t)(mt % p.stat_slices) * 2 + (e >> 3)) * p.Nout + n], v);
        else p.stats[((size_t)mt * 2 + (e >> 3)) * p.Nout + n] = v;
      }
    }
  }
  if (p.probe && tid == 0) {
    unsigned long long* o = p.probe + (size_t)blockIdx.x * 16;
    const unsigned long long t_end = __builtin_readcyclecounter();
    o[0] = t_end - pr_t0; o[1] = pr_wait; o[2] = pr_bar; o[3] = pr_comp; o[4] = t_end - pr_loop; o[5] = wall_clock64();
    o[6] = pr_rbar; o[7] = pr_riss; o[8] = pr_ew; o[9] = pr_er;
  }
#undef PROBE_NOW
#endif   // __HIP_DEVICE_COMPILE__
}

// ------------------------------------------------------------------------------------------------------------------
// LDS staging accesses as INLINE ASM: hipcc guards every LDS access it cannot prove disjoint from an LDS-DMA in flight with
// s_waitcnt vmcnt(0) -- in the persistent kernel that was the epilogue's first staging write waiting for the WHOLE window of
// the next tile, issued a moment earlier precisely to land underneath that epilogue (per-tile probe: epilogue 12.5 K of
// 27 K cycles).  The staging slices and the window are disjoint by construction; completion is the s_waitcnt lgkmcnt(0)
// the code already carries.
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
__device__ __forceinline__ void win_lds_write_b64(uint32_t addr, u32x2_t v) {
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ u32x4_t win_lds_read_b128(uint32_t addr) {
  u32x4_t v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void win_lds_write_b128(uint32_t addr, u32x4_t v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ u32x2_t win_lds_read_b64(uint32_t addr) {
  u32x2_t v;
  asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}

// Persistent form of the 256 x 64 tile (N <= 64: ResNet layer1, whose 9-chunk loop is shorter than the tile's own
// prologue + epilogue).  One workgroup per (CU, slot) walks tiles blockIdx.x, blockIdx.x + gridDim.x, ...:
//   * the NEXT tile's window DMA is issued as soon as the loop of the current tile ends -- it lands underneath the
//     epilogue instead of in front of the next loop;
//   * the epilogue needs no workgroup barrier: every wave stages its own 64 x 64 block, 32 rows at a time, in a private
//     4 KB slice of LDS (rounded to bf16; the fused residual path of the data gradient stays on the plain kernel, it
//     must see the unrounded accumulator) and stores finished 128-B rows;
//   * the weight ring (2 stages) simply runs on across tiles;
//   * BatchNorm partial sums stay in registers over all tiles of the workgroup: gridDim.x partial rows per conv.
// LDS: window | 2 x 8 KB weight stages | 4 x 4 KB staging  (<= 80 KB: two workgroups per CU).
template <bool DGRAD>
__global__ __launch_bounds__(256, 2) void conv_win_persist_kernel(const WinParams p) {      // (two workgroups per CU: <= 256 VGPRs)
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NW = 4, BM = 256, BN = 64, TM = 2, TN = 2;
  constexpr int WI_MAX = (BM + 2 * 58 + 7) / 8, WIW = (WI_MAX + NW - 1) / NW;
  constexpr int B_IT = BN / 8 / NW;
  constexpr int BSTAGE = BN * 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid;
  const int wrows8 = (p.wrows + 7) & ~7;
  unsigned char* const bring = smem + wrows8 * 128;
  unsigned char* const stage = bring + 2 * BSTAGE + wid * 4096;      // this wave's private 32 x 128 B
  const uint32_t stage_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)stage;
  const int ntiles = (p.Gtot + BM - 1) / BM;

  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, p.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpk, 0, p.wpk_bytes, 0x00020000);
  uint32_t boff[B_IT];
#pragma unroll
  for (int j = 0; j < B_IT; ++j) {
    const int q = wid * B_IT + j;
    const int lc = (lane & 7) ^ ((4 * q + (lane >> 4)) & 7);
    boff[j] = ((uint32_t)(8 * q + (lane >> 3)) * (uint32_t)p.Kgpad + lc * 8) * 2;
  }
  uint32_t wvoff[WIW];
  uint32_t wmask = 0;
  auto decode_window = [&](int G0) {
    wmask = 0;
#pragma unroll
    for (int k = 0; k < WIW; ++k) {
      const int I = wid + k * NW;
      const int j = 8 * I + (lane >> 3);
      const int G = G0 - p.halo + j;
      wvoff[k] = 0;
      if (j < p.wrows && G >= 0 && G < p.Gtot) {
        const uint32_t b = fdiv(G, p.div_img);
        const uint32_t pp = G - b * p.img;
        const uint32_t hh = fdiv(pp, p.div_wp);
        const uint32_t ww = pp - hh * p.Wp;
        if (hh < (uint32_t)p.H && ww < (uint32_t)p.W) {
          const uint32_t pix = (b * p.H + hh) * p.W + ww;
          const int lc = (lane & 7) ^ ((4 * I + (lane >> 4)) & 7);
          wvoff[k] = pix * (uint32_t)(2 * p.C) + lc * 16;
          wmask |= 1u << k;
        }
      }
    }
  };
  auto issue_window = [&](int cb) {
#pragma unroll
    for (int k = 0; k < WIW; ++k) {
      const int I = wid + k * NW;
      if (8 * I < wrows8) {
        const uint32_t v = ((wmask >> k) & 1u) ? wvoff[k] + cb * 128 : 0xFFFFFFF0u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(smem + I * 1024), 16, v,
                                                 0, 0, 0);
      }
    }
  };
  auto issue_weights = [&](int tap, int cb, int st) {
    const int soff = (tap * p.ncb + cb) * 128;
#pragma unroll
    for (int j = 0; j < B_IT; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rs_b, (__attribute__((address_space(3))) void*)(bring + st * BSTAGE + (wid * B_IT + j) * 1024), 16, boff[j],
          soff, 0, 0);
  };

  const int frow = lane & 31, fh = lane >> 5;
  const int baseA = wm * (TM * 32) + frow + p.halo;
  uint32_t b_rd[TN][4];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) b_rd[j][ks] = win_swz(j * 32 + frow, ks * 2 + fh);

  // epilogue read geometry: lane -> (row lane/8 + 8*it, 16-B chunk lane%8); its 8 channels are fixed over the kernel
  const int erow = lane >> 3, ech = lane & 7;
  const bool col_ok = ech * 8 < p.Nout;
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }

  const int nk = 9 * p.ncb;
  int gk = 0;                                   // chunks done by this workgroup (weight ring position)
  // (timing probe, wave 0: cycles waiting for DMA / at the chunk barrier / in the MFMA steps / in the epilogue; tiles)
  unsigned long long pq_wait = 0, pq_bar = 0, pq_comp = 0, pq_epi = 0, pq_tiles = 0, pq_e1 = 0, pq_e2 = 0, pq_e3 = 0;
#define PQ_NOW() (p.probe ? __builtin_readcyclecounter() : 0ull)
  const unsigned long long pq_t0 = PQ_NOW();
  // tiles of one round go to the XCDs in contiguous runs (as in the plain kernel): neighbouring tiles' windows overlap by
  // the halo (45 % of a window at W = 56), which then comes from that XCD's L2 instead of HBM a second time
  int tile = xcd_remap(blockIdx.x, gridDim.x);
  if (tile < ntiles) {
    decode_window(tile * BM);
    issue_window(0);
    issue_weights(0, 0, 0);
  }
  for (; tile < ntiles; tile += gridDim.x) {
    const int G0 = tile * BM;
    const bool more = tile + (int)gridDim.x < ntiles;
    f32x16 acc[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;
    int tap = 0, cb = 0, tr = 0, ts = 0;
    for (int kc = 0; kc < nk; ++kc, ++gk) {
      if (tap == 0 && cb > 0) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue_window(cb);
      }
      const unsigned long long pq0 = PQ_NOW();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long pq1 = PQ_NOW();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const unsigned long long pq2 = PQ_NOW();
      pq_wait += pq1 - pq0;
      pq_bar += pq2 - pq1;
      const int d = DGRAD ? (1 - tr) * p.Wp + (1 - ts) : (tr - 1) * p.Wp + (ts - 1);
      const int rowA = baseA + d;
      const int key = (rowA >> 1) & 7;
      const unsigned char* const arow = smem + rowA * 128;
      const unsigned char* const bst = bring + (gk & 1) * BSTAGE;
      bf16x8 af[2][TM], bfr[2][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[0][i] = *reinterpret_cast<const bf16x8*>(arow + i * 4096 + ((fh ^ key) << 4));
#pragma unroll
      for (int j = 0; j < TN; ++j) bfr[0][j] = *reinterpret_cast<const bf16x8*>(bst + b_rd[j][0]);
      __builtin_amdgcn_sched_barrier(0);
      {   // weights of the next chunk of the ring -- the first chunk of the next tile after the last one of this tile
        int ntap = tap + 1, ncb2 = cb;
        if (ntap == 9) { ntap = 0; ++ncb2; }
        if (ncb2 == p.ncb) { ntap = 0; ncb2 = 0; }
        if (kc + 1 < nk || more) issue_weights(ntap, ncb2, (gk + 1) & 1);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int cur = ks & 1, nxt = cur ^ 1;
        if (ks < 3) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
            af[nxt][i] = *reinterpret_cast<const bf16x8*>(arow + i * 4096 + (((2 * (ks + 1) + fh) ^ key) << 4));
#pragma unroll
          for (int j = 0; j < TN; ++j) bfr[nxt][j] = *reinterpret_cast<const bf16x8*>(bst + b_rd[j][ks + 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int i = 0; i < TM; ++i)
            acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[cur][j], af[cur][i], acc[j][i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (++ts == 3) { ts = 0; ++tr; }
      if (++tap == 9) { tap = 0; tr = 0; ++cb; }
      // one channel block: this tile's window offsets are dead once its DMA is out, so the NEXT tile's ~400 instructions
      // of raster decoding run here, underneath the MFMAs -- behind the post-loop barrier they sat in front of the DMA issue
      // with every wave of the workgroup idle (per-tile probe: ~2.5 K of 25 K cycles)
      if (kc == 3 && more && p.ncb == 1) decode_window(G0 + (int)gridDim.x * BM);
      pq_comp += PQ_NOW() - pq2;
    }
    const unsigned long long pq3 = PQ_NOW();
    // every wave is done with this tile's window: the next tile's goes out now and lands underneath the epilogue
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (more) {
      if (p.ncb != 1) decode_window(G0 + (int)gridDim.x * BM);
      issue_window(0);
    }
    // ---- epilogue, wave-private: 32 rows x 64 channels at a time through this wave's 4 KB of LDS
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const unsigned long long pe0 = PQ_NOW();
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          u32x2_t v;
          v.x = pack_bf16x2(acc[j][i][4 * g], acc[j][i][4 * g + 1]);
          v.y = pack_bf16x2(acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]);
          win_lds_write_b64(stage_lds + frow * 128 + (((4 * j + g) ^ (frow & 7)) << 4) + fh * 8, v);
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // wave-private: no barrier
      const unsigned long long pe1 = PQ_NOW();
      u32x4_t pkv[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int r = erow + 8 * it;
        pkv[it] = win_lds_read_b128(stage_lds + r * 128 + ((ech ^ (r & 7)) << 4));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pkv[0]), "+v"(pkv[1]), "+v"(pkv[2]), "+v"(pkv[3])::"memory");
      const unsigned long long pe2 = PQ_NOW();
      uint4 pk[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) pk[it] = make_uint4(pkv[it].x, pkv[it].y, pkv[it].z, pkv[it].w);
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int G = G0 + wm * 64 + i * 32 + erow + 8 * it;
        const uint32_t Gc = G < p.Gtot ? G : 0;
        const uint32_t b = fdiv(Gc, p.div_img);
        const uint32_t pp = Gc - b * p.img;
        const uint32_t hh = fdiv(pp, p.div_wp);
        const uint32_t ww = pp - hh * p.Wp;
        if (col_ok && G < p.Gtot && hh < (uint32_t)p.H && ww < (uint32_t)p.W) {
          *reinterpret_cast<uint4*>(p.dst + (size_t)((b * p.H + hh) * p.W + ww) * p.Nout + ech * 8) = pk[it];
          if (p.stats) {
            float q[8];
            unpack8(pk[it], q);
#pragma unroll
            for (int e = 0; e < 8; ++e) { s1[e] += q[e]; s2[e] += q[e] * q[e]; }
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // reads done before the slice is rewritten
      pq_e1 += pe1 - pe0; pq_e2 += pe2 - pe1; pq_e3 += PQ_NOW() - pe2;
    }
    pq_epi += PQ_NOW() - pq3;
    ++pq_tiles;
  }
  if (p.probe && tid == 0) {
    unsigned long long* o = p.probe + (size_t)blockIdx.x * 16;
    o[0] = PQ_NOW() - pq_t0; o[1] = pq_wait; o[2] = pq_bar; o[3] = pq_comp; o[4] = pq_epi; o[5] = pq_tiles;
    o[6] = pq_e1; o[7] = pq_e2; o[8] = pq_e3;
  }
#undef PQ_NOW
  if (p.stats) {
    // one partial row per workgroup: [T][16 (+1)] -> 8 channel groups x 16 values, 32 threads each
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[tid * 17 + e] = s1[e];
      red[tid * 17 + 8 + e] = s2[e];
    }
    __syncthreads();
    if (tid < 128) {
      const int c = tid >> 4, e = tid & 15;            // channel group (lane % 8 == c), value
      float v = 0.f;
      for (int q = 0; q < 32; ++q) v += red[(q * 8 + c) * 17 + e];
      const int n = c * 8 + (e & 7);
      if (n < p.Nout) {
        if (p.stat_slices > 0) atomicAdd(&p.stats[((size_t)(blockIdx.x % p.stat_slices) * 2 + (e >> 3)) * p.Nout + n], v);
        else p.stats[((size_t)blockIdx.x * 2 + (e >> 3)) * p.Nout + n] = v;
      }
    }
  }
#endif   // __HIP_DEVICE_COMPILE__
}


// ------------------------------------------------------------------------------------------------------------------
// 64 -> 64 channels (ResNet layer1) with the WHOLE FILTER IN REGISTERS (round 3).  The persistent kernel above streams the
// filter through its 2-stage ring once per tile: 72 KB of weights beside a 47 KB window, 119 KB of LDS-DMA per 18.9 MFLOP
// tile -- at the CU's ~30 B/clk fill rate that is 4 K cycles per tile, as long as the tile's 4.6 K cycles of MFMA work --
// with a workgroup barrier per tap.  These convolutions are HBM-bound by their arithmetic (288 FLOP per byte of activation
// moved: 65 us per launch at batch 512), so everything beside the activation stream has to go: here a wave owns
// 128 pixels x 32 output channels, its B operand -- 9 taps x 4 k-steps x one 32 x 16 fragment -- is 144 VGPRs loaded once
// per kernel, the loop over the nine taps has NO barrier and NO weight traffic, and one LDS fragment read feeds one MFMA
// (1 KB per 32 cycles and SIMD: half the LDS rate).  Workgroup = 4 waves (2 pixel halves x 2 channel halves) on a
// 256-position tile, persistent over tiles, two workgroups per CU (<= 256 VGPRs); per tile two barriers (window landed /
// window free), the next window's DMA under the wave-private epilogue, raster decode from the table.
template <int V>
struct WinInt { static constexpr int value = V; };
template <int I, int N, class F>
__device__ __forceinline__ void win_unroll(F&& f) {      // f(WinInt<I>) ... f(WinInt<N-1>), indices as compile-time constants
  if constexpr (I < N) {
    f(WinInt<I>{});
    win_unroll<I + 1, N>(f);
  }
}
template <bool DGRAD, bool PROBE = false, bool ADD = false, int BNB = 0>
__global__ __launch_bounds__(256, 2) void conv_win_l1_kernel(const WinParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NW = 4, BM = 256;
  constexpr int WI_MAX = (BM + 2 * 58 + 7) / 8, WIW = (WI_MAX + NW - 1) / NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [window][4 waves x 2 KB staging][4 waves x 4 KB sums][8 KB: ninth tap]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int wrows8 = (p.wrows + 7) & ~7;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const uint32_t stage_lds = lds0 + wrows8 * 128 + wid * 2048;
  const int ntiles = (p.Gtot + BM - 1) / BM;
  const int frow = lane & 31, fh = lane >> 5;

  // ---- the filter: breg[tap][ks] = rows wn*32 + frow of the panel, channels tap*64 + ks*16 + fh*8 .. +7
  //      (taps 0..7: 128 VGPRs; the ninth tap's four fragments live in LDS -- 8 KB per workgroup, one extra ds_read_b128 per
  //       four MFMAs in 4 of the 36 steps: with all 144 in registers the compiler spills table values around the window DMA)
  bf16x8 breg[8][4];
  unsigned char* const wtap8 = smem + wrows8 * 128 + NW * 2048 + NW * 4096;      // [wn][ks][64 lanes x 16 B]
  // 1 KB behind it: the sink of DMA instructions past the window -- or, with a fused data-gradient epilogue (ADD / BNB), the
  // BatchNorm coefficient table [mean | invstd | scale | shift][64]; the sink is then wave 0's staging slice: that epilogue
  // issues the next window behind a barrier at its END (see there), when no staging slice is in use
  constexpr bool FUSED = ADD || BNB != 0;
  static_assert(!FUSED || DGRAD, "fused epilogues belong to the data gradient");
  unsigned char* const sink = FUSED ? smem + wrows8 * 128 : wtap8 + 8192;
  float* const coef = reinterpret_cast<float*>(wtap8 + 8192);
  if (BNB && tid < 64) {
    const int n = tid < p.Nout ? tid : 0;
    coef[tid] = p.bn_mean[n];
    coef[64 + tid] = p.bn_invstd[n];
    if (BNB == 2) {
      coef[128 + tid] = p.bn_scale[n];
      coef[192 + tid] = p.bn_shift[n];
    }
  }
  {
    const bf16_t* wrow = p.wpk + (size_t)(wn * 32 + frow) * p.Kgpad + fh * 8;
#pragma unroll
    for (int tap = 0; tap < 8; ++tap)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) breg[tap][ks] = *reinterpret_cast<const bf16x8*>(wrow + tap * 64 + ks * 16);
    if (wm == 0) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        *reinterpret_cast<bf16x8*>(wtap8 + (wn * 4 + ks) * 1024 + lane * 16) = *reinterpret_cast<const bf16x8*>(wrow + 8 * 64 + ks * 16);
    }
    __syncthreads();
  }
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, p.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc((void*)p.rtab, 0, p.rtab_bytes, 0x00020000);

  // ---- window DMA: instruction I = wid + 4k covers window rows 8I..8I+7 (lane -> row lane/8, physical chunk lane%8, bank
  //      swizzle on the source side: logical chunk = physical ^ ((row >> 1) & 7), constant per lane: 4I & 7 = 4 (wid & 1))
  const int lane_r4 = (lane >> 3) * 4;
  const uint32_t wcol = (uint32_t)((((lane & 7) ^ ((4 * wid + (lane >> 4)) & 7)) * 16) - 2 * p.C);      // (- one pixel: the table holds pixel + 1)
  uint32_t wt[WIW];      // table values, then source offsets, of this wave's window instructions
  auto load_window_tab = [&](int G0) {
#pragma unroll
    for (int k = 0; k < WIW; ++k)
      wt[k] = __builtin_amdgcn_raw_buffer_load_b32(rs_t, lane_r4, (G0 - p.halo + 8 * (wid + k * NW) + WIN_RASTER_MARGIN) * 4, 0);
  };
  // All table values are converted BEFORE the first DMA instruction goes out, and pinned there by an empty asm: converted one by
  // one in front of their DMA instructions (where the optimizer sinks them otherwise) the last conversions wait with
  // s_waitcnt vmcnt(0) -- the counter is in order -- for every DMA instruction already issued to LAND: 4.7 K cycles per tile.
  auto issue_window = [&]() {
#pragma unroll
    for (int k = 0; k < WIW; ++k) wt[k] = wt[k] ? __umul24(wt[k], (uint32_t)(2 * p.C)) + wcol : 0xFFFFFFF0u;
#pragma unroll
    for (int k = 0; k < WIW; ++k) asm volatile("" : "+v"(wt[k]));
    // (no branches: an instruction past the window's last row group reads nothing -- out-of-range offset -- into a 1 KB sink
    //  behind everything else; as wave-uniform branches the twelve instructions cost 24 taken branches per tile and wave)
#pragma unroll
    for (int k = 0; k < WIW; ++k) {
      const int I = wid + k * NW;
      const bool in = 8 * I < wrows8;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(in ? smem + I * 1024 : sink), 16,
                                               in ? wt[k] : 0xFFFFFFF0u, 0, 0, 0);
    }
  };

  // epilogue geometry: lane -> (row lane/4 + 16 it, 16-B chunk lane%4) of a 32 x 32 block; its 8 channels are fixed
  const int erow = lane >> 2, ech = lane & 3;
  const int ecol = wn * 32 + ech * 8;
  const bool col_ok = ecol < p.Nout;
  // BatchNorm partial sums live in LDS between tiles (wave-private [16][64] floats behind the staging slices):
  // sixteen registers that would otherwise be live through the loop
  // (accessed by inline asm like the staging slices: a compiler-visible LDS access waits for the window DMA in flight)
  float* const sums = reinterpret_cast<float*>(smem + wrows8 * 128 + NW * 2048) + wid * 1024 + lane;
  const uint32_t sums_lds = lds0 + wrows8 * 128 + NW * 2048 + (wid * 1024 + lane) * 4;
  if (p.stats) {
#pragma unroll
    for (int e = 0; e < 16; ++e) sums[e * 64] = 0.f;
  }

  // (timing probe, wave 0: cycles waiting for the window / in the MFMA loop / at the post-loop barrier / issuing the next
  //  window / in the epilogue; tiles)
  unsigned long long pq[6] = {0, 0, 0, 0, 0, 0};
#define L1_NOW() (PROBE ? __builtin_readcyclecounter() : 0ull)
  const unsigned long long pq_t0 = L1_NOW();
  const unsigned long long pq_w0 = PROBE ? wall_clock64() : 0ull;
  int tile = xcd_remap(blockIdx.x, gridDim.x);
  if (tile < ntiles) {
    load_window_tab(tile * BM);
    issue_window();
  }
  for (; tile < ntiles; tile += gridDim.x) {
    const int G0 = tile * BM;
    const bool more = tile + (int)gridDim.x < ntiles;
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const unsigned long long q0 = L1_NOW();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const unsigned long long q1 = L1_NOW();
    // (an opaque zero per tile: without it the compiler hoists the 36 fragment addresses of a tile out of the tile loop --
    //  36 VGPRs beside 144 of filter and 64 of accumulators)
    int zero;
    asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
    const int baseA = wm * 128 + frow + p.halo + zero;
    // fragment n = (tap * 4 + ks) * 4 + i feeds MFMA n; six fragment slots, fragment n + 5 is read just before MFMA n
    // issues (into the slot MFMA n - 1 has consumed): five MFMAs = 160 cycles of look-ahead on 24 registers -- a second
    // full set (32) does not fit beside 144 of filter and 64 of accumulators
    bf16x8 af[6];
    auto read_frag = [&](auto N_) {
      constexpr int n = decltype(N_)::value;
      constexpr int t = n / 4, i = n % 4, tap = t / 4, ks = t % 4, tr = tap / 3, ts = tap % 3;
      const int d = DGRAD ? (1 - tr) * p.Wp + (1 - ts) : (tr - 1) * p.Wp + (ts - 1);
      const int rowA = baseA + d;
      const int key = (rowA >> 1) & 7;                     // rows of the four m-tiles are 32 apart: same key
      af[n % 6] = *reinterpret_cast<const bf16x8*>(smem + rowA * 128 + (((2 * ks + fh) ^ key) << 4) + i * 4096);
    };
    bf16x8 b8;      // the ninth tap's fragment of the current k-step (from LDS)
    auto mfma_frag = [&](auto N_) {
      constexpr int n = decltype(N_)::value;
      constexpr int t = n / 4, i = n % 4;
      if constexpr (t / 4 == 8) {
        if constexpr (i == 0) b8 = *reinterpret_cast<const bf16x8*>(wtap8 + (wn * 4 + t % 4) * 1024 + lane * 16);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b8, af[n % 6], acc[i], 0, 0, 0);
      } else {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(breg[t / 4][t % 4], af[n % 6], acc[i], 0, 0, 0);
      }
    };
    read_frag(WinInt<0>{}); read_frag(WinInt<1>{}); read_frag(WinInt<2>{}); read_frag(WinInt<3>{}); read_frag(WinInt<4>{});
    win_unroll<0, 144>([&](auto N_) {
      constexpr int n = decltype(N_)::value;
      if constexpr (n + 5 < 144) read_frag(WinInt<n + 5>{});
      __builtin_amdgcn_sched_barrier(0);
      mfma_frag(N_);
      __builtin_amdgcn_sched_barrier(0);
    });
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    unsigned long long q2 = 0, q3 = 0, q4 = 0;
    if constexpr (!FUSED) {
    // every table value of the epilogue (output pixels) and of the next window in ONE burst, issued before the barrier (its
    // wait hides their latency) and consumed -- pinned by an empty asm -- before the window DMA goes out: vmcnt counts in
    // order, so a load issued behind the DMA burst cannot be used before the whole window has landed
    uint32_t ot[8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
      ot[q] = __builtin_amdgcn_raw_buffer_load_b32(rs_t, erow * 4, (G0 + wm * 128 + q * 16 + WIN_RASTER_MARGIN) * 4, 0);
    if (more) load_window_tab(G0 + (int)gridDim.x * BM);
    // the tile rounded to bf16: 32 registers instead of 64 while the table values are live (a spilled table value comes back
    // through a scratch load in front of its DMA instruction -- and waits for the previous one to land)
    uint32_t pkacc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        pkacc[i][2 * g] = pack_bf16x2(acc[i][4 * g], acc[i][4 * g + 1]);
        pkacc[i][2 * g + 1] = pack_bf16x2(acc[i][4 * g + 2], acc[i][4 * g + 3]);
      }
    // every wave is done with this tile's window: the next tile's goes out now and lands underneath the epilogue
    q2 = L1_NOW();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    q3 = L1_NOW();
#pragma unroll
    for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(ot[q]));
    if (more) issue_window();
    q4 = L1_NOW();
    // ---- epilogue, wave-private: a 32-pixel x 32-channel block at a time through this wave's 2 KB of LDS (inline-asm LDS
    //      accesses: see above)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2_t v;
        v.x = pkacc[i][2 * g];
        v.y = pkacc[i][2 * g + 1];
        win_lds_write_b64(stage_lds + frow * 64 + ((g ^ ((frow >> 2) & 3)) << 4) + fh * 8, v);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // wave-private: no barrier
      u32x4_t pkv[2];
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int r = erow + 16 * it;
        pkv[it] = win_lds_read_b128(stage_lds + r * 64 + ((ech ^ ((r >> 2) & 3)) << 4));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pkv[0]), "+v"(pkv[1])::"memory");
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const uint32_t t = ot[2 * i + it];
        if (t && col_ok) {
          const uint4 pk = make_uint4(pkv[it].x, pkv[it].y, pkv[it].z, pkv[it].w);
          *reinterpret_cast<uint4*>(p.dst + (size_t)(t - 1) * p.Nout + ecol) = pk;
          if (p.stats) {
            float q[8];
            unpack8(pk, q);
#pragma unroll
            for (int e = 0; e < 8; ++e) { s1[e] += q[e]; s2[e] += q[e] * q[e]; }
          }
        }
      }
    }
    } else {
      // ---- fused data-gradient epilogue (skip-connection add of the UNROUNDED tile; ReLU mask + BatchNorm-backward sums of the
      //      layer that produced the conv's input: conv_win_kernel's ADD / BNB, the same arithmetic element by element).  Its
      //      operands -- residual, BatchNorm input, mask source: up to three maps -- are global loads that must not queue
      //      behind the next window's DMA (vmcnt is in order), so here the window goes out at the END, behind a barrier; the
      //      other workgroup of the CU covers the wait.  Loads of m-tile i + 1 are in flight while m-tile i is worked on.
      q2 = q3 = q4 = L1_NOW();
      uint32_t ot[8];
#pragma unroll
      for (int q = 0; q < 8; ++q)
        ot[q] = __builtin_amdgcn_raw_buffer_load_b32(rs_t, erow * 4, (G0 + wm * 128 + q * 16 + WIN_RASTER_MARGIN) * 4, 0);
      // (no skip add: the tile is rounded right away -- 32 registers instead of 64 beside the operands in flight)
      uint32_t pkacc[ADD ? 1 : 4][8];
      if constexpr (!ADD) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            pkacc[i][2 * g] = pack_bf16x2(acc[i][4 * g], acc[i][4 * g + 1]);
            pkacc[i][2 * g + 1] = pack_bf16x2(acc[i][4 * g + 2], acc[i][4 * g + 3]);
          }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int g = 0; g < 8; ++g) asm volatile("" : "+v"(pkacc[i][g]));
      }
      // (the residual is consumed first and single-buffered: its next pair goes out as soon as this one is in the accumulator)
      uint4 res[2], bx[2][2], my[2][2];      // [buffer][row of the pair]
      auto issue_res = [&](auto I_) {
        constexpr int i = decltype(I_)::value;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const uint32_t t = ot[2 * i + it];
          res[it] = *reinterpret_cast<const uint4*>(p.add + (size_t)(t ? t - 1 : 0u) * p.Nout + (col_ok ? ecol : 0));
        }
      };
      auto issue_loads = [&](auto BUF_, auto I_) {
        constexpr int buf = decltype(BUF_)::value, i = decltype(I_)::value;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const uint32_t t = ot[2 * i + it];
          const size_t o = (size_t)(t ? t - 1 : 0u) * p.Nout + (col_ok ? ecol : 0);
          if constexpr (BNB != 0) bx[buf][it] = *reinterpret_cast<const uint4*>(p.bn_x + o);
          if constexpr (BNB == 1) my[buf][it] = *reinterpret_cast<const uint4*>(p.mask_y + o);
        }
      };
      if constexpr (ADD) issue_res(WinInt<0>{});
      issue_loads(WinInt<0>{}, WinInt<0>{});
      win_unroll<0, 4>([&](auto I_) {
        constexpr int i = decltype(I_)::value, buf = i & 1;
        if constexpr (i + 1 < 4) issue_loads(WinInt<(i + 1) & 1>{}, WinInt<i + 1>{});
        if constexpr (ADD) {
          // the residual's two rows (this lane's 8 channels each) -> slice -> back in the accumulator's layout (lane = pixel,
          // four consecutive channels per register group), added to the unrounded tile
#pragma unroll
          for (int it = 0; it < 2; ++it) {
            const int r = erow + 16 * it;
            u32x4_t v;
            v.x = res[it].x; v.y = res[it].y; v.z = res[it].z; v.w = res[it].w;
            win_lds_write_b128(stage_lds + r * 64 + ((ech ^ ((r >> 2) & 3)) << 4), v);
          }
          if constexpr (i + 1 < 4) issue_res(WinInt<i + 1>{});
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          u32x2_t rv[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) rv[g] = win_lds_read_b64(stage_lds + frow * 64 + ((g ^ ((frow >> 2) & 3)) << 4) + fh * 8);
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rv[0]), "+v"(rv[1]), "+v"(rv[2]), "+v"(rv[3])::"memory");
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            acc[i][4 * g] += bf16lo(rv[g].x); acc[i][4 * g + 1] += bf16hi(rv[g].x);
            acc[i][4 * g + 2] += bf16lo(rv[g].y); acc[i][4 * g + 3] += bf16hi(rv[g].y);
          }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          u32x2_t v;
          if constexpr (ADD) {
            v.x = pack_bf16x2(acc[i][4 * g], acc[i][4 * g + 1]);
            v.y = pack_bf16x2(acc[i][4 * g + 2], acc[i][4 * g + 3]);
          } else {
            v.x = pkacc[i][2 * g];
            v.y = pkacc[i][2 * g + 1];
          }
          win_lds_write_b64(stage_lds + frow * 64 + ((g ^ ((frow >> 2) & 3)) << 4) + fh * 8, v);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        u32x4_t pkv[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int r = erow + 16 * it;
          pkv[it] = win_lds_read_b128(stage_lds + r * 64 + ((ech ^ ((r >> 2) & 3)) << 4));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pkv[0]), "+v"(pkv[1])::"memory");
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const uint32_t t = ot[2 * i + it];
          if (t && col_ok) {
            uint4 pk = make_uint4(pkv[it].x, pkv[it].y, pkv[it].z, pkv[it].w);
            if constexpr (BNB != 0) {
              // dz = mask * q (q: the bf16-rounded gradient values), sums of dz and dz * xhat (conv_win_kernel: bnb_group)
              float q[8], xf[8];
              unpack8(pk, q);
              unpack8(bx[buf][it], xf);
              if constexpr (BNB == 1) {
                float ym[8];
                unpack8(my[buf][it], ym);
#pragma unroll
                for (int e = 0; e < 8; ++e) q[e] = ym[e] > 0.f ? q[e] : 0.f;
              } else {
                const float4 c0 = *reinterpret_cast<const float4*>(coef + 128 + ecol), c1 = *reinterpret_cast<const float4*>(coef + 128 + ecol + 4);
                const float4 h0 = *reinterpret_cast<const float4*>(coef + 192 + ecol), h1 = *reinterpret_cast<const float4*>(coef + 192 + ecol + 4);
                const float sc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w}, sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) q[e] = fmaf(xf[e], sc[e], sh[e]) > 0.f ? q[e] : 0.f;
              }
              const float4 m0 = *reinterpret_cast<const float4*>(coef + ecol), m1 = *reinterpret_cast<const float4*>(coef + ecol + 4);
              const float4 i0 = *reinterpret_cast<const float4*>(coef + 64 + ecol), i1 = *reinterpret_cast<const float4*>(coef + 64 + ecol + 4);
              const float mu[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w}, is[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                s1[e] += q[e];
                s2[e] += q[e] * (xf[e] - mu[e]) * is[e];
              }
              pk = pack8(q);
            }
            *reinterpret_cast<uint4*>(p.dst + (size_t)(t - 1) * p.Nout + ecol) = pk;
          }
        }
      });
      // every wave is done with the window AND with its staging slice (wave 0's is the sink of the DMA below)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (more) {
        load_window_tab(G0 + (int)gridDim.x * BM);
        issue_window();
      }
    }
    if (p.stats) {      // lane-private addresses: plain read-modify-write (in two halves: registers)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float old[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
          asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(old[e]) : "v"(sums_lds + h * 2048), "n"(e * 256) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(old[0]), "+v"(old[1]), "+v"(old[2]), "+v"(old[3]), "+v"(old[4]), "+v"(old[5]), "+v"(old[6]), "+v"(old[7])
                     :: "memory");
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float a = old[e] + (h == 0 ? s1[e] : s2[e]);
          asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(sums_lds + h * 2048), "v"(a), "n"(e * 256) : "memory");
        }
      }
    }
    if (PROBE) {
      pq[0] += q1 - q0; pq[1] += q2 - q1; pq[2] += q3 - q2; pq[3] += q4 - q3; pq[4] += L1_NOW() - q4; pq[5] += 1;
    }
  }
  if (PROBE && p.probe && tid == 0) {
    unsigned long long* o = p.probe + (size_t)blockIdx.x * 16;
    o[0] = L1_NOW() - pq_t0; o[1] = pq[0]; o[2] = pq[1]; o[3] = pq[2]; o[4] = pq[3]; o[5] = pq[5]; o[6] = pq[4];
    o[7] = pq_w0; o[8] = wall_clock64();      // 100 MHz
    o[9] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
  }
#undef L1_NOW
  if (p.stats) {
    // one partial row per workgroup: channel group c = wn * 4 + ech lives in waves {wn, wn + 2}, lanes with lane % 4 == ech
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < 128) {
      const int c = tid >> 4, e = tid & 15;
      const float* base = reinterpret_cast<const float*>(smem + wrows8 * 128 + NW * 2048);
      float v = 0.f;
      for (int w2 = 0; w2 < 2; ++w2)
        for (int q = 0; q < 16; ++q) v += base[(w2 * 2 + (c >> 2)) * 1024 + e * 64 + q * 4 + (c & 3)];
      const int n = c * 8 + (e & 7);
      if (n < p.Nout) {
        if (p.stat_slices > 0) atomicAdd(&p.stats[((size_t)(blockIdx.x % p.stat_slices) * 2 + (e >> 3)) * p.Nout + n], v);
        else p.stats[((size_t)blockIdx.x * 2 + (e >> 3)) * p.Nout + n] = v;
      }
    }
  }
#endif   // __HIP_DEVICE_COMPILE__
}

extern "C" const uint32_t* mpr_raster_table(int B, int H, int W, long long* entries);   // runtime.cpp
static unsigned long long* g_win_probe = nullptr;
extern "C" int mpr_conv_debug_probe(void* buf) {   // 8 x uint64 per workgroup of the next window-kernel launches
  g_win_probe = (unsigned long long*)buf;
  return 0;
}
static int g_win_on = 1;
// measured best: 256 x 128 tile on 8 waves / 2-deep weight ring (N > 64), 256 x 64 / 4-deep (N <= 64); bit 9 set: the fused
// data-gradient epilogues of the 64 -> 64 kernel are OFF by default -- alone they are 27-46 us faster per launch than
// conv_win_kernel's (bit-identical results), inside the step the three launches that take them change nothing (9.86 vs 9.81,
// 10.03 vs 9.99 ms in two in-process A/Bs): kept, tested, not the default
static int g_win_variant = 5 | 512 | 1024;
extern "C" int mpr_conv_set_window_variant(int v) {
  const int old = g_win_variant;
  g_win_variant = v;
  return old;
}
extern "C" int mpr_conv_set_window(int on) {   // tuning / test knob; returns the previous setting
  const int old = g_win_on;
  g_win_on = on;
  return old;
}

struct WinBnb {   // BatchNorm-backward fusion of a data gradient (see WinParams), shared with conv_igemm.hip
  int mask_mode;
  const void *mask_y, *bn_x;
  const float *mean, *invstd, *scale, *shift;
  float* slices;
  int nslices, prezeroed;
};

extern "C" {   // (internal to the library: declared in conv_igemm.hip, not in the public header)

// Is (geometry, size) served by the shifted-window kernel?  Shared by the launchers and the stat-row query.
bool mpr_win_eligible(long long M, int H, int W, int srcC, int Nout, int R, int S, int sh, int sw, int ph, int pw,
                      long long min_rows) {
  return g_win_on && R == 3 && S == 3 && sh == 1 && sw == 1 && ph == 1 && pw == 1 && srcC % 64 == 0 && Nout % 8 == 0 &&
         W >= 2 && W <= 56 && H >= 2 && M >= min_rows && (long long)(M / (H * W)) * (H + 1) * (W + 1) < (1ll << 30);
}

// N <= 64 forward: persistent kernel (variant bit 6 switches it off, bit 7 extends it to plain data gradients).
// Measured at batch 512, 64 -> 64 @ 56x56: forward 192 us vs 204 us, data gradient 215 us vs 190 us -- forward only.
static inline bool win_persistent(bool dgrad, int Nout, const void* add) {
  return !(g_win_variant & 64) && Nout <= 64 && add == nullptr && (!dgrad || (g_win_variant & 128));
}

// Workgroups of the persistent filter-in-registers kernel: 512 are resident at once (2 per CU), 12.7 tiles each at batch 512.
// (More, shorter-lived workgroups -- so that a launch which starts beside another stream's kernel could rebalance -- were
//  measured inside the training step and lose: 9.96 / 10.03 / 10.15 ms per step at 512 / 1024 / 2048; the startup of a
//  workgroup, 32 filter loads + one exposed window, costs more than the rebalancing returns.)
static int g_win_l1_grid = 512;
extern "C" int mpr_conv_set_l1_grid(int workgroups) {
  const int old = g_win_l1_grid;
  if (workgroups > 0) g_win_l1_grid = workgroups;
  return old;
}
// 64 -> <= 64 channels, no fused epilogue: the filter-in-registers kernel (variant bit 8 switches it off; comparisons)
// (bit 9: the fused data-gradient epilogues -- skip add, BatchNorm-backward sums -- stay on conv_win_kernel)
static inline bool win_l1(int srcC, int Nout, bool fused) {
  return !(g_win_variant & 256) && srcC == 64 && Nout <= 64 && (!fused || !(g_win_variant & 512));
}

// rows of the BatchNorm partial-sum buffer the forward launch will write
int mpr_conv_stat_slices();
bool mpr_conv_take_prezeroed();
int mpr_win_stat_rows(int B, int H, int W, int Nout) {
  if (mpr_conv_stat_slices() > 0) return mpr_conv_stat_slices();
  const int tiles = ceil_div(B * (H + 1) * (W + 1), 256);
  if (win_l1(64, Nout, false)) return tiles < g_win_l1_grid ? tiles : g_win_l1_grid;
  return win_persistent(false, Nout, nullptr) ? (tiles < 512 ? tiles : 512) : tiles;
}

// src [B,H,W,srcC] (*) panel [Npad128][9*srcC] -> dst [B,H,W,Nout]  (dgrad: mirrored tap shifts)
int mpr_win_launch(bool dgrad, const void* src, const void* wpk, void* dst, const void* add, float* stats, int B, int H,
                   int W, int srcC, int Nout, hipStream_t st, const WinBnb* bnb) {
  WinParams p;
  p.src = (const bf16_t*)src; p.wpk = (const bf16_t*)wpk; p.dst = (bf16_t*)dst; p.add = (const bf16_t*)add;
  p.stats = stats;
  p.stat_slices = (stats && !dgrad) ? mpr_conv_stat_slices() : 0;
  p.mask_mode = 0; p.mask_y = nullptr; p.bn_x = nullptr;
  p.bn_mean = p.bn_invstd = p.bn_scale = p.bn_shift = nullptr;
  if (bnb) {
    // data gradient with the BatchNorm-backward reduction fused in: the sums go to `slices` rows (fp32 atomics)
    p.stats = bnb->slices; p.stat_slices = bnb->nslices;
    p.mask_mode = bnb->mask_mode; p.mask_y = (const bf16_t*)bnb->mask_y; p.bn_x = (const bf16_t*)bnb->bn_x;
    p.bn_mean = bnb->mean; p.bn_invstd = bnb->invstd; p.bn_scale = bnb->scale; p.bn_shift = bnb->shift;
    if (!bnb->prezeroed) MPR_HIP(hipMemsetAsync(p.stats, 0, sizeof(float) * 2 * (size_t)p.stat_slices * Nout, st));
  } else {
    const bool prezeroed = mpr_conv_take_prezeroed();
    if (p.stat_slices > 0 && !prezeroed)
      MPR_HIP(hipMemsetAsync(stats, 0, sizeof(float) * 2 * (size_t)p.stat_slices * Nout, st));
  }
  p.H = H; p.W = W; p.C = srcC; p.Nout = Nout;
  p.Wp = W + 1; p.img = (H + 1) * (W + 1); p.halo = W + 2; p.Gtot = B * p.img; p.wrows = 256 + 2 * p.halo;
  p.Kgpad = 9 * srcC; p.ncb = srcC / 64;
  p.src_bytes = (unsigned)((size_t)B * H * W * srcC * 2);
  p.wpk_bytes = (unsigned)((size_t)((Nout + 127) / 128 * 128) * p.Kgpad * 2);
  p.div_img = make_fastdiv(p.img); p.div_wp = make_fastdiv(p.Wp);
  p.probe = g_win_probe;
  // variants: 0 = 64 output channels per workgroup with a 4-deep weight ring (8 KB stages); 1 / 2 = the same with a
  // 2- / 3-deep ring; 3 / 4 = 128 channels (256 x 128 tile, 4 waves of 128 x 64) with a 3- / 2-deep ring of 16 KB
  // stages; 5 (default) / 6 = 256 x 128 on 8 waves of 64 x 64 with a 2- / 3-deep ring.  N <= 64 always takes the
  // 256 x 64 tile (4 waves of 64 x 64; 4-deep ring unless variant 1 / 2).
  const int BN = ((g_win_variant & 15) >= 3 && Nout > 64) ? 128 : 64;
  p.ntn = ceil_div(Nout, BN);
  const int tiles_m = ceil_div(p.Gtot, 256);
  const size_t wbytes = (size_t)((p.wrows + 7) / 8 * 8) * 128;
  // (skip add + BatchNorm-backward sums together stay on conv_win_kernel: beside 128 registers of filter and the unrounded
  //  tile, three operand streams in flight spill 22-36 registers)
  if (win_l1(srcC, Nout, bnb != nullptr || add != nullptr) && (dgrad || (!bnb && !add)) && !(bnb && add)) {
    long long rt_entries = 0;
    p.rtab = mpr_raster_table(B, H, W, &rt_entries);
    MPR_REQUIRE(p.rtab != nullptr, "conv (window, 64 channels): raster table allocation failed");
    p.rtab_bytes = (unsigned)(rt_entries * 4);
    p.ntn = 1;
    const size_t lds = wbytes + 4 * 2048 + 4 * 4096 + 8192 + 1024;      // window + staging + partial sums + the ninth tap + sink / coefficients
    const int grid_p = tiles_m < g_win_l1_grid ? tiles_m : g_win_l1_grid;
#define MPR_L1(DG_, PR_, ADD_, BNB_)                                                                                   \
  do {                                                                                                                 \
    static bool attr_set = false;                                                                                      \
    if (!attr_set) {                                                                                                   \
      hipFuncSetAttribute((const void*)conv_win_l1_kernel<DG_, PR_, ADD_, BNB_>,                                       \
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                     \
      attr_set = true;                                                                                                 \
    }                                                                                                                  \
    conv_win_l1_kernel<DG_, PR_, ADD_, BNB_><<<grid_p, 256, lds, st>>>(p);                                             \
  } while (0)
    if (bnb || add) {
      const int mm = bnb ? p.mask_mode : 0;
      if (add) { if (mm == 1) MPR_L1(true, false, true, 1); else if (mm == 2) MPR_L1(true, false, true, 2); else MPR_L1(true, false, true, 0); }
      else     { if (mm == 1) MPR_L1(true, false, false, 1); else MPR_L1(true, false, false, 2); }
    } else if (p.probe) { if (dgrad) MPR_L1(true, true, false, 0); else MPR_L1(false, true, false, 0); }
    else { if (dgrad) MPR_L1(true, false, false, 0); else MPR_L1(false, false, false, 0); }
#undef MPR_L1
    MPR_LAUNCH_CHECK("conv_win_l1_kernel");
    return MPR_OK;
  }
  if (!bnb && win_persistent(dgrad, Nout, add)) {
    p.ntn = 1;
    const size_t lds = wbytes + 2 * 8192 + 4 * 4096;
    const int grid_p = tiles_m < 512 ? tiles_m : 512;
    static bool attr_set[2] = {false, false};
    if (dgrad) {
      if (!attr_set[1]) {
        hipFuncSetAttribute((const void*)conv_win_persist_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set[1] = true;
      }
      conv_win_persist_kernel<true><<<grid_p, 256, lds, st>>>(p);
    } else {
      if (!attr_set[0]) {
        hipFuncSetAttribute((const void*)conv_win_persist_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set[0] = true;
      }
      conv_win_persist_kernel<false><<<grid_p, 256, lds, st>>>(p);
    }
    MPR_LAUNCH_CHECK("conv_win_persist_kernel");
    return MPR_OK;
  }
  if (bnb) {
    // default tiles only: 256 x 64 on 4 waves / 4-deep ring (N <= 64), 256 x 128 on 8 waves / 2-deep ring
    const int BNt = Nout > 64 ? 128 : 64;
    p.ntn = ceil_div(Nout, BNt);
    dim3 gridb(tiles_m * p.ntn);
#define MPR_WINB(WM_, WN_, TM_, TN_, ST_, ADD_, MM_) do { if (g_win_variant & 1024) MPR_WINBM(WM_, WN_, TM_, TN_, ST_, ADD_, MM_, true); else MPR_WINBM(WM_, WN_, TM_, TN_, ST_, ADD_, MM_, false); } while (0)
#define MPR_WINBM(WM_, WN_, TM_, TN_, ST_, ADD_, MM_, M16_)                                                \
  do {                                                                                                     \
    static bool attr_set = false;                                                                          \
    if (!attr_set) {                                                                                       \
      hipFuncSetAttribute((const void*)conv_win_kernel<WM_, WN_, TM_, TN_, ST_, true, ADD_, MM_, M16_>,    \
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                         \
      attr_set = true;                                                                                     \
    }                                                                                                      \
    const size_t ring_ = wbytes + (size_t)ST_ * (WN_ * TN_ * 32) * 128;                                    \
    const size_t epi_ = !ADD_ ? 0 : (size_t)((WN_ * TN_ * 32) == 64 ? 256 : 64) * ((WN_ * TN_ * 32) * 4 + 16); \
    /* coefficient table: behind what the epilogue stages (<= 36.9 KB without the skip add, 69.6 KB with it at 64    \
       channels, 33.8 KB at 128), inside the window + ring allocation whenever that is large enough: two workgroups  \
       of the 64-channel tile at W = 56 fill the CU's 160 KB to within 2 KB */                                       \
    const size_t tab_ = (size_t)(((ADD_ && (WN_ * TN_ * 32) == 64) || (!ADD_ && (WN_ * TN_ * 32) == 128)) ? 72 : 40) * 1024; \
    const size_t need_ = tab_ + 4 * (WN_ * TN_ * 32) * sizeof(float);                                      \
    size_t body_ = ring_ > epi_ ? ring_ : epi_;                                                            \
    if (body_ < need_) body_ = need_;                                                                      \
    p.coef_off = (int)tab_;                                                                                \
    conv_win_kernel<WM_, WN_, TM_, TN_, ST_, true, ADD_, MM_, M16_><<<gridb, 64 * WM_ * WN_, body_, st>>>(p); \
  } while (0)
#define MPR_WINB2(WM_, WN_, TM_, TN_, ST_, ADD_) do { if (p.mask_mode == 1) MPR_WINB(WM_, WN_, TM_, TN_, ST_, ADD_, 1); else MPR_WINB(WM_, WN_, TM_, TN_, ST_, ADD_, 2); } while (0)
    if (BNt == 64) { if (add) MPR_WINB2(4, 1, 2, 2, 4, true); else MPR_WINB2(4, 1, 2, 2, 4, false); }
    else           { if (add) MPR_WINB2(4, 2, 2, 2, 2, true); else MPR_WINB2(4, 2, 2, 2, 2, false); }
#undef MPR_WINB2
#undef MPR_WINB
#undef MPR_WINBM
    MPR_LAUNCH_CHECK("conv_win_kernel (BatchNorm-backward fusion)");
    return MPR_OK;
  }
  const size_t lds_pad = (g_win_variant & 16) ? 40 * 1024 : 0;   // experiment: force one workgroup per CU
  const int g_win_variant_ = g_win_variant & 15;
  dim3 grid(tiles_m * p.ntn);
#define MPR_WINM(WM_, WN_, TM_, TN_, ST_, DG_, M16_)                                                        \
  do {                                                                                                     \
    static bool attr_set = false;                                                                          \
    if (!attr_set) {                                                                                       \
      hipFuncSetAttribute((const void*)conv_win_kernel<WM_, WN_, TM_, TN_, ST_, DG_, false, 0, M16_>,      \
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                         \
      attr_set = true;                                                                                     \
    }                                                                                                      \
    const size_t lds = lds_pad + wbytes + (size_t)ST_ * (WN_ * TN_ * 32) * 128;                            \
    conv_win_kernel<WM_, WN_, TM_, TN_, ST_, DG_, false, 0, M16_><<<grid, 64 * WM_ * WN_, lds, st>>>(p);   \
  } while (0)
#define MPR_WINAM(WM_, WN_, TM_, TN_, ST_, M16_)                                                           \
  do {                                                                                                     \
    static bool attr_set = false;                                                                          \
    if (!attr_set) {                                                                                       \
      hipFuncSetAttribute((const void*)conv_win_kernel<WM_, WN_, TM_, TN_, ST_, true, true, 0, M16_>,      \
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                         \
      attr_set = true;                                                                                     \
    }                                                                                                      \
    const size_t ring_ = lds_pad + wbytes + (size_t)ST_ * (WN_ * TN_ * 32) * 128;                          \
    const size_t epi_ = (size_t)((WN_ * TN_ * 32) == 64 ? 256 : 64) * ((WN_ * TN_ * 32) * 4 + 16);         \
    conv_win_kernel<WM_, WN_, TM_, TN_, ST_, true, true, 0, M16_><<<grid, 64 * WM_ * WN_, ring_ > epi_ ? ring_ : epi_, st>>>(p); \
  } while (0)
#define MPR_WIN(WM_, WN_, TM_, TN_, ST_, DG_) MPR_WINM(WM_, WN_, TM_, TN_, ST_, DG_, false)
#define MPR_WINA(WM_, WN_, TM_, TN_, ST_) MPR_WINAM(WM_, WN_, TM_, TN_, ST_, false)
#define MPR_WIN2(WM_, WN_, TM_, TN_, ST_) do { if (dgrad && add) MPR_WINA(WM_, WN_, TM_, TN_, ST_); else if (dgrad) MPR_WIN(WM_, WN_, TM_, TN_, ST_, true); else MPR_WIN(WM_, WN_, TM_, TN_, ST_, false); } while (0)
  // (variant bit 10: the default tiles as v_mfma_f32_16x16x32_bf16)
#define MPR_WIN216(WM_, WN_, TM_, TN_, ST_) do { if (dgrad && add) MPR_WINAM(WM_, WN_, TM_, TN_, ST_, true); else if (dgrad) MPR_WINM(WM_, WN_, TM_, TN_, ST_, true, true); else MPR_WINM(WM_, WN_, TM_, TN_, ST_, false, true); } while (0)
  if (BN == 64) {
    switch (g_win_variant_) {
      case 1: MPR_WIN2(4, 1, 2, 2, 2); break;
      case 2: MPR_WIN2(4, 1, 2, 2, 3); break;
      default: if (g_win_variant & 1024) MPR_WIN216(4, 1, 2, 2, 4); else MPR_WIN2(4, 1, 2, 2, 4); break;
    }
  } else {
    switch (g_win_variant_) {
      case 3: MPR_WIN2(2, 2, 4, 2, 3); break;
      case 5: if (g_win_variant & 1024) MPR_WIN216(4, 2, 2, 2, 2); else MPR_WIN2(4, 2, 2, 2, 2); break;     // 8 waves of 64 x 64
      case 6: MPR_WIN2(4, 2, 2, 2, 3); break;
      default: MPR_WIN2(2, 2, 4, 2, 2); break;
    }
  }
#undef MPR_WIN216
#undef MPR_WIN2
#undef MPR_WINA
#undef MPR_WIN
#undef MPR_WINAM
#undef MPR_WINM
  MPR_LAUNCH_CHECK("conv_win_kernel");
  return MPR_OK;
}

}  // extern "C"
