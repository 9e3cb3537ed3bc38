// conv_wsplit.hip — 3x3 / stride 1 / dilation 1 convolution, 1-D Winograd F(2,3) along image rows,
// with the fp32 products carried out on the gfx950 BF16 matrix cores by a 3-way operand split.
//
// Why: on CDNA4 the fp32 MFMA runs at the vector rate (157 TF), the bf16 MFMA 16x faster.  Every
// fp32 value v is cut EXACTLY into three bf16 pieces v = h + m + l (8 significand bits each, by
// truncation: h = top 8 bits of v, m = top 8 bits of v-h, l = v-h-m), and
//     a*b = ah*bh + (ah*bm + am*bh) + (am*bm + ah*bl + al*bh)  [+ am*bl + al*bm + al*bl]
// is evaluated with SIX v_mfma_f32_32x32x16_bf16 (products exact, fp32 accumulation inside the
// matrix core) instead of eight fp32 MFMAs of the same shape-equivalent: 6 x 32 cycles against
// 16 x 32 for a 32x32x16 block.  The dropped terms are < 2^-21 |ab| in the worst case and 2^-25 |ab| rms
// (activations split by truncation: |am| < 2^-7, |al| < 2^-14; weights by round-to-nearest on the host:
// |bm| <= 2^-8, |bl| <= 2^-16), i.e. of the order of an fp32 ulp of the product (tests/test_split_arith_cpu.py); measured
// on hardware (scripts/probes/probe_bf16x3.hip, K = 4608 dot products of post-ReLU-like data):
// max error 1.56e-7 / rms 3.4e-8 of sum|ab| against 1.76e-7 / 2.9e-8 for the fp32 MFMA (== fmaf
// chain) — the same accuracy class, which tests/test_conv_gpu.py asserts against an fp64 oracle.
//
// 1-D Winograd F(2,3) along image rows (keras Conv2D 3x3 'same' + folded BN + ReLU,
// detection.py:87-103): for the output pair (x0, x0+1) of a row and every (ky, c)
//     d_i = in[y+ky-1][x0-1+i][c],   V = (d0-d2, d1+d2, d2-d1, d1-d3),   U = (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2)
//     M_xi[pair][o] = sum_{ky,c} V_xi U_xi,   out[x0] = M0+M1+M2,   out[x0+1] = M1-M2-M3.
//
// Block = 512 threads, persistent (one per CU, walking over tiles): waves 4-7 PRODUCE -- thread (pair,
// channel quad) loads the four raw pixels through raw buffer loads (out-of-range offset = zero padding),
// forms V in fp32, splits and writes bf16 operands to LDS in MFMA A-operand
// order As[buf][xi][piece][M-tile][k half][32 pairs][8 ch] (a 32x32x16 A fetch is two contiguous 512-B
// runs, bank-conflict free) -- and waves 0-3 CONSUME: wave (wm, wn) owns 64 pairs x 32 output channels
// for all four points (128 accumulator VGPRs), so the inverse transform is register-only.  Tile = 128 px x
// 128 couts (<1,4>) or 256 px x 64 couts (<2,2>); K-step = one (16-channel group, ky) = 48 MFMAs per
// consumer wave.  U is pre-transformed and pre-split on the host and packed in B-operand order
// [16-ch group][ky][32-cout tile][xi][piece][lane][8], fetched straight into registers with a rolling
// per-point prefetch; weights never touch LDS.  Requires W even.  POOL variant: tile = 2 image rows x
// 64*WM columns with the 2x2 max taken in-lane.  Details at the kernel below; measurements in DESIGN.md.
#include "split_common.h"
#include <cmath>
#include <algorithm>

struct WsParams {
  const float* in;
  const unsigned short* wgt;  // [Cin/16][3][Cout_pad/32][4 xi][3 pieces][64 lanes][8] bf16
  float* out;
  const float* pre_a;
  const float* pre_b;
  const float* post_a;
  const float* post_b;
  int H, W, Cin, in_cs, in_co;
  int Cout, Cout_pad, out_cs, out_co;
  int relu;
  int nsteps;  // 3 * Cin / 16
  int Mtotal;
  float* pool_out;
  int pool_cs, pool_co, write_full, tiles_per_row;
  int total_tiles;
  const unsigned* amax_in;  // HALF kernels: tracked max |input| (Tensor::amax), never null there
  int w_exp;                // HALF kernels: the weights are stored multiplied by 2^w_exp
  unsigned* amax_out;   // Tensor::amax of the output (and of the pooled output), or nullptr
  unsigned* amax_pool;
#ifdef KOCR_DEV_SWITCHES
  int dbg;  // developer timing experiments (wrong results): 1 = no stores, 2 = one K-step's weights
#endif
};

// Tile geometry shared by both roles.  POOL == 0: 128*WM consecutive pixels of the flattened (n, y, x)
// order.  POOL == 1: 2 image rows x 64*WM columns.
struct WsTile {
  long pm0;      // flattened index of the tile's first pixel
  int y0t, x0t;  // POOL: top row / left column
  int nt;        // output-channel tile
};

template <int POOL, int WM, int WN>
__device__ __forceinline__ WsTile ws_tile(const WsParams& p, int L, int total, int nblk_n) {
  WsTile t;
  const int tile = kocr_xcd_remap(L, total);
  const int mt = tile / nblk_n;
  t.nt = tile - mt * nblk_n;
  t.y0t = t.x0t = 0;
  if constexpr (POOL) {
    const int rp_lin = mt / p.tiles_per_row, cb = mt - rp_lin * p.tiles_per_row;
    const int hh = p.H >> 1;
    const int nimg = rp_lin / hh, rp = rp_lin - nimg * hh;
    t.y0t = 2 * rp;
    t.x0t = cb * 64 * WM;
    t.pm0 = ((long)nimg * p.H + t.y0t) * p.W + t.x0t;
  } else {
    t.pm0 = (long)mt * (128 * WM);
  }
  return t;
}

// Persistent, wave-specialised kernel: 512 threads = 4 consumer waves (MFMA + weight stream) and
// 4 producer waves (input gather, Winograd input transform, bf16x3 split, LDS fill); one block per CU
// walks over tiles L = blockIdx.x, blockIdx.x + gridDim.x, ...  The K-steps of consecutive tiles form
// ONE pipeline (LDS double buffer indexed by the global step parity), so a tile's first operands are
// already in LDS when the consumers finish the previous tile's epilogue.
//   WM x WN = 4 consumer waves: wave (wm, wn) owns M-tiles {2wm, 2wm+1} (64 pairs) x 32 couts.
//   <1,4>: tile 128 px x 128 couts      <2,2>: tile 256 px x 64 couts (weights shared by two waves)
//   KB = 16-channel blocks per K-step (1 or 2).  KB = 2 (fp16x2 mode, Cin % 32 == 0, W % 4 == 0): the
//   step is twice as long again (48 MFMAs, weights prefetched 1536 cycles ahead) and a producer thread
//   handles TWO adjacent pairs of one channel quad -- 6 raw pixel loads instead of 8.
template <int POOL, int WM, int WN, int HALF, int KB>
__global__ __launch_bounds__(512) void conv_ws_kernel(WsParams p) {
  constexpr int NP = HALF ? 2 : 3;                   // operand pieces: 2 x fp16 (3 products) or 3 x bf16 (6 products)
  constexpr int NMT = 2 * WM;                        // 32-pair M-tiles per block tile
  constexpr int IPT = WM;                            // gather items per producer thread
  constexpr int PPI = KB;                            // pairs per gather item
  constexpr int NPX = 2 * PPI + 2;                   // raw pixels per gather item
  constexpr int QPS = 4 * KB;                        // channel quads per K-step
  constexpr int KH_STRIDE = 256;                     // ushorts: 32 rows x 8 channels
  constexpr int PLANE = NMT * 2 * KH_STRIDE;         // one (xi, piece, k block) plane
  constexpr int BUF = 4 * NP * KB * PLANE;           // one K-step: 24 KB * WM (bf16x3) / 16 KB * WM * KB (fp16x2)
  constexpr int NPH = 4 * KB;                        // consumer phases (point, k block) per K-step
  // As[buf][xi][piece][M-tile][k half][32 rows x 8 ch bf16 = 512 B]: a 32x32x16 A fetch reads 2 x 512
  // contiguous bytes (16 B per lane, conflict-free); the rows of k half 1 are XOR-ed with 64 B so the two
  // k halves a 16-lane ds_write_b64 group touches fall into different halves of the 32 store banks.
  extern __shared__ __attribute__((aligned(16))) unsigned short As[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform by construction: keep it in an SGPR
  const int l31 = lane & 31, l5 = lane >> 5;
  const int nblk_n = p.Cout_pad / (32 * WN);
  const int total = p.total_tiles;
  const int ns = p.nsteps;
  const int G = gridDim.x;

  // ==================================================================================================
  // producer waves 4..7: raw pixels -> Winograd input transform -> bf16x3 split -> LDS
  // ==================================================================================================
  if (wave >= 4) {
    const int ptid = tid - 256;
    const int quad = ptid & (QPS - 1), q4 = quad & 3, kb = quad >> 2;
    constexpr unsigned OOB = 0x80000000u;
    const float in_scale = HALF ? kocr_pow2(kocr_scale_exp(p.amax_in, 12)) : 1.f;  // exact power of two
    (void)in_scale;
    // gather item it of this thread: PPI adjacent pairs starting at LDS row idx0, one channel quad
    int ldst[IPT][PPI];
#pragma unroll
    for (int it = 0; it < IPT; ++it)
#pragma unroll
      for (int pp = 0; pp < PPI; ++pp) {
        const int idx = ((ptid / QPS) + it * (256 / QPS)) * PPI + pp;
        ldst[it][pp] = ((idx >> 5) * 2 + (q4 >> 1)) * KH_STRIDE + ((((idx & 31) * 8) ^ ((q4 >> 1) * 32)) + (q4 & 1) * 4);
      }
    // position of the NEXT K-step to load: tile L_ld, step (ld_cg, ld_ky)
    int L_ld = blockIdx.x, ld_ky = 0, ld_cg = 0;
    unsigned goff[IPT][NPX];
    int gy[IPT];
    bool gok[IPT];
    __amdgpu_buffer_rsrc_t rsrc;
    // Raw buffer resource over the input, based one image row + one pixel BEFORE the tile so that every
    // byte offset is >= 0; an offset of 0x80000000 is out of range and the load returns 0, which is how
    // row/column zero padding and tiles past the end are expressed (no value masking, no 64-bit address
    // arithmetic in the loop).
    auto tile_geometry = [&]() __attribute__((always_inline)) {
      const WsTile t = ws_tile<POOL, WM, WN>(p, L_ld < total ? L_ld : 0, total, nblk_n);
      const float* bbase_v = p.in + (t.pm0 * p.in_cs + p.in_co) - (long)(p.W + 1) * p.in_cs;
      // block-uniform by construction; readfirstlane tells the compiler (no waterfall loop around the loads)
      const unsigned long long bb = (unsigned long long)bbase_v;
      const unsigned long long bbu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(bb >> 32)) << 32) |
                                     (unsigned)__builtin_amdgcn_readfirstlane((int)bb);
      rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)bbu, 0, 0x80000000, 0x00020000);
#pragma unroll
      for (int it = 0; it < IPT; ++it) {
        const int idx = ((ptid / QPS) + it * (256 / QPS)) * PPI;  // first LDS row (pair) of the item
        int rel;  // pixel offset of the item's first pair from pm0
        int x0;
        if constexpr (POOL) {
          const int i = idx & 31, row = i >> 4, pr = (idx >> 5) * 16 + (i & 15);
          rel = row * p.W + 2 * pr;
          x0 = t.x0t + 2 * pr;
          gy[it] = t.y0t + row;
          gok[it] = L_ld < total;
        } else {
          rel = 2 * idx;
          const long g = t.pm0 + rel;
          gok[it] = g < p.Mtotal && L_ld < total;
          x0 = (int)(g % p.W);
          gy[it] = (int)((g / p.W) % p.H);
        }
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
          // first / last raw pixel is column zero padding (KB = 2: W % 4 == 0, so the two pairs share a row)
          const bool pad = (i == 0 && x0 == 0) || (i == NPX - 1 && x0 + 2 * PPI >= p.W);
          goff[it][i] = pad ? OOB : (unsigned)(((rel + i) * p.in_cs + kb * 16 + q4 * 4) * 4);
        }
      }
    };
    auto load_raw = [&](v4f (&raw)[IPT][NPX]) __attribute__((always_inline)) {
      const int soff = (ld_ky * p.W * p.in_cs + ld_cg * 16 * KB) * 4;
#pragma unroll
      for (int it = 0; it < IPT; ++it) {
        const bool ok = gok[it] && (unsigned)(gy[it] + ld_ky - 1) < (unsigned)p.H;
#pragma unroll
        for (int i = 0; i < NPX; ++i)
          raw[it][i] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? goff[it][i] : OOB, soff, 0));
      }
      if (++ld_ky == 3) {
        ld_ky = 0;
        if (++ld_cg == p.Cin / (16 * KB)) {  // next tile
          ld_cg = 0;
          L_ld += G;
          tile_geometry();
        }
      }
    };
    auto produce = [&](const v4f (&raw)[IPT][NPX], int buf) __attribute__((always_inline)) {
      unsigned short* base = As + buf * BUF + kb * PLANE;
#pragma unroll
      for (int it = 0; it < IPT; ++it)
#pragma unroll
        for (int pp = 0; pp < PPI; ++pp) {
          const v4f d0 = raw[it][2 * pp], d1 = raw[it][2 * pp + 1], d2 = raw[it][2 * pp + 2], d3 = raw[it][2 * pp + 3];
          const v4f V[4] = {d0 - d2, d1 + d2, d2 - d1, d1 - d3};
#pragma unroll
          for (int xi = 0; xi < 4; ++xi) {
            unsigned short* dst = base + xi * NP * KB * PLANE + ldst[it][pp];
            if constexpr (HALF) {
              u2v h, l;
              kocr_split4_h(V[xi] * in_scale, h, l);
              *reinterpret_cast<u2v*>(dst) = h;
              *reinterpret_cast<u2v*>(dst + KB * PLANE) = l;
            } else {
              u2v h, m, l;
              kocr_split4(V[xi], h, m, l);
              *reinterpret_cast<u2v*>(dst) = h;
              *reinterpret_cast<u2v*>(dst + KB * PLANE) = m;
              *reinterpret_cast<u2v*>(dst + 2 * KB * PLANE) = l;
            }
          }
        }
    };
    // K-steps this block will run in total
    const int my_tiles = (total - (int)blockIdx.x + G - 1) / G;
    const int T = my_tiles * ns;
    tile_geometry();
    // branch-free two-step pipeline (a conditional load would force s_waitcnt vmcnt(0) at the join): the
    // loads of global step k+1 are always in flight while step k is transformed; past the end they are masked
    v4f rawA[IPT][NPX], rawB[IPT][NPX];
    load_raw(rawA);
    int k = 0;
    for (; k + 2 <= T; k += 2) {
      load_raw(rawB);
      produce(rawA, 0);
      __syncthreads();
      load_raw(rawA);
      produce(rawB, 1);
      __syncthreads();
    }
    if (k < T) {  // odd total: the last step is already in rawA
      produce(rawA, 0);
      __syncthreads();
    }
    __syncthreads();  // pairs with the consumers' barrier inside the last K-step
    return;
  }

  // ==================================================================================================
  // consumer waves 0..3: wave (wm, wn): 64 pairs x 32 output channels x 4 points; MFMA + weight stream
  // ==================================================================================================
  const int wn = (WN == 4) ? wave : (wave % WN), wm = (WM == 1) ? 0 : (wave / WN);
  const int ntiles32 = p.Cout_pad >> 5;
#ifdef KOCR_DEV_SWITCHES
  const size_t w_step = (p.dbg & 2) ? 0 : (size_t)ntiles32 * NPH * NP * 64 * 8;  // dbg 2: one step's weights
#else
  const size_t w_step = (size_t)ntiles32 * NPH * NP * 64 * 8;  // ushorts per K-step
#endif
  // weights: [step][ntile32][xi][k block][piece][lane][8]; 16 B per lane and (xi, k block, piece)
  auto w_tile = [&](int nt) { return p.wgt + ((size_t)(nt * WN + wn) * NPH * NP * 64 + lane) * 8; };

  bf8 bw[NPH][NP];   // [phase = xi * KB + k block][piece]
  f16v acc[4][2];    // [xi][own M-tile]
  const int a_lane = (wm * 2 * 2 + l5) * KH_STRIDE + ((l31 * 8) ^ (l5 * 32));
  // phase ph = (xi, k block): its operand planes are [(xi * NP + piece) * KB + k block]
  auto load_a = [&](bf8 (&a)[2][NP], const unsigned short* bufp, int ph) __attribute__((always_inline)) {
    const int xi = ph / KB, kblk = ph % KB;
    const unsigned short* base = bufp + (xi * NP * KB + kblk) * PLANE + a_lane;
    // fetch order = use order: the lowest piece of both M-tiles first (mfma6 starts with the smallest terms)
#pragma unroll
    for (int s = NP - 1; s >= 0; --s)
#pragma unroll
      for (int m = 0; m < 2; ++m) a[m][s] = *reinterpret_cast<const bf8*>(base + s * KB * PLANE + m * 2 * KH_STRIDE);
  };
  auto mfma6 = [&](const bf8 (&a)[2][NP], int ph) __attribute__((always_inline)) {
    const int xi = ph / KB;
    // smallest terms first; the two M-tiles alternate so consecutive MFMAs are independent
    if constexpr (HALF) {
      const hf8 b0 = __builtin_bit_cast(hf8, bw[ph][0]), b1 = __builtin_bit_cast(hf8, bw[ph][1]);
#pragma unroll
      for (int m = 0; m < 2; ++m)
        acc[xi][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf8, a[m][1]), b0, acc[xi][m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 2; ++m)
        acc[xi][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf8, a[m][0]), b1, acc[xi][m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 2; ++m)
        acc[xi][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf8, a[m][0]), b0, acc[xi][m], 0, 0, 0);
    } else {
      const bf8 b0 = bw[ph][0], b1 = bw[ph][1], b2 = bw[ph][NP - 1];
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[xi][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][NP - 1], b0, acc[xi][m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[xi][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][0], b2, acc[xi][m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[xi][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][1], b1, acc[xi][m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[xi][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][1], b0, acc[xi][m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[xi][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][0], b1, acc[xi][m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[xi][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][0], b0, acc[xi][m], 0, 0, 0);
    }
  };
  // One K-step = NPH phases (point, k block).  The A operands of phase ph+1 are fetched from LDS while
  // the MFMAs of phase ph run; a phase's weights are re-fetched (next K-step) as soon as its MFMAs are
  // issued.  The block barrier that publishes the NEXT K-step sits before the last phase's MFMAs (all
  // LDS reads of this step have landed by then), so the next step's first operands are fetched behind
  // those MFMAs instead of exposing the LDS latency after the barrier.  a0 carries phase 0 on entry.
  bf8 a0[2][NP], a1[2][NP];
  auto compute_step = [&](const unsigned short* bufp, const unsigned short* bufn,
                          const unsigned short* w_next) __attribute__((always_inline)) {
    auto load_b = [&](int ph) __attribute__((always_inline)) {
#pragma unroll
      for (int s = 0; s < NP; ++s) bw[ph][s] = *reinterpret_cast<const bf8*>(w_next + (size_t)(ph * NP + s) * 64 * 8);
    };
    // the fences pin the issue order: LDS fetch of the next phase, its MFMAs, weight fetch
    load_a(a1, bufp, 1);
#pragma unroll
    for (int q = 0; q < NPH / 2; ++q) {
      __builtin_amdgcn_sched_barrier(0);
      mfma6(a0, 2 * q);
      __builtin_amdgcn_sched_barrier(0);
      load_b(2 * q);
      if (2 * q + 2 < NPH) {
        load_a(a0, bufp, 2 * q + 2);
      } else {
        // unconditional (a branch here makes the compiler drain vmcnt at the loop head); the producers run
        // one extra barrier for the last K-step, whose prefetch reads stale but valid LDS
        __syncthreads();  // waits for this wave's LDS reads too: the current buffer is free, the next one is full
        load_a(a0, bufn, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma6(a1, 2 * q + 1);
      __builtin_amdgcn_sched_barrier(0);
      load_b(2 * q + 1);
      if (2 * q + 3 < NPH) load_a(a1, bufp, 2 * q + 3);
    }
  };

  int gs = 0;  // global K-step counter of this block (LDS buffer = gs & 1)
  {
    const WsTile t0 = ws_tile<POOL, WM, WN>(p, blockIdx.x, total, nblk_n);
    const unsigned short* w0 = w_tile(t0.nt);
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
      for (int s = 0; s < NP; ++s) bw[ph][s] = *reinterpret_cast<const bf8*>(w0 + (size_t)(ph * NP + s) * 64 * 8);
  }
  __syncthreads();  // global step 0 is in LDS
  load_a(a0, As, 0);
  for (int L = blockIdx.x; L < total; L += G) {
    const WsTile t = ws_tile<POOL, WM, WN>(p, L, total, nblk_n);
    const unsigned short* w_ptr = w_tile(t.nt);
    // the last K-step prefetches the first weights of this block's next tile (or re-reads its own)
    const unsigned short* w_after = (L + G < total) ? w_tile(ws_tile<POOL, WM, WN>(p, L + G, total, nblk_n).nt)
                                                    : w_ptr;
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][m][r] = 0.f;
    // drain the epilogue's stores here, once per tile: otherwise the compiler merges their unknown count
    // into the K-loop head and waits for vmcnt(0) -- i.e. for the weights just prefetched -- every K-step
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (int s = 0; s < ns; ++s, ++gs)
      compute_step(As + (gs & 1) * BUF, As + ((gs + 1) & 1) * BUF, s + 1 < ns ? w_ptr + (size_t)(s + 1) * w_step : w_after);

    // ---- epilogue: 32x32 C/D map: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) -------------
    // Two phases, no branches: (1) inverse transform + BN + ReLU IN PLACE in the accumulators, (2) all
    // stores back to back as raw buffer stores: one per-lane byte offset (never rewritten), the per-
    // register pixel offset in an SGPR, out-of-range pixels / padded couts dropped by the hardware range
    // check.  (With `if (pixel < M) o[..] = ..` per pair hipcc put `s_waitcnt vmcnt(0)` before every
    // store pair -- one full memory round trip each, 20-35 % of the kernel.)
    {
      const int n = (t.nt * WN + wn) * 32 + l31;
      const int nc = n < p.Cout ? n : p.Cout - 1;
      // HALF: the accumulators carry the exact factor 2^(input scale + weight scale); undo it in pre_a
      const float unscale = HALF ? kocr_pow2(-(kocr_scale_exp(p.amax_in, 12) + p.w_exp)) : 1.f;
      const float pa = p.pre_a[nc] * unscale, pb = p.pre_b[nc];
      const bool has_post = p.post_a != nullptr;
      const float qa = has_post ? p.post_a[nc] : 1.f, qb = has_post ? p.post_b[nc] : 0.f;
      auto finish = [&](float m0, float m1, float m2, float m3, float& o0, float& o1) {
        o0 = (m0 + m1) + m2;
        o1 = (m1 - m2) - m3;
        o0 = o0 * pa + pb;
        o1 = o1 * pa + pb;
        if (p.relu) {
          o0 = fmaxf(o0, 0.f);
          o1 = fmaxf(o1, 0.f);
        }
        if (has_post) {
          o0 = o0 * qa + qb;
          o1 = o1 * qa + qb;
        }
      };
      constexpr unsigned OOB = 0x80000000u;
#ifdef KOCR_DEV_SWITCHES
      const bool live = n < p.Cout && !(p.dbg & 1);
#else
      const bool live = n < p.Cout;
#endif
      auto uniform_rsrc = [&](const float* base, unsigned bytes) {
        const unsigned long long bb = (unsigned long long)base;
        const unsigned long long bbu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(bb >> 32)) << 32) |
                                       (unsigned)__builtin_amdgcn_readfirstlane((int)bb);
        return __builtin_amdgcn_make_buffer_rsrc((void*)bbu, 0, (int)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
      };
      const int ocs4 = p.out_cs * 4;
      if constexpr (POOL) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            float a0, a1, b0, b1;
            finish(acc[0][m][r], acc[1][m][r], acc[2][m][r], acc[3][m][r], a0, a1);                  // row y
            finish(acc[0][m][r + 8], acc[1][m][r + 8], acc[2][m][r + 8], acc[3][m][r + 8], b0, b1);  // row y+1
            acc[0][m][r] = a0;
            acc[1][m][r] = a1;
            acc[0][m][r + 8] = b0;
            acc[1][m][r + 8] = b1;
            acc[2][m][r] = fmaxf(fmaxf(a0, a1), fmaxf(b0, b1));
          }
        if (p.amax_out || p.amax_pool) {
          float mx = 0.f;
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fmaxf(fabsf(acc[0][m][r]), fabsf(acc[1][m][r])));
          mx = live ? mx : 0.f;
          if (p.amax_out) kocr_amax_update(p.amax_out, mx);
          if (p.amax_pool) kocr_amax_update(p.amax_pool, mx);
        }
        const long nimg = t.pm0 / ((long)p.H * p.W);
        const long pp0 = (nimg * (p.H >> 1) + (t.y0t >> 1)) * (p.W >> 1) + (t.x0t >> 1);
        if (p.write_full) {
          const __amdgpu_buffer_rsrc_t ro = uniform_rsrc(p.out + (t.pm0 * p.out_cs + p.out_co), 0x7FFFFFFFu);
          const unsigned vo = live ? (unsigned)((8 * l5 * p.out_cs + n) * 4) : OOB;
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              const int px = 2 * ((wm * 2 + m) * 16 + (r & 3) + 8 * (r >> 2));  // + 8*l5 in vo
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[0][m][r]), ro, vo, px * ocs4, 0);
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[1][m][r]), ro, vo, (px + 1) * ocs4, 0);
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[0][m][r + 8]), ro, vo, (px + p.W) * ocs4, 0);
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[1][m][r + 8]), ro, vo, (px + p.W + 1) * ocs4, 0);
            }
        }
        const __amdgpu_buffer_rsrc_t rp = uniform_rsrc(p.pool_out + (pp0 * p.pool_cs + p.pool_co), 0x7FFFFFFFu);
        const unsigned vp = live ? (unsigned)((4 * l5 * p.pool_cs + n) * 4) : OOB;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const int pp = (wm * 2 + m) * 16 + (r & 3) + 8 * (r >> 2);  // + 4*l5 in vp
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[2][m][r]), rp, vp, pp * p.pool_cs * 4, 0);
          }
      } else {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float o0, o1;
            finish(acc[0][m][r], acc[1][m][r], acc[2][m][r], acc[3][m][r], o0, o1);
            acc[0][m][r] = o0;
            acc[1][m][r] = o1;
          }
        if (p.amax_out) {
          float mx = 0.f;
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fmaxf(fabsf(acc[0][m][r]), fabsf(acc[1][m][r])));
          kocr_amax_update(p.amax_out, live ? mx : 0.f);
        }
        // bytes from the tile's first pixel to the end of the tensor: stores past it are dropped
        const long rem = ((long)p.Mtotal - t.pm0) * ocs4;
        const __amdgpu_buffer_rsrc_t ro =
            uniform_rsrc(p.out + (t.pm0 * p.out_cs + p.out_co), rem < 0x7FFFFFFFL ? (unsigned)rem : 0x7FFFFFFFu);
        const unsigned vo = live ? (unsigned)((8 * l5 * p.out_cs + n) * 4) : OOB;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int px = 2 * ((wm * 2 + m) * 32 + (r & 3) + 8 * (r >> 2));  // + 8*l5 in vo
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[0][m][r]), ro, vo, px * ocs4, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[1][m][r]), ro, vo, (px + 1) * ocs4, 0);
          }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
int prepare_wsplit(kocr_ctx* ctx, ConvLayer& L, const float* w, bool w_is_oihw) {
  // Cout <= 32 (the detector head) stays on the fp32 Winograd kernel: padded to a 64-cout tile the split kernel is
  // slower there (measured 1.53 vs 1.28 ms on upconv4.conv.3, 0.87 vs 0.73 ms on conv_cls.0 per 8 x 1536x1536)
  if (L.KH != 3 || L.KW != 3 || L.dil != 1 || L.Cin % 16 != 0 || L.Cout <= 32) return KOCR_OK;
  const int Cin = L.Cin, Cout = L.Cout;
  const int wcls = Cout > 64 ? 128 : 64;  // couts per block: 4 waves / 2 waves
  const int cp = (Cout + wcls - 1) / wcls * wcls;
  const int nt32 = cp / 32;
  std::vector<unsigned short> u((size_t)(Cin / 16) * 3 * nt32 * 12 * 64 * 8, 0);
  for (int c = 0; c < Cin; ++c)
    for (int ky = 0; ky < 3; ++ky)
      for (int o = 0; o < Cout; ++o) {
        float g[3];
        for (int kx = 0; kx < 3; ++kx)
          g[kx] = w_is_oihw ? w[(((size_t)o * Cin + c) * 3 + ky) * 3 + kx] : w[(((size_t)ky * 3 + kx) * Cin + c) * Cout + o];
        const float U[4] = {g[0], 0.5f * ((g[0] + g[1]) + g[2]), 0.5f * ((g[0] - g[1]) + g[2]), g[2]};
        // MFMA 32x32x16 B operand: lane = (k >> 3) * 32 + (o & 31) holds k = 8*(lane>>5) + j, j = 0..7
        const int k = c % 16, lane = (k >> 3) * 32 + (o & 31), j = k & 7;
        const size_t step = (size_t)(c / 16) * 3 + ky;
        for (int xi = 0; xi < 4; ++xi) {
          unsigned short pc[3];
          kocr_split3_host(U[xi], pc);
          for (int s = 0; s < 3; ++s)
            u[((((step * nt32 + o / 32) * 4 + xi) * 3 + s) * 64 + lane) * 8 + j] = pc[s];
        }
      }
  L.ws_cout_pad = cp;
  void* d = nullptr;
  KOCR_TRY(ctx->dev_alloc(&d, u.size() * sizeof(unsigned short)));
  KOCR_HIP(ctx, hipMemcpy(d, u.data(), u.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
  L.d_ws = (unsigned short*)d;

  return KOCR_OK;
}

bool wsplit_applicable(const ConvLayer& L, const Tensor& in) {
  static const bool off = getenv("KOCR_WSPLIT") && atoi(getenv("KOCR_WSPLIT")) == 0;
  return !off && L.d_ws && in.W % 2 == 0 && in.cs % 4 == 0 && in.co % 4 == 0 && ((uintptr_t)in.p & 15) == 0 &&
         L.Cin % 16 == 0;
}

template <int POOL, int WM, int WN, int HALF, int KB>
static int ws_launch(kocr_ctx* ctx, WsParams& p, size_t M) {
  constexpr int LDS_BYTES = 2 * 4 * (HALF ? 2 : 3) * KB * (2 * WM) * 2 * 256 * 2;  // 48 / 96 KB (bf16x3), 32 / 64 KB * KB (fp16x2)
  static std::atomic<bool> attr_done[64];  // per device (one process may hold contexts on several GPUs); a race only repeats the call
  const int dev = ctx->device & 63;
  if (!attr_done[dev]) {
    KOCR_HIP(ctx, hipFuncSetAttribute((const void*)conv_ws_kernel<POOL, WM, WN, HALF, KB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      LDS_BYTES));
    attr_done[dev] = true;
  }
  static std::atomic<int> n_cus[64];
  if (!n_cus[dev]) {
    hipDeviceProp_t prop;
    KOCR_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
    n_cus[dev] = prop.multiProcessorCount;
  }
  const int n_cu = n_cus[dev];
  const size_t mtiles = POOL ? M / (size_t)(128 * WM) : (M + 128 * WM - 1) / (128 * WM);
  p.total_tiles = (int)(mtiles * (p.Cout_pad / (32 * WN)));
  const int grid = p.total_tiles < n_cu ? p.total_tiles : n_cu;  // persistent: one block per CU
  hipLaunchKernelGGL((conv_ws_kernel<POOL, WM, WN, HALF, KB>), dim3(grid), dim3(512), LDS_BYTES, ctx->stream, p);
  KOCR_HIP(ctx, hipGetLastError());
  return KOCR_OK;
}

int launch_conv_wsplit(kocr_ctx* ctx, const ConvLayer& L, const Tensor& in, const Tensor& out, const Tensor* pool,
                       bool need_full) {
  const int wcls = L.Cout > 64 ? 128 : 64;     // <1,4>: 128 px x 128 couts, <2,2>: 256 px x 64 couts
  const int tile_w = wcls == 128 ? 64 : 128;   // fused pooling: tile = 2 rows x tile_w columns
  const bool fuse = pool && in.H % 2 == 0 && in.W % tile_w == 0;
  const size_t M = in.pixels();
  WsParams p;
  p.in = in.p;
  p.wgt = L.d_ws;
  p.out = out.p;
  p.pre_a = L.d_pre_a;
  p.pre_b = L.d_pre_b;
  p.post_a = L.d_post_a;
  p.post_b = L.d_post_b;
  p.H = in.H;
  p.W = in.W;
  p.Cin = L.Cin;
  p.in_cs = in.cs;
  p.in_co = in.co;
  p.Cout = L.Cout;
  p.Cout_pad = L.ws_cout_pad;
  p.out_cs = out.cs;
  p.out_co = out.co;
  p.relu = L.relu;
  p.nsteps = 3 * (L.Cin / 16);  // divided by the K-step's 16-channel blocks below
  p.Mtotal = (int)M;
  p.pool_out = nullptr;
  p.pool_cs = p.pool_co = p.write_full = p.tiles_per_row = 0;
  p.total_tiles = 0;
  p.amax_out = p.amax_pool = nullptr;  // per-image slots (Tensor::amax) are filled by a reduction pass after the launch
#ifdef KOCR_DEV_SWITCHES
  static const int dbg = getenv("KOCR_WS_DBG") ? atoi(getenv("KOCR_WS_DBG")) : 0;
  p.dbg = dbg;
#endif
  // (round 4: the fp16 arithmetic lives in conv_w43h.hip; the HALF / KB template paths of this kernel are no longer instantiated)
  const bool half = false;
  p.amax_in = nullptr;
  p.w_exp = 0;
  if (fuse) {
    p.pool_out = pool->p;
    p.pool_cs = pool->cs;
    p.pool_co = pool->co;
    p.write_full = need_full ? 1 : 0;
    p.tiles_per_row = in.W / tile_w;
  }
  static const bool per_layer = getenv("KOCR_PROF_LAYERS") != nullptr;
  char nm[64];
  if (per_layer)
    snprintf(nm, sizeof nm, "conv_w%s_%dx%d%s:%s", half ? "h" : "s", wcls == 128 ? 128 : 256, wcls, fuse ? "p" : "", L.name.c_str());
  else
    snprintf(nm, sizeof nm, "conv_w%s_%dx%d%s", half ? "h" : "s", wcls == 128 ? 128 : 256, wcls, fuse ? "_pool" : "");
  const double flops = 2.0 * (double)M * L.Kreal * L.Cout;  // algorithmic (direct-convolution) FLOPs
  const double bytes = 4.0 * ((double)M * L.Cin + (double)M * L.Cout + (double)L.Kreal * L.Cout);
  {
    ProfScope ps(ctx, nm, flops, bytes);
    if (wcls == 128) {
      if (fuse)
        KOCR_TRY((ws_launch<1, 1, 4, 0, 1>(ctx, p, M)));
      else
        KOCR_TRY((ws_launch<0, 1, 4, 0, 1>(ctx, p, M)));
    } else {
      if (fuse)
        KOCR_TRY((ws_launch<1, 2, 2, 0, 1>(ctx, p, M)));
      else
        KOCR_TRY((ws_launch<0, 2, 2, 0, 1>(ctx, p, M)));
    }
  }
  if (out.amax && (!fuse || need_full)) KOCR_TRY(launch_absmax(ctx, out, out.amax));
  if (fuse && pool->amax) KOCR_TRY(launch_absmax(ctx, *pool, pool->amax));
  if (pool && !fuse) return launch_maxpool2x2(ctx, out, *pool);
  return KOCR_OK;
}
