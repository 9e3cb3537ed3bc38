// conv_mfma.hip — fp32 implicit-GEMM convolution on the gfx950 matrix cores.
//
// Replaces every keras.layers.Conv2D / Dense on the hot path of the reference
// (CRAFT: detection.py:65-103, 365-410; CRNN: recognition.py:217-327).  One kernel
// template covers 3x3 (incl. dilation 6), 5x5, 1x1 and dense layers:
//
//   GEMM view   M = N*H*W output pixels, N = Cout, K = KH*KW*Cin  (k = (ky*KW+kx)*Cin + c)
//   A[m][k]     gathered on the fly from the NHWC activation ('same' zero padding)
//   B[k][n]     weights pre-packed as [Kpad][Cout_pad]
//   D           v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD), fp32 accumulate
//   epilogue    out = post_a * relu?(pre_a*acc + pre_b) + post_b   (bias + folded BN)
//
// Block = 256 threads = 4 waves; block tile 128x128 (Cout > 64), 128x64 (Cout <= 64) or 256x32
// (Cout <= 32); K-step 16.  K order for Cin % 16 == 0: [16-channel group][tap][16] — the taps of
// one channel group re-read the same 64-B pieces shifted by a pixel, so they hit L1/L2 (measured:
// 2.7x less L2-miss traffic than tap-major).  A is gathered with float4 loads through 32-bit
// block-relative offsets and a per-pixel tap-validity bit mask (no 64-bit multiplies, no coordinate
// compares, no divisions in the K loop), register-staged one K-step ahead into a double-buffered
// k-major LDS image so that both MFMA operand reads are conflict-free ds_read_b32 (lane ->
// consecutive m / n); one barrier per K-step.  blockIdx -> tile mapping is XCD-aware: each XCD
// (private L2) gets a contiguous range of tiles, n-tile fastest.  POOL kernels fuse the following
// 2x2/stride-2 max-pool: the tile is 2 image rows x 64 columns in window-major order, so the four
// pixels of a pooling window are the four r&3 accumulator registers of one lane.
//
// Measured on MI355X (profiles/): 125 TFLOP/s = 79 % of the 157.3 TF fp32 MFMA peak on the
// 128x128 instances inside the pipeline, matrix pipe busy 80 % at 2.38 GHz, 3.8 waves/SIMD.
// (Round 1's developer A/B variants -- KOCR_CONV_VARIANT: fragment prefetch, 8-wave tiles, ablations; KOCR_CONV_STAGGER --
// were removed in round 6: since round 2 this kernel is the fp32 fallback for narrow / odd layers only.)
#include "common.h"
#include <cmath>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
// native clang vector (HIP's float4 wrapper struct defeats SROA for arrays captured by lambdas
// and sends the staged tile through scratch)
typedef float v4f __attribute__((ext_vector_type(4)));

struct ConvParams {
  const float* in;
  const uint8_t* in_u8;
  const float* lut;  // [3][256] normalisation table for u8 input
  const float* wgt;
  float* out;
  const float* pre_a;
  const float* pre_b;
  const float* post_a;
  const float* post_b;
  int H, W, Cin, in_cs, in_co;
  int Cout, Cout_pad, out_cs, out_co;
  int KH, KW, dil, padh, padw;
  int relu;
  int Mtotal;
  int Kreal, nchunks;
  // fused 2x2/stride-2 max-pool epilogue (POOL kernels): tile = 2 image rows x BM/2 columns
  float* pool_out;
  int pool_cs, pool_co, write_full, tiles_per_row;
  int stagger;  // s_sleep units per stagger step (0 = off)
  unsigned* amax_out;   // Tensor::amax slots of the output / pooled output, or nullptr
  unsigned* amax_pool;
  int tap_inner, ntaps;  // K order: k = ((c/16)*ntaps + tap)*16 + c%16 (vector-gather layers)
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7;
  const int q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// Position in the K order [16-channel group][tap][16] of the vector-gather path, advanced one
// K-step at a time (no divisions in the steady state).
struct TapWalk {
  int tap, kx, dy, dx, c0;
};
__device__ __forceinline__ TapWalk walk_next(TapWalk w, const ConvParams& p, int bk) {
  if (++w.tap == p.ntaps) {
    w.tap = 0;
    w.kx = 0;
    w.dy = -p.padh;
    w.dx = -p.padw;
    w.c0 += bk;
  } else if (++w.kx == p.KW) {
    w.kx = 0;
    w.dx = -p.padw;
    w.dy += p.dil;
  } else {
    w.dx += p.dil;
  }
  return w;
}

// MODE 0: Cin % 16 == 0, float4 gathers.  MODE 1: generic scalar gather (f32).
// MODE 2: generic scalar gather from raw u8 RGB through the normalisation LUT.
template <int BM, int BN, int WM, int WN, int MODE, int BK = 16, int PF = 0, int POOL = 0>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN >= 8) ? 2 : 2) void conv_mfma_kernel(ConvParams p) {
  constexpr int NT = 64 * WM * WN;  // threads per block
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int QPR = BK / 4;             // float4 quads per pixel row of the K-step
  constexpr int A_PER_T = BM * QPR / NT;
  constexpr int A_MSTEP = NT / QPR;      // pixels covered by one pass of the block
  constexpr int B_F4 = BK * BN / 4;
  constexpr int B_PER_T = (B_F4 + NT - 1) / NT;
  static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves");
  static_assert(TM >= 1 && TN >= 1, "tile");

  __shared__ float As[2][BK][LDA];
  __shared__ float Bs[2][BK][LDB];
  __shared__ float lut_s[MODE == 2 ? 768 : 1];  // compute_input table (first layer, uint8 input)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: SGPR
  const int wm = wave / WN, wn = wave % WN;
  const int lr = lane & 31, lk = lane >> 5;

  const int nblk_n = p.Cout_pad / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = tile / nblk_n, nt = tile - mt * nblk_n;
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- per-thread gather coordinates (fixed over the K loop) --------------------------
  const int quad = tid % QPR;
  int a_oy[A_PER_T], a_ox[A_PER_T];
  long a_pm[A_PER_T];
  // MODE 0 fast path: 32-bit element offsets relative to the block's first pixel and a bit mask of
  // the taps that fall inside the image (KH*KW <= 25), so the K loop needs no 64-bit multiplies
  // and no coordinate compares.
  int a_off[A_PER_T];
  unsigned a_mask[A_PER_T];
  // tile -> pixels.  POOL == 0: BM consecutive pixels of the flattened (n,y,x) order.
  // POOL == 1: 2 image rows x BM/2 columns in window-major order, local m = 4*q + 2*ry + cx, so
  // that the four pixels of a 2x2 pooling window are MFMA rows 4q..4q+3 = registers r&3 of one lane.
  long pm0;      // linear pixel index of local pixel 0
  int y0t = 0, x0t = 0;
  if constexpr (POOL) {
    const int rp_lin = mt / p.tiles_per_row, cb = mt - rp_lin * p.tiles_per_row;
    const int hh = p.H >> 1;
    const int n = rp_lin / hh, rp = rp_lin - n * hh;
    y0t = 2 * rp;
    x0t = cb * (BM / 2);
    pm0 = ((long)n * p.H + y0t) * p.W + x0t;
  } else {
    pm0 = m0;
  }
  const float* blk_in = p.in + (pm0 * p.in_cs + p.in_co);
#pragma unroll
  for (int i = 0; i < A_PER_T; ++i) {
    const int ml = tid / QPR + A_MSTEP * i;
    int rel;  // pixel index relative to pm0
    bool inside;
    int ox, oy;
    if constexpr (POOL) {
      const int ry = (ml >> 1) & 1;
      rel = ry * p.W + 2 * (ml >> 2) + (ml & 1);
      oy = y0t + ry;
      ox = x0t + 2 * (ml >> 2) + (ml & 1);
      inside = true;
    } else {
      rel = ml;
      const int pm = m0 + ml;
      inside = pm < p.Mtotal;
      ox = pm % p.W;
      oy = (pm / p.W) % p.H;
    }
    unsigned mask = 0;
    if (inside) {
      a_ox[i] = ox;
      a_oy[i] = oy;
      a_pm[i] = pm0 + rel;
      if constexpr (MODE == 0) {
        for (int tap = 0; tap < p.KH * p.KW; ++tap) {
          const int ky = tap / p.KW, kx = tap - ky * p.KW;
          const int iy = oy + ky * p.dil - p.padh, ix = ox + kx * p.dil - p.padw;
          if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) mask |= 1u << tap;
        }
      }
    } else {
      a_ox[i] = -(1 << 28);  // every tap falls outside -> zeros
      a_oy[i] = -(1 << 28);
      a_pm[i] = 0;
    }
    a_mask[i] = mask;
    a_off[i] = rel * p.in_cs + quad * 4;
  }
  // weight-tile pointers advance by BK rows per K-step
  const float* b_ptr[B_PER_T];
#pragma unroll
  for (int j = 0; j < B_PER_T; ++j) {
    const int f = (B_F4 % NT == 0) ? tid + NT * j : (tid + NT * j) % B_F4;  // wrap: duplicates are harmless
    const int krow = f / (BN / 4), nc = f - krow * (BN / 4);
    b_ptr[j] = p.wgt + (size_t)krow * p.Cout_pad + n0 + nc * 4;
  }

  // staging registers; PF == 1 keeps two K-steps in flight (prefetch distance 2)
  if constexpr (MODE == 2) {
    for (int i = tid; i < 768; i += NT) lut_s[i] = p.lut[i];
    __syncthreads();
  }
  v4f ra0[A_PER_T], rb0[B_PER_T];
  v4f ra1[(PF == 1) ? A_PER_T : 1], rb1[(PF == 1) ? B_PER_T : 1];

  TapWalk walk{0, 0, -p.padh, -p.padw, 0};
  auto load_chunk = [&](int ch, v4f* __restrict__ ra, v4f* __restrict__ rb) __attribute__((always_inline)) {
    if constexpr (MODE == 0) {
      // K order = [16-channel group][tap][16]: the 9 taps of one channel group re-read (shifted by a
      // pixel) the same 64-B pieces back to back, so taps 2..9 hit L1/L2 instead of HBM/MALL
      // `walk` follows the K order incrementally: chunks are always requested in order 0,1,2,...
      (void)ch;
      const TapWalk w = walk;
      const int tap_off = (w.dy * p.W + w.dx) * p.in_cs + w.c0;  // wave-uniform
#pragma unroll
      for (int i = 0; i < A_PER_T; ++i) {
        const bool ok = (a_mask[i] >> w.tap) & 1u;
        const int off = ok ? a_off[i] + tap_off : 0;  // offset 0 = the block's first pixel: always mapped
        v4f v = *reinterpret_cast<const v4f*>(blk_in + off);
        ra[i] = ok ? v : v4f{0.f, 0.f, 0.f, 0.f};
      }
      walk = walk_next(w, p, BK);
    } else if constexpr (MODE == 2) {
      // first layer, raw uint8 RGB: K order = [tap][R, G, B, 0], so one float4 of the K-step is one
      // tap of one pixel: 3 byte loads + 3 LDS table look-ups, no divisions
      const int tap = ch * (BK / 4) + quad;
      const int ky = tap / 3, kx = tap - ky * 3;  // 3x3 only
      const int dy = ky - 1, dx = kx - 1;
#pragma unroll
      for (int i = 0; i < A_PER_T; ++i) {
        const int iy = a_oy[i] + dy, ix = a_ox[i] + dx;
        v4f v = {0.f, 0.f, 0.f, 0.f};
        if (tap < 9 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
          const uint8_t* px = p.in_u8 + (a_pm[i] + (long)dy * p.W + dx) * 3;
          v.x = lut_s[px[0]];
          v.y = lut_s[256 + px[1]];
          v.z = lut_s[512 + px[2]];
        }
        ra[i] = v;
      }
    } else {
#pragma unroll
      for (int i = 0; i < A_PER_T; ++i) {
        float e[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int k = ch * BK + quad * 4 + q;
          float v = 0.f;
          if (k < p.Kreal) {
            int tap, c;
            if (p.tap_inner) {
              const int cg = k / (16 * p.ntaps), rem = k - cg * 16 * p.ntaps;
              tap = rem >> 4;
              c = cg * 16 + (rem & 15);
            } else {
              tap = k / p.Cin;
              c = k - tap * p.Cin;
            }
            const int ky = tap / p.KW, kx = tap - ky * p.KW;
            const int dy = ky * p.dil - p.padh, dx = kx * p.dil - p.padw;
            const int iy = a_oy[i] + dy, ix = a_ox[i] + dx;
            if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
              const long off = (a_pm[i] + (long)dy * p.W + dx) * p.in_cs + p.in_co + c;
              v = p.in[off];
            }
          }
          e[q] = v;
        }
        ra[i] = v4f{e[0], e[1], e[2], e[3]};
      }
    }
#pragma unroll
    for (int j = 0; j < B_PER_T; ++j) {
      rb[j] = *reinterpret_cast<const v4f*>(b_ptr[j] + (size_t)ch * BK * p.Cout_pad);
    }
  };

  auto store_chunk = [&](int buf, const v4f* __restrict__ ra, const v4f* __restrict__ rb) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
      const int m = tid / QPR + A_MSTEP * i;
      As[buf][quad * 4 + 0][m] = ra[i].x;
      As[buf][quad * 4 + 1][m] = ra[i].y;
      As[buf][quad * 4 + 2][m] = ra[i].z;
      As[buf][quad * 4 + 3][m] = ra[i].w;
    }
#pragma unroll
    for (int j = 0; j < B_PER_T; ++j) {
      const int f = (B_F4 % NT == 0) ? tid + NT * j : (tid + NT * j) % B_F4;
      const int krow = f / (BN / 4), nc = f - krow * (BN / 4);
      *reinterpret_cast<v4f*>(&Bs[buf][krow][nc * 4]) = rb[j];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute_chunk = [&](int buf) {
    {
#pragma unroll
      for (int kp = 0; kp < BK / 2; ++kp) {
        float a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = As[buf][2 * kp + lk][wm * WTM + i * 32 + lr];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = Bs[buf][2 * kp + lk][wn * WTN + j * 32 + lr];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
  };

  if (p.stagger) {
    // de-correlate the co-resident workgroups of a CU: identical blocks launched together otherwise
    // run their MFMA and non-MFMA phases in lockstep and the matrix pipe idles in the gaps
    const int s4 = (blockIdx.x >> 8) & 3;
    for (int i = 0; i < s4 * p.stagger; ++i) __builtin_amdgcn_s_sleep(8);  // ~512 cycles per unit
  }
  load_chunk(0, ra0, rb0);
  store_chunk(0, ra0, rb0);
  __syncthreads();

  const int nch = p.nchunks;
  if constexpr (PF == 1) {
    // prefetch distance 2: while K-step ch is on the matrix cores, K-step ch+1 sits in one register
    // set (stored to LDS at the end of the step) and K-step ch+2 is being loaded into the other
    if (nch > 1) load_chunk(1, ra0, rb0);
    int ch = 0;
    while (ch + 1 < nch) {
      if (ch + 2 < nch) load_chunk(ch + 2, ra1, rb1);
      __builtin_amdgcn_sched_barrier(0);
      compute_chunk(ch & 1);
      __builtin_amdgcn_sched_barrier(0);
      store_chunk((ch + 1) & 1, ra0, rb0);
      __syncthreads();
      ++ch;
      if (!(ch + 1 < nch)) break;
      if (ch + 2 < nch) load_chunk(ch + 2, ra0, rb0);
      __builtin_amdgcn_sched_barrier(0);
      compute_chunk(ch & 1);
      __builtin_amdgcn_sched_barrier(0);
      store_chunk((ch + 1) & 1, ra1, rb1);
      __syncthreads();
      ++ch;
    }
  } else {
    // steady state: no conditionals around the staging registers
    for (int ch = 0; ch + 1 < nch; ++ch) {
      const int buf = ch & 1;
      if constexpr (PF < 2 || PF >= 4) load_chunk(ch + 1, ra0, rb0);
      // keep the prefetch loads ahead of the MFMAs: without this fence hipcc sinks the weight-tile
      // loads to just before their LDS store and the wave eats the full L2 latency every K-step
      __builtin_amdgcn_sched_barrier(0);
      compute_chunk(buf);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (PF == 4) {  // ablation: loads + barrier, no LDS stores (keep the loads live)
        for (int i = 0; i < A_PER_T; ++i) asm volatile("" ::"v"(ra0[i]));
        for (int j = 0; j < B_PER_T; ++j) asm volatile("" ::"v"(rb0[j]));
        __syncthreads();
      } else if constexpr (PF == 5) {  // ablation: loads + LDS stores, no barrier
        store_chunk(buf ^ 1, ra0, rb0);
      } else if constexpr (PF != 3) {
        store_chunk(buf ^ 1, ra0, rb0);
        __syncthreads();
      }
    }
  }
  compute_chunk((nch - 1) & 1);

  // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * WTN + j * 32 + lr;
    const bool live = n < p.Cout;
    const int nc = live ? n : p.Cout - 1;
    const float pa = p.pre_a[nc], pb = p.pre_b[nc];
    const bool has_post = p.post_a != nullptr;
    const float qa = has_post ? p.post_a[nc] : 1.f;
    const float qb = has_post ? p.post_b[nc] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if constexpr (POOL) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {  // g = r>>2: one 2x2 window per register group
          const int ml = wm * WTM + i * 32 + 8 * g + 4 * lk;  // local index of the window's first pixel
          float best = -INFINITY;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = acc[i][j][4 * g + e] * pa + pb;
            if (p.relu) v = fmaxf(v, 0.f);
            if (has_post) v = v * qa + qb;
            best = fmaxf(best, v);
            if (p.write_full && live) {
              const long pix = pm0 + (long)(e >> 1) * p.W + 2 * (ml >> 2) + (e & 1);
              p.out[pix * p.out_cs + p.out_co + n] = v;
            }
          }
          // pooled pixel: (n_img, y0t/2, x0t/2 + q)
          const long nimg = pm0 / ((long)p.H * p.W);
          const long pp = (nimg * (p.H >> 1) + (y0t >> 1)) * (p.W >> 1) + (x0t >> 1) + (ml >> 2);
          if (live) p.pool_out[pp * p.pool_cs + p.pool_co + n] = best;
        }
      } else {
        // in place; the stores follow below, back to back
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[i][j][r] * pa + pb;
          if (p.relu) v = fmaxf(v, 0.f);
          if (has_post) v = v * qa + qb;
          acc[i][j][r] = v;
        }
      }
    }
    if (p.amax_out || p.amax_pool) {  // accumulators hold the finished outputs (non-POOL) / pre-pool values are bounded by them
      float mx = 0.f;
      if constexpr (!POOL) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fabsf(acc[i][j][r]));
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[i][j][r] * pa + pb;
            if (p.relu) v = fmaxf(v, 0.f);
            if (has_post) v = v * qa + qb;
            mx = fmaxf(mx, fabsf(v));
          }
      }
      mx = live ? mx : 0.f;
      if (p.amax_out) kocr_amax_update(p.amax_out, mx);
      if (p.amax_pool) kocr_amax_update(p.amax_pool, mx);
    }
    if constexpr (!POOL) {
      // Raw buffer stores: one per-lane byte offset, the per-register pixel offset in an SGPR, pixels past
      // the end of the tensor and padded couts dropped by the range check.  (A per-pixel `if (m < M)`
      // around each store made hipcc wait for vmcnt(0) before every store.)
      const int ocs4 = p.out_cs * 4;
      const long rem = ((long)p.Mtotal - m0) * ocs4;
      const unsigned long long bb = (unsigned long long)(p.out + ((long)m0 * p.out_cs + p.out_co));
      const unsigned long long bbu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(bb >> 32)) << 32) |
                                     (unsigned)__builtin_amdgcn_readfirstlane((int)bb);
      const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
          (void*)bbu, 0, __builtin_amdgcn_readfirstlane((int)(rem < 0x7FFFFFFFL ? rem : 0x7FFFFFFFL)), 0x00020000);
      const unsigned vo = live ? (unsigned)((4 * lk * p.out_cs + n) * 4) : 0x80000000u;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int px = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2);  // + 4*lk in vo
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][j][r]), ro, vo, px * ocs4, 0);
        }
    }
  }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
static int round_up(int a, int b) { return (a + b - 1) / b * b; }

int prepare_conv(kocr_ctx* ctx, ConvLayer& L, const float* w, bool w_is_oihw, int Cin, int Cout,
                 int KH, int KW, int dil, const float* pre_a, const float* pre_b, int relu,
                 const float* post_a, const float* post_b) {
  // A layer object may be prepared again (kocr_load_* on a context that already holds weights): nothing derived from the
  // previous weights may survive.  In round 4 d_pre_a_h did -- the <= 32-cout fp16 kernel's per-cout exponents were
  // uploaded only `if (!L.d_pre_a_h)`, so a second load kept the FIRST load's exponents next to the new weights (found by
  // tests/test_range_gpu.py's re-loaded detector: conv_cls.0 / .2 a factor 2^k off per channel).  The old device buffers
  // are released here, after the stream has drained (kept until the context died, every kocr_load_* on a live
  // context leaked the whole previous weight set -- fp32 plus every split / transformed copy).
  {
    void* old[] = {L.d_w, L.d_pre_a, L.d_pre_b, L.d_post_a, L.d_post_b, L.d_w_rgb4, L.d_pre_a_h, L.d_ws, L.d_w4,
                   L.d_w4h, L.d_ds, L.d_first, L.d_hs, L.d_hs16, L.d_hsh, L.d_k5};
    bool any = false;
    for (void* q : old) any = any || q != nullptr;
    if (any) {
      KOCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
      for (void* q : old) ctx->release(q);
    }
  }
  L.d_w = L.d_pre_a = L.d_pre_b = L.d_post_a = L.d_post_b = L.d_w_rgb4 = L.d_pre_a_h = nullptr;
  L.d_ws = L.d_w4 = L.d_w4h = L.d_ds = L.d_first = L.d_hs = L.d_hs16 = L.d_hsh = L.d_k5 = nullptr;
  L.hs_wexp.clear();
  L.first_bound = 0.f;
  L.Cin = Cin;
  L.Cout = Cout;
  L.KH = KH;
  L.KW = KW;
  L.dil = dil;
  L.relu = relu;
  L.Kreal = KH * KW * Cin;
  L.tap_inner = (Cin % 16 == 0);
  L.Kpad = round_up(L.Kreal, 32);
  L.BN = Cout > 64 ? 128 : (Cout > 32 ? 64 : 32);
  L.Cout_pad = round_up(Cout, L.BN);
  std::vector<float> wp((size_t)L.Kpad * L.Cout_pad, 0.f);
  for (int ky = 0; ky < KH; ++ky)
    for (int kx = 0; kx < KW; ++kx)
      for (int c = 0; c < Cin; ++c)
        for (int o = 0; o < Cout; ++o) {
          const float v = w_is_oihw ? w[(((size_t)o * Cin + c) * KH + ky) * KW + kx]
                                    : w[(((size_t)ky * KW + kx) * Cin + c) * Cout + o];
          const int tap = ky * KW + kx;
          const size_t row = L.tap_inner ? (size_t)((c / 16) * KH * KW + tap) * 16 + (c % 16) : (size_t)tap * Cin + c;
          wp[row * L.Cout_pad + o] = v;
        }
  KOCR_TRY(ctx->upload(&L.d_w, wp));
  KOCR_TRY(prepare_wsplit(ctx, L, w, w_is_oihw));
  KOCR_TRY(prepare_w43(ctx, L, w, w_is_oihw));
  KOCR_TRY(prepare_w43h(ctx, L, w, w_is_oihw, pre_a));
  KOCR_TRY(prepare_dsplit(ctx, L, w, w_is_oihw));
  KOCR_TRY(prepare_hsplit(ctx, L, w, w_is_oihw));
  KOCR_TRY(prepare_k5(ctx, L, w, w_is_oihw));
  KOCR_TRY(prepare_first(ctx, L, w, w_is_oihw));
  if (Cin == 3 && KH == 3 && KW == 3 && dil == 1) {  // uint8 first layer: K order [tap][R,G,B,0], 48 rows
    std::vector<float> w4((size_t)48 * L.Cout_pad, 0.f);
    for (int tap = 0; tap < 9; ++tap)
      for (int c = 0; c < 3; ++c)
        for (int o = 0; o < Cout; ++o)
          w4[(size_t)(tap * 4 + c) * L.Cout_pad + o] =
              w_is_oihw ? w[(((size_t)o * 3 + c) * 3 + tap / 3) * 3 + tap % 3] : w[((size_t)tap * 3 + c) * Cout + o];
    KOCR_TRY(ctx->upload(&L.d_w_rgb4, w4));
  }
  std::vector<float> a(L.Cout_pad, 1.f), b(L.Cout_pad, 0.f);
  for (int o = 0; o < Cout; ++o) {
    if (pre_a) a[o] = pre_a[o];
    if (pre_b) b[o] = pre_b[o];
  }
  KOCR_TRY(ctx->upload(&L.d_pre_a, a));
  KOCR_TRY(ctx->upload(&L.d_pre_b, b));
  if (L.d_first) {
    // The first layer reads compute_input(uint8) (detection.py:34-42): |x_c| <= max(mean_c, 1 - mean_c) / variance_c, so
    // |out_o| <= |post_a| (|pre_a| sum_{tap,c} |w| xmax_c + |pre_b|) + |post_b| is a bound that needs no reduction at all:
    // launch_conv_first fills the output's per-image max-|x| slots with it (an upper bound is all the fp16 consumer needs,
    // and a constant is trivially independent of the batch).  In-kernel tracking cost 0.18 ms per 8 x 1536^2 images.
    const double xmax[3] = {0.515 / 0.229, 0.544 / 0.224, 0.594 / 0.225};
    double bound = 0;
    for (int o = 0; o < Cout; ++o) {
      double sw = 0;
      for (int ky = 0; ky < KH; ++ky)
        for (int kx = 0; kx < KW; ++kx)
          for (int c = 0; c < Cin; ++c)
            sw += std::fabs((double)(w_is_oihw ? w[(((size_t)o * Cin + c) * KH + ky) * KW + kx] : w[(((size_t)ky * KW + kx) * Cin + c) * Cout + o])) * xmax[c % 3];
      double v = std::fabs((double)a[o]) * sw + std::fabs((double)b[o]);
      if (post_a || post_b) v = std::fabs(post_a ? (double)post_a[o] : 1.0) * v + std::fabs(post_b ? (double)post_b[o] : 0.0);
      bound = std::max(bound, v);
    }
    L.first_bound = (float)(bound * (1.0 + 1e-6));
  }
  if (L.d_hsh && !L.d_pre_a_h) {  // <= 32-cout fp16 kernel: the per-cout weight scale is undone in pre_a
    std::vector<float> ah(L.Cout_pad, 1.f);
    for (int o = 0; o < Cout; ++o) ah[o] = std::ldexp(a[o], -L.hs_wexp[o]);
    KOCR_TRY(ctx->upload(&L.d_pre_a_h, ah));
  }
  L.d_post_a = L.d_post_b = nullptr;
  if (post_a || post_b) {
    std::vector<float> qa(L.Cout_pad, 1.f), qb(L.Cout_pad, 0.f);
    for (int o = 0; o < Cout; ++o) {
      if (post_a) qa[o] = post_a[o];
      if (post_b) qb[o] = post_b[o];
    }
    KOCR_TRY(ctx->upload(&L.d_post_a, qa));
    KOCR_TRY(ctx->upload(&L.d_post_b, qb));
  }
  return KOCR_OK;
}

template <int BM, int BN, int WM, int WN, int BK, int PF>
static void dispatch_mode(int mode, dim3 grid, hipStream_t s, const ConvParams& p) {
  if (mode == 0)
    hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, WM, WN, 0, BK, PF>), grid, dim3(64 * WM * WN), 0, s, p);
  else if (mode == 1)
    hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, WM, WN, 1, BK, PF>), grid, dim3(64 * WM * WN), 0, s, p);
  else
    hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, WM, WN, 2, BK, PF>), grid, dim3(64 * WM * WN), 0, s, p);
}


// BK=32 in vector mode needs Cin % 32 == 0

int launch_conv(kocr_ctx* ctx, const ConvLayer& L, const Tensor& in, const uint8_t* in_u8,
                const float* lut, const Tensor& out) {
  return launch_conv_pool(ctx, L, in, in_u8, lut, out, nullptr);
}

// Convolution with an optional fused 2x2/stride-2 max-pool epilogue.  `pool`: destination of the
// pooled tensor (H/2 x W/2) or nullptr.  `out.p == nullptr` with a pool destination means the
// full-resolution result is not needed (need_full == false lets the fused epilogue skip writing it).
// Falls back to conv + maxpool kernel when the shape does not tile (odd H, W not a multiple of 64,
// scalar-gather layers); `out` must always be a real buffer.
int launch_conv_pool(kocr_ctx* ctx, const ConvLayer& L, const Tensor& in, const uint8_t* in_u8,
                     const float* lut, const Tensor& out, const Tensor* pool, bool need_full) {
  if (!L.ready()) KOCR_FAIL(ctx, KOCR_ENOWEIGHTS, "conv layer " + L.name + " has no weights");
  if (in.C != L.Cin || out.C != L.Cout || in.N != out.N || in.H != out.H || in.W != out.W)
    KOCR_FAIL(ctx, KOCR_EINVAL, "conv " + L.name + ": shape mismatch");
  const size_t M = in.pixels();
  if (M == 0) return KOCR_OK;
  if (M > (size_t)0x7fffffff - 512) KOCR_FAIL(ctx, KOCR_EINVAL, "conv " + L.name + ": too many pixels");
  ConvParams p;
  p.in = in.p;
  p.in_u8 = in_u8;
  p.lut = lut;
  p.wgt = L.d_w;
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
  p.Cout_pad = L.Cout_pad;
  p.out_cs = out.cs;
  p.out_co = out.co;
  p.KH = L.KH;
  p.KW = L.KW;
  p.dil = L.dil;
  p.padh = L.dil * (L.KH - 1) / 2;
  p.padw = L.dil * (L.KW - 1) / 2;
  p.relu = L.relu;
  p.Mtotal = (int)M;
  p.Kreal = L.Kreal;
  const int bk = 16;
  p.nchunks = (L.Kreal + bk - 1) / bk;  // weight rows are zero padded up to Kpad (multiple of 32)
  p.pool_out = nullptr;
  p.tap_inner = L.tap_inner ? 1 : 0;
  p.amax_out = p.amax_pool = nullptr;  // per-image slots (Tensor::amax) are filled by a reduction pass after the launch
  p.ntaps = L.KH * L.KW;
  p.stagger = 0;
  p.pool_cs = p.pool_co = p.write_full = p.tiles_per_row = 0;
  int mode;
  if (in_u8) {
    if (!L.d_w_rgb4) KOCR_FAIL(ctx, KOCR_EINVAL, "conv " + L.name + ": uint8 input needs a 3x3, 3-channel layer");
    mode = 2;
    p.wgt = L.d_w_rgb4;
    p.nchunks = 3;
  }
  else if (L.Cin % 16 == 0 && in.cs % 4 == 0 && in.co % 4 == 0 && ((uintptr_t)in.p & 15) == 0)
    mode = 0;
  else
    mode = 1;
  const bool fuse_pool = pool && mode == 0 && L.BN >= 64 && (in.H % 2 == 0) && (in.W % 64 == 0);
  // Layouts with zero padding that belongs to the tensor (cell grids, Tensor::cellW; width-padded rows, Tensor::Wv) are
  // read and WRITTEN correctly by the fp16 F(4,3) kernels only: anything else fails here instead of silently writing
  // convolution values into the padding
  if (in.cellW || out.cellW || (pool && pool->cellW)) {
    if (in_u8 || !w43_applicable(ctx, L, in))
      KOCR_FAIL(ctx, KOCR_EINVAL, "conv " + L.name + ": a cell-grid tensor needs the fp16 vertical-reuse F(4,3) kernel (disabled by a switch or the arithmetic mode)");
    return launch_conv_w43(ctx, L, in, out, pool, need_full);
  }
  if (out.Wv && out.Wv < out.W && (in_u8 || !w43_applicable(ctx, L, in) || !w43_flat_h_ok(ctx, L)))
    KOCR_FAIL(ctx, KOCR_EINVAL, "conv " + L.name + ": a width-padded output (Tensor::Wv) needs the flattened fp16 F(4,3) kernel (disabled by a switch or the arithmetic mode)");
  if (!out.p) KOCR_FAIL(ctx, KOCR_EINVAL, "conv " + L.name + ": no output buffer");
  if (in_u8 && lut && first_applicable(ctx, L, in)) {  // first layer from raw uint8 on the split path
    kocr_note_dispatch("first", L, in);
    KOCR_TRY(launch_conv_first(ctx, L, in, in_u8, lut, out));  // maintains out.amax itself
    return pool ? launch_maxpool2x2(ctx, out, *pool) : KOCR_OK;
  }
  if (!in_u8 && !pool && k5_applicable(ctx, L, in, out)) {  // 5x5, 16 couts, small images
    kocr_note_dispatch("k5", L, in);
    return launch_conv_k5(ctx, L, in, out);
  }
  // bf16x3-split Winograd on the bf16 matrix cores (fp32-class accuracy, see conv_wsplit.hip)
  if (!in_u8 && w43_applicable(ctx, L, in)) {
    return launch_conv_w43(ctx, L, in, out, pool, need_full);
  }
  if (!in_u8 && wsplit_applicable(L, in)) {
    kocr_note_dispatch("wsplit", L, in);
    return launch_conv_wsplit(ctx, L, in, out, pool, need_full);
  }
  if (!in_u8 && dsplit_applicable(L, in)) {
    kocr_note_dispatch("dsplit", L, in);
    KOCR_TRY(launch_conv_dsplit(ctx, L, in, out));
    return pool ? launch_maxpool2x2(ctx, out, *pool) : KOCR_OK;
  }
  if (!in_u8 && hsplit_applicable(ctx, L, in)) {  // few couts: split once into LDS
    kocr_note_dispatch("hsplit", L, in);
    KOCR_TRY(launch_conv_hsplit(ctx, L, in, out));
    return pool ? launch_maxpool2x2(ctx, out, *pool) : KOCR_OK;
  }
  // (round 1's fp32 Winograd F(2,3) kernel, conv_wino.hip, stood here: by round 5 only <= 32-cout layers on images of fewer
  //  than 4096 pixels still reached it -- they take conv_hs_kernel now, everything else the fp32 MFMA kernel below.  Removed
  //  in round 6; profiles/r06_dispatch.txt)
  if (pool && !fuse_pool) {  // unfused: conv to full resolution, then the pooling kernel
    KOCR_TRY(launch_conv_pool(ctx, L, in, in_u8, lut, out, nullptr));
    return launch_maxpool2x2(ctx, out, *pool);
  }
  if (fuse_pool) {
    if (pool->H != in.H / 2 || pool->W != in.W / 2 || pool->C != L.Cout || pool->N != in.N)
      KOCR_FAIL(ctx, KOCR_EINVAL, "conv " + L.name + ": bad pooled shape");
    p.pool_out = pool->p;
    p.pool_cs = pool->cs;
    p.pool_co = pool->co;
    p.write_full = need_full ? 1 : 0;
    p.tiles_per_row = in.W / 64;
  }
  kocr_note_dispatch("mfma", L, in);
  const double flops = 2.0 * (double)M * L.Kreal * L.Cout;
  const double bytes = 4.0 * ((double)M * L.Cin + (double)M * L.Cout + (double)L.Kreal * L.Cout);
  char nm[64];
  const int BM = 128;
  static const bool per_layer = getenv("KOCR_PROF_LAYERS") != nullptr;  // developer: one row per layer
  if (per_layer)
    snprintf(nm, sizeof nm, "conv_%dx%d_m%d%s:%s", BM, L.BN, mode, fuse_pool ? "p" : "", L.name.c_str());
  else
    snprintf(nm, sizeof nm, "conv_mfma_%dx%d_m%d%s", BM, L.BN, mode, fuse_pool ? "_pool" : "");
  {
  ProfScope ps(ctx, nm, flops, bytes);
  const int mtiles = (int)((M + BM - 1) / BM);
  dim3 grid(mtiles * (L.Cout_pad / L.BN));
  if (fuse_pool) {  // BM == 128: tile = 2 rows x 64 columns; M is an exact multiple of 128
    if (L.BN == 128)
      hipLaunchKernelGGL((conv_mfma_kernel<128, 128, 2, 2, 0, 16, 0, 1>), grid, dim3(256), 0, ctx->stream, p);
    else
      hipLaunchKernelGGL((conv_mfma_kernel<128, 64, 2, 2, 0, 16, 0, 1>), grid, dim3(256), 0, ctx->stream, p);
  } else if (L.BN == 128)
    dispatch_mode<128, 128, 2, 2, 16, 0>(mode, grid, ctx->stream, p);
  else if (L.BN == 64)
    dispatch_mode<128, 64, 2, 2, 16, 0>(mode, grid, ctx->stream, p);
  else
    dispatch_mode<128, 32, 4, 1, 16, 0>(mode, grid, ctx->stream, p);
  KOCR_HIP(ctx, hipGetLastError());
  }
  if (out.amax && (!fuse_pool || need_full)) KOCR_TRY(launch_absmax(ctx, out, out.amax));
  if (fuse_pool && pool->amax) KOCR_TRY(launch_absmax(ctx, *pool, pool->amax));
  return KOCR_OK;
}
