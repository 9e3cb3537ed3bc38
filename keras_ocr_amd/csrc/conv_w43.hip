// conv_w43.hip — 3x3 / stride 1 / dilation 1 convolution, 1-D Winograd F(4,3) along image rows, fp32 products on
// the gfx950 BF16 matrix cores through the exact 3-way operand split of conv_wsplit.hip.
//
// Why a second Winograd kernel: conv_wsplit.hip (F(2,3)) sits on the chip's power budget at 0.55 of the bf16 issue
// peak, so only FEWER matrix-core products per output make it faster.  F(4,3) produces 4 outputs of a row from 6
// points per vertical tap -- 6/12 = 1/2 of the direct multiplies against F(2,3)'s 2/3: 3.0 instead of 4.0 issued
// bf16 FLOPs per algorithmic fp32 FLOP, with the same operand traffic per MFMA.  Six points need 6 x 32 accumulator
// registers per 64 quads x 32 couts, more than a wave has when two waves share a SIMD, so this kernel runs ONE wave
// per SIMD (256 threads, up to 512 VGPRs) and every wave both produces (gather, input transform, split, LDS fill:
// VALU work that issues in the shadow of the wave's own MFMAs) and consumes.
//
// Algebra (Lavin & Gray, points 0, +-1, +-2, inf).  For the output quad (x0 .. x0+3) of a row and every (ky, c):
//     d_i = in[y+ky-1][x0-1+i][c], i = 0..5
//     V0 = 4 d0 - 5 d2 + d4            U0 = g0 / 4
//     V1 = -4 (d1 + d2) + (d3 + d4)    U1 = -(g0 + g1 + g2) / 6
//     V2 =  4 (d1 - d2) - (d3 - d4)    U2 = -(g0 - g1 + g2) / 6
//     V3 = -2 (d1 - d3) - (d2 - d4)    U3 = g0 / 24 + g1 / 12 + g2 / 6
//     V4 =  2 (d1 - d3) - (d2 - d4)    U4 = g0 / 24 - g1 / 12 + g2 / 6
//     V5 = 4 d1 - 5 d3 + d5            U5 = g2
//     M_xi[quad][o] = sum_{ky,c} V_xi U_xi
//     out[x0]   = M0 + M1 + M2 + M3 + M4          out[x0+1] = (M1 - M2) + 2 (M3 - M4)
//     out[x0+2] = (M1 + M2) + 4 (M3 + M4)         out[x0+3] = (M1 - M2) + 8 (M3 - M4) + M5
// U is transformed in float64 on the host and rounded once.  fp32 error against an fp64 convolution: about 3x that
// of F(2,3) / of a direct fp32 fma chain (tests/test_split_arith_cpu.py restates it in numpy), well inside the
// bound tests/test_conv_gpu.py holds every split kernel to (1e-6 of |x| conv |w| elementwise, 1.5e-7 rms).
//
// Block = 256 threads = 4 waves, persistent (one per CU).  Tile = 64 quads (256 pixels: two M-tiles of 32 quads) x
// 128 couts; wave wn owns both M-tiles x couts [32 wn, 32 wn + 32) x 6 points = 192 accumulator VGPRs.  K-step = one
// (16-channel group, ky) = 72 MFMAs (v_mfma_f32_32x32x16_bf16) per wave.  A operands: LDS, same conflict-free layout
// as conv_wsplit.hip, As[buf][xi][piece][M-tile][k half][32 quads x 8 ch], double buffered (2 x 36 KB); thread
// (quad, channel quad) fills its 6 x 3 eight-byte slots per K-step.  B operands: pre-transformed, pre-split weights
// in MFMA order [16-ch group][ky][32-cout tile][xi][piece][lane][8], straight from L2 into registers one K-step ahead.
// One barrier per K-step: step k+1 is produced into the other LDS buffer while step k is consumed.
// Needs W % 4 == 0 (quads do not straddle rows), Cin % 32 == 0 (even K-step count), Cout > 64; the fused 2x2
// max-pool (POOL = 1) additionally needs even H and W % 64 == 0 (M-tile = 2 rows x 64 columns, the two rows of a
// pooling window are accumulator registers r and r + 8 of one lane).  Everything else stays on conv_wsplit.hip.
#include "split_common.h"
#include <cmath>
#include <algorithm>
#include <vector>

struct W4Params {
  const float* in;
  const unsigned short* wgt;  // [Cin/16][3][Cout_pad/32][6 xi][3 pieces][64 lanes][8] bf16
  float* out;
  const float* pre_a;
  const float* pre_b;
  const float* post_a;
  const float* post_b;
  int H, W, Cin, in_cs, in_co;
  int Cout, Cout_pad, out_cs, out_co;
  int relu;
  int nsteps;  // 3 * Cin / 16 (even)
  int Mtotal;  // pixels
  int total_mtiles;  // 32-quad M-tiles
  float* pool_out;
  int pool_cs, pool_co, write_full, tiles_per_row;
  int total_tiles;
  unsigned* amax_out;
  unsigned* amax_pool;
};

namespace {

constexpr int KH_STRIDE = 256;               // ushorts: 32 rows x 8 channels
constexpr int PLANE = 2 * 2 * KH_STRIDE;     // one (xi, piece) plane: 2 M-tiles x 2 k halves
constexpr int BUF = 6 * 3 * PLANE;           // one K-step: 36 KB
constexpr int LDS_BYTES = 2 * BUF * 2;       // 72 KB

// first pixel of M-tile `mt` (flattened (n, y, x) index); POOL: the M-tile is 2 rows x 64 columns
template <int POOL>
__device__ __forceinline__ long w4_mtile_pm0(const W4Params& p, int mt, int& y0, int& x0) {
  if constexpr (POOL) {
    const int rp_lin = mt / p.tiles_per_row, cb = mt - rp_lin * p.tiles_per_row;
    const int hh = p.H >> 1;
    const int nimg = rp_lin / hh, rp = rp_lin - nimg * hh;
    y0 = 2 * rp;
    x0 = cb * 64;
    return ((long)nimg * p.H + y0) * p.W + x0;
  } else {
    y0 = x0 = 0;
    return (long)mt * 128;
  }
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t w4_rsrc(const float* base, unsigned bytes) {
  const unsigned long long bb = (unsigned long long)base;
  const unsigned long long bbu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(bb >> 32)) << 32) |
                                 (unsigned)__builtin_amdgcn_readfirstlane((int)bb);
  return __builtin_amdgcn_make_buffer_rsrc((void*)bbu, 0, (int)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}

}  // namespace

template <int POOL>
__global__ __launch_bounds__(256) void conv_w43_kernel(W4Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned short As[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave = cout sub-tile
  const int l31 = lane & 31, l5 = lane >> 5;
  const int nblk_n = p.Cout_pad >> 7;
  const int total = p.total_tiles;
  const int ns = p.nsteps;
  const int G = gridDim.x;
  constexpr unsigned OOB = 0x80000000u;

  // ------------------------------------------------------------------------------------------------
  // producer state: this thread's gather item = (quad qi of the 64-quad tile, channel quad q4)
  // ------------------------------------------------------------------------------------------------
  const int qi = tid >> 2, q4 = tid & 3;
  const int ldst = ((qi >> 5) * 2 + (q4 >> 1)) * KH_STRIDE + ((((qi & 31) * 8) ^ ((q4 >> 1) * 32)) + (q4 & 1) * 4);
  int L_ld = blockIdx.x, ld_ky = 0, ld_cg = 0;  // position of the NEXT K-step to load
  unsigned goff[6];
  int gy = 0;
  bool gok = false;
  __amdgpu_buffer_rsrc_t rsrc_in;
  auto tile_geometry = [&]() __attribute__((always_inline)) {
    const int tile = kocr_xcd_remap(L_ld < total ? L_ld : 0, total);
    const int mtb = (tile / nblk_n) * 2;  // first of the tile's two M-tiles
    int y0a, x0a, y0b, x0b;
    const long pm_a = w4_mtile_pm0<POOL>(p, mtb, y0a, x0a);
    // resource based one image row + one pixel before the tile's first pixel: every valid offset is >= 0
    rsrc_in = w4_rsrc(p.in + (pm_a * p.in_cs + p.in_co) - (long)(p.W + 1) * p.in_cs, 0x80000000u);
    const int m = qi >> 5, i = qi & 31;
    const int mt = mtb + m;
    const long pm = m ? w4_mtile_pm0<POOL>(p, mt < p.total_mtiles ? mt : mtb, y0b, x0b) : pm_a;
    int rel, x0;
    if constexpr (POOL) {
      const int row = i >> 4, qc = i & 15;
      rel = (int)(pm - pm_a) + row * p.W + 4 * qc;
      x0 = (m ? x0b : x0a) + 4 * qc;
      gy = (m ? y0b : y0a) + row;
      gok = L_ld < total && mt < p.total_mtiles;
    } else {
      rel = 4 * qi;
      const long g = pm_a + rel;
      gok = L_ld < total && g < p.Mtotal;
      x0 = (int)(g % p.W);
      gy = (int)((g / p.W) % p.H);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const bool pad = (k == 0 && x0 == 0) || (k == 5 && x0 + 4 >= p.W);  // column zero padding
      goff[k] = pad ? OOB : (unsigned)(((rel + k) * p.in_cs + q4 * 4) * 4);
    }
  };
  auto load_raw = [&](v4f (&raw)[6]) __attribute__((always_inline)) {
    const int soff = (ld_ky * p.W * p.in_cs + ld_cg * 16) * 4;
    const bool ok = gok && (unsigned)(gy + ld_ky - 1) < (unsigned)p.H;
#pragma unroll
    for (int k = 0; k < 6; ++k)
      raw[k] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, ok ? goff[k] : OOB, soff, 0));
    if (++ld_ky == 3) {
      ld_ky = 0;
      if (++ld_cg == (p.Cin >> 4)) {  // next tile
        ld_cg = 0;
        L_ld += G;
        tile_geometry();
      }
    }
  };
  // input transform of point xi (fp32, fixed operation order), split, 3 x 8 bytes into LDS
  auto produce_point = [&](const v4f (&d)[6], unsigned short* bufp, int xi) __attribute__((always_inline)) {
    v4f V;
    switch (xi) {
      case 0: V = (4.f * d[0] - 5.f * d[2]) + d[4]; break;
      case 1: V = (d[3] + d[4]) - 4.f * (d[1] + d[2]); break;
      case 2: V = 4.f * (d[1] - d[2]) - (d[3] - d[4]); break;
      case 3: V = -2.f * (d[1] - d[3]) - (d[2] - d[4]); break;
      case 4: V = 2.f * (d[1] - d[3]) - (d[2] - d[4]); break;
      default: V = (4.f * d[1] - 5.f * d[3]) + d[5]; break;
    }
    u2v h, m, l;
    kocr_split4(V, h, m, l);
    unsigned short* dst = bufp + xi * 3 * PLANE + ldst;
    *reinterpret_cast<u2v*>(dst) = h;
    *reinterpret_cast<u2v*>(dst + PLANE) = m;
    *reinterpret_cast<u2v*>(dst + 2 * PLANE) = l;
  };

  // ------------------------------------------------------------------------------------------------
  // consumer state
  // ------------------------------------------------------------------------------------------------
  const int ntiles32 = p.Cout_pad >> 5;
  const size_t w_step = (size_t)ntiles32 * 18 * 64 * 8;  // ushorts per K-step
  auto w_tile = [&](int nt) { return p.wgt + ((size_t)(nt * 4 + wn) * 18 * 64 + lane) * 8; };
  bf8 bw[6][3];
  f16v acc[6][2];
  const int a_lane = l5 * KH_STRIDE + ((l31 * 8) ^ (l5 * 32));
  auto load_a = [&](bf8 (&a)[2][3], const unsigned short* bufp, int xi) __attribute__((always_inline)) {
    const unsigned short* base = bufp + xi * 3 * PLANE + a_lane;
#pragma unroll
    for (int s = 2; s >= 0; --s)
#pragma unroll
      for (int m = 0; m < 2; ++m) a[m][s] = *reinterpret_cast<const bf8*>(base + s * PLANE + m * 2 * KH_STRIDE);
  };
  auto mfma12 = [&](const bf8 (&a)[2][3], int xi) __attribute__((always_inline)) {
    const bf8 b0 = bw[xi][0], b1 = bw[xi][1], b2 = bw[xi][2];
    // smallest terms first; the two M-tiles alternate so consecutive MFMAs are independent
#pragma unroll
    for (int m = 0; m < 2; ++m) acc[xi][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][2], b0, acc[xi][m], 0, 0, 0);
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
  };
  // One K-step: consume `bufc` (6 points x 12 MFMAs) while producing the NEXT step from `raw` into `bufn`; the
  // weights of a point are re-fetched (next K-step) as soon as its MFMAs are issued.  a0 holds point 0 on entry.
  bf8 a0[2][3], a1[2][3];
  auto step = [&](const unsigned short* bufc, unsigned short* bufn, const v4f (&raw)[6],
                  const unsigned short* w_next) __attribute__((always_inline)) {
    auto load_b = [&](int xi) __attribute__((always_inline)) {
#pragma unroll
      for (int s = 0; s < 3; ++s) bw[xi][s] = *reinterpret_cast<const bf8*>(w_next + (size_t)(xi * 3 + s) * 64 * 8);
    };
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      load_a(a1, bufc, 2 * q + 1);
      produce_point(raw, bufn, 2 * q);
      mfma12(a0, 2 * q);
      load_b(2 * q);
      if (q < 2) load_a(a0, bufc, 2 * q + 2);
      produce_point(raw, bufn, 2 * q + 1);
      mfma12(a1, 2 * q + 1);
      load_b(2 * q + 1);
    }
  };

  // ------------------------------------------------------------------------------------------------
  // pipeline prologue
  // ------------------------------------------------------------------------------------------------
  const int my_tiles = (total - (int)blockIdx.x + G - 1) / G;
  (void)my_tiles;
  tile_geometry();
  v4f rawA[6], rawB[6];
  load_raw(rawA);  // global step 0
  load_raw(rawB);  // global step 1
  {
    const int t0 = kocr_xcd_remap(blockIdx.x, total);
    const unsigned short* w0 = w_tile(t0 % nblk_n);
#pragma unroll
    for (int xi = 0; xi < 6; ++xi)
#pragma unroll
      for (int s = 0; s < 3; ++s) bw[xi][s] = *reinterpret_cast<const bf8*>(w0 + (size_t)(xi * 3 + s) * 64 * 8);
  }
#pragma unroll
  for (int xi = 0; xi < 6; ++xi) produce_point(rawA, As, xi);
  load_raw(rawA);  // global step 2
  __syncthreads();

  for (int L = blockIdx.x; L < total; L += G) {
    const int tile = kocr_xcd_remap(L, total);
    const int mtb = (tile / nblk_n) * 2, nt = tile % nblk_n;
    const unsigned short* w_ptr = w_tile(nt);
    const unsigned short* w_after = (L + G < total) ? w_tile(kocr_xcd_remap(L + G, total) % nblk_n) : w_ptr;
#pragma unroll
    for (int x = 0; x < 6; ++x)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][m][r] = 0.f;
    // drain the previous tile's stores once here (their unknown count must not merge into the K loop's waits)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    load_a(a0, As, 0);
    for (int s = 0; s < ns; s += 2) {
      // even global step: consume buffer 0, produce the odd step (rawB) into buffer 1, then refill rawB (step + 3)
      step(As, As + BUF, rawB, w_ptr + (size_t)(s + 1) * w_step);
      load_raw(rawB);
      __syncthreads();
      load_a(a0, As + BUF, 0);
      // odd global step: consume buffer 1, produce the next even step (rawA; possibly the next tile's first) into 0
      step(As + BUF, As, rawA, s + 2 < ns ? w_ptr + (size_t)(s + 2) * w_step : w_after);
      load_raw(rawA);
      __syncthreads();
      if (s + 2 < ns) load_a(a0, As, 0);
    }

    // ---- epilogue: 32x32 C/D map: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) ----------------
    {
      const int n = (nt * 4 + wn) * 32 + l31;
      const int nc = n < p.Cout ? n : p.Cout - 1;
      const float pa = p.pre_a[nc], pb = p.pre_b[nc];
      const bool has_post = p.post_a != nullptr;
      const float qa = has_post ? p.post_a[nc] : 1.f, qb = has_post ? p.post_b[nc] : 0.f;
      const bool live = n < p.Cout;
      auto act = [&](float v) {
        v = v * pa + pb;
        if (p.relu) v = fmaxf(v, 0.f);
        if (has_post) v = v * qa + qb;
        return v;
      };
      // inverse transform + BN + ReLU in place: acc[0..3][m][r] become the quad's four outputs
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float m0 = acc[0][m][r], m1 = acc[1][m][r], m2 = acc[2][m][r], m3 = acc[3][m][r], m4 = acc[4][m][r],
                      m5 = acc[5][m][r];
          const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
          acc[0][m][r] = act((m0 + s12) + s34);
          acc[1][m][r] = act(d12 + 2.f * d34);
          acc[2][m][r] = act(s12 + 4.f * s34);
          acc[3][m][r] = act((d12 + 8.f * d34) + m5);
        }
      const int ocs4 = p.out_cs * 4;
      if (p.amax_out || p.amax_pool) {
        float mx = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fabsf(acc[j][m][r]));
        mx = live ? mx : 0.f;
        if (p.amax_out) kocr_amax_update(p.amax_out, mx);
        if (p.amax_pool) kocr_amax_update(p.amax_pool, mx);
      }
      if constexpr (POOL) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int mt = mtb + m;
          int y0, x0;
          const long pm = w4_mtile_pm0<POOL>(p, mt < p.total_mtiles ? mt : mtb, y0, x0);
          const bool mlive = live && mt < p.total_mtiles;
          if (p.write_full) {
            const __amdgpu_buffer_rsrc_t ro = w4_rsrc(p.out + (pm * p.out_cs + p.out_co), 0x7FFFFFFFu);
            const unsigned vo = mlive ? (unsigned)((16 * l5 * p.out_cs + n) * 4) : OOB;  // 4 quads = 16 px per l5
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              const int px = 4 * ((r & 3) + 8 * (r >> 2));  // quad column (r&3) + 8 (r>>2) [+ 4 l5] of row y
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[j][m][r]), ro, vo, (px + j) * ocs4, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[j][m][r + 8]), ro, vo, (px + j + p.W) * ocs4, 0);
              }
            }
          }
          // 2x2 max: rows y (r) and y+1 (r+8), columns (0,1) and (2,3) of the quad
          const long nimg = pm / ((long)p.H * p.W);
          const long pp0 = (nimg * (p.H >> 1) + (y0 >> 1)) * (p.W >> 1) + (x0 >> 1);
          const __amdgpu_buffer_rsrc_t rp = w4_rsrc(p.pool_out + (pp0 * p.pool_cs + p.pool_co), 0x7FFFFFFFu);
          const unsigned vp = mlive ? (unsigned)((8 * l5 * p.pool_cs + n) * 4) : OOB;  // 4 quads = 8 pooled px per l5
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const int pq = 2 * ((r & 3) + 8 * (r >> 2));
            const float v0 = fmaxf(fmaxf(acc[0][m][r], acc[1][m][r]), fmaxf(acc[0][m][r + 8], acc[1][m][r + 8]));
            const float v1 = fmaxf(fmaxf(acc[2][m][r], acc[3][m][r]), fmaxf(acc[2][m][r + 8], acc[3][m][r + 8]));
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v0), rp, vp, pq * p.pool_cs * 4, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v1), rp, vp, (pq + 1) * p.pool_cs * 4, 0);
          }
        }
      } else {
        int yy, xx;
        const long pm0 = w4_mtile_pm0<POOL>(p, mtb, yy, xx);
        // bytes from the tile's first pixel to the end of the tensor: stores past it are dropped
        const long rem = ((long)p.Mtotal - pm0) * ocs4;
        const __amdgpu_buffer_rsrc_t ro =
            w4_rsrc(p.out + (pm0 * p.out_cs + p.out_co), rem < 0x7FFFFFFFL ? (unsigned)(rem > 0 ? rem : 0) : 0x7FFFFFFFu);
        const unsigned vo = live ? (unsigned)((16 * l5 * p.out_cs + n) * 4) : OOB;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int px = 4 * (m * 32 + (r & 3) + 8 * (r >> 2));  // + 16 l5 in vo
#pragma unroll
            for (int j = 0; j < 4; ++j)
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[j][m][r]), ro, vo, (px + j) * ocs4, 0);
          }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
int prepare_w43(kocr_ctx* ctx, ConvLayer& L, const float* w, bool w_is_oihw) {
  if (L.KH != 3 || L.KW != 3 || L.dil != 1 || L.Cin % 32 != 0 || L.Cout <= 64) return KOCR_OK;
  const int Cin = L.Cin, Cout = L.Cout;
  const int cp = (Cout + 127) / 128 * 128;
  const int nt32 = cp / 32;
  std::vector<unsigned short> u((size_t)(Cin / 16) * 3 * nt32 * 18 * 64 * 8, 0);
  for (int c = 0; c < Cin; ++c)
    for (int ky = 0; ky < 3; ++ky)
      for (int o = 0; o < Cout; ++o) {
        double g[3];
        for (int kx = 0; kx < 3; ++kx)
          g[kx] = w_is_oihw ? w[(((size_t)o * Cin + c) * 3 + ky) * 3 + kx] : w[(((size_t)ky * 3 + kx) * Cin + c) * Cout + o];
        // G g in float64, rounded once to fp32
        const float U[6] = {(float)(g[0] / 4.0),
                            (float)(-(g[0] + g[1] + g[2]) / 6.0),
                            (float)(-(g[0] - g[1] + g[2]) / 6.0),
                            (float)(g[0] / 24.0 + g[1] / 12.0 + g[2] / 6.0),
                            (float)(g[0] / 24.0 - g[1] / 12.0 + g[2] / 6.0),
                            (float)g[2]};
        // MFMA 32x32x16 B operand: lane = (k >> 3) * 32 + (o & 31) holds k = 8 (lane >> 5) + j, j = 0..7
        const int k = c % 16, lane = (k >> 3) * 32 + (o & 31), j = k & 7;
        const size_t step = (size_t)(c / 16) * 3 + ky;
        for (int xi = 0; xi < 6; ++xi) {
          unsigned short pc[3];
          kocr_split3_host(U[xi], pc);
          for (int s = 0; s < 3; ++s) u[((((step * nt32 + o / 32) * 6 + xi) * 3 + s) * 64 + lane) * 8 + j] = pc[s];
        }
      }
  L.w4_cout_pad = cp;
  void* d = nullptr;
  KOCR_TRY(ctx->dev_alloc(&d, u.size() * sizeof(unsigned short)));
  KOCR_HIP(ctx, hipMemcpy(d, u.data(), u.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
  L.d_w4 = (unsigned short*)d;
  return KOCR_OK;
}

bool w43_applicable(const kocr_ctx* ctx, const ConvLayer& L, const Tensor& in) {
  static const bool off = getenv("KOCR_W43") && atoi(getenv("KOCR_W43")) == 0;
  return !off && ctx->split_mode == KOCR_SPLIT_BF16X3 && L.d_w4 && in.W % 4 == 0 && in.cs % 4 == 0 && in.co % 4 == 0 &&
         ((uintptr_t)in.p & 15) == 0 && L.Cin % 32 == 0;
}

template <int POOL>
static int w4_launch(kocr_ctx* ctx, W4Params& p) {
  static bool attr_done[64] = {};  // per device: one process may hold contexts on several GPUs
  const int dev = ctx->device & 63;
  if (!attr_done[dev]) {
    KOCR_HIP(ctx, hipFuncSetAttribute((const void*)conv_w43_kernel<POOL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr_done[dev] = true;
  }
  static int n_cus[64] = {};
  if (!n_cus[dev]) {
    hipDeviceProp_t prop;
    KOCR_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
    n_cus[dev] = prop.multiProcessorCount;
  }
  const int grid = p.total_tiles < n_cus[dev] ? p.total_tiles : n_cus[dev];  // persistent: one block per CU
  hipLaunchKernelGGL((conv_w43_kernel<POOL>), dim3(grid), dim3(256), LDS_BYTES, ctx->stream, p);
  KOCR_HIP(ctx, hipGetLastError());
  return KOCR_OK;
}

int launch_conv_w43(kocr_ctx* ctx, const ConvLayer& L, const Tensor& in, const Tensor& out, const Tensor* pool, bool need_full) {
  const bool fuse = pool && in.H % 2 == 0 && in.W % 64 == 0;
  const size_t M = in.pixels();
  W4Params p;
  p.in = in.p;
  p.wgt = L.d_w4;
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
  p.Cout_pad = L.w4_cout_pad;
  p.out_cs = out.cs;
  p.out_co = out.co;
  p.relu = L.relu;
  p.nsteps = 3 * (L.Cin / 16);
  p.Mtotal = (int)M;
  p.total_mtiles = (int)((M + 127) / 128);  // fuse: M % 128 == 0 (two rows x 64 columns)
  p.pool_out = nullptr;
  p.pool_cs = p.pool_co = p.write_full = p.tiles_per_row = 0;
  p.amax_out = out.amax;
  p.amax_pool = fuse ? pool->amax : nullptr;
  if (fuse) {
    p.pool_out = pool->p;
    p.pool_cs = pool->cs;
    p.pool_co = pool->co;
    p.write_full = need_full ? 1 : 0;
    p.tiles_per_row = in.W / 64;
  }
  p.total_tiles = ((p.total_mtiles + 1) / 2) * (p.Cout_pad / 128);
  static const bool per_layer = getenv("KOCR_PROF_LAYERS") != nullptr;
  char nm[64];
  if (per_layer)
    snprintf(nm, sizeof nm, "conv_w4s_256x128%s:%s", fuse ? "p" : "", L.name.c_str());
  else
    snprintf(nm, sizeof nm, "conv_w4s_256x128%s", fuse ? "_pool" : "");
  const double flops = 2.0 * (double)M * L.Kreal * L.Cout;  // algorithmic (direct-convolution) FLOPs
  const double bytes = 4.0 * ((double)M * L.Cin + (double)M * L.Cout + (double)L.Kreal * L.Cout);
  {
    ProfScope ps(ctx, nm, flops, bytes);
    if (fuse)
      KOCR_TRY(w4_launch<1>(ctx, p));
    else
      KOCR_TRY(w4_launch<0>(ctx, p));
  }
  if (pool && !fuse) return launch_maxpool2x2(ctx, out, *pool);
  return KOCR_OK;
}
