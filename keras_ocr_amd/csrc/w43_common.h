// w43_common.h -- launch parameters, tile-index helpers and the F(4,3) constants shared by the Winograd F(4,3) kernels
// (conv_w43.hip: bf16x3 arithmetic; conv_w43h.hip: fp16 arithmetic with per-image power-of-two scaling).
#pragma once
#include "split_common.h"
#include <type_traits>
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
  int n_mpairs;   // pixel tiles (pairs of M-tiles)
  int dil;        // dilation d (DIL kernels: taps d apart; a quad = 4 outputs d apart in one residue class mod d)
  int qpr;        // quads per image row = W / 4
  int m_fastest;  // tile order: 1 = consecutive tiles share the cout block (weights stay in the XCD's L2)
  unsigned* amax_out;   // per-image max-|x| slots of the output / pooled output (Tensor::amax[n]) or nullptr
  unsigned* amax_pool;
  // exact division of tile indices (< 2^31) by launch constants without the ~40-instruction integer division sequence:
  // q = (t + ((n - t) >> 1)) >> sh with t = mulhi(n, mul)  (Granlund-Montgomery); [0] = multiplier, [1] = shift
  unsigned dv_tpr[2];  // by tiles_per_row
  unsigned dv_hh[2];   // by the row blocks per image: H / 2 (2-row M-tiles), H / 8 (conv_w43v GEO 2)
  unsigned dv_mp[2];   // by n_mpairs
  unsigned dv_nb[2];   // by the number of cout blocks
  unsigned dv_hw[2];   // by H * W (pixel index -> image, for the per-image max-|x| slots)
  const unsigned* amax_in;  // fp16 kernels: per-image max-|x| slots of the INPUT (Tensor::amax[n]), never null there
  int Wv;              // conv_w43fh_kernel: valid output width (Tensor::Wv): columns >= Wv are written as zeros; 0 = all valid
  unsigned dv_w[2];    // by W
  // conv_w43vh / conv_w43rh MODE 1 (ragged images) and MODE 2 (cell grids): the tile grid of an image
  int rq_per_img;      // row blocks per image: ceil(H / 4) (4 x 64 tiles) or ceil(H / 8) (8 x 32 tiles)
  unsigned dv_rq[2];   // by rq_per_img
  unsigned dv_qpr[2], dv_dil[2], dv_h[2];  // conv_w43fh_kernel<.., DIL>: by qpr (quads per row), by dil, by H
  int cellW, cellWv, cells_per_row;  // MODE 2 (Tensor::cellW / cellWv): cell pitch, valid columns of a cell, W / cellW
  unsigned dv_wc[2];   // by cellW
};

// host: multiplier / shift of the division by d (1 <= d < 2^31)
static inline void w4_div_magic(unsigned d, unsigned (&out)[2]) {
  if (d <= 1) {
    out[0] = 0;
    out[1] = 0;  // q = (0 + (n >> 1)) >> 0 would be wrong: d == 1 is special-cased on the device through sh == 0 && mul == 0
    return;
  }
  unsigned L = 0;
  while ((1ull << L) < d) ++L;  // ceil(log2 d), >= 1
  out[0] = (unsigned)(((1ull << 32) * ((1ull << L) - d)) / d + 1);
  out[1] = L - 1;
}

namespace {

__device__ __forceinline__ unsigned w4_fdiv(unsigned n, const unsigned (&dv)[2]) {
  if (dv[0] == 0 && dv[1] == 0) return n;  // d == 1 (wave-uniform)
  const unsigned t = __umulhi(n, dv[0]);
  return (t + ((n - t) >> 1)) >> dv[1];
}

// interpolation points 0, +-a, +-b, inf (all constants exact in fp32)
constexpr double W4_PA = 0.625, W4_PB = 1.5;
constexpr float W4_A = (float)W4_PA, W4_B = (float)W4_PB;
constexpr float W4_A2 = (float)(W4_PA * W4_PA), W4_B2 = (float)(W4_PB * W4_PB);
constexpr float W4_A3 = (float)(W4_PA * W4_PA * W4_PA), W4_B3 = (float)(W4_PB * W4_PB * W4_PB);
constexpr float W4_A2B2 = (float)(W4_PA * W4_PA * W4_PB * W4_PB), W4_A2PB2 = (float)(W4_PA * W4_PA + W4_PB * W4_PB);

constexpr int KH_STRIDE = 256;               // ushorts: 32 rows x 8 channels
constexpr int PLANE = 2 * 2 * KH_STRIDE;     // one (xi, piece) plane: 2 M-tiles x 2 k halves
constexpr int BUF = 6 * 3 * PLANE;           // one K-step: 36 KB
constexpr int LDS_BYTES = 2 * BUF * 2;       // 72 KB

// first pixel of M-tile `mt` (flattened (n, y, x) index); POOL: the M-tile is 2 rows x 64 columns
template <int POOL>
__device__ __forceinline__ long w4_mtile_pm0(const W4Params& p, int mt, int& y0, int& x0) {
  if constexpr (POOL) {
    const int rp_lin = (int)w4_fdiv((unsigned)mt, p.dv_tpr), cb = mt - rp_lin * p.tiles_per_row;
    const int hh = p.H >> 1;
    const int nimg = (int)w4_fdiv((unsigned)rp_lin, p.dv_hh), rp = rp_lin - nimg * hh;
    y0 = 2 * rp;
    x0 = cb * 64;
    return ((long)nimg * p.H + y0) * p.W + x0;
  } else {
    y0 = x0 = 0;
    return (long)mt * 128;
  }
}

// tile index -> (pixel tile, cout block)
__device__ __forceinline__ void w4_decode(const W4Params& p, int tile, int nblk_n, int& mp, int& nt) {
  // selects, not an if / else over the two assignment orders: hipcc lowers that to a two-element stack array written at a
  // dynamic index and read back -- a scratch store + load with a full memory round trip in every tile of every F(4,3) kernel
  const bool mf = p.m_fastest != 0;
  const unsigned dv[2] = {mf ? p.dv_mp[0] : p.dv_nb[0], mf ? p.dv_mp[1] : p.dv_nb[1]};
  const int quot = (int)w4_fdiv((unsigned)tile, dv);
  const int rem = tile - quot * (mf ? p.n_mpairs : nblk_n);
  nt = mf ? quot : rem;
  mp = mf ? rem : quot;
}

// The epilogue's per-channel coefficients [pre_a | pre_b | post_a | post_b][Cout_pad] staged in LDS once per block (the
// padding channels repeat the last one; post = identity without one): fetched from global memory in every tile's epilogue
// they cost a persistent block one full memory latency per tile.  The caller's next block barrier publishes them.
__device__ __forceinline__ void w4_stage_coef(const W4Params& p, float* coef, int tid) {
  for (int i = tid; i < p.Cout_pad; i += 256) {
    const int c = i < p.Cout ? i : p.Cout - 1;
    coef[i] = p.pre_a[c];
    coef[p.Cout_pad + i] = p.pre_b[c];
    coef[2 * p.Cout_pad + i] = p.post_a ? p.post_a[c] : 1.f;
    coef[3 * p.Cout_pad + i] = p.post_a ? p.post_b[c] : 0.f;
  }
}
constexpr int W4_COEF_BYTES_MAX = 4 * 1024 * 4;  // what the launchers reserve: Cout_pad <= 1024

__device__ __forceinline__ __amdgpu_buffer_rsrc_t w4_rsrc(const float* base, unsigned bytes) {
  const unsigned long long bb = (unsigned long long)base;
  const unsigned long long bbu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(bb >> 32)) << 32) |
                                 (unsigned)__builtin_amdgcn_readfirstlane((int)bb);
  return __builtin_amdgcn_make_buffer_rsrc((void*)bbu, 0, (int)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}

}  // namespace

// input transform of point xi (fp32, fixed operation order; T = v4f or v2f)
template <class T>
__device__ __forceinline__ T w4_transform(const T (&d)[6], int xi) {
  switch (xi) {
    case 0: return (W4_A2B2 * d[0] - W4_A2PB2 * d[2]) + d[4];
    case 1: return (d[4] - W4_B2 * d[2]) + W4_A * (d[3] - W4_B2 * d[1]);
    case 2: return (d[4] - W4_B2 * d[2]) - W4_A * (d[3] - W4_B2 * d[1]);
    case 3: return (d[4] - W4_A2 * d[2]) + W4_B * (d[3] - W4_A2 * d[1]);
    case 4: return (d[4] - W4_A2 * d[2]) - W4_B * (d[3] - W4_A2 * d[1]);
    default: return (W4_A2B2 * d[1] - W4_A2PB2 * d[3]) + d[5];
  }
}
