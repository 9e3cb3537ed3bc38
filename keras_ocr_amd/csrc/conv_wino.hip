// conv_wino.hip — 3x3 / stride 1 / dilation 1 convolution with 1-D Winograd F(2,3) along image
// rows, on the gfx950 matrix cores (fp32 MFMA 16x16x4).
//
// Same operator as conv_mfma.hip (keras Conv2D 3x3 'same' + folded BN + ReLU, detection.py:87-103)
// with 2/3 of the matrix-core work: for an output pair (x0, x0+1) of row y and every (ky, c)
//     d_i = in[y+ky-1][x0-1+i][c],  i = 0..3            (zero outside the image)
//     m0 = (d0-d2) g0,  m1 = (d1+d2)(g0+g1+g2)/2,  m2 = (d2-d1)(g0-g1+g2)/2,  m3 = (d1-d3) g2
//     out[x0] = sum(m0+m1+m2),  out[x0+1] = sum(m1-m2-m3)
// i.e. four GEMMs  M_xi[pair][o] = sum_{ky,c} V_xi[pair][(ky,c)] * U_xi[(ky,c)][o]  instead of the
// direct [pixel] x [9*Cin] x [Cout] one: 4 multiplies per 2 outputs instead of 6 per (ky,c,o).
// The transform coefficients are +-1 and +-1/2, so the fp32 error stays at round-off level
// (tests bound it with the same tolerance as the direct kernel).
//
// Block = 256 threads = 4 waves; tile = 128 consecutive pixels of one image row (64 pairs) x 64
// output channels; K-step = one (16-channel group, ky) = 16 k.  The RAW input row piece (130 px x
// 16 ch) is staged k-major in LDS; every wave reads two aligned 8-B pieces (d0,d1),(d2,d3) per pair
// and forms the four V_xi values in registers (the transformed tile is never stored).  Wave w owns
// output channels [16w,16w+16) for ALL four points, so the inverse transform in the epilogue is
// register-only: the 16x16x4 MFMA C/D map puts M_0..M_3 of one (pair, channel) in the same lane.
// U_xi is pre-transformed on the host and packed in MFMA-operand order, [16-channel group][ky]
// [16-cout tile][lane][kq*4+xi], so a lane fetches the four B operands of a k-quad with one 16-B load
// straight into registers (4 KB contiguous per wave and K-step); the load for the NEXT K-step's
// k-quad is issued right after the MFMAs that consumed the current one (rolling prefetch, no second
// register set): the weights never touch LDS.  Requires W even (a pair never straddles an image row); a tile may cross rows
// and images: the flattened neighbours that fall across a row end are zeroed per pair in registers.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

struct WinoParams {
  const float* in;
  const float* wgt;  // U: [Cin/16][3][Cout_pad/16][64 lanes][16 = kq*4 + xi]
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
  // fused 2x2 max-pool (POOL kernel): tile = 2 image rows x 64 columns
  float* pool_out;
  int pool_cs, pool_co, write_full, tiles_per_row;
  unsigned* amax_out;   // Tensor::amax slots of the output / pooled output, or nullptr
  unsigned* amax_pool;
};

namespace {
constexpr int LDA = 160;  // floats per k row of the raw tile (130 used); 160*4 B = 128 mod 256 -> the two
                          // k rows a 32-lane ds_read_b64 group touches fall into different bank halves
}

__device__ __forceinline__ int wino_xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7;
  const int q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// NW waves per block, WC of them side by side along the output channels (16 couts each); the other
// NW / WC = WP groups split the tile's four 16-pair M-tiles: wave (wp, wc) owns M-tiles wp, wp+WP, ...
// <4,4>: 64 couts (default)   <8,8>: 128 couts   <4,2>: 32 couts   <4,1>: 16 couts (narrow head layers)
template <int POOL, int NW, int WC>
__global__ __launch_bounds__(64 * NW, (NW == 8) ? 4 : 2) void conv_wino_kernel(WinoParams p) {
  constexpr int NT = 64 * NW;
  constexpr int WP = NW / WC;   // pair groups
  constexpr int MT = 4 / WP;    // M-tiles (16 pairs each) per wave
  static_assert(WP * MT == 4 && (!POOL || WP <= 2), "tile split");
  __shared__ float As[2][16][LDA];        // raw input: As[buf][k][1 + pixel], pixel = -1 .. 128
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // SGPR
  const int l15 = lane & 15, l4 = lane >> 4;

  const int nblk_n = p.Cout_pad / (16 * WC);
  // compile-time zero when there is a single pair group, so the M-tile indices stay constants
  const int wc = (WC == NW) ? wave : wave % WC, wp = (WP == 1) ? 0 : wave / WC;
  const int tile = wino_xcd_remap(blockIdx.x, gridDim.x);
  const int mt = tile / nblk_n, nt = tile - mt * nblk_n;
  const int n0 = nt * 16 * WC;
  const int quad = tid & 3;
  constexpr int PPI = NT / 4;  // pixels covered by one gather item across the block
  // Tile -> pixels.  POOL == 0: 128 consecutive pixels of the flattened (n, y, x) order (64 pairs); LDS
  // position = 1 + pixel (pixel = -1 .. 128).  POOL == 1: rows (y, y+1) x 64 columns (32 pairs each);
  // LDS position = 80*row + 1 + pixel (pixel = -1 .. 64), so M-tiles 0,1 are row y and 2,3 row y+1.
  long pm0;
  int lpos[3], goff[3], yy[3];
  bool ex[3];
  unsigned lz = 0, rz = 0;
  int y0t = 0, x0t = 0;
  if constexpr (POOL) {
    const int rp_lin = mt / p.tiles_per_row, cb = mt - rp_lin * p.tiles_per_row;
    const int hh = p.H >> 1;
    const int nimg = rp_lin / hh, rp = rp_lin - nimg * hh;
    y0t = 2 * rp;
    x0t = cb * 64;
    pm0 = ((long)nimg * p.H + y0t) * p.W + x0t;
    // 2 rows x 66 positions: NW == 4: items {row 0, row 1} by all threads + 16 stragglers;
    // NW == 8: one item per thread (row = tid >> 8) + 16 stragglers
    const int pidx[3] = {(tid >> 2) & 63, (tid >> 2) & 63, 64 + ((tid >> 2) & 1)};
    const int prow[3] = {NW == 8 ? (tid >> 8) : 0, 1, (tid >> 3) & 1};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int x = x0t + pidx[j] - 1;
      ex[j] = ((j == 0) || (j == 1 && NW == 4) || (j == 2 && tid < 16)) && x >= 0 && x < p.W;
      yy[j] = y0t + prow[j];
      lpos[j] = prow[j] * 80 + pidx[j];
      goff[j] = (prow[j] * p.W + pidx[j] - 1) * p.in_cs + quad * 4;
    }
  } else {
    pm0 = (long)mt * 128;
    const int pidx[3] = {tid >> 2, (tid >> 2) + 64, 128 + ((tid >> 2) & 1)};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const long g = pm0 + pidx[j] - 1;
      ex[j] = ((j == 0) || (j == 1 && NW == 4) || (j == 2 && tid < 8)) && g >= 0 && g < p.Mtotal;
      yy[j] = ex[j] ? (int)((g / p.W) % p.H) : 0;
      lpos[j] = pidx[j];
      goff[j] = (pidx[j] - 1) * p.in_cs + quad * 4;
    }
    // pair t = 16*i + (lane & 15) at x0 = (pm0 + 2t) % W; its d0 (pixel x0-1) is zero padding when
    // x0 == 0, its d3 (pixel x0+2) when x0 + 2 == W (the flattened neighbour belongs to another row)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int x0 = (int)((pm0 + 2 * (i * 16 + l15)) % p.W);
      if (x0 == 0) lz |= 1u << i;
      if (x0 + 2 >= p.W) rz |= 1u << i;
    }
  }
  const bool has_c = POOL ? (tid < 16) : (tid < 8);
  (void)PPI;
  const float* blk_in = p.in + (pm0 * p.in_cs + p.in_co);
  // weights: this lane's 16 B-operand values of a K-step are 64 contiguous bytes
  const int ntiles16 = p.Cout_pad >> 4;
  const float* w_ptr = p.wgt + ((size_t)(nt * WC + wc) * 64 + lane) * 16;
  const size_t w_step = (size_t)ntiles16 * 64 * 16;  // floats per K-step

  v4f rr[3];
  v4f bw[4];  // B operands of the current K-step: [kq] -> .x/.y/.z/.w = xi 0..3
  const float* wp_next = nullptr;
  int st_ky = 0, st_cg = 0;  // position of the NEXT step to load
  auto load_step = [&]() __attribute__((always_inline)) {
    const int dy = st_ky - 1;
    const int soff = dy * p.W * p.in_cs + st_cg * 16;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (j == 1 && NW == 8) continue;
      const bool ok = ex[j] && (unsigned)(yy[j] + dy) < (unsigned)p.H;
      v4f v = *reinterpret_cast<const v4f*>(blk_in + (ok ? goff[j] + soff : 0));
      rr[j] = ok ? v : v4f{0.f, 0.f, 0.f, 0.f};
    }
    wp_next = w_ptr + (size_t)(st_cg * 3 + st_ky) * w_step;
    if (++st_ky == 3) {
      st_ky = 0;
      ++st_cg;
    }
  };
  auto store_step = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (j == 0 || (j == 1 && NW == 4) || (j == 2 && has_c)) {
        As[buf][quad * 4 + 0][lpos[j]] = rr[j].x;
        As[buf][quad * 4 + 1][lpos[j]] = rr[j].y;
        As[buf][quad * 4 + 2][lpos[j]] = rr[j].z;
        As[buf][quad * 4 + 3][lpos[j]] = rr[j].w;
      }
    }
  };

  f32x4 acc[4][MT];  // [xi][own m-tile]; own m-tile m <-> tile index i = wp + WP*m
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[x][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute_step = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
      const int k = kq * 4 + l4;  // 16x16x4: A[row = lane&15][k = lane>>4], B[k = lane>>4][col = lane&15]
      const v4f b = bw[kq];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int i = wp + WP * m;
        const int base = POOL ? (i >> 1) * 80 + 2 * ((i & 1) * 16 + l15) : 2 * (i * 16 + l15);
        const v2f d01 = *reinterpret_cast<const v2f*>(&As[buf][k][base]);      // pixels 2t-1, 2t
        const v2f d23 = *reinterpret_cast<const v2f*>(&As[buf][k][base + 2]);  // pixels 2t+1, 2t+2
        const float d0 = ((lz >> i) & 1u) ? 0.f : d01.x;  // row start: left neighbour is padding
        const float d3 = ((rz >> i) & 1u) ? 0.f : d23.y;  // row end: right neighbour is padding
        const float v0 = d0 - d23.x, v1 = d01.y + d23.x, v2 = d23.x - d01.y, v3 = d01.y - d3;
        acc[0][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(v0, b[0], acc[0][m], 0, 0, 0);
        acc[1][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(v1, b[1], acc[1][m], 0, 0, 0);
        acc[2][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(v2, b[2], acc[2][m], 0, 0, 0);
        acc[3][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(v3, b[3], acc[3][m], 0, 0, 0);
      }
      // this k-quad's operands are consumed: fetch the same k-quad of the next K-step into them
      bw[kq] = *reinterpret_cast<const v4f*>(wp_next + 4 * kq);
    }
  };

  load_step();
#pragma unroll
  for (int q = 0; q < 4; ++q) bw[q] = *reinterpret_cast<const v4f*>(wp_next + 4 * q);  // K-step 0
  store_step(0);
  __syncthreads();
  const int ns = p.nsteps;
  for (int s = 0; s + 1 < ns; ++s) {
    load_step();  // also points wp_next at K-step s+1
    __builtin_amdgcn_sched_barrier(0);
    compute_step(s & 1);
    __builtin_amdgcn_sched_barrier(0);
    store_step((s + 1) & 1);
    __syncthreads();
  }
  compute_step((ns - 1) & 1);

  // ---- epilogue: 16x16x4 C/D map: col = lane&15, row = (lane>>4)*4 + reg --------------------------
  const int n = n0 + wc * 16 + l15;
  float amx = 0.f;  // max |output| of this lane (Tensor::amax)
  if (n < p.Cout) {
    const float pa = p.pre_a[n], pb = p.pre_b[n];
    const bool has_post = p.post_a != nullptr;
    const float qa = has_post ? p.post_a[n] : 1.f, qb = has_post ? p.post_b[n] : 0.f;
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
    if constexpr (POOL) {
      const long nimg = pm0 / ((long)p.H * p.W);
      const long pp0 = (nimg * (p.H >> 1) + (y0t >> 1)) * (p.W >> 1) + (x0t >> 1);
#pragma unroll
      for (int mh = 0; mh < MT / 2; ++mh)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ih = (WP == 1) ? mh : wp;   // tile index inside the row (0 or 1)
          constexpr int MR = MT / 2;            // own m-tile of the same columns in row y+1
          const int pp = ih * 16 + l4 * 4 + r;  // pair inside the row == pooled column
          float a0, a1, b0, b1;
          finish(acc[0][mh][r], acc[1][mh][r], acc[2][mh][r], acc[3][mh][r], a0, a1);                          // row y
          finish(acc[0][mh + MR][r], acc[1][mh + MR][r], acc[2][mh + MR][r], acc[3][mh + MR][r], b0, b1);      // row y+1
          amx = fmaxf(amx, fmaxf(fmaxf(fabsf(a0), fabsf(a1)), fmaxf(fabsf(b0), fabsf(b1))));
          if (p.write_full) {
            float* o = p.out + ((pm0 + 2 * pp) * p.out_cs + p.out_co + n);
            o[0] = a0;
            o[p.out_cs] = a1;
            o[(long)p.W * p.out_cs] = b0;
            o[(long)p.W * p.out_cs + p.out_cs] = b1;
          }
          p.pool_out[(pp0 + pp) * p.pool_cs + p.pool_co + n] = fmaxf(fmaxf(a0, a1), fmaxf(b0, b1));
        }
    } else {
      // in place, then raw buffer stores back to back: one per-lane byte offset, the per-register pixel
      // offset in an SGPR, pixels past the end of the tensor dropped by the range check (a per-pair
      // `if (pixel < M)` made hipcc wait for vmcnt(0) before every store pair)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float o0, o1;
          finish(acc[0][m][r], acc[1][m][r], acc[2][m][r], acc[3][m][r], o0, o1);
          acc[0][m][r] = o0;
          acc[1][m][r] = o1;
          amx = fmaxf(amx, fmaxf(fabsf(o0), fabsf(o1)));
        }
      const int ocs4 = p.out_cs * 4;
      const long rem = ((long)p.Mtotal - pm0) * ocs4;
      const unsigned long long bb = (unsigned long long)(p.out + (pm0 * p.out_cs + p.out_co));
      const unsigned long long bbu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(bb >> 32)) << 32) |
                                     (unsigned)__builtin_amdgcn_readfirstlane((int)bb);
      const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
          (void*)bbu, 0, __builtin_amdgcn_readfirstlane((int)(rem < 0x7FFFFFFFL ? rem : 0x7FFFFFFFL)), 0x00020000);
      const unsigned vo = (unsigned)((8 * l4 * p.out_cs + n) * 4);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int px = 2 * ((wp + WP * m) * 16 + r);  // + 8*l4 in vo
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[0][m][r]), ro, vo, px * ocs4, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[1][m][r]), ro, vo, (px + 1) * ocs4, 0);
        }
    }
  }
  // all lanes take part in the wave reduction (padded couts contribute 0)
  if (p.amax_out) kocr_amax_update(p.amax_out, amx);
  if (p.amax_pool) kocr_amax_update(p.amax_pool, amx);
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
// g: the layer's 3x3 kernel, OIHW or HWIO.  Builds U[cg][ky][xi][16][Cout_pad] (Cout_pad multiple of 64).
int prepare_wino(kocr_ctx* ctx, ConvLayer& L, const float* w, bool w_is_oihw) {
  if (L.KH != 3 || L.KW != 3 || L.dil != 1 || L.Cin % 16 != 0) return KOCR_OK;
  const int Cin = L.Cin, Cout = L.Cout;
  const int wcls = Cout > 32 ? 64 : (Cout > 16 ? 32 : 16);  // couts per block: <4,4>, <4,2>, <4,1>
  const int cp = (Cout + wcls - 1) / wcls * wcls;
  const int nt16 = cp / 16;
  std::vector<float> u((size_t)(Cin / 16) * 3 * nt16 * 64 * 16, 0.f);
  for (int c = 0; c < Cin; ++c)
    for (int ky = 0; ky < 3; ++ky)
      for (int o = 0; o < Cout; ++o) {
        float g[3];
        for (int kx = 0; kx < 3; ++kx)
          g[kx] = w_is_oihw ? w[(((size_t)o * Cin + c) * 3 + ky) * 3 + kx] : w[(((size_t)ky * 3 + kx) * Cin + c) * Cout + o];
        const float U[4] = {g[0], 0.5f * ((g[0] + g[1]) + g[2]), 0.5f * ((g[0] - g[1]) + g[2]), g[2]};
        // MFMA 16x16x4 B operand: lane = (k & 3) * 16 + (o & 15) holds k-quad kq = (k >> 2) of the step
        const int k = c % 16, lane = (k & 3) * 16 + (o & 15), kq = k >> 2;
        for (int xi = 0; xi < 4; ++xi)
          u[((((size_t)(c / 16) * 3 + ky) * nt16 + o / 16) * 64 + lane) * 16 + kq * 4 + xi] = U[xi];
      }
  L.wino_cout_pad = cp;
  return ctx->upload(&L.d_wino, u);
}

bool wino_applicable(const ConvLayer& L, const Tensor& in) {
  static const bool off = getenv("KOCR_WINO") && atoi(getenv("KOCR_WINO")) == 0;
  return !off && L.d_wino && in.W % 2 == 0 && in.cs % 4 == 0 && in.co % 4 == 0 && ((uintptr_t)in.p & 15) == 0 &&
         L.Cin % 16 == 0;
}

int launch_conv_wino(kocr_ctx* ctx, const ConvLayer& L, const Tensor& in, const Tensor& out, const Tensor* pool,
                     bool need_full) {
  // fused pooling needs exact 2-row x 64-column tiles
  const bool fuse = pool && in.H % 2 == 0 && in.W % 64 == 0 && L.Cout > 16;
  const size_t M = in.pixels();
  WinoParams p;
  p.in = in.p;
  p.wgt = L.d_wino;
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
  p.Cout_pad = L.wino_cout_pad;
  p.out_cs = out.cs;
  p.out_co = out.co;
  p.relu = L.relu;
  p.nsteps = 3 * (L.Cin / 16);
  p.Mtotal = (int)M;
  p.pool_out = nullptr;
  p.pool_cs = p.pool_co = p.write_full = p.tiles_per_row = 0;
  p.amax_out = p.amax_pool = nullptr;  // per-image slots (Tensor::amax) are filled by a reduction pass after the launch
  if (fuse) {
    p.pool_out = pool->p;
    p.pool_cs = pool->cs;
    p.pool_co = pool->co;
    p.write_full = need_full ? 1 : 0;
    p.tiles_per_row = in.W / 64;
  }
  static const bool per_layer = getenv("KOCR_PROF_LAYERS") != nullptr;
  char nm[64];
  if (per_layer)
    snprintf(nm, sizeof nm, "conv_wino_128x%d%s:%s", L.Cout > 32 ? 64 : (L.Cout > 16 ? 32 : 16), fuse ? "p" : "", L.name.c_str());
  else
    snprintf(nm, sizeof nm, "conv_wino_128x%d%s", L.Cout > 32 ? 64 : (L.Cout > 16 ? 32 : 16), fuse ? "_pool" : "");
  const double flops = 2.0 * (double)M * L.Kreal * L.Cout;  // algorithmic (direct-convolution) FLOPs
  const double bytes = 4.0 * ((double)M * L.Cin + (double)M * L.Cout + (double)L.Kreal * L.Cout);
  {
    ProfScope ps(ctx, nm, flops, bytes);
    // 8-wave blocks (128 couts share one raw input tile) measured no faster than 4-wave ones;
    // kept as a developer A/B switch
    static const bool use8 = getenv("KOCR_WINO_NW8") != nullptr;
    const size_t mtiles = (M + 127) / 128;
    if (L.Cout > 32) {
      const bool wide = use8 && p.Cout_pad % 128 == 0;
      if (wide) {
        dim3 grid((unsigned)(mtiles * (p.Cout_pad / 128)));
        if (fuse)
          hipLaunchKernelGGL((conv_wino_kernel<1, 8, 8>), grid, dim3(512), 0, ctx->stream, p);
        else
          hipLaunchKernelGGL((conv_wino_kernel<0, 8, 8>), grid, dim3(512), 0, ctx->stream, p);
      } else {
        dim3 grid((unsigned)(mtiles * (p.Cout_pad / 64)));
        if (fuse)
          hipLaunchKernelGGL((conv_wino_kernel<1, 4, 4>), grid, dim3(256), 0, ctx->stream, p);
        else
          hipLaunchKernelGGL((conv_wino_kernel<0, 4, 4>), grid, dim3(256), 0, ctx->stream, p);
      }
    } else if (L.Cout > 16) {
      dim3 grid((unsigned)(mtiles * (p.Cout_pad / 32)));
      if (fuse)
        hipLaunchKernelGGL((conv_wino_kernel<1, 4, 2>), grid, dim3(256), 0, ctx->stream, p);
      else
        hipLaunchKernelGGL((conv_wino_kernel<0, 4, 2>), grid, dim3(256), 0, ctx->stream, p);
    } else {
      dim3 grid((unsigned)(mtiles * (p.Cout_pad / 16)));
      hipLaunchKernelGGL((conv_wino_kernel<0, 4, 1>), grid, dim3(256), 0, ctx->stream, p);
    }
    KOCR_HIP(ctx, hipGetLastError());
  }
  if (out.amax && (!fuse || need_full)) KOCR_TRY(launch_absmax(ctx, out, out.amax));
  if (fuse && pool->amax) KOCR_TRY(launch_absmax(ctx, *pool, pool->amax));
  if (pool && !fuse) return launch_maxpool2x2(ctx, out, *pool);
  return KOCR_OK;
}
