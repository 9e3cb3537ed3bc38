// conv_hsplit.hip — 3x3 / stride 1 / dilation 1 'same' convolution with FEW output channels (<= 32) on the gfx950 BF16
// matrix cores, fp32 operands split exactly into three bf16 pieces (six of the nine piece products kept, fp32
// accumulation): the arithmetic of conv_dsplit.hip, arranged for the CRAFT head (detection.py:392-406: conv_cls.0 / .2
// 32 -> 32, conv_cls.4 32 -> 16) and the decoder's last 3x3 (detection.py:109-113: upconv4.conv.3, 64 -> 32).
//
// With 32 output channels a (pixel, channel) operand is used by only 32 x 9 products, so the cost that decides the layout
// is the operand preparation, not the matrix pipe: conv_dsplit.hip gathers and splits the input once per TAP (9 x), the
// Winograd kernels transform it once per kernel row (3 x) and amortise that over >= 64 channels.  Here the haloed input
// tile is split ONCE into LDS and every tap reads it back at a shifted address:
//
//   block = 256 threads (4 waves), tile = 8 rows x 32 columns of one image; per 16-channel chunk the 10 x 34 halo tile is
//   loaded with raw buffer loads (out-of-image offsets return the zero padding), split, and stored as three bf16 planes
//   [pixel][16 ch] with a 48-byte pixel stride (odd multiple of 16 B: the 16-byte A-operand reads of 32 neighbouring
//   pixels are conflict-free);  wave w owns rows 2w, 2w+1 = two 32-pixel M-tiles x 32 couts (32 accumulators);  per
//   (chunk, tap): 6 ds_read_b128 + 3 weight loads (16 B per lane, L1/L2 resident: 27 KB per chunk) + 12 MFMA 32x32x16.
//   The next chunk's raw pixels are in flight while the current one is multiplied;  one buffer, two barriers per chunk --
//   48 KB of LDS and 164 registers let three blocks share a CU, which is what hides the barriers.
#include "split_common.h"
#include <algorithm>
#include <cmath>
#include <atomic>

struct HsParams {
  const float* in;
  const unsigned short* wgt;  // [Cin/16][9 taps][3 pieces][64 lanes][8]
  float* out;
  const float* pre_a;
  const float* pre_b;
  const float* post_a;
  const float* post_b;
  int H, W, Cin, in_cs, in_co;
  int Cout, out_cs, out_co;
  int relu;
  int tiles_x, tiles_y;
  unsigned* amax_out;
  const unsigned* amax_in;  // conv_hsh_kernel: per-image max-|x| slots of the input (Tensor::amax), never null there
};

namespace {
constexpr int HS_TH = 8, HS_TW = 32;
constexpr int HS_HH = HS_TH + 2, HS_HW = HS_TW + 2;
constexpr int HS_NPX = HS_HH * HS_HW;        // 340 halo pixels
constexpr int HS_PS = 24;                    // ushorts per pixel of a plane: 16 channels + 8 padding (48 bytes)
constexpr int HS_PLANE = HS_NPX * HS_PS;     // 8160 ushorts
constexpr int HS_IPT = (HS_NPX * 4 + 255) / 256;  // gather items (pixel, channel quad) per thread: 6
}  // namespace

__global__ __launch_bounds__(256, 3) void conv_hs_kernel(HsParams p) {
  __shared__ __attribute__((aligned(16))) unsigned short As[3 * HS_PLANE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, l5 = lane >> 5;
  int t = blockIdx.x;
  const int tx = t % p.tiles_x;
  t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int n = t / p.tiles_y;
  const int y0 = ty * HS_TH, x0 = tx * HS_TW;
  constexpr unsigned OOB = 0x80000000u;

  // one buffer resource per image: offsets stay below 2^31
  const float* img = p.in + (size_t)n * p.H * p.W * p.in_cs + p.in_co;
  const unsigned long long ib = (unsigned long long)img;
  const unsigned long long ibu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ib >> 32)) << 32) |
                                 (unsigned)__builtin_amdgcn_readfirstlane((int)ib);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)ibu, 0, 0x80000000, 0x00020000);

  unsigned goff[HS_IPT];
  int ldst[HS_IPT];
#pragma unroll
  for (int it = 0; it < HS_IPT; ++it) {
    const int item = tid + it * 256;
    const int px = item >> 2, c4 = item & 3;
    const int hy = px / HS_HW, hx = px - hy * HS_HW;
    const int gy = y0 + hy - 1, gx = x0 + hx - 1;
    const bool ok = px < HS_NPX && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    goff[it] = ok ? (unsigned)(((gy * p.W + gx) * p.in_cs + c4 * 4) * 4) : OOB;
    ldst[it] = px < HS_NPX ? px * HS_PS + c4 * 4 : -1;
  }
  auto load_raw = [&](v4f (&raw)[HS_IPT], int chunk) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < HS_IPT; ++it)
      raw[it] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff[it], chunk * 64, 0));
  };
  auto produce = [&](const v4f (&raw)[HS_IPT]) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < HS_IPT; ++it) {
      u2v h, m, l;
      kocr_split4(raw[it], h, m, l);
      if (it + 1 < HS_IPT || ldst[it] >= 0) {
        unsigned short* dst = As + ldst[it];
        *reinterpret_cast<u2v*>(dst) = h;
        *reinterpret_cast<u2v*>(dst + HS_PLANE) = m;
        *reinterpret_cast<u2v*>(dst + 2 * HS_PLANE) = l;
      }
    }
  };

  f16v acc[2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  // A operand of M-tile m (tile row 2*wave + m), tap (ky, kx): pixel ((2*wave + m + ky) * 34 + l31 + kx), k half l5
  const int a_lane = ((2 * wave) * HS_HW + l31) * HS_PS + l5 * 8;
  const unsigned short* w_lane = p.wgt + lane * 8;
  const int nchunks = p.Cin >> 4;

  v4f raw[HS_IPT];
  load_raw(raw, 0);
  produce(raw);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    if (c + 1 < nchunks) load_raw(raw, c + 1);
    const unsigned short* wc = w_lane + (size_t)c * 9 * 3 * 512;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - ky * 3;
      bf8 b[3], a[2][3];
#pragma unroll
      for (int s = 0; s < 3; ++s) b[s] = *reinterpret_cast<const bf8*>(wc + (tap * 3 + s) * 512);
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int s = 0; s < 3; ++s)
          a[m][s] = *reinterpret_cast<const bf8*>(As + s * HS_PLANE + a_lane + ((m + ky) * HS_HW + kx) * HS_PS);
      // smallest products first (the order of conv_dsplit.hip)
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][2], b[0], acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][0], b[2], acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][1], b[1], acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][1], b[0], acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][0], b[1], acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][0], b[0], acc[m], 0, 0, 0);
    }
    if (c + 1 < nchunks) {
      __syncthreads();  // every wave has read this chunk
      produce(raw);
      __syncthreads();
    }
  }

  // ---- epilogue: 32x32 C/D map: col (cout) = lane & 31, row (tile column) = (r&3) + 8*(r>>2) + 4*(lane>>5) ----------
  const int nc = l31 < p.Cout ? l31 : p.Cout - 1;
  const float pa = p.pre_a[nc], pb = p.pre_b[nc];
  const bool has_post = p.post_a != nullptr;
  const float qa = has_post ? p.post_a[nc] : 1.f, qb = has_post ? p.post_b[nc] : 0.f;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float o = acc[m][r] * pa + pb;
      if (p.relu) o = fmaxf(o, 0.f);
      if (has_post) o = o * qa + qb;
      acc[m][r] = o;
    }
  float* oimg = p.out + (size_t)n * p.H * p.W * p.out_cs + p.out_co;
  const unsigned long long ob = (unsigned long long)oimg;
  const unsigned long long obu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ob >> 32)) << 32) |
                                 (unsigned)__builtin_amdgcn_readfirstlane((int)ob);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)obu, 0, 0x80000000, 0x00020000);
  float mxv = 0.f;
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int y = y0 + 2 * wave + m;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * l5;
      const bool ok = y < p.H && x < p.W && l31 < p.Cout;
      const unsigned vo = ok ? (unsigned)(((y * p.W + x) * p.out_cs + l31) * 4) : OOB;
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[m][r]), ro, vo, 0, 0);
      if (ok) mxv = fmaxf(mxv, fabsf(acc[m][r]));
    }
  }
  if (p.amax_out) kocr_amax_update(p.amax_out + n, mxv);  // per-image slot (Tensor::amax): the tile lies inside image n
}

// ===================================================================================================
// conv_hs16_kernel -- the same for Cout <= 16 (conv_cls.4, 32 -> 16): on conv_hs_kernel's 32-column product tile half of
// every MFMA was padding (96 TF/s algorithmic against 188 for the 32-cout layers).  v_mfma_f32_16x16x32_bf16 with the
// WEIGHTS as the A operand (16 couts x 32 k) and the pixels as B (32 k x 16 pixels): a lane ends up with four consecutive
// couts of one pixel (16-byte stores, a half-wave writes 16 neighbouring pixels).  K = 32 of one MFMA = 16 channels x TWO
// taps: lanes 0..31 read tap 2j, lanes 32..63 tap 2j + 1 (the next halo pixel, or the first of the next halo row for the
// pair (2, 3); tap 9 does not exist: zero weights, its lanes re-read tap 8).  Same haloed tile, LDS planes and chunk
// pipeline as conv_hs_kernel; wave w owns tile rows 2w, 2w + 1 = four 16-pixel M-tiles (16 accumulator registers);
// per (chunk, pair): 3 weight loads + 12 ds_read_b128 + 24 MFMAs.
// ===================================================================================================
__global__ __launch_bounds__(256, 3) void conv_hs16_kernel(HsParams p) {
  __shared__ __attribute__((aligned(16))) unsigned short As[3 * HS_PLANE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  int t = blockIdx.x;
  const int tx = t % p.tiles_x;
  t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int n = t / p.tiles_y;
  const int y0 = ty * HS_TH, x0 = tx * HS_TW;
  constexpr unsigned OOB = 0x80000000u;

  const float* img = p.in + (size_t)n * p.H * p.W * p.in_cs + p.in_co;
  const unsigned long long ib = (unsigned long long)img;
  const unsigned long long ibu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ib >> 32)) << 32) |
                                 (unsigned)__builtin_amdgcn_readfirstlane((int)ib);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)ibu, 0, 0x80000000, 0x00020000);

  unsigned goff[HS_IPT];
  int ldst[HS_IPT];
#pragma unroll
  for (int it = 0; it < HS_IPT; ++it) {
    const int item = tid + it * 256;
    const int px = item >> 2, c4 = item & 3;
    const int hy = px / HS_HW, hx = px - hy * HS_HW;
    const int gy = y0 + hy - 1, gx = x0 + hx - 1;
    const bool ok = px < HS_NPX && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    goff[it] = ok ? (unsigned)(((gy * p.W + gx) * p.in_cs + c4 * 4) * 4) : OOB;
    ldst[it] = px < HS_NPX ? px * HS_PS + c4 * 4 : -1;
  }
  auto load_raw = [&](v4f (&raw)[HS_IPT], int chunk) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < HS_IPT; ++it)
      raw[it] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff[it], chunk * 64, 0));
  };
  auto produce = [&](const v4f (&raw)[HS_IPT]) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < HS_IPT; ++it) {
      u2v h, m, l;
      kocr_split4(raw[it], h, m, l);
      if (it + 1 < HS_IPT || ldst[it] >= 0) {
        unsigned short* dst = As + ldst[it];
        *reinterpret_cast<u2v*>(dst) = h;
        *reinterpret_cast<u2v*>(dst + HS_PLANE) = m;
        *reinterpret_cast<u2v*>(dst + 2 * HS_PLANE) = l;
      }
    }
  };

  v4f acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
  // B operand of M-tile i = (row 2 wave + (i >> 1), columns 16 (i & 1) ..): lane = pixel l15 x k-group g; g & 1 = channel
  // half, g >> 1 = tap of the pair
  const int b_lane = ((2 * wave) * HS_HW + l15) * HS_PS + (g & 1) * 8;
  const int d_next = (g >> 1) ? HS_PS : 0;                  // second tap of a pair: the next halo pixel ...
  const int d_cross = (g >> 1) ? (HS_HW - 2) * HS_PS : 0;   // ... or, from kx = 2, the first pixel of the next halo row
  const unsigned short* w_lane = p.wgt + lane * 8;
  const int nchunks = p.Cin >> 4;

  v4f raw[HS_IPT];
  load_raw(raw, 0);
  produce(raw);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    if (c + 1 < nchunks) load_raw(raw, c + 1);
    const unsigned short* wc = w_lane + (size_t)c * 5 * 3 * 512;
#pragma unroll
    for (int pr = 0; pr < 5; ++pr) {
      const int t0 = 2 * pr, ky = t0 / 3, kx = t0 - ky * 3;
      const int toff = (ky * HS_HW + kx) * HS_PS + (pr == 4 ? 0 : (kx == 2 ? d_cross : d_next));
      bf8 wv[3], x[4][3];
#pragma unroll
      for (int s = 0; s < 3; ++s) wv[s] = *reinterpret_cast<const bf8*>(wc + (pr * 3 + s) * 512);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int s = 0; s < 3; ++s)
          x[i][s] = *reinterpret_cast<const bf8*>(As + s * HS_PLANE + b_lane + ((i >> 1) * HS_HW + (i & 1) * 16) * HS_PS + toff);
      // smallest products first (the order of conv_dsplit.hip)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[0], x[i][2], acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[2], x[i][0], acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[1], x[i][1], acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[0], x[i][1], acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[1], x[i][0], acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[0], x[i][0], acc[i], 0, 0, 0);
    }
    if (c + 1 < nchunks) {
      __syncthreads();  // every wave has read this chunk
      produce(raw);
      __syncthreads();
    }
  }

  // ---- epilogue: 16x16 C/D map: row (cout) = 4 * (lane >> 4) + r, col (pixel of the M-tile) = lane & 15 --------------------
  float* oimg = p.out + (size_t)n * p.H * p.W * p.out_cs + p.out_co;
  const bool has_post = p.post_a != nullptr;
  const bool vec = p.Cout == 16 && (p.out_cs & 3) == 0 && (p.out_co & 3) == 0 && (((uintptr_t)p.out) & 15) == 0;
  float pa[4], pb[4], qa[4], qb[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int nc = 4 * g + r < p.Cout ? 4 * g + r : p.Cout - 1;
    pa[r] = p.pre_a[nc];
    pb[r] = p.pre_b[nc];
    qa[r] = has_post ? p.post_a[nc] : 1.f;
    qb[r] = has_post ? p.post_b[nc] : 0.f;
  }
  float mxv = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int y = y0 + 2 * wave + (i >> 1), x = x0 + (i & 1) * 16 + l15;
    const bool inside = y < p.H && x < p.W;
    v4f o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = acc[i][r] * pa[r] + pb[r];
      if (p.relu) v = fmaxf(v, 0.f);
      if (has_post) v = v * qa[r] + qb[r];
      o[r] = v;
    }
    float* dst = oimg + ((size_t)y * p.W + x) * p.out_cs + 4 * g;
    if (inside) {
      if (vec) {
        *reinterpret_cast<v4f*>(dst) = o;
#pragma unroll
        for (int r = 0; r < 4; ++r) mxv = fmaxf(mxv, fabsf(o[r]));
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * g + r < p.Cout) {
            dst[r] = o[r];
            mxv = fmaxf(mxv, fabsf(o[r]));
          }
      }
    }
  }
  if (p.amax_out) kocr_amax_update(p.amax_out + n, mxv);  // per-image slot (Tensor::amax): the tile lies inside image n
}

// ===================================================================================================
// conv_hsh_kernel -- conv_hs_kernel in fp16 arithmetic (round 4; KOCR_SPLIT_F16X2 / F16X1): the haloed tile is scaled by the
// image's exact power of two 2^e (e = 14 - exponent of the image's tracked max |x|: no transform here, so |x 2^e| < 2^15)
// and split ONCE into two fp16 planes (round to nearest), the weights are stored scaled per output channel and split the
// same way; per (chunk, tap) 4 ds_read_b128 + 2 weight loads + 6 v_mfma_f32_32x32x16_f16 (a_l b_h, a_h b_l, a_h b_h)
// instead of 6 + 3 + 12; 32 KB of LDS, four blocks per CU.  Same error class as the bf16x3 kernel (tests/test_conv_gpu.py).
// ===================================================================================================
__global__ __launch_bounds__(256, 4) void conv_hsh_kernel(HsParams p) {
  __shared__ __attribute__((aligned(16))) unsigned short As[2 * HS_PLANE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, l5 = lane >> 5;
  int t = blockIdx.x;
  const int tx = t % p.tiles_x;
  t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int n = t / p.tiles_y;
  const int y0 = ty * HS_TH, x0 = tx * HS_TW;
  constexpr unsigned OOB = 0x80000000u;
  const int e = __builtin_amdgcn_readfirstlane(kocr_scale_exp(p.amax_in + n, 14));
  const float sc = kocr_pow2(e);

  const float* img = p.in + (size_t)n * p.H * p.W * p.in_cs + p.in_co;
  const unsigned long long ib = (unsigned long long)img;
  const unsigned long long ibu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ib >> 32)) << 32) |
                                 (unsigned)__builtin_amdgcn_readfirstlane((int)ib);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)ibu, 0, 0x80000000, 0x00020000);

  unsigned goff[HS_IPT];
  int ldst[HS_IPT];
#pragma unroll
  for (int it = 0; it < HS_IPT; ++it) {
    const int item = tid + it * 256;
    const int px = item >> 2, c4 = item & 3;
    const int hy = px / HS_HW, hx = px - hy * HS_HW;
    const int gy = y0 + hy - 1, gx = x0 + hx - 1;
    const bool ok = px < HS_NPX && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    goff[it] = ok ? (unsigned)(((gy * p.W + gx) * p.in_cs + c4 * 4) * 4) : OOB;
    ldst[it] = px < HS_NPX ? px * HS_PS + c4 * 4 : -1;
  }
  auto load_raw = [&](v4f (&raw)[HS_IPT], int chunk) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < HS_IPT; ++it)
      raw[it] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff[it], chunk * 64, 0));
  };
  auto produce = [&](const v4f (&raw)[HS_IPT]) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < HS_IPT; ++it) {
      u2v h, l;
      kocr_split4_h(raw[it] * sc, h, l);  // exact scaling
      if (it + 1 < HS_IPT || ldst[it] >= 0) {
        unsigned short* dst = As + ldst[it];
        *reinterpret_cast<u2v*>(dst) = h;
        *reinterpret_cast<u2v*>(dst + HS_PLANE) = l;
      }
    }
  };

  f16v acc[2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  const int a_lane = ((2 * wave) * HS_HW + l31) * HS_PS + l5 * 8;
  const unsigned short* w_lane = p.wgt + lane * 8;  // [Cin/16][9 taps][2 pieces][64 lanes][8] fp16
  const int nchunks = p.Cin >> 4;

  v4f raw[HS_IPT];
  load_raw(raw, 0);
  produce(raw);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    if (c + 1 < nchunks) load_raw(raw, c + 1);
    const unsigned short* wc = w_lane + (size_t)c * 9 * 2 * 512;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - ky * 3;
      hf8 b[2], a[2][2];
#pragma unroll
      for (int s = 0; s < 2; ++s) b[s] = *reinterpret_cast<const hf8*>(wc + (tap * 2 + s) * 512);
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int s = 0; s < 2; ++s)
          a[m][s] = *reinterpret_cast<const hf8*>(As + s * HS_PLANE + a_lane + ((m + ky) * HS_HW + kx) * HS_PS);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m][1], b[0], acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m][0], b[1], acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m][0], b[0], acc[m], 0, 0, 0);
    }
    if (c + 1 < nchunks) {
      __syncthreads();  // every wave has read this chunk
      produce(raw);
      __syncthreads();
    }
  }

  // ---- epilogue (conv_hs_kernel's; pre_a carries 2^-wexp[o], the input scale is undone here) -----------------------------
  const int nc = l31 < p.Cout ? l31 : p.Cout - 1;
  const float pa = p.pre_a[nc] * kocr_pow2(-e), pb = p.pre_b[nc];
  const bool has_post = p.post_a != nullptr;
  const float qa = has_post ? p.post_a[nc] : 1.f, qb = has_post ? p.post_b[nc] : 0.f;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float o = acc[m][r] * pa + pb;
      if (p.relu) o = fmaxf(o, 0.f);
      if (has_post) o = o * qa + qb;
      acc[m][r] = o;
    }
  float* oimg = p.out + (size_t)n * p.H * p.W * p.out_cs + p.out_co;
  const unsigned long long ob = (unsigned long long)oimg;
  const unsigned long long obu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ob >> 32)) << 32) |
                                 (unsigned)__builtin_amdgcn_readfirstlane((int)ob);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)obu, 0, 0x80000000, 0x00020000);
  float mxv = 0.f;
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int y = y0 + 2 * wave + m;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * l5;
      const bool ok = y < p.H && x < p.W && l31 < p.Cout;
      const unsigned vo = ok ? (unsigned)(((y * p.W + x) * p.out_cs + l31) * 4) : OOB;
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[m][r]), ro, vo, 0, 0);
      if (ok) mxv = fmaxf(mxv, fabsf(acc[m][r]));
    }
  }
  if (p.amax_out) kocr_amax_update(p.amax_out + n, mxv);
}

// ===================================================================================================
// conv_first_kernel -- CRAFT's first layer (basenet.slice1.0, detection.py:312-322: 3x3, 3 -> 64, BN, ReLU) straight from
// the uint8 image, on the same bf16x3 split arithmetic.  K = 27 (tap, channel) values per output pixel, padded to 32 =
// two 16-k MFMA steps: the fp32-MFMA kernel that ran it before spent 48 padded K per pixel on the slow pipe (43 % busy,
// 2.6 TB/s of output against the 6.8 TB/s a plain fill reaches).  Block = 256 threads, tile = 8 rows x 32 columns; thread
// t gathers the 27 neighbours of ITS pixel (byte loads, out-of-image = the zero padding of the NORMALISED image),
// normalises them through the compute_input table (detection.py:34-42, bit-exact float32 values, copied to LDS), splits
// them and stores its im2col row, one 16-k step at a time, x 3 pieces (48-byte pixel stride: conflict-free 16-byte A reads);
// wave w owns tile rows 2w, 2w+1 = two M-tiles x 64 couts: 2 k-steps x 2 cout tiles x 12 = 48 MFMAs.  36 KB of LDS (the halo
// tile and the table live in the same buffer before the operand planes do): four blocks / CU.
// ===================================================================================================
struct FirstParams {
  const uint8_t* img;         // [N][H][W][3]
  const float* lut;           // [3][256]
  const unsigned short* wgt;  // [2 k-steps][2 cout tiles][3 pieces][64 lanes][8]
  float* out;
  const float* pre_a;
  const float* pre_b;
  const float* post_a;
  const float* post_b;
  int H, W, Cout, out_cs, out_co, relu;
  int tiles_x, tiles_y;
  unsigned* amax_out;  // per-image max-|x| slots of the output (Tensor::amax) or nullptr
};

namespace {
constexpr int F1_PS = 24;                    // ushorts per pixel of a plane: 16 k (one k-step) + 8 padding (48 bytes)
constexpr int F1_PLANE = 256 * F1_PS;        // 6144 ushorts
}  // namespace

__global__ __launch_bounds__(256, 4) void conv_first_kernel(FirstParams p) {
  // one 36 KB buffer: first the halo tile and the compute_input table, then (after a block barrier) the operand planes
  __shared__ __attribute__((aligned(16))) unsigned short As[3 * F1_PLANE];
  float* lut_s = reinterpret_cast<float*>(As) + 512;  // behind the 10 x 28 dword halo tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, l5 = lane >> 5;
  int t = blockIdx.x;
  const int tx = t % p.tiles_x;
  t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int n = t / p.tiles_y;
  const int y0 = ty * HS_TH, x0 = tx * HS_TW;
  constexpr unsigned OOB = 0x80000000u;
  for (int i = tid; i < 768; i += 256) lut_s[i] = p.lut[i];

  // ---- gather: the 10 x 34 x 3 byte halo tile goes to LDS once -- per halo row the 102 bytes [3 (x0 - 1), 3 (x0 + 33)) of the
  // image row as 27 aligned dwords (one load per thread; bytes outside the image are never used: `valid` below) -- then every
  // thread picks the 27 neighbours of its pixel from LDS.  (Round 2 issued 27 global byte loads per thread: 6.9 k vector-memory
  // instructions per tile against 270 here.)
  unsigned* tile_s = reinterpret_cast<unsigned*>(As);
  const uint8_t* img = p.img + (size_t)n * p.H * p.W * 3;
  const int py = y0 + (tid >> 5), px = x0 + (tid & 31);
  auto row_ptr = [&](int hr) { return (long)(uintptr_t)img + ((long)(y0 + hr - 1) * p.W + (x0 - 1)) * 3; };  // first byte of halo row hr
  for (int i = tid; i < HS_HH * 27; i += 256) {
    const int hr = i / 27, dw = i - hr * 27;  // halo row, dword of the row
    {
      const int gy = y0 + hr - 1;
      const long a = (row_ptr(hr) & ~3L) + 4 * dw;
      const long lo = (long)(uintptr_t)img, hi = lo + (long)p.H * p.W * 3;
      unsigned v = 0;
      if ((unsigned)gy < (unsigned)p.H) {
        if (a >= lo && a + 4 <= hi) {
          v = *reinterpret_cast<const unsigned*>((uintptr_t)a);
        } else {
          for (int b = 0; b < 4; ++b)
            if (a + b >= lo && a + b < hi) v |= (unsigned)*reinterpret_cast<const uint8_t*>((uintptr_t)(a + b)) << (8 * b);
        }
      }
      tile_s[hr * 28 + dw] = v;
    }
  }
  __syncthreads();
  unsigned char raw[27];
  unsigned valid = 0;
  {
    const unsigned char* tb = reinterpret_cast<const unsigned char*>(tile_s);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int hr = (tid >> 5) + ky;                                       // halo row of the neighbours of kernel row ky
      const int off = hr * 112 + (int)(row_ptr(hr) & 3L) + 3 * (tid & 31);  // byte of halo pixel (hr, tid & 31)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = py + ky - 1, ix = px + kx - 1;
        const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        valid |= (ok ? 1u : 0u) << (ky * 3 + kx);
#pragma unroll
        for (int c = 0; c < 3; ++c) raw[(ky * 3 + kx) * 3 + c] = tb[off + 3 * kx + c];
      }
    }
  }
  __syncthreads();  // table in LDS
  float v[32];
#pragma unroll
  for (int k = 0; k < 27; ++k) v[k] = ((valid >> (k / 3)) & 1u) ? lut_s[(k % 3) * 256 + raw[k]] : 0.f;
#pragma unroll
  for (int k = 27; k < 32; ++k) v[k] = 0.f;
  __syncthreads();  // every thread has its 27 table values: the tile / table area becomes the operand planes
  // ---- 2 M-tiles (tile rows 2 wave, 2 wave + 1) x 2 cout tiles x 2 k-steps ------------------------------------------
  // The im2col rows a wave multiplies are the rows its OWN threads produce (thread = pixel, wave w = tile rows 2w, 2w + 1), so
  // the exchange through LDS needs no block barrier (a wave's DS operations execute in order), and it runs one k-step at a
  // time through a 36 KB image (16 k x 3 pieces x 256 pixels) instead of 60 KB: four blocks per CU instead of two.
  f16v acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][nt][r] = 0.f;
  const unsigned short* w_lane = p.wgt + lane * 8;
#pragma unroll
  for (int kc = 0; kc < 2; ++kc) {
    {
      unsigned short* dst = As + tid * F1_PS;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        u2v h, m, l;
        kocr_split4(v4f{v[16 * kc + 4 * q], v[16 * kc + 4 * q + 1], v[16 * kc + 4 * q + 2], v[16 * kc + 4 * q + 3]}, h, m, l);
        *reinterpret_cast<u2v*>(dst + q * 4) = h;
        *reinterpret_cast<u2v*>(dst + F1_PLANE + q * 4) = m;
        *reinterpret_cast<u2v*>(dst + 2 * F1_PLANE + q * 4) = l;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    bf8 a[2][3], b[2][3];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int s = 0; s < 3; ++s)
        a[m][s] = *reinterpret_cast<const bf8*>(As + s * F1_PLANE + ((2 * wave + m) * 32 + l31) * F1_PS + l5 * 8);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();  // the second k-step overwrites these rows
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int s = 0; s < 3; ++s) b[nt][s] = *reinterpret_cast<const bf8*>(w_lane + ((kc * 2 + nt) * 3 + s) * 512);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][2], b[nt][0], acc[m][nt], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][0], b[nt][2], acc[m][nt], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][1], b[nt][1], acc[m][nt], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][1], b[nt][0], acc[m][nt], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][0], b[nt][1], acc[m][nt], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][0], b[nt][0], acc[m][nt], 0, 0, 0);
    }
  }

  // ---- epilogue: col (cout) = 32 nt + (lane & 31), row (tile column) = (r&3) + 8*(r>>2) + 4*(lane>>5) ------------------
  float* oimg = p.out + (size_t)n * p.H * p.W * p.out_cs + p.out_co;
  const unsigned long long ob = (unsigned long long)oimg;
  const unsigned long long obu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ob >> 32)) << 32) |
                                 (unsigned)__builtin_amdgcn_readfirstlane((int)ob);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)obu, 0, 0x80000000, 0x00020000);
  const bool has_post = p.post_a != nullptr;
  float mxv = 0.f;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int co = nt * 32 + l31;
    const int nc = co < p.Cout ? co : p.Cout - 1;
    const float pa = p.pre_a[nc], pb = p.pre_b[nc];
    const float qa = has_post ? p.post_a[nc] : 1.f, qb = has_post ? p.post_b[nc] : 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int y = y0 + 2 * wave + m;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * l5;
        float o = acc[m][nt][r] * pa + pb;
        if (p.relu) o = fmaxf(o, 0.f);
        if (has_post) o = o * qa + qb;
        const bool ok = y < p.H && x < p.W && co < p.Cout;
        const unsigned vo = ok ? (unsigned)(((y * p.W + x) * p.out_cs + co) * 4) : OOB;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o), ro, vo, 0, 0);
        mxv = fmaxf(mxv, ok ? fabsf(o) : 0.f);
      }
    }
  }
  if (p.amax_out) kocr_amax_update(p.amax_out + n, mxv);  // per-image slot (Tensor::amax): the tile lies inside image n
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
int prepare_hsplit(kocr_ctx* ctx, ConvLayer& L, const float* w, bool w_is_oihw) {
  if (L.KH != 3 || L.KW != 3 || L.dil != 1 || L.Cout > 32 || L.Cin % 16 != 0) return KOCR_OK;
  const int Cin = L.Cin, Cout = L.Cout;
  std::vector<unsigned short> u((size_t)(Cin / 16) * 9 * 3 * 512, 0);
  for (int c = 0; c < Cin; ++c)
    for (int tap = 0; tap < 9; ++tap)
      for (int o = 0; o < Cout; ++o) {
        const float g = w_is_oihw ? w[((size_t)o * Cin + c) * 9 + tap] : w[((size_t)tap * Cin + c) * Cout + o];
        // MFMA 32x32x16 B operand: lane = (k >> 3) * 32 + o holds k = 8 * (lane >> 5) + j
        const int k = c % 16, lane = (k >> 3) * 32 + o, j = k & 7;
        unsigned short pc[3];
        kocr_split3_host(g, pc);
        for (int s = 0; s < 3; ++s) u[((((size_t)(c / 16) * 9 + tap) * 3 + s) * 64 + lane) * 8 + j] = pc[s];
      }
  void* d = nullptr;
  KOCR_TRY(ctx->dev_alloc(&d, u.size() * sizeof(unsigned short)));
  KOCR_HIP(ctx, hipMemcpy(d, u.data(), u.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
  L.d_hs = (unsigned short*)d;
  {  // conv_hsh_kernel: w 2^wexp[o] in two fp16 pieces, [Cin/16][9 taps][2 pieces][64 lanes][8]; d_pre_a_h = pre_a 2^-wexp
     // (the field the F(4,3) fp16 kernels use for layers with more than 32 couts: a layer has one or the other)
    std::vector<int> wexp(32, 0);
    for (int o = 0; o < Cout; ++o) {
      float wmax = 0.f;
      for (int c = 0; c < Cin; ++c)
        for (int tap = 0; tap < 9; ++tap)
          wmax = std::max(wmax, std::fabs(w_is_oihw ? w[((size_t)o * Cin + c) * 9 + tap] : w[((size_t)tap * Cin + c) * Cout + o]));
      if (wmax > 0.f && std::isfinite(wmax)) {
        int E;
        std::frexp(wmax, &E);
        wexp[o] = std::max(-100, std::min(100, 15 - E));
      }
    }
    std::vector<unsigned short> v((size_t)(Cin / 16) * 9 * 2 * 512, 0);
    for (int c = 0; c < Cin; ++c)
      for (int tap = 0; tap < 9; ++tap)
        for (int o = 0; o < Cout; ++o) {
          const float g = std::ldexp(w_is_oihw ? w[((size_t)o * Cin + c) * 9 + tap] : w[((size_t)tap * Cin + c) * Cout + o], wexp[o]);
          const int k = c % 16, lane = (k >> 3) * 32 + o, j = k & 7;
          const _Float16 h = (_Float16)g, l = (_Float16)(g - (float)h);
          unsigned short hb, lb;
          memcpy(&hb, &h, 2);
          memcpy(&lb, &l, 2);
          v[((((size_t)(c / 16) * 9 + tap) * 2 + 0) * 64 + lane) * 8 + j] = hb;
          v[((((size_t)(c / 16) * 9 + tap) * 2 + 1) * 64 + lane) * 8 + j] = lb;
        }
    void* dh = nullptr;
    KOCR_TRY(ctx->dev_alloc(&dh, v.size() * sizeof(unsigned short)));
    KOCR_HIP(ctx, hipMemcpy(dh, v.data(), v.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
    L.d_hsh = (unsigned short*)dh;
    L.hs_wexp.assign(wexp.begin(), wexp.end());
  }
  if (Cout <= 16) {  // conv_hs16_kernel: [Cin/16][5 tap pairs][3 pieces][64 lanes][8]
    std::vector<unsigned short> v((size_t)(Cin / 16) * 5 * 3 * 512, 0);
    for (int c = 0; c < Cin; ++c)
      for (int tap = 0; tap < 9; ++tap)
        for (int o = 0; o < Cout; ++o) {
          const float g = w_is_oihw ? w[((size_t)o * Cin + c) * 9 + tap] : w[((size_t)tap * Cin + c) * Cout + o];
          // MFMA 16x16x32 A operand: lane = (k >> 3) * 16 + row holds k = 8 * (lane >> 4) + j;  k = (tap & 1) * 16 + channel
          const int k = (tap & 1) * 16 + c % 16, lane = (k >> 3) * 16 + o, j = k & 7;
          unsigned short pc[3];
          kocr_split3_host(g, pc);
          for (int s = 0; s < 3; ++s) v[((((size_t)(c / 16) * 5 + tap / 2) * 3 + s) * 64 + lane) * 8 + j] = pc[s];
        }
    void* d16 = nullptr;
    KOCR_TRY(ctx->dev_alloc(&d16, v.size() * sizeof(unsigned short)));
    KOCR_HIP(ctx, hipMemcpy(d16, v.data(), v.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
    L.d_hs16 = (unsigned short*)d16;
  }
  return KOCR_OK;
}

bool hsplit_applicable(kocr_ctx* ctx, const ConvLayer& L, const Tensor& in) {
  static const bool off = getenv("KOCR_HSPLIT") && atoi(getenv("KOCR_HSPLIT")) == 0;
  return !off && L.d_hs && in.cs % 4 == 0 && in.co % 4 == 0 &&
         ((uintptr_t)in.p & 15) == 0 && (size_t)in.H * in.W * in.cs * 4 < ((size_t)1 << 31) &&
         (size_t)in.H * in.W * 32 * 4 < ((size_t)1 << 31);
}

int launch_conv_hsplit(kocr_ctx* ctx, const ConvLayer& L, const Tensor& in, const Tensor& out) {
  if ((size_t)out.H * out.W * out.cs * 4 >= ((size_t)1 << 31)) KOCR_FAIL(ctx, KOCR_EINVAL, "conv " + L.name + ": image too large");
  HsParams p;
  p.in = in.p;
  p.wgt = L.d_hs;
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
  p.out_cs = out.cs;
  p.out_co = out.co;
  p.relu = L.relu;
  p.tiles_x = (in.W + HS_TW - 1) / HS_TW;
  p.tiles_y = (in.H + HS_TH - 1) / HS_TH;
  p.amax_out = out.amax;
  p.amax_in = nullptr;
  const size_t M = in.pixels();
  static const bool per_layer = getenv("KOCR_PROF_LAYERS") != nullptr;
  const bool no16 = !ctx->sw.hs16;
  const bool use16 = L.d_hs16 && !no16;  // <= 16 couts: the 16-wide product tile
  // fp16 arithmetic (conv_hsh_kernel) in the fp16 modes, for the 32-wide product tile
  const bool half = !use16 && ctx->split_mode != KOCR_SPLIT_BF16X3 && ctx->sw.w43h && L.d_hsh && L.d_pre_a_h;
  if (half) {
    const unsigned* slots = in.amax;
    if (!slots) {
      unsigned* tmp = ctx->amax_slots(in.N);
      if (!tmp) KOCR_FAIL(ctx, KOCR_ECAPACITY, "conv " + L.name + ": out of max-|x| slots");
      KOCR_TRY(launch_absmax(ctx, in, tmp));
      slots = tmp;
    }
    p.amax_in = slots;
    if (ctx->range_on) KOCR_TRY(launch_range_stats(ctx, L.name, in, slots, 14));
    p.wgt = L.d_hsh;
    p.pre_a = L.d_pre_a_h;
  }
  char nm[64];
  if (per_layer)
    snprintf(nm, sizeof nm, "conv_h%s_256x%d:%s", half ? "h" : "s", use16 ? 16 : 32, L.name.c_str());
  else
    snprintf(nm, sizeof nm, "conv_h%s_256x%d", half ? "h" : "s", use16 ? 16 : 32);
  const double flops = 2.0 * (double)M * L.Kreal * L.Cout;
  const double bytes = 4.0 * ((double)M * L.Cin + (double)M * L.Cout + (double)L.Kreal * L.Cout);
  ProfScope ps(ctx, nm, flops, bytes);
  const size_t grid = (size_t)in.N * p.tiles_y * p.tiles_x;
  if (use16) {
    p.wgt = L.d_hs16;
    hipLaunchKernelGGL(conv_hs16_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->stream, p);
  } else if (half)
    hipLaunchKernelGGL(conv_hsh_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->stream, p);
  else
    hipLaunchKernelGGL(conv_hs_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->stream, p);
  KOCR_HIP(ctx, hipGetLastError());
  return KOCR_OK;
}

// ---- first layer on raw uint8 --------------------------------------------------------------------------------------
int prepare_first(kocr_ctx* ctx, ConvLayer& L, const float* w, bool w_is_oihw) {
  if (L.KH != 3 || L.KW != 3 || L.dil != 1 || L.Cin != 3 || L.Cout > 64) return KOCR_OK;
  const int Cout = L.Cout;
  std::vector<unsigned short> u((size_t)2 * 2 * 3 * 512, 0);
  for (int tap = 0; tap < 9; ++tap)
    for (int c = 0; c < 3; ++c)
      for (int o = 0; o < Cout; ++o) {
        const float g = w_is_oihw ? w[((size_t)o * 3 + c) * 9 + tap] : w[((size_t)tap * 3 + c) * Cout + o];
        const int kk = tap * 3 + c, kc = kk / 16, k = kk % 16;
        const int lane = (k >> 3) * 32 + (o & 31), j = k & 7;
        unsigned short pc[3];
        kocr_split3_host(g, pc);
        for (int s = 0; s < 3; ++s) u[((((size_t)kc * 2 + o / 32) * 3 + s) * 64 + lane) * 8 + j] = pc[s];
      }
  void* d = nullptr;
  KOCR_TRY(ctx->dev_alloc(&d, u.size() * sizeof(unsigned short)));
  KOCR_HIP(ctx, hipMemcpy(d, u.data(), u.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
  L.d_first = (unsigned short*)d;
  return KOCR_OK;
}

bool first_applicable(kocr_ctx* ctx, const ConvLayer& L, const Tensor& in) {
  static const bool off = getenv("KOCR_FIRST") && atoi(getenv("KOCR_FIRST")) == 0;
  return !off && L.d_first && (size_t)in.H * in.W * 64 * 4 < ((size_t)1 << 31);
}

int launch_conv_first(kocr_ctx* ctx, const ConvLayer& L, const Tensor& in, const uint8_t* in_u8, const float* lut, const Tensor& out) {
  if ((size_t)out.H * out.W * out.cs * 4 >= ((size_t)1 << 31)) KOCR_FAIL(ctx, KOCR_EINVAL, "conv " + L.name + ": image too large");
  FirstParams p;
  p.img = in_u8;
  p.lut = lut;
  p.wgt = L.d_first;
  p.out = out.p;
  p.pre_a = L.d_pre_a;
  p.pre_b = L.d_pre_b;
  p.post_a = L.d_post_a;
  p.post_b = L.d_post_b;
  p.H = in.H;
  p.W = in.W;
  p.Cout = L.Cout;
  p.out_cs = out.cs;
  p.out_co = out.co;
  p.relu = L.relu;
  p.tiles_x = (in.W + HS_TW - 1) / HS_TW;
  p.tiles_y = (in.H + HS_TH - 1) / HS_TH;
  // per-image max-|x| slots of the output: the layer's constant bound (ConvLayer::first_bound) when it is known -- no
  // reduction in the kernel --, else tracked by the kernel
  p.amax_out = nullptr;
  if (out.amax) {
    if (L.first_bound > 0.f && std::isfinite(L.first_bound)) {
      unsigned bits;
      memcpy(&bits, &L.first_bound, 4);
      KOCR_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t)out.amax, (int)bits, (size_t)in.N, ctx->stream));
    } else {
      p.amax_out = out.amax;
    }
  }
  const size_t M = in.pixels();
  const double flops = 2.0 * (double)M * L.Kreal * L.Cout;
  const double bytes = (double)M * 3 + 4.0 * ((double)M * L.Cout + (double)L.Kreal * L.Cout);
  ProfScope ps(ctx, "conv_hs_first_256x64", flops, bytes);
  const size_t grid = (size_t)in.N * p.tiles_y * p.tiles_x;
  if (grid > 0x7fffffff) KOCR_FAIL(ctx, KOCR_EINVAL, "conv " + L.name + ": too many tiles");
  hipLaunchKernelGGL(conv_first_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->stream, p);
  KOCR_HIP(ctx, hipGetLastError());
  return KOCR_OK;
}
