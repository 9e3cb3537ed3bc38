// conv_dsplit.hip — direct (implicit-GEMM) stride-1 'same' convolution of any kernel size / dilation
// on the gfx950 BF16 matrix cores, fp32 operands split exactly into three bf16 pieces (six of the nine
// piece products kept, fp32 accumulation) — the same arithmetic as conv_wsplit.hip, without the
// Winograd transform.  It serves the layers the Winograd kernel cannot take: 1x1 convolutions
// (detection.py:349-353 slice5 1x1, :106-115 the upconv 1x1s) and the dilated 3x3 of slice5
// (detection.py:351, dilation 6).
//
//   out[pixel][o] = sum_{tap, c} in[pixel + offset(tap)][c] * w[tap][c][o]
//
// Block = 512 threads, persistent (one per CU): 4 consumer waves (MFMA + weight stream) and 4 producer
// waves (gather of the shifted input pixels through raw buffer loads whose out-of-range offset returns
// the zero padding, exact bf16x3 split, LDS fill).  Tile = 256*WM consecutive pixels of the flattened
// (n, y, x) order x 32*WN output channels; consumer wave (wm, wn) owns 8 M-tiles of 32 pixels x 32
// couts (128 accumulator registers).  K-step = one (16-channel group, tap): 48 MFMA 32x32x16 per wave.
// LDS: As[buf][piece][M-tile][k half][32 rows x 8 ch] bf16, 24 KB * WM per buffer, the same
// conflict-free layout as conv_wsplit.hip.  Weights: [16-ch group][tap][32-cout tile][piece][lane][8].
#include "split_common.h"
#include "w43_common.h"  // w4_div_magic / w4_fdiv
#include "probe_clock.h"
#include <cmath>
#include <algorithm>

struct DsParams {
  const float* in;
  const unsigned short* wgt;
  float* out;
  const float* pre_a;
  const float* pre_b;
  const float* post_a;
  const float* post_b;
  int H, W, Cin, in_cs, in_co;
  int Cout, Cout_pad, out_cs, out_co;
  int relu;
  int KH, KW, dil;
  int nsteps;  // KH * KW * Cin / 16
  int Mtotal;
  int total_tiles;
  const unsigned* amax_in;  // HALF kernels: tracked max |input| (Tensor::amax), never null there
  int w_exp;                // HALF kernels: the weights are stored multiplied by 2^w_exp
  unsigned* amax_out;  // Tensor::amax of the output or nullptr
  // UP kernels: out = epilogue(acc + bilinear_resize(up)[pixel][cout]) -- see launch_conv_dsplit
  const float* up;     // [N][up_H][up_W][Cout] contiguous
  int up_H, up_W;
  float up_sy, up_sx;  // up_H / H, up_W / W
  unsigned dv_hw[2];   // w4_div_magic(H W): the image of a flattened pixel without a 64-bit division
};

// probe_clock.h bins: consumer wave 0: [0] K loop  [1] up-sampled addend  [2] post + amax  [3] store issue;  producer wave 4:
// [4] LDS fill (with the wait for its raw loads)  [5] barrier  [6] load issue

// (WM, WN) = (1, 4) / (2, 2): 512 threads, one block per CU.  The kernel also builds as (1, 2) -- 256 threads, two consumer and
// two producer waves, 64 KB of LDS, TWO blocks per CU so that one block's epilogue overlaps the other's K loop; measured in
// round 6 (profiles/r06_ab_notes.txt item 2): no faster (1.165 -> 1.150 ms, 1.076 -> 1.066 ms with PAIR), the layer is bound
// by the memory system, not by the serial epilogue.  Not instantiated.
// PAIR = 1 (round 6; 1x1 convolutions with an even number of 16-channel groups): the producers fetch TWO consecutive K-steps
// with back-to-back loads per item -- the two 64-byte halves of the same 128-byte lines (a pixel's channels c .. c + 15 and
// c + 16 .. c + 31).  Fetched a K-step (~2 us) apart, as before, the second half found its line evicted from L2 again in ~40 %
// of the cases (PMC: upconv4.conv.0 read 3.69 GB for 2.72 GB of input + addend; profiles/r06_ab_notes.txt item 3).
template <int WM, int WN, int HALF, int UP, int PAIR = 0>
__global__ __launch_bounds__(128 * WM * WN, WM * WN == 2 ? 2 : 1) void conv_ds_kernel(DsParams p) {
  constexpr int NP = HALF ? 2 : 3;            // operand pieces
  constexpr int NCW = WM * WN;                // consumer waves = producer waves
  constexpr int NMT = 8 * WM;                 // 32-pixel M-tiles per block tile
  constexpr int TILE_PX = 256 * WM;
  constexpr int PXI = 16 * NCW;               // pixels one gather item of all producer threads covers
  constexpr int IPT = TILE_PX / PXI;          // gather items (pixel, channel quad) per producer thread
  constexpr int KH_STRIDE = 256;              // ushorts: 32 rows x 8 channels
  constexpr int PLANE = NMT * 2 * KH_STRIDE;  // one piece plane
  constexpr int BUF = NP * PLANE;             // one K-step: 24 KB * WM (bf16x3) / 16 KB * WM (fp16x2)
  extern __shared__ __attribute__((aligned(16))) unsigned short As[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: keep it in an SGPR
  const int l31 = lane & 31, l5 = lane >> 5;
  const int nblk_n = p.Cout_pad / (32 * WN);
  const int total = p.total_tiles;
  const int ns = p.nsteps;
  const int G = gridDim.x;
  const int pad_y = p.dil * (p.KH / 2), pad_x = p.dil * (p.KW / 2);

  // ==================================================================================================
  // producer waves 4..7
  // ==================================================================================================
  if (wave >= NCW) {
    const int ptid = tid - 64 * NCW;
    const int quad = ptid & 3;
    constexpr unsigned OOB = 0x80000000u;
    const float in_scale = HALF ? kocr_pow2(kocr_scale_exp(p.amax_in, 13)) : 1.f;  // exact power of two
    (void)in_scale;
    int ldst[IPT];
#pragma unroll
    for (int it = 0; it < IPT; ++it) {
      const int idx = (ptid >> 2) + it * PXI;
      ldst[it] = ((idx >> 5) * 2 + (quad >> 1)) * KH_STRIDE + ((((idx & 31) * 8) ^ ((quad >> 1) * 32)) + (quad & 1) * 4);
    }
    // position of the NEXT K-step to load: tile L_ld, step (ld_cg, ld_tap = (ld_ky, ld_kx))
    int L_ld = blockIdx.x, ld_ky = 0, ld_kx = 0, ld_cg = 0;
    unsigned goff[IPT];
    int gy[IPT], gx[IPT];
    bool gok[IPT];
    __amdgpu_buffer_rsrc_t rsrc;
    // buffer resource based (pad_y rows + pad_x pixels) BEFORE the tile: every tap offset is >= 0
    auto tile_geometry = [&]() __attribute__((always_inline)) {
      const int tile = kocr_xcd_remap(L_ld < total ? L_ld : 0, total);
      const long pm0 = (long)(tile / nblk_n) * TILE_PX;
      const float* bbase_v = p.in + (pm0 * p.in_cs + p.in_co) - (long)(pad_y * p.W + pad_x) * p.in_cs;
      const unsigned long long bb = (unsigned long long)bbase_v;
      const unsigned long long bbu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(bb >> 32)) << 32) |
                                     (unsigned)__builtin_amdgcn_readfirstlane((int)bb);
      rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)bbu, 0, 0x80000000, 0x00020000);
#pragma unroll
      for (int it = 0; it < IPT; ++it) {
        const int idx = (ptid >> 2) + it * PXI;
        const long g = pm0 + idx;
        gok[it] = g < p.Mtotal && L_ld < total;
        gx[it] = (int)(g % p.W);
        gy[it] = (int)((g / p.W) % p.H);
        goff[it] = (unsigned)((idx * p.in_cs + quad * 4) * 4);
      }
    };
    // PAIR: K-steps 2 j and 2 j + 1 of a 1x1 convolution (channel groups 2 j, 2 j + 1; no taps, no padding)
    auto load_raw2 = [&](v4f (&ra)[IPT], v4f (&rb)[IPT]) __attribute__((always_inline)) {
      const int soff = ld_cg * 64;
#pragma unroll
      for (int it = 0; it < IPT; ++it) {
        const unsigned off = gok[it] ? goff[it] : OOB;
        ra[it] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, soff, 0));
        rb[it] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, soff + 64, 0));
      }
      ld_cg += 2;
      if (ld_cg == p.Cin / 16) {  // next tile
        ld_cg = 0;
        L_ld += G;
        tile_geometry();
      }
    };
    auto load_raw = [&](v4f (&raw)[IPT]) __attribute__((always_inline)) {
      const int dy = ld_ky * p.dil - pad_y, dx = ld_kx * p.dil - pad_x;
      const int soff = (((dy + pad_y) * p.W + (dx + pad_x)) * p.in_cs + ld_cg * 16) * 4;
#pragma unroll
      for (int it = 0; it < IPT; ++it) {
        const bool ok = gok[it] && (unsigned)(gy[it] + dy) < (unsigned)p.H && (unsigned)(gx[it] + dx) < (unsigned)p.W;
        raw[it] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? goff[it] : OOB, soff, 0));
      }
      if (++ld_kx == p.KW) {
        ld_kx = 0;
        if (++ld_ky == p.KH) {
          ld_ky = 0;
          if (++ld_cg == p.Cin / 16) {  // next tile
            ld_cg = 0;
            L_ld += G;
            tile_geometry();
          }
        }
      }
    };
    auto produce = [&](const v4f (&raw)[IPT], int buf) __attribute__((always_inline)) {
      unsigned short* base = As + buf * BUF;
#pragma unroll
      for (int it = 0; it < IPT; ++it) {
        unsigned short* dst = base + ldst[it];
        if constexpr (HALF) {
          u2v h, l;
          kocr_split4_h(raw[it] * in_scale, h, l);
          *reinterpret_cast<u2v*>(dst) = h;
          *reinterpret_cast<u2v*>(dst + PLANE) = l;
        } else {
          u2v h, m, l;
          kocr_split4(raw[it], h, m, l);
          *reinterpret_cast<u2v*>(dst) = h;
          *reinterpret_cast<u2v*>(dst + PLANE) = m;
          *reinterpret_cast<u2v*>(dst + 2 * PLANE) = l;
        }
      }
    };
    const int my_tiles = (total - (int)blockIdx.x + G - 1) / G;
    const int T = my_tiles * ns;
    tile_geometry();
    // D - 1 K-steps of raw input in flight ahead of the LDS fill (the LDS ring itself stays two deep): these layers are
    // HBM-bound (K = 64 .. 256 per pixel), and one 32 KB step in flight per CU is 8 MB on the chip -- half of what 8 TB/s
    // times the loaded latency asks for.  Loads past the block's last step are out-of-range buffer loads (no traffic).
    constexpr int D = PAIR ? 6 : 4;  // PAIR: three pairs of steps, two of them (4 steps) in flight ahead of the fill
    v4f raw[D][IPT];
    int k = 0;
    if constexpr (PAIR) {
      load_raw2(raw[0], raw[1]);
      load_raw2(raw[2], raw[3]);
      PROBE_T0();
      for (; k + D <= T; k += D) {
#pragma unroll
        for (int d = 0; d < D; d += 2) {
          load_raw2(raw[(d + 4) % D], raw[(d + 5) % D]);
          PROBE_T(6);
          produce(raw[d], 0);  // k and d are even: step k + d fills buffer 0, the next one buffer 1
          PROBE_T(4);
          __syncthreads();
          PROBE_T(5);
          produce(raw[d + 1], 1);
          PROBE_T(4);
          __syncthreads();
          PROBE_T(5);
        }
      }
      PROBE_TEND(tid == 64 * NCW, 4, 7);
#pragma unroll
      for (int d = 0; d < 4; ++d)  // T is even: 0, 2 or 4 steps are left, all of them loaded
        if (k + d < T) {
          produce(raw[d], d & 1);
          __syncthreads();
        }
    } else {
#pragma unroll
    for (int d = 0; d < D - 1; ++d) load_raw(raw[d]);
    PROBE_T0();
    for (; k + D <= T; k += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        load_raw(raw[(d + D - 1) % D]);
        PROBE_T(6);
        produce(raw[d], d & 1);  // k is a multiple of D (even): step k + d fills buffer d & 1
        PROBE_T(4);
        __syncthreads();
        PROBE_T(5);
      }
    }
    PROBE_TEND(tid == 64 * NCW, 4, 7);
#pragma unroll
    for (int d = 0; d < D - 1; ++d)
      if (k + d < T) {
        produce(raw[d], d & 1);
        __syncthreads();
      }
    }
    __syncthreads();  // pairs with the consumers' barrier inside the last K-step
    return;
  }

  // ==================================================================================================
  // consumer waves 0..3: wave (wm, wn): 8 M-tiles (256 pixels) x 32 output channels
  // ==================================================================================================
  const int wn = (WN == 4) ? wave : (wave % WN), wm = (WM == 1) ? 0 : (wave / WN);
  const int ntiles32 = p.Cout_pad >> 5;
  const size_t w_step = (size_t)ntiles32 * NP * 64 * 8;  // ushorts per K-step
  auto w_tile = [&](int nt) { return p.wgt + ((size_t)(nt * WN + wn) * NP * 64 + lane) * 8; };

  bf8 bw[NP], bwn[NP];
  f16v acc[4][2];  // [M-tile pair][M-tile in pair]
  const int a_lane = (wm * 8 * 2 + l5) * KH_STRIDE + ((l31 * 8) ^ (l5 * 32));
  auto load_a = [&](bf8 (&a)[2][NP], const unsigned short* bufp, int g) __attribute__((always_inline)) {
    const unsigned short* base = bufp + a_lane + g * 4 * KH_STRIDE;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int s = 0; s < NP; ++s) a[m][s] = *reinterpret_cast<const bf8*>(base + s * PLANE + m * 2 * KH_STRIDE);
  };
  auto mfma6 = [&](const bf8 (&a)[2][NP], int g) __attribute__((always_inline)) {
    if constexpr (HALF) {
      const hf8 b0 = __builtin_bit_cast(hf8, bw[0]), b1 = __builtin_bit_cast(hf8, bw[1]);
#pragma unroll
      for (int m = 0; m < 2; ++m)
        acc[g][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf8, a[m][1]), b0, acc[g][m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 2; ++m)
        acc[g][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf8, a[m][0]), b1, acc[g][m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 2; ++m)
        acc[g][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf8, a[m][0]), b0, acc[g][m], 0, 0, 0);
    } else {
      const bf8 b0 = bw[0], b1 = bw[1], b2 = bw[NP - 1];
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[g][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][NP - 1], b0, acc[g][m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[g][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][0], b2, acc[g][m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[g][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][1], b1, acc[g][m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[g][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][1], b0, acc[g][m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[g][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][0], b1, acc[g][m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[g][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][0], b0, acc[g][m], 0, 0, 0);
    }
  };
  // One K-step (see conv_wsplit.hip): LDS fetch of the next M-tile pair behind the 12 MFMAs of the
  // current one; the barrier publishing the next K-step sits before the last pair's MFMAs; the next
  // step's weights (3 x 16 B per lane) are fetched at the start of the step.
  bf8 a0[2][NP], a1[2][NP];
  // last_c (UP kernels, a tile's last step): the first A fragments of the NEXT tile are fetched after the epilogue instead
  // of here -- 24 registers the up-sampling epilogue needs for its loads in flight (it spilled accumulators otherwise)
  auto compute_step = [&](auto last_c, const unsigned short* bufp, const unsigned short* bufn,
                          const unsigned short* w_next) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(last_c)::value;
#pragma unroll
    for (int s = 0; s < NP; ++s) bwn[s] = *reinterpret_cast<const bf8*>(w_next + (size_t)s * 64 * 8);
    load_a(a1, bufp, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma6(a0, 0);
    __builtin_amdgcn_sched_barrier(0);
    load_a(a0, bufp, 2);
    __builtin_amdgcn_sched_barrier(0);
    mfma6(a1, 1);
    __builtin_amdgcn_sched_barrier(0);
    load_a(a1, bufp, 3);
    __builtin_amdgcn_sched_barrier(0);
    mfma6(a0, 2);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    if constexpr (!LAST) load_a(a0, bufn, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfma6(a1, 3);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < NP; ++s) bw[s] = bwn[s];
  };

  int gs = 0;
  {
    const int tile0 = kocr_xcd_remap(blockIdx.x, total);
    const unsigned short* w0 = w_tile(tile0 % nblk_n);
#pragma unroll
    for (int s = 0; s < NP; ++s) bw[s] = *reinterpret_cast<const bf8*>(w0 + (size_t)s * 64 * 8);
  }
  __syncthreads();  // global step 0 is in LDS
  load_a(a0, As, 0);
  PROBE_T0();
  for (int L = blockIdx.x; L < total; L += G) {
    const int tile = kocr_xcd_remap(L, total);
    const int mt = tile / nblk_n, nt = tile - mt * nblk_n;
    const long pm0 = (long)mt * TILE_PX;
    const unsigned short* w_ptr = w_tile(nt);
    const unsigned short* w_after = (L + G < total) ? w_tile(kocr_xcd_remap(L + G, total) % nblk_n) : w_ptr;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][m][r] = 0.f;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // drain the epilogue's stores once per tile (see conv_wsplit.hip)
    if constexpr (UP != 0) {
      for (int s = 0; s < ns - 1; ++s, ++gs)
        compute_step(std::false_type{}, As + (gs & 1) * BUF, As + ((gs + 1) & 1) * BUF, w_ptr + (size_t)(s + 1) * w_step);
      compute_step(std::true_type{}, As + (gs & 1) * BUF, As + ((gs + 1) & 1) * BUF, w_after);
      ++gs;
    } else {
      for (int s = 0; s < ns; ++s, ++gs)
        compute_step(std::false_type{}, As + (gs & 1) * BUF, As + ((gs + 1) & 1) * BUF, s + 1 < ns ? w_ptr + (size_t)(s + 1) * w_step : w_after);
    }
    PROBE_T(0);

    // ---- epilogue: 32x32 C/D map: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) -------------
    // branch-free, in place, raw buffer stores back to back (see conv_wsplit.hip)
    {
      const int n = (nt * WN + wn) * 32 + l31;
      const int nc = n < p.Cout ? n : p.Cout - 1;
      const float unscale = HALF ? kocr_pow2(-(kocr_scale_exp(p.amax_in, 13) + p.w_exp)) : 1.f;  // exact
      const float pa = p.pre_a[nc] * unscale, pb = p.pre_b[nc];
      const bool has_post = p.post_a != nullptr;
      const float qa = has_post ? p.post_a[nc] : 1.f, qb = has_post ? p.post_b[nc] : 0.f;
      if constexpr (UP) {
        // acc += the low-resolution tensor `up` resized bilinearly to this layer's H x W (the resize_bilinear_kernel
        // formula, elementwise.hip).  The four tap offsets and two weights of a pixel are the same for every output
        // channel: each wave works them out once for its 256 pixels (4 per lane) into a private LDS table -- no
        // block barrier, a wave's DS operations execute in order -- and then reads one 32-byte entry per pixel.
        unsigned* tab = reinterpret_cast<unsigned*>(As + 2 * BUF) + wave * (256 * 8);
        const long wbase = pm0 + (long)wm * 256;
        const unsigned ucs4 = (unsigned)p.Cout * 4u;
        // byte offsets are relative to the image of the wave's first pixel (its 256 pixels reach into the next image at
        // most): the 32-bit offsets then only have to span two images, whatever the batch size
        const int hw = p.H * p.W;
        const long wb = wbase < (long)p.Mtotal - 1 ? wbase : (long)p.Mtotal - 1;
        const int n0 = __builtin_amdgcn_readfirstlane((int)(wb / hw));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int q = lane + 64 * i;
          long gl = wbase + q;
          if (gl > (long)p.Mtotal - 1) gl = (long)p.Mtotal - 1;
          const int gi = (int)gl;
          const int ox = gi % p.W, t2 = gi / p.W;
          const int oy = t2 % p.H, ni = t2 / p.H;
          const float fy = ((float)oy + 0.5f) * p.up_sy - 0.5f;
          const float fx = ((float)ox + 0.5f) * p.up_sx - 0.5f;
          const float fly = floorf(fy), flx = floorf(fx);
          const int y0 = max((int)fly, 0), y1 = min((int)ceilf(fy), p.up_H - 1);
          const int x0 = max((int)flx, 0), x1 = min((int)ceilf(fx), p.up_W - 1);
          const unsigned r0 = (unsigned)(((ni - n0) * p.up_H + y0) * p.up_W), r1 = (unsigned)(((ni - n0) * p.up_H + y1) * p.up_W);
          uint4 o4;
          o4.x = (r0 + x0) * ucs4;
          o4.y = (r0 + x1) * ucs4;
          o4.z = (r1 + x0) * ucs4;
          o4.w = (r1 + x1) * ucs4;
          *reinterpret_cast<uint4*>(tab + q * 8) = o4;
          float2 wl;
          wl.x = fx - flx;
          wl.y = fy - fly;
          *reinterpret_cast<float2*>(tab + q * 8 + 4) = wl;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const unsigned long long ub = (unsigned long long)(p.up + (size_t)n0 * p.up_H * p.up_W * p.Cout);
        const unsigned long long ubu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ub >> 32)) << 32) |
                                       (unsigned)__builtin_amdgcn_readfirstlane((int)ub);
        const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc((void*)ubu, 0, 0x7FFFFFFF, 0x00020000);
        const unsigned ch = (unsigned)nc * 4u;
        if constexpr (UP == 2) {
          // UP = 2: exactly 2x in both directions and W % 4 == 0 (the launcher checks).  The four outputs ox0 .. ox0 + 3
          // (ox0 = 4 q) of a register group read the source columns k - 1, k, k + 1, k + 2 (k = ox0 / 2; clamped at the image
          // edges by the table entries of the group's first and last pixel) of two source rows: 8 loads instead of 16, the
          // same blend arithmetic
          // One stage = one M-tile (g, m): its 32 loads are issued a whole stage ahead of their blend, so that 32 .. 64
          // loads per lane are always in flight (issued batch by batch behind their own use, the 32 batches of a tile each
          // exposed one full memory latency: 26 of the 42 us a tile of upconv4.conv.0 took).
          float tb[2][4][8];  // [stage parity][rr][top k-1 .. k+2, bottom k-1 .. k+2]
          // the lane's table base is made opaque once per tile: every entry address below is then base + an immediate DS
          // offset (left visible as loop-invariant, hipcc hoists ~100 precomputed addresses out of the tile loop and spills them)
          unsigned tab_lane = 4 * l5 * 8;
          asm volatile("" : "+v"(tab_lane));  // (an integer, not the pointer: an opaque pointer would lose its LDS address space)
          const unsigned* tab_l = tab + tab_lane;
          auto up_issue = [&](int gm, float (&v)[4][8]) __attribute__((always_inline)) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              const unsigned* e = tab_l + (gm * 32 + 8 * rr) * 8;
              const uint4 oa = *reinterpret_cast<const uint4*>(e);          // pixel 0: (y0,k-1) (y0,k) (y1,k-1) (y1,k)
              const uint4 od = *reinterpret_cast<const uint4*>(e + 3 * 8);  // pixel 3: (y0,k+1) (y0,k+2) (y1,k+1) (y1,k+2)
              v[rr][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ru, oa.x + ch, 0, 0));
              v[rr][1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ru, oa.y + ch, 0, 0));
              v[rr][2] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ru, od.x + ch, 0, 0));
              v[rr][3] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ru, od.y + ch, 0, 0));
              v[rr][4] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ru, oa.z + ch, 0, 0));
              v[rr][5] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ru, oa.w + ch, 0, 0));
              v[rr][6] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ru, od.z + ch, 0, 0));
              v[rr][7] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ru, od.w + ch, 0, 0));
            }
          };
          auto up_blend = [&](f16v& a, int gm, const float (&v)[4][8]) __attribute__((always_inline)) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              const unsigned* e = tab_l + (gm * 32 + 8 * rr) * 8;
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float2 wl = *reinterpret_cast<const float2*>(e + j * 8 + 4);
                const int c0 = (j + 1) >> 1;  // left tap of output j: columns (k-1,k) (k,k+1) (k,k+1) (k+1,k+2)
                const float top = v[rr][c0] + (v[rr][c0 + 1] - v[rr][c0]) * wl.x;
                const float bot = v[rr][4 + c0] + (v[rr][4 + c0 + 1] - v[rr][4 + c0]) * wl.x;
                a[rr * 4 + j] += top + (bot - top) * wl.y;
              }
            }
          };
          up_issue(0, tb[0]);
#pragma unroll
          for (int gm = 0; gm < 8; ++gm) {
            if (gm + 1 < 8) up_issue(gm + 1, tb[(gm + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            up_blend(acc[gm >> 1][gm & 1], gm, tb[gm & 1]);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              const unsigned* e = tab + ((g * 2 + m) * 32 + 8 * rr + 4 * l5) * 8;
              uint4 o4[4];
              float2 wl[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                o4[j] = *reinterpret_cast<const uint4*>(e + j * 8);
                wl[j] = *reinterpret_cast<const float2*>(e + j * 8 + 4);
              }
              float v[4][4];
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                v[j][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ru, o4[j].x + ch, 0, 0));
                v[j][1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ru, o4[j].y + ch, 0, 0));
                v[j][2] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ru, o4[j].z + ch, 0, 0));
                v[j][3] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ru, o4[j].w + ch, 0, 0));
              }
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float top = v[j][0] + (v[j][1] - v[j][0]) * wl[j].x;
                const float bot = v[j][2] + (v[j][3] - v[j][2]) * wl[j].x;
                acc[g][m][rr * 4 + j] += top + (bot - top) * wl[j].y;
              }
            }
        }
      }
      PROBE_T(1);
      {
        const float lo = p.relu ? 0.f : -INFINITY;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][m][r] = fmaxf(acc[g][m][r] * pa + pb, lo);
        if (has_post) {  // uniform
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[g][m][r] = acc[g][m][r] * qa + qb;
        }
      }
      if (p.amax_out) {
        // per-image max |x| (Tensor::amax[n]): the wave's 256 consecutive pixels lie inside one image unless H W is not a
        // multiple of 256 -- then every image the wave touches gets the maximum over ITS pixels only (a slot must not
        // depend on the neighbouring image, nor on the rows past the end of the tensor)
        const unsigned w0 = (unsigned)(pm0 + (long)wm * 256);
        const unsigned wend = w0 + 256 < (unsigned)p.Mtotal ? w0 + 256 : (unsigned)p.Mtotal;
        if (wend > w0) {
          const unsigned hw = (unsigned)(p.H * p.W);
          const int n_lo = __builtin_amdgcn_readfirstlane((int)w4_fdiv(w0, p.dv_hw)),
                    n_hi = __builtin_amdgcn_readfirstlane((int)w4_fdiv(wend - 1, p.dv_hw));
          if (n_lo == n_hi && wend == w0 + 256) {
            float mx = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
              for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fabsf(acc[g][m][r]));
            kocr_amax_update(p.amax_out + n_lo, n < p.Cout ? mx : 0.f);
          } else {
            int l5o = 4 * l5;
            asm volatile("" : "+v"(l5o));  // keeps the 128 row indices below out of the tile loop's preheader (and of scratch)
            for (int ni = n_lo; ni <= n_hi; ++ni) {
              const unsigned i0 = (unsigned)ni * hw, i1 = i0 + hw;
              const int lo_r = (int)((i0 > w0 ? i0 : w0) - w0), hi_r = (int)((i1 < wend ? i1 : wend) - w0);
              float mx = 0.f;
#pragma unroll
              for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                  for (int r = 0; r < 16; ++r) {
                    const int rel = (g * 2 + m) * 32 + (r & 3) + 8 * (r >> 2) + l5o;
                    if (rel >= lo_r && rel < hi_r) mx = fmaxf(mx, fabsf(acc[g][m][r]));
                  }
              kocr_amax_update(p.amax_out + ni, n < p.Cout ? mx : 0.f);
            }
          }
        }
      }
      PROBE_T(2);
      const int ocs4 = p.out_cs * 4;
      const long rem = ((long)p.Mtotal - pm0) * ocs4;  // stores past the end of the tensor are dropped
      const unsigned long long bb = (unsigned long long)(p.out + (pm0 * p.out_cs + p.out_co));
      const unsigned long long bbu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(bb >> 32)) << 32) |
                                     (unsigned)__builtin_amdgcn_readfirstlane((int)bb);
      const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
          (void*)bbu, 0, __builtin_amdgcn_readfirstlane((int)(rem < 0x7FFFFFFFL ? rem : 0x7FFFFFFFL)), 0x00020000);
      const unsigned vo = n < p.Cout ? (unsigned)((4 * l5 * p.out_cs + n) * 4) : 0x80000000u;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int px = ((wm * 4 + g) * 2 + m) * 32 + (r & 3) + 8 * (r >> 2);  // + 4*l5 in vo
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[g][m][r]), ro, vo, px * ocs4, 0);
          }
      PROBE_T(3);
    }
    if constexpr (UP != 0) load_a(a0, As + (gs & 1) * BUF, 0);  // the next tile's first fragments (see compute_step)
  }
  PROBE_TEND(tid == 0, 0, 4);
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
int prepare_dsplit(kocr_ctx* ctx, ConvLayer& L, const float* w, bool w_is_oihw) {
  if (L.Cin % 16 != 0 || L.Cout <= 32 || (L.KH == 3 && L.KW == 3 && L.dil == 1)) return KOCR_OK;
  const int Cin = L.Cin, Cout = L.Cout, ntaps = L.KH * L.KW;
  const int wcls = Cout > 64 ? 128 : 64;
  const int cp = (Cout + wcls - 1) / wcls * wcls;
  const int nt32 = cp / 32;
  std::vector<unsigned short> u((size_t)(Cin / 16) * ntaps * nt32 * 3 * 64 * 8, 0);
  for (int c = 0; c < Cin; ++c)
    for (int tap = 0; tap < ntaps; ++tap)
      for (int o = 0; o < Cout; ++o) {
        const float g = w_is_oihw ? w[((size_t)o * Cin + c) * ntaps + tap] : w[((size_t)tap * Cin + c) * Cout + o];
        // MFMA 32x32x16 B operand: lane = (k >> 3) * 32 + (o & 31) holds k = 8*(lane>>5) + j, j = 0..7
        const int k = c % 16, lane = (k >> 3) * 32 + (o & 31), j = k & 7;
        const size_t step = (size_t)(c / 16) * ntaps + tap;
        unsigned short pc[3];
        kocr_split3_host(g, pc);
        for (int s = 0; s < 3; ++s) u[(((step * nt32 + o / 32) * 3 + s) * 64 + lane) * 8 + j] = pc[s];
      }
  L.ds_cout_pad = cp;
  void* d = nullptr;
  KOCR_TRY(ctx->dev_alloc(&d, u.size() * sizeof(unsigned short)));
  KOCR_HIP(ctx, hipMemcpy(d, u.data(), u.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
  L.d_ds = (unsigned short*)d;

  return KOCR_OK;
}

// the kernel can run this layer on this input (any size)
bool dsplit_usable(const ConvLayer& L, const Tensor& in) {
  static const bool off = getenv("KOCR_DSPLIT") && atoi(getenv("KOCR_DSPLIT")) == 0;
  return !off && L.d_ds && in.cs % 4 == 0 && in.co % 4 == 0 && ((uintptr_t)in.p & 15) == 0 &&
         (size_t)in.pixels() * in.cs < ((size_t)1 << 40);
}

// ... and launch_conv's dispatch prefers it: small GEMMs (the CRNN's dense layers) stay on the fp32 kernel, nothing to
// gain below a few tiles
bool dsplit_applicable(const ConvLayer& L, const Tensor& in) { return dsplit_usable(L, in) && in.pixels() >= 4096; }

template <int WM, int WN, int HALF, int UP = 0, int PAIR = 0>
static int ds_launch(kocr_ctx* ctx, DsParams& p, size_t M) {
  // 48 / 96 KB (bf16x3), 32 / 64 KB (fp16x2); UP: + one 8 KB tap table per consumer wave
  constexpr int NCW = WM * WN;
  constexpr int LDS_BYTES = 2 * (HALF ? 2 : 3) * (8 * WM) * 2 * 256 * 2 + (UP ? NCW * 256 * 32 : 0);
  static std::atomic<bool> attr_done[64];  // per device (one process may hold contexts on several GPUs); a race only repeats the call
  const int dev = ctx->device & 63;
  if (!attr_done[dev]) {
    KOCR_HIP(ctx, hipFuncSetAttribute((const void*)conv_ds_kernel<WM, WN, HALF, UP, PAIR>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr_done[dev] = true;
  }
  static std::atomic<int> n_cus[64];
  if (!n_cus[dev]) {
    hipDeviceProp_t prop;
    KOCR_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
    n_cus[dev] = prop.multiProcessorCount;
  }
  const int n_cu = n_cus[dev];
  const size_t mtiles = (M + 256 * WM - 1) / (256 * WM);
  p.total_tiles = (int)(mtiles * (p.Cout_pad / (32 * WN)));
  const int slots = n_cu * (NCW == 2 ? 2 : 1);  // (1, 2): two blocks per CU
  const int grid = p.total_tiles < slots ? p.total_tiles : slots;
  PROBE_RESET(ctx);
  hipLaunchKernelGGL((conv_ds_kernel<WM, WN, HALF, UP, PAIR>), dim3(grid), dim3(128 * NCW), LDS_BYTES, ctx->stream, p);
  KOCR_HIP(ctx, hipGetLastError());
  {
    char what[64];
    snprintf(what, sizeof what, "conv_ds<%d,%d,%d,pair%d> tiles %d steps %d", WM, WN, UP, PAIR, p.total_tiles, p.nsteps);
    (void)what;
    PROBE_REPORT(ctx, what, grid);
  }
  return KOCR_OK;
}

// `up` (optional, bf16x3 mode only): a [N][h][w][Cout] tensor that is resized bilinearly to the output's H x W and added
// to the accumulators before the epilogue.  craft.cpp uses it to evaluate the decoder's
// conv1x1(concat(resize(y), skip)) (detection.py:106-115, 380-389) as resize(conv1x1_y(y)) + conv1x1_skip(skip): a
// 1x1 convolution commutes with the (linear, per-channel) resize, so the y half of the products runs at a quarter of
// the pixels and the up-sampled tensor is never written.
int launch_conv_dsplit(kocr_ctx* ctx, const ConvLayer& L, const Tensor& in, const Tensor& out, const Tensor* up) {
  const size_t M = in.pixels();
  DsParams p;
  p.in = in.p;
  p.wgt = L.d_ds;
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
  p.Cout_pad = L.ds_cout_pad;
  p.out_cs = out.cs;
  p.out_co = out.co;
  p.relu = L.relu;
  p.KH = L.KH;
  p.KW = L.KW;
  p.dil = L.dil;
  p.nsteps = L.KH * L.KW * (L.Cin / 16);
  p.Mtotal = (int)M;
  p.total_tiles = 0;
  p.amax_out = out.amax;
  const bool half = false;  // round 4: the fp16 arithmetic lives in conv_w43h.hip; the HALF template path is no longer instantiated
  p.up = nullptr;
  p.up_H = p.up_W = 0;
  p.up_sy = p.up_sx = 0.f;
  if (up) {
    if (half || up->C != L.Cout || up->cs != L.Cout || up->co != 0 || up->N != in.N || L.KH != 1 || L.KW != 1 ||
        2 * (size_t)up->H * up->W * L.Cout * 4 >= ((size_t)1 << 31))
      KOCR_FAIL(ctx, KOCR_EINVAL, "conv " + L.name + ": unsupported up-sampled addend");
    p.up = up->p;
    p.up_H = up->H;
    p.up_W = up->W;
    p.up_sy = (float)up->H / (float)in.H;
    p.up_sx = (float)up->W / (float)in.W;
  }
  p.amax_in = nullptr;
  p.w_exp = 0;
  w4_div_magic((unsigned)(in.H * in.W), p.dv_hw);
  const int wcls = L.Cout > 64 ? 128 : 64;
  static const bool per_layer = getenv("KOCR_PROF_LAYERS") != nullptr;
  char nm[64];
  if (per_layer)
    snprintf(nm, sizeof nm, "conv_d%s_%dx%d%s:%s", half ? "h" : "s", wcls == 128 ? 256 : 512, wcls, up ? "_up" : "", L.name.c_str());
  else
    snprintf(nm, sizeof nm, "conv_d%s_%dx%d%s", half ? "h" : "s", wcls == 128 ? 256 : 512, wcls, up ? "_up" : "");
  const double flops = 2.0 * (double)M * L.Kreal * L.Cout;
  const double bytes = 4.0 * ((double)M * L.Cin + (double)M * L.Cout + (double)L.Kreal * L.Cout + (up ? (double)up->pixels() * L.Cout : 0.0));
  ProfScope ps(ctx, nm, flops, bytes);
  // whole 128-byte lines per fetch (PAIR): 1x1 convolutions over an even number of channel groups (A/B against round 5's one
  // K-step per fetch: profiles/r06_ab_notes.txt item 3)
  const bool line_pairs = L.KH == 1 && L.KW == 1 && (L.Cin / 16) % 2 == 0 && in.cs % 32 == 0 && in.co % 32 == 0;
  if (up) {
    // exactly 2x up-sampling with W % 4 == 0: the shared-tap epilogue (UP = 2)
    const bool no_up2 = !ctx->sw.up2x;
    if (!no_up2 && in.H == 2 * up->H && in.W == 2 * up->W && in.W % 4 == 0) {
      if (line_pairs) return wcls == 128 ? ds_launch<1, 4, 0, 2, 1>(ctx, p, M) : ds_launch<2, 2, 0, 2, 1>(ctx, p, M);
      return wcls == 128 ? ds_launch<1, 4, 0, 2>(ctx, p, M) : ds_launch<2, 2, 0, 2>(ctx, p, M);
    }
    return wcls == 128 ? ds_launch<1, 4, 0, 1>(ctx, p, M) : ds_launch<2, 2, 0, 1>(ctx, p, M);
  }
  if (line_pairs) return wcls == 128 ? ds_launch<1, 4, 0, 0, 1>(ctx, p, M) : ds_launch<2, 2, 0, 0, 1>(ctx, p, M);
  return wcls == 128 ? ds_launch<1, 4, 0>(ctx, p, M) : ds_launch<2, 2, 0>(ctx, p, M);
}
