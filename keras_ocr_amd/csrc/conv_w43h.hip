// conv_w43h.hip — the Winograd F(4,3) convolutions of conv_w43.hip on the gfx950 FP16 matrix cores (round 4).
//
// Why: the bf16x3 kernels issue 3.0 matrix-core FLOPs per algorithmic FLOP (F(4,3): 1/2 of the multiplies x 6 split
// products) and sit on the chip's power ceiling, so only fewer products per output make them faster.  Two fp16 pieces per
// operand need three products (a_l b_h + a_h b_l + a_h b_h): 1.5 issued FLOPs per algorithmic FLOP.
//
// Arithmetic (KOCR_SPLIT_F16X2, NP = 2; tests/test_split_arith_cpu.py restates it in numpy):
//   * fp16 has 11 significand bits but only 5 exponent bits, so both operands are first scaled by EXACT powers of two:
//     the input by 2^e with e = 12 - exponent(max |x| of the image), read from the per-image slot its producer maintains
//     (Tensor::amax; the F(4,3) input transform grows a value by at most 5.28, so |V 2^e| < 43 300 < 65 504), the weights by
//     2^wexp[o] per output channel at load time (max |U 2^wexp| in [2^14, 2^15)); the epilogue undoes both in the affine
//     it applies anyway.  Per IMAGE: a result never depends on what else is in the batch.
//   * V 2^e = h + l with h = rn_fp16(V 2^e), l = rn_fp16(V 2^e - h): |V 2^e - h - l| <= 2^-22 |V 2^e| (2^-24 rms) while l is
//     a normal fp16, 2^-25 absolute (scaled units) below -- the second term of the bound tests/test_conv_gpu.py states
//     (elements more than 2^16 below their image's maximum lose low-piece bits); the same for the weights.
//   * the three kept products are exact in the fp16 MFMA and accumulate in fp32; the dropped a_l b_l is <= 2^-22 |ab|.
//   Measured against fp64 (numpy restatement, K = 4608): max 1.3e-7 / rms 2.1e-8 of |x| conv |w| -- the bf16x3 F(4,3)
//   kernel gives 1.9e-7 / 2.9e-8 on the same data.
// KOCR_SPLIT_F16X1 (NP = 1) keeps only h (a_h b_h, 0.5 issued FLOPs per algorithmic FLOP): the reduced-precision fast
// mode, relative error 2^-12 per operand (about 1e-4 of |x| conv |w| on the same probe); never the default.
//
// Kernel structure = conv_w43v_kernel (vertical reuse: tile = 4 rows x 64 columns or 8 rows x 32 columns of ONE image x
// 128 couts, the window's input rows transformed and split once per 16-channel group into LDS, the three vertical taps
// read them back at a row offset; wave wn owns both M-tiles x 32 couts x six points = 192 accumulators; weights straight
// from L2 into registers one (channel group, ky) step ahead), with NP operand planes instead of three, PR = 3 (or 1)
// v_mfma_f32_32x32x16_f16 per point and M-tile instead of six, and the power-of-two scale folded into the constants of the
// input transform (so it costs two extra multiplies per value and channel group, not one per point).
#include "w43_common.h"
#include "probe_clock.h"

namespace {

// V_xi * s with the power of two s folded into the (exact) constants of w4_transform: every product with s is exact.
// The seven scaled constants are plain scalars (kept in registers; a struct of them ended up in scratch memory).
#define W4H_SCALED_CONSTANTS(sc)                                                                       \
  const float k_s = (sc), k_sA = (sc) * W4_A, k_sB = (sc) * W4_B, k_sA2 = (sc) * W4_A2, k_sB2 = (sc) * W4_B2, \
              k_sA2B2 = (sc) * W4_A2B2, k_sA2PB2 = (sc) * W4_A2PB2
template <class T>
__device__ __forceinline__ T w4h_transform(const T (&d)[6], int xi, float k_s, float k_sA, float k_sB, float k_sA2, float k_sB2,
                                           float k_sA2B2, float k_sA2PB2) {
  switch (xi) {
    case 0: return (k_sA2B2 * d[0] - k_sA2PB2 * d[2]) + k_s * d[4];
    case 1: return (k_s * d[4] - k_sB2 * d[2]) + k_sA * (d[3] - W4_B2 * d[1]);
    case 2: return (k_s * d[4] - k_sB2 * d[2]) - k_sA * (d[3] - W4_B2 * d[1]);
    case 3: return (k_s * d[4] - k_sA2 * d[2]) + k_sB * (d[3] - W4_A2 * d[1]);
    case 4: return (k_s * d[4] - k_sA2 * d[2]) - k_sB * (d[3] - W4_A2 * d[1]);
    default: return (k_sA2B2 * d[1] - k_sA2PB2 * d[3]) + k_s * d[5];
  }
}

constexpr int W4H_TOP = 12;  // scaled inputs lie below 2^13: 5.28 x 2^13 < 65 504

}  // namespace

// DBG (developer timing experiments, WRONG results; only instantiated with -DKOCR_DEV_SWITCHES): 1 = no input transform /
// split VALU work, 2 = no weight stream, 4 = no MFMAs, 8 = no LDS operand fetches, 16 = no raw input loads, 32 = no block
// barrier in the K loop
//
// MODE (round 5): 0 = the image tiles exactly (H % 4 == 0, W % 64 == 0, or H % 8 == 0, W % 32 == 0): the code of round 4.
//   1 = RAGGED: any H, W (dense layout).  The tile grid covers ceil(H / 4) x ceil(W / 64) (ceil(H / 8) x ceil(W / 32)) tiles
//       per image; the gather masks the taps whose column lies outside the image (a bit per tap in Geo::ok -- in the dense
//       layout "column W" is the next row's first pixel), rows outside are masked as before, the epilogue drops the stores
//       (full-resolution and pooled) of positions outside the image.  The per-image max-|x| slot then also sees the values
//       computed at those positions (their inputs are the image's own border pixels): still an upper bound that depends
//       on nothing but the image.
//   2 = CELLS: every image is one row of `cells_per_row` cells of cellW columns (cellW % 4 == 0),
//       each holding one independent crop in its columns [0, cellWv) and rows [1, H): row 0 and the columns behind cellWv
//       are ZERO gutters -- exactly the 'same' padding between neighbouring crops -- which this kernel reads as data and
//       writes as zeros.  The recogniser's crop batch (crnn.cpp): 31 x 200 / 15 x 100 / 7 x 50 crops in cells of
//       32 x 208 / 16 x 104 / 8 x 52, so that the vertical-reuse arrangement (and its fused pooling: the zero row on top makes
//       the pooling windows of rows (2 i + 1, 2 i + 2) aligned) applies to them.  One input scale and one max-|x| slot per CELL
//       (slot index image * cells_per_row + cell): a tile must touch at most TWO cells (the launcher takes 8 x 32 tiles where a
//       cell is narrower than 64 columns -- a 64-column tile over 52-column cells can touch three), so a producer thread picks
//       the scale of ITS quad's cell and the epilogue undoes it per accumulator row.
template <int POOL, int GEO, int NP, int DBG = 0, int MODE = 0>
__global__ __launch_bounds__(256) void conv_w43vh_kernel(W4Params p) {
  static_assert(GEO == 1 || GEO == 2, "4 x 64 or 8 x 32 tiles");
  static_assert(!(POOL && GEO == 2), "no fused pooling on 8 x 32 tiles");
  static_assert(NP == 1 || NP == 2, "one or two fp16 pieces");
  static_assert(MODE == 0 || MODE == 1 || MODE == 2, "exact tiling, ragged, cells");
  constexpr int PR = NP == 2 ? 3 : 1;        // products per point and M-tile
  constexpr int NROWS = GEO == 2 ? 10 : 6;  // input rows of the tile's window
  constexpr int QPR = GEO == 2 ? 8 : 16;    // quads per tile row
  constexpr int KHS = QPR * 8;                             // ushorts of one k half of a row: QPR quads x 8 channels
  constexpr int ROW_STRIDE = 2 * KHS;                      // GEO 1: ushorts per input row of a plane
  constexpr int PLANE_R = NROWS * 2 * KHS;                 // one (xi, piece) plane
  constexpr int BUF_R = 6 * NP * PLANE_R;                  // one channel group: 36 / 30 KB (NP = 2)
  constexpr int TCOLS = QPR * 4;                           // tile columns
  extern __shared__ __attribute__((aligned(16))) unsigned short As[];
  float* coef = reinterpret_cast<float*>(As + 2 * BUF_R);  // behind the two channel-group buffers: see w4_stage_coef
  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave = cout sub-tile
  const int l31 = lane & 31, l5 = lane >> 5;
  const int nblk_n = p.Cout_pad >> 7;
  const int total = p.total_tiles;
  const int ncg = p.Cin >> 4;
  const int ns = 3 * ncg;
  const int G = gridDim.x;
  constexpr unsigned OOB = 0x80000000u;
  PROBE_T0();

  // ushort offset of the 16-byte slot (window row w, k half kh, quad q) inside a plane (see conv_w43v_kernel)
  auto slot = [&](int w, int kh, int q) {
    if constexpr (GEO == 2)
      return ((w >> 1) * 2 + kh) * 128 + (w & 1) * 64 + ((q * 8) ^ (kh * 32));
    else
      return w * ROW_STRIDE + kh * KHS + ((q * 8) ^ (kh * 32));
  };
  // first pixel of pixel tile mp (flattened (n, y, x) index), its row and column
  auto tile_org = [&](int mp, int& y0, int& x0, int& nimg_o) -> long {
    if constexpr (MODE != 0) {
      // ragged / cell grids: rq = (image, row block), rq_per_img = ceil(H / 4) (ceil(H / 8))
      const int rq = (int)w4_fdiv((unsigned)mp, p.dv_tpr), cb = mp - rq * p.tiles_per_row;
      const int nimg = (int)w4_fdiv((unsigned)rq, p.dv_rq), ro = rq - nimg * p.rq_per_img;
      y0 = (GEO == 2 ? 8 : 4) * ro;
      x0 = TCOLS * cb;
      nimg_o = nimg;
      return ((long)nimg * p.H + y0) * p.W + x0;
    } else if constexpr (GEO == 2) {
      const int rq = (int)w4_fdiv((unsigned)mp, p.dv_tpr), cb = mp - rq * p.tiles_per_row;  // (image, row octet), column block
      const int ho = p.H >> 3;
      const int nimg = (int)w4_fdiv((unsigned)rq, p.dv_hh), ro = rq - nimg * ho;
      y0 = 8 * ro;
      x0 = 32 * cb;
      return ((long)nimg * p.H + y0) * p.W + x0;
    } else {
      const int rq = (int)w4_fdiv((unsigned)mp, p.dv_tpr), cb = mp - rq * p.tiles_per_row;  // rq: (image, row quad)
      return w4_mtile_pm0<1>(p, 2 * rq * p.tiles_per_row + cb, y0, x0);
    }
  };

  // ---- producer state -----------------------------------------------------------------------------------------
  const int q4 = tid & 3, qd0 = (tid >> 2) & (QPR - 1), r0 = GEO == 2 ? (tid >> 5) : (tid >> 6);
  const int cp = tid & 7;
  const int qd1 = GEO == 2 ? ((tid >> 3) & 7) : ((tid >> 3) & 15);
  const int r1 = GEO == 2 ? 8 + ((tid >> 6) & 1) : 4 + (tid >> 7);
  int ldst[2];
  ldst[0] = slot(r0, q4 >> 1, qd0) + (q4 & 1) * 4;
  ldst[1] = slot(r1, cp >> 2, qd1) + (cp & 3) * 2;
  struct Geo {
    unsigned off0[2];  // byte offset of raw pixel d0 of each item
    unsigned ok;       // bit it: the item's input row lies inside the image (and the tile exists); MODE 1: bit 2 + 6 it + k
                       // set = tap k of item `it` lies in a column outside the image
    int e;             // exponent of the image's input scale 2^e (MODE 2: of the tile's first cell)
    int e1, qb;        // MODE 2: exponent of the tile's second cell; first quad of a tile row that lies in it (QPR: none)
    const float* base;
  };
  auto make_geo = [&](int L, Geo& g, bool& left, bool& right) __attribute__((always_inline)) {
    int mp, nt_unused;
    w4_decode(p, kocr_xcd_remap(L < total ? L : 0, total), nblk_n, mp, nt_unused);
    int y0, x0, nimg = 0;
    const long pm = tile_org(mp, y0, x0, nimg);
    g.base = p.in + (pm * p.in_cs + p.in_co) - (long)(p.W + 1) * p.in_cs;
    // the thread's item coordinates again, from a fresh lane id (kocr_fresh_lane: nothing of this per-tile code is kept
    // alive -- spilled -- across the K loop)
    const int tid = wn * 64 + kocr_fresh_lane();
    const int q4 = tid & 3, qd0 = (tid >> 2) & (QPR - 1), r0 = GEO == 2 ? (tid >> 5) : (tid >> 6);
    const int cp = tid & 7;
    const int qd1 = GEO == 2 ? ((tid >> 3) & 7) : ((tid >> 3) & 15);
    const int r1 = GEO == 2 ? 8 + ((tid >> 6) & 1) : 4 + (tid >> 7);
    g.off0[0] = (unsigned)(((r0 * p.W + 4 * qd0) * p.in_cs + q4 * 4) * 4);
    g.off0[1] = (unsigned)(((r1 * p.W + 4 * qd1) * p.in_cs + cp * 2) * 4);
    g.ok = ((L < total && (unsigned)(y0 - 1 + r0) < (unsigned)p.H) ? 1u : 0u) |
           ((L < total && (unsigned)(y0 - 1 + r1) < (unsigned)p.H) ? 2u : 0u);
    g.e1 = 0;
    g.qb = QPR;
    if constexpr (MODE == 2) {
      const int j0 = (int)w4_fdiv((unsigned)x0, p.dv_wc);
      const int j1 = j0 + 1 < p.cells_per_row ? j0 + 1 : j0;
      const int c0 = __builtin_amdgcn_readfirstlane(nimg * p.cells_per_row + j0), c1 = __builtin_amdgcn_readfirstlane(nimg * p.cells_per_row + j1);
      g.e = kocr_scale_exp_bits(kocr_sload(p.amax_in + c0), W4H_TOP);
      g.e1 = kocr_scale_exp_bits(kocr_sload(p.amax_in + c1), W4H_TOP);
      const int qb = ((j0 + 1) * p.cellW - x0) >> 2;
      g.qb = qb < QPR ? qb : QPR;
    } else if constexpr (MODE == 1) {
      g.e = kocr_scale_exp_bits(kocr_sload(p.amax_in + __builtin_amdgcn_readfirstlane(nimg)), W4H_TOP);
    } else {
      g.e = kocr_scale_exp_bits(kocr_sload(p.amax_in + __builtin_amdgcn_readfirstlane((int)w4_fdiv((unsigned)pm, p.dv_hw))), W4H_TOP);
    }
    if constexpr (MODE == 1) {
      // the taps' columns one by one: x0 - 1 + 4 qd + k must lie in [0, W)
      const int c0 = x0 - 1 + 4 * qd0, c1 = x0 - 1 + 4 * qd1;
      unsigned cm = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k)
        cm |= ((unsigned)(c0 + k) < (unsigned)p.W ? 0u : (4u << k)) | ((unsigned)(c1 + k) < (unsigned)p.W ? 0u : (256u << k));
      g.ok |= cm;
      left = right = false;
    } else {
      // d0 / d5 are column zero padding only at the image edges (W is a multiple of the tile width)
      left = x0 == 0;
      right = x0 + TCOLS >= p.W;
    }
  };
  Geo gc, gn;
  bool lc, rc, ln, rn;
  int ld_cg = 0;  // channel group of the NEXT load inside its tile
  bool ld_next = false;
  auto load_item0 = [&](v4f (&raw)[6]) __attribute__((always_inline)) {
    if constexpr (DBG & 16) return;
    const int soff = ld_cg * 64;
    const unsigned okb = ld_next ? gn.ok : gc.ok;
    const bool ok = okb & 1u;
    const unsigned off0 = (ld_next ? gn.off0[0] : gc.off0[0]) | (ok ? 0u : OOB);
    const bool left = (ld_next ? ln : lc) && qd0 == 0, right = (ld_next ? rn : rc) && qd0 == QPR - 1;
    const unsigned stride = (unsigned)(p.in_cs * 4);
    const __amdgpu_buffer_rsrc_t rsrc = w4_rsrc(ld_next ? gn.base : gc.base, 0x80000000u);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const unsigned padk = MODE == 1 ? ((okb << (29 - k)) & OOB)  // bit 2 + k -> bit 31
                                      : ((k == 0 ? (left ? OOB : 0u) : 0u) | (k == 5 ? (right ? OOB : 0u) : 0u));
      if constexpr (DBG & 128)  // the same instructions and bytes, lane-contiguous (1 KB per instruction), cache-resident
        raw[k] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(w4_rsrc(p.in, 0x80000000u), (unsigned)((tid & 63) * 16 + (tid >> 6) * 8192 + k * 1024), soff & 0xFFF, 0));
      else if constexpr (DBG & 64)  // the same instructions and bytes, but from a 256 KB cache-resident window
        raw[k] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(w4_rsrc(p.in, 0x80000000u), ((off0 + k * stride) | padk) & 0x8003FFF0u, soff & 0xFFF, 0));
      else
      raw[k] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (off0 + k * stride) | padk, soff, 0));
    }
  };
  auto load_item1 = [&](v2f (&raw)[6]) __attribute__((always_inline)) {
    if constexpr (DBG & 16) return;
    const int soff = ld_cg * 64;
    const unsigned okb = ld_next ? gn.ok : gc.ok;
    const bool ok = (okb >> 1) & 1u;
    const unsigned off0 = (ld_next ? gn.off0[1] : gc.off0[1]) | (ok ? 0u : OOB);
    const bool left = (ld_next ? ln : lc) && qd1 == 0, right = (ld_next ? rn : rc) && qd1 == QPR - 1;
    const unsigned stride = (unsigned)(p.in_cs * 4);
    const __amdgpu_buffer_rsrc_t rsrc = w4_rsrc(ld_next ? gn.base : gc.base, 0x80000000u);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const unsigned padk = MODE == 1 ? ((okb << (23 - k)) & OOB)  // bit 8 + k -> bit 31
                                      : ((k == 0 ? (left ? OOB : 0u) : 0u) | (k == 5 ? (right ? OOB : 0u) : 0u));
      if constexpr (DBG & 128)
        raw[k] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(w4_rsrc(p.in, 0x80000000u), (unsigned)((tid & 63) * 8 + (tid >> 6) * 8192 + 6144 + k * 512), soff & 0xFFF, 0));
      else if constexpr (DBG & 64)
        raw[k] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(w4_rsrc(p.in, 0x80000000u), ((off0 + k * stride) | padk) & 0x8003FFF8u, soff & 0xFFF, 0));
      else
      raw[k] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (off0 + k * stride) | padk, soff, 0));
    }
  };
  auto advance = [&]() __attribute__((always_inline)) {
    const bool wrap = ld_cg == ncg - 1;
    ld_cg = wrap ? 0 : ld_cg + 1;
    ld_next = ld_next || wrap;
  };
  // sc = the scale(s) 2^e of the channel group being PRODUCED (it may already belong to the next tile's image): one per tile,
  // or (MODE 2) those of the tile's two cells and the first quad of the second one -- a thread takes its quad's
  // (plain scalars: a struct of them captured by these lambdas ends up in scratch memory)
  auto produce4 = [&](const v4f (&d)[6], unsigned short* bufp, int xi, int it, float sc0, float sc1, int scq) __attribute__((always_inline)) {
    const float sk = MODE == 2 ? (qd0 < scq ? sc0 : sc1) : sc0;
    unsigned short* dst = bufp + xi * NP * PLANE_R + ldst[it];
    if constexpr (DBG & 1) {
      const u2v r = __builtin_bit_cast(u2v, __builtin_shufflevector(d[xi], d[xi], 0, 1));
      *reinterpret_cast<u2v*>(dst) = r;
      if constexpr (NP == 2) *reinterpret_cast<u2v*>(dst + PLANE_R) = r;
      return;
    }
    v4f V;
    if constexpr (MODE == 2) {
      // a per-THREAD scale: folded into the constants it would hold seven more registers across the step (they spill);
      // scaling the transformed value costs one multiply per component and gives the same bits (a power of two)
      V = w4_transform(d, xi) * sk;
    } else {
      W4H_SCALED_CONSTANTS(sk);
      V = w4h_transform(d, xi, k_s, k_sA, k_sB, k_sA2, k_sB2, k_sA2B2, k_sA2PB2);
    }
    if constexpr (NP == 2) {
      u2v h, l;
      kocr_split4_h(V, h, l);
      *reinterpret_cast<u2v*>(dst) = h;
      *reinterpret_cast<u2v*>(dst + PLANE_R) = l;
    } else {
      *reinterpret_cast<u2v*>(dst) = u2v{__builtin_bit_cast(unsigned, hf2{(_Float16)V[0], (_Float16)V[1]}),
                                         __builtin_bit_cast(unsigned, hf2{(_Float16)V[2], (_Float16)V[3]})};
    }
  };
  auto produce2 = [&](const v2f (&d)[6], unsigned short* bufp, int xi, float sc0, float sc1, int scq) __attribute__((always_inline)) {
    const float sk = MODE == 2 ? (qd1 < scq ? sc0 : sc1) : sc0;
    unsigned short* dst = bufp + xi * NP * PLANE_R + ldst[1];
    if constexpr (DBG & 1) {
      const unsigned r = __float_as_uint(d[xi][0]);
      *reinterpret_cast<unsigned*>(dst) = r;
      if constexpr (NP == 2) *reinterpret_cast<unsigned*>(dst + PLANE_R) = r;
      return;
    }
    v2f V;
    if constexpr (MODE == 2) {
      V = w4_transform(d, xi) * sk;
    } else {
      W4H_SCALED_CONSTANTS(sk);
      V = w4h_transform(d, xi, k_s, k_sA, k_sB, k_sA2, k_sB2, k_sA2B2, k_sA2PB2);
    }
    if constexpr (NP == 2) {
      unsigned h, l;
      kocr_split2_h(V, h, l);
      *reinterpret_cast<unsigned*>(dst) = h;
      *reinterpret_cast<unsigned*>(dst + PLANE_R) = l;
    } else {
      *reinterpret_cast<unsigned*>(dst) = __builtin_bit_cast(unsigned, hf2{(_Float16)V[0], (_Float16)V[1]});
    }
  };
  v4f raw0[6];
  v2f raw1[6];

  // ---- consumer state ------------------------------------------------------------------------------------------
  // weights: [16-ch group][ky][32-cout tile][xi][2 pieces][lane][8] fp16 (NP = 1 reads piece 0 only), fetched with raw
  // buffer loads (lane offset in one VGPR, everything else scalar) TWO (channel group, ky) steps ahead into a ring of two
  // register sets: one step is only 36 MFMAs (about 0.6 us) long in this arithmetic, less than an L2 round trip under load
  const int ntiles32 = p.Cout_pad >> 5;
  const unsigned w_step = (unsigned)ntiles32 * 12 * 64 * 16;  // bytes per (channel group, ky) step
  const __amdgpu_buffer_rsrc_t wrsrc = w4_rsrc(reinterpret_cast<const float*>(p.wgt), 0x7FFFFFFFu);
  const unsigned wlane = (unsigned)lane * 16u;
  auto w_tile = [&](int nt) { return (unsigned)__builtin_amdgcn_readfirstlane((nt * 4 + wn) * 12 * 64 * 16); };
  hf8 bw[2][6][NP];
  f16v acc[6][2];
  const int a_lane = GEO == 2 ? slot(l31 >> 3, l5, l31 & 7) : slot(l31 >> 4, l5, l31 & 15);
  const int a_lane1 = GEO == 2 ? slot((l31 >> 3) + 1, l5, l31 & 7) : 0;
  auto load_a = [&](hf8 (&a)[2][NP], const unsigned short* bufp, int ky, int xi) __attribute__((always_inline)) {
    if constexpr (DBG & 8) return;
    const unsigned short* plane = bufp + xi * NP * PLANE_R;
#pragma unroll
    for (int s = NP - 1; s >= 0; --s)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        int off;
        if constexpr (GEO == 2) {
          const int c = ky + 4 * m;
          off = ((c & 1) ? a_lane1 : a_lane) + (c >> 1) * 256;
        } else {
          off = a_lane + (ky + 2 * m) * ROW_STRIDE;
        }
        a[m][s] = *reinterpret_cast<const hf8*>(plane + s * PLANE_R + off);
      }
  };
  auto mfma_pt = [&](const hf8 (&a)[2][NP], const hf8 (&b)[NP], int xi) __attribute__((always_inline)) {
    if constexpr (DBG & 4) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int s = 0; s < NP; ++s) acc[xi][m][s] += __builtin_bit_cast(v4f, a[m][s])[0] + __builtin_bit_cast(v4f, b[s])[0];
      return;
    }
    // smallest terms first; the two M-tiles alternate so consecutive MFMAs are independent
    if constexpr (NP == 2) {
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[xi][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m][1], b[0], acc[xi][m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[xi][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m][0], b[1], acc[xi][m], 0, 0, 0);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) acc[xi][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m][0], b[0], acc[xi][m], 0, 0, 0);
  };
  // One (channel group, ky) step: consume rows ky .. (+ M-tile offset) of `bufc` (6 points x 2 PR MFMAs); KY = 0 / 1 also
  // transforms item 0 / 1 of the NEXT channel group into `bufn`, one point per MFMA group.  The step uses ring slot SLOT of
  // the weights; the weights of the step after next (byte offset w_next) replace them point by point.  a0 holds point 0 of this step on entry and point 0 of the next step on
  // exit; for KY = 2 the next step lives in `bufn`, published by the block barrier before the last point's MFMAs.
  hf8 a0[2][NP], a1[2][NP];
  auto step = [&](auto ky_c, auto slot_c, const unsigned short* bufc, unsigned short* bufn, unsigned w_next, float sc0, float sc1, int scq) __attribute__((always_inline)) {
    constexpr int KY = decltype(ky_c)::value;
    constexpr int SLOT = decltype(slot_c)::value;
    auto load_b = [&](int xi) __attribute__((always_inline)) {
      if constexpr (DBG & 2) return;
#pragma unroll
      for (int s = 0; s < NP; ++s)
        bw[SLOT][xi][s] = __builtin_bit_cast(hf8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, (int)(w_next + (unsigned)(xi * 2 + s) * 1024u), 0));
    };
    auto produce = [&](int xi) __attribute__((always_inline)) {
      if constexpr (KY == 0) produce4(raw0, bufn, xi, 0, sc0, sc1, scq);
      if constexpr (KY == 1) produce2(raw1, bufn, xi, sc0, sc1, scq);
    };
    auto interleave = [&]() __attribute__((always_inline)) {
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * NP, 0);  // the LDS fetches of the next point first
      if constexpr (KY < 2) {
        constexpr int VPG = (KY == 0 ? 30 : 16) / (2 * PR - 1) + 1;  // VALU per MFMA gap
#pragma unroll
        for (int i = 0; i < 2 * PR - 1; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, VPG, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x200, NP, 0);  // the point's LDS stores
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * PR, 0);
      }
    };
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      __builtin_amdgcn_sched_barrier(0);
      load_a(a1, bufc, KY, 2 * q + 1);
      produce(2 * q);
      mfma_pt(a0, bw[SLOT][2 * q], 2 * q);
      interleave();
      __builtin_amdgcn_sched_barrier(0);
      load_b(2 * q);
      __builtin_amdgcn_sched_barrier(0);
      if (q < 2) {
        load_a(a0, bufc, KY, 2 * q + 2);
        produce(2 * q + 1);
        mfma_pt(a1, bw[SLOT][2 * q + 1], 2 * q + 1);
        interleave();
      } else if constexpr (KY < 2) {
        load_a(a0, bufc, KY + 1, 0);
        produce(5);
        mfma_pt(a1, bw[SLOT][5], 5);
        interleave();
      } else {
        if constexpr (!(DBG & 32)) __syncthreads();  // the next channel group is complete in bufn, bufc is free
        load_a(a0, bufn, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_pt(a1, bw[SLOT][5], 5);
      }
      __builtin_amdgcn_sched_barrier(0);
      load_b(2 * q + 1);
    }
  };

  // ---- prologue ------------------------------------------------------------------------------------------------
  w4_stage_coef(p, coef, tid);
  make_geo(blockIdx.x, gc, lc, rc);
  make_geo(blockIdx.x + G, gn, ln, rn);
  load_item0(raw0);
  load_item1(raw1);
  advance();  // channel group 0 loaded
  {
    int mp0, nt0;
    w4_decode(p, kocr_xcd_remap(blockIdx.x, total), nblk_n, mp0, nt0);
    const unsigned w0 = w_tile(nt0);
#pragma unroll
    for (int st = 0; st < 2; ++st)  // steps 0 and 1 of the first tile (ns >= 6)
#pragma unroll
      for (int xi = 0; xi < 6; ++xi)
#pragma unroll
        for (int s = 0; s < NP; ++s)
          bw[st][xi][s] = __builtin_bit_cast(hf8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, (int)(w0 + st * w_step + (unsigned)(xi * 2 + s) * 1024u), 0));
  }
#pragma unroll
  for (int xi = 0; xi < 6; ++xi) {
    produce4(raw0, As, xi, 0, kocr_pow2(gc.e), kocr_pow2(gc.e1), gc.qb);
    produce2(raw1, As, xi, kocr_pow2(gc.e), kocr_pow2(gc.e1), gc.qb);
  }
  load_item0(raw0);
  load_item1(raw1);
  advance();  // channel group 1 loaded
  __syncthreads();
  load_a(a0, As, 0, 0);

  // bins: [0] K loop  [1] next tile's geometry  [2] epilogue arithmetic + amax  [3] store issue  (marks INSIDE the K loop
  // slowed it six-fold: the probe is for per-tile phases only)
  PROBE_RESTART();
  for (int L = blockIdx.x; L < total; L += G) {
    int mp, nt, mp_n, nt_n;
    w4_decode(p, kocr_xcd_remap(L, total), nblk_n, mp, nt);
    w4_decode(p, kocr_xcd_remap(L + G < total ? L + G : L, total), nblk_n, mp_n, nt_n);
    const unsigned w_ptr = w_tile(nt), w_after = w_tile(nt_n);
    // byte offset of step s of this tile; steps ns, ns + 1 are the next tile's first two
    auto w_at = [&](int s) { return s < ns ? w_ptr + (unsigned)s * w_step : w_after + (unsigned)(s - ns) * w_step; };
    const float s_cur = kocr_pow2(gc.e), s_nxt = kocr_pow2(gn.e);
    const float s_cur1 = kocr_pow2(gc.e1), s_nxt1 = kocr_pow2(gn.e1);  // MODE 2: the second cell's
    const int qb_nxt = gn.qb;
    const float unscale = kocr_pow2(-gc.e);  // this tile's accumulators carry 2^(e + wexp[o]); wexp is folded into pre_a
    const float unscale1 = kocr_pow2(-gc.e1);  // MODE 2: ... of the quads in the tile's second cell
    const int qb_cur = gc.qb;
#pragma unroll
    for (int x = 0; x < 6; ++x)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][m][r] = 0.f;
    PROBE_T(5);  // bin [5] accumulator zeroing, [4] the drain of the previous tile's stores (+ every load in flight)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    PROBE_T(4);
    for (int cg = 0; cg < ncg; cg += 2) {
      // even channel group: consume buffer 0, produce the odd one (this tile's) into buffer 1
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>;
      step(I0{}, I0{}, As, As + BUF_R, w_at(3 * cg + 2), s_cur, s_cur1, qb_cur);
      load_item0(raw0);
      step(I1{}, I1{}, As, As + BUF_R, w_at(3 * cg + 3), s_cur, s_cur1, qb_cur);
      load_item1(raw1);
      advance();
      step(I2{}, I0{}, As, As + BUF_R, w_at(3 * cg + 4), s_cur, s_cur1, qb_cur);
      // odd channel group: consume buffer 1, produce the next even one (the next tile's first after the last pair) into 0
      const bool same = cg + 2 < ncg;
      const float s_odd = same ? s_cur : s_nxt, s_odd1 = same ? s_cur1 : s_nxt1;
      const int qb_odd = same ? qb_cur : qb_nxt;
      step(I0{}, I1{}, As + BUF_R, As, w_at(3 * cg + 5), s_odd, s_odd1, qb_odd);
      load_item0(raw0);
      step(I1{}, I0{}, As + BUF_R, As, w_at(3 * cg + 6), s_odd, s_odd1, qb_odd);
      load_item1(raw1);
      advance();
      step(I2{}, I1{}, As + BUF_R, As, w_at(3 * cg + 7), s_odd, s_odd1, qb_odd);
    }
    PROBE_T(0);

    // ---- epilogue --------------------------------------------------------------------------------------------------
    {
      // every lane term of the epilogue derives from a fresh lane id (kocr_fresh_lane), the coefficients come from LDS and
      // the output slots are read here, long before they are compared: no memory round trip is waited for in a tile's epilogue
      const int lane_e = kocr_fresh_lane();
      const int l31e = lane_e & 31, l5e = lane_e >> 5;
      const int n = (nt * 4 + wn) * 32 + l31e;
      int ty0, tx0, timg = 0;
      const long tpm = tile_org(mp, ty0, tx0, timg);
      // slot index of the tile's image (MODE 2: of its first cell, nimg1 = of its second one)
      int nimg, nimg1 = 0, cell0_x = 0;
      if constexpr (MODE == 2) {
        const int j0 = (int)w4_fdiv((unsigned)tx0, p.dv_wc);
        const int j1 = j0 + 1 < p.cells_per_row ? j0 + 1 : j0;
        nimg = __builtin_amdgcn_readfirstlane(timg * p.cells_per_row + j0);
        nimg1 = __builtin_amdgcn_readfirstlane(timg * p.cells_per_row + j1);
        cell0_x = j0 * p.cellW;
      } else if constexpr (MODE == 1) {
        nimg = __builtin_amdgcn_readfirstlane(timg);
      } else {
        nimg = __builtin_amdgcn_readfirstlane((int)w4_fdiv((unsigned)tpm, p.dv_hw));  // the tile lies inside one image
      }
      const unsigned seen_out = p.amax_out ? kocr_amax_peek(p.amax_out + nimg) : 0u;
      const unsigned seen_pool = p.amax_pool ? kocr_amax_peek(p.amax_pool + nimg) : 0u;
      unsigned seen_out1 = 0u, seen_pool1 = 0u;
      if constexpr (MODE == 2) {
        seen_out1 = p.amax_out ? kocr_amax_peek(p.amax_out + nimg1) : 0u;
        seen_pool1 = p.amax_pool ? kocr_amax_peek(p.amax_pool + nimg1) : 0u;
      }
      const float pa = coef[n] * unscale, pb = coef[p.Cout_pad + n];  // pre_a here = pre_a 2^-wexp[o] (ConvLayer::d_pre_a_h)
      const float pa1 = coef[n] * unscale1;                             // MODE 2: for the quads of the tile's second cell
      const bool has_post = p.post_a != nullptr;
      const float qa = coef[2 * p.Cout_pad + n], qb = coef[3 * p.Cout_pad + n];
      const bool live = n < p.Cout;
      const float lo = p.relu ? 0.f : -INFINITY;
      auto act = [&](float v, float a) { return fmaxf(v * a + pb, lo); };
      // MODE 2: quad column (inside the tile row) of accumulator register r of M-tile m, and whether it lies in the tile's
      // first row: 4 x 64 tiles: M-tile = 2 rows x 16 quads, quad (r & 3) + 8 ((r >> 2) & 1) + 4 l5 of row r >> 3;
      // 8 x 32 tiles: M-tile = 4 rows x 8 quads, quad (r & 3) + 4 l5 of row r >> 2
      const int l5q = 4 * l5e;
      auto qx_of = [&](int r) { return GEO == 2 ? (r & 3) + l5q : (r & 3) + 8 * ((r >> 2) & 1) + l5q; };
      auto first_row = [&](int m, int r) { return m == 0 && (GEO == 2 ? (r >> 2) == 0 : r < 8); };
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float par = pa;
          if constexpr (MODE == 2) par = qx_of(r) < qb_cur ? pa : pa1;
          const float m0 = acc[0][m][r], m1 = acc[1][m][r], m2 = acc[2][m][r], m3 = acc[3][m][r], m4 = acc[4][m][r],
                      m5 = acc[5][m][r];
          const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
          acc[0][m][r] = act((m0 + s12) + s34, par);
          acc[1][m][r] = act(W4_A * d12 + W4_B * d34, par);
          acc[2][m][r] = act(W4_A2 * s12 + W4_B2 * s34, par);
          acc[3][m][r] = act((W4_A3 * d12 + W4_B3 * d34) + m5, par);
        }
      if (has_post) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][m][r] = acc[j][m][r] * qa + qb;
      }
      if constexpr (MODE == 2) {
        // the gutters are written as zeros: row 0 of the image (every cell's top row) and the columns behind cellWv
        const bool top = ty0 == 0;  // uniform
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int qx = qx_of(r);
          const int cx = tx0 + 4 * qx - (qx < qb_cur ? cell0_x : cell0_x + p.cellW);  // column inside the quad's cell
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool v = cx + j < p.cellWv;
#pragma unroll
            for (int m = 0; m < 2; ++m) acc[j][m][r] = (v && !(top && first_row(m, r))) ? acc[j][m][r] : 0.f;
          }
        }
      }
      int ocs4 = p.out_cs * 4;
      asm volatile("" : "+s"(ocs4));
      int pcs4 = p.pool_cs * 4;
      asm volatile("" : "+s"(pcs4));
      if (p.amax_out || p.amax_pool) {
        if constexpr (MODE == 2) {
          float mx0 = 0.f, mx1 = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float mq = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int m = 0; m < 2; ++m) mq = fmaxf(mq, fabsf(acc[j][m][r]));
            const bool first = qx_of(r) < qb_cur;
            mx0 = fmaxf(mx0, first ? mq : 0.f);
            mx1 = fmaxf(mx1, first ? 0.f : mq);
          }
          mx0 = live ? mx0 : 0.f;
          mx1 = live ? mx1 : 0.f;
          if (p.amax_out) kocr_amax_update_known(p.amax_out + nimg, mx0, seen_out);
          if (p.amax_pool) kocr_amax_update_known(p.amax_pool + nimg, mx0, seen_pool);
          if (qb_cur < QPR) {  // uniform: the tile has a second cell
            if (p.amax_out) kocr_amax_update_known(p.amax_out + nimg1, mx1, seen_out1);
            if (p.amax_pool) kocr_amax_update_known(p.amax_pool + nimg1, mx1, seen_pool1);
          }
        } else {
          float mx = 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
              for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fabsf(acc[j][m][r]));
          mx = live ? mx : 0.f;
          if (p.amax_out) kocr_amax_update_known(p.amax_out + nimg, mx, seen_out);
          if (p.amax_pool) kocr_amax_update_known(p.amax_pool + nimg, mx, seen_pool);
        }
      }
      PROBE_T(2);
      if constexpr (GEO == 2) {
        // M-tile m = rows 4 m .. 4 m + 3 x 8 quads: accumulator register r of lane half l5 is row r >> 2, quad (r & 3) + 4 l5
        const __amdgpu_buffer_rsrc_t ro = w4_rsrc(p.out + (tpm * p.out_cs + p.out_co), 0x7FFFFFFFu);
        const unsigned vo = live ? (unsigned)((16 * l5e * p.out_cs + n) * 4) : OOB;  // 4 quads = 16 px per l5
        const int wlim = p.W - tx0 - 16 * l5e;  // MODE 1: columns of the image right of this lane half's first one
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int px = (4 * m + (r >> 2)) * p.W + 4 * (r & 3);
            if (MODE == 1 && ty0 + 4 * m + (r >> 2) >= p.H) continue;  // uniform: the row lies below the image
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const unsigned vom = MODE == 1 ? (4 * (r & 3) + j < wlim ? vo : OOB) : vo;
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[j][m][r]), ro, vom, (px + j) * ocs4, 0);
            }
          }
      } else {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          // M-tile m: 2 rows x 64 columns, two rows below M-tile 0
          const long pm = tpm + (long)2 * m * p.W;
          const int x0 = tx0;
          if (!POOL || p.write_full) {
            const __amdgpu_buffer_rsrc_t ro = w4_rsrc(p.out + (pm * p.out_cs + p.out_co), 0x7FFFFFFFu);
            const unsigned vo = live ? (unsigned)((16 * l5e * p.out_cs + n) * 4) : OOB;  // 4 quads = 16 px per l5
            if constexpr (MODE == 1) {
              const int wlim = p.W - tx0 - 16 * l5e;  // columns of the image right of this lane half's first one
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                if (ty0 + 2 * m + h >= p.H) continue;  // uniform: the row lies below the image
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                  const int px = 4 * ((r & 3) + 8 * (r >> 2));
#pragma unroll
                  for (int j = 0; j < 4; ++j)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[j][m][r + 8 * h]), ro, px + j < wlim ? vo : OOB,
                                                          (px + j + h * p.W) * ocs4, 0);
                }
              }
            } else {
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
          }
          if constexpr (POOL) {
            // 2x2 max: rows y (r) and y+1 (r+8), columns (0,1) and (2,3) of the quad
            long pp0;
            if constexpr (MODE == 1)  // floor pooling of any H, W: (nimg (H >> 1) + y0 / 2 + m) (W >> 1) + x0 / 2
              pp0 = ((long)timg * (p.H >> 1) + (ty0 >> 1) + m) * (p.W >> 1) + (x0 >> 1);
            else
              pp0 = ((pm - x0) >> 2) + (x0 >> 1);  // (nimg H/2 + y0/2) W/2 + x0/2: H, W even
            const __amdgpu_buffer_rsrc_t rp = w4_rsrc(p.pool_out + (pp0 * p.pool_cs + p.pool_co), 0x7FFFFFFFu);
            const unsigned vp = live ? (unsigned)((8 * l5e * p.pool_cs + n) * 4) : OOB;  // 4 quads = 8 pooled px per l5
            if (MODE == 1 && (ty0 >> 1) + m >= (p.H >> 1)) continue;  // uniform: no pooled row here
            const int plim = (p.W >> 1) - (x0 >> 1) - 8 * l5e;        // MODE 1: pooled columns right of this lane half's first
            const bool ztop = MODE == 2 && ty0 == 0 && m == 0;        // MODE 2: pooled row 0 is the pooled cells' zero row
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              const int pq = 2 * ((r & 3) + 8 * (r >> 2));
              float v0 = fmaxf(fmaxf(acc[0][m][r], acc[1][m][r]), fmaxf(acc[0][m][r + 8], acc[1][m][r + 8]));
              float v1 = fmaxf(fmaxf(acc[2][m][r], acc[3][m][r]), fmaxf(acc[2][m][r + 8], acc[3][m][r + 8]));
              if constexpr (MODE == 2) {
                v0 = ztop ? 0.f : v0;
                v1 = ztop ? 0.f : v1;
              }
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v0), rp, (MODE != 1 || pq < plim) ? vp : OOB, pq * pcs4, 0);
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v1), rp, (MODE != 1 || pq + 1 < plim) ? vp : OOB, (pq + 1) * pcs4, 0);
            }
          }
        }
      }
      PROBE_T(3);
    }
    // the geometry of the tile after next, behind the stores (their drain at the head of the next K loop then has this
    // arithmetic in front of it; same-box A/B against the order geometry -> epilogue: 26.03 vs 26.07 ms per 8 x 1536^2, no
    // measurable difference)
    gc = gn;
    lc = ln;
    rc = rn;
    make_geo(L + 2 * G, gn, ln, rn);
    ld_next = false;
    PROBE_T(1);
  }
  PROBE_TEND(tid == 0, 0, 6);
}

// ===================================================================================================
// conv_w43rh_kernel -- the 64-cout row-reuse arrangement (conv_w43r_kernel, GEO 1: tile = 4 rows x 64 columns of one
// image x 64 couts; wave (ph, wn) owns both M-tiles x 32 couts x the three points 3 ph .. 3 ph + 2 = 96 accumulators; the
// partner waves exchange their partial output transforms through LDS) in fp16 arithmetic: NP planes per point, PR = 3 (1)
// products, the per-image power-of-two scale folded into the input transform, the per-cout weight scale into pre_a.
// CRAFT's slice1.3 (64 -> 64 at full resolution) and upconv3.conv.3.
// ===================================================================================================
// MODE 1 (round 5) = RAGGED, any H, W: see conv_w43vh_kernel (column bits in Geo::ok, masked stores).
// OCC = 2 (round 6, the fp16x2 default): TWO blocks per CU (two waves per SIMD, 256 registers each) -- one block's loads, transform
// and epilogue run under the other's matrix products, and the other wave hides what in-order vmcnt exposes: a weight fragment is
// fetched three MFMA groups ahead of its use and the raw pixels (HBM) are issued in between, so the wait for a fragment also
// waits for the raw loads issued before it (profiles/r06_ab_notes.txt items 1, 8).  LDS 2 x 36 KB + coefficients: the epilogue
// exchanges its partial output transforms in two halves of 32 KB through the free K-loop buffer instead of 64 KB behind it.
template <int POOL, int NP, int MODE = 0, int OCC = 1>
__global__ __launch_bounds__(256, OCC) void conv_w43rh_kernel(W4Params p) {
  static_assert(MODE == 0 || MODE == 1, "exact tiling or ragged");
  constexpr int NBW = 3;  // weight fragments held: one (ky, point) row
  constexpr int PR = NP == 2 ? 3 : 1;
  constexpr int NROWS = 6, QPR = 16, KHS = QPR * 8, ROW_STRIDE = 2 * KHS, PLANE_R = NROWS * ROW_STRIDE;
  constexpr int BUF_R = 6 * NP * PLANE_R;  // one channel group: 36 KB (NP = 2)
  constexpr int TCOLS = QPR * 4;
  extern __shared__ __attribute__((aligned(16))) unsigned short As[];
  // behind the buffer and the exchange area (OCC = 2: behind the two K-loop buffers): w4_stage_coef
  float* coef = reinterpret_cast<float*>(OCC == 2 ? As + 2 * BUF_R : As + BUF_R + 4 * 16 * 64 * 8);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, ph = wave >> 1;  // cout half, point half (points 3 ph .. 3 ph + 2)
  const int l31 = lane & 31, l5 = lane >> 5;
  const int total = p.total_tiles;  // pixel tiles; one cout block
  const int ncg = p.Cin >> 4;
  const int G = gridDim.x;
  constexpr unsigned OOB = 0x80000000u;

  auto tile_mt = [&](int mp, int m) {
    const int rq = (int)w4_fdiv((unsigned)mp, p.dv_tpr), cb = mp - rq * p.tiles_per_row;  // rq: (image, row quad)
    return (2 * rq + m) * p.tiles_per_row + cb;
  };
  // first pixel of M-tile m (2 rows x 64 columns) of pixel tile mp, its row, column and image
  auto mtile_org = [&](int mp, int m, int& y0, int& x0, int& nimg_o) -> long {
    if constexpr (MODE == 1) {
      const int rq = (int)w4_fdiv((unsigned)mp, p.dv_tpr), cb = mp - rq * p.tiles_per_row;
      const int nimg = (int)w4_fdiv((unsigned)rq, p.dv_rq), ro = rq - nimg * p.rq_per_img;  // rq_per_img = ceil(H / 4)
      y0 = 4 * ro + 2 * m;
      x0 = TCOLS * cb;
      nimg_o = nimg;
      return ((long)nimg * p.H + y0) * p.W + x0;
    } else {
      const long pm = w4_mtile_pm0<1>(p, tile_mt(mp, m), y0, x0);
      nimg_o = (int)w4_fdiv((unsigned)pm, p.dv_hw);
      return pm;
    }
  };

  // ---- producer state (conv_w43r_kernel GEO 1) ------------------------------------------------------------------
  const int q4 = tid & 3, qd0 = (tid >> 2) & (QPR - 1), r0 = tid >> 6;
  const int cp = tid & 7, qd1 = (tid >> 3) & 15, r1 = 4 + (tid >> 7);
  int ldst[2];
  ldst[0] = r0 * ROW_STRIDE + (q4 >> 1) * KHS + (((qd0 * 8) ^ ((q4 >> 1) * 32)) + (q4 & 1) * 4);
  ldst[1] = r1 * ROW_STRIDE + (cp >> 2) * KHS + (((qd1 * 8) ^ ((cp >> 2) * 32)) + (cp & 3) * 2);
  struct Geo {
    unsigned off0[2];
    unsigned ok;
    int e;  // exponent of the image's input scale
    const float* base;
  };
  auto make_geo = [&](int L, Geo& g, bool& left, bool& right) __attribute__((always_inline)) {
    const int mp = kocr_xcd_remap(L < total ? L : 0, total);
    int y0, x0, nimg;
    const long pm = mtile_org(mp, 0, y0, x0, nimg);
    g.base = p.in + (pm * p.in_cs + p.in_co) - (long)(p.W + 1) * p.in_cs;
    const int tid = wave * 64 + kocr_fresh_lane();  // see conv_w43vh_kernel's make_geo
    const int q4 = tid & 3, qd0 = (tid >> 2) & (QPR - 1), r0 = tid >> 6;
    const int cp = tid & 7, qd1 = (tid >> 3) & 15, r1 = 4 + (tid >> 7);
    g.off0[0] = (unsigned)(((r0 * p.W + 4 * qd0) * p.in_cs + q4 * 4) * 4);
    g.off0[1] = (unsigned)(((r1 * p.W + 4 * qd1) * p.in_cs + cp * 2) * 4);
    g.ok = ((L < total && (unsigned)(y0 - 1 + r0) < (unsigned)p.H) ? 1u : 0u) |
           ((L < total && (unsigned)(y0 - 1 + r1) < (unsigned)p.H) ? 2u : 0u);
    g.e = kocr_scale_exp_bits(kocr_sload(p.amax_in + __builtin_amdgcn_readfirstlane(nimg)), W4H_TOP);
    if constexpr (MODE == 1) {
      const int c0 = x0 - 1 + 4 * qd0, c1 = x0 - 1 + 4 * qd1;
      unsigned cm = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k)
        cm |= ((unsigned)(c0 + k) < (unsigned)p.W ? 0u : (4u << k)) | ((unsigned)(c1 + k) < (unsigned)p.W ? 0u : (256u << k));
      g.ok |= cm;
      left = right = false;
    } else {
      left = x0 == 0;
      right = x0 + TCOLS >= p.W;
    }
  };
  Geo gc, gn;
  bool lc, rc, ln, rn;
  int ld_cg = 0;
  bool ld_next = false;
  auto load_item0 = [&](v4f (&raw)[6]) __attribute__((always_inline)) {
    const int soff = ld_cg * 64;
    const unsigned okb = ld_next ? gn.ok : gc.ok;
    const bool ok = okb & 1u;
    const unsigned off0 = (ld_next ? gn.off0[0] : gc.off0[0]) | (ok ? 0u : OOB);
    const bool left = (ld_next ? ln : lc) && qd0 == 0, right = (ld_next ? rn : rc) && qd0 == QPR - 1;
    const unsigned stride = (unsigned)(p.in_cs * 4);
    const __amdgpu_buffer_rsrc_t rsrc = w4_rsrc(ld_next ? gn.base : gc.base, 0x80000000u);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const unsigned padk = MODE == 1 ? ((okb << (29 - k)) & OOB) : ((k == 0 ? (left ? OOB : 0u) : 0u) | (k == 5 ? (right ? OOB : 0u) : 0u));
      raw[k] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (off0 + k * stride) | padk, soff, 0));
    }
  };
  auto load_item1 = [&](v2f (&raw)[6]) __attribute__((always_inline)) {
    const int soff = ld_cg * 64;
    const unsigned okb = ld_next ? gn.ok : gc.ok;
    const bool ok = (okb >> 1) & 1u;
    const unsigned off0 = (ld_next ? gn.off0[1] : gc.off0[1]) | (ok ? 0u : OOB);
    const bool left = (ld_next ? ln : lc) && qd1 == 0, right = (ld_next ? rn : rc) && qd1 == QPR - 1;
    const unsigned stride = (unsigned)(p.in_cs * 4);
    const __amdgpu_buffer_rsrc_t rsrc = w4_rsrc(ld_next ? gn.base : gc.base, 0x80000000u);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const unsigned padk = MODE == 1 ? ((okb << (23 - k)) & OOB) : ((k == 0 ? (left ? OOB : 0u) : 0u) | (k == 5 ? (right ? OOB : 0u) : 0u));
      raw[k] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (off0 + k * stride) | padk, soff, 0));
    }
  };
  auto advance = [&]() __attribute__((always_inline)) {
    const bool wrap = ld_cg == ncg - 1;
    ld_cg = wrap ? 0 : ld_cg + 1;
    ld_next = ld_next || wrap;
  };
  auto produce4 = [&](const v4f (&d)[6], unsigned short* bufp, int xi, float sk) __attribute__((always_inline)) {
    W4H_SCALED_CONSTANTS(sk);
    const v4f V = w4h_transform(d, xi, k_s, k_sA, k_sB, k_sA2, k_sB2, k_sA2B2, k_sA2PB2);
    unsigned short* dst = bufp + xi * NP * PLANE_R + ldst[0];
    if constexpr (NP == 2) {
      u2v h, l;
      kocr_split4_h(V, h, l);
      *reinterpret_cast<u2v*>(dst) = h;
      *reinterpret_cast<u2v*>(dst + PLANE_R) = l;
    } else {
      *reinterpret_cast<u2v*>(dst) = u2v{__builtin_bit_cast(unsigned, hf2{(_Float16)V[0], (_Float16)V[1]}),
                                         __builtin_bit_cast(unsigned, hf2{(_Float16)V[2], (_Float16)V[3]})};
    }
  };
  auto produce2 = [&](const v2f (&d)[6], unsigned short* bufp, int xi, float sk) __attribute__((always_inline)) {
    W4H_SCALED_CONSTANTS(sk);
    const v2f V = w4h_transform(d, xi, k_s, k_sA, k_sB, k_sA2, k_sB2, k_sA2B2, k_sA2PB2);
    unsigned short* dst = bufp + xi * NP * PLANE_R + ldst[1];
    if constexpr (NP == 2) {
      unsigned h, l;
      kocr_split2_h(V, h, l);
      *reinterpret_cast<unsigned*>(dst) = h;
      *reinterpret_cast<unsigned*>(dst + PLANE_R) = l;
    } else {
      *reinterpret_cast<unsigned*>(dst) = __builtin_bit_cast(unsigned, hf2{(_Float16)V[0], (_Float16)V[1]});
    }
  };
  v4f raw0[1][6];
  v2f raw1[1][6];

  // ---- consumer state ------------------------------------------------------------------------------------------
  // weights through raw buffer loads (round 6, as conv_w43vh_kernel): one VGPR of lane offset, the (step, point, piece) offset a
  // scalar -- the 64-bit address arithmetic of a plain pointer cost five VALU / SALU instructions and a 64-bit-address
  // global_load issue per fragment pair of a kernel that is issue-bound
  constexpr unsigned w_step = 2u * 12 * 64 * 16;  // BYTES per (channel group, ky) step: two 32-cout tiles, 2 pieces
  const __amdgpu_buffer_rsrc_t wrsrc = w4_rsrc(reinterpret_cast<const float*>(p.wgt), 0x7FFFFFFFu);
  const unsigned wlane = (unsigned)lane * 16u;
  const unsigned w_wave = (unsigned)__builtin_amdgcn_readfirstlane(wn * 12 * 64 * 16 + 3 * ph * 2 * 64 * 16);  // cout tile, first point
  const int ns = 3 * ncg;
  hf8 bw[NBW][NP];
  f16v acc[3][2];  // [point of this wave's half][M-tile]
  const int a_lane = (l31 >> 4) * ROW_STRIDE + l5 * KHS + (((l31 & 15) * 8) ^ (l5 * 32));
  constexpr int M_OFF = 2 * ROW_STRIDE;  // M-tile 1: two rows down
  auto load_a = [&](hf8 (&a)[2][NP], const unsigned short* bufp, int ky, int pl) __attribute__((always_inline)) {
    const unsigned short* base = bufp + a_lane + ky * ROW_STRIDE + (3 * ph + pl) * NP * PLANE_R;
#pragma unroll
    for (int s = NP - 1; s >= 0; --s)
#pragma unroll
      for (int m = 0; m < 2; ++m) a[m][s] = *reinterpret_cast<const hf8*>(base + s * PLANE_R + m * M_OFF);
  };
  auto mfma_grp = [&](const hf8 (&a)[2][NP], int pl, int bi) __attribute__((always_inline)) {
    if constexpr (NP == 2) {
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[pl][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m][1], bw[bi][0], acc[pl][m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[pl][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m][0], bw[bi][1], acc[pl][m], 0, 0, 0);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) acc[pl][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m][0], bw[bi][0], acc[pl][m], 0, 0, 0);
  };
  // One channel group: nine groups (ky, point of the wave's half) of 2 PR MFMAs; the NEXT channel group is transformed
  // meanwhile (12 chunks: six points of item 0, six of the half item, spread 2,1,1 per three groups) with the scale `sk`
  // of the image it belongs to.  See conv_w43r_kernel for the a0 / a1 flip and the weight replacement.
  hf8 a0[2][NP], a1[2][NP];
  auto phase = [&](const unsigned short* bufc, unsigned short* bufn, int s0, int flip, float sk) __attribute__((always_inline)) {
    v4f(&r0)[6] = raw0[0];
    v2f(&r1)[6] = raw1[0];
#pragma unroll
    for (int g = 0; g < 9; ++g) {
      __builtin_amdgcn_sched_barrier(0);
      const int ky = g / 3, pp = g - ky * 3;
      hf8(&cur)[2][NP] = ((g & 1) ^ flip) ? a1 : a0;
      hf8(&nxt)[2][NP] = ((g & 1) ^ flip) ? a0 : a1;
      const int nchunks = (g % 3 == 0) ? 2 : 1;
      const int c0 = (g / 3) * 4 + (g % 3 == 0 ? 0 : g % 3 + 1);  // first chunk of this group: 0,2,3 | 4,6,7 | 8,10,11
      if (g < 8) load_a(nxt, bufc, (g + 1) / 3, (g + 1) % 3);
#pragma unroll
      for (int c = c0; c < c0 + nchunks; ++c) {
        if (c < 6)
          produce4(r0, bufn, c, sk);
        else
          produce2(r1, bufn, c - 6, sk);
      }
      if (g < 8) {
        mfma_grp(cur, pp, pp);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * NP, 0);  // the LDS fetches of the next group first
        // VALU per MFMA gap, in the scheduler's units (round 6: the ISA showed the five gaps of a group filled 7 / 7 / 0 / 0 / 0 --
        // the counts below are taken before instruction expansion, ~ 1.4 machine instructions each -- so three of six MFMAs ran
        // back to back with nothing to issue behind them); the group's LDS stores before its last MFMA
        auto ilv = [&](auto v_c, auto st_c) __attribute__((always_inline)) {
          constexpr int V = decltype(v_c)::value, ST = decltype(st_c)::value;
#pragma unroll
          for (int i = 0; i < 2 * PR - 1; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, V, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x200, ST, 0);
        };
        const bool half = c0 >= 6;
        if (nchunks == 2) {
          if (half)
            ilv(std::integral_constant<int, PR == 3 ? 3 : 20>{}, std::integral_constant<int, 2 * NP>{});
          else
            ilv(std::integral_constant<int, PR == 3 ? 5 : 44>{}, std::integral_constant<int, 2 * NP>{});
        } else {
          if (half)
            ilv(std::integral_constant<int, PR == 3 ? 2 : 10>{}, std::integral_constant<int, NP>{});
          else
            ilv(std::integral_constant<int, PR == 3 ? 2 : 22>{}, std::integral_constant<int, NP>{});
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      } else {
        __syncthreads();  // the next channel group is complete in bufn, bufc is free
        load_a(nxt, bufn, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_grp(cur, pp, pp);
      }
      __builtin_amdgcn_sched_barrier(0);
      {  // this point's weights of the next step
        int sn = s0 + ky + 1;
        sn = sn >= ns ? sn - ns : sn;
        const unsigned wq = w_wave + (unsigned)sn * w_step;
#pragma unroll
        for (int s = 0; s < NP; ++s)
          bw[pp][s] = __builtin_bit_cast(hf8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, (int)(wq + (unsigned)(pp * 2 + s) * 1024u), 0));
      }
      if (g == 3) load_item0(r0);
    }
    load_item1(r1);
    advance();
  };

  // ---- prologue ------------------------------------------------------------------------------------------------
  w4_stage_coef(p, coef, tid);
  make_geo(blockIdx.x, gc, lc, rc);
  make_geo(blockIdx.x + G, gn, ln, rn);
  load_item0(raw0[0]);
  load_item1(raw1[0]);
  advance();  // channel group 0 loaded
#pragma unroll
  for (int g = 0; g < NBW; ++g)
#pragma unroll
    for (int s = 0; s < NP; ++s)
      bw[g][s] = __builtin_bit_cast(hf8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, (int)(w_wave + (unsigned)(g / 3) * w_step + (unsigned)((g % 3) * 2 + s) * 1024u), 0));
#pragma unroll
  for (int xi = 0; xi < 6; ++xi) {
    produce4(raw0[0], As, xi, kocr_pow2(gc.e));
    produce2(raw1[0], As, xi, kocr_pow2(gc.e));
  }
  load_item0(raw0[0]);
  load_item1(raw1[0]);
  advance();  // channel group 1 loaded
  __syncthreads();
  load_a(a0, As, 0, 0);

  PROBE_T0();  // bins: [0] K loop  [1] next tile's geometry  [2] epilogue up to the stores  [3] store issue
  for (int L = blockIdx.x; L < total; L += G) {
    const int mp = kocr_xcd_remap(L, total);
    const float s_cur = kocr_pow2(gc.e), s_nxt = kocr_pow2(gn.e);
    const float unscale = kocr_pow2(-gc.e);
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][m][r] = 0.f;
    PROBE_T(5);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    PROBE_T(4);
    for (int cg = 0; cg < ncg; cg += 2) {
      phase(As, As + BUF_R, 3 * cg, 0, s_cur);                             // produces channel group cg + 1 (this tile's)
      phase(As + BUF_R, As, 3 * cg + 3, 1, cg + 2 < ncg ? s_cur : s_nxt);  // ... cg + 2, or the next tile's first
    }

    PROBE_T(0);
    // ---- epilogue (conv_w43r_kernel's: partial output transforms exchanged between the point halves) -----------------
    {
      // (fresh lane id, LDS coefficients, early slot reads: see conv_w43vh_kernel's epilogue)
      const int lane_e = kocr_fresh_lane();
      const int l31e = lane_e & 31, l5e = lane_e >> 5;
      const int n = wn * 32 + l31e;
      int y0, x0, timg;
      const long pm = mtile_org(mp, ph, y0, x0, timg);
      const int nimg = __builtin_amdgcn_readfirstlane(timg);
      const unsigned seen_out = p.amax_out ? kocr_amax_peek(p.amax_out + nimg) : 0u;
      const unsigned seen_pool = p.amax_pool ? kocr_amax_peek(p.amax_pool + nimg) : 0u;
      const float pa = coef[n] * unscale, pb = coef[p.Cout_pad + n];  // pre_a = pre_a 2^-wexp[o] (ConvLayer::d_pre_a_h)
      const bool has_post = p.post_a != nullptr;
      const float qa = coef[2 * p.Cout_pad + n], qb = coef[3 * p.Cout_pad + n];
      const bool live = n < p.Cout;
      const float lo = p.relu ? 0.f : -INFINITY;
      auto act = [&](float v) { return fmaxf(v * pa + pb, lo); };
      v4f* xch = reinterpret_cast<v4f*>(As + BUF_R);  // the K loop's second buffer: free since the last block barrier
      float out[4][16];
      auto halves = [&](auto ph_c) __attribute__((always_inline)) {
        constexpr int PH = decltype(ph_c)::value;
        auto partial = [&](auto m_c, int r, float (&o)[4]) __attribute__((always_inline)) {
          constexpr int M = decltype(m_c)::value;
          const float u = acc[0][M][r], v = acc[1][M][r], w = acc[2][M][r];
          if constexpr (PH == 0) {
            const float s12 = v + w, d12 = v - w;
            o[0] = u + s12;
            o[1] = W4_A * d12;
            o[2] = W4_A2 * s12;
            o[3] = W4_A3 * d12;
          } else {
            const float s34 = u + v, d34 = u - v;
            o[0] = s34;
            o[1] = W4_B * d34;
            o[2] = W4_B2 * s34;
            o[3] = W4_B3 * d34 + w;
          }
        };
        if constexpr (OCC == 1) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float o[4];
            partial(std::integral_constant<int, 1 - PH>{}, r, o);
            xch[(wave * 16 + r) * 64 + lane_e] = v4f{o[0], o[1], o[2], o[3]};
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float o[4];
          partial(std::integral_constant<int, PH>{}, r, o);
#pragma unroll
          for (int j = 0; j < 4; ++j) out[j][r] = o[j];
        }
        if constexpr (OCC == 2) {  // the exchange in two halves of eight accumulator rows: 32 KB, inside the free buffer
#pragma unroll
          for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int r = 8 * h; r < 8 * h + 8; ++r) {
              float o[4];
              partial(std::integral_constant<int, 1 - PH>{}, r, o);
              xch[(wave * 8 + r - 8 * h) * 64 + lane_e] = v4f{o[0], o[1], o[2], o[3]};
            }
            __syncthreads();
#pragma unroll
            for (int r = 8 * h; r < 8 * h + 8; ++r) {
              const v4f q = xch[((wave ^ 2) * 8 + r - 8 * h) * 64 + lane_e];
#pragma unroll
              for (int j = 0; j < 4; ++j) out[j][r] += q[j];
            }
            __syncthreads();  // the second half / the next channel group is written into this buffer
          }
        }
      };
      if (ph == 0)
        halves(std::integral_constant<int, 0>{});
      else
        halves(std::integral_constant<int, 1>{});
      if constexpr (OCC == 1) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const v4f q = xch[((wave ^ 2) * 16 + r) * 64 + lane_e];
#pragma unroll
          for (int j = 0; j < 4; ++j) out[j][r] = act(out[j][r] + q[j]);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int j = 0; j < 4; ++j) out[j][r] = act(out[j][r]);
      }
      if (has_post) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int j = 0; j < 4; ++j) out[j][r] = out[j][r] * qa + qb;
      }
      if constexpr (OCC == 1) __syncthreads();  // the next channel group is transformed into this buffer
      int ocs4 = p.out_cs * 4;
      asm volatile("" : "+s"(ocs4));
      int pcs4 = p.pool_cs * 4;
      asm volatile("" : "+s"(pcs4));
      if (p.amax_out || p.amax_pool) {
        float mx = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fabsf(out[j][r]));
        mx = live ? mx : 0.f;
        if (p.amax_out) kocr_amax_update_known(p.amax_out + nimg, mx, seen_out);
        if (p.amax_pool) kocr_amax_update_known(p.amax_pool + nimg, mx, seen_pool);
      }
      PROBE_T(2);
      if (!POOL || p.write_full) {
        const __amdgpu_buffer_rsrc_t ro = w4_rsrc(p.out + (pm * p.out_cs + p.out_co), 0x7FFFFFFFu);
        const unsigned vo = live ? (unsigned)((16 * l5e * p.out_cs + n) * 4) : OOB;
        if constexpr (MODE == 1) {
          const int wlim = p.W - x0 - 16 * l5e;  // columns of the image right of this lane half's first one
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            if (y0 + h >= p.H) continue;  // uniform: the row lies below the image
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              const int px = 4 * ((r & 3) + 8 * (r >> 2));
#pragma unroll
              for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(out[j][r + 8 * h]), ro, px + j < wlim ? vo : OOB, (px + j + h * p.W) * ocs4, 0);
            }
          }
        } else {
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const int px = 4 * ((r & 3) + 8 * (r >> 2));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(out[j][r]), ro, vo, (px + j) * ocs4, 0);
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(out[j][r + 8]), ro, vo, (px + j + p.W) * ocs4, 0);
            }
          }
        }
      }
      if constexpr (POOL) {
        long pp0;
        if constexpr (MODE == 1)  // floor pooling of any H, W
          pp0 = ((long)timg * (p.H >> 1) + (y0 >> 1)) * (p.W >> 1) + (x0 >> 1);
        else
          pp0 = ((pm - x0) >> 2) + (x0 >> 1);
        const __amdgpu_buffer_rsrc_t rp = w4_rsrc(p.pool_out + (pp0 * p.pool_cs + p.pool_co), 0x7FFFFFFFu);
        const unsigned vp = live ? (unsigned)((8 * l5e * p.pool_cs + n) * 4) : OOB;
        const bool prow = MODE != 1 || (y0 >> 1) < (p.H >> 1);  // uniform: a pooled row exists here
        const int plim = (p.W >> 1) - (x0 >> 1) - 8 * l5e;
        if (prow) {
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const int pq = 2 * ((r & 3) + 8 * (r >> 2));
            const float v0 = fmaxf(fmaxf(out[0][r], out[1][r]), fmaxf(out[0][r + 8], out[1][r + 8]));
            const float v1 = fmaxf(fmaxf(out[2][r], out[3][r]), fmaxf(out[2][r + 8], out[3][r + 8]));
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v0), rp, (MODE != 1 || pq < plim) ? vp : OOB, pq * pcs4, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v1), rp, (MODE != 1 || pq + 1 < plim) ? vp : OOB, (pq + 1) * pcs4, 0);
          }
        }
      }
      PROBE_T(3);
    }
    // (behind the stores: see conv_w43vh_kernel)
    gc = gn;
    lc = ln;
    rc = rn;
    make_geo(L + 2 * G, gn, ln, rn);
    ld_next = false;
    PROBE_T(1);
  }
  PROBE_TEND(tid == 0, 0, 6);
}

// ===================================================================================================
// conv_w43fh_kernel -- the flattened-pixel arrangement (conv_w43_kernel, no fused pooling, dilation 1) in fp16 arithmetic,
// for Cout > 64 layers whose images tile neither as 4 x 64 nor as 8 x 32: the recogniser's conv_2 ... conv_5 (31 x 200 and
// 15 x 100 crops), odd CRAFT sizes.  Tile = 64 quads (256 flattened pixels, may cross image rows and IMAGES) x 128 couts;
// K-step = (16-channel group, ky): every step gathers, transforms and splits the input row it needs (no vertical reuse).
// The input scale is per image, so here it is per QUAD: a producer thread looks up the slot of the image its quad lies
// in; a tile of 256 pixels touches at most two images (the launcher requires H W >= 256), so the epilogue undoes the
// scale with a per-accumulator-row select between two factors and maintains the two images' max-|x| slots separately.
// ===================================================================================================
// DIL = 1 (round 5): a dilated 3x3 convolution (taps p.dil pixels apart in x and y; CRAFT's composite slice5 layer, dilation 6,
// W % (4 dil) == 0) through the same algebra on the comb of pixels of conv_w43_kernel<.., DIL = 1>: the quads of a row are its
// W / 4 residue-class quads (quad q covers x0 + dil j, j = 0 .. 3, x0 = (q / dil) 4 dil + q % dil), a tile = 64 consecutive
// quads over the rows of the batch; everything per quad (image, scale, padding) follows from its row.
template <int NP, int DIL = 0>
__global__ __launch_bounds__(256) void conv_w43fh_kernel(W4Params p) {
  constexpr int PR = NP == 2 ? 3 : 1;
  constexpr int PLANE_F = PLANE;          // one (xi, piece) plane: 2 M-tiles x 2 k halves x 256 ushorts (w43_common.h)
  constexpr int BUF_F = 6 * NP * PLANE_F; // one K-step: 24 KB (NP = 2)
  extern __shared__ __attribute__((aligned(16))) unsigned short As[];
  // behind the two K-step buffers: the epilogue's per-channel coefficients [pre_a | pre_b | post_a | post_b][Cout_pad],
  // staged once per block -- fetched from global memory in every tile's epilogue they cost it one memory latency each time
  float* coef = reinterpret_cast<float*>(As + 2 * BUF_F);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, l5 = lane >> 5;
  const int nblk_n = p.Cout_pad >> 7;
  const int total = p.total_tiles;
  const int ns = p.nsteps;
  const int G = gridDim.x;
  const int hw = p.H * p.W;
  constexpr unsigned OOB = 0x80000000u;

  // ---- producer state: this thread's gather item = (quad qi of the 64-quad tile, channel quad q4) ----------------------
  const int qi = tid >> 2, q4 = tid & 3;
  const int ldst = ((qi >> 5) * 2 + (q4 >> 1)) * KH_STRIDE + ((((qi & 31) * 8) ^ ((q4 >> 1) * 32)) + (q4 & 1) * 4);
  struct Geo {
    unsigned goff[6];
    int gy;
    bool gok;
    float s;     // 2^e of the image this thread's quad lies in
    int e0, e1;  // (uniform) scale exponents of the tile's first image and of the next one (the tile's second, if it has one)
    const float* base;
  };
  const int dd = DIL ? p.dil : 1;
  auto make_geo = [&](int L, Geo& g) __attribute__((always_inline)) {
    int mp, nt_unused;
    w4_decode(p, kocr_xcd_remap(L < total ? L : 0, total), nblk_n, mp, nt_unused);
    const int tid_f = wn * 64 + kocr_fresh_lane();  // see kocr_fresh_lane
    const int qi = tid_f >> 2, q4 = tid_f & 3;
    if constexpr (DIL) {
      // quad index over all rows of the batch -> (row, quad of the row) -> comb position; offsets relative to the first
      // pixel of the tile's first row (Mtotal / 4 quads in all: < 2^29)
      const unsigned q0 = (unsigned)mp * 64u;
      const unsigned nq = (unsigned)(p.Mtotal >> 2);
      const unsigned row_a = w4_fdiv(q0 < nq ? q0 : nq - 1, p.dv_qpr);
      g.base = p.in + ((long)row_a * p.W * p.in_cs + p.in_co) - (long)(dd * p.W + dd) * p.in_cs;
      const unsigned qg = q0 + (unsigned)qi;
      g.gok = L < total && qg < nq;
      const unsigned qc = qg < nq ? qg : nq - 1;
      const unsigned row = w4_fdiv(qc, p.dv_qpr);
      const int qr = (int)(qc - row * (unsigned)p.qpr);
      const int qd = (int)w4_fdiv((unsigned)qr, p.dv_dil);
      const int x0 = qd * 4 * dd + (qr - qd * dd);
      const unsigned nimg = w4_fdiv(row, p.dv_h);
      g.gy = (int)(row - nimg * (unsigned)p.H);
      const int n0 = __builtin_amdgcn_readfirstlane((int)w4_fdiv(row_a, p.dv_h));
      const int n1 = (n0 + 1 < p.Mtotal / (p.H * p.W)) ? n0 + 1 : n0;
      g.e0 = kocr_scale_exp_bits(kocr_sload(p.amax_in + n0), W4H_TOP);
      g.e1 = kocr_scale_exp_bits(kocr_sload(p.amax_in + n1), W4H_TOP);
      g.s = kocr_pow2((int)nimg == n0 ? g.e0 : g.e1);
      const int rel = (int)(row - row_a) * p.W + x0;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const bool pad = (k == 0 && x0 < dd) || (k == 5 && x0 + 4 * dd >= p.W);  // column zero padding
        g.goff[k] = pad ? OOB : (unsigned)(((rel + k * dd) * p.in_cs + q4 * 4) * 4);
      }
      return;
    }
    const long pm_a = (long)mp * 256;
    g.base = p.in + (pm_a * p.in_cs + p.in_co) - (long)(p.W + 1) * p.in_cs;
    const int rel = 4 * qi;
    const long gp = pm_a + rel;
    g.gok = L < total && gp < p.Mtotal;
    // (row, image) of the quad's first pixel by multiply-high (gp < 2^31): a 64-bit % and / here compile to two software
    // division loops per tile and thread
    const unsigned gpc = (unsigned)(gp < p.Mtotal ? gp : (long)p.Mtotal - 1);
    const unsigned row = w4_fdiv(gpc, p.dv_w), nimg = w4_fdiv(gpc, p.dv_hw);
    const int x0 = (int)(gpc - row * (unsigned)p.W);
    g.gy = (int)(row - nimg * (unsigned)p.H);
    // the tile's (at most two) images' slots through the scalar cache
    const unsigned pmc = (unsigned)(pm_a < (long)p.Mtotal ? pm_a : (long)p.Mtotal - 1);
    const int n0 = __builtin_amdgcn_readfirstlane((int)w4_fdiv(pmc, p.dv_hw));
    const int n1 = ((long)(n0 + 1) * hw < (long)p.Mtotal) ? n0 + 1 : n0;
    g.e0 = kocr_scale_exp_bits(kocr_sload(p.amax_in + n0), W4H_TOP);
    g.e1 = kocr_scale_exp_bits(kocr_sload(p.amax_in + n1), W4H_TOP);
    g.s = kocr_pow2((int)nimg == n0 ? g.e0 : g.e1);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const bool pad = (k == 0 && x0 < 1) || (k == 5 && x0 + 4 >= p.W);  // column zero padding
      g.goff[k] = pad ? OOB : (unsigned)(((rel + k) * p.in_cs + q4 * 4) * 4);
    }
  };
  Geo gc, gn;
  int ld_ky = 0, ld_cg = 0;  // position of the NEXT K-step to load inside its tile
  bool ld_next = false;      // ... and whether that tile is already the next one
  const int ncg = p.Cin >> 4;
  auto load_raw = [&](v4f (&raw)[6]) __attribute__((always_inline)) {
    const int soff = (ld_ky * dd * p.W * p.in_cs + ld_cg * 16) * 4;
    const int gy = ld_next ? gn.gy : gc.gy;
    const bool ok = (ld_next ? gn.gok : gc.gok) & ((unsigned)(gy + dd * (ld_ky - 1)) < (unsigned)p.H);
    const unsigned kill = ok ? 0u : OOB;
    const __amdgpu_buffer_rsrc_t rsrc = w4_rsrc(ld_next ? gn.base : gc.base, 0x80000000u);
#pragma unroll
    for (int k = 0; k < 6; ++k)
      raw[k] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (ld_next ? gn.goff[k] : gc.goff[k]) | kill, soff, 0));
    const bool wrap_ky = ld_ky == 2;
    ld_ky = wrap_ky ? 0 : ld_ky + 1;
    const bool wrap_cg = wrap_ky && ld_cg == ncg - 1;
    ld_cg = wrap_cg ? 0 : (wrap_ky ? ld_cg + 1 : ld_cg);
    ld_next = ld_next || wrap_cg;
  };
  auto produce_point = [&](const v4f (&d)[6], unsigned short* bufp, int xi, float sk) __attribute__((always_inline)) {
    W4H_SCALED_CONSTANTS(sk);
    const v4f V = w4h_transform(d, xi, k_s, k_sA, k_sB, k_sA2, k_sB2, k_sA2B2, k_sA2PB2);
    unsigned short* dst = bufp + xi * NP * PLANE_F + ldst;
    if constexpr (NP == 2) {
      u2v h, l;
      kocr_split4_h(V, h, l);
      *reinterpret_cast<u2v*>(dst) = h;
      *reinterpret_cast<u2v*>(dst + PLANE_F) = l;
    } else {
      *reinterpret_cast<u2v*>(dst) = u2v{__builtin_bit_cast(unsigned, hf2{(_Float16)V[0], (_Float16)V[1]}),
                                         __builtin_bit_cast(unsigned, hf2{(_Float16)V[2], (_Float16)V[3]})};
    }
  };

  // ---- consumer state ------------------------------------------------------------------------------------------------
  const int ntiles32 = p.Cout_pad >> 5;
  const size_t w_step = (size_t)ntiles32 * 12 * 64 * 8;  // ushorts per K-step
  auto w_tile = [&](int nt) { return p.wgt + ((size_t)(nt * 4 + wn) * 12 * 64 + lane) * 8; };
  hf8 bw[6][NP];
  f16v acc[6][2];
  const int a_lane = l5 * KH_STRIDE + ((l31 * 8) ^ (l5 * 32));
  auto load_a = [&](hf8 (&a)[2][NP], const unsigned short* bufp, int xi) __attribute__((always_inline)) {
    const unsigned short* base = bufp + xi * NP * PLANE_F + a_lane;
#pragma unroll
    for (int s = NP - 1; s >= 0; --s)
#pragma unroll
      for (int m = 0; m < 2; ++m) a[m][s] = *reinterpret_cast<const hf8*>(base + s * PLANE_F + m * 2 * KH_STRIDE);
  };
  auto mfma_pt = [&](const hf8 (&a)[2][NP], int xi) __attribute__((always_inline)) {
    if constexpr (NP == 2) {
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[xi][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m][1], bw[xi][0], acc[xi][m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[xi][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m][0], bw[xi][1], acc[xi][m], 0, 0, 0);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) acc[xi][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m][0], bw[xi][0], acc[xi][m], 0, 0, 0);
  };
  // One K-step: consume `bufc` (6 points x 2 PR MFMAs) while producing the NEXT step from `raw` (scale sk) into `bufn`;
  // see conv_w43_kernel::step for the order of fetches, stores, the weight replacement and the barrier placement.
  hf8 a0[2][NP], a1[2][NP];
  auto step = [&](const unsigned short* bufc, unsigned short* bufn, const v4f (&raw)[6], const unsigned short* w_next, float sk)
                  __attribute__((always_inline)) {
    auto load_b = [&](int xi) __attribute__((always_inline)) {
#pragma unroll
      for (int s = 0; s < NP; ++s) bw[xi][s] = *reinterpret_cast<const hf8*>(w_next + (size_t)(xi * 2 + s) * 64 * 8);
    };
    auto interleave = [&]() __attribute__((always_inline)) {
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * NP, 0);
      constexpr int VPG = 30 / (2 * PR - 1) + 1;
#pragma unroll
      for (int i = 0; i < 2 * PR - 1; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, VPG, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x200, NP, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    };
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      __builtin_amdgcn_sched_barrier(0);
      load_a(a1, bufc, 2 * q + 1);
      produce_point(raw, bufn, 2 * q, sk);
      mfma_pt(a0, 2 * q);
      interleave();
      __builtin_amdgcn_sched_barrier(0);
      load_b(2 * q);
      __builtin_amdgcn_sched_barrier(0);
      if (q < 2) {
        load_a(a0, bufc, 2 * q + 2);
        produce_point(raw, bufn, 2 * q + 1, sk);
        mfma_pt(a1, 2 * q + 1);
        interleave();
      } else {
        produce_point(raw, bufn, 5, sk);
        __syncthreads();  // next step complete in bufn, bufc free
        load_a(a0, bufn, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_pt(a1, 5);
      }
      __builtin_amdgcn_sched_barrier(0);
      load_b(2 * q + 1);
    }
  };

  // ---- pipeline prologue ---------------------------------------------------------------------------------------------
  w4_stage_coef(p, coef, tid);
  make_geo(blockIdx.x, gc);
  make_geo(blockIdx.x + G, gn);
  v4f rawA[6], rawB[6];
  load_raw(rawA);  // global step 0
  load_raw(rawB);  // global step 1
  {
    int mp0, nt0;
    w4_decode(p, kocr_xcd_remap(blockIdx.x, total), nblk_n, mp0, nt0);
    const unsigned short* w0 = w_tile(nt0);
#pragma unroll
    for (int xi = 0; xi < 6; ++xi)
#pragma unroll
      for (int s = 0; s < NP; ++s) bw[xi][s] = *reinterpret_cast<const hf8*>(w0 + (size_t)(xi * 2 + s) * 64 * 8);
  }
#pragma unroll
  for (int xi = 0; xi < 6; ++xi) produce_point(rawA, As, xi, gc.s);
  load_raw(rawA);  // global step 2
  __syncthreads();
  load_a(a0, As, 0);

  PROBE_T0();  // bins: [0] K loop  [1] next tile's geometry  [2] epilogue arithmetic + amax  [3] store issue
  for (int L = blockIdx.x; L < total; L += G) {
    int mp, nt, mp_n, nt_n;
    w4_decode(p, kocr_xcd_remap(L, total), nblk_n, mp, nt);
    w4_decode(p, kocr_xcd_remap(L + G < total ? L + G : L, total), nblk_n, mp_n, nt_n);
    const unsigned short* w_ptr = w_tile(nt);
    const unsigned short* w_after = w_tile(nt_n);
    const float s_cur = gc.s, s_nxt = gn.s;
#pragma unroll
    for (int x = 0; x < 6; ++x)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][m][r] = 0.f;
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (int s = 0; s < ns; s += 2) {
      step(As, As + BUF_F, rawB, w_ptr + (size_t)(s + 1) * w_step, s_cur);  // produces the odd step s + 1 (this tile's)
      load_raw(rawB);
      // produces step s + 2: this tile's, or the next tile's first
      step(As + BUF_F, As, rawA, s + 2 < ns ? w_ptr + (size_t)(s + 2) * w_step : w_after, s + 2 < ns ? s_cur : s_nxt);
      load_raw(rawA);
    }
    PROBE_T(0);
    const int e0c = gc.e0, e1c = gc.e1;

    // ---- epilogue: 32x32 C/D map: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) ----------------------------
    {
      const int lane_e = kocr_fresh_lane();  // every lane term of the epilogue derives from this
      const int l31e = lane_e & 31, l5e = lane_e >> 5;
      const int n = (nt * 4 + wn) * 32 + l31e;
      const long pm0 = (long)mp * 256;
      // the tile's (at most two) images, the first quad of the second one, their unscale factors 2^-e
      const long pm0c = pm0 < (long)p.Mtotal ? pm0 : (long)p.Mtotal - 1;
      const int n0 = __builtin_amdgcn_readfirstlane((int)w4_fdiv((unsigned)pm0c, p.dv_hw));
      const long b1 = ((long)(n0 + 1) * hw - pm0) >> 2;  // quads of the tile that lie in image n0 (W % 4 == 0)
      const int qb = __builtin_amdgcn_readfirstlane((int)(b1 < 64 ? b1 : 64));
      const int n1 = ((long)(n0 + 1) * hw < (long)p.Mtotal) ? n0 + 1 : n0;
      // the output slots as they are now: read here, compared after the arithmetic below (kocr_amax_update_known)
      unsigned seen0 = 0, seen1 = 0;
      if (p.amax_out) {
        seen0 = kocr_amax_peek(p.amax_out + n0);
        seen1 = kocr_amax_peek(p.amax_out + n1);
      }
      const float u0 = kocr_pow2(-e0c), u1 = kocr_pow2(-e1c);
      const float pre_a = coef[n], pb = coef[p.Cout_pad + n];  // n < Cout_pad; the padding channels repeat the last one
      const float pa0 = pre_a * u0, pa1 = pre_a * u1;
      const bool has_post = p.post_a != nullptr;
      const bool live = n < p.Cout;
      const float lo = p.relu ? 0.f : -INFINITY;
      const int qlim = (int)((((long)p.Mtotal - pm0) >> 2) < 64 ? (((long)p.Mtotal - pm0) >> 2) : 64);  // quads inside the tensor
      // (from the fresh lane id: as loop invariants the 32 quad indices below and what derives from them are hoisted out of
      // the tile loop into scratch and come back through ~35 serialised scratch loads per tile)
      const int l5q = 4 * l5e;
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int q = m * 32 + (r & 3) + 8 * (r >> 2) + l5q;
          const float pa = q < qb ? pa0 : pa1;
          const float m0 = acc[0][m][r], m1 = acc[1][m][r], m2 = acc[2][m][r], m3 = acc[3][m][r], m4 = acc[4][m][r],
                      m5 = acc[5][m][r];
          const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
          acc[0][m][r] = fmaxf(((m0 + s12) + s34) * pa + pb, lo);
          acc[1][m][r] = fmaxf((W4_A * d12 + W4_B * d34) * pa + pb, lo);
          acc[2][m][r] = fmaxf((W4_A2 * s12 + W4_B2 * s34) * pa + pb, lo);
          acc[3][m][r] = fmaxf(((W4_A3 * d12 + W4_B3 * d34) + m5) * pa + pb, lo);
        }
      if (has_post) {  // uniform
        const float qa = coef[2 * p.Cout_pad + n], qb_ = coef[3 * p.Cout_pad + n];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][m][r] = acc[j][m][r] * qa + qb_;
      }
      if (!DIL && p.Wv) {  // width-padded output (Tensor::Wv): the columns behind the valid width are zero padding
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int q = m * 32 + (r & 3) + 8 * (r >> 2) + l5q;
            const unsigned pq = (unsigned)(pm0 + 4 * q);
            const int x0 = (int)(pq - w4_fdiv(pq, p.dv_w) * (unsigned)p.W);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j][m][r] = x0 + j < p.Wv ? acc[j][m][r] : 0.f;
          }
      }
      if (p.amax_out) {
        float mx0 = 0.f, mx1 = 0.f;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int q = m * 32 + (r & 3) + 8 * (r >> 2) + l5q;
            const float mq = (live && q < qlim) ? fmaxf(fmaxf(fabsf(acc[0][m][r]), fabsf(acc[1][m][r])), fmaxf(fabsf(acc[2][m][r]), fabsf(acc[3][m][r]))) : 0.f;
            mx0 = fmaxf(mx0, q < qb ? mq : 0.f);
            mx1 = fmaxf(mx1, q < qb ? 0.f : mq);
          }
        kocr_amax_update_known(p.amax_out + n0, mx0, seen0);
        if (qb < 64) kocr_amax_update_known(p.amax_out + n1, mx1, seen1);
      }
      PROBE_T(2);
      int ocs4 = p.out_cs * 4;
      asm volatile("" : "+s"(ocs4));
      if constexpr (DIL) {
        // quad (m, r, l5) of the tile -> (row, comb position), relative to the first pixel of the tile's first row; the quad's
        // four outputs are p.dil pixels apart
        const unsigned q0 = (unsigned)mp * 64u;
        const unsigned row_a = w4_fdiv(q0, p.dv_qpr);
        const __amdgpu_buffer_rsrc_t ro = w4_rsrc(p.out + ((long)row_a * p.W * p.out_cs + p.out_co), 0x7FFFFFFFu);
        const int dstep = dd * ocs4;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int q = m * 32 + (r & 3) + 8 * (r >> 2) + l5q;
            const unsigned qg = q0 + (unsigned)q;
            const unsigned row = w4_fdiv(qg, p.dv_qpr);
            const int qr = (int)(qg - row * (unsigned)p.qpr);
            const int qd = (int)w4_fdiv((unsigned)qr, p.dv_dil);
            const int x0 = qd * 4 * dd + (qr - qd * dd);
            const unsigned vo = (live && q < qlim) ? (unsigned)((((int)(row - row_a) * p.W + x0) * p.out_cs + n) * 4) : OOB;
#pragma unroll
            for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[j][m][r]), ro, vo, j * dstep, 0);
          }
      } else {
      // bytes from the tile's first pixel to the end of the tensor: stores past it are dropped
      const long rem = ((long)p.Mtotal - pm0) * ocs4;
      const __amdgpu_buffer_rsrc_t ro =
          w4_rsrc(p.out + (pm0 * p.out_cs + p.out_co), rem < 0x7FFFFFFFL ? (unsigned)(rem > 0 ? rem : 0) : 0x7FFFFFFFu);
      const unsigned vo = live ? (unsigned)((16 * l5e * p.out_cs + n) * 4) : OOB;
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
      PROBE_T(3);
    }
    // (behind the stores: see conv_w43vh_kernel)
    gc = gn;
    make_geo(L + 2 * G, gn);
    ld_next = false;
    PROBE_T(1);
  }
  PROBE_TEND(tid == 0, 0, 4);
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
// Weights of a layer prepare_w43 accepted (3x3, Cin % 32 == 0, Cout > 32; dilation 1 only here): U = G g in float64,
// scaled per output channel by 2^wexp[o] (max |U 2^wexp| over the channel's 18 Cin values in [2^14, 2^15)), rounded once to
// fp32, split into two fp16 pieces by round-to-nearest, packed in conv_w43's B-operand order with two pieces per point.
int prepare_w43h(kocr_ctx* ctx, ConvLayer& L, const float* w, bool w_is_oihw, const float* pre_a) {
  if (!L.d_w4 || (L.dil != 1 && L.w4_cout_pad == 64)) return KOCR_OK;  // dilated: the flattened 128-cout arrangement only
  const int Cin = L.Cin, Cout = L.Cout;
  const int cp = L.w4_cout_pad;
  const int nt32 = cp / 32;
  const double pa = W4_PA, pb = W4_PB, a2 = pa * pa, b2 = pb * pb;
  const double na = 2.0 * a2 * (a2 - b2), nb = 2.0 * b2 * (b2 - a2);
  auto U6 = [&](int c, int ky, int o, double (&U)[6]) {
    double g[3];
    for (int kx = 0; kx < 3; ++kx)
      g[kx] = w_is_oihw ? w[(((size_t)o * Cin + c) * 3 + ky) * 3 + kx] : w[(((size_t)ky * 3 + kx) * Cin + c) * Cout + o];
    U[0] = g[0] / (a2 * b2);
    U[1] = (g[0] + pa * g[1] + a2 * g[2]) / na;
    U[2] = (g[0] - pa * g[1] + a2 * g[2]) / na;
    U[3] = (g[0] + pb * g[1] + b2 * g[2]) / nb;
    U[4] = (g[0] - pb * g[1] + b2 * g[2]) / nb;
    U[5] = g[2];
  };
  std::vector<int> wexp(cp, 0);
  for (int o = 0; o < Cout; ++o) {
    double umax = 0;
    for (int c = 0; c < Cin; ++c)
      for (int ky = 0; ky < 3; ++ky) {
        double U[6];
        U6(c, ky, o, U);
        for (int xi = 0; xi < 6; ++xi) umax = std::max(umax, std::fabs(U[xi]));
      }
    if (umax > 0 && std::isfinite(umax)) {
      int E;
      std::frexp((float)umax, &E);  // umax = f 2^E, f in [0.5, 1): umax 2^(15 - E) in [2^14, 2^15)
      wexp[o] = std::max(-100, std::min(100, 15 - E));
    }
  }
  std::vector<unsigned short> u((size_t)(Cin / 16) * 3 * nt32 * 12 * 64 * 8, 0);
  for (int c = 0; c < Cin; ++c)
    for (int ky = 0; ky < 3; ++ky)
      for (int o = 0; o < Cout; ++o) {
        double U[6];
        U6(c, ky, o, U);
        const int k = c % 16, lane = (k >> 3) * 32 + (o & 31), j = k & 7;
        const size_t step = (size_t)(c / 16) * 3 + ky;
        for (int xi = 0; xi < 6; ++xi) {
          const float x = std::ldexp((float)U[xi], wexp[o]);  // the fp32 value the bf16x3 kernels split, times 2^wexp (exact)
          const _Float16 h = (_Float16)x, l = (_Float16)(x - (float)h);
          unsigned short hb, lb;
          memcpy(&hb, &h, 2);
          memcpy(&lb, &l, 2);
          u[((((step * nt32 + o / 32) * 6 + xi) * 2 + 0) * 64 + lane) * 8 + j] = hb;
          u[((((step * nt32 + o / 32) * 6 + xi) * 2 + 1) * 64 + lane) * 8 + j] = lb;
        }
      }
  void* d = nullptr;
  KOCR_TRY(ctx->dev_alloc(&d, u.size() * sizeof(unsigned short)));
  KOCR_HIP(ctx, hipMemcpy(d, u.data(), u.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
  L.d_w4h = (unsigned short*)d;
  std::vector<float> ah(std::max(cp, L.Cout_pad), 1.f);
  for (int o = 0; o < Cout; ++o) ah[o] = std::ldexp(pre_a ? pre_a[o] : 1.f, -wexp[o]);
  KOCR_TRY(ctx->upload(&L.d_pre_a_h, ah));
  return KOCR_OK;
}

template <int POOL, int GEO, int NP, int DBG = 0, int MODE = 0>
static int w4vh_launch(kocr_ctx* ctx, W4Params& p) {
  constexpr int LDSV0 = (GEO == 2 ? 2 * 6 * 10 * 128 * 2 : 2 * 6 * 6 * 256 * 2) * NP;  // 2 x 30 / 36 KB (NP = 2)
  const int LDSV = LDSV0 + 4 * p.Cout_pad * 4;                                         // + the epilogue's coefficients
  static std::atomic<bool> attr_done[64];
  const int dev = ctx->device & 63;
  if (!attr_done[dev]) {
    KOCR_HIP(ctx, hipFuncSetAttribute((const void*)conv_w43vh_kernel<POOL, GEO, NP, DBG, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSV0 + W4_COEF_BYTES_MAX));
    attr_done[dev] = true;
  }
  static std::atomic<int> n_cus[64];
  if (!n_cus[dev]) {
    hipDeviceProp_t prop;
    KOCR_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
    n_cus[dev] = prop.multiProcessorCount;
  }
  const int n_cu = n_cus[dev];
  const int grid = p.total_tiles < n_cu ? p.total_tiles : n_cu;  // persistent: one block per CU
  PROBE_RESET(ctx);
  hipLaunchKernelGGL((conv_w43vh_kernel<POOL, GEO, NP, DBG, MODE>), dim3(grid), dim3(256), LDSV, ctx->stream, p);
  KOCR_HIP(ctx, hipGetLastError());
  {
    char what[80];
    snprintf(what, sizeof what, "conv_w43vh<%d,%d,%d,%d> tiles %d steps %d cout %d", POOL, GEO, NP, MODE, p.total_tiles, p.nsteps, p.Cout);
    (void)what;
    PROBE_REPORT(ctx, what, grid);
  }
  return KOCR_OK;
}

// the vertical-reuse arrangement in fp16 arithmetic: p as launch_conv_w43 filled it for conv_w43v_kernel, with wgt =
// ConvLayer::d_w4h, pre_a = d_pre_a_h and amax_in set.  geo 1 / 2, pieces 2 / 1.
// mode 1 / 2 (ragged images, cell grids) exist with two pieces only: the one-piece fast mode runs those layers in fp16x2
int launch_w43vh(kocr_ctx* ctx, W4Params& p, bool fuse, int geo, int pieces, int mode) {
  if (mode == 2) {
    if (geo == 2) return w4vh_launch<0, 2, 2, 0, 2>(ctx, p);
    return fuse ? w4vh_launch<1, 1, 2, 0, 2>(ctx, p) : w4vh_launch<0, 1, 2, 0, 2>(ctx, p);
  }
  if (mode == 1) {
    if (geo == 2) return w4vh_launch<0, 2, 2, 0, 1>(ctx, p);
    return fuse ? w4vh_launch<1, 1, 2, 0, 1>(ctx, p) : w4vh_launch<0, 1, 2, 0, 1>(ctx, p);
  }
#ifdef KOCR_DEV_SWITCHES
  static const int dbg = getenv("KOCR_W43H_DBG") ? atoi(getenv("KOCR_W43H_DBG")) : 0;
  if (dbg && !fuse && geo == 1 && pieces == 2) {
    switch (dbg) {
      case 1: return w4vh_launch<0, 1, 2, 1>(ctx, p);
      case 2: return w4vh_launch<0, 1, 2, 2>(ctx, p);
      case 4: return w4vh_launch<0, 1, 2, 4>(ctx, p);
      case 8: return w4vh_launch<0, 1, 2, 8>(ctx, p);
      case 16: return w4vh_launch<0, 1, 2, 16>(ctx, p);
      case 32: return w4vh_launch<0, 1, 2, 32>(ctx, p);
      case 3: return w4vh_launch<0, 1, 2, 3>(ctx, p);
      case 18: return w4vh_launch<0, 1, 2, 18>(ctx, p);
      case 19: return w4vh_launch<0, 1, 2, 19>(ctx, p);
      case 27: return w4vh_launch<0, 1, 2, 27>(ctx, p);
      case 59: return w4vh_launch<0, 1, 2, 59>(ctx, p);
      case 12: return w4vh_launch<0, 1, 2, 12>(ctx, p);
      case 17: return w4vh_launch<0, 1, 2, 17>(ctx, p);
      case 64: return w4vh_launch<0, 1, 2, 64>(ctx, p);
      case 128: return w4vh_launch<0, 1, 2, 128>(ctx, p);
      case 66: return w4vh_launch<0, 1, 2, 66>(ctx, p);
      default: break;
    }
  }
#endif
  if (pieces == 2) {
    if (geo == 2) return w4vh_launch<0, 2, 2>(ctx, p);
    return fuse ? w4vh_launch<1, 1, 2>(ctx, p) : w4vh_launch<0, 1, 2>(ctx, p);
  }
  if (geo == 2) return w4vh_launch<0, 2, 1>(ctx, p);
  return fuse ? w4vh_launch<1, 1, 1>(ctx, p) : w4vh_launch<0, 1, 1>(ctx, p);
}

template <int POOL, int NP, int MODE = 0, int OCC = 1>
static int w4rh_launch(kocr_ctx* ctx, W4Params& p) {
  // one 36 KB buffer + the epilogue's 64 KB exchange area (its head is the second buffer); OCC = 2: the two buffers only
  constexpr int LDSR0 = OCC == 2 ? 2 * 6 * NP * 6 * 256 * 2 : 6 * NP * 6 * 256 * 2 + 4 * 16 * 64 * 16;
  const int LDSR = LDSR0 + 4 * p.Cout_pad * 4;                    // + the epilogue's coefficients
  static std::atomic<bool> attr_done[64];
  const int dev = ctx->device & 63;
  if (!attr_done[dev]) {
    KOCR_HIP(ctx, hipFuncSetAttribute((const void*)conv_w43rh_kernel<POOL, NP, MODE, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSR0 + W4_COEF_BYTES_MAX));
    attr_done[dev] = true;
  }
  static std::atomic<int> n_cus[64];
  if (!n_cus[dev]) {
    hipDeviceProp_t prop;
    KOCR_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
    n_cus[dev] = prop.multiProcessorCount;
  }
  const int n_cu = n_cus[dev];
  const int slots = n_cu * OCC;
  const int grid = p.total_tiles < slots ? p.total_tiles : slots;
  PROBE_RESET(ctx);
  hipLaunchKernelGGL((conv_w43rh_kernel<POOL, NP, MODE, OCC>), dim3(grid), dim3(256), LDSR, ctx->stream, p);
  KOCR_HIP(ctx, hipGetLastError());
  {
    char what[80];
    snprintf(what, sizeof what, "conv_w43rh<%d,%d,%d,occ%d> tiles %d steps %d", POOL, NP, MODE, OCC, p.total_tiles, p.nsteps);
    (void)what;
    PROBE_REPORT(ctx, what, grid);
  }
  return KOCR_OK;
}

// the 64-cout row-reuse arrangement (4 x 64 tiles) in fp16 arithmetic: p as launch_conv_w43 filled it for
// conv_w43r_kernel<POOL, 1>, with wgt = d_w4h, pre_a = d_pre_a_h and amax_in set
int launch_w43rh(kocr_ctx* ctx, W4Params& p, bool fuse, int pieces, int mode) {
  // fp16x2: two 256-register blocks per CU (OCC = 2; A/B against one 512-register block with two channel groups of loads in
  // flight: profiles/r06_ab_notes.txt items 1, 8)
  if (mode == 1) return fuse ? w4rh_launch<1, 2, 1, 2>(ctx, p) : w4rh_launch<0, 2, 1, 2>(ctx, p);
  if (pieces == 2) return fuse ? w4rh_launch<1, 2, 0, 2>(ctx, p) : w4rh_launch<0, 2, 0, 2>(ctx, p);
  return fuse ? w4rh_launch<1, 1>(ctx, p) : w4rh_launch<0, 1>(ctx, p);
}

template <int NP, int DIL = 0>
static int w4fh_launch(kocr_ctx* ctx, W4Params& p) {
  const int LDSF = 2 * 6 * NP * 2 * 2 * 256 * 2 + 4 * p.Cout_pad * 4;  // 2 x 24 KB (NP = 2) + the epilogue's coefficients
  static std::atomic<bool> attr_done[64];
  const int dev = ctx->device & 63;
  if (!attr_done[dev]) {
    KOCR_HIP(ctx, hipFuncSetAttribute((const void*)conv_w43fh_kernel<NP, DIL>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      2 * 6 * NP * 2 * 2 * 256 * 2 + W4_COEF_BYTES_MAX));
    attr_done[dev] = true;
  }
  static std::atomic<int> n_cus[64];
  if (!n_cus[dev]) {
    hipDeviceProp_t prop;
    KOCR_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
    n_cus[dev] = prop.multiProcessorCount;
  }
  const int n_cu = n_cus[dev];
  const int grid = p.total_tiles < n_cu ? p.total_tiles : n_cu;
  PROBE_RESET(ctx);
  hipLaunchKernelGGL((conv_w43fh_kernel<NP, DIL>), dim3(grid), dim3(256), LDSF, ctx->stream, p);
  KOCR_HIP(ctx, hipGetLastError());
  {
    char what[64];
    snprintf(what, sizeof what, "conv_w43fh<%d> tiles %d steps %d", NP, p.total_tiles, p.nsteps);
    (void)what;
    PROBE_REPORT(ctx, what, grid);
  }
  return KOCR_OK;
}

// the flattened-pixel arrangement (no fused pooling, dilation 1, H W >= 256) in fp16 arithmetic: p as launch_conv_w43
// filled it for conv_w43_kernel<0>, with wgt = d_w4h, pre_a = d_pre_a_h, amax_in set and amax_out = out.amax (per image)
// ... p.dil != 1: the dilated comb tiles (two pieces only: the one-piece fast mode runs dilated layers in fp16x2)
int launch_w43fh(kocr_ctx* ctx, W4Params& p, int pieces) {
  if (p.dil != 1) return w4fh_launch<2, 1>(ctx, p);
  return pieces == 2 ? w4fh_launch<2>(ctx, p) : w4fh_launch<1>(ctx, p);
}
