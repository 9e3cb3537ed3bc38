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
// Algebra: Cook-Toom / Winograd F(4,3) with the interpolation points 0, +-a, +-b, inf, a = 5/8, b = 3/2 (the textbook
// choice 0, +-1, +-2 amplifies the accumulators' fp32 round-off by up to 8 in the output transform; this symmetric
// set keeps the cheap butterfly structure and has 30 % less error -- restated and compared in numpy in
// tests/test_split_arith_cpu.py).  For the output quad (x0 .. x0+3) of a row and every (ky, c), with
// d_i = in[y+ky-1][x0-1+i][c], i = 0..5 and taps g0, g1, g2:
//     V0 = a^2 b^2 d0 - (a^2 + b^2) d2 + d4                      U0 = g0 / (a^2 b^2)
//     V1,2 = (d4 - b^2 d2) +- a (d3 - b^2 d1)                    U1,2 = (g0 +- a g1 + a^2 g2) / (2 a^2 (a^2 - b^2))
//     V3,4 = (d4 - a^2 d2) +- b (d3 - a^2 d1)                    U3,4 = (g0 +- b g1 + b^2 g2) / (2 b^2 (b^2 - a^2))
//     V5 = a^2 b^2 d1 - (a^2 + b^2) d3 + d5                      U5 = g2
//     M_xi[quad][o] = sum_{ky,c} V_xi U_xi
//     out[x0]   = M0 + (M1 + M2) + (M3 + M4)                     out[x0+1] = a (M1 - M2) + b (M3 - M4)
//     out[x0+2] = a^2 (M1 + M2) + b^2 (M3 + M4)                  out[x0+3] = a^3 (M1 - M2) + b^3 (M3 - M4) + M5
// U is transformed in float64 on the host and rounded once.  fp32 error against an fp64 convolution: about 1.7x that
// of F(2,3) / of a direct fp32 fma chain, inside the bound tests/test_conv_gpu.py states for this kernel.
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
#include "w43_common.h"


// DBG (developer timing experiments, wrong results; only instantiated with -DKOCR_DEV_SWITCHES): 1 = no input
// transform / split VALU work, 2 = no weight stream, 4 = no MFMAs, 8 = no LDS operand fetches
// DIL = 1: dilated 3x3 convolution (taps p.dil pixels apart in x and y, POOL = 0 only) -- the same F(4,3) algebra on
// the comb of pixels x = r + d t: quad q of a row covers x0 + d j, j = 0..3, with x0 = (q / d) 4 d + q % d.
template <int POOL, int DBG = 0, int DIL = 0>
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
  // Gather geometry of a tile: raw-buffer byte offsets of the item's six pixels (an offset of 0x80000000 is out of
  // range and the load returns 0: row / column zero padding and everything past the end), relative to a base one image
  // row + one pixel before the tile's first pixel.  Two sets are alive: the tile being consumed and the next one --
  // the loads run two K-steps ahead and cross the tile boundary first; the choice is a branch-free select so that the
  // K loop stays one basic block chain without vector-memory waits at joins.
  struct Geo {
    unsigned goff[6];
    int gy;
    bool gok;
    const float* base;
  };
  auto make_geo = [&](int L, Geo& g) __attribute__((always_inline)) {
    int mp, nt_unused;
    w4_decode(p, kocr_xcd_remap(L < total ? L : 0, total), nblk_n, mp, nt_unused);
    const int mtb = mp * 2;  // first of the tile's two M-tiles
    int y0a, x0a, y0b, x0b;
    const long pm_a = w4_mtile_pm0<POOL>(p, mtb, y0a, x0a);
    const int mt1 = mtb + 1 < p.total_mtiles ? mtb + 1 : mtb;
    const long pm_b = w4_mtile_pm0<POOL>(p, mt1, y0b, x0b);
    g.base = p.in + (pm_a * p.in_cs + p.in_co) - (long)(p.W + 1) * p.in_cs;
    const int m = qi >> 5, i = qi & 31;
    int rel, x0;
    if constexpr (POOL) {
      const int row = i >> 4, qc = i & 15;
      rel = (m ? (int)(pm_b - pm_a) : 0) + row * p.W + 4 * qc;
      x0 = (m ? x0b : x0a) + 4 * qc;
      g.gy = (m ? y0b : y0a) + row;
      g.gok = L < total && mtb + m < p.total_mtiles;
    } else if constexpr (DIL) {
      // quad index over all rows of the batch -> (row, quad of the row) -> comb position
      const long q0 = (long)mtb * 32;                   // first quad of the tile
      const long row_a = q0 / p.qpr;                    // the base row: offsets are relative to its first pixel
      const long qg = q0 + qi;
      const long row = qg / p.qpr;
      const int qr = (int)(qg - row * p.qpr);
      x0 = (qr / p.dil) * 4 * p.dil + qr % p.dil;
      rel = (int)((row - row_a) * p.W) + x0;
      g.gok = L < total && qg * 4 < p.Mtotal;
      g.gy = (int)(row % p.H);
      g.base = p.in + ((row_a * p.W) * p.in_cs + p.in_co) - (long)(p.dil * p.W + p.dil) * p.in_cs;
    } else {
      rel = 4 * qi;
      const long gp = pm_a + rel;
      g.gok = L < total && gp < p.Mtotal;
      x0 = (int)(gp % p.W);
      g.gy = (int)((gp / p.W) % p.H);
    }
    const int dd = DIL ? p.dil : 1;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const bool pad = (k == 0 && x0 < dd) || (k == 5 && x0 + 4 * dd >= p.W);  // column zero padding
      g.goff[k] = pad ? OOB : (unsigned)(((rel + k * dd) * p.in_cs + q4 * 4) * 4);
    }
  };
  Geo gc, gn;
  int ld_ky = 0, ld_cg = 0;  // position of the NEXT K-step to load inside its tile
  bool ld_next = false;      // ... and whether that tile is already the next one
  const int ncg = p.Cin >> 4;
  auto load_raw = [&](v4f (&raw)[6]) __attribute__((always_inline)) {
    const int dd = DIL ? p.dil : 1;
    const int soff = (ld_ky * dd * p.W * p.in_cs + ld_cg * 16) * 4;
    const int gy = ld_next ? gn.gy : gc.gy;
    const bool ok = (ld_next ? gn.gok : gc.gok) & ((unsigned)(gy + dd * (ld_ky - 1)) < (unsigned)p.H);
    const unsigned kill = ok ? 0u : OOB;  // valid offsets are < 2^31: OR-ing the top bit pushes the load out of range
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
  // input transform of point xi (fp32, fixed operation order), split, 3 x 8 bytes into LDS
  auto produce_point = [&](const v4f (&d)[6], unsigned short* bufp, int xi) __attribute__((always_inline)) {
    if constexpr (DBG & 1) {
      unsigned short* dst0 = bufp + xi * 3 * PLANE + ldst;
      const u2v r = __builtin_bit_cast(u2v, __builtin_shufflevector(d[xi], d[xi], 0, 1));
      *reinterpret_cast<u2v*>(dst0) = r;
      *reinterpret_cast<u2v*>(dst0 + PLANE) = r;
      *reinterpret_cast<u2v*>(dst0 + 2 * PLANE) = r;
      return;
    }
    v4f V;
    switch (xi) {
      case 0: V = (W4_A2B2 * d[0] - W4_A2PB2 * d[2]) + d[4]; break;
      case 1: V = (d[4] - W4_B2 * d[2]) + W4_A * (d[3] - W4_B2 * d[1]); break;
      case 2: V = (d[4] - W4_B2 * d[2]) - W4_A * (d[3] - W4_B2 * d[1]); break;
      case 3: V = (d[4] - W4_A2 * d[2]) + W4_B * (d[3] - W4_A2 * d[1]); break;
      case 4: V = (d[4] - W4_A2 * d[2]) - W4_B * (d[3] - W4_A2 * d[1]); break;
      default: V = (W4_A2B2 * d[1] - W4_A2PB2 * d[3]) + d[5]; break;
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
    if constexpr (DBG & 8) return;
    const unsigned short* base = bufp + xi * 3 * PLANE + a_lane;
#pragma unroll
    for (int s = 2; s >= 0; --s)
#pragma unroll
      for (int m = 0; m < 2; ++m) a[m][s] = *reinterpret_cast<const bf8*>(base + s * PLANE + m * 2 * KH_STRIDE);
  };
  auto mfma12 = [&](const bf8 (&a)[2][3], int xi) __attribute__((always_inline)) {
    if constexpr (DBG & 4) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int s = 0; s < 3; ++s) acc[xi][m][s] += __builtin_bit_cast(v4f, a[m][s])[0] + __builtin_bit_cast(v4f, bw[xi][s])[0];
      return;
    }
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
  // One K-step: consume `bufc` (6 points x 12 MFMAs) while producing the NEXT step from `raw` into `bufn`.  Per point:
  // fetch the next point's A operands from LDS, transform + split + store one point of the next step (VALU work that
  // issues in the shadow of this point's MFMAs -- the group barriers interleave 1 MFMA : 3 VALU), 12 MFMAs, then
  // re-fetch this point's weights for the next K-step (a full step ahead; the fences pin that order).  The block
  // barrier that publishes the next step sits BEFORE the last point's MFMAs: by then this wave has stored all six
  // points and all its reads of `bufc` have landed, so the next step's first operands are fetched behind 12 MFMAs
  // instead of exposing the LDS latency after the barrier.  a0 holds point 0 on entry and on exit (of the next step).
  bf8 a0[2][3], a1[2][3];
  auto step = [&](const unsigned short* bufc, unsigned short* bufn, const v4f (&raw)[6],
                  const unsigned short* w_next) __attribute__((always_inline)) {
    auto load_b = [&](int xi) __attribute__((always_inline)) {
      if constexpr (DBG & 2) return;
#pragma unroll
      for (int s = 0; s < 3; ++s) bw[xi][s] = *reinterpret_cast<const bf8*>(w_next + (size_t)(xi * 3 + s) * 64 * 8);
    };
    auto interleave = [&]() __attribute__((always_inline)) {
      __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);  // the 6 LDS fetches of the next point first
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);  // 3 VALU
      }
      __builtin_amdgcn_sched_group_barrier(0x200, 3, 0);  // the point's 3 LDS stores
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    };
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      __builtin_amdgcn_sched_barrier(0);
      load_a(a1, bufc, 2 * q + 1);
      produce_point(raw, bufn, 2 * q);
      mfma12(a0, 2 * q);
      interleave();
      __builtin_amdgcn_sched_barrier(0);
      load_b(2 * q);
      __builtin_amdgcn_sched_barrier(0);
      if (q < 2) {
        load_a(a0, bufc, 2 * q + 2);
        produce_point(raw, bufn, 2 * q + 1);
        mfma12(a1, 2 * q + 1);
        interleave();
      } else {
        produce_point(raw, bufn, 5);
        __syncthreads();  // next step complete in bufn, bufc free (waits for this wave's LDS traffic too)
        load_a(a0, bufn, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma12(a1, 5);
      }
      __builtin_amdgcn_sched_barrier(0);
      load_b(2 * q + 1);
    }
  };

  // ------------------------------------------------------------------------------------------------
  // pipeline prologue
  // ------------------------------------------------------------------------------------------------
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
      for (int s = 0; s < 3; ++s) bw[xi][s] = *reinterpret_cast<const bf8*>(w0 + (size_t)(xi * 3 + s) * 64 * 8);
  }
#pragma unroll
  for (int xi = 0; xi < 6; ++xi) produce_point(rawA, As, xi);
  load_raw(rawA);  // global step 2
  __syncthreads();
  load_a(a0, As, 0);

  for (int L = blockIdx.x; L < total; L += G) {
    int mp, nt, mp_n, nt_n;
    w4_decode(p, kocr_xcd_remap(L, total), nblk_n, mp, nt);
    w4_decode(p, kocr_xcd_remap(L + G < total ? L + G : L, total), nblk_n, mp_n, nt_n);
    const int mtb = mp * 2;
    const unsigned short* w_ptr = w_tile(nt);
    const unsigned short* w_after = w_tile(nt_n);
#pragma unroll
    for (int x = 0; x < 6; ++x)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][m][r] = 0.f;
    // drain the previous tile's stores once here (their unknown count must not merge into the K loop's waits)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (int s = 0; s < ns; s += 2) {
      // even global step: consume buffer 0, produce the odd step (rawB) into buffer 1, then refill rawB (step + 3)
      step(As, As + BUF, rawB, w_ptr + (size_t)(s + 1) * w_step);
      load_raw(rawB);
      // odd global step: consume buffer 1, produce the next even step (rawA; possibly the next tile's first) into 0
      step(As + BUF, As, rawA, s + 2 < ns ? w_ptr + (size_t)(s + 2) * w_step : w_after);
      load_raw(rawA);
    }
    // the loads are now inside the next tile: it becomes the current one
    gc = gn;
    make_geo(L + 2 * G, gn);
    ld_next = false;

    // ---- epilogue: 32x32 C/D map: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) ----------------
    {
      const int n = (nt * 4 + wn) * 32 + l31;
      const int nc = n < p.Cout ? n : p.Cout - 1;
      const float pa = p.pre_a[nc], pb = p.pre_b[nc];
      const bool has_post = p.post_a != nullptr;
      const float qa = has_post ? p.post_a[nc] : 1.f, qb = has_post ? p.post_b[nc] : 0.f;
      const bool live = n < p.Cout;
      // branch-free epilogue arithmetic: ReLU as a max with 0 or -inf; the CRNN's post-ReLU BatchNorm affine (has_post) runs
      // as its own pass under a wave-uniform branch instead of a per-element select
      const float lo = p.relu ? 0.f : -INFINITY;
      auto act = [&](float v) { return fmaxf(v * pa + pb, lo); };
      // inverse transform + BN + ReLU in place: acc[0..3][m][r] become the quad's four outputs
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float m0 = acc[0][m][r], m1 = acc[1][m][r], m2 = acc[2][m][r], m3 = acc[3][m][r], m4 = acc[4][m][r],
                      m5 = acc[5][m][r];
          const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
          acc[0][m][r] = act((m0 + s12) + s34);
          acc[1][m][r] = act(W4_A * d12 + W4_B * d34);
          acc[2][m][r] = act(W4_A2 * s12 + W4_B2 * s34);
          acc[3][m][r] = act((W4_A3 * d12 + W4_B3 * d34) + m5);
        }
      if (has_post) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][m][r] = acc[j][m][r] * qa + qb;
      }
      // the output pixel stride is made opaque per tile: hoisted out of the persistent loop the 128 store offsets would be
      // kept in (spilled) scalar registers and fetched back with one v_readlane per store
      int ocs4 = p.out_cs * 4;
      asm volatile("" : "+s"(ocs4));
      int pcs4 = p.pool_cs * 4;
      asm volatile("" : "+s"(pcs4));
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
          // pm = (nimg H + y0) W + x0  ->  pooled pixel (nimg H/2 + y0/2) W/2 + x0/2 = (pm - x0) / 4 ... exactly, H and W even
          const long pp0 = ((pm - x0) >> 2) + (x0 >> 1);
          const __amdgpu_buffer_rsrc_t rp = w4_rsrc(p.pool_out + (pp0 * p.pool_cs + p.pool_co), 0x7FFFFFFFu);
          const unsigned vp = mlive ? (unsigned)((8 * l5 * p.pool_cs + n) * 4) : OOB;  // 4 quads = 8 pooled px per l5
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const int pq = 2 * ((r & 3) + 8 * (r >> 2));
            const float v0 = fmaxf(fmaxf(acc[0][m][r], acc[1][m][r]), fmaxf(acc[0][m][r + 8], acc[1][m][r + 8]));
            const float v1 = fmaxf(fmaxf(acc[2][m][r], acc[3][m][r]), fmaxf(acc[2][m][r + 8], acc[3][m][r + 8]));
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v0), rp, vp, pq * pcs4, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v1), rp, vp, (pq + 1) * pcs4, 0);
          }
        }
      } else if constexpr (DIL) {
        // per-register pixel offsets: quad (m, r, l5) of the tile -> (row, comb position), relative to the first
        // pixel of the tile's base row; the quad's four outputs are p.dil pixels apart
        const long q0 = (long)mtb * 32;
        const long row_a = q0 / p.qpr;
        const __amdgpu_buffer_rsrc_t ro = w4_rsrc(p.out + ((row_a * p.W) * p.out_cs + p.out_co), 0x7FFFFFFFu);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const long qg = q0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * l5;
            const long row = qg / p.qpr;
            const int qr = (int)(qg - row * p.qpr);
            const int x0 = (qr / p.dil) * 4 * p.dil + qr % p.dil;
            const unsigned vo = (live && qg * 4 < p.Mtotal) ? (unsigned)((((int)((row - row_a) * p.W) + x0) * p.out_cs + n) * 4) : OOB;
#pragma unroll
            for (int j = 0; j < 4; ++j)
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[j][m][r]), ro, vo, j * p.dil * ocs4, 0);
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

// ===================================================================================================
// conv_w43n_kernel -- the 64-cout ("narrow") arrangement for layers with 32 < Cout <= 64 (slice1.3, upconv3.conv.3):
// tile = 128 quads (512 pixels, four M-tiles) x 64 couts; wave (wm, wn) owns M-tiles {2 wm, 2 wm + 1} x couts
// [32 wn, 32 wn + 32) x 6 points -- the same 192 accumulators and the same K-step of 72 MFMAs as conv_w43_kernel.
// The transform is amortised over half as many couts, so every thread has TWO gather items per K-step (quads q and
// q + 64).  Their raw pixels share two register sets without doubling them: item 0 is transformed during the first
// three points of a K-step (two points per MFMA group) and its registers are refilled at once with the K-step after
// next; item 1 follows in the last three.  LDS: 2 x 72 KB.  dilation 1 only.
// ===================================================================================================
template <int POOL>
__global__ __launch_bounds__(256) void conv_w43n_kernel(W4Params p) {
  constexpr int NMT = 4;                       // M-tiles per block tile
  constexpr int PLANE_N = NMT * 2 * KH_STRIDE;  // one (xi, piece) plane
  constexpr int BUF_N = 6 * 3 * PLANE_N;        // one K-step: 72 KB
  extern __shared__ __attribute__((aligned(16))) unsigned short As[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, wm = wave >> 1;
  const int l31 = lane & 31, l5 = lane >> 5;
  const int nblk_n = p.Cout_pad >> 6;
  const int total = p.total_tiles;
  const int ns = p.nsteps;
  const int G = gridDim.x;
  constexpr unsigned OOB = 0x80000000u;

  // ---- producer state: items it = 0, 1: quad qi + 64 it of the 128-quad tile, channel quad q4 ------------------
  const int qi = tid >> 2, q4 = tid & 3;
  int ldst[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int q = qi + 64 * it;
    ldst[it] = ((q >> 5) * 2 + (q4 >> 1)) * KH_STRIDE + ((((q & 31) * 8) ^ ((q4 >> 1) * 32)) + (q4 & 1) * 4);
  }
  struct Geo {
    unsigned off0[2];  // byte offset of raw pixel d0 of each item (the six pixels are in_cs * 4 bytes apart)
    unsigned flags;    // per item it: bit 4 it: valid, bit 4 it + 1: d0 is left padding, bit 4 it + 2: d5 is right padding
    int gy[2];
    const float* base;
  };
  auto make_geo = [&](int L, Geo& g) __attribute__((always_inline)) {
    int mp, nt_unused;
    w4_decode(p, kocr_xcd_remap(L < total ? L : 0, total), nblk_n, mp, nt_unused);
    const int mtb = mp * NMT;
    int y0a, x0a;
    const long pm_a = w4_mtile_pm0<POOL>(p, mtb, y0a, x0a);
    g.base = p.in + (pm_a * p.in_cs + p.in_co) - (long)(p.W + 1) * p.in_cs;
    g.flags = 0;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int q = qi + 64 * it;
      const int m = q >> 5, i = q & 31;
      int rel, x0;
      bool gok;
      if constexpr (POOL) {
        const int mt = mtb + m;
        int y0b, x0b;
        const long pm_b = w4_mtile_pm0<POOL>(p, mt < p.total_mtiles ? mt : mtb, y0b, x0b);
        const int row = i >> 4, qc = i & 15;
        rel = (int)(pm_b - pm_a) + row * p.W + 4 * qc;
        x0 = x0b + 4 * qc;
        g.gy[it] = y0b + row;
        gok = L < total && mt < p.total_mtiles;
      } else {
        rel = 4 * q;
        const long gp = pm_a + rel;
        gok = L < total && gp < p.Mtotal;
        x0 = (int)(gp % p.W);
        g.gy[it] = (int)((gp / p.W) % p.H);
      }
      g.off0[it] = (unsigned)((rel * p.in_cs + q4 * 4) * 4);
      g.flags |= ((gok ? 1u : 0u) | (x0 == 0 ? 2u : 0u) | (x0 + 4 >= p.W ? 4u : 0u)) << (4 * it);
    }
  };
  Geo gc, gn;
  int ld_ky = 0, ld_cg = 0;  // position of the NEXT K-step to load (shared by both items; advanced after item 1)
  bool ld_next = false;
  const int ncg = p.Cin >> 4;
  v4f raw0[6], raw1[6];
  auto load_item = [&](v4f (&raw)[6], int it) __attribute__((always_inline)) {
    const int soff = (ld_ky * p.W * p.in_cs + ld_cg * 16) * 4;
    const int gy = ld_next ? gn.gy[it] : gc.gy[it];
    const unsigned fl = (ld_next ? gn.flags : gc.flags) >> (4 * it);
    const bool ok = (fl & 1u) & ((unsigned)(gy + ld_ky - 1) < (unsigned)p.H);
    const unsigned kill = ok ? 0u : OOB;
    const unsigned off0 = (ld_next ? gn.off0[it] : gc.off0[it]) | kill;
    const unsigned stride = (unsigned)(p.in_cs * 4);
    const __amdgpu_buffer_rsrc_t rsrc = w4_rsrc(ld_next ? gn.base : gc.base, 0x80000000u);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const unsigned padk = (k == 0 ? ((fl & 2u) ? OOB : 0u) : 0u) | (k == 5 ? ((fl & 4u) ? OOB : 0u) : 0u);
      raw[k] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (off0 + k * stride) | padk, soff, 0));
    }
  };
  auto advance = [&]() __attribute__((always_inline)) {
    const bool wrap_ky = ld_ky == 2;
    ld_ky = wrap_ky ? 0 : ld_ky + 1;
    const bool wrap_cg = wrap_ky && ld_cg == ncg - 1;
    ld_cg = wrap_cg ? 0 : (wrap_ky ? ld_cg + 1 : ld_cg);
    ld_next = ld_next || wrap_cg;
  };
  auto produce_point = [&](const v4f (&d)[6], unsigned short* bufp, int xi, int it) __attribute__((always_inline)) {
    v4f V;
    switch (xi) {
      case 0: V = (W4_A2B2 * d[0] - W4_A2PB2 * d[2]) + d[4]; break;
      case 1: V = (d[4] - W4_B2 * d[2]) + W4_A * (d[3] - W4_B2 * d[1]); break;
      case 2: V = (d[4] - W4_B2 * d[2]) - W4_A * (d[3] - W4_B2 * d[1]); break;
      case 3: V = (d[4] - W4_A2 * d[2]) + W4_B * (d[3] - W4_A2 * d[1]); break;
      case 4: V = (d[4] - W4_A2 * d[2]) - W4_B * (d[3] - W4_A2 * d[1]); break;
      default: V = (W4_A2B2 * d[1] - W4_A2PB2 * d[3]) + d[5]; break;
    }
    u2v h, m, l;
    kocr_split4(V, h, m, l);
    unsigned short* dst = bufp + xi * 3 * PLANE_N + ldst[it];
    *reinterpret_cast<u2v*>(dst) = h;
    *reinterpret_cast<u2v*>(dst + PLANE_N) = m;
    *reinterpret_cast<u2v*>(dst + 2 * PLANE_N) = l;
  };

  // ---- consumer state ------------------------------------------------------------------------------------------
  const int ntiles32 = p.Cout_pad >> 5;
  const size_t w_step = (size_t)ntiles32 * 18 * 64 * 8;
  auto w_tile = [&](int nt) { return p.wgt + ((size_t)(nt * 2 + wn) * 18 * 64 + lane) * 8; };
  bf8 bw[6][3];
  f16v acc[6][2];
  const int a_lane = (wm * 2 * 2 + l5) * KH_STRIDE + ((l31 * 8) ^ (l5 * 32));
  auto load_a = [&](bf8 (&a)[2][3], const unsigned short* bufp, int xi) __attribute__((always_inline)) {
    const unsigned short* base = bufp + xi * 3 * PLANE_N + a_lane;
#pragma unroll
    for (int s = 2; s >= 0; --s)
#pragma unroll
      for (int m = 0; m < 2; ++m) a[m][s] = *reinterpret_cast<const bf8*>(base + s * PLANE_N + m * 2 * KH_STRIDE);
  };
  auto mfma12 = [&](const bf8 (&a)[2][3], int xi) __attribute__((always_inline)) {
    const bf8 b0 = bw[xi][0], b1 = bw[xi][1], b2 = bw[xi][2];
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
  // One K-step: per point, the LDS fetches of the next point, two transform-split-store chunks of the NEXT step
  // (item 0's six points during points 0-2, item 1's during points 3-5) interleaved 1 MFMA : 5 VALU, 12 MFMAs, the
  // re-fetch of this point's weights.  a0 holds point 0 on entry and on exit.
  bf8 a0[2][3], a1[2][3];
  auto step = [&](const unsigned short* bufc, unsigned short* bufn, const unsigned short* w_next) __attribute__((always_inline)) {
    auto load_b = [&](int xi) __attribute__((always_inline)) {
#pragma unroll
      for (int s = 0; s < 3; ++s) bw[xi][s] = *reinterpret_cast<const bf8*>(w_next + (size_t)(xi * 3 + s) * 64 * 8);
    };
    auto interleave = [&]() __attribute__((always_inline)) {
      __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
      for (int i = 0; i < 11; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x200, 6, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    };
#pragma unroll
    for (int ph = 0; ph < 6; ++ph) {
      __builtin_amdgcn_sched_barrier(0);
      bf8(&cur)[2][3] = (ph & 1) ? a1 : a0;
      bf8(&nxt)[2][3] = (ph & 1) ? a0 : a1;
      const int it = ph / 3, pt = 2 * (ph % 3);
      if (ph < 5) {
        load_a(nxt, bufc, ph + 1);
        if (it == 0) {
          produce_point(raw0, bufn, pt, 0);
          produce_point(raw0, bufn, pt + 1, 0);
        } else {
          produce_point(raw1, bufn, pt, 1);
          produce_point(raw1, bufn, pt + 1, 1);
        }
        mfma12(cur, ph);
        interleave();
      } else {
        produce_point(raw1, bufn, 4, 1);
        produce_point(raw1, bufn, 5, 1);
        __syncthreads();  // next step complete in bufn, bufc free
        load_a(nxt, bufn, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma12(cur, 5);
      }
      __builtin_amdgcn_sched_barrier(0);
      load_b(ph);
      if (ph == 2) load_item(raw0, 0);  // item 0 is done with its registers: refill them with the step after next
    }
    load_item(raw1, 1);
    advance();
  };

  // ---- prologue ------------------------------------------------------------------------------------------------
  make_geo(blockIdx.x, gc);
  make_geo(blockIdx.x + G, gn);
  load_item(raw0, 0);
  load_item(raw1, 1);
  advance();  // global step 0 loaded
  {
    int mp0, nt0;
    w4_decode(p, kocr_xcd_remap(blockIdx.x, total), nblk_n, mp0, nt0);
    const unsigned short* w0 = w_tile(nt0);
#pragma unroll
    for (int xi = 0; xi < 6; ++xi)
#pragma unroll
      for (int s = 0; s < 3; ++s) bw[xi][s] = *reinterpret_cast<const bf8*>(w0 + (size_t)(xi * 3 + s) * 64 * 8);
  }
#pragma unroll
  for (int xi = 0; xi < 6; ++xi) {
    produce_point(raw0, As, xi, 0);
    produce_point(raw1, As, xi, 1);
  }
  load_item(raw0, 0);
  load_item(raw1, 1);
  advance();  // global step 1 loaded
  __syncthreads();
  load_a(a0, As, 0);

  for (int L = blockIdx.x; L < total; L += G) {
    int mp, nt, mp_n, nt_n;
    w4_decode(p, kocr_xcd_remap(L, total), nblk_n, mp, nt);
    w4_decode(p, kocr_xcd_remap(L + G < total ? L + G : L, total), nblk_n, mp_n, nt_n);
    const int mtb = mp * NMT;
    const unsigned short* w_ptr = w_tile(nt);
    const unsigned short* w_after = w_tile(nt_n);
#pragma unroll
    for (int x = 0; x < 6; ++x)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][m][r] = 0.f;
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (int s = 0; s < ns; s += 2) {
      step(As, As + BUF_N, w_ptr + (size_t)(s + 1) * w_step);
      step(As + BUF_N, As, s + 2 < ns ? w_ptr + (size_t)(s + 2) * w_step : w_after);
    }
    gc = gn;
    make_geo(L + 2 * G, gn);
    ld_next = false;

    // ---- epilogue (as conv_w43_kernel, M-tiles 2 wm + m) --------------------------------------------------------
    {
      const int n = (nt * 2 + wn) * 32 + l31;
      const int nc = n < p.Cout ? n : p.Cout - 1;
      const float pa = p.pre_a[nc], pb = p.pre_b[nc];
      const bool has_post = p.post_a != nullptr;
      const float qa = has_post ? p.post_a[nc] : 1.f, qb = has_post ? p.post_b[nc] : 0.f;
      const bool live = n < p.Cout;
      // branch-free epilogue arithmetic: ReLU as a max with 0 or -inf; the CRNN's post-ReLU BatchNorm affine (has_post) runs
      // as its own pass under a wave-uniform branch instead of a per-element select
      const float lo = p.relu ? 0.f : -INFINITY;
      auto act = [&](float v) { return fmaxf(v * pa + pb, lo); };
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float m0 = acc[0][m][r], m1 = acc[1][m][r], m2 = acc[2][m][r], m3 = acc[3][m][r], m4 = acc[4][m][r],
                      m5 = acc[5][m][r];
          const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
          acc[0][m][r] = act((m0 + s12) + s34);
          acc[1][m][r] = act(W4_A * d12 + W4_B * d34);
          acc[2][m][r] = act(W4_A2 * s12 + W4_B2 * s34);
          acc[3][m][r] = act((W4_A3 * d12 + W4_B3 * d34) + m5);
        }
      if (has_post) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][m][r] = acc[j][m][r] * qa + qb;
      }
      // the output pixel stride is made opaque per tile: hoisted out of the persistent loop the 128 store offsets would be
      // kept in (spilled) scalar registers and fetched back with one v_readlane per store
      int ocs4 = p.out_cs * 4;
      asm volatile("" : "+s"(ocs4));
      int pcs4 = p.pool_cs * 4;
      asm volatile("" : "+s"(pcs4));
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
          const int mt = mtb + wm * 2 + m;
          int y0, x0;
          const long pm = w4_mtile_pm0<POOL>(p, mt < p.total_mtiles ? mt : mtb, y0, x0);
          const bool mlive = live && mt < p.total_mtiles;
          if (p.write_full) {
            const __amdgpu_buffer_rsrc_t ro = w4_rsrc(p.out + (pm * p.out_cs + p.out_co), 0x7FFFFFFFu);
            const unsigned vo = mlive ? (unsigned)((16 * l5 * p.out_cs + n) * 4) : OOB;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              const int px = 4 * ((r & 3) + 8 * (r >> 2));
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[j][m][r]), ro, vo, (px + j) * ocs4, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[j][m][r + 8]), ro, vo, (px + j + p.W) * ocs4, 0);
              }
            }
          }
          // pm = (nimg H + y0) W + x0  ->  pooled pixel (nimg H/2 + y0/2) W/2 + x0/2 = (pm - x0) / 4 ... exactly, H and W even
          const long pp0 = ((pm - x0) >> 2) + (x0 >> 1);
          const __amdgpu_buffer_rsrc_t rp = w4_rsrc(p.pool_out + (pp0 * p.pool_cs + p.pool_co), 0x7FFFFFFFu);
          const unsigned vp = mlive ? (unsigned)((8 * l5 * p.pool_cs + n) * 4) : OOB;
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const int pq = 2 * ((r & 3) + 8 * (r >> 2));
            const float v0 = fmaxf(fmaxf(acc[0][m][r], acc[1][m][r]), fmaxf(acc[0][m][r + 8], acc[1][m][r + 8]));
            const float v1 = fmaxf(fmaxf(acc[2][m][r], acc[3][m][r]), fmaxf(acc[2][m][r + 8], acc[3][m][r + 8]));
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v0), rp, vp, pq * pcs4, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v1), rp, vp, (pq + 1) * pcs4, 0);
          }
        }
      } else {
        int yy, xx;
        const long pm0 = w4_mtile_pm0<POOL>(p, mtb, yy, xx);
        const long rem = ((long)p.Mtotal - pm0) * ocs4;
        const __amdgpu_buffer_rsrc_t ro =
            w4_rsrc(p.out + (pm0 * p.out_cs + p.out_co), rem < 0x7FFFFFFFL ? (unsigned)(rem > 0 ? rem : 0) : 0x7FFFFFFFu);
        const unsigned vo = live ? (unsigned)((16 * l5 * p.out_cs + n) * 4) : OOB;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int px = 4 * ((wm * 2 + m) * 32 + (r & 3) + 8 * (r >> 2));  // + 16 l5 in vo
#pragma unroll
            for (int j = 0; j < 4; ++j)
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[j][m][r]), ro, vo, (px + j) * ocs4, 0);
          }
      }
    }
  }
}

// ===================================================================================================
// conv_w43r_kernel -- 32 < Cout <= 64 on images that tile as 2 rows x 128 columns (H even, W % 128 == 0: slice1.3,
// upconv3.conv.3): the "row reuse" arrangement.  conv_w43n_kernel above transforms and splits every input row once per
// vertical tap (its K-step is (channel group, ky)) and, with only 64 couts to amortise that over, is issue-bound (9 VALU
// instructions per MFMA, matrix pipe 34 % busy).  Here a tile is 2 output rows x 128 columns and the K loop runs over
// channel groups only: the FOUR input rows y0-1 .. y0+2 of a 16-channel group are transformed and split ONCE into LDS
// (4 rows x 32 quads x 6 points x 3 pieces = 72 KB, double buffered) and the three vertical taps read them back at a
// row offset -- 4 row transforms per 2 output rows instead of 6, spread over 3 x 36 MFMAs per wave.
//   M-tile = 2 rows x 64 columns (the geometry of the fused-pool tiles).  Wave (ph, wn) owns BOTH M-tiles x couts
//   [32 wn, 32 wn + 32) x the three points 3 ph .. 3 ph + 2 = 96 accumulators, so that a weight fragment serves two
//   M-tiles (with one M-tile per wave the weight stream was 10 TB/s from L2 and the chip clocked 1.5 GHz); a group = one
//   (ky, point) = 12 MFMAs alternating between the two M-tiles; 9 groups per channel group, 12 transform-split-store
//   chunks (2 items x 6 points) spread 2,1,1 over them; the weights (conv_w43n's layout and order, one (channel group,
//   ky) step ahead in registers) are re-fetched point by point; one block barrier per channel group.  The output
//   transform is linear in the six points: in the epilogue every wave forms the partial outputs of both M-tiles from its
//   three points, hands the partner wave (ph ^ 1) the partials of M-tile 1 - ph through the LDS buffer the K loop has
//   just released (two block barriers per tile) and finishes M-tile ph itself.
// POOL = 1: fused 2x2 max-pool (full-resolution store optional); POOL = 0: full-resolution store only.
// ===================================================================================================

// GEO (round 3): 0 = tile of 2 rows x 128 columns as described above; 1 = tile of 4 rows x 64 columns (H % 4 == 0,
// W % 64 == 0; the two M-tiles stacked): SIX input rows per FOUR output rows, one full gather item + one HALF item (two
// channels, 8-byte loads) per thread and channel group, 54 KB per buffer -- a quarter less transform / split work again.
template <int POOL, int GEO>
__global__ __launch_bounds__(256) void conv_w43r_kernel(W4Params p) {
  constexpr int NROWS = GEO ? 6 : 4;            // input rows of the tile's window
  constexpr int QPR = GEO ? 16 : 32;            // quads per tile row
  constexpr int KHS = QPR * 8;                  // ushorts of one k half of a row
  constexpr int ROW_STRIDE = 2 * KHS;           // ushorts per input row of a plane
  constexpr int PLANE_R = NROWS * ROW_STRIDE;   // one (xi, piece) plane
  constexpr int BUF_R = 6 * 3 * PLANE_R;        // one channel group: 72 KB / 54 KB
  constexpr int TCOLS = QPR * 4;
  extern __shared__ __attribute__((aligned(16))) unsigned short As[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, ph = wave >> 1;  // cout half, point half (points 3 ph .. 3 ph + 2)
  const int l31 = lane & 31, l5 = lane >> 5;
  const int total = p.total_tiles;  // pairs of M-tiles; one cout block
  const int ncg = p.Cin >> 4;
  const int G = gridDim.x;
  constexpr unsigned OOB = 0x80000000u;

  // pixel tile mp -> M-tile m in conv_w43_kernel's fused-pool numbering: side by side (GEO 0) or the next row pair (GEO 1)
  auto tile_mt = [&](int mp, int m) {
    if constexpr (GEO) {
      const int rq = (int)w4_fdiv((unsigned)mp, p.dv_tpr), cb = mp - rq * p.tiles_per_row;  // rq: (image, row quad)
      return (2 * rq + m) * p.tiles_per_row + cb;
    } else {
      return 2 * mp + m;
    }
  };

  // ---- producer state -----------------------------------------------------------------------------------------
  // item 0: input row r0 of the window, quad qd0, channel quad q4 (16-byte loads).  item 1: GEO 0: row r0 + 2, same quad
  // and channels; GEO 1: row 4 + (tid >> 7), quad (tid >> 3) & 15, channel PAIR tid & 7 (8-byte loads).
  const int q4 = tid & 3, qd0 = (tid >> 2) & (QPR - 1), r0 = GEO ? (tid >> 6) : (tid >> 7);
  const int cp = tid & 7, qd1 = GEO ? ((tid >> 3) & 15) : qd0, r1 = GEO ? 4 + (tid >> 7) : r0 + 2;
  int ldst[2];
  ldst[0] = r0 * ROW_STRIDE + (q4 >> 1) * KHS + (((qd0 * 8) ^ ((q4 >> 1) * 32)) + (q4 & 1) * 4);
  if constexpr (GEO)
    ldst[1] = r1 * ROW_STRIDE + (cp >> 2) * KHS + (((qd1 * 8) ^ ((cp >> 2) * 32)) + (cp & 3) * 2);
  else
    ldst[1] = r1 * ROW_STRIDE + (q4 >> 1) * KHS + (((qd1 * 8) ^ ((q4 >> 1) * 32)) + (q4 & 1) * 4);
  struct Geo {
    unsigned off0[2];  // byte offset of raw pixel d0 of each item
    unsigned ok;       // bit it: the item's input row lies inside the image (and the tile exists)
    const float* base;
  };
  auto make_geo = [&](int L, Geo& g, bool& left, bool& right) __attribute__((always_inline)) {
    const int mp = kocr_xcd_remap(L < total ? L : 0, total);
    int y0, x0;
    const long pm = w4_mtile_pm0<1>(p, tile_mt(mp, 0), y0, x0);
    g.base = p.in + (pm * p.in_cs + p.in_co) - (long)(p.W + 1) * p.in_cs;
    g.off0[0] = (unsigned)(((r0 * p.W + 4 * qd0) * p.in_cs + q4 * 4) * 4);
    g.off0[1] = (unsigned)(((r1 * p.W + 4 * qd1) * p.in_cs + (GEO ? cp * 2 : q4 * 4)) * 4);
    g.ok = ((L < total && (unsigned)(y0 - 1 + r0) < (unsigned)p.H) ? 1u : 0u) |
           ((L < total && (unsigned)(y0 - 1 + r1) < (unsigned)p.H) ? 2u : 0u);
    // d0 / d5 are column zero padding only at the image edges (W is a multiple of the tile width)
    left = x0 == 0;
    right = x0 + TCOLS >= p.W;
  };
  Geo gc, gn;
  bool lc, rc, ln, rn;
  int ld_cg = 0;  // channel group of the NEXT load inside its tile
  bool ld_next = false;
  typedef typename std::conditional<GEO != 0, v2f, v4f>::type raw1_t;
  auto load_item0 = [&](v4f (&raw)[6]) __attribute__((always_inline)) {
    const int soff = ld_cg * 64;
    const bool ok = (ld_next ? gn.ok : gc.ok) & 1u;
    const unsigned off0 = (ld_next ? gn.off0[0] : gc.off0[0]) | (ok ? 0u : OOB);
    const bool left = (ld_next ? ln : lc) && qd0 == 0, right = (ld_next ? rn : rc) && qd0 == QPR - 1;
    const unsigned stride = (unsigned)(p.in_cs * 4);
    const __amdgpu_buffer_rsrc_t rsrc = w4_rsrc(ld_next ? gn.base : gc.base, 0x80000000u);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const unsigned padk = (k == 0 ? (left ? OOB : 0u) : 0u) | (k == 5 ? (right ? OOB : 0u) : 0u);
      raw[k] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (off0 + k * stride) | padk, soff, 0));
    }
  };
  auto load_item1 = [&](raw1_t (&raw)[6]) __attribute__((always_inline)) {
    const int soff = ld_cg * 64;
    const bool ok = ((ld_next ? gn.ok : gc.ok) >> 1) & 1u;
    const unsigned off0 = (ld_next ? gn.off0[1] : gc.off0[1]) | (ok ? 0u : OOB);
    const bool left = (ld_next ? ln : lc) && qd1 == 0, right = (ld_next ? rn : rc) && qd1 == QPR - 1;
    const unsigned stride = (unsigned)(p.in_cs * 4);
    const __amdgpu_buffer_rsrc_t rsrc = w4_rsrc(ld_next ? gn.base : gc.base, 0x80000000u);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const unsigned padk = (k == 0 ? (left ? OOB : 0u) : 0u) | (k == 5 ? (right ? OOB : 0u) : 0u);
      if constexpr (GEO)
        raw[k] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (off0 + k * stride) | padk, soff, 0));
      else
        raw[k] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (off0 + k * stride) | padk, soff, 0));
    }
  };
  auto advance = [&]() __attribute__((always_inline)) {
    const bool wrap = ld_cg == ncg - 1;
    ld_cg = wrap ? 0 : ld_cg + 1;
    ld_next = ld_next || wrap;
  };
  auto produce4 = [&](const v4f (&d)[6], unsigned short* bufp, int xi, int it) __attribute__((always_inline)) {
    const v4f V = w4_transform(d, xi);
    u2v h, m, l;
    kocr_split4(V, h, m, l);
    unsigned short* dst = bufp + xi * 3 * PLANE_R + ldst[it];
    *reinterpret_cast<u2v*>(dst) = h;
    *reinterpret_cast<u2v*>(dst + PLANE_R) = m;
    *reinterpret_cast<u2v*>(dst + 2 * PLANE_R) = l;
  };
  auto produce2 = [&](const v2f (&d)[6], unsigned short* bufp, int xi) __attribute__((always_inline)) {
    const v2f V = w4_transform(d, xi);
    unsigned h, m, l;
    kocr_split2(V, h, m, l);
    unsigned short* dst = bufp + xi * 3 * PLANE_R + ldst[1];
    *reinterpret_cast<unsigned*>(dst) = h;
    *reinterpret_cast<unsigned*>(dst + PLANE_R) = m;
    *reinterpret_cast<unsigned*>(dst + 2 * PLANE_R) = l;
  };
  v4f raw0[6];
  raw1_t raw1[6];
  auto produce_item1 = [&](unsigned short* bufp, int xi) __attribute__((always_inline)) {
    if constexpr (GEO)
      produce2(raw1, bufp, xi);
    else
      produce4(raw1, bufp, xi, 1);
  };

  // ---- consumer state ------------------------------------------------------------------------------------------
  const size_t w_step = (size_t)2 * 18 * 64 * 8;  // ushorts per (channel group, ky) step: two 32-cout tiles
  const unsigned short* w_ptr = p.wgt + ((size_t)wn * 18 * 64 + lane) * 8;
  const int ns = 3 * ncg;
  bf8 bw[3][3];
  f16v acc[3][2];  // [point of this wave's half][M-tile]
  // M row l31 of M-tile m: window row (l31 >> 4) + ky [+ 2 m: GEO 1], quad (l31 & 15) [+ 16 m: GEO 0], k half l5
  const int a_lane = (l31 >> 4) * ROW_STRIDE + l5 * KHS + (((l31 & 15) * 8) ^ (l5 * 32));
  constexpr int M_OFF = GEO ? 2 * ROW_STRIDE : 128;  // M-tile 1: two rows down / 16 quads to the right
  auto load_a = [&](bf8 (&a)[2][3], const unsigned short* bufp, int ky, int pl) __attribute__((always_inline)) {
    const unsigned short* base = bufp + a_lane + ky * ROW_STRIDE + (3 * ph + pl) * 3 * PLANE_R;
#pragma unroll
    for (int s = 2; s >= 0; --s)
#pragma unroll
      for (int m = 0; m < 2; ++m) a[m][s] = *reinterpret_cast<const bf8*>(base + s * PLANE_R + m * M_OFF);
  };
  auto mfma12 = [&](const bf8 (&a)[2][3], int pl) __attribute__((always_inline)) {
    const bf8 b0 = bw[pl][0], b1 = bw[pl][1], b2 = bw[pl][2];
    // smallest terms first; the two M-tiles alternate so consecutive MFMAs are independent
#pragma unroll
    for (int m = 0; m < 2; ++m) acc[pl][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][2], b0, acc[pl][m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 2; ++m) acc[pl][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][0], b2, acc[pl][m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 2; ++m) acc[pl][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][1], b1, acc[pl][m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 2; ++m) acc[pl][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][1], b0, acc[pl][m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 2; ++m) acc[pl][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][0], b1, acc[pl][m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 2; ++m) acc[pl][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][0], b0, acc[pl][m], 0, 0, 0);
  };
  // One channel group: consume `bufc` in 9 groups (ky, point pair) while transforming the NEXT channel group from
  // raw0 / raw1 into `bufn` (chunks 2,1,1 per three groups; item 0 is finished after group 3 and its registers are refilled
  // at once with the channel group after next, item 1 after group 8).  s0 = index of this channel group's first weight
  // step; the weights of step s0 + ky + 1 replace those of (s0 + ky) point pair by point pair.  a0 (flip = 0) /
  // a1 (flip = 1) holds group 0 on entry, the other one the next channel group's group 0 on exit.
  bf8 a0[2][3], a1[2][3];
  auto phase = [&](const unsigned short* bufc, unsigned short* bufn, int s0, int flip) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < 9; ++g) {
      __builtin_amdgcn_sched_barrier(0);
      const int ky = g / 3, pp = g - ky * 3;
      // nine groups: the roles of a0 / a1 swap from one channel group to the next (flip), back after two
      bf8(&cur)[2][3] = ((g & 1) ^ flip) ? a1 : a0;
      bf8(&nxt)[2][3] = ((g & 1) ^ flip) ? a0 : a1;
      const int nchunks = (g % 3 == 0) ? 2 : 1;
      const int c0 = (g / 3) * 4 + (g % 3 == 0 ? 0 : g % 3 + 1);  // first chunk of this group: 0,2,3 | 4,6,7 | 8,10,11
      if (g < 8) load_a(nxt, bufc, (g + 1) / 3, (g + 1) % 3);
#pragma unroll
      for (int c = c0; c < c0 + nchunks; ++c) {
        if (c < 6)
          produce4(raw0, bufn, c, 0);
        else
          produce_item1(bufn, c - 6);
      }
      if (g < 8) {
        mfma12(cur, pp);
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);  // the LDS fetches of the next group first
        // VALU per MFMA: two full chunks 5, one full chunk 3, (GEO 1) two half chunks 3, one half chunk 2
        auto ilv = [&](auto v_c, auto st_c) __attribute__((always_inline)) {
          constexpr int V = decltype(v_c)::value, ST = decltype(st_c)::value;
#pragma unroll
          for (int i = 0; i < 11; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, V, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x200, ST, 0);
        };
        const bool half = GEO && c0 >= 6;
        if (nchunks == 2) {
          if (half)
            ilv(std::integral_constant<int, 3>{}, std::integral_constant<int, 6>{});
          else
            ilv(std::integral_constant<int, 5>{}, std::integral_constant<int, 6>{});
        } else {
          if (half)
            ilv(std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{});
          else
            ilv(std::integral_constant<int, 3>{}, std::integral_constant<int, 3>{});
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      } else {
        __syncthreads();  // the next channel group is complete in bufn, bufc is free
        load_a(nxt, bufn, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma12(cur, pp);
      }
      __builtin_amdgcn_sched_barrier(0);
      {  // this point pair's weights of the next step
        int sn = s0 + ky + 1;
        sn = sn >= ns ? sn - ns : sn;
        const unsigned short* wq = w_ptr + (size_t)sn * w_step;
#pragma unroll
        for (int s = 0; s < 3; ++s) bw[pp][s] = *reinterpret_cast<const bf8*>(wq + (size_t)((3 * ph + pp) * 3 + s) * 64 * 8);
      }
      if (g == 3) load_item0(raw0);
    }
    load_item1(raw1);
    advance();
  };

  // ---- prologue ------------------------------------------------------------------------------------------------
  make_geo(blockIdx.x, gc, lc, rc);
  make_geo(blockIdx.x + G, gn, ln, rn);
  load_item0(raw0);
  load_item1(raw1);
  advance();  // channel group 0 loaded
#pragma unroll
  for (int pl = 0; pl < 3; ++pl)
#pragma unroll
    for (int s = 0; s < 3; ++s) bw[pl][s] = *reinterpret_cast<const bf8*>(w_ptr + (size_t)((3 * ph + pl) * 3 + s) * 64 * 8);
#pragma unroll
  for (int xi = 0; xi < 6; ++xi) {
    produce4(raw0, As, xi, 0);
    produce_item1(As, xi);
  }
  load_item0(raw0);
  load_item1(raw1);
  advance();  // channel group 1 loaded
  __syncthreads();
  load_a(a0, As, 0, 0);

  for (int L = blockIdx.x; L < total; L += G) {
    const int mp = kocr_xcd_remap(L, total);
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][m][r] = 0.f;
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (int cg = 0; cg < ncg; cg += 2) {
      phase(As, As + BUF_R, 3 * cg, 0);
      phase(As + BUF_R, As, 3 * cg + 3, 1);
    }
    gc = gn;
    lc = ln;
    rc = rn;
    make_geo(L + 2 * G, gn, ln, rn);
    ld_next = false;

    // ---- epilogue (stores as conv_w43_kernel's fused-pool tiles; this wave finishes M-tile 2 mp + ph) ------------------
    {
      const int n = wn * 32 + l31;
      const int nc = n < p.Cout ? n : p.Cout - 1;
      const float pa = p.pre_a[nc], pb = p.pre_b[nc];
      const bool has_post = p.post_a != nullptr;
      const float qa = has_post ? p.post_a[nc] : 1.f, qb = has_post ? p.post_b[nc] : 0.f;
      const bool live = n < p.Cout;
      // branch-free epilogue arithmetic: ReLU as a max with 0 or -inf; the CRNN's post-ReLU BatchNorm affine (has_post) runs
      // as its own pass under a wave-uniform branch instead of a per-element select
      const float lo = p.relu ? 0.f : -INFINITY;
      auto act = [&](float v) { return fmaxf(v * pa + pb, lo); };
      // The output transform is linear in the six points: every wave forms the partial outputs of BOTH M-tiles from its
      // three points, hands the partials of the partner's M-tile (1 - ph) over through the LDS buffer the K loop has just
      // released (16 KB per wave: [wave][r][lane] float4, the partner = wave ^ 2 has the same lane <-> (quad, cout) map)
      // and finishes M-tile ph itself.  Point half 0 holds (m0, m1, m2), half 1 (m3, m4, m5).
      v4f* xch = reinterpret_cast<v4f*>(As + BUF_R);  // the K loop's second buffer: free since the last block barrier
      float out[4][16];
      // PH (= ph, wave-uniform) and the M-tile index are compile-time constants inside: no dynamic register indexing
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
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float o[4];
          partial(std::integral_constant<int, 1 - PH>{}, r, o);
          xch[(wave * 16 + r) * 64 + lane] = v4f{o[0], o[1], o[2], o[3]};
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float o[4];
          partial(std::integral_constant<int, PH>{}, r, o);
#pragma unroll
          for (int j = 0; j < 4; ++j) out[j][r] = o[j];
        }
      };
      if (ph == 0)
        halves(std::integral_constant<int, 0>{});
      else
        halves(std::integral_constant<int, 1>{});
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const v4f q = xch[((wave ^ 2) * 16 + r) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 4; ++j) out[j][r] = act(out[j][r] + q[j]);
      }
      if (has_post) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int j = 0; j < 4; ++j) out[j][r] = out[j][r] * qa + qb;
      }
      __syncthreads();  // the next channel group is transformed into this buffer
      int ocs4 = p.out_cs * 4;
      asm volatile("" : "+s"(ocs4));
      int pcs4 = p.pool_cs * 4;
      asm volatile("" : "+s"(pcs4));
      int y0, x0;
      const long pm = w4_mtile_pm0<1>(p, tile_mt(mp, ph), y0, x0);
      if (p.amax_out || p.amax_pool) {  // per-image max |x| (Tensor::amax): the M-tile lies inside one image
        float mx = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fabsf(out[j][r]));
        mx = live ? mx : 0.f;
        const unsigned nimg = w4_fdiv((unsigned)pm, p.dv_hw);
        if (p.amax_out) kocr_amax_update(p.amax_out + nimg, mx);
        if (p.amax_pool) kocr_amax_update(p.amax_pool + nimg, mx);
      }
      if (!POOL || p.write_full) {
        const __amdgpu_buffer_rsrc_t ro = w4_rsrc(p.out + (pm * p.out_cs + p.out_co), 0x7FFFFFFFu);
        const unsigned vo = live ? (unsigned)((16 * l5 * p.out_cs + n) * 4) : OOB;  // 4 quads = 16 px per l5
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int px = 4 * ((r & 3) + 8 * (r >> 2));  // quad column (r&3) + 8 (r>>2) [+ 4 l5] of row y0
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(out[j][r]), ro, vo, (px + j) * ocs4, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(out[j][r + 8]), ro, vo, (px + j + p.W) * ocs4, 0);
          }
        }
      }
      if constexpr (POOL) {
        // 2x2 max: rows y0 (r) and y0 + 1 (r + 8), columns (0,1) and (2,3) of the quad
        const long pp0 = ((pm - x0) >> 2) + (x0 >> 1);  // (nimg H/2 + y0/2) W/2 + x0/2: H, W even
        const __amdgpu_buffer_rsrc_t rp = w4_rsrc(p.pool_out + (pp0 * p.pool_cs + p.pool_co), 0x7FFFFFFFu);
        const unsigned vp = live ? (unsigned)((8 * l5 * p.pool_cs + n) * 4) : OOB;  // 4 quads = 8 pooled px per l5
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int pq = 2 * ((r & 3) + 8 * (r >> 2));
          const float v0 = fmaxf(fmaxf(out[0][r], out[1][r]), fmaxf(out[0][r + 8], out[1][r + 8]));
          const float v1 = fmaxf(fmaxf(out[2][r], out[3][r]), fmaxf(out[2][r + 8], out[3][r + 8]));
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v0), rp, vp, pq * pcs4, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v1), rp, vp, (pq + 1) * pcs4, 0);
        }
      }
    }
  }
}

// ===================================================================================================
// conv_w43v_kernel -- the row-reuse idea of conv_w43r_kernel for layers with Cout > 64 (round 3).  A tile is a block of
// output rows x columns of ONE image (256 pixels = two M-tiles) x 128 couts; the K loop runs over channel groups only:
// every input row the tile touches is transformed and split ONCE per 16-channel group into LDS (double buffered) and the
// three vertical taps read it back at a row offset:
//   GEO = 1: tile = 4 rows x 64 columns (M-tiles of 2 rows x 64 columns stacked), 6 input rows per 4 output rows
//            (conv_w43_kernel: 12), 54 KB per buffer, one full gather item (row, quad, channel quad: 16-byte loads) + one
//            HALF item (two channels, 8-byte loads) per thread and channel group;
//   GEO = 2: tile = 8 rows x 32 columns (M-tiles of 4 rows x 32 columns stacked; W % 32 == 0, H % 8 == 0: the 96-wide
//            layers), 10 input rows per 8 output rows, 45 KB per buffer, one full item + one half item per thread (the
//            64 half-row items are produced twice, by waves 0-1 and again by waves 2-3, so that every wave runs the same
//            straight-line stream).  Two window rows share one 256-byte LDS line (row pair, k half) so that the 16-byte A
//            reads of a 4-row M-tile stay conflict-free.  No fused pooling.
// That is a third to a half less transform + split VALU work, LDS stores and raw-pixel loads per MFMA than conv_w43_kernel
// and one block barrier per 216 MFMAs instead of three.  Wave wn owns both M-tiles x couts [32 wn, 32 wn + 32) x all six
// points = 192 accumulators, exactly as in conv_w43_kernel, so the output transform stays inside the wave; weights:
// conv_w43_kernel's layout and order ([16-ch group][ky][32-cout tile][xi][piece][lane][8]), one (channel group, ky)
// step ahead in registers.  Per channel group: ky = 0 consumes 6 points while item 0's six points of the NEXT channel
// group are produced (one per point, 1 MFMA : 3 VALU), ky = 1 the same with item 1, ky = 2 produces nothing.
// (A 2 rows x 128 columns geometry, GEO = 0, exists for the 64-cout kernel conv_w43r only.)
// Needs Cin % 32 == 0, dilation 1.  POOL = 1 (GEO 1): fused 2x2 max-pool (full-resolution store optional).
// ===================================================================================================
template <int POOL, int GEO>
__global__ __launch_bounds__(256) void conv_w43v_kernel(W4Params p) {
  static_assert(GEO == 1 || GEO == 2, "4 x 64 or 8 x 32 tiles");
  static_assert(!(POOL && GEO == 2), "no fused pooling on 8 x 32 tiles");
  constexpr int NROWS = GEO == 2 ? 10 : 6;  // input rows of the tile's window
  constexpr int QPR = GEO == 2 ? 8 : 16;    // quads per tile row
  constexpr int KHS = QPR * 8;                             // ushorts of one k half of a row: QPR quads x 8 channels
  constexpr int ROW_STRIDE = 2 * KHS;                      // GEO 1: ushorts per input row of a plane
  constexpr int PLANE_R = NROWS * 2 * KHS;                 // one (xi, piece) plane
  constexpr int BUF_R = 6 * 3 * PLANE_R;                   // one channel group: 54 / 45 KB
  constexpr int TCOLS = QPR * 4;                           // tile columns
  extern __shared__ __attribute__((aligned(16))) unsigned short As[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave = cout sub-tile
  const int l31 = lane & 31, l5 = lane >> 5;
  const int nblk_n = p.Cout_pad >> 7;
  const int total = p.total_tiles;
  const int ncg = p.Cin >> 4;
  const int ns = 3 * ncg;
  const int G = gridDim.x;
  constexpr unsigned OOB = 0x80000000u;

  // ushort offset of the 16-byte slot (window row w, k half kh, quad q) inside a plane.  GEO 2 puts the two rows of a
  // row pair into one 256-byte line per k half: with rows 256 bytes apart the four rows of an M-tile would hit the
  // same banks two by two.
  auto slot = [&](int w, int kh, int q) {
    if constexpr (GEO == 2)
      return ((w >> 1) * 2 + kh) * 128 + (w & 1) * 64 + ((q * 8) ^ (kh * 32));
    else
      return w * ROW_STRIDE + kh * KHS + ((q * 8) ^ (kh * 32));
  };
  // first pixel of pixel tile mp (flattened (n, y, x) index), its row and column
  auto tile_org = [&](int mp, int& y0, int& x0) -> long {
    if constexpr (GEO == 2) {
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
  // item 0: input row r0 of the window, quad qd0, channel quad q4 (16-byte loads).  item 1, a HALF item: row r1, quad qd1,
  // channel PAIR tid & 7 (8-byte loads).
  const int q4 = tid & 3, qd0 = (tid >> 2) & (QPR - 1), r0 = GEO == 2 ? (tid >> 5) : (tid >> 6);
  const int cp = tid & 7;
  const int qd1 = GEO == 2 ? ((tid >> 3) & 7) : ((tid >> 3) & 15);
  const int r1 = GEO == 2 ? 8 + ((tid >> 6) & 1) : 4 + (tid >> 7);
  int ldst[2];
  ldst[0] = slot(r0, q4 >> 1, qd0) + (q4 & 1) * 4;
  ldst[1] = slot(r1, cp >> 2, qd1) + (cp & 3) * 2;
  struct Geo {
    unsigned off0[2];  // byte offset of raw pixel d0 of each item
    unsigned ok;       // bit it: the item's input row lies inside the image (and the tile exists)
    const float* base;
  };
  auto make_geo = [&](int L, Geo& g, bool& left, bool& right) __attribute__((always_inline)) {
    int mp, nt_unused;
    w4_decode(p, kocr_xcd_remap(L < total ? L : 0, total), nblk_n, mp, nt_unused);
    int y0, x0;
    const long pm = tile_org(mp, y0, x0);
    g.base = p.in + (pm * p.in_cs + p.in_co) - (long)(p.W + 1) * p.in_cs;
    g.off0[0] = (unsigned)(((r0 * p.W + 4 * qd0) * p.in_cs + q4 * 4) * 4);
    g.off0[1] = (unsigned)(((r1 * p.W + 4 * qd1) * p.in_cs + cp * 2) * 4);
    g.ok = ((L < total && (unsigned)(y0 - 1 + r0) < (unsigned)p.H) ? 1u : 0u) |
           ((L < total && (unsigned)(y0 - 1 + r1) < (unsigned)p.H) ? 2u : 0u);
    // d0 / d5 are column zero padding only at the image edges (W is a multiple of the tile width)
    left = x0 == 0;
    right = x0 + TCOLS >= p.W;
  };
  Geo gc, gn;
  bool lc, rc, ln, rn;
  int ld_cg = 0;  // channel group of the NEXT load inside its tile
  bool ld_next = false;
  auto load_item0 = [&](v4f (&raw)[6]) __attribute__((always_inline)) {
    const int soff = ld_cg * 64;
    const bool ok = (ld_next ? gn.ok : gc.ok) & 1u;
    const unsigned off0 = (ld_next ? gn.off0[0] : gc.off0[0]) | (ok ? 0u : OOB);
    const bool left = (ld_next ? ln : lc) && qd0 == 0, right = (ld_next ? rn : rc) && qd0 == QPR - 1;
    const unsigned stride = (unsigned)(p.in_cs * 4);
    const __amdgpu_buffer_rsrc_t rsrc = w4_rsrc(ld_next ? gn.base : gc.base, 0x80000000u);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const unsigned padk = (k == 0 ? (left ? OOB : 0u) : 0u) | (k == 5 ? (right ? OOB : 0u) : 0u);
      raw[k] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (off0 + k * stride) | padk, soff, 0));
    }
  };
  auto load_item1 = [&](v2f (&raw)[6]) __attribute__((always_inline)) {
    const int soff = ld_cg * 64;
    const bool ok = ((ld_next ? gn.ok : gc.ok) >> 1) & 1u;
    const unsigned off0 = (ld_next ? gn.off0[1] : gc.off0[1]) | (ok ? 0u : OOB);
    const bool left = (ld_next ? ln : lc) && qd1 == 0, right = (ld_next ? rn : rc) && qd1 == QPR - 1;
    const unsigned stride = (unsigned)(p.in_cs * 4);
    const __amdgpu_buffer_rsrc_t rsrc = w4_rsrc(ld_next ? gn.base : gc.base, 0x80000000u);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const unsigned padk = (k == 0 ? (left ? OOB : 0u) : 0u) | (k == 5 ? (right ? OOB : 0u) : 0u);
      raw[k] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (off0 + k * stride) | padk, soff, 0));
    }
  };
  auto advance = [&]() __attribute__((always_inline)) {
    const bool wrap = ld_cg == ncg - 1;
    ld_cg = wrap ? 0 : ld_cg + 1;
    ld_next = ld_next || wrap;
  };
  auto produce4 = [&](const v4f (&d)[6], unsigned short* bufp, int xi, int it) __attribute__((always_inline)) {
    const v4f V = w4_transform(d, xi);
    u2v h, m, l;
    kocr_split4(V, h, m, l);
    unsigned short* dst = bufp + xi * 3 * PLANE_R + ldst[it];
    *reinterpret_cast<u2v*>(dst) = h;
    *reinterpret_cast<u2v*>(dst + PLANE_R) = m;
    *reinterpret_cast<u2v*>(dst + 2 * PLANE_R) = l;
  };
  auto produce2 = [&](const v2f (&d)[6], unsigned short* bufp, int xi) __attribute__((always_inline)) {
    const v2f V = w4_transform(d, xi);
    unsigned h, m, l;
    kocr_split2(V, h, m, l);
    unsigned short* dst = bufp + xi * 3 * PLANE_R + ldst[1];
    *reinterpret_cast<unsigned*>(dst) = h;
    *reinterpret_cast<unsigned*>(dst + PLANE_R) = m;
    *reinterpret_cast<unsigned*>(dst + 2 * PLANE_R) = l;
  };
  v4f raw0[6];
  v2f raw1[6];
  auto produce_item1 = [&](unsigned short* bufp, int xi) __attribute__((always_inline)) { produce2(raw1, bufp, xi); };

  // ---- consumer state ------------------------------------------------------------------------------------------
  const int ntiles32 = p.Cout_pad >> 5;
  const size_t w_step = (size_t)ntiles32 * 18 * 64 * 8;  // ushorts per (channel group, ky) step
  auto w_tile = [&](int nt) { return p.wgt + ((size_t)(nt * 4 + wn) * 18 * 64 + lane) * 8; };
  bf8 bw[6][3];
  f16v acc[6][2];
  // A operand: M row l31 of M-tile m, tap ky.  GEO 1: window row (l31 >> 4) + ky + 2 m, quad l31 & 15; GEO 2: window row
  // (l31 >> 3) + ky + 4 m, quad l31 & 7 -- rows advance in steps of two per 256 ushorts, so an odd row offset starts from
  // the lane's NEXT row (a_lane1).  k half l5.
  const int a_lane = GEO == 2 ? slot(l31 >> 3, l5, l31 & 7) : slot(l31 >> 4, l5, l31 & 15);
  const int a_lane1 = GEO == 2 ? slot((l31 >> 3) + 1, l5, l31 & 7) : 0;
  auto load_a = [&](bf8 (&a)[2][3], const unsigned short* bufp, int ky, int xi) __attribute__((always_inline)) {
    const unsigned short* plane = bufp + xi * 3 * PLANE_R;
#pragma unroll
    for (int s = 2; s >= 0; --s)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        int off;
        if constexpr (GEO == 2) {
          const int c = ky + 4 * m;
          off = ((c & 1) ? a_lane1 : a_lane) + (c >> 1) * 256;
        } else {
          off = a_lane + (ky + 2 * m) * ROW_STRIDE;
        }
        a[m][s] = *reinterpret_cast<const bf8*>(plane + s * PLANE_R + off);
      }
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
  // One (channel group, ky) step: consume rows ky .. (+ M-tile offset) of `bufc` (6 points x 12 MFMAs); KY = 0 / 1
  // also transforms item 0 / 1 of the NEXT channel group into `bufn`, one point per MFMA group.  The weights of the next
  // step (w_next) replace this step's point by point.  a0 holds point 0 of this step on entry and point 0 of the next
  // step on exit; for KY = 2 the next step lives in `bufn`, published by the block barrier before the last point's MFMAs.
  bf8 a0[2][3], a1[2][3];
  auto step = [&](auto ky_c, const unsigned short* bufc, unsigned short* bufn, const unsigned short* w_next) __attribute__((always_inline)) {
    constexpr int KY = decltype(ky_c)::value;
    auto load_b = [&](int xi) __attribute__((always_inline)) {
#pragma unroll
      for (int s = 0; s < 3; ++s) bw[xi][s] = *reinterpret_cast<const bf8*>(w_next + (size_t)(xi * 3 + s) * 64 * 8);
    };
    auto produce = [&](int xi) __attribute__((always_inline)) {
      if constexpr (KY == 0) produce4(raw0, bufn, xi, 0);
      if constexpr (KY == 1) produce_item1(bufn, xi);
    };
    auto interleave = [&]() __attribute__((always_inline)) {
      __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);  // the 6 LDS fetches of the next point first
      if constexpr (KY == 0) {
#pragma unroll
        for (int i = 0; i < 10; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);  // 3 VALU
        }
        __builtin_amdgcn_sched_group_barrier(0x200, 3, 0);  // the point's 3 LDS stores
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      } else if constexpr (KY == 1) {  // half item: about half the VALU work
#pragma unroll
        for (int i = 0; i < 10; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x200, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
      }
    };
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      __builtin_amdgcn_sched_barrier(0);
      load_a(a1, bufc, KY, 2 * q + 1);
      produce(2 * q);
      mfma12(a0, 2 * q);
      interleave();
      __builtin_amdgcn_sched_barrier(0);
      load_b(2 * q);
      __builtin_amdgcn_sched_barrier(0);
      if (q < 2) {
        load_a(a0, bufc, KY, 2 * q + 2);
        produce(2 * q + 1);
        mfma12(a1, 2 * q + 1);
        interleave();
      } else if constexpr (KY < 2) {
        load_a(a0, bufc, KY + 1, 0);
        produce(5);
        mfma12(a1, 5);
        interleave();
      } else {
        __syncthreads();  // the next channel group is complete in bufn, bufc is free
        load_a(a0, bufn, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma12(a1, 5);
      }
      __builtin_amdgcn_sched_barrier(0);
      load_b(2 * q + 1);
    }
  };

  // ---- prologue ------------------------------------------------------------------------------------------------
  make_geo(blockIdx.x, gc, lc, rc);
  make_geo(blockIdx.x + G, gn, ln, rn);
  load_item0(raw0);
  load_item1(raw1);
  advance();  // channel group 0 loaded
  {
    int mp0, nt0;
    w4_decode(p, kocr_xcd_remap(blockIdx.x, total), nblk_n, mp0, nt0);
    const unsigned short* w0 = w_tile(nt0);
#pragma unroll
    for (int xi = 0; xi < 6; ++xi)
#pragma unroll
      for (int s = 0; s < 3; ++s) bw[xi][s] = *reinterpret_cast<const bf8*>(w0 + (size_t)(xi * 3 + s) * 64 * 8);
  }
#pragma unroll
  for (int xi = 0; xi < 6; ++xi) {
    produce4(raw0, As, xi, 0);
    produce_item1(As, xi);
  }
  load_item0(raw0);
  load_item1(raw1);
  advance();  // channel group 1 loaded
  __syncthreads();
  load_a(a0, As, 0, 0);

  for (int L = blockIdx.x; L < total; L += G) {
    int mp, nt, mp_n, nt_n;
    w4_decode(p, kocr_xcd_remap(L, total), nblk_n, mp, nt);
    w4_decode(p, kocr_xcd_remap(L + G < total ? L + G : L, total), nblk_n, mp_n, nt_n);
    const unsigned short* w_ptr = w_tile(nt);
    const unsigned short* w_after = w_tile(nt_n);
    auto w_at = [&](int s) { return s < ns ? w_ptr + (size_t)s * w_step : w_after; };
#pragma unroll
    for (int x = 0; x < 6; ++x)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][m][r] = 0.f;
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (int cg = 0; cg < ncg; cg += 2) {
      // even channel group: consume buffer 0, produce the odd one into buffer 1 (items refilled with channel group + 2)
      step(std::integral_constant<int, 0>{}, As, As + BUF_R, w_at(3 * cg + 1));
      load_item0(raw0);
      step(std::integral_constant<int, 1>{}, As, As + BUF_R, w_at(3 * cg + 2));
      load_item1(raw1);
      advance();
      step(std::integral_constant<int, 2>{}, As, As + BUF_R, w_at(3 * cg + 3));
      // odd channel group: consume buffer 1, produce the next even one (possibly the next tile's first) into buffer 0
      step(std::integral_constant<int, 0>{}, As + BUF_R, As, w_at(3 * cg + 4));
      load_item0(raw0);
      step(std::integral_constant<int, 1>{}, As + BUF_R, As, w_at(3 * cg + 5));
      load_item1(raw1);
      advance();
      step(std::integral_constant<int, 2>{}, As + BUF_R, As, w_at(3 * cg + 6));
    }
    gc = gn;
    lc = ln;
    rc = rn;
    make_geo(L + 2 * G, gn, ln, rn);
    ld_next = false;

    // ---- epilogue --------------------------------------------------------------------------------------------------
    {
      const int n = (nt * 4 + wn) * 32 + l31;
      const int nc = n < p.Cout ? n : p.Cout - 1;
      const float pa = p.pre_a[nc], pb = p.pre_b[nc];
      const bool has_post = p.post_a != nullptr;
      const float qa = has_post ? p.post_a[nc] : 1.f, qb = has_post ? p.post_b[nc] : 0.f;
      const bool live = n < p.Cout;
      // branch-free epilogue arithmetic: ReLU as a max with 0 or -inf; the CRNN's post-ReLU BatchNorm affine (has_post) runs
      // as its own pass under a wave-uniform branch instead of a per-element select
      const float lo = p.relu ? 0.f : -INFINITY;
      auto act = [&](float v) { return fmaxf(v * pa + pb, lo); };
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float m0 = acc[0][m][r], m1 = acc[1][m][r], m2 = acc[2][m][r], m3 = acc[3][m][r], m4 = acc[4][m][r],
                      m5 = acc[5][m][r];
          const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
          acc[0][m][r] = act((m0 + s12) + s34);
          acc[1][m][r] = act(W4_A * d12 + W4_B * d34);
          acc[2][m][r] = act(W4_A2 * s12 + W4_B2 * s34);
          acc[3][m][r] = act((W4_A3 * d12 + W4_B3 * d34) + m5);
        }
      if (has_post) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][m][r] = acc[j][m][r] * qa + qb;
      }
      // the output pixel stride is made opaque per tile: hoisted out of the persistent loop the 128 store offsets would be
      // kept in (spilled) scalar registers and fetched back with one v_readlane per store
      int ocs4 = p.out_cs * 4;
      asm volatile("" : "+s"(ocs4));
      int pcs4 = p.pool_cs * 4;
      asm volatile("" : "+s"(pcs4));
      if (p.amax_out || p.amax_pool) {
        float mx = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fabsf(acc[j][m][r]));
        mx = live ? mx : 0.f;
        int uy, ux;
        const unsigned nimg = w4_fdiv((unsigned)tile_org(mp, uy, ux), p.dv_hw);  // per-image slots: the tile lies inside one image
        if (p.amax_out) kocr_amax_update(p.amax_out + nimg, mx);
        if (p.amax_pool) kocr_amax_update(p.amax_pool + nimg, mx);
      }
      int ty0, tx0;
      const long tpm = tile_org(mp, ty0, tx0);
      if constexpr (GEO == 2) {
        // M-tile m = rows 4 m .. 4 m + 3 x 8 quads: accumulator register r of lane half l5 is row r >> 2, quad (r & 3) + 4 l5
        const __amdgpu_buffer_rsrc_t ro = w4_rsrc(p.out + (tpm * p.out_cs + p.out_co), 0x7FFFFFFFu);
        const unsigned vo = live ? (unsigned)((16 * l5 * p.out_cs + n) * 4) : OOB;  // 4 quads = 16 px per l5
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int px = (4 * m + (r >> 2)) * p.W + 4 * (r & 3);
#pragma unroll
            for (int j = 0; j < 4; ++j)
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[j][m][r]), ro, vo, (px + j) * ocs4, 0);
          }
      } else {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          // M-tile m: 2 rows x 64 columns, two rows below M-tile 0
          const long pm = tpm + (long)2 * m * p.W;
          const int x0 = tx0;
          if (!POOL || p.write_full) {
            const __amdgpu_buffer_rsrc_t ro = w4_rsrc(p.out + (pm * p.out_cs + p.out_co), 0x7FFFFFFFu);
            const unsigned vo = live ? (unsigned)((16 * l5 * p.out_cs + n) * 4) : OOB;  // 4 quads = 16 px per l5
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
          if constexpr (POOL) {
            // 2x2 max: rows y (r) and y+1 (r+8), columns (0,1) and (2,3) of the quad
            const long pp0 = ((pm - x0) >> 2) + (x0 >> 1);  // (nimg H/2 + y0/2) W/2 + x0/2: H, W even
            const __amdgpu_buffer_rsrc_t rp = w4_rsrc(p.pool_out + (pp0 * p.pool_cs + p.pool_co), 0x7FFFFFFFu);
            const unsigned vp = live ? (unsigned)((8 * l5 * p.pool_cs + n) * 4) : OOB;  // 4 quads = 8 pooled px per l5
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              const int pq = 2 * ((r & 3) + 8 * (r >> 2));
              const float v0 = fmaxf(fmaxf(acc[0][m][r], acc[1][m][r]), fmaxf(acc[0][m][r + 8], acc[1][m][r + 8]));
              const float v1 = fmaxf(fmaxf(acc[2][m][r], acc[3][m][r]), fmaxf(acc[2][m][r + 8], acc[3][m][r + 8]));
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v0), rp, vp, pq * pcs4, 0);
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v1), rp, vp, (pq + 1) * pcs4, 0);
            }
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
int prepare_w43(kocr_ctx* ctx, ConvLayer& L, const float* w, bool w_is_oihw) {
  // any dilation: the taps' algebra is the same.  32 < Cout <= 64: the 64-cout arrangement (dilation 1 only)
  if (L.KH != 3 || L.KW != 3 || L.Cin % 32 != 0 || L.Cout <= 32 || (L.Cout <= 64 && L.dil != 1)) return KOCR_OK;
  const int Cin = L.Cin, Cout = L.Cout;
  const int cp = Cout <= 64 ? 64 : (Cout + 127) / 128 * 128;
  const int nt32 = cp / 32;
  std::vector<unsigned short> u((size_t)(Cin / 16) * 3 * nt32 * 18 * 64 * 8, 0);
  for (int c = 0; c < Cin; ++c)
    for (int ky = 0; ky < 3; ++ky)
      for (int o = 0; o < Cout; ++o) {
        double g[3];
        for (int kx = 0; kx < 3; ++kx)
          g[kx] = w_is_oihw ? w[(((size_t)o * Cin + c) * 3 + ky) * 3 + kx] : w[(((size_t)ky * 3 + kx) * Cin + c) * Cout + o];
        // G g in float64, rounded once to fp32
        const double pa = W4_PA, pb = W4_PB, a2 = pa * pa, b2 = pb * pb;
        const double na = 2.0 * a2 * (a2 - b2), nb = 2.0 * b2 * (b2 - a2);
        const float U[6] = {(float)(g[0] / (a2 * b2)),
                            (float)((g[0] + pa * g[1] + a2 * g[2]) / na),
                            (float)((g[0] - pa * g[1] + a2 * g[2]) / na),
                            (float)((g[0] + pb * g[1] + b2 * g[2]) / nb),
                            (float)((g[0] - pb * g[1] + b2 * g[2]) / nb),
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

// Round 5: the fp16 vertical- / row-reuse kernels take any H, W (their MODE 1, "ragged": masked gather and stores) and the
// recogniser's cell grids (MODE 2).  Which geometry a ragged image gets, or -1: `narrow` = the 64-cout row-reuse kernel
// (4 x 64 tiles only), else the vertical-reuse kernel on 4 x 64 or 8 x 32 tiles, whichever covers the image with fewer
// padding pixels (8 x 32 has no fused pooling).  A ragged grid is used when the image's width is not a multiple of 4 (the
// flattened-pixel arrangements then do not apply at all) or when the padding costs less than the flattened arrangement
// loses against vertical reuse (measured 20-30 %, plus the unfused pooling).
static bool w43_env_off(const char* name) {
  const char* e = getenv(name);
  return e && atoi(e) == 0;
}
static int w43_ragged_geo(const kocr_ctx* ctx, const ConvLayer& L, const Tensor& in, bool pool) {
  static const bool no_v = w43_env_off("KOCR_W43V"), no_rr = w43_env_off("KOCR_W43R"), no_rag = w43_env_off("KOCR_W43RAG");
  if (no_rag || ctx->split_mode == KOCR_SPLIT_BF16X3 || !ctx->sw.w43h || !L.d_w4h || L.dil != 1) return -1;
  if ((size_t)in.H * in.W * in.cs * 4 >= ((size_t)1 << 31)) return -1;
  const bool narrow = L.w4_cout_pad == 64;
  if (narrow ? no_rr : no_v) return -1;
  auto cover = [&](int rb, int tc) { return (double)((in.H + rb - 1) / rb * rb) * ((in.W + tc - 1) / tc * tc) / ((double)in.H * in.W); };
  const double c1 = cover(4, 64), c2 = cover(8, 32);
  const int geo = (narrow || pool || c1 <= c2 * 1.02) ? 1 : 2;
  const double c = geo == 1 ? c1 : c2;
  if (in.W % 4 != 0) return geo;                 // no flattened arrangement takes this width
  if ((size_t)in.H * in.W < 256) return -1;      // tiny images stay where they were (bf16x3 flattened tiles)
  return c <= 1.25 ? geo : -1;
}

// can layer L run on a cell grid (conv_w43vh_kernel MODE 2) in the context's current arithmetic?  KOCR_CELLS=0: never (the
// recogniser then keeps round 4's dense crop batch on the flattened-pixel kernel)
bool w43_cells_ok(const kocr_ctx* ctx, const ConvLayer& L) {
  static const bool off = w43_env_off("KOCR_W43"), no_v = w43_env_off("KOCR_W43V"), no_cells = w43_env_off("KOCR_CELLS");
  return !off && !no_v && !no_cells && ctx->split_mode != KOCR_SPLIT_BF16X3 && ctx->sw.w43h && L.d_w4h && L.w4_cout_pad > 64 &&
         L.dil == 1 && L.Cin % 32 == 0;
}
// ... and the width-padded layout (Tensor::Wv) on the flattened fp16 kernel?
bool w43_flat_h_ok(const kocr_ctx* ctx, const ConvLayer& L) {
  static const bool off = w43_env_off("KOCR_W43");
  return !off && ctx->split_mode != KOCR_SPLIT_BF16X3 && ctx->sw.w43h && L.d_w4h && L.w4_cout_pad > 64 && L.dil == 1 && L.Cin % 32 == 0;
}

bool w43_applicable(const kocr_ctx* ctx, const ConvLayer& L, const Tensor& in) {
  static const bool off = getenv("KOCR_W43") && atoi(getenv("KOCR_W43")) == 0;
  if (in.cellW) return w43_cells_ok(ctx, L);
  if (off || !L.d_w4 || in.cs % 4 != 0 || in.co % 4 != 0 || ((uintptr_t)in.p & 15) != 0 || L.Cin % 32 != 0 ||
      (size_t)in.pixels() >= ((size_t)1 << 29))
    return false;
  if (in.W % (4 * L.dil) == 0) return true;
  return w43_ragged_geo(ctx, L, in, false) > 0;  // a width the flattened arrangements do not take
}

template <int POOL, int DBG = 0, int DIL = 0>
static int w4_launch(kocr_ctx* ctx, W4Params& p) {
  static std::atomic<bool> attr_done[64];  // per device (one process may hold contexts on several GPUs); a race only repeats the call
  const int dev = ctx->device & 63;
  if (!attr_done[dev]) {
    KOCR_HIP(ctx, hipFuncSetAttribute((const void*)conv_w43_kernel<POOL, DBG, DIL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
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
  hipLaunchKernelGGL((conv_w43_kernel<POOL, DBG, DIL>), dim3(grid), dim3(256), LDS_BYTES, ctx->stream, p);
  KOCR_HIP(ctx, hipGetLastError());
  return KOCR_OK;
}

template <int POOL>
static int w4n_launch(kocr_ctx* ctx, W4Params& p) {
  constexpr int LDSN = 2 * LDS_BYTES;  // 2 x 72 KB
  static std::atomic<bool> attr_done[64];
  const int dev = ctx->device & 63;
  if (!attr_done[dev]) {
    KOCR_HIP(ctx, hipFuncSetAttribute((const void*)conv_w43n_kernel<POOL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSN));
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
  hipLaunchKernelGGL((conv_w43n_kernel<POOL>), dim3(grid), dim3(256), LDSN, ctx->stream, p);
  KOCR_HIP(ctx, hipGetLastError());
  return KOCR_OK;
}

template <int POOL, int GEO>
static int w4r_launch(kocr_ctx* ctx, W4Params& p) {
  // 2 x 72 KB, or 54 KB + the epilogue's 64 KB exchange area (which starts at the second 54 KB buffer)
  constexpr int LDSR = GEO ? 6 * 3 * 6 * 256 * 2 + 4 * 16 * 64 * 16 : 2 * LDS_BYTES;
  static std::atomic<bool> attr_done[64];
  const int dev = ctx->device & 63;
  if (!attr_done[dev]) {
    KOCR_HIP(ctx, hipFuncSetAttribute((const void*)conv_w43r_kernel<POOL, GEO>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSR));
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
  hipLaunchKernelGGL((conv_w43r_kernel<POOL, GEO>), dim3(grid), dim3(256), LDSR, ctx->stream, p);
  KOCR_HIP(ctx, hipGetLastError());
  return KOCR_OK;
}

template <int POOL, int GEO>
static int w4v_launch(kocr_ctx* ctx, W4Params& p) {
  constexpr int LDSV = GEO == 2 ? 2 * 6 * 3 * 10 * 128 * 2 : 2 * 6 * 3 * 6 * 256 * 2;  // 2 x 45 / 54 KB
  static std::atomic<bool> attr_done[64];
  const int dev = ctx->device & 63;
  if (!attr_done[dev]) {
    KOCR_HIP(ctx, hipFuncSetAttribute((const void*)conv_w43v_kernel<POOL, GEO>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSV));
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
  hipLaunchKernelGGL((conv_w43v_kernel<POOL, GEO>), dim3(grid), dim3(256), LDSV, ctx->stream, p);
  KOCR_HIP(ctx, hipGetLastError());
  return KOCR_OK;
}

int launch_conv_w43(kocr_ctx* ctx, const ConvLayer& L, const Tensor& in, const Tensor& out, const Tensor* pool, bool need_full) {
  // round 5: cell grids (the recogniser's crop batch) and ragged images on the fp16 vertical- / row-reuse kernels
  const bool cells = in.cellW > 0;
  if (cells || out.cellW || (pool && pool->cellW)) {
    // tiles of 4 x 64 where a cell is at least 64 columns wide, else 8 x 32 (a tile may touch two cells, not three)
    const bool cgeo_ok = in.cellW >= 64 ? (in.H % 4 == 0 && in.W % 64 == 0) : (in.cellW >= 32 && in.H % 8 == 0 && in.W % 32 == 0 && !pool);
    const bool ok = cells && out.cellW == in.cellW && out.cellWv == in.cellWv && cgeo_ok && in.cellW % 4 == 0 &&
                    in.W % in.cellW == 0 && in.amax && L.dil == 1 && L.w4_cout_pad > 64 && L.d_w4h && ctx->sw.w43h &&
                    ctx->split_mode != KOCR_SPLIT_BF16X3 && (size_t)in.H * in.W * in.cs * 4 < ((size_t)1 << 31) &&
                    (!pool || (pool->cellW * 2 == in.cellW && pool->cellWv * 2 == in.cellWv && in.cellWv % 2 == 0 && pool->H * 2 == in.H &&
                               pool->W * 2 == in.W));
    if (!ok) KOCR_FAIL(ctx, KOCR_EINVAL, "conv " + L.name + ": a cell-grid tensor (Tensor::cellW) needs the fp16 vertical-reuse F(4,3) kernel");
  }
  const int rag_geo = cells ? -1 : w43_ragged_geo(ctx, L, in, pool != nullptr);
  const bool exact_fuse = pool && L.dil == 1 && in.H % 2 == 0 && in.W % 64 == 0;
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
  p.dil = L.dil;
  p.qpr = in.W / 4;
  p.pool_out = nullptr;
  p.pool_cs = p.pool_co = p.write_full = p.tiles_per_row = 0;
  p.amax_out = p.amax_pool = nullptr;  // set below for the arrangements that maintain the per-image slots themselves
  p.amax_in = nullptr;
  p.Wv = (out.Wv && out.Wv < out.W) ? out.Wv : 0;
  w4_div_magic((unsigned)in.W, p.dv_w);
  const bool narrow = L.w4_cout_pad == 64;  // 64-cout arrangement: 4 M-tiles x 64 couts per tile
  // ... or, when the image tiles as 2 rows x 128 columns, the row-reuse arrangement (2 M-tiles of 2 rows x 64 columns)
  static const bool no_rr = getenv("KOCR_W43R") && atoi(getenv("KOCR_W43R")) == 0;
  const bool r_ok = narrow && !no_rr && L.dil == 1 && (!pool || exact_fuse) && (size_t)in.H * in.W * in.cs * 4 < ((size_t)1 << 31);
  const bool rgeo1_ok = r_ok && in.H % 4 == 0 && in.W % 64 == 0, rgeo0_ok = r_ok && in.H % 2 == 0 && in.W % 128 == 0;
  int rgeo = rgeo1_ok ? 1 : rgeo0_ok ? 0 : -1;  // 4 x 64 tiles where the image tiles that way, else 2 x 128
  // mode of the fp16 kernels: 0 = the image tiles exactly, 1 = ragged (masked gather / stores), 2 = cell grid
  int mode = cells ? 2 : 0;
  if (rgeo < 0 && narrow && rag_geo == 1) {
    rgeo = 1;
    mode = 1;
  }
  const bool rowreuse = rgeo >= 0;
  // Cout > 64 on the same image geometry: the vertical-reuse arrangement (conv_w43v_kernel)
  static const bool no_v = getenv("KOCR_W43V") && atoi(getenv("KOCR_W43V")) == 0;
  // geometries of conv_w43v_kernel: GEO 1 = 4 rows x 64 columns (H % 4 == 0, W % 64 == 0), GEO 2 = 8 rows x 32 columns
  // (H % 8 == 0, W % 32 == 0, no fused pooling)
  const bool v_ok = !narrow && !no_v && L.dil == 1 && (!pool || exact_fuse) && (size_t)in.H * in.W * in.cs * 4 < ((size_t)1 << 31);
  // 4 rows x 64 columns where the image tiles that way, else 8 rows x 32 columns (the 96-wide layers); anything else (e.g.
  // H % 4 != 0) stays on conv_w43_kernel
  const bool geo1_ok = v_ok && in.H % 4 == 0 && in.W % 64 == 0;
  const bool geo2_ok = v_ok && !pool && in.H % 8 == 0 && in.W % 32 == 0;
  int vgeo = geo1_ok ? 1 : geo2_ok ? 2 : -1;
  if (cells) vgeo = in.cellW >= 64 ? 1 : 2;
  if (vgeo < 0 && !narrow && rag_geo > 0) {
    vgeo = rag_geo;
    mode = 1;
  }
  const bool vreuse = vgeo >= 0;
  // the fused 2x2 pooling: images that tile exactly as 2 rows x 64 columns (any arrangement), or the 4 x 64 ragged / cell grids
  const bool fuse = exact_fuse || (pool && mode != 0 && (vreuse ? vgeo == 1 : rgeo == 1));
  if (fuse) {
    if (pool->H != in.H / 2 || pool->W != in.W / 2 || pool->N != in.N) KOCR_FAIL(ctx, KOCR_EINVAL, "conv " + L.name + ": bad pooled shape");
    p.pool_out = pool->p;
    p.pool_cs = pool->cs;
    p.pool_co = pool->co;
    p.write_full = need_full ? 1 : 0;
    p.tiles_per_row = in.W / 64;
  }
  if (!out.p && !(fuse && !need_full)) KOCR_FAIL(ctx, KOCR_EINVAL, "conv " + L.name + ": no output buffer");
  if ((rowreuse || vreuse) && !fuse) p.tiles_per_row = in.W / 64;  // the 2-row x 64-column M-tile geometry without the pooling
  p.n_mpairs = (rowreuse || vreuse) ? p.total_mtiles / 2 : narrow ? (p.total_mtiles + 3) / 4 : (p.total_mtiles + 1) / 2;
  if (vgeo == 2) p.tiles_per_row = in.W / 32;
  p.rq_per_img = 0;
  p.cellW = p.cellWv = p.cells_per_row = 0;
  if (mode != 0) {  // the ragged / cell tile grid: ceil(H / 4) x ceil(W / 64) (ceil(H / 8) x ceil(W / 32)) tiles per image
    const int rb = (vreuse && vgeo == 2) ? 8 : 4, tc = (vreuse && vgeo == 2) ? 32 : 64;
    p.tiles_per_row = (in.W + tc - 1) / tc;
    p.rq_per_img = (in.H + rb - 1) / rb;
    p.n_mpairs = in.N * p.rq_per_img * p.tiles_per_row;
    p.total_mtiles = 2 * p.n_mpairs;
    if (cells) {
      p.cellW = in.cellW;
      p.cellWv = in.cellWv;
      p.cells_per_row = in.W / in.cellW;
    }
  }
  w4_div_magic((unsigned)(p.qpr > 0 ? p.qpr : 1), p.dv_qpr);
  w4_div_magic((unsigned)L.dil, p.dv_dil);
  w4_div_magic((unsigned)in.H, p.dv_h);
  w4_div_magic((unsigned)(p.rq_per_img ? p.rq_per_img : 1), p.dv_rq);
  w4_div_magic((unsigned)(p.cellW ? p.cellW : 1), p.dv_wc);
  p.total_tiles = p.n_mpairs * (p.Cout_pad / (narrow ? 64 : 128));
  w4_div_magic((unsigned)p.tiles_per_row, p.dv_tpr);
  w4_div_magic((unsigned)(vgeo == 2 ? in.H / 8 : in.H / 2), p.dv_hh);
  w4_div_magic((unsigned)p.n_mpairs, p.dv_mp);
  w4_div_magic((unsigned)(in.H * in.W), p.dv_hw);
  // Per-image max-|x| slots (Tensor::amax): the one-image tiles of the row-reuse / vertical-reuse arrangements maintain
  // them in their epilogue; the flattened-pixel arrangements leave them to a reduction pass after the launch.
  // fp16 arithmetic (conv_w43h.hip) where an fp16 kernel exists for the arrangement; else the exact bf16x3 kernels
  const int pieces = ctx->split_mode == KOCR_SPLIT_F16X2 ? 2 : ctx->split_mode == KOCR_SPLIT_F16X1 ? 1 : 0;
  const bool no_h = !ctx->sw.w43h;
  // ... the flattened-pixel arrangement too (no fused pooling, dilation 1; a 256-pixel tile must not span three images)
  // (round 5: dilated layers too -- the comb tiles of conv_w43fh_kernel<.., DIL = 1>; KOCR_W43DILH=0: bf16x3 as before)
  static const bool no_dilh = w43_env_off("KOCR_W43DILH");
  const bool flat_h = pieces && !no_h && L.d_w4h && !narrow && !vreuse && !rowreuse && (L.dil == 1 || !no_dilh) && !fuse &&
                      (size_t)in.H * in.W >= 256;
  const bool use_h = (pieces && !no_h && L.d_w4h && (vreuse || (rowreuse && rgeo == 1))) || flat_h;
  if (mode != 0 && !use_h) KOCR_FAIL(ctx, KOCR_EINVAL, "conv " + L.name + ": ragged / cell grids exist in the fp16 kernels only");
  if (p.Wv && !flat_h)  // only the flattened fp16 kernel writes the zero columns of a width-padded output
    KOCR_FAIL(ctx, KOCR_EINVAL, "conv " + L.name + ": a width-padded output (Tensor::Wv) needs the flattened fp16 F(4,3) kernel");
  const bool tracks = rowreuse || vreuse || flat_h;
  if (tracks) {
    p.amax_out = (!fuse || need_full) ? out.amax : nullptr;
    p.amax_pool = fuse ? pool->amax : nullptr;
  }
  if (use_h) {
    const unsigned* slots = in.amax;
    if (!slots) {  // the producer did not track: reduce the input once, per image
      unsigned* tmp = ctx->amax_slots(in.N);
      if (!tmp) KOCR_FAIL(ctx, KOCR_ECAPACITY, "conv " + L.name + ": out of max-|x| slots");
      KOCR_TRY(launch_absmax(ctx, in, tmp));
      slots = tmp;
    }
    p.amax_in = slots;
    if (ctx->range_on) KOCR_TRY(launch_range_stats(ctx, L.name, in, slots, 12));  // W4H_TOP (conv_w43h.hip)
    p.wgt = L.d_w4h;
    p.pre_a = L.d_pre_a_h;
  }
  w4_div_magic((unsigned)(p.Cout_pad / (narrow ? 64 : 128)), p.dv_nb);
  // Tile order.  The split weights of one cout block are Cin * 3 * 128 * 36 B; with every cout block of a deep layer in
  // flight on an XCD they overflow its 4 MB L2 and are re-streamed from the Infinity Cache by every round of tiles.
  // Pixel-tile-fastest order keeps ONE cout block per XCD at a time (measured +4 % on 512 -> 512, neutral below).
  p.m_fastest = (size_t)L.Cin * L.w4_cout_pad * 3 * 36 > ((size_t)6 << 20) ? 1 : 0;
  static const bool per_layer = getenv("KOCR_PROF_LAYERS") != nullptr;
  char nm[64];
  if (per_layer)
    snprintf(nm, sizeof nm, "conv_w4%s%s_%s%s%s:%s", use_h ? ((pieces == 2 || mode) ? "h" : "q") : "", vreuse ? (vgeo == 2 ? "t" : "v") : (flat_h ? "f" : use_h ? "r" : "s"), rowreuse ? "256x64" : narrow ? "512x64" : "256x128", fuse ? "p" : (L.dil != 1 ? "d" : ""), mode == 1 ? "g" : mode == 2 ? "c" : "", L.name.c_str());
  else
    snprintf(nm, sizeof nm, "conv_w4%s%s_%s%s%s", use_h ? ((pieces == 2 || mode) ? "h" : "q") : "", vreuse ? (vgeo == 2 ? "t" : "v") : (flat_h ? "f" : use_h ? "r" : "s"), rowreuse ? "256x64" : narrow ? "512x64" : "256x128", fuse ? "_pool" : (L.dil != 1 ? "_dil" : ""), mode == 1 ? "_rag" : mode == 2 ? "_cells" : "");
  {
    char fam[48];
    snprintf(fam, sizeof fam, "w4%s%s%s%s", use_h ? ((pieces == 2 || mode) ? "h" : "q") : "", vreuse ? (vgeo == 2 ? "t" : "v") : (flat_h ? "f" : use_h ? "r" : "s"),
             fuse ? "_pool" : (L.dil != 1 ? "_dil" : ""), mode == 1 ? "_rag" : mode == 2 ? "_cells" : "");
    kocr_note_dispatch(fam, L, in);
  }
  // algorithmic (direct-convolution) FLOPs and bytes: of the crops' own pixels in a cell grid, not of the gutters
  const double Malg = cells ? (double)in.N * in.cells() * (in.H - 1) * in.cellWv : (double)M;
  const double flops = 2.0 * Malg * L.Kreal * L.Cout;
  const double bytes = 4.0 * (Malg * L.Cin + Malg * L.Cout * ((fuse && !need_full) ? 0.25 : 1.0) + (double)L.Kreal * L.Cout);
  {
    ProfScope ps(ctx, nm, flops, bytes);
#ifdef KOCR_DEV_SWITCHES
    static const int dbg = getenv("KOCR_W43_DBG") ? atoi(getenv("KOCR_W43_DBG")) : 0;
    if (dbg && !fuse) {
      switch (dbg) {
        case 1: return w4_launch<0, 1>(ctx, p);
        case 2: return w4_launch<0, 2>(ctx, p);
        case 3: return w4_launch<0, 3>(ctx, p);
        case 4: return w4_launch<0, 4>(ctx, p);
        case 5: return w4_launch<0, 5>(ctx, p);
        case 6: return w4_launch<0, 6>(ctx, p);
        case 7: return w4_launch<0, 7>(ctx, p);
        case 8: return w4_launch<0, 8>(ctx, p);
        case 9: return w4_launch<0, 9>(ctx, p);
        case 10: return w4_launch<0, 10>(ctx, p);
        case 11: return w4_launch<0, 11>(ctx, p);
        case 12: return w4_launch<0, 12>(ctx, p);
        case 13: return w4_launch<0, 13>(ctx, p);
        case 14: return w4_launch<0, 14>(ctx, p);
        case 15: return w4_launch<0, 15>(ctx, p);
        default: break;
      }
    }
#endif
    if (use_h && vreuse) {
      KOCR_TRY(launch_w43vh(ctx, p, fuse, vgeo, pieces, mode));
    } else if (flat_h) {
      KOCR_TRY(launch_w43fh(ctx, p, pieces));
    } else if (use_h) {
      KOCR_TRY(launch_w43rh(ctx, p, fuse, pieces, mode));
    } else if (vreuse) {
      if (vgeo == 2) {
        KOCR_TRY((w4v_launch<0, 2>(ctx, p)));
      } else {
        if (fuse)
          KOCR_TRY((w4v_launch<1, 1>(ctx, p)));
        else
          KOCR_TRY((w4v_launch<0, 1>(ctx, p)));
      }
    } else if (rowreuse) {
      if (rgeo == 1) {
        if (fuse)
          KOCR_TRY((w4r_launch<1, 1>(ctx, p)));
        else
          KOCR_TRY((w4r_launch<0, 1>(ctx, p)));
      } else {
        if (fuse)
          KOCR_TRY((w4r_launch<1, 0>(ctx, p)));
        else
          KOCR_TRY((w4r_launch<0, 0>(ctx, p)));
      }
    } else if (narrow) {
      if (fuse)
        KOCR_TRY(w4n_launch<1>(ctx, p));
      else
        KOCR_TRY(w4n_launch<0>(ctx, p));
    } else if (L.dil != 1)
      KOCR_TRY((w4_launch<0, 0, 1>(ctx, p)));
    else if (fuse)
      KOCR_TRY(w4_launch<1>(ctx, p));
    else
      KOCR_TRY(w4_launch<0>(ctx, p));
  }
  if (!tracks) {  // flattened-pixel arrangements: per-image max |x| of what was written, for an fp16 consumer
    if (out.amax && (!fuse || need_full)) KOCR_TRY(launch_absmax(ctx, out, out.amax));
    if (fuse && pool->amax) KOCR_TRY(launch_absmax(ctx, *pool, pool->amax));
  }
  if (pool && !fuse) return launch_maxpool2x2(ctx, out, *pool);
  return KOCR_OK;
}
