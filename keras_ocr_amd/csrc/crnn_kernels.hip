// crnn_kernels.hip — the non-GEMM kernels of the CRNN recogniser (recognition.py:187-333).
//
//   crnn_input_kernel   Permute((2,1,3)) + flip axis 2 (recognition.py:215-216)
//   stn_sample_kernel   _transform bilinear sampler (recognition.py:73-166)
//   lstm_kernel         keras LSTM recurrence on the matrix cores (recognition.py:292-319):
//                       one workgroup = 32 crops x one direction for all 50 steps; h lives in LDS,
//                       c in registers, h@U is 32x512x128 per step on v_mfma_f32_32x32x2_f32 with the
//                       four gates of a hidden unit in the same lane (so the cell update is
//                       register-only); x@W+b is precomputed by the conv/GEMM kernel.
//   ctc_kernel          fc_12 softmax + CTCDecoder (recognition.py:169-184, 322-328): one wave per
//                       crop, lanes = classes, per-step argmax as a wavefront reduction, repeat
//                       merge + blank removal by lane 0.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// in: [M][Hc=31][Wc=200] -> out: [M][Wc][Hc] with out[m][w][j] = in[m][Hc-1-j][w]
__global__ void crnn_input_kernel(const float* __restrict__ in, float* __restrict__ out, int M, int Hc, int Wc) {
  const size_t total = (size_t)M * Hc * Wc;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = i % Hc;
    const size_t t = i / Hc;
    const int w = t % Wc;
    const size_t m = t / Wc;
    out[i] = in[(m * Hc + (Hc - 1 - j)) * Wc + w];
  }
}

// in: [M][Hn][Wn][C] (natural crop orientation) -> out: [M][Wn][Hn][C] with out[m][w][j] = in[m][Hn-1-j][w]
// (the Permute((2,1,3)) + flip of recognition.py:215-216 applied after the conv stack instead of before)
// Wp = width pitch of `in` in pixels (>= Wn: a width-padded tensor, Tensor::Wv)
__global__ void crnn_to_keras_kernel(const float* __restrict__ in, float* __restrict__ out, int M, int Hn, int Wn,
                                     int C4, int Wp) {
  const size_t total = (size_t)M * Hn * Wn * C4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = i % C4;
    size_t t = i / C4;
    const int j = t % Hn;
    t /= Hn;
    const int w = t % Wn;
    const size_t m = t / Wn;
    reinterpret_cast<float4*>(out)[i] =
        reinterpret_cast<const float4*>(in)[((m * Hn + (Hn - 1 - j)) * Wp + w) * C4 + c4];
  }
}

// ---- the recogniser's crop batch as a CELL GRID (Tensor::cellW; round 5) ------------------------------------------------------
// conv_1 (recognition.py:217: 1 -> 64 channels, 3x3 'same', bias, ReLU; natural orientation, see crnn.cpp) straight from
// the crop batch [M][31][200] into the level-1 cell grid out[R][32][cn * 208][64]: cell (n, j) holds crop n * cn + j in its
// rows 1 .. 31 and columns 0 .. 199; row 0, columns 200 .. 207 and the cells behind crop M - 1 are written as zeros.  K = 9:
// plain fp32 FMAs (9 per output; the layer is bound by its 256 bytes of output per pixel), the same fma chain order for
// every output.  Also maintains the per-cell max-|x| slots (exact maximum, so a crop's scale does not depend on its cell).
// grid = (cells, 32 cell rows); thread = (column slot t >> 4 of 16, channel quad t & 15), 13 column groups per row.
__global__ __launch_bounds__(256) void crnn_conv1_cells_kernel(const float* __restrict__ crops, const float* __restrict__ w,
                                                               const float* __restrict__ pre_a, const float* __restrict__ pre_b,
                                                               float* __restrict__ out, unsigned* __restrict__ amax, int M, int cn,
                                                               int Hc, int Wc, int cellH, int cellW, int cout_pad) {
  const int cell = blockIdx.x, crow = blockIdx.y;
  const int n = cell / cn, j = cell - n * cn;
  const int t = threadIdx.x, cq = t & 15, xs = t >> 4;
  float wk[9][4], pa[4], pb[4];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int c = 0; c < 4; ++c) wk[k][c] = w[(size_t)k * cout_pad + cq * 4 + c];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    pa[c] = pre_a[cq * 4 + c];
    pb[c] = pre_b[cq * 4 + c];
  }
  const int y = crow - 1;  // crop row
  const bool live = cell < M && y >= 0 && y < Hc;
  const float* src = crops + (size_t)(cell < M ? cell : 0) * Hc * Wc;
  float4* dst = reinterpret_cast<float4*>(out + (((size_t)n * cellH + crow) * ((size_t)cn * cellW) + (size_t)j * cellW) * 64) + cq;
  float mx = 0.f;
  for (int x = xs; x < cellW; x += 16) {
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live && x < Wc) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int yy = y + ky - 1, xx = x + kx - 1;
          const float v = ((unsigned)yy < (unsigned)Hc && (unsigned)xx < (unsigned)Wc) ? src[yy * Wc + xx] : 0.f;
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[c] = fmaf(v, wk[ky * 3 + kx][c], acc[c]);
        }
      o.x = fmaxf(fmaf(acc[0], pa[0], pb[0]), 0.f);
      o.y = fmaxf(fmaf(acc[1], pa[1], pb[1]), 0.f);
      o.z = fmaxf(fmaf(acc[2], pa[2], pb[2]), 0.f);
      o.w = fmaxf(fmaf(acc[3], pa[3], pb[3]), 0.f);
      mx = fmaxf(mx, fmaxf(fmaxf(o.x, o.y), fmaxf(o.z, o.w)));
    }
    dst[(size_t)x * 16] = o;
  }
  if (amax) {
    const unsigned bits = kocr_wave_max_bits(mx);
    if (bits != 0 && (t & 63) == 0) atomicMax(amax + cell, bits);
  }
}

// level-3 cell grid in[R][cellH][cn * cellW][C] (crop rows at cell rows 1 .. Hn, natural orientation) -> Keras layout
// out[M][Wn][Hn][C] with out[m][w][j] = crop_m[Hn - 1 - j][w] (the Permute((2,1,3)) + flip of recognition.py:215-216)
__global__ void crnn_cells_to_keras_kernel(const float* __restrict__ in, float* __restrict__ out, int M, int Hn, int Wn, int C4,
                                           int cn, int cellH, int cellW) {
  const size_t total = (size_t)M * Hn * Wn * C4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = i % C4;
    size_t t = i / C4;
    const int j = t % Hn;
    t /= Hn;
    const int w = t % Wn;
    const size_t m = t / Wn;
    const size_t n = m / cn, jc = m - n * cn;
    reinterpret_cast<float4*>(out)[i] =
        reinterpret_cast<const float4*>(in)[((n * cellH + (size_t)(Hn - 1 - j) + 1) * ((size_t)cn * cellW) + jc * cellW + w) * C4 + c4];
  }
}

// x: [M][H][W][C], theta: [M][6] -> out [M][H][W][C]
__global__ void stn_sample_kernel(const float* __restrict__ x, const float* __restrict__ theta, float* __restrict__ out,
                                  int M, int H, int W, int C) {
  const int pix = blockIdx.x;  // m*H*W + oy*W + ox
  const int ox = pix % W;
  const int oy = (pix / W) % H;
  const int m = pix / (W * H);
  const float* th = theta + (size_t)m * 6;
  // tf.linspace(-1, 1, n): start + i*delta, last element exactly 1
  const float xt = (ox == W - 1) ? 1.f : -1.f + (float)ox * (2.f / (float)(W - 1));
  const float yt = (oy == H - 1) ? 1.f : -1.f + (float)oy * (2.f / (float)(H - 1));
  const float xs = (th[0] * xt + th[1] * yt) + th[2];
  const float ys = (th[3] * xt + th[4] * yt) + th[5];
  const float fx = 0.5f * (xs + 1.0f) * (float)W;   // scaled by W, not W-1 (recognition.py:109)
  const float fy = 0.5f * (ys + 1.0f) * (float)H;
  int x0 = (int)floorf(fx), y0 = (int)floorf(fy);
  int x1 = x0 + 1, y1 = y0 + 1;
  x0 = min(max(x0, 0), W - 1);
  x1 = min(max(x1, 0), W - 1);
  y0 = min(max(y0, 0), H - 1);
  y1 = min(max(y1, 0), H - 1);
  // weights from the CLIPPED corners (recognition.py:144-152)
  const float wa = ((float)x1 - fx) * ((float)y1 - fy);
  const float wb = ((float)x1 - fx) * (fy - (float)y0);
  const float wc = (fx - (float)x0) * ((float)y1 - fy);
  const float wd = (fx - (float)x0) * (fy - (float)y0);
  const float* base = x + (size_t)m * H * W * C;
  const float* pa = base + ((size_t)y0 * W + x0) * C;
  const float* pb = base + ((size_t)y1 * W + x0) * C;
  const float* pc = base + ((size_t)y0 * W + x1) * C;
  const float* pd = base + ((size_t)y1 * W + x1) * C;
  float* o = out + (size_t)pix * C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) o[c] = ((wa * pa[c] + wb * pb[c]) + wc * pc[c]) + wd * pd[c];
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

// xp: [M][T][2*4*U] (dir-major, gate order i,f,c,o); Uf/Ub: [U][4U]; out: [M][T][2U] in PROCESSING order
template <int UNITS>
__global__ __launch_bounds__(256) void lstm_kernel(const float* __restrict__ xp, const float* __restrict__ Uf,
                                                   const float* __restrict__ Ub, float* __restrict__ out, int M,
                                                   int T) {
  static_assert(UNITS == 128, "4 waves x 32 hidden units");
  constexpr int LD = UNITS + 1;
  __shared__ float hs[32][LD];
  const int dir = blockIdx.y;
  const int m0 = blockIdx.x * 32;
  const float* Ur = dir ? Ub : Uf;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lk = lane >> 5;
  const int j = wave * 32 + lr;  // hidden unit of this lane
  for (int i = tid; i < 32 * LD; i += 256) (&hs[0][0])[i] = 0.f;
  float c[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    const int tin = dir ? (T - 1 - t) : t;
    f32x16 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        acc[g][r] = m < M ? xp[((size_t)m * T + tin) * (8 * UNITS) + dir * 4 * UNITS + g * UNITS + j] : 0.f;
      }
#pragma unroll 4
    for (int kp = 0; kp < UNITS / 2; ++kp) {
      const int k = 2 * kp + lk;
      const float a = hs[lr][k];
      const float* ur = Ur + (size_t)k * (4 * UNITS) + j;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, ur[g * UNITS], acc[g], 0, 0, 0);
    }
    __syncthreads();  // every wave has finished reading h(t-1)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
      const float ig = sigmoidf_(acc[0][r]), fg = sigmoidf_(acc[1][r]);
      const float gg = tanhf(acc[2][r]), og = sigmoidf_(acc[3][r]);
      c[r] = fg * c[r] + ig * gg;
      const float h = og * tanhf(c[r]);
      hs[row][j] = h;
      const int m = m0 + row;
      if (m < M) out[((size_t)m * T + t) * (2 * UNITS) + dir * UNITS + j] = h;
    }
    __syncthreads();
  }
}

// Round 3: the recurrence is a chain of T = 50 dependent steps, so its time is (time of one step) x 50 whatever the
// batch: lstm16_kernel shortens the step.  One workgroup = 16 crops x one direction (twice as many workgroups, each with
// half the matrix work per step: 32 x 8 v_mfma_f32_16x16x4_f32 per wave), and the wave's slice of the recurrent kernel
// U (128 k x 32 hidden units x 4 gates = 256 values per lane) stays in REGISTERS for all 50 steps instead of being
// re-fetched from L2 every step (256 loads per wave and step before).  Same arithmetic: fp32 MFMA, gates i, f, c, o of a
// hidden unit in one lane, cell state in registers, h through LDS.
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int UNITS>
__global__ __launch_bounds__(256) void lstm16_kernel(const float* __restrict__ xp, const float* __restrict__ Uf,
                                                     const float* __restrict__ Ub, float* __restrict__ out, int M, int T) {
  static_assert(UNITS == 128, "4 waves x 32 hidden units");
  constexpr int LD = UNITS + 1;
  __shared__ float hs[16][LD];
  const int dir = blockIdx.y;
  const int m0 = blockIdx.x * 16;
  const float* Ur = dir ? Ub : Uf;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  // N-tile (g, uh): gate g, hidden units 32 wave + 16 uh + [0, 16); this lane's column is unit j[uh]
  const int j0 = wave * 32 + lr;
  float ureg[8][UNITS / 4];  // [g * 2 + uh][k step]: U[4 kp + lk][g UNITS + j0 + 16 uh]
#pragma unroll
  for (int gu = 0; gu < 8; ++gu)
#pragma unroll
    for (int kp = 0; kp < UNITS / 4; ++kp)
      ureg[gu][kp] = Ur[(size_t)(4 * kp + lk) * (4 * UNITS) + (gu >> 1) * UNITS + j0 + 16 * (gu & 1)];
  for (int i = tid; i < 16 * LD; i += 256) (&hs[0][0])[i] = 0.f;
  float c[2][4];
#pragma unroll
  for (int uh = 0; uh < 2; ++uh)
#pragma unroll
    for (int r = 0; r < 4; ++r) c[uh][r] = 0.f;
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    const int tin = dir ? (T - 1 - t) : t;
    f32x4 acc[8];
    // 16x16 C/D map: column = lane & 15, row = 4 (lane >> 4) + r
#pragma unroll
    for (int gu = 0; gu < 8; ++gu)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + 4 * lk + r;
        acc[gu][r] = m < M ? xp[((size_t)m * T + tin) * (8 * UNITS) + dir * 4 * UNITS + (gu >> 1) * UNITS + j0 + 16 * (gu & 1)] : 0.f;
      }
#pragma unroll
    for (int kp = 0; kp < UNITS / 4; ++kp) {
      const float a = hs[lr][4 * kp + lk];  // A: row (crop) = lane & 15, k = lane >> 4
#pragma unroll
      for (int gu = 0; gu < 8; ++gu) acc[gu] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, ureg[gu][kp], acc[gu], 0, 0, 0);
    }
    __syncthreads();  // every wave has finished reading h(t-1)
#pragma unroll
    for (int uh = 0; uh < 2; ++uh)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * lk + r;
        const float ig = sigmoidf_(acc[0 + uh][r]), fg = sigmoidf_(acc[2 + uh][r]);
        const float gg = tanhf(acc[4 + uh][r]), og = sigmoidf_(acc[6 + uh][r]);
        c[uh][r] = fg * c[uh][r] + ig * gg;
        const float h = og * tanhf(c[uh][r]);
        const int j = j0 + 16 * uh;
        hs[row][j] = h;
        const int m = m0 + row;
        if (m < M) out[((size_t)m * T + t) * (2 * UNITS) + dir * UNITS + j] = h;
      }
    __syncthreads();
  }
}

// logits: [M][T][C]; labels: [M][T-discard] (-1 padded); probs (nullable): [M][T-discard][C].
// One wave per crop; lane l owns classes l, l+64, ... (any alphabet size).
__global__ void ctc_kernel(const float* __restrict__ logits, int M, int T, int C, int discard, int* __restrict__ labels,
                           float* __restrict__ probs) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (m >= M) return;
  const int To = T - discard;
  const int blank = C - 1;
  int prev = -1, k = 0;
  for (int t = 0; t < To; ++t) {
    const float* row = logits + ((size_t)m * T + t + discard) * C;
    // per-lane argmax over its classes (ascending class index: first maximum wins), then the
    // wave-level reduction of (value, index) pairs with lowest index on ties
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < C; c += 64) {
      const float v = row[c];
      if (v > bv) {
        bv = v;
        bi = c;
      }
    }
    for (int o = 32; o; o >>= 1) {
      const float ov = __shfl_xor(bv, o);
      const int oi = __shfl_xor(bi, o);
      if (ov > bv || (ov == bv && oi < bi)) {
        bv = ov;
        bi = oi;
      }
    }
    if (probs) {
      float s = 0.f;
      for (int c = lane; c < C; c += 64) s += expf(row[c] - bv);
      for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
      for (int c = lane; c < C; c += 64) probs[((size_t)m * To + t) * C + c] = expf(row[c] - bv) / s;
    }
    if (lane == 0) {
      if (bi != prev && bi != blank) labels[(size_t)m * To + k++] = bi;
      prev = bi;
    }
  }
  if (lane == 0)
    for (; k < To; ++k) labels[(size_t)m * To + k] = -1;
}

int launch_crnn_input(kocr_ctx* ctx, const float* d_crops, float* d_x, int M, int Hc, int Wc) {
  const size_t total = (size_t)M * Hc * Wc;
  if (!total) return KOCR_OK;
  ProfScope ps(ctx, "crnn_input", 0, 8.0 * total);
  size_t b = (total + 255) / 256;
  if (b > 8192) b = 8192;
  hipLaunchKernelGGL(crnn_input_kernel, dim3((unsigned)b), dim3(256), 0, ctx->stream, d_crops, d_x, M, Hc, Wc);
  KOCR_HIP(ctx, hipGetLastError());
  return KOCR_OK;
}

int launch_crnn_to_keras(kocr_ctx* ctx, const Tensor& in, const Tensor& out) {
  if (in.C % 4 || in.cs != in.C || out.cs != out.C || out.H != in.wv() || out.W != in.H || out.C != in.C)
    KOCR_FAIL(ctx, KOCR_EINVAL, "crnn_to_keras: bad shapes");
  const size_t total = out.pixels() * (in.C / 4);
  if (!total) return KOCR_OK;
  ProfScope ps(ctx, "crnn_to_keras", 0, 8.0 * in.pixels() * in.C);
  size_t b = (total + 255) / 256;
  if (b > 8192) b = 8192;
  hipLaunchKernelGGL(crnn_to_keras_kernel, dim3((unsigned)b), dim3(256), 0, ctx->stream, in.p, out.p, in.N, in.H, in.wv(),
                     in.C / 4, in.W);
  KOCR_HIP(ctx, hipGetLastError());
  return KOCR_OK;
}

// conv_1 of M crops [M][Hc][Wc] into the cell grid `out` (N rows of cells() cells, cell height out.H = Hc + 1)
int launch_crnn_conv1_cells(kocr_ctx* ctx, const ConvLayer& L, const float* d_crops, int M, int Hc, int Wc, const Tensor& out) {
  if (L.Cin != 1 || L.Cout != 64 || L.KH != 3 || L.KW != 3 || !out.cellW || out.C != 64 || out.cs != 64 || out.co || out.H != Hc + 1 ||
      out.cellWv != Wc || out.cellW < Wc || (size_t)out.N * out.cells() < (size_t)M || L.d_post_a || !L.relu)
    KOCR_FAIL(ctx, KOCR_EINVAL, "crnn_conv1_cells: bad layer / shapes");
  ProfScope ps(ctx, "crnn_conv1_cells", 2.0 * M * Hc * Wc * 9 * 64, 4.0 * ((double)M * Hc * Wc + (double)out.pixels() * 64));
  hipLaunchKernelGGL(crnn_conv1_cells_kernel, dim3((unsigned)(out.N * out.cells()), (unsigned)out.H), dim3(256), 0, ctx->stream, d_crops,
                     L.d_w, L.d_pre_a, L.d_pre_b, out.p, out.amax, M, out.cells(), Hc, Wc, out.H, out.cellW, L.Cout_pad);
  KOCR_HIP(ctx, hipGetLastError());
  return KOCR_OK;
}

// `in` = cell grid (crop rows at cell rows 1 .. in.H - 1), out = [M][cellWv][in.H - 1][C]
int launch_crnn_cells_to_keras(kocr_ctx* ctx, const Tensor& in, const Tensor& out) {
  if (!in.cellW || in.C % 4 || in.cs != in.C || in.co || out.cs != out.C || out.co || out.H != in.cellWv || out.W != in.H - 1 || out.C != in.C ||
      (size_t)in.N * in.cells() < (size_t)out.N)
    KOCR_FAIL(ctx, KOCR_EINVAL, "crnn_cells_to_keras: bad shapes");
  const size_t total = out.pixels() * (in.C / 4);
  if (!total) return KOCR_OK;
  ProfScope ps(ctx, "crnn_to_keras", 0, 8.0 * out.pixels() * in.C);
  size_t b = (total + 255) / 256;
  if (b > 8192) b = 8192;
  hipLaunchKernelGGL(crnn_cells_to_keras_kernel, dim3((unsigned)b), dim3(256), 0, ctx->stream, in.p, out.p, out.N, in.H - 1, in.cellWv,
                     in.C / 4, in.cells(), in.H, in.cellW);
  KOCR_HIP(ctx, hipGetLastError());
  return KOCR_OK;
}

int launch_stn_sample(kocr_ctx* ctx, const Tensor& x, const float* d_theta, const Tensor& out) {
  if (x.cs != x.C || out.cs != out.C || x.co || out.co) KOCR_FAIL(ctx, KOCR_EINVAL, "stn: dense tensors only");
  const size_t pix = x.pixels();
  if (!pix) return KOCR_OK;
  ProfScope ps(ctx, "stn_sample", 0, 4.0 * pix * x.C * 5);
  hipLaunchKernelGGL(stn_sample_kernel, dim3((unsigned)pix), dim3(128), 0, ctx->stream, x.p, d_theta, out.p, x.N,
                     x.H, x.W, x.C);
  KOCR_HIP(ctx, hipGetLastError());
  return KOCR_OK;
}

// ---- stn_dense_1 (recognition.py:276: Dense(64, relu) over the flattened 50 x 7 x 32 localisation features) ---------------
// A 512 x 11200 x 64 GEMM: four 128-row tiles on the implicit-GEMM kernel left 252 CUs idle for 0.65 ms.  Split-K in two
// deterministic passes: partial[s][m][o] = sum over K slice s (fp32 FMA chain in k order), then out = act(sum_s partial).
constexpr int DSK_ROWS = 16, DSK_KCHUNK = 448;
__global__ __launch_bounds__(256) void dense_splitk_partial_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                  float* __restrict__ partial, int M, int K, int ldw) {
  __shared__ float xs[DSK_ROWS][DSK_KCHUNK + 1];
  const int m0 = blockIdx.x * DSK_ROWS, s = blockIdx.y, k0 = s * DSK_KCHUNK;
  const int kn = min(DSK_KCHUNK, K - k0);
  for (int i = threadIdx.x; i < DSK_ROWS * DSK_KCHUNK; i += 256) {
    const int r = i / DSK_KCHUNK, k = i - r * DSK_KCHUNK;
    xs[r][k] = (m0 + r < M && k < kn) ? x[(size_t)(m0 + r) * K + k0 + k] : 0.f;
  }
  __syncthreads();
  const int o = threadIdx.x & 63, rg = threadIdx.x >> 6;  // output column, group of four rows
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const float* wp = w + (size_t)k0 * ldw + o;
  for (int k = 0; k < kn; ++k) {
    const float wv = wp[(size_t)k * ldw];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = fmaf(xs[4 * rg + r][k], wv, acc[r]);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (m0 + 4 * rg + r < M) partial[((size_t)s * M + m0 + 4 * rg + r) * 64 + o] = acc[r];
}

__global__ void dense_splitk_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out, const float* __restrict__ pre_a,
                                           const float* __restrict__ pre_b, int M, int S, int relu) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * 64) return;
  float v = 0.f;
  for (int s = 0; s < S; ++s) v += partial[(size_t)s * M * 64 + i];
  v = v * pre_a[i & 63] + pre_b[i & 63];
  out[i] = relu ? fmaxf(v, 0.f) : v;
}

// in: [M][K] contiguous, L: a 1x1 layer with Cout == 64 and Cin == K (weights [Kpad][Cout_pad], k-major), out: [M][64] contiguous;
// d_partial: ceil(K / 448) * M * 64 floats
size_t dense_splitk_workspace(int M, int K) { return (size_t)((K + DSK_KCHUNK - 1) / DSK_KCHUNK) * M * 64 * sizeof(float); }
int launch_dense_splitk(kocr_ctx* ctx, const ConvLayer& L, const float* d_in, float* d_out, float* d_partial, int M) {
  if (L.Cout != 64 || L.KH != 1 || L.KW != 1 || L.d_post_a) KOCR_FAIL(ctx, KOCR_EINVAL, "dense_splitk: unsupported layer " + L.name);
  const int K = L.Cin, S = (K + DSK_KCHUNK - 1) / DSK_KCHUNK;
  ProfScope ps(ctx, "dense_splitk", 2.0 * M * (double)K * 64, 4.0 * ((double)M * K + (double)K * 64));
  hipLaunchKernelGGL(dense_splitk_partial_kernel, dim3((M + DSK_ROWS - 1) / DSK_ROWS, S), dim3(256), 0, ctx->stream, d_in, L.d_w,
                     d_partial, M, K, L.Cout_pad);
  hipLaunchKernelGGL(dense_splitk_reduce_kernel, dim3((M * 64 + 255) / 256), dim3(256), 0, ctx->stream, d_partial, d_out, L.d_pre_a,
                     L.d_pre_b, M, S, L.relu);
  KOCR_HIP(ctx, hipGetLastError());
  return KOCR_OK;
}

int launch_lstm(kocr_ctx* ctx, const float* d_xp, const float* d_Uf, const float* d_Ub, float* d_out, int M, int T) {
  if (M <= 0) return KOCR_OK;
  ProfScope ps(ctx, "lstm_recurrence", 2.0 * 2 * M * (double)T * 128 * 512, 0);
  const bool old = !ctx->sw.lstm16;
  if (old)
    hipLaunchKernelGGL(lstm_kernel<128>, dim3((M + 31) / 32, 2), dim3(256), 0, ctx->stream, d_xp, d_Uf, d_Ub, d_out, M, T);
  else
    hipLaunchKernelGGL(lstm16_kernel<128>, dim3((M + 15) / 16, 2), dim3(256), 0, ctx->stream, d_xp, d_Uf, d_Ub, d_out, M, T);
  KOCR_HIP(ctx, hipGetLastError());
  return KOCR_OK;
}

int launch_ctc(kocr_ctx* ctx, const float* d_logits, int M, int T, int C, int discard, int* d_labels, float* d_probs) {
  if (M <= 0) return KOCR_OK;
  ProfScope ps(ctx, "ctc_greedy", 0, 4.0 * M * T * C);
  hipLaunchKernelGGL(ctc_kernel, dim3((M + 3) / 4), dim3(256), 0, ctx->stream, d_logits, M, T, C, discard, d_labels,
                     d_probs);
  KOCR_HIP(ctx, hipGetLastError());
  return KOCR_OK;
}
