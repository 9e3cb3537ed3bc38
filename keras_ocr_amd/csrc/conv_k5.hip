// conv_k5.hip — 5x5 / stride 1 'same' convolution with 16 output channels on SMALL images (<= 384 pixels), on the gfx950
// BF16 matrix cores with the exact bf16x3 operand split of conv_dsplit.hip (six of the nine piece products, fp32
// accumulation).  This is the first layer of the recogniser's localisation network (recognition.py:259-262: Conv2D(16,
// (5, 5), padding='same', activation='relu') on the 7 x 50 x 512 feature map): K = 25 x 512 = 12800 per output against
// N = 16 -- on the 128 x 32 fp32-MFMA tile it ran with half of every product column empty on the slow pipe (39 TFLOP/s,
// 7 % of the recogniser).
//
//   block = 256 threads (4 waves) = ONE image.  Per 16-channel chunk the (H + 4) x (W + 4) haloed image is loaded with raw
//   buffer loads (out-of-image offsets return the zero padding), split once, and stored as three bf16 planes [pixel][16 ch]
//   (32-byte pixel stride: the 16 pixels of an operand are contiguous).  The products run on v_mfma_f32_16x16x32_bf16 with
//   the WEIGHTS as the A operand (16 couts x 32 k) and the pixels as B (32 k x 16 pixels), so that a lane ends up with four
//   consecutive couts of one pixel (one 16-byte store).  K = 32 of one MFMA = 16 channels x TWO taps: lanes 0..31 read tap
//   2j, lanes 32..63 tap 2j + 1 (the next pixel of the halo row, or the first of the next halo row for the two pairs that
//   straddle a kernel row; tap 25 does not exist: zero weights, and its lanes re-read tap 24).  13 pairs x 6 products per
//   (chunk, 16-pixel tile); the H x W pixels are flattened into ceil(HW / 16) tiles dealt round-robin to the four waves
//   (at most six each: 24 accumulator registers).  One LDS buffer (64 KB), two barriers per chunk, two blocks per CU: one
//   block's products hide the other's loads and splits.
#include "split_common.h"
#include <algorithm>
#include <atomic>

struct K5Params {
  const float* in;
  const unsigned short* wgt;  // [Cin/16][13 tap pairs][3 pieces][64 lanes][8]
  float* out;
  const float* pre_a;
  const float* pre_b;
  const float* post_a;
  const float* post_b;
  int H, W, Cin, in_cs, in_co;
  int out_cs, out_co;
  int relu;
  unsigned* amax_out;
};

namespace {
constexpr int K5_MAXHP = 680;              // halo pixels a block can hold (3 planes = 65280 bytes of LDS)
constexpr int K5_MAXM = 384;               // 4 waves x 6 tiles x 16 pixels
constexpr int K5_PLANE = K5_MAXHP * 16;    // ushorts per piece plane
constexpr int K5_IPT = (K5_MAXHP * 4 + 255) / 256;  // gather items (pixel, channel quad) per thread: 11
constexpr int K5_TPW = 6;                  // tiles per wave
}  // namespace

__global__ __launch_bounds__(256, 2) void conv_k5_kernel(K5Params p) {
  __shared__ __attribute__((aligned(16))) unsigned short As[3 * K5_PLANE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const int n = blockIdx.x;
  const int HW = p.W + 4, HP = (p.H + 4) * HW, M = p.H * p.W;
  constexpr unsigned OOB = 0x80000000u;

  const float* img = p.in + (size_t)n * M * p.in_cs + p.in_co;
  const unsigned long long ib = (unsigned long long)img;
  const unsigned long long ibu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ib >> 32)) << 32) |
                                 (unsigned)__builtin_amdgcn_readfirstlane((int)ib);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)ibu, 0, 0x80000000, 0x00020000);

  unsigned goff[K5_IPT];
  int ldst[K5_IPT];
#pragma unroll
  for (int it = 0; it < K5_IPT; ++it) {
    const int item = tid + it * 256;
    const int px = item >> 2, c4 = item & 3;
    const int hy = px / HW, hx = px - hy * HW;
    const int gy = hy - 2, gx = hx - 2;
    const bool ok = px < HP && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    goff[it] = ok ? (unsigned)(((gy * p.W + gx) * p.in_cs + c4 * 4) * 4) : OOB;
    ldst[it] = px < HP ? px * 16 + c4 * 4 : -1;
  }
  auto load_raw = [&](v4f (&raw)[K5_IPT], int chunk) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < K5_IPT; ++it)
      raw[it] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff[it], chunk * 64, 0));
  };
  auto produce = [&](const v4f (&raw)[K5_IPT]) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < K5_IPT; ++it) {
      u2v h, m, l;
      kocr_split4(raw[it], h, m, l);
      if (ldst[it] >= 0) {
        unsigned short* dst = As + ldst[it];
        *reinterpret_cast<u2v*>(dst) = h;
        *reinterpret_cast<u2v*>(dst + K5_PLANE) = m;
        *reinterpret_cast<u2v*>(dst + 2 * K5_PLANE) = l;
      }
    }
  };

  v4f acc[K5_TPW];
#pragma unroll
  for (int i = 0; i < K5_TPW; ++i) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
  // B operand of tile i: lane = pixel (l15) x k-group g; g & 1 = channel half, g >> 1 = tap of the pair
  int abase[K5_TPW];
#pragma unroll
  for (int i = 0; i < K5_TPW; ++i) {
    const int pp = std::min((wave + 4 * i) * 16 + l15, M - 1);
    const int py = pp / p.W, px = pp - py * p.W;
    abase[i] = (py * HW + px) * 16 + (g & 1) * 8;
  }
  const int d_next = (g >> 1) ? 16 : 0;              // second tap of a pair: the next halo pixel ...
  const int d_cross = (g >> 1) ? (HW - 4) * 16 : 0;  // ... or, from kx = 4, the first pixel of the next halo row
  const unsigned short* w_lane = p.wgt + lane * 8;
  const int nchunks = p.Cin >> 4;

  v4f raw[K5_IPT];
  load_raw(raw, 0);
  produce(raw);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    if (c + 1 < nchunks) load_raw(raw, c + 1);
    const unsigned short* wc = w_lane + (size_t)c * 13 * 3 * 512;
#pragma unroll
    for (int pr = 0; pr < 13; ++pr) {
      const int t0 = 2 * pr, ky = t0 / 5, kx = t0 - ky * 5;
      const int toff = (ky * HW + kx) * 16 + (pr == 12 ? 0 : (kx == 4 ? d_cross : d_next));
      bf8 wv[3], x[K5_TPW][3];
#pragma unroll
      for (int s = 0; s < 3; ++s) wv[s] = *reinterpret_cast<const bf8*>(wc + (pr * 3 + s) * 512);
#pragma unroll
      for (int i = 0; i < K5_TPW; ++i)
#pragma unroll
        for (int s = 0; s < 3; ++s) x[i][s] = *reinterpret_cast<const bf8*>(As + s * K5_PLANE + abase[i] + toff);
      // smallest products first (the order of conv_dsplit.hip)
#pragma unroll
      for (int i = 0; i < K5_TPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[0], x[i][2], acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < K5_TPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[2], x[i][0], acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < K5_TPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[1], x[i][1], acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < K5_TPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[0], x[i][1], acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < K5_TPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[1], x[i][0], acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < K5_TPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[0], x[i][0], acc[i], 0, 0, 0);
    }
    if (c + 1 < nchunks) {
      __syncthreads();  // every wave has read this chunk
      produce(raw);
      __syncthreads();
    }
  }

  // ---- epilogue: 16x16 C/D map: row (cout) = 4 * (lane >> 4) + r, col (pixel of the tile) = lane & 15 ---------------------
  const v4f pa = *reinterpret_cast<const v4f*>(p.pre_a + 4 * g), pb = *reinterpret_cast<const v4f*>(p.pre_b + 4 * g);
  const bool has_post = p.post_a != nullptr;
  v4f qa = v4f{1.f, 1.f, 1.f, 1.f}, qb = v4f{0.f, 0.f, 0.f, 0.f};
  if (has_post) {
    qa = *reinterpret_cast<const v4f*>(p.post_a + 4 * g);
    qb = *reinterpret_cast<const v4f*>(p.post_b + 4 * g);
  }
  float* oimg = p.out + (size_t)n * M * p.out_cs + p.out_co;
  float mxv = 0.f;
#pragma unroll
  for (int i = 0; i < K5_TPW; ++i) {
    const int pp = (wave + 4 * i) * 16 + l15;
    v4f o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = acc[i][r] * pa[r] + pb[r];
      if (p.relu) v = fmaxf(v, 0.f);
      if (has_post) v = v * qa[r] + qb[r];
      o[r] = v;
    }
    if (pp < M) {
      *reinterpret_cast<v4f*>(oimg + (size_t)pp * p.out_cs + 4 * g) = o;
#pragma unroll
      for (int r = 0; r < 4; ++r) mxv = fmaxf(mxv, fabsf(o[r]));
    }
  }
  if (p.amax_out) kocr_amax_update(p.amax_out + n, mxv);  // per-image slot (Tensor::amax)
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
int prepare_k5(kocr_ctx* ctx, ConvLayer& L, const float* w, bool w_is_oihw) {
  if (L.KH != 5 || L.KW != 5 || L.dil != 1 || L.Cout != 16 || L.Cin % 16 != 0) return KOCR_OK;
  const int Cin = L.Cin, Cout = L.Cout;
  std::vector<unsigned short> u((size_t)(Cin / 16) * 13 * 3 * 512, 0);
  for (int c = 0; c < Cin; ++c)
    for (int tap = 0; tap < 25; ++tap)
      for (int o = 0; o < Cout; ++o) {
        const float wv = w_is_oihw ? w[((size_t)o * Cin + c) * 25 + tap] : w[((size_t)tap * Cin + c) * Cout + o];
        // MFMA 16x16x32 A operand: lane = (k >> 3) * 16 + row holds k = 8 * (lane >> 4) + j;  k = (tap & 1) * 16 + channel
        const int k = (tap & 1) * 16 + c % 16, lane = (k >> 3) * 16 + o, j = k & 7;
        unsigned short pc[3];
        kocr_split3_host(wv, pc);
        for (int s = 0; s < 3; ++s) u[((((size_t)(c / 16) * 13 + tap / 2) * 3 + s) * 64 + lane) * 8 + j] = pc[s];
      }
  void* d = nullptr;
  KOCR_TRY(ctx->dev_alloc(&d, u.size() * sizeof(unsigned short)));
  KOCR_HIP(ctx, hipMemcpy(d, u.data(), u.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
  L.d_k5 = (unsigned short*)d;
  return KOCR_OK;
}

bool k5_applicable(kocr_ctx* ctx, const ConvLayer& L, const Tensor& in, const Tensor& out) {
  const bool off = !ctx->sw.k5;
  return !off && L.d_k5 && in.cs % 4 == 0 && in.co % 4 == 0 &&
         ((uintptr_t)in.p & 15) == 0 && out.cs % 4 == 0 && out.co % 4 == 0 && ((uintptr_t)out.p & 15) == 0 &&
         in.H * in.W <= K5_MAXM && (in.H + 4) * (in.W + 4) <= K5_MAXHP;
}

int launch_conv_k5(kocr_ctx* ctx, const ConvLayer& L, const Tensor& in, const Tensor& out) {
  K5Params p;
  p.in = in.p;
  p.wgt = L.d_k5;
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
  p.out_cs = out.cs;
  p.out_co = out.co;
  p.relu = L.relu;
  p.amax_out = out.amax;
  const size_t M = in.pixels();
  static const bool per_layer = getenv("KOCR_PROF_LAYERS") != nullptr;
  char nm[64];
  if (per_layer)
    snprintf(nm, sizeof nm, "conv_k5_352x16:%s", L.name.c_str());
  else
    snprintf(nm, sizeof nm, "conv_k5_352x16");
  const double flops = 2.0 * (double)M * L.Kreal * L.Cout;
  const double bytes = 4.0 * ((double)M * L.Cin + (double)M * L.Cout + (double)L.Kreal * L.Cout);
  ProfScope ps(ctx, nm, flops, bytes);
  hipLaunchKernelGGL(conv_k5_kernel, dim3((unsigned)in.N), dim3(256), 0, ctx->stream, p);
  KOCR_HIP(ctx, hipGetLastError());
  return KOCR_OK;
}
