// elementwise.hip — HBM-bound NHWC helpers of the CRAFT / CRNN graphs.
//   maxpool 2x2 / stride 2 / 'valid'   keras MaxPooling2D  (detection.py:99-102, recognition.py:227,235)
//   maxpool 3x3 / stride 1 / 'same'    basenet.slice5.0    (detection.py:365-367), padding ignored (-inf)
//   bilinear resize, half-pixel centres UpsampleLike        (detection.py:290-303) written straight into
//                                       the channel slice of the concat buffer it feeds (detection.py:380-389)
// All kernels move float4 (4 channels) per lane, lanes run along channels then pixels, so a
// wave touches whole 128-B lines of both the source and the destination.
#include "split_common.h"
#include <algorithm>

struct EwParams {
  const float* in;
  float* out;
  int N, Hi, Wi, Ho, Wo, C4;  // C4 = channels / 4
  int in_cs, in_co, out_cs, out_co;
  float sy, sx;
  int row_off;  // maxpool2x2: first input row of output row 0 (0, or 1 for the flipped CRNN layout)
  int Wov;      // maxpool2x2: valid output width (Tensor::Wv): output columns [Wov, Wo) are written as zeros
};

__global__ void maxpool2x2_kernel(EwParams p) {
  const size_t total = (size_t)p.N * p.Ho * p.Wo * p.C4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = i % p.C4;
    size_t t = i / p.C4;
    const int ox = t % p.Wo;
    t /= p.Wo;
    const int oy = t % p.Ho;
    const int n = t / p.Ho;
    const size_t o = (((size_t)n * p.Ho + oy) * p.Wo + ox) * p.out_cs + p.out_co + c4 * 4;
    if (ox >= p.Wov) {  // zero padding columns of a width-padded output
      *reinterpret_cast<float4*>(p.out + o) = float4{0.f, 0.f, 0.f, 0.f};
      continue;
    }
    const size_t base = (((size_t)n * p.Hi + 2 * oy + p.row_off) * p.Wi + 2 * ox) * p.in_cs + p.in_co + c4 * 4;
    const float4 a = *reinterpret_cast<const float4*>(p.in + base);
    const float4 b = *reinterpret_cast<const float4*>(p.in + base + p.in_cs);
    const float4 c = *reinterpret_cast<const float4*>(p.in + base + (size_t)p.Wi * p.in_cs);
    const float4 d = *reinterpret_cast<const float4*>(p.in + base + (size_t)p.Wi * p.in_cs + p.in_cs);
    float4 r;
    r.x = fmaxf(fmaxf(a.x, b.x), fmaxf(c.x, d.x));
    r.y = fmaxf(fmaxf(a.y, b.y), fmaxf(c.y, d.y));
    r.z = fmaxf(fmaxf(a.z, b.z), fmaxf(c.z, d.z));
    r.w = fmaxf(fmaxf(a.w, b.w), fmaxf(c.w, d.w));
    *reinterpret_cast<float4*>(p.out + o) = r;
  }
}

__global__ void maxpool3x3s1_kernel(EwParams p) {
  const size_t total = (size_t)p.N * p.Ho * p.Wo * p.C4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = i % p.C4;
    size_t t = i / p.C4;
    const int ox = t % p.Wo;
    t /= p.Wo;
    const int oy = t % p.Ho;
    const int n = t / p.Ho;
    float4 r = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int dy = -1; dy <= 1; ++dy) {
      const int iy = oy + dy;
      if (iy < 0 || iy >= p.Hi) continue;
      for (int dx = -1; dx <= 1; ++dx) {
        const int ix = ox + dx;
        if (ix < 0 || ix >= p.Wi) continue;
        const float4 v = *reinterpret_cast<const float4*>(
            p.in + (((size_t)n * p.Hi + iy) * p.Wi + ix) * p.in_cs + p.in_co + c4 * 4);
        r.x = fmaxf(r.x, v.x);
        r.y = fmaxf(r.y, v.y);
        r.z = fmaxf(r.z, v.z);
        r.w = fmaxf(r.w, v.w);
      }
    }
    const size_t o = (((size_t)n * p.Ho + oy) * p.Wo + ox) * p.out_cs + p.out_co + c4 * 4;
    *reinterpret_cast<float4*>(p.out + o) = r;
  }
}

// tf.compat.v1.image.resize_bilinear(half_pixel_centers=True): src = (dst+0.5)*(in/out)-0.5,
// lower = max(floor(src),0), upper = min(ceil(src), in-1), lerp = src - floor(src);
// top = tl + (tr-tl)*xl; bottom = bl + (br-bl)*xl; out = top + (bottom-top)*yl.
__global__ void resize_bilinear_kernel(EwParams p) {
  const size_t total = (size_t)p.N * p.Ho * p.Wo * p.C4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = i % p.C4;
    size_t t = i / p.C4;
    const int ox = t % p.Wo;
    t /= p.Wo;
    const int oy = t % p.Ho;
    const int n = t / p.Ho;
    const float fy = ((float)oy + 0.5f) * p.sy - 0.5f;
    const float fx = ((float)ox + 0.5f) * p.sx - 0.5f;
    const float fly = floorf(fy), flx = floorf(fx);
    const int y0 = max((int)fly, 0), y1 = min((int)ceilf(fy), p.Hi - 1);
    const int x0 = max((int)flx, 0), x1 = min((int)ceilf(fx), p.Wi - 1);
    const float yl = fy - fly, xl = fx - flx;
    const size_t rb = (size_t)n * p.Hi;
    const float* b0 = p.in + ((rb + y0) * p.Wi) * p.in_cs + p.in_co + c4 * 4;
    const float* b1 = p.in + ((rb + y1) * p.Wi) * p.in_cs + p.in_co + c4 * 4;
    const float4 tl = *reinterpret_cast<const float4*>(b0 + (size_t)x0 * p.in_cs);
    const float4 tr = *reinterpret_cast<const float4*>(b0 + (size_t)x1 * p.in_cs);
    const float4 bl = *reinterpret_cast<const float4*>(b1 + (size_t)x0 * p.in_cs);
    const float4 br = *reinterpret_cast<const float4*>(b1 + (size_t)x1 * p.in_cs);
    float4 r;
    {
      const float top = tl.x + (tr.x - tl.x) * xl, bot = bl.x + (br.x - bl.x) * xl;
      r.x = top + (bot - top) * yl;
    }
    {
      const float top = tl.y + (tr.y - tl.y) * xl, bot = bl.y + (br.y - bl.y) * xl;
      r.y = top + (bot - top) * yl;
    }
    {
      const float top = tl.z + (tr.z - tl.z) * xl, bot = bl.z + (br.z - bl.z) * xl;
      r.z = top + (bot - top) * yl;
    }
    {
      const float top = tl.w + (tr.w - tl.w) * xl, bot = bl.w + (br.w - bl.w) * xl;
      r.w = top + (bot - top) * yl;
    }
    const size_t o = (((size_t)n * p.Ho + oy) * p.Wo + ox) * p.out_cs + p.out_co + c4 * 4;
    *reinterpret_cast<float4*>(p.out + o) = r;
  }
}

__global__ void copy_channels_kernel(EwParams p) {
  const size_t total = (size_t)p.N * p.Ho * p.Wo * p.C4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = i % p.C4;
    const size_t px = i / p.C4;
    *reinterpret_cast<float4*>(p.out + px * p.out_cs + p.out_co + c4 * 4) =
        *reinterpret_cast<const float4*>(p.in + px * p.in_cs + p.in_co + c4 * 4);
  }
}

static int check_vec(kocr_ctx* ctx, const Tensor& in, const Tensor& out, const char* what) {
  if (in.C != out.C || in.N != out.N) KOCR_FAIL(ctx, KOCR_EINVAL, std::string(what) + ": shape mismatch");
  if (in.C % 4 || in.cs % 4 || in.co % 4 || out.cs % 4 || out.co % 4)
    KOCR_FAIL(ctx, KOCR_EINVAL, std::string(what) + ": channels must be multiples of 4");
  return KOCR_OK;
}

static EwParams make_params(const Tensor& in, const Tensor& out) {
  EwParams p;
  p.in = in.p;
  p.out = out.p;
  p.N = in.N;
  p.Hi = in.H;
  p.Wi = in.W;
  p.Ho = out.H;
  p.Wo = out.W;
  p.C4 = in.C / 4;
  p.in_cs = in.cs;
  p.in_co = in.co;
  p.out_cs = out.cs;
  p.out_co = out.co;
  p.sy = p.sx = 1.f;
  p.row_off = 0;
  p.Wov = out.W;
  return p;
}

static dim3 ew_grid(size_t total) {
  size_t b = (total + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  if (b == 0) b = 1;
  return dim3((unsigned)b);
}

int launch_maxpool2x2(kocr_ctx* ctx, const Tensor& in, const Tensor& out, int row_off) {
  KOCR_TRY(check_vec(ctx, in, out, "maxpool2x2"));
  if (out.H != (in.H - row_off) / 2 || out.wv() != in.wv() / 2) KOCR_FAIL(ctx, KOCR_EINVAL, "maxpool2x2: bad output size");
  EwParams p = make_params(in, out);
  p.row_off = row_off;
  p.Wov = out.wv();
  const size_t total = out.pixels() * p.C4;
  if (!total) return KOCR_OK;
  {
    ProfScope ps(ctx, "maxpool2x2", 0, 4.0 * (in.pixels() + out.pixels()) * in.C);
    hipLaunchKernelGGL(maxpool2x2_kernel, ew_grid(total), dim3(256), 0, ctx->stream, p);
    KOCR_HIP(ctx, hipGetLastError());
  }
  if (out.amax) {  // |out| <= max |in|
    if (in.amax) return launch_amax_copy(ctx, in.amax, out.amax, out.N);
    return launch_absmax(ctx, out, out.amax);
  }
  return KOCR_OK;
}

int launch_maxpool3x3s1(kocr_ctx* ctx, const Tensor& in, const Tensor& out) {
  KOCR_TRY(check_vec(ctx, in, out, "maxpool3x3s1"));
  if (out.H != in.H || out.W != in.W) KOCR_FAIL(ctx, KOCR_EINVAL, "maxpool3x3s1: bad output size");
  EwParams p = make_params(in, out);
  const size_t total = out.pixels() * p.C4;
  if (!total) return KOCR_OK;
  {
    ProfScope ps(ctx, "maxpool3x3s1", 0, 4.0 * (in.pixels() + out.pixels()) * in.C);
    hipLaunchKernelGGL(maxpool3x3s1_kernel, ew_grid(total), dim3(256), 0, ctx->stream, p);
    KOCR_HIP(ctx, hipGetLastError());
  }
  if (out.amax) {  // |out| <= max |in|
    if (in.amax) return launch_amax_copy(ctx, in.amax, out.amax, out.N);
    return launch_absmax(ctx, out, out.amax);
  }
  return KOCR_OK;
}

int launch_resize_bilinear(kocr_ctx* ctx, const Tensor& in, const Tensor& out) {
  KOCR_TRY(check_vec(ctx, in, out, "resize_bilinear"));
  EwParams p = make_params(in, out);
  p.sy = (float)in.H / (float)out.H;
  p.sx = (float)in.W / (float)out.W;
  const size_t total = out.pixels() * p.C4;
  if (!total) return KOCR_OK;
  {
    ProfScope ps(ctx, "resize_bilinear", 0, 4.0 * (in.pixels() + out.pixels()) * in.C);
    hipLaunchKernelGGL(resize_bilinear_kernel, ew_grid(total), dim3(256), 0, ctx->stream, p);
    KOCR_HIP(ctx, hipGetLastError());
  }
  if (out.amax) {  // |out| <= max |in|
    if (in.amax) return launch_amax_copy(ctx, in.amax, out.amax, out.N);
    return launch_absmax(ctx, out, out.amax);
  }
  return KOCR_OK;
}

int launch_copy_channels(kocr_ctx* ctx, const Tensor& in, const Tensor& out) {
  KOCR_TRY(check_vec(ctx, in, out, "copy_channels"));
  if (out.H != in.H || out.W != in.W) KOCR_FAIL(ctx, KOCR_EINVAL, "copy_channels: bad output size");
  EwParams p = make_params(in, out);
  const size_t total = out.pixels() * p.C4;
  if (!total) return KOCR_OK;
  ProfScope ps(ctx, "copy_channels", 0, 8.0 * out.pixels() * in.C);
  hipLaunchKernelGGL(copy_channels_kernel, ew_grid(total), dim3(256), 0, ctx->stream, p);
  KOCR_HIP(ctx, hipGetLastError());
  return KOCR_OK;
}

// ---------------------------------------------------------------------------------------
// CRAFT head tail, fused: conv_cls.6 (1x1, 16 -> 16, ReLU) + conv_cls.8 (1x1, 16 -> 2, linear),
// detection.py:407-410.  Both are far below the MFMA ridge (4 and 0.9 FLOP/B as separate layers):
// one thread per pixel reads its 16 channels once (64 B), keeps the 16 hidden values in registers
// and writes the two heat-map channels (8 B); weights and biases sit in LDS.  fp32 fma chains over
// the input channels in index order.
// ---------------------------------------------------------------------------------------
struct HeadTailParams {
  const float* in;   // [P][in_cs], 16 channels at in_co
  float* out;        // [P][2]
  const float* w6;   // [16][ld6] (k = cin, col = cout), pre_a6/pre_b6 per cout
  const float* a6;
  const float* b6;
  const float* w8;   // [16][ld8]
  const float* a8;
  const float* b8;
  int ld6, ld8, in_cs, in_co, relu6;
  size_t P;
};

__global__ __launch_bounds__(256) void head_tail_kernel(HeadTailParams p) {
  __shared__ float s_w6[16][16], s_a6[16], s_b6[16], s_w8[16][2], s_a8[2], s_b8[2];
  const int tid = threadIdx.x;
  s_w6[tid >> 4][tid & 15] = p.w6[(tid >> 4) * p.ld6 + (tid & 15)];
  if (tid < 16) {
    s_a6[tid] = p.a6[tid];
    s_b6[tid] = p.b6[tid];
    s_w8[tid][0] = p.w8[tid * p.ld8 + 0];
    s_w8[tid][1] = p.w8[tid * p.ld8 + 1];
  }
  if (tid < 2) {
    s_a8[tid] = p.a8[tid];
    s_b8[tid] = p.b8[tid];
  }
  __syncthreads();
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef float v2f __attribute__((ext_vector_type(2)));
  for (size_t i = blockIdx.x * (size_t)blockDim.x + tid; i < p.P; i += (size_t)gridDim.x * blockDim.x) {
    const float* src = p.in + i * p.in_cs + p.in_co;
    float x[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const v4f v = *reinterpret_cast<const v4f*>(src + 4 * q);
      x[4 * q + 0] = v.x;
      x[4 * q + 1] = v.y;
      x[4 * q + 2] = v.z;
      x[4 * q + 3] = v.w;
    }
    float y0 = 0.f, y1 = 0.f;
#pragma unroll
    for (int o = 0; o < 16; ++o) {
      float h = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) h = fmaf(x[c], s_w6[c][o], h);
      h = h * s_a6[o] + s_b6[o];
      if (p.relu6) h = fmaxf(h, 0.f);
      y0 = fmaf(h, s_w8[o][0], y0);
      y1 = fmaf(h, s_w8[o][1], y1);
    }
    *reinterpret_cast<v2f*>(p.out + 2 * i) = v2f{y0 * s_a8[0] + s_b8[0], y1 * s_a8[1] + s_b8[1]};
  }
}

int launch_head_tail(kocr_ctx* ctx, const ConvLayer& L6, const ConvLayer& L8, const Tensor& in, float* d_heat) {
  if (L6.Cin != 16 || L6.Cout != 16 || L6.KH != 1 || L8.Cin != 16 || L8.Cout != 2 || L8.KH != 1 || L8.relu ||
      L6.d_post_a || L8.d_post_a || !L6.tap_inner || !L8.tap_inner || in.C != 16 || in.cs % 4 || in.co % 4)
    KOCR_FAIL(ctx, KOCR_EINVAL, "head_tail: unexpected layer shapes");
  HeadTailParams p;
  p.in = in.p;
  p.out = d_heat;
  p.w6 = L6.d_w;
  p.a6 = L6.d_pre_a;
  p.b6 = L6.d_pre_b;
  p.w8 = L8.d_w;
  p.a8 = L8.d_pre_a;
  p.b8 = L8.d_pre_b;
  p.ld6 = L6.Cout_pad;
  p.ld8 = L8.Cout_pad;
  p.in_cs = in.cs;
  p.in_co = in.co;
  p.relu6 = L6.relu;
  p.P = in.pixels();
  if (!p.P) return KOCR_OK;
  const double flops = 2.0 * (double)p.P * (16 * 16 + 16 * 2);
  ProfScope ps(ctx, "conv_head_tail", flops, (double)p.P * (64 + 8));
  hipLaunchKernelGGL(head_tail_kernel, ew_grid(p.P), dim3(256), 0, ctx->stream, p);
  KOCR_HIP(ctx, hipGetLastError());
  return KOCR_OK;
}

// ---------------------------------------------------------------------------------------
// max |x| bookkeeping for the fp16-split convolutions (Tensor::amax)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void absmax_kernel(const float* in, size_t img_pixels, int C4, int cs, int co, unsigned* slots) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  float m = 0.f;
  const size_t total = img_pixels * C4;
  const float* base = in + (size_t)blockIdx.y * img_pixels * cs;  // image blockIdx.y
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t px = i / C4;
    const int c4 = (int)(i - px * C4);
    const v4f v = *reinterpret_cast<const v4f*>(base + px * cs + co + 4 * c4);
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  }
  kocr_amax_update(slots + blockIdx.y, m);
}

__global__ void amax_copy_kernel(const unsigned* from, unsigned* to, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicMax(to + i, from[i]);
}

int launch_absmax(kocr_ctx* ctx, const Tensor& t, unsigned* slots) {
  if (t.C % 4 || t.cs % 4 || t.co % 4 || ((uintptr_t)t.p & 15)) KOCR_FAIL(ctx, KOCR_EINVAL, "absmax: unaligned tensor");
  const size_t total = (size_t)t.H * t.W * (t.C / 4);
  if (!total || !t.N) return KOCR_OK;
  ProfScope ps(ctx, "absmax", 0, 4.0 * t.pixels() * t.C);
  // >= 16 sixteen-byte loads per thread and at most a few thousand blocks in all: a block per 256 values drowned small
  // images (the recogniser's 512 crops) in block launches and same-address atomics (0.3 TB/s)
  size_t b = (total + 4095) / 4096;
  const size_t cap = std::max<size_t>(1, 4096 / (size_t)t.N);
  if (b > cap) b = cap;
  // images are the grid's y dimension (<= 65 535 per launch: ADVICE r04 -- batches of more small images go in chunks)
  constexpr int NCHUNK = 32768;
  for (int n0 = 0; n0 < t.N; n0 += NCHUNK) {
    const int nn = std::min(NCHUNK, t.N - n0);
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)b, (unsigned)nn), dim3(256), 0, ctx->stream,
                       t.p + (size_t)n0 * t.H * t.W * t.cs, (size_t)t.H * t.W, t.C / 4, t.cs, t.co, slots + n0);
    KOCR_HIP(ctx, hipGetLastError());
  }
  return KOCR_OK;
}

// ---------------------------------------------------------------------------------------
// Range statistics of the fp16x2 arithmetic (developer instrumentation, OFF unless kocr_range_stats_enable; VERDICT r04
// item 5): for the INPUT tensor of an fp16-arithmetic convolution and the per-image (per-cell) scale 2^e the kernel is
// about to derive from the max-|x| slots, how many non-zero elements fall where the two-piece split no longer carries
// fp32's 24 bits -- s = |x| 2^e < 2^-4: the low piece is an fp16 subnormal with fewer than 20 bits below the high piece's
// lsb ... relative precision of the element worse than 2^-21; s < 2^-14: the HIGH piece is subnormal, worse than 2^-11 -- and
// how much of the tensor's sum |x| those elements carry.
// out[0..2] = counts (non-zero, s < 2^-4, s < 2^-14), sums[0..1] = sum |x| (all, those with s < 2^-4)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void range_stats_kernel(const float* in, int H, int W, int C4, int cs, int co, const unsigned* slots,
                                                          int cells, int cellW, int top, unsigned long long* cnt, double* sums, int n0) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const int n = n0 + blockIdx.y;
  const size_t img_pixels = (size_t)H * W, total = img_pixels * C4;
  const float* base = in + (size_t)n * img_pixels * cs;
  unsigned long long c0 = 0, c1 = 0, c2 = 0;
  double s0 = 0, s1 = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t px = i / C4;
    const int c4 = (int)(i - px * C4);
    const int slot = cells ? n * cells + (int)((px % W) / cellW) : n;
    const int e = kocr_scale_exp(slots + slot, top);
    const float sc = kocr_pow2(e);
    const v4f v = *reinterpret_cast<const v4f*>(base + px * cs + co + 4 * c4);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float a = fabsf(v[k]), s = a * sc;
      if (a > 0.f) {
        ++c0;
        s0 += a;
        if (s < 0.0625f) {
          ++c1;
          s1 += a;
        }
        if (s < 6.103515625e-05f) ++c2;
      }
    }
  }
  atomicAdd(cnt + 0, c0);
  atomicAdd(cnt + 1, c1);
  atomicAdd(cnt + 2, c2);
  atomicAdd(sums + 0, s0);
  atomicAdd(sums + 1, s1);
}

// accumulates into ctx->range[name]; `top` = the exponent the consumer places the image's maximum at (12 for the F(4,3)
// kernels, 14 for the <= 32-cout kernel)
int launch_range_stats(kocr_ctx* ctx, const std::string& name, const Tensor& t, const unsigned* slots, int top) {
  if (!ctx->range_on || !slots || t.C % 4 || t.cs % 4 || t.co % 4 || !t.pixels()) return KOCR_OK;
  if (!ctx->d_range) KOCR_FAIL(ctx, KOCR_EINVAL, "range statistics: call kocr_range_stats_enable first");
  unsigned long long* cnt = (unsigned long long*)ctx->d_range;
  double* sums = (double*)((char*)ctx->d_range + 32);
  KOCR_HIP(ctx, hipMemsetAsync(ctx->d_range, 0, 64, ctx->stream));
  size_t b = ((size_t)t.H * t.W * (t.C / 4) + 4095) / 4096;
  if (b > 64) b = 64;
  constexpr int NCHUNK = 32768;  // images are the grid's y dimension (<= 65 535 per launch), as in launch_absmax
  for (int n0 = 0; n0 < t.N; n0 += NCHUNK) {
    hipLaunchKernelGGL(range_stats_kernel, dim3((unsigned)b, (unsigned)std::min(NCHUNK, t.N - n0)), dim3(256), 0, ctx->stream, t.p, t.H,
                       t.W, t.C / 4, t.cs, t.co, slots, t.cells(), t.cellW ? t.cellW : 1, top, cnt, sums, n0);
    KOCR_HIP(ctx, hipGetLastError());
  }
  unsigned long long hc[4];
  double hs[4];
  KOCR_HIP(ctx, hipMemcpyAsync(hc, cnt, 32, hipMemcpyDeviceToHost, ctx->stream));
  KOCR_HIP(ctx, hipMemcpyAsync(hs, sums, 32, hipMemcpyDeviceToHost, ctx->stream));
  KOCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  RangeRow& r = ctx->range[name];
  r.launches += 1;
  r.elements += (double)t.pixels() * t.C;
  r.nonzero += (double)hc[0];
  r.below_m4 += (double)hc[1];
  r.below_m14 += (double)hc[2];
  r.sum_abs += hs[0];
  r.sum_abs_below_m4 += hs[1];
  return KOCR_OK;
}

int launch_amax_copy(kocr_ctx* ctx, const unsigned* from, unsigned* to, int n) {
  if (!from || !to || from == to || n <= 0) return KOCR_OK;
  hipLaunchKernelGGL(amax_copy_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, from, to, n);
  KOCR_HIP(ctx, hipGetLastError());
  return KOCR_OK;
}
