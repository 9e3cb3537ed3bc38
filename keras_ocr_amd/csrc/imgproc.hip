// imgproc.hip — tools.resize_image + tools.pad on the GPU (tools.py:356-398).
//
// cv2.resize(image, dsize) with the default INTER_LINEAR on uint8: half-pixel-centre mapping,
// 11-bit fixed-point coefficients, horizontal pass in int32, vertical pass
//   dst = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2
// (OpenCV's HResizeLinear / VResizeLinear for uchar).  The coefficient tables are built on the
// host in the same float/double steps as OpenCV (and as oracle/tools.py); the kernel is
// integer-only, so the result is bit-exact.  The kernel writes straight into the padded batch
// (tools.pad: bottom/right, cval 255) — one HBM pass: 3 B/px read (x4 taps, L2-resident), 3 B/px
// written.
#include "common.h"
#include <cmath>

struct ResizeTables {
  int* xi0;
  int* xi1;
  int* xa0;
  int* xa1;
  int* yi0;
  int* yi1;
  int* yb0;
  int* yb1;
};

namespace {
void axis_tables(int src, int dst, bool horizontal, std::vector<int>& i0, std::vector<int>& i1, std::vector<int>& c0,
                 std::vector<int>& c1) {
  const double scale = (double)src / (double)dst;
  i0.resize(dst);
  i1.resize(dst);
  c0.resize(dst);
  c1.resize(dst);
  for (int d = 0; d < dst; ++d) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)std::floor(f);
    f = f - (float)s;
    if (horizontal) {
      if (s < 0) {
        f = 0.f;
        s = 0;
      }
      if (s >= src - 1) {
        f = 0.f;
        s = src - 1;
      }
    }
    c1[d] = (int)std::nearbyint(f * 2048.f);          // saturate_cast<short>: round half to even
    c0[d] = (int)std::nearbyint((1.f - f) * 2048.f);
    i0[d] = std::min(std::max(s, 0), src - 1);
    i1[d] = std::min(std::max(s + 1, 0), src - 1);
  }
}
}  // namespace

// grid: (x blocks, dst rows (Hmax), N)
__global__ void resize_pad_kernel(const uint8_t* __restrict__ src, int sh, int sw, size_t src_img_stride,
                                  uint8_t* __restrict__ dst, int dh, int dw, int Hmax, int Wmax, int cval, ResizeTables t) {
  const int n = blockIdx.z, y = blockIdx.y;
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= Wmax) return;
  uint8_t* o = dst + (((size_t)n * Hmax + y) * Wmax + x) * 3;
  if (y >= dh || x >= dw) {  // tools.pad (cval 255) / tools.fit letterbox (cval 0)
    o[0] = o[1] = o[2] = (uint8_t)cval;
    return;
  }
  const uint8_t* im = src + (size_t)n * src_img_stride;
  const int x0 = t.xi0[x], x1 = t.xi1[x], a0 = t.xa0[x], a1 = t.xa1[x];
  const int y0 = t.yi0[y], y1 = t.yi1[y], b0 = t.yb0[y], b1 = t.yb1[y];
  const uint8_t* r0 = im + (size_t)y0 * sw * 3;
  const uint8_t* r1 = im + (size_t)y1 * sw * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int S0 = r0[x0 * 3 + c] * a0 + r0[x1 * 3 + c] * a1;
    const int S1 = r1[x0 * 3 + c] * a0 + r1[x1 * 3 + c] * a1;
    const int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
    o[c] = (uint8_t)min(max(v, 0), 255);
  }
}

// Resizes n images of identical size (sh,sw) (src stride = sh*sw*3) to (dh,dw) and writes them,
// padded to (Hmax,Wmax), at d_dst.  Tables are staged in ctx->io (caller must not hold io data).
int launch_resize_pad(kocr_ctx* ctx, const uint8_t* d_src, int n, int sh, int sw, uint8_t* d_dst, int dh, int dw,
                      int Hmax, int Wmax, int cval, Arena& tab_arena) {
  if (n <= 0) return KOCR_OK;
  if (dh <= 0 || dw <= 0 || dh > Hmax || dw > Wmax || sh <= 0 || sw <= 0)
    KOCR_FAIL(ctx, KOCR_EINVAL, "resize_pad: bad sizes");
  std::vector<int> tb[8];
  axis_tables(sw, dw, true, tb[0], tb[1], tb[2], tb[3]);
  axis_tables(sh, dh, false, tb[4], tb[5], tb[6], tb[7]);
  std::vector<int> flat;
  for (auto& v : tb) flat.insert(flat.end(), v.begin(), v.end());
  int* d_tab = (int*)arena_alloc(tab_arena, flat.size() * sizeof(int));
  if (!d_tab) KOCR_FAIL(ctx, KOCR_ENOMEM, "resize_pad: table arena exhausted");
  // pageable-host async copy is staged by the runtime before returning, so `flat` may go out of scope
  KOCR_HIP(ctx, hipMemcpyAsync(d_tab, flat.data(), flat.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  KOCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ResizeTables t;
  t.xi0 = d_tab;
  t.xi1 = t.xi0 + dw;
  t.xa0 = t.xi1 + dw;
  t.xa1 = t.xa0 + dw;
  t.yi0 = t.xa1 + dw;
  t.yi1 = t.yi0 + dh;
  t.yb0 = t.yi1 + dh;
  t.yb1 = t.yb0 + dh;
  ProfScope ps(ctx, "resize_pad", 0, 3.0 * n * ((double)sh * sw + (double)Hmax * Wmax));
  hipLaunchKernelGGL(resize_pad_kernel, dim3((Wmax + 255) / 256, Hmax, n), dim3(256), 0, ctx->stream, d_src, sh, sw,
                     (size_t)sh * sw * 3, d_dst, dh, dw, Hmax, Wmax, cval, t);
  KOCR_HIP(ctx, hipGetLastError());
  return KOCR_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// Float images (round 5): cv2.resize of a float32 image interpolates in float (tools.py:394 hands cv2 whatever dtype it
// is given): source coordinate (d + 0.5) (src / dst) - 0.5 in double, tap index = floor, weight a = float(f - floor(f)) (0
// left of the image), taps clamped to the image (replicated border), horizontal pass then vertical pass, float32
// arithmetic in exactly that order -- this file is built with -ffp-contract=off, so the result equals the numpy statement of
// the same steps (oracle/tools.py::resize_linear_float) bit for bit.  Pads to (Hmax, Wmax) with cval like resize_pad_kernel.
// ---------------------------------------------------------------------------------------------------------------------
struct ResizeTablesF {
  int* xi0;
  int* xi1;
  float* xa;
  int* yi0;
  int* yi1;
  float* ya;
};

__global__ void resize_pad_f32_kernel(const float* __restrict__ src, int sh, int sw, int C, float* __restrict__ dst, int dh, int dw,
                                      int Hmax, int Wmax, float cval, ResizeTablesF t) {
  const int n = blockIdx.z, y = blockIdx.y;
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= Wmax) return;
  float* o = dst + (((size_t)n * Hmax + y) * Wmax + x) * C;
  if (y >= dh || x >= dw) {
    for (int c = 0; c < C; ++c) o[c] = cval;
    return;
  }
  const float* im = src + (size_t)n * sh * sw * C;
  const int x0 = t.xi0[x], x1 = t.xi1[x], y0 = t.yi0[y], y1 = t.yi1[y];
  const float ax = t.xa[x], ay = t.ya[y];
  const float* r0 = im + (size_t)y0 * sw * C;
  const float* r1 = im + (size_t)y1 * sw * C;
  for (int c = 0; c < C; ++c) {
    const float h0 = r0[x0 * C + c] * (1.f - ax) + r0[x1 * C + c] * ax;
    const float h1 = r1[x0 * C + c] * (1.f - ax) + r1[x1 * C + c] * ax;
    o[c] = h0 * (1.f - ay) + h1 * ay;
  }
}

int launch_resize_pad_f32(kocr_ctx* ctx, const float* d_src, int n, int sh, int sw, int C, float* d_dst, int dh, int dw, int Hmax,
                          int Wmax, float cval, Arena& tab_arena) {
  if (n <= 0) return KOCR_OK;
  if (dh <= 0 || dw <= 0 || dh > Hmax || dw > Wmax || sh <= 0 || sw <= 0 || C <= 0) KOCR_FAIL(ctx, KOCR_EINVAL, "resize_pad_f32: bad sizes");
  std::vector<int> flat((size_t)3 * (dw + dh));
  auto taps = [&](int dst_n, int src_n, int* i0, int* i1, float* a) {
    for (int d = 0; d < dst_n; ++d) {
      const double f = (d + 0.5) * ((double)src_n / (double)dst_n) - 0.5;
      long s = (long)std::floor(f);
      float w = (float)(f - (double)s);
      if (s < 0) w = 0.f;
      const long c0 = std::min<long>(std::max<long>(s, 0), src_n - 1);
      i0[d] = (int)c0;
      i1[d] = (int)std::min<long>(c0 + 1, src_n - 1);
      a[d] = w;
    }
  };
  int* xi0 = flat.data();
  int* xi1 = xi0 + dw;
  float* xa = reinterpret_cast<float*>(xi1 + dw);
  int* yi0 = reinterpret_cast<int*>(xa + dw);
  int* yi1 = yi0 + dh;
  float* ya = reinterpret_cast<float*>(yi1 + dh);
  taps(dw, sw, xi0, xi1, xa);
  taps(dh, sh, yi0, yi1, ya);
  int* d_tab = (int*)arena_alloc(tab_arena, flat.size() * sizeof(int));
  if (!d_tab) KOCR_FAIL(ctx, KOCR_ENOMEM, "resize_pad_f32: table arena exhausted");
  KOCR_HIP(ctx, hipMemcpyAsync(d_tab, flat.data(), flat.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  KOCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ResizeTablesF t;
  t.xi0 = d_tab;
  t.xi1 = t.xi0 + dw;
  t.xa = reinterpret_cast<float*>(t.xi1 + dw);
  t.yi0 = reinterpret_cast<int*>(t.xa + dw);
  t.yi1 = t.yi0 + dh;
  t.ya = reinterpret_cast<float*>(t.yi1 + dh);
  ProfScope ps(ctx, "resize_pad_f32", 0, 4.0 * C * n * ((double)sh * sw + (double)Hmax * Wmax));
  hipLaunchKernelGGL(resize_pad_f32_kernel, dim3((Wmax + 255) / 256, Hmax, n), dim3(256), 0, ctx->stream, d_src, sh, sw, C, d_dst, dh, dw,
                     Hmax, Wmax, cval, t);
  KOCR_HIP(ctx, hipGetLastError());
  return KOCR_OK;
}
