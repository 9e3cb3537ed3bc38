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
