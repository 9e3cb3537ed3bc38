// warp.hip — word-box crops for the recogniser.
//
// Replaces the per-box host loop of Recognizer.recognize_from_boxes (recognition.py:506-526):
// cv2.cvtColor(RGB2GRAY) of the whole image, tools.warpBox per box (tools.py:61-117:
// get_rotated_box :533-581, get_rotated_width_height :41-57, cv2.getPerspectiveTransform,
// cv2.warpPerspective, paste into a zero 31x200 canvas) and the float32 /255.
//
// Set-up (float64, fixed operation order, -ffp-contract=off; the SAME functions compile for host and device, and
// the pipeline runs them ON THE DEVICE, one thread per box, straight from the box buffer getBoxes filled -- no
// host round trip between boxes and crops): box ordering, (w,h), scale, 8x8 LU solve for the homography, 3x3
// adjugate inverse.  Warp: one thread per crop pixel maps (x,y) through M^-1 in float64, rounds to 1/32 px (half
// to even), gathers the 4 RGB taps from the uint8 image (coalescing is bounded by the box
// orientation; the whole crop stage moves ~25 KB per word), converts each tap to gray with
// OpenCV's 15-bit integer coefficients, blends with 15-bit bilinear weights, BORDER_CONSTANT
// 0, and writes float32 gray/255 — the gray image is never materialised.
#include "common.h"
#include <algorithm>
#include <cmath>

#define HD __host__ __device__

namespace {

struct P2 {
  double x, y;
};

HD double cross2(P2 o, P2 a, P2 b) { return (a.x - o.x) * (b.y - o.y) - (a.y - o.y) * (b.x - o.x); }

// shapely MultiPoint.minimum_rotated_rectangle (tools.py:543-547): min-area rectangle over the
// convex hull's edges.  Returns false for degenerate input (the reference's AttributeError path).
HD bool min_rotated_rect(const P2* pts, P2* out) {
  P2 u[4];
  int n = 0;
  for (int i = 0; i < 4; ++i) {
    bool dup = false;
    for (int j = 0; j < n; ++j) dup |= (u[j].x == pts[i].x && u[j].y == pts[i].y);
    if (!dup) u[n++] = pts[i];
  }
  if (n < 3) return false;
  for (int i = 1; i < n; ++i)  // insertion sort by (x, y): at most 4 distinct points
    for (int j = i; j > 0 && (u[j].x < u[j - 1].x || (u[j].x == u[j - 1].x && u[j].y < u[j - 1].y)); --j) {
      const P2 t = u[j];
      u[j] = u[j - 1];
      u[j - 1] = t;
    }
  P2 lower[8], upper[8];
  int nl = 0, nu = 0;
  for (int i = 0; i < n; ++i) {
    while (nl >= 2 && cross2(lower[nl - 2], lower[nl - 1], u[i]) <= 0) --nl;
    lower[nl++] = u[i];
  }
  for (int i = n - 1; i >= 0; --i) {
    while (nu >= 2 && cross2(upper[nu - 2], upper[nu - 1], u[i]) <= 0) --nu;
    upper[nu++] = u[i];
  }
  P2 hull[8];
  int nh = 0;
  for (int i = 0; i + 1 < nl; ++i) hull[nh++] = lower[i];
  for (int i = 0; i + 1 < nu; ++i) hull[nh++] = upper[i];
  if (nh < 3) return false;
  bool have = false;
  double barea = 0, bux = 0, buy = 0, bumin = 0, bumax = 0, bvmin = 0, bvmax = 0;
  for (int i = 0; i < nh; ++i) {
    const P2 p0 = hull[i], p1 = hull[(i + 1) % nh];
    const double dx = p1.x - p0.x, dy = p1.y - p0.y;
    const double ln = sqrt(dx * dx + dy * dy);
    const double ux = dx / ln, uy = dy / ln;
    double umin = 0, umax = 0, vmin = 0, vmax = 0;
    for (int j = 0; j < nh; ++j) {
      const double uu = hull[j].x * ux + hull[j].y * uy;
      const double vv = -hull[j].x * uy + hull[j].y * ux;
      if (j == 0) {
        umin = umax = uu;
        vmin = vmax = vv;
      } else {
        umin = uu < umin ? uu : umin;
        umax = uu > umax ? uu : umax;
        vmin = vv < vmin ? vv : vmin;
        vmax = vv > vmax ? vv : vmax;
      }
    }
    const double area = (umax - umin) * (vmax - vmin);
    if (!have || area < barea) {
      have = true;
      barea = area;
      bux = ux;
      buy = uy;
      bumin = umin;
      bumax = umax;
      bvmin = vmin;
      bvmax = vmax;
    }
  }
  const double us[4] = {bumin, bumax, bumax, bumin}, vs[4] = {bvmin, bvmin, bvmax, bvmax};
  for (int i = 0; i < 4; ++i) {
    out[i].x = us[i] * bux - vs[i] * buy;
    out[i].y = us[i] * buy + vs[i] * bux;
  }
  return true;
}

HD double dist2(const float* a, const float* b) {
  const double dx = (double)a[0] - (double)b[0], dy = (double)a[1] - (double)b[1];
  return sqrt(dx * dx + dy * dy);
}

// Gaussian elimination with partial pivoting, float64 (cv2.getPerspectiveTransform's solve).
HD bool solve8(double A[8][8], double b[8], double x[8]) {
  const int n = 8;
  for (int col = 0; col < n; ++col) {
    int piv = col;
    for (int r = col + 1; r < n; ++r)
      if (fabs(A[r][col]) > fabs(A[piv][col])) piv = r;
    if (A[piv][col] == 0.0) return false;
    if (piv != col) {
      for (int c = 0; c < n; ++c) {
        const double t = A[piv][c];
        A[piv][c] = A[col][c];
        A[col][c] = t;
      }
      const double tb = b[piv];
      b[piv] = b[col];
      b[col] = tb;
    }
    for (int r = col + 1; r < n; ++r) {
      const double f = A[r][col] / A[col][col];
      if (f != 0.0) {
        for (int c = col; c < n; ++c) A[r][c] = A[r][c] - f * A[col][c];
        b[r] = b[r] - f * b[col];
      }
    }
  }
  for (int r = n - 1; r >= 0; --r) {
    double s = b[r];
    for (int c = r + 1; c < n; ++c) s = s - A[r][c] * x[c];
    x[r] = s / A[r][r];
  }
  return true;
}

HD void invert3(const double m[9], double t[9]) {
  double d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
             m[2] * (m[3] * m[7] - m[4] * m[6]);
  if (d == 0.0) {
    for (int i = 0; i < 9; ++i) t[i] = 0.0;
    return;
  }
  d = 1.0 / d;
  t[0] = (m[4] * m[8] - m[5] * m[7]) * d;
  t[1] = (m[2] * m[7] - m[1] * m[8]) * d;
  t[2] = (m[1] * m[5] - m[2] * m[4]) * d;
  t[3] = (m[5] * m[6] - m[3] * m[8]) * d;
  t[4] = (m[0] * m[8] - m[2] * m[6]) * d;
  t[5] = (m[2] * m[3] - m[0] * m[5]) * d;
  t[6] = (m[3] * m[7] - m[4] * m[6]) * d;
  t[7] = (m[1] * m[6] - m[0] * m[7]) * d;
  t[8] = (m[0] * m[4] - m[1] * m[3]) * d;
}

// cv2.getPerspectiveTransform(src, dst) (tools.py:96-106) and its inverse.  src / dst: 4x2 float32.  m_fwd may be null.
HD bool quad_homography(const float* src, const float* dst, double* m_fwd, double* m_inv) {
  double A[8][8], b[8], x[8];
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 8; ++j) A[i][j] = 0.0;
  for (int i = 0; i < 4; ++i) {
    const double sx = src[2 * i], sy = src[2 * i + 1], dx = dst[2 * i], dy = dst[2 * i + 1];
    A[i][0] = A[i + 4][3] = sx;
    A[i][1] = A[i + 4][4] = sy;
    A[i][2] = A[i + 4][5] = 1.0;
    A[i][6] = -sx * dx;
    A[i][7] = -sy * dx;
    A[i + 4][6] = -sx * dy;
    A[i + 4][7] = -sy * dy;
    b[i] = dx;
    b[i + 4] = dy;
  }
  if (!solve8(A, b, x)) return false;
  const double M[9] = {x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7], 1.0};
  if (m_fwd)
    for (int i = 0; i < 9; ++i) m_fwd[i] = M[i];
  invert3(M, m_inv);
  return true;
}

// tools.get_rotated_box (tools.py:533-581): 4 points -> [tl, tr, br, bl] float32
HD void rotated_box(const float* box, float* ob) {
  P2 in[4], pts[4];
  for (int i = 0; i < 4; ++i) in[i] = {(double)box[2 * i], (double)box[2 * i + 1]};
  if (!min_rotated_rect(in, pts))
    for (int i = 0; i < 4; ++i) pts[i] = in[i];
  int idx[4] = {0, 1, 2, 3};
  for (int i = 1; i < 4; ++i)  // stable insertion sort by x (np.argsort(kind="stable"))
    for (int j = i; j > 0 && pts[idx[j]].x < pts[idx[j - 1]].x; --j) {
      const int t = idx[j];
      idx[j] = idx[j - 1];
      idx[j - 1] = t;
    }
  const P2 l0 = pts[idx[0]], l1 = pts[idx[1]], r0 = pts[idx[2]], r1 = pts[idx[3]];
  P2 tl, bl, tr, br;
  if (l1.y < l0.y) {
    tl = l1;
    bl = l0;
  } else {
    tl = l0;
    bl = l1;
  }
  const double d0 = sqrt((tl.x - r0.x) * (tl.x - r0.x) + (tl.y - r0.y) * (tl.y - r0.y));
  const double d1 = sqrt((tl.x - r1.x) * (tl.x - r1.x) + (tl.y - r1.y) * (tl.y - r1.y));
  if (d0 > d1) {
    br = r0;
    tr = r1;
  } else {
    br = r1;
    tr = r0;
  }
  ob[0] = (float)tl.x;
  ob[1] = (float)tl.y;
  ob[2] = (float)tr.x;
  ob[3] = (float)tr.y;
  ob[4] = (float)br.x;
  ob[5] = (float)br.y;
  ob[6] = (float)bl.x;
  ob[7] = (float)bl.y;
}

// tools.warpBox's scalar half (tools.py:86-107), margin 0.  box: 4x2 float32.  rc: 0 ok, 1 zero width/height
// (the reference raises ZeroDivisionError), 2 singular system.
HD int warp_prepare_hd(const float* box, int target_h, int target_w, WarpParam* out, float* ordered_box) {
  float ob[8];
  rotated_box(box, ob);
  if (ordered_box)
    for (int i = 0; i < 8; ++i) ordered_box[i] = ob[i];
  // ---- get_rotated_width_height (tools.py:41-57) ----
  const int w = (int)((dist2(ob + 0, ob + 2) + dist2(ob + 4, ob + 6)) / 2);
  const int h = (int)((dist2(ob + 0, ob + 6) + dist2(ob + 2, ob + 4)) / 2);
  if (w == 0 || h == 0) return 1;
  // ---- scale, destination quad, homography (tools.py:95-106) ----
  const double sw = (double)target_w / (double)w, sh = (double)target_h / (double)h;
  const double scale = sw < sh ? sw : sh;
  const float dst[8] = {0.f, 0.f, (float)(scale * w), 0.f, (float)(scale * w), (float)(scale * h),
                        0.f, (float)(scale * h)};
  if (!quad_homography(ob, dst, nullptr, out->mi)) return 2;
  const int cw = (int)(scale * w), ch = (int)(scale * h);
  out->cw = cw < target_w ? cw : target_w;
  out->ch = ch < target_h ? ch : target_h;
  out->pad = 0;
  return 0;
}

}  // namespace

int warp_prepare(const float* box, int target_h, int target_w, WarpParam* out, float* ordered_box) {
  return warp_prepare_hd(box, target_h, target_w, out, ordered_box);
}

// One thread per box slot (image k, slot b < cap): boxes[k][b] -> prm[offset(k) + b], where offset(k) = number of
// boxes of the images before k.  status: atomicMax of the per-box return code (0 ok, 1 zero size, 2 singular).
__global__ void warp_prepare_kernel(const float* __restrict__ boxes, const int* __restrict__ counts, int N, int cap,
                                    int th, int tw, WarpParam* __restrict__ prm, int* __restrict__ status) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * cap) return;
  const int k = i / cap, b = i - k * cap;
  if (b >= counts[k]) return;
  long off = 0;
  for (int j = 0; j < k; ++j) off += counts[j];
  WarpParam p;
  const int rc = warp_prepare_hd(boxes + (size_t)i * 8, th, tw, &p, nullptr);
  if (rc != 0) {
    // keep the slot harmless: an empty crop
    for (int q = 0; q < 9; ++q) p.mi[q] = 0.0;
    p.cw = p.ch = 0;
    p.pad = 0;
    atomicMax(status, rc);
  }
  p.img = k;
  prm[off + b] = p;
}

int launch_warp_prepare(kocr_ctx* ctx, const float* d_boxes, const int* d_counts, int N, int cap, int th, int tw,
                        WarpParam* d_prm, int* d_status) {
  if (N <= 0 || cap <= 0) return KOCR_OK;
  ProfScope ps(ctx, "warp_prepare", 0, 0);
  const int n = N * cap;
  hipLaunchKernelGGL(warp_prepare_kernel, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, d_boxes, d_counts, N, cap, th, tw,
                     d_prm, d_status);
  KOCR_HIP(ctx, hipGetLastError());
  return KOCR_OK;
}

// General quads (tools.warpBox with margin / skip_rotate / target size from the box, tools.py:61-117): the caller
// supplies the ordered source quad and the destination quad of every crop; the device solves the homographies.
__global__ void warp_quads_kernel(const float* __restrict__ src, const float* __restrict__ dst, const int* __restrict__ img,
                                  const int* __restrict__ cw, const int* __restrict__ ch, int M, WarpParam* __restrict__ prm,
                                  double* __restrict__ m_fwd, int* __restrict__ status) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  WarpParam p;
  double fwd[9];
  if (!quad_homography(src + (size_t)m * 8, dst + (size_t)m * 8, fwd, p.mi)) {
    for (int q = 0; q < 9; ++q) p.mi[q] = fwd[q] = 0.0;
    atomicMax(status, 2);
  }
  p.img = img[m];
  p.cw = cw[m];
  p.ch = ch[m];
  p.pad = 0;
  prm[m] = p;
  if (m_fwd)
    for (int q = 0; q < 9; ++q) m_fwd[(size_t)m * 9 + q] = fwd[q];
}

int launch_warp_quads(kocr_ctx* ctx, const float* d_src, const float* d_dst, const int* d_img, const int* d_cw,
                      const int* d_ch, int M, WarpParam* d_prm, double* d_mfwd, int* d_status) {
  if (M <= 0) return KOCR_OK;
  hipLaunchKernelGGL(warp_quads_kernel, dim3((M + 63) / 64), dim3(64), 0, ctx->stream, d_src, d_dst, d_img, d_cw, d_ch, M,
                     d_prm, d_mfwd, d_status);
  KOCR_HIP(ctx, hipGetLastError());
  return KOCR_OK;
}

__global__ void warp_kernel(const uint8_t* __restrict__ img, int H, int W, const WarpParam* __restrict__ prm,
                            int th, int tw, float* __restrict__ crops) {
  const int m = blockIdx.y;
  const WarpParam p = prm[m];
  const uint8_t* im = img + (size_t)p.img * H * W * 3;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < th * tw; i += gridDim.x * blockDim.x) {
    const int y = i / tw, x = i - y * tw;
    float v = 0.f;
    if (x < p.cw && y < p.ch) {
      const double xd = (double)x, yd = (double)y;
      const double X0 = (p.mi[0] * xd + p.mi[1] * yd) + p.mi[2];
      const double Y0 = (p.mi[3] * xd + p.mi[4] * yd) + p.mi[5];
      const double W0 = (p.mi[6] * xd + p.mi[7] * yd) + p.mi[8];
      const double Wi = W0 != 0.0 ? 32.0 / W0 : 0.0;
      const double fX = fmax(-2147483648.0, fmin(2147483647.0, X0 * Wi));
      const double fY = fmax(-2147483648.0, fmin(2147483647.0, Y0 * Wi));
      const long X = (long)rint(fX), Y = (long)rint(fY);  // saturate_cast<int>: half to even
      const long sx = X >> 5, sy = Y >> 5;
      const int ax = (int)(X & 31), ay = (int)(Y & 31);
      auto tap = [&](long yy, long xx) -> int {
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) return 0;
        const uint8_t* q = im + ((size_t)yy * W + xx) * 3;
        // cv2.cvtColor(RGB2GRAY), uint8: 15-bit fixed point          (recognition.py:507-510)
        return (q[0] * 9798 + q[1] * 19235 + q[2] * 3735 + (1 << 14)) >> 15;
      };
      const int w00 = (32 - ax) * (32 - ay) * 32, w01 = ax * (32 - ay) * 32;
      const int w10 = (32 - ax) * ay * 32, w11 = ax * ay * 32;
      const int acc = w00 * tap(sy, sx) + w01 * tap(sy, sx + 1) + w10 * tap(sy + 1, sx) + w11 * tap(sy + 1, sx + 1);
      v = (float)((acc + (1 << 14)) >> 15) / 255.0f;  // recognition.py:524
    }
    crops[(size_t)m * th * tw + i] = v;
  }
}

int launch_warp(kocr_ctx* ctx, const uint8_t* d_img, int H, int W, const WarpParam* d_prm, int M, int th,
                int tw, float* d_crops) {
  if (M <= 0) return KOCR_OK;
  ProfScope ps(ctx, "warp_crops", 0, (double)M * th * tw * (4.0 + 12.0));
  const int bx = (th * tw + 255) / 256;
  for (int s = 0; s < M; s += 65535) {
    const int mb = std::min(65535, M - s);
    hipLaunchKernelGGL(warp_kernel, dim3(bx, mb), dim3(256), 0, ctx->stream, d_img, H, W, d_prm + s, th, tw,
                       d_crops + (size_t)s * th * tw);
  }
  KOCR_HIP(ctx, hipGetLastError());
  return KOCR_OK;
}


// Float images (round 5): cvtColor(RGB2GRAY) + warpPerspective of a float32 image work in float (recognition.py:507-526 hands
// cv2 the image's own type): gray = (0.299 R + 0.587 G) + 0.114 B per tap in float32, the same 1/32-pixel source coordinates
// as the uint8 kernel (OpenCV's remap tables), FLOAT weights a / 32, four products summed in the order
// t00 w00 + t01 w01 + t10 w10 + t11 w11, constant-0 border; NO division by 255 (the caller's, recognition.py:524).
// C = 3 (RGB) or 1 (already gray).  -ffp-contract=off: equals oracle/tools.py::warp_box_float's arithmetic.
__global__ void warp_f32_kernel(const float* __restrict__ img, int H, int W, int C, const WarpParam* __restrict__ prm, int th, int tw,
                                float* __restrict__ crops) {
  const int m = blockIdx.y;
  const WarpParam p = prm[m];
  const float* im = img + (size_t)p.img * H * W * C;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < th * tw; i += gridDim.x * blockDim.x) {
    const int y = i / tw, x = i - y * tw;
    float v = 0.f;
    if (x < p.cw && y < p.ch) {
      const double xd = (double)x, yd = (double)y;
      const double X0 = (p.mi[0] * xd + p.mi[1] * yd) + p.mi[2];
      const double Y0 = (p.mi[3] * xd + p.mi[4] * yd) + p.mi[5];
      const double W0 = (p.mi[6] * xd + p.mi[7] * yd) + p.mi[8];
      const double Wi = W0 != 0.0 ? 32.0 / W0 : 0.0;
      const double fX = fmax(-2147483648.0, fmin(2147483647.0, X0 * Wi));
      const double fY = fmax(-2147483648.0, fmin(2147483647.0, Y0 * Wi));
      const long X = (long)rint(fX), Y = (long)rint(fY);
      const long sx = X >> 5, sy = Y >> 5;
      const float ax = (float)((double)(X & 31) / 32.0), ay = (float)((double)(Y & 31) / 32.0);
      auto tap = [&](long yy, long xx) -> float {
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) return 0.f;
        const float* q = im + ((size_t)yy * W + xx) * C;
        return C == 3 ? (q[0] * 0.299f + q[1] * 0.587f) + q[2] * 0.114f : q[0];
      };
      const float one = 1.f;
      v = ((tap(sy, sx) * ((one - ax) * (one - ay)) + tap(sy, sx + 1) * (ax * (one - ay))) + tap(sy + 1, sx) * ((one - ax) * ay)) +
          tap(sy + 1, sx + 1) * (ax * ay);
    }
    crops[(size_t)m * th * tw + i] = v;
  }
}

int launch_warp_f32(kocr_ctx* ctx, const float* d_img, int H, int W, int C, const WarpParam* d_prm, int M, int th, int tw, float* d_crops) {
  if (M <= 0) return KOCR_OK;
  ProfScope ps(ctx, "warp_crops_f32", 0, (double)M * th * tw * (4.0 + 16.0 * C));
  const int bx = (th * tw + 255) / 256;
  for (int s = 0; s < M; s += 65535) {
    const int mb = std::min(65535, M - s);
    hipLaunchKernelGGL(warp_f32_kernel, dim3(bx, mb), dim3(256), 0, ctx->stream, d_img, H, W, C, d_prm + s, th, tw,
                       d_crops + (size_t)s * th * tw);
  }
  KOCR_HIP(ctx, hipGetLastError());
  return KOCR_OK;
}
