// warp.hip — word-box crops for the recogniser.
//
// Replaces the per-box host loop of Recognizer.recognize_from_boxes (recognition.py:506-526):
// cv2.cvtColor(RGB2GRAY) of the whole image, tools.warpBox per box (tools.py:61-117:
// get_rotated_box :533-581, get_rotated_width_height :41-57, cv2.getPerspectiveTransform,
// cv2.warpPerspective, paste into a zero 31x200 canvas) and the float32 /255.
//
// Host part (this file, float64, fixed operation order, -ffp-contract=off): box ordering,
// (w,h), scale, 8x8 LU solve for the homography, 3x3 adjugate inverse.  Device part: one
// thread per crop pixel maps (x,y) through M^-1 in float64, rounds to 1/32 px (half to
// even), gathers the 4 RGB taps from the uint8 image (coalescing is bounded by the box
// orientation; the whole crop stage moves ~25 KB per word), converts each tap to gray with
// OpenCV's 15-bit integer coefficients, blends with 15-bit bilinear weights, BORDER_CONSTANT
// 0, and writes float32 gray/255 — the gray image is never materialised.
#include "common.h"
#include <algorithm>
#include <cmath>

namespace {

struct P2 {
  double x, y;
};

double cross2(P2 o, P2 a, P2 b) { return (a.x - o.x) * (b.y - o.y) - (a.y - o.y) * (b.x - o.x); }

// shapely MultiPoint.minimum_rotated_rectangle (tools.py:543-547): min-area rectangle over the
// convex hull's edges.  Returns false for degenerate input (the reference's AttributeError path).
bool min_rotated_rect(const P2* pts, P2* out) {
  P2 u[4];
  int n = 0;
  for (int i = 0; i < 4; ++i) {
    bool dup = false;
    for (int j = 0; j < n; ++j) dup |= (u[j].x == pts[i].x && u[j].y == pts[i].y);
    if (!dup) u[n++] = pts[i];
  }
  if (n < 3) return false;
  std::sort(u, u + n, [](const P2& a, const P2& b) { return a.x < b.x || (a.x == b.x && a.y < b.y); });
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
    const double ln = std::sqrt(dx * dx + dy * dy);
    const double ux = dx / ln, uy = dy / ln;
    double umin = 0, umax = 0, vmin = 0, vmax = 0;
    for (int j = 0; j < nh; ++j) {
      const double uu = hull[j].x * ux + hull[j].y * uy;
      const double vv = -hull[j].x * uy + hull[j].y * ux;
      if (j == 0) {
        umin = umax = uu;
        vmin = vmax = vv;
      } else {
        umin = std::min(umin, uu);
        umax = std::max(umax, uu);
        vmin = std::min(vmin, vv);
        vmax = std::max(vmax, vv);
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

double dist2(const float* a, const float* b) {
  const double dx = (double)a[0] - (double)b[0], dy = (double)a[1] - (double)b[1];
  return std::sqrt(dx * dx + dy * dy);
}

// Gaussian elimination with partial pivoting, float64 (cv2.getPerspectiveTransform's solve).
bool solve8(double A[8][8], double b[8], double x[8]) {
  const int n = 8;
  for (int col = 0; col < n; ++col) {
    int piv = col;
    for (int r = col + 1; r < n; ++r)
      if (std::fabs(A[r][col]) > std::fabs(A[piv][col])) piv = r;
    if (A[piv][col] == 0.0) return false;
    if (piv != col) {
      for (int c = 0; c < n; ++c) std::swap(A[piv][c], A[col][c]);
      std::swap(b[piv], b[col]);
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

void invert3(const double m[9], double t[9]) {
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

}  // namespace

// tools.warpBox's scalar half (tools.py:86-107).  box: 4x2 float32.  rc: 0 ok, 1 zero width/height
// (the reference raises ZeroDivisionError), 2 singular system.
int warp_prepare(const float* box, int target_h, int target_w, WarpParam* out, float* ordered_box) {
  // ---- get_rotated_box (tools.py:533-581) ----
  P2 in[4], pts[4];
  for (int i = 0; i < 4; ++i) in[i] = {(double)box[2 * i], (double)box[2 * i + 1]};
  if (!min_rotated_rect(in, pts))
    for (int i = 0; i < 4; ++i) pts[i] = in[i];
  int idx[4] = {0, 1, 2, 3};
  std::stable_sort(idx, idx + 4, [&](int a, int b) { return pts[a].x < pts[b].x; });
  P2 l0 = pts[idx[0]], l1 = pts[idx[1]], r0 = pts[idx[2]], r1 = pts[idx[3]];
  P2 tl, bl, tr, br;
  if (l1.y < l0.y) {
    tl = l1;
    bl = l0;
  } else {
    tl = l0;
    bl = l1;
  }
  const double d0 = std::sqrt((tl.x - r0.x) * (tl.x - r0.x) + (tl.y - r0.y) * (tl.y - r0.y));
  const double d1 = std::sqrt((tl.x - r1.x) * (tl.x - r1.x) + (tl.y - r1.y) * (tl.y - r1.y));
  if (d0 > d1) {
    br = r0;
    tr = r1;
  } else {
    br = r1;
    tr = r0;
  }
  float ob[8] = {(float)tl.x, (float)tl.y, (float)tr.x, (float)tr.y,
                 (float)br.x, (float)br.y, (float)bl.x, (float)bl.y};
  if (ordered_box)
    for (int i = 0; i < 8; ++i) ordered_box[i] = ob[i];
  // ---- get_rotated_width_height (tools.py:41-57) ----
  const int w = (int)((dist2(ob + 0, ob + 2) + dist2(ob + 4, ob + 6)) / 2);
  const int h = (int)((dist2(ob + 0, ob + 6) + dist2(ob + 2, ob + 4)) / 2);
  if (w == 0 || h == 0) return 1;
  // ---- scale, destination quad, homography (tools.py:95-106) ----
  const double scale = std::min((double)target_w / (double)w, (double)target_h / (double)h);
  const float dst[8] = {0.f, 0.f, (float)(scale * w), 0.f, (float)(scale * w), (float)(scale * h),
                        0.f, (float)(scale * h)};
  double A[8][8], b[8], x[8];
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 8; ++j) A[i][j] = 0.0;
  for (int i = 0; i < 4; ++i) {
    const double sx = ob[2 * i], sy = ob[2 * i + 1], dx = dst[2 * i], dy = dst[2 * i + 1];
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
  if (!solve8(A, b, x)) return 2;
  const double M[9] = {x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7], 1.0};
  invert3(M, out->mi);
  out->cw = std::min((int)(scale * w), target_w);
  out->ch = std::min((int)(scale * h), target_h);
  out->pad = 0;
  return 0;
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
