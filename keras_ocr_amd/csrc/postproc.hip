// postproc.hip — heat-map -> word boxes on the GPU.
//
// Replaces keras_ocr.detection.getBoxes (detection.py:207-287): the per-image / per-component
// host loop of cv2.threshold, connectedComponentsWithStats, dilate, findContours,
// minAreaRect, boxPoints.  All N images of a batch are processed together:
//
//   K1  threshold text/link (strict >), init union-find labels            (detection.py:221-229)
//   K2  4-connectivity union-find merge, K3 flatten: label = raster-first pixel of the component
//   K4  per-component stats: area, bbox, max(textmap)                     (:233-241, 247-255)
//   K5  filter (area >= size_threshold, max >= detection_threshold) + per-image ordered
//       compaction -> slot = rank in raster order of first pixel == OpenCV label order
//   K6  niter / dilation ROI per kept component (:258-260), canvas + row offsets
//   K7  segmap on the ROI canvas: component minus (text AND link)         (:244-246)
//   K8/9 separable (1+niter)^2 RECT dilation, anchor k/2, ROI-isolated    (:261-264)
//   K10 8-connectivity union-find on the canvas, K11 pick the fragment findContours lists
//       first (max raster-first pixel), K12 per-row extents of that fragment
//   K13 convex hull (monotone chain over row extents), exact-integer min-area rectangle over
//       hull edges, diamond test, clockwise roll, x2                      (:273-285)
//
// HBM-bound integer/byte work: one pass over the heat-map (8 B/px algorithmic), everything
// after K5 touches only the kept components' ROIs.  Compiled with -ffp-contract=off: the
// float32/float64 geometry is compared bit-for-bit with the CPU oracle.
#include "common.h"
#include <climits>

namespace {

constexpr int SCAN_BLOCK = 1024;  // pixels per compaction block

struct CompInfo {
  int root;   // global pixel index of the component's raster-first pixel
  int img;    // image index
  int slot;   // rank of the component inside its image (OpenCV label order)
  int sx, sy, rw, rh;  // dilation ROI (image coords, size)
  int k, a;   // RECT kernel size 1+niter, anchor k/2
  int canvas_off;  // first canvas pixel
  int row_off;     // first canvas row
};

struct PPArgs {
  const float* heat;
  int N, h, w;
  float det_thr, text_thr, link_thr;
  int size_thr;
  unsigned char* flags;
  int* label;
  int* area;
  int* minx;
  int* maxx;
  int* miny;
  int* maxy;
  int* tmax;
  int* blk_cnt;   // [N][bpi]
  int* blk_off;   // [N][bpi]
  int* counts;    // [N] kept components per image (may exceed cap)
  int* comp_root; // [N][cap]
  int bpi, cap;
  CompInfo* info; // [N*cap]
  int* img_base;  // [N+1] first component index of each image
  int* totals;    // [4]: total canvas px, total rows, n_empty, n_comps
};

__device__ __forceinline__ int ld_relaxed(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ int uf_find(const int* L, int i) {
  while (true) {
    const int p = ld_relaxed(L + i);
    if (p == i) return i;
    i = p;
  }
}

__device__ __forceinline__ void uf_union(int* L, int a, int b) {
  while (true) {
    a = uf_find(L, a);
    b = uf_find(L, b);
    if (a == b) return;
    if (a < b) {
      const int t = a;
      a = b;
      b = t;
    }
    const int old = atomicMin(&L[a], b);
    if (old == a) return;
    a = old;
  }
}

__device__ __forceinline__ int float_key(float f) {
  const int b = __float_as_int(f);
  return b >= 0 ? b : b ^ 0x7fffffff;
}

// ---- K1 -------------------------------------------------------------------------------
__global__ void k_threshold(PPArgs p) {
  const size_t NP = (size_t)p.N * p.h * p.w;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < NP; i += (size_t)gridDim.x * blockDim.x) {
    const float2 v = reinterpret_cast<const float2*>(p.heat)[i];
    const bool ts = v.x > p.text_thr, ls = v.y > p.link_thr;
    const bool fg = ts || ls;
    p.flags[i] = (unsigned char)((fg ? 1 : 0) | ((ts && ls) ? 2 : 0));
    p.label[i] = fg ? (int)i : -1;
    if (fg) {
      p.area[i] = 0;
      p.minx[i] = INT_MAX;
      p.maxx[i] = -1;
      p.miny[i] = INT_MAX;
      p.maxy[i] = -1;
      p.tmax[i] = INT_MIN;
    }
  }
}

// ---- K2: 4-connectivity merge -----------------------------------------------------------
__global__ void k_merge4(PPArgs p) {
  const size_t NP = (size_t)p.N * p.h * p.w;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < NP; i += (size_t)gridDim.x * blockDim.x) {
    if (!(p.flags[i] & 1)) continue;
    const int x = (int)(i % p.w);
    const int y = (int)((i / p.w) % p.h);
    if (x > 0 && (p.flags[i - 1] & 1)) uf_union(p.label, (int)i, (int)i - 1);
    if (y > 0 && (p.flags[i - p.w] & 1)) uf_union(p.label, (int)i, (int)i - p.w);
  }
}

// ---- K3+K4: flatten + stats -------------------------------------------------------------
__global__ void k_flatten_stats(PPArgs p) {
  const size_t NP = (size_t)p.N * p.h * p.w;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < NP; i += (size_t)gridDim.x * blockDim.x) {
    if (!(p.flags[i] & 1)) continue;
    const int r = uf_find(p.label, (int)i);
    p.label[i] = r;  // only ever replaces a value by an ancestor: safe against concurrent finds
    const int x = (int)(i % p.w);
    const int y = (int)((i / p.w) % p.h);
    atomicAdd(&p.area[r], 1);
    atomicMin(&p.minx[r], x);
    atomicMax(&p.maxx[r], x);
    atomicMin(&p.miny[r], y);
    atomicMax(&p.maxy[r], y);
    atomicMax(&p.tmax[r], float_key(p.heat[2 * i]));
  }
}

__device__ __forceinline__ bool comp_kept(const PPArgs& p, size_t i) {
  if (!(p.flags[i] & 1) || p.label[i] != (int)i) return false;
  if (p.area[i] < p.size_thr) return false;                 // detection.py:233-236
  if (p.tmax[i] < float_key(p.det_thr)) return false;       // detection.py:240-241
  return true;
}

// ---- K5a: per-block kept-root counts ------------------------------------------------------
__global__ void k_count(PPArgs p) {
  const int img = blockIdx.y, blk = blockIdx.x;
  const int hw = p.h * p.w;
  __shared__ int s_cnt;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  int c = 0;
  for (int j = threadIdx.x; j < SCAN_BLOCK; j += blockDim.x) {
    const int px = blk * SCAN_BLOCK + j;
    if (px < hw && comp_kept(p, (size_t)img * hw + px)) ++c;
  }
  if (c) atomicAdd(&s_cnt, c);
  __syncthreads();
  if (threadIdx.x == 0) p.blk_cnt[img * p.bpi + blk] = s_cnt;
}

// ---- K5b: per-image exclusive scan of block counts (one workgroup per image) --------------
__global__ void k_scan_blocks(PPArgs p) {
  const int img = blockIdx.x;
  __shared__ int s_part[256];
  __shared__ int s_base;
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  for (int start = 0; start < p.bpi; start += 256) {
    const int j = start + threadIdx.x;
    const int v = j < p.bpi ? p.blk_cnt[img * p.bpi + j] : 0;
    s_part[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {  // Hillis-Steele inclusive scan
      const int t = threadIdx.x >= d ? s_part[threadIdx.x - d] : 0;
      __syncthreads();
      s_part[threadIdx.x] += t;
      __syncthreads();
    }
    if (j < p.bpi) p.blk_off[img * p.bpi + j] = s_base + s_part[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 0) s_base += s_part[255];
    __syncthreads();
  }
  if (threadIdx.x == 0) p.counts[img] = s_base;
}

// ---- K5c: ordered slot assignment ---------------------------------------------------------
__global__ void k_assign(PPArgs p) {
  const int img = blockIdx.y, blk = blockIdx.x;
  const int hw = p.h * p.w;
  if (p.blk_cnt[img * p.bpi + blk] == 0) return;
  // one wave walks the block's pixels in raster order, 64 at a time
  int base = p.blk_off[img * p.bpi + blk];
  const int lane = threadIdx.x;  // blockDim.x == 64
  for (int j0 = 0; j0 < SCAN_BLOCK; j0 += 64) {
    const int px = blk * SCAN_BLOCK + j0 + lane;
    const bool kept = px < hw && comp_kept(p, (size_t)img * hw + px);
    const unsigned long long m = __ballot(kept);
    if (kept) {
      const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
      if (slot < p.cap) p.comp_root[img * p.cap + slot] = img * hw + px;
    }
    base += __popcll(m);
  }
}

// ---- K6: ROI geometry + canvas offsets (one workgroup, chunked block scan) ------------------
__global__ void k_geometry(PPArgs p) {
  __shared__ long s_can[256];
  __shared__ int s_row[256];
  __shared__ long s_cbase;
  __shared__ int s_rbase, s_nc;
  const int tid = threadIdx.x;
  if (tid == 0) {
    int nc = 0;
    for (int img = 0; img < p.N; ++img) {
      p.img_base[img] = nc;
      nc += min(p.counts[img], p.cap);
    }
    p.img_base[p.N] = nc;
    s_nc = nc;
    s_cbase = 0;
    s_rbase = 0;
  }
  __syncthreads();
  const int nc = s_nc;
  for (int start = 0; start < nc; start += 256) {
    const int c = start + tid;
    CompInfo ci;
    long csz = 0;
    int rsz = 0;
    if (c < nc) {
      int lo = 0, hi = p.N - 1;  // image of component c
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (p.img_base[mid] <= c)
          lo = mid;
        else
          hi = mid - 1;
      }
      const int img = lo, slot = c - p.img_base[img];
      const int root = p.comp_root[img * p.cap + slot];
      const int x = p.minx[root], y = p.miny[root];
      const int w = p.maxx[root] - x + 1, h = p.maxy[root] - y + 1;
      const int size = p.area[root];
      // niter = int(sqrt(size * min(w, h) / (w * h)) * 2)            (detection.py:258)
      const int niter = (int)(sqrt((double)((long)size * min(w, h)) / (double)((long)w * h)) * 2.0);
      ci.root = root;
      ci.img = img;
      ci.slot = slot;
      ci.sx = max(x - niter, 0);
      ci.sy = max(y - niter, 0);
      const int ex = min(x + w + niter + 1, p.w), ey = min(y + h + niter + 1, p.h);
      ci.rw = ex - ci.sx;
      ci.rh = ey - ci.sy;
      ci.k = 1 + niter;
      ci.a = ci.k / 2;
      csz = (long)ci.rw * ci.rh;
      rsz = ci.rh;
    }
    s_can[tid] = csz;
    s_row[tid] = rsz;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
      const long tc = tid >= d ? s_can[tid - d] : 0;
      const int tr = tid >= d ? s_row[tid - d] : 0;
      __syncthreads();
      s_can[tid] += tc;
      s_row[tid] += tr;
      __syncthreads();
    }
    if (c < nc) {
      const long off = s_cbase + s_can[tid] - csz;
      ci.canvas_off = off > 0x7fff0000L ? 0x7fff0000 : (int)off;
      ci.row_off = s_rbase + s_row[tid] - rsz;
      p.info[c] = ci;
    }
    __syncthreads();
    if (tid == 0) {
      s_cbase += s_can[255];
      s_rbase += s_row[255];
    }
    __syncthreads();
  }
  if (tid == 0) {
    p.totals[0] = s_cbase > 0x7fff0000L ? -1 : (int)s_cbase;
    p.totals[1] = s_rbase;
    p.totals[2] = 0;
    p.totals[3] = nc;
  }
}

// ---- canvas helpers ---------------------------------------------------------------------------
struct CanvasArgs {
  const CompInfo* info;
  int ncomp, total, total_rows;
  const unsigned char* flags;
  const int* label;
  int h, w;
  unsigned char* seg0;
  unsigned char* seg1;
  int* clabel;
  int* sel;       // [ncomp] chosen fragment root (canvas index), -1 = empty
  int* rowmin;    // [total_rows] local x
  int* rowmax;
  int2* hullbuf;  // scratch, 2 * (2*rows+4) per component region
  float* boxes;   // [N][cap][4][2]
  int cap;
  const int* counts;
  int* totals;
};

__device__ __forceinline__ int find_comp(const CompInfo* info, int n, int p) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (info[mid].canvas_off <= p)
      lo = mid;
    else
      hi = mid - 1;
  }
  return lo;
}

__global__ void k_canvas_fill(CanvasArgs a) {
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < a.total; p += gridDim.x * blockDim.x) {
    const int c = find_comp(a.info, a.ncomp, p);
    const CompInfo ci = a.info[c];
    const int q = p - ci.canvas_off;
    const int ly = q / ci.rw, lx = q - ly * ci.rw;
    const size_t pix = ((size_t)ci.img * a.h + (ci.sy + ly)) * a.w + (ci.sx + lx);
    const unsigned char f = a.flags[pix];
    a.seg0[p] = ((f & 1) && !(f & 2) && a.label[pix] == ci.root) ? 1 : 0;
  }
}

// out(x) = OR_{j in [0,k)} in(x + j - a), clipped to the ROI
__global__ void k_dilate_h(CanvasArgs a) {
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < a.total; p += gridDim.x * blockDim.x) {
    const int c = find_comp(a.info, a.ncomp, p);
    const CompInfo ci = a.info[c];
    const int q = p - ci.canvas_off;
    const int ly = q / ci.rw, lx = q - ly * ci.rw;
    const int x0 = max(lx - ci.a, 0), x1 = min(lx + ci.k - 1 - ci.a, ci.rw - 1);
    const unsigned char* row = a.seg0 + ci.canvas_off + ly * ci.rw;
    unsigned char v = 0;
    for (int x = x0; x <= x1 && !v; ++x) v |= row[x];
    a.seg1[p] = v;
  }
}

__global__ void k_dilate_v(CanvasArgs a) {
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < a.total; p += gridDim.x * blockDim.x) {
    const int c = find_comp(a.info, a.ncomp, p);
    const CompInfo ci = a.info[c];
    const int q = p - ci.canvas_off;
    const int ly = q / ci.rw, lx = q - ly * ci.rw;
    const int y0 = max(ly - ci.a, 0), y1 = min(ly + ci.k - 1 - ci.a, ci.rh - 1);
    const unsigned char* col = a.seg1 + ci.canvas_off + lx;
    unsigned char v = 0;
    for (int y = y0; y <= y1 && !v; ++y) v |= col[y * ci.rw];
    a.seg0[p] = v;
    a.clabel[p] = v ? p : -1;
  }
}

__global__ void k_merge8(CanvasArgs a) {
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < a.total; p += gridDim.x * blockDim.x) {
    if (!a.seg0[p]) continue;
    const int c = find_comp(a.info, a.ncomp, p);
    const CompInfo ci = a.info[c];
    const int q = p - ci.canvas_off;
    const int ly = q / ci.rw, lx = q - ly * ci.rw;
    if (lx > 0 && a.seg0[p - 1]) uf_union(a.clabel, p, p - 1);
    if (ly > 0) {
      const int up = p - ci.rw;
      if (a.seg0[up]) uf_union(a.clabel, p, up);
      if (lx > 0 && a.seg0[up - 1]) uf_union(a.clabel, p, up - 1);
      if (lx + 1 < ci.rw && a.seg0[up + 1]) uf_union(a.clabel, p, up + 1);
    }
  }
}

__global__ void k_flatten_select(CanvasArgs a) {
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < a.total; p += gridDim.x * blockDim.x) {
    if (!a.seg0[p]) continue;
    const int r = uf_find(a.clabel, p);
    a.clabel[p] = r;
    const int c = find_comp(a.info, a.ncomp, p);
    atomicMax(&a.sel[c], r);  // the fragment whose raster-first pixel comes last
  }
}

// one wave per canvas row
__global__ void k_row_extents(CanvasArgs a) {
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < a.total_rows; row += gridDim.x * wpb) {
    // find component by row offset
    int lo = 0, hi = a.ncomp - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (a.info[mid].row_off <= row)
        lo = mid;
      else
        hi = mid - 1;
    }
    const CompInfo ci = a.info[lo];
    const int ly = row - ci.row_off;
    const int sel = a.sel[lo];
    const int* L = a.clabel + ci.canvas_off + ly * ci.rw;
    int mn = INT_MAX, mx = -1;
    for (int x = lane; x < ci.rw; x += 64) {
      if (sel >= 0 && L[x] == sel) {
        mn = min(mn, x);
        mx = max(mx, x);
      }
    }
    for (int o = 32; o; o >>= 1) {
      mn = min(mn, __shfl_xor(mn, o));
      mx = max(mx, __shfl_xor(mx, o));
    }
    if (lane == 0) {
      a.rowmin[row] = mx < 0 ? -1 : mn;
      a.rowmax[row] = mx;
    }
  }
}

__device__ __forceinline__ long cross3(int2 o, int2 a, int2 b) {
  return (long)(a.x - o.x) * (b.y - o.y) - (long)(a.y - o.y) * (b.x - o.x);
}

// ---- K13: hull + min-area rectangle, one WAVE per component ------------------------------------
// Lane 0 builds the hull (monotone chain over the row extents, sequential by nature); the O(n^2) search over the hull's
// edges for the smallest enclosing rectangle runs one edge per lane.  Everything is exact integer arithmetic and the
// reduction keeps the FIRST edge among equal minima, i.e. what the sequential loop over the edges returns.
__device__ __forceinline__ long shfl_long(long v, int src) {
  const int lo = __shfl((int)v, src), hi = __shfl((int)(v >> 32), src);
  return (long)(((unsigned long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ long shfl_xor_long(long v, int mask) {
  const int lo = __shfl_xor((int)v, mask), hi = __shfl_xor((int)(v >> 32), mask);
  return (long)(((unsigned long)(unsigned)hi << 32) | (unsigned)lo);
}
// num_a / L_a < num_b / L_b, exactly (all non-negative, L > 0)
__device__ __forceinline__ bool ratio_less(long num_a, long L_a, long num_b, long L_b) {
  return (unsigned __int128)(unsigned long)num_a * (unsigned long)L_b < (unsigned __int128)(unsigned long)num_b * (unsigned long)L_a;
}

constexpr int KB_CAP = 1024;  // candidate points up to which k_boxes builds its chains in LDS

// Monotone chain step: push `pt` onto a chain whose two newest points live in registers (t1 newest)
#define KB_PUSH(chain, cnt, pt)                                   \
  do {                                                            \
    while ((cnt) >= 2 && cross3(t2, t1, (pt)) >= 0) {             \
      --(cnt);                                                    \
      t1 = t2;                                                    \
      if ((cnt) >= 2) t2 = (chain)[(cnt)-2];                      \
    }                                                             \
    (chain)[(cnt)++] = (pt);                                      \
    t2 = t1;                                                      \
    t1 = (pt);                                                    \
  } while (0)

__global__ __launch_bounds__(64) void k_boxes(CanvasArgs a) {
  __shared__ int2 hull_s[2 * (KB_CAP + 4)];
  __shared__ int2 cand_s[KB_CAP];
  const int lane = threadIdx.x;
  for (int c = blockIdx.x; c < a.ncomp; c += gridDim.x) {
    const CompInfo ci = a.info[c];
    float* out = a.boxes + ((size_t)ci.img * a.cap + ci.slot) * 8;
    if (a.sel[c] < 0) {  // reference: contours[0] of an empty list -> IndexError
      if (lane == 0) {
        atomicAdd(&a.totals[2], 1);
        for (int i = 0; i < 8; ++i) out[i] = 0.f;
      }
      continue;
    }
    // The hull is a monotone chain over the row extents (min and max column of every row, sorted by (y, x)): a serial
    // dependency chain, about 1 us per point on one lane (a 700-row line artefact cost 1 ms).  The points are first reduced,
    // in parallel and exactly, to those that can be strict hull vertices: the min point of a row only if it lies strictly
    // left of every row above it or strictly left of every row below it (otherwise it is inside, or on an edge of, the
    // triangle of those two rows' points and its own row's max point), the max point symmetrically; a single-point row if
    // either holds.  The chain over this subset is the chain over all points; it runs in LDS.  Only a component with more
    // than KB_CAP candidates (or more than 4096 rows) runs the plain chain over every point from global scratch.
    const int* rowmin = a.rowmin + ci.row_off;
    const int* rowmax = a.rowmax + ci.row_off;
    int n = 0, ncand = -1;
    int l = INT_MAX, r = -1, t = INT_MAX, b = -1;
    __syncthreads();  // the previous component's LDS data is no longer read (one wave per block: a wave-level sync)
    if (ci.rh <= 4096) {
      const int R = (ci.rh + 63) >> 6;  // contiguous rows per lane, at most 64: keep flags are bit masks
      const int r_lo = min(lane * R, ci.rh), r_hi = min(r_lo + R, ci.rh);
      int cmin = INT_MAX, cmax = INT_MIN, cfirst = INT_MAX, clast = -1;
      for (int ly = r_lo; ly < r_hi; ++ly) {
        const int mn = rowmin[ly];
        if (mn < 0) continue;
        cmin = min(cmin, mn);
        cmax = max(cmax, rowmax[ly]);
        cfirst = min(cfirst, ly);
        clast = ly;
      }
      // exclusive prefix / suffix extremes over the lanes
      int ipmin = cmin, ipmax = cmax, ismin = cmin, ismax = cmax;
      for (int off = 1; off < 64; off <<= 1) {
        const int a0 = __shfl_up(ipmin, off), a1 = __shfl_up(ipmax, off);
        const int b0 = __shfl_down(ismin, off), b1 = __shfl_down(ismax, off);
        if (lane >= off) {
          ipmin = min(ipmin, a0);
          ipmax = max(ipmax, a1);
        }
        if (lane + off < 64) {
          ismin = min(ismin, b0);
          ismax = max(ismax, b1);
        }
      }
      int pmin = __shfl_up(ipmin, 1), pmax = __shfl_up(ipmax, 1), smin = __shfl_down(ismin, 1), smax = __shfl_down(ismax, 1);
      if (lane == 0) {
        pmin = INT_MAX;
        pmax = INT_MIN;
      }
      if (lane == 63) {
        smin = INT_MAX;
        smax = INT_MIN;
      }
      l = ci.sx + __shfl(ipmin, 63);
      r = ci.sx + __shfl(ipmax, 63);
      int tf = cfirst, bl = clast;
      for (int off = 32; off > 0; off >>= 1) {
        tf = min(tf, __shfl_xor(tf, off));
        bl = max(bl, __shfl_xor(bl, off));
      }
      t = ci.sy + tf;
      b = ci.sy + bl;
      // keep flags of this lane's rows: km = min points, kx = max points (bit = row - r_lo)
      unsigned long km = 0, kx = 0;
      int rmin = pmin, rmax = pmax;
      for (int ly = r_lo; ly < r_hi; ++ly) {
        const int mn = rowmin[ly];
        if (mn < 0) continue;
        const int mx = rowmax[ly];
        if (mn < rmin) km |= 1ul << (ly - r_lo);
        if (mx > rmax) kx |= 1ul << (ly - r_lo);
        rmin = min(rmin, mn);
        rmax = max(rmax, mx);
      }
      rmin = smin;
      rmax = smax;
      for (int ly = r_hi - 1; ly >= r_lo; --ly) {
        const int mn = rowmin[ly];
        if (mn < 0) continue;
        const int mx = rowmax[ly];
        const unsigned long bit = 1ul << (ly - r_lo);
        if (mn < rmin) km |= bit;
        if (mx > rmax) kx |= bit;
        if (mx == mn) {  // one point
          if (kx & bit) km |= bit;
          kx &= ~bit;
        }
        rmin = min(rmin, mn);
        rmax = max(rmax, mx);
      }
      const int cnt = __popcll(km) + __popcll(kx);
      int incl = cnt;
      for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
      }
      ncand = __shfl(incl, 63);
      if (ncand <= KB_CAP) {  // compact candidate list in (y, x) order
        int pos = incl - cnt;
        for (int ly = r_lo; ly < r_hi; ++ly) {
          const unsigned long bit = 1ul << (ly - r_lo);
          if (km & bit) cand_s[pos++] = make_int2(ci.sx + rowmin[ly], ci.sy + ly);
          if (kx & bit) cand_s[pos++] = make_int2(ci.sx + rowmax[ly], ci.sy + ly);
        }
      } else {
        ncand = -1;
      }
      __syncthreads();
    }
    const bool in_lds = ncand >= 0;
    int2* lower = in_lds ? hull_s : a.hullbuf + 2 * (2 * (size_t)ci.row_off + 4 * (size_t)c);
    if (lane == 0) {
      int2* upper = lower + (in_lds ? KB_CAP + 4 : 2 * ci.rh + 4);
      int nl = 0, nu = 0;
      int2 t1 = make_int2(0, 0), t2 = make_int2(0, 0);
      if (in_lds) {
        for (int i = 0; i < ncand; ++i) {
          const int2 pt = cand_s[i];
          KB_PUSH(lower, nl, pt);
        }
        for (int i = ncand - 1; i >= 0; --i) {
          const int2 pt = cand_s[i];
          KB_PUSH(upper, nu, pt);
        }
      } else {
        // points sorted by (y, x): row by row, min then max
        for (int ly = 0; ly < ci.rh; ++ly) {
          const int mn = rowmin[ly];
          if (mn < 0) continue;
          const int mx = rowmax[ly];
          const int y = ci.sy + ly;
          t = min(t, y);
          b = max(b, y);
          l = min(l, ci.sx + mn);
          r = max(r, ci.sx + mx);
          for (int e = 0; e < (mx != mn ? 2 : 1); ++e) {
            const int2 pt = make_int2(ci.sx + (e ? mx : mn), y);
            KB_PUSH(lower, nl, pt);
          }
        }
        for (int ly = ci.rh - 1; ly >= 0; --ly) {
          const int mn = rowmin[ly];
          if (mn < 0) continue;
          const int mx = rowmax[ly];
          const int y = ci.sy + ly;
          for (int e = 0; e < (mx != mn ? 2 : 1); ++e) {
            const int2 pt = make_int2(ci.sx + (e ? mn : mx), y);  // reversed (y,x) order
            KB_PUSH(upper, nu, pt);
          }
        }
      }
      // hull = lower[:-1] + upper[:-1]; write it contiguously into `lower`
      if (nl == 1) {
        n = 1;
      } else {
        n = nl - 1;
        for (int i = 0; i + 1 < nu; ++i) lower[n++] = upper[i];
      }
      if (n >= 3) {
        // orientation: clockwise on screen <=> positive shoelace in image coordinates
        long area2 = 0;
        for (int i = 0; i < n; ++i) {
          const int2 p0 = lower[i], p1 = lower[(i + 1) % n];
          area2 += (long)p0.x * p1.y - (long)p1.x * p0.y;
        }
        if (area2 < 0)
          for (int i = 1, j = n - 1; i < j; ++i, --j) {
            const int2 tmp = lower[i];
            lower[i] = lower[j];
            lower[j] = tmp;
          }
      }
    }
    __threadfence_block();  // the hull, written by lane 0, is read by the whole wave
    __syncthreads();
    n = __shfl(n, 0);
    const int2* H = lower;
    long bnum = 0, bL = 0, bumin = 0, bumax = 0, bvmin = 0, bvmax = 0;
    int bdx = 0, bdy = 0, bidx = INT_MAX;
    if (n >= 3) {
      for (int i = lane; i < n; i += 64) {
        const int2 p0 = H[i], p1 = H[i + 1 < n ? i + 1 : 0];
        const long dx = p1.x - p0.x, dy = p1.y - p0.y;
        const long L = dx * dx + dy * dy;
        long umin = LONG_MAX, umax = LONG_MIN, vmin = LONG_MAX, vmax = LONG_MIN;
        for (int j = 0; j < n; ++j) {
          const int2 q = H[j];
          const long u = q.x * dx + q.y * dy;
          const long v = -q.x * dy + q.y * dx;
          umin = min(umin, u);
          umax = max(umax, u);
          vmin = min(vmin, v);
          vmax = max(vmax, v);
        }
        const long num = (umax - umin) * (vmax - vmin);
        if (bidx == INT_MAX || ratio_less(num, L, bnum, bL)) {  // strictly smaller: the first of equal minima stays
          bidx = i;
          bnum = num;
          bL = L;
          bdx = (int)dx;
          bdy = (int)dy;
          bumin = umin;
          bumax = umax;
          bvmin = vmin;
          bvmax = vmax;
        }
      }
      long rnum = bnum, rL = bL;
      int ridx = bidx;
      for (int off = 32; off > 0; off >>= 1) {
        const long onum = shfl_xor_long(rnum, off), oL = shfl_xor_long(rL, off);
        const int oidx = __shfl_xor(ridx, off);
        bool take = false;
        if (oidx != INT_MAX) {
          if (ridx == INT_MAX)
            take = true;
          else if (ratio_less(onum, oL, rnum, rL))
            take = true;
          else if (!ratio_less(rnum, rL, onum, oL) && oidx < ridx)
            take = true;
        }
        if (take) {
          rnum = onum;
          rL = oL;
          ridx = oidx;
        }
      }
      const int wl = ridx & 63;  // the lane whose own first minimum is the global one
      bnum = shfl_long(bnum, wl);
      bL = shfl_long(bL, wl);
      bumin = shfl_long(bumin, wl);
      bumax = shfl_long(bumax, wl);
      bvmin = shfl_long(bvmin, wl);
      bvmax = shfl_long(bvmax, wl);
      bdx = __shfl(bdx, wl);
      bdy = __shfl(bdy, wl);
    }
    if (lane != 0) continue;
    float bx[4], by[4];
    if (n == 1) {
      for (int i = 0; i < 4; ++i) {
        bx[i] = (float)H[0].x;
        by[i] = (float)H[0].y;
      }
    } else if (n == 2) {
      bx[0] = bx[1] = (float)H[0].x;
      by[0] = by[1] = (float)H[0].y;
      bx[2] = bx[3] = (float)H[1].x;
      by[2] = by[3] = (float)H[1].y;
    } else {
      const long us[4] = {bumin, bumax, bumax, bumin};
      const long vs[4] = {bvmin, bvmin, bvmax, bvmax};
      for (int i = 0; i < 4; ++i) {
        bx[i] = (float)((double)(us[i] * bdx - vs[i] * bdy) / (double)bL);
        by[i] = (float)((double)(us[i] * bdy + vs[i] * bdx) / (double)bL);
      }
    }
    // diamond test (detection.py:276-281), float32 arithmetic
    const float wx = bx[0] - bx[1], wy = by[0] - by[1];
    const float hx = bx[1] - bx[2], hy = by[1] - by[2];
    const float wlen = sqrtf(wx * wx + wy * wy);
    const float hlen = sqrtf(hx * hx + hy * hy);
    const float ratio = fmaxf(wlen, hlen) / (fminf(wlen, hlen) + 1e-5f);
    float ox[4], oy[4];
    if (fabsf(1.f - ratio) <= 0.1f) {
      ox[0] = (float)l; oy[0] = (float)t;
      ox[1] = (float)r; oy[1] = (float)t;
      ox[2] = (float)r; oy[2] = (float)b;
      ox[3] = (float)l; oy[3] = (float)b;
    } else {
      int k = 0;
      float best = bx[0] + by[0];
      for (int i = 1; i < 4; ++i) {
        const float s = bx[i] + by[i];
        if (s < best) {
          best = s;
          k = i;
        }
      }
      for (int i = 0; i < 4; ++i) {
        ox[i] = bx[(i + k) & 3];
        oy[i] = by[(i + k) & 3];
      }
    }
    for (int i = 0; i < 4; ++i) {
      out[2 * i] = 2.f * ox[i];      // detection.py:285
      out[2 * i + 1] = 2.f * oy[i];
    }
  }
}

__global__ void k_fill_int(int* p, int n, int v) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = v;
}

dim3 grid_for(size_t n, int block = 256) {
  size_t b = (n + block - 1) / block;
  if (b > 256 * 32) b = 256 * 32;
  if (b < 1) b = 1;
  return dim3((unsigned)b);
}

}  // namespace

// d_heat: device [N][h][w][2]; d_boxes: device [N][cap][4][2]; h_counts: host [N].
// Returns KOCR_OK, or KOCR_ECAPACITY (counts still filled with the true numbers).
int postproc_get_boxes(kocr_ctx* ctx, const float* d_heat, int N, int h, int w, float det_thr,
                       float text_thr, float link_thr, int size_thr, float* d_boxes, int cap,
                       int* h_counts, int* n_empty_out, PPDeviceOut* dev) {
  if (n_empty_out) *n_empty_out = 0;
  if (N <= 0) return KOCR_OK;
  if (h <= 0 || w <= 0 || h > 4096 || w > 4096)
    KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_get_boxes: heat-map side must be in [1, 4096]");
  const size_t NP = (size_t)N * h * w;
  if (NP > 0x7ffffff0u) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_get_boxes: batch too large (N*h*w must fit int32)");
  if (cap <= 0) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_get_boxes: cap must be positive");
  hipStream_t s = ctx->stream;
  PPArgs p;
  p.heat = d_heat;
  p.N = N;
  p.h = h;
  p.w = w;
  p.det_thr = det_thr;
  p.text_thr = text_thr;
  p.link_thr = link_thr;
  p.size_thr = size_thr;
  p.bpi = (h * w + SCAN_BLOCK - 1) / SCAN_BLOCK;
  p.cap = cap;
  const size_t need = NP * (1 + 7 * 4) + (size_t)N * p.bpi * 8 + (size_t)N * 4 + (size_t)N * cap * 4 +
                      (size_t)N * cap * sizeof(CompInfo) + (size_t)(N + 1) * 4 + 64 + 20 * 256;
  KOCR_TRY(arena_reserve(ctx, ctx->pp, need));
  ctx->pp.off = 0;
  auto A = [&](size_t bytes) { return arena_alloc(ctx->pp, bytes); };
  p.flags = (unsigned char*)A(NP);
  p.label = (int*)A(NP * 4);
  p.area = (int*)A(NP * 4);
  p.minx = (int*)A(NP * 4);
  p.maxx = (int*)A(NP * 4);
  p.miny = (int*)A(NP * 4);
  p.maxy = (int*)A(NP * 4);
  p.tmax = (int*)A(NP * 4);
  p.blk_cnt = (int*)A((size_t)N * p.bpi * 4);
  p.blk_off = (int*)A((size_t)N * p.bpi * 4);
  p.counts = (int*)A((size_t)N * 4);
  p.comp_root = (int*)A((size_t)N * cap * 4);
  p.info = (CompInfo*)A((size_t)N * cap * sizeof(CompInfo));
  p.img_base = (int*)A((size_t)(N + 1) * 4);
  p.totals = (int*)A(64);
  if (!p.totals) KOCR_FAIL(ctx, KOCR_ENOMEM, "kocr_get_boxes: scratch exhausted");
  if (dev) {
    dev->d_counts = p.counts;
    dev->d_totals = p.totals;
  }

  const double hb = 8.0 * NP;
  {
    ProfScope ps(ctx, "pp_threshold_ccl", 0, hb);
    hipLaunchKernelGGL(k_threshold, grid_for(NP), dim3(256), 0, s, p);
    hipLaunchKernelGGL(k_merge4, grid_for(NP), dim3(256), 0, s, p);
    hipLaunchKernelGGL(k_flatten_stats, grid_for(NP), dim3(256), 0, s, p);
  }
  {
    ProfScope ps(ctx, "pp_compact", 0, 0);
    hipLaunchKernelGGL(k_count, dim3(p.bpi, N), dim3(256), 0, s, p);
    hipLaunchKernelGGL(k_scan_blocks, dim3(N), dim3(256), 0, s, p);
    hipLaunchKernelGGL(k_assign, dim3(p.bpi, N), dim3(64), 0, s, p);
    hipLaunchKernelGGL(k_geometry, dim3(1), dim3(256), 0, s, p);
  }
  KOCR_HIP(ctx, hipGetLastError());
  int totals[4];
  KOCR_HIP(ctx, hipMemcpyAsync(h_counts, p.counts, (size_t)N * 4, hipMemcpyDeviceToHost, s));
  KOCR_HIP(ctx, hipMemcpyAsync(totals, p.totals, 16, hipMemcpyDeviceToHost, s));
  KOCR_HIP(ctx, hipStreamSynchronize(s));
  bool over = false;
  for (int i = 0; i < N; ++i) over |= h_counts[i] > cap;
  if (over) KOCR_FAIL(ctx, KOCR_ECAPACITY, "kocr_get_boxes: more boxes than cap in at least one image");
  if (totals[0] < 0) KOCR_FAIL(ctx, KOCR_ECAPACITY, "kocr_get_boxes: dilation canvases exceed 2^31 pixels");
  const int ncomp = totals[3];
  if (ncomp == 0) return KOCR_OK;

  CanvasArgs a;
  a.info = p.info;
  a.ncomp = ncomp;
  a.total = totals[0];
  a.total_rows = totals[1];
  a.flags = p.flags;
  a.label = p.label;
  a.h = h;
  a.w = w;
  a.cap = cap;
  a.boxes = d_boxes;
  a.counts = p.counts;
  a.totals = p.totals;
  const size_t cneed = (size_t)a.total * (1 + 1 + 4) + (size_t)ncomp * 4 + (size_t)a.total_rows * 8 +
                       ((size_t)a.total_rows * 4 + (size_t)ncomp * 8 + 16) * sizeof(int2) + 16 * 256;
  KOCR_TRY(arena_reserve(ctx, ctx->pp2, cneed));
  ctx->pp2.off = 0;
  auto B = [&](size_t bytes) { return arena_alloc(ctx->pp2, bytes); };
  a.seg0 = (unsigned char*)B(a.total);
  a.seg1 = (unsigned char*)B(a.total);
  a.clabel = (int*)B((size_t)a.total * 4);
  a.sel = (int*)B((size_t)ncomp * 4);
  a.rowmin = (int*)B((size_t)a.total_rows * 4);
  a.rowmax = (int*)B((size_t)a.total_rows * 4);
  a.hullbuf = (int2*)B(((size_t)a.total_rows * 4 + (size_t)ncomp * 8 + 16) * sizeof(int2));
  if (!a.hullbuf) KOCR_FAIL(ctx, KOCR_ENOMEM, "kocr_get_boxes: canvas scratch exhausted");
  {
    ProfScope ps(ctx, "pp_canvas_boxes", 0, 0);
    hipLaunchKernelGGL(k_fill_int, grid_for(ncomp), dim3(256), 0, s, a.sel, ncomp, -1);
    hipLaunchKernelGGL(k_canvas_fill, grid_for(a.total), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_dilate_h, grid_for(a.total), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_dilate_v, grid_for(a.total), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_merge8, grid_for(a.total), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_flatten_select, grid_for(a.total), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_row_extents, grid_for((size_t)a.total_rows * 64), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_boxes, grid_for(ncomp, 1), dim3(64), 0, s, a);
  }
  KOCR_HIP(ctx, hipGetLastError());
  if (n_empty_out && !dev) {
    KOCR_HIP(ctx, hipMemcpyAsync(totals, p.totals, 16, hipMemcpyDeviceToHost, s));
    KOCR_HIP(ctx, hipStreamSynchronize(s));
    *n_empty_out = totals[2];
  }
  return KOCR_OK;
}
