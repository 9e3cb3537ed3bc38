// craft.cpp — the CRAFT detector graph (VGG16-BN backbone + U-Net decoder + 5-conv head).
//
// Follows detection.build_keras_model / build_vgg_backbone / upconv (detection.py:65-103,
// 312-335, 353-413) layer for layer; layer names are the Keras layer names, which equal the
// PyTorch state-dict keys load_torch_weights maps them from (detection.py:428-468).
// BatchNorm (eps = 1e-5, detection.py:69-71, 95-97) is folded with the conv bias into the
// per-channel affine of the conv kernel's epilogue.  Skip tensors are written by their
// producer straight into the channel slice of the concat buffer that consumes them, so no
// concat kernel exists:  cat1 = [s5 | s4], cat2 = [up(y1) | s3], cat3 = [up(y2) | s2],
// cat4 = [up(y3) | s1]  (detection.py:380-389).
// In bf16x3 mode two chains of convolutions WITHOUT a non-linearity between them are evaluated in their algebraically
// identical shorter form (same sums, fp32 round-off apart): slice5.1 -> slice5.2 -> upconv1.conv.0 as one composed
// dilated 3x3 plus a 1x1 over s4 (craft_load / craft_run, kocr_set_schedule / KOCR_LINFOLD), and conv1x1(concat(resize(y), skip)) as
// resize(conv1x1_y(y)) + conv1x1_skip(skip) (up_conv, kocr_set_schedule / KOCR_UPFOLD); the up-sampled halves of cat2..4 and s5 are then
// never written.
#include "common.h"
#include <algorithm>
#include <cmath>
#include <thread>

struct CraftNet {
  std::map<std::string, ConvLayer> L;
  float* d_lut = nullptr;  // [3][256] compute_input table (detection.py:34-42)
  bool loaded = false;
};

namespace {

struct Spec {
  const char* conv;
  const char* bn;  // nullptr: no BN
  int cin, cout, k, dil, relu;
};

const Spec kSpecs[] = {
    {"basenet.slice1.0", "basenet.slice1.1", 3, 64, 3, 1, 1},
    {"basenet.slice1.3", "basenet.slice1.4", 64, 64, 3, 1, 1},
    {"basenet.slice1.7", "basenet.slice1.8", 64, 128, 3, 1, 1},
    {"basenet.slice1.10", "basenet.slice1.11", 128, 128, 3, 1, 1},
    {"basenet.slice2.14", "basenet.slice2.15", 128, 256, 3, 1, 1},
    {"basenet.slice2.17", "basenet.slice2.18", 256, 256, 3, 1, 1},
    {"basenet.slice3.20", "basenet.slice3.21", 256, 256, 3, 1, 1},
    {"basenet.slice3.24", "basenet.slice3.25", 256, 512, 3, 1, 1},
    {"basenet.slice3.27", "basenet.slice3.28", 512, 512, 3, 1, 1},
    {"basenet.slice4.30", "basenet.slice4.31", 512, 512, 3, 1, 1},
    {"basenet.slice4.34", "basenet.slice4.35", 512, 512, 3, 1, 1},
    // s4 is the BN output, NOT its ReLU (detection.py:333)
    {"basenet.slice4.37", "basenet.slice4.38", 512, 512, 3, 1, 0},
    {"basenet.slice5.1", nullptr, 512, 1024, 3, 6, 0},
    {"basenet.slice5.2", nullptr, 1024, 1024, 1, 1, 0},
    {"upconv1.conv.0", "upconv1.conv.1", 1536, 512, 1, 1, 1},
    {"upconv1.conv.3", "upconv1.conv.4", 512, 256, 3, 1, 1},
    {"upconv2.conv.0", "upconv2.conv.1", 768, 256, 1, 1, 1},
    {"upconv2.conv.3", "upconv2.conv.4", 256, 128, 3, 1, 1},
    {"upconv3.conv.0", "upconv3.conv.1", 384, 128, 1, 1, 1},
    {"upconv3.conv.3", "upconv3.conv.4", 128, 64, 3, 1, 1},
    {"upconv4.conv.0", "upconv4.conv.1", 192, 64, 1, 1, 1},
    {"upconv4.conv.3", "upconv4.conv.4", 64, 32, 3, 1, 1},
    {"conv_cls.0", nullptr, 32, 32, 3, 1, 1},
    {"conv_cls.2", nullptr, 32, 32, 3, 1, 1},
    {"conv_cls.4", nullptr, 32, 16, 3, 1, 1},
    {"conv_cls.6", nullptr, 16, 16, 1, 1, 1},
    {"conv_cls.8", nullptr, 16, 2, 1, 1, 0},
};

// channels of the up-sampled decoder tensor y at the head of the concat a 1x1 layer reads (detection.py:380-389)
int fold_channels(const char* conv) {
  const std::string n = conv;
  return n == "upconv2.conv.0" ? 256 : n == "upconv3.conv.0" ? 128 : n == "upconv4.conv.0" ? 64 : 0;
}

// C[m][n] = A[m][k] * B[k][n] in float64 (row-major), rows of C spread over a few host threads; load-time only
void matmul_f64(const double* A, const double* B, double* C, int m, int k, int n) {
  const int nt = std::max(1, std::min(16, (int)std::thread::hardware_concurrency()));
  std::vector<std::thread> th;
  for (int t = 0; t < nt; ++t)
    th.emplace_back([=]() {
      for (int i = t; i < m; i += nt) {
        double* c = C + (size_t)i * n;
        std::fill(c, c + n, 0.0);
        for (int kk = 0; kk < k; ++kk) {
          const double a = A[(size_t)i * k + kk];
          const double* b = B + (size_t)kk * n;
          for (int j = 0; j < n; ++j) c[j] += a * b[j];
        }
      }
    });
  for (auto& x : th) x.join();
}

struct Blob {
  const float* p;
  int64_t shape[4];
  int rank;
  size_t numel() const {
    size_t n = 1;
    for (int i = 0; i < rank; ++i) n *= (size_t)shape[i];
    return n;
  }
};

}  // namespace

int craft_load(kocr_ctx* ctx, int n, const char* const* names, const float* const* data,
               const int64_t* shapes, const int* ranks) {
  std::map<std::string, Blob> blobs;
  for (int i = 0; i < n; ++i) {
    Blob b;
    b.p = data[i];
    b.rank = ranks[i];
    for (int d = 0; d < 4; ++d) b.shape[d] = shapes[i * 4 + d];
    std::string nm = names[i];
    if (nm.rfind("module.", 0) == 0) nm = nm.substr(7);
    blobs[nm] = b;
  }
  auto need = [&](const std::string& k, size_t numel, const Blob** out) -> int {
    auto it = blobs.find(k);
    if (it == blobs.end()) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_load_craft: missing tensor " + k);
    if (it->second.numel() != numel)
      KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_load_craft: tensor " + k + " has wrong size");
    *out = &it->second;
    return KOCR_OK;
  };
  if (!ctx->craft) ctx->craft = new CraftNet();
  CraftNet* net = ctx->craft;
  for (const Spec& s : kSpecs) {
    const Blob *w, *b;
    KOCR_TRY(need(std::string(s.conv) + ".weight", (size_t)s.cout * s.cin * s.k * s.k, &w));
    KOCR_TRY(need(std::string(s.conv) + ".bias", (size_t)s.cout, &b));
    std::vector<float> pa(s.cout, 1.f), pb(b->p, b->p + s.cout);
    if (s.bn) {
      const Blob *g, *be, *mu, *var;
      KOCR_TRY(need(std::string(s.bn) + ".weight", s.cout, &g));
      KOCR_TRY(need(std::string(s.bn) + ".bias", s.cout, &be));
      KOCR_TRY(need(std::string(s.bn) + ".running_mean", s.cout, &mu));
      KOCR_TRY(need(std::string(s.bn) + ".running_var", s.cout, &var));
      for (int o = 0; o < s.cout; ++o) {
        const float sc = g->p[o] / std::sqrt(var->p[o] + 1e-5f);
        pa[o] = sc;
        pb[o] = (b->p[o] - mu->p[o]) * sc + be->p[o];
      }
    }
    ConvLayer& L = net->L[s.conv];
    L.name = s.conv;
    KOCR_TRY(prepare_conv(ctx, L, w->p, /*oihw=*/true, s.cin, s.cout, s.k, s.k, s.dil, pa.data(),
                          pb.data(), s.relu, nullptr, nullptr));
    // The decoder's 1x1 convolutions over concat(resize(y), skip), split by input-channel range (see up_conv below):
    // "<name>#y" = the y columns, no bias / BN / ReLU; "<name>#skip" = the skip columns with the layer's epilogue.
    const int c_y = fold_channels(s.conv);
    if (c_y) {
      std::vector<float> wy((size_t)s.cout * c_y), wsk((size_t)s.cout * (s.cin - c_y));
      for (int o = 0; o < s.cout; ++o)
        for (int c = 0; c < s.cin; ++c)
          (c < c_y ? wy[(size_t)o * c_y + c] : wsk[(size_t)o * (s.cin - c_y) + (c - c_y)]) = w->p[(size_t)o * s.cin + c];
      ConvLayer& Ly = net->L[std::string(s.conv) + "#y"];
      Ly.name = std::string(s.conv) + "#y";
      KOCR_TRY(prepare_conv(ctx, Ly, wy.data(), true, c_y, s.cout, 1, 1, 1, nullptr, nullptr, 0, nullptr, nullptr));
      ConvLayer& Ls = net->L[std::string(s.conv) + "#skip"];
      Ls.name = std::string(s.conv) + "#skip";
      KOCR_TRY(prepare_conv(ctx, Ls, wsk.data(), true, s.cin - c_y, s.cout, 1, 1, 1, pa.data(), pb.data(), s.relu, nullptr, nullptr));
    }
  }
  // slice5.1 (3x3, dilation 6) -> slice5.2 (1x1) -> [concat with s4] -> upconv1.conv.0 (1x1) has NO non-linearity between
  // its three convolutions (detection.py:349-353: Conv2D, Conv2D without activation; :106-108 the upconv's first Conv2D
  // precedes its BatchNorm/ReLU), so the chain is one linear map of (pool(s4), s4):
  //     Wu_a (W2 (W1 * h0 + b1) + b2) + Wu_b s4 + bu  =  (Wu_a W2 W1) * h0  +  Wu_b s4  +  Wu_a (W2 b1 + b2) + bu
  // The composite 3x3 dilated 512 -> 512 kernel is formed once, in float64, here; the forward then runs it plus a
  // 512 -> 512 1x1 over s4 instead of 512 -> 1024 (3x3), 1024 -> 1024 and 1536 -> 512: 40 % of the products, and the two
  // 1024-channel tensors are never written.  Same sum up to fp32 round-off (the intermediates are no longer rounded to
  // fp32); craft_run uses it in bf16x3 mode unless KOCR_LINFOLD=0.
  {
    const Blob *w1, *b1, *w2, *b2, *wu, *bu;
    KOCR_TRY(need("basenet.slice5.1.weight", (size_t)1024 * 512 * 9, &w1));
    KOCR_TRY(need("basenet.slice5.1.bias", 1024, &b1));
    KOCR_TRY(need("basenet.slice5.2.weight", (size_t)1024 * 1024, &w2));
    KOCR_TRY(need("basenet.slice5.2.bias", 1024, &b2));
    KOCR_TRY(need("upconv1.conv.0.weight", (size_t)512 * 1536, &wu));
    KOCR_TRY(need("upconv1.conv.0.bias", 512, &bu));
    std::vector<double> A((size_t)512 * 1024), B(w2->p, w2->p + (size_t)1024 * 1024), P((size_t)512 * 1024);
    std::vector<float> wskip((size_t)512 * 512);
    for (int o = 0; o < 512; ++o)
      for (int c = 0; c < 1536; ++c) {
        if (c < 1024)
          A[(size_t)o * 1024 + c] = wu->p[(size_t)o * 1536 + c];
        else
          wskip[(size_t)o * 512 + (c - 1024)] = wu->p[(size_t)o * 1536 + c];
      }
    matmul_f64(A.data(), B.data(), P.data(), 512, 1024, 1024);  // Wu_a W2
    std::vector<double> W1(w1->p, w1->p + (size_t)1024 * 4608), Wc((size_t)512 * 4608);
    matmul_f64(P.data(), W1.data(), Wc.data(), 512, 1024, 4608);  // (Wu_a W2) W1: [512][512][3][3]
    std::vector<float> wc(Wc.begin(), Wc.end());
    std::vector<double> c0(512);
    for (int o = 0; o < 512; ++o) {
      double acc = 0;
      for (int c = 0; c < 1024; ++c) acc += P[(size_t)o * 1024 + c] * (double)b1->p[c] + A[(size_t)o * 1024 + c] * (double)b2->p[c];
      c0[o] = acc;
    }
    const Blob *g, *be, *mu, *var;
    KOCR_TRY(need("upconv1.conv.1.weight", 512, &g));
    KOCR_TRY(need("upconv1.conv.1.bias", 512, &be));
    KOCR_TRY(need("upconv1.conv.1.running_mean", 512, &mu));
    KOCR_TRY(need("upconv1.conv.1.running_var", 512, &var));
    std::vector<float> pa(512), pb(512);
    for (int o = 0; o < 512; ++o) {
      const float sc = g->p[o] / std::sqrt(var->p[o] + 1e-5f);
      pa[o] = sc;
      pb[o] = (float)(((double)bu->p[o] + c0[o] - (double)mu->p[o]) * (double)sc + (double)be->p[o]);
    }
    ConvLayer& Lf = net->L["basenet.slice5#fold"];
    Lf.name = "basenet.slice5#fold";
    KOCR_TRY(prepare_conv(ctx, Lf, wc.data(), true, 512, 512, 3, 3, 6, nullptr, nullptr, 0, nullptr, nullptr));
    ConvLayer& Ls = net->L["upconv1.conv.0#skip"];
    Ls.name = "upconv1.conv.0#skip";
    KOCR_TRY(prepare_conv(ctx, Ls, wskip.data(), true, 512, 512, 1, 1, 1, pa.data(), pb.data(), 1, nullptr, nullptr));
  }
  // compute_input (detection.py:34-42): float32 image; image -= mean*255 (float64 math, stored
  // back as float32); image /= variance*255 (same).
  const double mean[3] = {0.485, 0.456, 0.406}, var[3] = {0.229, 0.224, 0.225};
  std::vector<float> lut(3 * 256);
  for (int c = 0; c < 3; ++c)
    for (int v = 0; v < 256; ++v) {
      const float t = (float)((double)(float)v - mean[c] * 255);
      lut[c * 256 + v] = (float)((double)t / (var[c] * 255));
    }
  KOCR_TRY(ctx->upload(&net->d_lut, lut));
  net->loaded = true;
  return KOCR_OK;
}

void craft_free(kocr_ctx* ctx) {
  delete ctx->craft;
  ctx->craft = nullptr;
}

namespace {
struct Dims {
  int H, W, H2, W2, H4, W4, H8, W8, H16, W16;
  explicit Dims(int h, int w) {
    H = h;
    W = w;
    H2 = H / 2;
    W2 = W / 2;
    H4 = H2 / 2;
    W4 = W2 / 2;
    H8 = H4 / 2;
    W8 = W4 / 2;
    H16 = H8 / 2;
    W16 = W8 / 2;
  }
};
size_t al(size_t b) { return (b + 255) & ~(size_t)255; }

// Activation memory by LIFETIME: every tensor is returned to the pool as soon as its last consumer has been
// launched (one stream: launch order = execution order), so the peak live set -- the two full-resolution tensors
// around slice1.3 plus the pooled one -- sizes the workspace, not the sum of all layers: 1.36 GB per 1536x1536 image
// instead of 2.6 GB, which lets BASELINE's 32-image batches (and 2048x2048 inputs) run as ONE micro-batch.
struct LivePool {
  struct Blk {
    size_t off, size;
  };
  std::vector<Blk> free_;  // sorted by offset, coalesced
  size_t top = 0, peak = 0;
  size_t alloc(size_t bytes) {
    bytes = al(bytes);
    for (size_t i = 0; i < free_.size(); ++i)
      if (free_[i].size >= bytes) {  // first fit
        const size_t off = free_[i].off;
        free_[i].off += bytes;
        free_[i].size -= bytes;
        if (free_[i].size == 0) free_.erase(free_.begin() + i);
        return off;
      }
    const size_t off = top;
    top += bytes;
    peak = std::max(peak, top);
    return off;
  }
  void release(size_t off, size_t bytes) {
    bytes = al(bytes);
    size_t i = 0;
    while (i < free_.size() && free_[i].off < off) ++i;
    free_.insert(free_.begin() + i, Blk{off, bytes});
    if (i + 1 < free_.size() && free_[i].off + free_[i].size == free_[i + 1].off) {
      free_[i].size += free_[i + 1].size;
      free_.erase(free_.begin() + i + 1);
    }
    if (i > 0 && free_[i - 1].off + free_[i - 1].size == free_[i].off) {
      free_[i - 1].size += free_[i].size;
      free_.erase(free_.begin() + i);
    }
    if (!free_.empty() && free_.back().off + free_.back().size == top) {
      top = free_.back().off;
      free_.pop_back();
    }
  }
};

// The CRAFT graph, once: with DRY = true nothing is launched and only the pool's peak is computed (this IS
// craft_workspace_bytes), with DRY = false the same allocation sequence runs against the ctx workspace.
template <bool DRY>
int craft_run(kocr_ctx* ctx, CraftNet* net, const void* d_img, int dtype, int N, int H, int W, float* d_heat, size_t* peak_out) {
  Dims d(H, W);
  LivePool pool;
  char* base = nullptr;
  size_t room = 0;
  if constexpr (!DRY) {
    const size_t o = (ctx->ws.off + 255) & ~(size_t)255;
    base = ctx->ws.base + o;
    room = ctx->ws.cap > o ? ctx->ws.cap - o : 0;
    KOCR_TRY(ctx->amax_begin());
  }
  // track = true: the tensor feeds a 3x3 convolution that may run in fp16 arithmetic (conv_w43h.hip); it carries one
  // max-|x| slot per image (Tensor::amax) which its producer maintains, so that the consumer can pick the exact
  // power-of-two input scale of every image.  (A consumer whose input has no slots reduces it on demand.)
  std::map<const float*, std::pair<size_t, size_t>> live;  // tensor base -> (offset, bytes)
  auto mk = [&](int h, int w, int c, Tensor* t, bool track = false) -> int {
    const size_t bytes = (size_t)N * h * w * c * sizeof(float);
    const size_t off = pool.alloc(bytes);
    t->N = N;
    t->H = h;
    t->W = w;
    t->C = c;
    t->cs = c;
    t->co = 0;
    if constexpr (DRY) {
      t->amax = nullptr;
      t->p = reinterpret_cast<float*>(off + 256);  // a distinct non-null key; never dereferenced
    } else {
      t->amax = track ? ctx->amax_slots(N) : nullptr;
      if (off + al(bytes) > room) KOCR_FAIL(ctx, KOCR_ENOMEM, "kocr_craft_forward: workspace exhausted");
      t->p = reinterpret_cast<float*>(base + off);
    }
    live[t->p] = {off, bytes};
    return KOCR_OK;
  };
  auto done = [&](const Tensor& t) {  // last consumer launched: the memory may be reused by later layers
    auto it = live.find(t.p);
    if (it != live.end()) {
      pool.release(it->second.first, it->second.second);
      live.erase(it);
    }
  };
#define RUN(expr)               \
  do {                          \
    if constexpr (!DRY) KOCR_TRY(expr); \
  } while (0)
  auto L = [&](const char* name) -> const ConvLayer& { return net->L.at(name); };

  Tensor x0;
  x0.N = N;
  x0.H = H;
  x0.W = W;
  x0.C = 3;
  x0.cs = 3;
  x0.co = 0;
  x0.p = (dtype == KOCR_F32) ? (float*)d_img : nullptr;
  const uint8_t* u8 = (dtype == KOCR_U8) ? (const uint8_t*)d_img : nullptr;
  const float* lut = DRY ? nullptr : net->d_lut;

  // ---- backbone (detection.py:312-335) ------------------------------------------------
  // conv + 2x2 max-pool: fused epilogue when the shape tiles, else two kernels; need_full: the pre-pool tensor is
  // consumed elsewhere (skip connection).  The full-resolution buffer of an un-needed tensor is returned at once.
  Tensor a1, a2, p1, b1, cat4, p2, c1, cat3, c3, p3, e1, cat2, f1, p4, g1, cat1, h0, h1;
  KOCR_TRY(mk(d.H, d.W, 64, &a1, true));
  RUN(launch_conv(ctx, L("basenet.slice1.0"), x0, u8, lut, a1));
  KOCR_TRY(mk(d.H, d.W, 64, &a2));
  KOCR_TRY(mk(d.H2, d.W2, 64, &p1, true));
  RUN(launch_conv_pool(ctx, L("basenet.slice1.3"), a1, nullptr, nullptr, a2, &p1, /*need_full=*/false));
  done(a1);
  done(a2);
  KOCR_TRY(mk(d.H2, d.W2, 128, &b1, true));
  RUN(launch_conv(ctx, L("basenet.slice1.7"), p1, nullptr, nullptr, b1));
  done(p1);
  KOCR_TRY(mk(d.H2, d.W2, 192, &cat4));
  KOCR_TRY(mk(d.H4, d.W4, 128, &p2, true));
  const Tensor s1 = cat4.slice(64, 128);
  RUN(launch_conv_pool(ctx, L("basenet.slice1.10"), b1, nullptr, nullptr, s1, &p2, /*need_full=*/true));  // s1: skip tensor
  done(b1);
  KOCR_TRY(mk(d.H4, d.W4, 256, &c1, true));
  RUN(launch_conv(ctx, L("basenet.slice2.14"), p2, nullptr, nullptr, c1));
  done(p2);
  KOCR_TRY(mk(d.H4, d.W4, 384, &cat3, true));
  const Tensor s2 = cat3.slice(128, 256);
  RUN(launch_conv(ctx, L("basenet.slice2.17"), c1, nullptr, nullptr, s2));
  done(c1);
  KOCR_TRY(mk(d.H4, d.W4, 256, &c3));
  KOCR_TRY(mk(d.H8, d.W8, 256, &p3, true));
  RUN(launch_conv_pool(ctx, L("basenet.slice3.20"), s2, nullptr, nullptr, c3, &p3, false));
  done(c3);
  KOCR_TRY(mk(d.H8, d.W8, 512, &e1, true));
  RUN(launch_conv(ctx, L("basenet.slice3.24"), p3, nullptr, nullptr, e1));
  done(p3);
  KOCR_TRY(mk(d.H8, d.W8, 768, &cat2, true));
  const Tensor s3 = cat2.slice(256, 512);
  RUN(launch_conv(ctx, L("basenet.slice3.27"), e1, nullptr, nullptr, s3));
  done(e1);
  KOCR_TRY(mk(d.H8, d.W8, 512, &f1));
  KOCR_TRY(mk(d.H16, d.W16, 512, &p4, true));
  RUN(launch_conv_pool(ctx, L("basenet.slice4.30"), s3, nullptr, nullptr, f1, &p4, false));
  done(f1);
  KOCR_TRY(mk(d.H16, d.W16, 512, &g1, true));
  RUN(launch_conv(ctx, L("basenet.slice4.34"), p4, nullptr, nullptr, g1));
  done(p4);
  KOCR_TRY(mk(d.H16, d.W16, 1536, &cat1, true));
  const Tensor s4 = cat1.slice(1024, 512);
  RUN(launch_conv(ctx, L("basenet.slice4.37"), g1, nullptr, nullptr, s4));
  done(g1);
  // ---- slice5 (detection.py:365-378) ----------------------------------------------------
  KOCR_TRY(mk(d.H16, d.W16, 512, &h0, true));
  RUN(launch_maxpool3x3s1(ctx, s4, h0));
  KOCR_TRY(mk(d.H16, d.W16, 1024, &h1));
  Tensor u1a;
  KOCR_TRY(mk(d.H16, d.W16, 512, &u1a, true));
  bool lin_fold = false;
  if constexpr (!DRY) {
    lin_fold = ctx->opt_linfold && dsplit_usable(L("upconv1.conv.0#skip"), s4) &&
               2 * (size_t)h0.H * h0.W * 512 * 4 < ((size_t)1 << 31);
    if (lin_fold) {  // see craft_load: slice5.1 -> slice5.2 -> upconv1.conv.0 as one dilated 3x3 plus a 1x1 over s4
      Tensor t = h1;  // the first half of h1's buffer, as a contiguous 512-channel tensor
      t.C = t.cs = 512;
      KOCR_TRY(launch_conv(ctx, L("basenet.slice5#fold"), h0, nullptr, nullptr, t));
      KOCR_TRY(launch_conv_dsplit(ctx, L("upconv1.conv.0#skip"), s4, u1a, &t));
    } else {
      KOCR_TRY(launch_conv(ctx, L("basenet.slice5.1"), h0, nullptr, nullptr, h1));
      KOCR_TRY(launch_conv(ctx, L("basenet.slice5.2"), h1, nullptr, nullptr, cat1.slice(0, 1024)));
      KOCR_TRY(launch_conv(ctx, L("upconv1.conv.0"), cat1, nullptr, nullptr, u1a));
    }
  }
  done(h0);
  done(h1);
  done(cat1);
  // ---- U-Net decoder (detection.py:380-390) ---------------------------------------------
  // conv1x1(concat(resize(y), skip)) -> BN -> ReLU.  A 1x1 convolution commutes with the bilinear resize (both are linear
  // and the resize acts per channel), so the y columns of the weight matrix are applied at y's own resolution -- a
  // quarter of the pixels -- and conv_dsplit's epilogue adds the resized product to the skip columns' sum: the
  // up-sampled tensor is never written, and a quarter of the layer's products disappear (same sum, different rounding
  // order: fp32 round-off, like the Winograd layers).  bf16x3 mode only; KOCR_UPFOLD=0, the fp16 mode and shapes the
  // split kernel does not take run resize + convolution over the concat buffer as before.  `t` is allocated either
  // way so that the dry run sizes the workspace for both.
  auto up_conv = [&](const char* name, Tensor& y, Tensor& cat, Tensor& out) -> int {
    const int c_y = y.C;
    Tensor t;
    KOCR_TRY(mk(y.H, y.W, out.C, &t));
    if constexpr (!DRY) {
      const ConvLayer& Ls = L((std::string(name) + "#skip").c_str());
      const Tensor skip = cat.slice(c_y, cat.C - c_y);
      const bool fold = ctx->opt_upfold && dsplit_usable(Ls, skip) &&
                        2 * (size_t)t.H * t.W * t.C * 4 < ((size_t)1 << 31);  // two images of t within 32-bit offsets
      if (fold) {
        KOCR_TRY(launch_conv(ctx, L((std::string(name) + "#y").c_str()), y, nullptr, nullptr, t));
        KOCR_TRY(launch_conv_dsplit(ctx, Ls, skip, out, &t));
      } else {
        KOCR_TRY(launch_resize_bilinear(ctx, y, cat.slice(0, c_y)));
        KOCR_TRY(launch_conv(ctx, L(name), cat, nullptr, nullptr, out));
      }
    }
    done(t);
    done(y);
    done(cat);
    return KOCR_OK;
  };
  Tensor u1b, u2a, u2b, u3a, u3b, u4a, feat, k0, k1, k2;
  KOCR_TRY(mk(d.H16, d.W16, 256, &u1b));
  RUN(launch_conv(ctx, L("upconv1.conv.3"), u1a, nullptr, nullptr, u1b));
  done(u1a);
  KOCR_TRY(mk(d.H8, d.W8, 256, &u2a, true));
  KOCR_TRY(up_conv("upconv2.conv.0", u1b, cat2, u2a));
  KOCR_TRY(mk(d.H8, d.W8, 128, &u2b));
  RUN(launch_conv(ctx, L("upconv2.conv.3"), u2a, nullptr, nullptr, u2b));
  done(u2a);
  KOCR_TRY(mk(d.H4, d.W4, 128, &u3a, true));
  KOCR_TRY(up_conv("upconv3.conv.0", u2b, cat3, u3a));
  KOCR_TRY(mk(d.H4, d.W4, 64, &u3b));
  RUN(launch_conv(ctx, L("upconv3.conv.3"), u3a, nullptr, nullptr, u3b));
  done(u3a);
  KOCR_TRY(mk(d.H2, d.W2, 64, &u4a, true));
  KOCR_TRY(up_conv("upconv4.conv.0", u3b, cat4, u4a));
  KOCR_TRY(mk(d.H2, d.W2, 32, &feat, true));
  RUN(launch_conv(ctx, L("upconv4.conv.3"), u4a, nullptr, nullptr, feat));
  done(u4a);
  // ---- head (detection.py:392-410), linear output ---------------------------------------
  KOCR_TRY(mk(d.H2, d.W2, 32, &k0, true));
  RUN(launch_conv(ctx, L("conv_cls.0"), feat, nullptr, nullptr, k0));
  done(feat);
  KOCR_TRY(mk(d.H2, d.W2, 32, &k1));
  RUN(launch_conv(ctx, L("conv_cls.2"), k0, nullptr, nullptr, k1));
  done(k0);
  KOCR_TRY(mk(d.H2, d.W2, 16, &k2));
  RUN(launch_conv(ctx, L("conv_cls.4"), k1, nullptr, nullptr, k2));
  done(k1);
  // conv_cls.6 + conv_cls.8 (1x1 16 -> 16 ReLU, 1x1 16 -> 2): one fused pass, no 16-channel round trip
  RUN(launch_head_tail(ctx, L("conv_cls.6"), L("conv_cls.8"), k2, d_heat));
  done(k2);
#undef RUN
  if (peak_out) *peak_out = pool.peak;
  return KOCR_OK;
}

}  // namespace

size_t craft_workspace_bytes(int N, int H, int W) {
  size_t peak = 0;
  static CraftNet empty;  // layer look-ups are never evaluated in the dry run
  craft_run<true>(nullptr, &empty, nullptr, KOCR_U8, N, H, W, nullptr, &peak);
  return peak + 4096;
}

int craft_forward(kocr_ctx* ctx, const void* d_img, int dtype, int N, int H, int W, float* d_heat) {
  CraftNet* net = ctx->craft;
  if (!net || !net->loaded) KOCR_FAIL(ctx, KOCR_ENOWEIGHTS, "kocr_craft_forward: call kocr_load_craft first");
  if (N <= 0) return KOCR_OK;
  if (H < 16 || W < 16) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_craft_forward: image smaller than 16x16");
  return craft_run<false>(ctx, net, d_img, dtype, N, H, W, d_heat, nullptr);
}
