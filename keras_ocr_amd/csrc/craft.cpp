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
#include "common.h"
#include <cmath>

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
}  // namespace

size_t craft_workspace_bytes(int N, int H, int W) {
  Dims d(H, W);
  const size_t n = (size_t)N, f = sizeof(float);
  size_t t = 0;
  const size_t P1 = n * d.H * d.W, P2 = n * d.H2 * d.W2, P4 = n * d.H4 * d.W4, P8 = n * d.H8 * d.W8,
               P16 = n * d.H16 * d.W16;
  t += al(P1 * 64 * f) * 2;                     // slice1.0, slice1.3
  t += al(P2 * 64 * f) + al(P2 * 128 * f);      // pool, slice1.7
  t += al(P2 * 192 * f);                        // cat4
  t += al(P4 * 128 * f) + al(P4 * 256 * f);     // pool, slice2.14
  t += al(P4 * 384 * f);                        // cat3
  t += al(P4 * 256 * f) + al(P8 * 256 * f);     // slice3.20, pool
  t += al(P8 * 512 * f);                        // slice3.24
  t += al(P8 * 768 * f);                        // cat2
  t += al(P8 * 512 * f) + al(P16 * 512 * f);    // slice4.30, pool
  t += al(P16 * 512 * f);                       // slice4.34
  t += al(P16 * 1536 * f);                      // cat1
  t += al(P16 * 512 * f) + al(P16 * 1024 * f);  // slice5.0, slice5.1
  t += al(P16 * 512 * f) + al(P16 * 256 * f);   // upconv1
  t += al(P8 * 256 * f) + al(P8 * 128 * f);     // upconv2
  t += al(P4 * 128 * f) + al(P4 * 64 * f);      // upconv3
  t += al(P2 * 64 * f) + al(P2 * 32 * f);       // upconv4
  t += al(P2 * 32 * f) * 2 + al(P2 * 16 * f) * 2;  // conv_cls
  return t + 4096;
}

int craft_forward(kocr_ctx* ctx, const void* d_img, int dtype, int N, int H, int W, float* d_heat) {
  CraftNet* net = ctx->craft;
  if (!net || !net->loaded) KOCR_FAIL(ctx, KOCR_ENOWEIGHTS, "kocr_craft_forward: call kocr_load_craft first");
  if (N <= 0) return KOCR_OK;
  if (H < 16 || W < 16) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_craft_forward: image smaller than 16x16");
  Dims d(H, W);
  KOCR_TRY(ctx->amax_begin());
  // every tensor produced by a convolution / pooling / up-sampling kernel carries a max-|x| slot
  // (Tensor::amax) so that an fp16-split consumer can pick its exact power-of-two input scale
  auto mk = [&](int h, int w, int c, Tensor* t) -> int {
    t->amax = ctx->amax_slot();
    t->N = N;
    t->H = h;
    t->W = w;
    t->C = c;
    t->cs = c;
    t->co = 0;
    t->p = (float*)ctx->ws_alloc((size_t)N * h * w * c * sizeof(float));
    if (!t->p) KOCR_FAIL(ctx, KOCR_ENOMEM, "kocr_craft_forward: workspace exhausted");
    return KOCR_OK;
  };
  auto conv = [&](const char* name, const Tensor& in, const Tensor& out) -> int {
    return launch_conv(ctx, net->L[name], in, nullptr, nullptr, out);
  };

  // conv + 2x2 max-pool: fused epilogue when the shape tiles (even H, W % 64 == 0), else two kernels.
  // need_full: the pre-pool tensor is consumed elsewhere (skip connection).
  auto conv_pool = [&](const char* name, const Tensor& in, const Tensor& full, const Tensor& pooled,
                       bool need_full) -> int {
    return launch_conv_pool(ctx, net->L[name], in, nullptr, nullptr, full, &pooled, need_full);
  };

  Tensor x0;
  x0.N = N;
  x0.H = H;
  x0.W = W;
  x0.C = 3;
  x0.cs = 3;
  x0.co = 0;
  x0.p = (dtype == KOCR_F32) ? (float*)d_img : nullptr;
  const uint8_t* u8 = (dtype == KOCR_U8) ? (const uint8_t*)d_img : nullptr;

  Tensor a1, a2, p1, b1, cat4, p2, c1, cat3, c3, p3, e1, cat2, f1, p4, g1, cat1, h0, h1;
  KOCR_TRY(mk(d.H, d.W, 64, &a1));
  KOCR_TRY(mk(d.H, d.W, 64, &a2));
  KOCR_TRY(mk(d.H2, d.W2, 64, &p1));
  KOCR_TRY(mk(d.H2, d.W2, 128, &b1));
  KOCR_TRY(mk(d.H2, d.W2, 192, &cat4));
  KOCR_TRY(mk(d.H4, d.W4, 128, &p2));
  KOCR_TRY(mk(d.H4, d.W4, 256, &c1));
  KOCR_TRY(mk(d.H4, d.W4, 384, &cat3));
  KOCR_TRY(mk(d.H4, d.W4, 256, &c3));
  KOCR_TRY(mk(d.H8, d.W8, 256, &p3));
  KOCR_TRY(mk(d.H8, d.W8, 512, &e1));
  KOCR_TRY(mk(d.H8, d.W8, 768, &cat2));
  KOCR_TRY(mk(d.H8, d.W8, 512, &f1));
  KOCR_TRY(mk(d.H16, d.W16, 512, &p4));
  KOCR_TRY(mk(d.H16, d.W16, 512, &g1));
  KOCR_TRY(mk(d.H16, d.W16, 1536, &cat1));
  KOCR_TRY(mk(d.H16, d.W16, 512, &h0));
  KOCR_TRY(mk(d.H16, d.W16, 1024, &h1));

  // ---- backbone (detection.py:312-335) ------------------------------------------------
  KOCR_TRY(launch_conv(ctx, net->L["basenet.slice1.0"], x0, u8, net->d_lut, a1));
  KOCR_TRY(conv_pool("basenet.slice1.3", a1, a2, p1, /*need_full=*/false));
  KOCR_TRY(conv("basenet.slice1.7", p1, b1));
  const Tensor s1 = cat4.slice(64, 128);
  KOCR_TRY(conv_pool("basenet.slice1.10", b1, s1, p2, /*need_full=*/true));  // s1 is a skip tensor
  KOCR_TRY(conv("basenet.slice2.14", p2, c1));
  const Tensor s2 = cat3.slice(128, 256);
  KOCR_TRY(conv("basenet.slice2.17", c1, s2));
  KOCR_TRY(conv_pool("basenet.slice3.20", s2, c3, p3, false));
  KOCR_TRY(conv("basenet.slice3.24", p3, e1));
  const Tensor s3 = cat2.slice(256, 512);
  KOCR_TRY(conv("basenet.slice3.27", e1, s3));
  KOCR_TRY(conv_pool("basenet.slice4.30", s3, f1, p4, false));
  KOCR_TRY(conv("basenet.slice4.34", p4, g1));
  const Tensor s4 = cat1.slice(1024, 512);
  KOCR_TRY(conv("basenet.slice4.37", g1, s4));
  // ---- slice5 (detection.py:365-378) ----------------------------------------------------
  KOCR_TRY(launch_maxpool3x3s1(ctx, s4, h0));
  KOCR_TRY(conv("basenet.slice5.1", h0, h1));
  KOCR_TRY(conv("basenet.slice5.2", h1, cat1.slice(0, 1024)));
  // ---- U-Net decoder (detection.py:380-390) ---------------------------------------------
  Tensor u1a, u1b, u2a, u2b, u3a, u3b, u4a, feat, k0, k1, k2;
  KOCR_TRY(mk(d.H16, d.W16, 512, &u1a));
  KOCR_TRY(mk(d.H16, d.W16, 256, &u1b));
  KOCR_TRY(conv("upconv1.conv.0", cat1, u1a));
  KOCR_TRY(conv("upconv1.conv.3", u1a, u1b));
  KOCR_TRY(launch_resize_bilinear(ctx, u1b, cat2.slice(0, 256)));
  KOCR_TRY(mk(d.H8, d.W8, 256, &u2a));
  KOCR_TRY(mk(d.H8, d.W8, 128, &u2b));
  KOCR_TRY(conv("upconv2.conv.0", cat2, u2a));
  KOCR_TRY(conv("upconv2.conv.3", u2a, u2b));
  KOCR_TRY(launch_resize_bilinear(ctx, u2b, cat3.slice(0, 128)));
  KOCR_TRY(mk(d.H4, d.W4, 128, &u3a));
  KOCR_TRY(mk(d.H4, d.W4, 64, &u3b));
  KOCR_TRY(conv("upconv3.conv.0", cat3, u3a));
  KOCR_TRY(conv("upconv3.conv.3", u3a, u3b));
  KOCR_TRY(launch_resize_bilinear(ctx, u3b, cat4.slice(0, 64)));
  KOCR_TRY(mk(d.H2, d.W2, 64, &u4a));
  KOCR_TRY(mk(d.H2, d.W2, 32, &feat));
  KOCR_TRY(conv("upconv4.conv.0", cat4, u4a));
  KOCR_TRY(conv("upconv4.conv.3", u4a, feat));
  // ---- head (detection.py:392-410), linear output ---------------------------------------
  KOCR_TRY(mk(d.H2, d.W2, 32, &k0));
  KOCR_TRY(mk(d.H2, d.W2, 32, &k1));
  KOCR_TRY(mk(d.H2, d.W2, 16, &k2));
  KOCR_TRY(conv("conv_cls.0", feat, k0));
  KOCR_TRY(conv("conv_cls.2", k0, k1));
  KOCR_TRY(conv("conv_cls.4", k1, k2));
  // conv_cls.6 + conv_cls.8 (1x1 16 -> 16 ReLU, 1x1 16 -> 2): one fused pass, no 16-channel round trip
  return launch_head_tail(ctx, net->L.at("conv_cls.6"), net->L.at("conv_cls.8"), k2, d_heat);
}
