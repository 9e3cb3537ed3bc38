// crnn.cpp — the CRNN recogniser graph (default build parameters, recognition.py:13-23).
//
// Follows recognition.build_model (recognition.py:187-333) layer for layer; names are the Keras
// layer names ("conv_1".."conv_7", "bn_3/5/7", "fc_9", "lstm_10[_back]", "lstm_11[_back]",
// "fc_12"; the unnamed STN localisation layers are "stn_conv_1/2", "stn_dense_1/2").
//   - every Conv2D / Dense runs on the MFMA implicit-GEMM kernel; conv bias + ReLU, and the
//     BatchNorm that FOLLOWS the ReLU at conv_3/5/7 (Keras default eps = 1e-3,
//     recognition.py:226-242), are the kernel's pre/post affine epilogue;
//   - conv_1..conv_7 and the two poolings run in the crop's NATURAL orientation [M,31,200,C]: the
//     reference first permutes to (200,31) and flips the 31-axis (recognition.py:215-216), i.e.
//     x'[w][j] = x[30-j][w]; a 3x3 convolution on x' with kernel K[a][b] equals one on x with
//     Knat[p][q] = K[q][2-p], and 'valid' 2x2 pooling of the flipped odd axis (31 -> 15 -> 7, last
//     row dropped) equals pooling rows (2i+1, 2i+2) of the natural tensor (first row dropped).
//     W = 200 / 100 / 50 is even, so the whole stack takes the Winograd F(2,3) kernel; one small
//     transpose+flip kernel after bn_7 restores the Keras layout for the STN and the recurrent part;
//   - Flatten / Reshape are free re-interpretations of the NHWC buffers;
//   - the forward and backward LSTM input projections of a layer are one GEMM (N = 2*4*128);
//     Add (recognition.py:305) is folded into the next projection by stacking its kernel over
//     the [forward | backward] channel halves; Concatenate (:319) is the same buffer.
#include "common.h"
#include <cmath>

int launch_crnn_input(kocr_ctx* ctx, const float* d_crops, float* d_x, int M, int Hc, int Wc);
int launch_crnn_to_keras(kocr_ctx* ctx, const Tensor& in, const Tensor& out);
int launch_crnn_conv1_cells(kocr_ctx* ctx, const ConvLayer& L, const float* d_crops, int M, int Hc, int Wc, const Tensor& out);
int launch_crnn_cells_to_keras(kocr_ctx* ctx, const Tensor& in, const Tensor& out);
int launch_stn_sample(kocr_ctx* ctx, const Tensor& x, const float* d_theta, const Tensor& out);
int launch_lstm(kocr_ctx* ctx, const float* d_xp, const float* d_Uf, const float* d_Ub, float* d_out, int M, int T);
size_t dense_splitk_workspace(int M, int K);
int launch_dense_splitk(kocr_ctx* ctx, const ConvLayer& L, const float* d_in, float* d_out, float* d_partial, int M);
int launch_ctc(kocr_ctx* ctx, const float* d_logits, int M, int T, int C, int discard, int* d_labels, float* d_probs);

struct CrnnNet {
  std::map<std::string, ConvLayer> L;
  float* U[4] = {nullptr, nullptr, nullptr, nullptr};  // recurrent kernels 10, 10_back, 11, 11_back
  int n_classes = 0;
  bool loaded = false;
  bool stn = true;   // build_params["stn"] (recognition.py:243-281): false when the weight set carries no stn_* tensors
  int discard = 2;   // build_params["rnn_steps_to_discard"] (recognition.py:328)
};

namespace {
constexpr int HC = 31, WC = 200, UNITS = 128, T = 50, DISCARD = 2;
// the crop batch as a cell grid (Tensor::cellW): CN crops side by side per image; cell = (HC + 1) x 208 at full resolution
// (one zero row on top, eight zero columns behind the crop), 16 x 104 and 8 x 52 after the two poolings.  208 CN, 104 CN and
// 52 CN are multiples of 64: the grid tiles as 4 rows x 64 columns at every level.
// CN = 16 cells per row; batches of at most 8 crops (Recognizer.recognize, a page with a few words) take 8 -- 208 * 8, 104 * 8 and
// 52 * 8 are multiples of 64 / 64 / 32 as well -- so that a single crop convolves 8 cells, not 16.  A crop's result
// does not depend on its cell or its neighbours (tests/test_cells_gpu.py), hence not on this choice.
constexpr int CN = 16, CELL_W = 208;
static inline int cells_per_row(int M) { return M <= 8 ? 8 : CN; }
const int kFilters[7] = {64, 128, 256, 256, 512, 512, 512};

struct Blob {
  const float* p;
  size_t numel;
};
}  // namespace

int crnn_classes(kocr_ctx* ctx) { return ctx->crnn && ctx->crnn->loaded ? ctx->crnn->n_classes : 0; }
int crnn_label_width(kocr_ctx* ctx) { return T - (ctx->crnn ? ctx->crnn->discard : DISCARD); }
int crnn_set_discard(kocr_ctx* ctx, int d) {
  if (d < 0 || d >= T) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_crnn_set_rnn_steps_to_discard: 0 <= steps < 50 (recognition.py:328)");
  if (!ctx->crnn) ctx->crnn = new CrnnNet();
  ctx->crnn->discard = d;
  return KOCR_OK;
}

int crnn_load(kocr_ctx* ctx, int n, const char* const* names, const float* const* data, const int64_t* shapes,
              const int* ranks) {
  std::map<std::string, Blob> blobs;
  for (int i = 0; i < n; ++i) {
    size_t ne = 1;
    for (int d = 0; d < ranks[i]; ++d) ne *= (size_t)shapes[i * 4 + d];
    blobs[names[i]] = Blob{data[i], ne};
  }
  auto need = [&](const std::string& k, size_t numel, const float** out) -> int {
    auto it = blobs.find(k);
    if (it == blobs.end()) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_load_crnn: missing tensor " + k);
    if (numel && it->second.numel != numel) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_load_crnn: tensor " + k + " has wrong size");
    *out = it->second.p;
    return KOCR_OK;
  };
  if (!ctx->crnn) ctx->crnn = new CrnnNet();
  CrnnNet* net = ctx->crnn;
  auto add = [&](const std::string& name, const float* w, int cin, int cout, int k, const float* bias, int relu,
                 const float* post_a, const float* post_b) -> int {
    ConvLayer& L = net->L[name];
    L.name = name;
    return prepare_conv(ctx, L, w, /*oihw=*/false, cin, cout, k, k, 1, nullptr, bias, relu, post_a, post_b);
  };
  int cin = 1;
  for (int i = 1; i <= 7; ++i) {
    const int cout = kFilters[i - 1];
    const std::string nm = "conv_" + std::to_string(i);
    const float *w, *b;
    KOCR_TRY(need(nm + "/kernel", (size_t)9 * cin * cout, &w));
    KOCR_TRY(need(nm + "/bias", cout, &b));
    // natural-orientation kernel: Knat[p][q][c][o] = K[q][2-p][c][o]  (HWIO)
    std::vector<float> wn((size_t)9 * cin * cout);
    for (int pp = 0; pp < 3; ++pp)
      for (int q = 0; q < 3; ++q)
        memcpy(&wn[((size_t)pp * 3 + q) * cin * cout], &w[((size_t)q * 3 + (2 - pp)) * cin * cout],
               sizeof(float) * cin * cout);
    w = wn.data();
    std::vector<float> qa, qb;
    if (i == 3 || i == 5 || i == 7) {
      const std::string bn = "bn_" + std::to_string(i);
      const float *g, *be, *mu, *var;
      KOCR_TRY(need(bn + "/gamma", cout, &g));
      KOCR_TRY(need(bn + "/beta", cout, &be));
      KOCR_TRY(need(bn + "/moving_mean", cout, &mu));
      KOCR_TRY(need(bn + "/moving_variance", cout, &var));
      qa.resize(cout);
      qb.resize(cout);
      for (int o = 0; o < cout; ++o) {
        qa[o] = g[o] / std::sqrt(var[o] + 1e-3f);
        qb[o] = be[o] - mu[o] * qa[o];
      }
    }
    KOCR_TRY(add(nm, w, cin, cout, 3, b, 1, qa.empty() ? nullptr : qa.data(), qb.empty() ? nullptr : qb.data()));
    cin = cout;
  }
  // build_params["stn"] = False (recognition.py:243): a weight set without the localisation network -> no transformer
  net->stn = blobs.count("stn_conv_1/kernel") != 0;
  if (net->stn) {
    const float *w, *b;
    KOCR_TRY(need("stn_conv_1/kernel", (size_t)25 * 512 * 16, &w));
    KOCR_TRY(need("stn_conv_1/bias", 16, &b));
    KOCR_TRY(add("stn_conv_1", w, 512, 16, 5, b, 1, nullptr, nullptr));
    KOCR_TRY(need("stn_conv_2/kernel", (size_t)25 * 16 * 32, &w));
    KOCR_TRY(need("stn_conv_2/bias", 32, &b));
    KOCR_TRY(add("stn_conv_2", w, 16, 32, 5, b, 1, nullptr, nullptr));
    KOCR_TRY(need("stn_dense_1/kernel", (size_t)11200 * 64, &w));
    KOCR_TRY(need("stn_dense_1/bias", 64, &b));
    KOCR_TRY(add("stn_dense_1", w, 11200, 64, 1, b, 1, nullptr, nullptr));
    KOCR_TRY(need("stn_dense_2/kernel", (size_t)64 * 6, &w));
    KOCR_TRY(need("stn_dense_2/bias", 6, &b));
    KOCR_TRY(add("stn_dense_2", w, 64, 6, 1, b, 0, nullptr, nullptr));
  }
  {
    const float *w, *b;
    KOCR_TRY(need("fc_9/kernel", (size_t)3584 * UNITS, &w));
    KOCR_TRY(need("fc_9/bias", UNITS, &b));
    KOCR_TRY(add("fc_9", w, 3584, UNITS, 1, b, 1, nullptr, nullptr));
  }
  // LSTM layers: merged [fwd | back] input projection; layer 2 reads [f | b] = 256 channels and
  // its kernel is stacked twice so that (f + b) @ W == [f | b] @ [W; W]   (recognition.py:305)
  const char* lname[4] = {"lstm_10", "lstm_10_back", "lstm_11", "lstm_11_back"};
  for (int layer = 0; layer < 2; ++layer) {
    const float *wf, *wb, *bf, *bb;
    KOCR_TRY(need(std::string(lname[2 * layer]) + "/kernel", (size_t)UNITS * 4 * UNITS, &wf));
    KOCR_TRY(need(std::string(lname[2 * layer + 1]) + "/kernel", (size_t)UNITS * 4 * UNITS, &wb));
    KOCR_TRY(need(std::string(lname[2 * layer]) + "/bias", 4 * UNITS, &bf));
    KOCR_TRY(need(std::string(lname[2 * layer + 1]) + "/bias", 4 * UNITS, &bb));
    const int kin = layer == 0 ? UNITS : 2 * UNITS;
    std::vector<float> w((size_t)kin * 8 * UNITS), b(8 * UNITS);
    for (int k = 0; k < kin; ++k)
      for (int o = 0; o < 4 * UNITS; ++o) {
        w[(size_t)k * 8 * UNITS + o] = wf[(size_t)(k % UNITS) * 4 * UNITS + o];
        w[(size_t)k * 8 * UNITS + 4 * UNITS + o] = wb[(size_t)(k % UNITS) * 4 * UNITS + o];
      }
    for (int o = 0; o < 4 * UNITS; ++o) {
      b[o] = bf[o];
      b[4 * UNITS + o] = bb[o];
    }
    KOCR_TRY(add(layer == 0 ? "lstm_10_xproj" : "lstm_11_xproj", w.data(), kin, 8 * UNITS, 1, b.data(), 0, nullptr, nullptr));
    for (int d = 0; d < 2; ++d) {
      const float* u;
      KOCR_TRY(need(std::string(lname[2 * layer + d]) + "/recurrent_kernel", (size_t)UNITS * 4 * UNITS, &u));
      KOCR_TRY(ctx->upload(&net->U[2 * layer + d], std::vector<float>(u, u + (size_t)UNITS * 4 * UNITS)));
    }
  }
  {
    auto it = blobs.find("fc_12/bias");
    if (it == blobs.end()) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_load_crnn: missing tensor fc_12/bias");
    net->n_classes = (int)it->second.numel;
    const float *w, *b;
    KOCR_TRY(need("fc_12/kernel", (size_t)2 * UNITS * net->n_classes, &w));
    KOCR_TRY(need("fc_12/bias", net->n_classes, &b));
    KOCR_TRY(add("fc_12", w, 2 * UNITS, net->n_classes, 1, b, 0, nullptr, nullptr));
  }
  net->loaded = true;
  return KOCR_OK;
}

void crnn_free(kocr_ctx* ctx) {
  delete ctx->crnn;
  ctx->crnn = nullptr;
}

size_t crnn_workspace_bytes(int M, int n_classes) {
  const size_t m = (size_t)M, f = sizeof(float);
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  size_t t = 0;
  {  // the cell-grid schedule instead of the dense conv stack below (whichever is larger): whole rows of CN cells
    const size_t mc = (m + CN - 1) / CN * CN;
    const size_t cells = al(mc * (HC + 1) * CELL_W * 64 * f) + al(mc * (HC + 1) * CELL_W * 128 * f) +
                         al(mc * 16 * (CELL_W / 2) * 256 * f) * 2 + al(mc * 8 * (CELL_W / 4) * 512 * f) * 3;
    const size_t dense = al(m * WC * HC * 64 * f) + al(m * WC * HC * 128 * f) + al(m * WC * HC * 256 * f) +
                         al(m * 100 * 15 * 256 * f) * 2 + al(m * 100 * 15 * 512 * f) + al(m * 52 * 7 * 512 * f) * 3;
    if (cells > dense) t += cells - dense;
  }
  t += al(m * WC * HC * 64 * f) + al(m * WC * HC * 128 * f) + al(m * WC * HC * 256 * f);
  t += al(m * 100 * 15 * 256 * f) * 2 + al(m * 100 * 15 * 512 * f);
  t += al(m * 52 * 7 * 512 * f) * 5;                              // p5, c6, c7 (both layouts; width padded to 52), stn
  t += al(m * 50 * 7 * 16 * f) + al(m * 50 * 7 * 32 * f) + al(m * 64 * f) + al(m * 6 * f);
  t += al(dense_splitk_workspace(M, 50 * 7 * 32));
  t += al(m * T * UNITS * f) + al(m * T * 8 * UNITS * f) + al(m * T * 2 * UNITS * f) * 2;
  t += al(m * T * (size_t)n_classes * f);
  return t + 8192;
}

// d_crops: device [M][31][200]; d_labels: device [M][LW]; d_probs: device [M][LW][C] or null, LW = crnn_label_width (48)
int crnn_forward(kocr_ctx* ctx, const float* d_crops, int M, int* d_labels, float* d_probs) {
  CrnnNet* net = ctx->crnn;
  if (!net || !net->loaded) KOCR_FAIL(ctx, KOCR_ENOWEIGHTS, "kocr_crnn_forward: call kocr_load_crnn first");
  if (M <= 0) return KOCR_OK;
  KOCR_TRY(ctx->amax_begin());
  // Arithmetic: conv_2 ... conv_5 take the flattened-pixel F(4,3) kernel in the context's mode -- fp16x2 with one input scale
  // per CROP (the "images" of this batch), so a crop's result does not depend on its neighbours in the batch; the
  // reduced-precision KOCR_SPLIT_F16X1 is a detector-only mode: the recogniser then runs fp16x2 (its decoded strings hang on
  // top-1 margins).  conv_6 / conv_7 (width 50) stay on the exact bf16x3 F(2,3) kernel.
  struct ModeGuard {
    kocr_ctx* c;
    int old;
    explicit ModeGuard(kocr_ctx* ctx) : c(ctx), old(ctx->split_mode) {
      if (c->split_mode == KOCR_SPLIT_F16X1) c->split_mode = KOCR_SPLIT_F16X2;
    }
    ~ModeGuard() { c->split_mode = old; }
  } mode_guard(ctx);
  auto mk = [&](int n, int h, int w, int c, Tensor* t) -> int {
    t->amax = nullptr;
    t->N = n;
    t->H = h;
    t->W = w;
    t->C = c;
    t->cs = c;
    t->co = 0;
    t->p = (float*)ctx->ws_alloc((size_t)n * h * w * c * sizeof(float));
    if (!t->p) KOCR_FAIL(ctx, KOCR_ENOMEM, "kocr_crnn_forward: workspace exhausted");
    return KOCR_OK;
  };
  auto view = [](const Tensor& t, int n, int h, int w, int c) {
    Tensor v = t;
    v.N = n;
    v.H = h;
    v.W = w;
    v.C = c;
    v.cs = c;
    v.co = 0;
    return v;
  };
  auto conv = [&](const char* name, const Tensor& in, const Tensor& out) -> int {
    return launch_conv(ctx, net->L[name], in, nullptr, nullptr, out);
  };
  Tensor x0, c1, c2, c3, p3, c4, c5, p5, c6, c7n, c7, s1, s2, d1, th, st, f9, xp, r1, r2, lg;
  // Round 5: in the fp16 arithmetic the conv stack runs on a CELL GRID (Tensor::cellW): the crops side by side, CN per image,
  // with zero gutters, so that conv_2 ... conv_7 take the vertical-reuse F(4,3) kernel (conv_w43vh_kernel MODE 2) with both
  // poolings fused into conv_3 / conv_5 (whose full-resolution outputs are never written) and the max-|x| slots maintained
  // per cell by every producer.  One path for every M (a crop's result must not depend on the batch it comes in).
  bool cells = true;
  for (int i = 2; i <= 7; ++i) cells = cells && w43_cells_ok(ctx, net->L["conv_" + std::to_string(i)]);
  if (cells) {
    const int cn = cells_per_row(M);
    const int R = (M + cn - 1) / cn;
    auto mkc = [&](int hc, int wc, int wv, int c, bool alloc, Tensor* t) -> int {
      t->N = R;
      t->H = hc;
      t->W = cn * wc;
      t->C = t->cs = c;
      t->co = 0;
      t->cellW = wc;
      t->cellWv = wv;
      t->amax = ctx->amax_slots(R * cn);
      if (!t->amax) KOCR_FAIL(ctx, KOCR_ECAPACITY, "kocr_crnn_forward: out of max-|x| slots");
      t->p = alloc ? (float*)ctx->ws_alloc((size_t)R * hc * cn * wc * c * sizeof(float)) : nullptr;
      if (alloc && !t->p) KOCR_FAIL(ctx, KOCR_ENOMEM, "kocr_crnn_forward: workspace exhausted");
      return KOCR_OK;
    };
    KOCR_TRY(mkc(HC + 1, CELL_W, WC, 64, true, &c1));
    KOCR_TRY(launch_crnn_conv1_cells(ctx, net->L["conv_1"], d_crops, M, HC, WC, c1));
    KOCR_TRY(mkc(HC + 1, CELL_W, WC, 128, true, &c2));
    KOCR_TRY(conv("conv_2", c1, c2));
    KOCR_TRY(mkc(HC + 1, CELL_W, WC, 256, false, &c3));  // conv_3's full-resolution output is only ever pooled
    KOCR_TRY(mkc(16, CELL_W / 2, WC / 2, 256, true, &p3));
    KOCR_TRY(launch_conv_pool(ctx, net->L["conv_3"], c2, nullptr, nullptr, c3, &p3, /*need_full=*/false));  // ReLU, bn_3, pool
    KOCR_TRY(mkc(16, CELL_W / 2, WC / 2, 256, true, &c4));
    KOCR_TRY(conv("conv_4", p3, c4));
    KOCR_TRY(mkc(16, CELL_W / 2, WC / 2, 512, false, &c5));
    KOCR_TRY(mkc(8, CELL_W / 4, WC / 4, 512, true, &p5));
    KOCR_TRY(launch_conv_pool(ctx, net->L["conv_5"], c4, nullptr, nullptr, c5, &p5, /*need_full=*/false));
    KOCR_TRY(mkc(8, CELL_W / 4, WC / 4, 512, true, &c6));
    KOCR_TRY(conv("conv_6", p5, c6));
    KOCR_TRY(mkc(8, CELL_W / 4, WC / 4, 512, true, &c7n));
    c7n.amax = nullptr;  // nothing downstream reads its scale
    KOCR_TRY(conv("conv_7", c6, c7n));
    KOCR_TRY(mk(M, WC / 4, HC / 4, 512, &c7));
    KOCR_TRY(launch_crnn_cells_to_keras(ctx, c7n, c7));
  } else {
  // conv stack in the crop's natural orientation (see the header): [M,31,200,C]
  x0.N = M;
  x0.H = HC;
  x0.W = WC;
  x0.C = x0.cs = 1;
  x0.co = 0;
  x0.p = const_cast<float*>(d_crops);
  KOCR_TRY(mk(M, HC, WC, 64, &c1));
  c1.amax = ctx->amax_slots(M);  // per-crop max |x| for the fp16 consumer (conv_2); nullptr in bf16x3 mode
  KOCR_TRY(conv("conv_1", x0, c1));
  KOCR_TRY(mk(M, HC, WC, 128, &c2));
  c2.amax = ctx->amax_slots(M);
  KOCR_TRY(conv("conv_2", c1, c2));
  KOCR_TRY(mk(M, HC, WC, 256, &c3));
  c3.amax = ctx->amax_slots(M);
  KOCR_TRY(conv("conv_3", c2, c3));  // ReLU then bn_3
  KOCR_TRY(mk(M, HC / 2, WC / 2, 256, &p3));
  p3.amax = ctx->amax_slots(M);
  KOCR_TRY(launch_maxpool2x2(ctx, c3, p3, /*row_off=*/1));
  KOCR_TRY(mk(M, HC / 2, WC / 2, 256, &c4));
  c4.amax = ctx->amax_slots(M);
  KOCR_TRY(conv("conv_4", p3, c4));
  KOCR_TRY(mk(M, HC / 2, WC / 2, 512, &c5));
  // conv_6 / conv_7 work on 7 x 50 maps: 50 is not a multiple of 4, so the F(4,3) kernels do not take them as they are.
  // In the fp16 modes the three tensors around them are stored 52 wide with two zero columns (Tensor::Wv = 50): the pooling
  // kernel and the flattened fp16 F(4,3) kernel write the zeros, which are exactly the 'same' padding of column 49's right
  // neighbours -- 4 % more pixels on a kernel 1.4x faster than the F(2,3) bf16x3 one that takes the 50-wide tensors.
  // (the same predicate the dispatcher uses: a 52-wide tensor through any other kernel would get convolution values written
  //  into its padding columns -- launch_conv now refuses that)
  const bool pad52 = w43_flat_h_ok(ctx, net->L["conv_6"]) && w43_flat_h_ok(ctx, net->L["conv_7"]);
  const int W6 = pad52 ? 52 : WC / 4;
  if (pad52) c5.amax = ctx->amax_slots(M);
  KOCR_TRY(conv("conv_5", c4, c5));
  KOCR_TRY(mk(M, HC / 4, W6, 512, &p5));
  p5.Wv = WC / 4;
  if (pad52) p5.amax = ctx->amax_slots(M);
  KOCR_TRY(launch_maxpool2x2(ctx, c5, p5, /*row_off=*/1));
  KOCR_TRY(mk(M, HC / 4, W6, 512, &c6));
  c6.Wv = WC / 4;
  if (pad52) c6.amax = ctx->amax_slots(M);
  KOCR_TRY(conv("conv_6", p5, c6));
  KOCR_TRY(mk(M, HC / 4, W6, 512, &c7n));
  c7n.Wv = WC / 4;
  KOCR_TRY(conv("conv_7", c6, c7n));
  // back to the Keras layout (M, 50, 7, 512) for the STN and everything after it
  KOCR_TRY(mk(M, WC / 4, HC / 4, 512, &c7));
  KOCR_TRY(launch_crnn_to_keras(ctx, c7n, c7));
  }
  // STN (recognition.py:268-281)
  if (!net->stn) {
    st = c7;
  } else {
  KOCR_TRY(mk(M, WC / 4, HC / 4, 16, &s1));
  KOCR_TRY(conv("stn_conv_1", c7, s1));
  KOCR_TRY(mk(M, WC / 4, HC / 4, 32, &s2));
  KOCR_TRY(conv("stn_conv_2", s1, s2));
  KOCR_TRY(mk(M, 1, 1, 64, &d1));
  {
    // Dense(64) over 11 200 features: split-K (crnn_kernels.hip); the weight rows must be in plain k order
    const ConvLayer& Ld = net->L["stn_dense_1"];
    const bool no_sk = !ctx->sw.dense_splitk;
    float* part = no_sk ? nullptr : (float*)ctx->ws_alloc(dense_splitk_workspace(M, 11200));
    if (part && Ld.Cout == 64 && Ld.Cin == 11200)
      KOCR_TRY(launch_dense_splitk(ctx, Ld, s2.p, d1.p, part, M));
    else
      KOCR_TRY(conv("stn_dense_1", view(s2, M, 1, 1, 11200), d1));
  }
  KOCR_TRY(mk(M, 1, 1, 6, &th));
  KOCR_TRY(conv("stn_dense_2", d1, th));
  KOCR_TRY(mk(M, WC / 4, HC / 4, 512, &st));
  KOCR_TRY(launch_stn_sample(ctx, c7, th.p, st));
  }
  // Reshape + fc_9 (recognition.py:282-290)
  KOCR_TRY(mk(M, T, 1, UNITS, &f9));
  KOCR_TRY(conv("fc_9", view(st, M, T, 1, 7 * 512), f9));
  // BiLSTM x2 (recognition.py:292-319)
  KOCR_TRY(mk(M, T, 1, 8 * UNITS, &xp));
  KOCR_TRY(mk(M, T, 1, 2 * UNITS, &r1));
  KOCR_TRY(mk(M, T, 1, 2 * UNITS, &r2));
  KOCR_TRY(conv("lstm_10_xproj", f9, xp));
  KOCR_TRY(launch_lstm(ctx, xp.p, net->U[0], net->U[1], r1.p, M, T));
  KOCR_TRY(conv("lstm_11_xproj", r1, xp));
  KOCR_TRY(launch_lstm(ctx, xp.p, net->U[2], net->U[3], r2.p, M, T));
  // fc_12 + softmax + decode (recognition.py:321-328, 169-184)
  KOCR_TRY(mk(M, T, 1, net->n_classes, &lg));
  KOCR_TRY(conv("fc_12", r2, lg));
  KOCR_TRY(launch_ctc(ctx, lg.p, M, T, net->n_classes, net->discard, d_labels, d_probs));
  return KOCR_OK;
}
