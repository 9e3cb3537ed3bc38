// api.cpp — the extern "C" surface declared in include/kocr.h (context, memory, profiler,
// and the entry points that wrap the graphs).  No torch types: plain pointers and sizes.
#include "common.h"
#include <cmath>
#include <algorithm>
#include <mutex>
#include <set>

// ---------------------------------------------------------------------------------------
// ctx internals
// ---------------------------------------------------------------------------------------
int arena_reserve(kocr_ctx* ctx, Arena& a, size_t bytes) {
  ctx->last_pl.valid = false;  // whoever sizes an arena is about to overwrite it (kocr_pipeline re-validates at its end)
  if (bytes <= a.cap) return KOCR_OK;
  if (a.base) {
    KOCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    KOCR_HIP(ctx, hipFree(a.base));
    a.base = nullptr;
    a.cap = 0;
  }
  const size_t want = bytes + (bytes >> 3) + 4096;
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, want);
  if (e != hipSuccess) {
    ctx->set_err("workspace hipMalloc(" + std::to_string(want) + " B) failed: " + hipGetErrorString(e));
    return KOCR_ENOMEM;
  }
  a.base = (char*)p;
  a.cap = want;
  a.off = 0;
  return KOCR_OK;
}

int kocr_ctx::ws_reserve(size_t bytes) { return arena_reserve(this, ws, bytes); }

int kocr_ctx::amax_begin() {
  if (!d_amax) {
    void* d = nullptr;
    KOCR_TRY(dev_alloc(&d, AMAX_SLOTS * sizeof(unsigned)));
    d_amax = (unsigned*)d;
  }
  amax_used = 0;
  KOCR_HIP(this, hipMemsetAsync(d_amax, 0, AMAX_SLOTS * sizeof(unsigned), stream));
  return KOCR_OK;
}

// Only the fp16 kernels read a tensor's tracked max |x| (their exact power-of-two input scale); in bf16x3 mode no slot is
// handed out, so the producers' epilogues skip the reduction and the atomic altogether.
unsigned* kocr_ctx::amax_slots(int n) {
  if (split_mode == KOCR_SPLIT_BF16X3 || !d_amax || n <= 0 || amax_used + n > AMAX_SLOTS) return nullptr;
  unsigned* s = d_amax + amax_used;
  amax_used += n;
  return s;
}

int kocr_ctx::dev_alloc(void** out, size_t bytes) {
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes ? bytes : 4);
  if (e != hipSuccess) {
    set_err(std::string("hipMalloc failed: ") + hipGetErrorString(e));
    return KOCR_ENOMEM;
  }
  owned.push_back(p);
  *out = p;
  return KOCR_OK;
}

void kocr_note_dispatch(const char* family, const ConvLayer& L, const Tensor& in) {
  static const char* path = getenv("KOCR_DISPATCH_LOG");
  if (!path) return;
  static std::mutex mu;
  static std::set<std::string> seen;
  char line[256];
  snprintf(line, sizeof line, "%-14s %-28s N %d H %d W %d Cin %d Cout %d k %dx%d dil %d%s", family, L.name.c_str(), in.N, in.H, in.W, L.Cin,
           L.Cout, L.KH, L.KW, L.dil, in.cellW ? " cells" : "");
  std::lock_guard<std::mutex> g(mu);
  if (!seen.insert(line).second) return;
  if (FILE* f = fopen(path, "a")) {
    fprintf(f, "%s\n", line);
    fclose(f);
  }
}

void kocr_ctx::release(void* p) {
  if (!p) return;
  for (size_t i = owned.size(); i-- > 0;)
    if (owned[i] == p) {
      owned.erase(owned.begin() + (long)i);
      hipFree(p);
      return;
    }
}

int kocr_ctx::upload(float** out, const std::vector<float>& host) {
  void* p = nullptr;
  KOCR_TRY(dev_alloc(&p, host.size() * sizeof(float)));
  KOCR_HIP(this, hipMemcpy(p, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
  *out = (float*)p;
  return KOCR_OK;
}

hipEvent_t kocr_ctx::get_event() {
  if (!ev_pool.empty()) {
    hipEvent_t e = ev_pool.back();
    ev_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}

void kocr_ctx::prof_begin(const char* name, double flops, double bytes) {
  // bounded: a caller that leaves profiling on without ever asking for the report pays one synchronisation per 4096
  // launches instead of an event pool that grows without limit
  if (pending.size() >= 4096) prof_flush();
  Pending pd;
  pd.name = name;
  pd.a = get_event();
  pd.b = get_event();
  hipEventRecord(pd.a, stream);
  pending.push_back(pd);
  ProfRow& r = prof[name];
  r.launches += 1;
  r.flops += flops;
  r.bytes += bytes;
}

void kocr_ctx::prof_end() { hipEventRecord(pending.back().b, stream); }

int kocr_ctx::prof_flush() {
  if (pending.empty()) return KOCR_OK;
  KOCR_HIP(this, hipStreamSynchronize(stream));
  for (Pending& pd : pending) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, pd.a, pd.b);
    prof[pd.name].ms += ms;
    ev_pool.push_back(pd.a);
    ev_pool.push_back(pd.b);
  }
  pending.clear();
  return KOCR_OK;
}

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
extern "C" {

int kocr_create(kocr_ctx** out, int hip_device) {
  if (!out) return KOCR_EINVAL;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || hip_device < 0 || hip_device >= count) return KOCR_EHIP;
  if (hipSetDevice(hip_device) != hipSuccess) return KOCR_EHIP;
  kocr_ctx* c = new kocr_ctx();
  if (const char* e = getenv("KOCR_SPLIT"))
    c->split_mode = (!strcmp(e, "f16") || !strcmp(e, "fp16") || !strcmp(e, "f16x2")) ? KOCR_SPLIT_F16X2
                    : (!strcmp(e, "f16x1") || !strcmp(e, "fast"))                     ? KOCR_SPLIT_F16X1
                                                                                      : KOCR_SPLIT_BF16X3;
  {
    auto on = [](const char* name, bool dflt) {
      const char* e = getenv(name);
      return e ? atoi(e) != 0 : dflt;
    };
    c->sw.dense_splitk = on("KOCR_DENSE_SPLITK", true);
    c->sw.lstm16 = on("KOCR_LSTM16", true);
    c->sw.hs16 = on("KOCR_HS16", true);
    c->sw.k5 = on("KOCR_K5", true);
    c->sw.up2x = on("KOCR_UP2X", true);
    c->sw.w43h = on("KOCR_W43H", true);
  }
  if (const char* e = getenv("KOCR_LINFOLD")) c->opt_linfold = atoi(e) != 0;
  if (const char* e = getenv("KOCR_UPFOLD")) c->opt_upfold = atoi(e) != 0;
  c->device = hip_device;
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return KOCR_EHIP;
  }
  c->stream = c->own_stream;
  *out = c;
  return KOCR_OK;
}

void kocr_destroy(kocr_ctx* ctx) {
  if (!ctx) return;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  craft_free(ctx);
  crnn_free(ctx);
  for (void* p : ctx->owned) hipFree(p);
  for (Arena* a : {&ctx->ws, &ctx->pp, &ctx->pp2, &ctx->io, &ctx->pl, &ctx->bx})
    if (a->base) hipFree(a->base);
  for (auto& pd : ctx->pending) {
    hipEventDestroy(pd.a);
    hipEventDestroy(pd.b);
  }
  for (hipEvent_t e : ctx->ev_pool) hipEventDestroy(e);
  hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

const char* kocr_last_error(const kocr_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int kocr_set_stream(kocr_ctx* ctx, void* hip_stream) {
  if (!ctx) return KOCR_EINVAL;
  KOCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  return KOCR_OK;
}

int kocr_synchronize(kocr_ctx* ctx) {
  if (!ctx) return KOCR_EINVAL;
  KOCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return KOCR_OK;
}

int kocr_device_alloc(kocr_ctx* ctx, void** out, uint64_t bytes) {
  if (!ctx || !out) return KOCR_EINVAL;
  KOCR_HIP(ctx, hipSetDevice(ctx->device));
  KOCR_HIP(ctx, hipMalloc(out, bytes ? bytes : 4));
  return KOCR_OK;
}

int kocr_device_free(kocr_ctx* ctx, void* p) {
  if (!ctx) return KOCR_EINVAL;
  KOCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  KOCR_HIP(ctx, hipFree(p));
  return KOCR_OK;
}

int kocr_memcpy_h2d(kocr_ctx* ctx, void* dst, const void* src, uint64_t bytes) {
  if (!ctx) return KOCR_EINVAL;
  KOCR_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  KOCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return KOCR_OK;
}

int kocr_memcpy_d2h(kocr_ctx* ctx, void* dst, const void* src, uint64_t bytes) {
  if (!ctx) return KOCR_EINVAL;
  KOCR_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  KOCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return KOCR_OK;
}

int kocr_load_craft(kocr_ctx* ctx, int n, const char* const* names, const float* const* data,
                    const int64_t* shapes, const int* ranks) {
  if (!ctx || n <= 0 || !names || !data || !shapes || !ranks) return KOCR_EINVAL;
  KOCR_HIP(ctx, hipSetDevice(ctx->device));
  return craft_load(ctx, n, names, data, shapes, ranks);
}

int kocr_craft_forward(kocr_ctx* ctx, const void* img, int dtype, int N, int H, int W, float* heat,
                       int micro_batch, int on_device) {
  if (!ctx) return KOCR_EINVAL;
  if (N < 0 || (N > 0 && (!img || !heat))) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_craft_forward: null buffer");
  if (dtype != KOCR_U8 && dtype != KOCR_F32) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_craft_forward: bad dtype");
  if (N == 0) return KOCR_OK;
  KOCR_HIP(ctx, hipSetDevice(ctx->device));
  int mb = micro_batch > 0 ? micro_batch : 32;  // Keras predict default batch_size (detection.py:779)
  // keep the per-micro-batch workspace under ~64 GiB of the 288 GB HBM
  while (mb > 1 && craft_workspace_bytes(mb, H, W) > ((size_t)96 << 30)) mb = (mb + 1) / 2;
  mb = std::min(mb, N);
  const size_t esz = dtype == KOCR_U8 ? 1 : 4;
  const size_t in_img = (size_t)H * W * 3 * esz;
  const size_t out_img = (size_t)(H / 2) * (W / 2) * 2 * sizeof(float);
  size_t need = craft_workspace_bytes(mb, H, W);
  if (!on_device) need += (in_img + out_img) * mb + 1024;
  KOCR_TRY(ctx->ws_reserve(need));
  for (int s = 0; s < N; s += mb) {
    const int nb = std::min(mb, N - s);
    ctx->ws_reset();
    const void* d_in;
    float* d_out;
    if (on_device) {
      d_in = (const char*)img + (size_t)s * in_img;
      d_out = (float*)((char*)heat + (size_t)s * out_img);
    } else {
      void* di = ctx->ws_alloc(in_img * nb);
      void* dout = ctx->ws_alloc(out_img * nb);
      if (!di || !dout) KOCR_FAIL(ctx, KOCR_ENOMEM, "kocr_craft_forward: workspace exhausted");
      KOCR_HIP(ctx, hipMemcpyAsync(di, (const char*)img + (size_t)s * in_img, in_img * nb,
                                   hipMemcpyHostToDevice, ctx->stream));
      d_in = di;
      d_out = (float*)dout;
    }
    KOCR_TRY(craft_forward(ctx, d_in, dtype, nb, H, W, d_out));
    if (!on_device) {
      KOCR_HIP(ctx, hipMemcpyAsync((char*)heat + (size_t)s * out_img, d_out, out_img * nb,
                                   hipMemcpyDeviceToHost, ctx->stream));
      KOCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
  }
  return KOCR_OK;
}

int kocr_load_crnn(kocr_ctx* ctx, int n, const char* const* names, const float* const* data,
                   const int64_t* shapes, const int* ranks) {
  if (!ctx || n <= 0 || !names || !data || !shapes || !ranks) return KOCR_EINVAL;
  KOCR_HIP(ctx, hipSetDevice(ctx->device));
  return crnn_load(ctx, n, names, data, shapes, ranks);
}

int kocr_crnn_classes(kocr_ctx* ctx) { return ctx ? crnn_classes(ctx) : KOCR_EINVAL; }
int kocr_crnn_label_width(kocr_ctx* ctx) { return ctx ? crnn_label_width(ctx) : KOCR_EINVAL; }
int kocr_crnn_set_rnn_steps_to_discard(kocr_ctx* ctx, int steps) { return ctx ? crnn_set_discard(ctx, steps) : KOCR_EINVAL; }

int kocr_crnn_forward(kocr_ctx* ctx, const float* crops, int M, int32_t* labels, float* probs, int on_device) {
  if (!ctx) return KOCR_EINVAL;
  if (M < 0 || (M > 0 && (!crops || !labels))) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_crnn_forward: null buffer");
  const int C = crnn_classes(ctx);
  if (C == 0) KOCR_FAIL(ctx, KOCR_ENOWEIGHTS, "kocr_crnn_forward: call kocr_load_crnn first");
  if (M == 0) return KOCR_OK;
  KOCR_HIP(ctx, hipSetDevice(ctx->device));
  const int mb = std::min(M, 1024);
  const int LW = crnn_label_width(ctx);
  const size_t cb = (size_t)31 * 200 * sizeof(float), lb = LW * sizeof(int32_t), pb = (size_t)LW * C * sizeof(float);
  KOCR_TRY(ctx->ws_reserve(crnn_workspace_bytes(mb, C) + (on_device ? 0 : (cb + lb + pb) * mb + 2048)));
  for (int s = 0; s < M; s += mb) {
    const int nb = std::min(mb, M - s);
    ctx->ws_reset();
    const float* d_c = crops + (size_t)s * 31 * 200;
    int32_t* d_l = labels + (size_t)s * LW;
    float* d_p = probs ? probs + (size_t)s * LW * C : nullptr;
    if (!on_device) {
      float* dc = (float*)ctx->ws_alloc(cb * nb);
      d_l = (int32_t*)ctx->ws_alloc(lb * nb);
      d_p = probs ? (float*)ctx->ws_alloc(pb * nb) : nullptr;
      if (!dc || !d_l || (probs && !d_p)) KOCR_FAIL(ctx, KOCR_ENOMEM, "kocr_crnn_forward: workspace exhausted");
      KOCR_HIP(ctx, hipMemcpyAsync(dc, d_c, cb * nb, hipMemcpyHostToDevice, ctx->stream));
      d_c = dc;
    }
    KOCR_TRY(crnn_forward(ctx, d_c, nb, d_l, d_p));
    if (!on_device) {
      KOCR_HIP(ctx, hipMemcpyAsync(labels + (size_t)s * LW, d_l, lb * nb, hipMemcpyDeviceToHost, ctx->stream));
      if (probs)
        KOCR_HIP(ctx, hipMemcpyAsync(probs + (size_t)s * LW * C, d_p, pb * nb, hipMemcpyDeviceToHost, ctx->stream));
      KOCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
  }
  return KOCR_OK;
}

int kocr_get_boxes(kocr_ctx* ctx, const float* heat, int N, int h, int w, float detection_threshold,
                   float text_threshold, float link_threshold, int size_threshold, float* boxes,
                   int32_t* counts, int cap, int on_device) {
  if (!ctx) return KOCR_EINVAL;
  if (N < 0 || (N > 0 && (!heat || !boxes || !counts))) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_get_boxes: null buffer");
  if (N == 0) return KOCR_OK;
  KOCR_HIP(ctx, hipSetDevice(ctx->device));
  const size_t hb = (size_t)N * h * w * 2 * sizeof(float), bb = (size_t)N * cap * 8 * sizeof(float);
  const float* d_heat = heat;
  float* d_boxes = boxes;
  if (!on_device) {
    KOCR_TRY(arena_reserve(ctx, ctx->io, hb + bb + 1024));
    ctx->io.off = 0;
    float* dh = (float*)arena_alloc(ctx->io, hb);
    d_boxes = (float*)arena_alloc(ctx->io, bb);
    KOCR_HIP(ctx, hipMemcpyAsync(dh, heat, hb, hipMemcpyHostToDevice, ctx->stream));
    d_heat = dh;
  }
  int n_empty = 0;
  const int rc = postproc_get_boxes(ctx, d_heat, N, h, w, detection_threshold, text_threshold, link_threshold,
                                    size_threshold, d_boxes, cap, counts, &n_empty);
  if (rc != KOCR_OK) return rc;
  if (!on_device) {
    KOCR_HIP(ctx, hipMemcpyAsync(boxes, d_boxes, bb, hipMemcpyDeviceToHost, ctx->stream));
    KOCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  if (n_empty > 0)
    KOCR_FAIL(ctx, KOCR_EEMPTYCONTOUR,
              "kocr_get_boxes: a component has no pixels left after removing text&link overlap "
              "(the reference raises IndexError at detection.py:272)");
  return KOCR_OK;
}

int kocr_warp_crops(kocr_ctx* ctx, const uint8_t* img_rgb, int N, int H, int W, const float* boxes,
                    const int32_t* counts, int target_h, int target_w, float* crops, int on_device) {
  if (!ctx) return KOCR_EINVAL;
  if (N < 0 || target_h <= 0 || target_w <= 0 || (N > 0 && (!img_rgb || !counts)))
    KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_warp_crops: bad argument");
  long M = 0;
  for (int i = 0; i < N; ++i) {
    if (counts[i] < 0) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_warp_crops: negative count");
    M += counts[i];
  }
  if (M == 0) return KOCR_OK;
  if (!boxes || !crops) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_warp_crops: null buffer");
  KOCR_HIP(ctx, hipSetDevice(ctx->device));
  std::vector<WarpParam> prm((size_t)M);
  long m = 0;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < counts[i]; ++j, ++m) {
      const int rc = warp_prepare(boxes + m * 8, target_h, target_w, &prm[m], nullptr);
      if (rc == 1)
        KOCR_FAIL(ctx, KOCR_EZERODIV, "kocr_warp_crops: box with zero width or height (ZeroDivisionError at tools.py:95)");
      if (rc != 0) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_warp_crops: singular perspective transform");
      prm[m].img = i;
    }
  const size_t ib = (size_t)N * H * W * 3, cb = (size_t)M * target_h * target_w * sizeof(float);
  const size_t pb = (size_t)M * sizeof(WarpParam);
  KOCR_TRY(arena_reserve(ctx, ctx->io, pb + (on_device ? 0 : ib + cb) + 2048));
  ctx->io.off = 0;
  WarpParam* d_prm = (WarpParam*)arena_alloc(ctx->io, pb);
  KOCR_HIP(ctx, hipMemcpyAsync(d_prm, prm.data(), pb, hipMemcpyHostToDevice, ctx->stream));
  const uint8_t* d_img = img_rgb;
  float* d_crops = crops;
  if (!on_device) {
    uint8_t* di = (uint8_t*)arena_alloc(ctx->io, ib);
    d_crops = (float*)arena_alloc(ctx->io, cb);
    KOCR_HIP(ctx, hipMemcpyAsync(di, img_rgb, ib, hipMemcpyHostToDevice, ctx->stream));
    d_img = di;
  }
  KOCR_TRY(launch_warp(ctx, d_img, H, W, d_prm, (int)M, target_h, target_w, d_crops));
  if (!on_device) KOCR_HIP(ctx, hipMemcpyAsync(crops, d_crops, cb, hipMemcpyDeviceToHost, ctx->stream));
  // prm is host memory about to go out of scope: the H2D copy must have completed
  KOCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return KOCR_OK;
}

// float images (round 5): the same two image operations in float, see imgproc.hip / warp.hip.  Host pointers.
int kocr_resize_pad_f32(kocr_ctx* ctx, const float* src, int n, int sh, int sw, int channels, int dh, int dw, int Hmax, int Wmax,
                        float cval, float* dst) {
  if (!ctx) return KOCR_EINVAL;
  if (n < 0 || channels <= 0 || (n > 0 && (!src || !dst))) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_resize_pad_f32: bad argument");
  if (n == 0) return KOCR_OK;
  KOCR_HIP(ctx, hipSetDevice(ctx->device));
  const size_t sb = (size_t)n * sh * sw * channels * sizeof(float), db = (size_t)n * Hmax * Wmax * channels * sizeof(float);
  const size_t tb = (size_t)(3 * (Wmax + Hmax) + 64) * sizeof(int);
  KOCR_TRY(arena_reserve(ctx, ctx->io, tb + sb + db + 4096));
  ctx->io.off = 0;
  float* ds = (float*)arena_alloc(ctx->io, sb);
  float* dd = (float*)arena_alloc(ctx->io, db);
  KOCR_HIP(ctx, hipMemcpyAsync(ds, src, sb, hipMemcpyHostToDevice, ctx->stream));
  KOCR_TRY(launch_resize_pad_f32(ctx, ds, n, sh, sw, channels, dd, dh, dw, Hmax, Wmax, cval, ctx->io));
  KOCR_HIP(ctx, hipMemcpyAsync(dst, dd, db, hipMemcpyDeviceToHost, ctx->stream));
  KOCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return KOCR_OK;
}

int kocr_warp_crops_f32(kocr_ctx* ctx, const float* img, int N, int H, int W, int channels, const float* boxes, const int32_t* counts,
                        int target_h, int target_w, float* crops) {
  if (!ctx) return KOCR_EINVAL;
  if (N < 0 || target_h <= 0 || target_w <= 0 || (channels != 1 && channels != 3) || (N > 0 && (!img || !counts)))
    KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_warp_crops_f32: bad argument");
  long M = 0;
  for (int i = 0; i < N; ++i) {
    if (counts[i] < 0) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_warp_crops_f32: negative count");
    M += counts[i];
  }
  if (M == 0) return KOCR_OK;
  if (!boxes || !crops) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_warp_crops_f32: null buffer");
  KOCR_HIP(ctx, hipSetDevice(ctx->device));
  std::vector<WarpParam> prm((size_t)M);
  long m = 0;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < counts[i]; ++j, ++m) {
      const int rc = warp_prepare(boxes + m * 8, target_h, target_w, &prm[m], nullptr);
      if (rc == 1) KOCR_FAIL(ctx, KOCR_EZERODIV, "kocr_warp_crops_f32: box with zero width or height (ZeroDivisionError at tools.py:95)");
      if (rc != 0) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_warp_crops_f32: singular perspective transform");
      prm[m].img = i;
    }
  const size_t ib = (size_t)N * H * W * channels * sizeof(float), cb = (size_t)M * target_h * target_w * sizeof(float);
  const size_t pb = (size_t)M * sizeof(WarpParam);
  KOCR_TRY(arena_reserve(ctx, ctx->io, pb + ib + cb + 4096));
  ctx->io.off = 0;
  WarpParam* d_prm = (WarpParam*)arena_alloc(ctx->io, pb);
  float* di = (float*)arena_alloc(ctx->io, ib);
  float* d_crops = (float*)arena_alloc(ctx->io, cb);
  KOCR_HIP(ctx, hipMemcpyAsync(d_prm, prm.data(), pb, hipMemcpyHostToDevice, ctx->stream));
  KOCR_HIP(ctx, hipMemcpyAsync(di, img, ib, hipMemcpyHostToDevice, ctx->stream));
  KOCR_TRY(launch_warp_f32(ctx, di, H, W, channels, d_prm, (int)M, target_h, target_w, d_crops));
  KOCR_HIP(ctx, hipMemcpyAsync(crops, d_crops, cb, hipMemcpyDeviceToHost, ctx->stream));
  KOCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return KOCR_OK;
}

int kocr_warp_quads(kocr_ctx* ctx, const uint8_t* img_rgb, int N, int H, int W, int M, const float* src_quads,
                    const float* dst_quads, const int32_t* image_index, const int32_t* crop_w, const int32_t* crop_h,
                    int target_h, int target_w, float* crops, double* transforms) {
  if (!ctx) return KOCR_EINVAL;
  if (N <= 0 || M < 0 || target_h <= 0 || target_w <= 0 || !img_rgb)
    KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_warp_quads: bad argument");
  if (M == 0) return KOCR_OK;
  if (!src_quads || !dst_quads || !image_index || !crop_w || !crop_h || !crops)
    KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_warp_quads: null buffer");
  for (int m = 0; m < M; ++m)
    if (image_index[m] < 0 || image_index[m] >= N || crop_w[m] < 0 || crop_h[m] < 0 || crop_w[m] > target_w ||
        crop_h[m] > target_h)
      KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_warp_quads: image index or crop size out of range");
  KOCR_HIP(ctx, hipSetDevice(ctx->device));
  const size_t ib = (size_t)N * H * W * 3, cb = (size_t)M * target_h * target_w * sizeof(float);
  const size_t qb = (size_t)M * 8 * sizeof(float), nb = (size_t)M * sizeof(int), tb = (size_t)M * 9 * sizeof(double);
  KOCR_TRY(arena_reserve(ctx, ctx->io, ib + cb + 2 * qb + 3 * nb + tb + (size_t)M * sizeof(WarpParam) + 8192));
  ctx->io.off = 0;
  uint8_t* d_img = (uint8_t*)arena_alloc(ctx->io, ib);
  float* d_crops = (float*)arena_alloc(ctx->io, cb);
  float* d_src = (float*)arena_alloc(ctx->io, qb);
  float* d_dst = (float*)arena_alloc(ctx->io, qb);
  int* d_idx = (int*)arena_alloc(ctx->io, nb);
  int* d_cw = (int*)arena_alloc(ctx->io, nb);
  int* d_ch = (int*)arena_alloc(ctx->io, nb);
  double* d_tf = (double*)arena_alloc(ctx->io, tb);
  WarpParam* d_prm = (WarpParam*)arena_alloc(ctx->io, (size_t)M * sizeof(WarpParam));
  int* d_status = (int*)arena_alloc(ctx->io, 256);
  if (!d_status) KOCR_FAIL(ctx, KOCR_ENOMEM, "kocr_warp_quads: arena exhausted");
  hipStream_t s = ctx->stream;
  KOCR_HIP(ctx, hipMemcpyAsync(d_img, img_rgb, ib, hipMemcpyHostToDevice, s));
  KOCR_HIP(ctx, hipMemcpyAsync(d_src, src_quads, qb, hipMemcpyHostToDevice, s));
  KOCR_HIP(ctx, hipMemcpyAsync(d_dst, dst_quads, qb, hipMemcpyHostToDevice, s));
  KOCR_HIP(ctx, hipMemcpyAsync(d_idx, image_index, nb, hipMemcpyHostToDevice, s));
  KOCR_HIP(ctx, hipMemcpyAsync(d_cw, crop_w, nb, hipMemcpyHostToDevice, s));
  KOCR_HIP(ctx, hipMemcpyAsync(d_ch, crop_h, nb, hipMemcpyHostToDevice, s));
  KOCR_HIP(ctx, hipMemsetAsync(d_status, 0, sizeof(int), s));
  KOCR_TRY(launch_warp_quads(ctx, d_src, d_dst, d_idx, d_cw, d_ch, M, d_prm, d_tf, d_status));
  KOCR_TRY(launch_warp(ctx, d_img, H, W, d_prm, M, target_h, target_w, d_crops));
  int status = 0;
  KOCR_HIP(ctx, hipMemcpyAsync(crops, d_crops, cb, hipMemcpyDeviceToHost, s));
  if (transforms) KOCR_HIP(ctx, hipMemcpyAsync(transforms, d_tf, tb, hipMemcpyDeviceToHost, s));
  KOCR_HIP(ctx, hipMemcpyAsync(&status, d_status, sizeof(int), hipMemcpyDeviceToHost, s));
  KOCR_HIP(ctx, hipStreamSynchronize(s));
  if (status != 0) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_warp_quads: singular perspective transform");
  return KOCR_OK;
}

int kocr_conv2d_nhwc(kocr_ctx* ctx, const float* in, int N, int H, int W, int Cin, const float* w_hwio,
                     int KH, int KW, int dilation, int Cout, const float* pre_a, const float* pre_b,
                     int relu, const float* post_a, const float* post_b, float* out) {
  if (!ctx || !in || !w_hwio || !out) return KOCR_EINVAL;
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || dilation <= 0 ||
      !(KH & 1) || !(KW & 1))
    KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_conv2d_nhwc: bad shape (odd kernels, positive sizes)");
  KOCR_HIP(ctx, hipSetDevice(ctx->device));
  KOCR_TRY(ctx->amax_begin());  // allocates the slot pool BEFORE the mark below: it outlives this call
  const size_t first_owned = ctx->owned.size();
  ConvLayer L;
  L.name = "kocr_conv2d_nhwc";
  int rc = prepare_conv(ctx, L, w_hwio, /*oihw=*/false, Cin, Cout, KH, KW, dilation, pre_a, pre_b, relu,
                        post_a, post_b);
  const size_t nin = (size_t)N * H * W * Cin, nout = (size_t)N * H * W * Cout;
  if (rc == KOCR_OK) rc = ctx->ws_reserve((nin + nout) * sizeof(float) + 4096);
  if (rc == KOCR_OK) {
    ctx->ws_reset();
    Tensor ti, to;
    ti.N = to.N = N;
    ti.H = to.H = H;
    ti.W = to.W = W;
    ti.C = ti.cs = Cin;
    to.C = to.cs = Cout;
    ti.p = (float*)ctx->ws_alloc(nin * sizeof(float));
    to.p = (float*)ctx->ws_alloc(nout * sizeof(float));
    auto run = [&]() -> int {
      KOCR_HIP(ctx, hipMemcpyAsync(ti.p, in, nin * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
      KOCR_TRY(launch_conv(ctx, L, ti, nullptr, nullptr, to));
      KOCR_HIP(ctx, hipMemcpyAsync(out, to.p, nout * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
      KOCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
      return KOCR_OK;
    };
    rc = run();
  }
  // release the temporary layer's device buffers
  hipStreamSynchronize(ctx->stream);
  while (ctx->owned.size() > first_owned) {
    hipFree(ctx->owned.back());
    ctx->owned.pop_back();
  }
  return rc;
}

int kocr_conv2d_cells(kocr_ctx* ctx, const float* in, int N, int H, int W, int Cin, const float* w_hwio, int Cout,
                      const float* pre_a, const float* pre_b, int relu, const float* post_a, const float* post_b,
                      int cellW, int cellWv, int pool, float* out, float* pool_out, float* amax_out) {
  if (!ctx || !in || !w_hwio || (!out && !pool) || (pool && !pool_out)) return KOCR_EINVAL;
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || cellW <= 0 || cellWv <= 0 || W % cellW)
    KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_conv2d_cells: bad shape");
  // the constraints include/kocr.h documents, checked here with their own messages: the columns behind cellWv are
  // the 'same' padding between neighbouring crops -- without at least one of them a crop would read its neighbour's edge
  if (cellWv >= cellW)
    KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_conv2d_cells: cellWv must be smaller than cellW (at least one zero gutter column per cell)");
  if (cellW % 4 != 0) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_conv2d_cells: cellW must be a multiple of 4 (F(4,3) quads)");
  if (pool && (cellWv % 2 != 0 || cellW % 8 != 0 || H % 2 != 0))
    KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_conv2d_cells: pooling needs an even cellWv, cellW % 8 == 0 and an even H");
  if (Cin % 32 != 0 || Cout <= 64)
    KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_conv2d_cells: the cell-grid kernel takes Cin % 32 == 0 and Cout > 64");
  if (!((H % 4 == 0 && W % 64 == 0) || (H % 8 == 0 && W % 32 == 0)))
    KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_conv2d_cells: the grid must tile as 4 x 64 (H % 4 == 0, W % 64 == 0) or 8 x 32");
  KOCR_HIP(ctx, hipSetDevice(ctx->device));
  KOCR_TRY(ctx->amax_begin());
  const size_t first_owned = ctx->owned.size();
  ConvLayer L;
  L.name = "kocr_conv2d_cells";
  int rc = prepare_conv(ctx, L, w_hwio, /*oihw=*/false, Cin, Cout, 3, 3, 1, pre_a, pre_b, relu, post_a, post_b);
  const int cn = W / cellW;
  const size_t nin = (size_t)N * H * W * Cin, nout = (size_t)N * H * W * Cout, npool = nout / 4;
  if (rc == KOCR_OK) rc = ctx->ws_reserve((nin + (out ? nout : 0) + (pool ? npool : 0)) * sizeof(float) + 8192);
  if (rc == KOCR_OK) {
    ctx->ws_reset();
    Tensor ti, to, tp;
    ti.N = to.N = tp.N = N;
    ti.H = to.H = H;
    ti.W = to.W = W;
    tp.H = H / 2;
    tp.W = W / 2;
    ti.C = ti.cs = Cin;
    to.C = to.cs = tp.C = tp.cs = Cout;
    ti.cellW = to.cellW = cellW;
    ti.cellWv = to.cellWv = cellWv;
    tp.cellW = cellW / 2;
    tp.cellWv = cellWv / 2;
    ti.p = (float*)ctx->ws_alloc(nin * sizeof(float));
    to.p = out ? (float*)ctx->ws_alloc(nout * sizeof(float)) : nullptr;
    tp.p = pool ? (float*)ctx->ws_alloc(npool * sizeof(float)) : nullptr;
    ti.amax = ctx->amax_slots(N * cn);
    to.amax = ctx->amax_slots(N * cn);
    tp.amax = ctx->amax_slots(N * cn);
    // the input's per-cell max |x| (its producer's job in the recogniser): computed here on the host
    std::vector<float> am((size_t)N * cn, 0.f);
    for (int n = 0; n < N; ++n)
      for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
          const float* px = in + (((size_t)n * H + y) * W + x) * Cin;
          float m = 0.f;
          for (int c = 0; c < Cin; ++c) m = std::max(m, std::fabs(px[c]));
          float& a = am[(size_t)n * cn + x / cellW];
          a = std::max(a, m);
        }
    auto run = [&]() -> int {
      if (!ti.amax || !to.amax || !tp.amax) KOCR_FAIL(ctx, KOCR_ECAPACITY, "kocr_conv2d_cells: out of max-|x| slots");
      KOCR_HIP(ctx, hipMemcpyAsync(ti.p, in, nin * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
      KOCR_HIP(ctx, hipMemcpyAsync(ti.amax, am.data(), am.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
      KOCR_TRY(launch_conv_pool(ctx, L, ti, nullptr, nullptr, to, pool ? &tp : nullptr, /*need_full=*/out != nullptr));
      if (out) KOCR_HIP(ctx, hipMemcpyAsync(out, to.p, nout * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
      if (pool) KOCR_HIP(ctx, hipMemcpyAsync(pool_out, tp.p, npool * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
      if (amax_out)
        KOCR_HIP(ctx, hipMemcpyAsync(amax_out, pool && !out ? tp.amax : to.amax, am.size() * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
      KOCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
      return KOCR_OK;
    };
    rc = run();
  }
  hipStreamSynchronize(ctx->stream);
  while (ctx->owned.size() > first_owned) {
    hipFree(ctx->owned.back());
    ctx->owned.pop_back();
  }
  return rc;
}

int kocr_set_split_mode(kocr_ctx* ctx, int mode) {
  if (!ctx) return KOCR_EINVAL;
  if (mode != KOCR_SPLIT_BF16X3 && mode != KOCR_SPLIT_F16X2 && mode != KOCR_SPLIT_F16X1) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_set_split_mode: unknown mode");
  ctx->split_mode = mode;
  return KOCR_OK;
}

int kocr_get_split_mode(const kocr_ctx* ctx) { return ctx ? ctx->split_mode : KOCR_EINVAL; }

int kocr_set_schedule(kocr_ctx* ctx, int fold_linear_chain, int fold_upsample) {
  if (!ctx) return KOCR_EINVAL;
  ctx->opt_linfold = fold_linear_chain != 0;
  ctx->opt_upfold = fold_upsample != 0;
  return KOCR_OK;
}

int kocr_profile_enable(kocr_ctx* ctx, int on) {
  if (!ctx) return KOCR_EINVAL;
  KOCR_TRY(ctx->prof_flush());
  ctx->prof_on = on != 0;
  return KOCR_OK;
}

int kocr_profile_reset(kocr_ctx* ctx) {
  if (!ctx) return KOCR_EINVAL;
  KOCR_TRY(ctx->prof_flush());
  ctx->prof.clear();
  return KOCR_OK;
}

int kocr_profile_report(kocr_ctx* ctx, int cap, char* names, int64_t* launches, double* total_ms,
                        double* flops, double* bytes) {
  if (!ctx) return KOCR_EINVAL;
  KOCR_TRY(ctx->prof_flush());
  int i = 0;
  for (auto& kv : ctx->prof) {
    if (i < cap && names && launches && total_ms && flops && bytes) {
      snprintf(names + (size_t)i * 64, 64, "%s", kv.first.c_str());
      launches[i] = kv.second.launches;
      total_ms[i] = kv.second.ms;
      flops[i] = kv.second.flops;
      bytes[i] = kv.second.bytes;
    }
    ++i;
  }
  return i;
}

int kocr_range_stats_enable(kocr_ctx* ctx, int on) {
  if (!ctx) return KOCR_EINVAL;
  if (on && !ctx->d_range) {  // allocated here, not at first use: kocr_conv2d_nhwc frees what a call allocates
    KOCR_HIP(ctx, hipSetDevice(ctx->device));
    void* d = nullptr;
    KOCR_TRY(ctx->dev_alloc(&d, 64));
    ctx->d_range = d;
  }
  ctx->range_on = on != 0;
  if (on) ctx->range.clear();
  return KOCR_OK;
}

int kocr_range_stats_report(kocr_ctx* ctx, int cap, char* names, double* values) {
  if (!ctx) return KOCR_EINVAL;
  int i = 0;
  for (auto& kv : ctx->range) {
    if (i < cap && names && values) {
      snprintf(names + (size_t)i * 64, 64, "%s", kv.first.c_str());
      const RangeRow& r = kv.second;
      const double v[7] = {r.launches, r.elements, r.nonzero, r.below_m4, r.below_m14, r.sum_abs, r.sum_abs_below_m4};
      for (int k = 0; k < 7; ++k) values[(size_t)i * 7 + k] = v[k];
    }
    ++i;
  }
  return i;
}

}  // extern "C"
