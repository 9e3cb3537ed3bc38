// pipeline.cpp — the fused device path of Pipeline.recognize (pipeline.py:28-75):
//   resize_image + pad (pipeline.py:44-57)  ->  Detector.detect (:62; detection.py:745-785)
//   ->  Recognizer.recognize_from_boxes (:63-65; recognition.py:491-537).
// Everything stays resident in HBM between the stages; the host sees only the per-image box
// counts (a few bytes, needed to size the crop batch), the boxes and the decoded label rows.
#include "common.h"
#include <algorithm>

extern "C" int kocr_resize_pad(kocr_ctx* ctx, const uint8_t* src, int n, int sh, int sw, int dh, int dw, int Hmax,
                               int Wmax, int cval, uint8_t* dst, int on_device) {
  if (!ctx) return KOCR_EINVAL;
  if (n < 0 || (n > 0 && (!src || !dst))) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_resize_pad: null buffer");
  if (n == 0) return KOCR_OK;
  ctx->last_pl.valid = false;
  KOCR_HIP(ctx, hipSetDevice(ctx->device));
  const size_t sb = (size_t)n * sh * sw * 3, db = (size_t)n * Hmax * Wmax * 3;
  const size_t tb = (size_t)(4 * (Wmax + Hmax) + 64) * sizeof(int);
  KOCR_TRY(arena_reserve(ctx, ctx->io, tb + (on_device ? 0 : sb + db) + 4096));
  ctx->io.off = 0;
  const uint8_t* d_src = src;
  uint8_t* d_dst = dst;
  if (!on_device) {
    uint8_t* ds = (uint8_t*)arena_alloc(ctx->io, sb);
    d_dst = (uint8_t*)arena_alloc(ctx->io, db);
    KOCR_HIP(ctx, hipMemcpyAsync(ds, src, sb, hipMemcpyHostToDevice, ctx->stream));
    d_src = ds;
  }
  KOCR_TRY(launch_resize_pad(ctx, d_src, n, sh, sw, d_dst, dh, dw, Hmax, Wmax, cval, ctx->io));
  if (!on_device) {
    KOCR_HIP(ctx, hipMemcpyAsync(dst, d_dst, db, hipMemcpyDeviceToHost, ctx->stream));
    KOCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return KOCR_OK;
}

// Detector.detect's device half in one call: CRAFT forward + getBoxes, heat-maps never leave HBM.
extern "C" int kocr_detect(kocr_ctx* ctx, const void* img, int dtype, int N, int H, int W, float detection_threshold,
                           float text_threshold, float link_threshold, int size_threshold, int micro_batch,
                           float* boxes, int32_t* counts, int cap, int on_device) {
  if (!ctx) return KOCR_EINVAL;
  if (N < 0 || (N > 0 && (!img || !boxes || !counts))) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_detect: null buffer");
  if (dtype != KOCR_U8 && dtype != KOCR_F32) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_detect: bad dtype");
  if (N == 0) return KOCR_OK;
  if (cap <= 0) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_detect: cap must be positive");
  ctx->last_pl.valid = false;
  KOCR_HIP(ctx, hipSetDevice(ctx->device));
  const int h2 = H / 2, w2 = W / 2;
  const size_t esz = dtype == KOCR_U8 ? 1 : 4;
  const size_t in_b = (size_t)N * H * W * 3 * esz;
  const size_t heat_b = (size_t)N * h2 * w2 * 2 * sizeof(float);
  const size_t box_b = (size_t)N * cap * 8 * sizeof(float);
  KOCR_TRY(arena_reserve(ctx, ctx->pl, heat_b + box_b + (on_device ? 0 : in_b) + 4096));
  ctx->pl.off = 0;
  float* d_heat = (float*)arena_alloc(ctx->pl, heat_b);
  float* d_boxes = on_device ? boxes : (float*)arena_alloc(ctx->pl, box_b);
  const char* d_in = (const char*)img;
  if (!on_device) {
    char* di = (char*)arena_alloc(ctx->pl, in_b);
    if (!di) KOCR_FAIL(ctx, KOCR_ENOMEM, "kocr_detect: arena exhausted");
    KOCR_HIP(ctx, hipMemcpyAsync(di, img, in_b, hipMemcpyHostToDevice, ctx->stream));
    d_in = di;
  }
  if (!d_heat || !d_boxes) KOCR_FAIL(ctx, KOCR_ENOMEM, "kocr_detect: arena exhausted");
  int mb = micro_batch > 0 ? micro_batch : 32;
  while (mb > 1 && craft_workspace_bytes(mb, H, W) > ((size_t)96 << 30)) mb = (mb + 1) / 2;
  mb = std::min(mb, N);
  KOCR_TRY(ctx->ws_reserve(craft_workspace_bytes(mb, H, W)));
  for (int s = 0; s < N; s += mb) {
    const int nb = std::min(mb, N - s);
    ctx->ws_reset();
    KOCR_TRY(craft_forward(ctx, d_in + (size_t)s * H * W * 3 * esz, dtype, nb, H, W, d_heat + (size_t)s * h2 * w2 * 2));
  }
  int n_empty = 0;
  KOCR_TRY(postproc_get_boxes(ctx, d_heat, N, h2, w2, detection_threshold, text_threshold, link_threshold,
                              size_threshold, d_boxes, cap, counts, &n_empty));
  if (!on_device) {
    KOCR_HIP(ctx, hipMemcpyAsync(boxes, d_boxes, box_b, hipMemcpyDeviceToHost, ctx->stream));
    KOCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  if (n_empty > 0)
    KOCR_FAIL(ctx, KOCR_EEMPTYCONTOUR, "kocr_detect: empty contour list (IndexError at detection.py:272)");
  return KOCR_OK;
}

// Recognizer.recognize_from_boxes' device half in one call: crops are warped and recognised without
// leaving HBM.  All N images share one size; boxes/counts/labels are HOST buffers.
extern "C" int kocr_recognize_boxes(kocr_ctx* ctx, const uint8_t* img_rgb, int N, int H, int W, const float* boxes,
                                    const int32_t* counts, int32_t* labels, int on_device) {
  if (!ctx) return KOCR_EINVAL;
  if (N < 0 || (N > 0 && (!img_rgb || !counts))) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_recognize_boxes: null buffer");
  const int C = crnn_classes(ctx);
  if (C == 0) KOCR_FAIL(ctx, KOCR_ENOWEIGHTS, "kocr_recognize_boxes: call kocr_load_crnn first");
  long M = 0;
  for (int i = 0; i < N; ++i) {
    if (counts[i] < 0) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_recognize_boxes: negative count");
    M += counts[i];
  }
  if (M == 0) return KOCR_OK;
  if (!boxes || !labels) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_recognize_boxes: null buffer");
  ctx->last_pl.valid = false;
  KOCR_HIP(ctx, hipSetDevice(ctx->device));
  std::vector<WarpParam> prm((size_t)M);
  long m = 0;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < counts[i]; ++j, ++m) {
      const int rc = warp_prepare(boxes + m * 8, 31, 200, &prm[m], nullptr);
      if (rc == 1) KOCR_FAIL(ctx, KOCR_EZERODIV, "kocr_recognize_boxes: box with zero width or height (tools.py:95)");
      if (rc != 0) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_recognize_boxes: singular perspective transform");
      prm[m].img = i;
    }
  const size_t ib = (size_t)N * H * W * 3, crop_b = (size_t)M * 31 * 200 * sizeof(float);
  const size_t lab_b = (size_t)M * crnn_label_width(ctx) * sizeof(int32_t), pb = (size_t)M * sizeof(WarpParam);
  KOCR_TRY(arena_reserve(ctx, ctx->io, pb + crop_b + lab_b + (on_device ? 0 : ib) + 4096));
  ctx->io.off = 0;
  WarpParam* d_prm = (WarpParam*)arena_alloc(ctx->io, pb);
  float* d_crops = (float*)arena_alloc(ctx->io, crop_b);
  int32_t* d_lab = (int32_t*)arena_alloc(ctx->io, lab_b);
  const uint8_t* d_img = img_rgb;
  if (!on_device) {
    uint8_t* di = (uint8_t*)arena_alloc(ctx->io, ib);
    if (!di) KOCR_FAIL(ctx, KOCR_ENOMEM, "kocr_recognize_boxes: arena exhausted");
    KOCR_HIP(ctx, hipMemcpyAsync(di, img_rgb, ib, hipMemcpyHostToDevice, ctx->stream));
    d_img = di;
  }
  KOCR_HIP(ctx, hipMemcpyAsync(d_prm, prm.data(), pb, hipMemcpyHostToDevice, ctx->stream));
  KOCR_TRY(launch_warp(ctx, d_img, H, W, d_prm, (int)M, 31, 200, d_crops));
  const int cmb = (int)std::min<long>(M, 1024);
  KOCR_TRY(ctx->ws_reserve(crnn_workspace_bytes(cmb, C)));
  for (long s = 0; s < M; s += cmb) {
    const int nb = (int)std::min<long>(cmb, M - s);
    ctx->ws_reset();
    KOCR_TRY(crnn_forward(ctx, d_crops + (size_t)s * 31 * 200, nb, d_lab + (size_t)s * crnn_label_width(ctx), nullptr));
  }
  KOCR_HIP(ctx, hipMemcpyAsync(labels, d_lab, lab_b, hipMemcpyDeviceToHost, ctx->stream));
  KOCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return KOCR_OK;
}

extern "C" int kocr_pipeline(kocr_ctx* ctx, int N, const uint8_t* const* imgs, const int32_t* hs, const int32_t* ws,
                             const int32_t* dhs, const int32_t* dws, int Hmax, int Wmax, float detection_threshold,
                             float text_threshold, float link_threshold, int size_threshold, int micro_batch,
                             float* boxes, int32_t* counts, int cap, int32_t* labels, int max_crops,
                             int32_t* n_crops, int on_device) {
  if (!ctx) return KOCR_EINVAL;
  if (n_crops) *n_crops = 0;
  if (N < 0 || (N > 0 && (!imgs || !hs || !ws || !dhs || !dws || !boxes || !counts)))
    KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_pipeline: null buffer");
  ctx->last_pl.valid = false;
  if (N == 0) return KOCR_OK;
  if (!ctx->craft) KOCR_FAIL(ctx, KOCR_ENOWEIGHTS, "kocr_pipeline: call kocr_load_craft first");
  if (crnn_classes(ctx) == 0) KOCR_FAIL(ctx, KOCR_ENOWEIGHTS, "kocr_pipeline: call kocr_load_crnn first");
  if (Hmax < 16 || Wmax < 16 || cap <= 0) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_pipeline: bad sizes");
  KOCR_HIP(ctx, hipSetDevice(ctx->device));
  const int h2 = Hmax / 2, w2 = Wmax / 2;
  const size_t bat_b = (size_t)N * Hmax * Wmax * 3;
  const size_t heat_b = (size_t)N * h2 * w2 * 2 * sizeof(float);
  const size_t box_b = (size_t)N * cap * 8 * sizeof(float);
  // ---- persistent buffers of this call ----
  KOCR_TRY(arena_reserve(ctx, ctx->pl, bat_b + heat_b + box_b + 4096));
  ctx->pl.off = 0;
  uint8_t* d_bat = (uint8_t*)arena_alloc(ctx->pl, bat_b);
  float* d_heat = (float*)arena_alloc(ctx->pl, heat_b);
  float* d_boxes = (float*)arena_alloc(ctx->pl, box_b);
  if (!d_bat || !d_heat || !d_boxes) KOCR_FAIL(ctx, KOCR_ENOMEM, "kocr_pipeline: arena exhausted");
  // ---- resize + pad, runs of identically-shaped contiguous images in one launch ----
  size_t max_src = 0;
  for (int i = 0; i < N; ++i) max_src = std::max(max_src, (size_t)hs[i] * ws[i] * 3);
  int i = 0;
  while (i < N) {
    int j = i + 1;
    while (j < N && hs[j] == hs[i] && ws[j] == ws[i] && dhs[j] == dhs[i] && dws[j] == dws[i] &&
           imgs[j] == imgs[j - 1] + (size_t)hs[i] * ws[i] * 3)
      ++j;
    const int run = j - i;
    const size_t sb = (size_t)run * hs[i] * ws[i] * 3;
    const size_t tb = (size_t)(4 * (Wmax + Hmax) + 64) * sizeof(int);
    KOCR_TRY(arena_reserve(ctx, ctx->io, tb + (on_device ? 0 : sb) + 4096));
    ctx->io.off = 0;
    const uint8_t* d_src = imgs[i];
    if (!on_device) {
      uint8_t* ds = (uint8_t*)arena_alloc(ctx->io, sb);
      KOCR_HIP(ctx, hipMemcpyAsync(ds, imgs[i], sb, hipMemcpyHostToDevice, ctx->stream));
      d_src = ds;
    }
    KOCR_TRY(launch_resize_pad(ctx, d_src, run, hs[i], ws[i], d_bat + (size_t)i * Hmax * Wmax * 3, dhs[i], dws[i],
                               Hmax, Wmax, 255, ctx->io));
    i = j;
  }
  // ---- detector forward (micro-batched) ----
  int mb = micro_batch > 0 ? micro_batch : 32;
  while (mb > 1 && craft_workspace_bytes(mb, Hmax, Wmax) > ((size_t)96 << 30)) mb = (mb + 1) / 2;
  mb = std::min(mb, N);
  KOCR_TRY(ctx->ws_reserve(craft_workspace_bytes(mb, Hmax, Wmax)));
  for (int s = 0; s < N; s += mb) {
    const int nb = std::min(mb, N - s);
    ctx->ws_reset();
    KOCR_TRY(craft_forward(ctx, d_bat + (size_t)s * Hmax * Wmax * 3, KOCR_U8, nb, Hmax, Wmax,
                           d_heat + (size_t)s * h2 * w2 * 2));
  }
  // ---- boxes: the host learns only the per-image counts here (needed to size the crop batch); the boxes themselves
  // go to the caller asynchronously while the device already derives the crop homographies from its own copy ----
  // CAPACITY WITHOUT RECOMPUTATION (round 6; the reference has no cap, detection.py:230-286): `cap` and `max_crops` size the
  // CALLER's buffers only.  A page with more boxes than cap gets a larger device box buffer (its own arena: growing the pl
  // arena would free the heat-maps) and ONLY the post-processing runs again on the resident heat-maps -- it stops at its
  // counting pass when the capacity does not suffice, so the first attempt cost the threshold + labelling kernels.  Crops and
  // recogniser then run on all M crops whatever max_crops is; when the caller's buffers are too small the results stay in
  // HBM, KOCR_ECAPACITY reports the true counts, and kocr_pipeline_results copies them into larger buffers: one detector
  // forward, one recogniser pass, always.
  PPDeviceOut dv;
  int d_cap = cap;
  int rc_pp = postproc_get_boxes(ctx, d_heat, N, h2, w2, detection_threshold, text_threshold, link_threshold,
                                 size_threshold, d_boxes, d_cap, counts, nullptr, &dv);
  if (rc_pp == KOCR_ECAPACITY) {
    int need = 0;
    for (int k = 0; k < N; ++k) need = std::max(need, (int)counts[k]);
    if (need <= d_cap) return rc_pp;  // the other capacity error (dilation canvases beyond 2^31 pixels): not a matter of cap
    d_cap = need;
    KOCR_TRY(arena_reserve(ctx, ctx->bx, (size_t)N * d_cap * 8 * sizeof(float) + 256));
    ctx->bx.off = 0;
    d_boxes = (float*)arena_alloc(ctx->bx, (size_t)N * d_cap * 8 * sizeof(float));
    if (!d_boxes) KOCR_FAIL(ctx, KOCR_ENOMEM, "kocr_pipeline: box arena exhausted");
    rc_pp = postproc_get_boxes(ctx, d_heat, N, h2, w2, detection_threshold, text_threshold, link_threshold, size_threshold,
                               d_boxes, d_cap, counts, nullptr, &dv);
  }
  KOCR_TRY(rc_pp);
  long M = 0;
  for (int k = 0; k < N; ++k) M += counts[k];
  if (n_crops) *n_crops = (int32_t)M;
  const bool host_fits = d_cap == cap && labels && M <= max_crops;
  if (d_cap == cap) KOCR_HIP(ctx, hipMemcpyAsync(boxes, d_boxes, box_b, hipMemcpyDeviceToHost, ctx->stream));
  int host_flags[5] = {0, 0, 0, 0, 0};  // totals[0..3] of the post-processing, warp status
  auto finish = [&]() -> int {
    KOCR_HIP(ctx, hipMemcpyAsync(host_flags, dv.d_totals, 16, hipMemcpyDeviceToHost, ctx->stream));
    KOCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (host_flags[2] > 0)
      KOCR_FAIL(ctx, KOCR_EEMPTYCONTOUR, "kocr_pipeline: empty contour list (IndexError at detection.py:272)");
    if (host_flags[4] == 1) KOCR_FAIL(ctx, KOCR_EZERODIV, "kocr_pipeline: box with zero width or height (tools.py:95)");
    if (host_flags[4] != 0) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_pipeline: singular perspective transform");
    return KOCR_OK;
  };
  if (M == 0) {
    KOCR_TRY(finish());
    ctx->last_pl = {d_boxes, dv.d_counts, nullptr, N, cap, 0, true};
    return KOCR_OK;
  }
  if (!labels && d_cap == cap) {
    KOCR_TRY(finish());  // an empty contour list (the reference's IndexError) takes precedence over the capacity error
    KOCR_FAIL(ctx, KOCR_ECAPACITY, "kocr_pipeline: more crops than max_crops");
  }
  // ---- crops: homographies on the device (warp.hip), no host round trip ----
  const size_t crop_b = (size_t)M * 31 * 200 * sizeof(float), lab_b = (size_t)M * crnn_label_width(ctx) * sizeof(int32_t);
  KOCR_TRY(arena_reserve(ctx, ctx->io, (size_t)M * sizeof(WarpParam) + crop_b + lab_b + 8192));
  ctx->io.off = 0;
  WarpParam* d_prm = (WarpParam*)arena_alloc(ctx->io, (size_t)M * sizeof(WarpParam));
  float* d_crops = (float*)arena_alloc(ctx->io, crop_b);
  int32_t* d_lab = (int32_t*)arena_alloc(ctx->io, lab_b);
  int* d_status = (int*)arena_alloc(ctx->io, 256);
  KOCR_HIP(ctx, hipMemsetAsync(d_status, 0, sizeof(int), ctx->stream));
  KOCR_TRY(launch_warp_prepare(ctx, d_boxes, dv.d_counts, N, d_cap, 31, 200, d_prm, d_status));
  KOCR_TRY(launch_warp(ctx, d_bat, Hmax, Wmax, d_prm, (int)M, 31, 200, d_crops));
  // ---- recogniser ----
  const int C = crnn_classes(ctx);
  const int cmb = (int)std::min<long>(M, 1024);
  KOCR_TRY(ctx->ws_reserve(crnn_workspace_bytes(cmb, C)));
  for (long s = 0; s < M; s += cmb) {
    const int nb = (int)std::min<long>(cmb, M - s);
    ctx->ws_reset();
    KOCR_TRY(crnn_forward(ctx, d_crops + (size_t)s * 31 * 200, nb, d_lab + (size_t)s * crnn_label_width(ctx), nullptr));
  }
  if (host_fits) KOCR_HIP(ctx, hipMemcpyAsync(labels, d_lab, lab_b, hipMemcpyDeviceToHost, ctx->stream));
  KOCR_HIP(ctx, hipMemcpyAsync(&host_flags[4], d_status, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  KOCR_TRY(finish());
  ctx->last_pl = {d_boxes, dv.d_counts, d_lab, N, d_cap, (int)M, true};
  if (!host_fits) {
    ctx->set_err("kocr_pipeline: an image has more boxes than cap, or there are more crops than max_crops; the results are "
                 "resident -- fetch them with kocr_pipeline_results into buffers sized from counts / n_crops");
    return KOCR_ECAPACITY;
  }
  return KOCR_OK;
}

// The resident results of the last kocr_pipeline call copied into the caller's (larger) buffers: see include/kocr.h
extern "C" int kocr_pipeline_results(kocr_ctx* ctx, float* boxes, int cap, int32_t* labels, int max_crops) {
  if (!ctx) return KOCR_EINVAL;
  const auto& r = ctx->last_pl;
  if (!r.valid) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_pipeline_results: no kocr_pipeline result is resident (call it right after kocr_pipeline)");
  if (!boxes || cap < r.cap || (r.M > 0 && (!labels || max_crops < r.M)))
    KOCR_FAIL(ctx, KOCR_ECAPACITY, "kocr_pipeline_results: buffers smaller than the resident results (cap >= " + std::to_string(r.cap) +
                                       ", max_crops >= " + std::to_string(r.M) + ")");
  KOCR_HIP(ctx, hipSetDevice(ctx->device));
  // device rows are r.cap boxes apart, the caller's cap boxes
  KOCR_HIP(ctx, hipMemcpy2DAsync(boxes, (size_t)cap * 32, r.d_boxes, (size_t)r.cap * 32, (size_t)r.cap * 32, (size_t)r.N,
                                 hipMemcpyDeviceToHost, ctx->stream));
  if (r.M > 0)
    KOCR_HIP(ctx, hipMemcpyAsync(labels, r.d_labels, (size_t)r.M * crnn_label_width(ctx) * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  KOCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return KOCR_OK;
}

// The results of the last successful kocr_pipeline call as they lie in HBM (see include/kocr.h)
extern "C" int kocr_pipeline_device_results(kocr_ctx* ctx, const float** d_boxes, const int32_t** d_counts, const int32_t** d_labels,
                                            int32_t* N, int32_t* cap, int32_t* M) {
  if (!ctx) return KOCR_EINVAL;
  if (!ctx->last_pl.valid) KOCR_FAIL(ctx, KOCR_EINVAL, "kocr_pipeline_device_results: no kocr_pipeline result is resident (call it right after kocr_pipeline)");
  if (d_boxes) *d_boxes = ctx->last_pl.d_boxes;
  if (d_counts) *d_counts = ctx->last_pl.d_counts;
  if (d_labels) *d_labels = ctx->last_pl.d_labels;
  if (N) *N = ctx->last_pl.N;
  if (cap) *cap = ctx->last_pl.cap;
  if (M) *M = ctx->last_pl.M;
  return KOCR_OK;
}
