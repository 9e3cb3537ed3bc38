/* kocr.h — C-ABI of libkocr.so: the MI355X (gfx950) hot path of
 * keras_ocr.pipeline.Pipeline.recognize().
 *
 * The reference (faustomorales/keras-ocr) is pure Python and has NO FFI; the seams this
 * library replaces are the two Keras `predict` calls and the OpenCV/shapely host loops
 * between them.  Every entry point cites the reference interface it replaces
 * (file:line relative to the reference checkout).
 *
 * Conventions
 *   - every call returns 0 on success, a negative KOCR_E* code on failure;
 *     kocr_last_error() gives the message of the last failure on that ctx.  After a non-zero
 *     return the contents of every OUTPUT buffer of that call are undefined (partial results may
 *     have been written); only documented exceptions hold (KOCR_ECAPACITY: the counts).  When
 *     several failures apply, a data error the reference would raise (KOCR_EEMPTYCONTOUR,
 *     KOCR_EZERODIV) takes precedence over KOCR_ECAPACITY.
 *   - the caller owns every buffer passed in or out; the library owns weights and
 *     workspace.  `on_device` != 0 means the pointers are HIP device pointers on the
 *     ctx's device (e.g. torch.Tensor.data_ptr()); 0 means host pointers (numpy).
 *   - one ctx per device; calls on a ctx are serialised on its HIP stream and are not
 *     re-entrant.  A ctx (its stream, workspace arenas and profiler) belongs to ONE host thread
 *     at a time: the library takes no locks, exactly as the reference documents nothing as
 *     thread-safe (single caller thread, synchronous).  Use one ctx per thread, or serialise
 *     the calls yourself; separate contexts (also on one device) are independent.  Device-pointer calls are asynchronous on that stream unless they
 *     return host-side counts (documented per call).
 *   - tensors are dense, row-major, channels-last (NHWC), exactly the layouts the
 *     reference hands to / receives from Keras.
 */
#ifndef KOCR_H
#define KOCR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kocr_ctx kocr_ctx;

enum {
  KOCR_OK = 0,
  KOCR_EINVAL = -1,   /* bad argument (the reference would raise AssertionError/ValueError) */
  KOCR_EHIP = -2,     /* HIP runtime error */
  KOCR_ENOWEIGHTS = -3, /* forward called before kocr_load_* */
  KOCR_ECAPACITY = -4,  /* caller-provided output capacity too small */
  KOCR_ENOMEM = -5,
  KOCR_EEMPTYCONTOUR = -6, /* reference: IndexError at detection.py:272 */
  KOCR_EZERODIV = -7       /* reference: ZeroDivisionError at tools.py:95 */
};

enum { KOCR_U8 = 0, KOCR_F32 = 1 };

/* ---- context ------------------------------------------------------------------------ */
int kocr_create(kocr_ctx** out, int hip_device);
void kocr_destroy(kocr_ctx* ctx);
const char* kocr_last_error(const kocr_ctx* ctx);
/* Run on a caller-owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream);
 * NULL restores the ctx's own stream. */
int kocr_set_stream(kocr_ctx* ctx, void* hip_stream);
int kocr_synchronize(kocr_ctx* ctx);
/* Device-buffer helpers so a host without torch can still keep data resident. */
int kocr_device_alloc(kocr_ctx* ctx, void** out, uint64_t bytes);
int kocr_device_free(kocr_ctx* ctx, void* p);
int kocr_memcpy_h2d(kocr_ctx* ctx, void* dst, const void* src, uint64_t bytes);
int kocr_memcpy_d2h(kocr_ctx* ctx, void* dst, const void* src, uint64_t bytes);

/* ---- weights ------------------------------------------------------------------------ */
/* CRAFT detector weights.  Replaces detection.build_keras_model(weights_path) +
 * load_torch_weights (detection.py:353-424, 428-468).  `names[i]` are the PyTorch
 * state-dict keys minus the "module." prefix (the naming load_torch_weights uses,
 * detection.py:432-461), e.g. "basenet.slice1.0.weight" (OIHW), "...bias",
 * BN "...weight/bias/running_mean/running_var".  shapes is n x 4 (unused dims = 1). */
int kocr_load_craft(kocr_ctx* ctx, int n, const char* const* names,
                    const float* const* data, const int64_t* shapes, const int* ranks);
/* CRNN recogniser weights.  Replaces recognition.build_model + Recognizer.__init__
 * load_weights (recognition.py:187-350, 365-404).  Names are Keras layer/variable
 * names: "conv_1/kernel" (HWIO) "conv_1/bias" ... "conv_7/...", "bn_3|5|7/gamma|beta|
 * moving_mean|moving_variance", STN localisation net "stn_conv_1/...", "stn_conv_2/...",
 * "stn_dense_1/kernel|bias", "stn_dense_2/...", "fc_9/...", "lstm_10|lstm_10_back|
 * lstm_11|lstm_11_back/kernel|recurrent_kernel|bias", "fc_12/kernel|bias". */
int kocr_load_crnn(kocr_ctx* ctx, int n, const char* const* names,
                   const float* const* data, const int64_t* shapes, const int* ranks);

/* ---- inner seam #1: detector.model.predict (detection.py:779) ------------------------ */
/* img: N x H x W x 3.  dtype KOCR_U8 = raw RGB bytes, normalised in the first conv's
 * loader exactly as detection.compute_input (detection.py:34-42); KOCR_F32 = already
 * normalised.  heat: N x (H/2) x (W/2) x 2 float32 (ch0 text, ch1 link), linear output
 * (detection.py:408-413).  micro_batch <= 0 selects the default (Keras predict's
 * batch_size=32 analogue, detection.py:779). */
int kocr_craft_forward(kocr_ctx* ctx, const void* img, int dtype, int N, int H, int W,
                       float* heat, int micro_batch, int on_device);

/* ---- detection.getBoxes (detection.py:207-287) ---------------------------------------- */
/* heat: N x h x w x 2 float32.  Thresholds as Detector.detect's keyword arguments
 * (detection.py:748-751).  boxes: N x cap x 4 x 2 float32, corner order and x2 scaling as the
 * reference (clockwise from the min(x+y) corner, or l,t,r,b for near-square boxes); counts:
 * HOST int32[N] (the call synchronises).  Box order inside an image = connected-component
 * label order of cv2.connectedComponentsWithStats (raster order of the first pixel).
 * KOCR_ECAPACITY if an image has more than cap boxes (counts still hold the true numbers);
 * KOCR_EEMPTYCONTOUR where the reference would raise IndexError (a component whose
 * segmentation map is empty after removing text AND link pixels, detection.py:246, 272). */
int kocr_get_boxes(kocr_ctx* ctx, const float* heat, int N, int h, int w, float detection_threshold,
                   float text_threshold, float link_threshold, int size_threshold, float* boxes,
                   int32_t* counts, int cap, int on_device);

/* ---- crops: recognize_from_boxes' cvtColor + tools.warpBox loop (recognition.py:506-526,
 * tools.py:61-117) ---------------------------------------------------------------------- */
/* img_rgb: N x H x W x 3 uint8 (device if on_device).  boxes: HOST float32 [M][4][2], the
 * images' boxes concatenated in image order; counts: HOST int32[N], sum = M.  crops:
 * M x target_h x target_w float32 = gray/255, zero outside the warped region (device if
 * on_device).  KOCR_EZERODIV where the reference raises ZeroDivisionError (box with integer
 * width or height 0, tools.py:95). */
int kocr_warp_crops(kocr_ctx* ctx, const uint8_t* img_rgb, int N, int H, int W, const float* boxes,
                    const int32_t* counts, int target_h, int target_w, float* crops, int on_device);

/* ---- float images (round 5) --------------------------------------------------------------------------------------------
 * The reference hands cv2 whatever dtype it is given: a float image is resized, converted to gray and warped IN FLOAT
 * (tools.py:394, recognition.py:507-526).  These are the float forms of kocr_resize_pad / kocr_warp_crops: bilinear with
 * half-pixel centres and replicated border, float32 arithmetic (horizontal then vertical pass); gray = 0.299 R + 0.587 G +
 * 0.114 B, perspective warp with 1/32-pixel source coordinates, float weights and constant-0 border, the crop NOT divided by
 * 255 (the caller's, recognition.py:524).  channels = 3 (RGB) or 1 (gray) for the warp, any for the resize.  Host pointers. */
int kocr_resize_pad_f32(kocr_ctx* ctx, const float* src, int n, int sh, int sw, int channels, int dh, int dw, int Hmax, int Wmax,
                        float cval, float* dst);
int kocr_warp_crops_f32(kocr_ctx* ctx, const float* img, int N, int H, int W, int channels, const float* boxes,
                        const int32_t* counts, int target_h, int target_w, float* crops);

/* The general form of tools.warpBox (tools.py:61-117: margin, skip_rotate, target size taken from the box,
 * return_transform): the caller states the ordered source quad and the destination quad of each of the M crops;
 * cv2.getPerspectiveTransform (8x8 float64 LU, on the device) + cv2.warpPerspective as above.  All buffers are HOST
 * arrays: src_quads / dst_quads float32 [M][4][2]; image_index int32[M] (which of the N images); crop_w / crop_h
 * int32[M] = dsize of the warp (clipped to the target); crops M x target_h x target_w float32 gray/255, zero
 * outside the crop; transforms (optional) float64 [M][3][3] = the matrices M the reference returns. */
int kocr_warp_quads(kocr_ctx* ctx, const uint8_t* img_rgb, int N, int H, int W, int M, const float* src_quads,
                    const float* dst_quads, const int32_t* image_index, const int32_t* crop_w, const int32_t* crop_h,
                    int target_h, int target_w, float* crops, double* transforms);

/* ---- inner seam #2: recognizer.prediction_model.predict (recognition.py:535) ------------ */
/* crops: M x 31 x 200 float32 in [0,1] (the (M,31,200,1) array recognize_from_boxes builds,
 * recognition.py:524-526).  labels: M x 48 int32, the CTCDecoder output: greedy decode,
 * repeats merged, blank (= n_classes-1) removed, -1 padded (recognition.py:169-184).
 * probs (may be NULL): M x 48 x n_classes float32 = recognizer.model.predict, the softmax
 * after dropping the first 2 steps (recognition.py:322-328). */
int kocr_crnn_forward(kocr_ctx* ctx, const float* crops, int M, int32_t* labels, float* probs,
                      int on_device);
/* len(alphabet) + 1 of the loaded recogniser (recognition.py:323), 0 if none is loaded. */
int kocr_crnn_classes(kocr_ctx* ctx);
/* Non-default recogniser builds (recognition.py:187-198 build_params; round 6).  `stn=False` (recognition.py:243) needs no call:
 * kocr_load_crnn on a weight set WITHOUT the stn_* tensors builds the model without the spatial transformer.
 * `rnn_steps_to_discard` (recognition.py:328; default 2): every "48" in this header -- the label rows of kocr_crnn_forward,
 * kocr_recognize_boxes, kocr_pipeline and the probability rows -- is kocr_crnn_label_width() = 50 - steps columns. */
int kocr_crnn_set_rnn_steps_to_discard(kocr_ctx* ctx, int steps);
int kocr_crnn_label_width(kocr_ctx* ctx);

/* ---- Detector.detect (detection.py:745-785): compute_input + predict + getBoxes in one call; the
 * heat-maps stay in HBM.  Arguments as kocr_craft_forward + kocr_get_boxes; counts is a HOST array. */
int kocr_detect(kocr_ctx* ctx, const void* img, int dtype, int N, int H, int W,
                float detection_threshold, float text_threshold, float link_threshold,
                int size_threshold, int micro_batch, float* boxes, int32_t* counts, int cap,
                int on_device);

/* ---- Recognizer.recognize_from_boxes (recognition.py:491-537) for N same-sized images: gray
 * conversion + warpBox crops + prediction_model.predict without leaving HBM.  boxes [M][4][2],
 * counts [N] and labels [M][48] are HOST buffers; img_rgb is a device pointer if on_device. */
int kocr_recognize_boxes(kocr_ctx* ctx, const uint8_t* img_rgb, int N, int H, int W,
                         const float* boxes, const int32_t* counts, int32_t* labels, int on_device);

/* ---- tools.resize_image + tools.pad (tools.py:356-398; pipeline.py:44-57) -------------- */
/* src: n x sh x sw x 3 uint8 (n images of one size); each is resized to dh x dw exactly as
 * cv2.resize(image, dsize=(dw, dh)) (INTER_LINEAR, uint8 fixed point) and written to the
 * top-left of an Hmax x Wmax canvas filled with cval (255 for tools.pad, tools.py:356; 0 for the
 * letterbox of Recognizer.recognize, tools.py:442, recognition.py:473-478).  dst: n x Hmax x Wmax x 3. */
int kocr_resize_pad(kocr_ctx* ctx, const uint8_t* src, int n, int sh, int sw, int dh, int dw,
                    int Hmax, int Wmax, int cval, uint8_t* dst, int on_device);

/* ---- the fused path of Pipeline.recognize (pipeline.py:28-75) ---------------------------- */
/* imgs[i]: RGB uint8 image i of size hs[i] x ws[i] (device pointers if on_device); dhs/dws:
 * its size after tools.resize_image (the caller applies the scale rule of tools.py:387-397,
 * int(h*scale), int(w*scale)); Hmax/Wmax: the batch's padded size (pipeline.py:48-57).
 * Runs resize+pad -> CRAFT -> getBoxes -> warp crops -> CRNN -> CTC decode without leaving HBM.
 * Outputs (HOST): boxes N x cap x 4 x 2 in detector-input pixels (the caller divides by its
 * scale, tools.adjust_boxes / pipeline.py:66-71), counts[N], labels [sum(counts)] x 48 (-1
 * padded, image-major, box order), n_crops = sum(counts).  KOCR_ECAPACITY if an image has more
 * than cap boxes or sum(counts) > max_crops (counts / n_crops hold the true numbers). */
int kocr_pipeline(kocr_ctx* ctx, int N, const uint8_t* const* imgs, const int32_t* hs,
                  const int32_t* ws, const int32_t* dhs, const int32_t* dws, int Hmax, int Wmax,
                  float detection_threshold, float text_threshold, float link_threshold,
                  int size_threshold, int micro_batch, float* boxes, int32_t* counts, int cap,
                  int32_t* labels, int max_crops, int32_t* n_crops, int on_device);

/* The same results where kocr_pipeline left them in HBM (round 5): d_boxes [N][cap][4][2] float32 (rows >= counts[i] of an
 * image undefined), d_counts [N] int32, d_labels [M][48] int32 (M = sum of the counts; NULL when M == 0).  Valid only until
 * the next libkocr call on this context that processes images (the buffers live in the context's arenas); KOCR_EINVAL when
 * no result is resident.  For callers that hand the results to another device-side consumer -- keras_ocr_amd.dist packs
 * them on the device and all-gathers them over RCCL without a host round trip. */
/* Capacity without recomputation (round 6).  cap / max_crops size the CALLER's buffers only: when an image has more boxes than
 * cap, or there are more crops than max_crops, kocr_pipeline still runs the whole chain once (a larger device box buffer, the
 * post-processing alone repeated on the resident heat-maps), leaves the results in HBM and returns KOCR_ECAPACITY with the
 * true counts / n_crops.  This call copies them into buffers sized from those numbers: boxes N x cap x 4 x 2 (cap >= the
 * largest count), labels n_crops x 48.  Valid until the next libkocr call on this context that processes images. */
int kocr_pipeline_results(kocr_ctx* ctx, float* boxes, int cap, int32_t* labels, int max_crops);

int kocr_pipeline_device_results(kocr_ctx* ctx, const float** d_boxes, const int32_t** d_counts, const int32_t** d_labels,
                                 int32_t* N, int32_t* cap, int32_t* M);

/* ---- single fused-epilogue convolution (unit-test seam for the MFMA kernel) ---------- */
/* out = post_a * act(pre_a * conv(in, w) + pre_b) + post_b, NHWC, stride 1, 'same'
 * padding; w is HWIO (the Keras kernel layout, detection.py:461).  pre_a/pre_b/post_a/
 * post_b are per-Cout vectors or NULL (identity).  Host pointers only. */
int kocr_conv2d_nhwc(kocr_ctx* ctx, const float* in, int N, int H, int W, int Cin,
                     const float* w_hwio, int KH, int KW, int dilation, int Cout,
                     const float* pre_a, const float* pre_b, int relu,
                     const float* post_a, const float* post_b, float* out);

/* The same seam for a CELL GRID (round 5; the layout the recogniser's conv stack runs in, keras_ocr_amd/csrc/crnn.cpp): every
 * image is one row of W / cellW cells of cellW columns, each holding an independent crop in its columns [0, cellWv) and rows
 * [1, H); row 0 and the columns behind cellWv are zero gutters (the caller passes zeros there, the result has zeros there).
 * 3x3, dilation 1, Cin % 32 == 0, Cout > 64, H % 4 == 0, W % 64 == 0, fp16 arithmetic modes only.  pool != 0: the 2x2 max
 * pooling of cell rows (2 i, 2 i + 1) -- crop rows (2 i - 1, 2 i): the flipped 'valid' pooling of recognition.py:228, 238 -- is
 * fused and pool_out [N][H/2][W/2][Cout] (cells of cellW / 2) is written instead of out (out may be NULL).  amax_out (or NULL):
 * the per-cell max |x| the kernel tracked for what it wrote, N * (W / cellW) floats.  Host pointers only. */
int kocr_conv2d_cells(kocr_ctx* ctx, const float* in, int N, int H, int W, int Cin, const float* w_hwio, int Cout,
                      const float* pre_a, const float* pre_b, int relu, const float* post_a, const float* post_b,
                      int cellW, int cellWv, int pool, float* out, float* pool_out, float* amax_out);

/* ---- arithmetic of the wide convolutions --------------------------------------------- */
/* The 3x3 / 1x1 / dilated convolutions with Cout > 32 run on the 16-bit matrix cores with fp32 operands split into
 * 16-bit pieces and fp32 accumulation:
 *   KOCR_SPLIT_BF16X3: 3 bf16 pieces, exact split, 6 products everywhere (dropped terms < 2^-21 |ab| worst case,
 *                      2^-25 |ab| rms; no operand bit is dropped);
 *   KOCR_SPLIT_F16X2:  the Winograd F(4,3) layers whose images tile as 4 x 64 or 8 x 32 pixels (the bulk of CRAFT) run on
 *                      the fp16 cores instead: 2 fp16 pieces per operand (round to nearest, <= 2^-22 |a| worst case, 2^-24
 *                      rms, while the low piece is a normal fp16), 3 products, operands scaled by exact powers of two --
 *                      per IMAGE from the max |x| its producer tracked, per output channel for the weights -- so a result
 *                      never depends on the rest of the batch.  Same measured error against fp64 as bf16x3; half the
 *                      matrix-core work of those layers.  Every other layer runs as in KOCR_SPLIT_BF16X3;
 *   KOCR_SPLIT_F16X1:  REDUCED PRECISION fast mode (opt-in, never a default): the same layers with ONE fp16 piece per
 *                      operand and one product (relative operand error 2^-12); tolerance stated in DESIGN.md section 3.
 * The environment variable KOCR_SPLIT=bf16|f16|f16x1 sets the initial mode of new contexts.  There is no reference
 * counterpart (the reference computes in TensorFlow fp32). */
#define KOCR_SPLIT_BF16X3 0
#define KOCR_SPLIT_F16X2 1
#define KOCR_SPLIT_F16X1 2
int kocr_set_split_mode(kocr_ctx* ctx, int mode);
int kocr_get_split_mode(const kocr_ctx* ctx);

/* CRAFT schedule (every arithmetic mode): two chains of convolutions without a non-linearity between them are evaluated in
 * their algebraically identical shorter form ("Folded linear layers"):
 * fold_linear_chain -- slice5.1 -> slice5.2 -> upconv1.conv.0 (detection.py:349-353, :106-108) as one composed
 * dilated 3x3 + a 1x1 over s4; fold_upsample -- conv1x1(concat(resize(y), skip)) (detection.py:106-115, 380-389)
 * as resize(conv1x1_y(y)) + conv1x1_skip(skip).  Both default to on (KOCR_LINFOLD=0 / KOCR_UPFOLD=0 in the
 * environment of kocr_create turn them off for new contexts); results differ by fp32 round-off only. */
int kocr_set_schedule(kocr_ctx* ctx, int fold_linear_chain, int fold_upsample);

/* ---- measurement -------------------------------------------------------------------- */
/* When enabled, every kernel launch on the ctx is bracketed by hipEvents on the ctx
 * stream; kocr_profile_report fills parallel arrays (up to cap rows) with per-kernel-name
 * launch count, total milliseconds and algorithmic FLOPs / bytes.  Returns the number of
 * rows available. */
int kocr_profile_enable(kocr_ctx* ctx, int on);
int kocr_profile_reset(kocr_ctx* ctx);
int kocr_profile_report(kocr_ctx* ctx, int cap, char* names /* cap x 64 */, int64_t* launches,
                        double* total_ms, double* flops, double* bytes);

/* ---- range statistics of the fp16x2 arithmetic (developer instrumentation; off = no cost) ------------- */
/* KOCR_SPLIT_F16X2 keeps fp32's 24 bits of an element only while its magnitude lies within 2^16 of its image's maximum
 * (DESIGN.md section 3).  With the statistics enabled, every fp16-arithmetic convolution first counts, over its INPUT tensor
 * and with the per-image scale 2^e it is about to use, the non-zero elements with |x| 2^e < 2^-4 (two-piece precision worse
 * than 2^-21 relative to the element) and < 2^-14 (the high piece is an fp16 subnormal: worse than 2^-11), and the share
 * of the tensor's sum |x| they carry -- one synchronising side launch per layer, so timing runs keep it off.
 * kocr_range_stats_report fills parallel arrays (up to cap rows, one per layer name, accumulated since the last enable):
 * values[i * 7 + k] = launches, elements, non-zero elements, elements below 2^-4, below 2^-14, sum |x|, sum |x| of the
 * elements below 2^-4.  Returns the number of rows. */
int kocr_range_stats_enable(kocr_ctx* ctx, int on);
int kocr_range_stats_report(kocr_ctx* ctx, int cap, char* names /* cap x 64 */, double* values /* cap x 7 */);

#ifdef __cplusplus
}
#endif
#endif /* KOCR_H */
