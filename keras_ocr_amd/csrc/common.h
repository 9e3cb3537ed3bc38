// common.h — internal (non-ABI) declarations shared by the libkocr translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <map>
#include <atomic>
#include <cstdio>
#include <cstring>
#include "../../include/kocr.h"

// ---------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------
#define KOCR_HIP(ctx, expr)                                                         \
  do {                                                                              \
    hipError_t _e = (expr);                                                         \
    if (_e != hipSuccess) {                                                         \
      (ctx)->set_err(std::string(#expr) + ": " + hipGetErrorString(_e) + " at " +   \
                     __FILE__ + ":" + std::to_string(__LINE__));                    \
      return KOCR_EHIP;                                                             \
    }                                                                               \
  } while (0)

#define KOCR_TRY(expr)            \
  do {                            \
    int _rc = (expr);             \
    if (_rc != KOCR_OK) return _rc; \
  } while (0)

#define KOCR_FAIL(ctx, code, msg) \
  do {                            \
    (ctx)->set_err(msg);          \
    return (code);                \
  } while (0)

// ---------------------------------------------------------------------------------------
// device tensors (NHWC fp32, optionally a channel slice of a wider buffer)
// ---------------------------------------------------------------------------------------
struct Tensor {
  float* p = nullptr;  // base of the (wider) buffer
  int N = 0, H = 0, W = 0, C = 0;
  int cs = 0;  // channel stride of the underlying buffer (floats per pixel)
  int co = 0;  // channel offset of this view inside the buffer
  // Optional device slots, ONE PER IMAGE (amax[n], n < N), each holding (as uint bits of a non-negative float) an UPPER
  // BOUND of max |x| over everything written into image n of the underlying buffer during this forward; every producer
  // kernel atomicMax-es into them.  The fp16 F(4,3) convolutions (conv_w43h.hip) derive the exact power-of-two input scale
  // of an image from its slot -- per image, so that a result never depends on the rest of the batch; a tensor without
  // slots is reduced on demand (launch_absmax).  Slices / views share the buffer's slots.
  unsigned* amax = nullptr;
  // Valid width (0 = W): columns [Wv, W) of every row are ZERO padding that belongs to the tensor -- its producer writes
  // zeros there, a convolution reading it sees exactly the 'same' padding it would see at the image edge.  Lets a tensor
  // whose width is not a multiple of 4 (the recogniser's 50-wide conv_6 / conv_7) run on the F(4,3) kernels at width 52.
  int Wv = 0;
  int wv() const { return Wv ? Wv : W; }
  // Cell grid (0 = none; round 5, the recogniser's crop batch): every image is ONE ROW of W / cellW cells of cellW columns,
  // each holding an independent crop in its columns [0, cellWv) and rows [1, H); row 0 and the columns [cellWv, cellW) of a
  // cell are ZERO gutters that belong to the tensor (its producer writes them) -- the 'same' padding between neighbouring
  // crops.  amax then holds one slot per CELL (image * cells() + cell).  Only conv_w43vh_kernel<.., MODE 2> takes and
  // writes such tensors; launch_conv fails loudly for anything else.
  int cellW = 0, cellWv = 0;
  int cells() const { return cellW ? W / cellW : 0; }
  size_t pixels() const { return (size_t)N * H * W; }
  Tensor slice(int off, int c) const {
    Tensor t = *this;
    t.co = co + off;
    t.C = c;
    return t;
  }
};

// ---------------------------------------------------------------------------------------
// a prepared convolution / dense layer
// ---------------------------------------------------------------------------------------
struct ConvLayer {
  std::string name;
  int Cin = 0, Cout = 0, KH = 1, KW = 1, dil = 1;
  int Kreal = 0, Kpad = 0, Cout_pad = 0, BN = 0;
  float* d_w = nullptr;       // [Kpad][Cout_pad], k = (ky*KW + kx)*Cin + c
  float* d_pre_a = nullptr;   // [Cout_pad]  v = acc*pre_a + pre_b
  float* d_pre_b = nullptr;
  float* d_post_a = nullptr;  // [Cout_pad] or nullptr: out = relu?(v)*post_a + post_b
  float* d_post_b = nullptr;
  int relu = 0;
  float* d_w_rgb4 = nullptr;  // first layer on raw uint8: [9 taps][R,G,B,0][Cout_pad] (48 rows)
  unsigned short* d_ws = nullptr;  // Winograd F(2,3) weights, 3-way bf16 split, conv_wsplit.hip order (Cout > 32)
  int ws_cout_pad = 0;
  unsigned short* d_w4 = nullptr;  // Winograd F(4,3) weights, 3-way bf16 split, conv_w43.hip order (Cout > 64, Cin % 32 == 0)
  int w4_cout_pad = 0;
  // the same weights for the fp16 kernels (conv_w43h.hip): U * 2^wexp[o] split into two fp16 pieces (round to nearest), the
  // per-cout exponent chosen so that max |U 2^wexp| over the cout's weights lies in [2^14, 2^15); d_pre_a_h[o] =
  // pre_a[o] * 2^-wexp[o] undoes it in the epilogue
  unsigned short* d_w4h = nullptr;
  float* d_pre_a_h = nullptr;
  unsigned short* d_ds = nullptr;  // direct-conv weights, 3-way bf16 split, conv_dsplit.hip order
  int ds_cout_pad = 0;
  unsigned short* d_first = nullptr;  // 3 -> <= 64 first layer on raw uint8, im2col K = 27 -> 32, conv_hsplit.hip order
  float first_bound = 0.f;            // ... an upper bound of |output| over every uint8 image (prepare_conv)
  unsigned short* d_hs = nullptr;  // 3x3, <= 32 couts: 3-way bf16 split, conv_hsplit.hip order
  unsigned short* d_hs16 = nullptr;  // the same for <= 16 couts: tap pairs on the 16x16x32 MFMA (conv_hs16_kernel)
  unsigned short* d_hsh = nullptr;   // conv_hsh_kernel: the <= 32-cout weights scaled per cout and split into two fp16 pieces
  std::vector<int> hs_wexp;          // ... their exponents (d_pre_a_h = pre_a 2^-wexp is uploaded by prepare_conv)
  unsigned short* d_k5 = nullptr;  // 5x5, 16 couts, small images: 3-way bf16 split, conv_k5.hip order
  bool tap_inner = false;  // K order [16-channel group][tap][16] (Cin % 16 == 0) instead of [tap][Cin]
  bool ready() const { return d_w != nullptr; }
};

// ---------------------------------------------------------------------------------------
// profiler: per kernel name, HIP-event bracketed launches on the ctx stream
// ---------------------------------------------------------------------------------------
struct ProfRow {
  int64_t launches = 0;
  double ms = 0, flops = 0, bytes = 0;
};

// range statistics of the fp16x2 arithmetic per convolution layer (elementwise.hip: launch_range_stats)
struct RangeRow {
  double launches = 0, elements = 0, nonzero = 0, below_m4 = 0, below_m14 = 0, sum_abs = 0, sum_abs_below_m4 = 0;
};

struct Arena {
  char* base = nullptr;
  size_t cap = 0, off = 0;
};

struct CraftNet;
struct CrnnNet;

struct kocr_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string err;
  void set_err(const std::string& s) { err = s; }

  int split_mode = KOCR_SPLIT_F16X2;  // KOCR_SPLIT_BF16X3 / KOCR_SPLIT_F16X2 (default since round 4) / KOCR_SPLIT_F16X1
  // CRAFT schedule options (craft.cpp: folded linear layers); read ONCE from KOCR_LINFOLD / KOCR_UPFOLD when the
  // context is created, changed through kocr_set_schedule
  bool opt_linfold = true, opt_upfold = true;
  // kernel selection switches, read ONCE per context in kocr_create from the environment variable of the same name in
  // upper case with the KOCR_ prefix (KOCR_LSTM16=0 ...): the fallback kernels can be exercised per context, inside
  // one process (the older switches of the convolution dispatch are still process-wide)
  struct Switches {
    bool dense_splitk = true;  // stn_dense_1 on the split-K kernel
    bool lstm16 = true;        // round-3 LSTM kernel (16 crops per workgroup, U resident in registers)
    bool hs16 = true;          // <= 16-cout 3x3 layers on the 16x16x32 MFMA
    bool k5 = true;            // 5x5 / 16-cout kernel (stn_conv_1)
    bool up2x = true;          // shared-tap epilogue for exact 2x up-sampling in the decoder 1x1s
    bool w43h = true;          // fp16 F(4,3) kernels in the fp16 arithmetic modes
  } sw;

  // max-|x| slots of the tensors of the current forward (see Tensor::amax): zeroed by amax_begin()
  unsigned* d_amax = nullptr;
  int amax_used = 0;
  static constexpr int AMAX_SLOTS = 1 << 16;
  int amax_begin();            // (re)start slot allocation, zero the slots on the ctx stream
  unsigned* amax_slots(int n); // the next n slots (one per image; nullptr when exhausted or when the arithmetic in use never
                               // reads them: the consumer then reduces on demand)

  // workspace arenas (bump allocated per call, grown on demand): ws = network activations,
  // pp = post-processing per-pixel scratch, pp2 = post-processing canvases, io = staging,
  // pl = buffers that live for a whole kocr_pipeline call (padded batch, heat-maps, boxes)
  Arena ws, pp, pp2, io, pl;
  Arena bx;  // kocr_pipeline: the box buffer of a page with more boxes than the caller's cap (the pl arena holds the heat-maps)
  int ws_reserve(size_t bytes);
  void ws_reset() { ws.off = 0; }
  void* ws_alloc(size_t bytes);

  // persistent allocations (weights)
  std::vector<void*> owned;
  int dev_alloc(void** out, size_t bytes);
  int upload(float** out, const std::vector<float>& host);
  void release(void* p);  // hipFree one persistent allocation (a re-prepared layer's previous weights); nullptr is ignored

  CraftNet* craft = nullptr;
  CrnnNet* crnn = nullptr;

  // the device-resident results of the last successful kocr_pipeline call (kocr_pipeline_device_results): boxes in the pl
  // arena, counts in the post-processing scratch, label rows in the io arena -- all valid until the next call that uses
  // those arenas, which clears `valid` first
  struct LastPipeline {
    const float* d_boxes = nullptr;
    const int32_t* d_counts = nullptr;
    const int32_t* d_labels = nullptr;
    int N = 0, cap = 0, M = 0;
    bool valid = false;
  } last_pl;

  // developer instrumentation of the fp16x2 range assumption (kocr_range_stats_enable): off = no cost
  bool range_on = false;
  void* d_range = nullptr;
  std::map<std::string, RangeRow> range;

  // profiler
  bool prof_on = false;
  std::map<std::string, ProfRow> prof;
  struct Pending {
    std::string name;
    hipEvent_t a, b;
  };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> ev_pool;
  hipEvent_t get_event();
  void prof_begin(const char* name, double flops, double bytes);
  void prof_end();
  int prof_flush();
};

int arena_reserve(kocr_ctx* ctx, Arena& a, size_t bytes);
inline void* arena_alloc(Arena& a, size_t bytes) {
  const size_t o = (a.off + 255) & ~(size_t)255;
  if (o + bytes > a.cap) return nullptr;
  a.off = o + bytes;
  return a.base + o;
}
inline void* kocr_ctx::ws_alloc(size_t bytes) { return arena_alloc(ws, bytes); }

// RAII-less helper used around launches
struct ProfScope {
  kocr_ctx* c;
  ProfScope(kocr_ctx* ctx, const char* name, double flops, double bytes) : c(ctx) {
    if (c->prof_on) c->prof_begin(name, flops, bytes);
  }
  ~ProfScope() {
    if (c->prof_on) c->prof_end();
  }
};

// wave-level max |x| into a Tensor::amax slot (device code; values are non-negative, so uint order = float order)
#if defined(__HIPCC__)
// max over the wave of a value that is >= 0 (not NaN) in every lane, as its bits, wave-uniform.  Non-negative floats order
// like their bit patterns, so the cross-row step is four v_readlane + scalar max; the in-row steps are DPP modifiers.  No
// lane-index operand anywhere: the __shfl_xor version kept six (lane ^ o) * 4 addresses alive across a persistent kernel's
// K loop -- spilled there, and reloaded one by one (each a full memory round trip) in every tile's epilogue.
__device__ __forceinline__ unsigned kocr_wave_max_bits(float m) {
  int v = __float_as_int(m);
  int t;
  t = __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);  // quad_perm [1, 0, 3, 2]
  v = v > t ? v : t;
  t = __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);  // quad_perm [2, 3, 0, 1]
  v = v > t ? v : t;
  t = __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);  // row_half_mirror
  v = v > t ? v : t;
  t = __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);  // row_mirror: every lane holds its row's maximum
  v = v > t ? v : t;
  const unsigned a = (unsigned)__builtin_amdgcn_readlane(v, 0), b = (unsigned)__builtin_amdgcn_readlane(v, 16),
                 c = (unsigned)__builtin_amdgcn_readlane(v, 32), d = (unsigned)__builtin_amdgcn_readlane(v, 48);
  const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
  return ab > cd ? ab : cd;
}
__device__ __forceinline__ unsigned kocr_amax_peek(const unsigned* slot) {
  // a relaxed agent-scope atomic load: read at L2 like a volatile read, but without the s_waitcnt vmcnt(0) hipcc puts
  // behind every volatile access
  return __hip_atomic_load(const_cast<unsigned*>(slot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The slot's value as the caller read it beforehand (kocr_amax_peek at the start of its epilogue, consumed here after the
// epilogue's arithmetic): the read's latency is hidden instead of being waited for tile after tile.
// One address for a whole image: the atomic is issued only when it can still raise the slot (a stale read merely costs a
// redundant atomic).  Unconditional atomics from every tile serialise in L2 -- measured 4x on the first layer.
__device__ __forceinline__ void kocr_amax_update_known(unsigned* slot, float m, unsigned seen) {
  const unsigned bits = kocr_wave_max_bits(m);
  if (bits > seen) {  // uniform
    int l = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(l));  // not a loop invariant to hoist (and spill) out of the caller's tile loop
    if (l == 0) atomicMax(slot, bits);
  }
}
__device__ __forceinline__ void kocr_amax_update(unsigned* slot, float m) { kocr_amax_update_known(slot, m, kocr_amax_peek(slot)); }
#endif

// ---------------------------------------------------------------------------------------
// kernels (launchers)
// ---------------------------------------------------------------------------------------
// conv_mfma.hip
int prepare_conv(kocr_ctx* ctx, ConvLayer& L, const float* w, bool w_is_oihw, int Cin, int Cout,
                 int KH, int KW, int dil, const float* pre_a, const float* pre_b, int relu,
                 const float* post_a, const float* post_b);
// in_u8 != nullptr: first-layer mode, raw RGB bytes + LUT normalisation.
// developer instrumentation (KOCR_DISPATCH_LOG=<file>): one line per distinct (kernel family, layer, input shape) the dispatcher
// of launch_conv_pool chose in this process -- the table of DESIGN.md section 3 "which shapes reach which family" comes from it
void kocr_note_dispatch(const char* family, const ConvLayer& L, const Tensor& in);
int launch_conv(kocr_ctx* ctx, const ConvLayer& L, const Tensor& in, const uint8_t* in_u8,
                const float* lut, const Tensor& out);
int launch_conv_pool(kocr_ctx* ctx, const ConvLayer& L, const Tensor& in, const uint8_t* in_u8,
                     const float* lut, const Tensor& out, const Tensor* pool, bool need_full = true);
// Split arithmetic of conv_wsplit.hip / conv_dsplit.hip: 0 = bf16 x 3 pieces / 6 products (exact split),
// 1 = fp16 x 2 pieces / 3 products (RNE split at 2^-24, exact power-of-two scaling from Tensor::amax).
// kocr_ctx::split_mode; KOCR_SPLIT=bf16|f16 sets the initial value, kocr_set_split_mode changes it.
// conv_wsplit.hip
int prepare_wsplit(kocr_ctx* ctx, ConvLayer& L, const float* w, bool w_is_oihw);
bool wsplit_applicable(const ConvLayer& L, const Tensor& in);
int launch_conv_wsplit(kocr_ctx* ctx, const ConvLayer& L, const Tensor& in, const Tensor& out, const Tensor* pool,
                       bool need_full);
// conv_w43.hip: Winograd F(4,3) on the bf16 cores (bf16x3 mode; 3.0 issued FLOPs per algorithmic FLOP instead of 4.0)
int prepare_w43(kocr_ctx* ctx, ConvLayer& L, const float* w, bool w_is_oihw);
// conv_w43h.hip: the same algebra on the fp16 cores, operands scaled by exact powers of two (per image / per cout) and
// split into two fp16 pieces, three products (KOCR_SPLIT_F16X2: 1.5 issued FLOPs per algorithmic FLOP) or cut to ONE
// fp16 piece, one product (KOCR_SPLIT_F16X1, the reduced-precision fast mode: 0.5).  launch_conv_w43 dispatches to it.
struct W4Params;
int prepare_w43h(kocr_ctx* ctx, ConvLayer& L, const float* w, bool w_is_oihw, const float* pre_a);
// mode: 0 = the image tiles exactly, 1 = ragged (any H, W), 2 = cell grid (launch_w43vh, geo 1 only)
int launch_w43vh(kocr_ctx* ctx, W4Params& p, bool fuse, int geo, int pieces, int mode = 0);
int launch_w43rh(kocr_ctx* ctx, W4Params& p, bool fuse, int pieces, int mode = 0);
int launch_w43fh(kocr_ctx* ctx, W4Params& p, int pieces);
bool w43_applicable(const kocr_ctx* ctx, const ConvLayer& L, const Tensor& in);
bool w43_cells_ok(const kocr_ctx* ctx, const ConvLayer& L);   // a cell-grid tensor (Tensor::cellW) can run through layer L
bool w43_flat_h_ok(const kocr_ctx* ctx, const ConvLayer& L);  // a width-padded tensor (Tensor::Wv) can
int launch_conv_w43(kocr_ctx* ctx, const ConvLayer& L, const Tensor& in, const Tensor& out, const Tensor* pool,
                    bool need_full);
// conv_dsplit.hip
int prepare_dsplit(kocr_ctx* ctx, ConvLayer& L, const float* w, bool w_is_oihw);
bool dsplit_applicable(const ConvLayer& L, const Tensor& in);
bool dsplit_usable(const ConvLayer& L, const Tensor& in);
// conv_hsplit.hip
int prepare_hsplit(kocr_ctx* ctx, ConvLayer& L, const float* w, bool w_is_oihw);
bool hsplit_applicable(kocr_ctx* ctx, const ConvLayer& L, const Tensor& in);
int launch_conv_hsplit(kocr_ctx* ctx, const ConvLayer& L, const Tensor& in, const Tensor& out);
int prepare_first(kocr_ctx* ctx, ConvLayer& L, const float* w, bool w_is_oihw);
bool first_applicable(kocr_ctx* ctx, const ConvLayer& L, const Tensor& in);
int launch_conv_first(kocr_ctx* ctx, const ConvLayer& L, const Tensor& in, const uint8_t* in_u8, const float* lut, const Tensor& out);
int launch_conv_dsplit(kocr_ctx* ctx, const ConvLayer& L, const Tensor& in, const Tensor& out, const Tensor* up = nullptr);
// conv_k5.hip
int prepare_k5(kocr_ctx* ctx, ConvLayer& L, const float* w, bool w_is_oihw);
bool k5_applicable(kocr_ctx* ctx, const ConvLayer& L, const Tensor& in, const Tensor& out);
int launch_conv_k5(kocr_ctx* ctx, const ConvLayer& L, const Tensor& in, const Tensor& out);
// elementwise.hip
// row_off: input row that starts output row 0 (0 = keras 'valid' pooling; 1 = the same pooling seen
// through a vertical flip of an odd-height tensor, as in the CRNN's natural-orientation conv stack)
int launch_maxpool2x2(kocr_ctx* ctx, const Tensor& in, const Tensor& out, int row_off = 0);
int launch_maxpool3x3s1(kocr_ctx* ctx, const Tensor& in, const Tensor& out);
int launch_resize_bilinear(kocr_ctx* ctx, const Tensor& in, const Tensor& out);
int launch_copy_channels(kocr_ctx* ctx, const Tensor& in, const Tensor& out);
// per-image max |x| of a tensor into t.N slots (atomicMax; the slots are NOT cleared), and slot-to-slot propagation
int launch_absmax(kocr_ctx* ctx, const Tensor& t, unsigned* slots);
int launch_amax_copy(kocr_ctx* ctx, const unsigned* from, unsigned* to, int n);
int launch_range_stats(kocr_ctx* ctx, const std::string& name, const Tensor& t, const unsigned* slots, int top);
// conv_cls.6 + conv_cls.8 of the CRAFT head in one pass (16 -> 16 ReLU -> 2), heat-map written densely
int launch_head_tail(kocr_ctx* ctx, const ConvLayer& L6, const ConvLayer& L8, const Tensor& in, float* d_heat);

// craft.cpp
int craft_load(kocr_ctx* ctx, int n, const char* const* names, const float* const* data,
               const int64_t* shapes, const int* ranks);
int craft_forward(kocr_ctx* ctx, const void* d_img, int dtype, int N, int H, int W, float* d_heat);
size_t craft_workspace_bytes(int N, int H, int W);
void craft_free(kocr_ctx* ctx);

// crnn.cpp
void crnn_free(kocr_ctx* ctx);
int crnn_load(kocr_ctx* ctx, int n, const char* const* names, const float* const* data, const int64_t* shapes,
              const int* ranks);
int crnn_classes(kocr_ctx* ctx);
int crnn_label_width(kocr_ctx* ctx);  // 50 - rnn_steps_to_discard (48)
int crnn_set_discard(kocr_ctx* ctx, int d);
size_t crnn_workspace_bytes(int M, int n_classes);
int crnn_forward(kocr_ctx* ctx, const float* d_crops, int M, int* d_labels, float* d_probs);

// postproc.hip
// dev (optional): device-side results for a caller that keeps going on the stream -- the per-image counts and the
// totals block (totals[2] = components whose contour list is empty); with dev set, the final synchronisation that
// fetches n_empty is skipped and the caller reads totals[2] itself.  Both pointers live until the next call.
struct PPDeviceOut {
  int* d_counts = nullptr;
  int* d_totals = nullptr;
};
int postproc_get_boxes(kocr_ctx* ctx, const float* d_heat, int N, int h, int w, float det_thr,
                       float text_thr, float link_thr, int size_thr, float* d_boxes, int cap,
                       int* h_counts, int* n_empty_out, PPDeviceOut* dev = nullptr);

// warp.hip
struct WarpParam {
  double mi[9];  // inverse homography (crop px -> image px)
  int img;       // image index
  int cw, ch;    // crop size (<= target); outside -> 0
  int pad;
};
int warp_prepare(const float* box, int target_h, int target_w, WarpParam* out, float* ordered_box);
int launch_warp(kocr_ctx* ctx, const uint8_t* d_img, int H, int W, const WarpParam* d_prm, int M, int th,
                int tw, float* d_crops);
// the same set-up on the device, one thread per box slot of d_boxes[N][cap][4][2] (d_counts[N] on the device);
// *d_status = max return code of warp_prepare over all boxes (caller zeroes it)
int launch_warp_prepare(kocr_ctx* ctx, const float* d_boxes, const int* d_counts, int N, int cap, int th, int tw,
                        WarpParam* d_prm, int* d_status);
int launch_warp_quads(kocr_ctx* ctx, const float* d_src, const float* d_dst, const int* d_img, const int* d_cw,
                      const int* d_ch, int M, WarpParam* d_prm, double* d_mfwd, int* d_status);

// imgproc.hip
int launch_resize_pad_f32(kocr_ctx* ctx, const float* d_src, int n, int sh, int sw, int C, float* d_dst, int dh, int dw, int Hmax,
                          int Wmax, float cval, Arena& tab_arena);
int launch_warp_f32(kocr_ctx* ctx, const float* d_img, int H, int W, int C, const WarpParam* d_prm, int M, int th, int tw, float* d_crops);
int launch_resize_pad(kocr_ctx* ctx, const uint8_t* d_src, int n, int sh, int sw, uint8_t* d_dst, int dh, int dw,
                      int Hmax, int Wmax, int cval, Arena& tab_arena);
