// probe_clock.h -- developer probe, never defined in the product build (-DKOCR_CLOCK_PROBE on one translation unit, linked
// into a scratch copy of the library): where a persistent block's time goes.  PROBE_T(i) adds the shader clocks since the
// previous mark to bin i (s_memtime is issued in program order, so a bin holds the waits of the instructions before its
// mark); PROBE_TEND adds one thread's bins to a per-translation-unit device array that the launcher prints per launch
// (profiles/r04_ab_notes.txt items 10, 11 were measured with it).
#pragma once
#ifdef KOCR_CLOCK_PROBE
#include <cstdio>
static __device__ unsigned long long kocr_probe_clk[8];
#define PROBE_T0() unsigned long long probe_tl_ = __builtin_amdgcn_s_memtime(), probe_ta_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PROBE_T(i)                                                   \
  {                                                                  \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    probe_ta_[i] += t_ - probe_tl_;                                  \
    probe_tl_ = t_;                                                  \
  }
#define PROBE_RESTART() probe_tl_ = __builtin_amdgcn_s_memtime()
#define PROBE_TEND(cond, a, b) \
  if (cond)                    \
    for (int i_ = a; i_ < b; ++i_) atomicAdd(&kocr_probe_clk[i_], probe_ta_[i_])
#define PROBE_RESET(ctx)                                                         \
  unsigned long long probe_z_[8] = {0, 0, 0, 0, 0, 0, 0, 0};                     \
  KOCR_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(kocr_probe_clk), probe_z_, sizeof probe_z_))
// prints bins / blocks in microseconds
#define PROBE_REPORT(ctx, what, blocks)                                                                  \
  {                                                                                                      \
    KOCR_HIP(ctx, hipStreamSynchronize(ctx->stream));                                                    \
    KOCR_HIP(ctx, hipMemcpyFromSymbol(probe_z_, HIP_SYMBOL(kocr_probe_clk), sizeof probe_z_));           \
    fprintf(stderr, "PROBE %s blocks %d: kilo-clocks (s_memtime) per block:", what, (int)(blocks));                          \
    for (int i_ = 0; i_ < 8; ++i_) fprintf(stderr, " [%d] %.1f", i_, probe_z_[i_] * 0.001 / (blocks));   \
    fprintf(stderr, "\n");                                                                               \
  }
#else
#define PROBE_T0()
#define PROBE_T(i)
#define PROBE_RESTART()
#define PROBE_TEND(cond, a, b)
#define PROBE_RESET(ctx)
#define PROBE_REPORT(ctx, what, blocks)
#endif
