// split_common.h — device / host helpers shared by the split-arithmetic convolution kernels (conv_wsplit.hip,
// conv_dsplit.hip, conv_w43.hip): operand splitting, exact power-of-two scaling, XCD-aware tile order.
#pragma once
#include "common.h"
#include <cstring>

typedef short bf8 __attribute__((ext_vector_type(8)));
typedef _Float16 hf8 __attribute__((ext_vector_type(8)));
typedef _Float16 hf2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));

// blockIdx -> tile: workgroups are dealt round-robin to the 8 XCDs (each with a private L2); this bijection gives
// every XCD a CONTIGUOUS range of tiles so that neighbouring tiles share one L2.
__device__ __forceinline__ int kocr_xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7;
  const int q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// Exact 3-way split of four fp32 values by TRUNCATION, packed as 4 bf16 (8 bytes) per piece: h = top 8
// significand bits of v, m = top 8 bits of v - h, l = the rest; |m| < 2^-7 |v|, |l| < 2^-14 |v|.  (A
// round-to-nearest split -- the weights use one, on the host -- would give |m| <= 2^-9, |l| <= 2^-18 at the
// same instruction count, but its mixed-sign pieces cost 5 % end to end on these power-bound kernels.)
__device__ __forceinline__ void kocr_split4(const v4f v, u2v& h, u2v& m, u2v& l) {
  unsigned uh[4], um[4], ul[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    uh[c] = __float_as_uint(v[c]) & 0xFFFF0000u;
    const float r = v[c] - __uint_as_float(uh[c]);
    um[c] = __float_as_uint(r) & 0xFFFF0000u;
    ul[c] = __float_as_uint(r - __uint_as_float(um[c]));
  }
  // perm(a, b, 0x07060302) = (a & 0xFFFF0000) | (b >> 16)
  h = u2v{__builtin_amdgcn_perm(uh[1], uh[0], 0x07060302u), __builtin_amdgcn_perm(uh[3], uh[2], 0x07060302u)};
  m = u2v{__builtin_amdgcn_perm(um[1], um[0], 0x07060302u), __builtin_amdgcn_perm(um[3], um[2], 0x07060302u)};
  l = u2v{__builtin_amdgcn_perm(ul[1], ul[0], 0x07060302u), __builtin_amdgcn_perm(ul[3], ul[2], 0x07060302u)};
}

// the same for two values: one dword (2 bf16) per piece
__device__ __forceinline__ void kocr_split2(const v2f v, unsigned& h, unsigned& m, unsigned& l) {
  unsigned uh[2], um[2], ul[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    uh[c] = __float_as_uint(v[c]) & 0xFFFF0000u;
    const float r = v[c] - __uint_as_float(uh[c]);
    um[c] = __float_as_uint(r) & 0xFFFF0000u;
    ul[c] = __float_as_uint(r - __uint_as_float(um[c]));
  }
  h = __builtin_amdgcn_perm(uh[1], uh[0], 0x07060302u);
  m = __builtin_amdgcn_perm(um[1], um[0], 0x07060302u);
  l = __builtin_amdgcn_perm(ul[1], ul[0], 0x07060302u);
}

// fp16 mode: 2-way RNE fp16 split of four values, v ~ h + l with |v - h - l| <= 2^-22 |v| (2^-24 rms) while l is a
// normal fp16 (the caller scales the tensor by an exact power of two so that max |v| ~ 2^14)
// v - float(h) for the fp16 value in the low / high half of `hh`, as ONE mixed-precision fma (v_fma_mix_f32: the f16 source is
// widened inside the instruction; the result is exact, h being v rounded to 11 bits) instead of a conversion plus a
// subtraction -- hipcc canonicalises fma(float(h), -1, v) back into the two-instruction form, hence the inline assembly
__device__ __forceinline__ float kocr_resid_lo(unsigned hh, float v) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hh), "v"(v));
  return r;
}
__device__ __forceinline__ float kocr_resid_hi(unsigned hh, float v) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hh), "v"(v));
  return r;
}
__device__ __forceinline__ void kocr_split2_h(const v2f v, unsigned& h, unsigned& l) {
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, hf2));  // v_cvt_pk_f16_f32, round to nearest
  const v2f r = {kocr_resid_lo(h, v[0]), kocr_resid_hi(h, v[1])};
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, hf2));
}
__device__ __forceinline__ void kocr_split4_h(const v4f v, u2v& h, u2v& l) {
  unsigned h0, l0, h1, l1;
  kocr_split2_h(v2f{v[0], v[1]}, h0, l0);
  kocr_split2_h(v2f{v[2], v[3]}, h1, l1);
  h = u2v{h0, h1};
  l = u2v{l0, l1};
}

// exponent e of the exact input scale 2^e: with E = exponent of the tracked max |x| (Tensor::amax), e = top - E puts
// the operands below 2^(top + 1 + growth of the input transform).  amax == 0 -> e = 0.
__device__ __forceinline__ int kocr_scale_exp(const unsigned* amax, int top) {
  const unsigned b = *amax;
  if (b == 0) return 0;
  int e = top - ((int)(b >> 23) - 127);
  return e < -100 ? -100 : (e > 100 ? 100 : e);
}
// ... from the bits of a slot the caller fetched itself
__device__ __forceinline__ int kocr_scale_exp_bits(unsigned b, int top) {
  if (b == 0) return 0;
  int e = top - ((int)(b >> 23) - 127);
  return e < -100 ? -100 : (e > 100 ? 100 : e);
}
// A dword at a wave-uniform address in memory that nothing writes while this kernel runs (the max-|x| slots of an INPUT
// tensor): a scalar load through the constant address space, scheduled and waited for by the compiler -- no VGPR and no
// vmcnt round trip (a vector load of it in a persistent kernel's per-tile code costs one full memory latency per tile: one
// wave per SIMD has nothing else to run meanwhile).
__device__ __forceinline__ unsigned kocr_sload(const unsigned* p) {
  typedef const unsigned __attribute__((address_space(4))) cu4;
  return *(cu4*)(unsigned long long)p;
}
// The lane id recomputed (two mbcnt) and made opaque: per-tile code of a persistent kernel that derives its lane terms
// from this instead of from values computed before the tile loop does not keep those alive across the K loop -- where the
// F(4,3) kernels have no register to spare, so they are spilled and come back through one serialised scratch load each,
// a full memory round trip per value and tile at one wave per SIMD.
__device__ __forceinline__ int kocr_fresh_lane() {
  int l = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  asm volatile("" : "+v"(l));
  return l;
}
__device__ __forceinline__ float kocr_pow2(int e) { return __uint_as_float((unsigned)(127 + e) << 23); }

// host: round-to-nearest-even 3-way bf16 split of a weight (finite inputs)
static inline void kocr_split3_host(float v, unsigned short out[3]) {
  float r = v;
  for (int s = 0; s < 3; ++s) {
    uint32_t u;
    memcpy(&u, &r, 4);
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
    float h;
    memcpy(&h, &u, 4);
    out[s] = (unsigned short)(u >> 16);
    r = r - h;
  }
}
