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
typedef float v2f __attribute__((ext_vector_type(2)));
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
__device__ __forceinline__ void kocr_split4_h(const v4f v, u2v& h, u2v& l) {
  _Float16 hh[4], ll[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    hh[c] = (_Float16)v[c];
    ll[c] = (_Float16)(v[c] - (float)hh[c]);
  }
  h = u2v{__builtin_bit_cast(unsigned, hf2{hh[0], hh[1]}), __builtin_bit_cast(unsigned, hf2{hh[2], hh[3]})};
  l = u2v{__builtin_bit_cast(unsigned, hf2{ll[0], ll[1]}), __builtin_bit_cast(unsigned, hf2{ll[2], ll[3]})};
}

// exponent e of the exact input scale 2^e: with E = exponent of the tracked max |x| (Tensor::amax), e = top - E puts
// the operands below 2^(top + 1 + growth of the input transform).  amax == 0 -> e = 0.
__device__ __forceinline__ int kocr_scale_exp(const unsigned* amax, int top) {
  const unsigned b = *amax;
  if (b == 0) return 0;
  int e = top - ((int)(b >> 23) - 127);
  return e < -100 ? -100 : (e > 100 ? 100 : e);
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
