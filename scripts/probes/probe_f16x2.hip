// probe_f16x2.hip — developer probe: fp32 emulation by a 2-way fp16 split (RNE) with 3 products on
// v_mfma_f32_32x32x16_f16: accuracy against fp64, and whether fp16 denormal operands are honoured.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void k_split(const float* A, const float* B, float* C, int K, int terms, float sa, float sb) {
  const int lane = threadIdx.x, row = lane & 31, kg = lane >> 5;
  f16v acc = {0};
  for (int k0 = 0; k0 < K; k0 += 16) {
    h8 ah, al, bh, bl;
    for (int j = 0; j < 8; ++j) {
      const float a = A[row * K + k0 + kg * 8 + j] * sa, b = B[(k0 + kg * 8 + j) * 32 + row] * sb;
      const _Float16 h = (_Float16)a; ah[j] = h; al[j] = (_Float16)(a - (float)h);
      const _Float16 g = (_Float16)b; bh[j] = g; bl[j] = (_Float16)(b - (float)g);
    }
    if (terms >= 4) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bl, acc, 0, 0, 0);
    if (terms >= 3) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
  }
  const float inv = 1.f / (sa * sb);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[r] * inv;
}

int main() {
  const int K = 4608;
  std::vector<float> A(32 * K), B(K * 32), C(1024);
  srand(1);
  auto rnd = []() { float s = 0; for (int i = 0; i < 12; ++i) s += rand() / (float)RAND_MAX; return s - 6.f; };
  for (auto& v : A) v = fmaxf(rnd(), 0.f);
  for (auto& v : B) v = 0.02f * rnd();
  float *dA, *dB, *dC;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
  auto run = [&](const char* nm, float ascale, int terms, float sa, float sb) {
    std::vector<float> A2(A);
    for (auto& v : A2) v *= ascale;
    std::vector<double> T(1024), S(1024);
    std::vector<float> F(1024);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double t = 0, s = 0; float f = 0;
      for (int k = 0; k < K; ++k) { t += (double)A2[i * K + k] * B[k * 32 + j]; s += fabs((double)A2[i * K + k] * B[k * 32 + j]); f = fmaf(A2[i * K + k], B[k * 32 + j], f); }
      T[i * 32 + j] = t; S[i * 32 + j] = s; F[i * 32 + j] = f;
    }
    hipMemcpy(dA, A2.data(), A2.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_split, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, terms, sa, sb);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    double mx = 0, rms = 0, mxf = 0, rmsf = 0;
    for (int i = 0; i < 1024; ++i) {
      const double e = (C[i] - T[i]) / S[i], ef = (F[i] - T[i]) / S[i];
      mx = fmax(mx, fabs(e)); rms += e * e; mxf = fmax(mxf, fabs(ef)); rmsf += ef * ef;
    }
    printf("%-44s max %.3e rms %.3e   (fp32 fma chain: max %.3e rms %.3e)\n", nm, mx, sqrt(rms / 1024), mxf, sqrt(rmsf / 1024));
  };
  run("f16x2, 3 products, data O(1), no scaling", 1.f, 3, 1.f, 1.f);
  run("f16x2, 4 products, data O(1), no scaling", 1.f, 4, 1.f, 1.f);
  run("f16x2, 1 product", 1.f, 1, 1.f, 1.f);
  run("f16x2, 3 products, A*1e-4, no scaling", 1e-4f, 3, 1.f, 1.f);
  run("f16x2, 3 products, A*1e-4, scaled to 2^12", 1e-4f, 3, 4096.f * 8192.f, 4096.f * 32.f);
  run("f16x2, 3 products, data O(1), scaled to 2^12", 1.f, 3, 1024.f, 4096.f * 32.f);
  // denormal operand check: a = 2^-20 (fp16 denormal), b = 1 -> hi-only product should be 16 * 2^-20 per row
  std::vector<float> A3(32 * 16, ldexpf(1.f, -20)), B3(16 * 32, 1.f);
  hipMemcpy(dA, A3.data(), A3.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B3.data(), B3.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_split, dim3(1), dim3(64), 0, 0, dA, dB, dC, 16, 1, 1.f, 1.f);
  hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
  printf("denormal fp16 operand 2^-20 x 16: got %.6e expect %.6e (%s)\n", C[0], 16 * ldexp(1.0, -20), C[0] == 0.f ? "FLUSHED" : "honoured");
  return 0;
}
