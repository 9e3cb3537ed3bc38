// probe_bf16x3.hip — developer probe (not part of libkocr): (1) operand layout of v_mfma_f32_32x32x16_bf16,
// (2) accuracy of fp32 emulation by a 3-way bf16 split with 6 products against an fp64 truth and a plain fp32
// fma chain, incl. the SIGNED mean error (would reveal truncating accumulation inside the matrix core).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstdint>
typedef short bf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ inline void split3(float v, unsigned short& h, unsigned short& m, unsigned short& l) {
  const unsigned uh = __float_as_uint(v) & 0xFFFF0000u;
  const float r = v - __uint_as_float(uh);
  const unsigned um = __float_as_uint(r) & 0xFFFF0000u;
  const float r2 = r - __uint_as_float(um);
  h = uh >> 16; m = um >> 16; l = __float_as_uint(r2) >> 16;
}

// C[32][32] = A[32][K] * B[K][32]; A, B fp32 row-major; one wave.  terms: 1 (hi only), 3, 6, 9
__global__ void k_split(const float* A, const float* B, float* C, int K, int terms) {
  const int lane = threadIdx.x, row = lane & 31, kg = lane >> 5;
  f16v acc = {0};
  for (int k0 = 0; k0 < K; k0 += 16) {
    bf8 a[3], b[3];
    for (int j = 0; j < 8; ++j) {
      unsigned short h, m, l;
      split3(A[row * K + k0 + kg * 8 + j], h, m, l);
      a[0][j] = h; a[1][j] = m; a[2][j] = l;
      split3(B[(k0 + kg * 8 + j) * 32 + row], h, m, l);
      b[0][j] = h; b[1][j] = m; b[2][j] = l;
    }
    if (terms >= 9) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[2], acc, 0, 0, 0);
    }
    if (terms >= 6) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
    }
    if (terms >= 3) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[r];
}

// same with the fp32 matrix-core instruction (the path libkocr uses today)
typedef float f4v __attribute__((ext_vector_type(4)));
__global__ void k_f32(const float* A, const float* B, float* C, int K) {
  const int lane = threadIdx.x;
  f16v acc = {0};
  for (int k0 = 0; k0 < K; k0 += 2) {
    const float a = A[(lane & 31) * K + k0 + (lane >> 5)], b = B[(k0 + (lane >> 5)) * 32 + (lane & 31)];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[r];
}

// throughput: 8 independent accumulators, back to back
__global__ void k_rate(float* out, int iters) {
  bf8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = 0x3f80 + threadIdx.x; b[j] = 0x3f80; }
  f16v acc[8] = {};
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
  float s = 0;
  for (int t = 0; t < 8; ++t) s += acc[t][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  const int K = 4608;
  std::vector<float> A(32 * K), B(K * 32), C(1024);
  srand(1);
  auto rnd = []() { float s = 0; for (int i = 0; i < 12; ++i) s += rand() / (float)RAND_MAX; return s - 6.f; };
  for (auto& v : A) v = fmaxf(rnd(), 0.f);
  for (auto& v : B) v = 0.02f * rnd();
  std::vector<double> T(1024, 0.0), S(1024, 0.0);
  std::vector<float> F(1024, 0.f);
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    double t = 0, s = 0; float f = 0;
    for (int k = 0; k < K; ++k) { t += (double)A[i * K + k] * B[k * 32 + j]; s += fabs((double)A[i * K + k] * B[k * 32 + j]); f = fmaf(A[i * K + k], B[k * 32 + j], f); }
    T[i * 32 + j] = t; S[i * 32 + j] = s; F[i * 32 + j] = f;
  }
  float *dA, *dB, *dC;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  auto report = [&](const char* nm, const float* c) {
    double mx = 0, bias = 0, rms = 0;
    for (int i = 0; i < 1024; ++i) { const double e = (c[i] - T[i]) / S[i]; mx = fmax(mx, fabs(e)); bias += e; rms += e * e; }
    printf("%-22s max|err|/sum|ab| = %.3e   mean signed = %+.3e   rms = %.3e\n", nm, mx, bias / 1024, sqrt(rms / 1024));
  };
  report("host fmaf chain", F.data());
  hipLaunchKernelGGL(k_f32, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
  hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost); report("mfma f32 32x32x2", C.data());
  for (int terms : {1, 3, 6, 9}) {
    hipLaunchKernelGGL(k_split, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, terms);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    char nm[32]; snprintf(nm, sizeof nm, "bf16 split x%d", terms); report(nm, C.data());
  }
  // integer layout check (K = 16, so A is stored with stride 16): A[i][k] = i + 3k, B[k][j] = 2k - j: asymmetric
  std::vector<float> A2(32 * 16), B2(16 * 32);
  for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A2[i * 16 + k] = (float)(i + 3 * k);
  for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) B2[k * 32 + j] = (float)(2 * k - j);
  hipMemcpy(dA, A2.data(), A2.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B2.data(), B2.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_split, dim3(1), dim3(64), 0, 0, dA, dB, dC, 16, 1);
  hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float t = 0; for (int k = 0; k < 16; ++k) t += (i + 3 * k) * (2 * k - j); if (t != C[i * 32 + j]) ++bad; }
  printf("layout check (A row=lane&31,k=(lane>>5)*8+j; B col=lane&31): %d mismatches\n", bad);
  // rate
  float* dO; hipMalloc(&dO, 1024 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  hipLaunchKernelGGL(k_rate, dim3(1024), dim3(256), 0, 0, dO, 10);
  hipEventRecord(e0); hipLaunchKernelGGL(k_rate, dim3(1024), dim3(256), 0, 0, dO, iters); hipEventRecord(e1);
  hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("32x32x16 bf16 rate: %.1f TFLOP/s\n", 1024.0 * 4 * iters * 8 * 32768.0 / (ms * 1e-3) / 1e12);
  return 0;
}
