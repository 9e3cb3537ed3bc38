// Micro-probe: what does this chip sustain on v_mfma_f32_32x32x2_f32 under the conv kernel's
// inner-loop shapes?  (developer tool, not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>  // 0: registers only, 1: ds_read2 pairs per 4 mfma (conv pattern), 2: as 1 but fragments double-buffered
__global__ __launch_bounds__(256, 2) void probe(float* out, int iters) {
  __shared__ float As[2][16][132];
  __shared__ float Bs[2][16][132];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lk = lane >> 5, wm = wave >> 1, wn = wave & 1;
  for (int i = tid; i < 2 * 16 * 132; i += 256) { (&As[0][0][0])[i] = 0.001f * (i % 7); (&Bs[0][0][0])[i] = 0.002f * (i % 5); }
  __syncthreads();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float a0 = lane * 0.01f, b0 = lane * 0.02f;
  for (int it = 0; it < iters; ++it) {
    const int buf = it & 1;
    if (MODE == 0) {
#pragma unroll
      for (int kp = 0; kp < 8; ++kp)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int kp = 0; kp < 8; ++kp) {
        float a[2], b[2];
        for (int i = 0; i < 2; ++i) a[i] = As[buf][2 * kp + lk][wm * 64 + i * 32 + lr];
        for (int j = 0; j < 2; ++j) b[j] = Bs[buf][2 * kp + lk][wn * 64 + j * 32 + lr];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      if (MODE == 2) __syncthreads();
    }
  }
  float s = 0;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
void run(const char* name, int blocks, int iters) {
  float* d; hipMalloc(&d, (size_t)blocks * 256 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double flops = (double)blocks * 4 /*waves*/ * iters * 32.0 * 2 * 32 * 32 * 2;
  printf("%-28s blocks=%5d iters=%d  %.3f ms  %.1f TF/s\n", name, blocks, iters, ms, flops / ms / 1e9);
  hipFree(d);
}
int main() {
  for (int bpc : {1, 2, 4}) {
    int blocks = 256 * bpc;
    printf("--- %d block(s) per CU\n", bpc);
    run<0>("mfma registers only", blocks, 4000 / bpc);
    run<1>("mfma + ds_read2 (conv loop)", blocks, 4000 / bpc);
    run<2>("  + barrier per K-step", blocks, 4000 / bpc);
  }
  return 0;
}
