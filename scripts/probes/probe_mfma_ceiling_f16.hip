// probe_mfma_ceiling_f16.hip — developer probe (not part of libkocr), round 4.  The fp16 sibling of probe_mfma_ceiling.hip:
// what does an MI355X sustain on v_mfma_f32_32x32x16_f16 with the operand statistics of an F(4,3) convolution in the
// fp16x2 split (h = rn_fp16(V 2^e), l = rn_fp16(V 2^e - h); products a_l b_h, a_h b_l, a_h b_h) or with one fp16 piece
// (fast mode: a_h b_h only), as a function of
//   * the wave tile: MT M-tiles x NT N-tiles x PTS points (192 accumulators each): 2x1x6 (conv_w43v's tile: every weight
//     fragment serves two M-tiles), 2x2x3 (each fragment serves two tiles on both sides; needs a cross-wave output
//     transform), 4x1x3,
//   * the operand delivery (NP MT ds_read_b128 of A + NP NT 16-byte weight loads of B per point, against PR MT NT MFMAs,
//     PR = 3 products for two pieces, 1 for one piece),
//   * FILL plain VALU operations per MFMA (the transform / split work that hides in the MFMA shadows) and NP ds_write_b64
//     per point.
// Each line: TFLOP/s of fp16 MFMA work issued, its fraction of the 2 500 TF dense peak, the shader clock.  Algorithmic rate
// of a convolution built on the variant = issued / 1.5 (F(4,3) x 3 products) or / 0.5 (one product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <cstdint>
typedef _Float16 hf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));

template <int MT, int NT, int PTS, int NP, int LOADS, int FILL, int STORES>
__global__ __launch_bounds__(256) void k_probe(const unsigned short* __restrict__ Asrc, const unsigned short* __restrict__ Wsrc,
                                               float* out, unsigned long long* clk, int ksteps, int wsteps, int a_ushorts) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  constexpr int PR = NP == 2 ? 3 : 1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int PLANE = MT * 512;  // ushorts of one (point, piece) plane: MT M-tiles x 2 k halves x 32 rows x 8
  for (int i = tid * 8; i < PTS * NP * PLANE; i += 256 * 8)
    *reinterpret_cast<hf8*>(lds + i) = *reinterpret_cast<const hf8*>(Asrc + (i % a_ushorts));
  __syncthreads();
  const int l31 = lane & 31, l5 = lane >> 5;
  const unsigned short* a_base = lds + l5 * 256 + ((l31 * 8) ^ (l5 * 32));
  unsigned short* st_base = lds + PTS * NP * PLANE + tid * 4;
  const size_t w_step = (size_t)4 * PTS * NP * NT * 512;
  const unsigned short* w_base = Wsrc + ((size_t)wave * PTS * NP * NT * 64 + lane) * 8;
  f16v acc[PTS][MT][NT];
#pragma unroll
  for (int p = 0; p < PTS; ++p)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][m][n][r] = 0.f;
  hf8 a[2][MT][NP], b[PTS][NT][NP];
  float fl[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) fl[i] = 1.f + 0.001f * (lane + i);
  auto load_a = [&](hf8 (&d)[MT][NP], int p) __attribute__((always_inline)) {
#pragma unroll
    for (int s = NP - 1; s >= 0; --s)
#pragma unroll
      for (int m = 0; m < MT; ++m) d[m][s] = *reinterpret_cast<const hf8*>(a_base + (p * NP + s) * PLANE + m * 512);
  };
  auto load_b = [&](hf8 (&d)[NT][NP], const unsigned short* w, int p) __attribute__((always_inline)) {
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int s = 0; s < NP; ++s) d[n][s] = *reinterpret_cast<const hf8*>(w + (size_t)((p * NT + n) * NP + s) * 512);
  };
  load_a(a[0], 0);
  load_a(a[1], PTS > 1 ? 1 : 0);
#pragma unroll
  for (int d = 0; d < PTS; ++d) load_b(b[d], w_base, d);
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  int ws = 0;
  for (int k = 0; k < ksteps; ++k) {
    const int wn = ws + 1 == wsteps ? 0 : ws + 1;
    const unsigned short* w_nxt = w_base + (size_t)wn * w_step;
    ws = wn;
#pragma unroll
    for (int p = 0; p < PTS; ++p) {
      __builtin_amdgcn_sched_barrier(0);
      hf8(&cur)[MT][NP] = a[p & 1];
      hf8(&nxt)[MT][NP] = a[(p + 1) & 1];
      if (LOADS == 1 || LOADS == 2) load_a(nxt, (p + 1) % PTS);
      if (FILL) {
#pragma unroll
        for (int i = 0; i < FILL * PR * MT * NT; ++i) fl[i & 7] = __builtin_fmaf(fl[i & 7], 1.0000001f, 1e-9f);
      }
      hf8(&bb)[NT][NP] = b[p];
      // smallest terms first; tiles alternate so that consecutive MFMAs are independent
      if constexpr (NP == 2) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[p][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[m][1], bb[n][0], acc[p][m][n], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[p][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[m][0], bb[n][1], acc[p][m][n], 0, 0, 0);
      }
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[p][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[m][0], bb[n][0], acc[p][m][n], 0, 0, 0);
      if (STORES) {
        const u2v v = u2v{__float_as_uint(fl[0]), __float_as_uint(fl[1])};
#pragma unroll
        for (int s = 0; s < NP; ++s) *reinterpret_cast<u2v*>(st_base + s * 256 * 4) = v;
      }
      if (LOADS == 1 || LOADS == 2) __builtin_amdgcn_sched_group_barrier(0x100, NP * MT, 0);
      if (FILL) {
#pragma unroll
        for (int i = 0; i < PR * MT * NT - 1; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, FILL, 0);
        }
        if (STORES) __builtin_amdgcn_sched_group_barrier(0x200, NP, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (LOADS == 1 || LOADS == 3) load_b(b[p], w_nxt, p);  // this point's weights of the next step, a full step ahead
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = fl[0] + fl[1] + fl[2] + fl[3] + fl[4] + fl[5] + fl[6] + fl[7];
#pragma unroll
  for (int p = 0; p < PTS; ++p)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[p][m][n][r];
  out[(size_t)blockIdx.x * 256 + tid] = s;
  if (tid == 0) {
    clk[2 * blockIdx.x] = t1 - t0;
    clk[2 * blockIdx.x + 1] = r1 - r0;
  }
}

static unsigned short f2h(float v) {
  _Float16 h = (_Float16)v;
  unsigned short u;
  memcpy(&u, &h, 2);
  return u;
}
static float h2f(unsigned short u) {
  _Float16 h;
  memcpy(&h, &u, 2);
  return (float)h;
}
static void split_h(float v, unsigned short o[2]) {
  o[0] = f2h(v);
  o[1] = f2h(v - h2f(o[0]));
}

struct Pool {
  std::vector<unsigned short> act;  // [value][2]
  std::vector<unsigned short> W;    // [value][2]
  unsigned short *dA, *dW, *dWz;
  float* dO;
  unsigned long long* dC;
};

template <int MT, int NT, int PTS, int NP, int LOADS, int FILL, int STORES>
static void run(const char* what, const Pool& b, int ksteps, bool real) {
  constexpr int PR = NP == 2 ? 3 : 1;
  constexpr int PLANE = MT * 512;
  const int lds = PTS * NP * PLANE * 2 + NP * 256 * 8;
  const int img_n = PTS * NP * PLANE;
  std::vector<unsigned short> img(img_n, 0);
  const size_t nv = b.act.size() / 2;
  if (real)
    for (int p = 0; p < PTS; ++p)
      for (int s = 0; s < NP; ++s)
        for (int i = 0; i < PLANE; ++i) img[(p * NP + s) * PLANE + i] = b.act[(((size_t)p * PLANE + i) % nv) * 2 + s];
  hipMemcpy(b.dA, img.data(), img_n * 2, hipMemcpyHostToDevice);
  // weights: planes of 512 values, plane index ... ((p * NT + n) * NP + s); about 6.9 MB per wsteps like the bf16 probe
  const size_t per_step = (size_t)4 * PTS * NP * NT * 512;
  const int wsteps = (int)((size_t)(3456 * 1024) / per_step);  // ushorts: 6.9 MB
  std::vector<unsigned short> W(per_step * wsteps, 0);
  const size_t nw = b.W.size() / 2;
  if (real)
    for (size_t pl = 0; pl < W.size() / 512; ++pl)
      for (int j = 0; j < 512; ++j) W[pl * 512 + j] = b.W[(((pl / NP) * 512 + j) % nw) * 2 + (pl % NP)];
  hipMemcpy(b.dW, W.data(), W.size() * 2, hipMemcpyHostToDevice);
  auto kern = k_probe<MT, NT, PTS, NP, LOADS, FILL, STORES>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int grid = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, b.dA, b.dW, b.dO, b.dC, 64, wsteps, img_n);
  hipDeviceSynchronize();
  float best = 1e30f;
  std::vector<unsigned long long> clk(2 * grid);
  double ghz = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, b.dA, b.dW, b.dO, b.dC, ksteps, wsteps, img_n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) {
      best = ms;
      hipMemcpy(clk.data(), b.dC, clk.size() * 8, hipMemcpyDeviceToHost);
      double c = 0, r = 0;
      for (int i = 0; i < grid; ++i) {
        c += (double)clk[2 * i];
        r += (double)clk[2 * i + 1];
      }
      ghz = c / r * 0.1;
    }
  }
  const hipError_t err = hipGetLastError();
  const double flop = (double)grid * 4 * ksteps * PTS * MT * NT * PR * 32768.0;
  const double tf = flop / (best * 1e-3) / 1e12;
  printf("%-66s data=%-4s %8.3f ms  %7.1f TF/s issued  %.3f of 2500  clk %.2f GHz  pipe busy %.3f  -> %6.1f TF/s algorithmic%s\n", what,
         real ? "real" : "zero", best, tf, tf / 2500.0, ghz, tf / (256.0 * 4 * 1024 * ghz * 1e9 / 1e12), tf / (NP == 2 ? 1.5 : 0.5),
         err == hipSuccess ? "" : "  [HIP ERROR]");
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int ksteps = argc > 1 ? atoi(argv[1]) : 6000;
  const int NV = 24 * 1024;
  Pool pool;
  pool.act.resize((size_t)NV * 2);
  pool.W.resize((size_t)NV * 2);
  srand(7);
  auto rnd = []() {
    float s = 0;
    for (int i = 0; i < 12; ++i) s += rand() / (float)RAND_MAX;
    return s - 6.f;
  };
  const float pa = 0.625f, pb = 1.5f, a2 = pa * pa, b2 = pb * pb;
  // activations: x = relu(N(0,1)), max ~ 5 -> scaled by 2^10 (max |x| 2^e < 2^13), transformed, split to nearest
  for (int v = 0; v < NV; ++v) {
    float d[6];
    for (int i = 0; i < 6; ++i) d[i] = fmaxf(rnd(), 0.f) * 1024.f;
    float V;
    switch (v % 6) {
      case 0: V = (a2 * b2 * d[0] - (a2 + b2) * d[2]) + d[4]; break;
      case 1: V = (d[4] - b2 * d[2]) + pa * (d[3] - b2 * d[1]); break;
      case 2: V = (d[4] - b2 * d[2]) - pa * (d[3] - b2 * d[1]); break;
      case 3: V = (d[4] - a2 * d[2]) + pb * (d[3] - a2 * d[1]); break;
      case 4: V = (d[4] - a2 * d[2]) - pb * (d[3] - a2 * d[1]); break;
      default: V = (a2 * b2 * d[1] - (a2 + b2) * d[3]) + d[5]; break;
    }
    split_h(V, &pool.act[(size_t)v * 2]);
  }
  // weights: g = 0.02 N(0,1) -> G g, scaled by 2^17 (max |U| 2^e in [2^14, 2^15))
  for (int v = 0; v < NV; ++v) {
    const float t[3] = {0.02f * rnd(), 0.02f * rnd(), 0.02f * rnd()};
    const float na = 2.f * a2 * (a2 - b2);
    const float U = (v % 3 == 0) ? t[0] / (a2 * b2) : (v % 3 == 1) ? (t[0] + pa * t[1] + a2 * t[2]) / na : t[2];
    split_h(U * 131072.f, &pool.W[(size_t)v * 2]);
  }
  hipMalloc(&pool.dO, 256 * 256 * 4);
  hipMalloc(&pool.dC, 2 * 256 * 8);
  hipMalloc(&pool.dA, 1 << 20);
  hipMalloc(&pool.dW, 16 << 20);
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  printf("# %s, %d CUs; %d K-steps per launch, grid 256 blocks of 256 threads (one wave per SIMD); issued fp16 MFMA FLOPs / time\n", prop.gcnArchName,
         prop.multiProcessorCount, ksteps);
  const Pool& P = pool;
  // ---- fp16x2, conv_w43v's wave tile: 2 M-tiles x 1 N-tile x 6 points: 4 ds_read + 2 weight loads per 6 MFMAs ------------
  run<2, 1, 6, 2, 0, 0, 0>("f16x2 2Mx1Nx6pt registers only", P, ksteps, true);
  run<2, 1, 6, 2, 0, 0, 0>("f16x2 2Mx1Nx6pt registers only", P, ksteps, false);
  run<2, 1, 6, 2, 2, 0, 0>("  + A from LDS (4 ds_read_b128 / 6 MFMA)", P, ksteps, true);
  run<2, 1, 6, 2, 3, 0, 0>("  + B from L2 (2 loads / 6 MFMA, one step ahead)", P, ksteps, true);
  run<2, 1, 6, 2, 1, 0, 0>("  + both", P, ksteps, true);
  run<2, 1, 6, 2, 1, 0, 0>("  + both", P, ksteps, false);
  run<2, 1, 6, 2, 1, 2, 1>("  + both + 2 VALU / MFMA + 2 ds_write_b64 / point", P, ksteps, true);
  run<2, 1, 6, 2, 1, 3, 1>("  + both + 3 VALU / MFMA + 2 ds_write_b64 / point", P, ksteps, true);
  run<2, 1, 6, 2, 1, 4, 1>("  + both + 4 VALU / MFMA + 2 ds_write_b64 / point", P, ksteps, true);
  run<2, 1, 6, 2, 1, 5, 1>("  + both + 5 VALU / MFMA + 2 ds_write_b64 / point", P, ksteps, true);
  run<2, 1, 6, 2, 1, 6, 1>("  + both + 6 VALU / MFMA + 2 ds_write_b64 / point", P, ksteps, true);
  run<2, 1, 6, 2, 0, 3, 0>("  registers only + 3 VALU / MFMA", P, ksteps, true);
  // ---- fp16x2, 2 M-tiles x 2 N-tiles x 3 points: 4 ds_read + 4 weight loads per 12 MFMAs ----------------------------------
  run<2, 2, 3, 2, 0, 0, 0>("f16x2 2Mx2Nx3pt registers only", P, ksteps, true);
  run<2, 2, 3, 2, 1, 0, 0>("  + A from LDS + B from L2 (4 + 4 / 12 MFMA)", P, ksteps, true);
  run<2, 2, 3, 2, 1, 2, 1>("  + both + 2 VALU / MFMA + stores", P, ksteps, true);
  run<2, 2, 3, 2, 1, 3, 1>("  + both + 3 VALU / MFMA + stores", P, ksteps, true);
  run<2, 2, 3, 2, 1, 4, 1>("  + both + 4 VALU / MFMA + stores", P, ksteps, true);
  // ---- fp16x2, 4 M-tiles x 1 N-tile x 3 points: 8 ds_read + 2 weight loads per 12 MFMAs -----------------------------------
  run<4, 1, 3, 2, 1, 0, 0>("f16x2 4Mx1Nx3pt A from LDS + B from L2 (8 + 2 / 12 MFMA)", P, ksteps, true);
  run<4, 1, 3, 2, 1, 3, 1>("  + both + 3 VALU / MFMA + stores", P, ksteps, true);
  // ---- one fp16 piece (fast mode): 1 product --------------------------------------------------------------------------------
  run<2, 1, 6, 1, 0, 0, 0>("f16x1 2Mx1Nx6pt registers only", P, ksteps, true);
  run<2, 1, 6, 1, 1, 0, 0>("  + A from LDS + B from L2 (2 + 1 / 2 MFMA)", P, ksteps, true);
  run<2, 1, 6, 1, 1, 4, 1>("  + both + 4 VALU / MFMA + 1 ds_write_b64 / point", P, ksteps, true);
  run<2, 1, 6, 1, 1, 8, 1>("  + both + 8 VALU / MFMA + 1 ds_write_b64 / point", P, ksteps, true);
  run<2, 2, 3, 1, 1, 0, 0>("f16x1 2Mx2Nx3pt A from LDS + B from L2 (2 + 2 / 4 MFMA)", P, ksteps, true);
  run<2, 2, 3, 1, 1, 4, 1>("  + both + 4 VALU / MFMA + stores", P, ksteps, true);
  run<2, 2, 3, 1, 1, 8, 1>("  + both + 8 VALU / MFMA + stores", P, ksteps, true);
  return 0;
}
