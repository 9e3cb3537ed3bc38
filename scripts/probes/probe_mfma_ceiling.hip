// probe_mfma_ceiling.hip — developer probe (not part of libkocr).  What does an MI355X sustain on
// v_mfma_f32_32x32x16_bf16 with the OPERAND STATISTICS of the bf16x3 split convolutions (Winograd-transformed post-ReLU
// activations split by truncation, transformed weights split by round-to-nearest), separated from the convolution
// kernels themselves:
//   * one wave per SIMD (256 threads, 192 or 384 accumulator registers) against two waves per SIMD (512 threads, 96),
//   * registers-only operands against the kernels' operand delivery (ds_read_b128 of A fragments from a conflict-free
//     LDS image + 16-byte weight loads from an L2-resident array, rolling ahead as the kernels do),
//   * with and without the transform / split VALU work (FILL plain VALU operations per MFMA, interleaved with
//     sched_group_barrier as in conv_w43.hip) and the LDS stores of the produced operands,
//   * the F(4,3) wave tile (2 M-tiles x 6 points: 6 ds_read + 3 weight loads per 12 MFMAs) against the wave tile a nested
//     F(2x4, 3x3) kernel would have (1 M-tile x 24 points: 3 ds_read + 3 weight loads per 6 MFMAs),
//   * real operands against zeros (the DVFS / power reference).
// Each line: TFLOP/s of bf16 MFMA work issued, its fraction of the 2 500 TF dense peak, and the shader clock measured
// inside the kernel (s_memtime ticks per s_memrealtime tick x 100 MHz).  The matrix-pipe busy counter of the same
// variants comes from a separate rocprofv3 --pmc pass (scripts/probe_ceiling.sh).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <cstdint>
#include <string>
typedef short bf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));

// MT M-tiles per point, PTS points per wave and K-step, BD = depth of the weight ring in points (weights are fetched BD
// points ahead), LOADS: 0 registers only, 1 A from LDS + B from L2, 2 A only, 3 B only; FILL VALU per MFMA; STORES: three
// ds_write_b64 per point; NTHR threads per block.
template <int MT, int PTS, int BD, int LOADS, int FILL, int STORES, int NTHR>
__global__ __launch_bounds__(NTHR) void k_probe(const unsigned short* __restrict__ Asrc, const unsigned short* __restrict__ Wsrc,
                                                float* out, unsigned long long* clk, int ksteps, int wsteps, int a_ushorts) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NW = NTHR / 64;
  constexpr int PLANE = MT * 512;  // ushorts of one (point, piece) plane: MT M-tiles x 2 k halves x 32 rows x 8
  constexpr int LPTS = NTHR == 512 ? 2 * PTS : PTS;  // points resident in LDS (two-wave arrangement: the partner's too)
  for (int i = tid * 8; i < LPTS * 3 * PLANE; i += NTHR * 8)
    *reinterpret_cast<bf8*>(lds + i) = *reinterpret_cast<const bf8*>(Asrc + (i % a_ushorts));
  __syncthreads();
  const int l31 = lane & 31, l5 = lane >> 5;
  const int pbase = NTHR == 512 ? (wave >> 2) * PTS : 0;  // two waves per SIMD: waves 4-7 own the other points
  const unsigned short* a_base = lds + pbase * 3 * PLANE + l5 * 256 + ((l31 * 8) ^ (l5 * 32));
  unsigned short* st_base = lds + LPTS * 3 * PLANE + tid * 4;  // store sink behind the operand image
  const size_t w_step = (size_t)NW * PTS * 3 * 512;
  const unsigned short* w_base = Wsrc + ((size_t)wave * PTS * 3 * 64 + lane) * 8;
  f16v acc[PTS][MT];
#pragma unroll
  for (int p = 0; p < PTS; ++p)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][m][r] = 0.f;
  bf8 a[2][MT][3], b[BD][3];
  float fl[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) fl[i] = 1.f + 0.001f * (lane + i);
  auto load_a = [&](bf8 (&d)[MT][3], int p) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 2; s >= 0; --s)
#pragma unroll
      for (int m = 0; m < MT; ++m) d[m][s] = *reinterpret_cast<const bf8*>(a_base + (p * 3 + s) * PLANE + m * 512);
  };
  auto load_b = [&](bf8 (&d)[3], const unsigned short* w, int p) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < 3; ++s) d[s] = *reinterpret_cast<const bf8*>(w + (size_t)(p * 3 + s) * 512);
  };
  // initial fragments (register-only variants keep them for the whole run)
  load_a(a[0], 0);
  load_a(a[1], PTS > 1 ? 1 : 0);
#pragma unroll
  for (int d = 0; d < BD; ++d) load_b(b[d], w_base, d % PTS);
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  int ws = 0;
  for (int k = 0; k < ksteps; ++k) {
    const unsigned short* w_cur = w_base + (size_t)ws * w_step;
    const int wn = ws + 1 == wsteps ? 0 : ws + 1;
    const unsigned short* w_nxt = w_base + (size_t)wn * w_step;
    ws = wn;
#pragma unroll
    for (int p = 0; p < PTS; ++p) {
      __builtin_amdgcn_sched_barrier(0);
      bf8(&cur)[MT][3] = a[p & 1];
      bf8(&nxt)[MT][3] = a[(p + 1) & 1];
      if (LOADS == 1 || LOADS == 2) load_a(nxt, (p + 1) % PTS);
      if (FILL) {
#pragma unroll
        for (int i = 0; i < FILL * 6 * MT; ++i) fl[i & 7] = __builtin_fmaf(fl[i & 7], 1.0000001f, 1e-9f);
      }
      bf8(&bb)[3] = b[p % BD];
      // the kernels' product order: smallest terms first, M-tiles alternating
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[p][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[m][2], bb[0], acc[p][m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[p][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[m][0], bb[2], acc[p][m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[p][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[m][1], bb[1], acc[p][m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[p][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[m][1], bb[0], acc[p][m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[p][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[m][0], bb[1], acc[p][m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[p][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[m][0], bb[0], acc[p][m], 0, 0, 0);
      if (STORES) {
        const u2v v = u2v{__float_as_uint(fl[0]), __float_as_uint(fl[1])};
#pragma unroll
        for (int s = 0; s < 3; ++s) *reinterpret_cast<u2v*>(st_base + s * NTHR * 4) = v;
      }
      if (LOADS == 1 || LOADS == 2) __builtin_amdgcn_sched_group_barrier(0x100, 3 * MT, 0);
      if (FILL) {
#pragma unroll
        for (int i = 0; i < 6 * MT - 1; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, FILL, 0);
        }
        if (STORES) __builtin_amdgcn_sched_group_barrier(0x200, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      // this ring slot's weights, BD points ahead
      if (LOADS == 1 || LOADS == 3) {
        const int pn = p + BD;
        load_b(b[p % BD], pn < PTS ? w_cur : w_nxt, pn % PTS);
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = fl[0] + fl[1] + fl[2] + fl[3] + fl[4] + fl[5] + fl[6] + fl[7];
#pragma unroll
  for (int p = 0; p < PTS; ++p)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[p][m][r];
  out[(size_t)blockIdx.x * NTHR + tid] = s;
  if (tid == 0) {
    clk[2 * blockIdx.x] = t1 - t0;
    clk[2 * blockIdx.x + 1] = r1 - r0;
  }
}

static void split_trunc(float v, unsigned short o[3]) {
  float r = v;
  for (int s = 0; s < 3; ++s) {
    uint32_t u;
    memcpy(&u, &r, 4);
    u &= 0xFFFF0000u;
    float h;
    memcpy(&h, &u, 4);
    o[s] = (unsigned short)(u >> 16);
    r -= h;
  }
  uint32_t u;
  memcpy(&u, &r, 4);  // unreachable remainder is zero after three truncations of a 24-bit significand
  (void)u;
}
static void split_rne(float v, unsigned short o[3]) {
  float r = v;
  for (int s = 0; s < 3; ++s) {
    uint32_t u;
    memcpy(&u, &r, 4);
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
    float h;
    memcpy(&h, &u, 4);
    o[s] = (unsigned short)(u >> 16);
    r -= h;
  }
}

struct Pool {
  std::vector<unsigned short> act;  // [value][3] split activations
  std::vector<unsigned short> W;    // weight array in plane order (512-value planes, piece = plane % 3)
  unsigned short *dA, *dW, *dWz;
  float* dO;
  unsigned long long* dC;
};

template <int MT, int PTS, int BD, int LOADS, int FILL, int STORES, int NTHR>
static void run(const char* what, const Pool& b, int wsteps, int ksteps, bool real) {
  constexpr int PLANE = MT * 512;
  constexpr int LPTS = NTHR == 512 ? 2 * PTS : PTS;
  const int lds = LPTS * 3 * PLANE * 2 + 3 * NTHR * 8;
  // LDS operand image: plane (p, s) holds piece s of PLANE transformed values
  const int img_n = LPTS * 3 * PLANE;
  std::vector<unsigned short> img(img_n, 0);
  if (real) {
    const size_t nv = b.act.size() / 3;
    for (int p = 0; p < LPTS; ++p)
      for (int s = 0; s < 3; ++s)
        for (int i = 0; i < PLANE; ++i) img[(p * 3 + s) * PLANE + i] = b.act[(((size_t)p * PLANE + i) % nv) * 3 + s];
  }
  hipMemcpy(b.dA, img.data(), img_n * 2, hipMemcpyHostToDevice);
  const unsigned short* dW = real ? b.dW : b.dWz;
  auto kern = k_probe<MT, PTS, BD, LOADS, FILL, STORES, NTHR>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int grid = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHR), lds, 0, b.dA, dW, b.dO, b.dC, 64, wsteps, img_n);
  hipDeviceSynchronize();
  float best = 1e30f;
  std::vector<unsigned long long> clk(2 * grid);
  double ghz = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHR), lds, 0, b.dA, dW, b.dO, b.dC, ksteps, wsteps, img_n);
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
  const double flop = (double)grid * (NTHR / 64) * ksteps * PTS * MT * 6 * 32768.0;
  const double tf = flop / (best * 1e-3) / 1e12;
  printf("%-60s data=%-4s %8.3f ms  %7.1f TF/s issued  %.3f of 2500  clk %.2f GHz  pipe busy (TF / clk) %.3f%s\n", what,
         real ? "real" : "zero", best, tf, tf / 2500.0, ghz, tf / (256.0 * 4 * 1024 * ghz * 1e9 / 1e12), err == hipSuccess ? "" : "  [HIP ERROR]");
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int ksteps = argc > 1 ? atoi(argv[1]) : 6000;
  // operands: x = relu(N(0,1)) pixels of 6-wide row pieces -> F(4,3) input transform (points 0, +-5/8, +-3/2, inf) -> split by
  // truncation; weights g = 0.02 N(0,1) -> G g -> split to nearest (the operand ORDER is irrelevant for the probe, the value
  // statistics are not).
  const int NV = 24 * 1024;      // transformed activation values in the pool
  const int NWS = 96;            // weight steps of the 6-point layout (4 waves x 18 KB per step: 6.9 MB in all)
  const size_t NWU = (size_t)NWS * 4 * 6 * 3 * 512 * 2;  // ushorts; also covers the 24-point layout and 8 waves
  Pool pool;
  pool.act.resize((size_t)NV * 3);
  pool.W.resize(NWU);
  srand(7);
  auto rnd = []() {
    float s = 0;
    for (int i = 0; i < 12; ++i) s += rand() / (float)RAND_MAX;
    return s - 6.f;
  };
  const float pa = 0.625f, pb = 1.5f, a2 = pa * pa, b2 = pb * pb;
  for (int v = 0; v < NV; ++v) {
    float d[6];
    for (int i = 0; i < 6; ++i) d[i] = fmaxf(rnd(), 0.f);
    float V;
    switch (v % 6) {
      case 0: V = (a2 * b2 * d[0] - (a2 + b2) * d[2]) + d[4]; break;
      case 1: V = (d[4] - b2 * d[2]) + pa * (d[3] - b2 * d[1]); break;
      case 2: V = (d[4] - b2 * d[2]) - pa * (d[3] - b2 * d[1]); break;
      case 3: V = (d[4] - a2 * d[2]) + pb * (d[3] - a2 * d[1]); break;
      case 4: V = (d[4] - a2 * d[2]) - pb * (d[3] - a2 * d[1]); break;
      default: V = (a2 * b2 * d[1] - (a2 + b2) * d[3]) + d[5]; break;
    }
    split_trunc(V, &pool.act[(size_t)v * 3]);
  }
  for (size_t g = 0; g < NWU / 1536; ++g)
    for (int j = 0; j < 512; ++j) {
      const float t[3] = {0.02f * rnd(), 0.02f * rnd(), 0.02f * rnd()};
      const float na = 2.f * a2 * (a2 - b2);
      const float U = (j % 3 == 0) ? t[0] / (a2 * b2) : (j % 3 == 1) ? (t[0] + pa * t[1] + a2 * t[2]) / na : t[2];
      unsigned short o[3];
      split_rne(U, o);
      for (int s = 0; s < 3; ++s) pool.W[(g * 3 + s) * 512 + j] = o[s];
    }
  hipMalloc(&pool.dO, 256 * 512 * 4);
  hipMalloc(&pool.dC, 2 * 256 * 8);
  hipMalloc(&pool.dA, 24 * 3 * 1024 * 2 * 2);
  hipMalloc(&pool.dW, NWU * 2);
  hipMalloc(&pool.dWz, NWU * 2);
  hipMemcpy(pool.dW, pool.W.data(), NWU * 2, hipMemcpyHostToDevice);
  hipMemset(pool.dWz, 0, NWU * 2);
  const Pool& real = pool;
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  printf("# %s, %d CUs; %d K-steps per launch, grid 256 blocks; issued bf16 MFMA FLOPs / time\n", prop.gcnArchName, prop.multiProcessorCount, ksteps);

  // ---- the F(4,3) wave tile: 2 M-tiles x 6 points, one wave per SIMD ------------------------------------------------
  run<2, 6, 6, 0, 0, 0, 256>("w43 tile 2Mx6pt, 1 wave/SIMD, registers only", real, NWS, ksteps, true);
  run<2, 6, 6, 0, 0, 0, 256>("w43 tile 2Mx6pt, 1 wave/SIMD, registers only", real, NWS, ksteps, false);
  run<2, 6, 6, 2, 0, 0, 256>("  + A fragments from LDS (6 ds_read_b128 / 12 MFMA)", real, NWS, ksteps, true);
  run<2, 6, 6, 3, 0, 0, 256>("  + B fragments from L2 (3 loads / 12 MFMA, 1 step ahead)", real, NWS, ksteps, true);
  run<2, 6, 6, 1, 0, 0, 256>("  + both", real, NWS, ksteps, true);
  run<2, 6, 6, 1, 0, 0, 256>("  + both", real, NWS, ksteps, false);
  run<2, 6, 6, 1, 3, 1, 256>("  + both + 3 VALU / MFMA + 3 ds_write_b64 / point", real, NWS, ksteps, true);
  run<2, 6, 6, 1, 5, 1, 256>("  + both + 5 VALU / MFMA + 3 ds_write_b64 / point", real, NWS, ksteps, true);
  run<2, 6, 6, 1, 2, 1, 256>("  + both + 2 VALU / MFMA + 3 ds_write_b64 / point", real, NWS, ksteps, true);
  run<2, 6, 6, 1, 4, 1, 256>("  + both + 4 VALU / MFMA + 3 ds_write_b64 / point", real, NWS, ksteps, true);
  run<2, 6, 6, 0, 3, 0, 256>("  registers only + 3 VALU / MFMA", real, NWS, ksteps, true);
  run<2, 6, 6, 0, 5, 0, 256>("  registers only + 5 VALU / MFMA", real, NWS, ksteps, true);
  // ---- the same work at two waves per SIMD: 2 M-tiles x 3 points per wave ------------------------------------------
  run<2, 3, 3, 0, 0, 0, 512>("2 waves/SIMD 2Mx3pt each, registers only", real, NWS / 2, ksteps, true);
  run<2, 3, 3, 0, 0, 0, 512>("2 waves/SIMD 2Mx3pt each, registers only", real, NWS / 2, ksteps, false);
  run<2, 3, 3, 1, 0, 0, 512>("  + A from LDS + B from L2", real, NWS / 2, ksteps, true);
  run<2, 3, 3, 1, 3, 1, 512>("  + both + 3 VALU / MFMA + stores", real, NWS / 2, ksteps, true);
  run<2, 3, 3, 1, 5, 1, 512>("  + both + 5 VALU / MFMA + stores", real, NWS / 2, ksteps, true);
  // ---- the wave tile a nested F(2x4, 3x3) kernel could have.  1 M-tile x 24 points = 384 accumulators does not compile
  // without spilling (hipcc -O3: 34 spilled VGPRs with register-only operands, 200-350 with any operand delivery), so
  // the feasible tile is 1 M-tile x 12 points (192 accumulators, the 24 points over two waves): 3 ds_read + 3 weight loads
  // per 6 MFMAs, and a block tile of 256 pixels x 64 couts, i.e. the transform amortised over half as many couts
  // (about 6 VALU per MFMA instead of 3) ------------------------------------------------------------------------------
  run<1, 12, 4, 0, 0, 0, 256>("nested tile 1Mx12pt, 1 wave/SIMD, registers only", real, NWS / 2, ksteps, true);
  run<1, 12, 4, 2, 0, 0, 256>("  + A from LDS (3 ds_read / 6 MFMA)", real, NWS / 2, ksteps, true);
  run<1, 12, 4, 3, 0, 0, 256>("  + B from L2 (3 loads / 6 MFMA, 4 points ahead)", real, NWS / 2, ksteps, true);
  run<1, 12, 4, 1, 0, 0, 256>("  + both", real, NWS / 2, ksteps, true);
  run<1, 12, 4, 1, 3, 1, 256>("  + both + 3 VALU / MFMA + stores", real, NWS / 2, ksteps, true);
  run<1, 12, 4, 1, 6, 1, 256>("  + both + 6 VALU / MFMA + stores (64-cout amortisation)", real, NWS / 2, ksteps, true);
  run<1, 12, 4, 1, 8, 1, 256>("  + both + 8 VALU / MFMA + stores", real, NWS / 2, ksteps, true);
  // the same 24 points over two waves per SIMD x 4 SIMDs: 1 M-tile x 6 points (96 accumulators) per wave, 64 couts per block
  run<1, 6, 3, 0, 0, 0, 512>("nested, 2 waves/SIMD 1Mx6pt each, registers only", real, NWS / 2, ksteps, true);
  run<1, 6, 3, 1, 0, 0, 512>("  + A from LDS + B from L2", real, NWS / 2, ksteps, true);
  run<1, 6, 3, 1, 3, 1, 512>("  + both + 3 VALU / MFMA + stores", real, NWS / 2, ksteps, true);
  run<1, 6, 3, 1, 6, 1, 512>("  + both + 6 VALU / MFMA + stores", real, NWS / 2, ksteps, true);
  run<1, 6, 3, 1, 8, 1, 512>("  + both + 8 VALU / MFMA + stores", real, NWS / 2, ksteps, true);
  return 0;
}
