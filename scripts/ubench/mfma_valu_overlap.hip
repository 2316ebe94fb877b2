// Micro-benchmark (diagnostic, not part of the library): does the vector work of a wave hide
// under the shadow of its OWN bf16 MFMAs on gfx950, and how does that change with 1-3 waves per
// SIMD?  Each wave runs REPS rounds of {36 MFMAs (three accumulator chains), NV dependent-chain
// v_fma (four chains)} either clustered (all MFMAs, then all VALU) or interleaved (one MFMA, then
// NV / 36 VALU).  Prints cycles per round per wave.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o /tmp/ub scripts/ubench/mfma_valu_overlap.hip && /tmp/ub
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int VPM>  // MODE 0: clustered, 1: interleaved, 2: MFMA only, 3: VALU only; VPM: VALU per MFMA
__global__ void __launch_bounds__(768) kern(float* out, long long* cycles, int reps) {
  f32x16 acc[3];
  for (int i = 0; i < 3; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = (float)(threadIdx.x + r);
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * e); }
  float v[8] = {1.0f + threadIdx.x, 2.0f, 3.0f, 4.0f, 5.0f, 6.0f, 7.0f, 8.0f};
  const float k = 1.0000001f, c = 1e-7f;
  const long long t0 = clock64();
  for (int rep = 0; rep < reps; ++rep) {
    if (MODE == 0 || MODE == 2 || MODE == 3) {
      if (MODE != 3) {
#pragma unroll
        for (int m = 0; m < 36; ++m) acc[m % 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m % 3], 0, 0, 0);
      }
      if (MODE != 2) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 36 * VPM; ++i) v[i & 7] = __builtin_fmaf(v[i & 7], k, c);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int m = 0; m < 36; ++m) {
        acc[m % 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m % 3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < VPM; ++i) v[i & 7] = __builtin_fmaf(v[i & 7], k, c);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const long long t1 = clock64();
  float s = v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7];
  for (int i = 0; i < 3; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE, int VPM>
double run(int waves_per_simd, int reps) {
  const int threads = 256 * waves_per_simd, blocks = 256;
  float* out; long long* cyc;
  hipMalloc(&out, sizeof(float) * threads * blocks);
  hipMalloc(&cyc, sizeof(long long) * blocks);
  kern<MODE, VPM><<<blocks, threads>>>(out, cyc, reps);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  kern<MODE, VPM><<<blocks, threads>>>(out, cyc, reps);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks);
  hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  double mean = 0; for (auto x : h) mean += x; mean /= blocks;
  hipFree(out); hipFree(cyc);
  // clock64 counts at 100 MHz on this part; report wall microseconds per round instead
  return 1e3 * ms / reps;
}

int main() {
  const int reps = 2000;
  printf("us per round (36 MFMA 32x32x16 bf16 + 36*VPM v_fma) per workgroup, all CUs busy\n");
  printf("%-22s %8s %8s %8s\n", "variant", "1w/SIMD", "2w/SIMD", "3w/SIMD");
#define ROW(name, MODE, VPM) printf("%-22s %8.3f %8.3f %8.3f\n", name, run<MODE, VPM>(1, reps), run<MODE, VPM>(2, reps), run<MODE, VPM>(3, reps));
  ROW("mfma only", 2, 10)
  ROW("valu only x10", 3, 10)
  ROW("clustered x10", 0, 10)
  ROW("interleaved x10", 1, 10)
  ROW("valu only x5", 3, 5)
  ROW("clustered x5", 0, 5)
  ROW("interleaved x5", 1, 5)
  return 0;
}
