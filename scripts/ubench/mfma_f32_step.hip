// Micro-benchmark (diagnostic, not part of the library): what ONE step of the training kernels' layers costs a wave -
// 16 B operands from global memory (one float per lane each, rows of a [128][128] matrix), 16 A operands from LDS
// (odd row stride), a relu on them, 16 v_mfma_f32_32x32x2_f32 in two accumulator chains - by ingredient, with one and
// with two waves per SIMD and 32 workgroups on the chip (the geometry of a 1000-row minibatch).  Prints shader cycles
// per step (s_memtime of wave 0 of workgroup 0 over STEPS steps).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/mfma_f32_step scripts/ubench/mfma_f32_step.hip && scripts/ubench/mfma_f32_step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define STEPS 64
#define LD 129

// what a step contains
#define W_GLOBAL 1   // B operands loaded from global memory (else: constants)
#define W_AHEAD 2    // ... requested one step ahead, behind a scheduling barrier
#define W_LDS 4      // A operands read from LDS (else: constants)
#define W_RELU 8
#define W_MFMA 16
#define W_LDS_AHEAD 32  // A operands of the next step read before this step's MFMAs
#define W_GROUPS 64     // ... and the scheduler told to put them BETWEEN the MFMAs: 1 MFMA, 1 load, 1 LDS read, 3 VALU, 16 times

template <int WHAT>
__global__ void __launch_bounds__(512) step_kernel(const float* __restrict__ W, float* __restrict__ out, long long* cycles,
                                                   int waves_active) {
  extern __shared__ float sm[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, j = lane & 31;
  for (int i = tid; i < 32 * LD; i += 512) sm[i] = 0.001f * (float)(i % 97) - 0.04f;
  __syncthreads();
  if (wave >= waves_active) return;
  const int ot = wave & 3;
  f32x16 acc = {0}, acc2 = {0};
  float wv[16], wn[16], av[16], an[16];
  auto load_w = [&](float (&w)[16], int s) {
    unsigned offb = 4u * ((unsigned)(2 * 16 * (s & 3) + half) * 128u + (unsigned)(ot * 32 + j));
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if constexpr ((WHAT & W_GLOBAL) != 0) w[q] = *(const float*)((const char*)W + offb);
      else w[q] = 1e-3f * (float)(q + lane);
      offb += 8u * 128u;
    }
  };
  auto load_a = [&](float (&a)[16], int s) {
    const float* ak = sm + j * LD + 2 * 16 * (s & 3) + half;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if constexpr ((WHAT & W_LDS) != 0) a[q] = ak[2 * q];
      else a[q] = 1e-3f * (float)(q - lane);
    }
  };
  load_w(wv, 0);
  if constexpr ((WHAT & W_LDS_AHEAD) != 0) load_a(av, 0);
  const long long t0 = clock64();
  for (int s = 0; s < STEPS; ++s) {
    if constexpr ((WHAT & W_AHEAD) != 0) {
      load_w(wn, s + 1);
      if constexpr ((WHAT & W_GROUPS) == 0) __builtin_amdgcn_sched_barrier(0);
    } else {
      load_w(wv, s);
    }
    if constexpr ((WHAT & W_LDS_AHEAD) != 0) {
      load_a(an, s + 1);
      if constexpr ((WHAT & W_GROUPS) == 0) __builtin_amdgcn_sched_barrier(0);
    } else {
      load_a(av, s);
    }
    if constexpr ((WHAT & W_RELU) != 0) {
#pragma unroll
      for (int q = 0; q < 16; ++q) av[q] = fmaxf(av[q], 0.0f);
    }
    if constexpr ((WHAT & W_MFMA) != 0) {
#pragma unroll
      for (int q = 0; q < 16; q += 2) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], wv[q], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q + 1], wv[q + 1], acc2, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] += av[q] * wv[q];
    }
    if constexpr ((WHAT & W_AHEAD) != 0) {
#pragma unroll
      for (int q = 0; q < 16; ++q) wv[q] = wn[q];
    }
    if constexpr ((WHAT & W_LDS_AHEAD) != 0) {
#pragma unroll
      for (int q = 0; q < 16; ++q) av[q] = an[q];
    }
    if constexpr ((WHAT & W_GROUPS) != 0) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // one global load
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one LDS read
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);  // four vector instructions
      }
    }
  }
  float sum = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) sum += acc[r] + acc2[r];
  out[(blockIdx.x * 8 + wave) * 64 + lane] = sum;  // (the accumulators are needed: the store waits for the last MFMA)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t1 = clock64();
  if (blockIdx.x == 0 && tid == 0) *cycles = t1 - t0;
}

// The same work written the way that overlaps: the next step's operands requested, read and activated one at a time
// BETWEEN this step's MFMAs (a scheduling barrier after each group pins the order), two steps per loop trip with the
// operand buffers swapping roles so that nothing is copied.
__global__ void __launch_bounds__(512) step_kernel_hand(const float* __restrict__ W, float* __restrict__ out,
                                                        long long* cycles, int waves_active) {
  extern __shared__ float sm[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, j = lane & 31;
  for (int i = tid; i < 32 * LD; i += 512) sm[i] = 0.001f * (float)(i % 97) - 0.04f;
  __syncthreads();
  if (wave >= waves_active) return;
  const int ot = wave & 3;
  f32x16 acc = {0}, acc2 = {0};
  float w0[16], w1[16], a0[16], a1[16];
  auto w_off = [&](int s) { return 4u * ((unsigned)(2 * 16 * (s & 3) + half) * 128u + (unsigned)(ot * 32 + j)); };
  auto a_ptr = [&](int s) { return sm + j * LD + 2 * 16 * (s & 3) + half; };
  {
    unsigned offb = w_off(0);
    const float* ak = a_ptr(0);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      w0[q] = *(const float*)((const char*)W + offb + q * 8u * 128u);
      a0[q] = fmaxf(ak[2 * q], 0.0f);
    }
  }
  const long long t0 = clock64();
  auto step = [&](float (&wc)[16], float (&ac)[16], float (&wn)[16], float (&an)[16], int s) {
    const unsigned offb = w_off(s + 1);
    const float* ak = a_ptr(s + 1);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if (q & 1) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[q], wc[q], acc2, 0, 0, 0);
      else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[q], wc[q], acc, 0, 0, 0);
      wn[q] = *(const float*)((const char*)W + offb + q * 8u * 128u);
      an[q] = ak[2 * q];
      if (q > 0) an[q - 1] = fmaxf(an[q - 1], 0.0f);  // (the read of the previous group has had an MFMA to land)
      __builtin_amdgcn_sched_barrier(0);
    }
    an[15] = fmaxf(an[15], 0.0f);
  };
  for (int s = 0; s < STEPS; s += 2) {
    step(w0, a0, w1, a1, s);
    step(w1, a1, w0, a0, s + 1);
  }
  float sum = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) sum += acc[r] + acc2[r];
  out[(blockIdx.x * 8 + wave) * 64 + lane] = sum;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t1 = clock64();
  if (blockIdx.x == 0 && tid == 0) *cycles = t1 - t0;
}

template <int WHAT>
static void run(const char* name, const float* W, float* out, long long* cyc) {
  for (int waves : {4, 8}) {
    long long best = 1LL << 60;
    for (int rep = 0; rep < 5; ++rep) {
      hipLaunchKernelGGL(step_kernel<WHAT>, dim3(32), dim3(512), 32 * LD * sizeof(float), 0, W, out, cyc, waves);
      hipDeviceSynchronize();
      long long c;
      hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
      if (c < best) best = c;
    }
    printf("%-72s %d wave(s) per SIMD: %7.0f cycles per step\n", name, waves / 4, (double)best / STEPS);
  }
}

int main() {
  float *W, *out;
  long long* cyc;
  hipMalloc(&W, 128 * 128 * sizeof(float));
  hipMalloc(&out, 32 * 8 * 64 * sizeof(float));
  hipMalloc(&cyc, sizeof(long long));
  std::vector<float> h(128 * 128, 0.01f);
  hipMemcpy(W, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
  run<W_MFMA>("16 MFMAs (operands in registers)", W, out, cyc);
  run<W_MFMA | W_RELU>("+ relu", W, out, cyc);
  run<W_MFMA | W_RELU | W_LDS>("+ 16 LDS reads", W, out, cyc);
  run<W_MFMA | W_RELU | W_LDS | W_LDS_AHEAD>("+ 16 LDS reads, a step ahead", W, out, cyc);
  run<W_MFMA | W_RELU | W_LDS | W_GLOBAL>("+ 16 LDS reads + 16 global loads, used at once", W, out, cyc);
  run<W_MFMA | W_RELU | W_LDS | W_GLOBAL | W_AHEAD>("+ 16 LDS reads + 16 global loads, a step ahead", W, out, cyc);
  run<W_MFMA | W_RELU | W_LDS | W_LDS_AHEAD | W_GLOBAL | W_AHEAD>("+ both a step ahead", W, out, cyc);
  run<W_MFMA | W_RELU | W_LDS | W_LDS_AHEAD | W_GLOBAL | W_AHEAD | W_GROUPS>("+ both a step ahead, interleaved with the MFMAs (sched_group_barrier)", W, out, cyc);
  for (int waves : {4, 8}) {
    long long best = 1LL << 60;
    for (int rep = 0; rep < 5; ++rep) {
      hipLaunchKernelGGL(step_kernel_hand, dim3(32), dim3(512), 32 * LD * sizeof(float), 0, W, out, cyc, waves);
      hipDeviceSynchronize();
      long long c;
      hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
      if (c < best) best = c;
    }
    printf("%-72s %d wave(s) per SIMD: %7.0f cycles per step\n", "+ both a step ahead, one of each between two MFMAs, no copies (by hand)", waves / 4,
           (double)best / STEPS);
  }
  run<W_RELU | W_LDS | W_GLOBAL | W_AHEAD>("the same without the MFMAs (16 v_fma)", W, out, cyc);
  return 0;
}
