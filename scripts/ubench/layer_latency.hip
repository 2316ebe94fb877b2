// What does ONE hidden layer of the conditioner (32 rows x 64 -> 64, bias, relu) cost a workgroup when
// its weights are already in LDS?  The training kernels (nsa_train_fused.h) spend ~9 k cycles per
// hidden layer; this measures the floor of the same product under several work splits, as a chain of
// NL dependent layers on one 32-row tile (one workgroup; or G workgroups for the loaded case).
//   variant 0  f32 MFMA 32x32x2, out tile per wave (2 waves busy of WAVES), one barrier per layer
//   variant 1  f32 MFMA, out tile x 4-way split of k over 8 waves, partial sums meet in LDS (2 barriers)
//   variant 2  f32 MFMA, out tile x 2-way split of k over 4 waves (2 barriers)
//   variant 3  split-f16 (3 products, 32x32x16), out tile per wave (2 waves), one barrier per layer
//   variant 4  split-f16, out tile x 2-way split of k over 4 waves (2 barriers)
// SAVE = 1 also writes the layer's output to global memory (the training forward saves activations).
// build: hipcc --offload-arch=gfx950 -O2 -o layer_latency layer_latency.hip ; run: ./layer_latency
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

#define H 64
#define LD 65   // f32 variants: odd stride, conflict-free column reads
#define LDV 68  // f16 variants: rows 16-byte aligned for ds_read_b128

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int VARIANT, int WAVES, int SAVE>
__global__ void __launch_bounds__(WAVES * 64) layers(const float* __restrict__ w, const float* __restrict__ x,
                                                      float* __restrict__ out, float* __restrict__ save, int NL) {
  __shared__ __attribute__((aligned(16))) float act[2][32 * LDV];
  __shared__ __attribute__((aligned(16))) float wl[H * H];          // f32: W^T[k][o]; f16: hi/lo halves [o][k]
  __shared__ __attribute__((aligned(16))) float part[4][2][32 * 32];  // split-k partial sums
  __shared__ float bias[H];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, j = lane & 31;
  constexpr bool F16 = VARIANT >= 3;
  constexpr int ld = F16 ? LDV : LD;
  for (int i = tid; i < H * H; i += WAVES * 64) {
    if (F16) {  // [o][k] as f16 hi in the first 8 KB, lo in the second
      const int o = i / H, k = i % H;
      const float v = w[k * H + o];
      const _Float16 hi = (_Float16)v;
      ((_Float16*)wl)[o * H + k] = hi;
      ((_Float16*)wl)[H * H + o * H + k] = (_Float16)(v - (float)hi);
    } else {
      wl[i] = w[i];
    }
  }
  for (int i = tid; i < 32 * H; i += WAVES * 64) act[0][(i / H) * ld + (i % H)] = x[blockIdx.x * 32 * H + i];
  if (tid < H) bias[tid] = 0.01f * (tid & 7);
  __syncthreads();
  int cur = 0;
  for (int l = 0; l < NL; ++l) {
    const float* A = act[cur];
    float* O = act[cur ^ 1];
    if constexpr (VARIANT == 0) {
      if (wave < 2) {
        const int ot = wave;
        f32x16 acc = {0}, acc2 = {0};
        const float* ax = A + j * ld;
#pragma unroll
        for (int u0 = 0; u0 < 32; u0 += 16) {
          float av[16], wv[16];
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int k = 2 * (u0 + q) + half;
            av[q] = fmaxf(ax[k], 0.0f);
            wv[q] = wl[k * H + ot * 32 + j];
          }
#pragma unroll
          for (int q = 0; q < 16; q += 2) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], wv[q], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q + 1], wv[q + 1], acc2, 0, 0, 0);
          }
        }
        const float b = bias[ot * 32 + j];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
          const float v = acc[r] + acc2[r] + b;
          O[row * ld + ot * 32 + j] = v;
          if (SAVE) save[((size_t)l * gridDim.x * 32 + blockIdx.x * 32 + row) * H + ot * 32 + j] = v;
        }
      }
      lds_barrier();
    } else if constexpr (VARIANT == 1 || VARIANT == 2) {
      constexpr int KS = VARIANT == 1 ? 4 : 2;  // k slices
      constexpr int PAIRS = 32 / KS;             // k pairs per slice
      const int ot = wave & 1, ks = wave >> 1;
      {
        f32x16 acc = {0}, acc2 = {0};
        const float* ax = A + j * ld;
        float av[PAIRS], wv[PAIRS];
#pragma unroll
        for (int q = 0; q < PAIRS; ++q) {
          const int k = 2 * (ks * PAIRS + q) + half;
          av[q] = fmaxf(ax[k], 0.0f);
          wv[q] = wl[k * H + ot * 32 + j];
        }
#pragma unroll
        for (int q = 0; q < PAIRS; q += 2) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], wv[q], acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q + 1], wv[q + 1], acc2, 0, 0, 0);
        }
        float* pp = part[ks][ot];
#pragma unroll
        for (int r = 0; r < 16; ++r) pp[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + j] = acc[r] + acc2[r];
      }
      lds_barrier();
      // 2048 outputs over WAVES * 64 threads
      for (int e = tid; e < 32 * H; e += WAVES * 64) {
        const int row = e >> 6, o = e & 63;
        float v = bias[o];
#pragma unroll
        for (int s = 0; s < KS; ++s) v += part[s][o >> 5][row * 32 + (o & 31)];
        O[row * ld + o] = v;
        if (SAVE) save[((size_t)l * gridDim.x * 32 + blockIdx.x * 32 + row) * H + o] = v;
      }
      lds_barrier();
    } else {
      constexpr int KS = VARIANT == 3 ? 1 : 2;
      constexpr int STEPS = 4 / KS;  // k steps of 16 per slice
      const int ot = wave & 1, ks = wave >> 1;
      if (VARIANT == 4 || wave < 2) {
        f32x16 acc = {0};
        const _Float16* whi = (const _Float16*)wl + (ot * 32 + j) * H;
        const _Float16* wlo = whi + H * H;
        const float* ax = A + j * ld;
        f16x8 ah[STEPS], al[STEPS], bh[STEPS], bl[STEPS];
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
          const int k0 = (ks * STEPS + s) * 16 + 8 * half;
          const f32x4 v0 = *(const f32x4*)(ax + k0), v1 = *(const f32x4*)(ax + k0 + 4);
          bh[s] = *(const f16x8*)(whi + k0);
          bl[s] = *(const f16x8*)(wlo + k0);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float a0 = fmaxf(v0[e], 0.0f), a1 = fmaxf(v1[e], 0.0f);
            const _Float16 h0 = (_Float16)a0, h1 = (_Float16)a1;
            ah[s][e] = h0;
            ah[s][e + 4] = h1;
            al[s][e] = (_Float16)(a0 - (float)h0);
            al[s][e + 4] = (_Float16)(a1 - (float)h1);
          }
        }
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], bh[s], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bl[s], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh[s], acc, 0, 0, 0);
        }
        if constexpr (VARIANT == 3) {
          const float b = bias[ot * 32 + j];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
            const float v = acc[r] + b;
            O[row * ld + ot * 32 + j] = v;
            if (SAVE) save[((size_t)l * gridDim.x * 32 + blockIdx.x * 32 + row) * H + ot * 32 + j] = v;
          }
        } else {
          float* pp = part[ks][ot];
#pragma unroll
          for (int r = 0; r < 16; ++r) pp[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + j] = acc[r];
        }
      }
      lds_barrier();
      if constexpr (VARIANT == 4) {
        for (int e = tid; e < 32 * H; e += WAVES * 64) {
          const int row = e >> 6, o = e & 63;
          const float v = bias[o] + part[0][o >> 5][row * 32 + (o & 31)] + part[1][o >> 5][row * 32 + (o & 31)];
          O[row * ld + o] = v;
          if (SAVE) save[((size_t)l * gridDim.x * 32 + blockIdx.x * 32 + row) * H + o] = v;
        }
        lds_barrier();
      }
    }
    cur ^= 1;
  }
  __syncthreads();
  for (int i = tid; i < 32 * H; i += WAVES * 64) out[blockIdx.x * 32 * H + i] = act[cur][(i / H) * ld + (i % H)];
}

template <int VARIANT, int WAVES, int SAVE>
static void run(const char* name, const float* w, const float* x, float* out, float* save, int grid, float* ref) {
  const int NL = 200;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  layers<VARIANT, WAVES, SAVE><<<grid, WAVES * 64>>>(w, x, out, save, NL);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int it = 0; it < 5; ++it) layers<VARIANT, WAVES, SAVE><<<grid, WAVES * 64>>>(w, x, out, save, NL);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  float h[8];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  if (ref[0] == 0.0f) ref[0] = h[3];
  printf("%-44s grid %3d save %d: %8.1f ns per layer (%6.0f cycles at 2.4 GHz)  out[3] %.6f (d %.1e)\n", name, grid, SAVE,
         ms * 1e6 / (5.0 * NL), ms * 1e6 / (5.0 * NL) * 2.4, h[3], h[3] - ref[0]);
}

int main() {
  const int G = 64, NL = 200;
  float *w, *x, *out, *save;
  hipMalloc(&w, H * H * 4);
  hipMalloc(&x, G * 32 * H * 4);
  hipMalloc(&out, G * 32 * H * 4);
  hipMalloc(&save, (size_t)NL * G * 32 * H * 4);
  float* hw = (float*)malloc(H * H * 4);
  float* hx = (float*)malloc(G * 32 * H * 4);
  srand(1);
  for (int i = 0; i < H * H; ++i) hw[i] = ((rand() % 2001) - 1000) * 1.7e-4f;  // spectral radius ~ 1 under relu
  for (int i = 0; i < G * 32 * H; ++i) hx[i] = ((rand() % 2001) - 1000) * 1e-3f;
  hipMemcpy(w, hw, H * H * 4, hipMemcpyHostToDevice);
  hipMemcpy(x, hx, G * 32 * H * 4, hipMemcpyHostToDevice);
  float ref[1] = {0.0f};
  for (int grid : {1, 32}) {
    run<0, 2, 0>("f32, tile per wave, 2 waves", w, x, out, save, grid, ref);
    run<0, 8, 0>("f32, tile per wave, 2 of 8 waves", w, x, out, save, grid, ref);
    run<1, 8, 0>("f32, 4-way k split, 8 waves", w, x, out, save, grid, ref);
    run<2, 4, 0>("f32, 2-way k split, 4 waves", w, x, out, save, grid, ref);
    run<3, 2, 0>("f16x3, tile per wave, 2 waves", w, x, out, save, grid, ref);
    run<4, 4, 0>("f16x3, 2-way k split, 4 waves", w, x, out, save, grid, ref);
    run<0, 2, 1>("f32, tile per wave, 2 waves", w, x, out, save, grid, ref);
    run<1, 8, 1>("f32, 4-way k split, 8 waves", w, x, out, save, grid, ref);
    run<3, 2, 1>("f16x3, tile per wave, 2 waves", w, x, out, save, grid, ref);
  }
  return 0;
}
