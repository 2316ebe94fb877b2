// Micro-benchmark (diagnostic, not part of the library): issue rate of the vector instructions the flow
// kernel is made of, per SIMD, with 1 / 2 / 4 waves per SIMD each running 8 independent chains.  Prints
// nanoseconds per wave-instruction per SIMD and the rate relative to v_fma_f32 - i.e. which instructions
// a VALU-bound kernel (the weights-resident geometry on the reference's default RealNVP) should avoid.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/valu_rates scripts/ubench/valu_rates.hip && scripts/ubench/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHAINS 8
#define UNROLL 32

#define KERNEL(name, asm_text, ...)                                                            \
  __global__ void __launch_bounds__(1024) name(float* out, int reps) {                          \
    float v[CHAINS];                                                                            \
    for (int i = 0; i < CHAINS; ++i) v[i] = 1.0f + 0.001f * (float)(threadIdx.x + i);           \
    float k = 1.0000001f, c = 1e-7f;                                                            \
    asm volatile("" : "+v"(k), "+v"(c));                                                        \
    for (int rep = 0; rep < reps; ++rep) {                                                      \
      _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                                      \
        _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile(asm_text : "+v"(v[i]) : "v"(k), "v"(c) __VA_ARGS__); \
      }                                                                                         \
    }                                                                                           \
    float s = 0.0f;                                                                             \
    for (int i = 0; i < CHAINS; ++i) s += v[i];                                                 \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                             \
  }

KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2", )
KERNEL(k_add, "v_add_f32 %0, %0, %2", )
KERNEL(k_sub, "v_sub_f32 %0, %0, %2", )
KERNEL(k_mul, "v_mul_f32 %0, %0, %1", )
KERNEL(k_max_i32, "v_max_i32 %0, %0, %2", )
KERNEL(k_and, "v_and_b32 %0, %0, %1", )
KERNEL(k_mov, "v_mov_b32 %0, %1", )
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, s[10:11]", : "s10", "s11")
KERNEL(k_cmp, "v_cmp_lt_f32 vcc, %0, %1", : "vcc")
KERNEL(k_cmp_cnd, "v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc", : "vcc")
KERNEL(k_max_f32, "v_max_f32 %0, %0, %2", )
KERNEL(k_med3_f32, "v_med3_f32 %0, %0, %1, %2", )
KERNEL(k_add_u32, "v_add_u32 %0, %0, %1", )
KERNEL(k_lshlrev, "v_lshlrev_b32 %0, 1, %0", )
KERNEL(k_bfi, "v_bfi_b32 %0, %1, %0, %2", )
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2", )
KERNEL(k_ashr, "v_ashrrev_i32 %0, 31, %0", )
KERNEL(k_cvt_pk_f16, "v_cvt_pk_f16_f32 %0, %0, %1", )
KERNEL(k_cvt_f32_f16, "v_cvt_f32_f16 %0, %0", )
KERNEL(k_cvt_f32_f16_sdwa, "v_cvt_f32_f16_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1", )
KERNEL(k_fma_mix, "v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]", )
KERNEL(k_exp, "v_exp_f32 %0, %0", )
KERNEL(k_log, "v_log_f32 %0, %0", )
KERNEL(k_rcp, "v_rcp_f32 %0, %0", )
KERNEL(k_sqrt, "v_sqrt_f32 %0, %0", )
KERNEL(k_sin, "v_sin_f32 %0, %0", )
KERNEL(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1", )
KERNEL(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1", )
KERNEL(k_xor, "v_xor_b32 %0, %0, %1", )
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 1, %1", )
KERNEL(k_cvt_f64_f32, "v_cvt_f32_u32 %0, %0", )
KERNEL(k_pk_max_f16, "v_pk_max_f16 %0, %0, %1", )

// packed f32 (two values per lane and instruction): 64-bit register pairs
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define KERNEL2(name, asm_text)                                                                 \
  __global__ void __launch_bounds__(1024) name(float* out, int reps) {                          \
    f32x2 v[CHAINS];                                                                            \
    for (int i = 0; i < CHAINS; ++i) v[i] = f32x2{1.0f + 0.001f * (float)(threadIdx.x + i), 2.0f}; \
    f32x2 k = {1.0000001f, 0.9999999f}, c = {1e-7f, 2e-7f};                                     \
    asm volatile("" : "+v"(k), "+v"(c));                                                        \
    for (int rep = 0; rep < reps; ++rep) {                                                      \
      _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                                      \
        _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile(asm_text : "+v"(v[i]) : "v"(k), "v"(c)); \
      }                                                                                         \
    }                                                                                           \
    float s = 0.0f;                                                                             \
    for (int i = 0; i < CHAINS; ++i) s += v[i][0] + v[i][1];                                    \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                             \
  }
KERNEL2(k_pk_add_f32, "v_pk_add_f32 %0, %0, %2")
KERNEL2(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %2")
KERNEL2(k_pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
KERNEL2(k_fma_f64, "v_fma_f64 %0, %0, %1, %2")
KERNEL2(k_add_f64, "v_add_f64 %0, %0, %2")

typedef void (*kern_t)(float*, int);

static double time_kernel(kern_t k, int waves_per_simd, int reps, float* out) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int threads = 256 * waves_per_simd;
  k<<<256, threads>>>(out, 2);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<<<256, threads>>>(out, reps);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0.0f;
  hipEventElapsedTime(&ms, a, b);
  const double insts_per_simd = (double)reps * UNROLL * CHAINS * waves_per_simd;
  return ms * 1e6 / insts_per_simd;  // ns per wave-instruction per SIMD
}

int main() {
  float* out;
  hipMalloc(&out, sizeof(float) * 256 * 1024);
  struct { const char* name; kern_t k; } ks[] = {
      {"v_fma_f32", k_fma}, {"v_add_f32", k_add}, {"v_sub_f32", k_sub}, {"v_mul_f32", k_mul}, {"v_max_i32", k_max_i32},
      {"v_and_b32", k_and}, {"v_mov_b32", k_mov}, {"v_cndmask_b32 (sgpr mask)", k_cndmask}, {"v_cmp_lt_f32", k_cmp}, {"v_cmp + v_cndmask (2 insts)", k_cmp_cnd},
      {"v_max_f32", k_max_f32}, {"v_med3_f32", k_med3_f32}, {"v_add_u32", k_add_u32}, {"v_lshlrev_b32", k_lshlrev},
      {"v_bfi_b32", k_bfi}, {"v_perm_b32", k_perm}, {"v_ashrrev_i32", k_ashr},
      {"v_cvt_pk_f16_f32", k_cvt_pk_f16}, {"v_cvt_f32_f16", k_cvt_f32_f16}, {"v_cvt_f32_f16_sdwa", k_cvt_f32_f16_sdwa},
      {"v_fma_mix_f32", k_fma_mix}, {"v_pk_add_f32 (2 values)", k_pk_add_f32}, {"v_pk_fma_f32 (2 values)", k_pk_fma_f32},
      {"v_pk_mul_f32 (2 values)", k_pk_mul_f32}, {"v_fma_f64", k_fma_f64}, {"v_add_f64", k_add_f64}, {"v_pk_max_f16", k_pk_max_f16}, {"v_exp_f32", k_exp}, {"v_log_f32", k_log},
      {"v_rcp_f32", k_rcp}, {"v_sqrt_f32", k_sqrt}, {"v_sin_f32", k_sin}, {"v_mul_hi_u32", k_mul_hi_u32},
      {"v_mul_lo_u32", k_mul_lo_u32}, {"v_xor_b32", k_xor}, {"v_lshl_add_u32", k_lshl_add}, {"v_cvt_f32_u32", k_cvt_f64_f32}};
  printf("%-26s %10s %10s %10s   (ns per wave-instruction per SIMD; x = relative to v_fma_f32 at 4 waves)\n", "instruction",
         "1 wave", "2 waves", "4 waves");
  double base = 0.0;
  for (auto& e : ks) {
    const double t1 = time_kernel(e.k, 1, 400, out), t2 = time_kernel(e.k, 2, 400, out), t4 = time_kernel(e.k, 4, 400, out);
    if (base == 0.0) base = t4;
    printf("%-26s %10.3f %10.3f %10.3f   x%.2f\n", e.name, t1, t2, t4, t4 / base);
  }
  return 0;
}
