// Hardware facts the split-f16 mode (NSA_PREC_F16X3) relies on, checked on the box:
//   1. v_mfma_f32_32x32x16_f16 does not flush f16 subnormal operands (the lo parts of small values)
//   2. v_cvt_pk_f16_f32 overflow sets the sticky TRAPSTS.EXCP overflow bit (3); ordinary conversions,
//      MFMAs and transcendental instructions on ordinary values do not; s_setreg clears it
//   3. what a v_exp_f32 / v_rcp_f32 overflow does to the bit (needless-repeat sources)
//   4. sign and propagation of the NaN an out-of-range operand produces (inf - inf in the accumulator)
// build: hipcc --offload-arch=gfx950 -O2 -o f16_facts f16_facts.hip ; run: ./f16_facts
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int excp_overflow() {
  int v;
  asm volatile("s_nop 7\n s_nop 7\n s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 3, 1)" : "=s"(v)::"memory");
  return v;
}
__device__ __forceinline__ void excp_clear() {
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_TRAPSTS, 0, 9), 0" ::: "memory");
}
__device__ __forceinline__ int excp_all() {
  int v;
  asm volatile("s_nop 7\n s_nop 7\n s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=s"(v)::"memory");
  return v;
}

__global__ void facts(const float* in, float* out, int* flags) {
  const int lane = threadIdx.x;
  // 1. subnormal operands: A = 2^-20 (f16 subnormal) in every slot, B = 1 -> D = 16 * 2^-20
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (_Float16)9.5367431640625e-07f;  // 2^-20
    b[e] = (_Float16)1.0f;
  }
  f32x16 c = {};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (lane == 0) out[0] = c[0];
  // subnormal on the B side
  c = f32x16{};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c, 0, 0, 0);
  if (lane == 0) out[1] = c[0];

  // 2. conversion overflow and the sticky flag
  excp_clear();
  flags[0] = excp_all();  // after clear
  float x0 = in[lane], x1 = in[64 + lane];  // ordinary values
  f32x2 x = {x0, x1};
  f16x2 h = __builtin_convertvector(x, f16x2);
  asm volatile("" ::"v"(h));
  flags[1] = excp_all();  // after an ordinary conversion (inexact expected: bit 5)
  excp_clear();
  f32x2 big = {in[128 + lane], in[192 + lane]};  // lane 5 holds 70000
  f16x2 hb = __builtin_convertvector(big, f16x2);
  asm volatile("" ::"v"(hb));
  flags[2] = excp_all();  // overflow bit 3 expected
  if (lane == 5) out[2] = (float)hb[0];
  // still set after other work (sticky)
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  float t = __expf(x0) + __logf(fabsf(x1) + 1.0f) + __builtin_amdgcn_rcpf(x0 + 3.0f);
  asm volatile("" ::"v"(t), "v"(c));
  flags[3] = excp_overflow();
  excp_clear();
  // ordinary MFMA + transcendental work alone
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, b, c, 0, 0, 0);
  t = __builtin_amdgcn_exp2f(-fabsf(x0)) + __builtin_amdgcn_logf(1.0f + fabsf(x1)) + __builtin_amdgcn_sqrtf(fabsf(x0));
  asm volatile("" ::"v"(t), "v"(c));
  flags[4] = excp_all();
  // 3. exp / rcp overflow
  excp_clear();
  t = __builtin_amdgcn_exp2f(in[256 + lane]);  // lane 7: 200 -> inf
  asm volatile("" ::"v"(t));
  flags[5] = excp_all();
  excp_clear();
  t = __builtin_amdgcn_rcpf(in[320 + lane]);  // lane 9: 1e-40 (subnormal) -> inf
  asm volatile("" ::"v"(t));
  flags[6] = excp_all();
  excp_clear();
  // float64 exp overflow (populate epilogue)
  double dd = exp((double)in[256 + lane] * 10.0);
  asm volatile("" ::"v"(dd));
  flags[7] = excp_all();

  // 4. NaN of an out-of-range operand: B = {inf (hi), -inf (lo)} against finite weights
  f16x8 wpos, wneg, binf, bninf, zero;
  for (int e = 0; e < 8; ++e) {
    wpos[e] = (_Float16)0.5f;
    wneg[e] = (_Float16)-0.5f;
    binf[e] = (_Float16)0.0f;
    bninf[e] = (_Float16)0.0f;
    zero[e] = (_Float16)0.0f;
  }
  binf[0] = __builtin_bit_cast(_Float16, (unsigned short)0x7c00);
  bninf[0] = __builtin_bit_cast(_Float16, (unsigned short)0xfc00);
  c = f32x16{};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wpos, bninf, c, 0, 0, 0);  // a_h * b_l = -inf
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wpos, binf, c, 0, 0, 0);   // a_h * b_h = +inf -> NaN
  if (lane == 0) out[3] = c[0];
  f32x16 c2 = {};
  c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(zero, binf, c2, 0, 0, 0);  // 0 * inf
  if (lane == 0) out[4] = c2[0];
  f32x16 c3 = {};
  c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wneg, bninf, c3, 0, 0, 0);
  c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wneg, binf, c3, 0, 0, 0);
  if (lane == 0) out[5] = c3[0];
  // propagation: NaN operand through a product with a negative weight, and through an add
  f16x8 bn;
  for (int e = 0; e < 8; ++e) bn[e] = (_Float16)c[0];
  f32x16 c4 = {};
  c4 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wneg, bn, c4, 0, 0, 0);
  if (lane == 0) out[6] = c4[0];
  if (lane == 0) out[7] = c[0] + 1.0f;
}

int main() {
  float h_in[384];
  for (int i = 0; i < 384; ++i) h_in[i] = 0.25f + 0.01f * (i % 64);
  h_in[128 + 5] = 70000.0f;
  for (int i = 0; i < 64; ++i) h_in[256 + i] = -1.0f;
  h_in[256 + 7] = 200.0f;
  for (int i = 0; i < 64; ++i) h_in[320 + i] = 2.0f;
  h_in[320 + 9] = 1e-40f;
  float *d_in, *d_out;
  int* d_flags;
  hipMalloc(&d_in, sizeof(h_in));
  hipMalloc(&d_out, 16 * sizeof(float));
  hipMalloc(&d_flags, 16 * sizeof(int));
  hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
  hipMemset(d_out, 0, 16 * sizeof(float));
  hipMemset(d_flags, 0, 16 * sizeof(int));
  hipLaunchKernelGGL(facts, dim3(1), dim3(64), 0, 0, d_in, d_out, d_flags);
  float out[16];
  int flags[16];
  hipMemcpy(out, d_out, sizeof(out), hipMemcpyDeviceToHost);
  hipMemcpy(flags, d_flags, sizeof(flags), hipMemcpyDeviceToHost);
  uint32_t bits[16];
  memcpy(bits, out, sizeof(out));
  printf("subnormal A operand: D = %.9g (expect %.9g)\n", out[0], 16.0 * 9.5367431640625e-07);
  printf("subnormal B operand: D = %.9g (expect %.9g)\n", out[1], 16.0 * 9.5367431640625e-07);
  printf("TRAPSTS.EXCP after clear               0x%03x\n", flags[0]);
  printf("TRAPSTS.EXCP after ordinary cvt        0x%03x (bit 5 = inexact)\n", flags[1]);
  printf("TRAPSTS.EXCP after cvt of 70000        0x%03x (bit 3 = overflow expected); f16 value %g\n", flags[2], out[2]);
  printf("overflow bit after more work (sticky)  %d\n", flags[3]);
  printf("TRAPSTS.EXCP ordinary mfma+exp+log     0x%03x\n", flags[4]);
  printf("TRAPSTS.EXCP after v_exp_f32 overflow  0x%03x\n", flags[5]);
  printf("TRAPSTS.EXCP after v_rcp_f32(1e-40)    0x%03x\n", flags[6]);
  printf("TRAPSTS.EXCP after f64 exp overflow    0x%03x\n", flags[7]);
  printf("NaN of +w*(-inf) + w*inf      bits 0x%08x\n", bits[3]);
  printf("NaN of 0*inf                  bits 0x%08x\n", bits[4]);
  printf("NaN of -w*(-inf) + -w*inf     bits 0x%08x\n", bits[5]);
  printf("NaN through -w*NaN            bits 0x%08x\n", bits[6]);
  printf("NaN + 1                       bits 0x%08x\n", bits[7]);
  return 0;
}
