// Device-side building blocks for the gfx950 flow kernels.
//
// Wave-level data layout (one wave = one 32-row tile, wave64):
//   lane = 32*half + j  ->  batch row j of the tile, "half" selects which of
//   the two k-slots / output-feature groups of the 32x32x2 f32 MFMA the lane
//   holds.  v_mfma_f32_32x32x2_f32 computes D[i][j] += sum_k A[i][k]*B[k][j]
//   with A[i][k] in lane 32*k+i, B[k][j] in lane 32*k+j and D[i][j] in lane
//   32*((i>>2)&1) + j, register (i&3) + 4*(i>>3).  Because the D layout of one
//   layer is exactly a valid B layout for the next (with the k order permuted,
//   which the host bakes into the packed weights), conditioner activations
//   never leave registers between layers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nsa_format.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define NSA_ROWS 32  // rows per wave tile

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------

__device__ __forceinline__ float nsa_bits_to_float(uint32_t u) { return __uint_as_float(u); }

__device__ __forceinline__ f32x16 nsa_mfma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// 16-bit matrix cores: D[i][j] += sum_k A[i][k] B[k][j], k = 8*half + e (16 per instruction).  The
// operands travel as 128-bit words (eight bf16 or eight f16); PREC selects the instruction:
//   NSA_PREC_BF16X3  v_mfma_f32_32x32x16_bf16      NSA_PREC_F16X3  v_mfma_f32_32x32x16_f16
// (same rate, same lane layout).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef u32x4 nsa_op8;  // eight 16-bit operand values of one lane

template <int PREC>
__device__ __forceinline__ f32x16 nsa_mfma16(nsa_op8 a, nsa_op8 b, f32x16 c) {
#if defined(NSA_ABLATE) && (NSA_ABLATE & 8)
  asm volatile("" ::"v"(a), "v"(b));  // diagnostic build: operands stay live, no matrix work
  return c;
#else
  if constexpr (PREC == NSA_PREC_F16X3)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
#endif
}

// x = hi + lo with hi = r16(x), lo = r16(x - hi)   (round to nearest even; r16 = bf16 or f16)
// bf16: two values per v_cvt_pk_bf16_f32; float(hi) comes back out of the packed pair with one shift or
// mask per value (written per element, the compiler converted every value twice - once alone for the
// subtraction, once packed for the operand: 4 VALU per value instead of 3).  8 + 8 significant bits:
// ~2^-17 relative per operand.
// f16: v_cvt_pk_f16_f32, float(hi) by v_cvt_f32_f16 (the upper half through SDWA) - also 3 VALU per
// value.  11 + 11 significant bits: x - hi is exact in float32 and |lo| <= 2^-11 |x|, so hi + lo carries x
// to 2^-24 relative (float32 itself) while lo stays a normal f16 (|x| >= 2^-3), and to 2^-25 ABSOLUTE
// below that (f16 subnormals, which the matrix cores do not flush).  The price is the range: |x| must
// stay below 65520, where f16 rounds to infinity.  A launch that split such a value must not be used, and
// the kernel says so in status[1] without an instruction per value: hi = +-inf makes lo = x - hi = -+inf,
// so EVERY product of that batch row's column - a_l*b_h, a_h*b_l, a_h*b_h with a zero or non-zero weight -
// is inf - inf or 0 * inf: all accumulators of the row come out NaN.  A NaN travels to the row's outputs
// through every path of the flow except an activation (the default NaN of the matrix cores is negative,
// a relu returns 0 for it: scripts/ubench/f16_facts.hip), so one accumulator per row is tested after the
// GEMMs that feed an activation and the row's log-det / outputs in the epilogue.  (The wave's sticky IEEE
// flags do not work for this: TRAPSTS.EXCP stays 0 on gfx950 whatever overflows.  A running maximum over the
// split values, even as one v_pk_max_u16 per pair on the hi words, cost 25-120 more spilled VGPRs in the
// two-tiles instances: NSA_F16_DETECT=2 keeps that form for cross-checks.)  The host repeats such a call
// in exact float32 (HipFlow: precision "auto") or raises.  A NaN of any other origin is reported too: a
// needless repeat, never a miss.
#ifndef NSA_F16_DETECT
#define NSA_F16_DETECT 1
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

#define NSA_F16_INF 0x7c00

template <int PREC>
__device__ __forceinline__ void nsa_split8(const float (&v)[8], nsa_op8& hi, nsa_op8& lo, u16x2& amax) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x2 x = {v[2 * q], v[2 * q + 1]};
    if constexpr (PREC == NSA_PREC_F16X3) {
      // x - float(hi) as fma(float(hi), -1, x): with the -1 opaque to the optimiser (it would turn the fma
      // back into cvt + sub: 3 VALU per value) this selects v_fma_mix_f32, which reads the f16 half
      // directly - cvt_pk, 2 x fma_mix, cvt_pk: 2 VALU per value; the result is the same exact difference
      float m1 = -1.0f;
      asm("" : "+s"(m1));
      const f16x2 h = __builtin_convertvector(x, f16x2);
      const f32x2 d = {__builtin_fmaf((float)h[0], m1, x[0]), __builtin_fmaf((float)h[1], m1, x[1])};
      hi[q] = __builtin_bit_cast(uint32_t, h);
      lo[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(d, f16x2));
#if NSA_F16_DETECT == 2
      amax = __builtin_elementwise_max(amax, __builtin_bit_cast(u16x2, hi[q] & 0x7fff7fffu));
#endif
    } else {
      const uint32_t hp = __builtin_bit_cast(uint32_t, __builtin_convertvector(x, bf16x2));
      const f32x2 d = {x[0] - __builtin_bit_cast(float, hp << 16), x[1] - __builtin_bit_cast(float, hp & 0xffff0000u)};
      hi[q] = hp;
      lo[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(d, bf16x2));
    }
  }
}

// three-product split multiply-accumulate: a*b ~ ah*bh + ah*bl + al*bh
template <int PREC>
__device__ __forceinline__ f32x16 nsa_mfma_x3(nsa_op8 ah, nsa_op8 al, nsa_op8 bh, nsa_op8 bl, f32x16 c) {
  c = nsa_mfma16<PREC>(al, bh, c);
  c = nsa_mfma16<PREC>(ah, bl, c);
  return nsa_mfma16<PREC>(ah, bh, c);
}

// relu on the integer ALU: negative floats are negative ints.  One instruction (fmaxf(x, 0) is two:
// the compiler canonicalises a possibly-signalling input first); a NaN with the sign bit set
// becomes 0, exactly as fmaxf returns the non-NaN operand.
__device__ __forceinline__ float nsa_relu(float x) {
  const int b = __builtin_bit_cast(int, x);
  return __builtin_bit_cast(float, b > 0 ? b : 0);
}

// tanh and silu on the bare transcendental units (v_exp_f32, v_rcp_f32: 1 ulp each): tanhf / expf + an IEEE division
// are ~40 instructions per value, and a conditioner has 64-256 values per row and layer - a tanh flow ran 2x the time
// of the same flow with relu.
//   silu(x) = x / (1 + 2^(-x log2 e));  overflow of the power gives x * 0
//   tanh(x) = 1 - 2 / (1 + 2^(2 x log2 e)) away from zero, where the subtraction is harmless;
//             x (1 - x^2/3 + 2 x^4/15 - 17 x^6/315 + 62 x^8/2835) for |x| < 1/4 (next term < 1e-8 relative);
//             worst error 1.3e-7 absolute, 5e-7 relative (just above 1/4), against float64
__device__ __forceinline__ float nsa_silu_fast(float x) {
  const float e = __builtin_amdgcn_exp2f(x * -1.4426950408889634f);
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float nsa_tanh_fast(float x) {
  const float t = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  const float big = fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + t), 1.0f);  // t = inf -> 1, t = 0 -> -1
  const float s = x * x;
  const float small = x * fmaf(s, fmaf(s, fmaf(s, fmaf(s, 0.021869488536155203f, -0.053968253968253971f), 0.13333333333333333f),
                                        -0.33333333333333333f), 1.0f);
  return __builtin_fabsf(x) < 0.25f ? small : big;
}

// activation of the conditioner (src/nessai/flows/utils.py:200-205)
__device__ __forceinline__ float nsa_act(float x, int kind) {
  if (kind == NSA_ACT_RELU) return nsa_relu(x);
  if (kind == NSA_ACT_TANH) return nsa_tanh_fast(x);
  return nsa_silu_fast(x);  // silu: x * sigmoid(x)
}

// Async global -> LDS copy of one 1 KiB piece per wave instruction
// (global_load_lds_dwordx4: LDS destination = wave-uniform base + lane*16).
__device__ __forceinline__ void nsa_glds16(const uint32_t* gsrc_lane, float* lds_base_uniform) {
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void*)gsrc_lane,
      (__attribute__((address_space(3))) void*)lds_base_uniform, 16, 0, 0);
}

// The same copy as one the COMPILER does not know to be a vector-memory write to LDS (inline assembly; M0 - the
// LDS base of an LDS-DMA - saved and restored inside it).  hipcc orders LDS reads it cannot tell from the copy's
// destination - and register reuse next to it - behind every pending LDS-DMA with an s_waitcnt vmcnt(0) of its own
// (scripts/isa_waits.py shows them): harmless with ONE copy in flight that the stage waits for anyway, fatal for a
// ring that keeps several stages on their way (NsaRingPipe, nsa_flow_kernel.h) - there every wait on the counter is
// written out (nsa_wait_vmcnt) and the reads of a buffer sit behind the counted wait and a barrier.
__device__ __forceinline__ void nsa_glds16_raw(const uint32_t* gsrc_lane, float* lds_base_uniform) {
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)lds_base_uniform;
  // (no scalar operand: the base goes in as a VECTOR register and is made scalar inside, with VCC as the scratch pair -
  // with an "s" operand, in or out, the instances that spill - 256-wide conditioners - did not compile: "illegal
  // VGPR to SGPR copy")
  asm volatile("s_mov_b32 vcc_hi, m0\n\tv_readfirstlane_b32 vcc_lo, %0\n\ts_mov_b32 m0, vcc_lo\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, vcc_hi"
               :
               : "v"(base), "v"(gsrc_lane)
               : "vcc");
}

// ---------------------------------------------------------------------------
// Philox4x32-10 counter-based generator (Salmon et al. 2011) + Box-Muller.
// Counter = (global row lo, global row hi, quad index, stream); key = seed.
// The same function is restated in numpy in oracle/philox.py for the tests.
// ---------------------------------------------------------------------------

__device__ __forceinline__ void nsa_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                  uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += W0; k1 += W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// u32 -> float in (0, 1]: (x + 0.5) * 2^-32 evaluated in f32
__device__ __forceinline__ float nsa_u01(uint32_t x) {
  return fmaf((float)x, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
}

// two u32 -> two standard normals: r = sqrt(-2 ln u1), angle = u2 revolutions.  On the hardware's
// transcendental unit - v_log_f32 (log2), v_sqrt_f32, v_sin_f32 / v_cos_f32 (argument in revolutions) - 12
// instructions where logf + sqrtf + sincospif took ~80 per pair; u1 >= 2^-33 is a normal number, u2 in (0, 1]
// needs no range reduction.  (A random stream is defined by what the kernel does: oracle/philox.py holds the
// same map - Box-Muller on the (x + 0.5) 2^-32 uniforms - in numpy's float arithmetic, and the tests hold the
// device to it within the transcendental units' error, a few 1e-6 absolute.)
__device__ __forceinline__ void nsa_box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
  const float u1 = nsa_u01(a), u2 = nsa_u01(b);
  const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));  // -2 ln2 log2(u1)
  n0 = r * __builtin_amdgcn_cosf(u2);
  n1 = r * __builtin_amdgcn_sinf(u2);
}

// ---------------------------------------------------------------------------
// Element-wise coupling transforms (one feature of one row per call).
// Policies expose: C (MFMA tiles per chunk), FPC (features per lane per
// chunk), NP (parameters per feature) and apply().
// ---------------------------------------------------------------------------

struct NsaTransformConsts {
  float tail_bound, two_b;          // B and (right - left) = 2B
  float min_bw, bw_scale;           // min_bin_width, 1 - min_bin_width*K
  float min_bh, bh_scale;
  float min_d, edge_d;              // min_derivative, boundary derivative (== 1 up to f32 rounding)
  float edge_raw;                   // its logit log(exp(1 - min_d) - 1): edge_d == min_d + softplus(edge_raw)
  float inv_logit_div;              // 1/sqrt(hidden_features) or 1
  float aff_a, aff_b;               // affine coupling: scale = sigmoid(raw + aff_a) + aff_b
  float aff_arg0, aff_1pb;          // -aff_a log2(e), 1 + aff_b
  int k_real;                       // spline bins of the flow (= K of the instance; <= K in the padded generic instance)
  float aff_cap;                    // cap of -(raw + a) log2(e): 64 (sigmoid is 0 beside a floor b >= 1e-9), 126 for smaller floors
  bool ld_group4;                   // Jacobian factors of an LD_FACTOR transform may be multiplied four at a time
};

// Fast-but-accurate scalar helpers.  The spline's arithmetic *structure* follows
// oracle/nflows_restated/splines.py; transcendental pieces use the hardware
// v_exp/v_log/v_rcp units (<= 2 ulp) instead of the multi-instruction IEEE
// sequences, which keeps log_prob within the 1e-5 relative parity bar (measured
// in tests/test_gpu_flows.py) at a third of the VALU instruction count.
__device__ __forceinline__ float nsa_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float nsa_exp(float x) { return __expf(x); }
__device__ __forceinline__ float nsa_log(float x) { return __logf(x); }

// log(1 + u) for u in [0, 1]: 2*atanh(u / (2 + u)) as an odd series, relative
// error < 1e-7 over the whole range (no cancellation for tiny u).
__device__ __forceinline__ float nsa_log1p_unit(float u) {
  const float s = u * nsa_rcp(2.0f + u);  // <= 1/3
  const float t = s * s;
  float p = fmaf(t, 1.0f / 15.0f, 1.0f / 13.0f);
  p = fmaf(t, p, 1.0f / 11.0f);
  p = fmaf(t, p, 1.0f / 9.0f);
  p = fmaf(t, p, 1.0f / 7.0f);
  p = fmaf(t, p, 1.0f / 5.0f);
  p = fmaf(t, p, 1.0f / 3.0f);
  p = fmaf(t, p, 1.0f);
  return 2.0f * s * p;
}

// F.softplus(x) = max(x, 0) + log1p(exp(-|x|)); equals x for x > 20 in f32,
// which is torch's threshold behaviour.
__device__ __forceinline__ float nsa_softplus(float x) {
  return fmaxf(x, 0.0f) + nsa_log1p_unit(nsa_exp(-fabsf(x)));
}

// Element-wise transforms.  A policy provides
//   C, FPC, NP, NPH (phases per feature), State, and phase<P, OFF>(...);
// running phases 0..NPH-1 in order transforms one feature of one row (all three policies have a
// single phase now; the hook is kept so that a transform can be split again).

// ln(x) for normal positive x on the bare v_log_f32 (log2, <= 1 ulp); __logf adds a denormal
// rescue and a two-term log2 -> ln conversion (12 instructions) that the spline does not need
__device__ __forceinline__ float nsa_log2(float x) { return __builtin_amdgcn_logf(x); }

// F.softplus(x) = max(x, 0) + log1p(exp(-|x|)) with log1p(u) = ln2 * log2(1 + u): the rounding of
// 1 + u costs <= 6e-8 absolute, i.e. <= 6e-5 relative on min_derivative + softplus = 1e-3 at the
// far negative end and ~1e-7 where trained derivative logits live
__device__ __forceinline__ float nsa_softplus_fast(float x) {
  const float u = __builtin_amdgcn_exp2f(-1.4426950408889634f * fabsf(x));
  return fmaf(0.6931471805599453f, nsa_log2(1.0f + u), nsa_relu(x));
}

// Bin search on monotone knots by halving, carried out on THREE parallel tables (the knots that are
// searched, the knots of the other axis, the derivative logits): a window of N bins keeps N + 1
// entries of each; one compare against the middle knot keeps the half that holds x.
// 3 + 30 instructions for 8 bins (7 + 42 for the linear scan).  Same index as
// searchsorted's count of knots <= x because the knots increase strictly (min bin size > 0).
template <int N>
struct NsaBinSelect {
  __device__ __forceinline__ static void run(const float (&a)[N + 1], const float (&b)[N + 1],
                                             const float (&r)[N + 1], float x, float (&o)[6]) {
    constexpr int S = (N + 1) / 2;  // bins kept
    constexpr int M = N / 2;        // bin >= M  <=>  x >= a[M]; note N - S == M
    const bool ge = x >= a[M];
    float a2[S + 1], b2[S + 1], r2[S + 1];
#pragma unroll
    for (int i = 0; i <= S; ++i) {
      a2[i] = ge ? a[M + i] : a[i];
      b2[i] = ge ? b[M + i] : b[i];
      r2[i] = ge ? r[M + i] : r[i];
    }
    NsaBinSelect<S>::run(a2, b2, r2, x, o);
  }
};
template <>
struct NsaBinSelect<1> {
  __device__ __forceinline__ static void run(const float (&a)[2], const float (&b)[2],
                                             const float (&r)[2], float, float (&o)[6]) {
    o[0] = a[0]; o[1] = a[1]; o[2] = b[0]; o[3] = b[1]; o[4] = r[0]; o[5] = r[1];
  }
};

// Monotonic rational-quadratic spline with linear tails.  Restates the
// arithmetic of oracle/nflows_restated/splines.py for one scalar input.
//   v[0..K)      unnormalised widths
//   v[K..2K)     unnormalised heights
//   v[2K..3K-1)  unnormalised interior derivatives
// PAD (the generic instance, NsaRQ<16, true>: any num_bins from 2 to 16 - the reference passes num_bins through,
// flows/nsf.py:66,103): the flow has c.k_real <= K bins and the packer (pack.py, _CouplingBlobs) deals its 3 k_real - 1
// parameters to the first slots of each of the three groups of K; the logits of the slots beyond k_real carry the
// bias -1e30 (their exponentials are 0: they take no part in a softmax), the derivative slot of knot k_real the
// boundary logit.  Knot k_real is the right end B, the knots beyond it +inf: the bin search never enters them.
// Costs two selects per knot over the exact instances; the exact ones (K = 4, 8, 10, 16) are untouched.
template <int K, bool PAD = false>
struct NsaRQ {
  static constexpr int NP = 3 * K - 1;
  static constexpr int FPC = (K == 8) ? 2 : 1;
  static constexpr int C = (FPC * NP + 15) / 16;
  static constexpr int KIND = NSA_T_RQ;
  static constexpr int NPH = 1;
  static constexpr bool LD_FACTOR = false;

  struct State {};

  // knots of one axis: softmax -> min + scale*p -> cumulative sum mapped to [-B, B]:
  //   cum[i] = -B + i*(2B*min) + (2B*scale/sum) * prefix_i,   prefix_i = sum_{j<i} exp(l_j - shift)
  // The logits are scaled by log2(e)/sqrt(H) once so that the exponentials are bare v_exp_f32;
  // the prefix sums double as the softmax denominator; the ends are exact.
  // SHIFTED = false (round 5): no maximum is subtracted - a softmax does not depend on the shift, and the logits
  // of a conditioner are O(1-10), so the plain exponentials are as exact as the shifted ones (8 of ~180 vector
  // instructions per evaluation; 2 % of the draw kernel).  The caller checks that the two sums stayed where
  // nothing overflowed and no term of weight was flushed (`sums_in_range`) and repeats the evaluation with
  // SHIFTED = true - torch's max-subtracted softmax, safe for any logits - when a lane of the wave left it.
  // Returns the sum of the exponentials.
  template <int OFF, int NV, bool SHIFTED>
  __device__ __forceinline__ static float knots(const float (&v)[NV], float inv_div, float mn,
                                                float sc, float two_b, float B, float (&cum)[K + 1], int k_real = K) {
    const float k2 = inv_div * 1.4426950408889634f;
    float mk = 0.0f;
    if constexpr (SHIFTED) {
      // max over the raw logits (the scale is positive), then scale and shift in ONE fma per logit
      float m = v[OFF];
#pragma unroll
      for (int i = 1; i < K; ++i) m = fmaxf(m, v[OFF + i]);
      mk = m * k2;
    }
    float pre[K + 1];
    pre[0] = 0.0f;
#pragma unroll
    for (int i = 0; i < K; ++i)
      pre[i + 1] = pre[i] + __builtin_amdgcn_exp2f(SHIFTED ? fmaf(v[OFF + i], k2, -mk) : v[OFF + i] * k2);
    const float g = (two_b * sc) * nsa_rcp(pre[K]);
    const float c0 = two_b * mn;
#pragma unroll
    for (int i = 1; i < K; ++i) cum[i] = fmaf(g, pre[i], fmaf((float)i, c0, -B));
    cum[0] = -B;
    cum[K] = B;
    if constexpr (PAD) {
      // (k_real is wave-uniform: the conditions are scalar compares, the selects v_cndmask on a scalar mask)
#pragma unroll
      for (int i = 1; i < K; ++i) {
        const float top = (i == k_real) ? B : __builtin_huge_valf();
        cum[i] = (i < k_real) ? cum[i] : top;
      }
      cum[K] = (k_real == K) ? B : __builtin_huge_valf();
    }
    return pre[K];
  }

  // the sum of an unshifted softmax inside [2^-100, 2^100]: no exponential overflowed (each <= the sum) and none
  // that carries weight was flushed (a term below 2^-126 is below 2^-26 of such a sum); NaN -> false
  __device__ __forceinline__ static bool sum_in_range(float s) {
    return (s >= 7.888609052210118e-31f) & (s <= 1.2676506002282294e30f);
  }

  // The evaluation in three pieces (round 6: a caller can place other work - the next chunk's MFMAs - inside each of
  // them; `phase` below runs them back to back, the same instruction stream as before the split):
  //   knots_checked<OFF>      the knots of one axis (widths: OFF, bin sizes min_bw / bw_scale; heights: OFF + K, ...)
  //   evaluate<OFF, INV>      bin search, the two derivatives, the rational-quadratic map and its log|J|
  // The two wave-uniform "redo" branches (below) end a basic block each: three straight-line pieces per evaluation.
  template <int OFF, int NV>
  __device__ __forceinline__ static void knots_checked(const float (&v)[NV], const NsaTransformConsts& c, float mn,
                                                       float sc, float (&cum)[K + 1]) {
    const float B = c.tail_bound;
    // (checked per axis, right behind its knots: the eight logits of an axis then stay live a few instructions
    // longer, not across the other axis - checked together they cost the two-tiles instance two spilled registers;
    // the branches are wave-uniform and practically never taken)
#ifdef NSA_SHIFTED_SOFTMAX  // A/B builds (scripts/build_variant.sh): the max-subtracted softmax of rounds 1-4
    knots<OFF, NV, true>(v, c.inv_logit_div, mn, sc, c.two_b, B, cum, c.k_real);
#else
    // A row's result must not depend on the rows that share its wave (the same rows come back bit for bit from any
    // launch geometry and from nsa_populate_replay): the branch is taken by the wave, the repeat is KEPT per lane.
    const float sw = knots<OFF, NV, false>(v, c.inv_logit_div, mn, sc, c.two_b, B, cum, c.k_real);
    const bool redo = !sum_in_range(sw);
    if (__builtin_expect(__any(redo), 0)) {
      float alt[K + 1];
      knots<OFF, NV, true>(v, c.inv_logit_div, mn, sc, c.two_b, B, alt, c.k_real);
#pragma unroll
      for (int i = 0; i <= K; ++i) cum[i] = redo ? alt[i] : cum[i];
    }
#endif
  }

  template <int OFF, bool INV, int NV>
  __device__ __forceinline__ static void evaluate(const float (&v)[NV], float x, const NsaTransformConsts& c,
                                                  const float (&cw)[K + 1], const float (&ch)[K + 1], float& y,
                                                  float& ld, bool& bad) {
#pragma clang fp contract(off)
    const float B = c.tail_bound;
    float raw[K + 1];
    // derivative logits per knot; the boundary knots carry the logit of derivative 1
    raw[0] = c.edge_raw;
    raw[K] = c.edge_raw;
#pragma unroll
    for (int i = 1; i < K; ++i) raw[i] = v[OFF + 2 * K + i - 1];
    const bool inside = (x >= -B) && (x <= B);
    const float xin = inside ? x : 0.0f;
    float o[6];  // searched axis k, k+1; other axis k, k+1; derivative logits k, k+1
    // PAD: knot k_real (= B) is an INTERIOR entry of the searched table when k_real < K, so an input exactly equal to
    // +B would pass `x >= a[k_real]` into the empty bin behind it (width inf, 1 / width 0: NaN) where nflows and the
    // exact instances keep it in the last real bin (y = B, log|J| = 0).  The search - compares only - takes the
    // largest float below B for such an input; the arithmetic keeps xin.
    float xkey = xin;
    if constexpr (PAD) xkey = fminf(xin, __uint_as_float(__float_as_uint(B) - 1u));
    if constexpr (INV) NsaBinSelect<K>::run(ch, cw, raw, xkey, o);
    else NsaBinSelect<K>::run(cw, ch, raw, xkey, o);
    const float in_cw = INV ? o[2] : o[0], in_cw1 = INV ? o[3] : o[1];
    const float in_ch = INV ? o[0] : o[2], in_ch1 = INV ? o[1] : o[3];
    // only the two derivatives that are used are evaluated
    const float in_d = c.min_d + nsa_softplus_fast(o[4]);
    const float in_d1 = c.min_d + nsa_softplus_fast(o[5]);
    // Every multiply-add below is an explicit fmaf and nothing else is contracted (pragma at the top of the function):
    // left to -ffp-contract=fast, ONE of the four copies of this code in the two-tiles instance (skewed / plain wave x
    // first / second tile) fused the log-det differently from the others in round 5, and a quarter of the rows came
    // back with a log q one ulp away from the other launch geometries' (test_launch_geometry_..._never_changes_the_result).
    const float in_w = in_cw1 - in_cw;
    const float in_h = in_ch1 - in_ch;
    const float inv_w = nsa_rcp(in_w);
    const float delta = in_h * inv_w;
    const float curv = fmaf(-2.0f, delta, in_d + in_d1);
    float out, dnum, den;
    if constexpr (INV) {
      const float sh = xin - in_ch;
      const float a = fmaf(sh, curv, in_h * (delta - in_d));
      const float b = fmaf(-sh, curv, in_h * in_d);
      const float cc = -delta * sh;
      const float disc = fmaf(-4.0f * a, cc, b * b);
      bad = inside && !(disc >= 0.0f);
      const float root = (2.0f * cc) * nsa_rcp(-b - __builtin_amdgcn_sqrtf(disc));
      out = fmaf(root, in_w, in_cw);
      const float tomt = root * (1.0f - root);
      const float omr = 1.0f - root;
      den = fmaf(curv, tomt, delta);
      dnum = (delta * delta) * fmaf(in_d, omr * omr, fmaf(in_d1, root * root, (2.0f * delta) * tomt));
    } else {
      bad = false;
      const float theta = (xin - in_cw) * inv_w;
      const float tomt = theta * (1.0f - theta);
      const float omt = 1.0f - theta;
      den = fmaf(curv, tomt, delta);
      out = fmaf(in_h * fmaf(delta, theta * theta, in_d * tomt), nsa_rcp(den), in_ch);
      dnum = (delta * delta) * fmaf(in_d, omt * omt, fmaf(in_d1, theta * theta, (2.0f * delta) * tomt));
    }
    // log(dnum) - 2 log(den) on the hardware log2
    float lad = 0.6931471805599453f * fmaf(-2.0f, nsa_log2(den), nsa_log2(dnum));
    lad = INV ? -lad : lad;
    y = inside ? out : x;
    ld = inside ? lad : 0.0f;
  }

  template <int P, int OFF, bool INV, int NV>
  __device__ __forceinline__ static void phase(State&, const float (&v)[NV], float x,
                                               const NsaTransformConsts& c, float& y, float& ld,
                                               bool& bad) {
#pragma clang fp contract(off)
    float cw[K + 1], ch[K + 1];
    knots_checked<OFF, NV>(v, c, c.min_bw, c.bw_scale, cw);
    knots_checked<OFF + K, NV>(v, c, c.min_bh, c.bh_scale, ch);
    evaluate<OFF, INV, NV>(v, x, c, cw, ch, y, ld, bad);
  }
};

// Affine coupling: scale = sigmoid(raw + a) + b; a = 2, b = 1e-3 is the nflows default
// scale_activation, other (a, b) come from a user callable of that family
// (flows/realnvp.py).  v[OFF] = shift, v[OFF+1] = raw scale.
//
// With e = exp(-(raw + a)):  scale = (1 + b + b e) / (1 + e),  1 / scale = (1 + e) / (1 + b + b e)  - ONE
// reciprocal for the direction at hand (the plain form needs two in the inverse direction: of 1 + e and of
// the scale), on the bare v_exp_f32 / v_rcp_f32; and log|det J| = sum log(scale) is taken as the log of a
// PRODUCT of the factors a lane holds (LD_FACTOR: the caller multiplies the `ld` of up to four features -
// >= b^4; for floors b below 1e-9, a plain sigmoid included, it takes one logarithm per feature instead:
// NsaTransformConsts::ld_group4 - and takes one v_log_f32), since a transcendental costs 2.7 plain vector
// instructions here (scripts/ubench/valu_rates).  e is capped at 2^64, where sigmoid is 0 to
// float32 beside b (at 2^126 for floors below 1e-9, a plain sigmoid included: the scale then follows the sigmoid
// down to the smallest normal number - NsaTransformConsts::aff_cap); the cap is a compare + select, not a min: a
// NaN stays a NaN.
struct NsaAffine {
  static constexpr int NP = 2;
  static constexpr int FPC = 8;
  static constexpr int C = 1;
  static constexpr int KIND = NSA_T_AFFINE;
  static constexpr int NPH = 1;
  static constexpr bool LD_FACTOR = true;  // `ld` is the Jacobian FACTOR of the feature, not its logarithm
  struct State {};

  template <int P, int OFF, bool INV, int NV>
  __device__ __forceinline__ static void phase(State&, const float (&v)[NV], float x,
                                               const NsaTransformConsts& c, float& y, float& ld,
                                               bool& bad) {
    const float shift = v[OFF];
    float arg = fmaf(v[OFF + 1], -1.4426950408889634f, c.aff_arg0);  // -(raw + a) log2(e)
    arg = arg > c.aff_cap ? c.aff_cap : arg;
    const float e = __builtin_amdgcn_exp2f(arg);
    const float num = 1.0f + e, den = fmaf(c.aff_b, e, c.aff_1pb);
    bad = false;
    if constexpr (INV) {
      const float inv_scale = num * nsa_rcp(den);
      y = (x - shift) * inv_scale;
      ld = inv_scale;
    } else {
      const float scale = den * nsa_rcp(num);
      y = fmaf(x, scale, shift);
      ld = scale;
    }
  }
};

// Additive (volume preserving) coupling: shift only.
struct NsaAdditive {
  static constexpr int NP = 1;
  static constexpr int FPC = 16;
  static constexpr int C = 1;
  static constexpr int KIND = NSA_T_ADDITIVE;
  static constexpr int NPH = 1;
  static constexpr bool LD_FACTOR = false;
  struct State {};

  template <int P, int OFF, bool INV, int NV>
  __device__ __forceinline__ static void phase(State&, const float (&v)[NV], float x,
                                               const NsaTransformConsts&, float& y, float& ld,
                                               bool& bad) {
    bad = false;
    ld = 0.0f;
    y = INV ? (x - v[OFF]) : (x + v[OFF]);  // scale == 1: exact
  }
};
