// Third generation of the fused training kernels (included by nsa_train.hip after nsa_train_fused2.h; same TrfArgs,
// same workspace layout, same results up to the order of the float32 sums): conditioners 68 to 128 wide, in steps
// of 4 (config 4's importance-sampling flows: 64 inputs, 128-wide residual nets, 736 spline parameters per row; the
// reference's default n_neurons = 2 n_inputs for 34 to 64 inputs), one coupling layer per launch, forward and
// backward, on 16-ROW workgroups and v_mfma_f32_16x16x4_f32 tiles - config 4's optimiser step 1.24 -> 0.63 ms.
// The 128-wide instances have every size at compile time; the others take the width at run time and skip the MFMA
// steps and tiles beyond it (a last tile may be partial: its lanes beyond the width write nothing, and the tiles'
// columns from the width on are cleared once, so that steps reaching past it multiply zeros).
//
// What the first generation spends its time on at this width (profiles/r04_train_c4_kernels.txt: forward 65 us,
// backward 118 us per coupling for a 32-row workgroup whose MFMAs take 20 / 40 us): a hidden layer has four output
// tiles, so four of the eight waves own one and four idle; every wave requests 4 KB of weights per step and waits a
// round trip (~2300 cycles: the flow's 4 MB of weights do not stay in the 4 MB L2 of an XCD across a step) for them;
// the final layer's gradient travels through global memory in chunks, each a round trip and two barriers, and its
// weight gradient is added with device-scope float atomics from every workgroup.
//
// A minibatch of 1000 rows is 32 workgroups of 32 rows on a 256-CU chip, and what bounds a 32-row workgroup - its
// f32 MFMAs (40 us per coupling at the peak of ONE CU), the spline's reverse mode (26 k cycles of VALU work), the
// glue between the barriers of a split-k layer - is all per workgroup: a 32-row kernel pair on the second generation's
// split-k scheme (all 8 waves in every layer, operands a layer ahead, parameter tile in LDS, gradient copies) measured
// 0.90 ms per step (git history, commit 6e8af15), and a 32-row forward kernel for the widths between 65 and 127 no
// gain at all over the first generation (0.91 / 0.87 ms at 96 wide, 1.07 / 1.08 at 100; profiles/r06_train_configs.jsonl).
// With 16 rows a workgroup has half the matrix and vector work (63 CUs busy instead of 32), and a wave owns a whole
// 16 x 16 output tile over the full k: no partial sums, no reduction, ONE barrier per layer, bias / residual / act'
// applied to the accumulators in registers.
//   A[i][k]: lane i + 16 k (i < 16 rows, k < 4)     B[k][j]: lane j + 16 k     D[i][j]: lane j + 16 (i / 4), register i % 4
// LDS tiles [16][128] have row stride 132 (4 mod 64) where they are read as A operands (row = lane % 16, k = lane / 16:
// 64 banks) and 144 (16 mod 64) where they are read transposed for a weight gradient (row = lane / 16, column = lane % 16).
// Reference semantics: glasflow.nflows ResidualNet (call sites src/nessai/flows/nsf.py:85-95, realnvp.py:136-146)
// under FlowModel._train (src/nessai/flowmodel/base.py:365-452; the importance sampler's
// src/nessai/flowmodel/importance.py:80-142).
#pragma once

#define TRF4_ROWS 16
#define TRF4_LD 132
#define TRF4_LDA 144
#define TRF4_Q 32  // MFMA steps over k = 128

__device__ __forceinline__ f32x4v trf4_mfma(float a, float b, f32x4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// The operand streams go through buffer loads: ONE descriptor over the flat parameter vector (and one over its
// transposed copy), the lane's part of an address in a 32-bit VGPR offset, everything else - matrix, k step - in the
// scalar offset.  As flat loads each of the 32 requests of a layer carried its own 64-bit address (two carry-chained
// VALU adds and their wait states): 2.4 k cycles per layer went into ISSUING a prefetch.
#define TRF4_RSRC(ptr) __builtin_amdgcn_make_buffer_rsrc((void*)(ptr), 0, 0x7ffffff0, 0x00020000)
template <typename R>
__device__ __forceinline__ float trf4_bload(R rsrc, unsigned voff_bytes, unsigned soff_bytes) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff_bytes, soff_bytes, 0));
}
template <typename R>
__device__ __forceinline__ f32x4v trf4_bload4(R rsrc, unsigned voff_bytes, unsigned soff_bytes) {
  return __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff_bytes, soff_bytes, 0));
}

// B operands of the wave's output tile (columns 16 wave + lane % 16) of a layer with 128 outputs from the transposed
// weights W^T[Kd][128] at flat offset offW: step q holds k = 4 q + lane / 16.  FULL: Kd == 128.
// H: the layer's width (a multiple of 4, at most 128; steps beyond Kd = H re-read the last one, columns beyond H read on
// into the next row - finite values no MFMA result that is kept depends on).
template <bool FULL, typename RT, typename RP>
__device__ __forceinline__ void trf4_load_w(float (&w)[TRF4_Q], float& bias, RT rsrcT, RP rsrcP, long long offW,
                                            long long offB, int Kd, int H, int wave, int lane) {
  const int i16 = lane & 15, kg = lane >> 4;
  const unsigned s0 = 4u * ((unsigned)offW + 16u * (unsigned)wave);
  if constexpr (FULL) {
    const unsigned voff = 4u * ((unsigned)kg * (unsigned)H + (unsigned)i16);
    const unsigned step = 16u * (unsigned)H;
    const int nq = (Kd + 3) >> 2;  // (steps beyond the matrix's Kd rows re-read its last one: nothing is read past the flat vector)
#pragma unroll
    for (int q = 0; q < TRF4_Q; ++q) w[q] = trf4_bload(rsrcT, voff, s0 + step * (unsigned)(q < nq ? q : nq - 1));
  } else {
#pragma unroll
    for (int q = 0; q < TRF4_Q; ++q) {
      const int k = 4 * q + kg;
      w[q] = trf4_bload(rsrcT, 4u * ((unsigned)(k < Kd ? k : Kd - 1) * (unsigned)H + (unsigned)i16), s0);
    }
  }
  bias = trf4_bload(rsrcP, 4u * (unsigned)i16, 4u * ((unsigned)offB + 16u * (unsigned)wave));
  __builtin_amdgcn_sched_barrier(0);  // (left alone, the scheduler sinks these loads to their first use: a round trip per layer)
}

// out[row][col] = sum_k act(Ain[row][k]) W[col][k] + b[col] (+ R[row][col]) for the wave's 16 columns of the
// workgroup's 16 rows; the tile is WRITTEN, not yet published.  GATHER: A from the x tile, column of step q in cidx[q].
// HH: 128 (the steps and tiles are all there: no guards) or 0 (H at run time, a multiple of 4: steps 4 q >= Kd and
// tiles beyond H skipped; the A operands of k >= H are the zeros the kernel left in the tiles' padding columns).
template <bool GATHER, int HH>
__device__ __forceinline__ void trf4_layer(const float* __restrict__ Ain, int act_in, const float (&w)[TRF4_Q],
                                           const int (&cidx)[TRF4_Q], float bias, int Kd, int H, const float* R,
                                           float* out_tile, float* __restrict__ out_glb, int r0, int rows_here,
                                           int wave, int lane) {
  const int i16 = lane & 15, kg = lane >> 4;
  if (HH == 0 && 16 * wave >= H) return;  // wave-uniform
  float av[TRF4_Q], rv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  f32x4v acc = {0}, acc2 = {0};
  if constexpr (GATHER) {
#pragma unroll
    for (int q = 0; q < TRF4_Q; ++q) av[q] = Ain[cidx[q]];
#pragma unroll
    for (int q = 0; q < TRF4_Q; ++q) av[q] *= (4 * q + kg < Kd) ? 1.0f : 0.0f;
#pragma unroll
    for (int q = 0; q < TRF4_Q; q += 2) {
      if (4 * q < Kd) {  // uniform
        acc = trf4_mfma(av[q], w[q], acc);
        acc2 = trf4_mfma(av[q + 1], w[q + 1], acc2);
      }
    }
  } else {
    const float* ax = Ain + i16 * TRF4_LD + kg;
#pragma unroll
    for (int q = 0; q < TRF4_Q; ++q) av[q] = ax[4 * q];
    if (R) {
#pragma unroll
      for (int r = 0; r < 4; ++r) rv[r] = R[(4 * kg + r) * TRF4_LD + 16 * wave + i16];
    }
    __builtin_amdgcn_sched_barrier(0);  // all the LDS reads in front of the first MFMA (one read per pair of MFMAs
                                        // otherwise, each pair waiting for its own)
    trf_act_batch(av, act_in);
#pragma unroll
    for (int q = 0; q < TRF4_Q; q += 2) {
      if (HH != 0 || 4 * q < Kd) {  // uniform (a last pair may reach past Kd: zeros of the padding columns)
        acc = trf4_mfma(av[q], w[q], acc);
        acc2 = trf4_mfma(av[q + 1], w[q + 1], acc2);
      }
    }
  }
  const int col = 16 * wave + i16;
  if (HH != 0 || col < H) {  // (a width that is not a multiple of 16: the last tile's lanes beyond it write nothing -
                             // the tiles' columns from H on stay zero, as the kernels cleared them)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * kg + r;
      const float v = acc[r] + acc2[r] + bias + rv[r];
      out_tile[row * TRF4_LD + col] = v;
      if (row < rows_here) out_glb[(unsigned)(r0 + row) * (unsigned)H + (unsigned)col] = v;
    }
  }
}

static inline size_t trf4_forward_floats(const TrfArgs& a) {
  return (size_t)TRF4_ROWS * trf2_ldx(a) + (size_t)2 * TRF4_ROWS * TRF4_LD + (size_t)TRF4_ROWS * a.np + TRF4_ROWS;
}

template <int K, int HH>  // spline bins (0: affine / additive); hidden width 128, or 0: a.H (68 ... 124, a multiple of 4)
__global__ void __launch_bounds__(TRF_THREADS) trf4_forward(TrfArgs a) {
  extern __shared__ float trf_sm[];
  const int tid = threadIdx.x, lane = tid & 63, i16 = lane & 15, kg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r0 = blockIdx.x * TRF4_ROWS;
  const int H = HH != 0 ? HH : a.H;
  const int n = a.n, D = a.D, L = a.L, np = a.np, d_id = a.d_id, d_tr = a.d_tr;
  const int ldx = (D + 1) | 1;
  const int rows_here = n - r0 < TRF4_ROWS ? n - r0 : TRF4_ROWS;
  float* XT = trf_sm;                      // [16][ldx]
  float* HC = XT + TRF4_ROWS * ldx;        // h_l
  float* U = HC + TRF4_ROWS * TRF4_LD;     // u_l
  float* PT = U + TRF4_ROWS * TRF4_LD;     // [16][np] the transform's parameters
  float* ldsum = PT + TRF4_ROWS * np;      // [16] log|det J| per row
  const float* P = a.params;
  const auto rsrcP = TRF4_RSRC(a.params);
  const auto rsrcT = TRF4_RSRC(a.wt);
  const long long nH = (long long)n * H;
  const long long t_start = clock64();
  if (tid < TRF4_ROWS) ldsum[tid] = 0.0f;
  if constexpr (HH == 0) {  // the columns from H on of h_l / u_l: zero A operands for the steps that reach past H
    for (int i = tid; i < 2 * TRF4_ROWS * TRF4_LD; i += TRF_THREADS) HC[i] = 0.0f;
  }

  // the rows of x (async, 4 bytes per lane, rows wave and wave + 8; rows beyond n read row n - 1)
  for (int r = wave; r < TRF4_ROWS; r += TRF_WAVES) {
    const float* g = a.X + (long long)(r0 + r < n ? r0 + r : n - 1) * D;
    for (int c0 = 0; c0 < D; c0 += 64) {
      if (c0 + lane < D)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + c0 + lane),
                                         (__attribute__((address_space(3))) void*)(XT + r * ldx + c0), 4, 0, 0);
    }
  }
  int cidx[TRF4_Q], none[TRF4_Q];
  float wc[TRF4_Q], wn[TRF4_Q], bc, bn = 0.0f;
#pragma unroll
  for (int q = 0; q < TRF4_Q; ++q) {
    const int k = 4 * q + kg;
    cidx[q] = a.id_cols[k < d_id ? k : 0] + i16 * ldx;
    none[q] = 0;
  }
  trf4_load_w<false>(wc, bc, rsrcT, rsrcP, a.off[0], a.off[1], d_id, H, wave, lane);
  // the final layer: output tiles wave + 8 s (16 columns each), whole k per tile, two operand buffers
  const float* bf = P + a.off[3 + 4 * L];
  auto load_final = [&](float (&buf)[TRF4_Q], int s) {
    const int o = (wave + TRF_WAVES * s) * 16 + i16;
    const unsigned voff = 4u * ((unsigned)kg * (unsigned)np + (unsigned)(o < np ? o : np - 1));
    const unsigned s0 = 4u * (unsigned)a.off[2 + 4 * L], step = 16u * (unsigned)np;
    const int nqh = (H + 3) >> 2;
#pragma unroll
    for (int q = 0; q < TRF4_Q; ++q) buf[q] = trf4_bload(rsrcT, voff, s0 + step * (unsigned)(q < nqh ? q : nqh - 1));
    __builtin_amdgcn_sched_barrier(0);
  };
  float bfv[6];  // (loaded inside the rounds, a bias behind the refill loads made every round wait for them)
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    const int o = (wave + TRF_WAVES * s) * 16 + i16;
    bfv[s] = bf[o < np ? o : np - 1];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the async rows (and layer 0)
  trf_barrier_lds();
  TRF_MARK(0);

  float fb[TRF4_Q];
  trf4_load_w<true>(wn, bn, rsrcT, rsrcP, a.off[L > 0 ? 2 : 0], a.off[L > 0 ? 3 : 1], L > 0 ? H : d_id, H, wave, lane);
  // h_0 = W0 x_id + b0
  trf4_layer<true, HH>(XT, ACT_NONE, wc, cidx, bc, d_id, H, nullptr, HC, a.h, r0, rows_here, wave, lane);
  TRF_MARK(1);
  for (int l = 0; l < L; ++l) {
    // u_l = W1 act(h_l) + b1;  h_{l+1} = h_l + W2 act(u_l) + b2
#pragma unroll
    for (int q = 0; q < TRF4_Q; ++q) wc[q] = wn[q];
    bc = bn;
    trf4_load_w<true>(wn, bn, rsrcT, rsrcP, a.off[4 + 4 * l], a.off[5 + 4 * l], H, H, wave, lane);
    trf_barrier_lds();
    trf4_layer<false, HH>(HC, a.act, wc, none, bc, H, H, nullptr, U, a.u + l * nH, r0, rows_here, wave, lane);
    TRF_MARK(2 + 2 * l);
#pragma unroll
    for (int q = 0; q < TRF4_Q; ++q) wc[q] = wn[q];
    bc = bn;
    {
      // (no branch round a prefetch: where paths with and without loads in flight meet, the compiler's wait counts
      // fall back to "everything" - and the next use of ANY prefetched operand waits for the newest request.  The
      // last block requests its own second layer again instead.)
      const int ln = l + 1 < L ? 6 + 4 * l : 4 + 4 * l;
      trf4_load_w<true>(wn, bn, rsrcT, rsrcP, a.off[ln], a.off[ln + 1], H, H, wave, lane);
    }
    trf_barrier_lds();
    trf4_layer<false, HH>(U, a.act, wc, none, bc, H, H, HC, HC, a.h + (l + 1) * nH, r0, rows_here, wave, lane);
    TRF_MARK(3 + 2 * l);
  }
  load_final(wn, 0);
  load_final(fb, 1);
  trf_barrier_lds();  // h_L published
  // parameters = Wf h_L + bf
  const bool p_vec = (np & 3) == 0 && (((unsigned long long)a.p) & 15) == 0;
  {
    float av[TRF4_Q];
    const float* ax = HC + i16 * TRF4_LD + kg;
#pragma unroll
    for (int q = 0; q < TRF4_Q; ++q) av[q] = ax[4 * q];
    __builtin_amdgcn_sched_barrier(0);
    trf2_static_for<6>([&](auto si) {
      constexpr int s = decltype(si)::value;
      float (&buf)[TRF4_Q] = (s & 1) ? fb : wn;
      const int t = wave + TRF_WAVES * s;
      {
        // (every wave runs every round, on clamped columns where its tile is beyond np: see the hidden layers)
        f32x4v acc = {0}, acc2 = {0};
#pragma unroll
        for (int q = 0; q < TRF4_Q; q += 2) {
          if (HH != 0 || 4 * q < H) {  // uniform
            acc = trf4_mfma(av[q], buf[q], acc);
            acc2 = trf4_mfma(av[q + 1], buf[q + 1], acc2);
          }
        }
        if (s + 2 < 6) load_final(buf, s + 2);
        const int o = t * 16 + i16;
        if (o < np) {
          const float bias = bfv[s];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 4 * kg + r;
            const float v = acc[r] + acc2[r] + bias;
            PT[row * np + o] = v;
            if (!p_vec && row < rows_here) a.p[(unsigned)(r0 + row) * (unsigned)np + (unsigned)o] = v;
          }
        }
      }
    });
  }
  TRF_MARK(20);
  trf_barrier_lds();
  TRF_MARK(21);
  if (p_vec) {
    const int quads = rows_here * (np >> 2);
    float* dst = a.p + (long long)r0 * np;
    for (int i = tid; i < quads; i += TRF_THREADS) *(f32x4v*)(dst + 4 * i) = *(const f32x4v*)(PT + 4 * i);
  }
  const int items = TRF4_ROWS * d_tr;
  for (int e = tid; e < items; e += TRF_THREADS) {
    const int r = e / d_tr, mi = e - r * d_tr;
    const int row = r0 + r;
    if (row >= n) continue;
    const int col = a.tr_cols[mi];
    const float x = XT[r * ldx + col];
    float y, ldj;
    if constexpr (K > 0) {
      tr_rq_forward_core<K>(PT + r * np + mi * (3 * K - 1), x, a.c, y, ldj);
    } else {
      const float* pr = PT + r * np;
      float scale = 1.0f;
      ldj = 0.0f;
      if (a.transform != NSA_T_ADDITIVE) {
        scale = 1.0f / (1.0f + expf(-(pr[d_tr + mi] + a.c.aff_a))) + a.c.aff_b;
        ldj = logf(scale);
      }
      y = x * scale + pr[mi];
    }
    a.xsave[row * d_tr + mi] = x;
    a.X[(long long)row * D + col] = y;
    if (K > 0 || a.transform != NSA_T_ADDITIVE) atomicAdd(&ldsum[r], ldj);
  }
  TRF_MARK(22);
  trf_barrier_lds();
  if (tid < TRF4_ROWS && r0 + tid < n) a.logdet[r0 + tid] += ldsum[tid];
}

// ---- backward pass, 16 rows per workgroup.  LDS (floats):
//   PT [16][pld] parameters -> their gradient | ACT [2 L + 1][16][144]: h_0 .. h_L, u_0 .. u_{L-1} | XT [16][ldx]
//   | G0, G1 [16][132] - everything the pass reads, resident from the first barrier on.
// row stride of the parameter tile: rows of 16-byte units, an odd number of them (16-byte reads without bank
// conflicts: trf2_pld), where the rows are read in 16-byte units (PV); otherwise 4 mod 64 (the tile is then read as an
// A operand - row = lane % 16, k = lane / 16 - like the activation tiles)
__host__ __device__ static inline int trf4_pld(int np, bool pv) {
  // (element-wise tile: the final layer's dX walks it in groups of 16 parameters - zeros up to the end of the last group)
  return pv ? trf2_pld(np, true) : ((((np + 15) & ~15) + 59) / 64) * 64 + 4;
}
static inline size_t trf4_backward_floats(const TrfArgs& a, bool pv) {
  return (size_t)TRF4_ROWS * trf4_pld(a.np, pv) + (size_t)(2 * a.L + 1) * TRF4_ROWS * TRF4_LDA +
         (size_t)TRF4_ROWS * trf2_ldx(a) + (size_t)2 * TRF4_ROWS * TRF4_LD;
}

// B operands of the wave's dX tile (input columns 16 wave + lane % 16) of a layer with 128 outputs: W[out = 4 q +
// lane / 16][in], natural layout [128][Kin] at flat offset offW
template <typename RP>
__device__ __forceinline__ void trf4_load_wx(float (&w)[TRF4_Q], RP rsrcP, long long offW, int Kin, int H, int wave,
                                             int lane) {
  const int i16 = lane & 15, kg = lane >> 4;
  const int c = 16 * wave + i16;
  const unsigned voff = 4u * ((unsigned)kg * (unsigned)Kin + (unsigned)(c < Kin ? c : Kin - 1));
  const unsigned s0 = 4u * (unsigned)offW, step = 16u * (unsigned)Kin;
  const int nq = (H + 3) >> 2;  // (the matrix has H rows: steps beyond re-read the last one)
#pragma unroll
  for (int q = 0; q < TRF4_Q; ++q) w[q] = trf4_bload(rsrcP, voff, s0 + step * (unsigned)(q < nq ? q : nq - 1));
  __builtin_amdgcn_sched_barrier(0);  // (left alone, the scheduler sinks these loads to their first use: a round trip per layer)
}

// a weight-gradient tile out of D[i = in][j = out]: the lane owns inputs in0 + 4 (lane / 16) .. + 3 of ITS output
// row o - one 16-byte store where the rows allow it.  gW: [O][Kin] row-major.
template <bool SLAB>
__device__ __forceinline__ void trf4_dw_store(float* __restrict__ gW, int O, int Kin, int o, int in0, int kg,
                                              const f32x4v& acc, bool vec) {
  if (o >= O) return;
  const int c = in0 + 4 * kg;
  float* dst = gW + (unsigned)o * (unsigned)Kin + (unsigned)c;
  if (vec) {
    if (c < Kin) *(f32x4v*)dst = acc;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c + e < Kin) trf2_grad_out<SLAB>(dst + e, acc[e]);
  }
}

// One linear layer (128 outputs) of the backward pass on 16 rows:
//   dX  the wave's 16 input columns over all 128 outputs, act'(z) and the residual path applied to the accumulators;
//   dW  output tile = wave, every input tile: A = act(Ain)^T, B = G, four MFMA steps (16 rows) per tile;
//   db  column sums of G out of the B operands' registers.
//   G [16][132];  Ain: row stride a_ld (ACT tiles: 144; GATHER: the x tile, column of slot c = cols[c]);  Kin inputs
//   Z [16][144]: pre-activation whose act' multiplies dX (null: none);  Rs [16][132] added to the result (null: none)
// H: the layer's number of outputs (HH: 128, or 0 - at run time: MFMA steps and tiles beyond it skipped).
template <bool GATHER, bool SLAB, int HH>
__device__ __forceinline__ void trf4_back_linear(const float* __restrict__ G, const float (&w)[TRF4_Q],
                                                 const float* __restrict__ Ain, int a_ld, const int* __restrict__ cols,
                                                 int act_in, int Kin, int H, const float* __restrict__ Z, int act_z,
                                                 const float* Rs, float* out, float* __restrict__ gW,
                                                 float* __restrict__ gb, int wave, int lane) {
  const int i16 = lane & 15, kg = lane >> 4;
  const int o = 16 * wave + i16;  // dW, db: output columns 16 wave + lane % 16
  const bool tile_on = HH != 0 || 16 * wave < H;  // wave-uniform: this wave's 16 columns exist
  if constexpr (!GATHER) {
    if (!tile_on) return;
    // Kin == H.  Every LDS read of the layer in front of its first MFMA - the dX operands, the pre-activations
    // and the residual tile of the epilogue, the dW operands: one LDS latency per layer (read where they are used,
    // the epilogue's eight reads were a chain of 2.2 k cycles and the dW tiles' another 1.8 k)
    const bool vec = SLAB && (((unsigned long long)gW) & 15) == 0;  // uniform
    float av[TRF4_Q], zv[4], rv[4] = {0.0f, 0.0f, 0.0f, 0.0f}, bvv[4], at[8][4];
    const float* ag = G + i16 * TRF4_LD + kg;
#pragma unroll
    for (int q = 0; q < TRF4_Q; ++q) av[q] = ag[4 * q];
#pragma unroll
    for (int r = 0; r < 4; ++r) zv[r] = Z[(4 * kg + r) * TRF4_LDA + o];
    if (Rs) {
#pragma unroll
      for (int r = 0; r < 4; ++r) rv[r] = Rs[(4 * kg + r) * TRF4_LD + o];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) bvv[s] = G[(4 * s + kg) * TRF4_LD + o];
#pragma unroll
    for (int it = 0; it < 8; ++it)
#pragma unroll
      for (int s = 0; s < 4; ++s) at[it][s] = Ain[(4 * s + kg) * a_ld + 16 * it + i16];
    __builtin_amdgcn_sched_barrier(0);
    f32x4v acc = {0}, acc2 = {0};
#pragma unroll
    for (int q = 0; q < TRF4_Q; q += 2) {
      if (HH != 0 || 4 * q < H) {  // uniform
        acc = trf4_mfma(av[q], w[q], acc);
        acc2 = trf4_mfma(av[q + 1], w[q + 1], acc2);
      }
    }
    const bool lane_on = HH != 0 || o < H;  // (the last tile of a width that is not a multiple of 16)
    {  // db under the MFMAs
      float cs = (bvv[0] + bvv[1]) + (bvv[2] + bvv[3]);
      cs += __shfl_xor(cs, 16);
      cs += __shfl_xor(cs, 32);
      if (kg == 0 && lane_on) trf2_grad_out<SLAB>(&gb[o], cs);
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) trf_act_batch(at[it], act_in);
    if (!lane_on) {
    } else if (act_z == NSA_ACT_RELU) {
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(4 * kg + r) * TRF4_LD + o] = (zv[r] > 0.0f ? acc[r] + acc2[r] : 0.0f) + rv[r];
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        out[(4 * kg + r) * TRF4_LD + o] = (acc[r] + acc2[r]) * tr_act_grad(zv[r], act_z) + rv[r];
    }
#pragma unroll
    for (int it = 0; it < 8; it += 2) {
      if (HH != 0 || 16 * it < Kin) {  // uniform (Kin / 16 input tiles)
        f32x4v d0 = {0}, d1 = {0};
#pragma unroll
        for (int s = 0; s < 4; ++s) {  // D[i = in][j = out], two tiles interleaved
          d0 = trf4_mfma(at[it][s], bvv[s], d0);
          d1 = trf4_mfma(at[it + 1][s], bvv[s], d1);
        }
        trf4_dw_store<SLAB>(gW, H, Kin, o, 16 * it, kg, d0, vec);
        if (HH != 0 || 16 * (it + 1) < Kin) trf4_dw_store<SLAB>(gW, H, Kin, o, 16 * (it + 1), kg, d1, vec);
      }
    }
    return;
  }
  // the initial layer (GATHER): Kin = d_id inputs read out of the x tile
  if (16 * wave < Kin) {  // wave-uniform
    float av[TRF4_Q];
    const float* ag = G + i16 * TRF4_LD + kg;
#pragma unroll
    for (int q = 0; q < TRF4_Q; ++q) av[q] = ag[4 * q];
    __builtin_amdgcn_sched_barrier(0);
    f32x4v acc = {0}, acc2 = {0};
#pragma unroll
    for (int q = 0; q < TRF4_Q; q += 2) {
      if (HH != 0 || 4 * q < H) {  // uniform
        acc = trf4_mfma(av[q], w[q], acc);
        acc2 = trf4_mfma(av[q + 1], w[q + 1], acc2);
      }
    }
    const int c = 16 * wave + i16;
    if (c < Kin) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * kg + r;
        float v = acc[r] + acc2[r];
        if (Z) {
          const float z = Z[row * TRF4_LDA + c];
          if (act_z == NSA_ACT_RELU) v = z > 0.0f ? v : 0.0f;
          else v *= tr_act_grad(z, act_z);
        }
        if (Rs) v += Rs[row * TRF4_LD + c];
        out[row * TRF4_LD + c] = v;
      }
    }
  }
  if (!tile_on) return;
  float bvv[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) bvv[s] = G[(4 * s + kg) * TRF4_LD + o];
  {
    float cs = (bvv[0] + bvv[1]) + (bvv[2] + bvv[3]);
    cs += __shfl_xor(cs, 16);
    cs += __shfl_xor(cs, 32);
    if (kg == 0 && (HH != 0 || o < H)) trf2_grad_out<SLAB>(&gb[o], cs);
  }
  const bool vec = SLAB && (Kin & 3) == 0 && (((unsigned long long)gW) & 15) == 0;  // uniform
  const int NI = (Kin + 15) >> 4;  // <= 8
#pragma unroll
  for (int it = 0; it < 8; it += 2) {
    if (it < NI) {  // uniform
      float at[2][4];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int kc = 16 * (it + e) + i16;
        const int kcl = kc < Kin ? kc : 0;
        const int acol = GATHER ? cols[kcl] : kcl;
#pragma unroll
        for (int s = 0; s < 4; ++s) at[e][s] = Ain[(4 * s + kg) * a_ld + acol];
        trf_act_batch(at[e], act_in);
      }
      f32x4v d0 = {0}, d1 = {0};
#pragma unroll
      for (int s = 0; s < 4; ++s) {  // D[i = in][j = out], two tiles interleaved
        d0 = trf4_mfma(at[0][s], bvv[s], d0);
        d1 = trf4_mfma(at[1][s], bvv[s], d1);
      }
      trf4_dw_store<SLAB>(gW, H, Kin, o, 16 * it, kg, d0, vec);
      if (it + 1 < NI) trf4_dw_store<SLAB>(gW, H, Kin, o, 16 * (it + 1), kg, d1, vec);
    }
  }
}

// PV: the parameter rows are 16-byte multiples (np % 4 == 0: 16-byte loads of the tile, the final layer's dX on
// 16-byte LDS / transposed-weight operands); otherwise the tile is filled element by element and that product runs on
// the natural weight layout like a hidden layer's (23 parameters per feature: np % 4 != 0 unless d_tr % 4 == 0)
template <int K, bool SLAB, int HH, bool PV>  // HH: 128, or 0 - the hidden width a.H (68 ... 128, a multiple of 4) at run time
__global__ void __launch_bounds__(TRF_THREADS) trf4_backward(TrfArgs a) {
  extern __shared__ float trf_sm[];
  const int tid = threadIdx.x, lane = tid & 63, i16 = lane & 15, kg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r0 = blockIdx.x * TRF4_ROWS;
  const int H = HH != 0 ? HH : a.H;
  const int n = a.n, np = a.np, D = a.D, L = a.L;
  const int ldx = (D + 1) | 1, pld = trf4_pld(np, PV);
  float* PT = trf_sm;                                        // [16][pld]
  float* ACT = PT + TRF4_ROWS * pld;                         // [2 L + 1][16][144]
  float* XT = ACT + (2 * L + 1) * TRF4_ROWS * TRF4_LDA;      // [16][ldx]
  float* G0 = XT + TRF4_ROWS * ldx;                          // [16][132]
  float* G1 = G0 + TRF4_ROWS * TRF4_LD;
  const auto rsrcP = TRF4_RSRC(a.params);
  const auto rsrcT = TRF4_RSRC(a.wt);
  float* GR = SLAB ? a.gslab + (long long)blockIdx.x * a.npar : a.grads;
  const long long nH = (long long)n * H;
  const long long t_start = clock64();
  const int rows_here = (n - r0 < TRF4_ROWS) ? (n - r0) : TRF4_ROWS;
  if constexpr (HH == 0) {  // the gradient tiles' columns from H on: zero A operands for the steps that reach past H
    for (int i = tid; i < 2 * TRF4_ROWS * TRF4_LD; i += TRF_THREADS) G0[i] = 0.0f;
  }

  // ---- everything the pass reads, 16-byte loads through registers: the parameter tile, the activation tiles
  // (one unit per thread each), the rows of x; and what the transform item of this thread needs ----
  // the transform item of this thread: (row, feature) = (tid / d_tr, tid % d_tr), d_tr <= 32.  Its column index is
  // the first load of the kernel: dL/dy is read through it, a second round trip that then runs beside the tiles'
  const int it_r = tid / a.d_tr, it_m = tid - it_r * a.d_tr;
  const bool it_on = tid < TRF4_ROWS * a.d_tr;
  const int colv = a.tr_cols[it_m];
  constexpr int PQ4 = 7;   // 16 (768 + 4) / 4 <= 7 * 512
  constexpr int PQS = 24;  // element by element: 16 * 768 <= 24 * 512
  f32x4v pq[PV ? PQ4 : 1];
  float ps[PV ? 1 : PQS];
  const int rq = pld >> 2, nq = np >> 2;
  if constexpr (PV) {
#pragma unroll
    for (int i = 0; i < PQ4; ++i) {
      if (TRF_THREADS * i >= TRF4_ROWS * rq) break;  // uniform
      int Q = tid + TRF_THREADS * i;
      Q = Q < TRF4_ROWS * rq ? Q : TRF4_ROWS * rq - 1;
      const int r = Q / rq, cq = Q - r * rq;
      const int row = r0 + r < n ? r0 + r : n - 1;
      pq[i] = *(const f32x4v*)(a.p + (long long)row * np + 4 * (cq < nq ? cq : nq - 1));
    }
  } else {
#pragma unroll
    for (int i = 0; i < PQS; ++i) {
      if (TRF_THREADS * i >= TRF4_ROWS * np) break;  // uniform
      int idx = tid + TRF_THREADS * i;
      idx = idx < TRF4_ROWS * np ? idx : TRF4_ROWS * np - 1;
      const int r = idx / np;
      const int row = r0 + r < n ? r0 + r : n - 1;
      ps[i] = a.p[(long long)row * np + (idx - r * np)];
    }
  }
  constexpr int MAXT = 7;  // 2 L + 1 <= 7
  f32x4v aq[MAXT];
  {
    // (the [16][H] tile is contiguous in global memory: unit tid of its 4 H units; threads beyond load unit 4 H - 1)
    long long g = (long long)r0 * H + 4 * (tid < 4 * H ? tid : 4 * H - 1);
    g = g < nH - 4 ? g : nH - 4;
    trf2_static_for<MAXT>([&](auto ti) {
      constexpr int t = decltype(ti)::value;
      const int tc = t < 2 * L + 1 ? t : 2 * L;
      const float* src = (tc <= L) ? a.h + tc * nH : a.u + (tc - L - 1) * nH;
      aq[t] = *(const f32x4v*)(src + g);
    });
  }
  constexpr int XQ = 4;  // x elements per thread: 16 D <= 4 * 512
  float xq[XQ];
#pragma unroll
  for (int i = 0; i < XQ; ++i) {
    if (TRF_THREADS * i >= TRF4_ROWS * D) break;  // uniform
    int idx = tid + TRF_THREADS * i;
    idx = idx < TRF4_ROWS * D ? idx : TRF4_ROWS * D - 1;
    const int r = idx / D;
    const int row = r0 + r < n ? r0 + r : n - 1;
    xq[i] = a.X[(long long)row * D + (idx - r * D)];
  }
  float xs, gldv, gyv;
  {
    const int row = (it_on && it_r < rows_here) ? r0 + it_r : n - 1;
    xs = a.xsave[row * a.d_tr + it_m];
    gldv = a.gld[row];
    gyv = a.Gx[(long long)row * D + colv];
  }
  TRF_MARK(35);
  if constexpr (PV) {
#pragma unroll
    for (int i = 0; i < PQ4; ++i) {
      if (TRF_THREADS * i >= TRF4_ROWS * rq) break;  // uniform
      const int Q = tid + TRF_THREADS * i;
      const int r = Q / rq, cq = Q - r * rq;
      if (Q < TRF4_ROWS * rq && cq < nq) *(f32x4v*)(PT + 4 * Q) = pq[i];
    }
  } else {
#pragma unroll
    for (int i = 0; i < PQS; ++i) {
      if (TRF_THREADS * i >= TRF4_ROWS * np) break;  // uniform
      const int idx = tid + TRF_THREADS * i;
      const int r = idx / np;
      if (idx < TRF4_ROWS * np) PT[r * pld + (idx - r * np)] = ps[i];
    }
    // the rows' padding: zero A operands for the steps of the final layer's dX that reach past np
    const int pad = pld - np;
    for (int idx = tid; idx < TRF4_ROWS * pad; idx += TRF_THREADS) PT[(idx / pad) * pld + np + idx % pad] = 0.0f;
  }
  trf2_static_for<MAXT>([&](auto ti) {
    constexpr int t = decltype(ti)::value;
    if (t < 2 * L + 1 && tid < 4 * H)
      *(f32x4v*)(ACT + t * TRF4_ROWS * TRF4_LDA + ((4 * tid) / H) * TRF4_LDA + (4 * tid) % H) = aq[t];
  });
#pragma unroll
  for (int i = 0; i < XQ; ++i) {
    if (TRF_THREADS * i >= TRF4_ROWS * D) break;  // uniform
    const int idx = tid + TRF_THREADS * i;
    const int r = idx / D;
    if (idx < TRF4_ROWS * D) XT[r * ldx + (idx - r * D)] = xq[i];
  }
  TRF_MARK(36);
  // (requested behind the stores of the tiles: in front of them, the registers of 70 loads in flight ran out and the
  // wave waited for ALL of them before it could form the last addresses - three round trips in a row at kernel
  // start; here the operands' round trip runs under the transform items)
  // the final layer's dX: the wave's 16 hidden columns over all np parameters, in groups of 16 parameters - a lane's
  // quarter of a group is one 16-byte LDS read (A) and one 16-byte load from the transposed weights (B) for four
  // MFMA steps; eight groups per chunk, two chunks in registers
  const int NG = (np + 15) >> 4;  // <= 48: six chunks
  // (a group beyond np reads on into the next row of W^T - the bias behind it for the last row - finite values that
  // meet a zero A operand)
  const int wf_row = 16 * wave + i16 < H ? 16 * wave + i16 : H - 1;  // (waves beyond H stay inside the matrix)
  const unsigned wf_voff = 4u * ((unsigned)wf_row * (unsigned)np + 4u * (unsigned)kg);  // W^T[in][o]
  const unsigned wf_s0 = 4u * (unsigned)a.off[2 + 4 * L];
  // (PV == false: W[o = 128 c + 16 t + 4 e + lane / 16][in] of the natural layout, one value per MFMA step)
  const int wf_col = 16 * wave + i16 < H ? 16 * wave + i16 : H - 1;
  auto load_chunk = [&](f32x4v (&b)[8], int c) {
    if constexpr (PV) {
#pragma unroll
      for (int t = 0; t < 8; ++t)  // (groups past the last one re-read it: nothing is read past the matrix's rows)
        b[t] = trf4_bload4(rsrcT, wf_voff, wf_s0 + 64u * (unsigned)(8 * c + t < NG ? 8 * c + t : NG - 1));
    } else {
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e)
        {
          const int o = 4 * (32 * c + 4 * t + e) + kg;  // (rows past the matrix re-read its last one)
          b[t][e] = trf4_bload(rsrcP, 4u * ((unsigned)(o < np ? o : np - 1) * (unsigned)H + (unsigned)wf_col), wf_s0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  f32x4v wa[8], wb[8];
  load_chunk(wa, 0);
  load_chunk(wb, 1);
  float wc[TRF4_Q], wn[TRF4_Q];
  trf4_load_wx(wn, rsrcP, a.off[L > 0 ? 4 + 4 * (L - 1) : 0], L > 0 ? H : a.d_id, H, wave, lane);

  TRF_MARK(37);
  trf_barrier_lds();
  TRF_MARK(30);

  // ---- transform backward: parameters -> their gradient in place, dL/dx_tr, x restored ----
  if (it_on) {
    float* pit = PT + it_r * pld;
    const int m = it_m;
    if (it_r < rows_here) {
      float gx;
      if constexpr (K > 0) {
        gx = tr_rq_backward_core<K>(pit + m * (3 * K - 1), xs, gyv, gldv, a.c, pit + m * (3 * K - 1));
      } else {
        gx = gyv;
        if (a.transform != NSA_T_ADDITIVE) {
          const float sig = 1.0f / (1.0f + expf(-(pit[a.d_tr + m] + a.c.aff_a)));
          const float scale = sig + a.c.aff_b;
          pit[a.d_tr + m] = (gyv * xs + gldv / scale) * sig * (1.0f - sig);
          gx = gyv * scale;
        }
        pit[m] = gyv;
      }
      const long long g = (long long)(r0 + it_r) * D + colv;
      a.X[g] = xs;
      a.Gx[g] = gx;
    } else {  // rows beyond the batch contribute nothing to the weight gradients
      if constexpr (K > 0) {
#pragma unroll
        for (int q = 0; q < 3 * K - 1; ++q) pit[m * (3 * K - 1) + q] = 0.0f;
      } else {
        pit[m] = 0.0f;
        if (a.transform != NSA_T_ADDITIVE) pit[a.d_tr + m] = 0.0f;
      }
    }
  }
  TRF_MARK(41);
  trf_barrier_lds();
  TRF_MARK(31);

  // ---- final layer, dX: gh = gp . Wf -> G0 ----
  if (HH != 0 || 16 * wave < H) {
    f32x4v acc = {0}, acc2 = {0};
    const float* ax = PT + i16 * pld;
    trf2_static_for<6>([&](auto ci) {
      constexpr int c = decltype(ci)::value;
      f32x4v (&b)[8] = (c & 1) ? wb : wa;
      if constexpr (PV) {
        f32x4v av[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const int qd = 4 * (8 * c + t) + kg;
          av[t] = *(const f32x4v*)(ax + 4 * (qd < nq ? qd : 0));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const float keep = (4 * (8 * c + t) + kg < nq) ? 1.0f : 0.0f;
          av[t] *= keep;
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          if (8 * c + t < NG) {  // uniform
            acc = trf4_mfma(av[t][0], b[t][0], acc);
            acc2 = trf4_mfma(av[t][1], b[t][1], acc2);
            acc = trf4_mfma(av[t][2], b[t][2], acc);
            acc2 = trf4_mfma(av[t][3], b[t][3], acc2);
          }
        }
        if (c + 2 < 6) load_chunk(b, c + 2);
      } else {
        // step 32 c + 4 t + e: parameters o = 4 (32 c + 4 t + e) + lane / 16 (zeros of the padding past np)
        float av[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) av[q] = ax[128 * c + 4 * q + kg];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          if (128 * c + 16 * t < np) {  // uniform
            acc = trf4_mfma(av[4 * t], b[t][0], acc);
            acc2 = trf4_mfma(av[4 * t + 1], b[t][1], acc2);
            acc = trf4_mfma(av[4 * t + 2], b[t][2], acc);
            acc2 = trf4_mfma(av[4 * t + 3], b[t][3], acc2);
          }
        }
        if (c + 2 < 6) load_chunk(b, c + 2);
      }
    });
    if (HH != 0 || 16 * wave + i16 < H) {
#pragma unroll
      for (int r = 0; r < 4; ++r) G0[(4 * kg + r) * TRF4_LD + 16 * wave + i16] = acc[r] + acc2[r];
    }
  }
  TRF_MARK(42);
  // ---- final layer, dW: dWf = gp^T . h_L (output tiles wave + 8 s, the eight input tiles of each), dbf ----
  {
    const int OTf = (np + 15) >> 4;
    float* gWf = GR + a.off[2 + 4 * L];
    float* gbf = GR + a.off[3 + 4 * L];
    const float* HL = ACT + L * TRF4_ROWS * TRF4_LDA;
    const bool vec = SLAB && (((unsigned long long)gWf) & 15) == 0;  // (H % 4 == 0)
    float at[8][4];
#pragma unroll
    for (int it = 0; it < 8; ++it)
#pragma unroll
      for (int s = 0; s < 4; ++s) at[it][s] = HL[(4 * s + kg) * TRF4_LDA + 16 * it + i16];
    for (int ot = wave; ot < OTf; ot += TRF_WAVES) {
      const int o = 16 * ot + i16;
      const int ocl = o < np ? o : 0;
      float bvv[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) bvv[s] = PT[(4 * s + kg) * pld + ocl];
      {
        float cs = (bvv[0] + bvv[1]) + (bvv[2] + bvv[3]);
        cs += __shfl_xor(cs, 16);
        cs += __shfl_xor(cs, 32);
        if (kg == 0 && o < np) trf2_grad_out<SLAB>(&gbf[o], cs);
      }
#pragma unroll
      for (int it = 0; it < 8; it += 2) {
        if (HH != 0 || 16 * it < H) {  // uniform
          f32x4v d0 = {0}, d1 = {0};
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            d0 = trf4_mfma(at[it][s], bvv[s], d0);
            d1 = trf4_mfma(at[it + 1][s], bvv[s], d1);
          }
          trf4_dw_store<SLAB>(gWf, np, H, o, 16 * it, kg, d0, vec);
          if (HH != 0 || 16 * (it + 1) < H) trf4_dw_store<SLAB>(gWf, np, H, o, 16 * (it + 1), kg, d1, vec);
        }
      }
    }
  }
  TRF_MARK(43);
  trf_barrier_lds();
  TRF_MARK(32);

  // ---- residual blocks, last to first ----
  for (int l = L - 1; l >= 0; --l) {
    const float* hl = ACT + l * TRF4_ROWS * TRF4_LDA;
    const float* ul = ACT + (L + 1 + l) * TRF4_ROWS * TRF4_LDA;
    // h_{l+1} = h_l + W2 act(u_l) + b2:  g_u = (G0 . W2) act'(u_l) -> G1
#pragma unroll
    for (int q = 0; q < TRF4_Q; ++q) wc[q] = wn[q];
    trf4_load_wx(wn, rsrcP, a.off[2 + 4 * l], H, H, wave, lane);
    trf4_back_linear<false, SLAB, HH>(G0, wc, ul, TRF4_LDA, nullptr, a.act, H, H, ul, a.act, nullptr, G1,
                                  GR + a.off[4 + 4 * l], GR + a.off[5 + 4 * l], wave, lane);
    trf_barrier_lds();
    // u_l = W1 act(h_l) + b1:  g_h_l = G0 + (G1 . W1) act'(h_l) -> G0
#pragma unroll
    for (int q = 0; q < TRF4_Q; ++q) wc[q] = wn[q];
    trf4_load_wx(wn, rsrcP, a.off[l > 0 ? 4 + 4 * (l - 1) : 0], l > 0 ? H : a.d_id, H, wave, lane);  // (no branch: see trf4_forward)
    trf4_back_linear<false, SLAB, HH>(G1, wc, hl, TRF4_LDA, nullptr, a.act, H, H, hl, a.act, G0, G0,
                                  GR + a.off[2 + 4 * l], GR + a.off[3 + 4 * l], wave, lane);
    trf_barrier_lds();
    TRF_MARK(33 + (L - 1 - l));
  }

  // ---- initial layer: dW0 = G0^T . x_id, g_x[:, id_cols] += G0 . W0 ----
#pragma unroll
  for (int q = 0; q < TRF4_Q; ++q) wc[q] = wn[q];
  trf4_back_linear<true, SLAB, HH>(G0, wc, XT, ldx, a.id_cols, ACT_NONE, a.d_id, H, nullptr, ACT_NONE, nullptr, G1,
                               GR + a.off[0], GR + a.off[1], wave, lane);
  trf_barrier_lds();
  for (int idx = tid; idx < TRF4_ROWS * a.d_id; idx += TRF_THREADS) {
    const int r = idx / a.d_id, k = idx - r * a.d_id;
    if (r < rows_here) {
      float* g = a.Gx + (long long)(r0 + r) * D + a.id_cols[k];
      *g = *g + G1[r * TRF4_LD + k];
    }
  }
  TRF_MARK(60);
}
