// Second generation of the fused training kernels (included by nsa_train.hip after nsa_train_fused.h; same
// TrfArgs, same workspace layout, same results up to the order of the float32 sums): the forward pass of a
// RUN of coupling layers in one launch, the backward pass of one coupling layer per launch.  Used when the
// conditioner is at most 64 wide (tr_v2_fits); the first generation stays for wider nets and as the
// cross-check (NESSAI_AMD_TRAIN_V2=0).  DESIGN.md section 4.3 has the measurements behind every choice below.
//
// What the first generation spends its time on (scripts/ubench/layer_latency.hip measures the floor:
// a 32 x 64 -> 64 layer costs a workgroup 1.07 us with its operands at hand, the first generation
// ~4 us) is a chain of exposed round trips: per layer the second half of the weights and the bias
// are requested where they are used, and only 2 of the 8 waves own an output tile.  Here
//   * every layer's B operands and bias are requested ONE LAYER AHEAD into registers (plain loads:
//     the compiler's own wait counting stays valid next to the activation saves), the operands of the
//     final layer - 60 % of the weight bytes - right after layer 0, two output tiles per wave;
//   * a hidden layer is split over all 8 waves, output tile x slice of k, and the partial sums meet
//     in LDS where bias, residual and the activation save are applied by all 512 threads;
//   * the parameters of the transform stay in an LDS tile for the spline / affine items (and are saved
//     for the backward pass with stores nobody waits for).
// Reference semantics: glasflow.nflows ResidualNet (call sites src/nessai/flows/nsf.py:85-95,
// realnvp.py:136-146) under FlowModel._train (src/nessai/flowmodel/base.py:365-452).
#pragma once
#include <type_traits>
#include <utility>

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): indices that are constants from the start (an
// array indexed inside a "#pragma unroll" loop can end up on the stack when the loop is unrolled after the pass that
// would have turned it into registers - and a loaded value stored to the stack waits for its load)
typedef float f32x4v __attribute__((ext_vector_type(4)));  // a plain vector register quad (float4 is a struct)

template <typename F, int... I>
__device__ __forceinline__ void trf2_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void trf2_static_for(F&& f) {
  trf2_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

#ifndef TRF2_ABL
#define TRF2_ABL 0  // diagnostic builds only: 1 drops the weight-gradient writes (and, the results being unused, the MFMAs behind them)
#endif
#define TRF2_PP 8  // k pairs per split-k item: 32 pairs / 4 slices (H <= 64), 32 / 8 (H <= 32)

// split of a [rows x Kd] . [Kd x O] product, O <= 32 HT, over the 8 waves: tile ot = wave % HT, slice ks = wave / HT
template <int HT>
struct Trf2Item {
  static constexpr int KS = TRF_WAVES / HT;
  int ot, ks, pp;  // pp: k pairs per slice (wave-uniform, <= TRF2_PP)
  __device__ __forceinline__ Trf2Item(int wave, int Kd) {
    ot = wave % HT;
    ks = wave / HT;
    const int pairs = (Kd + 1) >> 1;
    pp = (pairs + KS - 1) / KS;
  }
};

// B operands of this wave's item (W^T[k][o], lanes along o: coalesced) and the bias of the column this
// THREAD finishes in the reduction (column tid % (32 HT)).  Every load is unconditional from a clamped
// address and its value is first touched by the MFMA: a select on the loaded value makes the compiler
// branch round the load and wait for it on the spot - 64 serialised round trips in the first version of
// this kernel.  Out-of-range k carry a zero A operand instead; out-of-range columns are never written.
// a zero the compiler cannot see through (see its use in trf2_forward)
__device__ __forceinline__ int trf2_opaque_zero() {
  int z;
  asm volatile("s_mov_b32 %0, 0" : "=s"(z));
  return z;
}

template <int HT>
__device__ __forceinline__ void trf2_load_w(float (&w)[TRF2_PP], float& bias, const float* __restrict__ Wt,
                                            const float* __restrict__ b, int Kd, int O, int wave, int lane,
                                            int tid) {
  const Trf2Item<HT> it(wave, Kd);
  const int half = lane >> 5, j = lane & 31;
  const int o = it.ot * 32 + j;
  const unsigned oc = (unsigned)(o < O ? o : O - 1);
#pragma unroll
  for (int q = 0; q < TRF2_PP; ++q) {
    const int k = 2 * (it.ks * it.pp + q) + half;
    w[q] = Wt[(unsigned)(k < Kd ? k : Kd - 1) * (unsigned)O + oc];
  }
  const int oc2 = tid & (32 * HT - 1);
  bias = b[oc2 < O ? oc2 : O - 1];
}

// out[row][o] = sum_k act(Ain[row][k]) W[o][k] + b[o] (+ R[row][o]) for the workgroup's 32 rows.
// Ends with the output tile WRITTEN but not yet published (the caller's barrier does that).
// GATHER: the A operand's column of slot q is cidx[q] (layer 0 reads the identity columns out of the x tile).
template <int HT, bool GATHER = false>
__device__ __forceinline__ void trf2_layer(const float* __restrict__ Ain, int act_in, const float (&w)[TRF2_PP],
                                           const int (&cidx)[TRF2_PP],
                                           float bias, int Kd, int O, const float* R, float* out_tile,
                                           float* __restrict__ out_glb, int ld, int r0, int n,
                                           float* __restrict__ part, int wave, int lane, int tid) {
  constexpr int KS = Trf2Item<HT>::KS;
  const Trf2Item<HT> it(wave, Kd);
  const int half = lane >> 5, j = lane & 31;
  float av[TRF2_PP];
#pragma unroll
  for (int q = 0; q < TRF2_PP; ++q) {
    const int k = 2 * (it.ks * it.pp + q) + half;
    av[q] = GATHER ? Ain[cidx[q]] : Ain[j * ld + (k < Kd ? k : 0)];
  }
  trf_act_batch(av, act_in);
#pragma unroll
  for (int q = 0; q < TRF2_PP; ++q) {  // a factor, not a select: see trf2_load_w
    const int k = 2 * (it.ks * it.pp + q) + half;
    av[q] *= (q < it.pp && k < Kd) ? 1.0f : 0.0f;
  }
  f32x16 acc = {0}, acc2 = {0};
#pragma unroll
  for (int q = 0; q < TRF2_PP; q += 2) {
    if (q < it.pp) {  // wave-uniform
      acc = nsa_mfma(av[q], w[q], acc);
      acc2 = nsa_mfma(av[q + 1], w[q + 1], acc2);
    }
  }
  float* pp_ = part + wave * 1024;
#pragma unroll
  for (int r = 0; r < 16; ++r) pp_[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + j] = acc[r] + acc2[r];
  trf_barrier_lds();
  // 32 x 32 HT outputs over 512 threads: column tid % (32 HT), rows tid / (32 HT) + (16 / HT) i
  constexpr int NC = 32 * HT, STEP = TRF_THREADS / NC;
  const int o = tid & (NC - 1), rb = tid / NC;
  const float* pc = part + (o >> 5) * 1024 + (o & 31);
  if (o < O) {
#pragma unroll
    for (int i = 0; i < 32 / STEP; ++i) {
      const int row = rb + STEP * i;
      float v = bias;
#pragma unroll
      for (int s = 0; s < KS; ++s) v += pc[(s * HT) * 1024 + row * 32];
      if (R) v += R[row * ld + o];
      out_tile[row * ld + o] = v;
      if (r0 + row < n) out_glb[(unsigned)(r0 + row) * (unsigned)O + (unsigned)o] = v;
    }
  }
}

// Async copy (global_load_lds, 4 bytes per lane) of the workgroup's 32 rows of a row-major [n][W] global matrix
// into an LDS tile with row stride `lds_ld`: rows wave, wave + 8, ..., 64 floats per instruction, no registers
// and no index arithmetic per element.  Rows beyond n read row n - 1.  The caller waits (vmcnt) before its barrier.
__device__ __forceinline__ void trf2_rows_async(float* tile, int lds_ld, const float* __restrict__ src, int W,
                                                int r0, int n, int wave, int lane) {
  for (int r = wave; r < TRF_ROWS; r += TRF_WAVES) {
    const int row = r0 + r < n ? r0 + r : n - 1;
    const float* g = src + (long long)row * W;
    for (int c0 = 0; c0 < W; c0 += 64) {
      if (c0 + lane < W)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + c0 + lane),
                                         (__attribute__((address_space(3))) void*)(tile + r * lds_ld + c0), 4, 0, 0);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// forward pass: ONE launch for COUNT consecutive coupling layers (permutations are folded into the column
// tables, so the reference's default flows are runs of couplings).  The workgroup's 32 rows of x stay in LDS
// from one coupling to the next - a launch, its argument fetch, the first round trip to the batch and the
// wait for the last stores are paid once per run - and the next coupling's first operands are requested while
// the current one's transform runs.  COUNT is a template parameter and the couplings are laid out one after
// the other in the code: written as a run-time loop, the lane-dependent halves of every address were hoisted
// out of it and held in registers across it (157 spilled registers, and a spilled operand waits for its
// load on the spot - slower than a launch per coupling).
// ---------------------------------------------------------------------------------------------------
#define TRF2_MAXC 6
struct Trf2One {  // what differs from one coupling layer of the run to the next
  int d_id, d_tr, np, pad_;
  const int* id_cols;
  const int* tr_cols;
  float* xsave;  // [n][d_tr]
  float* h;      // [L + 1][n][H]
  float* u;      // [L][n][H]
  float* p;      // [n][np]
  int off[2 + 4 * TRF_MAX_L + 2];  // flat offsets of W0, b0, {W1, b1, W2, b2} x L, Wf, bf
};
struct Trf2Multi {
  TrfArgs a;  // the fields common to the run; np = the largest np of the run (LDS tile)
  Trf2One one[TRF2_MAXC];
};
static_assert(sizeof(Trf2Multi) <= 4096, "kernel arguments are limited to 4 KiB");

#define TRF2_MARK(i)                                                      \
  do {                                                                    \
    if (dbg != nullptr && blockIdx.x == 0 && threadIdx.x == 0)            \
      dbg[i] = (float)(long long)(clock64() - t_start);                   \
  } while (0)

// LDS floats of trf2_forward: the x tile, two activation tiles, the partial sums, the parameter tile
static inline int trf2_ldx(const TrfArgs& a) { return (a.D + 1) | 1; }
static inline size_t trf2_forward_floats(const TrfArgs& a) {
  return (size_t)TRF_ROWS * trf2_ldx(a) + (size_t)2 * TRF_ROWS * a.ld + (size_t)TRF_WAVES * 1024 +
         (size_t)TRF_ROWS * a.np + TRF_ROWS;
}

template <int K, int HT, int COUNT>  // spline bins (0: affine / additive); hidden width <= 32 HT; couplings
__global__ void __launch_bounds__(TRF_THREADS) trf2_forward(Trf2Multi m) {
  extern __shared__ float trf_sm[];
  const TrfArgs& a = m.a;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, j = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // in an SGPR: the item tests below are scalar branches
  const int r0 = blockIdx.x * TRF_ROWS;
  const int ld = a.ld, n = a.n, H = a.H, D = a.D, L = a.L;
  const int ldx = (D + 1) | 1;
  float* XT = trf_sm;                  // [32][ldx] the workgroup's rows of x, all columns
  float* HC = XT + TRF_ROWS * ldx;     // h_l
  float* U = HC + TRF_ROWS * ld;       // u_l
  float* part = U + TRF_ROWS * ld;     // [8 items][32][32] partial sums
  float* PT = part + TRF_WAVES * 1024;  // [32][np] transform parameters
  float* ldsum = PT + TRF_ROWS * a.np;  // [32] log|det J| of the run, per row (one global update per row at the end)
  const float* P = a.params;
  const float* WT = a.wt;
  const long long nH = (long long)n * H;
  long long t_start = clock64();
  float* dbg = COUNT == 1 ? a.dbg : nullptr;
  if (tid < TRF_ROWS) ldsum[tid] = 0.0f;

  // the x rows (async, straight into the padded tile), the identity column table and layer 0 of the first
  // coupling; everything else is requested behind the first barrier, so that it waits for these only
  trf2_rows_async(XT, ldx, a.X, D, r0, n, wave, lane);
  int cidx[TRF2_PP], none[TRF2_PP];
  float wc[TRF2_PP], wn[TRF2_PP], bc, bn = 0.0f;
  {
    const Trf2One& c0 = m.one[0];
    const Trf2Item<HT> it0(wave, c0.d_id);
#pragma unroll
    for (int q = 0; q < TRF2_PP; ++q) {
      const int k = 2 * (it0.ks * it0.pp + q) + half;
      cidx[q] = c0.id_cols[k < c0.d_id ? k : 0] + j * ldx;
      none[q] = 0;
    }
    trf2_load_w<HT>(wc, bc, WT + c0.off[0], P + c0.off[1], c0.d_id, H, wave, lane, tid);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the async rows (and layer 0)
  trf_barrier_lds();

  const int tid_launch = tid;
  trf2_static_for<COUNT>([&](auto bi) {
    constexpr int b = decltype(bi)::value;
    // lane indices the compiler cannot prove equal from one coupling to the next: what is derived from them is
    // recomputed per coupling, as in a launch per coupling, instead of being shared between the copies of the
    // code and held in registers across them (94 spilled registers for three couplings otherwise)
    const int tid = tid_launch + trf2_opaque_zero(), lane = tid & 63, half = lane >> 5, j = lane & 31;
    const Trf2One& c = m.one[b];
    const int np = c.np, d_id = c.d_id, d_tr = c.d_tr;
    // the parameter tile is saved for the backward pass with 16-byte stores out of LDS where its rows allow it
    // (6 store instructions per thread instead of 32)
    const bool p_vec = (np & 3) == 0 && (((unsigned long long)c.p) & 15) == 0;
    if (b == COUNT - 1 && COUNT > 1) {  // phase stamps of the last coupling of the run
      t_start = clock64();
      dbg = a.dbg;
    }
    // the first hidden layer and - used only after the hidden layers - the final layer's operands for
    // output tiles wave and wave + 8
    if (L > 0) trf2_load_w<HT>(wn, bn, WT + c.off[2], P + c.off[3], H, H, wave, lane, tid);
    constexpr int FP = 16 * HT;  // k pairs of the final layer
    const int OTf = (np + 31) >> 5;
    const float* Wf = WT + c.off[2 + 4 * L];
    const float* bf = P + c.off[3 + 4 * L];
    TRF2_MARK(0);
    // h_0 = W0 x_id + b0
    trf2_layer<HT, true>(XT, ACT_NONE, wc, cidx, bc, d_id, H, nullptr, HC, c.h, ld, r0, n, part, wave, lane, tid);
    // (requested here rather than at the top of the coupling: a wave has 63 memory operations in flight at most,
    // and behind a previous coupling the queue still holds its stores - 64 loads at the top waited for them)
    float wf[2][FP], bfv[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int o = (wave + TRF_WAVES * s) * 32 + j;
      const unsigned oc = (unsigned)(o < np ? o : np - 1);
      bfv[s] = bf[oc];
#pragma unroll
      for (int q = 0; q < FP; ++q) {
        const int k = 2 * q + half;
        wf[s][q] = Wf[(unsigned)(k < H ? k : H - 1) * (unsigned)np + oc];
      }
    }
    TRF2_MARK(1);
    for (int l = 0; l < L; ++l) {
      // u_l = W1 act(h_l) + b1;  h_{l+1} = h_l + W2 act(u_l) + b2
#pragma unroll
      for (int q = 0; q < TRF2_PP; ++q) wc[q] = wn[q];
      bc = bn;
      trf2_load_w<HT>(wn, bn, WT + c.off[4 + 4 * l], P + c.off[5 + 4 * l], H, H, wave, lane, tid);
      trf_barrier_lds();
      trf2_layer<HT>(HC, a.act, wc, none, bc, H, H, nullptr, U, c.u + l * nH, ld, r0, n, part, wave, lane, tid);
      TRF2_MARK(2 + 2 * l);
#pragma unroll
      for (int q = 0; q < TRF2_PP; ++q) wc[q] = wn[q];
      bc = bn;
      if (l + 1 < L) trf2_load_w<HT>(wn, bn, WT + c.off[6 + 4 * l], P + c.off[7 + 4 * l], H, H, wave, lane, tid);
      trf_barrier_lds();
      trf2_layer<HT>(U, a.act, wc, none, bc, H, H, HC, HC, c.h + (l + 1) * nH, ld, r0, n, part, wave, lane, tid);
      TRF2_MARK(3 + 2 * l);
    }
    trf_barrier_lds();
    // parameters = Wf h_L + bf: whole output tiles, no split (waves w and w + 4 share a SIMD: 3 tiles each of 12)
    {
      float av[FP];
      const float* ax = HC + j * ld;
#pragma unroll
      for (int q = 0; q < FP; ++q) {
        const int k = 2 * q + half;
        av[q] = ax[k < H ? k : 0];
      }
#pragma unroll
      for (int q = 0; q < FP; ++q) av[q] *= (2 * q + half < H) ? 1.0f : 0.0f;
      const int pairs = (H + 1) >> 1;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int t = wave + TRF_WAVES * s;
        if (t < OTf) {  // wave-uniform
          f32x16 acc = {0}, acc2 = {0};
#pragma unroll
          for (int q = 0; q < FP; q += 2) {
            if (q < pairs) {
              acc = nsa_mfma(av[q], wf[s][q], acc);
              acc2 = nsa_mfma(av[q + 1], wf[s][q + 1], acc2);
            }
          }
          const int o = t * 32 + j;
          if (o < np) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int rl = (r & 3) + 8 * (r >> 2) + 4 * half;
              const float v = acc[r] + acc2[r] + bfv[s];
              PT[rl * np + o] = v;
              if (!p_vec && r0 + rl < n) c.p[(unsigned)(r0 + rl) * (unsigned)np + (unsigned)o] = v;
            }
          }
        }
      }
    }
    TRF2_MARK(20);
    // the next coupling's identity columns and layer 0, requested before this one's transform
    if constexpr (b + 1 < COUNT) {
      const Trf2One& cn = m.one[b + 1];
      const Trf2Item<HT> itn(wave, cn.d_id);
#pragma unroll
      for (int q = 0; q < TRF2_PP; ++q) {
        const int k = 2 * (itn.ks * itn.pp + q) + half;
        cidx[q] = cn.id_cols[k < cn.d_id ? k : 0] + j * ldx;
      }
      trf2_load_w<HT>(wc, bc, WT + cn.off[0], P + cn.off[1], cn.d_id, H, wave, lane, tid);
    }
    trf_barrier_lds();
    TRF2_MARK(21);
    if (p_vec) {
      const int rows_here = n - r0 < TRF_ROWS ? n - r0 : TRF_ROWS;
      const int quads = rows_here * (np >> 2);
      float* dst = c.p + (long long)r0 * np;
      for (int i = tid; i < quads; i += TRF_THREADS) *(f32x4v*)(dst + 4 * i) = *(const f32x4v*)(PT + 4 * i);
    }
    // the transform on the x tile's own copy of x (no load behind the stores above); y goes back into the
    // tile for the next coupling and to global memory for the loss / the backward pass
    const int items = TRF_ROWS * d_tr;
    for (int e = tid; e < items; e += TRF_THREADS) {
      const int r = e / d_tr, mi = e - r * d_tr;
      const int row = r0 + r;
      if (row >= n) continue;
      const int col = c.tr_cols[mi];
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
      if (b + 1 < COUNT) XT[r * ldx + col] = y;
      c.xsave[row * d_tr + mi] = x;
      a.X[(long long)row * D + col] = y;
      if (K > 0 || a.transform != NSA_T_ADDITIVE) atomicAdd(&ldsum[r], ldj);
    }
    TRF2_MARK(22);
    if (b + 1 < COUNT) trf_barrier_lds();  // the x tile of the next coupling
  });
  trf_barrier_lds();
  if (tid < TRF_ROWS && r0 + tid < n) a.logdet[r0 + tid] += ldsum[tid];  // (this workgroup's rows only)
}

// ---------------------------------------------------------------------------------------------------
// backward pass of the coupling layer
//
// The workgroup's slice of everything the forward pass saved - transform parameters, h_0..h_L, u_0..u_{L-1},
// its rows of x and of dL/dx - is requested at kernel start and lives in LDS; the parameter tile is
// overwritten in place by its own gradient (the spline / affine items), so the gradient of the final
// layer's 3K - 1 parameters per feature never travels through global memory.  Every linear layer is then
// one step for all 8 waves:
//   dX  partial products  G[row][out] . W[out][in], in-tile x slice of `out`   (B from registers, requested
//       one step ahead; the final layer's, 60 % of the bytes, at kernel start)
//   dW  G^T[out][row] . act(A)[row][in], output tile x slice of the rows, both operands from LDS, added to
//       the flat gradient with float atomics (which also merge the row slices)
//   barrier, then all 512 threads sum the dX slices, apply act'(z) and the residual path, write the next
//   gradient tile and add its column sums to the bias gradient of the layer that produced it.
// ---------------------------------------------------------------------------------------------------

#define TRF2_FQ 64   // k pairs per slice of the final layer's dX: 256 pairs (np <= 512) / 4 slices

// row stride of the backward kernel's parameter tile
__host__ __device__ static inline int trf2_pld(int np, bool vec) {
  return vec ? ((np >> 2) & 1 ? np : np + 4) : (np | 1);
}
static inline size_t trf2_backward_floats(const TrfArgs& a) {
  const size_t ldx = (size_t)trf2_ldx(a);
  return (size_t)TRF_ROWS * (a.np + 4) + (size_t)(2 * a.L + 1) * TRF_ROWS * a.H + 2 * TRF_ROWS * ldx +
         (size_t)2 * TRF_ROWS * a.ld + (size_t)TRF_WAVES * 1024;
}

// a workgroup's contribution to a weight gradient: its own copy of the flat vector (plain store, summed by
// tr_reduce_slab) or a float atomic on the shared one
template <bool SLAB>
__device__ __forceinline__ void trf2_grad_out(float* dst, float v) {
  if (TRF2_ABL & 1) return;
  if constexpr (SLAB) *dst = v;
  else atomicAdd(dst, v);
}

// One 32 x 32 tile of a weight gradient out of the accumulators of D[i = in][jj = out] (the `in` index runs down
// the registers of a lane, so that a lane owns runs of 4 consecutive inputs of ITS output row: four 16-byte stores
// per tile instead of sixteen 4-byte ones when the rows allow it).  gW: [O][Kin] row-major.
template <bool SLAB>
__device__ __forceinline__ void trf2_dw_store(float* __restrict__ gW, int O, int Kin, int o, int in0, int half,
                                              const f32x16& acc, const f32x16& acc2) {
  if (o >= O) return;
  float* row = gW + (unsigned)o * (unsigned)Kin;
  const bool vec = SLAB && (Kin & 3) == 0 && (((unsigned long long)gW) & 15) == 0 && !(TRF2_ABL & 1);  // uniform
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int c = in0 + 8 * g + 4 * half;
    if (vec) {
      if (c < Kin) {
        f32x4v v = {acc[4 * g] + acc2[4 * g], acc[4 * g + 1] + acc2[4 * g + 1], acc[4 * g + 2] + acc2[4 * g + 2],
                    acc[4 * g + 3] + acc2[4 * g + 3]};
        *(f32x4v*)(row + c) = v;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c + e < Kin) trf2_grad_out<SLAB>(row + c + e, acc[4 * g + e] + acc2[4 * g + e]);
    }
  }
}

// B operands of a dX item: W[out = k][in = it * 32 + j] (natural layout, lanes along `in`: coalesced)
template <int HT>
__device__ __forceinline__ void trf2_load_wx(float (&w)[TRF2_PP], const float* __restrict__ W, int O, int Kin,
                                             int wave, int lane) {
  const Trf2Item<HT> it(wave, O);
  const int half = lane >> 5, j = lane & 31;
  const int c = it.ot * 32 + j;
  const unsigned cc = (unsigned)(c < Kin ? c : Kin - 1);
#pragma unroll
  for (int q = 0; q < TRF2_PP; ++q) {
    const int k = 2 * (it.ks * it.pp + q) + half;
    w[q] = W[(unsigned)(k < O ? k : O - 1) * (unsigned)Kin + cc];
  }
}

// One linear layer of the backward pass (see the header of this section).
//   G [32][ld]: gradient w.r.t. the layer's output (O = H columns);  w: this wave's dX operands
//   Ain: the layer's input tile (A = act_in(Ain)), row stride a_ld, column of slot c = acol(c);  Kin inputs
//   Z: pre-activation whose act' multiplies dX (null: none), row stride H;  Rs: tile added to the result (null: none)
//   out [32][ld] <- result;  gW / gb: flat gradients of this layer's weights and bias
template <int HT, bool GATHER, bool SLAB>
__device__ __forceinline__ void trf2_back_linear(const float* __restrict__ G, const float (&w)[TRF2_PP], int O,
                                                 const float* __restrict__ Ain, int a_ld, const int* __restrict__ cols,
                                                 int act_in, int Kin, const float* __restrict__ Z, int act_z,
                                                 const float* Rs, float* out, float* __restrict__ gW,
                                                 float* __restrict__ gb, int ld, float* __restrict__ part,
                                                 int wave, int lane, int tid) {
  constexpr int KS = Trf2Item<HT>::KS;
  const int half = lane >> 5, j = lane & 31;
  const int IT = (Kin + 31) >> 5;  // <= HT
  // dX partial product of item (in-tile, slice of `out`)
  {
    const Trf2Item<HT> it(wave, O);
    float av[TRF2_PP];
#pragma unroll
    for (int q = 0; q < TRF2_PP; ++q) {
      const int k = 2 * (it.ks * it.pp + q) + half;
      av[q] = G[j * ld + (k < O ? k : 0)];
    }
#pragma unroll
    for (int q = 0; q < TRF2_PP; ++q) {
      const int k = 2 * (it.ks * it.pp + q) + half;
      av[q] *= (q < it.pp && k < O) ? 1.0f : 0.0f;
    }
    f32x16 acc = {0}, acc2 = {0};
    if (it.ot < IT) {
#pragma unroll
      for (int q = 0; q < TRF2_PP; q += 2) {
        if (q < it.pp) {
          acc = nsa_mfma(av[q], w[q], acc);
          acc2 = nsa_mfma(av[q + 1], w[q + 1], acc2);
        }
      }
    }
    float* pp_ = part + wave * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) pp_[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + j] = acc[r] + acc2[r];
  }
  // dW: one (out-tile, in-tile) per wave, all 32 rows (every element of the gradient is written once per
  // workgroup); the last wave sums the columns of G for the bias instead
  {
    const int NT = HT * IT;  // 1, 2 or 4
    if (wave < NT) {
      const int ot = wave % HT, itl = wave / HT;
      const int oc = ot * 32 + j, kc = itl * 32 + j;
      const int ocl = oc < O ? oc : 0, kcl = kc < Kin ? kc : 0;
      const int acol = GATHER ? cols[kcl] : kcl;
      float av[16], bv[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = 2 * q + half;
        av[q] = G[row * ld + ocl];
        bv[q] = Ain[row * a_ld + acol];
      }
      trf_act_batch(bv, act_in);
      f32x16 acc = {0}, acc2 = {0};
#pragma unroll
      for (int q = 0; q < 16; q += 2) {  // D[i = in][jj = out]
        acc = nsa_mfma(bv[q], av[q], acc);
        acc2 = nsa_mfma(bv[q + 1], av[q + 1], acc2);
      }
      trf2_dw_store<SLAB>(gW, O, Kin, oc, itl * 32, half, acc, acc2);
    } else if (wave == TRF_WAVES - 1 && lane < O) {
      float s = 0.0f;
#pragma unroll 8
      for (int row = 0; row < TRF_ROWS; ++row) s += G[row * ld + lane];
      trf2_grad_out<SLAB>(&gb[lane], s);
    }
  }
  trf_barrier_lds();
  // sum the slices; act'(z), residual path, bias gradient of the producing layer
  constexpr int NC = 32 * HT, STEP = TRF_THREADS / NC;
  const int c = tid & (NC - 1), rb = tid / NC;
  const float* pc = part + (c >> 5) * 1024 + (c & 31);
  if (c < Kin) {
#pragma unroll
    for (int i = 0; i < 32 / STEP; ++i) {
      const int row = rb + STEP * i;
      float v = 0.0f;
#pragma unroll
      for (int s = 0; s < KS; ++s) v += pc[(s * HT) * 1024 + row * 32];
      if (Z) {
        const float z = Z[row * O + c];  // the tiles of h / u are [32][H] and Kin == H here
        if (act_z == NSA_ACT_RELU) v = z > 0.0f ? v : 0.0f;
        else v *= tr_act_grad(z, act_z);
      }
      if (Rs) v += Rs[row * ld + c];
      out[row * ld + c] = v;
    }
  }
}

// VEC (np % 4 == 0, 16-byte aligned parameter blocks: tr_v2_vec): the parameter tile has rows of pld floats with
// pld / 4 odd - 16-byte aligned rows whose 16-byte reads are conflict-free - and is filled by 1 KiB async pieces;
// the final layer's contraction index is dealt out so that every lane owns runs of 4 consecutive parameters:
// one 16-byte LDS read (A) and one 16-byte load from the transposed weights (B) feed four MFMA steps.
template <int K, int HT, bool SLAB, bool VEC>
__global__ void __launch_bounds__(TRF_THREADS) trf2_backward(TrfArgs a) {
  extern __shared__ float trf_sm[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, j = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // in an SGPR: the item tests below are scalar branches
  const int r0 = blockIdx.x * TRF_ROWS;
  const int ld = a.ld, n = a.n, H = a.H, np = a.np, D = a.D, L = a.L;
  const int ldx = (D + 1) | 1, pld = trf2_pld(np, VEC);
  float* PT = trf_sm;                          // [32][pld] parameters, then their gradient
  float* ACT = PT + TRF_ROWS * pld;            // h_0 .. h_L, u_0 .. u_{L-1}: [32][H] each
  float* XT = ACT + (2 * L + 1) * TRF_ROWS * H;  // [32][ldx] rows of x (as the forward pass left them)
  float* GXT = XT + TRF_ROWS * ldx;            // [32][ldx] rows of dL/dx
  float* G0 = GXT + TRF_ROWS * ldx;
  float* G1 = G0 + TRF_ROWS * ld;
  float* part = G1 + TRF_ROWS * ld;
  const float* P = a.params;
  float* GR = SLAB ? a.gslab + (long long)blockIdx.x * a.npar : a.grads;
  const long long nH = (long long)n * H;
  const long long t_start = clock64();
  const int rows_here = (n - r0 < TRF_ROWS) ? (n - r0) : TRF_ROWS;

  // ---- what the transform items need: parameter tile, dL/dx rows, saved x ----
  // (VEC: plain 16-byte loads through registers - the LDS-DMA path of a CU retires about one instruction per 85
  // cycles whatever its size, and the ~80 of them this tile took were 6 k cycles of the prologue)
  constexpr int PQ4 = 9;  // 16-byte units of the parameter tile per thread: 32 (np + 4) / 4 <= 9 * 512
  constexpr int XQ = 8;   // x / dL/dx elements per thread: 32 D <= 8 * 512
  f32x4v pq[PQ4];
  float gq[XQ];
  const int rq = pld >> 2, nq = np >> 2;  // 16-byte units per tile row / per row of parameters
  if constexpr (VEC) {
#pragma unroll
    for (int i = 0; i < PQ4; ++i) {
      if (TRF_THREADS * i >= TRF_ROWS * rq) break;  // uniform
      int Q = tid + TRF_THREADS * i;
      Q = Q < TRF_ROWS * rq ? Q : TRF_ROWS * rq - 1;
      const int r = Q / rq, cq = Q - r * rq;
      const int row = r0 + r < n ? r0 + r : n - 1;
      pq[i] = *(const f32x4v*)(a.p + (long long)row * np + 4 * (cq < nq ? cq : nq - 1));
    }
#pragma unroll
    for (int i = 0; i < XQ; ++i) {
      if (TRF_THREADS * i >= TRF_ROWS * D) break;  // uniform
      int idx = tid + TRF_THREADS * i;
      idx = idx < TRF_ROWS * D ? idx : TRF_ROWS * D - 1;
      const int r = idx / D;
      const int row = r0 + r < n ? r0 + r : n - 1;
      gq[i] = a.Gx[(long long)row * D + (idx - r * D)];
    }
  } else {
    trf2_rows_async(PT, pld, a.p, np, r0, n, wave, lane);
    trf2_rows_async(GXT, ldx, a.Gx, D, r0, n, wave, lane);
  }
  // items of the transform: (row, feature) = (e / d_tr, e % d_tr), e = tid + 512 i
  constexpr int IQ = 2;  // d_tr <= 32
  float xs[IQ], gldv[IQ];
  int colv[IQ];
#pragma unroll
  for (int i = 0; i < IQ; ++i) {
    const int e = tid + TRF_THREADS * i;
    const int r = e / a.d_tr, m = e - r * a.d_tr;
    const int row = (r < rows_here) ? r0 + r : n - 1;
    xs[i] = a.xsave[row * a.d_tr + m];
    gldv[i] = a.gld[row];
    colv[i] = a.tr_cols[m];
  }
  if constexpr (VEC) {
#pragma unroll
    for (int i = 0; i < PQ4; ++i) {
      if (TRF_THREADS * i >= TRF_ROWS * rq) break;  // uniform
      const int Q = tid + TRF_THREADS * i;
      const int r = Q / rq, cq = Q - r * rq;
      if (Q < TRF_ROWS * rq && cq < nq) *(f32x4v*)(PT + 4 * Q) = pq[i];
    }
#pragma unroll
    for (int i = 0; i < XQ; ++i) {
      if (TRF_THREADS * i >= TRF_ROWS * D) break;  // uniform
      const int idx = tid + TRF_THREADS * i;
      const int r = idx / D;
      if (idx < TRF_ROWS * D) GXT[r * ldx + (idx - r * D)] = gq[i];
    }
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  trf_barrier_lds();
  TRF_MARK(30);
  // ---- requested now, used after the items: x rows (identity columns), activation tiles ([32][H] contiguous in
  // global memory and in LDS: 1 KiB pieces; H % 4 == 0 and 16-byte aligned blocks, tr_v2_fits), the final
  // layer's dX operands Wf[o][in] of item (in-tile wave % HT, slice wave / HT of the np parameters), and
  // the first backward step of the hidden layers: W2 of the last block (or W0 when there is none)
  constexpr int MAXT = 7;  // activation tiles through registers: 2 L + 1 <= 7 (one 16-byte unit per thread each)
  f32x4v aq[MAXT];  // (VEC implies 2 L + 1 <= MAXT: tr_launch_fused2)
  float xq2[XQ];
  if constexpr (VEC) {
#pragma unroll
    for (int i = 0; i < XQ; ++i) {
      if (TRF_THREADS * i >= TRF_ROWS * D) break;  // uniform
      int idx = tid + TRF_THREADS * i;
      idx = idx < TRF_ROWS * D ? idx : TRF_ROWS * D - 1;
      const int r = idx / D;
      const int row = r0 + r < n ? r0 + r : n - 1;
      xq2[i] = a.X[(long long)row * D + (idx - r * D)];
    }
    const int quads = TRF_ROWS * H / 4;
    long long g = (long long)r0 * H + 4 * (tid < quads ? tid : quads - 1);
    g = g < nH - 4 ? g : nH - 4;
    trf2_static_for<MAXT>([&](auto ti) {
      constexpr int t = decltype(ti)::value;
      const int tc = t < 2 * L + 1 ? t : 2 * L;
      const float* src = (tc <= L) ? a.h + tc * nH : a.u + (tc - L - 1) * nH;
      aq[t] = *(const f32x4v*)(src + g);
    });
  } else {
    trf2_rows_async(XT, ldx, a.X, D, r0, n, wave, lane);
    const int words = TRF_ROWS * H;
    // 16-byte pieces where the tiles allow it (H % 4 == 0, 16-byte aligned blocks); row by row otherwise (a hidden
    // width of 2 x an odd number of dimensions, the reference's default for such models)
    const bool quads = (H & 3) == 0 && (nH & 3) == 0 && ((((unsigned long long)a.h) | ((unsigned long long)a.u)) & 15) == 0;
    for (int t = 0; t < 2 * L + 1; ++t) {
      const float* src = (t <= L) ? a.h + t * nH : a.u + (t - L - 1) * nH;
      if (quads) {
        for (int piece = wave; piece * 256 < words; piece += TRF_WAVES) {
          const int word = piece * 256 + lane * 4;
          if (word < words) {
            long long g = (long long)r0 * H + word;
            g = g < nH - 4 ? g : nH - 4;
            nsa_glds16((const uint32_t*)(src + g), ACT + t * words + piece * 256);
          }
        }
      } else {
        trf2_rows_async(ACT + t * words, H, src, H, r0, n, wave, lane);
      }
    }
  }
  constexpr int KSF = TRF_WAVES / HT;
  const float* Wf = P + a.off[2 + 4 * L];
  const int pairs_f = (np + 1) >> 1;
  const int ppf = (pairs_f + KSF - 1) / KSF;  // <= TRF2_FQ
  const int itf = wave % HT, ksf = wave / HT;
  // VEC: group g = 2 ksf + half of the 2 KSF groups owns the 16-byte units [g QG, (g + 1) QG) of the np parameters
  const int nq_f = np >> 2;
  const int QG = (nq_f + 2 * KSF - 1) / (2 * KSF);  // <= TRF2_FQ / 4
  const int qd0 = (2 * ksf + half) * QG;
  float wfd[TRF2_FQ];
  {
    const int cI = itf * 32 + j;
    const unsigned cc = (unsigned)(cI < H ? cI : H - 1);
    if constexpr (VEC) {
      const float* WfT = a.wt + a.off[2 + 4 * L] + (long long)cc * np;  // W^T[in][o]: a lane's run is contiguous
#pragma unroll
      for (int t = 0; t < TRF2_FQ / 4; ++t) {
        const int qd = qd0 + t;
        const float4 v = *(const float4*)(WfT + 4 * (qd < nq_f ? qd : nq_f - 1));
        wfd[4 * t] = v.x; wfd[4 * t + 1] = v.y; wfd[4 * t + 2] = v.z; wfd[4 * t + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int q = 0; q < TRF2_FQ; ++q) {
        const int k = 2 * (ksf * ppf + q) + half;
        wfd[q] = Wf[(unsigned)(k < np ? k : np - 1) * (unsigned)H + cc];
      }
    }
  }
  float wc[TRF2_PP], wn[TRF2_PP];
  if (L > 0) trf2_load_wx<HT>(wn, P + a.off[4 + 4 * (L - 1)], H, H, wave, lane);
  else trf2_load_wx<HT>(wn, P + a.off[0], H, a.d_id, wave, lane);

  TRF_MARK(40);
  // ---- transform backward: parameters -> their gradient in place, dL/dx_tr, x restored ----
#pragma unroll
  for (int i = 0; i < IQ; ++i) {
    const int e = tid + TRF_THREADS * i;
    if (e < TRF_ROWS * a.d_tr) {
      const int r = e / a.d_tr, m = e - r * a.d_tr;
      const int col = colv[i];
      float* pit = PT + r * pld;
      if (r < rows_here) {
        const float gy = GXT[r * ldx + col];
        float gx;
        if constexpr (K > 0) {
          gx = tr_rq_backward_core<K>(pit + m * (3 * K - 1), xs[i], gy, gldv[i], a.c, pit + m * (3 * K - 1));
        } else {
          gx = gy;
          if (a.transform != NSA_T_ADDITIVE) {
            const float sig = 1.0f / (1.0f + expf(-(pit[a.d_tr + m] + a.c.aff_a)));
            const float scale = sig + a.c.aff_b;
            pit[a.d_tr + m] = (gy * xs[i] + gldv[i] / scale) * sig * (1.0f - sig);
            gx = gy * scale;
          }
          pit[m] = gy;
        }
        const long long g = (long long)(r0 + r) * D + col;
        a.X[g] = xs[i];
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
  }
  TRF_MARK(41);
  if constexpr (VEC) {
#pragma unroll
    for (int i = 0; i < XQ; ++i) {
      if (TRF_THREADS * i >= TRF_ROWS * D) break;  // uniform
      const int idx = tid + TRF_THREADS * i;
      const int r = idx / D;
      if (idx < TRF_ROWS * D) XT[r * ldx + (idx - r * D)] = xq2[i];
    }
    const int quads = TRF_ROWS * H / 4;
    trf2_static_for<MAXT>([&](auto ti) {
      constexpr int t = decltype(ti)::value;
      if (t < 2 * L + 1 && tid < quads) *(f32x4v*)(ACT + t * TRF_ROWS * H + 4 * tid) = aq[t];
    });
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the async x rows and activation tiles
  }
  trf_barrier_lds();
  TRF_MARK(31);

  // ---- final layer: gh = gp . Wf (slices of the np parameters), dWf = gp^T . h_L, dbf ----
  const float* hL = ACT + L * TRF_ROWS * H;
  if constexpr (VEC) {
    f32x16 acc = {0}, acc2 = {0};
    const float* ax = PT + j * pld;
#pragma unroll
    for (int t0 = 0; t0 < TRF2_FQ / 4; t0 += 4) {
      if (t0 < QG) {  // uniform
        float4 av[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int qd = qd0 + t0 + t;
          av[t] = *(const float4*)(ax + 4 * (qd < nq_f ? qd : 0));
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float keep = (t0 + t < QG && qd0 + t0 + t < nq_f) ? 1.0f : 0.0f;
          av[t].x *= keep; av[t].y *= keep; av[t].z *= keep; av[t].w *= keep;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (t0 + t < QG) {
            acc = nsa_mfma(av[t].x, wfd[4 * (t0 + t)], acc);
            acc2 = nsa_mfma(av[t].y, wfd[4 * (t0 + t) + 1], acc2);
            acc = nsa_mfma(av[t].z, wfd[4 * (t0 + t) + 2], acc);
            acc2 = nsa_mfma(av[t].w, wfd[4 * (t0 + t) + 3], acc2);
          }
        }
      }
    }
    float* pp_ = part + wave * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) pp_[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + j] = acc[r] + acc2[r];
  } else {
    f32x16 acc = {0}, acc2 = {0};
    const float* ax = PT + j * pld;
#pragma unroll
    for (int q0 = 0; q0 < TRF2_FQ; q0 += 16) {
      if (q0 < ppf) {  // uniform
        float av[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int k = 2 * (ksf * ppf + q0 + q) + half;
          av[q] = ax[k < np ? k : 0];
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int k = 2 * (ksf * ppf + q0 + q) + half;
          av[q] *= (q0 + q < ppf && k < np) ? 1.0f : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < 16; q += 2) {
          if (q0 + q < ppf) {
            acc = nsa_mfma(av[q], wfd[q0 + q], acc);
            acc2 = nsa_mfma(av[q + 1], wfd[q0 + q + 1], acc2);
          }
        }
      }
    }
    float* pp_ = part + wave * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) pp_[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + j] = acc[r] + acc2[r];
  }
  {
    const int OTf = (np + 31) >> 5;
    float* gWf = GR + a.off[2 + 4 * L];
    for (int t = wave; t < OTf * HT; t += TRF_WAVES) {
      const int ot = t / HT, itl = t - ot * HT;
      const int oc = ot * 32 + j, kc = itl * 32 + j;
      const int ocl = oc < np ? oc : 0, kcl = kc < H ? kc : 0;
      float av[16], bv[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = 2 * q + half;
        av[q] = PT[row * pld + ocl];
        bv[q] = hL[row * H + kcl];
      }
      f32x16 acc = {0}, acc2 = {0};
#pragma unroll
      for (int q = 0; q < 16; q += 2) {  // D[i = in][jj = out]
        acc = nsa_mfma(bv[q], av[q], acc);
        acc2 = nsa_mfma(bv[q + 1], av[q + 1], acc2);
      }
      trf2_dw_store<SLAB>(gWf, np, H, oc, itl * 32, half, acc, acc2);
    }
    float* gbf = GR + a.off[3 + 4 * L];
    for (int o = tid; o < np; o += TRF_THREADS) {
      float s = 0.0f;
#pragma unroll 8
      for (int row = 0; row < TRF_ROWS; ++row) s += PT[row * pld + o];
      trf2_grad_out<SLAB>(&gbf[o], s);
    }
  }
  trf_barrier_lds();
  {
    // gh -> G0
    constexpr int NC = 32 * HT, STEP = TRF_THREADS / NC;
    const int c = tid & (NC - 1), rb = tid / NC;
    const float* pc = part + (c >> 5) * 1024 + (c & 31);
    if (c < H) {
#pragma unroll
      for (int i = 0; i < 32 / STEP; ++i) {
        const int row = rb + STEP * i;
        float v = 0.0f;
#pragma unroll
        for (int s = 0; s < KSF; ++s) v += pc[(s * HT) * 1024 + row * 32];
        G0[row * ld + c] = v;
      }
    }
  }
  TRF_MARK(32);

  // ---- residual blocks, last to first ----
  for (int l = L - 1; l >= 0; --l) {
    const float* hl = ACT + l * TRF_ROWS * H;
    const float* ul = ACT + (L + 1 + l) * TRF_ROWS * H;
    // h_{l+1} = h_l + W2 act(u_l) + b2:  g_u = (G0 . W2) act'(u_l) -> G1
#pragma unroll
    for (int q = 0; q < TRF2_PP; ++q) wc[q] = wn[q];
    trf2_load_wx<HT>(wn, P + a.off[2 + 4 * l], H, H, wave, lane);
    trf_barrier_lds();
    trf2_back_linear<HT, false, SLAB>(G0, wc, H, ul, H, nullptr, a.act, H, ul, a.act, nullptr, G1, GR + a.off[4 + 4 * l],
                                      GR + a.off[5 + 4 * l], ld, part, wave, lane, tid);
    // u_l = W1 act(h_l) + b1:  g_h_l = G0 + (G1 . W1) act'(h_l) -> G0
#pragma unroll
    for (int q = 0; q < TRF2_PP; ++q) wc[q] = wn[q];
    if (l > 0) trf2_load_wx<HT>(wn, P + a.off[4 + 4 * (l - 1)], H, H, wave, lane);
    else trf2_load_wx<HT>(wn, P + a.off[0], H, a.d_id, wave, lane);
    trf_barrier_lds();
    trf2_back_linear<HT, false, SLAB>(G1, wc, H, hl, H, nullptr, a.act, H, hl, a.act, G0, G0, GR + a.off[2 + 4 * l],
                                      GR + a.off[3 + 4 * l], ld, part, wave, lane, tid);
    TRF_MARK(33 + (L - 1 - l));
  }

  // ---- initial layer: dW0 = G0^T . x_id, g_x[:, id_cols] += G0 . W0 ----
#pragma unroll
  for (int q = 0; q < TRF2_PP; ++q) wc[q] = wn[q];
  trf_barrier_lds();
  trf2_back_linear<HT, true, SLAB>(G0, wc, H, XT, ldx, a.id_cols, ACT_NONE, a.d_id, nullptr, ACT_NONE, nullptr, G1,
                                   GR + a.off[0], GR + a.off[1], ld, part, wave, lane, tid);
  trf_barrier_lds();
  for (int idx = tid; idx < TRF_ROWS * a.d_id; idx += TRF_THREADS) {
    const int r = idx / a.d_id, k = idx - r * a.d_id;
    if (r < rows_here) {
      const int col = a.id_cols[k];
      a.Gx[(long long)(r0 + r) * D + col] = GXT[r * ldx + col] + G1[r * ld + k];
    }
  }
  TRF_MARK(60);
}
