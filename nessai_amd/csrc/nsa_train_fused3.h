// Third generation of the fused training kernels (included by nsa_train.hip after nsa_train_fused2.h; same TrfArgs,
// same workspace layout, same results up to the order of the float32 sums): the second generation's scheme for
// conditioners 65 to 128 wide (config 4's importance-sampling flows: 64 inputs, 128-wide residual nets, 736 spline
// parameters per row), one coupling layer per launch.
//
// What the first generation spends its time on at this width (profiles/r04_train_c4_kernels.txt: forward 65 us,
// backward 118 us per coupling for a 32-row workgroup whose MFMAs take 20 / 40 us): a hidden layer has four output
// tiles, so four of the eight waves own one and four idle; every wave requests 4 KB of weights per step and waits a
// round trip (~2300 cycles: the flow's 4 MB of weights do not stay in the 4 MB L2 of an XCD across a step) for them;
// the final layer's gradient travels through global memory in chunks, each a round trip and two barriers, and its
// weight gradient is added with device-scope float atomics from every workgroup.  Here
//   * a hidden layer is split over all 8 waves - output tile x half of k - with the whole layer's operands (64 KB
//     per workgroup) requested ONE LAYER AHEAD into registers, the partial sums meeting in LDS (trf2_layer's scheme);
//   * the final layer (23 output tiles: three rounds of eight) streams its operands through two register buffers
//     of half a tile's k each, refilled two steps ahead, and leaves the parameters in an LDS tile that takes the
//     place of the partial sums (the transform reads it there; 16-byte stores save it for the backward pass);
//   * the backward pass keeps the parameter tile in LDS, overwrites it by its own gradient, and runs dX and dW of
//     every layer on all 8 waves; weight gradients go to the workgroup's own copy of the flat vector (plain
//     stores, summed by tr_finish_grads in a fixed order).
// Reference semantics: glasflow.nflows ResidualNet (call sites src/nessai/flows/nsf.py:85-95, realnvp.py:136-146)
// under FlowModel._train (src/nessai/flowmodel/base.py:365-452; the importance sampler's
// src/nessai/flowmodel/importance.py:80-142).
#pragma once

#define TRF3_HT 4   // output tiles of a hidden layer (H <= 128)
#define TRF3_PP 32  // k pairs per item of a hidden layer: 64 pairs / 2 slices

// B operands of this wave's item of a [rows x Kd] . [Kd x O] product, O <= 128: tile ot = wave % 4, k slice
// ks = wave / 4 (see trf2_load_w for why every load is unconditional from a clamped address)
// FULL: Kd == O == 128 - every operand of every item is inside the matrix: one base address per lane, the loads a
// constant stride apart, nothing to clamp (and no clamped offsets for the compiler to keep across the layers)
template <bool FULL>
__device__ __forceinline__ void trf3_load_w(float (&w)[TRF3_PP], float& bias, const float* __restrict__ Wt,
                                            const float* __restrict__ b, int Kd, int O, int wave, int lane, int tid) {
  const int half = lane >> 5, j = lane & 31;
  if constexpr (FULL) {
    // (a uniform base per load - scalar arithmetic - plus ONE 32-bit lane offset: written as a lane-dependent
    // pointer a constant stride apart, the compiler kept 32 64-bit addresses per matrix, on the stack)
    const unsigned lane_off = 4u * ((unsigned)half * 128u + (unsigned)j);
    const float* base = Wt + (2 * (wave >> 2) * TRF3_PP) * 128 + (wave & 3) * 32;
#pragma unroll
    for (int q = 0; q < TRF3_PP; ++q) w[q] = *(const float*)((const char*)(base + q * 256) + lane_off);
    bias = b[tid & 127];
  } else {
    const Trf2Item<TRF3_HT> it(wave, Kd);
    const int o = it.ot * 32 + j;
    const unsigned oc = (unsigned)(o < O ? o : O - 1);
#pragma unroll
    for (int q = 0; q < TRF3_PP; ++q) {
      const int k = 2 * (it.ks * it.pp + q) + half;
      w[q] = Wt[(unsigned)(k < Kd ? k : Kd - 1) * (unsigned)O + oc];
    }
    const int oc2 = tid & 127;
    bias = b[oc2 < O ? oc2 : O - 1];
  }
}

// out[row][o] = sum_k act(Ain[row][k]) W[o][k] + b[o] (+ R[row][o]) for the workgroup's 32 rows, O <= 128.
// Ends with the output tile WRITTEN but not yet published (the caller's barrier does that).
template <bool GATHER, bool FULL>
__device__ __forceinline__ void trf3_layer(const float* __restrict__ Ain, int act_in, const float (&w)[TRF3_PP],
                                           const int (&cidx)[TRF3_PP], float bias, int Kd, int O, const float* R,
                                           float* out_tile, float* __restrict__ out_glb, int ld, int r0, int n,
                                           float* __restrict__ part, int wave, int lane, int tid) {
  const int half = lane >> 5, j = lane & 31;
  float av[TRF3_PP];
  f32x16 acc = {0}, acc2 = {0};
  if constexpr (FULL && !GATHER) {
    const float* ax = Ain + j * ld + 2 * (wave >> 2) * TRF3_PP + half;
#pragma unroll
    for (int q = 0; q < TRF3_PP; ++q) av[q] = ax[2 * q];
    trf_act_batch(av, act_in);
#pragma unroll
    for (int q = 0; q < TRF3_PP; q += 2) {
      acc = nsa_mfma(av[q], w[q], acc);
      acc2 = nsa_mfma(av[q + 1], w[q + 1], acc2);
    }
  } else {
    const Trf2Item<TRF3_HT> it(wave, Kd);
#pragma unroll
    for (int q = 0; q < TRF3_PP; ++q) {
      const int k = 2 * (it.ks * it.pp + q) + half;
      av[q] = GATHER ? Ain[cidx[q]] : Ain[j * ld + (k < Kd ? k : 0)];
    }
    trf_act_batch(av, act_in);
#pragma unroll
    for (int q = 0; q < TRF3_PP; ++q) {  // a factor, not a select: see trf2_load_w
      const int k = 2 * (it.ks * it.pp + q) + half;
      av[q] *= (q < it.pp && k < Kd) ? 1.0f : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < TRF3_PP; q += 2) {
      if (q < it.pp) {  // wave-uniform
        acc = nsa_mfma(av[q], w[q], acc);
        acc2 = nsa_mfma(av[q + 1], w[q + 1], acc2);
      }
    }
  }
  float* pp_ = part + wave * 1024;
#pragma unroll
  for (int r = 0; r < 16; ++r) pp_[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + j] = acc[r] + acc2[r];
  trf_barrier_lds();
  // 32 x 128 outputs over 512 threads: column tid % 128, rows 8 (tid / 128) + i (one address per thread and
  // immediate offsets for the saves: with rows 4 i apart the compiler kept eight 64-bit addresses per saved tile
  // across the layers - on the stack)
  const int o = tid & 127, rb = 8 * (tid >> 7);
  const float* pc = part + (o >> 5) * 1024 + rb * 32 + (o & 31);
  if (o < O) {
    float* og = out_glb + (unsigned)(r0 + rb) * (unsigned)O + (unsigned)o;
    float* ot = out_tile + rb * ld + o;
    const float* rt = R ? R + rb * ld + o : nullptr;
    const int rows_ok = n - r0 - rb;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float v = bias + pc[i * 32] + pc[TRF3_HT * 1024 + i * 32];
      if (rt) v += rt[i * ld];
      ot[i * ld] = v;
      if (i < rows_ok) og[i * O] = v;
    }
  }
}

// LDS floats of trf3_forward: the x tile, two activation tiles, the partial sums / the parameter tile, log|det J|
static inline size_t trf3_forward_floats(const TrfArgs& a) {
  const size_t un = (size_t)TRF_ROWS * a.np > (size_t)TRF_WAVES * 1024 ? (size_t)TRF_ROWS * a.np : (size_t)TRF_WAVES * 1024;
  return (size_t)TRF_ROWS * trf2_ldx(a) + (size_t)2 * TRF_ROWS * a.ld + un + TRF_ROWS;
}

template <int K, bool FULL>  // spline bins (0: affine / additive); FULL: H == 128
__global__ void __launch_bounds__(TRF_THREADS) trf3_forward(TrfArgs a) {
  extern __shared__ float trf_sm[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, j = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r0 = blockIdx.x * TRF_ROWS;
  const int ld = a.ld, n = a.n, H = a.H, D = a.D, L = a.L, np = a.np, d_id = a.d_id, d_tr = a.d_tr;
  const int ldx = (D + 1) | 1;
  float* XT = trf_sm;                // [32][ldx] the workgroup's rows of x, all columns
  float* HC = XT + TRF_ROWS * ldx;   // h_l
  float* U = HC + TRF_ROWS * ld;     // u_l
  float* part = U + TRF_ROWS * ld;   // [8 items][32][32] partial sums of the conditioner's layers, then
  float* PT = part;                  // [32][np] the transform's parameters
  const size_t un = (size_t)TRF_ROWS * np > (size_t)TRF_WAVES * 1024 ? (size_t)TRF_ROWS * np : (size_t)TRF_WAVES * 1024;
  float* ldsum = part + un;          // [32] log|det J| per row
  const float* P = a.params;
  const float* WT = a.wt;
  const long long nH = (long long)n * H;
  const long long t_start = clock64();
  if (tid < TRF_ROWS) ldsum[tid] = 0.0f;

  trf2_rows_async(XT, ldx, a.X, D, r0, n, wave, lane);
  int cidx[TRF3_PP], none[TRF3_PP];
  float wc[TRF3_PP], wn[TRF3_PP], fb[TRF3_PP], bc, bn = 0.0f;
  {
    const Trf2Item<TRF3_HT> it0(wave, d_id);
#pragma unroll
    for (int q = 0; q < TRF3_PP; ++q) {
      const int k = 2 * (it0.ks * it0.pp + q) + half;
      cidx[q] = a.id_cols[k < d_id ? k : 0] + j * ldx;
      none[q] = 0;
    }
    trf3_load_w<false>(wc, bc, WT + a.off[0], P + a.off[1], d_id, H, wave, lane, tid);
  }
  // the final layer: output tiles wave, wave + 8, wave + 16 (np <= 768), each in two steps of half the k; the
  // operands of step (round s, half h) - W^T[64 h + 2 q + half][32 (wave + 8 s) + j], 32 per lane
  const float* Wf = WT + a.off[2 + 4 * L];
  const float* bf = P + a.off[3 + 4 * L];
  const int OTf = (np + 31) >> 5;
  auto load_final = [&](float (&buf)[TRF3_PP], int s, int h) {
    const int o = (wave + TRF_WAVES * s) * 32 + j;
    const unsigned oc = (unsigned)(o < np ? o : np - 1);
    if constexpr (FULL) {
      const unsigned lane_off = 4u * ((unsigned)half * (unsigned)np + oc);
#pragma unroll
      for (int q = 0; q < TRF3_PP; ++q)
        buf[q] = *(const float*)((const char*)(Wf + (long long)(64 * h + 2 * q) * np) + lane_off);
    } else {
#pragma unroll
      for (int q = 0; q < TRF3_PP; ++q) {
        const int k = 64 * h + 2 * q + half;
        buf[q] = Wf[(unsigned)(k < H ? k : H - 1) * (unsigned)np + oc];
      }
    }
  };
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the async rows (and layer 0)
  trf_barrier_lds();
  TRF_MARK(0);

  if (L > 0) trf3_load_w<FULL>(wn, bn, WT + a.off[2], P + a.off[3], H, H, wave, lane, tid);
  else { load_final(wn, 0, 0); load_final(fb, 0, 1); }
  // h_0 = W0 x_id + b0
  trf3_layer<true, false>(XT, ACT_NONE, wc, cidx, bc, d_id, H, nullptr, HC, a.h, ld, r0, n, part, wave, lane, tid);
  TRF_MARK(1);
  // u_l = W1 act(h_l) + b1;  h_{l+1} = h_l + W2 act(u_l) + b2.  The last block stands outside the loop: the final
  // layer's first operands are requested in front of its second layer, and requested inside the loop their 64
  // addresses are loop invariants the compiler computes ahead of it and keeps - on the stack
  auto first = [&](int l) {
#pragma unroll
    for (int q = 0; q < TRF3_PP; ++q) wc[q] = wn[q];
    bc = bn;
    trf3_load_w<FULL>(wn, bn, WT + a.off[4 + 4 * l], P + a.off[5 + 4 * l], H, H, wave, lane, tid);
    trf_barrier_lds();
    trf3_layer<false, FULL>(HC, a.act, wc, none, bc, H, H, nullptr, U, a.u + l * nH, ld, r0, n, part, wave, lane, tid);
    TRF_MARK(2 + 2 * l);
#pragma unroll
    for (int q = 0; q < TRF3_PP; ++q) wc[q] = wn[q];
    bc = bn;
  };
  for (int l = 0; l + 1 < L; ++l) {
    first(l);
    trf3_load_w<FULL>(wn, bn, WT + a.off[6 + 4 * l], P + a.off[7 + 4 * l], H, H, wave, lane, tid);
    trf_barrier_lds();
    trf3_layer<false, FULL>(U, a.act, wc, none, bc, H, H, HC, HC, a.h + (l + 1) * nH, ld, r0, n, part, wave, lane, tid);
    TRF_MARK(3 + 2 * l);
  }
  if (L > 0) {
    first(L - 1);
    load_final(wn, 0, 0);
    load_final(fb, 0, 1);
    trf_barrier_lds();
    trf3_layer<false, FULL>(U, a.act, wc, none, bc, H, H, HC, HC, a.h + L * nH, ld, r0, n, part, wave, lane, tid);
    TRF_MARK(1 + 2 * L);
  }
  trf_barrier_lds();  // h_L published; the partial sums are dead: the parameter tile takes their place
  // parameters = Wf h_L + bf
  const bool p_vec = (np & 3) == 0 && (((unsigned long long)a.p) & 15) == 0;
  {
    f32x16 acc = {0}, acc2 = {0};
    const float* ax = HC + j * ld;
    trf2_static_for<3>([&](auto si) {
      constexpr int s = decltype(si)::value;
      const int t = wave + TRF_WAVES * s;
      trf2_static_for<2>([&](auto hi) {
        constexpr int h = decltype(hi)::value;
        float (&buf)[TRF3_PP] = h == 0 ? wn : fb;
        if (t < OTf) {  // wave-uniform
          float av[TRF3_PP];
          if (h == 0) {
            acc = f32x16{0};
            acc2 = f32x16{0};
          }
          if constexpr (FULL) {
#pragma unroll
            for (int q = 0; q < TRF3_PP; ++q) av[q] = ax[64 * h + 2 * q + half];
#pragma unroll
            for (int q = 0; q < TRF3_PP; q += 2) {
              acc = nsa_mfma(av[q], buf[q], acc);
              acc2 = nsa_mfma(av[q + 1], buf[q + 1], acc2);
            }
          } else {
#pragma unroll
            for (int q = 0; q < TRF3_PP; ++q) {
              const int k = 64 * h + 2 * q + half;
              av[q] = ax[k < H ? k : 0];
            }
#pragma unroll
            for (int q = 0; q < TRF3_PP; ++q) av[q] *= (64 * h + 2 * q + half < H) ? 1.0f : 0.0f;
#pragma unroll
            for (int q = 0; q < TRF3_PP; q += 2) {
              if (64 * h + 2 * q < H) {  // uniform
                acc = nsa_mfma(av[q], buf[q], acc);
                acc2 = nsa_mfma(av[q + 1], buf[q + 1], acc2);
              }
            }
          }
        }
        // the operands of the step two ahead (the same half of the next round's tile) into the buffer just read
        if (s + 1 < 3) {
          if (wave + TRF_WAVES * (s + 1) < OTf) load_final(buf, s + 1, h);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (h == 1 && t < OTf) {
          const int o = t * 32 + j;
          if (o < np) {
            const float bias = bf[o];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int rl = (r & 3) + 8 * (r >> 2) + 4 * half;
              const float v = acc[r] + acc2[r] + bias;
              PT[rl * np + o] = v;
              if (!p_vec && r0 + rl < n) a.p[(unsigned)(r0 + rl) * (unsigned)np + (unsigned)o] = v;
            }
          }
        }
      });
    });
  }
  TRF_MARK(20);
  trf_barrier_lds();
  TRF_MARK(21);
  if (p_vec) {  // the parameter tile for the backward pass: 16-byte stores out of LDS
    const int rows_here = n - r0 < TRF_ROWS ? n - r0 : TRF_ROWS;
    const int quads = rows_here * (np >> 2);
    float* dst = a.p + (long long)r0 * np;
    for (int i = tid; i < quads; i += TRF_THREADS) *(f32x4v*)(dst + 4 * i) = *(const f32x4v*)(PT + 4 * i);
  }
  const int items = TRF_ROWS * d_tr;
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
  if (tid < TRF_ROWS && r0 + tid < n) a.logdet[r0 + tid] += ldsum[tid];
}

// ---------------------------------------------------------------------------------------------------
// backward pass of the coupling layer, H == 128 (see the header of this file)
//
// LDS (floats): R [region]  |  HL [32][128] h_L  |  part [8][1024]  |  G0 [32][ld]
//   region, first:  PT [32][pld] the parameters, overwritten in place by their gradient (transform items), read by
//                   the final layer's two products;
//   region, then:   ACT [2 L][32][128] (h_0 .. h_{L-1}, u_0 .. u_{L-1}) | G1 [32][ld] | XT [32][ldx] - the tiles of
//                   the residual blocks and of the initial layer, requested into REGISTERS while the final layer's
//                   weight gradient runs and stored once the parameter tile is dead.
// ---------------------------------------------------------------------------------------------------
__host__ __device__ static inline int trf3_backward_region(const TrfArgs& a) {
  const int pt = TRF_ROWS * trf2_pld(a.np, true);
  const int blocks = 2 * a.L * TRF_ROWS * 128 + TRF_ROWS * a.ld + TRF_ROWS * ((a.D + 1) | 1);
  return pt > blocks ? pt : blocks;
}
static inline size_t trf3_backward_floats(const TrfArgs& a) {
  return (size_t)trf3_backward_region(a) + (size_t)TRF_ROWS * 128 + (size_t)TRF_WAVES * 1024 + (size_t)TRF_ROWS * a.ld;
}

// B operands of a dX item of a layer with 128 outputs: W[out = 64 ks + 2 q + half][in = 32 it + j], natural layout
__device__ __forceinline__ void trf3_load_wx(float (&w)[TRF3_PP], const float* __restrict__ W, int Kin, int wave,
                                             int lane) {
  const int half = lane >> 5, j = lane & 31;
  const int c = (wave & 3) * 32 + j;
  const unsigned lane_off = 4u * ((unsigned)half * (unsigned)Kin + (unsigned)(c < Kin ? c : Kin - 1));
  const float* base = W + (long long)(64 * (wave >> 2)) * Kin;
#pragma unroll
  for (int q = 0; q < TRF3_PP; ++q) w[q] = *(const float*)((const char*)(base + (long long)(2 * q) * Kin) + lane_off);
}

// One linear layer (128 outputs) of the backward pass: see trf2_back_linear; here dX is in-tile x HALF of `out`
// over the 8 waves and dW two (out-tile, in-tile) pairs per wave.
//   G [32][ld]: gradient w.r.t. the layer's output;  w: this wave's dX operands
//   Ain: the layer's input tile, row stride a_ld (GATHER: column of slot c = cols[c]);  Kin inputs
//   Z [32][128]: pre-activation whose act' multiplies dX (null: none);  Rs: tile added to the result (null: none)
template <bool GATHER, bool SLAB>
__device__ __forceinline__ void trf3_back_linear(const float* __restrict__ G, const float (&w)[TRF3_PP],
                                                 const float* __restrict__ Ain, int a_ld, const int* __restrict__ cols,
                                                 int act_in, int Kin, const float* __restrict__ Z, int act_z,
                                                 const float* Rs, float* out, float* __restrict__ gW,
                                                 float* __restrict__ gb, int ld, float* __restrict__ part, int wave,
                                                 int lane, int tid) {
  const int half = lane >> 5, j = lane & 31;
  const int IT = (Kin + 31) >> 5;  // <= 4
  const int it = wave & 3, ks = wave >> 2;
  {
    f32x16 acc = {0}, acc2 = {0};
    if (it < IT) {  // wave-uniform
      float av[TRF3_PP];
      const float* ag = G + j * ld + 64 * ks + half;
#pragma unroll
      for (int q = 0; q < TRF3_PP; ++q) av[q] = ag[2 * q];
#pragma unroll
      for (int q = 0; q < TRF3_PP; q += 2) {
        acc = nsa_mfma(av[q], w[q], acc);
        acc2 = nsa_mfma(av[q + 1], w[q + 1], acc2);
      }
    }
    float* pp_ = part + wave * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) pp_[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + j] = acc[r] + acc2[r];
  }
  {
    // dW: out-tile ot = wave % 4, in-tiles 2 (wave / 4) and 2 (wave / 4) + 1, all 32 rows
    const int ot = wave & 3;
    float av[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) av[q] = G[(2 * q + half) * ld + ot * 32 + j];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int itl = 2 * (wave >> 2) + s;
      if (itl < IT) {  // wave-uniform
        const int kc = itl * 32 + j;
        const int kcl = kc < Kin ? kc : 0;
        const int acol = GATHER ? cols[kcl] : kcl;
        float bv[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) bv[q] = Ain[(2 * q + half) * a_ld + acol];
        trf_act_batch(bv, act_in);
        f32x16 acc = {0}, acc2 = {0};
#pragma unroll
        for (int q = 0; q < 16; q += 2) {  // D[i = in][jj = out]
          acc = nsa_mfma(bv[q], av[q], acc);
          acc2 = nsa_mfma(bv[q + 1], av[q + 1], acc2);
        }
        trf2_dw_store<SLAB>(gW, 128, Kin, ot * 32 + j, itl * 32, half, acc, acc2);
      }
    }
    if (tid < 128) {
      float s = 0.0f;
#pragma unroll 8
      for (int row = 0; row < TRF_ROWS; ++row) s += G[row * ld + tid];
      trf2_grad_out<SLAB>(&gb[tid], s);
    }
  }
  trf_barrier_lds();
  // sum the two slices; act'(z), residual path
  const int c = tid & 127, rb = 8 * (tid >> 7);
  const float* pc = part + (c >> 5) * 1024 + rb * 32 + (c & 31);
  if (c < Kin) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float v = pc[i * 32] + pc[TRF3_HT * 1024 + i * 32];
      if (Z) {
        const float z = Z[(rb + i) * 128 + c];
        if (act_z == NSA_ACT_RELU) v = z > 0.0f ? v : 0.0f;
        else v *= tr_act_grad(z, act_z);
      }
      if (Rs) v += Rs[(rb + i) * ld + c];
      out[(rb + i) * ld + c] = v;
    }
  }
}

#define TRF3_CQ 8  // 16-byte units per chunk of the final layer's dX operands (two chunks in registers)

template <int K, bool SLAB>
__global__ void __launch_bounds__(TRF_THREADS) trf3_backward(TrfArgs a) {
  extern __shared__ float trf_sm[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, j = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r0 = blockIdx.x * TRF_ROWS;
  constexpr int H = 128;
  const int ld = a.ld, n = a.n, np = a.np, D = a.D, L = a.L;
  const int ldx = (D + 1) | 1, pld = trf2_pld(np, true);
  float* PT = trf_sm;  // [32][pld] parameters, then their gradient
  float* HL = PT + trf3_backward_region(a);  // [32][128] h_L
  float* part = HL + TRF_ROWS * H;
  float* G0 = part + TRF_WAVES * 1024;
  float* ACT = PT;                            // once the parameter tile is dead
  float* G1 = ACT + 2 * L * TRF_ROWS * H;
  float* XT = G1 + TRF_ROWS * ld;
  const float* P = a.params;
  float* GR = SLAB ? a.gslab + (long long)blockIdx.x * a.npar : a.grads;
  const long long nH = (long long)n * H;
  const long long t_start = clock64();
  const int rows_here = (n - r0 < TRF_ROWS) ? (n - r0) : TRF_ROWS;

  // ---- the parameter tile (16-byte loads through registers), h_L, what the transform items need ----
  constexpr int PQ4 = 13;  // 16-byte units of the parameter tile per thread: 32 (768 + 4) / 4 <= 13 * 512
  f32x4v pq[PQ4];
  const int rq = pld >> 2, nq = np >> 2;
#pragma unroll
  for (int i = 0; i < PQ4; ++i) {
    if (TRF_THREADS * i >= TRF_ROWS * rq) break;  // uniform
    int Q = tid + TRF_THREADS * i;
    Q = Q < TRF_ROWS * rq ? Q : TRF_ROWS * rq - 1;
    const int r = Q / rq, cq = Q - r * rq;
    const int row = r0 + r < n ? r0 + r : n - 1;
    pq[i] = *(const f32x4v*)(a.p + (long long)row * np + 4 * (cq < nq ? cq : nq - 1));
  }
  // a [32][128] tile of the saved activations: two 16-byte units per thread
  auto tile_load = [&](f32x4v (&q)[2], const float* src) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      long long g = (long long)r0 * H + 4 * (tid + TRF_THREADS * i);
      g = g < nH - 4 ? g : nH - 4;
      q[i] = *(const f32x4v*)(src + g);
    }
  };
  auto tile_store = [&](const f32x4v (&q)[2], float* dst) {
#pragma unroll
    for (int i = 0; i < 2; ++i) *(f32x4v*)(dst + 4 * (tid + TRF_THREADS * i)) = q[i];
  };
  f32x4v hq[2];
  tile_load(hq, a.h + L * nH);
  constexpr int IQ = 2;  // d_tr <= 32
  float xs[IQ], gldv[IQ], gyv[IQ];
  int colv[IQ];
#pragma unroll
  for (int i = 0; i < IQ; ++i) {
    const int e = tid + TRF_THREADS * i;
    const int r = e / a.d_tr, m = e - r * a.d_tr;
    const int row = (r < rows_here) ? r0 + r : n - 1;
    const int mc = m < a.d_tr ? m : 0;
    xs[i] = a.xsave[row * a.d_tr + mc];
    gldv[i] = a.gld[row];
    colv[i] = a.tr_cols[mc];
    gyv[i] = a.Gx[(long long)row * D + colv[i]];
  }
#pragma unroll
  for (int i = 0; i < PQ4; ++i) {
    if (TRF_THREADS * i >= TRF_ROWS * rq) break;  // uniform
    const int Q = tid + TRF_THREADS * i;
    const int r = Q / rq, cq = Q - r * rq;
    if (Q < TRF_ROWS * rq && cq < nq) *(f32x4v*)(PT + 4 * Q) = pq[i];
  }
  tile_store(hq, HL);
  trf_barrier_lds();
  TRF_MARK(30);

  // ---- requested now, used after the items: the first two chunks of the final layer's dX operands.  Group
  // g = 2 ksf + half of the four groups of an in-tile owns the 16-byte units [g QG, (g + 1) QG) of the np parameters:
  // a lane's run of four consecutive parameters is one 16-byte LDS read (A) and one 16-byte load from the
  // transposed weights (B) for four MFMA steps
  const int itf = wave & 3, ksf = wave >> 2;
  const int QG = (nq + 3) >> 2;  // <= 48: six chunks
  const int qd0 = (2 * ksf + half) * QG;
  const float* WfT = a.wt + a.off[2 + 4 * L] + (long long)(itf * 32 + j) * np;  // W^T[in][o]
  auto load_chunk = [&](f32x4v (&b)[TRF3_CQ], int c) {
#pragma unroll
    for (int t = 0; t < TRF3_CQ; ++t) {
      const int qd = qd0 + TRF3_CQ * c + t;
      b[t] = *(const f32x4v*)(WfT + 4 * (qd < nq ? qd : nq - 1));
    }
  };
  f32x4v wa[TRF3_CQ], wb[TRF3_CQ];
  load_chunk(wa, 0);
  load_chunk(wb, 1);

  // ---- transform backward: parameters -> their gradient in place, dL/dx_tr, x restored ----
#pragma unroll
  for (int i = 0; i < IQ; ++i) {
    const int e = tid + TRF_THREADS * i;
    if (e < TRF_ROWS * a.d_tr) {
      const int r = e / a.d_tr, m = e - r * a.d_tr;
      float* pit = PT + r * pld;
      if (r < rows_here) {
        const float gy = gyv[i];
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
        const long long g = (long long)(r0 + r) * D + colv[i];
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
  trf_barrier_lds();
  TRF_MARK(31);

  // ---- final layer, dX: gh = gp . Wf, in-tile x half of the parameters per wave, operands streamed ----
  {
    f32x16 acc = {0}, acc2 = {0};
    const float* ax = PT + j * pld;
    trf2_static_for<6>([&](auto ci) {
      constexpr int c = decltype(ci)::value;
      f32x4v (&b)[TRF3_CQ] = (c & 1) ? wb : wa;
      if (TRF3_CQ * c < QG) {  // uniform
        f32x4v av[TRF3_CQ];
#pragma unroll
        for (int t = 0; t < TRF3_CQ; ++t) {
          const int qd = qd0 + TRF3_CQ * c + t;
          av[t] = *(const f32x4v*)(ax + 4 * (qd < nq ? qd : 0));
        }
#pragma unroll
        for (int t = 0; t < TRF3_CQ; ++t) {
          const float keep = (TRF3_CQ * c + t < QG && qd0 + TRF3_CQ * c + t < nq) ? 1.0f : 0.0f;
          av[t] *= keep;
        }
#pragma unroll
        for (int t = 0; t < TRF3_CQ; ++t) {
          acc = nsa_mfma(av[t][0], b[t][0], acc);
          acc2 = nsa_mfma(av[t][1], b[t][1], acc2);
          acc = nsa_mfma(av[t][2], b[t][2], acc);
          acc2 = nsa_mfma(av[t][3], b[t][3], acc2);
        }
        if (c + 2 < 6) {
          if (TRF3_CQ * (c + 2) < QG) load_chunk(b, c + 2);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    });
    float* pp_ = part + wave * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) pp_[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + j] = acc[r] + acc2[r];
  }
  TRF_MARK(42);
  // ---- requested now, stored once the parameter tile is dead: the tiles of the residual blocks, the rows of x,
  // and the first backward step of the hidden layers (W2 of the last block, or W0 when there is none)
  constexpr int MAXT = 6;  // 2 L <= 6
  f32x4v aq[MAXT][2];
  trf2_static_for<MAXT>([&](auto ti) {
    constexpr int t = decltype(ti)::value;
    const int tc = t < 2 * L ? t : (L > 0 ? 2 * L - 1 : 0);
    const float* src = (tc < L) ? a.h + tc * nH : a.u + (tc - L) * nH;
    if (L > 0) tile_load(aq[t], src);
  });
  constexpr int XQ = 8;  // x elements per thread: 32 D <= 8 * 512
  float xq[XQ];
#pragma unroll
  for (int i = 0; i < XQ; ++i) {
    if (TRF_THREADS * i >= TRF_ROWS * D) break;  // uniform
    int idx = tid + TRF_THREADS * i;
    idx = idx < TRF_ROWS * D ? idx : TRF_ROWS * D - 1;
    const int r = idx / D;
    const int row = r0 + r < n ? r0 + r : n - 1;
    xq[i] = a.X[(long long)row * D + (idx - r * D)];
  }
  float wc[TRF3_PP], wn[TRF3_PP];
  if (L > 0) trf3_load_wx(wn, P + a.off[4 + 4 * (L - 1)], H, wave, lane);
  else trf3_load_wx(wn, P + a.off[0], a.d_id, wave, lane);
  // ---- final layer, dW: dWf = gp^T . h_L - out-tiles wave, wave + 8, wave + 16, the four in-tiles of each ----
  {
    const int OTf = (np + 31) >> 5;
    float* gWf = GR + a.off[2 + 4 * L];
    float bv[4][16];
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
      for (int q = 0; q < 16; ++q) bv[it][q] = HL[(2 * q + half) * H + it * 32 + j];
    for (int ot = wave; ot < OTf; ot += TRF_WAVES) {
      const int oc = ot * 32 + j;
      const int ocl = oc < np ? oc : 0;
      float av[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) av[q] = PT[(2 * q + half) * pld + ocl];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        f32x16 acc = {0}, acc2 = {0};
#pragma unroll
        for (int q = 0; q < 16; q += 2) {  // D[i = in][jj = out]
          acc = nsa_mfma(bv[it][q], av[q], acc);
          acc2 = nsa_mfma(bv[it][q + 1], av[q + 1], acc2);
        }
        trf2_dw_store<SLAB>(gWf, np, H, oc, it * 32, half, acc, acc2);
      }
    }
    float* gbf = GR + a.off[3 + 4 * L];
    for (int o = tid; o < np; o += TRF_THREADS) {
      float s = 0.0f;
#pragma unroll 8
      for (int row = 0; row < TRF_ROWS; ++row) s += PT[row * pld + o];
      trf2_grad_out<SLAB>(&gbf[o], s);
    }
  }
  trf_barrier_lds();  // the partial sums of gh are complete, the parameter tile is dead
  {
    const int c = tid & 127, rb = 8 * (tid >> 7);
    const float* pc = part + (c >> 5) * 1024 + rb * 32 + (c & 31);
#pragma unroll
    for (int i = 0; i < 8; ++i) G0[(rb + i) * ld + c] = pc[i * 32] + pc[TRF3_HT * 1024 + i * 32];
  }
  trf2_static_for<MAXT>([&](auto ti) {
    constexpr int t = decltype(ti)::value;
    if (t < 2 * L) tile_store(aq[t], ACT + t * TRF_ROWS * H);
  });
#pragma unroll
  for (int i = 0; i < XQ; ++i) {
    if (TRF_THREADS * i >= TRF_ROWS * D) break;  // uniform
    const int idx = tid + TRF_THREADS * i;
    const int r = idx / D;
    if (idx < TRF_ROWS * D) XT[r * ldx + (idx - r * D)] = xq[i];
  }
  TRF_MARK(32);

  // ---- residual blocks, last to first ----
  for (int l = L - 1; l >= 0; --l) {
    const float* hl = ACT + l * TRF_ROWS * H;
    const float* ul = ACT + (L + l) * TRF_ROWS * H;
    // h_{l+1} = h_l + W2 act(u_l) + b2:  g_u = (G0 . W2) act'(u_l) -> G1
#pragma unroll
    for (int q = 0; q < TRF3_PP; ++q) wc[q] = wn[q];
    trf3_load_wx(wn, P + a.off[2 + 4 * l], H, wave, lane);
    trf_barrier_lds();
    trf3_back_linear<false, SLAB>(G0, wc, ul, H, nullptr, a.act, H, ul, a.act, nullptr, G1, GR + a.off[4 + 4 * l],
                                  GR + a.off[5 + 4 * l], ld, part, wave, lane, tid);
    // u_l = W1 act(h_l) + b1:  g_h_l = G0 + (G1 . W1) act'(h_l) -> G0
#pragma unroll
    for (int q = 0; q < TRF3_PP; ++q) wc[q] = wn[q];
    if (l > 0) trf3_load_wx(wn, P + a.off[4 + 4 * (l - 1)], H, wave, lane);
    else trf3_load_wx(wn, P + a.off[0], a.d_id, wave, lane);
    trf_barrier_lds();
    trf3_back_linear<false, SLAB>(G1, wc, hl, H, nullptr, a.act, H, hl, a.act, G0, G0, GR + a.off[2 + 4 * l],
                                  GR + a.off[3 + 4 * l], ld, part, wave, lane, tid);
    TRF_MARK(33 + (L - 1 - l));
  }

  // ---- initial layer: dW0 = G0^T . x_id, g_x[:, id_cols] += G0 . W0 ----
#pragma unroll
  for (int q = 0; q < TRF3_PP; ++q) wc[q] = wn[q];
  trf_barrier_lds();
  trf3_back_linear<true, SLAB>(G0, wc, XT, ldx, a.id_cols, ACT_NONE, a.d_id, nullptr, ACT_NONE, nullptr, G1,
                               GR + a.off[0], GR + a.off[1], ld, part, wave, lane, tid);
  trf_barrier_lds();
  for (int idx = tid; idx < TRF_ROWS * a.d_id; idx += TRF_THREADS) {
    const int r = idx / a.d_id, k = idx - r * a.d_id;
    if (r < rows_here) {
      float* g = a.Gx + (long long)(r0 + r) * D + a.id_cols[k];
      *g = *g + G1[r * ld + k];
    }
  }
  TRF_MARK(60);
}
