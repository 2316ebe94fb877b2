// Fused training kernels of one coupling layer (included by nsa_train.hip).
//
// The layer-wise path of nsa_train.hip spends a minibatch step in ~150 launches of
// 5-15 us kernels.  Here one workgroup (8 waves) owns 32 batch rows and runs a whole coupling
// layer - conditioner (residual net), transform, and their reverse mode - with the
// activations of its rows in LDS:
//
//   trf_forward   x_id -> h_0 -> {u_l, h_{l+1}} -> parameters -> spline/affine forward
//                 (saves h_l, u_l, parameters and the transformed inputs for the backward pass,
//                  exactly the workspace layout of the layer-wise path)
//   trf_backward  transform backward -> final layer -> residual blocks -> initial layer:
//                 dX chains stay in LDS, weight gradients are per-workgroup partial products
//                 added to the flat gradient with float atomics (the flat gradient is
//                 zeroed once per step)
//
// GEMMs are v_mfma_f32_32x32x2_f32 with exact float32 products (training keeps full
// precision).  Operand roles are chosen so that LDS reads are conflict-free (odd row
// stride) and global accesses run along rows:
//   forward  D[i = row][jj = out]   A = act(LDS[row][k])         B = W^T[k][out] (global, coalesced;
//                                                                    the transposed copy is rebuilt once per step)
//   dX       D[i = row][jj = in]    A = G LDS[row][out]          B = W[out][in]  (global, coalesced)
//   dW       D[i = out][jj = in]    A = G LDS[row][out] (k=row)  B = act(LDS[row][in])
// Reference semantics: glasflow.nflows ResidualNet (call sites src/nessai/flows/nsf.py:85-95,
// realnvp.py:136-146) under FlowModel._train (src/nessai/flowmodel/base.py:365-452).
#pragma once

#ifndef TRF_ROWS
#define TRF_ROWS 32
#endif
#define TRF_RT (TRF_ROWS / 32)  // MFMA row tiles per workgroup
#define TRF_THREADS 512
#define TRF_WAVES 8
#define TRF_MAX_L 8

struct TrfArgs {
  int n, D, H, L, d_id, d_tr, np, act, transform, ld;  // ld: odd LDS row stride >= max(H, d_id) + 1
  int ch;  // parameters per chunk of the last layer's backward pass (64 or 128)
  const int* id_cols;
  const int* tr_cols;
  const float* params;
  const float* wt;  // the weight matrices transposed ([in][out]) at the same flat offsets
  float* grads;
  float* X;       // [n][D] current values (in/out)
  float* Gx;      // [n][D] dL/dX (backward)
  float* xsave;   // [n][d_tr]
  float* h;       // [L + 1][n][H]
  float* u;       // [L][n][H]
  float* p;       // [n][np] transform parameters
  float* gp;      // [n][np] their gradient
  float* logdet;  // [n]
  const float* gld;  // [n] dL/dlogdet
  float* dbg;        // optional: 32 phase timestamps (cycles since kernel start) of workgroup 0
  float* gslab;      // second generation: [grid][npar] per-workgroup weight gradients (null: float atomics on grads)
  long long npar;
  TrConsts c;
  long long off[2 + 4 * TRF_MAX_L + 2];  // flat offsets of W0, b0, {W1, b1, W2, b2} x L, Wf, bf
};

#define TRF_MARK(i)                                                              \
  do {                                                                           \
    if (a.dbg != nullptr && blockIdx.x == 0 && threadIdx.x == 0)                 \
      a.dbg[i] = (float)(long long)(clock64() - t_start);                        \
  } while (0)

// activation of a register batch with the (wave-uniform) kind tested ONCE: a per-element
// tr_act() is if-converted into evaluating tanh AND the sigmoid for every element
template <int N>
__device__ __forceinline__ void trf_act_batch(float (&v)[N], int act) {
  if (act == ACT_NONE) return;
  if (act == NSA_ACT_RELU) {
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] = fmaxf(v[q], 0.0f);
  } else if (act == NSA_ACT_TANH) {
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] = tanhf(v[q]);
  } else {
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] = v[q] * (1.0f / (1.0f + expf(-v[q])));
  }
}

// barrier that publishes LDS tiles only: global stores / loads in flight (activation saves,
// weight preloads) are NOT waited for, unlike __syncthreads()
__device__ __forceinline__ void trf_barrier_lds() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// wt[tmap[i]] = params[i]: every weight matrix [out][in] lands transposed at its own offset
__global__ void tr_transpose_weights(long long n, const float* __restrict__ p, const int* __restrict__ tmap,
                                     float* __restrict__ wt) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i < n) wt[tmap[i]] = p[i];
}

// rows [r0, r0 + 64) x cols [c0, c0 + nc) of a row-major global matrix -> LDS tile[r][c], zero padded
__device__ __forceinline__ void trf_stage(float* __restrict__ tile, int ld, const float* __restrict__ src,
                                          int src_ld, int r0, int n, int c0, int nc, int nc_valid,
                                          const int* __restrict__ cols, int tid) {
  // flattened (row, col) index; 8 loads in flight per thread before the LDS stores (a 32 x 128 tile in ONE round trip,
  // a 384-parameter chunk of the final layer's gradient in three)
  const int total = TRF_ROWS * nc;
  for (int base = tid; base < total; base += TRF_THREADS * 8) {
    float v[8];
    int dst[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int idx = base + TRF_THREADS * q;
      const int r = idx / nc, c = idx - r * nc;
      const int row = r0 + r;
      dst[q] = (idx < total) ? r * ld + c : -1;
      v[q] = 0.0f;
      if (idx < total && row < n && c < nc_valid)
        v[q] = src[(unsigned)row * (unsigned)src_ld + (unsigned)(cols ? cols[c0 + c] : c0 + c)];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (dst[q] >= 0) tile[dst[q]] = v[q];
  }
}

// The weights of a layer do not depend on the previous layer's result: every wave loads the first
// 64 k of its first tile BEFORE the barrier that publishes its input tile (TrfPre), so the global
// latency overlaps the previous layer's tail instead of sitting on the critical path.
struct TrfPre {
  float v[16];
};

// first batch (k < 32) of W^T[k][o] for tile t = wave of trf_linear
__device__ __forceinline__ void trf_linear_preload(TrfPre& pre, const float* __restrict__ Wt, int Kd, int O,
                                                   int wave, int lane) {
  const int half = lane >> 5, j = lane & 31;
  const int OT = (O + 31) >> 5;
  if (wave >= TRF_RT * OT) return;
  const int o = (wave / TRF_RT) * 32 + j;
  const unsigned oc = (unsigned)(o < O ? o : O - 1);
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int k = 2 * q + half;
    pre.v[q] = Wt[(unsigned)(k < Kd ? k : Kd - 1) * (unsigned)O + oc];
  }
}

// w[q] = M[r0 + 2 q][c], q < 16, of a row-major matrix [rows][ldm] (c < cols valid): 16 B operands of an MFMA batch.
// `inside` (wave-uniform): the whole batch of the whole tile is inside the matrix - one 32-bit byte offset per load, a
// constant stride apart, nothing else.  Otherwise rows and column are clamped (the caller zeroes what it must not use).
// (With the clamps on every load, a batch was ~230 vector instructions around 16 MFMAs - and a single wave issues a
// vector instruction per ~5 cycles: as long as the MFMAs themselves; scripts/train_phase_timing.py --c4.)
__device__ __forceinline__ void trf_load16(float (&w)[16], const float* __restrict__ M, int r0, int rows, int ldm,
                                           int c, int cols, bool inside) {
  if (inside) {
    unsigned offb = 4u * ((unsigned)r0 * (unsigned)ldm + (unsigned)c);
    const unsigned strideb = 8u * (unsigned)ldm;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      w[q] = *(const float*)((const char*)M + offb);
      offb += strideb;
    }
  } else {
    const unsigned cc = (unsigned)(c < cols ? c : cols - 1);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int r = r0 + 2 * q;
      w[q] = M[(unsigned)(r < rows ? r : rows - 1) * (unsigned)ldm + cc];
    }
  }
}

// out[row][o] = sum_k act(Ain[row][k]) W[o][k] + b[o] (+ R[row][o]); rows of the workgroup.
// Wt = W^T [Kd][O].  out_lds / R may be null; out_lds == R is allowed (in-place residual update).
__device__ __forceinline__ void trf_linear(const float* __restrict__ Ain, int Kd, int act_in,
                                           const float* __restrict__ Wt, const float* __restrict__ b, int O,
                                           const float* R, float* out_lds, float* __restrict__ out_glb,
                                           int ld_glb, int ld, int r0, int n, int wave, int lane,
                                           const TrfPre& pre, float* fine = nullptr, long long t_start = 0) {
  // diagnostic stamps of one layer (scripts/train_phase_timing.py --fine): everything outstanding is waited for first
  int fine_i = 0;
  auto stamp = [&]() {
    if (fine != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      fine[fine_i++] = (float)(long long)(clock64() - t_start);
    }
  };
  stamp();
  const int half = lane >> 5, j = lane & 31;
  const int OT = (O + 31) >> 5;
  const int pairs = (Kd + 1) >> 1;
  // B operands of step (tile t, pairs [u0, u0 + 16)): lanes along the row of W^T - coalesced
  auto inside = [&](int t, int u0) { return 2 * (u0 + 16) <= Kd && (t / TRF_RT) * 32 + 32 <= O; };  // wave-uniform
  auto load_w = [&](float (&w)[16], int t, int u0) {
    trf_load16(w, Wt, 2 * u0 + half, Kd, O, (t / TRF_RT) * 32 + j, O, inside(t, u0));
  };
  // The operands of step s + 1 (a step = 32 k of a tile; the wave's tiles one after the other) are requested BEFORE
  // the MFMAs of step s, and a scheduling barrier keeps them there (left alone, the compiler sinks the loads below
  // the MFMAs and waits for them at the top of the next step): a layer used to be a chain of exposed round trips to
  // L2, one per step - 4 per tile of a 128-wide layer, 12 per wave in the final layer of config 4's flow.  (Two steps
  // ahead costs a third operand buffer and its addresses - 29 / 71 spilled registers - for copies that wait anyway.)
  const int n_tiles = TRF_RT * OT;
  int tn = wave, un = 0;  // the step the next request is for
  auto advance = [&]() {  // (stays on the last step once there: re-reads that are dropped)
    const bool more_k = un + 16 < pairs, more_t = tn + TRF_WAVES < n_tiles;
    un = more_k ? un + 16 : (more_t ? 0 : un);
    tn = more_k ? tn : (more_t ? tn + TRF_WAVES : tn);
  };
  float wv[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) wv[q] = pre.v[q];
  stamp();
  for (int t = wave; t < n_tiles; t += TRF_WAVES) {
    const int rt = t % TRF_RT, ot = t / TRF_RT;
    const int o = ot * 32 + j;
    const bool o_ok = o < O;
    f32x16 acc = {0}, acc2 = {0};
    const float* ax = Ain + (rt * 32 + j) * ld;
    for (int u0 = 0; u0 < pairs; u0 += 16) {
      float wn[16];
      advance();
      load_w(wn, tn, un);
      __builtin_amdgcn_sched_barrier(0);
      // all LDS reads of the batch are issued before the first MFMA (a dependent read per step
      // would serialise the chain on the LDS latency)
      float av[16];
      if (inside(t, u0)) {  // wave-uniform: nothing to clamp or to zero
        const float* ak = ax + 2 * u0 + half;
#pragma unroll
        for (int q = 0; q < 16; ++q) av[q] = ak[2 * q];
        trf_act_batch(av, act_in);
      } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int k = 2 * (u0 + q) + half;
          av[q] = ax[k < Kd ? k : 0];
        }
        trf_act_batch(av, act_in);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int k = 2 * (u0 + q) + half;
          const bool ok = k < Kd;
          av[q] = ok ? av[q] : 0.0f;
          wv[q] = (ok && o_ok) ? wv[q] : 0.0f;
        }
      }
      // two independent accumulation chains, no branches (k beyond Kd carries zero operands):
      // guarded, single-chain MFMAs issued at 100-145 cycles each instead of 64
#pragma unroll
      for (int q = 0; q < 16; q += 2) {
        acc = nsa_mfma(av[q], wv[q], acc);
        acc2 = nsa_mfma(av[q + 1], wv[q + 1], acc2);
      }
      if (fine != nullptr) {
        __builtin_amdgcn_sched_barrier(0);
        if (blockIdx.x == 0 && threadIdx.x == 0) fine[fine_i++] = (float)(long long)(clock64() - t_start);  // MFMAs issued
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) wv[q] = wn[q];
      stamp();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
    if (o_ok) {
      const float bias = b[o];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rl = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = acc[r] + bias;
        if (R) v += R[rl * ld + o];
        if (out_lds) out_lds[rl * ld + o] = v;
        if (r0 + rl < n) out_glb[(unsigned)(r0 + rl) * (unsigned)ld_glb + (unsigned)o] = v;
      }
    }
    stamp();
  }
}

// out[row][k] (op)= (sum_o G[row][g-col o] W[o][k]) * act'(Z[row][k]);  add != 0: out += value
// G, Z, out: LDS tiles of the workgroup's rows; W global [O][Kd].  Run by waves [w0, w0 + nw).
__device__ __forceinline__ void trf_dx(const float* __restrict__ G, int g_ld, int O, const float* __restrict__ W,
                                       int Kd, const float* Z, int act, float* out, int add, int ld, int wave,
                                       int nw, int lane) {
  const int half = lane >> 5, j = lane & 31;
  const int KT = (Kd + 31) >> 5;
  const int pairs = (O + 1) >> 1;
  // B operands of step (tile t, pairs [u0, u0 + 16)): lanes along the row of W - coalesced
  auto inside = [&](int t, int u0) { return 2 * (u0 + 16) <= O && (t / TRF_RT) * 32 + 32 <= Kd; };  // wave-uniform
  auto load_w = [&](float (&w)[16], int t, int u0) {
    trf_load16(w, W, 2 * u0 + half, O, Kd, (t / TRF_RT) * 32 + j, Kd, inside(t, u0));
  };
  const int n_tiles = TRF_RT * KT;
  int tn = wave, un = 0;  // the step the next request is for (trf_linear has the reasoning)
  auto advance = [&]() {
    const bool more_o = un + 16 < pairs, more_t = tn + nw < n_tiles;
    un = more_o ? un + 16 : (more_t ? 0 : un);
    tn = more_o ? tn : (more_t ? tn + nw : tn);
  };
  float wv[16];
  if (wave < n_tiles) load_w(wv, tn, un);
  for (int t = wave; t < n_tiles; t += nw) {
    const int rt = t % TRF_RT, kt = t / TRF_RT;
    const int kcol = kt * 32 + j;
    const bool col_ok = kcol < Kd;
    f32x16 acc = {0}, acc2 = {0};
    const float* ag = G + (rt * 32 + j) * g_ld;
    for (int u0 = 0; u0 < pairs; u0 += 16) {
      float wn[16], av[16];
      advance();
      load_w(wn, tn, un);
      __builtin_amdgcn_sched_barrier(0);
      if (inside(t, u0)) {  // wave-uniform: nothing to clamp or to zero
        const float* ao = ag + 2 * u0 + half;
#pragma unroll
        for (int q = 0; q < 16; ++q) av[q] = ao[2 * q];  // LDS reads issued together, not one per MFMA
      } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int o = 2 * (u0 + q) + half;
          av[q] = ag[o < O ? o : 0];
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int o = 2 * (u0 + q) + half;
          const bool ok = o < O;
          av[q] = ok ? av[q] : 0.0f;
          wv[q] = (ok && col_ok) ? wv[q] : 0.0f;
        }
      }
#pragma unroll
      for (int q = 0; q < 16; q += 2) {  // two independent chains; o beyond O carries zero operands
        acc = nsa_mfma(av[q], wv[q], acc);
        acc2 = nsa_mfma(av[q + 1], wv[q + 1], acc2);
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) wv[q] = wn[q];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
    if (col_ok) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rl = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = acc[r];
        if (Z) {
          const float z = Z[rl * ld + kcol];
          if (act == NSA_ACT_RELU) v = z > 0.0f ? v : 0.0f;  // wave-uniform kind test first
          else v *= tr_act_grad(z, act);
        }
        if (add) v += out[rl * ld + kcol];
        out[rl * ld + kcol] = v;
      }
    }
  }
}

// gW[o][k] += sum_rows G[row][o] act(Ain[row][k]);  gb[o] += sum_rows G[row][o]
// (64-row partial sums, float atomics).  o < O, k < Kd.  Run by waves [w0, w0 + nw).
__device__ __forceinline__ void trf_dw(const float* __restrict__ G, int g_ld, int O, const float* __restrict__ Ain,
                                       int a_ld, int Kd, int act_in, float* __restrict__ gW,
                                       float* __restrict__ gb, int wave, int nw, int lane) {
  const int half = lane >> 5, j = lane & 31;
  const int OT = (O + 31) >> 5, KT = (Kd + 31) >> 5;
  for (int t = wave; t < OT * KT; t += nw) {
    const int ot = t / KT, kt = t - ot * KT;
    const int oc = ot * 32 + j, kc = kt * 32 + j;
    const bool o_ok = oc < O, k_ok = kc < Kd;
    f32x16 acc = {0}, acc2 = {0};
    const int ocl = o_ok ? oc : 0, kcl = k_ok ? kc : 0;
    const bool full = ot * 32 + 32 <= O && kt * 32 + 32 <= Kd;  // wave-uniform: no operand to zero
    for (int ub = 0; ub < TRF_ROWS / 2; ub += 16) {  // 32 LDS reads issued before each 16-MFMA chain
      float av[16], bv[16];
      {  // rows 2 (ub + q) + half: one address each, a constant stride apart
        const float* gp_ = G + (2 * ub + half) * g_ld + ocl;
        const float* ap_ = Ain + (2 * ub + half) * a_ld + kcl;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          av[q] = *gp_;
          bv[q] = *ap_;
          gp_ += 2 * g_ld;
          ap_ += 2 * a_ld;
        }
      }
      trf_act_batch(bv, act_in);
      if (!full) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          av[q] = o_ok ? av[q] : 0.0f;
          bv[q] = k_ok ? bv[q] : 0.0f;
        }
      }
#pragma unroll
      for (int q = 0; q < 16; q += 2) {  // D[i = o][jj = k], two independent chains
        acc = nsa_mfma(av[q], bv[q], acc);
        acc2 = nsa_mfma(av[q + 1], bv[q + 1], acc2);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
    if (k_ok) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (o < O) atomicAdd(&gW[(unsigned)o * (unsigned)Kd + (unsigned)kc], acc[r]);
      }
    }
  }
  for (int o = wave * 64 + lane; o < O; o += nw * 64) {
    float s = 0.0f;
    for (int row = 0; row < TRF_ROWS; ++row) s += G[row * g_ld + o];
    atomicAdd(&gb[o], s);
  }
}

template <int K>  // spline bins; 0: affine / additive (a.transform)
__global__ void __launch_bounds__(TRF_THREADS) trf_forward(TrfArgs a) {
  extern __shared__ float trf_sm[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r0 = blockIdx.x * TRF_ROWS;
  const int ld = a.ld, n = a.n, H = a.H;
  float* A0 = trf_sm;
  float* A1 = A0 + TRF_ROWS * ld;
  const float* P = a.params;
  const float* WT = a.wt;
  const long long nH = (long long)n * H;

  const long long t_start = clock64();
  TrfPre pre;
  trf_linear_preload(pre, WT + a.off[0], a.d_id, H, wave, lane);
  trf_stage(A0, ld, a.X, a.D, r0, n, 0, a.d_id, a.d_id, a.id_cols, tid);
  trf_barrier_lds();
  TRF_MARK(0);
  // h_0 = W0 x_id + b0
  trf_linear(A0, a.d_id, ACT_NONE, WT + a.off[0], P + a.off[1], H, nullptr, A1, a.h, H, ld, r0, n, wave, lane, pre);
  TRF_MARK(1);
  for (int l = 0; l < a.L; ++l) {
    // u_l = W1 act(h_l) + b1;  h_{l+1} = h_l + W2 act(u_l) + b2
    trf_linear_preload(pre, WT + a.off[2 + 4 * l], H, H, wave, lane);
    trf_barrier_lds();
    trf_linear(A1, H, a.act, WT + a.off[2 + 4 * l], P + a.off[3 + 4 * l], H, nullptr, A0, a.u + l * nH, H, ld, r0,
               n, wave, lane, pre);
    TRF_MARK(2 + 2 * l);
    trf_linear_preload(pre, WT + a.off[4 + 4 * l], H, H, wave, lane);
    trf_barrier_lds();
    trf_linear(A0, H, a.act, WT + a.off[4 + 4 * l], P + a.off[5 + 4 * l], H, A1, A1, a.h + (l + 1) * nH, H, ld,
               r0, n, wave, lane, pre, (a.dbg != nullptr && l == 0) ? a.dbg + 44 : nullptr, t_start);
    TRF_MARK(3 + 2 * l);
  }
  // parameters = Wf h_L + bf (global only)
  trf_linear_preload(pre, WT + a.off[2 + 4 * a.L], H, a.np, wave, lane);
  trf_barrier_lds();
  trf_linear(A1, H, ACT_NONE, WT + a.off[2 + 4 * a.L], P + a.off[3 + 4 * a.L], a.np, nullptr, nullptr, a.p, a.np,
             ld, r0, n, wave, lane, pre);
  TRF_MARK(20);
  __syncthreads();  // the parameters of this workgroup's rows are visible to all its threads
  TRF_MARK(21);
  const int items = TRF_ROWS * a.d_tr;
  for (int e = tid; e < items; e += TRF_THREADS) {
    const int r = e / a.d_tr, m = e - r * a.d_tr;
    const int row = r0 + r;
    if (row >= n) continue;
    if constexpr (K > 0) {
      tr_rq_forward_item<K>(row, m, a.d_tr, a.tr_cols, a.p + (long long)row * a.np + m * (3 * K - 1), a.X, a.D, a.xsave,
                            a.logdet, a.c);
    } else {
      tr_affine_forward_item(row, m, a.d_tr, a.transform == NSA_T_ADDITIVE ? 1 : 0, a.tr_cols,
                             a.p + (long long)row * a.np, a.X, a.D, a.xsave, a.logdet, a.c.aff_a, a.c.aff_b);
    }
  }
  TRF_MARK(22);
}

template <int K>
__global__ void __launch_bounds__(TRF_THREADS) trf_backward(TrfArgs a) {
  extern __shared__ float trf_sm[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, j = lane & 31;
  const int r0 = blockIdx.x * TRF_ROWS;
  const int ld = a.ld, n = a.n, H = a.H, np = a.np;
  float* A0 = trf_sm;
  float* A1 = A0 + TRF_ROWS * ld;
  float* G0 = A1 + TRF_ROWS * ld;
  float* G1 = G0 + TRF_ROWS * ld;
  float* gc = G1 + TRF_ROWS * ld;  // [64][ch + 1] chunk of the parameter gradient
  const float* P = a.params;
  float* GR = a.grads;
  const long long nH = (long long)n * H;
  // the waves split into a dX group (0-3) and a dW group (4-7): the two products of every layer
  // read the same tiles and write disjoint ones
  const bool dx_group = wave < 4;
  const int gw = wave & 3;
  const long long t_start = clock64();

  // transform backward: dL/dparameters -> a.gp, dL/dx_tr -> a.Gx, x restored in a.X
  const int items = TRF_ROWS * a.d_tr;
  for (int e = tid; e < items; e += TRF_THREADS) {
    const int r = e / a.d_tr, m = e - r * a.d_tr;
    const int row = r0 + r;
    if (row >= n) continue;
    if constexpr (K > 0) {
      const long long it = (long long)row * a.np + m * (3 * K - 1);
      tr_rq_backward_item<K>(row, m, a.d_tr, a.tr_cols, a.p + it, a.xsave, a.X, a.D, a.Gx, a.gld, a.gp + it, a.c);
    } else {
      tr_affine_backward_item(row, m, a.d_tr, a.transform == NSA_T_ADDITIVE ? 1 : 0, a.tr_cols,
                              a.p + (long long)row * a.np, a.xsave, a.X, a.D, a.Gx, a.gld, a.gp + (long long)row * a.np,
                              a.c.aff_a, a.c.aff_b);
    }
  }
  TRF_MARK(31);
  trf_stage(A1, ld, a.h + a.L * nH, H, r0, n, 0, H, H, nullptr, tid);  // h_L
  __syncthreads();
  TRF_MARK(30);

  // final layer, a.ch parameters at a time: dWf, dbf (waves 4-7) and gh = gp . Wf (waves 0-3)
  const int KT = (H + 31) >> 5;  // <= 4: 2 KT <= 8 tiles of gh, two per wave of the dX group
  f32x16 acc_dx[2];
  acc_dx[0] = f32x16{0};
  acc_dx[1] = f32x16{0};
  const float* Wf = P + a.off[2 + 4 * a.L];
  const int CH = a.ch, CLD = a.ch + 1;
  // B operands of step (chunk c0, tile t, pairs [u0, u0 + 16)) of gh = gp . Wf: lanes along the row of Wf
  auto inside_f = [&](int c0, int t, int u0) {  // wave-uniform
    return c0 + 2 * (u0 + 16) <= np && 2 * (u0 + 16) <= a.ch && (t / TRF_RT) * 32 + 32 <= H;
  };
  auto load_wf = [&](float (&w)[16], int c0, int t, int u0) {
    trf_load16(w, Wf + (long long)c0 * H, 2 * u0 + half, np - c0, H, (t / TRF_RT) * 32 + j, H, inside_f(c0, t, u0));
  };
  // the dX group's steps in the order they run, each requesting the operands of the one behind it (see trf_linear):
  // a chunk used to be 8 exposed round trips to L2 per wave, 48 per launch with config 4's 736 parameters
  const bool two_tiles = gw + 4 < TRF_RT * KT;
  float wv[16];
  if (dx_group && gw < TRF_RT * KT) load_wf(wv, 0, gw, 0);
  for (int c0 = 0; c0 < np; c0 += CH) {
    const int nc = (np - c0 < CH) ? (np - c0) : CH;
    if (c0 > 0) __syncthreads();
    trf_stage(gc, CLD, a.gp, np, r0, n, c0, CH, nc, nullptr, tid);
    __syncthreads();
    if (dx_group) {
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) {
        const int t = gw + 4 * s_;
        if (t < TRF_RT * KT) {
          const int rt = t % TRF_RT, kt = t / TRF_RT;
          const int kcol = kt * 32 + j;
          const bool col_ok = kcol < H;
          const float* ag = gc + (rt * 32 + j) * CLD;
          for (int u0 = 0; u0 < CH / 2; u0 += 16) {
            if (2 * u0 < nc) {  // wave-uniform
              float wn[16];
              {
                const bool more_u = 2 * (u0 + 16) < nc && u0 + 16 < CH / 2;
                const bool more_t = s_ == 0 && two_tiles;
                const bool more_c = c0 + CH < np;
                // next: same tile; else the other tile of this chunk; else tile gw of the next chunk; else a re-read
                const int c2 = (more_u || more_t) ? c0 : (more_c ? c0 + CH : c0);
                const int t2 = more_u ? t : (more_t ? gw + 4 : gw);
                load_wf(wn, c2, t2, more_u ? u0 + 16 : 0);
              }
              __builtin_amdgcn_sched_barrier(0);
              float av[16];
              if (inside_f(c0, t, u0)) {  // wave-uniform: nothing to clamp or to zero
                const float* ao = ag + 2 * u0 + half;
#pragma unroll
                for (int q = 0; q < 16; ++q) av[q] = ao[2 * q];
              } else {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                  const int oc = 2 * (u0 + q) + half;
                  av[q] = ag[oc < nc ? oc : 0];
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                  const int oc = 2 * (u0 + q) + half;
                  const bool ok = oc < nc;
                  av[q] = ok ? av[q] : 0.0f;
                  wv[q] = (ok && col_ok) ? wv[q] : 0.0f;
                }
              }
#pragma unroll
              for (int q = 0; q < 16; ++q) acc_dx[s_] = nsa_mfma(av[q], wv[q], acc_dx[s_]);
#pragma unroll
              for (int q = 0; q < 16; ++q) wv[q] = wn[q];
            }
          }
        }
      }
    } else {
      trf_dw(gc, CLD, nc, A1, ld, H, ACT_NONE, GR + a.off[2 + 4 * a.L] + (long long)c0 * H,
             GR + a.off[3 + 4 * a.L] + c0, gw, 4, lane);
    }
  }
  if (dx_group) {
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
      const int t = gw + 4 * s_;
      if (t < TRF_RT * KT) {
        const int rt = t % TRF_RT, kt = t / TRF_RT;
        const int kcol = kt * 32 + j;
        if (kcol < H) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            G0[(rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * ld + kcol] = acc_dx[s_][r];
        }
      }
    }
  }
  __syncthreads();
  TRF_MARK(32);

  for (int l = a.L - 1; l >= 0; --l) {
    trf_stage(A0, ld, a.u + l * nH, H, r0, n, 0, H, H, nullptr, tid);  // u_l
    trf_stage(A1, ld, a.h + l * nH, H, r0, n, 0, H, H, nullptr, tid);  // h_l
    trf_barrier_lds();
    // h_{l+1} = h_l + W2 act(u_l) + b2
    if (dx_group) trf_dx(G0, ld, H, P + a.off[4 + 4 * l], H, A0, a.act, G1, 0, ld, gw, 4, lane);
    else trf_dw(G0, ld, H, A0, ld, H, a.act, GR + a.off[4 + 4 * l], GR + a.off[5 + 4 * l], gw, 4, lane);
    trf_barrier_lds();
    // u_l = W1 act(h_l) + b1
    if (dx_group) trf_dx(G1, ld, H, P + a.off[2 + 4 * l], H, A1, a.act, G0, 1, ld, gw, 4, lane);
    else trf_dw(G1, ld, H, A1, ld, H, a.act, GR + a.off[2 + 4 * l], GR + a.off[3 + 4 * l], gw, 4, lane);
    trf_barrier_lds();
    TRF_MARK(33 + (a.L - 1 - l));
  }

  // initial layer: dW0, db0 and g_x[:, id_cols] += gh . W0 (x was restored above)
  trf_stage(A0, ld, a.X, a.D, r0, n, 0, a.d_id, a.d_id, a.id_cols, tid);
  trf_barrier_lds();
  if (dx_group) trf_dx(G0, ld, H, P + a.off[0], a.d_id, nullptr, ACT_NONE, G1, 0, ld, gw, 4, lane);
  else trf_dw(G0, ld, H, A0, ld, a.d_id, ACT_NONE, GR + a.off[0], GR + a.off[1], gw, 4, lane);
  trf_barrier_lds();
  for (int idx = tid; idx < TRF_ROWS * a.d_id; idx += TRF_THREADS) {
    const int r = idx / a.d_id, k = idx - r * a.d_id;
    const int row = r0 + r;
    if (row < n) a.Gx[(unsigned)row * (unsigned)a.D + (unsigned)a.id_cols[k]] += G1[r * ld + k];
  }
  TRF_MARK(60);
}
