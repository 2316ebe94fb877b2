// Training kernels: forward + backward of the coupling flow, loss, gradient-norm
// clipping and AdamW, all enqueued by ONE C entry point per minibatch.
//
// Reference semantics: FlowModel._train (src/nessai/flowmodel/base.py:365-452):
//   loss = -model.log_prob(x).mean(); loss.backward();
//   clip_grad_norm_(parameters, 5.0); AdamW.step()
// with log_prob = StandardNormal.log_prob(T(x)) + log|det J| (flows/base.py:236-245).
//
// Batches are tiny (<= ~1000 rows x <= 64 features), so the work is
// launch-latency-bound rather than throughput-bound: the step is a chain of
// small kernels on one stream (no host round trip inside a step), each GEMM a
// 32x32-tile v_mfma_f32_32x32x2_f32 kernel reading its operands straight from
// L2.  Activations of the forward pass are kept in a workspace and reused by the
// backward pass; the input of every coupling layer is reconstructed by undoing
// the layer in place (only the transformed columns are saved).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/nessai_amd.h"
#include "nsa_device.h"

// ---------------------------------------------------------------------------
// small GEMMs on MFMA (one wave = one 32x32 output tile)
// ---------------------------------------------------------------------------

enum { ACT_NONE = -1 };

__device__ __forceinline__ float tr_act(float x, int act) {
  if (act == ACT_NONE) return x;
  if (act == NSA_ACT_RELU) return fmaxf(x, 0.0f);
  if (act == NSA_ACT_TANH) return tanhf(x);
  return x * (1.0f / (1.0f + expf(-x)));
}
// derivative of the activation at pre-activation x
__device__ __forceinline__ float tr_act_grad(float x, int act) {
  if (act == ACT_NONE) return 1.0f;
  if (act == NSA_ACT_RELU) return x > 0.0f ? 1.0f : 0.0f;
  if (act == NSA_ACT_TANH) { const float t = tanhf(x); return 1.0f - t * t; }
  const float s = 1.0f / (1.0f + expf(-x));
  return s * (1.0f + x * (1.0f - s));
}

// Y[n][O] = act_in(X[:, cols])[n][I] . W[O][I]^T + b (+ R)
// M (optional, [n][I]): a multiplier of the activated input - the dropout mask of this product's input
// (0 or 1 / (1 - p): tr_dropout_mask), training mode only
__global__ void __launch_bounds__(64) tr_gemm_nt(int n, int I, int O, const float* __restrict__ X, int ldx,
                                                 const int* __restrict__ cols, int act_in,
                                                 const float* __restrict__ W, const float* __restrict__ b,
                                                 const float* __restrict__ R, float* __restrict__ Y, int ldy,
                                                 const float* __restrict__ M = nullptr) {
  const int lane = threadIdx.x, half = lane >> 5, j = lane & 31;
  const int row0 = blockIdx.x * 32, col0 = blockIdx.y * 32;
  const int row = row0 + j, col = col0 + j;
  f32x16 acc = {0};
  const long long xrow = (long long)(row < n ? row : n - 1) * ldx;
  const long long wrow = (long long)(col < O ? col : O - 1) * I;
  const bool row_ok = row < n, col_ok = col < O;
  for (int k0 = 0; k0 < I; k0 += 16) {  // 8 MFMA steps per trip: all 16 loads issued before use
    float xv[8], wv[8], mv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int kk = k0 + 2 * u + half;
      const int kc = kk < I ? kk : I - 1;  // clamped: loads are unconditional
      xv[u] = X[xrow + (cols ? cols[kc] : kc)];
      wv[u] = W[wrow + kc];
      mv[u] = M ? M[(long long)(row < n ? row : n - 1) * I + kc] : 1.0f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int kk = k0 + 2 * u + half;
      const float a = (row_ok && kk < I) ? tr_act(xv[u], act_in) * mv[u] : 0.0f;
      const float bb = (col_ok && kk < I) ? wv[u] : 0.0f;
      // D[i][jj] = sum_k A[i][k] B[k][jj]: i = output column (W row), jj = batch row
      acc = nsa_mfma(bb, a, acc);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int o = col0 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (o < O && row < n) {
      float v = acc[r] + b[o];
      if (R) v += R[(long long)row * ldy + o];
      Y[(long long)row * ldy + o] = v;
    }
  }
}

// Gin[n][I] (op)= (Gout[n][O] . W[O][I]) * act'(Z[n][I])
//   mode 0: Gin = value; mode 1: Gin += value; mode 2: Gx[:, cols[i]] += value (scatter-add);
//   mode 3: Gx[:, cols[i]] = value (scatter-assign)
__global__ void __launch_bounds__(64) tr_gemm_nn(int n, int O, int I, const float* __restrict__ Gout, int ldg,
                                                 const float* __restrict__ W, const float* __restrict__ Z,
                                                 int ldz, int act, float* __restrict__ Gin, int ldi,
                                                 const int* __restrict__ cols, int mode,
                                                 const float* __restrict__ M = nullptr) {
  const int lane = threadIdx.x, half = lane >> 5, j = lane & 31;
  const int row0 = blockIdx.x * 32, col0 = blockIdx.y * 32;
  const int row = row0 + j, col = col0 + j;
  f32x16 acc = {0};
  const long long grow = (long long)(row < n ? row : n - 1) * ldg;
  const int wc = col < I ? col : I - 1;
  const bool row_ok = row < n, col_ok = col < I;
  for (int k0 = 0; k0 < O; k0 += 16) {
    float gv[8], wv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int kk = k0 + 2 * u + half;
      const int kc = kk < O ? kk : O - 1;
      gv[u] = Gout[grow + kc];
      wv[u] = W[(long long)kc * I + wc];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int kk = k0 + 2 * u + half;
      const float a = (row_ok && kk < O) ? gv[u] : 0.0f;
      const float bb = (col_ok && kk < O) ? wv[u] : 0.0f;
      acc = nsa_mfma(bb, a, acc);  // D[i = input feature][jj = batch row]
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = col0 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (i < I && row < n) {
      float v = acc[r];
      if (M) v *= M[(long long)row * I + i];  // (the dropout mask sits between the activation and the product)
      if (Z) v *= tr_act_grad(Z[(long long)row * ldz + i], act);
      if (mode == 0) Gin[(long long)row * ldi + i] = v;
      else if (mode == 1) Gin[(long long)row * ldi + i] += v;
      else if (mode == 2) Gin[(long long)row * ldi + cols[i]] += v;
      else Gin[(long long)row * ldi + cols[i]] = v;
    }
  }
}

// dW[O][I] = sum_rows Gout[row][o] * act_in(X[row][cols[i]]);  db[o] = sum_rows Gout[row][o]
// One workgroup = one 32x32 tile of dW; its 4 waves split the batch (the K dimension of
// this product) and reduce through LDS in a fixed order (deterministic).
__global__ void __launch_bounds__(256) tr_gemm_tn(int n, int O, int I, const float* __restrict__ Gout, int ldg,
                                                  const float* __restrict__ X, int ldx,
                                                  const int* __restrict__ cols, int act_in,
                                                  float* __restrict__ dW, float* __restrict__ db,
                                                  const float* __restrict__ M = nullptr) {
  __shared__ float red[3][64][17];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5, j = lane & 31;
  const int o0 = blockIdx.x * 32, i0 = blockIdx.y * 32;
  const int o = o0 + j, i = i0 + j;
  const int oc = o < O ? o : O - 1;
  const int xcol = (i < I) ? (cols ? cols[i] : i) : (cols ? cols[0] : 0);
  const bool o_ok = o < O, i_ok = i < I;
  const int per = (((n + 3) / 4) + 1) & ~1;  // rows per wave, even
  const int k_lo = wave * per, k_hi = (k_lo + per < n) ? (k_lo + per) : n;
  f32x16 acc = {0};
  float bsum = 0.0f;
  // 64 rows of operands are requested before the first of their MFMAs (this product is a chain of round trips to
  // the batch: 16 rows per trip made it 16 us for a 1000-row batch), from clamped addresses and with the range
  // tests as factors - a select on a loaded value makes the compiler wait for the load where it is issued
  for (int k0 = k_lo; k0 < k_hi; k0 += 64) {
    float gv[32], xv[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int row = k0 + 2 * u + half;
      const int rc = row < n ? row : n - 1;
      gv[u] = Gout[(long long)rc * ldg + oc];
      xv[u] = X[(long long)rc * ldx + xcol];
    }
    if (M) {  // (dropout: the activated input of the product was masked in the forward pass)
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        const int row = k0 + 2 * u + half;
        const int rc = row < n ? row : n - 1;
        xv[u] = tr_act(xv[u], act_in) * M[(long long)rc * I + (i < I ? i : I - 1)];
      }
    }
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int row = k0 + 2 * u + half;
      const float a = gv[u] * ((o_ok && row < k_hi) ? 1.0f : 0.0f);
      const float bb = (M ? xv[u] : tr_act(xv[u], act_in)) * ((i_ok && row < k_hi) ? 1.0f : 0.0f);
      bsum += a;
      acc = nsa_mfma(a, bb, acc);  // D[ii = o][jj = i]
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave - 1][lane][r] = acc[r];
    red[wave - 1][lane][16] = bsum;
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int w = 0; w < 3; ++w) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += red[w][lane][r];
      bsum += red[w][lane][16];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int oo = o0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (oo < O && i_ok) dW[(long long)oo * I + i] = acc[r];
    }
    if (blockIdx.y == 0) {
      bsum += __shfl_xor(bsum, 32);
      if (half == 0 && o_ok) db[o] = bsum;
    }
  }
}

// ---------------------------------------------------------------------------
// element-wise coupling transforms, forward (training) and backward
// ---------------------------------------------------------------------------

struct TrConsts {
  float tail_bound, min_bw, min_bh, min_d, edge_d, inv_div;
  float aff_a, aff_b;  // affine coupling: scale = sigmoid(raw + aff_a) + aff_b (nflows default 2, 1e-3)
};

// the spline of one (row, transformed feature) item: y and log|dy/dx| from x and the item's 3K - 1 parameters
// (global row-major, or an LDS tile)
template <int K>
__device__ __forceinline__ void tr_rq_forward_core(const float* __restrict__ Pit, float x, const TrConsts& c, float& y,
                                                   float& ld) {
  constexpr int NP = 3 * K - 1;
  float v[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) v[i] = Pit[i];
  NsaTransformConsts tc;
  tc.tail_bound = c.tail_bound; tc.two_b = 2.0f * c.tail_bound;
  tc.min_bw = c.min_bw; tc.bw_scale = (float)(1.0 - (double)c.min_bw * K);
  tc.min_bh = c.min_bh; tc.bh_scale = (float)(1.0 - (double)c.min_bh * K);
  tc.min_d = c.min_d; tc.edge_d = c.edge_d; tc.inv_logit_div = c.inv_div;
  tc.edge_raw = logf(expf(1.0f - c.min_d) - 1.0f);
  tc.k_real = K;
  typename NsaRQ<K>::State st;
  y = 0.0f;
  ld = 0.0f;
  bool bad = false;
  NsaRQ<K>::template phase<0, 0, false, NP>(st, v, x, tc, y, ld, bad);
}

// one thread per (row, transformed feature).  Forward: saves x, writes y in place,
// atomically accumulates log|det J| per row.
template <int K>
__device__ __forceinline__ void tr_rq_forward_item(int row, int m, int d_tr, const int* __restrict__ tr_cols,
                                                   const float* __restrict__ Pit, float* __restrict__ X, int ldx,
                                                   float* __restrict__ xsave, float* __restrict__ logdet,
                                                   const TrConsts& c) {
  const int col = tr_cols[m];
  const float x = X[(long long)row * ldx + col];
  xsave[row * d_tr + m] = x;
  float y, ld;
  tr_rq_forward_core<K>(Pit, x, c, y, ld);
  X[(long long)row * ldx + col] = y;
  atomicAdd(&logdet[row], ld);
}

template <int K>
__global__ void tr_rq_forward(int n, int d_tr, const int* __restrict__ tr_cols, const float* __restrict__ P,
                              float* __restrict__ X, int ldx, float* __restrict__ xsave,
                              float* __restrict__ logdet, TrConsts c) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * d_tr) return;
  const int row = t / d_tr;
  tr_rq_forward_item<K>(row, t - row * d_tr, d_tr, tr_cols, P + (long long)t * (3 * K - 1), X, ldx, xsave, logdet, c);
}

// Backward of the spline (hand-derived reverse mode of NsaRQ::phase, forward
// direction).  Inputs: saved x, parameters, upstream gy = dL/dy (read from
// Gx[row][col]) and gld = dL/dlogdet (same for every row).  Outputs: dL/dx
// (overwrites Gx[row][col]), dL/dparams (GP), and X[row][col] restored to x.
// the item's arithmetic: dL/dx (returned) and dL/dparameters (GPit, which may be Pit itself: every
// parameter is read before the first gradient is written) from x, gy = dL/dy and gld = dL/dlogdet
template <int K>
__device__ __forceinline__ float tr_rq_backward_core(const float* Pit, float x, float gy, float gld,
                                                     const TrConsts& c, float* GPit) {
  constexpr int NP = 3 * K - 1;
  const float B = c.tail_bound;
  float gp[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) gp[i] = 0.0f;
  float gx = gy;
  if ((x >= -B) && (x <= B)) {
    float v[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) v[i] = Pit[i];
    const float two_b = 2.0f * B;
    const float sc_w = (float)(1.0 - (double)c.min_bw * K), sc_h = (float)(1.0 - (double)c.min_bh * K);
    // forward recompute: softmax probabilities and knots
    float pw[K], ph[K], cw[K + 1], ch[K + 1], dv[K + 1], sg[K + 1];
    {
      float mx = -INFINITY, s = 0.0f;
#pragma unroll
      for (int i = 0; i < K; ++i) { pw[i] = v[i] * c.inv_div; mx = fmaxf(mx, pw[i]); }
#pragma unroll
      for (int i = 0; i < K; ++i) { pw[i] = expf(pw[i] - mx); s += pw[i]; }
      float run = 0.0f;
      cw[0] = -B;
#pragma unroll
      for (int i = 0; i < K; ++i) { pw[i] /= s; run += c.min_bw + sc_w * pw[i]; cw[i + 1] = two_b * run - B; }
      cw[K] = B;
    }
    {
      float mx = -INFINITY, s = 0.0f;
#pragma unroll
      for (int i = 0; i < K; ++i) { ph[i] = v[K + i] * c.inv_div; mx = fmaxf(mx, ph[i]); }
#pragma unroll
      for (int i = 0; i < K; ++i) { ph[i] = expf(ph[i] - mx); s += ph[i]; }
      float run = 0.0f;
      ch[0] = -B;
#pragma unroll
      for (int i = 0; i < K; ++i) { ph[i] /= s; run += c.min_bh + sc_h * ph[i]; ch[i + 1] = two_b * run - B; }
      ch[K] = B;
    }
    dv[0] = c.edge_d; dv[K] = c.edge_d; sg[0] = 0.0f; sg[K] = 0.0f;
#pragma unroll
    for (int i = 1; i < K; ++i) {
      const float u = v[2 * K + i - 1];
      dv[i] = c.min_d + nsa_softplus(u);
      sg[i] = 1.0f / (1.0f + expf(-u));  // d softplus / du
    }
    int k = 0;
#pragma unroll
    for (int i = 1; i < K; ++i) k += (x >= cw[i]) ? 1 : 0;
    float c0 = cw[0], c1 = cw[1], e0 = ch[0], e1 = ch[1], d0 = dv[0], d1 = dv[1];
#pragma unroll
    for (int i = 1; i < K; ++i) {
      const bool sel = (k == i);
      c0 = sel ? cw[i] : c0; c1 = sel ? cw[i + 1] : c1;
      e0 = sel ? ch[i] : e0; e1 = sel ? ch[i + 1] : e1;
      d0 = sel ? dv[i] : d0; d1 = sel ? dv[i + 1] : d1;
    }
    const float w = c1 - c0, hgt = e1 - e0, delta = hgt / w;
    const float theta = (x - c0) / w, T = theta * (1.0f - theta), omt = 1.0f - theta;
    const float curv = d0 + d1 - 2.0f * delta;
    const float Rr = delta * theta * theta + d0 * T;
    const float num = hgt * Rr, den = delta + curv * T;
    const float Q = d1 * theta * theta + 2.0f * delta * T + d0 * omt * omt;
    const float dnum = delta * delta * Q;
    // reverse mode
    const float dnum_b = gld / dnum;
    float den_b = -2.0f * gld / den - gy * num / (den * den);
    const float num_b = gy / den;
    float e0_b = gy, e1_b = 0.0f, c0_b = 0.0f, c1_b = 0.0f, d0_b = 0.0f, d1_b = 0.0f;
    float delta_b = dnum_b * 2.0f * delta * Q;
    const float Q_b = dnum_b * delta * delta;
    d1_b += Q_b * theta * theta;
    float theta_b = Q_b * (2.0f * d1 * theta - 2.0f * d0 * omt);
    float T_b = Q_b * 2.0f * delta;
    delta_b += Q_b * 2.0f * T;
    d0_b += Q_b * omt * omt;
    delta_b += den_b;
    const float curv_b = den_b * T;
    T_b += den_b * curv;
    float hgt_b = num_b * Rr;
    const float R_b = num_b * hgt;
    delta_b += R_b * theta * theta;
    theta_b += R_b * 2.0f * delta * theta;
    d0_b += R_b * T;
    T_b += R_b * d0;
    d0_b += curv_b; d1_b += curv_b; delta_b -= 2.0f * curv_b;
    theta_b += T_b * (1.0f - 2.0f * theta);
    gx = theta_b / w;
    c0_b -= theta_b / w;
    float w_b = -theta_b * theta / w;
    hgt_b += delta_b / w;
    w_b -= delta_b * delta / w;
    c1_b += w_b; c0_b -= w_b;
    e1_b += hgt_b; e0_b -= hgt_b;
    // knots -> cumulative sums -> softmax -> logits (interior knots 1..K-1 only; the
    // end knots are constants, as in the reference's in-place assignment)
    float gcw[K + 1], gch[K + 1];
#pragma unroll
    for (int i = 0; i <= K; ++i) {
      gcw[i] = (i == k ? c0_b : 0.0f) + (i == k + 1 ? c1_b : 0.0f);
      gch[i] = (i == k ? e0_b : 0.0f) + (i == k + 1 ? e1_b : 0.0f);
    }
    {
      float gw[K], suffix = 0.0f, dot = 0.0f;
#pragma unroll
      for (int jj = K - 1; jj >= 0; --jj) {  // w_j feeds S_i for i > j, i <= K-1
        gw[jj] = suffix;
        if (jj >= 1) suffix += two_b * gcw[jj];
      }
#pragma unroll
      for (int jj = 0; jj < K; ++jj) { gw[jj] *= sc_w; dot += pw[jj] * gw[jj]; }
#pragma unroll
      for (int jj = 0; jj < K; ++jj) gp[jj] = pw[jj] * (gw[jj] - dot) * c.inv_div;
    }
    {
      float gh[K], suffix = 0.0f, dot = 0.0f;
#pragma unroll
      for (int jj = K - 1; jj >= 0; --jj) {
        gh[jj] = suffix;
        if (jj >= 1) suffix += two_b * gch[jj];
      }
#pragma unroll
      for (int jj = 0; jj < K; ++jj) { gh[jj] *= sc_h; dot += ph[jj] * gh[jj]; }
#pragma unroll
      for (int jj = 0; jj < K; ++jj) gp[K + jj] = ph[jj] * (gh[jj] - dot) * c.inv_div;
    }
#pragma unroll
    for (int i = 1; i < K; ++i) {
      const float gd = (i == k ? d0_b : 0.0f) + (i == k + 1 ? d1_b : 0.0f);
      gp[2 * K + i - 1] = gd * sg[i];
    }
  }
#pragma unroll
  for (int i = 0; i < NP; ++i) GPit[i] = gp[i];
  return gx;
}

template <int K>
__device__ __forceinline__ void tr_rq_backward_item(int row, int m, int d_tr, const int* __restrict__ tr_cols,
                                                    const float* __restrict__ Pit, const float* __restrict__ xsave,
                                                    float* __restrict__ X, int ldx, float* __restrict__ Gx,
                                                    const float* __restrict__ gld_rows, float* __restrict__ GPit,
                                                    const TrConsts& c) {
  const int col = tr_cols[m];
  const float x = xsave[row * d_tr + m];
  const float gy = Gx[(long long)row * ldx + col];
  X[(long long)row * ldx + col] = x;  // undo the layer
  Gx[(long long)row * ldx + col] = tr_rq_backward_core<K>(Pit, x, gy, gld_rows[row], c, GPit);
}

template <int K>
__global__ void tr_rq_backward(int n, int d_tr, const int* __restrict__ tr_cols, const float* __restrict__ P,
                               const float* __restrict__ xsave, float* __restrict__ X, int ldx,
                               float* __restrict__ Gx, const float* __restrict__ gld_rows,
                               float* __restrict__ GP, TrConsts c) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * d_tr) return;
  const int row = t / d_tr;
  tr_rq_backward_item<K>(row, t - row * d_tr, d_tr, tr_cols, P + (long long)t * (3 * K - 1), xsave, X, ldx, Gx, gld_rows,
                         GP + (long long)t * (3 * K - 1), c);
}

// affine coupling: P row-major [n][2*d_tr] = [shift block | raw-scale block]
__device__ __forceinline__ void tr_affine_forward_item(int row, int m, int d_tr, int additive,
                                                       const int* __restrict__ tr_cols,
                                                       const float* __restrict__ Prow, float* __restrict__ X, int ldx,
                                                       float* __restrict__ xsave, float* __restrict__ logdet,
                                                       float aff_a, float aff_b) {
  const int t = row * d_tr + m;
  const int col = tr_cols[m];
  const float x = X[(long long)row * ldx + col];
  xsave[t] = x;
  const float shift = Prow[m];  // the row's parameters (global row-major, or an LDS tile)
  float scale = 1.0f, ls = 0.0f;
  if (!additive) {
    scale = 1.0f / (1.0f + expf(-(Prow[d_tr + m] + aff_a))) + aff_b;
    ls = logf(scale);
  }
  X[(long long)row * ldx + col] = x * scale + shift;
  if (!additive) atomicAdd(&logdet[row], ls);
}

__global__ void tr_affine_forward(int n, int d_tr, int additive, const int* __restrict__ tr_cols,
                                  const float* __restrict__ P, float* __restrict__ X, int ldx,
                                  float* __restrict__ xsave, float* __restrict__ logdet, float aff_a,
                                  float aff_b) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * d_tr) return;
  const int row = t / d_tr;
  tr_affine_forward_item(row, t - row * d_tr, d_tr, additive, tr_cols, P + (long long)row * (additive ? d_tr : 2 * d_tr), X,
                         ldx, xsave, logdet, aff_a, aff_b);
}

__device__ __forceinline__ void tr_affine_backward_item(int row, int m, int d_tr, int additive,
                                                        const int* __restrict__ tr_cols,
                                                        const float* __restrict__ Prow,
                                                        const float* __restrict__ xsave, float* __restrict__ X,
                                                        int ldx, float* __restrict__ Gx,
                                                        const float* __restrict__ gld_rows,
                                                        float* __restrict__ GProw, float aff_a, float aff_b) {
  const int t = row * d_tr + m;
  const float gld = gld_rows[row];
  const int col = tr_cols[m];
  const float x = xsave[t];
  const float gy = Gx[(long long)row * ldx + col];
  X[(long long)row * ldx + col] = x;
  GProw[m] = gy;
  if (additive) return;
  const float sig = 1.0f / (1.0f + expf(-(Prow[d_tr + m] + aff_a)));
  const float scale = sig + aff_b;
  const float s_b = gy * x + gld / scale;
  GProw[d_tr + m] = s_b * sig * (1.0f - sig);
  Gx[(long long)row * ldx + col] = gy * scale;
}

__global__ void tr_affine_backward(int n, int d_tr, int additive, const int* __restrict__ tr_cols,
                                   const float* __restrict__ P, const float* __restrict__ xsave,
                                   float* __restrict__ X, int ldx, float* __restrict__ Gx,
                                   const float* __restrict__ gld_rows, float* __restrict__ GP, float aff_a,
                                   float aff_b) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * d_tr) return;
  const int row = t / d_tr;
  const long long np = additive ? d_tr : 2 * d_tr;
  tr_affine_backward_item(row, t - row * d_tr, d_tr, additive, tr_cols, P + row * np, xsave, X, ldx, Gx, gld_rows,
                          GP + row * np, aff_a, aff_b);
}

// ---------------------------------------------------------------------------
// batch norm between coupling layers (nflows BatchNorm transform, training mode:
// batch mean / unbiased variance, running statistics updated with `momentum`)
// ---------------------------------------------------------------------------

// one workgroup per feature: mean and unbiased variance over the batch
__global__ void __launch_bounds__(256) tr_bn_stats(int n, const float* __restrict__ X, int ldx,
                                                   const int* __restrict__ cols, float* __restrict__ mean,
                                                   float* __restrict__ var) {
  __shared__ float sm[256];
  const int i = blockIdx.x, col = cols[i];
  float s = 0.0f;
  for (int r = threadIdx.x; r < n; r += 256) s += X[(long long)r * ldx + col];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) { if (threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off]; __syncthreads(); }
  const float mu = sm[0] / (float)n;
  __syncthreads();
  float q = 0.0f;
  for (int r = threadIdx.x; r < n; r += 256) { const float d = X[(long long)r * ldx + col] - mu; q += d * d; }
  sm[threadIdx.x] = q;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) { if (threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off]; __syncthreads(); }
  if (threadIdx.x == 0) { mean[i] = mu; var[i] = sm[0] / (float)(n - 1); }
}

// y = w * (x - mean) / sqrt(var + eps) + b;  w = softplus(uw) + eps
// (thread 0 also updates the running statistics and adds the layer's constant log|det J|)
__global__ void tr_bn_forward(int n, int D, float* __restrict__ X, int ldx, const int* __restrict__ cols,
                              const float* __restrict__ uw, const float* __restrict__ bias,
                              const float* __restrict__ mean, const float* __restrict__ var, float eps,
                              float* __restrict__ xsave, float* __restrict__ run_mean,
                              float* __restrict__ run_var, float momentum, int update_running,
                              float* __restrict__ ld_const) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0) {
    float ld = 0.0f;
    for (int i = 0; i < D; ++i) {
      const float w = nsa_softplus(uw[i]) + eps;
      ld += logf(w) - 0.5f * logf(var[i] + eps);
      if (update_running) {
        run_mean[i] = run_mean[i] * (1.0f - momentum) + mean[i] * momentum;
        run_var[i] = run_var[i] * (1.0f - momentum) + var[i] * momentum;
      }
    }
    ld_const[0] += ld;
  }
  if (t >= n * D) return;
  const int row = t / D, i = t - row * D;
  const int col = cols[i];
  const float x = X[(long long)row * ldx + col];
  xsave[t] = x;
  const float w = nsa_softplus(uw[i]) + eps;
  X[(long long)row * ldx + col] = w * ((x - mean[i]) / sqrtf(var[i] + eps)) + bias[i];
}

// per feature: T1 = sum gy, T2 = sum gy * xhat
__global__ void __launch_bounds__(256) tr_bn_backward_reduce(int n, const float* __restrict__ Gx, int ldx,
                                                             const int* __restrict__ cols, const float* __restrict__ xsave,
                                                             int D, const float* __restrict__ mean,
                                                             const float* __restrict__ var, float eps,
                                                             float* __restrict__ T1, float* __restrict__ T2) {
  __shared__ float s1[256], s2[256];
  const int i = blockIdx.x, col = cols[i];
  const float mu = mean[i], istd = 1.0f / sqrtf(var[i] + eps);
  float a = 0.0f, b = 0.0f;
  for (int r = threadIdx.x; r < n; r += 256) {
    const float gy = Gx[(long long)r * ldx + col];
    a += gy;
    b += gy * ((xsave[(long long)r * D + i] - mu) * istd);
  }
  s1[threadIdx.x] = a; s2[threadIdx.x] = b;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) { s1[threadIdx.x] += s1[threadIdx.x + off]; s2[threadIdx.x] += s2[threadIdx.x + off]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { T1[i] = s1[0]; T2[i] = s2[0]; }
}

// gl_sum[0] = sum_i d loss / d logdet_i: the layer's constant log|det J| enters every row's
// log_prob (-1 for the normalised losses of nsa_train_step; the sum of the upstream gradients
// under autograd, nsa_train_backward)
__global__ void tr_bn_backward_apply(int n, int D, float* __restrict__ X, float* __restrict__ Gx, int ldx,
                                     const int* __restrict__ cols, const float* __restrict__ xsave,
                                     const float* __restrict__ uw, const float* __restrict__ mean,
                                     const float* __restrict__ var, float eps, const float* __restrict__ T1,
                                     const float* __restrict__ T2, float* __restrict__ g_uw,
                                     float* __restrict__ g_bias, const float* __restrict__ gl_sum,
                                     int batch_stats) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * D) return;
  const int row = t / D, i = t - row * D;
  const int col = cols[i];
  const float w = nsa_softplus(uw[i]) + eps;
  const float ve = var[i] + eps, istd = 1.0f / sqrtf(ve);
  const float x = xsave[t], xc = x - mean[i];
  const float S1 = w * T1[i], S2 = w * T2[i];
  const float gl = gl_sum[0];
  // batch statistics depend on the rows (training mode); running statistics are constants
  const float g_var = batch_stats ? (-0.5f * S2 - 0.5f * gl) / ve : 0.0f;
  const float g_mean = batch_stats ? -S1 * istd : 0.0f;
  const float g = Gx[(long long)row * ldx + col] * w;
  Gx[(long long)row * ldx + col] = g * istd + g_var * 2.0f * xc / (float)(n - 1) + g_mean / (float)n;
  X[(long long)row * ldx + col] = x;
  if (row == 0) {
    const float sig = 1.0f / (1.0f + expf(-uw[i]));
    g_uw[i] = (T2[i] + gl / w) * sig;
    g_bias[i] = T1[i];
  }
}

// Training-mode batch norm in ONE launch per direction (the fused trainer; the two-launch pairs above stay for the
// deterministic layer-wise path): the transform is independent per feature, so workgroup i owns column i - its
// statistics (the sums of tr_bn_stats / tr_bn_backward_reduce, same order) and then its own column of the batch.
// A launch of these sizes is ~4.5 us of latency whatever it does: the reference's default RealNVP paid it four
// times per batch-norm layer and step.  The layer's constant log|det J| is added per column with a float atomic.
__global__ void __launch_bounds__(256) tr_bn_forward_fused(int n, int D, float* __restrict__ X, int ldx,
                                                           const int* __restrict__ cols, const float* __restrict__ uw,
                                                           const float* __restrict__ bias, float eps,
                                                           float* __restrict__ mean, float* __restrict__ var,
                                                           float* __restrict__ xsave, float* __restrict__ run_mean,
                                                           float* __restrict__ run_var, float momentum,
                                                           int update_running, float* __restrict__ ld_const) {
  __shared__ float sm[256];
  const int i = blockIdx.x, col = cols[i];
  float s = 0.0f;
  for (int r = threadIdx.x; r < n; r += 256) s += X[(long long)r * ldx + col];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) { if (threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off]; __syncthreads(); }
  const float mu = sm[0] / (float)n;
  __syncthreads();
  float q = 0.0f;
  for (int r = threadIdx.x; r < n; r += 256) { const float d = X[(long long)r * ldx + col] - mu; q += d * d; }
  sm[threadIdx.x] = q;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) { if (threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off]; __syncthreads(); }
  const float v = sm[0] / (float)(n - 1);
  const float w = nsa_softplus(uw[i]) + eps;
  if (threadIdx.x == 0) {
    mean[i] = mu;
    var[i] = v;
    if (update_running) {
      run_mean[i] = run_mean[i] * (1.0f - momentum) + mu * momentum;
      run_var[i] = run_var[i] * (1.0f - momentum) + v * momentum;
    }
    atomicAdd(ld_const, logf(w) - 0.5f * logf(v + eps));
  }
  const float istd_den = sqrtf(v + eps), b = bias[i];
  for (int r = threadIdx.x; r < n; r += 256) {
    const float x = X[(long long)r * ldx + col];
    xsave[(long long)r * D + i] = x;
    X[(long long)r * ldx + col] = w * ((x - mu) / istd_den) + b;
  }
}

__global__ void __launch_bounds__(256) tr_bn_backward_fused(int n, int D, float* __restrict__ X, float* __restrict__ Gx,
                                                            int ldx, const int* __restrict__ cols,
                                                            const float* __restrict__ xsave, const float* __restrict__ uw,
                                                            const float* __restrict__ mean, const float* __restrict__ var,
                                                            float eps, float* __restrict__ g_uw, float* __restrict__ g_bias,
                                                            const float* __restrict__ gl_sum, int batch_stats) {
  __shared__ float s1[256], s2[256];
  const int i = blockIdx.x, col = cols[i];
  const float mu = mean[i], ve = var[i] + eps, istd = 1.0f / sqrtf(ve);
  float a = 0.0f, b = 0.0f;
  for (int r = threadIdx.x; r < n; r += 256) {
    const float gy = Gx[(long long)r * ldx + col];
    a += gy;
    b += gy * ((xsave[(long long)r * D + i] - mu) * istd);
  }
  s1[threadIdx.x] = a; s2[threadIdx.x] = b;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) { s1[threadIdx.x] += s1[threadIdx.x + off]; s2[threadIdx.x] += s2[threadIdx.x + off]; }
    __syncthreads();
  }
  const float T1 = s1[0], T2 = s2[0];
  const float w = nsa_softplus(uw[i]) + eps;
  const float S1 = w * T1, S2 = w * T2, gl = gl_sum[0];
  const float g_var = batch_stats ? (-0.5f * S2 - 0.5f * gl) / ve : 0.0f;
  const float g_mean = batch_stats ? -S1 * istd : 0.0f;
  for (int r = threadIdx.x; r < n; r += 256) {
    const float x = xsave[(long long)r * D + i], xc = x - mu;
    const float g = Gx[(long long)r * ldx + col] * w;
    Gx[(long long)r * ldx + col] = g * istd + g_var * 2.0f * xc / (float)(n - 1) + g_mean / (float)n;
    X[(long long)r * ldx + col] = x;
  }
  if (threadIdx.x == 0) {
    const float sig = 1.0f / (1.0f + expf(-uw[i]));
    g_uw[i] = (T2 + gl / w) * sig;
    g_bias[i] = T1;
  }
}

// eval-mode batch norm (running statistics), used by the validation pass
__global__ void tr_bn_eval(int n, int D, float* __restrict__ X, int ldx, const int* __restrict__ cols,
                           const float* __restrict__ uw, const float* __restrict__ bias,
                           const float* __restrict__ run_mean, const float* __restrict__ run_var, float eps,
                           float* __restrict__ xsave, float* __restrict__ ld_const) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0) {
    float ld = 0.0f;
    for (int i = 0; i < D; ++i) ld += logf(nsa_softplus(uw[i]) + eps) - 0.5f * logf(run_var[i] + eps);
    ld_const[0] += ld;
  }
  if (t >= n * D) return;
  const int row = t / D, i = t - row * D;
  const int col = cols[i];
  const float w = nsa_softplus(uw[i]) + eps;
  const float x = X[(long long)row * ldx + col];
  xsave[t] = x;
  X[(long long)row * ldx + col] = w * ((x - run_mean[i]) / sqrtf(run_var[i] + eps)) + bias[i];
}

// ---------------------------------------------------------------------------
// act norm (nflows ActNorm: y = exp(log_scale) x + shift, log|det J| = sum log_scale, with the
// data-dependent initialisation of the first training batch)
// ---------------------------------------------------------------------------

// one workgroup per feature: std = unbiased std of the column, log_scale = -log(std),
// shift = -mean(x / std)
__global__ void __launch_bounds__(256) tr_actnorm_init(int n, const float* __restrict__ X, int ldx,
                                                       const int* __restrict__ cols, float* __restrict__ log_scale,
                                                       float* __restrict__ shift) {
  __shared__ float sm[256];
  const int i = blockIdx.x, col = cols[i];
  float s = 0.0f;
  for (int row = threadIdx.x; row < n; row += 256) s += X[(long long)row * ldx + col];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
    __syncthreads();
  }
  const float mean = sm[0] / (float)n;
  __syncthreads();
  float q = 0.0f;
  for (int row = threadIdx.x; row < n; row += 256) {
    const float d = X[(long long)row * ldx + col] - mean;
    q += d * d;
  }
  sm[threadIdx.x] = q;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
    __syncthreads();
  }
  const float std_ = sqrtf(sm[0] / (float)(n - 1));
  __syncthreads();
  float m2 = 0.0f;
  for (int row = threadIdx.x; row < n; row += 256) m2 += X[(long long)row * ldx + col] / std_;
  sm[threadIdx.x] = m2;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    log_scale[i] = -logf(std_);
    shift[i] = -(sm[0] / (float)n);
  }
}

// forward in place; xin (if not null) keeps the input for the backward pass
__global__ void tr_actnorm_forward(int n, int D, float* __restrict__ X, int ldx, const int* __restrict__ cols,
                                   const float* __restrict__ log_scale, const float* __restrict__ shift,
                                   float* __restrict__ xin, float* __restrict__ ld_const) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0) {
    float ld = 0.0f;
    for (int i = 0; i < D; ++i) ld += log_scale[i];
    ld_const[0] += ld;
  }
  if (t >= n * D) return;
  const int row = t / D, i = t - row * D;
  const int col = cols[i];
  const float x = X[(long long)row * ldx + col];
  if (xin) xin[t] = x;
  X[(long long)row * ldx + col] = expf(log_scale[i]) * x + shift[i];
}

// one workgroup per feature: d log_scale = sum gy x scale + gl_sum (the layer's log|det J| enters
// every row's log_prob; gl_sum = sum_i d loss / d logdet_i), d shift = sum gy; gx = gy scale; x restored
__global__ void __launch_bounds__(256) tr_actnorm_backward(int n, float* __restrict__ X, float* __restrict__ Gx,
                                                           int ldx, const int* __restrict__ cols,
                                                           const float* __restrict__ xin, int D,
                                                           const float* __restrict__ log_scale,
                                                           float* __restrict__ g_log_scale,
                                                           float* __restrict__ g_shift,
                                                           const float* __restrict__ gl_sum) {
  __shared__ float sa[256], sb[256];
  const int i = blockIdx.x, col = cols[i];
  const float scale = expf(log_scale[i]);
  float a = 0.0f, b = 0.0f;
  for (int row = threadIdx.x; row < n; row += 256) {
    const float gy = Gx[(long long)row * ldx + col];
    const float x = xin[(long long)row * D + i];
    a += gy * x * scale;
    b += gy;
    Gx[(long long)row * ldx + col] = gy * scale;
    X[(long long)row * ldx + col] = x;
  }
  sa[threadIdx.x] = a;
  sb[threadIdx.x] = b;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) { sa[threadIdx.x] += sa[threadIdx.x + off]; sb[threadIdx.x] += sb[threadIdx.x + off]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    g_log_scale[i] = sa[0] + gl_sum[0];
    g_shift[i] = sb[0];
  }
}

// ---------------------------------------------------------------------------
// LU-linear layer: W = L U, unit-diagonal L, diag(U) = softplus(raw) + eps
// ---------------------------------------------------------------------------

__global__ void tr_lu_build(int D, const float* __restrict__ lower, const float* __restrict__ upper,
                            const float* __restrict__ raw, float eps, float* __restrict__ Lm,
                            float* __restrict__ Um, float* __restrict__ ld_const) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0) {
    float ld = 0.0f;
    for (int i = 0; i < D; ++i) ld += logf(nsa_softplus(raw[i]) + eps);
    ld_const[0] += ld;
  }
  if (t >= D * D) return;
  const int i = t / D, j = t - i * D;
  float l = 0.0f, u = 0.0f;
  if (j < i) l = lower[i * (i - 1) / 2 + j];
  else if (j == i) { l = 1.0f; u = nsa_softplus(raw[i]) + eps; }
  else u = upper[i * D - i * (i + 1) / 2 + (j - i - 1)];
  Lm[t] = l;
  Um[t] = u;
}

// dst[row][k] = src[row][cols[k]]  (gather)  /  dst[row][cols[k]] = src[row][k]  (scatter)
// ---- the LU-linear layer in one launch per direction (D <= 64; the layer-wise sequence above is 5 + 6 launches of
// 4-16 us each, and the reference's default RealNVP has one such layer per coupling) ----
// A workgroup takes groups of R = 256 / D rows: thread (r, i) owns one output element, the factors sit in LDS
// (transposed for the forward products so that lanes run along the contiguous index).
//   forward:  xin = x[:, cols];  t = xin U^T;  y = t L^T + b   (xin, t saved; block 0 adds log|det| = sum log diag U
//             and clears the accumulators of the backward pass)
__global__ void __launch_bounds__(256) tr_lu_forward_fused(int n, int D, float* __restrict__ X, const int* __restrict__ cols,
                                                           const float* __restrict__ lower, const float* __restrict__ upper,
                                                           const float* __restrict__ raw, float eps,
                                                           const float* __restrict__ bias, float* __restrict__ xin,
                                                           float* __restrict__ tsave, float* __restrict__ gL,
                                                           float* __restrict__ gU, float* __restrict__ ld_const,
                                                           unsigned* __restrict__ done) {
  extern __shared__ float lu_sm[];
  float* UT = lu_sm;            // UT[j * D + i] = U[i][j]
  float* LT = UT + D * D;       // LT[j * D + i] = L[i][j]
  float* xs = LT + D * D;       // [R][D]
  float* ts = xs + 256;
  const int tid = threadIdx.x;
  const int R = 256 / D;
  for (int e = tid; e < D * D; e += 256) {
    const int i = e / D, j = e - i * D;
    float l = 0.0f, u = 0.0f;
    if (j < i) l = lower[i * (i - 1) / 2 + j];
    else if (j == i) { l = 1.0f; u = nsa_softplus(raw[i]) + eps; }
    else u = upper[i * D - i * (i + 1) / 2 + (j - i - 1)];
    LT[j * D + i] = l;
    UT[j * D + i] = u;
    if (blockIdx.x == 0) { gL[e] = 0.0f; gU[e] = 0.0f; }
    if (blockIdx.x == 0 && e == 0) *done = 0u;  // ticket counter of tr_lu_backward_fused
  }
  if (blockIdx.x == 0 && tid == 0) {
    float ld = 0.0f;
    for (int i = 0; i < D; ++i) ld += logf(nsa_softplus(raw[i]) + eps);
    ld_const[0] += ld;
  }
  const int r = tid / D, i = tid - r * D;
  const bool mine = r < R;
  const int col = mine ? cols[i] : 0;
  const float b = mine ? bias[i] : 0.0f;
  __syncthreads();
  for (int g = blockIdx.x; g * R < n; g += gridDim.x) {
    const int row = g * R + r;
    const bool ok = mine && row < n;
    const float xv = ok ? X[(long long)row * D + col] : 0.0f;
    if (mine) xs[r * D + i] = xv;
    if (ok) xin[(long long)row * D + i] = xv;
    __syncthreads();
    float t = 0.0f;
    if (mine)
      for (int j = 0; j < D; ++j) t += UT[j * D + i] * xs[r * D + j];
    if (mine) ts[r * D + i] = t;
    if (ok) tsave[(long long)row * D + i] = t;
    __syncthreads();
    if (ok) {
      float y = b;
      for (int j = 0; j < D; ++j) y += LT[j * D + i] * ts[r * D + j];
      X[(long long)row * D + i] = y;
    }
    __syncthreads();
  }
}

//   backward: gt = gy L;  g_x[:, cols] = gt U;  dL += gy^T t;  dU += gt^T xin;  db += colsum gy;  x[:, cols] restored
//   (per-workgroup sums in registers, one float atomic per element and workgroup at the end; tr_lu_grads follows)
__global__ void __launch_bounds__(256) tr_lu_backward_fused(int n, int D, float* __restrict__ X, float* __restrict__ Gx,
                                                            const int* __restrict__ cols, const float* __restrict__ lower,
                                                            const float* __restrict__ upper, const float* __restrict__ raw,
                                                            float eps, const float* __restrict__ xin,
                                                            const float* __restrict__ tsave, float* __restrict__ gL,
                                                            float* __restrict__ gU, float* __restrict__ gb,
                                                            float* __restrict__ g_lower, float* __restrict__ g_upper,
                                                            float* __restrict__ g_raw, const float* __restrict__ gl_sum,
                                                            unsigned* __restrict__ done) {
  extern __shared__ float lu_sm[];
  float* Lm = lu_sm;          // Lm[i * D + j]
  float* Um = Lm + D * D;
  float* gys = Um + D * D;    // [R][D] each
  float* gts = gys + 256;
  float* xs = gts + 256;
  float* ts = xs + 256;
  const int tid = threadIdx.x;
  const int R = 256 / D;
  for (int e = tid; e < D * D; e += 256) {
    const int i = e / D, j = e - i * D;
    float l = 0.0f, u = 0.0f;
    if (j < i) l = lower[i * (i - 1) / 2 + j];
    else if (j == i) { l = 1.0f; u = nsa_softplus(raw[i]) + eps; }
    else u = upper[i * D - i * (i + 1) / 2 + (j - i - 1)];
    Lm[e] = l;
    Um[e] = u;
  }
  const int r = tid / D, j = tid - r * D;
  const bool mine = r < R;
  const int col = mine ? cols[j] : 0;
  constexpr int PER = 16;  // (i, j) pairs of the D x D gradients per thread: 64 * 64 / 256
  float accL[PER], accU[PER], accb = 0.0f;
#pragma unroll
  for (int p = 0; p < PER; ++p) { accL[p] = 0.0f; accU[p] = 0.0f; }
  __syncthreads();
  for (int g = blockIdx.x; g * R < n; g += gridDim.x) {
    const int row = g * R + r;
    const bool ok = mine && row < n;
    if (mine) {
      gys[r * D + j] = ok ? Gx[(long long)row * D + j] : 0.0f;
      xs[r * D + j] = ok ? xin[(long long)row * D + j] : 0.0f;
      ts[r * D + j] = ok ? tsave[(long long)row * D + j] : 0.0f;
    }
    __syncthreads();
    float gt = 0.0f;
    if (mine)
      for (int i = 0; i < D; ++i) gt += gys[r * D + i] * Lm[i * D + j];
    if (mine) gts[r * D + j] = gt;
    __syncthreads();
    if (ok) {
      float gx = 0.0f;
      for (int i = 0; i < D; ++i) gx += gts[r * D + i] * Um[i * D + j];
      Gx[(long long)row * D + col] = gx;
      X[(long long)row * D + col] = xs[r * D + j];
    }
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const int e = tid + 256 * p;
      if (e < D * D) {
        const int i = e / D, jj = e - i * D;
        float a = 0.0f, u = 0.0f;
        for (int rr = 0; rr < R; ++rr) {
          a += gys[rr * D + i] * ts[rr * D + jj];
          u += gts[rr * D + i] * xs[rr * D + jj];
        }
        accL[p] += a;
        accU[p] += u;
      }
    }
    if (tid < D)
      for (int rr = 0; rr < R; ++rr) accb += gys[rr * D + tid];
    __syncthreads();
  }
#pragma unroll
  for (int p = 0; p < PER; ++p) {
    const int e = tid + 256 * p;
    if (e < D * D) {
      atomicAdd(&gL[e], accL[p]);
      atomicAdd(&gU[e], accU[p]);
    }
  }
  if (tid < D) atomicAdd(&gb[tid], accb);
  // The workgroup that finishes LAST turns the summed D x D gradients into those of the triangular
  // parameters (what tr_lu_grads did in a launch of its own): a ticket counter behind the sums.
  __shared__ unsigned ticket;
  __threadfence();
  __syncthreads();
  if (tid == 0) ticket = atomicAdd(done, 1u);
  __syncthreads();
  if (ticket != gridDim.x - 1) return;
  __threadfence();
  for (int e = tid; e < D * D; e += 256) {
    const int i = e / D, j = e - i * D;
    const float l = __hip_atomic_load(&gL[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float u = __hip_atomic_load(&gU[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (j < i) g_lower[i * (i - 1) / 2 + j] = l;
    else if (j > i) g_upper[i * D - i * (i + 1) / 2 + (j - i - 1)] = u;
    else {
      const float d = nsa_softplus(raw[i]) + eps;
      g_raw[i] = (u + gl_sum[0] / d) / (1.0f + expf(-raw[i]));
    }
  }
  if (tid == 0) *done = 0u;  // ready for the next step
}

__global__ void tr_gather_cols(int n, int D, const float* __restrict__ src, int lds_, const int* __restrict__ cols,
                               float* __restrict__ dst, int scatter) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * D) return;
  const int row = t / D, k = t - row * D;
  if (scatter) dst[(long long)row * lds_ + cols[k]] = src[t];
  else dst[t] = src[(long long)row * lds_ + cols[k]];
}

__global__ void tr_lu_grads(int D, const float* __restrict__ gL, const float* __restrict__ gU,
                            const float* __restrict__ raw, float eps, float* __restrict__ g_lower,
                            float* __restrict__ g_upper, float* __restrict__ g_raw,
                            const float* __restrict__ gl_sum) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= D * D) return;
  const int i = t / D, j = t - i * D;
  if (j < i) g_lower[i * (i - 1) / 2 + j] = gL[t];
  else if (j > i) g_upper[i * D - i * (i + 1) / 2 + (j - i - 1)] = gU[t];
  else {
    const float d = nsa_softplus(raw[i]) + eps;
    g_raw[i] = (gU[t] + gl_sum[0] / d) / (1.0f + expf(-raw[i]));
  }
}

// ---------------------------------------------------------------------------
// loss, gradient seed, gradient norm, AdamW
// ---------------------------------------------------------------------------

// per row: log_prob = -0.5/var * sum z^2 - log_z + logdet
// loss = -sum_i w_i log_prob_i / sum_i w_i   (w_i = 1 when `weights` is NULL: -mean(log_prob),
// flowmodel/base.py:404-412); seeds the backward pass: Gx = z * w_i / (var * W),
// gld_rows[i] = d loss / d logdet_i = -w_i / W.
// One workgroup per 64 rows, 4 threads per row (coalesced row segments); every workgroup sums
// the weights itself (fixed order) and adds its part of the loss with one atomic.
#define TR_LOSS_ROWS 64
__global__ void __launch_bounds__(256) tr_loss(int n, int D, const float* __restrict__ X, int ldx,
                                               const float* __restrict__ logdet,
                                               const float* __restrict__ ld_const,
                                               const float* __restrict__ weights, float inv_var, float log_z,
                                               float* __restrict__ Gx, float* __restrict__ gld_rows,
                                               float* __restrict__ loss_acc,
                                               float* __restrict__ log_prob_out, int want_grad,
                                               float* __restrict__ gl_sum) {
  __shared__ float sm[256];
  if (want_grad && blockIdx.x == 0 && threadIdx.x == 0) gl_sum[0] = -1.0f;  // sum_i -w_i / W
  float W = (float)n;
  if (weights != nullptr) {
    float wsum = 0.0f;
    for (int row = threadIdx.x; row < n; row += 256) wsum += weights[row];
    sm[threadIdx.x] = wsum;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if (threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
      __syncthreads();
    }
    W = sm[0];
    __syncthreads();
  }
  const int row = blockIdx.x * TR_LOSS_ROWS + (threadIdx.x >> 2), part = threadIdx.x & 3;
  float local = 0.0f;
  if (row < n) {
    const float wi = (weights ? weights[row] : 1.0f) / W;
    float e = 0.0f;
    for (int c = part; c < D; c += 4) {
      const float z = X[(long long)row * ldx + c];
      e += z * z;
      if (want_grad) Gx[(long long)row * ldx + c] = z * inv_var * wi;
    }
    e += __shfl_xor(e, 1);
    e += __shfl_xor(e, 2);
    if (part == 0) {
      if (want_grad) gld_rows[row] = -wi;
      const float lp = -0.5f * inv_var * e - log_z + (logdet[row] + ld_const[0]);
      if (log_prob_out) log_prob_out[row] = lp;
      local = lp * wi;
    }
  }
  sm[threadIdx.x] = local;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(&loss_acc[0], -sm[0]);
}

// sum of squares of the flat gradient: TR_SQ_BLOCKS partial sums, added up in a fixed order by
// the consumer (tr_adamw) => deterministic
#define TR_SQ_BLOCKS 64
#define TR_SQ_MAX 1024  // partial sums of tr_finish_grads (one per block), reduced in a fixed order by every block of tr_adamw
__global__ void __launch_bounds__(256) tr_grad_sqnorm(long long n, const float* __restrict__ g,
                                                      float* __restrict__ partial) {
  __shared__ float sm[256];
  float s = 0.0f;
  for (long long i = blockIdx.x * 256 + threadIdx.x; i < n; i += 256LL * TR_SQ_BLOCKS) s += g[i] * g[i];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
}

// clip_grad_norm_ (coef = min(1, max_norm / (norm + 1e-6))) fused with the optimiser update of
// FlowModel.get_optimiser (src/nessai/flowmodel/base.py:105-135):
//   mode 0 torch.optim.AdamW (decoupled decay p *= 1 - lr wd), 1 torch.optim.Adam (L2: g += wd p),
//   2 torch.optim.SGD without momentum (g += wd p; p -= lr g)
//   opt (optional, device): {lr, steps done, 1 - beta1^step, sqrt(1 - beta2^step)} maintained by
//   tr_opt_tick - the form a captured HIP graph replays (no per-step host scalars).
__global__ void tr_opt_tick(float* __restrict__ opt, float beta1, float beta2) {
  const float step = opt[1] + 1.0f;
  opt[1] = step;
  opt[2] = 1.0f - powf(beta1, step);
  opt[3] = sqrtf(1.0f - powf(beta2, step));
}

__global__ void __launch_bounds__(256) tr_adamw(long long n, float* __restrict__ p, const float* __restrict__ g,
                                                float* __restrict__ m, float* __restrict__ v,
                                                const float* __restrict__ sq_partial, int n_partial, float max_norm,
                                                float lr, float beta1, float beta2, float eps, float wd, float bc1,
                                                float bc2_sqrt, int mode, const float* __restrict__ opt) {
  __shared__ float sm[256];
  float coef = 1.0f;
  if (max_norm > 0.0f) {  // block-uniform; every block sums the partials in the same order
    float sq = 0.0f;
    for (int b = threadIdx.x; b < n_partial; b += 256) sq += sq_partial[b];
    sm[threadIdx.x] = sq;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if (threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
      __syncthreads();
    }
    coef = max_norm / (sqrtf(sm[0]) + 1e-6f);
    coef = coef < 1.0f ? coef : 1.0f;
  }
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (opt != nullptr) {
    lr = opt[0];
    bc1 = opt[2];
    bc2_sqrt = opt[3];
  }
  float grad = g[i] * coef;
  float pi = p[i];
  if (mode == 0) pi *= 1.0f - lr * wd;
  else if (wd != 0.0f) grad = grad + wd * pi;
  if (mode == 2) {
    p[i] = pi - lr * grad;
    return;
  }
  const float mi = beta1 * m[i] + (1.0f - beta1) * grad;
  const float vi = beta2 * v[i] + (1.0f - beta2) * grad * grad;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  pi -= (lr / bc1) * (mi / denom);
  p[i] = pi;
}

__global__ void tr_copy_rows(long long n, const float* __restrict__ src, float* __restrict__ dst) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

#include "nsa_train_fused.h"
#include "nsa_train_fused2.h"
#include "nsa_train_fused3.h"

// Everything a pass needs before its first coupling layer, in one launch instead of five: the batch copied into
// the workspace, the transposed weight copies of the fused kernels, the flat gradient, log|det J| and the scalar
// block zeroed.  wt / grads may be null.
__global__ void tr_prepare(long long n_x, const float* __restrict__ x, float* __restrict__ wx, long long n_params,
                           const float* __restrict__ p, const int* __restrict__ tmap, float* __restrict__ wt,
                           float* __restrict__ grads, long long n, float* __restrict__ logdet,
                           float* __restrict__ scal) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i < n_x) wx[i] = x[i];
  if (i < n_params) {
    if (wt != nullptr) wt[tmap[i]] = p[i];
    if (grads != nullptr) grads[i] = 0.0f;
  }
  if (i < n) logdet[i] = 0.0f;
  if (i < 64) scal[i] = 0.0f;
}

// ---------------------------------------------------------------------------
// conditional flows (layer-wise path): the four element-wise steps around the GEMMs
// ---------------------------------------------------------------------------

// xc[row] = {x[row][id_cols[0 .. d_id)], ctx[row][0 .. C)}: the input of a conditional coupling's initial layer
__global__ void tr_concat_ctx(int n, int d_id, int C, const float* __restrict__ X, int ldx,
                              const int* __restrict__ id_cols, const float* __restrict__ ctx,
                              float* __restrict__ xc) {
  const int W = d_id + C;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n * W) return;
  const int row = (int)(t / W), i = (int)(t - (long long)row * W);
  xc[t] = i < d_id ? X[(long long)row * ldx + id_cols[i]] : ctx[(long long)row * C + (i - d_id)];
}

// h_next = h + t * sigmoid(g)   (glu(cat(temps, context_layer(context))) of a residual block)
__global__ void tr_gate_forward(long long m, const float* __restrict__ h, const float* __restrict__ t,
                                const float* __restrict__ g, float* __restrict__ h_next) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  h_next[i] = h[i] + t[i] * (1.0f / (1.0f + expf(-g[i])));
}

// in place: t <- dL/dt = gh * sigmoid(g);  g <- dL/dg = gh * t * sigmoid(g) (1 - sigmoid(g))
__global__ void tr_gate_backward(long long m, const float* __restrict__ gh, float* __restrict__ t, float* __restrict__ g) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const float sg = 1.0f / (1.0f + expf(-g[i]));
  const float up = gh[i], tv = t[i];
  t[i] = up * sg;
  g[i] = up * tv * sg * (1.0f - sg);
}

// Gx[row][id_cols[i]] += G[row][i] for i < d_id (the context columns of G carry no gradient anyone needs)
__global__ void tr_scatter_add(int n, int d_id, const float* __restrict__ G, int ldg, float* __restrict__ Gx, int ldx,
                               const int* __restrict__ id_cols) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n * d_id) return;
  const int row = (int)(t / d_id), i = (int)(t - (long long)row * d_id);
  Gx[(long long)row * ldx + id_cols[i]] += G[(long long)row * ldg + i];
}

// ---------------------------------------------------------------------------
// SVD-linear layer (linear_transform = "svd": src/nessai/flows/utils.py:316-323; nflows SVDLinear):
//   y = H1( diag(s) H2(x) ) + b,   H(.) = the Householder sequence x -> x (I - 2 q q^T / |q|^2) for its vectors in
//   turn,  s = eps + softplus(raw),  log|det J| = sum log s.
// One WAVE per row (D <= 128: two elements per lane), the vectors, 2 / |q|^2 and s in LDS.  The backward pass keeps no
// intermediate: a reflection is its own inverse, so the inputs of the reflections are walked back from the output.
//   y = x - (2 a / n2) q, a = q.x, n2 = q.q, b = q.g:   g_x = g - (2 b / n2) q,
//   g_q = -(2 / n2) (a g + b x) + (4 a b / n2^2) q
// Parameter gradients: per-workgroup sums in LDS, one float atomic per element and workgroup at the end.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float tr_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// LDS: Q2[nh][D] | Q1[nh][D] | two_over_n2[2 nh] | s[D]   (+ the gradient sums of the backward kernel behind them)
__device__ __forceinline__ void tr_svd_load(int D, int nh, const float* __restrict__ q1, const float* __restrict__ raw,
                                            const float* __restrict__ q2, float eps, float* sm) {
  float* Q2 = sm;
  float* Q1 = Q2 + nh * D;
  float* ton = Q1 + nh * D;
  float* sd = ton + 2 * nh;
  for (int e = threadIdx.x; e < nh * D; e += blockDim.x) { Q2[e] = q2[e]; Q1[e] = q1[e]; }
  for (int e = threadIdx.x; e < D; e += blockDim.x) sd[e] = eps + nsa_softplus(raw[e]);
  __syncthreads();
  for (int k = threadIdx.x; k < 2 * nh; k += blockDim.x) {
    const float* q = k < nh ? Q2 + k * D : Q1 + (k - nh) * D;
    float n2 = 0.0f;
    for (int e = 0; e < D; ++e) n2 += q[e] * q[e];
    ton[k] = 2.0f / n2;
  }
  __syncthreads();
}

__global__ void __launch_bounds__(256) tr_svd_forward(int n, int D, int nh, float* __restrict__ X,
                                                      const int* __restrict__ cols, const float* __restrict__ q1,
                                                      const float* __restrict__ raw, const float* __restrict__ q2,
                                                      const float* __restrict__ bias, float eps,
                                                      float* __restrict__ xin, float* __restrict__ ld_const) {
  extern __shared__ float svd_sm[];
  tr_svd_load(D, nh, q1, raw, q2, eps, svd_sm);
  const float* Q2 = svd_sm;
  const float* Q1 = Q2 + nh * D;
  const float* ton = Q1 + nh * D;
  const float* sd = ton + 2 * nh;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float ld = 0.0f;
    for (int e = 0; e < D; ++e) ld += logf(sd[e]);
    ld_const[0] += ld;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int e0 = lane, e1 = lane + 64;
  const bool ok0 = e0 < D, ok1 = e1 < D;
  const int c0 = ok0 ? cols[e0] : 0, c1 = ok1 ? cols[e1] : 0;
  for (long long row = (long long)blockIdx.x * 4 + wave; row < n; row += (long long)gridDim.x * 4) {
    float v0 = ok0 ? X[row * D + c0] : 0.0f, v1 = ok1 ? X[row * D + c1] : 0.0f;
    if (ok0) xin[row * D + e0] = v0;
    if (ok1) xin[row * D + e1] = v1;
    for (int k = 0; k < 2 * nh; ++k) {
      if (k == nh) {
        v0 *= ok0 ? sd[e0] : 0.0f;
        v1 *= ok1 ? sd[e1] : 0.0f;
      }
      const float* q = k < nh ? Q2 + k * D : Q1 + (k - nh) * D;
      const float qa = ok0 ? q[e0] : 0.0f, qb = ok1 ? q[e1] : 0.0f;
      const float c = tr_wave_sum(v0 * qa + v1 * qb) * ton[k];
      v0 -= c * qa;
      v1 -= c * qb;
    }
    // (every lane of the wave read its inputs before any lane writes: X's columns are permuted by `cols`)
    if (ok0) X[row * D + e0] = v0 + bias[e0];
    if (ok1) X[row * D + e1] = v1 + bias[e1];
  }
}

// Gx holds d loss / d y on entry (natural order) and d loss / d x in the columns `cols` on exit; X is restored to the
// layer's input; g_q1 / g_raw / g_q2 / g_bias += (the flat gradient was zeroed by tr_prepare).
__global__ void __launch_bounds__(256) tr_svd_backward(int n, int D, int nh, float* __restrict__ X, float* __restrict__ Gx,
                                                       const int* __restrict__ cols, const float* __restrict__ q1,
                                                       const float* __restrict__ raw, const float* __restrict__ q2,
                                                       float eps, const float* __restrict__ xin,
                                                       float* __restrict__ g_q1, float* __restrict__ g_raw,
                                                       float* __restrict__ g_q2, float* __restrict__ g_bias,
                                                       const float* __restrict__ gl_sum) {
  extern __shared__ float svd_sm[];
  tr_svd_load(D, nh, q1, raw, q2, eps, svd_sm);
  const float* Q2 = svd_sm;
  const float* Q1 = Q2 + nh * D;
  const float* ton = Q1 + nh * D;
  const float* sd = ton + 2 * nh;
  float* GQ2 = svd_sm + 2 * nh * D + 2 * nh + D;  // [nh][D]
  float* GQ1 = GQ2 + nh * D;
  float* GS = GQ1 + nh * D;                       // d loss / d s
  float* GB = GS + D;
  for (int e = threadIdx.x; e < 2 * nh * D + 2 * D; e += blockDim.x) GQ2[e] = 0.0f;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int e0 = lane, e1 = lane + 64;
  const bool ok0 = e0 < D, ok1 = e1 < D;
  const int c0 = ok0 ? cols[e0] : 0, c1 = ok1 ? cols[e1] : 0;
  for (long long row = (long long)blockIdx.x * 4 + wave; row < n; row += (long long)gridDim.x * 4) {
    const float x0 = ok0 ? xin[row * D + e0] : 0.0f, x1 = ok1 ? xin[row * D + e1] : 0.0f;
    float g0 = ok0 ? Gx[row * D + e0] : 0.0f, g1 = ok1 ? Gx[row * D + e1] : 0.0f;
    if (ok0) atomicAdd(&GB[e0], g0);
    if (ok1) atomicAdd(&GB[e1], g1);
    // forward again (registers only) to the layer's output without the bias
    float v0 = x0, v1 = x1;
    for (int k = 0; k < 2 * nh; ++k) {
      if (k == nh) {
        v0 *= ok0 ? sd[e0] : 0.0f;
        v1 *= ok1 ? sd[e1] : 0.0f;
      }
      const float* q = k < nh ? Q2 + k * D : Q1 + (k - nh) * D;
      const float qa = ok0 ? q[e0] : 0.0f, qb = ok1 ? q[e1] : 0.0f;
      const float c = tr_wave_sum(v0 * qa + v1 * qb) * ton[k];
      v0 -= c * qa;
      v1 -= c * qb;
    }
    // ... and back: v = output of reflection k, its input is the reflection of v
    for (int k = 2 * nh - 1; k >= 0; --k) {
      const float* q = k < nh ? Q2 + k * D : Q1 + (k - nh) * D;
      float* gq = k < nh ? GQ2 + k * D : GQ1 + (k - nh) * D;
      const float qa = ok0 ? q[e0] : 0.0f, qb = ok1 ? q[e1] : 0.0f;
      const float t = ton[k];                                  // 2 / n2
      const float cy = tr_wave_sum(v0 * qa + v1 * qb) * t;      // q.y 2 / n2 ( = -q.x 2 / n2 )
      const float in0 = v0 - cy * qa, in1 = v1 - cy * qb;       // the reflection's input x
      const float ax = -cy / t;                                 // a = q.x = -q.y (q.(H y) = q.y - 2 q.y)
      const float bg = tr_wave_sum(g0 * qa + g1 * qb);          // b = q.g
      const float k4 = ax * bg * t * t;                         // 4 a b / n2^2
      if (ok0) atomicAdd(&gq[e0], -t * (ax * g0 + bg * in0) + k4 * qa);
      if (ok1) atomicAdd(&gq[e1], -t * (ax * g1 + bg * in1) + k4 * qb);
      g0 -= bg * t * qa;
      g1 -= bg * t * qb;
      v0 = in0;
      v1 = in1;
      if (k == nh) {  // through the diagonal: v = s * u
        const float s0 = ok0 ? sd[e0] : 1.0f, s1 = ok1 ? sd[e1] : 1.0f;
        const float u0 = v0 / s0, u1 = v1 / s1;
        if (ok0) atomicAdd(&GS[e0], g0 * u0);
        if (ok1) atomicAdd(&GS[e1], g1 * u1);
        g0 *= s0;
        g1 *= s1;
        v0 = u0;
        v1 = u1;
      }
    }
    // the layer's input is known exactly (xin): restore it, scatter its gradient
    if (ok0) { X[row * D + c0] = x0; }
    if (ok1) { X[row * D + c1] = x1; }
    __builtin_amdgcn_wave_barrier();
    if (ok0) Gx[row * D + c0] = g0;
    if (ok1) Gx[row * D + c1] = g1;
  }
  __syncthreads();
  const float gl = gl_sum[0];
  for (int e = threadIdx.x; e < nh * D; e += blockDim.x) {
    atomicAdd(&g_q2[e], GQ2[e]);
    atomicAdd(&g_q1[e], GQ1[e]);
  }
  for (int e = threadIdx.x; e < D; e += blockDim.x) {
    // s = eps + softplus(raw): d s / d raw = sigmoid(raw); the constant log|det J| = sum log s enters every row
    const float sig = 1.0f / (1.0f + expf(-raw[e]));
    const float extra = blockIdx.x == 0 ? gl / sd[e] : 0.0f;
    atomicAdd(&g_raw[e], (GS[e] + extra) * sig);
    atomicAdd(&g_bias[e], GB[e]);
  }
}

// batch_norm_within_layers (nflows ResidualBlock with use_batch_norm: torch.nn.BatchNorm1d(features, eps = 1e-3) in
// front of each of the block's two activations; src/nessai/flows/realnvp.py:145, nsf.py:94).  One workgroup per
// feature; sums over the batch in a fixed tree (deterministic).
//   training: batch mean and BIASED variance normalise, the running statistics move by `momentum` (0.1) towards the
//   batch mean and the UNBIASED variance - torch's BatchNorm1d;  otherwise the running statistics normalise.
// Y = gamma (X - mean) invstd + beta;  {mean, invstd} kept for the backward pass.
__global__ void __launch_bounds__(256) tr_bn1d_forward(int n, int H, const float* __restrict__ X,
                                                       float* __restrict__ Y, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ run_mean,
                                                       float* __restrict__ run_var, float eps, float momentum,
                                                       int training, float* __restrict__ save_mean,
                                                       float* __restrict__ save_invstd) {
  __shared__ float sm[256];
  const int col = blockIdx.x;
  float mu, invstd;
  if (training) {
    float a = 0.0f;
    for (int r = threadIdx.x; r < n; r += 256) a += X[(long long)r * H + col];
    sm[threadIdx.x] = a;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if (threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off]; __syncthreads(); }
    mu = sm[0] / (float)n;
    __syncthreads();
    float q = 0.0f;
    for (int r = threadIdx.x; r < n; r += 256) { const float d = X[(long long)r * H + col] - mu; q += d * d; }
    sm[threadIdx.x] = q;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if (threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off]; __syncthreads(); }
    const float ss = sm[0];
    invstd = 1.0f / sqrtf(ss / (float)n + eps);
    if (threadIdx.x == 0) {
      run_mean[col] = run_mean[col] * (1.0f - momentum) + mu * momentum;
      run_var[col] = run_var[col] * (1.0f - momentum) + (ss / (float)(n > 1 ? n - 1 : 1)) * momentum;
    }
  } else {
    mu = run_mean[col];
    invstd = 1.0f / sqrtf(run_var[col] + eps);
  }
  if (threadIdx.x == 0) { save_mean[col] = mu; save_invstd[col] = invstd; }
  const float g = gamma[col], b = beta[col];
  for (int r = threadIdx.x; r < n; r += 256) Y[(long long)r * H + col] = (X[(long long)r * H + col] - mu) * invstd * g + b;
}

// Gx (op)= d loss / d X of the above given Gy; dgamma = sum Gy xhat, dbeta = sum Gy.
//   training: Gx = gamma invstd / n (n Gy - dbeta - xhat dgamma);  otherwise Gx = Gy gamma invstd.
// accumulate != 0: Gx += (the skip path's gradient is already there).  Gy and Gx may be the same buffer.
__global__ void __launch_bounds__(256) tr_bn1d_backward(int n, int H, const float* Gy, const float* __restrict__ X,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ save_mean,
                                                        const float* __restrict__ save_invstd, int training,
                                                        float* Gx, int accumulate, float* __restrict__ dgamma,
                                                        float* __restrict__ dbeta) {
  __shared__ float s1[256], s2[256];
  const int col = blockIdx.x;
  const float mu = save_mean[col], invstd = save_invstd[col];
  float a = 0.0f, b = 0.0f;
  for (int r = threadIdx.x; r < n; r += 256) {
    const float gy = Gy[(long long)r * H + col];
    a += gy;
    b += gy * ((X[(long long)r * H + col] - mu) * invstd);
  }
  s1[threadIdx.x] = a;
  s2[threadIdx.x] = b;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) { s1[threadIdx.x] += s1[threadIdx.x + off]; s2[threadIdx.x] += s2[threadIdx.x + off]; }
    __syncthreads();
  }
  const float db = s1[0], dg = s2[0];
  if (threadIdx.x == 0) { dgamma[col] = dg; dbeta[col] = db; }
  const float g = gamma[col];
  const float k = g * invstd;
  const float inv_n = 1.0f / (float)n;
  for (int r = threadIdx.x; r < n; r += 256) {
    const long long at = (long long)r * H + col;
    const float gy = Gy[at];
    float v;
    if (training) {
      const float xhat = (X[at] - mu) * invstd;
      v = k * (gy - inv_n * (db + xhat * dg));
    } else {
      v = k * gy;
    }
    Gx[at] = accumulate ? Gx[at] + v : v;
  }
}

// Dropout of the residual blocks (nflows ResidualBlock: act -> dropout -> second linear layer; the reference passes
// dropout_probability through, src/nessai/flows/realnvp.py:141, nsf.py:90), training mode only.  The mask of one
// block over the batch, M[n][H] = keep ? 1 / (1 - p) : 0, comes from a counter-based stream so that a step is
// reproducible and independent of how the rows are dealt to workgroups (or ranks):
//   Philox4x32-10, counter = (row lo, row hi, column >> 2, tag), key = (seed lo, seed hi ^ step);
//   element column & 3 of the four outputs; keep <=> output >= threshold (= round(p 2^32))
// tag = 0x44000000 | coupling << 8 | block; {seed, step} live in DEVICE memory (state: int64[2]) - the step ticks
// on the device (tr_dropout_tick) so that an epoch captured into a HIP graph draws new masks at every replay.
// tests/test_gpu_train_modes.py restates the stream in numpy (oracle/philox.py) and feeds the oracle the same masks.
__global__ void tr_dropout_mask(long long n, int H, const long long* __restrict__ state, uint32_t tag,
                                uint32_t threshold, float scale, float* __restrict__ M) {
  const int quads = (H + 3) >> 2;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * quads) return;
  const long long row = t / quads;
  const int q = (int)(t - row * quads);
  const unsigned long long seed = (unsigned long long)state[0], step = (unsigned long long)state[1];
  uint32_t out[4];
  nsa_philox4x32_10((uint32_t)row, (uint32_t)((unsigned long long)row >> 32), (uint32_t)q, tag, (uint32_t)seed,
                    (uint32_t)(seed >> 32) ^ (uint32_t)step, out);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int col = 4 * q + e;
    if (col < H) M[row * H + col] = out[e] >= threshold ? scale : 0.0f;
  }
}
__global__ void tr_dropout_tick(long long* __restrict__ state) { state[1] += 1; }

// host orchestration
// ---------------------------------------------------------------------------

// plan (int64, host): header then one fixed-size record per layer, forward order
//   [0] D  [1] H  [2] L (residual blocks)  [3] n_layers  [4] transform  [5] bins
//   [6] activation  [7] params per transformed feature  [8] n trainable floats  [9] record words
//   [10] offset (in the int32 device table) of the transposition map of the trainable floats
//   [11] conditioner: 0 residual net {W0, b0, {W1, b1, W2, b2} x L, Wf, bf};
//        1 plain MLP {W_in, b_in, {W_h, b_h} x L, W_out, b_out} (src/nessai/flows/nets.py:12-126)
//   [12] offset (int32 device table) of the final column table: logical output i of the flow is
//        physical column table[i] of the batch (permutations never move data)
//   [13] context features C of a conditional flow (0: none).  The context rows of the batch, [n][C] float32, are
//        handed over INSIDE the workspace, at nsa_train_context_offset(plan, n) floats from its start (the entry
//        points keep their signatures); W0 then takes d_id + C inputs - identity features first - and every
//        residual block is gated, h + (W2 act(u) + b2) * sigmoid(Wc context + bc)  (nflows ResidualNet:
//        src/nessai/flows/nsf.py:85-95, flows/realnvp.py:136-146); such a flow trains on the layer-wise kernels
//   [14] dropout probability p of the residual blocks (float bits; 0: none) - applied in training-mode passes only;
//        a flow with p > 0 trains on the layer-wise kernels
//   [15] device address of the dropout state {seed, step} (int64[2]; see tr_dropout_mask), 0 when [14] is 0
//   [16] 1: batch norm within the residual blocks (batch_norm_within_layers; see tr_bn1d_forward); [17] its eps
//        (float bits).  A coupling record then ends with {gamma, beta, running_mean, running_var} of the block's
//        first and second batch norm, L times.  Such a flow trains on the layer-wise kernels.
//   coupling  : {1, d_id, d_tr, id_cols offset, tr_cols offset (int32 device table),
//                flat offsets of W0, b0, {W1, b1, W2, b2} x L, Wf, bf[, {Wc, bc} x L of a conditional flow]}
//   batch norm: {2, cols offset, off unconstrained_weight, off bias, off running_mean,
//                off running_var, eps (float bits), momentum (float bits)}
//   LU linear : {3, cols offset, off lower_entries, off upper_entries, off unconstrained_upper_diag,
//                off bias, eps (float bits)}
//   act norm  : {4, cols offset, off log_scale, off shift, 1 if the data-dependent initialisation is
//                still due (the host clears it after the first training step)}
//   SVD linear: {5, cols offset, off orthogonal_1.q_vectors, off unconstrained_diagonal, off orthogonal_2.q_vectors,
//                off bias, eps (float bits), Householder vectors per sequence}
#define TR_HDR 18
enum { TR_COUPLING = 1, TR_BATCHNORM = 2, TR_LU = 3, TR_ACTNORM = 4, TR_SVD = 5 };

static inline float tr_bits(long long v) {
  const uint32_t u = (uint32_t)v;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

#define TR_SCAL_WORDS (128 + 1024)  // 128 + TR_SQ_MAX
struct TrWs {  // workspace carve (floats)
  float *scal;      // [1] constant log|det J| of the pass, [2] sum_i d loss / d logdet_i, [64..127] phase
                    // timestamps, [128..191] sq-norm partials
  float *x, *gx, *logdet, *gld, *gp, *gh, *gu;
  float* wt;        // transposed copy of the weight matrices (fused kernels)
  float* slab;      // [ceil(n / 32)][npar] per-workgroup weight gradients of the fused backward kernels (or null)
  float* layer_base;
  long long per_layer;
  long long xsave_words;  // n * max(d_tr): size of a coupling layer's saved-input block
  float* ctx;             // [n][C] context rows of a conditional flow (written by the caller), or null
  long long np_words;     // al4(n * np_max): size of a coupling layer's parameter block
  long long mask_off;     // dropout: a coupling layer's masks M[L][al4(n * H)] start this many floats into its block
  long long bn_off;       // batch norm within: {a[L], v[L]} (al4(n * H) each) then {mean0, invstd0, mean1, invstd1}[L] (al4(H) each)
};

static inline bool tr_has_dropout(const long long* plan) { return (uint32_t)plan[14] != 0u; }
static inline bool tr_has_inner_bn(const long long* plan) { return plan[16] != 0; }

// every block of the workspace starts on a 16-byte boundary (the fused kernels copy tiles with 16-byte pieces)
static inline long long al4(long long words) { return (words + 3) & ~3LL; }

static void tr_sizes(const long long* plan, long long n, long long& d_tr_max, long long& np_max,
                     long long& per_layer, long long* mask_off = nullptr, long long* bn_off = nullptr) {
  const long long D = plan[0], H = plan[1], L = plan[2], NL = plan[3], rec = plan[9];
  d_tr_max = 1;
  for (long long b = 0; b < NL; ++b) {
    const long long* r = plan + TR_HDR + b * rec;
    if (r[0] == TR_COUPLING && r[2] > d_tr_max) d_tr_max = r[2];
  }
  np_max = d_tr_max * plan[7];
  long long coupling = al4(n * d_tr_max) + (2 * L + 1) * al4(n * H) + al4(n * np_max);
  if (plan[13] > 0)  // conditional: t[L], g[L] behind the parameters, then xc and its gradient [n][D + C]
    coupling += 2 * L * al4(n * H) + 2 * al4(n * (D + plan[13]));
  if (mask_off != nullptr) *mask_off = coupling;
  if (tr_has_dropout(plan)) coupling += L * al4(n * H);  // the blocks' dropout masks, kept for the backward pass
  if (bn_off != nullptr) *bn_off = coupling;
  if (tr_has_inner_bn(plan)) coupling += L * (2 * al4(n * H) + 4 * al4(H));  // normalised activations + statistics
  const long long linear = 3 * n * D + 4 * D * D + 8 * D;  // x, t, gt | L, U, gL, gU | small vectors
  per_layer = al4(coupling > linear ? coupling : linear);
}

// The fused backward kernels of the second generation leave each workgroup's share of the weight gradients
// in its own copy of the flat gradient vector and one kernel sums the copies in a fixed order (below a cap
// on the copies' size; above it the second generation adds to the flat gradient with float atomics as the first
// generation does, and 128-wide conditioners go back to the first generation's backward kernel)
// rows per workgroup of the fused backward kernels that write gradient copies: 16 for conditioners 68 to 128 wide, in
// steps of 4 (the 16-row kernels of nsa_train_fused3.h), 32 otherwise (second generation)
static inline bool tr_v4_width(long long H) { return H > 64 && H <= 128 && H % 4 == 0; }
static inline long long tr_slab_rows(const long long* plan) { return tr_v4_width(plan[1]) ? 16 : 32; }
static inline int tr_slab_copies(const long long* plan, long long n) {
  return (int)((n + tr_slab_rows(plan) - 1) / tr_slab_rows(plan));
}
// (the cap on the copies' size: NESSAI_AMD_TRAIN_SLAB_MB, default 4096 - read ONCE per process: the workspace a caller
// allocated and the carve of a later call must agree on it)
static long long tr_slab_floats(const long long* plan, long long n) {
  const long long grid = tr_slab_copies(plan, n);
  const long long words = grid * al4(plan[8]);
  static const long long cap = [] {
    const char* e = getenv("NESSAI_AMD_TRAIN_SLAB_MB");
    return (e != nullptr ? atoll(e) : 4096LL) << 18;  // MB -> floats
  }();
  return (plan[11] == NSA_NET_RESNET && plan[1] <= 128 && words <= cap) ? words : 0;
}

extern "C" NSA_API size_t nsa_train_workspace_floats(const long long* plan, long long n) {
  long long d_tr_max, np_max, per_layer;
  tr_sizes(plan, n, d_tr_max, np_max, per_layer);
  const long long D = plan[0], H = plan[1], NL = plan[3];
  return (size_t)(TR_SCAL_WORDS + 2 * al4(n * D) + 2 * al4(n) + al4(n * np_max) + 2 * al4(n * H) + NL * per_layer + al4(plan[8]) +
                  tr_slab_floats(plan, n) + al4(n * plan[13]));
}

// where the caller puts the [n][C] context rows of a conditional flow's batch: floats from the workspace's start
extern "C" NSA_API size_t nsa_train_context_offset(const long long* plan, long long n) {
  return nsa_train_workspace_floats(plan, n) - (size_t)al4(n * plan[13]);
}

static void tr_carve(const long long* plan, long long n, float* ws, TrWs& w) {
  long long d_tr_max, np_max, per_layer, mask_off, bn_off;
  tr_sizes(plan, n, d_tr_max, np_max, per_layer, &mask_off, &bn_off);
  w.mask_off = mask_off;
  w.bn_off = bn_off;
  const long long D = plan[0], H = plan[1];
  float* p = ws;
  w.scal = p; p += TR_SCAL_WORDS;  // [0..63] scalars, [64..127] optional phase timestamps, [128..] sq-norm partials
  w.x = p; p += al4(n * D);
  w.gx = p; p += al4(n * D);
  w.logdet = p; p += al4(n);
  w.gld = p; p += al4(n);
  w.gp = p; p += al4(n * np_max);
  w.gh = p; p += al4(n * H);
  w.gu = p; p += al4(n * H);
  w.layer_base = p; p += plan[3] * per_layer;
  w.wt = p; p += al4(plan[8]);
  w.slab = tr_slab_floats(plan, n) > 0 ? p : nullptr;
  p += tr_slab_floats(plan, n);
  w.ctx = plan[13] > 0 ? p : nullptr;
  w.per_layer = per_layer;
  w.xsave_words = al4(n * d_tr_max);
  w.np_words = al4(n * np_max);
}

static inline dim3 tiles(long long rows, long long cols) {
  return dim3((unsigned)((rows + 31) / 32), (unsigned)((cols + 31) / 32));
}
static inline dim3 blocks_for(long long work) { return dim3((unsigned)((work + 255) / 256)); }

// the per-item spline kernels exist for every bin count from 2 to 16 (the reference passes num_bins through,
// src/nessai/flows/nsf.py:66,103); the fused coupling kernels for the four counts with an exact inference instance
#define TR_SPLINE_CASE(KERNEL, KK, ...) case KK: hipLaunchKernelGGL(KERNEL<KK>, __VA_ARGS__); break;
#define TR_SPLINE_DISPATCH(KERNEL, K, ...)                                                          \
  do {                                                                                              \
    switch (K) {                                                                                    \
      TR_SPLINE_CASE(KERNEL, 2, __VA_ARGS__) TR_SPLINE_CASE(KERNEL, 3, __VA_ARGS__)                 \
      TR_SPLINE_CASE(KERNEL, 4, __VA_ARGS__) TR_SPLINE_CASE(KERNEL, 5, __VA_ARGS__)                 \
      TR_SPLINE_CASE(KERNEL, 6, __VA_ARGS__) TR_SPLINE_CASE(KERNEL, 7, __VA_ARGS__)                 \
      TR_SPLINE_CASE(KERNEL, 8, __VA_ARGS__) TR_SPLINE_CASE(KERNEL, 9, __VA_ARGS__)                 \
      TR_SPLINE_CASE(KERNEL, 10, __VA_ARGS__) TR_SPLINE_CASE(KERNEL, 11, __VA_ARGS__)               \
      TR_SPLINE_CASE(KERNEL, 12, __VA_ARGS__) TR_SPLINE_CASE(KERNEL, 13, __VA_ARGS__)               \
      TR_SPLINE_CASE(KERNEL, 14, __VA_ARGS__) TR_SPLINE_CASE(KERNEL, 15, __VA_ARGS__)               \
      default: hipLaunchKernelGGL(KERNEL<16>, __VA_ARGS__); break;                                  \
    }                                                                                               \
  } while (0)

// fused coupling kernels (nsa_train_fused.h): used when the conditioner fits their LDS tiles;
// NESSAI_AMD_TRAIN_FUSED=0 selects the layer-wise kernels (kept as the cross-check of the tests)
static bool tr_use_fused(const long long* plan) {
  const char* e = getenv("NESSAI_AMD_TRAIN_FUSED");  // read per call: the tests flip it
  const bool enabled = !(e != nullptr && e[0] == '0');
  // (a spline with a bin count other than 4 / 8 / 10 / 16 trains on the layer-wise kernels)
  const bool exact_bins = plan[4] != NSA_T_RQ || plan[5] == 4 || plan[5] == 8 || plan[5] == 10 || plan[5] == 16;
  return enabled && exact_bins && plan[13] == 0 && plan[11] == NSA_NET_RESNET && plan[1] <= 128 &&
         plan[2] <= TRF_MAX_L && !tr_has_dropout(plan) && !tr_has_inner_bn(plan);  // H <= 128: <= 8 gh tiles; conditional flows: the layer-wise kernels
}

static void tr_prepare_pass(const long long* plan, const int* tables_dev, const float* params_dev, float* grads_dev,
                            const float* x_dev, long long n, TrWs& w, hipStream_t s) {
  const long long D = plan[0], n_params = plan[8];
  const bool fused = tr_use_fused(plan);
  const long long work = (fused || grads_dev) && n_params > n * D ? n_params : n * D;
  hipLaunchKernelGGL(tr_prepare, blocks_for(work > 64 ? work : 64), dim3(256), 0, s, n * D, x_dev, w.x,
                     (fused || grads_dev) ? n_params : 0LL, params_dev, tables_dev + plan[10],
                     fused ? w.wt : (float*)nullptr, grads_dev, n, w.logdet, w.scal);
}

// grads[start + i] += sum over the workgroups' copies, in workgroup order (deterministic)
struct TrRanges {
  int count;
  long long start[32];
  int len[32];
};
__global__ void tr_reduce_slab(const float* __restrict__ slab, long long npar, int copies, float* __restrict__ grads,
                               TrRanges rg) {
  const int r = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rg.len[r]) return;
  const float* src = slab + rg.start[r] + i;
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
  int b = 0;
  for (; b + 4 <= copies; b += 4) {
    s0 += src[(long long)b * npar];
    s1 += src[(long long)(b + 1) * npar];
    s2 += src[(long long)(b + 2) * npar];
    s3 += src[(long long)(b + 3) * npar];
  }
  for (; b < copies; ++b) s0 += src[(long long)b * npar];
  grads[rg.start[r] + i] += (s0 + s1) + (s2 + s3);
}

// The optimiser step's version of the above, fused with what follows it: the copies are summed into the flat
// gradient, the sum of squares of the WHOLE flat gradient is left as one partial sum per block for tr_adamw
// (fixed order: deterministic) and the device-side optimiser state ticks - one launch instead of three.
__global__ void __launch_bounds__(256) tr_finish_grads(long long n, float* __restrict__ g, const float* __restrict__ slab,
                                                       long long npar, int copies, TrRanges rg,
                                                       float* __restrict__ partial, float* __restrict__ opt,
                                                       float beta1, float beta2) {
  __shared__ float sm[256];
  float s = 0.0f;
  for (long long i = blockIdx.x * 256 + threadIdx.x; i < n; i += 256LL * gridDim.x) {
    float v = g[i];
    bool mine = false;
    for (int r = 0; r < rg.count; ++r) mine = mine || (i >= rg.start[r] && i < rg.start[r] + rg.len[r]);
    if (mine) {
      const float* src = slab + i;
      float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
      int b = 0;
      for (; b + 4 <= copies; b += 4) {
        s0 += src[(long long)b * npar];
        s1 += src[(long long)(b + 1) * npar];
        s2 += src[(long long)(b + 2) * npar];
        s3 += src[(long long)(b + 3) * npar];
      }
      for (; b < copies; ++b) s0 += src[(long long)b * npar];
      v += (s0 + s1) + (s2 + s3);
      g[i] = v;
    }
    s += v * v;
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
  if (opt != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {  // tr_opt_tick
    const float step = opt[1] + 1.0f;
    opt[1] = step;
    opt[2] = 1.0f - powf(beta1, step);
    opt[3] = sqrtf(1.0f - powf(beta2, step));
  }
}

// second-generation kernels (nsa_train_fused2.h): conditioner at most 64 wide, tiles within the LDS;
// NESSAI_AMD_TRAIN_V2=0 keeps the first generation (the cross-check of the tests)
static bool tr_v2_fits(bool backward, const TrfArgs& a) {
  const char* e = getenv("NESSAI_AMD_TRAIN_V2");
  if (e != nullptr && e[0] == '0') return false;
  if (a.H > 64 || a.D > 128 || a.np > 512) return false;
  if (!backward) return a.d_id <= 64 && trf2_forward_floats(a) * sizeof(float) <= 160 * 1024;
  // backward: identity columns within the hidden width's tiles, <= 2 items per thread
  const int ht = a.H <= 32 ? 1 : 2;
  return a.d_id <= 32 * ht && a.d_tr <= 32 && trf2_backward_floats(a) * sizeof(float) <= 160 * 1024;
}

// forward pass of a run of `count` coupling layers in one launch (m.a.np = the largest np of the run)
template <int K, int HT, int COUNT>
static hipError_t tr_launch_forward2(const Trf2Multi& m, hipStream_t s) {
  const size_t lds = trf2_forward_floats(m.a) * sizeof(float);
  auto kern = trf2_forward<K, HT, COUNT>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  const unsigned grid = (unsigned)((m.a.n + TRF_ROWS - 1) / TRF_ROWS);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(TRF_THREADS), lds, s, m);
  return hipGetLastError();
}
template <int K, int HT>
static hipError_t tr_forward_run_c(const Trf2Multi& m, int count, hipStream_t s) {
  if (count == 1) return tr_launch_forward2<K, HT, 1>(m, s);
  if (count == 2) return tr_launch_forward2<K, HT, 2>(m, s);
  if (count == 4) return tr_launch_forward2<K, HT, 4>(m, s);
  return tr_launch_forward2<K, HT, 6>(m, s);
}
template <int K>
static hipError_t tr_forward_run_k(const Trf2Multi& m, int count, hipStream_t s) {
  return m.a.H <= 32 ? tr_forward_run_c<K, 1>(m, count, s) : tr_forward_run_c<K, 2>(m, count, s);
}
static hipError_t tr_forward_run(int transform, int K, const Trf2Multi& m, int count, hipStream_t s) {
#ifdef NSA_TRAIN_DEV_K8  // development builds (make TRAIN_DEV=1): the 8-bin instances only - a fifth of the compile time
  return tr_forward_run_k<8>(m, count, s);
#else
  if (transform != NSA_T_RQ) return tr_forward_run_k<0>(m, count, s);
  if (K == 4) return tr_forward_run_k<4>(m, count, s);
  if (K == 8) return tr_forward_run_k<8>(m, count, s);
  if (K == 10) return tr_forward_run_k<10>(m, count, s);
  return tr_forward_run_k<16>(m, count, s);
#endif
}

template <int K, int HT>
static hipError_t tr_launch_fused2(bool backward, const TrfArgs& a, hipStream_t s) {
  const size_t lds = trf2_backward_floats(a) * sizeof(float);
  // 16-byte accesses to the parameter tile and to the transposed final-layer weights
  const bool vec = a.np % 4 == 0 && a.H % 4 == 0 && a.L <= 3 && a.off[2 + 4 * a.L] % 4 == 0 && ((uintptr_t)a.p % 16 == 0) &&
                   ((uintptr_t)a.wt % 16 == 0) && ((uintptr_t)a.h % 16 == 0) && ((uintptr_t)a.u % 16 == 0) &&
                   (((long long)a.n * a.H) % 4 == 0);
  auto kern = a.gslab != nullptr ? (vec ? trf2_backward<K, HT, true, true> : trf2_backward<K, HT, true, false>)
                                 : (vec ? trf2_backward<K, HT, false, true> : trf2_backward<K, HT, false, false>);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  const unsigned grid = (unsigned)((a.n + TRF_ROWS - 1) / TRF_ROWS);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(TRF_THREADS), lds, s, a);
  return hipGetLastError();
}

// third-generation kernels (nsa_train_fused3.h): conditioners 68 to 128 wide (multiples of 4) on 16-row workgroups; NESSAI_AMD_TRAIN_V3=0
// keeps the first generation (the cross-check of the tests)
static bool tr_v3_enabled() {
  const char* e = getenv("NESSAI_AMD_TRAIN_V3");
  return !(e != nullptr && e[0] == '0');
}
// parameter rows in 16-byte units (the tile and the transposed final-layer weights through 16-byte accesses); otherwise
// the backward kernel fills the tile element by element and takes the final layer's weights in their natural layout
static bool tr_v4_param_rows_vec(const TrfArgs& a) {
  return a.np % 4 == 0 && a.off[2 + 4 * a.L] % 4 == 0 && ((uintptr_t)a.p % 16 == 0) && ((uintptr_t)a.wt % 16 == 0);
}
static bool tr_v4_fits(bool backward, const TrfArgs& a) {  // 16-row kernels
  if (!tr_v3_enabled()) return false;
  if (!tr_v4_width(a.H) || a.D > 128 || a.np > 768 || a.d_id > 128) return false;
  if (!backward) return trf4_forward_floats(a) * sizeof(float) <= 160 * 1024;
  // activation tiles in 16-byte units, one transform item per thread, <= 7 activation tiles through registers
  return a.d_tr <= 32 && a.L <= 3 && ((uintptr_t)a.h % 16 == 0) && ((uintptr_t)a.u % 16 == 0) &&
         (((long long)a.n * a.H) % 4 == 0) &&
         trf4_backward_floats(a, tr_v4_param_rows_vec(a)) * sizeof(float) <= 160 * 1024;
}

template <int K>
static hipError_t tr_launch_fused(bool backward, const TrfArgs& a, hipStream_t s) {
  if (backward && tr_v2_fits(true, a))
    return a.H <= 32 ? tr_launch_fused2<K, 1>(backward, a, s) : tr_launch_fused2<K, 2>(backward, a, s);
  // (the 16-row backward kernel only with gradient copies: adding to the flat gradient with float atomics from twice
  // as many workgroups it is slower than the first generation - 1.36 against 1.13 ms per step at 1100 rows of config
  // 4's flow, 7.3 against 1.55 at 8000; profiles/r06_train_batch_sizes.jsonl)
  if (tr_v4_fits(backward, a) && (!backward || a.gslab != nullptr)) {
    const size_t lds4 = (backward ? trf4_backward_floats(a, tr_v4_param_rows_vec(a)) : trf4_forward_floats(a)) * sizeof(float);
    const bool pv = tr_v4_param_rows_vec(a);
    auto kern4 = backward ? (pv ? (a.H == 128 ? trf4_backward<K, true, 128, true> : trf4_backward<K, true, 0, true>)
                                : trf4_backward<K, true, 0, false>)
                          : (a.H == 128 ? trf4_forward<K, 128> : trf4_forward<K, 0>);
    hipError_t e4 = hipFuncSetAttribute(reinterpret_cast<const void*>(kern4),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
    if (e4 != hipSuccess) return e4;
    hipLaunchKernelGGL(kern4, dim3((unsigned)((a.n + TRF4_ROWS - 1) / TRF4_ROWS)), dim3(TRF_THREADS), lds4, s, a);
    return hipGetLastError();
  }
  const size_t floats = backward ? (size_t)(4 * TRF_ROWS * a.ld + TRF_ROWS * (a.ch + 1))
                                 : (size_t)(2 * TRF_ROWS * a.ld);
  const size_t lds = floats * sizeof(float);
  auto kern = backward ? trf_backward<K> : trf_forward<K>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  const unsigned grid = (unsigned)((a.n + TRF_ROWS - 1) / TRF_ROWS);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(TRF_THREADS), lds, s, a);
  return hipGetLastError();
}

static hipError_t tr_fused_coupling(bool backward, int transform, int K, const TrfArgs& a, hipStream_t s) {
#ifdef NSA_TRAIN_DEV_K8
  return tr_launch_fused<8>(backward, a, s);
#else
  if (transform != NSA_T_RQ) return tr_launch_fused<0>(backward, a, s);
  if (K == 4) return tr_launch_fused<4>(backward, a, s);
  if (K == 8) return tr_launch_fused<8>(backward, a, s);
  if (K == 10) return tr_launch_fused<10>(backward, a, s);
  return tr_launch_fused<16>(backward, a, s);
#endif
}

// linear-layer block of the workspace
struct TrLin {
  float *xin, *t, *gt, *Lm, *Um, *gL, *gU, *vec;  // vec: 8*D floats (mean, var, T1, T2, zero bias, ...)
};
static inline TrLin tr_lin(float* lay, long long n, long long D) {
  TrLin q;
  q.xin = lay; q.t = q.xin + n * D; q.gt = q.t + n * D;
  q.Lm = q.gt + n * D; q.Um = q.Lm + D * D; q.gL = q.Um + D * D; q.gU = q.gL + D * D;
  q.vec = q.gU + D * D;
  return q;
}

// forward of the whole flow over the batch in w.x; keeps what the backward pass needs.
// training != 0: batch-norm uses batch statistics and updates the running ones.
static void tr_forward(const long long* plan, const int* tabs, float* params, const float* consts,
                       long long n, TrWs& w, int training, hipStream_t s) {
  const int D = (int)plan[0], H = (int)plan[1], L = (int)plan[2], NL = (int)plan[3];
  const int transform = (int)plan[4], K = (int)plan[5], act = (int)plan[6], npf = (int)plan[7];
  const long long rec = plan[9];
  TrConsts c = {consts[0], consts[1], consts[2], consts[3], consts[4], consts[5], consts[8], consts[9]};
  // dropout of the residual blocks: training-mode passes only (header words [14], [15])
  const bool drop = training != 0 && tr_has_dropout(plan);
  const float drop_p = tr_bits(plan[14]);
  const double thr_d = (double)drop_p * 4294967296.0 + 0.5;
  const uint32_t drop_thr = thr_d >= 4294967295.0 ? 0xffffffffu : (uint32_t)thr_d;
  const float drop_scale = drop_p < 1.0f ? 1.0f / (1.0f - drop_p) : 0.0f;
  const long long* drop_state = (const long long*)(uintptr_t)plan[15];
  int coupling_index = 0;
  // (w.logdet and the scalar block were zeroed by tr_prepare)
  for (int b = 0; b < NL; ++b) {
    const long long* r = plan + TR_HDR + b * rec;
    float* lay = w.layer_base + b * w.per_layer;
    if (r[0] == TR_COUPLING) {
      const int d_id = (int)r[1], d_tr = (int)r[2];
      const int* id_cols = tabs + r[3];
      const int* tr_cols = tabs + r[4];
      const long long* off = r + 5;
      const uint32_t drop_tag = 0x44000000u | ((uint32_t)coupling_index << 8);
      ++coupling_index;
      float* masks = lay + w.mask_off;  // [L][al4(n * H)] (dropout only)
      // batch norm within the blocks: a_l = BN0(h_l) feeds the first linear layer, v_l = BN1(u_l) the second
      const bool bnw = tr_has_inner_bn(plan);
      const float bn_eps = tr_bits(plan[17]);
      float* bn_a = lay + w.bn_off;
      float* bn_v = bn_a + (long long)L * al4(n * H);
      float* bn_st = bn_v + (long long)L * al4(n * H);  // per block: mean0, invstd0, mean1, invstd1 (al4(H) each)
      const long long* bn_par = off + 4 + 4 * L + (plan[13] > 0 ? 2 * L : 0);  // per block: {g, b, rm, rv} x 2
      auto inner_bn = [&](int l, int which, const float* X, float* Y) {
        const long long* q = bn_par + 8 * l + 4 * which;
        float* st = bn_st + ((long long)l * 4 + 2 * which) * al4(H);
        hipLaunchKernelGGL(tr_bn1d_forward, dim3(H), dim3(256), 0, s, (int)n, H, X, Y, params + q[0], params + q[1],
                           params + q[2], params + q[3], bn_eps, 0.1f, training, st, st + al4(H));
      };
      // layer block: xsave[n*d_tr_max] | h[L+1][n*H] | u[L][n*H] | p[n*np_max]
      float* xsave = lay;
      float* h = xsave + w.xsave_words;
      float* u = h + (long long)(L + 1) * n * H;
      float* p = u + (long long)L * n * H;
      const int np = d_tr * npf;
      if (tr_use_fused(plan)) {
        TrfArgs a;
        memset(&a, 0, sizeof(a));
        a.n = (int)n; a.D = D; a.H = H; a.L = L; a.d_id = d_id; a.d_tr = d_tr; a.np = np; a.act = act;
        a.transform = transform;
        a.ld = ((H > d_id ? H : d_id) + 1) | 1;
        a.wt = w.wt;
        a.dbg = getenv("NESSAI_AMD_TRAIN_TIMING") ? w.scal + 64 : nullptr;
        a.id_cols = id_cols; a.tr_cols = tr_cols; a.params = params; a.X = w.x; a.xsave = xsave;
        a.h = h; a.u = u; a.p = p; a.logdet = w.logdet; a.c = c;
        for (int i = 0; i < 4 + 4 * L; ++i) a.off[i] = off[i];
        if (tr_v2_fits(false, a) && plan[8] < (1LL << 31)) {
          // second generation: this coupling and up to TRF2_MAXC - 1 couplings that follow it directly, in one
          // launch; how many (NESSAI_AMD_TRAIN_RUN, default all three) balances the launches saved against the length
          // of the tail a 32-workgroup grid leaves idle
          const char* e = getenv("NESSAI_AMD_TRAIN_RUN");
          int want = e != nullptr ? atoi(e) : TRF2_MAXC;
          want = want < 1 ? 1 : (want > TRF2_MAXC ? TRF2_MAXC : want);
          {  // instantiated run lengths: 6 (config 3), 4 (the reference's default n_blocks), 2, 1
            int ahead = 0;
            for (int bb = b; bb < NL && ahead < want && plan[TR_HDR + bb * rec] == TR_COUPLING; ++bb) ++ahead;
            want = ahead >= 6 ? 6 : (ahead >= 4 ? 4 : (ahead >= 2 ? 2 : 1));
          }
          Trf2Multi m;
          memset(&m, 0, sizeof(m));
          m.a = a;
          int np_max = 0, count = 0, bb = b;
          for (; bb < NL && count < want; ++bb) {
            const long long* rr = plan + TR_HDR + bb * rec;
            if (rr[0] != TR_COUPLING) break;
            TrfArgs t = a;
            t.d_id = (int)rr[1]; t.d_tr = (int)rr[2]; t.np = t.d_tr * npf;
            t.np = t.np > np_max ? t.np : np_max;  // the parameter tile holds the largest np of the run
            if (!tr_v2_fits(false, t) || (((H > t.d_id ? H : t.d_id) + 1) | 1) > a.ld) break;
            float* layb = w.layer_base + bb * w.per_layer;
            Trf2One& o = m.one[count++];
            o.d_id = (int)rr[1]; o.d_tr = (int)rr[2]; o.np = o.d_tr * npf;
            if (o.np > np_max) np_max = o.np;
            o.id_cols = tabs + rr[3]; o.tr_cols = tabs + rr[4];
            o.xsave = layb;
            o.h = o.xsave + w.xsave_words;
            o.u = o.h + (long long)(L + 1) * n * H;
            o.p = o.u + (long long)L * n * H;
            for (int i = 0; i < 4 + 4 * L; ++i) o.off[i] = (int)rr[5 + i];
          }
          // The loop can stop early (a coupling that does not fit the second generation, or whose rows are
          // wider than the run's tile): only run lengths 6 / 4 / 2 / 1 are instantiated, so a run of 3 or 5
          // is cut back to the next instantiated length - launching the 6-coupling kernel on it would walk
          // the zeroed records behind it - and the couplings cut off start the next run.
          if (count >= 1) {
            const int keep = count >= 6 ? 6 : (count >= 4 ? 4 : (count >= 2 ? 2 : 1));
            if (keep != count) {
              count = keep;
              bb = b + count;
              np_max = 0;
              for (int i = 0; i < count; ++i) np_max = m.one[i].np > np_max ? m.one[i].np : np_max;
              for (int i = count; i < TRF2_MAXC; ++i) memset(&m.one[i], 0, sizeof(m.one[i]));
            }
            m.a.np = np_max;
            (void)tr_forward_run(transform, K, m, count, s);
            b = bb - 1;
            continue;
          }
        }
        (void)tr_fused_coupling(false, transform, K, a, s);
        continue;
      }
      const int C = (int)plan[13];
      if (C > 0) {
        // conditional residual net: h0 = W0 cat(x_id, ctx) + b0;  h_{l+1} = h_l + (W2 act(u_l) + b2) * sigmoid(Wc ctx + bc)
        float* tb = p + w.np_words;                       // t[L][n*H]: W2 act(u) + b2, then its gradient
        float* gb = tb + (long long)L * al4(n * H);       // g[L][n*H]: Wc ctx + bc, then its gradient
        float* xc = gb + (long long)L * al4(n * H);       // [n][d_id + C]
        const int W = d_id + C;
        hipLaunchKernelGGL(tr_concat_ctx, blocks_for(n * W), dim3(256), 0, s, (int)n, d_id, C, w.x, D, id_cols, w.ctx, xc);
        hipLaunchKernelGGL(tr_gemm_nt, tiles(n, H), dim3(64), 0, s, (int)n, W, H, xc, W, (const int*)nullptr, ACT_NONE,
                           params + off[0], params + off[1], (const float*)nullptr, h, H);
        for (int l = 0; l < L; ++l) {
          float* hl = h + (long long)l * n * H;
          float* ul = u + (long long)l * n * H;
          float* tl = tb + (long long)l * al4(n * H);
          float* gl = gb + (long long)l * al4(n * H);
          const float* in0 = hl;
          if (bnw) { inner_bn(l, 0, hl, bn_a + (long long)l * al4(n * H)); in0 = bn_a + (long long)l * al4(n * H); }
          hipLaunchKernelGGL(tr_gemm_nt, tiles(n, H), dim3(64), 0, s, (int)n, H, H, in0, H, (const int*)nullptr,
                             act, params + off[2 + 4 * l], params + off[3 + 4 * l], (const float*)nullptr, ul, H);
          const float* in1 = ul;
          if (bnw) { inner_bn(l, 1, ul, bn_v + (long long)l * al4(n * H)); in1 = bn_v + (long long)l * al4(n * H); }
          float* ml = drop ? masks + (long long)l * al4(n * H) : nullptr;
          if (drop)
            hipLaunchKernelGGL(tr_dropout_mask, blocks_for(n * ((H + 3) / 4)), dim3(256), 0, s, n, H, drop_state,
                               drop_tag | (uint32_t)l, drop_thr, drop_scale, ml);
          hipLaunchKernelGGL(tr_gemm_nt, tiles(n, H), dim3(64), 0, s, (int)n, H, H, in1, H, (const int*)nullptr,
                             act, params + off[4 + 4 * l], params + off[5 + 4 * l], (const float*)nullptr, tl, H,
                             (const float*)ml);
          hipLaunchKernelGGL(tr_gemm_nt, tiles(n, H), dim3(64), 0, s, (int)n, C, H, w.ctx, C, (const int*)nullptr,
                             ACT_NONE, params + off[4 + 4 * L + 2 * l], params + off[5 + 4 * L + 2 * l],
                             (const float*)nullptr, gl, H);
          hipLaunchKernelGGL(tr_gate_forward, blocks_for(n * H), dim3(256), 0, s, n * H, hl, tl, gl, hl + (long long)n * H);
        }
        float* hL = h + (long long)L * n * H;
        hipLaunchKernelGGL(tr_gemm_nt, tiles(n, np), dim3(64), 0, s, (int)n, H, np, hL, H, (const int*)nullptr,
                           ACT_NONE, params + off[2 + 4 * L], params + off[3 + 4 * L], (const float*)nullptr, p, np);
      } else {
      hipLaunchKernelGGL(tr_gemm_nt, tiles(n, H), dim3(64), 0, s, (int)n, d_id, H, w.x, D, id_cols, ACT_NONE,
                         params + off[0], params + off[1], (const float*)nullptr, h, H);
      }
      if (C > 0) {
      } else if (plan[11] == NSA_NET_MLP) {
        // z_0 = W_in x_id + b_in;  z_{l+1} = W_h act(z_l) + b_h;  parameters = W_out act(z_L) + b_out
        for (int l = 0; l < L; ++l)
          hipLaunchKernelGGL(tr_gemm_nt, tiles(n, H), dim3(64), 0, s, (int)n, H, H, h + (long long)l * n * H, H,
                             (const int*)nullptr, act, params + off[2 + 2 * l], params + off[3 + 2 * l],
                             (const float*)nullptr, h + (long long)(l + 1) * n * H, H);
        hipLaunchKernelGGL(tr_gemm_nt, tiles(n, np), dim3(64), 0, s, (int)n, H, np, h + (long long)L * n * H, H,
                           (const int*)nullptr, act, params + off[2 + 2 * L], params + off[3 + 2 * L],
                           (const float*)nullptr, p, np);
      } else {
      for (int l = 0; l < L; ++l) {
        float* hl = h + (long long)l * n * H;
        float* ul = u + (long long)l * n * H;
        const float* in0 = hl;
        if (bnw) { inner_bn(l, 0, hl, bn_a + (long long)l * al4(n * H)); in0 = bn_a + (long long)l * al4(n * H); }
        hipLaunchKernelGGL(tr_gemm_nt, tiles(n, H), dim3(64), 0, s, (int)n, H, H, in0, H, (const int*)nullptr,
                           act, params + off[2 + 4 * l], params + off[3 + 4 * l], (const float*)nullptr, ul, H);
        const float* in1 = ul;
        if (bnw) { inner_bn(l, 1, ul, bn_v + (long long)l * al4(n * H)); in1 = bn_v + (long long)l * al4(n * H); }
        float* ml = drop ? masks + (long long)l * al4(n * H) : nullptr;
        if (drop)
          hipLaunchKernelGGL(tr_dropout_mask, blocks_for(n * ((H + 3) / 4)), dim3(256), 0, s, n, H, drop_state,
                             drop_tag | (uint32_t)l, drop_thr, drop_scale, ml);
        hipLaunchKernelGGL(tr_gemm_nt, tiles(n, H), dim3(64), 0, s, (int)n, H, H, in1, H, (const int*)nullptr,
                           act, params + off[4 + 4 * l], params + off[5 + 4 * l], hl, hl + (long long)n * H, H,
                           (const float*)ml);
      }
      float* hL = h + (long long)L * n * H;
      hipLaunchKernelGGL(tr_gemm_nt, tiles(n, np), dim3(64), 0, s, (int)n, H, np, hL, H, (const int*)nullptr,
                         ACT_NONE, params + off[2 + 4 * L], params + off[3 + 4 * L], (const float*)nullptr, p,
                         np);
      }
      if (transform == NSA_T_RQ) {
        TR_SPLINE_DISPATCH(tr_rq_forward, K, blocks_for(n * d_tr), dim3(256), 0, s, (int)n, d_tr, tr_cols, p,
                           w.x, D, xsave, w.logdet, c);
      } else {
        hipLaunchKernelGGL(tr_affine_forward, blocks_for(n * d_tr), dim3(256), 0, s, (int)n, d_tr,
                           transform == NSA_T_ADDITIVE ? 1 : 0, tr_cols, p, w.x, D, xsave, w.logdet, c.aff_a, c.aff_b);
      }
    } else if (r[0] == TR_BATCHNORM) {
      const int* cols = tabs + r[1];
      const float eps = tr_bits(r[6]), momentum = tr_bits(r[7]);
      TrLin q = tr_lin(lay, n, D);
      float* mean = q.vec;
      float* var = q.vec + D;
      if (training && tr_use_fused(plan)) {
        hipLaunchKernelGGL(tr_bn_forward_fused, dim3(D), dim3(256), 0, s, (int)n, D, w.x, D, cols, params + r[2],
                           params + r[3], eps, mean, var, q.xin, params + r[4], params + r[5], momentum, 1, w.scal + 1);
      } else if (training) {
        hipLaunchKernelGGL(tr_bn_stats, dim3(D), dim3(256), 0, s, (int)n, w.x, D, cols, mean, var);
        hipLaunchKernelGGL(tr_bn_forward, blocks_for(n * D), dim3(256), 0, s, (int)n, D, w.x, D, cols,
                           params + r[2], params + r[3], mean, var, eps, q.xin, params + r[4], params + r[5],
                           momentum, 1, w.scal + 1);
      } else {
        hipLaunchKernelGGL(tr_bn_eval, blocks_for(n * D), dim3(256), 0, s, (int)n, D, w.x, D, cols,
                           params + r[2], params + r[3], params + r[4], params + r[5], eps, q.xin, w.scal + 1);
      }
    } else if (r[0] == TR_ACTNORM) {
      const int* cols = tabs + r[1];
      TrLin q = tr_lin(lay, n, D);
      if (training && r[4] != 0)
        hipLaunchKernelGGL(tr_actnorm_init, dim3(D), dim3(256), 0, s, (int)n, w.x, D, cols, params + r[2],
                           params + r[3]);
      hipLaunchKernelGGL(tr_actnorm_forward, blocks_for(n * D), dim3(256), 0, s, (int)n, D, w.x, D, cols,
                         params + r[2], params + r[3], q.xin, w.scal + 1);
    } else if (r[0] == TR_LU) {  // y = (x U^T) L^T + b, outputs in logical order
      const int* cols = tabs + r[1];
      const float eps = tr_bits(r[6]);
      TrLin q = tr_lin(lay, n, D);
      if (tr_use_fused(plan) && D <= 64) {
        const int R = 256 / D;
        const long long groups = (n + R - 1) / R;
        hipLaunchKernelGGL(tr_lu_forward_fused, dim3((unsigned)(groups < 256 ? groups : 256)), dim3(256),
                           (size_t)(2 * D * D + 512) * sizeof(float), s, (int)n, D, w.x, cols, params + r[2],
                           params + r[3], params + r[4], eps, params + r[5], q.xin, q.t, q.gL, q.gU, w.scal + 1,
                           (unsigned*)(q.vec + 4 * D));
        continue;
      }
      float* zero = q.vec;  // zero bias for the first product
      (void)hipMemsetAsync(zero, 0, D * sizeof(float), s);
      hipLaunchKernelGGL(tr_lu_build, blocks_for(D * D), dim3(256), 0, s, D, params + r[2], params + r[3],
                         params + r[4], eps, q.Lm, q.Um, w.scal + 1);
      hipLaunchKernelGGL(tr_gather_cols, blocks_for(n * D), dim3(256), 0, s, (int)n, D, w.x, D, cols, q.xin, 0);
      hipLaunchKernelGGL(tr_gemm_nt, tiles(n, D), dim3(64), 0, s, (int)n, D, D, q.xin, D, (const int*)nullptr,
                         ACT_NONE, q.Um, zero, (const float*)nullptr, q.t, D);
      hipLaunchKernelGGL(tr_gemm_nt, tiles(n, D), dim3(64), 0, s, (int)n, D, D, q.t, D, (const int*)nullptr,
                         ACT_NONE, q.Lm, params + r[5], (const float*)nullptr, w.x, D);
    } else if (r[0] == TR_SVD) {
      const int* cols = tabs + r[1];
      const int nh = (int)r[7];
      TrLin q = tr_lin(lay, n, D);
      const long long groups = (n + 3) / 4;
      hipLaunchKernelGGL(tr_svd_forward, dim3((unsigned)(groups < 256 ? groups : 256)), dim3(256),
                         (size_t)(2 * nh * D + 2 * nh + D) * sizeof(float), s, (int)n, D, nh, w.x, cols, params + r[2],
                         params + r[3], params + r[4], params + r[5], tr_bits(r[6]), q.xin, w.scal + 1);
    }
  }
  // the next training-mode pass draws new masks (on the device: the step survives a HIP-graph replay)
  if (drop) hipLaunchKernelGGL(tr_dropout_tick, dim3(1), dim3(1), 0, s, (long long*)(uintptr_t)plan[15]);
}

// training != 0: the forward pass used batch statistics (batch norm)
// ranges_out: the caller sums the per-workgroup gradient copies itself (tr_finish_grads); null: done here
static void tr_backward(const long long* plan, const int* tabs, const float* params, float* grads,
                        const float* consts, long long n, TrWs& w, int training, hipStream_t s,
                        TrRanges* ranges_out = nullptr) {
  const int D = (int)plan[0], H = (int)plan[1], L = (int)plan[2], NL = (int)plan[3];
  const int transform = (int)plan[4], K = (int)plan[5], act = (int)plan[6], npf = (int)plan[7];
  const long long rec = plan[9];
  TrConsts c = {consts[0], consts[1], consts[2], consts[3], consts[4], consts[5], consts[8], consts[9]};
  TrRanges ranges;
  ranges.count = 0;
  const bool drop = training != 0 && tr_has_dropout(plan);  // the masks of the forward pass are in the workspace
  for (int b = NL - 1; b >= 0; --b) {
    const long long* r = plan + TR_HDR + b * rec;
    float* lay = w.layer_base + b * w.per_layer;
    if (r[0] == TR_COUPLING) {
      const int d_id = (int)r[1], d_tr = (int)r[2];
      const int* id_cols = tabs + r[3];
      const int* tr_cols = tabs + r[4];
      const long long* off = r + 5;
      const float* masks = lay + w.mask_off;
      const bool bnw = tr_has_inner_bn(plan);
      float* bn_a = lay + w.bn_off;
      float* bn_v = bn_a + (long long)L * al4(n * H);
      float* bn_st = bn_v + (long long)L * al4(n * H);
      const long long* bn_par = off + 4 + 4 * L + (plan[13] > 0 ? 2 * L : 0);
      // d loss / d (input of batch norm `which` of block l) from Gy = d loss / d (its output); the gradients of its
      // gamma / beta go to the flat gradient
      auto inner_bn_back = [&](int l, int which, const float* Gy, const float* X, float* Gx, int accumulate) {
        const long long* q = bn_par + 8 * l + 4 * which;
        const float* st = bn_st + ((long long)l * 4 + 2 * which) * al4(H);
        hipLaunchKernelGGL(tr_bn1d_backward, dim3(H), dim3(256), 0, s, (int)n, H, Gy, X, params + q[0], st, st + al4(H),
                           training, Gx, accumulate, grads + q[0], grads + q[1]);
      };
      float* xsave = lay;
      float* h = xsave + w.xsave_words;
      float* u = h + (long long)(L + 1) * n * H;
      float* p = u + (long long)L * n * H;
      const int np = d_tr * npf;
      if (tr_use_fused(plan)) {
        TrfArgs a;
        memset(&a, 0, sizeof(a));
        a.n = (int)n; a.D = D; a.H = H; a.L = L; a.d_id = d_id; a.d_tr = d_tr; a.np = np; a.act = act;
        a.transform = transform;
        a.ld = ((H > d_id ? H : d_id) + 1) | 1;
        {
          // chunk of the final layer's backward pass (first generation): as few chunks as fit next to the four
          // activation / gradient tiles in 150 KB of LDS, equal in size, a multiple of 32 parameters - every chunk
          // is a staging round trip and two workgroup barriers (config 4's 736 parameters: two chunks, twelve before)
          const int tiles = 4 * TRF_ROWS * a.ld;
          int ch_max = (((150 * 1024) / 4 - tiles) / TRF_ROWS - 1) & ~31;
          ch_max = ch_max < 64 ? 64 : ch_max;
          const int chunks = (np + ch_max - 1) / ch_max;
          a.ch = (((np + chunks - 1) / chunks) + 31) & ~31;
          a.ch = a.ch < 64 ? 64 : a.ch;
        }
        a.dbg = getenv("NESSAI_AMD_TRAIN_TIMING") ? w.scal + 64 : nullptr;
        a.id_cols = id_cols; a.tr_cols = tr_cols; a.params = params; a.grads = grads; a.X = w.x; a.Gx = w.gx;
        a.xsave = xsave; a.h = h; a.u = u; a.p = p; a.gp = w.gp; a.gld = w.gld; a.c = c;
        a.wt = w.wt;
        for (int i = 0; i < 4 + 4 * L; ++i) a.off[i] = off[i];
        // per-workgroup gradient copies when this coupling's parameters are one contiguous block of the flat
        // vector (they are: the module registers them in order) and there is room for one more range
        if (w.slab != nullptr && (tr_v2_fits(true, a) || tr_v4_fits(true, a)) && ranges.count < 32) {
          long long expect = off[0];
          bool contiguous = true;
          for (int i = 0; i < 4 + 4 * L; ++i) {
            contiguous = contiguous && off[i] == expect;
            const bool bias = (i & 1) != 0;
            const long long rows = (i < 2) ? H : (i >= 2 + 4 * L ? np : H);
            const long long cols = (i < 2) ? d_id : H;
            expect += bias ? rows : rows * cols;
          }
          if (contiguous && expect - off[0] < (1LL << 31)) {
            a.gslab = w.slab;
            a.npar = al4(plan[8]);
            ranges.start[ranges.count] = off[0];
            ranges.len[ranges.count] = (int)(expect - off[0]);
            ++ranges.count;
          }
        }
        (void)tr_fused_coupling(true, transform, K, a, s);
        continue;
      }
      if (transform == NSA_T_RQ) {
        TR_SPLINE_DISPATCH(tr_rq_backward, K, blocks_for(n * d_tr), dim3(256), 0, s, (int)n, d_tr, tr_cols, p,
                           xsave, w.x, D, w.gx, w.gld, w.gp, c);
      } else {
        hipLaunchKernelGGL(tr_affine_backward, blocks_for(n * d_tr), dim3(256), 0, s, (int)n, d_tr,
                           transform == NSA_T_ADDITIVE ? 1 : 0, tr_cols, p, xsave, w.x, D, w.gx, w.gld, w.gp, c.aff_a,
                           c.aff_b);
      }
      if (plan[11] == NSA_NET_MLP) {
        float* zL = h + (long long)L * n * H;
        hipLaunchKernelGGL(tr_gemm_tn, tiles(np, H), dim3(256), 0, s, (int)n, np, H, w.gp, np, zL, H,
                           (const int*)nullptr, act, grads + off[2 + 2 * L], grads + off[3 + 2 * L]);
        hipLaunchKernelGGL(tr_gemm_nn, tiles(n, H), dim3(64), 0, s, (int)n, np, H, w.gp, np,
                           params + off[2 + 2 * L], zL, H, act, w.gh, H, (const int*)nullptr, 0);
        float* gcur = w.gh;
        float* gnext = w.gu;
        for (int l = L - 1; l >= 0; --l) {  // z_{l+1} = W_h act(z_l) + b_h
          float* zl = h + (long long)l * n * H;
          hipLaunchKernelGGL(tr_gemm_tn, tiles(H, H), dim3(256), 0, s, (int)n, H, H, gcur, H, zl, H,
                             (const int*)nullptr, act, grads + off[2 + 2 * l], grads + off[3 + 2 * l]);
          hipLaunchKernelGGL(tr_gemm_nn, tiles(n, H), dim3(64), 0, s, (int)n, H, H, gcur, H,
                             params + off[2 + 2 * l], zl, H, act, gnext, H, (const int*)nullptr, 0);
          float* t_ = gcur; gcur = gnext; gnext = t_;
        }
        hipLaunchKernelGGL(tr_gemm_tn, tiles(H, d_id), dim3(256), 0, s, (int)n, H, d_id, gcur, H, w.x, D, id_cols,
                           ACT_NONE, grads + off[0], grads + off[1]);
        hipLaunchKernelGGL(tr_gemm_nn, tiles(n, d_id), dim3(64), 0, s, (int)n, H, d_id, gcur, H, params + off[0],
                           (const float*)nullptr, 0, ACT_NONE, w.gx, D, id_cols, 2);
        continue;
      }
      // final layer: dWf, dbf, gh = gp . Wf
      float* hL = h + (long long)L * n * H;
      hipLaunchKernelGGL(tr_gemm_tn, tiles(np, H), dim3(256), 0, s, (int)n, np, H, w.gp, np, hL, H,
                         (const int*)nullptr, ACT_NONE, grads + off[2 + 4 * L], grads + off[3 + 4 * L]);
      hipLaunchKernelGGL(tr_gemm_nn, tiles(n, H), dim3(64), 0, s, (int)n, np, H, w.gp, np,
                         params + off[2 + 4 * L], (const float*)nullptr, 0, ACT_NONE, w.gh, H,
                         (const int*)nullptr, 0);
      const int C = (int)plan[13];
      if (C > 0) {  // conditional residual net (see tr_forward)
        float* tb = p + w.np_words;
        float* gb = tb + (long long)L * al4(n * H);
        float* xc = gb + (long long)L * al4(n * H);
        const int W = d_id + C;
        float* gxc = xc + al4(n * (D + C));
        for (int l = L - 1; l >= 0; --l) {
          float* hl = h + (long long)l * n * H;
          float* ul = u + (long long)l * n * H;
          float* tl = tb + (long long)l * al4(n * H);
          float* gl = gb + (long long)l * al4(n * H);
          // the gate: tl <- dL/dt, gl <- dL/dg (w.gh = dL/dh_{l+1}, which is also the gradient of the skip path)
          hipLaunchKernelGGL(tr_gate_backward, blocks_for(n * H), dim3(256), 0, s, n * H, w.gh, tl, gl);
          hipLaunchKernelGGL(tr_gemm_tn, tiles(H, C), dim3(256), 0, s, (int)n, H, C, gl, H, w.ctx, C,
                             (const int*)nullptr, ACT_NONE, grads + off[4 + 4 * L + 2 * l], grads + off[5 + 4 * L + 2 * l]);
          const float* ml = drop ? masks + (long long)l * al4(n * H) : nullptr;
          float* al_ = bn_a + (long long)l * al4(n * H);
          float* vl_ = bn_v + (long long)l * al4(n * H);
          const float* in1 = bnw ? vl_ : ul;
          const float* in0 = bnw ? al_ : hl;
          hipLaunchKernelGGL(tr_gemm_tn, tiles(H, H), dim3(256), 0, s, (int)n, H, H, tl, H, in1, H,
                             (const int*)nullptr, act, grads + off[4 + 4 * l], grads + off[5 + 4 * l], ml);
          hipLaunchKernelGGL(tr_gemm_nn, tiles(n, H), dim3(64), 0, s, (int)n, H, H, tl, H,
                             params + off[4 + 4 * l], in1, H, act, w.gu, H, (const int*)nullptr, 0, ml);
          if (bnw) inner_bn_back(l, 1, w.gu, ul, w.gu, 0);
          hipLaunchKernelGGL(tr_gemm_tn, tiles(H, H), dim3(256), 0, s, (int)n, H, H, w.gu, H, in0, H,
                             (const int*)nullptr, act, grads + off[2 + 4 * l], grads + off[3 + 4 * l]);
          if (bnw) {  // (v_l is free now: it holds d loss / d a_l on its way through the first batch norm)
            hipLaunchKernelGGL(tr_gemm_nn, tiles(n, H), dim3(64), 0, s, (int)n, H, H, w.gu, H,
                               params + off[2 + 4 * l], in0, H, act, vl_, H, (const int*)nullptr, 0);
            inner_bn_back(l, 0, vl_, hl, w.gh, 1);
          } else {
          hipLaunchKernelGGL(tr_gemm_nn, tiles(n, H), dim3(64), 0, s, (int)n, H, H, w.gu, H,
                             params + off[2 + 4 * l], hl, H, act, w.gh, H, (const int*)nullptr, 1);
          }
        }
        hipLaunchKernelGGL(tr_gemm_tn, tiles(H, W), dim3(256), 0, s, (int)n, H, W, w.gh, H, xc, W, (const int*)nullptr,
                           ACT_NONE, grads + off[0], grads + off[1]);
        hipLaunchKernelGGL(tr_gemm_nn, tiles(n, W), dim3(64), 0, s, (int)n, H, W, w.gh, H, params + off[0],
                           (const float*)nullptr, 0, ACT_NONE, gxc, W, (const int*)nullptr, 0);
        hipLaunchKernelGGL(tr_scatter_add, blocks_for(n * d_id), dim3(256), 0, s, (int)n, d_id, gxc, W, w.gx, D, id_cols);
        continue;
      }
      for (int l = L - 1; l >= 0; --l) {
        float* hl = h + (long long)l * n * H;
        float* ul = u + (long long)l * n * H;
        // second linear of the block: h_{l+1} = h_l + W2 act(u_l) + b2
        const float* ml = drop ? masks + (long long)l * al4(n * H) : nullptr;
        float* al_ = bn_a + (long long)l * al4(n * H);
        float* vl_ = bn_v + (long long)l * al4(n * H);
        const float* in1 = bnw ? vl_ : ul;  // (batch norm within: the linear layers saw BN1(u_l) and BN0(h_l))
        const float* in0 = bnw ? al_ : hl;
        hipLaunchKernelGGL(tr_gemm_tn, tiles(H, H), dim3(256), 0, s, (int)n, H, H, w.gh, H, in1, H,
                           (const int*)nullptr, act, grads + off[4 + 4 * l], grads + off[5 + 4 * l], ml);
        hipLaunchKernelGGL(tr_gemm_nn, tiles(n, H), dim3(64), 0, s, (int)n, H, H, w.gh, H,
                           params + off[4 + 4 * l], in1, H, act, w.gu, H, (const int*)nullptr, 0, ml);
        if (bnw) inner_bn_back(l, 1, w.gu, ul, w.gu, 0);
        // first linear: u_l = W1 act(h_l) + b1
        hipLaunchKernelGGL(tr_gemm_tn, tiles(H, H), dim3(256), 0, s, (int)n, H, H, w.gu, H, in0, H,
                           (const int*)nullptr, act, grads + off[2 + 4 * l], grads + off[3 + 4 * l]);
        if (bnw) {  // (v_l is free now: it holds d loss / d a_l on its way through the first batch norm)
          hipLaunchKernelGGL(tr_gemm_nn, tiles(n, H), dim3(64), 0, s, (int)n, H, H, w.gu, H,
                             params + off[2 + 4 * l], in0, H, act, vl_, H, (const int*)nullptr, 0);
          inner_bn_back(l, 0, vl_, hl, w.gh, 1);
        } else {
        hipLaunchKernelGGL(tr_gemm_nn, tiles(n, H), dim3(64), 0, s, (int)n, H, H, w.gu, H,
                           params + off[2 + 4 * l], hl, H, act, w.gh, H, (const int*)nullptr, 1);
        }
      }
      // initial layer: dW0, db0, g_x[:, id_cols] += gh . W0 (x was restored by the transform's backward)
      hipLaunchKernelGGL(tr_gemm_tn, tiles(H, d_id), dim3(256), 0, s, (int)n, H, d_id, w.gh, H, w.x, D, id_cols,
                         ACT_NONE, grads + off[0], grads + off[1]);
      hipLaunchKernelGGL(tr_gemm_nn, tiles(n, d_id), dim3(64), 0, s, (int)n, H, d_id, w.gh, H, params + off[0],
                         (const float*)nullptr, 0, ACT_NONE, w.gx, D, id_cols, 2);
    } else if (r[0] == TR_BATCHNORM) {
      const int* cols = tabs + r[1];
      const float eps = tr_bits(r[6]);
      TrLin q = tr_lin(lay, n, D);
      float *T1 = q.vec + 2 * D, *T2 = q.vec + 3 * D;
      const float* mean = training ? q.vec : params + r[4];  // batch / running statistics
      const float* var = training ? q.vec + D : params + r[5];
      if (tr_use_fused(plan)) {
        hipLaunchKernelGGL(tr_bn_backward_fused, dim3(D), dim3(256), 0, s, (int)n, D, w.x, w.gx, D, cols, q.xin,
                           params + r[2], mean, var, eps, grads + r[2], grads + r[3], w.scal + 2, training ? 1 : 0);
      } else {
        hipLaunchKernelGGL(tr_bn_backward_reduce, dim3(D), dim3(256), 0, s, (int)n, w.gx, D, cols, q.xin, D, mean,
                           var, eps, T1, T2);
        hipLaunchKernelGGL(tr_bn_backward_apply, blocks_for(n * D), dim3(256), 0, s, (int)n, D, w.x, w.gx, D, cols,
                           q.xin, params + r[2], mean, var, eps, T1, T2, grads + r[2], grads + r[3], w.scal + 2,
                           training ? 1 : 0);
      }
    } else if (r[0] == TR_ACTNORM) {
      const int* cols = tabs + r[1];
      TrLin q = tr_lin(lay, n, D);
      hipLaunchKernelGGL(tr_actnorm_backward, dim3(D), dim3(256), 0, s, (int)n, w.x, w.gx, D, cols, q.xin, D,
                         params + r[2], grads + r[2], grads + r[3], w.scal + 2);
    } else if (r[0] == TR_SVD) {
      const int* cols = tabs + r[1];
      const int nh = (int)r[7];
      TrLin q = tr_lin(lay, n, D);
      const long long groups = (n + 3) / 4;
      hipLaunchKernelGGL(tr_svd_backward, dim3((unsigned)(groups < 64 ? groups : 64)), dim3(256),
                         (size_t)(4 * nh * D + 2 * nh + 3 * D) * sizeof(float), s, (int)n, D, nh, w.x, w.gx, cols,
                         params + r[2], params + r[3], params + r[4], tr_bits(r[6]), q.xin, grads + r[2], grads + r[3],
                         grads + r[4], grads + r[5], w.scal + 2);
    } else {  // TR_LU
      const int* cols = tabs + r[1];
      const float eps = tr_bits(r[6]);
      TrLin q = tr_lin(lay, n, D);
      if (tr_use_fused(plan) && D <= 64) {  // (grads were zeroed by tr_prepare / nsa_train_backward: db accumulates)
        const int R = 256 / D;
        const long long groups = (n + R - 1) / R;
        hipLaunchKernelGGL(tr_lu_backward_fused, dim3((unsigned)(groups < 32 ? groups : 32)), dim3(256),
                           (size_t)(2 * D * D + 1024) * sizeof(float), s, (int)n, D, w.x, w.gx, cols, params + r[2],
                           params + r[3], params + r[4], eps, q.xin, q.t, q.gL, q.gU, grads + r[5], grads + r[2],
                           grads + r[3], grads + r[4], w.scal + 2, (unsigned*)(q.vec + 4 * D));
        continue;
      }
      // gy = w.gx (logical order).  dL = gy^T t, db = colsum gy;  gt = gy L;  dU = gt^T x;  gx = gt U
      hipLaunchKernelGGL(tr_gemm_tn, tiles(D, D), dim3(256), 0, s, (int)n, D, D, w.gx, D, q.t, D,
                         (const int*)nullptr, ACT_NONE, q.gL, grads + r[5]);
      hipLaunchKernelGGL(tr_gemm_nn, tiles(n, D), dim3(64), 0, s, (int)n, D, D, w.gx, D, q.Lm,
                         (const float*)nullptr, 0, ACT_NONE, q.gt, D, (const int*)nullptr, 0);
      hipLaunchKernelGGL(tr_gemm_tn, tiles(D, D), dim3(256), 0, s, (int)n, D, D, q.gt, D, q.xin, D,
                         (const int*)nullptr, ACT_NONE, q.gU, q.vec + D);
      hipLaunchKernelGGL(tr_gemm_nn, tiles(n, D), dim3(64), 0, s, (int)n, D, D, q.gt, D, q.Um,
                         (const float*)nullptr, 0, ACT_NONE, w.gx, D, cols, 3);
      hipLaunchKernelGGL(tr_lu_grads, blocks_for(D * D), dim3(256), 0, s, D, q.gL, q.gU, params + r[4], eps,
                         grads + r[2], grads + r[3], grads + r[4], w.scal + 2);
      hipLaunchKernelGGL(tr_gather_cols, blocks_for(n * D), dim3(256), 0, s, (int)n, D, q.xin, D, cols, w.x, 1);
    }
  }
  if (ranges_out != nullptr) {
    *ranges_out = ranges;
  } else if (ranges.count > 0) {
    int longest = 0;
    for (int i = 0; i < ranges.count; ++i) longest = ranges.len[i] > longest ? ranges.len[i] : longest;
    hipLaunchKernelGGL(tr_reduce_slab, dim3((unsigned)((longest + 255) / 256), (unsigned)ranges.count), dim3(256), 0, s,
                       (const float*)w.slab, al4(plan[8]), tr_slab_copies(plan, n), grads, ranges);
  }
}

// One minibatch: forward, loss, backward, clip, AdamW.  loss_acc_dev[0] += loss.
static int tr_step_impl(const long long* plan, const int* tables_dev, const float* consts, float* params_dev,
                        float* grads_dev, float* adam_m_dev, float* adam_v_dev, const float* x_dev,
                        const float* weights_dev, long long n, float* workspace_dev, float lr, float beta1,
                        float beta2, float eps, float weight_decay, float clip_norm, long long step, int optimiser,
                        float* loss_acc_dev, float* opt_dev, void* stream);

extern "C" NSA_API int nsa_train_step(const long long* plan, const int* tables_dev, const float* consts,
                                      float* params_dev, float* grads_dev, float* adam_m_dev,
                                      float* adam_v_dev, const float* x_dev, const float* weights_dev,
                                      long long n, float* workspace_dev, float lr, float beta1, float beta2,
                                      float eps,
                                      float weight_decay, float clip_norm, long long step,
                                      int optimiser, float* loss_acc_dev, void* stream) {
  return tr_step_impl(plan, tables_dev, consts, params_dev, grads_dev, adam_m_dev, adam_v_dev, x_dev, weights_dev, n,
                      workspace_dev, lr, beta1, beta2, eps, weight_decay, clip_norm, step, optimiser, loss_acc_dev,
                      nullptr, stream);
}

// The same minibatch with the learning rate and the step counter on the DEVICE (opt_dev[0] = lr,
// opt_dev[1] = optimiser steps done so far; [2], [3] scratch): nothing in the launch sequence depends
// on a per-step host value, so a whole epoch can be captured once as a HIP graph and replayed.
extern "C" NSA_API int nsa_train_step_dev(const long long* plan, const int* tables_dev, const float* consts,
                                          float* params_dev, float* grads_dev, float* adam_m_dev,
                                          float* adam_v_dev, const float* x_dev, const float* weights_dev,
                                          long long n, float* workspace_dev, float* opt_dev, float beta1,
                                          float beta2, float eps, float weight_decay, float clip_norm,
                                          int optimiser, float* loss_acc_dev, void* stream) {
  if (!opt_dev) return NSA_E_ARG;
  return tr_step_impl(plan, tables_dev, consts, params_dev, grads_dev, adam_m_dev, adam_v_dev, x_dev, weights_dev, n,
                      workspace_dev, 0.0f, beta1, beta2, eps, weight_decay, clip_norm, 1, optimiser, loss_acc_dev,
                      opt_dev, stream);
}

static int tr_step_impl(const long long* plan, const int* tables_dev, const float* consts, float* params_dev,
                        float* grads_dev, float* adam_m_dev, float* adam_v_dev, const float* x_dev,
                        const float* weights_dev, long long n, float* workspace_dev, float lr, float beta1,
                        float beta2, float eps, float weight_decay, float clip_norm, long long step, int optimiser,
                        float* loss_acc_dev, float* opt_dev, void* stream) {
  if (!plan || !tables_dev || !params_dev || !grads_dev || !x_dev || !workspace_dev || !loss_acc_dev || n <= 1 ||
      optimiser < 0 || optimiser > 2)
    return NSA_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  const long long D = plan[0], n_params = plan[8];
  TrWs w;
  tr_carve(plan, n, workspace_dev, w);
  // (the fused kernels accumulate into the flat gradient; the layer-wise ones overwrite it)
  tr_prepare_pass(plan, tables_dev, params_dev, tr_use_fused(plan) ? grads_dev : nullptr, x_dev, n, w, s);
  tr_forward(plan, tables_dev, params_dev, consts, n, w, 1, s);
  hipLaunchKernelGGL(tr_loss, dim3((unsigned)((n + TR_LOSS_ROWS - 1) / TR_LOSS_ROWS)), dim3(256), 0, s, (int)n, (int)D, w.x, (int)D, w.logdet, w.scal + 1,
                     weights_dev, consts[6], consts[7], w.gx, w.gld, loss_acc_dev, (float*)nullptr, 1, w.scal + 2);
  TrRanges ranges;
  tr_backward(plan, tables_dev, params_dev, grads_dev, consts, n, w, 1, s, &ranges);
  const long long want_blocks = (n_params + 255) / 256;
  const int sq_blocks = (int)(want_blocks < TR_SQ_MAX ? want_blocks : TR_SQ_MAX);
  hipLaunchKernelGGL(tr_finish_grads, dim3(sq_blocks), dim3(256), 0, s, n_params, grads_dev, (const float*)w.slab,
                     al4(n_params), tr_slab_copies(plan, n), ranges, w.scal + 128, opt_dev, beta1, beta2);
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.0f - powf(beta2, (float)step));
  hipLaunchKernelGGL(tr_adamw, blocks_for(n_params), dim3(256), 0, s, n_params, params_dev, grads_dev, adam_m_dev,
                     adam_v_dev, w.scal + 128, sq_blocks, clip_norm, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, optimiser,
                     (const float*)opt_dev);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? NSA_OK : NSA_E_HIP;
}

// Forward only, eval mode: loss_acc_dev[0] += -mean(log_prob(x)); optional per-row log_prob.
extern "C" NSA_API int nsa_train_eval(const long long* plan, const int* tables_dev, const float* consts,
                                      const float* params_dev, const float* x_dev, const float* weights_dev,
                                      long long n, float* workspace_dev, float* loss_acc_dev,
                                      float* log_prob_out_dev, void* stream) {
  if (!plan || !tables_dev || !params_dev || !x_dev || !workspace_dev || !loss_acc_dev || n <= 0)
    return NSA_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  const long long D = plan[0];
  TrWs w;
  tr_carve(plan, n, workspace_dev, w);
  tr_prepare_pass(plan, tables_dev, params_dev, nullptr, x_dev, n, w, s);
  tr_forward(plan, tables_dev, const_cast<float*>(params_dev), consts, n, w, 0, s);
  hipLaunchKernelGGL(tr_loss, dim3((unsigned)((n + TR_LOSS_ROWS - 1) / TR_LOSS_ROWS)), dim3(256), 0, s, (int)n, (int)D, w.x, (int)D, w.logdet, w.scal + 1,
                     weights_dev, consts[6], consts[7], w.gx, w.gld, loss_acc_dev, log_prob_out_dev, 0, w.scal + 2);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? NSA_OK : NSA_E_HIP;
}

// ---------------------------------------------------------------------------
// differentiable evaluation for torch autograd (nessai_amd/ops.py): the forward pass keeps its
// activations in the caller's workspace, the backward pass takes arbitrary upstream gradients
// ---------------------------------------------------------------------------

// z (logical column order), logdet and log_prob of the rows after tr_forward
__global__ void tr_outputs(int n, int D, const float* __restrict__ X, const int* __restrict__ out_cols,
                           const float* __restrict__ logdet, const float* __restrict__ ld_const, float inv_var,
                           float log_z, float* __restrict__ z_out, float* __restrict__ logdet_out,
                           float* __restrict__ log_prob_out) {
  const int row = blockIdx.x * (blockDim.x >> 2) + (threadIdx.x >> 2), part = threadIdx.x & 3;
  if (row >= n) return;
  float e = 0.0f;
  for (int i = part; i < D; i += 4) {
    const float z = X[(long long)row * D + out_cols[i]];
    e += z * z;
    if (z_out) z_out[(long long)row * D + i] = z;
  }
  e += __shfl_xor(e, 1);
  e += __shfl_xor(e, 2);
  if (part == 0) {
    const float ld = logdet[row] + ld_const[0];
    if (logdet_out) logdet_out[row] = ld;
    if (log_prob_out) log_prob_out[row] = -0.5f * inv_var * e - log_z + ld;
  }
}

// seeds of the backward pass from the upstream gradients (any of them may be null):
//   Gx[row][out_cols[i]] = g_z[row][i] - g_lp[row] * z / var;   gld[row] = g_ld[row] + g_lp[row]
__global__ void tr_seed(int n, int D, const float* __restrict__ X, const int* __restrict__ out_cols,
                        const float* __restrict__ g_z, const float* __restrict__ g_ld,
                        const float* __restrict__ g_lp, float inv_var, float* __restrict__ Gx,
                        float* __restrict__ gld) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * D) return;
  const int row = t / D, i = t - row * D;
  const int col = out_cols[i];
  float g = g_z ? g_z[t] : 0.0f;
  if (g_lp) g -= g_lp[row] * X[(long long)row * D + col] * inv_var;
  Gx[(long long)row * D + col] = g;
  if (i == 0) gld[row] = (g_ld ? g_ld[row] : 0.0f) + (g_lp ? g_lp[row] : 0.0f);
}

// gl_sum[0] = sum_i gld[i], one workgroup, fixed order (deterministic)
__global__ void __launch_bounds__(256) tr_seed_sum(int n, const float* __restrict__ gld, float* __restrict__ gl_sum) {
  __shared__ float sm[256];
  float a = 0.0f;
  for (int row = threadIdx.x; row < n; row += 256) a += gld[row];
  sm[threadIdx.x] = a;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) gl_sum[0] = sm[0];
}

extern "C" NSA_API int nsa_train_forward(const long long* plan, const int* tables_dev, const float* consts,
                                         float* params_dev, const float* x_dev, long long n,
                                         float* workspace_dev, int training, float* z_out_dev,
                                         float* logdet_out_dev, float* log_prob_out_dev, void* stream) {
  if (!plan || !tables_dev || !consts || !params_dev || !x_dev || !workspace_dev || n <= 0) return NSA_E_ARG;
  if (training && n <= 1) return NSA_E_ARG;  // batch statistics need two rows
  hipStream_t s = (hipStream_t)stream;
  const long long D = plan[0];
  TrWs w;
  tr_carve(plan, n, workspace_dev, w);
  tr_prepare_pass(plan, tables_dev, params_dev, nullptr, x_dev, n, w, s);
  tr_forward(plan, tables_dev, params_dev, consts, n, w, training ? 1 : 0, s);
  hipLaunchKernelGGL(tr_outputs, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, s, (int)n, (int)D, w.x,
                     tables_dev + plan[12], w.logdet, w.scal + 1, consts[6], consts[7], z_out_dev,
                     logdet_out_dev, log_prob_out_dev);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? NSA_OK : NSA_E_HIP;
}

extern "C" NSA_API int nsa_train_backward(const long long* plan, const int* tables_dev, const float* consts,
                                          const float* params_dev, float* grads_dev, long long n,
                                          float* workspace_dev, int training, const float* g_z_dev,
                                          const float* g_logdet_dev, const float* g_log_prob_dev,
                                          float* gx_out_dev, void* stream) {
  if (!plan || !tables_dev || !consts || !params_dev || !grads_dev || !workspace_dev || n <= 0) return NSA_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  const long long D = plan[0], n_params = plan[8];
  TrWs w;
  tr_carve(plan, n, workspace_dev, w);
  (void)hipMemsetAsync(grads_dev, 0, n_params * sizeof(float), s);
  hipLaunchKernelGGL(tr_seed, blocks_for(n * D), dim3(256), 0, s, (int)n, (int)D, w.x, tables_dev + plan[12],
                     g_z_dev, g_logdet_dev, g_log_prob_dev, consts[6], w.gx, w.gld);
  hipLaunchKernelGGL(tr_seed_sum, dim3(1), dim3(256), 0, s, (int)n, w.gld, w.scal + 2);
  tr_backward(plan, tables_dev, params_dev, grads_dev, consts, n, w, training ? 1 : 0, s);
  if (gx_out_dev != nullptr)
    hipLaunchKernelGGL(tr_copy_rows, blocks_for(n * D), dim3(256), 0, s, n * D, w.gx, gx_out_dev);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? NSA_OK : NSA_E_HIP;
}
