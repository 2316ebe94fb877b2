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
__global__ void __launch_bounds__(64) tr_gemm_nt(int n, int I, int O, const float* __restrict__ X, int ldx,
                                                 const int* __restrict__ cols, int act_in,
                                                 const float* __restrict__ W, const float* __restrict__ b,
                                                 const float* __restrict__ R, float* __restrict__ Y, int ldy) {
  const int lane = threadIdx.x, half = lane >> 5, j = lane & 31;
  const int row0 = blockIdx.x * 32, col0 = blockIdx.y * 32;
  const int row = row0 + j, col = col0 + j;
  f32x16 acc = {0};
  for (int k = 0; k < I; k += 2) {
    const int kk = k + half;
    float a = 0.0f, bb = 0.0f;
    if (kk < I) {
      if (row < n) a = tr_act(X[(long long)row * ldx + (cols ? cols[kk] : kk)], act_in);
      if (col < O) bb = W[(long long)col * I + kk];
    }
    // D[i][jj] = sum_k A[i][k] B[k][jj]: i = output column (W row), jj = batch row
    acc = nsa_mfma(bb, a, acc);
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
//   mode 0: Gin = value; mode 1: Gin += value; mode 2: Gx[:, cols[i]] += value (scatter-add)
__global__ void __launch_bounds__(64) tr_gemm_nn(int n, int O, int I, const float* __restrict__ Gout, int ldg,
                                                 const float* __restrict__ W, const float* __restrict__ Z,
                                                 int ldz, int act, float* __restrict__ Gin, int ldi,
                                                 const int* __restrict__ cols, int mode) {
  const int lane = threadIdx.x, half = lane >> 5, j = lane & 31;
  const int row0 = blockIdx.x * 32, col0 = blockIdx.y * 32;
  const int row = row0 + j, col = col0 + j;
  f32x16 acc = {0};
  for (int k = 0; k < O; k += 2) {
    const int kk = k + half;
    float a = 0.0f, bb = 0.0f;
    if (kk < O) {
      if (row < n) a = Gout[(long long)row * ldg + kk];
      if (col < I) bb = W[(long long)kk * I + col];
    }
    acc = nsa_mfma(bb, a, acc);  // D[i = input feature][jj = batch row]
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = col0 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (i < I && row < n) {
      float v = acc[r];
      if (Z) v *= tr_act_grad(Z[(long long)row * ldz + i], act);
      if (mode == 0) Gin[(long long)row * ldi + i] = v;
      else if (mode == 1) Gin[(long long)row * ldi + i] += v;
      else Gin[(long long)row * ldi + cols[i]] += v;
    }
  }
}

// dW[O][I] = sum_rows Gout[row][o] * act_in(X[row][cols[i]]);  db[o] = sum_rows Gout[row][o]
__global__ void __launch_bounds__(64) tr_gemm_tn(int n, int O, int I, const float* __restrict__ Gout, int ldg,
                                                 const float* __restrict__ X, int ldx,
                                                 const int* __restrict__ cols, int act_in,
                                                 float* __restrict__ dW, float* __restrict__ db) {
  const int lane = threadIdx.x, half = lane >> 5, j = lane & 31;
  const int o0 = blockIdx.x * 32, i0 = blockIdx.y * 32;
  const int o = o0 + j, i = i0 + j;
  const int xcol = (i < I) ? (cols ? cols[i] : i) : 0;
  f32x16 acc = {0};
  float bsum = 0.0f;
  for (int k = 0; k < n; k += 2) {
    const int row = k + half;
    float a = 0.0f, bb = 0.0f;
    if (row < n) {
      if (o < O) a = Gout[(long long)row * ldg + o];
      if (i < I) bb = tr_act(X[(long long)row * ldx + xcol], act_in);
    }
    bsum += a;
    acc = nsa_mfma(a, bb, acc);  // D[ii = o][jj = i]
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int oo = o0 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (oo < O && i < I) dW[(long long)oo * I + i] = acc[r];
  }
  if (blockIdx.y == 0) {
    bsum += __shfl_xor(bsum, 32);
    if (half == 0 && o < O) db[o] = bsum;
  }
}

// ---------------------------------------------------------------------------
// element-wise coupling transforms, forward (training) and backward
// ---------------------------------------------------------------------------

struct TrConsts {
  float tail_bound, min_bw, min_bh, min_d, edge_d, inv_div;
};

// one thread per (row, transformed feature).  Forward: saves x, writes y in place,
// atomically accumulates log|det J| per row.
template <int K>
__global__ void tr_rq_forward(int n, int d_tr, const int* __restrict__ tr_cols, const float* __restrict__ P,
                              float* __restrict__ X, int ldx, float* __restrict__ xsave,
                              float* __restrict__ logdet, TrConsts c) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * d_tr) return;
  const int row = t / d_tr, m = t - row * d_tr;
  constexpr int NP = 3 * K - 1;
  float v[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) v[i] = P[(long long)t * NP + i];
  const int col = tr_cols[m];
  const float x = X[(long long)row * ldx + col];
  xsave[t] = x;
  NsaTransformConsts tc;
  tc.tail_bound = c.tail_bound; tc.two_b = 2.0f * c.tail_bound;
  tc.min_bw = c.min_bw; tc.bw_scale = (float)(1.0 - (double)c.min_bw * K);
  tc.min_bh = c.min_bh; tc.bh_scale = (float)(1.0 - (double)c.min_bh * K);
  tc.min_d = c.min_d; tc.edge_d = c.edge_d; tc.inv_logit_div = c.inv_div;
  typename NsaRQ<K>::State st;
  float y = 0.0f, ld = 0.0f;
  bool bad = false;
  NsaRQ<K>::template phase<0, 0, false, NP>(st, v, x, tc, y, ld, bad);
  NsaRQ<K>::template phase<1, 0, false, NP>(st, v, x, tc, y, ld, bad);
  NsaRQ<K>::template phase<2, 0, false, NP>(st, v, x, tc, y, ld, bad);
  NsaRQ<K>::template phase<3, 0, false, NP>(st, v, x, tc, y, ld, bad);
  X[(long long)row * ldx + col] = y;
  atomicAdd(&logdet[row], ld);
}

// Backward of the spline (hand-derived reverse mode of NsaRQ::phase, forward
// direction).  Inputs: saved x, parameters, upstream gy = dL/dy (read from
// Gx[row][col]) and gld = dL/dlogdet (same for every row).  Outputs: dL/dx
// (overwrites Gx[row][col]), dL/dparams (GP), and X[row][col] restored to x.
template <int K>
__global__ void tr_rq_backward(int n, int d_tr, const int* __restrict__ tr_cols, const float* __restrict__ P,
                               const float* __restrict__ xsave, float* __restrict__ X, int ldx,
                               float* __restrict__ Gx, float gld, float* __restrict__ GP, TrConsts c) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * d_tr) return;
  const int row = t / d_tr, m = t - row * d_tr;
  constexpr int NP = 3 * K - 1;
  const int col = tr_cols[m];
  const float x = xsave[t];
  const float gy = Gx[(long long)row * ldx + col];
  X[(long long)row * ldx + col] = x;  // undo the layer
  const float B = c.tail_bound;
  float gp[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) gp[i] = 0.0f;
  float gx = gy;
  if ((x >= -B) && (x <= B)) {
    float v[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) v[i] = P[(long long)t * NP + i];
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
  Gx[(long long)row * ldx + col] = gx;
#pragma unroll
  for (int i = 0; i < NP; ++i) GP[(long long)t * NP + i] = gp[i];
}

// affine coupling: P row-major [n][2*d_tr] = [shift block | raw-scale block]
__global__ void tr_affine_forward(int n, int d_tr, int additive, const int* __restrict__ tr_cols,
                                  const float* __restrict__ P, float* __restrict__ X, int ldx,
                                  float* __restrict__ xsave, float* __restrict__ logdet) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * d_tr) return;
  const int row = t / d_tr, m = t - row * d_tr;
  const int np = additive ? d_tr : 2 * d_tr;
  const int col = tr_cols[m];
  const float x = X[(long long)row * ldx + col];
  xsave[t] = x;
  const float shift = P[(long long)row * np + m];
  float scale = 1.0f, ls = 0.0f;
  if (!additive) {
    scale = 1.0f / (1.0f + expf(-(P[(long long)row * np + d_tr + m] + 2.0f))) + 1e-3f;
    ls = logf(scale);
  }
  X[(long long)row * ldx + col] = x * scale + shift;
  if (!additive) atomicAdd(&logdet[row], ls);
}

__global__ void tr_affine_backward(int n, int d_tr, int additive, const int* __restrict__ tr_cols,
                                   const float* __restrict__ P, const float* __restrict__ xsave,
                                   float* __restrict__ X, int ldx, float* __restrict__ Gx, float gld,
                                   float* __restrict__ GP) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * d_tr) return;
  const int row = t / d_tr, m = t - row * d_tr;
  const int np = additive ? d_tr : 2 * d_tr;
  const int col = tr_cols[m];
  const float x = xsave[t];
  const float gy = Gx[(long long)row * ldx + col];
  X[(long long)row * ldx + col] = x;
  GP[(long long)row * np + m] = gy;
  if (additive) return;
  const float sig = 1.0f / (1.0f + expf(-(P[(long long)row * np + d_tr + m] + 2.0f)));
  const float scale = sig + 1e-3f;
  const float s_b = gy * x + gld / scale;
  GP[(long long)row * np + d_tr + m] = s_b * sig * (1.0f - sig);
  Gx[(long long)row * ldx + col] = gy * scale;
}

// ---------------------------------------------------------------------------
// loss, gradient seed, gradient norm, AdamW
// ---------------------------------------------------------------------------

// per row: log_prob = -0.5/var * sum z^2 - log_z + logdet;  Gx = z / (var * n)
// loss_acc[0] += -sum(log_prob) / n   (one workgroup, fixed summation order)
__global__ void __launch_bounds__(256) tr_loss(int n, int D, const float* __restrict__ X, int ldx,
                                               const float* __restrict__ logdet, float inv_var, float log_z,
                                               float* __restrict__ Gx, float* __restrict__ loss_acc,
                                               float* __restrict__ log_prob_out, int want_grad) {
  __shared__ float sm[256];
  float local = 0.0f;
  for (int row = threadIdx.x; row < n; row += 256) {
    float e = 0.0f;
    for (int c = 0; c < D; ++c) {
      const float z = X[(long long)row * ldx + c];
      e += z * z;
      if (want_grad) Gx[(long long)row * ldx + c] = z * inv_var / (float)n;
    }
    const float lp = -0.5f * inv_var * e - log_z + logdet[row];
    if (log_prob_out) log_prob_out[row] = lp;
    local += lp;
  }
  sm[threadIdx.x] = local;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss_acc[0] += -sm[0] / (float)n;
}

// sum of squares of the flat gradient (one workgroup => deterministic)
__global__ void __launch_bounds__(1024) tr_grad_sqnorm(long long n, const float* __restrict__ g,
                                                       float* __restrict__ out) {
  __shared__ float sm[1024];
  float s = 0.0f;
  for (long long i = threadIdx.x; i < n; i += 1024) s += g[i] * g[i];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if (threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sm[0];
}

// clip_grad_norm_ (coef = min(1, max_norm / (norm + 1e-6))) fused with torch.optim.AdamW
__global__ void tr_adamw(long long n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                         float* __restrict__ v, const float* __restrict__ sqnorm, float max_norm, float lr,
                         float beta1, float beta2, float eps, float wd, float bc1, float bc2_sqrt) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  float coef = 1.0f;
  if (max_norm > 0.0f) {
    coef = max_norm / (sqrtf(sqnorm[0]) + 1e-6f);
    coef = coef < 1.0f ? coef : 1.0f;
  }
  const float grad = g[i] * coef;
  float pi = p[i] * (1.0f - lr * wd);
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

// ---------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------

// plan (int64, host): header then one record per coupling layer
//   [0] D  [1] H  [2] L (residual blocks)  [3] n_layers (coupling)  [4] transform  [5] bins
//   [6] activation  [7] params per transformed feature  [8] n trainable floats  [9] record words
//   record: d_id, d_tr, id_cols offset, tr_cols offset (into the int32 device table),
//           offsets (into the flat parameter buffer) of W0, b0, {W1, b1, W2, b2} x L, Wf, bf
#define TR_HDR 10
static thread_local char g_terr[256];

struct TrWs {  // workspace carve (floats)
  float *x, *gx, *logdet, *gp, *gh, *gu, *scal;
  float* layer_base;
  long long per_layer;
  long long xsave_words;  // n * max(d_tr): size of a layer's saved-input block
};

static long long tr_per_layer(long long n, long long d_tr_max, long long H, long long L, long long np_max) {
  return n * d_tr_max + (L + 1) * n * H + L * n * H + n * np_max;
}

extern "C" NSA_API size_t nsa_train_workspace_floats(const long long* plan, long long n) {
  const long long D = plan[0], H = plan[1], L = plan[2], NL = plan[3], rec = plan[9];
  long long d_tr_max = 0;
  for (long long b = 0; b < NL; ++b) {
    const long long d_tr = plan[TR_HDR + b * rec + 1];
    if (d_tr > d_tr_max) d_tr_max = d_tr;
  }
  const long long np_max = d_tr_max * plan[7];
  long long total = 2 * n * D + n + n * np_max + 2 * n * H + 64;
  total += NL * tr_per_layer(n, d_tr_max, H, L, np_max);
  return (size_t)total;
}

static void tr_carve(const long long* plan, long long n, float* ws, TrWs& w) {
  const long long D = plan[0], H = plan[1], L = plan[2], NL = plan[3], rec = plan[9];
  long long d_tr_max = 0;
  for (long long b = 0; b < NL; ++b) {
    const long long d_tr = plan[TR_HDR + b * rec + 1];
    if (d_tr > d_tr_max) d_tr_max = d_tr;
  }
  const long long np_max = d_tr_max * plan[7];
  float* p = ws;
  w.scal = p; p += 64;
  w.x = p; p += n * D;
  w.gx = p; p += n * D;
  w.logdet = p; p += n;
  w.gp = p; p += n * np_max;
  w.gh = p; p += n * H;
  w.gu = p; p += n * H;
  w.layer_base = p;
  w.per_layer = tr_per_layer(n, d_tr_max, H, L, np_max);
  w.xsave_words = n * d_tr_max;
}

static inline dim3 tiles(long long rows, long long cols) {
  return dim3((unsigned)((rows + 31) / 32), (unsigned)((cols + 31) / 32));
}

#define TR_SPLINE_DISPATCH(KERNEL, K, ...)                                                          \
  do {                                                                                              \
    if ((K) == 4) hipLaunchKernelGGL(KERNEL<4>, __VA_ARGS__);                                       \
    else if ((K) == 8) hipLaunchKernelGGL(KERNEL<8>, __VA_ARGS__);                                  \
    else if ((K) == 10) hipLaunchKernelGGL(KERNEL<10>, __VA_ARGS__);                                \
    else hipLaunchKernelGGL(KERNEL<16>, __VA_ARGS__);                                               \
  } while (0)

// forward of the whole flow over a batch already copied into w.x; fills the
// per-layer activations when `train` (they are needed by the backward pass)
static void tr_forward(const long long* plan, const int* tabs, const float* params, const float* consts,
                       long long n, TrWs& w, hipStream_t s) {
  const int D = (int)plan[0], H = (int)plan[1], L = (int)plan[2], NL = (int)plan[3];
  const int transform = (int)plan[4], K = (int)plan[5], act = (int)plan[6], npf = (int)plan[7];
  const long long rec = plan[9];
  TrConsts c = {consts[0], consts[1], consts[2], consts[3], consts[4], consts[5]};
  hipMemsetAsync(w.logdet, 0, n * sizeof(float), s);
  for (int b = 0; b < NL; ++b) {
    const long long* r = plan + TR_HDR + b * rec;
    const int d_id = (int)r[0], d_tr = (int)r[1];
    const int* id_cols = tabs + r[2];
    const int* tr_cols = tabs + r[3];
    const long long* off = r + 4;
    float* lay = w.layer_base + b * w.per_layer;
    // layer block: xsave[n*d_tr_max] | h[L+1][n*H] | u[L][n*H] | p[n*np_max]
    float* xsave = lay;
    float* h = xsave + w.xsave_words;
    float* u = h + (long long)(L + 1) * n * H;
    float* p = u + (long long)L * n * H;
    const int np = d_tr * npf;
    // initial layer
    hipLaunchKernelGGL(tr_gemm_nt, tiles(n, H), dim3(64), 0, s, (int)n, d_id, H, w.x, D, id_cols, ACT_NONE,
                       params + off[0], params + off[1], (const float*)nullptr, h, H);
    for (int l = 0; l < L; ++l) {
      float* hl = h + (long long)l * n * H;
      float* ul = u + (long long)l * n * H;
      hipLaunchKernelGGL(tr_gemm_nt, tiles(n, H), dim3(64), 0, s, (int)n, H, H, hl, H, (const int*)nullptr, act,
                         params + off[2 + 4 * l], params + off[3 + 4 * l], (const float*)nullptr, ul, H);
      hipLaunchKernelGGL(tr_gemm_nt, tiles(n, H), dim3(64), 0, s, (int)n, H, H, ul, H, (const int*)nullptr, act,
                         params + off[4 + 4 * l], params + off[5 + 4 * l], hl, hl + (long long)n * H, H);
    }
    float* hL = h + (long long)L * n * H;
    hipLaunchKernelGGL(tr_gemm_nt, tiles(n, np), dim3(64), 0, s, (int)n, H, np, hL, H, (const int*)nullptr,
                       ACT_NONE, params + off[2 + 4 * L], params + off[3 + 4 * L], (const float*)nullptr, p, np);
    const int threads = 256, blocks = (int)((n * d_tr + threads - 1) / threads);
    if (transform == NSA_T_RQ) {
      TR_SPLINE_DISPATCH(tr_rq_forward, K, dim3(blocks), dim3(threads), 0, s, (int)n, d_tr, tr_cols, p, w.x, D,
                         xsave, w.logdet, c);
    } else {
      hipLaunchKernelGGL(tr_affine_forward, dim3(blocks), dim3(threads), 0, s, (int)n, d_tr,
                         transform == NSA_T_ADDITIVE ? 1 : 0, tr_cols, p, w.x, D, xsave, w.logdet);
    }
  }
}

static void tr_backward(const long long* plan, const int* tabs, const float* params, float* grads,
                        const float* consts, long long n, TrWs& w, hipStream_t s) {
  const int D = (int)plan[0], H = (int)plan[1], L = (int)plan[2], NL = (int)plan[3];
  const int transform = (int)plan[4], K = (int)plan[5], act = (int)plan[6], npf = (int)plan[7];
  const long long rec = plan[9];
  TrConsts c = {consts[0], consts[1], consts[2], consts[3], consts[4], consts[5]};
  const float gld = -1.0f / (float)n;  // d loss / d logdet
  for (int b = NL - 1; b >= 0; --b) {
    const long long* r = plan + TR_HDR + b * rec;
    const int d_id = (int)r[0], d_tr = (int)r[1];
    const int* id_cols = tabs + r[2];
    const int* tr_cols = tabs + r[3];
    const long long* off = r + 4;
    float* lay = w.layer_base + b * w.per_layer;
    float* xsave = lay;
    float* h = xsave + w.xsave_words;
    float* u = h + (long long)(L + 1) * n * H;
    float* p = u + (long long)L * n * H;
    const int np = d_tr * npf;
    const int threads = 256, blocks = (int)((n * d_tr + threads - 1) / threads);
    if (transform == NSA_T_RQ) {
      TR_SPLINE_DISPATCH(tr_rq_backward, K, dim3(blocks), dim3(threads), 0, s, (int)n, d_tr, tr_cols, p, xsave,
                         w.x, D, w.gx, gld, w.gp, c);
    } else {
      hipLaunchKernelGGL(tr_affine_backward, dim3(blocks), dim3(threads), 0, s, (int)n, d_tr,
                         transform == NSA_T_ADDITIVE ? 1 : 0, tr_cols, p, xsave, w.x, D, w.gx, gld, w.gp);
    }
    // final layer: dWf, dbf, gh = gp . Wf
    float* hL = h + (long long)L * n * H;
    hipLaunchKernelGGL(tr_gemm_tn, tiles(np, H), dim3(64), 0, s, (int)n, np, H, w.gp, np, hL, H,
                       (const int*)nullptr, ACT_NONE, grads + off[2 + 4 * L], grads + off[3 + 4 * L]);
    hipLaunchKernelGGL(tr_gemm_nn, tiles(n, H), dim3(64), 0, s, (int)n, np, H, w.gp, np, params + off[2 + 4 * L],
                       (const float*)nullptr, 0, ACT_NONE, w.gh, H, (const int*)nullptr, 0);
    for (int l = L - 1; l >= 0; --l) {
      float* hl = h + (long long)l * n * H;
      float* ul = u + (long long)l * n * H;
      // second linear of the block: h_{l+1} = h_l + W2 act(u_l) + b2
      hipLaunchKernelGGL(tr_gemm_tn, tiles(H, H), dim3(64), 0, s, (int)n, H, H, w.gh, H, ul, H,
                         (const int*)nullptr, act, grads + off[4 + 4 * l], grads + off[5 + 4 * l]);
      hipLaunchKernelGGL(tr_gemm_nn, tiles(n, H), dim3(64), 0, s, (int)n, H, H, w.gh, H, params + off[4 + 4 * l], ul,
                         H, act, w.gu, H, (const int*)nullptr, 0);
      // first linear: u_l = W1 act(h_l) + b1
      hipLaunchKernelGGL(tr_gemm_tn, tiles(H, H), dim3(64), 0, s, (int)n, H, H, w.gu, H, hl, H,
                         (const int*)nullptr, act, grads + off[2 + 4 * l], grads + off[3 + 4 * l]);
      hipLaunchKernelGGL(tr_gemm_nn, tiles(n, H), dim3(64), 0, s, (int)n, H, H, w.gu, H, params + off[2 + 4 * l], hl,
                         H, act, w.gh, H, (const int*)nullptr, 1);
    }
    // initial layer: dW0, db0, g_x[:, id_cols] += gh . W0   (x already restored by the spline backward)
    hipLaunchKernelGGL(tr_gemm_tn, tiles(H, d_id), dim3(64), 0, s, (int)n, H, d_id, w.gh, H, w.x, D, id_cols,
                       ACT_NONE, grads + off[0], grads + off[1]);
    hipLaunchKernelGGL(tr_gemm_nn, tiles(n, d_id), dim3(64), 0, s, (int)n, H, d_id, w.gh, H, params + off[0],
                       (const float*)nullptr, 0, ACT_NONE, w.gx, D, id_cols, 2);
  }
}

// One minibatch: forward, loss, backward, clip, AdamW.  loss_acc_dev[0] += loss.
extern "C" NSA_API int nsa_train_step(const long long* plan, const int* tables_dev, const float* consts,
                                      float* params_dev, float* grads_dev, float* adam_m_dev,
                                      float* adam_v_dev, const float* x_dev, long long n,
                                      float* workspace_dev, float lr, float beta1, float beta2, float eps,
                                      float weight_decay, float clip_norm, long long step,
                                      float* loss_acc_dev, void* stream) {
  if (!plan || !tables_dev || !params_dev || !grads_dev || !x_dev || !workspace_dev || !loss_acc_dev || n <= 0)
    return NSA_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  const long long D = plan[0], n_params = plan[8];
  TrWs w;
  tr_carve(plan, n, workspace_dev, w);
  hipLaunchKernelGGL(tr_copy_rows, dim3((unsigned)((n * D + 255) / 256)), dim3(256), 0, s, n * D, x_dev, w.x);
  tr_forward(plan, tables_dev, params_dev, consts, n, w, s);
  hipLaunchKernelGGL(tr_loss, dim3(1), dim3(256), 0, s, (int)n, (int)D, w.x, (int)D, w.logdet, consts[6], consts[7],
                     w.gx, loss_acc_dev, (float*)nullptr, 1);
  tr_backward(plan, tables_dev, params_dev, grads_dev, consts, n, w, s);
  hipLaunchKernelGGL(tr_grad_sqnorm, dim3(1), dim3(1024), 0, s, n_params, grads_dev, w.scal);
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.0f - powf(beta2, (float)step));
  hipLaunchKernelGGL(tr_adamw, dim3((unsigned)((n_params + 255) / 256)), dim3(256), 0, s, n_params, params_dev,
                     grads_dev, adam_m_dev, adam_v_dev, w.scal, clip_norm, lr, beta1, beta2, eps, weight_decay, bc1,
                     bc2s);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? NSA_OK : NSA_E_HIP;
}

// Forward only: loss_acc_dev[0] += -mean(log_prob(x)); optional per-row log_prob.
extern "C" NSA_API int nsa_train_eval(const long long* plan, const int* tables_dev, const float* consts,
                                      const float* params_dev, const float* x_dev, long long n,
                                      float* workspace_dev, float* loss_acc_dev, float* log_prob_out_dev,
                                      void* stream) {
  if (!plan || !tables_dev || !params_dev || !x_dev || !workspace_dev || !loss_acc_dev || n <= 0)
    return NSA_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  const long long D = plan[0];
  TrWs w;
  tr_carve(plan, n, workspace_dev, w);
  hipLaunchKernelGGL(tr_copy_rows, dim3((unsigned)((n * D + 255) / 256)), dim3(256), 0, s, n * D, x_dev, w.x);
  tr_forward(plan, tables_dev, params_dev, consts, n, w, s);
  hipLaunchKernelGGL(tr_loss, dim3(1), dim3(256), 0, s, (int)n, (int)D, w.x, (int)D, w.logdet, consts[6], consts[7],
                     w.gx, loss_acc_dev, log_prob_out_dev, 0);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? NSA_OK : NSA_E_HIP;
}
