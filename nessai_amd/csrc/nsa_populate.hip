// Rejection-sampling kernels of the populate path: importance weights + global
// maximum, accept test, order-preserving stream compaction.
//
// Reference semantics (src/nessai/proposal/flowproposal/flowproposal.py:469,
// 492-498): log_w = log_prior(x) - log_q; log_w -= log_w.max();
// accept = log_w > log(U[0,1)); samples[...] = x[accept][:m].
// HBM-bound byte shuffling: one pass over log_q (8 B/row), one over log_w, and
// D*8 B per ACCEPTED row only.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nessai_amd.h"
#include "nsa_device.h"

#define NSA_ACC_THREADS 256
#define NSA_ACC_ITEMS 4
#define NSA_ACC_ROWS (NSA_ACC_THREADS * NSA_ACC_ITEMS)

// monotone (signed) integer encoding of a double, usable with atomicMax
// loop control of nsa_populate_passes (see NsaLaunch::stop): true when an earlier pass of the
// train already reached the wanted count or the pass limit
#define NSA_STOPPED(stop) ((stop) != nullptr && ((stop)[0] >= (stop)[2] || (stop)[1] >= (stop)[5]))

__host__ __device__ inline long long nsa_key_encode(double v) {
  long long b;
  __builtin_memcpy(&b, &v, 8);
  return (b >= 0) ? b : (b ^ 0x7fffffffffffffffLL);
}
__host__ __device__ inline double nsa_key_decode(long long k) {
  const long long b = (k >= 0) ? k : (k ^ 0x7fffffffffffffffLL);
  double v;
  __builtin_memcpy(&v, &b, 8);
  return v;
}

__global__ void __launch_bounds__(256) nsa_weights_kernel(long long n, const double* __restrict__ log_q,
                                                          const double* __restrict__ log_prior,
                                                          double log_prior_const,
                                                          double* __restrict__ log_w,
                                                          long long* max_key,
                                                          const long long* __restrict__ stop) {
  if (NSA_STOPPED(stop)) return;
  const double ninf = -__builtin_huge_val();
  double m = ninf;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const double lq = log_q[i];
    const double lp = (log_prior != nullptr) ? log_prior[i] : log_prior_const;
    double w = lp - lq;
    if (lq != lq || w != w) w = ninf;  // dropped row (NaN log_q) or NaN weight
    log_w[i] = w;
    m = fmax(m, w);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
  __shared__ double sm[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) sm[wave] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
    // thousands of same-address 64-bit atomics serialise (they were 20 of this kernel's 27 us):
    // only workgroups that can still raise the maximum issue one
    const long long key = nsa_key_encode(m);
    if (key > __atomic_load_n(max_key, __ATOMIC_RELAXED)) atomicMax(max_key, key);
  }
}

// uniform in [0,1) with 53 random bits (as numpy's Generator.random)
__device__ __forceinline__ double nsa_uniform53(unsigned long long seed, unsigned long long grow) {
  uint32_t r[4];
  nsa_philox4x32_10((uint32_t)grow, (uint32_t)(grow >> 32), 0u, 1u, (uint32_t)seed,
                    (uint32_t)(seed >> 32), r);
  const unsigned long long hi = r[0] >> 5, lo = r[1] >> 6;
  return (double)((hi << 26) | lo) * (1.0 / 9007199254740992.0);
}

// pass 1: accept flags + per-block counts
__global__ void __launch_bounds__(NSA_ACC_THREADS) nsa_accept_flag_kernel(
    long long n, const double* __restrict__ log_w, const long long* __restrict__ max_key,
    unsigned long long seed, unsigned long long row_offset, const double* __restrict__ u_in,
    unsigned char* __restrict__ flags, int* __restrict__ block_counts, const long long* __restrict__ stop) {
  if (NSA_STOPPED(stop)) return;
  const double mx = nsa_key_decode(*max_key);
  const long long base = (long long)blockIdx.x * NSA_ACC_ROWS;
  int cnt = 0;
#pragma unroll
  for (int it = 0; it < NSA_ACC_ITEMS; ++it) {
    const long long i = base + it * NSA_ACC_THREADS + threadIdx.x;
    if (i < n) {
      const double u = (u_in != nullptr) ? u_in[i] : nsa_uniform53(seed, row_offset + (unsigned long long)i);
      const double lu = log(u);
      const bool acc = (log_w[i] - mx) > lu;
      flags[i] = acc ? 1 : 0;
      cnt += acc ? 1 : 0;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
  __shared__ int sm[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) sm[wave] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

// pass 2: exclusive scan of the block counts (one workgroup)
__global__ void __launch_bounds__(1024) nsa_accept_scan_kernel(int n_blocks, const int* __restrict__ counts,
                                                               long long* __restrict__ offsets,
                                                               long long* __restrict__ total,
                                                               const long long* __restrict__ stop) {
  if (NSA_STOPPED(stop)) return;
  __shared__ long long sm[1024];
  __shared__ long long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int start = 0; start < n_blocks; start += 1024) {
    const int i = start + threadIdx.x;
    const long long v = (i < n_blocks) ? counts[i] : 0;
    sm[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const long long t = (threadIdx.x >= off) ? sm[threadIdx.x - off] : 0;
      __syncthreads();
      sm[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < n_blocks) offsets[i] = carry + sm[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += sm[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

// pass 3: scatter accepted rows in draw order
__global__ void __launch_bounds__(NSA_ACC_THREADS) nsa_accept_scatter_kernel(
    long long n, int D, const unsigned char* __restrict__ flags, const long long* __restrict__ offsets,
    const double* __restrict__ x64, long long capacity, double* __restrict__ x_acc,
    long long* __restrict__ idx_acc, long long idx_base, const long long* __restrict__ stop) {
  // (x64 == null: the lean passes - nsa_populate_passes_lean - keep no candidate rows; an accepted row is its
  // global Philox row idx_base + i, which nsa_populate_replay turns back into x)
  if (NSA_STOPPED(stop)) return;
  const long long written = (stop != nullptr) ? stop[3] : 0;  // rows of earlier passes of the train
  const long long base = (long long)blockIdx.x * NSA_ACC_ROWS;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __shared__ int wave_cnt[NSA_ACC_ITEMS][4];
  int my_flag[NSA_ACC_ITEMS], my_rank[NSA_ACC_ITEMS];
#pragma unroll
  for (int it = 0; it < NSA_ACC_ITEMS; ++it) {
    const long long i = base + it * NSA_ACC_THREADS + threadIdx.x;
    const int f = (i < n) ? (int)flags[i] : 0;
    const unsigned long long ball = __ballot(f);
    my_flag[it] = f;
    my_rank[it] = __popcll(ball & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[it][wave] = __popcll(ball);
  }
  __syncthreads();
  long long running = written + offsets[blockIdx.x];
#pragma unroll
  for (int it = 0; it < NSA_ACC_ITEMS; ++it) {
    int before = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) before += (w < wave) ? wave_cnt[it][w] : 0;
    const int tot = wave_cnt[it][0] + wave_cnt[it][1] + wave_cnt[it][2] + wave_cnt[it][3];
    if (my_flag[it]) {
      const long long dst = running + before + my_rank[it];
      if (dst < capacity) {
        const long long i = base + it * NSA_ACC_THREADS + threadIdx.x;
        if (x64 != nullptr)
          for (int c = 0; c < D; ++c) x_acc[dst * D + c] = x64[i * D + c];
        if (idx_acc != nullptr) idx_acc[dst] = idx_base + i;
      }
    }
    running += tot;
  }
}

// host-side launchers ---------------------------------------------------------

extern "C" __attribute__((visibility("hidden"))) hipError_t nsa_launch_weights(
    long long n, const double* log_q, const double* log_prior, double log_prior_const,
    double* log_w, long long* max_key, hipStream_t stream, const long long* stop) {
  if (n <= 0) return hipSuccess;
  long long blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(nsa_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, n, log_q,
                     log_prior, log_prior_const, log_w, max_key, stop);
  return hipGetLastError();
}

// Small passes (n <= NSA_SMALL_N): weights, maximum, accept flags, scan, compaction and the commit
// of one pass in ONE workgroup - five launches of a few microseconds each would otherwise
// dominate a pass whose draw kernel takes ~20 us.  Thread t owns the contiguous rows
// [t c, (t + 1) c), so thread order is draw order; the arithmetic is that of the separate kernels.
#define NSA_SMALL_N 16384
#define NSA_SMALL_C 16
__global__ void __launch_bounds__(1024) nsa_small_tail_kernel(long long n, int D, const double* __restrict__ log_q,
                                                              const double* __restrict__ log_prior,
                                                              double log_prior_const, unsigned long long seed,
                                                              unsigned long long row_offset,
                                                              const double* __restrict__ x64, long long capacity,
                                                              double* __restrict__ x_acc, long long* state) {
  if (NSA_STOPPED(state)) return;
  __shared__ double smax[16];
  __shared__ int scnt[1024];
  const double ninf = -__builtin_huge_val();
  const int c = (int)((n + 1023) / 1024);
  const long long lo = (long long)threadIdx.x * c;
  double w[NSA_SMALL_C];
  double m = ninf;
#pragma unroll
  for (int k = 0; k < NSA_SMALL_C; ++k) {
    const long long i = lo + k;
    double v = ninf;
    if (k < c && i < n) {
      const double lq = log_q[i];
      const double lp = (log_prior != nullptr) ? log_prior[i] : log_prior_const;
      v = lp - lq;
      if (lq != lq || v != v) v = ninf;
    }
    w[k] = v;
    m = fmax(m, v);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = m;
  __syncthreads();
  double mx = smax[0];
#pragma unroll
  for (int k = 1; k < 16; ++k) mx = fmax(mx, smax[k]);
  unsigned flags = 0;
  int cnt = 0;
#pragma unroll
  for (int k = 0; k < NSA_SMALL_C; ++k) {
    const long long i = lo + k;
    if (k < c && i < n) {
      const double lu = log(nsa_uniform53(seed, row_offset + (unsigned long long)i));
      const bool acc = (w[k] - mx) > lu;
      flags |= acc ? (1u << k) : 0u;
      cnt += acc ? 1 : 0;
    }
  }
  scnt[threadIdx.x] = cnt;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {  // inclusive scan
    const int t = (threadIdx.x >= off) ? scnt[threadIdx.x - off] : 0;
    __syncthreads();
    scnt[threadIdx.x] += t;
    __syncthreads();
  }
  const long long total = scnt[1023];
  long long dst = state[3] + scnt[threadIdx.x] - cnt;
#pragma unroll
  for (int k = 0; k < NSA_SMALL_C; ++k) {
    if (flags & (1u << k)) {
      if (dst < capacity) {
        const long long i = lo + k;
        for (int cc = 0; cc < D; ++cc) x_acc[dst * D + cc] = x64[i * D + cc];
      }
      ++dst;
    }
  }
  __syncthreads();  // every thread has read state[3]
  if (threadIdx.x == 0) {
    state[4] = total;
    state[0] += total;
    state[1] += 1;
    const long long wr = state[3] + total;
    state[3] = wr < capacity ? wr : capacity;
  }
}

extern "C" __attribute__((visibility("hidden"))) hipError_t nsa_launch_small_tail(
    long long n, int D, const double* log_q, const double* log_prior, double log_prior_const,
    unsigned long long seed, unsigned long long row_offset, const double* x64, long long capacity, double* x_acc,
    long long* state, hipStream_t stream) {
  hipLaunchKernelGGL(nsa_small_tail_kernel, dim3(1), dim3(1024), 0, stream, n, D, log_q, log_prior,
                     log_prior_const, seed, row_offset, x64, capacity, x_acc, state);
  return hipGetLastError();
}

// A whole TRAIN of small passes at once (nsa_populate_passes_small, include/nessai_amd.h): the draw kernel
// has already run over the n_passes * n rows of the train in one launch; workgroup i does the tail of pass i
// - weights, the pass's own maximum, accept flags with the uniforms of the global rows, compaction into the
// pass's staging block - and nsa_small_commit_kernel then walks the passes in order against the loop state,
// appending the staged rows of every pass that counts.  Same arithmetic and row order as the pass-at-a-time
// kernel above; surplus passes only cost their share of two launches.
__global__ void __launch_bounds__(1024) nsa_small_tails_kernel(long long n, int D, const double* __restrict__ log_q_many,
                                                               const double* __restrict__ log_prior_many,
                                                               double log_prior_const, unsigned long long seed,
                                                               unsigned long long row_offset,
                                                               const double* __restrict__ x64_many, long long cap,
                                                               double* __restrict__ stage, long long* __restrict__ counts,
                                                               const long long* __restrict__ state) {
  if (NSA_STOPPED(state)) return;
  const long long pass = blockIdx.x;
  const double* log_q = log_q_many + pass * n;
  const double* log_prior = (log_prior_many != nullptr) ? log_prior_many + pass * n : nullptr;
  const double* x64 = x64_many + pass * n * D;
  double* out = stage + pass * cap * D;
  const unsigned long long off = row_offset + (unsigned long long)(pass * n);
  __shared__ double smax[16];
  __shared__ int scnt[1024];
  const double ninf = -__builtin_huge_val();
  const int c = (int)((n + 1023) / 1024);
  const long long lo = (long long)threadIdx.x * c;
  double w[NSA_SMALL_C];
  double m = ninf;
#pragma unroll
  for (int k = 0; k < NSA_SMALL_C; ++k) {
    const long long i = lo + k;
    double v = ninf;
    if (k < c && i < n) {
      const double lq = log_q[i];
      const double lp = (log_prior != nullptr) ? log_prior[i] : log_prior_const;
      v = lp - lq;
      if (lq != lq || v != v) v = ninf;
    }
    w[k] = v;
    m = fmax(m, v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = m;
  __syncthreads();
  double mx = smax[0];
#pragma unroll
  for (int k = 1; k < 16; ++k) mx = fmax(mx, smax[k]);
  unsigned flags = 0;
  int cnt = 0;
#pragma unroll
  for (int k = 0; k < NSA_SMALL_C; ++k) {
    const long long i = lo + k;
    if (k < c && i < n) {
      const double lu = log(nsa_uniform53(seed, off + (unsigned long long)i));
      const bool acc = (w[k] - mx) > lu;
      flags |= acc ? (1u << k) : 0u;
      cnt += acc ? 1 : 0;
    }
  }
  scnt[threadIdx.x] = cnt;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {  // inclusive scan
    const int t = (threadIdx.x >= o) ? scnt[threadIdx.x - o] : 0;
    __syncthreads();
    scnt[threadIdx.x] += t;
    __syncthreads();
  }
  long long dst = scnt[threadIdx.x] - cnt;
#pragma unroll
  for (int k = 0; k < NSA_SMALL_C; ++k) {
    if (flags & (1u << k)) {
      if (dst < cap) {
        const long long i = lo + k;
        for (int cc = 0; cc < D; ++cc) out[dst * D + cc] = x64[i * D + cc];
      }
      ++dst;
    }
  }
  if (threadIdx.x == 1023) counts[pass] = scnt[1023];
}

// The stop test of a train of small passes, replayed in pass order against the loop state
// (FlowProposal.populate's `while n_accepted < n_samples`, flowproposal.py:447-518): ONE thread walks the
// counts - reads of the state once, no barrier per pass (the first version crossed three per pass and took
// 73 us for sixty passes) - and leaves where each counted pass's rows go; then every thread copies its share
// of ALL staged rows (an element finds its pass by bisection of <= 64 offsets).
#define NSA_SMALL_MAX_PASSES 64
__global__ void __launch_bounds__(1024) nsa_small_commit_kernel(int n_passes, int D, long long cap,
                                                                const double* __restrict__ stage,
                                                                const long long* __restrict__ counts,
                                                                long long capacity, double* __restrict__ x_acc,
                                                                long long* state) {
  __shared__ long long s_first[NSA_SMALL_MAX_PASSES + 1];  // first output ELEMENT of pass i (prefix of take * D)
  __shared__ long long s_dst[NSA_SMALL_MAX_PASSES];        // first output row of pass i
  __shared__ int s_np;
  if (threadIdx.x == 0) {
    long long acc = state[0], passes = state[1], written = state[3], last = state[4];
    const long long want = state[2], limit = state[5];
    int np = 0;
    long long first = 0;
    for (int i = 0; i < n_passes && i < NSA_SMALL_MAX_PASSES; ++i) {
      if (acc >= want || passes >= limit) break;
      const long long got = counts[i];
      long long take = got < cap ? got : cap;
      if (take > capacity - written) take = capacity - written;
      s_first[np] = first;
      s_dst[np] = written;
      first += take * D;
      ++np;
      last = got;
      acc += got;
      passes += 1;
      written = (written + got < capacity) ? written + got : capacity;
    }
    s_first[np] = first;
    s_np = np;
    state[0] = acc;
    state[1] = passes;
    state[3] = written;
    state[4] = last;
  }
  __syncthreads();
  const int np = s_np;
  const long long total = s_first[np];
  for (long long k = threadIdx.x; k < total; k += blockDim.x) {
    int lo = 0, hi = np;  // pass with s_first[lo] <= k < s_first[lo + 1]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (s_first[mid] <= k) lo = mid;
      else hi = mid;
    }
    x_acc[s_dst[lo] * D + (k - s_first[lo])] = stage[(long long)lo * cap * D + (k - s_first[lo])];
  }
}

// ---- the measured precision policy's probe (HipFlow._pack_auto): one comparison kernel instead of a dozen
// torch element-wise operators, each of which loads its own code object the first time it runs (20-120 ms
// apiece: the first populate() of a process spent ~450 ms there) ------------------------------------------------
// out[0] = max over rows of max(|ld - ld_ref| / max(|log N(z) - ld_ref|, 1e-3), |lp - lp_ref| / max(|lp_ref|, 1e-3))
// as a sign-ordered int64 key of the double (NaN -> +inf: a probe that produced one never passes).
__global__ void __launch_bounds__(256) nsa_probe_compare_kernel(long long n, int D, const float* __restrict__ z,
                                                                const float* __restrict__ ld, const float* __restrict__ lp,
                                                                const float* __restrict__ ld_ref,
                                                                const float* __restrict__ lp_ref,
                                                                unsigned long long* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  double err = 0.0;
  if (i < n) {
    double r2 = 0.0;
    for (int c = 0; c < D; ++c) {
      const double v = (double)z[i * D + c];
      r2 += v * v;
    }
    const double log_pz = -0.5 * r2 - 0.5 * D * 1.8378770664093453;  // log(2 pi)
    const double lq_ref = log_pz - (double)ld_ref[i];
    const double a = fabs((double)ld[i] - (double)ld_ref[i]) / fmax(fabs(lq_ref), 1e-3);
    const double b = fabs((double)lp[i] - (double)lp_ref[i]) / fmax(fabs((double)lp_ref[i]), 1e-3);
    err = fmax(a, b);
    if (a != a || b != b) err = __builtin_huge_val();
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) err = fmax(err, __shfl_xor(err, o));
  if ((threadIdx.x & 63) == 0 && err > 0.0) atomicMax(out, (unsigned long long)__double_as_longlong(err));  // err >= 0: bit order = value order
}

// x_out = x_in moved one unit in the last place away from zero (finite non-zero values; the probe's "same
// flow, inputs one ulp further out" twin)
__global__ void __launch_bounds__(256) nsa_probe_nudge_kernel(long long n, const float* __restrict__ x_in,
                                                              float* __restrict__ x_out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned u = __float_as_uint(x_in[i]);
  const unsigned mag = u & 0x7fffffffu;
  x_out[i] = __uint_as_float((mag != 0u && mag < 0x7f7fffffu) ? u + 1u : u);
}

// flag[0] |= 1 when a value is NaN or at / beyond `limit` in magnitude (the weights of a split-f16 image)
__global__ void __launch_bounds__(256) nsa_probe_range_kernel(long long n, const float* __restrict__ v, float limit,
                                                              int* __restrict__ flag) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool bad = i < n && !(fabsf(v[i]) < limit);
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

extern "C" __attribute__((visibility("hidden"))) hipError_t nsa_launch_probe_compare(
    long long n, int D, const float* z, const float* ld, const float* lp, const float* ld_ref, const float* lp_ref,
    unsigned long long* out, hipStream_t stream) {
  hipLaunchKernelGGL(nsa_probe_compare_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, D, z, ld, lp,
                     ld_ref, lp_ref, out);
  return hipGetLastError();
}
extern "C" __attribute__((visibility("hidden"))) hipError_t nsa_launch_probe_nudge(long long n, const float* x_in,
                                                                                  float* x_out, hipStream_t stream) {
  hipLaunchKernelGGL(nsa_probe_nudge_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, x_in, x_out);
  return hipGetLastError();
}
extern "C" __attribute__((visibility("hidden"))) hipError_t nsa_launch_probe_range(long long n, const float* v,
                                                                                  float limit, int* flag,
                                                                                  hipStream_t stream) {
  hipLaunchKernelGGL(nsa_probe_range_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, v, limit, flag);
  return hipGetLastError();
}

extern "C" __attribute__((visibility("hidden"))) hipError_t nsa_launch_small_train(
    long long n, int D, int n_passes, const double* log_q_many, const double* log_prior_many, double log_prior_const,
    unsigned long long seed, unsigned long long row_offset, const double* x64_many, long long cap, double* stage,
    long long* counts, long long capacity, double* x_acc, long long* state, hipStream_t stream) {
  hipLaunchKernelGGL(nsa_small_tails_kernel, dim3((unsigned)n_passes), dim3(1024), 0, stream, n, D, log_q_many,
                     log_prior_many, log_prior_const, seed, row_offset, x64_many, cap, stage, counts, state);
  hipLaunchKernelGGL(nsa_small_commit_kernel, dim3(1), dim3(1024), 0, stream, n_passes, D, cap, stage, counts, capacity,
                     x_acc, state);
  return hipGetLastError();
}

// end of one pass of a train: publish its counts (the only writer of the loop state)
// Sharded runs: `global_count` holds the pass's accepted count summed over the ranks (the stop test
// is on the global count; rows are appended locally) and `history[pass]` keeps this rank's count
// so that the host can interleave the ranks' rows pass by pass at the end.
__global__ void nsa_pass_commit_kernel(long long* state, long long capacity, const long long* global_count,
                                       long long* history) {
  if (NSA_STOPPED(state)) return;
  const long long got = state[4];
  if (history != nullptr) history[state[1]] = got;
  state[0] += (global_count != nullptr) ? global_count[0] : got;
  state[1] += 1;
  const long long w = state[3] + got;
  state[3] = w < capacity ? w : capacity;
}

extern "C" __attribute__((visibility("hidden"))) hipError_t nsa_launch_pass_commit(
    long long* state, long long capacity, const long long* global_count, long long* history, hipStream_t stream) {
  hipLaunchKernelGGL(nsa_pass_commit_kernel, dim3(1), dim3(1), 0, stream, state, capacity, global_count, history);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Flow image rebuilt on the device (nsa_repack_image, include/nessai_amd.h; maps from nessai_amd/pack.py
// repack_map): parameter word k of the image is one float32 parameter, or the hi / lo 16-bit halves of two
// parameters - the conversions of pack.split_bf16 / pack.split_f16 (round to nearest even), bit for bit.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t nsa_bf16_rne(float x) {
  const uint32_t u = __float_as_uint(x);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;  // pack.bf16_bits
}
__device__ __forceinline__ void nsa_split16_host_exact(float x, int precision, uint32_t& hi, uint32_t& lo) {
  if (precision == NSA_PREC_F16X3) {
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)(x - (float)h);
    hi = (uint32_t)__builtin_bit_cast(unsigned short, h);
    lo = (uint32_t)__builtin_bit_cast(unsigned short, l);
  } else {
    hi = nsa_bf16_rne(x);
    lo = nsa_bf16_rne(x - __uint_as_float(hi << 16));
  }
}

__global__ void __launch_bounds__(256) nsa_repack_kernel(long long n, const int* __restrict__ pos,
                                                         const int* __restrict__ e0, const int* __restrict__ e1,
                                                         const unsigned char* __restrict__ kind,
                                                         const float* __restrict__ params, int precision,
                                                         uint32_t* __restrict__ image) {
#pragma clang fp contract(off)
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int i0 = e0[k], i1 = e1[k];
  const float a = i0 >= 0 ? params[i0] : 0.0f;
  const int kd = kind[k];
  if (kd == 1) {
    image[pos[k]] = __float_as_uint(a);
    return;
  }
  const float b = i1 >= 0 ? params[i1] : 0.0f;
  uint32_t ah, al, bh, bl;
  nsa_split16_host_exact(a, precision, ah, al);
  nsa_split16_host_exact(b, precision, bh, bl);
  image[pos[k]] = (kd == 2) ? (ah | (bh << 16)) : (al | (bl << 16));
}

extern "C" __attribute__((visibility("hidden"))) hipError_t nsa_launch_repack(
    long long n, const int* pos, const int* e0, const int* e1, const unsigned char* kind, const float* params,
    int precision, uint32_t* image, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(nsa_repack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, pos, e0, e1, kind,
                     params, precision, image);
  return hipGetLastError();
}

// accumulate_weights (flowproposal.py:471-476): sum_r exp(log_w[r] - max) of each pass of a train, the
// term a pass adds to the expected number of accepted rows.  One workgroup per pass, fixed summation
// order (thread-strided partial sums, then a tree over the 1024 threads): the value does not depend on
// the launch, only on how the rows are dealt to the ranks.
__global__ void __launch_bounds__(1024) nsa_sumexp_kernel(long long n, const double* __restrict__ log_w_many,
                                                          const long long* __restrict__ max_keys,
                                                          double* __restrict__ out) {
  const double* log_w = log_w_many + (long long)blockIdx.x * n;
  const double m = nsa_key_decode(max_keys[blockIdx.x]);
  double s = 0.0;
  for (long long i = threadIdx.x; i < n; i += 1024) {
    const double w = log_w[i];
    if (w > -__builtin_huge_val()) s += exp(w - m);
  }
  __shared__ double sm[1024];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = sm[0];
}

extern "C" __attribute__((visibility("hidden"))) hipError_t nsa_launch_sumexp(
    long long n, int n_passes, const double* log_w_many, const long long* max_keys, double* out,
    hipStream_t stream) {
  hipLaunchKernelGGL(nsa_sumexp_kernel, dim3((unsigned)n_passes), dim3(1024), 0, stream, n, log_w_many, max_keys,
                     out);
  return hipGetLastError();
}

// Trains of sharded passes (nsa_populate_train_*, include/nessai_amd.h): the tails of a train run on a
// scratch copy of the loop state that never stops; the commit then replays the passes against the
// real state with the global counts.
__global__ void nsa_train_scratch_kernel(const long long* state, long long* scratch) {
  scratch[0] = 0;
  scratch[1] = 0;
  scratch[2] = 0x7fffffffffffffffLL;
  scratch[3] = state[3];
  scratch[4] = 0;
  scratch[5] = 0x7fffffffffffffffLL;
}

__global__ void nsa_train_commit_kernel(long long* state, long long capacity, int n_passes,
                                        const long long* local_counts, const long long* global_counts,
                                        long long* history) {
  for (int i = 0; i < n_passes; ++i) {
    if (NSA_STOPPED(state)) return;
    const long long got = local_counts[i];
    if (history != nullptr) history[state[1]] = got;
    state[0] += (global_counts != nullptr) ? global_counts[i] : got;
    state[1] += 1;
    const long long w = state[3] + got;
    state[3] = w < capacity ? w : capacity;
  }
}

extern "C" __attribute__((visibility("hidden"))) hipError_t nsa_launch_train_scratch(
    const long long* state, long long* scratch, hipStream_t stream) {
  hipLaunchKernelGGL(nsa_train_scratch_kernel, dim3(1), dim3(1), 0, stream, state, scratch);
  return hipGetLastError();
}

extern "C" __attribute__((visibility("hidden"))) hipError_t nsa_launch_train_commit(
    long long* state, long long capacity, int n_passes, const long long* local_counts,
    const long long* global_counts, long long* history, hipStream_t stream) {
  hipLaunchKernelGGL(nsa_train_commit_kernel, dim3(1), dim3(1), 0, stream, state, capacity, n_passes, local_counts,
                     global_counts, history);
  return hipGetLastError();
}

extern "C" __attribute__((visibility("hidden"))) size_t nsa_accept_workspace_bytes(long long n) {
  const long long nb = (n + NSA_ACC_ROWS - 1) / NSA_ACC_ROWS;
  size_t bytes = ((size_t)n + 255) & ~(size_t)255;      // flags
  bytes += (((size_t)nb * 4) + 255) & ~(size_t)255;     // block counts
  bytes += (((size_t)nb * 8) + 255) & ~(size_t)255;     // block offsets
  return bytes + 256;
}

extern "C" __attribute__((visibility("hidden"))) hipError_t nsa_launch_accept(
    long long n, int D, const double* log_w, const long long* max_key, unsigned long long seed,
    unsigned long long row_offset, const double* u_in, const double* x64, long long capacity,
    double* x_acc, long long* idx_acc, long long* n_accept, void* workspace, hipStream_t stream,
    const long long* stop, long long idx_base) {
  if (n <= 0) return hipMemsetAsync(n_accept, 0, 8, stream);
  const long long nb = (n + NSA_ACC_ROWS - 1) / NSA_ACC_ROWS;
  unsigned char* flags = (unsigned char*)workspace;
  size_t off = ((size_t)n + 255) & ~(size_t)255;
  int* counts = (int*)((char*)workspace + off);
  off += (((size_t)nb * 4) + 255) & ~(size_t)255;
  long long* offsets = (long long*)((char*)workspace + off);
  hipLaunchKernelGGL(nsa_accept_flag_kernel, dim3((unsigned)nb), dim3(NSA_ACC_THREADS), 0, stream, n,
                     log_w, max_key, seed, row_offset, u_in, flags, counts, stop);
  hipLaunchKernelGGL(nsa_accept_scan_kernel, dim3(1), dim3(1024), 0, stream, (int)nb, counts, offsets,
                     n_accept, stop);
  hipLaunchKernelGGL(nsa_accept_scatter_kernel, dim3((unsigned)nb), dim3(NSA_ACC_THREADS), 0, stream, n,
                     D, flags, offsets, x64, capacity, x_acc, idx_acc, idx_base, stop);
  return hipGetLastError();
}

extern "C" long long nsa_max_key_init(void) { return nsa_key_encode(-__builtin_huge_val()); }
extern "C" long long nsa_max_key_encode(double v) { return nsa_key_encode(v); }
extern "C" double nsa_max_key_decode(long long key) { return nsa_key_decode(key); }

// ---------------------------------------------------------------------------
// analytic benchmark likelihoods on the accepted rows
// ---------------------------------------------------------------------------
// One thread per row, float64, no contraction, and numpy's reduction order for a row of <= 128
// terms (eight strided partial sums combined pairwise, then the tail) so that the values equal the
// host Model.log_likelihood bit for bit.
//   kind 0  Rosenbrock  -sum_i 100 (x_{i+1} - x_i^2)^2 + (1 - x_i)^2      (examples/rosenbrock.py:33-45)
//   kind 1  diagonal Gaussian  -0.5 sum_i ((x_i - mu_i) / sigma_i)^2 - cst  (mu/sigma NULL: 0 / 1)

__device__ __forceinline__ double nsa_ll_term(int kind, const double* __restrict__ x, int i,
                                              const double* __restrict__ mu, const double* __restrict__ sigma) {
#pragma clang fp contract(off)
  if (kind == 0) {
    const double x0 = x[i], x1 = x[i + 1];
    const double a = x1 - x0 * x0;
    const double b = 1.0 - x0;
    return 100.0 * (a * a) + b * b;
  }
  double t = x[i];
  if (mu != nullptr) t = (t - mu[i]) / sigma[i];
  return t * t;
}

// log L of one row
__device__ __forceinline__ double nsa_ll_row(int kind, int D, const double* __restrict__ xr,
                                             const double* __restrict__ mu, const double* __restrict__ sigma,
                                             double cst) {
#pragma clang fp contract(off)
  const int m = (kind == 0) ? D - 1 : D;  // number of terms
  double s;
  if (m < 8) {
    s = 0.0;
    for (int i = 0; i < m; ++i) s += nsa_ll_term(kind, xr, i, mu, sigma);
  } else {
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = nsa_ll_term(kind, xr, j, mu, sigma);
    int i = 8;
    for (; i + 8 <= m; i += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] += nsa_ll_term(kind, xr, i + j, mu, sigma);
    }
    s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < m; ++i) s += nsa_ll_term(kind, xr, i, mu, sigma);
  }
  return (kind == 0) ? -s : (-0.5 * s - cst);
}

__global__ void nsa_log_likelihood_kernel(int kind, long long n, int D, const double* __restrict__ x,
                                          const double* __restrict__ mu, const double* __restrict__ sigma,
                                          double cst, double* __restrict__ out) {
  const long long row = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (row >= n) return;
  out[row] = nsa_ll_row(kind, D, x + row * D, mu, sigma, cst);
}

extern "C" hipError_t nsa_launch_log_likelihood(int kind, long long n, int D, const double* x, const double* mu,
                                                const double* sigma, double cst, double* out, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(nsa_log_likelihood_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, kind, n, D,
                     x, mu, sigma, cst, out);
  return hipGetLastError();
}


// ---------------------------------------------------------------------------
// Truncation rules applied after the flow (src/nessai/proposal/flowproposal/truncation.py):
//   min_log_q             keep log_q > min_log_q            (apply_after_backward, :385-392)
//   likelihood_threshold  keep logL  > threshold            (apply_after_likelihood, :423-430)
// A dropped row is marked the way the draw kernel marks rows outside the prior bounds
// (log_q = NaN): it takes no part in the batch maximum and is never accepted.  8 (+8) bytes in,
// at most 8 bytes out per row - HBM-bound byte work.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) nsa_truncate_kernel(long long n, double* __restrict__ log_q,
                                                           double min_log_q,
                                                           const double* __restrict__ log_l,
                                                           double log_l_threshold) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double lq = log_q[i];
  bool keep = lq > min_log_q;  // false for NaN as well (already dropped)
  if (log_l != nullptr) keep = keep && (log_l[i] > log_l_threshold);
  if (!keep && lq == lq) log_q[i] = __longlong_as_double(0x7ff8000000000000LL);
}

// The same two rules inside a train of passes (nsa_post_rules, include/nessai_amd.h): the likelihood of a
// candidate that is still alive is evaluated in place (the built-in analytic likelihoods, the arithmetic of
// nsa_log_likelihood_kernel) - no (n,) log L array, and the kernel honours the loop state of the train.
__global__ void __launch_bounds__(256) nsa_post_rules_kernel(long long n, int D, double* __restrict__ log_q,
                                                             double min_log_q, const double* __restrict__ x64,
                                                             int ll_kind, const double* __restrict__ mu,
                                                             const double* __restrict__ sigma, double cst,
                                                             double log_l_threshold,
                                                             const long long* __restrict__ stop) {
  if (NSA_STOPPED(stop)) return;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double lq = log_q[i];
  if (lq != lq) return;  // already dropped
  bool keep = lq > min_log_q;
  if (keep && ll_kind >= 0) keep = nsa_ll_row(ll_kind, D, x64 + i * D, mu, sigma, cst) > log_l_threshold;
  if (!keep) log_q[i] = __longlong_as_double(0x7ff8000000000000LL);
}

extern "C" hipError_t nsa_launch_post_rules(long long n, int D, double* log_q, const nsa_post_rules* r,
                                            const double* x64, const long long* stop, hipStream_t stream) {
  if (n == 0 || r == nullptr) return hipSuccess;
  const unsigned grid = (unsigned)((n + 255) / 256);
  nsa_post_rules_kernel<<<grid, 256, 0, stream>>>(n, D, log_q, r->min_log_q, x64, r->ll_kind, r->ll_mu_dev,
                                                  r->ll_sigma_dev, r->ll_const, r->log_l_threshold, stop);
  return hipGetLastError();
}

extern "C" hipError_t nsa_launch_truncate(long long n, double* log_q, double min_log_q, const double* log_l,
                                          double log_l_threshold, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const unsigned grid = (unsigned)((n + 255) / 256);
  nsa_truncate_kernel<<<grid, 256, 0, stream>>>(n, log_q, min_log_q, log_l, log_l_threshold);
  return hipGetLastError();
}


// log_q_all[0 .. P) of one row into `row` and its log Q (the arithmetic of nsa_importance_mix_kernel, as ONE function
// with contraction off: the kernels that share it give the same bits)
__device__ __forceinline__ double nsa_mix_row(long long n, long long r, int n_flows, const float* __restrict__ log_prob_all,
                                              double lj, const double* __restrict__ weights, double* row) {
#pragma clang fp contract(off)
  const int P = n_flows + 1;
  // scipy masks the entries of a zero-weight proposal (a[b == 0] = -inf) BEFORE it takes the maximum:
  // such a proposal takes no part in log Q, whatever its density (NaN and inf included)
  const double ninf = -__builtin_huge_val();
  double a_max = (weights[0] == 0.0) ? ninf : 0.0;  // column 0
  row[0] = 0.0;
  for (int j = 0; j < n_flows; ++j) {
    const double a = (double)log_prob_all[(long long)j * n + r] + lj;
    row[1 + j] = a;
    const double am = (weights[1 + j] == 0.0) ? ninf : a;
    a_max = (am > a_max || am != am) ? am : a_max;  // numpy.amax propagates NaN
  }
  if (!((a_max - a_max) == 0.0)) a_max = 0.0;   // scipy: a_max[~isfinite(a_max)] = 0
  double s = 0.0;
  for (int k = 0; k < P; ++k)
    if (weights[k] != 0.0) s += weights[k] * exp(row[k] - a_max);
  return log(s) + a_max;
}

// ---------------------------------------------------------------------------
// Meta-proposal of the importance sampler (src/nessai/proposal/importance.py:415-441):
//   log_q_all[r][0]     = 0                                   (the prior, proposal -1)
//   log_q_all[r][1 + j] = log_prob_j(x'_r) + log|J|_r         (flow j in the unit hypercube)
//   log_Q[r]            = logsumexp(log_q_all[r], b = weights)
// in float64 as the reference's numpy / scipy glue: scipy's logsumexp subtracts the row maximum
// (0 if it is not finite), sums b * exp(a - max) and adds the maximum back.  One thread per row;
// 4 F bytes in (the (F, n) float32 log-probabilities, coalesced across rows), 8 (F + 2) bytes out.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) nsa_importance_mix_kernel(long long n, int n_flows,
                                                                 const float* __restrict__ log_prob_all,
                                                                 const double* __restrict__ log_j,
                                                                 const double* __restrict__ weights,
                                                                 double* __restrict__ log_q_all,
                                                                 double* __restrict__ log_Q) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int P = n_flows + 1;
  const double lj = log_j[r];
  double* row = log_q_all + r * P;
  log_Q[r] = nsa_mix_row(n, r, n_flows, log_prob_all, lj, weights, row);
}

// The same mix, written as ONE record per drawn row for a single read-back by the host
// (ImportanceFlowProposal.draw, src/nessai/proposal/importance.py:443-526):
//   packed[r] = { x[r][0 .. D) (clipped to [0, 1] when `clip`), log_q_all[r][0 .. P), log_Q[r], log_j[r] }
// (row stride D + P + 2 doubles; log_j is NaN for the rows the reference's first filter drops: the host keeps
// the others).  Replaces the mix kernel, the clamp and three strided copies of the draw's device epilogue.
__global__ void __launch_bounds__(256) nsa_importance_pack_kernel(long long n, int n_flows, int D,
                                                                  const float* __restrict__ log_prob_all,
                                                                  const double* __restrict__ log_j,
                                                                  const double* __restrict__ weights,
                                                                  const double* __restrict__ x64, int clip,
                                                                  double* __restrict__ packed) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int P = n_flows + 1;
  double* rec = packed + r * (D + P + 2);
  const double* xr = x64 + r * D;
  for (int i = 0; i < D; ++i) {
    double v = xr[i];
    if (clip) v = fmin(fmax(v, 0.0), 1.0);  // (numpy.clip: NaN stays NaN)
    rec[i] = v;
  }
  double* row = rec + D;
  const double lj = log_j[r];
  row[P] = nsa_mix_row(n, r, n_flows, log_prob_all, lj, weights, row);
  row[P + 1] = lj;
}

extern "C" hipError_t nsa_launch_importance_pack(long long n, int n_flows, int D, const float* log_prob_all,
                                                 const double* log_j, const double* weights, const double* x64,
                                                 int clip, double* packed, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const unsigned grid = (unsigned)((n + 255) / 256);
  nsa_importance_pack_kernel<<<grid, 256, 0, stream>>>(n, n_flows, D, log_prob_all, log_j, weights, x64, clip, packed);
  return hipGetLastError();
}

// The mix again, with the draw's records written in the LIVE-POINT WIRE FORMAT itself:
// row r of `points` is one element of the importance sampler's structured dtype for D float64 parameters,
//   { x[0 .. D) f8, logP f8, logL f8, it i4, logQ f8, logW f8, logU f8 }   (8 D + 44 bytes, packed: written as
// 32-bit words), for a model with a UNIFORM prior on a box (lower / upper, `log_prior_total` = the host's own
// -sum log(upper - lower)), i.e. the fields ImportanceFlowProposal.draw fills one numpy pass at a time
// (src/nessai/proposal/importance.py:488-503, model.py:497-620):
//   logP = log(in_bounds(lower + (upper - lower) x)) + total,  logU = log_prior_unit_hypercube(x) (0 on [0, 1)^D),
//   logQ = the meta-proposal, logW = logU - logQ, logL = NaN, it = 0
// (comparisons as numpy makes them: a NaN coordinate is "inside").  `aux` row r = { log_q_all[0 .. P), log_j } -
// log_j NaN marks the rows of the reference's first filter.  The host views the read-back as the array it returns.
__global__ void __launch_bounds__(256) nsa_importance_points_kernel(long long n, int n_flows, int D,
                                                                    const float* __restrict__ log_prob_all,
                                                                    const double* __restrict__ log_j,
                                                                    const double* __restrict__ weights,
                                                                    const double* __restrict__ x64, int clip,
                                                                    const double* __restrict__ lower,
                                                                    const double* __restrict__ upper,
                                                                    double log_prior_total,
                                                                    uint32_t* __restrict__ points,
                                                                    double* __restrict__ aux) {
#pragma clang fp contract(off)
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int P = n_flows + 1;
  uint32_t* rec = points + r * (2 * (long long)D + 11);
  auto put = [&](int word, double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    rec[word] = (uint32_t)b;
    rec[word + 1] = (uint32_t)(b >> 32);
  };
  const double* xr = x64 + r * D;
  bool in_box = true, in_cube = true;
  for (int i = 0; i < D; ++i) {
    double v = xr[i];
    if (clip) v = fmin(fmax(v, 0.0), 1.0);  // (numpy.clip: NaN stays NaN)
    put(2 * i, v);
    in_cube = in_cube && !((v < 0.0) || (v >= 1.0));  // the cube prior's half-open support (src/nessai/model.py:593-601)
    const double lo = lower[i], hi = upper[i];
    const double phys = lo + (hi - lo) * v;
    in_box = in_box && !((phys < lo) || (phys > hi));
  }
  double* row = aux + r * (P + 1);
  const double lj = log_j[r];
  const double ninf = -__builtin_huge_val();
  const double log_Q = nsa_mix_row(n, r, n_flows, log_prob_all, lj, weights, row);
  row[P] = lj;
  const double log_u = in_cube ? 0.0 : ninf;
  const int w = 2 * D;
  put(w, (in_box ? 0.0 : ninf) + log_prior_total);                     // logP
  put(w + 2, __longlong_as_double(0x7ff8000000000000LL));              // logL
  rec[w + 4] = 0u;                                                     // it
  put(w + 5, log_Q);                                                   // logQ
  put(w + 7, log_u - log_Q);                                           // logW
  put(w + 9, log_u);                                                   // logU
}

extern "C" hipError_t nsa_launch_importance_points(long long n, int n_flows, int D, const float* log_prob_all,
                                                   const double* log_j, const double* weights, const double* x64,
                                                   int clip, const double* lower, const double* upper,
                                                   double log_prior_total, uint32_t* points, double* aux,
                                                   hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const unsigned grid = (unsigned)((n + 255) / 256);
  nsa_importance_points_kernel<<<grid, 256, 0, stream>>>(n, n_flows, D, log_prob_all, log_j, weights, x64, clip, lower,
                                                         upper, log_prior_total, points, aux);
  return hipGetLastError();
}

// Unit hypercube -> x' of the importance sampler, one thread per row (proposal/importance.py:219-245,
// utils/rescaling.py:290-313): xc = clip(x, eps, 1 - eps), x' = log(xc) - log1p(-xc) (float32 out: what
// the flows read), log|J| = sum_i (-log(xc_i) - log1p(-xc_i)) in float64, summed in numpy's order for a
// contiguous row (8 interleaved partial sums combined as a tree, then the remainder).
__global__ void nsa_importance_to_prime_kernel(long long n, int D, const double* __restrict__ x, double eps,
                                               float* __restrict__ xp, double* __restrict__ log_j) {
#pragma clang fp contract(off)
  const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  const double* xr = x + row * D;
  float* xo = xp + row * D;
  auto term = [&](int i) {
    double v = xr[i];
    if (eps > 0.0) v = fmin(fmax(v, eps), 1.0 - eps);
    const double a = log(v), b = log1p(-v);
    xo[i] = (float)(a - b);
    return -a - b;
  };
  double total;
  if (D < 8) {
    total = 0.0;
    for (int i = 0; i < D; ++i) total += term(i);
  } else {
    double r[8];
    for (int k = 0; k < 8; ++k) r[k] = term(k);
    int i = 8;
    for (; i + 8 <= D; i += 8)
      for (int k = 0; k < 8; ++k) r[k] += term(i + k);
    total = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < D; ++i) total += term(i);
  }
  log_j[row] = total;
}

extern "C" hipError_t nsa_launch_importance_to_prime(long long n, int D, const double* x, double eps, float* xp,
                                                     double* log_j, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const unsigned grid = (unsigned)((n + 127) / 128);
  nsa_importance_to_prime_kernel<<<grid, 128, 0, stream>>>(n, D, x, eps, xp, log_j);
  return hipGetLastError();
}

extern "C" hipError_t nsa_launch_importance_mix(long long n, int n_flows, const float* log_prob_all,
                                                const double* log_j, const double* weights, double* log_q_all,
                                                double* log_Q, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const unsigned grid = (unsigned)((n + 255) / 256);
  nsa_importance_mix_kernel<<<grid, 256, 0, stream>>>(n, n_flows, log_prob_all, log_j, weights, log_q_all, log_Q);
  return hipGetLastError();
}
