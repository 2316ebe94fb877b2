// Fused normalising-flow kernel for gfx950 (MI355X).
//
// One launch runs the WHOLE flow (every coupling layer, LU-linear and
// batch-norm transform) over a batch of rows, in either direction, with one
// of three epilogues:
//   mode APPLY    : x_out = T(x_in) (or T^-1), ld_out = log|det J|
//                   (reference: NFlow.forward / NFlow.inverse,
//                    src/nessai/flows/base.py:209-221)
//   mode LOGPROB  : forward + base log-density
//                   (NFlow.log_prob / forward_and_log_prob, base.py:236-273)
//   mode POPULATE : latent draw (Philox) or injected latents -> radius
//                   truncation -> inverse flow -> z-score inverse -> prior
//                   bounds -> log_q, in float64 where the reference's glue is
//                   float64 (FlowProposal.populate / backward_pass,
//                   src/nessai/proposal/flowproposal/flowproposal.py:345-389,
//                   431-447; truncation.py:358-365; rescale.py:263-291;
//                   model.py:512-518)
//
// Work decomposition: a workgroup of NW waves; each wave owns a 32-row tile
// whose D features live in a private LDS tile for the whole flow.  The
// conditioner MLP of every coupling layer runs on v_mfma_f32_32x32x2_f32
// with activations held in registers between layers; the weights of the
// current layer are staged global(L2) -> LDS with global_load_lds,
// double-buffered one layer ahead, and shared by all waves of the workgroup.
// The spline / affine element-wise transform consumes the conditioner output
// straight from the MFMA accumulators (the host permutes the rows of the last
// layer so that every lane receives complete parameter sets).
#pragma once
#include <type_traits>
#include <utility>

#include "nsa_device.h"

enum { NSA_MODE_APPLY = 0, NSA_MODE_LOGPROB = 1, NSA_MODE_POPULATE = 2 };

// Diagnostic builds (scripts/ablate.sh; never the shipped library): -DNSA_ABLATE=<bits> removes one
// ingredient of the kernel so that its share of the run time can be read off the clock.
//   1 stage copies after the first   2 stage barriers   4 element-wise transform
//   8 MFMAs of the hidden and last layers   16 latent draw (Philox / Box-Muller)   32 epilogue stores
//   64 the populate epilogue's per-feature loop (log q only)   128 the populate prologue (latent norm / density)
//   256 the f16 range check of the outputs   512 LU-linear / batch-norm ops (stage skipped)
//   1024 the hidden layers of the conditioner (stages skipped)   2048 the last layer's chunks (stages skipped)
#ifndef NSA_ABLATE
#define NSA_ABLATE 0
#endif
// LDS prefetch distance of the weight operands, in units of one k-step of one output tile
// (3 MFMAs = 96 matrix-pipe cycles per unit against ~100+ cycles of ds_read_b128 latency)
#ifndef NSA_PF
#define NSA_PF 2
#endif

// Diagnostic build -DNSA_TIMELINE (scripts/timeline.py; never the shipped library): every wave sums
// the shader-clock cycles it spends in each section of the tile loop (s_memtime at the section
// boundaries, fenced so that the compiler does not move work across them) and adds them to
// NsaLaunch::dbg at exit.  Slots: 0 fill + prologue, 1 layer-0 GEMM issue, 2 layer-0 tail,
// 3 hidden GEMM issue, 4 hidden tail (activation / residual), 5 chunk GEMM issue, 6 element-wise
// transform, 7 wait for the next stage's copy, 8 stage barrier, 9 epilogue, 10 other ops, 11 loop top,
// 12 issue of the next stage's copy, 13 bias / table reads in front of a GEMM;
// 14 = tiles, 15 = waves.
#ifdef NSA_TIMELINE
struct NsaTl {
  unsigned long long acc[14];
  unsigned long long last;
};
__device__ __forceinline__ void nsa_tl_mark(NsaTl& t, int slot) {
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long now = __builtin_readcyclecounter();
  __builtin_amdgcn_sched_barrier(0);
  t.acc[slot] += now - t.last;
  t.last = now;
}
#define NSA_TL(pipe, slot) nsa_tl_mark((pipe).tl, slot)
#else
#define NSA_TL(pipe, slot)
#endif

// Sub-stages along k of a register-fed GEMM with `tiles` output tiles (same rule as pack.ksplit):
// for the 128-wide conditioners (HT = 4) packed weights above 16 KiB are dealt to 2 or 4 consecutive
// stages, the accumulators live across them: 2 x 17 KiB of stage buffers instead of 2 x 64 KiB.
#ifndef NSA_WIDE_HT
#define NSA_WIDE_HT 8  // hidden tiles from which an instance is built for one wave per SIMD (512 registers)
#endif
constexpr int nsa_ksplit(int tiles, int ht) {
  if (ht < 4) return 1;  // <= 64 hidden units: the largest stage is 25 KiB, splitting only adds barriers
  int ks = 1;
  while (tiles * ht * 4096 / ks > 16384 && ks < 2 * ht) ks *= 2;
  return ks;
}

// f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N - 1>)
template <class F, int... I>
__device__ __forceinline__ void nsa_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void nsa_static_for(F&& f) {
  nsa_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

struct NsaLaunch {
  const uint32_t* image;  // device flow image (nsa_format.h)
  int direction;          // NSA_DIR_*
  int mode;               // NSA_MODE_*
  int precision;          // NSA_PREC_* (copied from the image header by the host)
  long long n;            // rows
  const float* x_in;      // [n][D] or null (POPULATE: null -> draw with Philox)
  float* x_out;           // [n][D] or null
  float* ld_out;          // [n] logabsdet (APPLY) / log_prob (LOGPROB) or null
  int* status;            // int[4]: [0] rows with a negative spline discriminant
  // POPULATE
  unsigned long long seed;
  unsigned long long row_offset;  // global index of row 0 (sharded draws)
  double radius;                  // keep ||z|| <= radius
  double latent_scale;            // sqrt(latent_temperature); 1.0 = off
  double log_j_latent;            // D * log(latent_scale)
  const double* scale;            // [D] z-score sigma
  const double* shift;            // [D] z-score mu
  const double* lower;            // [D] prior bounds
  const double* upper;            // [D]
  double log_j_rescale;           // sum log|sigma| accumulated as the reference does
  double* x64_out;                // [n][D]
  double* log_q_out;              // [n]; NaN marks a dropped row
  float* z_out;                   // [n][D] or null
  // optional separable analytic prior evaluated in the epilogue (float64):
  // log_prior = sum_i f_kind[i](x_i); kinds NSA_PRIOR_*, 3 doubles per dimension
  const int* prior_kind;          // [D] or null
  const double* prior_par;        // [D][3]
  double* log_prior_out;          // [n]
  // optional per-dimension reparameterisation table (NSA_RP_*); null = z-score everywhere
  const int* rp_kind;             // [D]
  const double* rp_par;           // [D][5] {lo, hi, rescale shift, rescale factor, offset}
  // device-side loop control of nsa_populate_passes: int64 state {accepted so far, passes executed,
  // accepted wanted, rows written, accepted in the current pass, pass limit}; a pass whose
  // predecessors already reached a limit does nothing
  const long long* stop;
  // grouped launch (APPLY / LOGPROB): gridDim.y flows of identical structure on the same
  // input; flow f reads image_list[f] and writes x_out + f*n*D, ld_out + f*n
  const uint32_t* const* image_list;  // device array of n_flows image pointers, or null
  // waves per SIMD the kernel variant is compiled for (2: <= 256 VGPRs, 3: <= 168 VGPRs); set by
  // the host from the launch geometry (nsa_api.hip)
  int occupancy;
  // importance sampler draw (POPULATE with xp_out != null; ImportanceFlowProposal.draw,
  // src/nessai/proposal/importance.py:443-487): x64_out = sigmoid(x') (the unit hypercube),
  // xp_out = x' (float32, the point the OTHER flows are evaluated at), log_q_out = log|J| of the
  // logit map at x clipped to [unit_eps, 1 - unit_eps] (utils/rescaling.py:290-313), NaN for rows
  // that fail the reference's finiteness / unit-hypercube checks
  float* xp_out;
  double unit_eps;
  // 32-row tiles per wave: 1, or 2 in the split-bf16 instances of <= 64-wide conditioners (`occupancy`
  // 2, 8-wave workgroups)
  int row_tiles;
  // the two-tiles instances: which waves cross the stage barriers between a stage's MFMAs and its vector
  // part (NsaStagePipe::skew): 0 none, 1 the upper half of the workgroup's waves, 2 every other wave
  int skew;
  // weights-resident geometry (NsaResidentPipe; host-side selector of the RES instances): the whole program
  // of the direction in LDS, `occupancy` 2 / 3 / 4 waves per SIMD in ONE workgroup per CU
  int resident;
  // POPULATE without materialised candidates (nsa_populate_passes_lean / nsa_populate_replay): with
  // x64_out == null the pass writes log_q (8 bytes per row) and nothing else - bounds, prior and log q are
  // still evaluated row by row in registers.  Only accepted rows are ever used (flowproposal.py:431-498), and
  // a row is a pure function of its Philox counter, so the accepted ones are computed AGAIN from their global
  // row indices: row_list[r] replaces row_offset + r as the counter of local row r (replay launch).
  const long long* row_list;
  // conditional flows (NSA_H_CTX > 0): [n][C] float32 context rows, constant through the flow
  // (NFlow.forward(x, context=...), src/nessai/flows/base.py:209-221)
  const float* ctx_in;
  // 1: the flow needs the extended-conditioner instances (NSA_H_CTX > 0 or NSA_H_BNW != 0; set by nsa_launch_flow)
  int ext;
  // small-batch geometry (extended instances only - nsa_launch_flow): `team` computing waves per 32-row tile, >= 1 (0:
  // a plain launch of the extended instances, every wave its own tile and its share of the stage copies).  With
  // team > 1 the waves of a team share ONE tile: every member runs the conditioner's trunk, the chunks of its last
  // layer - the parameters and the element-wise transform of 2 FPC features each - are dealt round the members; the
  // Jacobian terms meet in LDS and are summed by member 0 in the order one wave sums them, so the rows come out bit
  // for bit as in any other geometry.
  int team;
  // extended instances: buffers of the stage ring (NsaRingPipe), 2 ... 8; 0 = 2
  int ring;
  // extended instances: 1 = teams never work in rounds (NsaTeam::rounds; NESSAI_AMD_ROUNDS=0)
  int rounds_off;
  // extended instances: 1 / 2 = the LAST one / two waves of every workgroup are loaders - they request the stage copies
  // (every second piece each, when two) and compute nothing, the others request none.  Requesting a 1 KiB piece costs
  // the requesting wave 60 - 200 cycles (the CU's LDS-DMA path takes one per ~60 cycles, MI355X_MICROARCH.md): spread
  // over the computing waves of a small batch that was a quarter of their time, and the copies' landing waited for the
  // arithmetic to get out of the way.  A wave has at most 63 vector-memory operations in flight (the counter's width):
  // two loaders keep twice as many pieces on their way.
  int loader;
  // static wave priority (MI355X_MICROARCH.md, "static priority for the younger half"): 1 = the upper half of the
  // workgroup's waves runs at s_setprio 1 for the whole kernel (the default of the staged two-tiles geometry: nsa_api.hip),
  // 2 = the lower half, 0 = off; NESSAI_AMD_PRIO pins it
  int prio;
  // The lean pass with the weights folded in (log_w_out != null): instead of log_q the pass writes
  //   log_w = log_prior - log_q        (FlowProposal.compute_weights, proposal/flowproposal/base.py:1069-1098)
  // -inf for dropped rows, after the min_log_q rule (truncation.py:385-392: keep log_q > min_log_q; -inf = off), and
  // raises *max_key_out (sign-ordered int64 key of the double, nsa_populate.hip) to the launch's largest weight -
  // one atomic per wave at most.  log_prior = the separable device prior of the row when prior_kind is set,
  // else log_prior_const (a uniform prior).
  double* log_w_out;
  long long* max_key_out;
  double log_prior_const;
  double min_log_q;
#ifdef NSA_TIMELINE
  unsigned long long* dbg;
#endif
};

// x' -> x maps of the populate epilogue (inverse_reparameterise of the reference):
//   SCALE_SHIFT  x = x' * scale + shift                       (rescale.py:263-291; z-score)
//   NULL         x = x'                                       (null.py)
//   TO_BOUNDS    x = (hi - lo) * (x' - rs) / rf + lo + offset (rescale.py:537-547, 673-700)
//   TO_BOUNDS_LOGIT  the same after u = sigmoid(x'), log|J| += log u + log1p(-u)
//                (utils/rescaling.py:318-337; the "logit" entry of reparameterisations/__init__.py:62-70)
//   TO_BOUNDS_LOG    the same after u = exp(x'), log|J| += x'
//                (utils/rescaling.py:359-395: the "log-rescale" entry, reparameterisations/__init__.py:71-79)
//   INVERT       boundary inversion, inverse direction (rescale.py:585-612): u = |x'| (only negative
//                values are negated), u = 1 - u at the upper edge, x = (hi - lo) * u + lo + offset;
//                par = {lo, hi, upper ? 1 : 0, -, offset}
//   MINUS_ONE_ONE  the same parameter while no edge is detected: x = (hi - lo) * ((x' + 1) / 2) + lo + offset
//                (utils/rescaling.py:84-105)
// 7 / 8 (round 6): an angle as a point of the plane (src/nessai/reparameterisations/angle.py:153-179) - column i
// holds the x coordinate and gives the angle atan2(y, x) [mod 2 pi] / scale (par = {scale, modulo?}), column i + 1
// holds the y coordinate and gives the auxiliary radius, whose -log is the row's share of the Jacobian
enum { NSA_RP_SCALE_SHIFT = 0, NSA_RP_NULL = 1, NSA_RP_TO_BOUNDS = 2, NSA_RP_TO_BOUNDS_LOGIT = 3,
       NSA_RP_TO_BOUNDS_LOG = 4, NSA_RP_INVERT = 5, NSA_RP_MINUS_ONE_ONE = 6, NSA_RP_ANGLE = 7,
       NSA_RP_ANGLE_RADIAL = 8, NSA_RP_PAIR_H = 9, NSA_RP_PAIR_V = 10, NSA_RP_PAIR_R = 11,
       NSA_RP_DEQUANTISE = 12, NSA_RP_DEQUANTISE_LOGIT = 13, NSA_RP_DEQUANTISE_LOG = 14, NSA_RP_SS_SIGMOID = 15,
       NSA_RP_SS_EXP = 16, NSA_RP_SS_GAUSSIAN_CDF = 17, NSA_RP_INV_GAUSSIAN_CDF_SS = 18 };

enum { NSA_PRIOR_UNIFORM = 0, NSA_PRIOR_SINE = 1, NSA_PRIOR_COSINE = 2, NSA_PRIOR_POWER_LAW = 3,
       NSA_PRIOR_GAUSSIAN = 4, NSA_PRIOR_CHI = 5 };

// log density of one separable prior term; par = {a, b, log normalisation}
__device__ __forceinline__ double nsa_log_prior_term(int kind, double x, const double* par) {
  switch (kind) {
    case NSA_PRIOR_SINE: return log(sin(x)) + par[2];
    case NSA_PRIOR_COSINE: return log(cos(x)) + par[2];
    case NSA_PRIOR_POWER_LAW: return par[0] * log(x) + par[2];
    case NSA_PRIOR_GAUSSIAN: { const double t = (x - par[0]) / par[1]; return -0.5 * t * t + par[2]; }
    // chi distribution with par[0] + 1 degrees of freedom (the auxiliary radius of an angle: 2; angle.py:181-183;
    // scipy: log(2) (1 - k/2) - gammaln(k/2) + (k - 1) log x - x^2 / 2, the constants in par[2])
    case NSA_PRIOR_CHI: return par[0] * log(x) - 0.5 * x * x + par[2];
    default: return par[2];  // uniform: -log(width)
  }
}

// double-buffered weight-stage pipeline -------------------------------------

// The copy of a stage is issued in NSA_COPY_PARTS instalments: the LDS-DMA path of a CU lands about
// one 1 KiB piece per ~85 cycles, and a wave that issues a global_load_lds behind a full queue is
// held at that instruction - 25 pieces issued by all waves at the top of a stage kept every wave
// of the workgroup there for ~1.5 k cycles (scripts/timeline.py).  Part 0 goes out at the top of the
// stage, the rest after the stage's MFMAs have been issued.
#ifndef NSA_COPY_PARTS
#define NSA_COPY_PARTS 2
#endif
// which stages of a coupling the skewed waves split at the M | V boundary: 1 layer 0, 2 hidden layers, 4 chunks
#ifndef NSA_SKEW_MASK
#define NSA_SKEW_MASK 7
#endif

// 1: the stage table is read from the image by scalar loads (NsaStagePipe::load_entry; 0 = from its LDS copy).  Same
// box, interleaved twice (gpurun_out r05k): config 3's lean draw kernel 1.513 against 1.533 ms per 1e6 rows; the
// copies as LDS-DMAs the compiler does not see (NSA_GLDS_RAW) add nothing to it.
#ifndef NSA_TABLE_SCALAR
#define NSA_TABLE_SCALAR 1
#endif
#ifndef NSA_GLDS_RAW
#define NSA_GLDS_RAW 0  // 1: the stage copies as LDS-DMAs the compiler does not see (nsa_glds16_raw)
#endif
struct NsaStagePipe {
  const uint32_t* img;
  const uint32_t* table;  // {src word offset, n words} per stage (the workgroup's LDS copy)
  const uint32_t* gtable; // ... and in the image (NSA_TABLE_SCALAR)
  int n_stages;
  float* buf0;
  float* buf1;
  int cur;      // stage index within the program
  int parity;   // LDS buffer holding stage `cur`
  int wave, nw, lane;
  // table entry of the stage after `cur` (read one stage ahead: the scalar load is off the path)
  uint32_t nxt_off, nxt_words;
  // the stage being copied: its entry, and the instalment to issue next
  uint32_t cp_off, cp_words;
  int cp_part;
  // Skewed waves (the two-tiles geometry; nsa_coupling<..., SKEW = true>): a stage is {M: its MFMAs, reading the stage buffer} then {V: the vector
  // work on their results, registers and the wave's own x tile only}.  A skewed wave crosses the stage
  // barrier BETWEEN M and V instead of after V, so that from the second stage on it runs V(s-1) while the
  // plain waves of its SIMD run M(s), then M(s) while they run V(s): the matrix pipe and the vector ALUs of
  // a SIMD are busy at the same time instead of in turns.  Same barrier count, same buffers: every wave
  // finishes M(s) - its last read of stage s - before barrier s.  The two kinds of wave run two copies of the
  // coupling code (template parameter): with one copy and the barrier position tested at run time the M / V
  // boundary is a scheduling boundary for every wave, which cost the plain ones 3 %.
  bool skew;
  bool issued;  // the copy of the next stage went out right after the barrier (skewed waves)
  // NSA_PREC_F16X3: an accumulator that feeds an activation came out NaN in this tile group (an operand
  // beyond the f16 range: nsa_device.h); with NSA_F16_DETECT=2 instead the largest |hi| split (nsa_split8)
  bool nan_seen;
  u16x2 amax;
  template <int PREC, int RT, int OT>
  __device__ __forceinline__ void watch(const f32x16 (&acc)[RT][OT]) {
    if constexpr (PREC == NSA_PREC_F16X3 && NSA_F16_DETECT == 1) {
#pragma unroll
      for (int t = 0; t < RT; ++t) nan_seen |= acc[t][0][0] != acc[t][0][0];
    }
  }
#ifdef NSA_TIMELINE
  NsaTl tl;
#endif

  // pieces wave + nw * (part + parts * k)
  __device__ __forceinline__ void issue(uint32_t off, uint32_t words, float* dst, int part, int parts) {
    if ((NSA_ABLATE & 1) && cur >= 0) return;
    const uint32_t* src = img + off;
    const int pieces = (int)((words + 255) >> 8);  // the last piece may run over the stage's end (nsa_format.h)
#if NSA_GLDS_RAW
    for (int c = wave + part * nw; c < pieces; c += parts * nw) nsa_glds16_raw(src + c * 256 + lane * 4, dst + c * 256);
#else
    for (int c = wave + part * nw; c < pieces; c += parts * nw) nsa_glds16(src + c * 256 + lane * 4, dst + c * 256);
#endif
  }
  __device__ __forceinline__ void load_entry(int s) {
#if NSA_TABLE_SCALAR
    // (scalar loads from the image instead of the workgroup's LDS copy: hipcc puts an s_waitcnt vmcnt(0) of its own in
    // front of an LDS read it cannot tell from the pending LDS-DMA's destination - right behind the copy's request
    // in the stages where it does not sink the read below the MFMAs; scripts/isa_waits.py)
    const auto* e = (const __attribute__((address_space(4))) uint32_t*)gtable + 2 * s;
#if NSA_TABLE_SCALAR == 2
    nxt_off = (uint32_t)__builtin_amdgcn_readfirstlane((int)e[0]);
    nxt_words = (uint32_t)__builtin_amdgcn_readfirstlane((int)e[1]);
#else
    nxt_off = e[0];  // (no readfirstlane: the loaded words are uniform as they are, and the intrinsic waits for the load on the spot)
    nxt_words = e[1];
#endif
#else
    const uint2 e = *(const uint2*)(table + 2 * s);
    nxt_off = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.x);
    nxt_words = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.y);
#endif
  }
  // first stage of the program for this workgroup
  __device__ __forceinline__ void prime() {
    cur = -1;
    parity = 0;
    cp_part = NSA_COPY_PARTS;
    load_entry(0);
    issue(nxt_off, nxt_words, buf0, 0, 1);
    cur = 0;
    load_entry(n_stages > 1 ? 1 : 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // current stage's data; starts the copy of the next stage (wrapping to stage 0 for the
  // workgroup's next tile group)
  __device__ __forceinline__ void issue_next(int parts) {
    const int nxt = (cur + 1 == n_stages) ? 0 : cur + 1;
    cp_off = nxt_off;
    cp_words = nxt_words;
    issue(cp_off, cp_words, parity ? buf0 : buf1, 0, parts);
    cp_part = parts == 1 ? NSA_COPY_PARTS : 1;
    load_entry((nxt + 1 == n_stages) ? 0 : nxt + 1);
  }
  // (SKEW: compiled for skewed waves.  The flag is honoured either way: the ops that are not split into M | V -
  // LU-linear, batch norm - share one copy of their code, and a skewed wave reaches them with the copy of
  // their stage already sent off at the end of the coupling before.)
  template <bool SKEW = false>
  __device__ __forceinline__ const float* begin() {
    if (issued) issued = false;
    else issue_next(NSA_COPY_PARTS);
    NSA_TL(*this, 12);
    return parity ? buf1 : buf0;
  }
  // next instalment of the copy begin() started (after the stage's MFMAs; end() otherwise)
  __device__ __forceinline__ void more() {
#pragma unroll
    for (int part = 1; part < NSA_COPY_PARTS; ++part)
      if (cp_part == part) {
        issue(cp_off, cp_words, parity ? buf0 : buf1, part, NSA_COPY_PARTS);
        cp_part = part + 1;
      }
  }
  __device__ __forceinline__ void end() {
    more();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    NSA_TL(*this, 7);
    if (!(NSA_ABLATE & 2)) __syncthreads();
    NSA_TL(*this, 8);
    cur = (cur + 1 == n_stages) ? 0 : cur + 1;
    parity ^= 1;
  }
  // end of a stage's M part / of its V part: the barrier is at the first for skewed waves (which then send
  // the whole next copy off at once - its second instalment would otherwise go out right in front of the
  // wait for it), at the second for the others
  template <bool SKEW>
  __device__ __forceinline__ void end_m() {
    if constexpr (SKEW) {
      end();
      issue_next(1);
      issued = true;
    }
  }
  template <bool SKEW>
  __device__ __forceinline__ void end_v() {
    if constexpr (!SKEW) end();
  }
};

// s_waitcnt vmcnt(k) for a wave-uniform k that is only known at run time (the instruction takes an immediate):
// waits until at most min(k, 63) of the wave's vector-memory operations are outstanding
__device__ __forceinline__ void nsa_wait_vmcnt(int k) {
#define NSA_VMW(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
  // (exact up to 23 outstanding, in coarser steps - a little more waiting than asked for - above: every call site
  // carries the whole table)
  switch (k < 24 ? k : (k < 32 ? 24 : (k < 40 ? 32 : (k < 48 ? 40 : (k < 56 ? 48 : (k < 63 ? 56 : 63)))))) {
    NSA_VMW(0) NSA_VMW(1) NSA_VMW(2) NSA_VMW(3) NSA_VMW(4) NSA_VMW(5) NSA_VMW(6) NSA_VMW(7)
    NSA_VMW(8) NSA_VMW(9) NSA_VMW(10) NSA_VMW(11) NSA_VMW(12) NSA_VMW(13) NSA_VMW(14) NSA_VMW(15)
    NSA_VMW(16) NSA_VMW(17) NSA_VMW(18) NSA_VMW(19) NSA_VMW(20) NSA_VMW(21) NSA_VMW(22) NSA_VMW(23)
    NSA_VMW(24) NSA_VMW(32) NSA_VMW(40) NSA_VMW(48) NSA_VMW(56)
    default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
  }
#undef NSA_VMW
}

// Ring of stage buffers (the extended instances, template parameter CTX of the kernel): the double-buffered pipe
// with `nb` >= 2 buffers and nb - 1 stages on their way at any time.  A small batch - one tile, or one team, per
// workgroup - spends ~1 k cycles of arithmetic on a stage whose copy takes 2-3 k cycles from request to landing
// (12-17 KiB through the CU's ~12 B / cycle LDS-DMA path, MI355X_MICROARCH.md): with the copy requested ONE stage
// ahead the stage ring of config 4's 294 stages ran at the copy's latency, 0.44 ms for 5050 rows; requested nb - 1
// stages ahead it runs at the copy's throughput.  The waits are counted (s_waitcnt vmcnt(N) with the pieces of the
// later stages still outstanding, nsa_wait_vmcnt) and the barrier is a bare s_barrier: __syncthreads() drains the
// vector-memory counter while LDS-DMA writes are pending (cdna_hip_programming.md).  Vector-memory STORES complete
// out of order with the loads - the counts hold only while no store is in flight, so the kernel drains the counter
// behind the stores of its prologue and epilogue (settle()).  nb = 2 is the old pipe without instalments and skew.
template <bool RAW>
struct NsaRingPipe {
  const uint32_t* img;
  const uint32_t* table;   // {src word offset, n words} per stage (the workgroup's LDS copy; unused here)
  const uint32_t* gtable;  // ... and in the image
  int n_stages;
  float* base;    // nb buffers of buf_words floats
  int buf_words;
  int nb;
  int cur;        // stage index within the program
  int slot;       // buffer holding stage `cur`
  int nxt;        // stage index of the next copy to request ...
  int nxt_slot;   // ... and its buffer
  int wave, nw, lane;
  // who requests the copies: 0 every wave its share of the pieces; with LOADER waves in the workgroup
  // (NsaLaunch::loader of them, the last waves) 2 = this wave is loader number `ld_index` of `ld_count` and requests
  // every ld_count-th piece, 1 = this wave requests none and waits for none
  int role, ld_index, ld_count;
  // stages beyond the next one that must have landed at a barrier as well (a team in ROUNDS reads up to this many
  // stages ahead of the ring's position: nsa_coupling)
  int lookahead;
  uint32_t ent_off, ent_words;  // table entry of stage `nxt` (read when the copy before it was requested)
  // pieces THIS wave requested for the stages cur + 1 ... cur + 7 (zero beyond the ring's depth)
  int q0, q1, q2, q3, q4, q5, q6;
  bool skew, issued;  // (interface of NsaStagePipe; never set)
  bool nan_seen;
  u16x2 amax;
  template <int PREC, int RT, int OT>
  __device__ __forceinline__ void watch(const f32x16 (&acc)[RT][OT]) {
    if constexpr (PREC == NSA_PREC_F16X3 && NSA_F16_DETECT == 1) {
#pragma unroll
      for (int t = 0; t < RT; ++t) nan_seen |= acc[t][0][0] != acc[t][0][0];
    }
  }
#ifdef NSA_TIMELINE
  NsaTl tl;
#endif
  // (from the image by a SCALAR load, not from the workgroup's LDS copy of the table: the compiler orders an LDS
  // read it cannot tell from the copies' destination behind every pending LDS-DMA - an s_waitcnt vmcnt(0) of its
  // own right behind the request, i.e. the whole ring drained at every stage; the entry is needed nb - 1 stages
  // from here, so the scalar load's latency is nobody's)
  __device__ __forceinline__ void load_entry(int s) {
    // (no readfirstlane on the loaded words: a scalar load's result is uniform as it is, and the intrinsic made the
    // wave wait for the load on the spot - ~350 cycles per stage, 102 k of a 708 k-cycle trip through config 4's flow)
    const auto* e = (const __attribute__((address_space(4))) uint32_t*)gtable + 2 * s;
    ent_off = e[0];
    ent_words = e[1];
  }
  // request stage `nxt`; returns the number of pieces this wave sent for it
  __device__ __forceinline__ int request() {
    if (role == 1) return 0;  // (a loader wave does the requesting)
    const uint32_t* src = img + ent_off;
    float* dst = base + nxt_slot * buf_words;
    const int pieces = (int)((ent_words + 255) >> 8);  // the last piece may run over the stage's end (nsa_format.h)
    const int first = role == 2 ? ld_index : wave, step = role == 2 ? ld_count : nw;
    // (every workgroup of a small batch walks the SAME program at the same time, asking its XCD's L2 for the same
    // lines as its neighbours: that is a help, not a hot spot - with every workgroup starting a stage at a piece of
    // its own, config 4's 5050-row launch measured 0.40 ms against 0.33; gpurun_out r05m)
    int n = 0;
    for (int c = first; c < pieces; c += step) {
      // (RAW = false - the 256-wide instances, which spill and do not compile with the raw copy: "illegal VGPR to
      // SGPR copy" - takes the compiler's own LDS-DMA and with it the compiler's own waits for it: correct, and a
      // ring in name only)
      if constexpr (RAW) nsa_glds16_raw(src + c * 256 + lane * 4, dst + c * 256);
      else nsa_glds16(src + c * 256 + lane * 4, dst + c * 256);
      ++n;
    }
    nxt = (nxt + 1 == n_stages) ? 0 : nxt + 1;
    nxt_slot = (nxt_slot + 1 == nb) ? 0 : nxt_slot + 1;
    load_entry(nxt);
    return n;
  }
  __device__ __forceinline__ void push(int n) {  // behind the nb - 2 stages already on their way
    const int at = nb - 2;
    q0 = at == 0 ? n : q0;
    q1 = at == 1 ? n : q1;
    q2 = at == 2 ? n : q2;
    q3 = at == 3 ? n : q3;
    q4 = at == 4 ? n : q4;
    q5 = at == 5 ? n : q5;
    q6 = at == 6 ? n : q6;
  }
  __device__ __forceinline__ void barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  __device__ __forceinline__ void prime() {
    cur = 0;
    slot = 0;
    nxt = 0;
    nxt_slot = 0;
    q0 = q1 = q2 = q3 = q4 = q5 = q6 = 0;
    load_entry(0);
    (void)request();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    barrier();
    // stages 1 ... nb - 2 (begin() sends the one after them)
    const int ahead = nb - 2;
    if (ahead > 0) q0 = request();
    if (ahead > 1) q1 = request();
    if (ahead > 2) q2 = request();
    if (ahead > 3) q3 = request();
    if (ahead > 4) q4 = request();
    if (ahead > 5) q5 = request();
  }
  // the vector-memory counter is in order again (no store of this wave in flight)
  __device__ __forceinline__ void settle() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  template <bool SKEW = false>
  __device__ __forceinline__ const float* begin() {
    push(request());  // into the buffer stage cur - 1 left at the last barrier
    NSA_TL(*this, 12);
    return base + slot * buf_words;
  }
  // the buffer of the stage `off` positions ahead of the current one (off < nb)
  __device__ __forceinline__ const float* at(int off) const {
    int k = slot + off;
    k = k >= nb ? k - nb : k;
    return base + k * buf_words;
  }
  __device__ __forceinline__ void more() {}
  __device__ __forceinline__ void end() {
    // stage cur + 1 (+ lookahead) has landed once only the pieces of the stages behind it are outstanding
    if (role != 1) {
      const int L = lookahead;
      nsa_wait_vmcnt((L < 1 ? q1 : 0) + (L < 2 ? q2 : 0) + (L < 3 ? q3 : 0) + (L < 4 ? q4 : 0) + (L < 5 ? q5 : 0) +
                     (L < 6 ? q6 : 0));
    }
    NSA_TL(*this, 7);
    barrier();
    NSA_TL(*this, 8);
    q0 = q1;
    q1 = q2;
    q2 = q3;
    q3 = q4;
    q4 = q5;
    q5 = q6;
    q6 = 0;
    cur = (cur + 1 == n_stages) ? 0 : cur + 1;
    slot = (slot + 1 == nb) ? 0 : slot + 1;
  }
  template <bool SKEW>
  __device__ __forceinline__ void end_m() {}
  template <bool SKEW>
  __device__ __forceinline__ void end_v() {
    end();
  }
};

// Weights-resident pipe (template parameter RES of the kernel): every stage blob of the direction's
// program was copied to LDS once, in the kernel's prologue, and stays there for the life of the
// workgroup.  begin() is an address, end() a counter: no copy, no wait and no barrier in the tile loop,
// so the waves of a workgroup drift apart and one wave's matrix phase runs under another wave's vector
// phase by itself.  For flows whose packed program fits beside the x tiles (the reference's default
// RealNVP: ~98 KB of split-f16 stages; small spline flows) - nsa_launch_flow decides.
struct NsaResidentPipe {
  const uint32_t* table;  // {src word offset, n words} per stage (the workgroup's LDS copy)
  const float* base;      // LDS address of stage 0
  uint32_t off0;          // image word offset of stage 0
  int n_stages;
  int cur;
  uint32_t cur_rel, nxt_rel;  // word offsets of stage `cur` / the stage after it, relative to stage 0
  bool nan_seen;
  u16x2 amax;
  bool skew;  // never set: there is no barrier to cross early
  template <int PREC, int RT, int OT>
  __device__ __forceinline__ void watch(const f32x16 (&acc)[RT][OT]) {
    if constexpr (PREC == NSA_PREC_F16X3 && NSA_F16_DETECT == 1) {
#pragma unroll
      for (int t = 0; t < RT; ++t) nan_seen |= acc[t][0][0] != acc[t][0][0];
    }
  }
#ifdef NSA_TIMELINE
  NsaTl tl;
#endif
  __device__ __forceinline__ uint32_t rel_of(int s) const {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)(table[2 * s] - off0));
  }
  __device__ __forceinline__ void prime() {
    cur = 0;
    cur_rel = 0;
    nxt_rel = rel_of(n_stages > 1 ? 1 : 0);
  }
  template <bool SKEW = false>
  __device__ __forceinline__ const float* begin() {
    return base + cur_rel;
  }
  __device__ __forceinline__ void more() {}
  // the offset of the stage after next is read a whole stage before it is needed
  __device__ __forceinline__ void end() {
    cur = (cur + 1 == n_stages) ? 0 : cur + 1;
    cur_rel = nxt_rel;
    nxt_rel = rel_of((cur + 1 == n_stages) ? 0 : cur + 1);
  }
  template <bool SKEW>
  __device__ __forceinline__ void end_m() {}
  template <bool SKEW>
  __device__ __forceinline__ void end_v() {
    end();
  }
};

// dense layers on MFMA --------------------------------------------------------

// acc[ot] = bias; lane reads its 16 biases for each output tile
template <int OT>
__device__ __forceinline__ void nsa_load_bias(const float* bias, int half, f32x16 (&acc)[OT]) {
#pragma unroll
  for (int ot = 0; ot < OT; ++ot) {
    const f32x4* b = (const f32x4*)(bias + ot * 32 + half * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 t = b[q];
      acc[ot][4 * q + 0] = t[0];
      acc[ot][4 * q + 1] = t[1];
      acc[ot][4 * q + 2] = t[2];
      acc[ot][4 * q + 3] = t[3];
    }
  }
}

// Input from registers (D layout of the previous layer): in[16*HT] -> OT tiles.
// W layout of the sub-stage: [OT][NS4][64 lanes][4]; it covers the step groups [S0, S0 + NS4) of `in`.
// RT row tiles per wave share every weight operand read from LDS.
template <int HT, int OT, int RT, int S0 = 0, int NS4 = 4 * HT>
__device__ __forceinline__ void nsa_dense_regs(const float* w, const float (&in)[RT][16 * HT], int lane,
                                               f32x16 (&acc)[RT][OT]) {
#pragma unroll
  for (int s4 = 0; s4 < NS4; ++s4) {
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
      const f32x4 a = *(const f32x4*)(w + ((ot * NS4 + s4) * 64 + lane) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[t][ot] = nsa_mfma(a[e], in[t][4 * (S0 + s4) + e], acc[t][ot]);
    }
  }
}

// Input gathered from the wave's x tile through a column table
// (tab[2*s + half] = byte offset of the physical column in the row; unused entries: the row's zero cell),
// optional pre-shift.
template <int OT, bool PRE, int RT>
__device__ __forceinline__ void nsa_dense_tile(const float* w, const int* tab, const float* pre,
                                               float* const (&xrow)[RT], int n_steps, int lane, int half,
                                               f32x16 (&acc)[RT][OT]) {
  const int n_s4 = n_steps >> 2;
  for (int s4 = 0; s4 < n_s4; ++s4) {
    // the 8 table entries of this step group in two 16-byte reads (a lane uses those of its half),
    // then 4 independent tile reads - no per-element branch, no read -> wait -> read chains
    const int4 c0 = *(const int4*)(tab + 8 * s4), c1 = *(const int4*)(tab + 8 * s4 + 4);
    const int col[4] = {half ? c0.y : c0.x, half ? c0.w : c0.z, half ? c1.y : c1.x, half ? c1.w : c1.z};  // byte offsets
    float sh[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (PRE) {
      const f32x4 p0 = *(const f32x4*)(pre + 8 * s4), p1 = *(const f32x4*)(pre + 8 * s4 + 4);
      sh[0] = half ? p0[1] : p0[0];
      sh[1] = half ? p0[3] : p0[2];
      sh[2] = half ? p1[1] : p1[0];
      sh[3] = half ? p1[3] : p1[2];
    }
    float b[RT][4];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) b[t][e] = *(const float*)((const char*)xrow[t] + col[e]);
    if (PRE) {
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) b[t][e] = b[t][e] - sh[e];
    }
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
      const f32x4 a = *(const f32x4*)(w + ((ot * n_s4 + s4) * 64 + lane) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[t][ot] = nsa_mfma(a[e], b[t][e], acc[t][ot]);
    }
  }
}

template <int HT>
__device__ __forceinline__ void nsa_acc_to_regs(const f32x16 (&acc)[HT], float (&out)[16 * HT]) {
#pragma unroll
  for (int ot = 0; ot < HT; ++ot)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[16 * ot + r] = acc[ot][r];
}

// ---- split 16-bit (NSA_PREC_BF16X3 / NSA_PREC_F16X3) variants --------------------

// activations of one layer, split once and reused for every output tile
template <int HT>
struct NsaSplit {
  nsa_op8 hi[2 * HT], lo[2 * HT];
};

template <int PREC, int HT>
__device__ __forceinline__ void nsa_split_regs(const float (&in)[16 * HT], NsaSplit<HT>& sp, u16x2& amax) {
#pragma unroll
  for (int s = 0; s < 2 * HT; ++s) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = in[8 * s + e];
    nsa_split8<PREC>(v, sp.hi[s], sp.lo[s], amax);
  }
}

// W layout of the sub-stage: [OT][NS steps][part][64 lanes][8 bf16] (4 words per lane and part); it
// covers the k steps [S0, S0 + NS) of the split activations.  The A operands (LDS) of unit
// u + NSA_PF are fetched before the MFMAs of unit u are issued, so that the LDS latency runs under the
// matrix pipe instead of in front of it.
template <int PREC, int HT, int OT, int RT, int S0 = 0, int NS = 2 * HT>
__device__ __forceinline__ void nsa_dense_regs16(const float* w, const NsaSplit<HT> (&sp)[RT], int lane,
                                                 f32x16 (&acc)[RT][OT]) {
  constexpr int U = NS * OT;  // unit u = s * OT + ot
  nsa_op8 ah[NSA_PF + 1], al[NSA_PF + 1];
#pragma unroll
  for (int u = 0; u < NSA_PF && u < U; ++u) {
    ah[u] = *(const nsa_op8*)(w + ((((u % OT) * NS + u / OT) * 2 + 0) * 64 + lane) * 4);
    al[u] = *(const nsa_op8*)(w + ((((u % OT) * NS + u / OT) * 2 + 1) * 64 + lane) * 4);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int s = u / OT, ot = u % OT;
    if (u + NSA_PF < U) {
      const int s2 = (u + NSA_PF) / OT, ot2 = (u + NSA_PF) % OT;
      ah[(u + NSA_PF) % (NSA_PF + 1)] = *(const nsa_op8*)(w + (((ot2 * NS + s2) * 2 + 0) * 64 + lane) * 4);
      al[(u + NSA_PF) % (NSA_PF + 1)] = *(const nsa_op8*)(w + (((ot2 * NS + s2) * 2 + 1) * 64 + lane) * 4);
    }
    const nsa_op8 a_h = ah[u % (NSA_PF + 1)], a_l = al[u % (NSA_PF + 1)];
    // the row tiles of the wave take turns product by product: independent accumulators back to back
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t][ot] = nsa_mfma16<PREC>(a_l, sp[t].hi[S0 + s], acc[t][ot]);
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t][ot] = nsa_mfma16<PREC>(a_h, sp[t].lo[S0 + s], acc[t][ot]);
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t][ot] = nsa_mfma16<PREC>(a_h, sp[t].hi[S0 + s], acc[t][ot]);
  }
}

// input gathered from the x tile: tab[16*s + 8*half + e]
template <int PREC, int OT, bool PRE, int RT>
__device__ __forceinline__ void nsa_dense_tile16(const float* w, const int* tab, const float* pre,
                                                 float* const (&xrow)[RT], int n_steps, int lane, int half,
                                                 f32x16 (&acc)[RT][OT], u16x2& amax) {
  for (int s = 0; s < n_steps; ++s) {
    // 8 table entries (and pre-shifts) in two 16-byte reads, then 8 independent tile reads: with a
    // branch per element the compiler issued read -> wait -> read -> wait sixteen times in a row
    const int base = 16 * s + 8 * half;
    const int4 c0 = *(const int4*)(tab + base), c1 = *(const int4*)(tab + base + 4);
    const int col[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    float sh[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (PRE) {
      const f32x4 p0 = *(const f32x4*)(pre + base), p1 = *(const f32x4*)(pre + base + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sh[e] = p0[e];
        sh[4 + e] = p1[e];
      }
    }
    nsa_op8 bh[RT], bl[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = *(const float*)((const char*)xrow[t] + col[e]);  // byte offsets
      if (PRE) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] - sh[e];
      }
      nsa_split8<PREC>(v, bh[t], bl[t], amax);
    }
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
      const nsa_op8 ah = *(const nsa_op8*)(w + (((ot * n_steps + s) * 2 + 0) * 64 + lane) * 4);
      const nsa_op8 al = *(const nsa_op8*)(w + (((ot * n_steps + s) * 2 + 1) * 64 + lane) * 4);
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[t][ot] = nsa_mfma_x3<PREC>(ah, al, bh[t], bl[t], acc[t][ot]);
    }
  }
}

// activation over a whole register array; the (wave-uniform) kind is tested
// once, not per element
#ifndef NSA_ACT_VARIANT
#define NSA_ACT_VARIANT 0  // (A/B build of round 6: 3 = the activation kind branched on once around the residual blocks)
#endif
// ACT_KNOWN: 0 = `act` decides at run time (three-way branch), 1 = the caller is on a path where act == relu,
// 2 = ... where act != relu (tanh / silu)
template <int N, int ACT_KNOWN = 0>
__device__ __forceinline__ void nsa_act_regs(float (&a)[N], int act) {
  if constexpr (ACT_KNOWN == 1) {
#pragma unroll
    for (int i = 0; i < N; ++i) a[i] = nsa_relu(a[i]);
  } else if constexpr (ACT_KNOWN == 2) {
    if (act == NSA_ACT_TANH) {
#pragma unroll
      for (int i = 0; i < N; ++i) a[i] = nsa_tanh_fast(a[i]);
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i) a[i] = nsa_silu_fast(a[i]);
    }
  } else if (act == NSA_ACT_RELU) {
#pragma unroll
    for (int i = 0; i < N; ++i) a[i] = nsa_relu(a[i]);
  } else if (act == NSA_ACT_TANH) {
#pragma unroll
    for (int i = 0; i < N; ++i) a[i] = nsa_tanh_fast(a[i]);
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) a[i] = nsa_silu_fast(a[i]);
  }
}

// element-wise transform of the features a lane owns in one chunk ------------
//
// A chunk holds TP::FPC features per lane, each transformed in TP::NPH phases;
// global phase index PH = feature * NPH + phase.

template <class TP, bool INV, int PH, int NV, bool REC = false>
__device__ __forceinline__ void nsa_run_phase(typename TP::State (&st)[TP::FPC], const float (&v)[NV],
                                              const int (&cols)[TP::FPC], float* xrow,
                                              int dummy_col, const NsaTransformConsts& tc,
                                              float& logdet, int& n_bad, float (&fac)[TP::FPC]) {
  // (REC callers - nsa_run_phases - pass the feature's Jacobian term out through fac[] instead of adding it)
  constexpr int Q = PH / TP::NPH, P = PH % TP::NPH;
  // unused feature slots (col < 0) run on the tile's spare column and add nothing
  const int col = cols[Q];
  const bool live = col >= 0;
  const int c = live ? col : dummy_col;
  float x_in = 0.0f;
  if constexpr (P == 0) x_in = xrow[c];
  float y = 0.0f, ld = 0.0f;
  bool bad = false;
  TP::template phase<P, Q * TP::NP, INV, NV>(st[Q], v, x_in, tc, y, ld, bad);
  if constexpr (P == TP::NPH - 1) {
    xrow[c] = y;
    if constexpr (TP::LD_FACTOR) fac[Q] = live ? ld : 1.0f;
    else if constexpr (REC) fac[Q] = live ? ld : 0.0f;
    else logdet += live ? ld : 0.0f;
    n_bad += (live && bad) ? 1 : 0;
  }
}

template <class TP, bool INV, int LO, int NV, int... I>
__device__ __forceinline__ void nsa_run_phases(typename TP::State (&st)[TP::FPC], const float (&v)[NV],
                                               const int (&cols)[TP::FPC], float* xrow,
                                               int dummy_col, const NsaTransformConsts& tc,
                                               float& logdet, int& n_bad,
                                               std::integer_sequence<int, I...>) {
  // LD_FACTOR transforms hand back Jacobian factors: multiplied in groups of four, one logarithm per group -
  // while a product of four cannot leave the float32 range (tc.ld_group4: the floor b of the scale activation
  // sigmoid(raw + a) + b is >= 1e-9, so a factor lies in [b, 1 + b] or its reciprocal; with b = 0, a plain
  // sigmoid, four saturated scales of 2^-64 would underflow to 0 and log|det J| to -inf where the reference's
  // sum of logarithms is finite): otherwise one logarithm per feature
  float fac[TP::FPC];
#pragma unroll
  for (int q = 0; q < TP::FPC; ++q) fac[q] = 1.0f;
  (nsa_run_phase<TP, INV, LO + I, NV>(st, v, cols, xrow, dummy_col, tc, logdet, n_bad, fac), ...);
  if constexpr (TP::LD_FACTOR) {
    float l2 = 0.0f;
    if (tc.ld_group4) {
#pragma unroll
      for (int g = 0; g < (TP::FPC + 3) / 4; ++g) {
        float pr = fac[4 * g];
#pragma unroll
        for (int q = 4 * g + 1; q < 4 * g + 4 && q < TP::FPC; ++q) pr *= fac[q];
        l2 += nsa_log2(pr);
      }
    } else {
#pragma unroll
      for (int q = 0; q < TP::FPC; ++q) l2 += nsa_log2(fac[q]);
    }
    logdet = fmaf(0.6931471805599453f, l2, logdet);
  }
}

// The same, with the chunk's Jacobian terms handed out instead of added (the extended instances: team geometry):
// rec[q] = the term of feature q (added to the running log|det J| one by one, in this order, by nsa_add_terms), or,
// for LD_FACTOR transforms, rec[0] = the chunk's sum of base-2 logarithms.
template <class TP, bool INV, int NV, int... I>
__device__ __forceinline__ void nsa_run_phases_rec(typename TP::State (&st)[TP::FPC], const float (&v)[NV],
                                                   const int (&cols)[TP::FPC], float* xrow, int dummy_col,
                                                   const NsaTransformConsts& tc, float (&rec)[TP::FPC], int& n_bad,
                                                   std::integer_sequence<int, I...>) {
  float unused = 0.0f;
#pragma unroll
  for (int q = 0; q < TP::FPC; ++q) rec[q] = 1.0f;
  (nsa_run_phase<TP, INV, I, NV, true>(st, v, cols, xrow, dummy_col, tc, unused, n_bad, rec), ...);
  if constexpr (TP::LD_FACTOR) {
    float l2 = 0.0f;
    if (tc.ld_group4) {
#pragma unroll
      for (int g = 0; g < (TP::FPC + 3) / 4; ++g) {
        float pr = rec[4 * g];
#pragma unroll
        for (int q = 4 * g + 1; q < 4 * g + 4 && q < TP::FPC; ++q) pr *= rec[q];
        l2 += nsa_log2(pr);
      }
    } else {
#pragma unroll
      for (int q = 0; q < TP::FPC; ++q) l2 += nsa_log2(rec[q]);
    }
    rec[0] = l2;
  }
}
// ... and the additions nsa_run_phases makes with them
template <class TP>
__device__ __forceinline__ void nsa_add_terms(const float (&rec)[TP::FPC], float& logdet) {
  if constexpr (TP::LD_FACTOR) {
    logdet = fmaf(0.6931471805599453f, rec[0], logdet);
  } else {
#pragma unroll
    for (int q = 0; q < TP::FPC; ++q) logdet += rec[q];
  }
}

// the wave's place in its team (NsaLaunch::team), and the team's LDS record of Jacobian terms: [slot][64 lanes]
struct NsaTeam {
  int size, member;
  float* rec;
  // 1: the chunks of the last layer in ROUNDS - the members work on `size` consecutive chunks at the same time, each
  // reading its chunk's stages member * KS positions ahead of the ring (loader waves keep that far ahead:
  // NsaRingPipe::lookahead); 0: one chunk - one member - per stage barrier
  int rounds;
};

// MFMAs of sub-stage J of one chunk of the last conditioner layer (-> acc; cols and the bias come
// with sub-stage 0).
template <int HT, class TP, int PREC, int J, int RT, class PIPE>
__device__ __forceinline__ void nsa_chunk_gemm([[maybe_unused]] PIPE& pipe, const float* stg,
                                               const float (&h)[RT][16 * HT], const NsaSplit<HT> (&hs)[RT],
                                               int lane, int half, f32x16 (&acc)[RT][TP::C],
                                               int (&cols)[TP::FPC]) {
  constexpr int TABW = (2 * TP::FPC + 3) & ~3;
  constexpr int KS = nsa_ksplit(TP::C, HT);
  const float* w = stg;
  if constexpr (J == 0) {
    const int* tr_tab = (const int*)stg;
    const float* bias = stg + TABW;
    w = bias + 32 * TP::C;
#pragma unroll
    for (int q = 0; q < TP::FPC; ++q) cols[q] = tr_tab[half * TP::FPC + q];
    nsa_load_bias<TP::C>(bias, half, acc[0]);
#pragma unroll
    for (int t = 1; t < RT; ++t)
#pragma unroll
      for (int ot = 0; ot < TP::C; ++ot) acc[t][ot] = acc[0][ot];
    NSA_TL(pipe, 13);
  }
  if constexpr (PREC != NSA_PREC_F32) nsa_dense_regs16<PREC, HT, TP::C, RT, J * (2 * HT / KS), 2 * HT / KS>(w, hs, lane, acc);
  else nsa_dense_regs<HT, TP::C, RT, J * (4 * HT / KS), 4 * HT / KS>(w, h, lane, acc);
}

// one coupling layer (over the RT row tiles of the wave) ---------------------------

// sigmoid on the bare transcendental units (the gate of a conditional residual block)
__device__ __forceinline__ float nsa_sigmoid_fast(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}

// CTX: a conditional flow (nflows ResidualNet with context_features: src/nessai/flows/nsf.py:85-95,
// realnvp.py:136-146) - layer 0 gathers the context columns of the x tile behind the identity features (its table
// says so: nothing to do here), and every residual block ends in
//   temps = glu(cat(temps, context_layer(context)))  =  temps * sigmoid(Wc context + bc)
// one more gathered GEMM (stage "context gate", nsa_format.h) and a sigmoid per hidden value.
template <int HT, class TP, bool INV, int PREC, int RT, bool SKEW = false, bool CTX = false, class PIPE>
__device__ __forceinline__ void nsa_coupling(PIPE& pipe, const uint32_t* op, float* const (&xrow)[RT],
                                             int dummy_col, int lane, int half, int act, int net,
                                             int n_layers, const NsaTransformConsts& tc,
                                             float (&logdet)[RT], int (&n_bad)[RT], bool bn_within = false,
                                             NsaTeam team = NsaTeam{1, 0, nullptr, 0}) {
  constexpr bool BF = (PREC != NSA_PREC_F32);  // split 16-bit operands (bf16 or f16 pairs)
  const int n_steps0 = __builtin_amdgcn_readfirstlane((int)op[1]);
  const int n_chunks = __builtin_amdgcn_readfirstlane((int)op[2]);
  float h[RT][16 * HT];

  // one H -> H layer from registers: out = W in + b
  // (nsa_ksplit(HT, HT) stages along k; the caller closes the LAST one after its own epilogue)
  auto hidden_layer = [&](const float (&in)[RT][16 * HT], f32x16 (&acc)[RT][HT]) {
    constexpr int KS = nsa_ksplit(HT, HT);
    NsaSplit<HT> sp[RT];
    if constexpr (BF) {
#pragma unroll
      for (int t = 0; t < RT; ++t) nsa_split_regs<PREC, HT>(in[t], sp[t], pipe.amax);
    }
    nsa_static_for<KS>([&](auto jc) {
      constexpr int J = decltype(jc)::value;
      const float* st = pipe.template begin<SKEW>();
      const float* w = st;
      if constexpr (J == 0) {
        nsa_load_bias<HT>(st, half, acc[0]);
#pragma unroll
        for (int t = 1; t < RT; ++t)
#pragma unroll
          for (int ot = 0; ot < HT; ++ot) acc[t][ot] = acc[0][ot];
        w = st + 32 * HT;
        NSA_TL(pipe, 13);
      }
      if constexpr (BF) nsa_dense_regs16<PREC, HT, HT, RT, J * (2 * HT / KS), 2 * HT / KS>(w, sp, lane, acc);
      else nsa_dense_regs<HT, HT, RT, J * (4 * HT / KS), 4 * HT / KS>(w, in, lane, acc);
      pipe.more();
      NSA_TL(pipe, 3);
      if constexpr (J + 1 < KS) pipe.end();
      else pipe.template end_m<SKEW && (NSA_SKEW_MASK & 2) != 0>();
    });
    pipe.template watch<PREC>(acc);
  };

  // layer 0: identity features (x tile) -> hidden
  {
    const float* st = pipe.template begin<SKEW>();
    const int* id_tab = (const int*)st;
    const float* bias = st + (BF ? 16 : 2) * n_steps0;
    const float* w = bias + 32 * HT;
    f32x16 acc[RT][HT];
    nsa_load_bias<HT>(bias, half, acc[0]);
#pragma unroll
    for (int t = 1; t < RT; ++t)
#pragma unroll
      for (int ot = 0; ot < HT; ++ot) acc[t][ot] = acc[0][ot];
    if constexpr (BF) nsa_dense_tile16<PREC, HT, false, RT>(w, id_tab, nullptr, xrow, n_steps0, lane, half, acc, pipe.amax);
    else nsa_dense_tile<HT, false, RT>(w, id_tab, nullptr, xrow, n_steps0, lane, half, acc);
    pipe.more();
    NSA_TL(pipe, 1);
    pipe.template end_m<SKEW && (NSA_SKEW_MASK & 1) != 0>();
    pipe.template watch<PREC>(acc);
#pragma unroll
    for (int t = 0; t < RT; ++t) nsa_acc_to_regs<HT>(acc[t], h[t]);
    NSA_TL(pipe, 2);
    pipe.template end_v<SKEW && (NSA_SKEW_MASK & 1) != 0>();
  }

  if constexpr ((NSA_ABLATE & 1024) != 0) {
    const int skip = (net == NSA_NET_RESNET ? 2 * n_layers : n_layers - 1) * nsa_ksplit(HT, HT);
    for (int k = 0; k < skip; ++k) {
      (void)pipe.template begin<false>();
      pipe.end();
    }
  } else if (net == NSA_NET_RESNET) {
    // pre-activation residual blocks: h += W2 act(W1 act(h) + b1) + b2
    // (Round 6: the stack of blocks is a generic lambda over the activation class.  As plain inline code - the state
    // of round 5 - config 2's kernel took 0.36-0.37 ms per 1e6 rows; it had taken 0.28 before round 4's last commit
    // replaced tanhf / expf in nsa_act_regs by the short inline forms, which changed how hipcc lays out the three
    // activation paths of EVERY flow, relu included.  Inside the lambda the same source compiles to 0.28 ms again
    // (act0 in profiles/r06_c2_activation_paths_ab.jsonl); branching on the kind ONCE around the stack
    // (NSA_ACT_VARIANT 3: the relu path one straight stream per block) gives the same 0.27-0.28 for config 2 and costs
    // config 3's kernel 1 % (1.557 against 1.543 ms), so the run-time branch stays.  A compiler layout effect, pinned
    // by nothing but this measurement: re-measure config 2 when this function or hipcc changes.)
    auto blocks = [&](auto known) {
    constexpr int AK = decltype(known)::value;
    for (int l = 0; l < n_layers; ++l) {
      float tmp[RT][16 * HT];
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int i = 0; i < 16 * HT; ++i) tmp[t][i] = h[t][i];
      if constexpr (CTX) {
        if (bn_within) {
          // batch_norm_within_layers, eval mode: the block's FIRST batch norm acts on the residual stream in front
          // of the activation - a per-feature a h + c from a stage of its own ({a[32 HT], c[32 HT]}, the bias
          // layout); the second one follows linear 0 and is folded into its weights and bias by the packer
          const float* st = pipe.template begin<false>();
          f32x16 sa[HT], sc[HT];
          nsa_load_bias<HT>(st, half, sa);
          nsa_load_bias<HT>(st + 32 * HT, half, sc);
#pragma unroll
          for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int ot = 0; ot < HT; ++ot)
#pragma unroll
              for (int r = 0; r < 16; ++r) tmp[t][16 * ot + r] = fmaf(tmp[t][16 * ot + r], sa[ot][r], sc[ot][r]);
          pipe.end();
        }
      }
#pragma unroll
      for (int t = 0; t < RT; ++t) nsa_act_regs<16 * HT, AK>(tmp[t], act);
      {
        f32x16 acc[RT][HT];
        hidden_layer(tmp, acc);
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          nsa_acc_to_regs<HT>(acc[t], tmp[t]);
          nsa_act_regs<16 * HT, AK>(tmp[t], act);
        }
        NSA_TL(pipe, 4);
        pipe.template end_v<SKEW && (NSA_SKEW_MASK & 2) != 0>();
      }
      {
        f32x16 acc[RT][HT];
        hidden_layer(tmp, acc);
        const int n_steps_c = CTX ? __builtin_amdgcn_readfirstlane((int)op[3]) : 0;
        if (CTX && n_steps_c > 0) {
          // the block's gate: g = Wc context + bc, gathered from the context columns of the x tile
          static_assert(!(CTX && SKEW), "conditional flows run one tile per wave, unskewed");
          pipe.end();
          const float* st = pipe.template begin<false>();
          const int* c_tab = (const int*)st;
          const float* gbias = st + (BF ? 16 : 2) * n_steps_c;
          const float* gw = gbias + 32 * HT;
          f32x16 g[RT][HT];
          nsa_load_bias<HT>(gbias, half, g[0]);
#pragma unroll
          for (int t = 1; t < RT; ++t)
#pragma unroll
            for (int ot = 0; ot < HT; ++ot) g[t][ot] = g[0][ot];
          if constexpr (BF) nsa_dense_tile16<PREC, HT, false, RT>(gw, c_tab, nullptr, xrow, n_steps_c, lane, half, g, pipe.amax);
          else nsa_dense_tile<HT, false, RT>(gw, c_tab, nullptr, xrow, n_steps_c, lane, half, g);
          pipe.more();
          pipe.template watch<PREC>(g);
#pragma unroll
          for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int ot = 0; ot < HT; ++ot)
#pragma unroll
              for (int r = 0; r < 16; ++r)
                h[t][16 * ot + r] = fmaf(acc[t][ot][r], nsa_sigmoid_fast(g[t][ot][r]), h[t][16 * ot + r]);
          pipe.end();
        } else {
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
          for (int ot = 0; ot < HT; ++ot)
#pragma unroll
            for (int r = 0; r < 16; ++r) h[t][16 * ot + r] = h[t][16 * ot + r] + acc[t][ot][r];
        NSA_TL(pipe, 4);
        pipe.template end_v<SKEW && (NSA_SKEW_MASK & 2) != 0>();
        }
      }
    }
    };
#if NSA_ACT_VARIANT == 3
    if (act == NSA_ACT_RELU) blocks(std::integral_constant<int, 1>{});
    else blocks(std::integral_constant<int, 2>{});
#else
    blocks(std::integral_constant<int, 0>{});
#endif
  } else {
    // plain MLP (src/nessai/flows/nets.py:110-124): act after every layer but the last
#pragma unroll
    for (int t = 0; t < RT; ++t) nsa_act_regs<16 * HT>(h[t], act);
    for (int l = 1; l < n_layers; ++l) {
      f32x16 acc[RT][HT];
      hidden_layer(h, acc);
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        nsa_acc_to_regs<HT>(acc[t], h[t]);
        nsa_act_regs<16 * HT>(h[t], act);
      }
      NSA_TL(pipe, 4);
      pipe.template end_v<SKEW && (NSA_SKEW_MASK & 2) != 0>();
    }
  }

  // Last layer in chunks of TP::C tiles: every lane gets TP::FPC complete parameter sets per
  // chunk and transforms those features in place (registers + the lane's own x-tile cells) before
  // the stage barrier, so that the next stage's copy has the whole element-wise part to land.
  // (A software-pipelined version - the MFMAs of chunk c+1 interleaved with the element-wise
  // phases of chunk c - needed 48 more registers and was 0.7 % slower in split-bf16, 1.7 % in f32.)
  NsaSplit<HT> hs[RT];
  if constexpr (BF) {
#pragma unroll
    for (int t = 0; t < RT; ++t) nsa_split_regs<PREC, HT>(h[t], hs[t], pipe.amax);
  }
  // (team geometry: sequentially, chunk c belongs to member c mod size and the others only keep the stage ring turning;
  // in rounds, iteration `it` is the round of chunks it * size ... and this member's chunk lies member * KS stages
  // ahead of the ring's position - the positions of the other members' chunks pass as idle stages behind it)
  constexpr int KSC = nsa_ksplit(TP::C, HT);
  const bool in_rounds = CTX && team.size > 1 && team.rounds != 0;
  const int n_iter = in_rounds ? (n_chunks + team.size - 1) / team.size : n_chunks;
  for (int it = 0; it < n_iter; ++it) {
    const int c = in_rounds ? it * team.size + team.member : it;
    int idle_after = 0, ahead = 0;
    bool active = true;
    if constexpr (CTX) {
      if (in_rounds) {
        const int left = n_chunks - it * team.size;
        idle_after = ((left < team.size ? left : team.size) - 1) * KSC;
        ahead = team.member * KSC;
        active = c < n_chunks;
      } else if (team.size > 1) {
        active = (c & (team.size - 1)) == team.member;
      }
    }
    if constexpr ((NSA_ABLATE & 2048) != 0) {
      for (int k = 0; k < nsa_ksplit(TP::C, HT); ++k) {
        (void)pipe.template begin<false>();
        pipe.end();
      }
      logdet[0] += h[0][c & 15];  // (keeps the conditioner alive)
      continue;
    }
    if constexpr (CTX) {
      if (!active) {
        for (int k = 0; k < KSC + idle_after; ++k) {
          (void)pipe.template begin<false>();
          pipe.end();
        }
        continue;
      }
    }
    f32x16 acc[RT][TP::C];
    int cols[TP::FPC];
    nsa_static_for<nsa_ksplit(TP::C, HT)>([&](auto jc) {
      constexpr int J = decltype(jc)::value;
      const float* st = pipe.template begin<SKEW>();
      if constexpr (CTX) st = pipe.at(ahead);
      nsa_chunk_gemm<HT, TP, PREC, J, RT>(pipe, st, h, hs, lane, half, acc, cols);
      pipe.more();
      NSA_TL(pipe, 5);
      if constexpr (J + 1 < nsa_ksplit(TP::C, HT)) pipe.end();
      else pipe.template end_m<SKEW && (NSA_SKEW_MASK & 4) != 0>();
    });
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      float v[16 * TP::C];
#pragma unroll
      for (int q = 0; q < TP::C; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[16 * q + r] = acc[t][q][r];
      if constexpr ((NSA_ABLATE & 4) != 0) {
#pragma unroll
        for (int i = 0; i < 16 * TP::C; ++i) asm volatile("" ::"v"(v[i]));
      } else if constexpr (CTX) {
        // (same arithmetic, the Jacobian terms by themselves: added right here by a wave that has its tile to
        // itself, left in the team's record otherwise - summed below in chunk order)
        typename TP::State sts[TP::FPC];
        float rec[TP::FPC];
        nsa_run_phases_rec<TP, INV>(sts, v, cols, xrow[t], dummy_col, tc, rec, n_bad[t],
                                    std::make_integer_sequence<int, TP::NPH * TP::FPC>{});
        if (team.size == 1) {
          nsa_add_terms<TP>(rec, logdet[t]);
        } else {
          float* const slot = team.rec + (c * TP::FPC) * 64 + lane;
#pragma unroll
          for (int q = 0; q < (TP::LD_FACTOR ? 1 : TP::FPC); ++q) slot[q * 64] = rec[q];
        }
      } else {
        typename TP::State sts[TP::FPC];
        nsa_run_phases<TP, INV, 0>(sts, v, cols, xrow[t], dummy_col, tc, logdet[t], n_bad[t],
                                   std::make_integer_sequence<int, TP::NPH * TP::FPC>{});
      }
    }
    NSA_TL(pipe, 6);
    pipe.template end_v<SKEW && (NSA_SKEW_MASK & 4) != 0>();
    if constexpr (CTX) {
      for (int k = 0; k < idle_after; ++k) {  // (the stages of the round's other chunks)
        (void)pipe.template begin<false>();
        pipe.end();
      }
    }
  }
  if constexpr (CTX) {
    // (behind the barrier of the last chunk's stage: every member's terms are in the record; the next coupling
    // writes it again several barriers from here)
    if (team.size > 1 && team.member == 0) {
      for (int c = 0; c < n_chunks; ++c) {
        float rec[TP::FPC];
        const float* const slot = team.rec + (c * TP::FPC) * 64 + lane;
#pragma unroll
        for (int q = 0; q < (TP::LD_FACTOR ? 1 : TP::FPC); ++q) rec[q] = slot[q * 64];
        nsa_add_terms<TP>(rec, logdet[0]);
      }
    }
  }
}

// an int the optimiser cannot trace back: keeps the per-lane 64-bit addresses of the tile copy
// loops from being hoisted out of the persistent loop (they were spilled to scratch there)
__device__ __forceinline__ int nsa_opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

// One output tile of a dense op back into the wave's x tile: accumulator register r = 4 q + e of a lane holds
// output feature first + 8 q + 4 half + e.  The features end at D anywhere inside the tile; a test per element
// is a per-LANE predicate (`half`) - sixteen exec-mask round trips, each with its mask kept in (spilled)
// scalar registers.  Groups of eight features that lie wholly below D (a wave-uniform test) are written with
// plain stores at immediate offsets; only the one group that straddles D is predicated per lane.
__device__ __forceinline__ void nsa_dense_store(float* xrow, const f32x16& acc, int first, int D, int half) {
  float* const xh = xrow + 4 * half;
  asm volatile("" : "+s"(D));  // (the group tests below are compared here, not precomputed per tile as 64-bit masks)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int lo = first + 8 * q;  // wave-uniform
    if (lo + 8 <= D) {
#pragma unroll
      for (int e = 0; e < 4; ++e) xh[lo + e] = acc[4 * q + e];
    } else if (lo < D) {
      // (opaque: as plain comparisons of loop invariants the sixteen predicates per output tile were computed
      // at the top of EVERY tile for every quad of every dense op - ~300 vector instructions per tile - and
      // parked in scalar registers; they are needed by the one quad that straddles D)
      const int lim = nsa_opaque(D - lo - 4 * half);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (e < lim) xh[lo + e] = acc[4 * q + e];
    }
  }
}

// dense (LU-linear, eval-mode cached weight) -----------------------------------
// y = W (x - pre) + bias over all D features: one stage per output tile of 32 features, up to four (D <= 128).
template <int PREC, int RT, class PIPE>
__device__ __forceinline__ void nsa_dense_op(PIPE& pipe, const uint32_t* op, float* const (&xrow)[RT],
                                             int D, int lane, int half, float (&logdet)[RT]) {
  constexpr bool BF = (PREC != NSA_PREC_F32);  // split 16-bit operands (bf16 or f16 pairs)
  const int n_steps = __builtin_amdgcn_readfirstlane((int)op[1]);
  const int n_ot = __builtin_amdgcn_readfirstlane((int)op[2]);
  const int tab_len = (BF ? 16 : 2) * n_steps;
  // one stage per output tile (up to four: 128 features), each with the gather table, the pre-shift, its
  // bias and its weights; every tile reads the x tile, so nothing is written back before the last GEMM
  f32x16 acc[4][RT][1];
  nsa_static_for<4>([&](auto oc) {
    constexpr int OT = decltype(oc)::value;
    if (OT < n_ot) {
      const float* st = pipe.template begin<false>();
      const int* in_tab = (const int*)st;
      const float* pre = st + tab_len;
      const float* bias = pre + tab_len;
      const float* w = bias + 32;
#pragma unroll
      for (int t = 0; t < RT; ++t) nsa_load_bias<1>(bias, half, acc[OT][t]);
      if constexpr (BF) nsa_dense_tile16<PREC, 1, true, RT>(w, in_tab, pre, xrow, n_steps, lane, half, acc[OT], pipe.amax);
      else nsa_dense_tile<1, true, RT>(w, in_tab, pre, xrow, n_steps, lane, half, acc[OT]);
      if (OT + 1 < n_ot) pipe.end();
    }
  });
  nsa_static_for<4>([&](auto oc) {
    constexpr int OT = decltype(oc)::value;
    if (OT < n_ot) {
#pragma unroll
      for (int t = 0; t < RT; ++t) nsa_dense_store(xrow[t], acc[OT][t][0], 32 * OT, D, half);
    }
  });
  if (half == 0) {
#pragma unroll
    for (int t = 0; t < RT; ++t) logdet[t] += nsa_bits_to_float(op[3]);
  }
  NSA_TL(pipe, 10);
  pipe.end();
}

// per-column ((x - a) / b) * c + d (batch norm / act norm in eval mode) ----------
template <int RT, class PIPE>
__device__ __forceinline__ void nsa_elementwise_op(PIPE& pipe, const uint32_t* op,
                                                   float* const (&xrow)[RT], int D, int half, float (&logdet)[RT]) {
#pragma clang fp contract(off)
  const float* st = pipe.template begin<false>();
  const int D4 = (D + 3) & ~3;
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    for (int c = half; c < D; c += 2) {
      const float x = xrow[t][c];
      xrow[t][c] = ((x - st[c]) / st[D4 + c]) * st[2 * D4 + c] + st[3 * D4 + c];
    }
    if (half == 0) logdet[t] += nsa_bits_to_float(op[3]);
  }
  pipe.end();
}

// sign-ordered int64 key of a double (nsa_key_encode, nsa_populate.hip): integer order = value order
__device__ __forceinline__ long long nsa_weight_key(double v) {
  const long long b = __double_as_longlong(v);
  return (b >= 0) ? b : (b ^ 0x7fffffffffffffffLL);
}

// the kernel ------------------------------------------------------------------


// The kernel's argument block as it lies in the kernarg segment, through a pointer the optimiser cannot see
// through: the loads of the fields stay where they are written (see the tile loop).
__device__ __forceinline__ const NsaLaunch* nsa_args() {
  const void* k = (const void*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(k));
  return (const NsaLaunch*)k;
}

// A float of the flow image's header, read where it is used by a scalar load (the image is read-only for
// the kernel: constant address space) through a pointer the optimiser cannot hoist the load with.
__device__ __forceinline__ float nsa_header_float(const uint32_t* img, int word) {
  asm volatile("" : "+s"(img));
  return __uint_as_float(((const __attribute__((address_space(4))) uint32_t*)img)[word]);
}

// OCC = waves per SIMD the instance is register-budgeted for: 2 (<= 256 VGPRs, workgroups of up to
// 8 waves) or 3 (<= 168 VGPRs, workgroups of 6 or 12 waves; only the prologue / epilogue spill).
// (256-wide conditioners - HT >= NSA_WIDE_HT - are built for ONE wave per SIMD whatever OCC says, workgroups of up to
// 4 waves: hidden state + temporaries + accumulators of a 256-wide layer are 384 registers per lane, and at the 256
// of two waves per SIMD the instances spilled ~1000 of them to scratch; one wave per SIMD has the unified file's 512)
// RT = 32-row tiles per wave: the tiles of a wave go through every layer together, so that one read of a
// weight operand from LDS feeds RT x 3 MFMAs, a stage is copied and a barrier crossed once per RT x NW
// tiles, and the wave always has RT independent accumulator chains and element-wise chains to issue from.
// RES = weights resident in LDS (NsaResidentPipe): one workgroup per CU of up to 16 waves (OCC 4: <= 128
// VGPRs), no barrier after the prologue.
// CTX = the extended conditioner: a conditional flow (context columns in the x tile, gated residual blocks:
// nsa_coupling) and / or batch norm within the residual blocks (eval mode); built for the staged one-tile geometry at
// 2 waves per SIMD only.  NsaLaunch::ext selects it.
template <int HT, class TP, bool INV, int PREC, int OCC, int RT = 1, bool RES = false, bool CTX = false>
__global__ void __launch_bounds__(HT >= NSA_WIDE_HT ? 256 : (OCC == 4 ? 1024 : (OCC == 3 ? 768 : 512)),
                                  HT >= NSA_WIDE_HT ? 1 : OCC) nsa_flow_kernel(NsaLaunch p) {
  static_assert(!CTX || (RT == 1 && !RES && OCC == 2), "conditional flows: staged, one tile per wave");
  // the staged two-tiles instances carry the skewed copy of the coupling code
  constexpr bool SK = RT == 2 && !RES;
  using PIPE = std::conditional_t<RES, NsaResidentPipe, std::conditional_t<CTX, NsaRingPipe<(HT < 8)>, NsaStagePipe>>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (p.stop != nullptr && (p.stop[0] >= p.stop[2] || p.stop[1] >= p.stop[5])) return;
  if (p.image_list != nullptr) p.image = p.image_list[blockIdx.y];
  const uint32_t* img = p.image;
  const int D0 = (int)img[NSA_H_D];
  // odd row stride (conflict-free column reads) with two spare columns behind the features: D0 takes the
  // writes of unused feature slots, D0 + 1 always holds 0.0 - the gather tables of layer 0 / the LU-linear op
  // point their unused entries at it (byte offsets: nsa_format.h), so a gathered operand is one add and
  // one read, without the clamp, the shift and the two selects a "-1 = nothing" entry cost per element
  const int C0 = CTX ? (int)img[NSA_H_CTX] : 0;  // context features: columns D0 + 2 ... of the tile
  const int XS0 = (D0 + 2 + C0) | 1;
  const int nw = (int)(blockDim.x >> 6);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane0 = (int)(threadIdx.x & 63);
  constexpr bool inverse = INV;  // p.direction selects the instantiation on the host

  // team geometry (NsaLaunch::team): the waves of a workgroup in teams of `team_n`, one tile per team
  int team_n = 1;
  if constexpr (CTX) team_n = __builtin_amdgcn_readfirstlane(p.team > 1 ? p.team : 1);
  int loaders = 0;  // the workgroup's last `loaders` waves only request the stage copies (NsaLaunch::loader: 0 ... 4)
  if constexpr (CTX) loaders = __builtin_amdgcn_readfirstlane(p.loader > 0 ? (p.loader < 4 ? p.loader : 4) : 0);
  const int n_teams = (nw - loaders) / team_n;
  // a team works in rounds (NsaTeam::rounds) when loader waves do the requesting and the ring is deep enough for them
  // to keep a round's stages landed: NESSAI_AMD_ROUNDS=0 (NsaLaunch::rounds_off) keeps one chunk per barrier
  const bool team_rounds = CTX && team_n > 1 && loaders > 0 && p.rounds_off == 0 &&
                           (CTX ? (p.ring > 2 ? (p.ring < 8 ? p.ring : 8) : 2) : 2) >= (team_n - 1) * nsa_ksplit(TP::C, HT) + 2;
  const int tw = wave / team_n;            // the wave's tile slot in the workgroup
  const int member = wave - tw * team_n;   // ... and its place in the team
  const int tiles_words = (n_teams * RT * NSA_ROWS * XS0 + 3) & ~3;
  const int max_stage = (int)img[NSA_H_MAX_STAGE];

  PIPE pipe;
  pipe.n_stages = (int)(inverse ? img[NSA_H_NSTAGES_INV] : img[NSA_H_NSTAGES_FWD]);
  // stage area: two buffers of the largest stage, or (RES) every stage of the direction's program
  float* const stage_area = smem + tiles_words;
  // (extended instances: a ring of NsaLaunch::ring >= 2 buffers, NsaRingPipe)
  const int ring_n = CTX ? __builtin_amdgcn_readfirstlane(p.ring > 2 ? (p.ring < 8 ? p.ring : 8) : 2) : 2;
  const int stage_words = RES ? (((int)(inverse ? img[NSA_H_RES_INV] : img[NSA_H_RES_FWD]) + 255) & ~255) : ring_n * max_stage;
  if constexpr (CTX) {
    pipe.img = img;
    pipe.gtable = img + (inverse ? img[NSA_H_STAGES_INV] : img[NSA_H_STAGES_FWD]);
    pipe.base = stage_area;
    pipe.buf_words = max_stage;
    pipe.nb = ring_n;
  } else if constexpr (!RES) {
    pipe.img = img;
    pipe.gtable = img + (inverse ? img[NSA_H_STAGES_INV] : img[NSA_H_STAGES_FWD]);
    pipe.buf0 = stage_area;
    pipe.buf1 = stage_area + max_stage;
  }
  const int n_ops = (int)(inverse ? img[NSA_H_NOPS_INV] : img[NSA_H_NOPS_FWD]);
  // POPULATE: the per-column constants of the epilogue (z-score sigma / mu, prior bounds, output
  // column table) live in LDS behind the stage buffers - the epilogue reads them per feature, and
  // from global memory every read was a separate exposed round trip
  const int D4e = (D0 + 3) & ~3;
  double* epi = (double*)(stage_area + stage_words);  // [4][D4e] doubles + [D4e] ints (populate only)
  // The program of this direction - stage table {offset, words} and op table - also lives in LDS
  // (NSA_PROG_WORDS): both are read at every stage / op, the loop's global stores make the compiler
  // fetch them with vector loads, and each was an exposed L2 round trip in front of the stage's copy.
  uint32_t* prog = (uint32_t*)(stage_area + stage_words) + (p.mode == NSA_MODE_POPULATE ? 9 * D4e : 0);
  uint32_t* prog_ops = prog + 2 * pipe.n_stages;
  // one float per lane and tile behind the program: the latent log-density of the row (NaN = row outside
  // the latent radius), written by the tile's prologue and read by its epilogue
  float* const park = (float*)(prog + ((2 * pipe.n_stages + NSA_OP_WORDS * n_ops + 3) & ~3));
  // (lean pass with folded weights: the running maximum of each wave's weights, one double per wave behind the
  // parked floats - live in registers across the flow it would be two more VGPRs in instances that have none)
  // (as sign-ordered int64 keys - nsa_key_encode, nsa_populate.hip - so that the rows raise it with one LDS atomic)
  long long* const wmax = (long long*)(park + (size_t)(blockDim.x >> 6) * RT * 64);
  if (p.log_w_out != nullptr && (threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = nsa_weight_key(-__builtin_huge_val());
  // (team geometry: one count of bad spline roots per lane and wave, then each team's record of Jacobian terms -
  // (D0 + 1) / 2 + 16 slots of 64 lanes: a coupling transforms at most half the features, FPC <= 16 per chunk)
  int* const team_bad = (int*)(wmax + nw);
  float* const team_rec = (float*)(team_bad + nw * 64) + (size_t)tw * 64 * ((D0 + 1) / 2 + 16);
  {
    const uint32_t* gtab = img + (inverse ? img[NSA_H_STAGES_INV] : img[NSA_H_STAGES_FWD]);
    const uint32_t* gops = img + (inverse ? img[NSA_H_OPS_INV] : img[NSA_H_OPS_FWD]);
    for (int i = (int)threadIdx.x; i < 2 * pipe.n_stages; i += (int)blockDim.x) prog[i] = gtab[i];
    for (int i = (int)threadIdx.x; i < n_ops * NSA_OP_WORDS; i += (int)blockDim.x) prog_ops[i] = gops[i];
  }
  pipe.table = prog;
  if (p.mode == NSA_MODE_POPULATE) {
    int* epi_tab = (int*)(epi + 4 * D4e);
    const int* otab = (const int*)(img + (inverse ? img[NSA_H_OUT_INV] : img[NSA_H_OUT_FWD]));
    for (int i = (int)threadIdx.x; i < D0; i += (int)blockDim.x) {
      epi[i] = p.scale != nullptr ? p.scale[i] : 1.0;
      epi[D4e + i] = p.shift != nullptr ? p.shift[i] : 0.0;
      epi[2 * D4e + i] = p.lower != nullptr ? p.lower[i] : 0.0;
      epi[3 * D4e + i] = p.upper != nullptr ? p.upper[i] : 1.0;
      epi_tab[i] = otab[i];
    }
  }
  if constexpr (RES) {
    // the whole program -> LDS, 1 KiB pieces dealt round the waves; the barrier below is the LAST one
    if (pipe.n_stages > 0) {
      const uint32_t off0 = (inverse ? img + img[NSA_H_STAGES_INV] : img + img[NSA_H_STAGES_FWD])[0];
      pipe.off0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)off0);
      pipe.base = stage_area;
      const int pieces = (stage_words + 255) >> 8;
      for (int c = wave; c < pieces; c += nw) nsa_glds16(img + pipe.off0 + c * 256 + lane0 * 4, stage_area + c * 256);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    pipe.skew = false;
  }
  __syncthreads();
  if constexpr (!RES) {
    pipe.wave = wave;
    pipe.nw = nw;
    pipe.lane = lane0;
    pipe.issued = false;
    if constexpr (CTX) {
      pipe.role = loaders ? (wave >= nw - loaders ? 2 : 1) : 0;
      pipe.ld_index = wave - (nw - loaders);
      pipe.ld_count = loaders;
      pipe.lookahead = team_rounds ? (team_n - 1) * nsa_ksplit(TP::C, HT) : 0;
    }
    // p.skew: 1 = the upper half of the workgroup's waves, 2 = every other wave
    pipe.skew = SK && (p.skew == 1 ? (wave >= (nw >> 1)) : (p.skew == 2 ? (wave & 1) != 0 : false));
  }

  if (p.prio == 1 && wave >= (nw >> 1)) __builtin_amdgcn_s_setprio(1);
  if (p.prio == 2 && wave < (nw >> 1)) __builtin_amdgcn_s_setprio(1);

  const uint32_t* ops = prog_ops;
  // (the offset through readfirstlane: loaded by a vector load, the table's address was a VGPR pair live across
  // the whole tile loop - the one value the two-tiles instance spilled once the spline grew by a branch)
  const int* out_tab = (const int*)(img + __builtin_amdgcn_readfirstlane((int)(inverse ? img[NSA_H_OUT_INV] : img[NSA_H_OUT_FWD])));
  // (the base density constants - header words NSA_H_BASE_INV_VAR / NSA_H_LOG_Z - are read where they are used by
  // scalar loads from the image, nsa_header_float)

  NsaTransformConsts tc;
  tc.tail_bound = nsa_bits_to_float(img[NSA_H_TAIL_BOUND]);
  tc.two_b = tc.tail_bound + tc.tail_bound;
  tc.min_bw = nsa_bits_to_float(img[NSA_H_MIN_BW]);
  tc.min_bh = nsa_bits_to_float(img[NSA_H_MIN_BH]);
  tc.min_d = nsa_bits_to_float(img[NSA_H_MIN_D]);
  tc.edge_raw = nsa_bits_to_float(img[NSA_H_EDGE_RAW]);
  tc.edge_d = tc.min_d + nsa_softplus(tc.edge_raw);
  tc.inv_logit_div = 1.0f / nsa_bits_to_float(img[NSA_H_LOGIT_DIV]);
  tc.aff_a = nsa_bits_to_float(img[NSA_H_AFF_A]);
  tc.aff_b = nsa_bits_to_float(img[NSA_H_AFF_B]);
  tc.aff_arg0 = -1.4426950408889634f * tc.aff_a;
  tc.aff_1pb = 1.0f + tc.aff_b;
  tc.ld_group4 = tc.aff_b >= 1e-9f;
  tc.aff_cap = tc.ld_group4 ? 64.0f : 126.0f;
  {
    const int K = (int)img[NSA_H_BINS];
    tc.k_real = K;
    tc.bw_scale = (float)(1.0 - (double)tc.min_bw * K);
    tc.bh_scale = (float)(1.0 - (double)tc.min_bh * K);
  }
  const int act = (int)img[NSA_H_ACT];
  const int net = (int)img[NSA_H_NET];
  const int n_layers = (int)img[NSA_H_NLAYERS];

  const long long n_tiles = (p.n + NSA_ROWS - 1) / NSA_ROWS;
  const long long n_groups = (n_tiles + n_teams * RT - 1) / (n_teams * RT);

#ifdef NSA_TIMELINE
  for (int i = 0; i < 14; ++i) pipe.tl.acc[i] = 0;
  unsigned long long tl_tiles = 0;
#endif
  if (pipe.n_stages > 0) pipe.prime();
#ifdef NSA_TIMELINE
  pipe.tl.last = __builtin_readcyclecounter();
#endif

  if constexpr (CTX) {
    if (loaders && wave >= nw - loaders) {
      // a loader: the stage ring's request side and every barrier of the tile loop, nothing else
      for (long long g = blockIdx.x; g < n_groups; g += gridDim.x) {
        pipe.barrier();  // (the tile is filled)
        for (int st = 0; st < pipe.n_stages; ++st) {
          (void)pipe.template begin<false>();
          pipe.end();
        }
        pipe.barrier();  // (the team's counts of bad roots)
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      return;
    }
  }
  for (long long g = blockIdx.x; g < n_groups; g += gridDim.x) {
    NSA_TL(pipe, 11);
#ifdef NSA_TIMELINE
    ++tl_tiles;
#endif
    // ---- fill the x tiles --------------------------------------------------
    // Kernel arguments are re-read from the kernarg segment (scalar loads) by the section that uses them:
    // held in scalar registers across the flow - ~80 of them for this argument block - they were spilled to
    // VGPR lanes and came back through v_readlane, vector instructions, on every tile.
    const NsaLaunch& a = *nsa_args();
    // The lane index is opaque to the optimiser from here on: every per-lane quantity that depends on the lane
    // and the flow's shape alone - tile addresses, the "column < D" predicates of the ragged paths, first-round
    // products of the Philox counters - is otherwise computed once in front of the tile loop and kept alive
    // across it: ~60 scalar registers of predicates alone (spilled to VGPR lanes) and the vector registers the
    // 256-register instances then spilled to scratch.
    // (... and not even kept: mbcnt counts the lanes below this one - two instructions per tile)
    const int lane = nsa_opaque((int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
    const int half = lane >> 5;
    const int j = lane & 31;
    // (the feature count likewise: the wave-uniform "group below D" tests of the ragged paths were kept as
    // 64-bit masks, two scalar registers each, from the kernel's entry on)
    int D = D0;
    asm volatile("" : "+s"(D));
    const int XS = (D + 2 + C0) | 1;
    float* xt_a[RT];
    float* xrow_a[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      xt_a[t] = smem + (tw * RT + t) * (NSA_ROWS * XS);
      xrow_a[t] = xt_a[t] + j * XS;
    }
    if (!CTX || member == 0) {  // (a team's tile is filled by its first wave)
#pragma unroll
    for (int t = 0; t < RT; ++t) {
    bool keep = true;
    float log_pz = 0.0f;
    float* const xt = xt_a[t];
    float* const xrow = xrow_a[t];
    const long long row0 = ((g * n_teams + tw) * RT + t) * NSA_ROWS;
    const long long my_row = row0 + j;
    if (a.x_in != nullptr) {
      const long long base = row0 * D;
      const long long limit = a.n * (long long)D;
      for (int idx = nsa_opaque(lane); idx < NSA_ROWS * D; idx += 64) {
        const int r = idx / D, c = idx - r * D;
        const long long gi = base + idx;
        xt[r * XS + c] = (gi < limit) ? a.x_in[gi] : 0.0f;
      }
      xrow[D + 1] = 0.0f;
    } else {
      // latent draw: 4 normals per Philox call; the two lanes of a row split
      // the quads (row-indexed counter => independent of launch geometry)
      // (pairs of quads, one per lane of the row: while both lie wholly below D - a wave-uniform test - the
      // four normals are stored without a per-lane predicate; only the pair that straddles D pays for one)
      unsigned long long grow = a.row_offset + (unsigned long long)my_row;
      if (a.row_list != nullptr) grow = (unsigned long long)a.row_list[my_row < a.n ? my_row : a.n - 1];
      const int n_quads = (D + 3) >> 2;
      xrow[D + 1] = 0.0f;
      for (int q0 = 0; q0 < n_quads; q0 += 2) {
        const int q = q0 + half;
        float z0, z1, z2, z3;
        if (NSA_ABLATE & 16) {
          z0 = 0.25f * (float)q - 1.0f, z1 = z0 + 0.25f, z2 = z0 + 0.5f, z3 = z0 + 0.75f;
        } else {
          uint32_t r4[4];
          nsa_philox4x32_10((uint32_t)grow, (uint32_t)(grow >> 32), (uint32_t)q, 0u,
                            (uint32_t)a.seed, (uint32_t)(a.seed >> 32), r4);
          nsa_box_muller(r4[0], r4[1], z0, z1);
          nsa_box_muller(r4[2], r4[3], z2, z3);
        }
        float* const xq = xrow + 4 * q;
        if (4 * q0 + 8 <= D) {
          xq[0] = z0;
          xq[1] = z1;
          xq[2] = z2;
          xq[3] = z3;
        } else {
          const int lim = nsa_opaque(D - 4 * q);  // (opaque: see nsa_dense_store)
          if (0 < lim) xq[0] = z0;
          if (1 < lim) xq[1] = z1;
          if (2 < lim) xq[2] = z2;
          if (3 < lim) xq[3] = z3;
        }
      }
    }

    if constexpr (CTX) {
      // the row's context, constant through the flow
      const long long cbase = row0 * C0, climit = a.n * (long long)C0;
      for (int idx = nsa_opaque(lane); idx < NSA_ROWS * C0; idx += 64) {
        const int r = idx / C0, c = idx - r * C0;
        xt[r * XS + D + 2 + c] = (cbase + idx < climit) ? a.ctx_in[cbase + idx] : 0.0f;
      }
    }
    // ---- POPULATE prologue: latent norm, latent log-density, optional z out --
    if constexpr (RES) NSA_TL(pipe, 12);  // (timeline of the resident geometry: slot 12 = fill / latent draw)
    if (a.mode == NSA_MODE_POPULATE && !(NSA_ABLATE & 128)) {
      // latent temperature (proposal/flowproposal/base.py:393-414): the flow sees sqrt(T) z0,
      // the latent density is evaluated at z / sqrt(T) (= z0 exactly for drawn latents)
      const double ls = a.latent_scale;
      const bool tempered = ls != 1.0;
      double r2 = 0.0;
      float e = 0.0f;
      if (!tempered) {
        // the usual case as straight-line code: D / 2 columns per lane at immediate offsets from the lane's
        // own start (same columns in the same order as the general loop below)
        const float* const xh = xrow + half;
        const int nk = D >> 1;
#pragma unroll 8
        for (int k = 0; k < nk; ++k) {
          const float z = xh[2 * k];
          r2 += (double)z * (double)z;
          e = fmaf(z, z, e);  // (explicit: the same sum in every kernel instance)
        }
        if ((D & 1) && half == 0) {
          const float z = xrow[D - 1];
          r2 += (double)z * (double)z;
          e = fmaf(z, z, e);  // (explicit: the same sum in every kernel instance)
        }
      } else
      for (int c = half; c < D; c += 2) {
        float z = xrow[c];
        float zin = z;
        if (a.x_in == nullptr) {
          z = (float)((double)z * ls);
          xrow[c] = z;
        } else {
          zin = (float)((double)z / ls);
        }
        r2 += (double)z * (double)z;
        e = fmaf(zin, zin, e);
      }
      r2 += __shfl_xor(r2, 32);
      e += __shfl_xor(e, 32);
      keep = sqrt(r2) <= a.radius;
      log_pz = -(0.5f * nsa_header_float(img, NSA_H_BASE_INV_VAR)) * e - nsa_header_float(img, NSA_H_LOG_Z);
      if (a.z_out != nullptr) {
        const long long base = row0 * D;
        const long long limit = a.n * (long long)D;
        for (int idx = nsa_opaque(lane); idx < NSA_ROWS * D; idx += 64) {
          const int r = idx / D, c = idx - r * D;
          const long long gi = base + idx;
          if (gi < limit) a.z_out[gi] = xt[r * XS + c];
        }
      }
    }
    // parked in LDS until the epilogue (one cell per lane and tile): live in registers across the whole flow
    // these per-row values were what the 256-register instances spilled to scratch
    park[(wave * RT + t) * 64 + lane] = keep ? log_pz : __uint_as_float(0x7fc00000u);
    }
    }
    if constexpr (CTX) {
      pipe.settle();  // (the prologue's stores - z_out - are done: NsaRingPipe counts loads only)
      if (team_n > 1 || loaders) pipe.barrier();  // the tile is there for the rest of the team
    }
    NSA_TL(pipe, 0);
    // ---- run the flow --------------------------------------------------------
    pipe.amax = u16x2{0, 0};
    pipe.nan_seen = false;
    float logdet_a[RT];
    int n_bad_a[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      logdet_a[t] = 0.0f;
      n_bad_a[t] = 0;
    }
    for (int o = 0; o < n_ops; ++o) {
      const uint32_t* op = ops + o * NSA_OP_WORDS;
      const int kind = __builtin_amdgcn_readfirstlane((int)op[0]);
      if (kind == NSA_OP_COUPLING) {
        if constexpr (SK) {
          if (pipe.skew)
            nsa_coupling<HT, TP, INV, PREC, RT, true>(pipe, op, xrow_a, D, lane, half, act, net, n_layers, tc, logdet_a, n_bad_a);
          else
            nsa_coupling<HT, TP, INV, PREC, RT, false>(pipe, op, xrow_a, D, lane, half, act, net, n_layers, tc, logdet_a, n_bad_a);
        } else {
          nsa_coupling<HT, TP, INV, PREC, RT, false, CTX>(pipe, op, xrow_a, D, lane, half, act, net, n_layers, tc, logdet_a, n_bad_a,
                                                          CTX && img[NSA_H_BNW] != 0, NsaTeam{team_n, member, team_rec, team_rounds ? 1 : 0});
        }
      } else if (CTX && member != 0) {
        // team geometry: the ops between the couplings are the first member's (they rewrite the whole tile; their
        // last stage's barrier hands it back to the team)
        const int skip = kind == NSA_OP_DENSE ? __builtin_amdgcn_readfirstlane((int)op[2]) : 1;
        for (int k = 0; k < skip; ++k) {
          (void)pipe.template begin<false>();
          pipe.end();
        }
      } else if (NSA_ABLATE & 512) {
        const int skip = kind == NSA_OP_DENSE ? __builtin_amdgcn_readfirstlane((int)op[2]) : 1;
        for (int k = 0; k < skip; ++k) {
          (void)pipe.template begin<false>();
          pipe.end();
        }
      } else if (kind == NSA_OP_DENSE) {
        nsa_dense_op<PREC, RT>(pipe, op, xrow_a, D, lane, half, logdet_a);
      } else {
        nsa_elementwise_op<RT>(pipe, op, xrow_a, D, half, logdet_a);
      }
    }

    // ---- epilogues -----------------------------------------------------------
    // (the arguments the epilogues use are re-read from the kernarg segment here - nsa_args)
    const NsaLaunch& e = *nsa_args();
    float* e_ld_out = e.ld_out;
    float* e_x_out = e.x_out;
    if (e.image_list != nullptr) {  // grouped launch: flow blockIdx.y writes its own block of the outputs
      if (e_ld_out != nullptr) e_ld_out += (long long)blockIdx.y * e.n;
      if (e_x_out != nullptr) e_x_out += (long long)blockIdx.y * e.n * D;
    }
    if constexpr (PREC == NSA_PREC_F16X3) {
      // a value beyond the f16 range went into a matrix product somewhere in this wave's tiles: the
      // results of the call are not to be used (status[1]; the host repeats it in float32 or raises)
      // (NSA_F16_DETECT == 1: the outputs are tested by the epilogue that reads them anyway - `poison` below)
      if constexpr (NSA_F16_DETECT != 1) {
        if (__any(pipe.amax[0] >= NSA_F16_INF || pipe.amax[1] >= NSA_F16_INF) && lane == 0) atomicAdd(e.status + 1, 1);
      }
    }
    // NaN or +-inf in the log-det or in any output column of the wave's tiles -> NaN here (split-f16 only)
    float poison = 0.0f;
    constexpr bool POISON = PREC == NSA_PREC_F16X3 && NSA_F16_DETECT == 1 && !(NSA_ABLATE & 256);
    if constexpr (CTX) {
      if (team_n > 1 || loaders) {  // the members' counts of bad spline roots -> the first member's
        team_bad[wave * 64 + lane] = n_bad_a[0];
        pipe.barrier();
        if (member == 0)
          for (int m = 1; m < team_n; ++m) n_bad_a[0] += team_bad[(wave + m) * 64 + lane];
      }
    }
    if (!CTX || member == 0) {  // (team geometry: the rows leave through the team's first wave)
#pragma unroll
    for (int t = 0; t < RT; ++t) {
    float* const xt = xt_a[t];
    float* const xrow = xrow_a[t];
    const float log_pz = park[(wave * RT + t) * 64 + nsa_opaque(lane)];
    const bool keep = log_pz == log_pz;  // NaN = outside the latent radius
    const float logdet = logdet_a[t] + __shfl_xor(logdet_a[t], 32);
    const int n_bad = n_bad_a[t] + __shfl_xor(n_bad_a[t], 32);
    // row indices are recomputed from the (opaque) group counter instead of staying live in
    // VGPRs through the whole flow, where they were spilled to scratch
    long long g_epi = g;
    asm volatile("" : "+s"(g_epi));
    const long long row0 = ((g_epi * n_teams + tw) * RT + t) * NSA_ROWS;
    const long long my_row = row0 + j;
    const bool row_ok = my_row < e.n;
    // (opaque too: with `half` the per-lane table addresses of every epilogue variant were hoisted out
    // of the tile loop and lived - spilled, in the 168-register instances - across the whole flow)
    const int half_e = nsa_opaque(half);
    if (e.mode == NSA_MODE_POPULATE && e.xp_out != nullptr) {
#pragma clang fp contract(off)
      // importance-sampler draw: x = 1 / (1 + exp(-x')), log|J_inv| = sum log x + log1p(-x)
      // (utils/rescaling.py:316-336), log|J| = sum -log xc - log1p(-xc) on the clipped x
      const int* epi_tab = (const int*)(epi + 4 * D4e);
      double lj_inv = 0.0, lj = 0.0;
      int good = 1;
      double* xo = e.x64_out + my_row * D;
      float* xpo = e.xp_out + my_row * D;
      const double eps = e.unit_eps;
      for (int i = half_e; i < D; i += 2) {
        const float xpf = xrow[epi_tab[i]];
        if constexpr (POISON) poison = fmaf(xpf, 0.0f, poison);
        const double xp = (double)xpf;
        const double u = 1.0 / (1.0 + exp(-xp));
        lj_inv += log(u) + log1p(-u);
        const double uc = fmin(fmax(u, eps), 1.0 - eps);
        lj += -log(uc) - log1p(-uc);
        good &= (int)((xp - xp) == 0.0) & (int)!(u < 0.0) & (int)!(u > 1.0);
        if (row_ok) {
          xo[i] = u;
          xpo[i] = xpf;
        }
      }
      lj_inv += __shfl_xor(lj_inv, 32);
      lj += __shfl_xor(lj, 32);
      good &= __shfl_xor(good, 32);
      const bool ok = good != 0 && (lj - lj) == 0.0 && (lj_inv - lj_inv) == 0.0 && n_bad == 0;
      if (row_ok && half == 0) e.log_q_out[my_row] = ok ? lj : __longlong_as_double(0x7ff8000000000000LL);
      if (row_ok && half == 0 && n_bad != 0) atomicAdd(e.status, 1);
    } else if (e.mode == NSA_MODE_POPULATE) {
#pragma clang fp contract(off)

      // x = x' * sigma + mu in float64 (rescale.py:276-279), prior bounds
      // (model.py:512-518), log_q = (log N(z) - log_j) - log_J
      // (flowproposal.py:356, 383)
      bool in_bounds = true;
      double log_prior = 0.0;
      double log_j_row = 0.0;  // row-dependent part of the rescaling log-Jacobian
      if (NSA_ABLATE & 64) {
      } else if (e.rp_kind == nullptr && e.prior_kind == nullptr) {
        // the plain case (z-score everywhere, prior evaluated elsewhere) without the per-column
        // dispatch: one multiply, one add, two compares and one store per feature
        // D / 2 features per lane - the lane's `half`, then every other one - as straight-line code at
        // immediate offsets from per-lane bases (tables, output row), eight per round of the loop
        const int* const tab_h = (const int*)(epi + 4 * D4e) + half_e;
        const double* const epi_h = epi + half_e;
        double* const xo_h = e.x64_out + my_row * D + half_e;
        int inb = 1;
        auto feature = [&](const int* tab, const double* c, double* xo, bool store) {
          const float xpf = xrow[tab[0]];
          if constexpr (POISON) poison = fmaf(xpf, 0.0f, poison);
          const double xv = ((double)xpf * c[0]) + c[D4e];
          inb &= (int)!(xv < c[2 * D4e]) & (int)!(xv > c[3 * D4e]);
          // (plain stores: as non-temporal ones these 8-byte pieces - two lanes per 16 bytes of a row - went to
          // memory one partial line at a time: WRITE_SIZE 1092 MB per launch against 264, kernel 1.72 against
          // 1.58 ms; the L2 merges them.  What they evict of the flow image costs 35 MB of re-fetches per launch.)
          if (store) xo[0] = xv;
        };
        const int nk = D >> 1;
        const bool tail = (D & 1) && half_e == 0;
        if (row_ok && e.x64_out != nullptr && !(NSA_ABLATE & 32)) {
#pragma unroll 8
          for (int k = 0; k < nk; ++k) feature(tab_h + 2 * k, epi_h + 2 * k, xo_h + 2 * k, true);
          if (tail) feature(tab_h + D - 1, epi_h + D - 1, xo_h + D - 1, true);
        } else {
#pragma unroll 8
          for (int k = 0; k < nk; ++k) feature(tab_h + 2 * k, epi_h + 2 * k, xo_h + 2 * k, false);
          if (tail) feature(tab_h + D - 1, epi_h + D - 1, xo_h + D - 1, false);
        }
        in_bounds = inb != 0;
      } else
      for (int i = half_e; i < D; i += 2) {
        const float xpf = xrow[out_tab[i]];
        if constexpr (POISON) poison = fmaf(xpf, 0.0f, poison);
        const double xp = (double)xpf;
        double xv;
        const int kind = (e.rp_kind != nullptr) ? e.rp_kind[i] : NSA_RP_SCALE_SHIFT;
        if (kind == NSA_RP_SCALE_SHIFT) {
          xv = (xp * e.scale[i]) + e.shift[i];
        } else if (kind == NSA_RP_NULL) {
          xv = xp;
        } else if (kind >= NSA_RP_SS_SIGMOID && kind <= NSA_RP_INV_GAUSSIAN_CDF_SS) {
          // the z-score of a rescaled parameter (reparameterisations/__init__.py:99-162; utils/rescaling.py:290-415):
          // the affine map, then the inverse of the rescaling that sat in front of it - or, for the Gaussian CDF
          // BEHIND the z-score, its inverse first
          if (kind == NSA_RP_INV_GAUSSIAN_CDF_SS) {
            const double v = -1.4142135623730951 * erfcinv(2.0 * xp);
            log_j_row += 0.9189385332046727 + 0.5 * (v * v);
            xv = (v * e.scale[i]) + e.shift[i];
            // x' outside [0, 1] has no inverse: the reference gets NaN for the parameter and for log q, and KEEPS the
            // row (its bounds test lets NaN through - as the one below does; such a row is never accepted and still
            // counts as proposed).  Here: the parameter NaN, log q = +inf - this file's mark of such rows (a NaN
            // log q is the mark of a DROPPED row); the device library's erfcinv promises nothing outside [0, 2].
            if (!(xp >= 0.0 && xp <= 1.0)) {
              xv = __longlong_as_double(0x7ff8000000000000LL);
              log_j_row = -__builtin_huge_val();
            }
          } else {
            const double v = (xp * e.scale[i]) + e.shift[i];
            if (kind == NSA_RP_SS_SIGMOID) {
              xv = 1.0 / (1.0 + exp(-v));
              log_j_row += log(xv) + log1p(-xv);
            } else if (kind == NSA_RP_SS_EXP) {
              xv = exp(v);
              log_j_row += v;
            } else {
              xv = 0.5 * erfc(-v / 1.4142135623730951);
              log_j_row += -0.9189385332046727 - 0.5 * (v * v);
            }
          }
        } else if (kind == NSA_RP_INVERT) {
          const double* c = e.rp_par + 5 * i;
          double u = (xp < 0.0) ? -xp : xp;
          if (c[2] != 0.0) u = 1.0 - u;
          xv = (c[1] - c[0]) * u + c[0];
          xv = xv + c[4];
        } else if (kind == NSA_RP_MINUS_ONE_ONE) {
          const double* c = e.rp_par + 5 * i;
          xv = (c[1] - c[0]) * ((xp + 1.0) / 2.0) + c[0];
          xv = xv + c[4];
        } else if (kind == NSA_RP_ANGLE) {
          const double* c = e.rp_par + 5 * i;
          const double cy = (double)xrow[out_tab[i + 1]];
          double a = atan2(cy, xp);
          if (c[1] != 0.0 && a < 0.0) a = a + 6.283185307179586;  // numpy's % (2 pi) of a value in [-pi, pi]
          xv = a / c[0];
          // ToCartesian (angle.py:223-232): the mirrored parameter folded back, |angle| / scale in [0, 1] onto its bounds
          if (c[2] != 0.0) xv = c[4] * fabs(xv) + c[3];
        } else if (kind == NSA_RP_ANGLE_RADIAL) {
          const double cx = (double)xrow[out_tab[i - 1]];
          xv = sqrt(cx * cx + xp * xp);
          log_j_row -= log(xv);
        } else if (kind == NSA_RP_PAIR_H) {
          // a direction from a point of space (the reference's AnglePair, angle.py:434-507): columns i, i + 1, i + 2
          // hold x, y, z; par = {modulo 2 pi?, ra-dec?}
          const double* c = e.rp_par + 5 * i;
          const double cy = (double)xrow[out_tab[i + 1]];
          xv = atan2(cy, xp);
          if (c[0] != 0.0 && xv < 0.0) xv = xv + 6.283185307179586;
        } else if (kind == NSA_RP_PAIR_V) {
          const double* c = e.rp_par + 5 * i;
          const double cx = (double)xrow[out_tab[i - 1]];
          const double cz = (double)xrow[out_tab[i + 1]];
          const double rho = sqrt(cx * cx + xp * xp);
          if (c[1] != 0.0) { xv = atan2(cz, rho); log_j_row -= log(cos(xv)); }   // declination in [-pi/2, pi/2]
          else             { xv = atan2(rho, cz); log_j_row -= log(sin(xv)); }   // zenith in [0, pi]
        } else if (kind == NSA_RP_PAIR_R) {
          const double cx = (double)xrow[out_tab[i - 2]];
          const double cy = (double)xrow[out_tab[i - 1]];
          xv = sqrt(cx * cx + cy * cy + xp * xp);
          log_j_row -= 2.0 * log(xv);
        } else {
          // rescale-to-bounds, its logit / log forms, and the same three followed by floor() for a discrete
          // parameter (the reference's Dequantise: discrete.py:9-78)
          const double* c = e.rp_par + 5 * i;
          const bool discrete = kind >= NSA_RP_DEQUANTISE;
          const int base = discrete ? kind - (NSA_RP_DEQUANTISE - NSA_RP_TO_BOUNDS) : kind;
          double u = xp;
          if (base == NSA_RP_TO_BOUNDS_LOGIT) {
            u = 1.0 / (1.0 + exp(-xp));
            log_j_row += log(u) + log1p(-u);
          } else if (base == NSA_RP_TO_BOUNDS_LOG) {
            u = exp(xp);
            log_j_row += xp;
          }
          xv = (c[1] - c[0]) * (u - c[2]) / c[3] + c[0];
          xv = xv + c[4];
          if (discrete) xv = floor(xv);
        }
        in_bounds = in_bounds && !((xv < e.lower[i]) || (xv > e.upper[i]));
        if (row_ok && e.x64_out != nullptr && !(NSA_ABLATE & 32)) e.x64_out[my_row * D + i] = xv;
        if (e.prior_kind != nullptr) log_prior += nsa_log_prior_term(e.prior_kind[i], xv, e.prior_par + 3 * i);
      }
      if (e.prior_kind != nullptr) {
        log_prior += __shfl_xor(log_prior, 32);
        if (row_ok && half == 0) e.log_prior_out[my_row] = log_prior;
      }
      if (e.rp_kind != nullptr) log_j_row += __shfl_xor(log_j_row, 32);
      const int ib = in_bounds ? 1 : 0;
      const bool ok = (ib + __shfl_xor(ib, 32)) == 2;
      // finite check on the flow's log_q (flowproposal.py:366-372), THEN the rescaling
      // Jacobian (flowproposal.py:383): a saturated sigmoid gives log_q = +inf, kept as the
      // reference keeps it (never accepted, still counted as proposed)
      double lq = ((double)log_pz - e.log_j_latent) - (double)logdet;
      const bool finite = (lq - lq) == 0.0;
      lq -= (e.log_j_rescale + log_j_row);
      if (!(keep && ok && finite && n_bad == 0)) lq = __longlong_as_double(0x7ff8000000000000LL);
      if (e.log_w_out != nullptr) {
        // weights of the lean pass (nsa_weights_kernel's arithmetic): log_w = log_prior - log_q
        if (!(lq > e.min_log_q)) lq = __longlong_as_double(0x7ff8000000000000LL);
        const double lp = (e.prior_kind != nullptr) ? log_prior : e.log_prior_const;
        double w = lp - lq;
        if (lq != lq || w != w) w = -__builtin_huge_val();
        if (row_ok && half == 0) e.log_w_out[my_row] = w;
        if (row_ok && half == 0 && w > -__builtin_huge_val()) atomicMax(wmax + wave, nsa_weight_key(w));
      } else if (row_ok && half == 0) {
        e.log_q_out[my_row] = lq;
      }
      if (row_ok && half == 0 && n_bad != 0) atomicAdd(e.status, 1);
    } else {
      float out_ld = logdet;
      if (e.mode == NSA_MODE_LOGPROB) {
        // log_prob = (-0.5/var * sum z^2 - log_z) + logabsdet (base.py:243-245)
        float e = 0.0f;
        for (int i = half_e; i < D; i += 2) {
          const float z = xrow[out_tab[i]];
          if constexpr (POISON) poison = fmaf(z, 0.0f, poison);
          e = fmaf(z, z, e);  // (explicit: the same sum in every kernel instance)
        }
        e += __shfl_xor(e, 32);
        const float lp = -(0.5f * nsa_header_float(img, NSA_H_BASE_INV_VAR)) * e - nsa_header_float(img, NSA_H_LOG_Z);
        out_ld = lp + logdet;
      }
      if (row_ok && half == 0 && e_ld_out != nullptr) e_ld_out[my_row] = out_ld;
      if (row_ok && half == 0 && n_bad != 0) atomicAdd(e.status, 1);
      if (e_x_out != nullptr) {
        const long long base = row0 * D;
        const long long limit = e.n * (long long)D;
        for (int idx = nsa_opaque(lane); idx < NSA_ROWS * D; idx += 64) {
          const int r = idx / D, c = idx - r * D;
          const long long gi = base + idx;
          const float v = xt[r * XS + out_tab[c]];
          if constexpr (POISON) poison = fmaf(v, 0.0f, poison);
          if (gi < limit) e_x_out[gi] = v;
        }
      } else if (e.mode != NSA_MODE_LOGPROB) {
        if constexpr (POISON)
          for (int c = half_e; c < D; c += 2) poison = fmaf(xrow[c], 0.0f, poison);
      }
    }
    if constexpr (POISON) poison = fmaf(logdet_a[t], 0.0f, poison);
    }
    }
    if constexpr (PREC == NSA_PREC_F16X3 && NSA_F16_DETECT == 1) {
      // a value beyond the f16 range went into a matrix product somewhere in this wave's tiles
      if (__any(pipe.nan_seen || (poison - poison) != 0.0f) && lane == 0) atomicAdd(e.status + 1, 1);
    }
    if constexpr (CTX) pipe.settle();  // (the epilogue's stores and atomics are done)
    NSA_TL(pipe, 9);
  }
  // drain the wrap-around prefetch before the workgroup exits
  if constexpr (!RES) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // (arguments re-read from the kernarg segment: held in scalar registers across the tile loop they were spilled)
  const NsaLaunch& fin = *nsa_args();
  if (fin.log_w_out != nullptr && lane0 == 0) {
    // the wave's largest weight -> the launch's; same-address atomics serialise, so only a wave that can still
    // raise the maximum issues one
    long long* const max_key_out = fin.max_key_out;
    const long long key = wmax[wave];
    if (key > __atomic_load_n(max_key_out, __ATOMIC_RELAXED)) atomicMax(max_key_out, key);
  }
#ifdef NSA_TIMELINE
  if (p.dbg != nullptr && lane0 == 0) {
    for (int i = 0; i < 14; ++i) atomicAdd(p.dbg + i, pipe.tl.acc[i]);
    atomicAdd(p.dbg + 14, tl_tiles);
    atomicAdd(p.dbg + 15, 1ull);
  }
#endif
}
