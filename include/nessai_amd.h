/* nessai_amd — C ABI of the MI355X (gfx950) normalising-flow proposal library.
 *
 * The reference (mj-will/nessai) has no native code and no FFI: its hot path is
 * eager PyTorch called through three Python plug-in surfaces.
 * This header is the C boundary the build adds underneath those surfaces.  Each
 * entry point names the reference interface whose work it replaces; paths are
 * relative to the reference tree.  All pointers named *_dev are device (HBM)
 * pointers, everything else is host memory.  Outputs are caller-allocated.
 * Every function returns 0 on success and a negative NSA_E_* code otherwise
 * (nsa_last_error() gives the message); a per-row domain failure inside the
 * spline inverse never aborts — the row is flagged (NaN log_q / status counter).
 *
 * INTEGRATION.md shows the ctypes binding a maintainer would add on the
 * reference side.
 */
#ifndef NESSAI_AMD_H
#define NESSAI_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NSA_OK 0
#define NSA_E_ARG (-1)       /* bad argument / unsupported configuration */
#define NSA_E_HIP (-2)       /* HIP runtime error */
#define NSA_E_IMAGE (-3)     /* malformed flow image */

#define NSA_HEADER_WORDS 40

#define NSA_API __attribute__((visibility("default")))

/* Library / ABI version (NSA_VERSION of the flow image format). */
NSA_API int nsa_version(void);

/* Message of the last failing call on this thread. */
NSA_API const char* nsa_last_error(void);

/* 1 if a kernel instance exists for this configuration (hidden tiles HT,
 * transform kind NSA_T_*, spline bins), else 0.  Lets the Python flow classes
 * reject unsupported constructor arguments up front
 * (src/nessai/flows/utils.py:208-246, configure_model). */
NSA_API int nsa_flow_supported(int hidden_tiles, int transform, int bins);

/* 1 if a rational-quadratic spline flow of `bins` bins has no exact kernel instance and runs on the generic
 * one (any num_bins from 2 to 16 - src/nessai/flows/nsf.py:66,103 passes num_bins through to nflows - on
 * parameter sets padded to 16 bins: the packer, nessai_amd/pack.py, must lay the last conditioner layer out
 * for it); 0 for an exact instance or an unsupported count. */
NSA_API int nsa_flow_generic_bins(int hidden_tiles, int bins);

/* Launch geometry of the last flow kernel this thread enqueued: waves per SIMD the instance is
 * built for (2 or 3), waves per workgroup, workgroups.  The library picks it per launch (batch
 * size, conditioner width, precision; for big split-bf16 batches by timing the three candidates
 * once per process) - reported so that a benchmark line can say what it measured.  No reference
 * counterpart (torch picks its own launch shapes). */
NSA_API void nsa_last_geometry(int* waves_per_simd, int* waves_per_group, int* groups);

/* 32-row tiles each wave of that launch carried through the flow together: 1, or 2 (the third
 * candidate geometry of big split-bf16 batches: one 8-wave workgroup per CU, every weight operand
 * read from LDS feeding both tiles). */
NSA_API int nsa_last_row_tiles(void);

/* 1 if that launch ran in the weights-resident geometry: the whole program of the direction (every
 * stage blob of the flow image) copied to LDS once per workgroup, one workgroup of up to 16 waves per
 * CU, no stage copy, wait or barrier in the tile loop.  Chosen by the library whenever the program fits
 * in LDS beside the x tiles (the reference's default RealNVP, flows/realnvp.py:76-214: ~98 KB); the rows
 * are those of the staged geometries bit for bit.  NESSAI_AMD_RESIDENT=0 switches it off.  No reference
 * counterpart. */
NSA_API int nsa_last_resident(void);

/* Computing waves per 32-row tile of that launch when it ran in the SMALL-BATCH geometry (0: a staged / resident
 * launch).  Batches of up to five tiles per CU are latency-bound - one workgroup per tile walks the whole program - and
 * the library then launches, per tile, one computing wave on a ring of up to eight stage buffers, with loader waves
 * that only request the stage copies (three per workgroup; several tiles share a workgroup above one tile per CU).
 * Chosen for conditioners of 64 features and more whose program does not stay resident in LDS
 * (ImportanceFlowProposal.draw / log_prob_all on a few thousand rows, src/nessai/proposal/importance.py:443-526; the
 * live points under a spline flow); same rows bit for bit.  NESSAI_AMD_TEAM=0 switches it off; 2 / 4 / 8 give a tile
 * to a team of that many computing waves (the chunks of the last conditioner layer dealt round the team, the Jacobian
 * terms summed in single-wave order).  No reference counterpart. */
NSA_API int nsa_last_team(void);

/* Stage buffers of that launch's ring when it ran in the small-batch geometry (0 otherwise): the copy of a stage is
 * requested ring - 1 stages before its arithmetic, so that a small batch runs at the throughput of the CU's LDS-DMA
 * path instead of at the latency of one copy per stage.  NESSAI_AMD_RING pins it (2 = double buffer),
 * NESSAI_AMD_LOADER the loader waves (0 ... 4), NESSAI_AMD_TILES the tiles per workgroup. */
NSA_API int nsa_last_ring(void);

/* Run a whole flow over n rows.
 *
 * Replaces NFlow.forward / NFlow.inverse / NFlow.log_prob /
 * NFlow.forward_and_log_prob (src/nessai/flows/base.py:209-273) and under them
 * glasflow.nflows' CompositeTransform of RandomPermutation / LULinear /
 * {Affine,Additive,PiecewiseRationalQuadratic}CouplingTransform / BatchNorm
 * (call sites src/nessai/flows/realnvp.py:186-213, src/nessai/flows/nsf.py:110-128).
 *
 *   header     first NSA_HEADER_WORDS words of the flow image (host copy)
 *   image_dev  the flow image in HBM (built by nessai_amd.pack)
 *   direction  0 forward (x -> z), 1 inverse (z -> x)
 *   mode       0: x_out = T(x_in), ld_out = log|det J|
 *              1: x_out = z (may be NULL), ld_out = log_prob(x_in)   [direction 0]
 *   x_in_dev   [n][D] float32 row-major
 *   x_out_dev  [n][D] float32 or NULL
 *   ld_out_dev [n] float32 or NULL
 *   status_dev int32[4]; [0] += rows whose spline inverse hit a negative
 *              discriminant (nflows asserts; flowproposal.py:357-364 drops the batch)
 *   waves_per_group  waves (32-row tiles) per workgroup, 0 = default
 *   max_groups       grid cap, 0 = default
 *   stream     hipStream_t
 */
NSA_API int nsa_flow_run(const uint32_t* header, const void* image_dev, int direction, int mode,
                 long long n, const float* x_in_dev, float* x_out_dev, float* ld_out_dev,
                 int* status_dev, int waves_per_group, int max_groups, void* stream);

/* nsa_flow_run for a CONDITIONAL flow (context_features > 0: src/nessai/flows/nsf.py:85-95,
 * flows/realnvp.py:136-146; NFlow.forward / inverse / log_prob with context=..., flows/base.py:209-245):
 *   ctx_in_dev  [n][C] float32, the context row of every input row (C = header word NSA_H_CTX); it reaches
 *               layer 0 of every coupling's conditioner behind the identity features and gates every residual
 *               block, temps * sigmoid(Wc context + bc) (nflows ResidualNet)
 * An unconditional image is refused here, a conditional one by nsa_flow_run. */
NSA_API int nsa_flow_run_ctx(const uint32_t* header, const void* image_dev, int direction, int mode,
                     long long n, const float* x_in_dev, const float* ctx_in_dev, float* x_out_dev,
                     float* ld_out_dev, int* status_dev, int waves_per_group, int max_groups, void* stream);

/* The same batch through n_flows flows of identical structure in ONE launch.
 *
 * Replaces the loop over flows of ImportanceFlowModel.log_prob_all
 * (src/nessai/flowmodel/importance.py:114-129: ``for i, m in enumerate(self.models):
 * log_prob[:, i] = m.log_prob(x)``), the dominant cost of the importance nested sampler's
 * meta-proposal (proposal/importance.py:585-606, 694-735).
 *   header           host copy of flow 0's header; every flow must share its structure
 *                    (D, hidden tiles, transform, bins, precision, stage sizes)
 *   image_list_dev   device array of n_flows device pointers, one flow image each
 *   x_in_dev   [n][D] float32, shared by all flows
 *   x_out_dev  [n_flows][n][D] float32 or NULL
 *   ld_out_dev [n_flows][n] float32 or NULL
 *   other arguments as nsa_flow_run.
 */
NSA_API int nsa_flow_run_many(const uint32_t* header, const void* const* image_list_dev, int n_flows,
                      int direction, int mode, long long n, const float* x_in_dev,
                      float* x_out_dev, float* ld_out_dev, int* status_dev,
                      int waves_per_group, int max_groups, void* stream);

/* Fused draw -> truncate -> inverse flow -> inverse z-score -> prior bounds -> log_q.
 *
 * Replaces one pass of the body of FlowProposal.populate up to compute_weights
 * (src/nessai/proposal/flowproposal/flowproposal.py:431-447), i.e.
 * sample_latent_distribution (proposal/flowproposal/base.py:393-399),
 * LatentRadiusTruncation.apply_latent (truncation.py:358-365),
 * FlowProposal.backward_pass (flowproposal.py:345-389) with
 * FlowModel.inverse (src/nessai/flowmodel/base.py:813-840),
 * latent_log_prob (proposal/flowproposal/base.py:401-414),
 * ScaleAndShift.inverse_reparameterise (reparameterisations/rescale.py:263-291)
 * and Model.in_bounds (src/nessai/model.py:512-518).
 *
 *   z_in_dev    [n][D] float32 latents to use, or NULL to draw them on-device
 *               with Philox4x32-10(seed; counter = row_offset + row, quad)
 *   radius      keep rows with ||z||_2 <= radius (float64 compare)
 *   latent_temperature  T > 0 (1.0 = off): drawn latents are scaled by sqrt(T) and the latent
 *               density is evaluated at z / sqrt(T) minus D log sqrt(T)
 *               (BaseFlowProposal.sample_latent_distribution / latent_log_prob,
 *               proposal/flowproposal/base.py:393-414)
 *   scale/shift [D] float64 z-score sigma / mu        (x = x' * sigma + mu)
 *   lower/upper [D] float64 prior bounds
 *   log_j_rescale  sum_i log|sigma_i| accumulated in the reference's order
 *   x64_out_dev [n][D] float64 samples in the physical space; NULL = the lean pass (log_q_out_dev alone is
 *               written, every test of the row still made: see nsa_populate_passes_lean)
 *   log_q_out_dev [n] float64 log proposal density; NaN marks a dropped row
 *               (outside radius / non-finite / out of bounds / spline failure)
 *   z_out_dev   [n][D] float32 or NULL
 *   prior_kind_dev / prior_par_dev / log_prior_out_dev   optional separable analytic prior
 *               evaluated in float64 in the same pass (replaces the user's vectorised
 *               Model.log_prior for the common bilby-style cases): per dimension a kind
 *               (0 uniform, 1 sine, 2 cosine, 3 power law, 4 Gaussian) and three doubles
 *               {a, b, log normalisation}; log_prior_out_dev [n] receives sum_i log p_i(x_i).
 *               Pass NULL to skip (then give nsa_populate_weights a constant or a host-
 *               evaluated per-row prior).
 *   rp_kind_dev / rp_par_dev   optional per-dimension reparameterisation table (NULL = the
 *               z-score map above for every dimension): kind 0 scale-and-shift (uses
 *               scale/shift), 1 null (null.py), 2 rescale-to-bounds
 *               x = (hi - lo) * (x' - rs) / rf + lo + offset
 *               (RescaleToBounds._inverse_rescale_to_bounds, rescale.py:537-547, 673-700),
 *               3 the same after u = sigmoid(x') with log|J| += log u + log1p(-u)
 *               (utils/rescaling.py:318-337: the registry's "logit" entry,
 *               reparameterisations/__init__.py:62-70), 4 the same after u = exp(x') with
 *               log|J| += x' (the "log-rescale" entry); rp_par_dev [D][5] float64
 *               {lo, hi, rs, rf, offset}.  Boundary inversion, inverse direction
 *               (RescaleToBounds._reverse_inversion, rescale.py:585-612; the registry's
 *               "inversion" / "inversion-duplicate" entries): 5 fold and measure from the
 *               detected edge, u = |x'|, u = 1 - u at the upper edge, x = (hi - lo) u + lo + offset,
 *               rp_par {lo, hi, upper ? 1 : 0, -, offset}; 6 no edge detected,
 *               x = (hi - lo) ((x' + 1) / 2) + lo + offset.  log_j_rescale then carries the
 *               row-independent terms only; the sigmoid terms are added per row in float64.
 */
NSA_API int nsa_populate_draw(const uint32_t* header, const void* image_dev, long long n,
                      const float* z_in_dev, unsigned long long seed,
                      unsigned long long row_offset, double radius, double latent_temperature,
                      const double* scale_dev,
                      const double* shift_dev, const double* lower_dev, const double* upper_dev,
                      double log_j_rescale, double* x64_out_dev, double* log_q_out_dev,
                      float* z_out_dev, const int* prior_kind_dev, const double* prior_par_dev,
                      double* log_prior_out_dev, const int* rp_kind_dev, const double* rp_par_dev,
                      int* status_dev, int waves_per_group, int max_groups, void* stream);

/* Rebuild the parameter words of a flow image on the device after the weights changed (FlowModel.train ->
 * the next FlowProposal.populate; the reference re-reads its torch Parameters directly, here the kernels
 * read a packed image: src/nessai/flowmodel/base.py:681-685 is the matching "flush the cache" point).
 * For a flow of couplings and permutations every image word is constant, one float32 parameter, or the hi /
 * lo 16-bit halves of two parameters; the host computes that map once per structure
 * (nessai_amd/pack.py repack_map) and this one launch replaces the host-side pack (16-33 ms + two copies).
 *   pos_dev[k]   image word to write            kind_dev[k]  1 float32, 2 pair hi halves, 3 pair lo halves
 *   e0_dev[k], e1_dev[k]  indices into params_dev (flat float32 parameter vector), -1 = zero padding
 *   precision    image precision code (0 f32, 1 bf16x3, 2 f16x3; selects bf16 / f16 rounding for kinds 2, 3)
 * The result equals pack.pack_flow of the same weights bit for bit. */
NSA_API int nsa_repack_image(long long n_words, const int* pos_dev, const int* e0_dev, const int* e1_dev,
                     const unsigned char* kind_dev, const float* params_dev, int precision,
                     void* image_dev, void* stream);

/* The probe of the measured precision policy (precision = "auto"; no reference counterpart - the reference
 * has one arithmetic, torch float32: flows/base.py:236-245 is the log_prob the modes are held to): after
 * every change of the weights a few thousand latent rows go through the float32 kernels and through each
 * 16-bit candidate in both directions, and a candidate is kept only if it agrees.
 *   nsa_probe_compare  out_dev[0] (a double, zeroed by the caller) = max over the n rows of
 *                      max(|ld - ld_ref| / max(|log N(z) - ld_ref|, 1e-3), |lp - lp_ref| / max(|lp_ref|, 1e-3)),
 *                      z_dev [n][D] the probe latents, ld the inverse pass's log|det J|, lp the forward
 *                      pass's log_prob; +inf if either holds a NaN.  In float64.
 *   nsa_probe_nudge    x_out = x_in moved one ulp away from zero (the "inputs one ulp further out" twin
 *                      that measures the float32 kernel's own sensitivity)
 *   nsa_probe_range    flag_dev[0] |= 1 when a value is NaN or >= limit in magnitude (the weights of a
 *                      split-f16 image must stay below 65520: pack.split_f16 raises on the host) */
NSA_API int nsa_probe_compare(long long n, int D, const float* z_dev, const float* ld_dev, const float* lp_dev,
                              const float* ld_ref_dev, const float* lp_ref_dev, double* out_dev, void* stream);
NSA_API int nsa_probe_nudge(long long n, const float* x_in_dev, float* x_out_dev, void* stream);
NSA_API int nsa_probe_range(long long n, const float* values_dev, float limit, int* flag_dev, void* stream);

/* The truncation rules that act AFTER the flow, for the trains of passes below
 * (src/nessai/proposal/flowproposal/truncation.py:368-446; nsa_populate_truncate is the one-pass form):
 *   min_log_q        keep log_q > min_log_q (-inf: rule off)
 *   ll_kind          -1: no likelihood rule; 0 / 1: keep rows whose built-in analytic log-likelihood
 *                    (kinds and constants of nsa_log_likelihood) exceeds log_l_threshold
 * A pointer to this struct is optional everywhere (NULL = no rule after the flow). */
typedef struct nsa_post_rules {
  double min_log_q;
  int ll_kind;
  const double* ll_mu_dev;
  const double* ll_sigma_dev;
  double ll_const;
  double log_l_threshold;
} nsa_post_rules;

/* A TRAIN of populate passes with the loop control on the device: for i < n_passes
 * { nsa_populate_draw(rows row_offset + i n ...); nsa_populate_weights; nsa_populate_accept
 *   appending to x_acc; commit }  - enqueued back to back with no host round trip.  Replaces the
 * `while n_accepted < n_samples` loop of FlowProposal.populate (flowproposal.py:431-518) for
 * on-device draws: a pass whose predecessors already reached the wanted count, or the
 * `n_proposed > max_samples` exit (flowproposal.py:499-503), does nothing, so the result and the
 * counters are those of the one-pass-at-a-time loop.
 *   state_dev  int64[8], set by the caller before the FIRST train of a populate call:
 *              [0] accepted so far = 0, [1] passes executed = 0, [2] accepted wanted (n_samples),
 *              [3] rows written to x_acc = 0, [4] scratch, [5] pass limit
 *              (floor(max_samples / n) + 1); read back after the train
 *   max_keys_dev  int64[n_passes], each initialised to nsa_max_key_init()
 *   x_acc_dev  [capacity][D] float64, accepted rows in draw order across the passes
 *   log_prior_const is used when prior_kind_dev is NULL (uniform prior), otherwise the draw
 *   kernel's per-row log_prior_dev
 *   other arguments as nsa_populate_draw / nsa_populate_accept.
 */
NSA_API int nsa_populate_passes(const uint32_t* header, const void* image_dev, long long n, int n_passes,
                        unsigned long long seed, unsigned long long row_offset, double radius,
                        double latent_temperature, const double* scale_dev, const double* shift_dev,
                        const double* lower_dev, const double* upper_dev, double log_j_rescale,
                        double* x64_dev, double* log_q_dev, const int* prior_kind_dev,
                        const double* prior_par_dev, double* log_prior_dev, double log_prior_const,
                        const int* rp_kind_dev, const double* rp_par_dev, double* log_w_dev,
                        long long* max_keys_dev, long long capacity, double* x_acc_dev,
                        long long* state_dev, void* workspace_dev, size_t workspace_bytes,
                        int* status_dev, int waves_per_group, int max_groups, void* stream,
                        const nsa_post_rules* rules);

/* nsa_populate_passes for SMALL passes (n <= 16384 rows - the reference's default drawsize = nlive): a pass
 * of a few thousand rows is one dependent chain through the flow (~40 us) however few rows it has, and the
 * passes of a train are independent until the stop test, so the train is ONE draw launch over its
 * n_passes * n rows (pass i = rows [row_offset + i n, + n)), one tail launch (workgroup i: weights, the pass's
 * own maximum, accept, compaction into stage_dev + i * stage_rows * D; counts_dev[i]) and one commit launch
 * that walks the passes in order against the loop state.  Rows, order, counters and Philox offsets are those
 * of nsa_populate_passes; passes beyond the one that reaches the wanted count are computed and ignored.
 *   x64_many_dev [n_passes n][D], log_q_many_dev / log_prior_many_dev [n_passes n],
 *   stage_dev [n_passes][stage_rows][D] with stage_rows >= min(n, capacity), counts_dev [n_passes];
 *   n_passes <= 64. */
NSA_API int nsa_populate_passes_small(const uint32_t* header, const void* image_dev, long long n, int n_passes,
                              unsigned long long seed, unsigned long long row_offset, double radius,
                              double latent_temperature, const double* scale_dev, const double* shift_dev,
                              const double* lower_dev, const double* upper_dev, double log_j_rescale,
                              double* x64_many_dev, double* log_q_many_dev, const int* prior_kind_dev,
                              const double* prior_par_dev, double* log_prior_many_dev, double log_prior_const,
                              const int* rp_kind_dev, const double* rp_par_dev, long long stage_rows,
                              double* stage_dev, long long* counts_dev, long long capacity, double* x_acc_dev,
                              long long* state_dev, int* status_dev, int waves_per_group, int max_groups,
                              void* stream, const nsa_post_rules* rules);

/* The same pass in pieces, for draw batches SHARDED over ranks: the collectives of the host
 * framework (RCCL through torch.distributed, stream ordered, no host synchronisation) go between
 * them:  head (draw + weights + this rank's maximum)  ->  all-reduce(max) of max_key_dev  ->
 * tail (accept + append; state[4] = this rank's count)  ->  all-reduce(sum) of a copy of state[4]
 * ->  commit (state[0] += the global count, history[pass] = this rank's count).  Every piece
 * honours the loop state like nsa_populate_passes.  flowproposal.py:431-518 + SURVEY.md §8e. */
NSA_API int nsa_populate_pass_head(const uint32_t* header, const void* image_dev, long long n,
                           unsigned long long seed, unsigned long long row_offset, double radius,
                           double latent_temperature, const double* scale_dev, const double* shift_dev,
                           const double* lower_dev, const double* upper_dev, double log_j_rescale,
                           double* x64_dev, double* log_q_dev, const int* prior_kind_dev,
                           const double* prior_par_dev, double* log_prior_dev, double log_prior_const,
                           const int* rp_kind_dev, const double* rp_par_dev, double* log_w_dev,
                           long long* max_key_dev, const long long* state_dev, int* status_dev,
                           int waves_per_group, int max_groups, void* stream);
NSA_API int nsa_populate_pass_tail(long long n, int D, const double* log_w_dev, const long long* max_key_dev,
                           unsigned long long seed, unsigned long long row_offset, const double* x64_dev,
                           long long capacity, double* x_acc_dev, long long* state_dev,
                           void* workspace_dev, size_t workspace_bytes, void* stream);
NSA_API int nsa_populate_pass_commit(long long* state_dev, long long capacity,
                             const long long* global_count_dev, long long* history_dev, void* stream);

/* A TRAIN of sharded passes with two collectives per train instead of two per pass
 * (flowproposal.py:431-518 + SURVEY.md §8e; the passes of a train are independent until the stop test):
 *   nsa_populate_train_heads   n_passes heads back to back: pass i draws rows
 *                              [row_offset + i * row_stride, + n) into x64_many_dev + i n D and
 *                              log_w_many_dev + i n, its maximum into max_keys_dev[i]
 *   -> ONE all-reduce(max) of max_keys_dev[0 .. n_passes)
 *   nsa_populate_train_tails   n_passes accept + append steps against the global maxima, in pass order,
 *                              on a scratch copy of the loop state that never stops;
 *                              local_counts_dev[i] = this rank's accepted rows of pass i
 *   -> ONE all-reduce(sum) of a copy of local_counts_dev[0 .. n_passes) (global_counts_dev)
 *   nsa_populate_train_commit  walks the passes in order against the REAL state: pass i counts
 *                              (state[0] += global count, state[1] += 1, history[pass] = local count,
 *                              state[3] advances) until a stop condition holds; the rows later passes
 *                              appended to x_acc_dev lie beyond state[3] and are never read.
 * Rows, order, counters and Philox offsets equal those of the one-pass-at-a-time pieces above; a train
 * that overshoots only spends the kernels of its surplus passes.  scratch_dev: int64[8]. */
NSA_API int nsa_populate_train_heads(const uint32_t* header, const void* image_dev, long long n, int n_passes,
                             unsigned long long seed, unsigned long long row_offset,
                             unsigned long long row_stride, double radius, double latent_temperature,
                             const double* scale_dev, const double* shift_dev, const double* lower_dev,
                             const double* upper_dev, double log_j_rescale, double* x64_many_dev,
                             double* log_q_dev, const int* prior_kind_dev, const double* prior_par_dev,
                             double* log_prior_dev, double log_prior_const, const int* rp_kind_dev,
                             const double* rp_par_dev, double* log_w_many_dev, long long* max_keys_dev,
                             const long long* state_dev, int* status_dev, int waves_per_group,
                             int max_groups, void* stream, const nsa_post_rules* rules);
NSA_API int nsa_populate_train_tails(long long n, int D, int n_passes, const double* log_w_many_dev,
                             const long long* max_keys_dev, unsigned long long seed,
                             unsigned long long row_offset, unsigned long long row_stride,
                             const double* x64_many_dev, long long capacity, double* x_acc_dev,
                             const long long* state_dev, long long* scratch_dev,
                             long long* local_counts_dev, void* workspace_dev, size_t workspace_bytes,
                             void* stream);
NSA_API int nsa_populate_train_commit(long long* state_dev, long long capacity, int n_passes,
                              const long long* local_counts_dev, const long long* global_counts_dev,
                              long long* history_dev, void* stream);

/* LEAN trains of passes: the same loops WITHOUT materialised candidates.
 *
 * Only accepted rows are ever used (`samples[...] = x[accept][:m]`, flowproposal.py:492-498) - at config 3's
 * acceptance that is ~86 of the 1e6 rows of a pass - and a candidate row is a pure function of its Philox
 * counter (the global row) and the flow.  So a lean pass keeps every row's float64 arithmetic (inverse
 * reparameterisation, prior bounds, separable prior, log q, the min_log_q rule) in registers and writes the
 * weight log_w = log_prior - log_q alone - 8 bytes per row instead of 8 (D + 1), the pass's maximum through one
 * atomic per wave (log_q_dev is not written); the accept step lists the accepted rows by their GLOBAL Philox rows in idx_acc_dev
 * (int64, draw order across the passes, same loop state as nsa_populate_passes), and nsa_populate_replay
 * computes exactly those rows again - the same kernel instance reading row_list_dev[r] in place of
 * row_offset + r - into x64_out_dev [m][D] (and log_q_out_dev [m]), bit for bit the rows the full pass would
 * have written.  Not for the likelihood-threshold rule (rules->ll_kind >= 0 is refused: it needs the
 * candidate rows) nor for accumulate_weights, which keep nsa_populate_passes / the arena.
 *   nsa_populate_passes_lean       = nsa_populate_passes minus x64_dev, with idx_acc_dev [capacity] for x_acc_dev
 *   nsa_populate_train_heads_lean  = nsa_populate_train_heads minus x64_many_dev
 *   nsa_populate_train_tails_lean  = nsa_populate_train_tails minus x64_many_dev, with idx_acc_dev for x_acc_dev
 *   nsa_populate_replay            m rows of row_list_dev through the populate pass (arguments as
 *                                  nsa_populate_draw; the latent radius test is repeated and holds) */
/* The head of ONE lean pass on its own (what nsa_populate_passes_lean enqueues per pass before the accept
 * step): draw -> radius -> inverse flow -> inverse reparameterisation -> bounds -> log q, then
 * log_w = log_prior - log_q with -inf for dropped rows (FlowProposal.compute_weights,
 * proposal/flowproposal/base.py:1069-1098) after the min_log_q rule (truncation.py:385-392; -inf = off), and
 * *max_key_dev raised to the largest weight of the launch (initialise it with nsa_max_key_init();
 * flowproposal.py:492 `log_w -= log_w.max()`).  8 bytes per row leave the chip.  log_prior is the separable
 * device prior when prior_kind_dev is given (log_prior_out_dev [n] also receives it), else log_prior_const. */
NSA_API int nsa_populate_draw_weights(const uint32_t* header, const void* image_dev, long long n,
                              unsigned long long seed, unsigned long long row_offset, double radius,
                              double latent_temperature, const double* scale_dev, const double* shift_dev,
                              const double* lower_dev, const double* upper_dev, double log_j_rescale,
                              const int* prior_kind_dev, const double* prior_par_dev, double* log_prior_out_dev,
                              double log_prior_const, double min_log_q, const int* rp_kind_dev,
                              const double* rp_par_dev, double* log_w_out_dev, long long* max_key_dev,
                              int* status_dev, int waves_per_group, int max_groups, void* stream);
NSA_API int nsa_populate_passes_lean(const uint32_t* header, const void* image_dev, long long n, int n_passes,
                             unsigned long long seed, unsigned long long row_offset, double radius,
                             double latent_temperature, const double* scale_dev, const double* shift_dev,
                             const double* lower_dev, const double* upper_dev, double log_j_rescale,
                             double* log_q_dev, const int* prior_kind_dev, const double* prior_par_dev,
                             double* log_prior_dev, double log_prior_const, const int* rp_kind_dev,
                             const double* rp_par_dev, double* log_w_dev, long long* max_keys_dev,
                             long long capacity, long long* idx_acc_dev, long long* state_dev,
                             void* workspace_dev, size_t workspace_bytes, int* status_dev,
                             int waves_per_group, int max_groups, void* stream, const nsa_post_rules* rules);
NSA_API int nsa_populate_replay(const uint32_t* header, const void* image_dev, long long m,
                        const long long* row_list_dev, unsigned long long seed, double radius,
                        double latent_temperature, const double* scale_dev, const double* shift_dev,
                        const double* lower_dev, const double* upper_dev, double log_j_rescale,
                        double* x64_out_dev, double* log_q_out_dev, const int* prior_kind_dev,
                        const double* prior_par_dev, double* log_prior_out_dev, const int* rp_kind_dev,
                        const double* rp_par_dev, int* status_dev, int waves_per_group, int max_groups,
                        void* stream);
NSA_API int nsa_populate_train_heads_lean(const uint32_t* header, const void* image_dev, long long n, int n_passes,
                                  unsigned long long seed, unsigned long long row_offset,
                                  unsigned long long row_stride, double radius, double latent_temperature,
                                  const double* scale_dev, const double* shift_dev, const double* lower_dev,
                                  const double* upper_dev, double log_j_rescale, double* log_q_dev,
                                  const int* prior_kind_dev, const double* prior_par_dev, double* log_prior_dev,
                                  double log_prior_const, const int* rp_kind_dev, const double* rp_par_dev,
                                  double* log_w_many_dev, long long* max_keys_dev, const long long* state_dev,
                                  int* status_dev, int waves_per_group, int max_groups, void* stream,
                                  const nsa_post_rules* rules);
NSA_API int nsa_populate_train_tails_lean(long long n, int D, int n_passes, const double* log_w_many_dev,
                                  const long long* max_keys_dev, unsigned long long seed,
                                  unsigned long long row_offset, unsigned long long row_stride,
                                  long long capacity, long long* idx_acc_dev, const long long* state_dev,
                                  long long* scratch_dev, long long* local_counts_dev, void* workspace_dev,
                                  size_t workspace_bytes, void* stream);

/* accumulate_weights = True (flowproposal.py:471-476): out_dev[i] = sum over the n rows of pass i of
 * exp(log_w - max_i), max_i = decoded max_keys_dev[i] - what pass i adds to the expected number of
 * accepted rows `exp(logsumexp(log_weights - log_constant))`; dropped rows (-inf) add nothing.  Fixed
 * summation order.  The host (or the ranks of a sharded proposal, after one all-reduce of the maxima and
 * one of these sums) folds the passes into the running log_constant. */
NSA_API int nsa_populate_sumexp(long long n, int n_passes, const double* log_w_many_dev,
                        const long long* max_keys_dev, double* out_dev, void* stream);

/* Importance nested sampler (config 4), ImportanceFlowProposal.draw
 * (src/nessai/proposal/importance.py:443-487), first half: draw rows [row_offset, row_offset + n)
 * from ONE flow and map them to the unit hypercube.  Same fused kernel as nsa_populate_draw
 * (Philox latents indexed by the global row, or z_in_dev), with the importance epilogue:
 *   x64_out_dev [n][D]  x = 1 / (1 + exp(-x'))        (inverse_rescale, utils/rescaling.py:316-336)
 *   xp_out_dev  [n][D]  x' in float32, logical column order: the point every stored flow is then
 *                       evaluated at (nsa_flow_run_many), as flowmodel/importance.py:114-129 does
 *   log_j_out_dev [n]   log|J| of the logit map at x clipped to [unit_eps, 1 - unit_eps]
 *                       (rescale, utils/rescaling.py:290-313; eps = 1e-8, config.py:159);
 *                       NaN marks a row that fails the reference's checks (non-finite x', log|J|
 *                       or log|J_inv|, x outside [0, 1]: importance.py:476-483) */
NSA_API int nsa_importance_draw(const uint32_t* header, const void* image_dev, long long n,
                                const float* z_in_dev, unsigned long long seed, unsigned long long row_offset,
                                double unit_eps, double* x64_out_dev, float* xp_out_dev, double* log_j_out_dev,
                                int* status_dev, int waves_per_group, int max_groups, void* stream);

/* Unit hypercube -> x' of the importance sampler for stored samples: x' = logit(clip(x, eps, 1 - eps))
 * (float32, what the flows read) and log|J| = sum_i -log(xc_i) - log1p(-xc_i) (float64).
 *
 * Replaces ImportanceFlowProposal.to_prime / rescale (proposal/importance.py:219-245, the "logit"
 * branch; utils/rescaling.py:290-313) on the path of update_log_q (importance.py:528-543), which the
 * sampler runs over EVERY stored sample at every level. */
NSA_API int nsa_importance_to_prime(long long n, int dims, const double* x_unit_dev, double eps,
                                    float* x_prime_dev, double* log_j_dev, void* stream);

/* ... second half, compute_log_Q (importance.py:415-441): log_q_all[r][0] = 0 (the prior),
 * log_q_all[r][1 + j] = log_prob_all[j][r] + log_j[r], log_Q[r] = logsumexp(log_q_all[r], b = weights)
 * in float64 (scipy.special.logsumexp's arithmetic).  log_prob_all_dev is the (n_flows, n) float32
 * output of nsa_flow_run_many; weights_dev has n_flows + 1 entries (prior first). */
NSA_API int nsa_importance_mix(long long n, int n_flows, const float* log_prob_all_dev, const double* log_j_dev,
                               const double* weights_dev, double* log_q_all_dev, double* log_Q_dev,
                               void* stream);

/* nsa_importance_mix for the rows of ONE draw batch, written as one record per row so that the host reads the
 * batch back with a single copy (ImportanceFlowProposal.draw, proposal/importance.py:443-526: the first filter,
 * the NaN check of compute_log_Q and the structured-array fill then run on that copy):
 *   packed_dev [n][dims + n_flows + 3] float64 = { x (clipped to [0, 1] when clip != 0: importance.py:470-471),
 *   log_q_all (n_flows + 1 columns), log_Q, log_j }; log_j is NaN for the rows nsa_importance_draw dropped. */
NSA_API int nsa_importance_mix_packed(long long n, int n_flows, int dims, const float* log_prob_all_dev,
                              const double* log_j_dev, const double* weights_dev, const double* x_unit_dev,
                              int clip, double* packed_dev, void* stream);

/* nsa_importance_mix_packed with the records in the LIVE-POINT WIRE FORMAT (the structured dtype of the importance
 * sampler: dims float64 parameters, then logP f8, logL f8, it i4, logQ f8, logW f8, logU f8 - 8 dims + 44 bytes per
 * row, packed), for models with a uniform prior on the box [lower, upper] whose prior methods are the framework's own:
 * the fields ImportanceFlowProposal.draw fills after the batch (proposal/importance.py:488-503) -
 *   logP = Model.batch_evaluate_log_prior(Model.from_unit_hypercube(points)) = log(in_bounds(lower + (upper - lower) x))
 *          + log_prior_total (model.py:497-518; the caller passes its own -sum log(upper - lower)),
 *   logU = Model.batch_evaluate_log_prior_unit_hypercube(points) = log(in_unit_hypercube(x)) (model.py:520-620),
 *   logQ = the meta-proposal, logW = logU - logQ, logL = NaN, it = 0 -
 * are computed per row on the device (float64, numpy's comparisons) and the host VIEWS the read-back as the array it
 * returns; aux_dev [n][n_flows + 2] float64 = { log_q_all (n_flows + 1 columns), log_j } (log_j NaN: first filter). */
NSA_API int nsa_importance_mix_points(long long n, int n_flows, int dims, const float* log_prob_all_dev,
                              const double* log_j_dev, const double* weights_dev, const double* x_unit_dev,
                              int clip, const double* lower_dev, const double* upper_dev, double log_prior_total,
                              void* points_dev, double* aux_dev, void* stream);

/* log_w = log_prior - log_q and its maximum.
 *
 * Replaces BaseFlowProposal.compute_weights (proposal/flowproposal/base.py:1069-1098)
 * and `log_w.max()` (flowproposal.py:492).  log_prior_dev may be NULL, in which
 * case log_prior_const is used for every kept row (uniform box priors).
 * Dropped rows (NaN log_q) get log_w = -inf.  max_key_dev is an int64 holding
 * the running maximum in a sign-magnitude-ordered encoding (initialise with
 * nsa_max_key_init(); atomically updated, so several calls / shards accumulate).
 */
NSA_API int nsa_populate_weights(long long n, const double* log_q_dev, const double* log_prior_dev,
                         double log_prior_const, double* log_w_dev, long long* max_key_dev,
                         void* stream);
/* The truncation rules that act after the flow: MinLogQTruncation.apply_after_backward
 * (`log_q > min_log_q`, proposal/flowproposal/truncation.py:385-392) and
 * LikelihoodThresholdTruncation.apply_after_likelihood (`logL > threshold`, :423-430).  A row that
 * fails is marked dropped (log_q = NaN), exactly like a row outside the prior bounds.  Pass
 * min_log_q = -inf and/or log_l_dev = NULL for an absent rule.  Call between nsa_populate_draw
 * and nsa_populate_weights. */
NSA_API int nsa_populate_truncate(long long n, double* log_q_dev, double min_log_q, const double* log_l_dev,
                                  double log_l_threshold, void* stream);
NSA_API long long nsa_max_key_init(void);
NSA_API long long nsa_max_key_encode(double v);
NSA_API double nsa_max_key_decode(long long key);

/* Rejection step + order-preserving compaction.
 *
 * Replaces `log_w -= log_w.max(); log_u = log(rng.random(n)); accept = log_w > log_u;
 * samples[...] = x[accept][:m]` (flowproposal.py:492-498).
 *
 *   max_key_dev  encoded global maximum (after any cross-rank all-reduce)
 *   u_in_dev     [n] float64 uniforms to use, or NULL to draw them on-device
 *                (Philox stream 1, counter = row_offset + row)
 *   capacity     rows available in x_acc_dev / idx_acc_dev
 *   x_acc_dev    [capacity][D] float64: accepted rows, in draw order
 *   idx_acc_dev  [capacity] int64 source row of each accepted row (may be NULL)
 *   n_accept_dev int64[1]: total accepted (can exceed capacity)
 *   workspace_dev / workspace_bytes   scratch, >= nsa_populate_accept_workspace(n)
 */
NSA_API size_t nsa_populate_accept_workspace(long long n);
NSA_API int nsa_populate_accept(long long n, int D, const double* log_w_dev, const long long* max_key_dev,
                        unsigned long long seed, unsigned long long row_offset,
                        const double* u_in_dev, const double* x64_dev, long long capacity,
                        double* x_acc_dev, long long* idx_acc_dev, long long* n_accept_dev,
                        void* workspace_dev, size_t workspace_bytes, void* stream);

/* Analytic benchmark likelihoods on device rows (the pool right after compaction).
 *
 * Replaces Model.batch_evaluate_log_likelihood (src/nessai/model.py:644-677) at
 * flowproposal.py:521 for models that declare one of the built-in forms; float64 in numpy's
 * reduction order, so the values equal the host evaluation bit for bit.
 *   kind 0  Rosenbrock (examples/rosenbrock.py:33-45); kind 1 diagonal Gaussian
 *           -0.5 sum ((x - mu) / sigma)^2 - constant  (mu_dev / sigma_dev [D], or both NULL)
 *   x64_dev [n][D] float64, out_dev [n] float64; 2 <= D <= 129
 */
NSA_API int nsa_log_likelihood(int kind, long long n, int D, const double* x64_dev, const double* mu_dev,
                               const double* sigma_dev, double constant, double* out_dev, void* stream);

/* ---- training (nessai_amd/csrc/nsa_train.hip) ---------------------------------
 *
 * One minibatch of FlowModel._train (src/nessai/flowmodel/base.py:365-452):
 * forward, loss = -mean(log_prob(x)) (flows/base.py:236-245), backward,
 * clip_grad_norm_(clip_norm) and torch.optim.AdamW.step(), enqueued on `stream`
 * with no host round trip.
 *
 *   plan        int64 host array describing the flow: header {D, hidden, residual
 *               blocks, coupling layers, transform, bins, activation, params per
 *               feature, n trainable floats, record words, table offset of the weight
 *               transposition map, conditioner kind, table offset of the final column
 *               table} then one record per
 *               coupling layer {d_id, d_tr, id table offset, tr table offset,
 *               flat-buffer offsets of W0, b0, {W1, b1, W2, b2} x blocks, Wf, bf}
 *   tables_dev  int32 physical column tables (permutations folded in)
 *   consts      float host array {tail_bound, min_bin_width, min_bin_height,
 *               min_derivative, boundary derivative, 1/sqrt(hidden), 1/var, log_z}
 *   params/grads/adam_m/adam_v   flat float32 buffers (n trainable floats)
 *   x_dev       [n][D] float32 minibatch;  step = 1-based optimiser step count
 *   weights_dev [n] float32 sample weights for the weighted loss
 *               -sum(w log_prob) / sum(w) (flowmodel/base.py:404-407), or NULL
 *   optimiser   0 AdamW, 1 Adam (L2 weight decay), 2 SGD without momentum - the three choices of
 *               FlowModel.get_optimiser (flowmodel/base.py:105-135)
 *   loss_acc_dev  float[1], the minibatch loss is ADDED (read once per epoch)
 *
 * The workspace (nsa_train_workspace_floats floats, 16-byte aligned blocks) holds the batch, the saved
 * activations of every layer and, for conditioners up to 64 wide, one copy of the flat gradient per
 * 32-row workgroup of the fused backward kernels (summed in a fixed order at the end of the pass; left
 * out, in favour of float atomics, above 2^26 floats).
 */
NSA_API size_t nsa_train_workspace_floats(const long long* plan, long long n);
/* Conditional flows (plan[13] = context features C > 0; FlowModel.train(conditional=...),
 * src/nessai/flowmodel/base.py:286-302, 404-436): the [n][C] float32 context rows of a batch are handed to
 * every training entry point INSIDE the workspace, this many floats from its start - written by the caller
 * before nsa_train_step / _step_dev / _eval / _forward (nsa_train_backward finds them where the forward pass
 * left them). */
NSA_API size_t nsa_train_context_offset(const long long* plan, long long n);
NSA_API int nsa_train_step(const long long* plan, const int* tables_dev, const float* consts,
                           float* params_dev, float* grads_dev, float* adam_m_dev, float* adam_v_dev,
                           const float* x_dev, const float* weights_dev, long long n,
                           float* workspace_dev, float lr, float beta1, float beta2, float eps,
                           float weight_decay, float clip_norm, long long step, int optimiser,
                           float* loss_acc_dev, void* stream);

/* The same minibatch with the learning rate and the optimiser step counter read from device memory
 * (opt_dev: float[4] = {lr, steps done so far, scratch, scratch}; the call increments [1]).  No
 * argument changes from step to step, so the launches of a whole epoch - the reference runs two
 * minibatches and one validation pass per epoch on 2000 live points, flowmodel/base.py:195-236,
 * 607-640 - can be captured once into a HIP graph and replayed with one launch. */
NSA_API int nsa_train_step_dev(const long long* plan, const int* tables_dev, const float* consts,
                               float* params_dev, float* grads_dev, float* adam_m_dev, float* adam_v_dev,
                               const float* x_dev, const float* weights_dev, long long n, float* workspace_dev,
                               float* opt_dev, float beta1, float beta2, float eps, float weight_decay,
                               float clip_norm, int optimiser, float* loss_acc_dev, void* stream);
/* Forward only (FlowModel._validate, flowmodel/base.py:454-523):
 * loss_acc_dev[0] += -mean(log_prob(x)); log_prob_out_dev [n] may be NULL. */
NSA_API int nsa_train_eval(const long long* plan, const int* tables_dev, const float* consts,
                           const float* params_dev, const float* x_dev, const float* weights_dev,
                           long long n, float* workspace_dev, float* loss_acc_dev,
                           float* log_prob_out_dev, void* stream);

/* Differentiable evaluation for torch autograd (the B1 surface under the reference's own
 * training loop, src/nessai/flowmodel/base.py:400-444: `loss = -model.log_prob(x).mean();
 * loss.backward()`).  nsa_train_forward runs NFlow.forward / log_prob (flows/base.py:209-245) in
 * exact float32 and keeps the activations of every layer in `workspace_dev`
 * (nsa_train_workspace_floats(plan, n) floats, owned by the caller until the matching backward
 * call); `training` != 0 selects batch statistics for batch norm (and updates the running ones)
 * and runs the data-dependent act-norm initialisation when the plan says it is due.
 *   z_out_dev [n][D] (logical column order), logdet_out_dev [n], log_prob_out_dev [n]: any may be NULL
 * nsa_train_backward (same `training` flag as the forward call) takes the upstream gradients of
 * those outputs (any may be NULL) and writes
 * the gradient of EVERY trainable float to grads_dev (overwritten, same layout as params_dev)
 * and, if gx_out_dev is not NULL, the gradient with respect to the input rows [n][D]. */
NSA_API int nsa_train_forward(const long long* plan, const int* tables_dev, const float* consts,
                              float* params_dev, const float* x_dev, long long n, float* workspace_dev,
                              int training, float* z_out_dev, float* logdet_out_dev,
                              float* log_prob_out_dev, void* stream);
NSA_API int nsa_train_backward(const long long* plan, const int* tables_dev, const float* consts,
                               const float* params_dev, float* grads_dev, long long n,
                               float* workspace_dev, int training, const float* g_z_dev,
                               const float* g_logdet_dev, const float* g_log_prob_dev, float* gx_out_dev,
                               void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NESSAI_AMD_H */
