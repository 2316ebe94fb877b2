// Flow image format shared by the host packer (nessai_amd/pack.py) and the
// gfx950 kernels.  A flow (RealNVP / NeuralSplineFlow of the reference,
// src/nessai/flows/realnvp.py:76-214, src/nessai/flows/nsf.py:60-130) is
// compiled on the host into ONE flat buffer of 32-bit words:
//
//   header | op table (forward) | op table (inverse) | stage table (forward)
//   | stage table (inverse) | output column tables | stage blobs
//
// The kernel executes the op table of the requested direction in order.  Every
// op reads its index tables / biases / MFMA-packed weights from "stages":
// contiguous blobs copied global -> LDS (double-buffered) ahead of use.
// Permutations never move data: they are folded into the column tables.
#pragma once
#include <stdint.h>

#define NSA_MAGIC 0x3141534e /* "NSA1" */
#define NSA_VERSION 10

// header word indices
enum {
  NSA_H_MAGIC = 0,
  NSA_H_VERSION = 1,
  NSA_H_D = 2,             // features
  NSA_H_HT = 3,            // hidden tiles: hidden width padded to 32*HT
  NSA_H_TRANSFORM = 4,     // NSA_T_*
  NSA_H_BINS = 5,          // spline bins K (RQ only)
  NSA_H_NOPS_FWD = 6,
  NSA_H_OPS_FWD = 7,       // word offset
  NSA_H_NOPS_INV = 8,
  NSA_H_OPS_INV = 9,
  NSA_H_NSTAGES_FWD = 10,
  NSA_H_STAGES_FWD = 11,   // word offset, 2 words per stage: {src word offset, n words}
  NSA_H_NSTAGES_INV = 12,
  NSA_H_STAGES_INV = 13,
  NSA_H_MAX_STAGE = 14,    // words of the largest stage, rounded up to a multiple of 256
  NSA_H_TOTAL = 15,        // total words
  NSA_H_OUT_FWD = 16,      // word offset: D ints, physical column of logical output i
  NSA_H_OUT_INV = 17,
  NSA_H_LOG_Z = 18,        // float: 0.5*D*log(2*pi*var) rounded to f32
  NSA_H_TAIL_BOUND = 19,   // float
  NSA_H_MIN_BW = 20,       // float min_bin_width
  NSA_H_MIN_BH = 21,       // float min_bin_height
  NSA_H_MIN_D = 22,        // float min_derivative
  NSA_H_EDGE_RAW = 23,     // float: log(exp(1-min_derivative)-1) rounded to f32
  NSA_H_LOGIT_DIV = 24,    // float: sqrt(hidden_features) or 1.0 (no scaling)
  NSA_H_ACT = 25,          // NSA_ACT_*
  NSA_H_NET = 26,          // NSA_NET_*
  NSA_H_NLAYERS = 27,      // residual blocks (resnet) / hidden layers (mlp)
  NSA_H_BASE_INV_VAR = 28, // float: 1/var of the base Gaussian (1.0 for StandardNormal)
  NSA_H_PREC = 29,         // NSA_PREC_*: arithmetic of the conditioner GEMMs
  NSA_H_AFF_A = 30,        // float: affine coupling scale = sigmoid(raw + a) + b (default 2)
  NSA_H_AFF_B = 31,        // float: ... b (default 1e-3)
  NSA_H_RES_FWD = 32,      // words of ALL stage blobs of the forward program, which lie back to back in stage
                           // order from stage 0's offset on (the weights-resident geometry copies that range
                           // to LDS once per workgroup); 0 = not contiguous, staged geometries only
  NSA_H_RES_INV = 33,      // the same for the inverse program
  NSA_H_CTX = 34,          // context features C of a conditional flow (0: none).  The context of a row lies in
                           // columns D + 2 ... D + 1 + C of the x tile (row stride (D + 2 + C) | 1); layer 0 of
                           // every coupling gathers it behind the identity features, every residual block gates
                           // its output with sigmoid(Wc context + bc) (format 10)
  NSA_H_BNW = 35,          // 1: batch norm within the residual blocks (ResidualNet use_batch_norm, eval mode): every
                           // block starts with a stage {a[32 HT], c[32 HT]} - temps = act(a h + c) - and its second
                           // batch norm is folded into linear 0 by the packer
  NSA_H_WORDS = 40
};

enum { NSA_T_NONE = 0, NSA_T_RQ = 1, NSA_T_AFFINE = 2, NSA_T_ADDITIVE = 3 };
enum { NSA_ACT_RELU = 0, NSA_ACT_TANH = 1, NSA_ACT_SILU = 2 };
enum { NSA_NET_RESNET = 0, NSA_NET_MLP = 1 };
enum { NSA_OP_COUPLING = 1, NSA_OP_DENSE = 2, NSA_OP_ELEMENTWISE = 3 };
// NSA_PREC_F32   : v_mfma_f32_32x32x2_f32, exact f32 products (runs on the f32 vector ALUs)
// NSA_PREC_BF16X3: operands split x = hi + lo (two bf16), products hi*hi + hi*lo + lo*hi on
//                  v_mfma_f32_32x32x16_bf16 with f32 accumulation (relative error ~2^-16 per
//                  product; log_prob stays within the 1e-5 parity bar)
// NSA_PREC_F16X3 : the same three products on f16 pairs (v_mfma_f32_32x32x16_f16): 11 + 11 significant
//                  bits carry an operand to 2^-24 relative (2^-25 absolute below 2^-3) - float32-class
//                  products at the bf16 rate - for operands below 65520 in magnitude; a launch that
//                  splits a larger value says so in status[1]
enum { NSA_PREC_F32 = 0, NSA_PREC_BF16X3 = 1, NSA_PREC_F16X3 = 2 };

// op record: 4 words
//   COUPLING    {kind, n_steps0 (multiple of 4), n_chunks, n_steps of the context gate (0: unconditional)}
//   DENSE       {kind, n_steps  (multiple of 4), n_out_tiles, logdet (float bits)}
//   ELEMENTWISE {kind, 0, 0, logdet (float bits)}
#define NSA_OP_WORDS 4

// Stage blob layouts (word offsets inside the blob; every section is padded
// to a multiple of 4 words, every blob to a multiple of 32 words = one 128-byte line; the copies move
// 1 KiB pieces, so the last piece of a stage runs over into the next blob - or into the 1 KiB of slack
// behind the last one - and over the stage's end in an LDS buffer that is sized in whole pieces):
//   coupling layer 0 : id_tab[2*n_steps0] | bias[32*HT] | W[HT][n_steps0/4][64][4]
//   hidden layer     : bias[32*HT]        | W[HT][4*HT/KS][64][4]      (sub-stage 0 of KS; the others: W only)
//   context gate     : ctx_tab[2*n_steps_c] | bias[32*HT] | W[HT][n_steps_c/4][64][4]   (conditional flows: one stage
//                      behind the second layer of every residual block, laid out like layer 0; its table holds
//                      the byte offsets of the context columns)
//   final chunk      : tr_tab[pad4(2*FPC)] | bias[32*C] | W[C][4*HT/KS][64][4]   (likewise)
//   KS = nsa_ksplit(output tiles, HT): a GEMM whose packed weights exceed 16 KiB is dealt to 2 or 4
//   consecutive stages along k (its accumulators live across them).
//   dense            : in_tab[2*n_steps] | pre[2*n_steps] | bias[32] | W[1][n_steps/4][64][4]  - ONE OUTPUT TILE per
//                      stage, OT stages per op, each with its own copy of the tables (format 9; up to 128 features)
//   elementwise      : a[D4] | b[D4] | c[D4] | d[D4]  (D4 = pad4(D); y = ((x-a)/b)*c + d per physical column)
//
// MFMA packing (v_mfma_f32_32x32x2_f32): W[ot][s4][lane][e] with step s = 4*s4+e,
// lane = 32*half + i holds weight(out_row(ot, i), in_col(s, half)).
//   in_col from registers (hidden/final): 32*(s>>4) + (r&3) + 8*(r>>2) + 4*half, r = s&15
//   in_col from the x tile (layer 0, dense): 2*s + half
// id_tab / in_tab entries are BYTE offsets into a row of the x tile (4 * physical column); an unused entry
// holds 4 * (D + 1), the row's always-zero cell (format 8; it was -1 and cost the kernel two selects).
// bias[ot][half][r] is the bias of out feature 32*ot + (r&3) + 8*(r>>2) + 4*half.
//
// NSA_PREC_BF16X3 / NSA_PREC_F16X3 images keep the same tables and biases but count k in steps of 16
// (v_mfma_f32_32x32x16_bf16 / _f16: lane = 32*half + i holds 8 consecutive k of its half) and
// store W as bf16 (f16) pairs: W[ot][s][part][64][8], part 0 = hi, part 1 = lo, with
//   in_col from registers: 32*(s>>1) + (r&3) + 8*(r>>2) + 4*half, r = 8*(s&1) + e
//   in_col from the x tile: 16*s + 8*half + e            (tables hold 16 entries per step)

// flags for nsa_flow_* entry points
#define NSA_DIR_FORWARD 0
#define NSA_DIR_INVERSE 1
