"""Compile a flow description into the device image consumed by the HIP kernels.

The image format is specified in ``nessai_amd/csrc/nsa_format.h``.  The packer
is pure host-side bookkeeping (numpy): it folds permutations into column
tables, lays the conditioner weights out in the lane order of
``v_mfma_f32_32x32x2_f32`` and permutes the rows of each coupling layer's last
linear layer so that every lane receives complete spline / affine parameter
sets straight from the MFMA accumulators.

A *flow description* is a dict::

    dict(D=..., hidden=H, transform="rq"|"affine"|"additive", bins=K,
         net="resnet"|"mlp", n_layers=L, activation="relu"|"tanh"|"silu",
         tail_bound=5.0, min_bin_width=1e-3, min_bin_height=1e-3,
         min_derivative=1e-3, scale_logits=True, base_var=1.0,
         transforms=[...])           # forward order

with transforms (mirroring the reference's ``CompositeTransform`` lists,
``src/nessai/flows/realnvp.py:186-213``, ``src/nessai/flows/nsf.py:110-128``):

* ``("permutation", perm)``                       out[:, i] = in[:, perm[i]]
* ``("lu", W, W_inv, bias, logabsdet)``            y = W x + b (eval-mode cache)
* ``("coupling", identity_idx, transform_idx, layers)`` with ``layers`` a list
  of ``(weight, bias)`` float32 arrays: initial, hidden..., final
* ``("elementwise", fwd, inv)`` with ``fwd = (a, b, c, d, logdet)`` meaning
  ``y = ((x - a) / b) * c + d`` per feature (batch norm / act norm, eval mode)
"""

import math

import numpy as np

MAGIC = 0x3141534E
VERSION = 10
HEADER_WORDS = 40
STAGE_ALIGN = 32   # words: stage blobs are padded to whole 128-byte lines
PIECE_WORDS = 256  # words moved by one global_load_lds of a wave (64 lanes x 16 bytes)
OP_WORDS = 4

H_MAGIC, H_VERSION, H_D, H_HT, H_TRANSFORM, H_BINS = 0, 1, 2, 3, 4, 5
H_NOPS_FWD, H_OPS_FWD, H_NOPS_INV, H_OPS_INV = 6, 7, 8, 9
H_NSTAGES_FWD, H_STAGES_FWD, H_NSTAGES_INV, H_STAGES_INV = 10, 11, 12, 13
H_MAX_STAGE, H_TOTAL, H_OUT_FWD, H_OUT_INV = 14, 15, 16, 17
H_LOG_Z, H_TAIL_BOUND, H_MIN_BW, H_MIN_BH, H_MIN_D, H_EDGE_RAW = 18, 19, 20, 21, 22, 23
H_LOGIT_DIV, H_ACT, H_NET, H_NLAYERS, H_BASE_INV_VAR, H_PREC, H_AFF_A, H_AFF_B = 24, 25, 26, 27, 28, 29, 30, 31
H_RES_FWD, H_RES_INV = 32, 33
H_CTX = 34  # context features of a conditional flow (format 10)
H_BNW = 35  # 1: batch norm within the residual blocks - every block starts with an {a, c} stage (format 10)

T_NONE, T_RQ, T_AFFINE, T_ADDITIVE = 0, 1, 2, 3
TRANSFORMS = {"rq": T_RQ, "affine": T_AFFINE, "additive": T_ADDITIVE}
ACTIVATIONS = {"relu": 0, "tanh": 1, "silu": 2, "swish": 2}
NETS = {"resnet": 0, "mlp": 1}
OP_COUPLING, OP_DENSE, OP_ELEMENTWISE = 1, 2, 3
PRECISIONS = {"f32": 0, "bf16x3": 1, "f16x3": 2}
#: precisions whose operands are pairs of 16-bit values (k in steps of 16, W stored as hi / lo parts)
SPLIT16 = ("bf16x3", "f16x3")
#: |values| from here on round to infinity in f16 (``NSA_F16_LIMIT``)
F16_LIMIT = 65520.0


def _f32_bits(value):
    return int(np.array([value], dtype=np.float32).view(np.uint32)[0])


def pad4(n):
    return (n + 3) & ~3


#: spline bin counts with an exact kernel instance (csrc/Makefile); any other count from 2 to 16 runs on the
#: generic instance NsaRQ<16, true> (csrc/nsa_device.h) with its parameters padded to 16 bins
EXACT_BINS = (4, 8, 10, 16)
GENERIC_BINS = 16
#: bias of a padded width / height logit: its exponential is 0.0, it takes no part in the softmax
PAD_LOGIT = -1.0e30


def kernel_bins(bins):
    """Bins of the kernel instance a spline flow of ``bins`` bins runs on."""
    if bins in EXACT_BINS:
        return bins
    if 2 <= bins <= GENERIC_BINS:
        return GENERIC_BINS
    raise ValueError(f"num_bins={bins} is not supported by the gfx950 kernels (2 <= num_bins <= {GENERIC_BINS})")


def transform_policy(transform, bins):
    """(tiles per chunk C, features per lane per chunk FPC, params per feature NP) of the KERNEL INSTANCE."""
    if transform == "rq":
        bins = kernel_bins(bins)
        n_params = 3 * bins - 1
        fpc = 2 if bins == 8 else 1
        return (fpc * n_params + 15) // 16, fpc, n_params
    if transform == "affine":
        return 1, 8, 2
    if transform == "additive":
        return 1, 16, 1
    raise ValueError(f"unknown transform {transform}")


def ksplit(tiles, ht):
    """Number of sub-stages along k of a register-fed GEMM with ``tiles`` output tiles.

    A stage must fit an LDS buffer; the packed weights of one GEMM are ``tiles * ht * 4`` KiB
    (32 outputs x 32 inputs x 4 bytes per tile pair, either precision).  Above 16 KiB the k steps
    are dealt to 2 or 4 consecutive stages (the accumulators live across them).  Applied to the
    128-wide conditioners (hidden tiles = 4, config 4), whose 64 KiB hidden-layer stages left room
    for 2 waves per workgroup only: with 17 KiB stages two 4-wave workgroups share a CU.  Same rule
    as ``nsa_ksplit`` in ``csrc/nsa_flow_kernel.h``."""
    if ht < 4:
        # a 64-wide conditioner's largest stage is 25 KiB: 12 waves per CU already run from one
        # double buffer, and the extra stage barriers of a split cost more than a third workgroup
        # per CU brings (1.92 against 1.76 ms per 1e6 rows of config 3)
        return 1
    nbytes = tiles * ht * 4096
    ks = 1
    while nbytes // ks > 16384 and ks < 2 * ht:
        ks *= 2
    return ks


def hidden_tiles(hidden):
    """Hidden tiles of the kernel instance a conditioner of this width runs on: 1, 2, 4 or 8 (widths up to 256;
    the reference's default ``n_neurons = 2 * n_inputs``, src/nessai/flows/utils.py:148-149, for up to 128 inputs)."""
    ht = (hidden + 31) // 32
    for instance in (1, 2, 4, 8):
        if ht <= instance:
            return instance
    return ht


def d_reg_feature(tile, reg, half):
    """Feature index held by (tile, accumulator register, lane half) in the D layout."""
    return 32 * tile + (reg & 3) + 8 * (reg >> 2) + 4 * half


def _row_of_tile_row(i):
    """Tile row i (0..31) -> (accumulator register, lane half)."""
    return (i & 3) + 4 * (i >> 3), (i >> 2) & 1


def pack_matrix(weight, out_rows, in_cols):
    """Lay ``weight`` out for the MFMA A operand.

    ``out_rows[ot][i]``  original output row for tile ``ot`` row ``i`` (or -1)
    ``in_cols[s][half]`` original input column for step ``s`` (or -1)
    Returns float32 ``[OT][n_steps/4][64][4]`` flattened.
    """
    n_ot = len(out_rows)
    n_steps = len(in_cols)
    assert n_steps % 4 == 0
    w = np.zeros((n_ot, n_steps // 4, 64, 4), dtype=np.float32)
    rows = np.asarray(out_rows, dtype=np.int64)  # [OT][32]
    cols = np.asarray(in_cols, dtype=np.int64)  # [S][2]
    for ot in range(n_ot):
        for half in range(2):
            r = rows[ot]  # [32]
            c = cols[:, half]  # [S]
            valid = (r[:, None] >= 0) & (c[None, :] >= 0)
            block = np.where(valid, weight[np.maximum(r, 0)[:, None], np.maximum(c, 0)[None, :]], 0.0)
            # block[i][s] -> w[ot][s//4][32*half + i][s%4]
            w[ot, :, 32 * half : 32 * half + 32, :] = block.reshape(32, n_steps // 4, 4).transpose(1, 0, 2)
    return w.reshape(-1)


def pack_bias(bias, out_rows, fill=None):
    """bias laid out as [OT][half][16] following the D layout; ``fill[ot][i]``: the constant of a row without
    an original (-1), default 0."""
    n_ot = len(out_rows)
    out = np.zeros((n_ot, 2, 16), dtype=np.float32)
    for ot in range(n_ot):
        for i in range(32):
            row = out_rows[ot][i]
            reg, half = _row_of_tile_row(i)
            if row >= 0:
                out[ot, half, reg] = bias[row]
            elif fill is not None:
                out[ot, half, reg] = fill[ot][i]
    return out.reshape(-1)


def bf16_bits(x):
    """float32 -> bfloat16 bit patterns, round-to-nearest-even (v_cvt_pk_bf16_f32)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def bf16_to_f32(bits):
    return (bits.astype(np.uint32) << 16).view(np.float32)


def split_bf16(x):
    """x ~= hi + lo with hi = bf16(x), lo = bf16(x - hi); returns the two bit arrays."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    hi = bf16_bits(x)
    lo = bf16_bits(x - bf16_to_f32(hi))
    return hi, lo


def split_f16(x):
    """x ~= hi + lo with hi = f16(x), lo = f16(x - hi) (round to nearest even, subnormals kept);
    returns the two bit arrays.  ``x - hi`` is exact in float32, so the pair carries x to 2^-24
    relative down to |x| = 2^-3 and to 2^-25 absolute below."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    if x.size and not np.all(np.abs(x) < F16_LIMIT):
        raise ValueError("precision 'f16x3' needs every weight below 65520 in magnitude")
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    return hi.view(np.uint16), lo.view(np.uint16)


def f16_to_f32(bits):
    return bits.view(np.float16).astype(np.float32)


def pack_matrix_bf16(weight, out_rows, in_cols, split=split_bf16):
    """A operand of v_mfma_f32_32x32x16_bf16 (or _f16: ``split=split_f16``), split in hi/lo parts.

    ``in_cols[s][half][e]`` original input column of k-step ``s`` (or -1).
    Returns uint32 words of the layout ``[OT][S][part][64 lanes][8 x 16 bit]``.
    """
    n_ot, n_steps = len(out_rows), len(in_cols)
    w = np.zeros((n_ot, n_steps, 64, 8), dtype=np.float32)
    rows = np.asarray(out_rows, dtype=np.int64)  # [OT][32]
    cols = np.asarray(in_cols, dtype=np.int64).reshape(n_steps, 2, 8)
    for ot in range(n_ot):
        for half in range(2):
            r = rows[ot]
            c = cols[:, half, :]  # [S][8]
            valid = (r[:, None, None] >= 0) & (c[None, :, :] >= 0)
            block = np.where(valid, weight[np.maximum(r, 0)[:, None, None], np.maximum(c, 0)[None, :, :]], 0.0)
            w[ot, :, 32 * half : 32 * half + 32, :] = block.transpose(1, 0, 2)  # [S][32][8]
    hi, lo = split(w)
    parts = np.stack([hi, lo], axis=2)  # [OT][S][2][64][8] uint16
    return np.ascontiguousarray(parts).reshape(-1).view(np.uint32)


def _pack_matrix16(precision, split=None):
    """Packer of the 16-bit A operands of ``precision`` (``split``: another splitter, see
    ``repack_map``)."""
    if split is None and precision == "f16x3":
        split = split_f16
    if split is not None:
        return lambda w, out_rows, in_cols: pack_matrix_bf16(w, out_rows, in_cols, split=split)
    return pack_matrix_bf16


def _reg_in_cols16(n_tiles, limit):
    """Input columns for register-fed layers in k-steps of 16: [s][half][e]."""
    cols = []
    for s in range(2 * n_tiles):
        step = []
        for half in range(2):
            step.append([
                (f if f < limit else -1)
                for f in (d_reg_feature(s >> 1, 8 * (s & 1) + e, half) for e in range(8))
            ])
        cols.append(step)
    return cols


def _tile_in_cols16(n_steps, limit):
    """Input columns for layers fed from the x tile in k-steps of 16: [s][half][e]."""
    return [[[(16 * s + 8 * h + e) if (16 * s + 8 * h + e) < limit else -1 for e in range(8)]
             for h in range(2)] for s in range(n_steps)]


def _reg_in_cols(n_tiles, limit):
    """Input columns for layers fed from registers: step s=(tile, reg)."""
    cols = []
    for s in range(16 * n_tiles):
        pair = []
        for half in range(2):
            f = d_reg_feature(s >> 4, s & 15, half)
            pair.append(f if f < limit else -1)
        cols.append(pair)
    return cols


def _plain_out_rows(n_tiles, limit):
    return [[(32 * ot + i) if (32 * ot + i) < limit else -1 for i in range(32)] for ot in range(n_tiles)]


def _stage(*sections):
    """Concatenate sections (each padded to 4 words) and pad to STAGE_ALIGN words.

    (Format 7: 32 words = one 128-byte line.  The copies still move 1 KiB pieces - the last piece of a
    stage runs over into the next blob / the image's tail slack and over the end of the stage in LDS,
    inside a buffer that is sized in whole pieces - but a blob no longer carries up to 1 KiB of padding,
    which is a quarter of the reference's default RealNVP when its whole program is kept in LDS.)"""
    parts = []
    for sec in sections:
        sec = np.ascontiguousarray(sec)
        if sec.dtype == np.float32:
            words = sec.view(np.uint32)
        elif sec.dtype == np.int32:
            words = sec.view(np.uint32)
        elif sec.dtype == np.uint32:
            words = sec
        else:
            raise TypeError(sec.dtype)
        pad = pad4(len(words)) - len(words)
        parts.append(words)
        if pad:
            parts.append(np.zeros(pad, dtype=np.uint32))
    blob = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint32)
    pad = (-len(blob)) % STAGE_ALIGN
    if pad:
        blob = np.concatenate([blob, np.zeros(pad, dtype=np.uint32)])
    return blob


class _CouplingBlobs:
    """Direction-independent packed weights of one coupling layer."""

    def __init__(self, desc, identity_idx, transform_idx, layers, gates=(), norms=()):
        """``gates``: the ``(Wc, bc)`` of the context layers of a conditional flow's residual blocks, in block
        order; layer 0 of such a flow takes ``d_id + C`` inputs, identity features first (nflows ResidualNet:
        ``initial_layer(cat(inputs, context))``).  ``norms``: the eval-mode ``(a, c)`` of the first batch norm of
        every residual block (``batch_norm_within_layers``; the second one is already folded into ``layers``)."""
        self.d_id = len(identity_idx)
        self.d_tr = len(transform_idx)
        hidden = desc["hidden"]
        ht = hidden_tiles(hidden)
        n_ctx = int(desc.get("context_features", 0) or 0)
        n_tiles_c, fpc, n_params = transform_policy(desc["transform"], desc.get("bins", 0))
        bf16 = desc.get("precision", "f32") in SPLIT16  # k in steps of 16, weights as 16-bit pairs
        self.bf16 = bf16
        self.n_in0 = self.d_id + n_ctx  # inputs of layer 0: identity features, then the context
        self.n_steps0 = (self.n_in0 + 15) // 16 if bf16 else pad4((self.n_in0 + 1) // 2)
        self.n_steps_c = ((n_ctx + 15) // 16 if bf16 else pad4((n_ctx + 1) // 2)) if n_ctx else 0
        self.n_chunks = (self.d_tr + 2 * fpc - 1) // (2 * fpc)
        self.fpc = fpc
        if bf16:
            pack_w = _pack_matrix16(desc["precision"], desc.get("_split16"))
            reg_cols = _reg_in_cols16(ht, hidden)
        else:
            pack_w = pack_matrix
            reg_cols = _reg_in_cols(ht, hidden)

        w0, b0 = layers[0]
        assert w0.shape == (hidden, self.n_in0), (w0.shape, hidden, self.n_in0)
        if bf16:
            in_cols0 = _tile_in_cols16(self.n_steps0, self.n_in0)
        else:
            in_cols0 = [[(2 * s + h) if (2 * s + h) < self.n_in0 else -1 for h in range(2)] for s in range(self.n_steps0)]
        # context gates of the residual blocks: (bias, W) laid out like layer 0, inputs = the C context columns
        self.gates = []
        for wc, bc in gates:
            assert wc.shape == (hidden, n_ctx), (wc.shape, hidden, n_ctx)
            if bf16:
                in_cols_c = _tile_in_cols16(self.n_steps_c, n_ctx)
            else:
                in_cols_c = [[(2 * s + h) if (2 * s + h) < n_ctx else -1 for h in range(2)] for s in range(self.n_steps_c)]
            self.gates.append((pack_bias(bc, _plain_out_rows(ht, hidden)), pack_w(wc, _plain_out_rows(ht, hidden), in_cols_c)))
        self.norms = [(pack_bias(a, _plain_out_rows(ht, hidden)), pack_bias(c, _plain_out_rows(ht, hidden)))
                      for a, c in norms]
        self.layer0 = (
            pack_bias(b0, _plain_out_rows(ht, hidden)),
            pack_w(w0, _plain_out_rows(ht, hidden), in_cols0),
        )

        def pack_split(w, out_rows, tiles):
            """Packed weights dealt to ksplit(tiles, ht) sub-stages along k."""
            ks = ksplit(tiles, ht)
            per = len(reg_cols) // ks
            return [pack_w(w, out_rows, reg_cols[j * per : (j + 1) * per]) for j in range(ks)]

        self.hidden_layers = []  # (bias, [W of sub-stage 0, 1, ...])
        for w, b in layers[1:-1]:
            assert w.shape == (hidden, hidden)
            self.hidden_layers.append(
                (
                    pack_bias(b, _plain_out_rows(ht, hidden)),
                    pack_split(w, _plain_out_rows(ht, hidden), ht),
                )
            )

        wf, bf = layers[-1]
        assert wf.shape[1] == hidden
        self.chunks = []  # (feature slots [2*fpc] (index into transform_idx or -1), bias, [W of sub-stage 0, 1, ...])
        # a spline flow on the generic instance: the kernel's parameter slot p of a feature (3 groups of
        # k_kern: widths, heights, derivatives) is the flow's parameter `real` or a padded slot with a constant
        k_real = int(desc.get("bins", 0)) if desc["transform"] == "rq" else 0
        k_kern = kernel_bins(k_real) if desc["transform"] == "rq" else 0
        edge_raw = float(np.float32(np.log(np.exp(1.0 - float(desc.get("min_derivative", 1e-3))) - 1.0)))

        def real_param(p):
            """(index among the flow's 3 k_real - 1 parameters or -1, constant of a padded slot)."""
            if k_kern == k_real:
                return p, 0.0
            group, j = divmod(p, k_kern)
            if group < 2:  # width / height logit j
                return (group * k_real + j, 0.0) if j < k_real else (-1, PAD_LOGIT)
            if j < k_real - 1:  # interior derivative logit of knot j + 1
                return 2 * k_real + j, 0.0
            # knot k_real is the right end of the spline: boundary derivative; beyond it: never selected
            return -1, (edge_raw if j == k_real - 1 else 0.0)

        n_real = 3 * k_real - 1 if desc["transform"] == "rq" else n_params
        for c in range(self.n_chunks):
            out_rows, fill = [], []
            for t in range(n_tiles_c):
                rows, consts = [], []
                for i in range(32):
                    reg, half = _row_of_tile_row(i)
                    v = 16 * t + reg
                    q, p = self._decode(desc["transform"], v, n_params)
                    m = c * 2 * fpc + half * fpc + q
                    row, const = -1, 0.0
                    if q < fpc and m < self.d_tr:
                        if desc["transform"] == "rq":
                            p, const = real_param(p)
                        row = self._orig_row(desc["transform"], m, p, n_real) if p >= 0 else -1
                    rows.append(row)
                    consts.append(const if row < 0 else 0.0)
                out_rows.append(rows)
                fill.append(consts)
            slots = []
            for half in range(2):
                for q in range(fpc):
                    m = c * 2 * fpc + half * fpc + q
                    slots.append(m if m < self.d_tr else -1)
            self.chunks.append((slots, pack_bias(bf, out_rows, fill), pack_split(wf, out_rows, n_tiles_c)))

    @staticmethod
    def _decode(transform, v, n_params):
        if transform == "rq":
            return v // n_params, v % n_params
        if transform == "affine":
            return v >> 1, v & 1
        return v, 0

    def _orig_row(self, transform, m, p, n_params):
        if transform == "rq":
            return m * n_params + p
        if transform == "affine":
            return p * self.d_tr + m  # shift block first, raw-scale block second
        return m


def _fold_elementwise(transforms, inverse, D, fold=True):
    """The transforms of one direction in execution order, as ``("permutation", p)`` (new logical feature i =
    old logical feature ``p[i]``), ``("dense", M, pre, post, log|det|)`` - ``y = M (x - pre) + post`` on the
    logical features - ``("elementwise", (a, b, c, d, log|det|))`` and couplings.

    In eval mode a batch norm / act norm (``((x - a) / b) c + d`` per feature) next to an LU-linear layer -
    only permutations between them, as in every block of the reference's default RealNVP
    (src/nessai/flows/realnvp.py:186-206: permutation, LU, coupling, batch norm) - is one affine map with it:
    the per-feature scale and shift are folded into the matrix and the bias on the host (in float64, rounded
    once), and the kernel runs ONE dense op where it ran a dense op and an element-wise op (the latter: eight
    IEEE divisions per lane and a stage of its own)."""
    seq = []
    for tr in (transforms[::-1] if inverse else transforms):
        if tr[0] == "permutation":
            perm = np.asarray(tr[1], dtype=np.int64)
            seq.append(("permutation", np.argsort(perm) if inverse else perm))
        elif tr[0] == "lu":
            _, W, W_inv, bias, logabsdet = tr
            bias = np.asarray(bias, dtype=np.float32)
            zero = np.zeros(D, dtype=np.float32)
            if inverse:
                seq.append(("dense", np.asarray(W_inv, dtype=np.float32), bias, zero, -np.float32(logabsdet)))
            else:
                seq.append(("dense", np.asarray(W, dtype=np.float32), zero, bias, np.float32(logabsdet)))
        elif tr[0] == "elementwise":
            seq.append(("elementwise", tr[2] if inverse else tr[1]))
        else:
            seq.append(tr)
    if not fold:
        return seq

    def neighbour(i, step):
        """Index of the dense op reached from position i through permutations only, and the composed map
        ``q`` of those permutations: the feature that is k AFTER them is feature ``q[k]`` BEFORE them."""
        q = np.arange(D)
        j = i + step
        while 0 <= j < len(seq) and seq[j][0] == "permutation":
            p = seq[j][1]
            q = q[p] if step > 0 else p[q]
            j += step
        if 0 <= j < len(seq) and seq[j][0] == "dense":
            return j, q
        return None, None

    i = 0
    while i < len(seq):
        if seq[i][0] != "elementwise":
            i += 1
            continue
        a, b, c, d, ld_e = seq[i][1]
        sc = np.asarray(c, dtype=np.float64) / np.asarray(b, dtype=np.float64)
        sh = np.asarray(d, dtype=np.float64) - np.asarray(a, dtype=np.float64) * sc
        j, q = neighbour(i, -1)
        if j is not None:
            # dense op first, then (through permutations) the element-wise map: feature k here is output row
            # q[k] of the dense op   (q composed backwards: logical k now = row q[k] then)
            _, M, pre, post, ld = seq[j]
            row_s, row_t = np.empty(D), np.empty(D)
            row_s[q], row_t[q] = sc, sh
            M2 = row_s[:, None] * M.astype(np.float64)
            post2 = row_s * post.astype(np.float64) + row_t
            seq[j] = ("dense", M2.astype(np.float32), pre, post2.astype(np.float32), np.float32(np.float32(ld) + np.float32(ld_e)))
            del seq[i]
            continue
        j, q = neighbour(i, +1)
        if j is not None:
            # element-wise map first: input k of the dense op is feature q[k] here
            _, M, pre, post, ld = seq[j]
            col_s, col_t = sc[q], sh[q]
            M64 = M.astype(np.float64)
            M2 = M64 * col_s[None, :]
            post2 = M64 @ (col_t - pre.astype(np.float64)) + post.astype(np.float64)
            seq[j] = ("dense", M2.astype(np.float32), np.zeros(D, dtype=np.float32), post2.astype(np.float32),
                      np.float32(np.float32(ld) + np.float32(ld_e)))
            del seq[i]
            continue
        i += 1
    return seq


def pack_flow(desc):
    """Return the flow image as a ``numpy.uint32`` array."""
    D = int(desc["D"])
    hidden = int(desc["hidden"])
    transform = desc["transform"]
    bins = int(desc.get("bins", 0)) if transform == "rq" else 0
    ht = hidden_tiles(hidden)
    if ht > 8:
        raise ValueError(f"hidden width {hidden} > 256 is not supported by the gfx950 kernels")
    if D < 2:
        raise ValueError("at least two features are required")
    n_tiles_c, fpc, n_params = transform_policy(transform, bins)
    tabw = pad4(2 * fpc)
    D4 = pad4(D)
    n_ctx = int(desc.get("context_features", 0) or 0)
    if n_ctx < 0 or n_ctx > 256:
        raise ValueError(f"context_features={n_ctx} is not supported by the gfx950 kernels (0 ... 256)")
    precision = desc.get("precision", "f32")
    if precision not in PRECISIONS:
        raise ValueError(f"unknown precision {precision!r}; choose from {sorted(PRECISIONS)}")
    bf16 = precision in SPLIT16

    blobs = []  # list of uint32 arrays
    blob_offsets = []

    def add_blob(blob):
        blobs.append(blob)
        blob_offsets.append(None)
        return len(blobs) - 1

    def build_direction(inverse):
        """Returns (ops [n][4] uint32, stage blob ids, out table)."""
        cur = list(range(D))  # physical column of logical feature i
        ops = []
        stages = []
        seq = _fold_elementwise(list(desc["transforms"]), inverse, D, desc.get("fold_elementwise", True))
        for tr in seq:
            kind = tr[0]
            if kind == "permutation":
                perm = np.asarray(tr[1], dtype=np.int64)
                cur = [cur[int(perm[i])] for i in range(D)]
            elif kind == "dense":
                _, mat, pre_v, post_v, ld = tr
                if D > 128:
                    raise ValueError("LU-linear transforms support at most 128 features")
                n_ot = (D + 31) // 32
                if bf16:
                    n_steps = (D + 15) // 16
                    in_cols = _tile_in_cols16(n_steps, D)
                    tab_len = 16 * n_steps
                else:
                    n_steps = pad4((D + 1) // 2)
                    in_cols = [[(2 * s + h) if (2 * s + h) < D else -1 for h in range(2)] for s in range(n_steps)]
                    tab_len = 2 * n_steps
                in_tab = np.full(tab_len, 4 * (D + 1), dtype=np.int32)  # byte offsets; unused: the zero cell
                pre = np.zeros(tab_len, dtype=np.float32)
                for k in range(D):
                    in_tab[k] = 4 * cur[k]
                mat = np.asarray(mat, dtype=np.float32)
                pre[:D] = np.asarray(pre_v, dtype=np.float32)
                post = np.asarray(post_v, dtype=np.float32)
                ld = np.float32(ld)
                out_rows = _plain_out_rows(n_ot, D)
                pack_w = _pack_matrix16(precision, desc.get("_split16")) if bf16 else pack_matrix
                # one self-contained stage per OUTPUT TILE (format 9: gather table, pre-shift, the tile's bias and
                # weights): a 128-feature layer is 4 stages of ~17 KiB of split weights instead of one of 64 KiB
                # that no double buffer holds; the tables are repeated (1 KiB) because a stage's buffer is gone
                # two stages later
                for ot in range(n_ot):
                    rows = out_rows[ot:ot + 1]
                    stages.append(add_blob(_stage(in_tab, pre, pack_bias(post, rows), pack_w(mat, rows, in_cols))))
                ops.append([OP_DENSE, n_steps, n_ot, _f32_bits(ld)])
                cur = list(range(D))
            elif kind == "coupling":
                _, identity_idx, transform_idx, packed = tr
                id_tab = np.full((16 if bf16 else 2) * packed.n_steps0, 4 * (D + 1), dtype=np.int32)
                for k, f in enumerate(identity_idx):
                    id_tab[k] = 4 * cur[int(f)]  # byte offset in the row; unused entries: the zero cell
                for j in range(n_ctx):  # the context columns lie behind the two spare cells of the row
                    id_tab[len(identity_idx) + j] = 4 * (D + 2 + j)
                stages.append(add_blob(_stage(id_tab, *packed.layer0)))
                for i, (bias, subs) in enumerate(packed.hidden_layers):
                    if packed.norms and i % 2 == 0:  # in front of a residual block: its first batch norm
                        stages.append(add_blob(_stage(*packed.norms[i // 2])))
                    stages.append(add_blob(_stage(bias, subs[0])))
                    for w in subs[1:]:
                        stages.append(add_blob(_stage(w)))
                    if packed.gates and i % 2 == 1:  # behind the second layer of a residual block: its context gate
                        c_tab = np.full((16 if bf16 else 2) * packed.n_steps_c, 4 * (D + 1), dtype=np.int32)
                        for j in range(n_ctx):
                            c_tab[j] = 4 * (D + 2 + j)
                        stages.append(add_blob(_stage(c_tab, *packed.gates[i // 2])))
                for slots, bias, subs in packed.chunks:
                    tr_tab = np.full(tabw, -1, dtype=np.int32)
                    for k, m in enumerate(slots):
                        if m >= 0:
                            tr_tab[k] = cur[int(transform_idx[m])]
                    stages.append(add_blob(_stage(tr_tab, bias, subs[0])))
                    for w in subs[1:]:
                        stages.append(add_blob(_stage(w)))
                ops.append([OP_COUPLING, packed.n_steps0, packed.n_chunks, packed.n_steps_c])
            elif kind == "elementwise":
                a, b, c, d, ld = tr[1]
                tabs = np.zeros((4, D4), dtype=np.float32)
                tabs[1, :] = 1.0
                tabs[2, :] = 1.0
                for i in range(D):
                    col = cur[i]
                    tabs[0, col] = a[i]
                    tabs[1, col] = b[i]
                    tabs[2, col] = c[i]
                    tabs[3, col] = d[i]
                stages.append(add_blob(_stage(tabs.reshape(-1))))
                ops.append([OP_ELEMENTWISE, 0, 0, _f32_bits(ld)])
            else:
                raise ValueError(f"unknown transform kind {kind}")
        return np.asarray(ops, dtype=np.uint32).reshape(-1, OP_WORDS), stages, np.asarray(cur, dtype=np.int32)

    # pack the coupling weights once (shared by both directions)
    prepared = []
    n_net_layers = None
    for tr in desc["transforms"]:
        if tr[0] == "coupling":
            _, identity_idx, transform_idx, layers = tr[:4]
            layers = [(np.asarray(w, dtype=np.float32), np.asarray(b, dtype=np.float32)) for w, b in layers]
            gates = [(np.asarray(w, dtype=np.float32), np.asarray(b, dtype=np.float32)) for w, b in (tr[4] if len(tr) > 4 else ())]
            if n_ctx and (desc["net"] != "resnet" or 2 * len(gates) != len(layers) - 2):
                raise ValueError("a conditional flow needs a residual conditioner with one context layer per block")
            norms = [(np.asarray(a, dtype=np.float32), np.asarray(c, dtype=np.float32)) for a, c in (tr[5] if len(tr) > 5 else ())]
            if bool(desc.get("batch_norm_within")) != bool(norms) or (norms and (desc["net"] != "resnet" or 2 * len(norms) != len(layers) - 2)):
                raise ValueError("batch norm within layers needs a residual conditioner with one (a, c) pair per block")
            prepared.append(("coupling", list(identity_idx), list(transform_idx),
                             _CouplingBlobs(desc, identity_idx, transform_idx, layers, gates, norms)))
            n_hidden = len(layers) - 2
            if desc["net"] == "resnet":
                assert n_hidden % 2 == 0
                nl = n_hidden // 2
            else:
                nl = n_hidden + 1
            if n_net_layers is None:
                n_net_layers = nl
            elif n_net_layers != nl:
                raise ValueError("all coupling layers must share one conditioner depth")
        else:
            prepared.append(tr)
    desc = dict(desc, transforms=prepared)

    ops_fwd, stages_fwd, out_fwd = build_direction(False)
    ops_inv, stages_inv, out_inv = build_direction(True)

    # layout
    pos = HEADER_WORDS
    off_ops_fwd = pos
    pos += ops_fwd.size
    off_ops_inv = pos
    pos += ops_inv.size
    off_st_fwd = pos
    pos += 2 * len(stages_fwd)
    off_st_inv = pos
    pos += 2 * len(stages_inv)
    off_out_fwd = pos
    pos += D4
    off_out_inv = pos
    pos += D4
    pos = (pos + 255) & ~255  # blobs start 1 KiB aligned
    for i, blob in enumerate(blobs):
        blob_offsets[i] = pos
        pos += len(blob)
    pos += PIECE_WORDS  # slack: the last 1 KiB piece of the last stage reads past its end
    total = pos
    max_stage = max(len(b) for b in blobs) if blobs else PIECE_WORDS
    max_stage = (max_stage + PIECE_WORDS - 1) // PIECE_WORDS * PIECE_WORDS  # LDS stage buffers: whole pieces

    img = np.zeros(total, dtype=np.uint32)
    img[H_MAGIC] = MAGIC
    img[H_VERSION] = VERSION
    img[H_D] = D
    img[H_HT] = ht
    img[H_TRANSFORM] = TRANSFORMS[transform]
    img[H_BINS] = bins
    img[H_NOPS_FWD] = len(ops_fwd)
    img[H_OPS_FWD] = off_ops_fwd
    img[H_NOPS_INV] = len(ops_inv)
    img[H_OPS_INV] = off_ops_inv
    img[H_NSTAGES_FWD] = len(stages_fwd)
    img[H_STAGES_FWD] = off_st_fwd
    img[H_NSTAGES_INV] = len(stages_inv)
    img[H_STAGES_INV] = off_st_inv
    img[H_MAX_STAGE] = max_stage
    img[H_TOTAL] = total
    img[H_OUT_FWD] = off_out_fwd
    img[H_OUT_INV] = off_out_inv
    base_var = float(desc.get("base_var", 1.0))
    img[H_LOG_Z] = _f32_bits(0.5 * D * math.log(2 * math.pi * base_var))
    img[H_BASE_INV_VAR] = _f32_bits(1.0 / base_var)
    min_d = float(desc.get("min_derivative", 1e-3))
    img[H_TAIL_BOUND] = _f32_bits(desc.get("tail_bound", 5.0))
    img[H_MIN_BW] = _f32_bits(desc.get("min_bin_width", 1e-3))
    img[H_MIN_BH] = _f32_bits(desc.get("min_bin_height", 1e-3))
    img[H_MIN_D] = _f32_bits(min_d)
    img[H_EDGE_RAW] = _f32_bits(math.log(math.exp(1 - min_d) - 1))
    img[H_LOGIT_DIV] = _f32_bits(math.sqrt(hidden) if desc.get("scale_logits", True) else 1.0)
    img[H_AFF_A] = _f32_bits(float(desc.get("affine_shift", 2.0)))   # scale = sigmoid(raw + a) + b
    img[H_AFF_B] = _f32_bits(float(desc.get("affine_eps", 1e-3)))
    img[H_ACT] = ACTIVATIONS[desc.get("activation", "relu")]
    img[H_NET] = NETS[desc.get("net", "resnet")]
    img[H_NLAYERS] = n_net_layers if n_net_layers is not None else 0
    img[H_PREC] = PRECISIONS[precision]
    img[H_CTX] = n_ctx
    img[H_BNW] = 1 if desc.get("batch_norm_within") else 0

    img[off_ops_fwd : off_ops_fwd + ops_fwd.size] = ops_fwd.reshape(-1)
    img[off_ops_inv : off_ops_inv + ops_inv.size] = ops_inv.reshape(-1)
    for k, bid in enumerate(stages_fwd):
        img[off_st_fwd + 2 * k] = blob_offsets[bid]
        img[off_st_fwd + 2 * k + 1] = len(blobs[bid])
    for k, bid in enumerate(stages_inv):
        img[off_st_inv + 2 * k] = blob_offsets[bid]
        img[off_st_inv + 2 * k + 1] = len(blobs[bid])
    # the stage blobs of a direction lie back to back in stage order: the weights-resident geometry of the
    # kernel copies that whole range into LDS once per workgroup (0 = not contiguous, never resident)
    for word, stage_ids in ((H_RES_FWD, stages_fwd), (H_RES_INV, stages_inv)):
        span, expect = 0, blob_offsets[stage_ids[0]] if stage_ids else 0
        for bid in stage_ids:
            if blob_offsets[bid] != expect:
                span = 0
                break
            span += len(blobs[bid])
            expect += len(blobs[bid])
        img[word] = span
    img[off_out_fwd : off_out_fwd + D] = out_fwd.view(np.uint32)
    img[off_out_inv : off_out_inv + D] = out_inv.view(np.uint32)
    for i, blob in enumerate(blobs):
        img[blob_offsets[i] : blob_offsets[i] + len(blob)] = blob
    return img


# ---------------------------------------------------------------------------
# Re-packing on the device
# ---------------------------------------------------------------------------
# The image of a flow whose transforms are couplings and permutations only is, apart from its weights
# and biases, a function of the flow's STRUCTURE: every parameter word is either one float32 parameter or
# the hi / lo halves of two parameters, everything else (header, tables, padding) is constant.  After a
# training run the image is therefore rebuilt on the device by a gather (``nsa_repack_image``: ~0.5 M
# words, one launch) instead of on the host (16 ms for the float32 image of config 3, 33 ms for the
# split-f16 one, plus the copy of the weights to the host and of the image back - per precision the
# ``auto`` policy probes, inside the first ``populate`` after every ``train``).

REPACK_F32, REPACK_PAIR_HI, REPACK_PAIR_LO = 1, 2, 3
_INDEX_A, _INDEX_B = 1, 1 + 32768 + 1  # two index offsets whose low AND high 15 bits differ for every index


def _split_index(x):
    """Stand-in for ``split_f16`` / ``split_bf16`` when the "weights" are parameter indices: the low 15 bits
    travel in the hi part (bit 15 clear), the next 15 bits in the lo part with bit 15 set as a tag."""
    idx = np.ascontiguousarray(x, dtype=np.float32).astype(np.int64)
    if idx.size and (idx.min() < 0 or idx.max() >= 1 << 30):
        raise ValueError("parameter index out of range")
    return (idx & 0x7FFF).astype(np.uint16), (((idx >> 15) & 0x7FFF) | 0x8000).astype(np.uint16)


def repack_map(index_desc, precision):
    """Where the parameters sit in the image of ``precision``.

    ``index_desc(offset)`` must return the flow description with every weight / bias array replaced by
    ``offset + (index of the parameter in the flow's flat parameter vector)`` as float32 (exact below
    2^24).  Packing it twice with different offsets marks the parameter words (they differ) and spells
    their indices out.  Returns ``(pos, e0, e1, kind)`` int32 / uint8 arrays, one entry per parameter
    word: image word ``pos`` is float32 parameter ``e0`` (``REPACK_F32``), or the hi (``REPACK_PAIR_HI``)
    / lo (``REPACK_PAIR_LO``) halves of parameters ``e0`` (low halfword) and ``e1`` (high halfword);
    an index of -1 is a padding element (zero)."""
    if precision not in PRECISIONS:
        raise ValueError(f"unknown precision {precision!r}")
    extra = {"_split16": _split_index} if precision in SPLIT16 else {}
    desc_a = index_desc(_INDEX_A)
    # the indices travel as float32 VALUES through the packer: exact below 2^24 only (a bigger flow keeps
    # the host pack - the caller treats ValueError as "no device map")
    largest = max((float(np.max(w)) for tr in desc_a["transforms"] if tr[0] == "coupling"
                   for pair in tr[3] for w in pair if np.size(w)), default=0.0)
    if largest + (_INDEX_B - _INDEX_A) >= float(1 << 24):
        raise ValueError("too many parameters for the float32-encoded index map")
    img_a = pack_flow(dict(desc_a, precision=precision, **extra))
    img_b = pack_flow(dict(index_desc(_INDEX_B), precision=precision, **extra))
    if img_a.shape != img_b.shape:
        raise RuntimeError("index images of different sizes")
    changed = img_a != img_b

    def f32_index(words):
        return np.rint(words.view(np.float32).astype(np.float64)).astype(np.int64) - _INDEX_A

    if precision not in SPLIT16:
        pos = np.nonzero(changed)[0]
        e0 = f32_index(img_a[pos])
        return (pos.astype(np.int32), e0.astype(np.int32), np.full(len(pos), -1, np.int32),
                np.full(len(pos), REPACK_F32, np.uint8))
    lo_pos = np.nonzero(changed & ((img_a & 0x80008000) == 0x80008000))[0]  # tagged: the lo-part words
    hi_pos = lo_pos - 256  # [part][64 lanes][4 words]: the hi part of the same elements
    if len(lo_pos) and (hi_pos.min() < 0 or not changed[hi_pos].all()):
        raise RuntimeError("split-16 weight blocks are not laid out as [hi | lo] pairs of 256 words")
    low, high = img_a[hi_pos].astype(np.int64), img_a[lo_pos].astype(np.int64)
    e0 = (((high & 0x7FFF) << 15) | (low & 0x7FFF)) - _INDEX_A
    e1 = ((((high >> 16) & 0x7FFF) << 15) | ((low >> 16) & 0x7FFF)) - _INDEX_A
    e0[e0 < 0] = -1
    e1[e1 < 0] = -1
    pair = np.zeros(len(img_a), dtype=bool)
    pair[hi_pos] = True
    pair[lo_pos] = True
    f_pos = np.nonzero(changed & ~pair)[0]
    pos = np.concatenate([hi_pos, lo_pos, f_pos])
    return (pos.astype(np.int32),
            np.concatenate([e0, e0, f32_index(img_a[f_pos])]).astype(np.int32),
            np.concatenate([e1, e1, np.full(len(f_pos), -1, np.int64)]).astype(np.int32),
            np.concatenate([np.full(len(hi_pos), REPACK_PAIR_HI, np.uint8), np.full(len(lo_pos), REPACK_PAIR_LO, np.uint8),
                            np.full(len(f_pos), REPACK_F32, np.uint8)]))


def apply_repack_map(image, repack, params, precision):
    """Host emulation of ``nsa_repack_image`` (tests): ``image`` with its parameter words rebuilt from the
    flat float32 vector ``params``."""
    pos, e0, e1, kind = repack
    params = np.ascontiguousarray(params, dtype=np.float32)
    a = np.where(e0 >= 0, params[np.maximum(e0, 0)], np.float32(0.0)).astype(np.float32)
    b = np.where(e1 >= 0, params[np.maximum(e1, 0)], np.float32(0.0)).astype(np.float32)
    out = np.array(image, dtype=np.uint32, copy=True)
    f = kind == REPACK_F32
    out[pos[f]] = a[f].view(np.uint32)
    if precision in SPLIT16:
        split = split_f16 if precision == "f16x3" else split_bf16
        (ah, al), (bh, bl) = split(a), split(b)
        for code, (x, y) in ((REPACK_PAIR_HI, (ah, bh)), (REPACK_PAIR_LO, (al, bl))):
            m = kind == code
            out[pos[m]] = x[m].astype(np.uint32) | (y[m].astype(np.uint32) << 16)
    return out
