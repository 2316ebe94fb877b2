"""Lane-level numpy emulation of ``nsa_flow_kernel`` (test infrastructure).

Interprets a flow image exactly as the gfx950 kernel does — stage blobs, column
tables, MFMA lane layouts (A[i][k] in lane 32k+i, B[k][j] in lane 32k+j, D[i][j]
in lane 32*((i>>2)&1)+j register (i&3)+4*(i>>3)) — so that the host packer
(``nessai_amd/pack.py``) and the kernel's data flow can be validated against the
oracle on a machine without a GPU.  The element-wise spline / affine maths is
delegated to the oracle's functions: what is under test here is the layout.
"""

import numpy as np
import torch

from nessai_amd import pack as P
from oracle.nflows_restated import splines


def _f32(bits):
    return np.array([bits], dtype=np.uint32).view(np.float32)[0]


def mfma_32x32x2(a, b, c):
    """a, b: [64] per-lane scalars; c: [64][16] accumulators -> new [64][16]."""
    A = np.stack([a[:32], a[32:]], axis=1)  # A[i][k]
    B = np.stack([b[:32], b[32:]], axis=0)  # B[k][j]
    Dm = (A.astype(np.float64) @ B.astype(np.float64)).astype(np.float32)  # [i][j]
    out = c.copy()
    for lane in range(64):
        half, j = lane >> 5, lane & 31
        for r in range(16):
            i = (r & 3) + 8 * (r >> 2) + 4 * half
            out[lane, r] += Dm[i, j]
    return out


def mfma_32x32x16_x3(a_hi, a_lo, b_hi, b_lo, c):
    """Split-bf16 product on v_mfma_f32_32x32x16_bf16: operands [64][8] float32 values that are
    exactly representable in bf16; lane 32*half + i holds k = 8*half + e."""
    def mat_a(a):  # A[i][k]
        return np.concatenate([a[:32], a[32:]], axis=1).astype(np.float64)  # [32][16]

    def mat_b(b):  # B[k][j]
        return np.concatenate([b[:32].T, b[32:].T], axis=0).astype(np.float64)  # [16][32]

    Dm = (mat_a(a_lo) @ mat_b(b_hi) + mat_a(a_hi) @ mat_b(b_lo) + mat_a(a_hi) @ mat_b(b_hi)).astype(np.float32)
    out = c.copy()
    for lane in range(64):
        half, j = lane >> 5, lane & 31
        for r in range(16):
            i = (r & 3) + 8 * (r >> 2) + 4 * half
            out[lane, r] += Dm[i, j]
    return out


def split_f32(x, f16=False):
    if f16:
        hi, lo = P.split_f16(x)
        return P.f16_to_f32(hi), P.f16_to_f32(lo)
    hi, lo = P.split_bf16(x)
    return P.bf16_to_f32(hi), P.bf16_to_f32(lo)


class Emulator:
    def __init__(self, image):
        self.img = np.asarray(image, dtype=np.uint32)
        h = self.img
        assert h[P.H_MAGIC] == P.MAGIC
        self.D = int(h[P.H_D])
        self.HT = int(h[P.H_HT])
        self.transform = int(h[P.H_TRANSFORM])
        self.bins = int(h[P.H_BINS])
        self.act = int(h[P.H_ACT])
        self.net = int(h[P.H_NET])
        self.n_layers = int(h[P.H_NLAYERS])
        self.bf16 = int(h[P.H_PREC]) in (1, 2)  # split 16-bit operands
        self.f16 = int(h[P.H_PREC]) == 2
        name = {1: "rq", 2: "affine", 3: "additive"}[self.transform]
        self.C, self.FPC, self.NP = P.transform_policy(name, self.bins)
        self.bn_within = bool(h[P.H_BNW])  # batch norm within the residual blocks: an {a, c} stage per block
        self.n_ctx = int(h[P.H_CTX])  # context features of a conditional flow: columns D + 2 ... of a tile row

    # -- helpers ---------------------------------------------------------------
    def _stage_words(self, inverse, s):
        tab = int(self.img[P.H_STAGES_INV if inverse else P.H_STAGES_FWD])
        off, n = int(self.img[tab + 2 * s]), int(self.img[tab + 2 * s + 1])
        return self.img[off : off + n]

    def _bias(self, words_f32, n_ot):
        acc = np.zeros((n_ot, 64, 16), dtype=np.float32)
        for ot in range(n_ot):
            for lane in range(64):
                half = lane >> 5
                acc[ot, lane, :] = words_f32[ot * 32 + half * 16 : ot * 32 + half * 16 + 16]
        return acc

    def _dense(self, w, n_ot, n_steps, b_of_step, acc):
        """w: float32 [n_ot][n_steps/4][64][4]; b_of_step(s) -> [64] per-lane B values."""
        w = w.reshape(n_ot, n_steps // 4, 64, 4)
        for s in range(n_steps):
            b = b_of_step(s)
            for ot in range(n_ot):
                a = w[ot, s // 4, :, s % 4]
                acc[ot] = mfma_32x32x2(a, b, acc[ot])
        return acc

    def _dense16(self, w_words, n_ot, n_steps, b_of_step, acc):
        """w_words: uint32 [n_ot][n_steps][2][64][4]; b_of_step(s) -> [64][8] float32."""
        w16 = np.ascontiguousarray(w_words).view(np.uint16).reshape(n_ot, n_steps, 2, 64, 8)
        wf = P.f16_to_f32(w16) if self.f16 else P.bf16_to_f32(w16)
        for s in range(n_steps):
            b_hi, b_lo = split_f32(b_of_step(s), self.f16)
            for ot in range(n_ot):
                acc[ot] = mfma_32x32x16_x3(wf[ot, s, 0], wf[ot, s, 1], b_hi, b_lo, acc[ot])
        return acc

    def _activate(self, x):
        if self.act == 0:
            return np.maximum(x, 0.0).astype(np.float32)
        if self.act == 1:
            return np.tanh(x).astype(np.float32)
        return (x * (1.0 / (1.0 + np.exp(-x)))).astype(np.float32)

    # -- one tile ----------------------------------------------------------------
    def run_tile(self, x_tile, inverse, ctx_tile=None):
        """x_tile: [32][D] float32 (ctx_tile: [32][C]) -> (out [32][D], logdet [32])."""
        img = self.img
        D, HT = self.D, self.HT
        xt = x_tile.astype(np.float32).copy()
        assert (ctx_tile is not None) == (self.n_ctx > 0)
        ctx = np.zeros((xt.shape[0], 0), dtype=np.float32) if ctx_tile is None else ctx_tile.astype(np.float32)
        ops_off = int(img[P.H_OPS_INV if inverse else P.H_OPS_FWD])
        n_ops = int(img[P.H_NOPS_INV if inverse else P.H_NOPS_FWD])
        out_off = int(img[P.H_OUT_INV if inverse else P.H_OUT_FWD])
        out_tab = img[out_off : out_off + D].view(np.int32)
        logdet = np.zeros(64, dtype=np.float32)
        lanes = np.arange(64)
        half_of = lanes >> 5
        row_of = lanes & 31
        stage = 0
        for o in range(n_ops):
            op = img[ops_off + 4 * o : ops_off + 4 * o + 4]
            kind = int(op[0])
            if kind == P.OP_COUPLING:
                n_steps0, n_chunks, n_steps_c = int(op[1]), int(op[2]), int(op[3])
                assert (n_steps_c > 0) == (self.n_ctx > 0)
                st = self._stage_words(inverse, stage)
                stage += 1
                tl = (16 if self.bf16 else 2) * n_steps0
                id_tab = st[:tl].view(np.int32)
                fl = st.view(np.float32)
                bias = fl[tl : tl + 32 * HT]

                def gather(tab, idx, pre=None):
                    # byte offsets into a row; an unused entry points at the row's zero cell, column D + 1
                    cols = tab[idx] // 4
                    assert np.all(tab[idx] % 4 == 0) and np.all((cols < D) | (cols == D + 1) | ((cols >= D + 2) & (cols < D + 2 + self.n_ctx)))
                    # a tile row: D features, the spare cell, the zero cell, the context
                    xz = np.concatenate([xt, np.zeros((xt.shape[0], 2), dtype=np.float32), ctx], axis=1)
                    vals = xz[row_of[:, None] if idx.ndim == 2 else row_of, cols]
                    if pre is not None:
                        vals = vals - pre[idx]
                    return vals.astype(np.float32)

                if self.bf16:
                    w = st[tl + 32 * HT : tl + 32 * HT + HT * n_steps0 * 512]
                    b0 = lambda s: gather(id_tab, 16 * s + 8 * half_of[:, None] + np.arange(8)[None, :])  # noqa: E731
                    acc = self._dense16(w, HT, n_steps0, b0, self._bias(bias, HT))
                else:
                    w = fl[tl + 32 * HT : tl + 32 * HT + HT * n_steps0 * 64]
                    acc = self._dense(w, HT, n_steps0, lambda s: gather(id_tab, 2 * s + half_of), self._bias(bias, HT))
                h = np.concatenate([acc[ot] for ot in range(HT)], axis=1)  # [64][16*HT]

                def from_regs(words, n_ot, inp, bias):
                    """A register-fed GEMM: ``words`` = the first sub-stage behind its tables and
                    bias; further sub-stages along k (pack.ksplit) are whole stages of their own."""
                    nonlocal stage
                    ks = P.ksplit(n_ot, HT)
                    acc = self._bias(bias, n_ot)
                    for j in range(ks):
                        if j:
                            words = self._stage_words(inverse, stage)
                            stage += 1
                        if self.bf16:
                            per = 2 * HT // ks
                            acc = self._dense16(words[: n_ot * per * 512], n_ot, per,
                                                lambda s, j=j, per=per: inp[:, 8 * (j * per + s) : 8 * (j * per + s) + 8],
                                                acc)
                        else:
                            per = 16 * HT // ks
                            acc = self._dense(words.view(np.float32)[: n_ot * per * 64], n_ot, per,
                                              lambda s, j=j, per=per: inp[:, j * per + s], acc)
                    return acc

                def hidden(inp):
                    nonlocal stage
                    st = self._stage_words(inverse, stage)
                    stage += 1
                    bias = st.view(np.float32)[: 32 * HT]
                    acc = from_regs(st[32 * HT :], HT, inp, bias)
                    return np.concatenate([acc[ot] for ot in range(HT)], axis=1)

                def gate():
                    """sigmoid(Wc context + bc) of a conditional residual block: a stage laid out like layer 0"""
                    nonlocal stage
                    st = self._stage_words(inverse, stage)
                    stage += 1
                    tl = (16 if self.bf16 else 2) * n_steps_c
                    c_tab = st[:tl].view(np.int32)
                    fl = st.view(np.float32)
                    bias = fl[tl : tl + 32 * HT]
                    if self.bf16:
                        w = st[tl + 32 * HT : tl + 32 * HT + HT * n_steps_c * 512]
                        acc = self._dense16(w, HT, n_steps_c,
                                            lambda s: gather(c_tab, 16 * s + 8 * half_of[:, None] + np.arange(8)[None, :]),
                                            self._bias(bias, HT))
                    else:
                        w = fl[tl + 32 * HT : tl + 32 * HT + HT * n_steps_c * 64]
                        acc = self._dense(w, HT, n_steps_c, lambda s: gather(c_tab, 2 * s + half_of), self._bias(bias, HT))
                    g = np.concatenate([acc[ot] for ot in range(HT)], axis=1)
                    return (1.0 / (1.0 + np.exp(-g.astype(np.float64)))).astype(np.float32)

                def norm(v):
                    """first batch norm of a residual block (batch_norm_within_layers): a stage {a, c} in the bias layout"""
                    nonlocal stage
                    fl = self._stage_words(inverse, stage).view(np.float32)
                    stage += 1
                    a = np.concatenate(self._bias(fl[: 32 * HT], HT), axis=1)
                    c = np.concatenate(self._bias(fl[32 * HT : 64 * HT], HT), axis=1)
                    return (v * a + c).astype(np.float32)

                if self.net == 0:
                    for _ in range(self.n_layers):
                        t = self._activate(norm(h) if self.bn_within else h)
                        t = self._activate(hidden(t))
                        t = hidden(t)
                        if self.n_ctx:
                            t = (t * gate()).astype(np.float32)
                        h = (h + t).astype(np.float32)
                else:
                    h = self._activate(h)
                    for _ in range(1, self.n_layers):
                        h = self._activate(hidden(h))

                tabw = P.pad4(2 * self.FPC)
                for _c in range(n_chunks):
                    st = self._stage_words(inverse, stage)
                    stage += 1
                    tr_tab = st[:tabw].view(np.int32)
                    fl = st.view(np.float32)
                    bias = fl[tabw : tabw + 32 * self.C]
                    acc = from_regs(st[tabw + 32 * self.C :], self.C, h, bias)
                    v = np.concatenate([acc[t] for t in range(self.C)], axis=1)  # [64][16*C]
                    for q in range(self.FPC):
                        cols = tr_tab[half_of * self.FPC + q]
                        ok = cols >= 0
                        if not ok.any():
                            continue
                        l_ok = lanes[ok]
                        x = xt[row_of[ok], cols[ok]]
                        params = v[l_ok, q * self.NP : (q + 1) * self.NP]
                        y, ld = self._transform(params, x, inverse)
                        xt[row_of[ok], cols[ok]] = y
                        logdet[l_ok] += ld
            elif kind == P.OP_DENSE:
                n_steps, n_ot = int(op[1]), int(op[2])
                tl = (16 if self.bf16 else 2) * n_steps
                acc = []
                for ot in range(n_ot):  # one self-contained stage per output tile (format 9)
                    st = self._stage_words(inverse, stage)
                    stage += 1
                    in_tab = st[:tl].view(np.int32)
                    fl = st.view(np.float32)
                    pre = fl[tl : 2 * tl]
                    bias = fl[2 * tl : 2 * tl + 32]

                    def gather_d(idx, in_tab=in_tab, pre=pre):
                        cols = in_tab[idx] // 4
                        assert np.all(in_tab[idx] % 4 == 0) and np.all((cols < D) | (cols == D + 1))
                        xz = np.concatenate([xt, np.zeros((xt.shape[0], 2), dtype=np.float32)], axis=1)
                        return (xz[row_of[:, None] if idx.ndim == 2 else row_of, cols] - pre[idx]).astype(np.float32)

                    if self.bf16:
                        w = st[2 * tl + 32 : 2 * tl + 32 + n_steps * 512]
                        acc.extend(self._dense16(w, 1, n_steps,
                                                 lambda s: gather_d(16 * s + 8 * half_of[:, None] + np.arange(8)[None, :]),
                                                 self._bias(bias, 1)))
                    else:
                        w = fl[2 * tl + 32 : 2 * tl + 32 + n_steps * 64]
                        acc.extend(self._dense(w, 1, n_steps, lambda s: gather_d(2 * s + half_of), self._bias(bias, 1)))
                new = xt.copy()
                for ot in range(n_ot):
                    for lane in range(64):
                        for r in range(16):
                            i = 32 * ot + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
                            if i < D:
                                new[lane & 31, i] = acc[ot][lane, r]
                xt = new
                logdet[:32] += _f32(op[3])
            else:
                st = self._stage_words(inverse, stage).view(np.float32)
                stage += 1
                D4 = P.pad4(D)
                a, b, c, d = (st[k * D4 : k * D4 + D] for k in range(4))
                xt = (((xt - a) / b) * c + d).astype(np.float32)
                logdet[:32] += _f32(op[3])
        ld = logdet[:32] + logdet[32:]
        return xt[:, out_tab], ld

    def _transform(self, params, x, inverse):
        img = self.img
        params = torch.from_numpy(np.ascontiguousarray(params))
        x = torch.from_numpy(np.ascontiguousarray(x))
        if self.transform == P.T_RQ:
            K = self.bins
            KK = P.kernel_bins(K)  # the generic instance: three groups of 16 slots, the flow's parameters first
            div = float(_f32(img[P.H_LOGIT_DIV]))
            if KK != K:  # what the padded slots must hold for the kernel's arithmetic (csrc/nsa_device.h, NsaRQ<16, true>)
                assert (params[:, K:KK] == P.PAD_LOGIT).all() and (params[:, KK + K : 2 * KK] == P.PAD_LOGIT).all()
                assert (params[:, 2 * KK + K - 1] == _f32(img[P.H_EDGE_RAW])).all()
            uw = params[:, None, :K] / div
            uh = params[:, None, KK : KK + K] / div
            ud = params[:, None, 2 * KK : 2 * KK + K - 1]
            y, ld = splines.unconstrained_rational_quadratic_spline(
                x[:, None], uw, uh, ud, inverse=inverse, tails="linear",
                tail_bound=float(_f32(img[P.H_TAIL_BOUND])),
                min_bin_width=float(_f32(img[P.H_MIN_BW])),
                min_bin_height=float(_f32(img[P.H_MIN_BH])),
                min_derivative=float(_f32(img[P.H_MIN_D])),
            )
            return y[:, 0].numpy(), ld[:, 0].numpy()
        if self.transform == P.T_AFFINE:
            shift = params[:, 0]
            scale = torch.sigmoid(params[:, 1] + 2) + 1e-3
            if inverse:
                return ((x - shift) / scale).numpy(), (-torch.log(scale)).numpy()
            return (x * scale + shift).numpy(), torch.log(scale).numpy()
        shift = params[:, 0]
        y = x - shift if inverse else x + shift
        return y.numpy(), np.zeros(len(x), dtype=np.float32)

    def run(self, x, inverse, context=None):
        x = np.asarray(x, dtype=np.float32)
        n = x.shape[0]
        out = np.zeros_like(x)
        ld = np.zeros(n, dtype=np.float32)
        for r0 in range(0, n, 32):
            tile = np.zeros((32, self.D), dtype=np.float32)
            m = min(32, n - r0)
            tile[:m] = x[r0 : r0 + m]
            ctile = None
            if context is not None:
                ctile = np.zeros((32, self.n_ctx), dtype=np.float32)
                ctile[:m] = np.asarray(context, dtype=np.float32)[r0 : r0 + m]
            o, l = self.run_tile(tile, inverse, ctile)
            out[r0 : r0 + m] = o[:m]
            ld[r0 : r0 + m] = l[:m]
        return out, ld
