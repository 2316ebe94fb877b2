"""Host packer + kernel data flow (lane-level emulation) against the oracle, on the CPU.

``tests/emulator.py`` interprets the flow image exactly like ``nsa_flow_kernel``
(stage blobs, column tables, MFMA lane layouts); agreement with the oracle
validates ``nessai_amd/pack.py`` and the image format without a GPU.
"""
import numpy as np
import pytest
import torch

from emulator import Emulator, mfma_32x32x2
from helpers import randomise
from nessai_amd import pack as P
from nessai_amd.flows import NeuralSplineFlow, RealNVP
from nessai_amd.flows.modules import describe
from oracle import flows as O

CONFIGS = {
    "nsf_d4": (O.NeuralSplineFlow, (4, 8, 2, 2), {}),
    "nsf_d5_k10_bn": (O.NeuralSplineFlow, (5, 40, 2, 1), dict(num_bins=10, batch_norm_between_layers=True)),
    "rnvp_lu_bn": (O.RealNVP, (6, 16, 3, 2), {}),
    "rnvp_mlp_tanh": (O.RealNVP, (7, 33, 2, 2), dict(net="mlp", linear_transform="permutation",
                                                     batch_norm_between_layers=False, activation=torch.tanh)),
    "rnvp_additive_actnorm": (O.RealNVP, (4, 8, 2, 2), dict(use_volume_preserving=True, linear_transform=None,
                                                            actnorm=True, batch_norm_between_layers=False)),
    "nsf_d32": (O.NeuralSplineFlow, (32, 64, 2, 2), {}),
    # bin counts without an exact kernel instance: the generic one, parameters padded to 16 bins by the packer
    "nsf_k5": (O.NeuralSplineFlow, (4, 8, 2, 2), dict(num_bins=5)),
    "nsf_k2_d7": (O.NeuralSplineFlow, (7, 40, 2, 1), dict(num_bins=2)),
    "nsf_k13_lu": (O.NeuralSplineFlow, (5, 16, 2, 2), dict(num_bins=13, linear_transform="lu")),
    # batch norm within the residual blocks (eval mode): an {a, c} stage in front of every block, the second
    # batch norm folded into the block's first linear layer
    "nsf_bnw": (O.NeuralSplineFlow, (4, 8, 2, 2), dict(batch_norm_within_layers=True)),
    "rnvp_bnw_lu_h40": (O.RealNVP, (6, 40, 2, 3), dict(batch_norm_within_layers=True, activation=torch.tanh)),
    "rnvp_svd": (O.RealNVP, (6, 16, 2, 2), dict(linear_transform="svd")),
    # act norm -> permutation -> LU per block (the element-wise map is folded into the dense op BEHIND it in
    # the forward direction and into the one in FRONT of it in the inverse direction)
    "rnvp_actnorm_lu": (O.RealNVP, (5, 16, 3, 1), dict(actnorm=True, batch_norm_between_layers=False)),
    "rnvp_pre_bn_lu": (O.RealNVP, (4, 8, 2, 2), dict(pre_transform="batch_norm")),
    # the reference's defaults beyond 64 inputs: LU-linear over 100 features (four output tiles, one stage each)
    # and a conditioner 192 wide (the eight-hidden-tiles instances, every GEMM dealt to 16 / 8 stages along k)
    "rnvp_d100_lu": (O.RealNVP, (100, 24, 2, 1), {}),
    "nsf_h192": (O.NeuralSplineFlow, (6, 192, 2, 1), {}),
}


def test_mfma_emulation_matches_matmul():
    """A=I check with an ASYMMETRIC B catches transposed layouts."""
    rng = np.random.default_rng(0)
    A = rng.normal(size=(32, 2)).astype(np.float32)
    B = rng.normal(size=(2, 32)).astype(np.float32)
    a = np.concatenate([A[:, 0], A[:, 1]])
    b = np.concatenate([B[0], B[1]])
    d = mfma_32x32x2(a, b, np.zeros((64, 16), dtype=np.float32))
    ref = A @ B
    for lane in range(64):
        for r in range(16):
            i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
            assert abs(d[lane, r] - ref[i, lane & 31]) < 1e-6


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x3"])
@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_emulated_kernel_matches_oracle(name, precision):
    cls, args, kw = CONFIGS[name]
    torch.manual_seed(3)
    flow = cls(*args, **kw)
    randomise(flow, 3)
    flow.train()
    flow.eval()
    em = Emulator(P.pack_flow(dict(describe(flow), precision=precision)))
    n = 40 if args[0] >= 32 or args[1] > 128 else 70
    x = torch.randn(n, args[0]) * 2.5
    with torch.no_grad():
        z, ld = flow.forward(x)
        xi, ldi = flow.inverse(x)
    ze, lde = em.run(x.numpy(), False)
    xe, ldie = em.run(x.numpy(), True)
    scale = max(1.0, float(xi.abs().max()))
    # split-bf16 products carry ~2^-16 relative error; split-f16 ones are float32-class (2^-24)
    tol = 10.0 if precision == "bf16x3" else 1.0
    assert np.abs(ze - z.numpy()).max() < 5e-5 * tol
    assert np.abs(lde - ld.numpy()).max() < 1e-4 * tol
    assert np.abs(xe - xi.numpy()).max() < 5e-5 * scale * tol
    assert np.abs(ldie - ldi.numpy()).max() < 1e-4 * tol


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("case", ["nsf_ctx1", "rnvp_ctx3_lu", "nsf_ctx2_bnw"])
def test_emulated_conditional_flow_matches_oracle(case, precision):
    """context_features: the context joins layer 0 of every conditioner and gates every residual block
    (nflows ResidualNet; src/nessai/flows/nsf.py:85-95, flows/realnvp.py:136-146)."""
    cls, args, kw = {"nsf_ctx1": (O.NeuralSplineFlow, (4, 8, 2, 2), dict(context_features=1)),
                     "rnvp_ctx3_lu": (O.RealNVP, (6, 16, 2, 2), dict(context_features=3)),
                     "nsf_ctx2_bnw": (O.NeuralSplineFlow, (5, 16, 2, 2), dict(context_features=2, batch_norm_within_layers=True)),
                     }[case]
    torch.manual_seed(4)
    flow = cls(*args, **kw)
    randomise(flow, 4)
    with torch.no_grad():  # the blocks' last layers start at ~0: make the gates matter
        for name, p in flow.named_parameters():
            if "linear_layers.1" in name:
                p.add_(0.3 * torch.randn(p.shape))
    flow.train()
    flow.eval()
    em = Emulator(P.pack_flow(dict(describe(flow), precision=precision)))
    assert em.n_ctx == kw["context_features"]
    x = torch.randn(70, args[0]) * 1.5
    c = torch.randn(70, kw["context_features"])
    with torch.no_grad():
        z, ld = flow.forward(x, context=c)
        xi, ldi = flow.inverse(x, context=c)
        z0, _ = flow.forward(x, context=torch.zeros_like(c))
    assert float((z - z0).abs().max()) > 1e-2  # the context does something
    ze, lde = em.run(x.numpy(), False, context=c.numpy())
    xe, ldie = em.run(x.numpy(), True, context=c.numpy())
    scale = max(1.0, float(xi.abs().max()))
    assert np.abs(ze - z.numpy()).max() < 5e-5
    assert np.abs(lde - ld.numpy()).max() < 1e-4
    assert np.abs(xe - xi.numpy()).max() < 5e-5 * scale
    assert np.abs(ldie - ldi.numpy()).max() < 1e-4


def test_product_and_oracle_trees_pack_identically():
    """Same seed -> same initial weights -> identical device image."""
    for Pc, Oc, args, kw in [
        (NeuralSplineFlow, O.NeuralSplineFlow, (6, 16, 3, 2), {}),
        (RealNVP, O.RealNVP, (6, 16, 3, 2), {}),
        (RealNVP, O.RealNVP, (5, 16, 2, 2), dict(net="mlp", linear_transform="permutation")),
    ]:
        torch.manual_seed(11)
        a = Pc(*args, **kw)
        torch.manual_seed(11)
        b = Oc(*args, **kw)
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa) == list(sb)
        assert all(torch.equal(sa[k], sb[k]) for k in sa)
        a.eval()
        b.eval()
        assert np.array_equal(P.pack_flow(describe(a)), P.pack_flow(describe(b)))


def test_batch_norm_next_to_an_lu_layer_is_folded_into_it():
    """The reference's default RealNVP (permutation, LU, coupling, batch norm per block): in the image every
    batch norm but the one with no LU-linear neighbour has become part of a dense op, in both directions;
    ``fold_elementwise=False`` keeps the one-op-per-transform program (same results: the emulator test above
    runs the folded form against the oracle)."""
    torch.manual_seed(2)
    flow = O.RealNVP(16, 32, 4, 2)
    randomise(flow, 2)
    flow.eval()

    def kinds(img, inverse):
        n = int(img[P.H_NOPS_INV if inverse else P.H_NOPS_FWD])
        at = int(img[P.H_OPS_INV if inverse else P.H_OPS_FWD])
        return [int(img[at + P.OP_WORDS * i]) for i in range(n)]

    img = P.pack_flow(describe(flow))
    plain = P.pack_flow(dict(describe(flow), fold_elementwise=False))
    for inverse in (False, True):
        assert kinds(plain, inverse).count(P.OP_ELEMENTWISE) == 4 and len(kinds(plain, inverse)) == 12
        assert kinds(img, inverse).count(P.OP_ELEMENTWISE) == 1 and len(kinds(img, inverse)) == 9
    x = torch.randn(50, 16) * 2
    for inverse in (False, True):
        a, la = Emulator(img).run(x.numpy(), inverse)
        b, lb = Emulator(plain).run(x.numpy(), inverse)
        assert np.abs(a - b).max() < 2e-5 * max(1.0, np.abs(b).max()) and np.abs(la - lb).max() < 2e-5


def test_image_header_and_alignment():
    torch.manual_seed(0)
    flow = O.NeuralSplineFlow(32, 64, 6, 2).eval()
    img = P.pack_flow(describe(flow))
    assert img[P.H_MAGIC] == P.MAGIC and img[P.H_VERSION] == P.VERSION
    assert img[P.H_PREC] == 0
    img16 = P.pack_flow(dict(describe(flow), precision="bf16x3"))
    assert img16[P.H_PREC] == 1 and img16[P.H_MAX_STAGE] == img[P.H_MAX_STAGE]
    imgf = P.pack_flow(dict(describe(flow), precision="f16x3"))
    assert imgf[P.H_PREC] == 2 and len(imgf) == len(img16)
    assert img[P.H_D] == 32 and img[P.H_HT] == 2 and img[P.H_BINS] == 8
    assert img[P.H_TOTAL] == len(img)
    assert img[P.H_MAX_STAGE] % 256 == 0
    # 6 couplings x (1 + 4 + 4) stages per direction
    assert img[P.H_NSTAGES_FWD] == 54 and img[P.H_NSTAGES_INV] == 54
    # blobs are padded to whole 128-byte lines and lie back to back; the copies move 1 KiB pieces, so the
    # last piece of every stage must still lie inside the image (the tail slack) and inside a stage buffer
    tab = int(img[P.H_STAGES_FWD])
    expect, span = int(img[tab]), 0
    for s in range(54):
        off, n = int(img[tab + 2 * s]), int(img[tab + 2 * s + 1])
        assert off % P.STAGE_ALIGN == 0 and n % P.STAGE_ALIGN == 0 and off == expect
        assert off + (n + 255) // 256 * 256 <= len(img) and (n + 255) // 256 * 256 <= img[P.H_MAX_STAGE]
        expect, span = off + n, span + n
    # ... which is what lets the weights-resident geometry copy a direction's whole program in one go
    assert img[P.H_RES_FWD] == span and img[P.H_RES_INV] == span


def test_wide_conditioner_stages_are_split_along_k():
    """128 hidden units (config 4): the 64 KiB hidden-layer GEMMs and the 48 KiB chunk GEMMs are
    dealt to four stages along k each (pack.ksplit), so that no stage exceeds 17 KiB; the emulator,
    which walks the stages exactly as the kernel does, still reproduces the oracle."""
    assert P.ksplit(3, 2) == 1 and P.ksplit(4, 4) == 4 and P.ksplit(3, 4) == 4 and P.ksplit(1, 4) == 1
    torch.manual_seed(1)
    flow = O.NeuralSplineFlow(12, 128, 2, 1).eval()
    for precision in ("f32", "bf16x3", "f16x3"):
        img = P.pack_flow(dict(describe(flow), precision=precision))
        assert img[P.H_HT] == 4 and img[P.H_MAX_STAGE] * 4 <= 17 * 1024
        # 2 couplings x (1 layer-0 + 2 x 4 hidden + 2 chunks x 4) stages
        assert img[P.H_NSTAGES_FWD] == 2 * (1 + 8 + 8)
        x = torch.randn(32, 12)
        with torch.no_grad():
            z_ref, ld_ref = flow.forward(x)
        z, ld = Emulator(img).run(x.numpy(), inverse=False)
        assert np.abs(z - z_ref.numpy()).max() < 2e-4 and np.abs(ld - ld_ref.numpy()).max() < 5e-4


def test_unsupported_configurations_fail_loudly():
    with pytest.raises(ValueError, match="No gfx950 kernel instance"):
        NeuralSplineFlow(4, 8, 2, 2, num_bins=17)
    NeuralSplineFlow(4, 8, 2, 2, num_bins=7)  # the generic spline instance
    with pytest.raises(ValueError):
        NeuralSplineFlow(4, 300, 2, 2)
    assert NeuralSplineFlow(4, 8, 2, 2, context_features=2).context_features == 2  # conditional flows: built
    with pytest.raises(NotImplementedError, match="MLP with conditional inputs"):
        RealNVP(4, 8, 2, 2, net="mlp", context_features=2)
    RealNVP(6, 8, 2, 2, linear_transform="svd")  # SVD-linear: one dense op in eval mode (trains since round 6)
    bnw = NeuralSplineFlow(4, 8, 2, 2, batch_norm_within_layers=True)  # batch norm within the blocks: built
    assert P.pack_flow(describe(bnw.eval()))[P.H_BNW] == 1
    # in training mode it runs on batch statistics through the training kernels (round 6) - on a device
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        bnw.train().forward(torch.zeros(2, 4))
    with pytest.raises(ValueError, match="at least 2"):
        RealNVP(1, 8, 2, 2)
    with pytest.raises(RuntimeError, match="actnorm and batchnorm"):
        RealNVP(4, 8, 2, 2, actnorm=True)
    with pytest.raises(ValueError, match="Unknown nn type"):
        RealNVP(4, 8, 2, 2, net="cnn")
    with pytest.raises(ValueError, match="Mask does not match number of features"):
        RealNVP(4, 8, 2, 2, mask=[1, -1, 1])


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x3"])
def test_repack_map_rebuilds_the_image_from_the_flat_parameter_vector(precision):
    """``pack.repack_map``: for a flow of couplings and permutations every image word is constant, one float32
    parameter, or the hi / lo halves of two parameters.  The map (what ``nsa_repack_image`` executes on the
    device) applied to NEW weights reproduces ``pack_flow`` of those weights bit for bit; LU-linear / batch-norm
    flows have derived words and are refused."""
    for cls, args, kw in [
        (O.NeuralSplineFlow, (32, 64, 6, 2), {}),
        (O.NeuralSplineFlow, (12, 128, 2, 1), dict(num_bins=4)),  # k-split stages of the 128-wide conditioner
        (O.NeuralSplineFlow, (5, 20, 2, 3), dict(num_bins=10)),
        (O.RealNVP, (6, 16, 3, 2), dict(linear_transform="permutation", batch_norm_between_layers=False)),
        (O.RealNVP, (5, 12, 2, 3), dict(net="mlp", linear_transform="permutation", batch_norm_between_layers=False,
                                        use_volume_preserving=True)),
    ]:
        torch.manual_seed(1)
        flow = cls(*args, **kw).eval()
        repack = P.repack_map(lambda off: describe(flow, index_offset=off), precision)
        template = P.pack_flow(dict(describe(flow), precision=precision))
        with torch.no_grad():
            for p in flow.parameters():
                p.mul_(1.7).add_(0.013)
        flat = torch.cat([p.detach().reshape(-1) for p in flow.parameters()]).numpy()
        expected = P.pack_flow(dict(describe(flow), precision=precision))
        assert not np.array_equal(template, expected)
        assert np.array_equal(P.apply_repack_map(template, repack, flat, precision), expected)
    with pytest.raises(ValueError, match="not copies of parameters"):
        describe(O.RealNVP(6, 16, 2, 2).eval(), index_offset=1)
