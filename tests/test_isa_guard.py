"""Guards on the compiled gfx950 code objects (CPU side: reads the built objects, launches nothing).

The performance of rounds 2-4 rests on compiler behaviour nothing else pins: ``-mllvm -disable-machine-licm`` and
``-fno-slp-vectorize`` (``csrc/Makefile``), the opaque ``-1`` that makes the f16 split select ``v_fma_mix_f32``
(``csrc/nsa_device.h``), the opaque lane index that keeps per-lane invariants out of the persistent loop.  A ROCm bump
that undoes one of them brings spills back silently - the results stay right and the kernels get slower.  This test
reads the register / scratch metadata of the instances that carry the measured numbers (``scripts/kernel_regs.py``)
and the disassembly of the headline kernel and fails when a budget documented in DESIGN.md is exceeded.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "nessai_amd", "csrc", "build")
LLVM = "/opt/rocm/lib/llvm/bin"
sys.path.insert(0, os.path.join(ROOT, "scripts"))

pytestmark = pytest.mark.skipif(not os.path.isdir(BUILD) or not os.path.exists(f"{LLVM}/llvm-objdump"),
                                reason="needs the in-tree build objects (make -C nessai_amd/csrc) and llvm-objdump")


def _kernels(obj):
    import kernel_regs

    rows = kernel_regs.kernels(os.path.join(BUILD, obj))
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True,
                           text=True).stdout.split("\n")
    out = {}
    for r, n in zip(rows, names):
        out[n.replace("void ", "").replace("(NsaLaunch)", "").strip()] = r
    return out


def _one(table, name):
    assert name in table, f"{name} not built; have e.g. {sorted(table)[:4]}"
    return table[name]


# (object, kernel, VGPR budget): the waves-per-SIMD step the instance is launched at (MI355X_MICROARCH.md, register
# files: <= 128 -> 4 waves, <= 168 -> 3, <= 256 -> 2) - and never a byte of scratch
FLOW_INSTANCES = [
    # headline (config 3): staged, two 32-row tiles per wave, split-f16, both directions
    ("inst_h2_t1_8.o", "nsa_flow_kernel<2, NsaRQ<8, false>, true, 2, 2, 2, false, false>", 256),
    ("inst_h2_t1_8.o", "nsa_flow_kernel<2, NsaRQ<8, false>, false, 2, 2, 2, false, false>", 256),
    # the same flow at 3 waves per SIMD and in exact f32
    ("inst_h2_t1_8.o", "nsa_flow_kernel<2, NsaRQ<8, false>, true, 2, 3, 1, false, false>", 168),
    ("inst_h2_t1_8.o", "nsa_flow_kernel<2, NsaRQ<8, false>, true, 0, 2, 1, false, false>", 256),
    # config 2 (the reference's default RealNVP): weights-resident, 4 waves per SIMD
    ("inst_h1_t2_0.o", "nsa_flow_kernel<1, NsaAffine, true, 2, 4, 1, true, false>", 128),
    ("inst_h1_t2_0.o", "nsa_flow_kernel<1, NsaAffine, false, 2, 4, 1, true, false>", 128),
    ("inst_h1_t2_0.o", "nsa_flow_kernel<1, NsaAffine, true, 2, 2, 2, true, false>", 256),
    # config 5 (15-D NSF, 30 -> 32 hidden units)
    ("inst_h1_t1_8.o", "nsa_flow_kernel<1, NsaRQ<8, false>, true, 2, 2, 2, false, false>", 256),
]


@pytest.mark.parametrize("obj,kernel,budget", FLOW_INSTANCES)
def test_flow_instances_keep_their_register_budget_and_use_no_scratch(obj, kernel, budget):
    r = _one(_kernels(obj), kernel)
    assert r["spill"] == 0 and r["scratch"] == 0, r
    assert r["vgpr"] + r["agpr"] <= budget, r


def test_small_batch_and_wide_instances_stay_inside_their_budgets():
    """The extended instances the small-batch geometry launches (loader waves, stage ring, teams) and the 256-wide
    ones: no spilled register in the 64-wide class (config 3's live points), at most a handful outside the coupling
    loop in the 128-wide class (config 4), the 256-wide class built for ONE wave per SIMD - accumulation registers in
    use, a few hundred bytes of scratch instead of the kilobyte of the two-waves build."""
    t2 = _kernels("inst_h2_t1_8.o")
    for inv in ("true", "false"):
        for prec in (0, 1, 2):
            r = _one(t2, f"nsa_flow_kernel<2, NsaRQ<8, false>, {inv}, {prec}, 2, 1, false, true>")
            assert r["spill"] == 0 and r["scratch"] == 0 and r["vgpr"] <= 256, r
    t4 = _kernels("inst_h4_t1_8.o")
    for inv in ("true", "false"):
        r = _one(t4, f"nsa_flow_kernel<4, NsaRQ<8, false>, {inv}, 2, 2, 1, false, false>")
        assert r["spill"] == 0 and r["scratch"] == 0, r
        for prec in (0, 1, 2):
            r = _one(t4, f"nsa_flow_kernel<4, NsaRQ<8, false>, {inv}, {prec}, 2, 1, false, true>")
            assert r["spill"] <= 16 and r["scratch"] <= 64, r
    t8 = _kernels("inst_h8_t1_8.o")
    for name, r in t8.items():
        assert r["agpr"] > 0 and r["vgpr"] <= 512 and r["scratch"] <= 400, (name, r)  # (vgpr: the unified total)


def test_second_generation_training_kernels_do_not_spill():
    table = _kernels("nsa_train.o")
    for name in ("trf2_forward<8, 2, 6>(Trf2Multi)", "trf2_forward<8, 2, 1>(Trf2Multi)", "trf2_forward<0, 1, 4>(Trf2Multi)"):
        r = _one(table, name)
        assert r["spill"] == 0 and r["scratch"] == 0 and r["vgpr"] <= 256, (name, r)
    back = [n for n in table if n.startswith("trf2_backward<8, 2,")]
    assert back, "config 3's backward instance (K = 8, two hidden tiles) is not built"
    for n in back:
        assert table[n]["spill"] == 0 and table[n]["scratch"] == 0, (n, table[n])


def test_third_generation_training_kernels():
    """The 16-row kernels of 128-wide conditioners (config 4; csrc/nsa_train_fused3.h): no scratch, registers that
    leave two waves per SIMD; in the disassembly of the 8-bin backward instance - 16x16x4 f32 matrix instructions,
    the operand streams as buffer loads (no 64-bit address arithmetic per request), and no wait for EVERY
    outstanding memory operation in front of an MFMA of the residual blocks (a prefetch under a branch, or sunk to
    its use, shows up there as vmcnt(0))."""
    table = _kernels("nsa_train.o")
    for k in (0, 4, 8, 10, 16):
        for name in (f"trf4_forward<{k}, 128>(TrfArgs)", f"trf4_backward<{k}, true, 128, true>(TrfArgs)",
                     f"trf4_forward<{k}, 0>(TrfArgs)", f"trf4_backward<{k}, true, 0, true>(TrfArgs)",
                     f"trf4_backward<{k}, true, 0, false>(TrfArgs)"):
            r = _one(table, name)
            assert r["spill"] == 0 and r["scratch"] == 0 and r["vgpr"] <= 256, (name, r)
    asm = _disassemble("nsa_train.o", "trf4_backwardILi8ELb1ELi128ELb1EEv7TrfArgs")
    lines = [ln.split("//")[0].strip() for ln in asm.split("\n")]
    ops = [ln.split()[0] for ln in lines if ln and not ln.endswith(":")]
    assert sum(1 for o in ops if o == "v_mfma_f32_16x16x4_f32") >= 300 and not any(o.startswith("v_mfma_f32_32x32") for o in ops)
    assert sum(1 for o in ops if o == "buffer_load_dword") >= 64 and sum(1 for o in ops if o == "buffer_load_dwordx4") >= 16
    assert not any(o.startswith("scratch_") for o in ops)
    barriers = [i for i, ln in enumerate(lines) if ln.startswith("s_barrier")]
    assert len(barriers) == 6  # tiles | transform items | final layer | (first, second layer of a block) | initial layer
    # the loop over the residual blocks: an MFMA never follows a full drain of the vector-memory counter directly
    # (there is always a younger prefetch in flight)
    for i in range(barriers[2], barriers[4]):
        if lines[i].startswith("s_waitcnt") and "vmcnt(0)" in lines[i]:
            nxt = next(ln for ln in lines[i + 1:] if ln and not ln.startswith("s_"))
            assert not nxt.startswith("v_mfma"), (i, lines[i], nxt)


def _disassemble(obj, mangled_fragment):
    tmp = tempfile.mkdtemp()
    try:
        local = os.path.join(tmp, obj)
        shutil.copy(os.path.join(BUILD, obj), local)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], check=True, capture_output=True)
        co = [f for f in os.listdir(tmp) if "amdgcn" in f]
        text = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", os.path.join(tmp, co[0])], check=True,
                              capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp)
    blocks = re.split(r"\n(?=[0-9a-f]+ <)", text)
    body = [b for b in blocks if mangled_fragment in b.split("\n", 1)[0]]
    assert len(body) == 1, f"{len(body)} symbols match {mangled_fragment}"
    return body[0]


def test_headline_kernel_instruction_stream():
    """What the measured 1.55-1.6 ms per 1e6 rows depend on, read off the disassembly of
    nsa_flow_kernel<2, NsaRQ<8, false>, inverse, f16x3, 2 waves per SIMD, 2 tiles per wave, staged>."""
    asm = _disassemble("inst_h2_t1_8.o", "nsa_flow_kernelILi2E5NsaRQILi8ELb0EELb1ELi2ELi2ELi2ELb0ELb0EE")
    ops = re.findall(r"^\s+([a-z_0-9]+)", asm, flags=re.M)
    count = lambda pat: sum(1 for o in ops if re.fullmatch(pat, o))  # noqa: E731
    # the f16 split through v_fma_mix_f32 (2 vector instructions per value; cvt + sub would be 3)
    assert count(r"v_fma_mix_f32") >= 200
    # f16 matrix cores, three products per operand pair
    assert count(r"v_mfma_f32_32x32x16_f16") >= 400 and count(r"v_mfma_f32_32x32x16_f16") % 3 == 0
    # no scratch traffic, no packed-f32 vector code (SLP vectorisation is an anti-lever beside MFMAs)
    assert count(r"scratch_.*") == 0
    assert count(r"v_pk_(fma|mul|add)_f32.*") == 0
    # transcendentals on the bare units; the only IEEE float32 divisions are the kernel's set-up constants
    # (1 / sqrt(H), the elementwise op: a few sequences outside the tile loop), none per spline evaluation
    assert count(r"v_div_fixup_f32.*") <= 4
    assert count(r"v_exp_f32.*") > 0 and count(r"v_log_f32.*") > 0 and count(r"v_rcp_f32.*") > 0
    # the weight stream goes global -> LDS without passing registers
    assert count(r"global_load_lds_dwordx4") > 0


RING_INSTANCES = [
    # the small-batch geometry's instances (stage ring + loader waves, NsaRingPipe<true>): config 3's flow over the live
    # points (inverse, split-f16) and config 4's log_prob_all / draw (128-wide, forward and inverse)
    ("inst_h2_t1_8.o", "nsa_flow_kernelILi2E5NsaRQILi8ELb0EELb1ELi2ELi2ELi1ELb0ELb1EE"),
    ("inst_h4_t1_8.o", "nsa_flow_kernelILi4E5NsaRQILi8ELb0EELb0ELi2ELi2ELi1ELb0ELb1EE"),
    ("inst_h4_t1_8.o", "nsa_flow_kernelILi4E5NsaRQILi8ELb0EELb1ELi2ELi2ELi1ELb0ELb1EE"),
]
#: the immediates of nsa_wait_vmcnt's switch (csrc/nsa_flow_kernel.h): exact to 23, then in steps of eight, then 63
WAIT_TABLE = set(range(1, 24)) | {24, 32, 40, 48, 56, 63}


@pytest.mark.parametrize("obj,fragment", RING_INSTANCES)
def test_ring_pipe_instances_wait_where_the_source_says(obj, fragment):
    """The stage ring's correctness rests on waits the COMPILER does not see: the copies are LDS-DMAs inside inline
    assembly (``nsa_glds16_raw``), the waits hand-counted ``s_waitcnt vmcnt(N)`` (``nsa_wait_vmcnt``) in front of bare
    ``s_barrier``s.  A compiler that started to add waits of its own, to move or drop a barrier's LDS wait, or to
    break the M0 save / restore around a copy would leave the results right on most runs and the ring drained or racy.
    Read off the disassembly:

    * every counted wait comes from the switch table of ``nsa_wait_vmcnt`` - the table's immediates and no others, each
      as often as the table has call sites (every call site carries the whole table);
    * a ``vmcnt(0)`` behind a copy and directly in front of an LDS READ - the wait hipcc puts in front of a read it
      cannot tell from a pending LDS-DMA's destination, which drains the ring (``scripts/isa_waits.py``) - occurs a
      handful of times (the kernel's prologue and epilogue), never once per stage;
    * every ``s_barrier`` sits behind a wait that drains the LDS counter, with no LDS operation in between;
    * every copy has its M0 (the LDS base of an LDS-DMA) set right in front of it and restored right behind it."""
    asm = _disassemble(obj, fragment)
    lines = [l.split("//")[0].strip() for l in asm.split("\n")[1:]]
    lines = [l for l in lines if l]
    waits = {}
    zero_after_copy = 0
    last_vmem = None
    for i, l in enumerate(lines):
        op = l.split()[0]
        if re.match(r"(global_|buffer_|flat_|scratch_)", op):
            last_vmem = op
        m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", l)
        if m and last_vmem == "global_load_lds_dwordx4":
            # (waits behind the compiler's own loads - the prologue's and epilogue's global reads - are the compiler's
            # business; the ones whose last vector-memory instruction is a copy are the ring's)
            n = int(m.group(1))
            waits[n] = waits.get(n, 0) + 1
            if n == 0:
                nxt = next((lines[j].split()[0] for j in range(i + 1, min(i + 40, len(lines)))
                            if re.match(r"(ds_|global_|buffer_|flat_|s_barrier|v_mfma|scratch_)", lines[j])), "")
                zero_after_copy += nxt.startswith("ds_read")  # ... and the next thing is an LDS read, not a barrier
    counted = {n: c for n, c in waits.items() if n != 0}
    assert set(counted) == WAIT_TABLE, sorted(set(counted) ^ WAIT_TABLE)
    # every call site carries the whole table: each immediate appears once per site (+ at most a few waits of the
    # compiler's own behind ITS loads that happen to follow a copy: a hidden copy only ever makes those over-wait -
    # the counter is in order and the copy is one more operation outstanding than the compiler counted)
    n_sites = min(counted.values())
    assert n_sites >= 4
    assert max(counted.values()) <= n_sites + 3, counted
    assert sum(1 for c in counted.values() if c == n_sites) >= len(WAIT_TABLE) - 3, counted
    assert waits.get(0, 0) >= n_sites  # (the table's own case 0)
    # 3-4 today (the kernel's prologue and epilogue); the regression round 5 removed had one at 13 of 17 stage sites
    assert zero_after_copy <= 6, (zero_after_copy, n_sites)
    copies = [i for i, l in enumerate(lines) if l.startswith("global_load_lds_dwordx4")]
    assert len(copies) >= 4
    for i in copies:
        before, after = lines[max(i - 3, 0):i], lines[i + 1:i + 2]
        assert any(re.match(r"s_mov_b32 m0, vcc_lo", b) for b in before), lines[i - 4:i + 2]
        assert after and re.match(r"s_mov_b32 m0, vcc_hi", after[0]), lines[i - 4:i + 2]
    barriers = [i for i, l in enumerate(lines) if l.startswith("s_barrier")]
    assert len(barriers) >= n_sites
    for i in barriers:
        # walking back from the barrier: the LDS counter is drained before any LDS operation is met
        for j in range(i - 1, max(i - 16, 0), -1):
            if lines[j].startswith("s_waitcnt") and "lgkmcnt(0)" in lines[j]:
                break
            assert not lines[j].startswith("ds_"), lines[j:i + 1]
        else:
            raise AssertionError(f"no LDS wait in front of the barrier: {lines[i - 16:i + 1]}")
