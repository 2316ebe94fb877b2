"""The C-ABI library loads and exports every symbol ``include/nessai_amd.h`` declares."""
import ctypes
import os
import re


from nessai_amd import lib, pack

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "nessai_amd.h")).read()
    return sorted(set(re.findall(r"NSA_API[^;(]*?\b(nsa_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    names = declared_symbols()
    for required in ("nsa_flow_run", "nsa_populate_draw", "nsa_populate_weights",
                     "nsa_populate_accept", "nsa_version", "nsa_last_error"):
        assert required in names


def test_library_exports_every_declared_symbol():
    handle = ctypes.CDLL(lib.library_path())
    for name in declared_symbols():
        assert hasattr(handle, name), f"{name} missing from libnessai_amd.so"


def test_binding_covers_every_declared_symbol():
    assert sorted(lib.SIGNATURES) == declared_symbols()


def test_version_and_instance_query():
    L = lib.load()
    assert L.nsa_version() == pack.VERSION == 10
    assert L.nsa_flow_supported(2, 1, 8) == 1  # C3: H=64, RQ, 8 bins
    assert L.nsa_flow_supported(1, 2, 0) == 1  # C2: H=32, affine
    assert L.nsa_flow_supported(2, 1, 7) == 1 and L.nsa_flow_generic_bins(2, 7) == 1  # the generic spline instance
    assert L.nsa_flow_generic_bins(2, 8) == 0 and L.nsa_flow_supported(2, 1, 17) == 0
    assert L.nsa_flow_supported(9, 1, 8) == 0


def test_max_key_roundtrip_is_monotone():
    L = lib.load()
    vals = [-float("inf"), -1e300, -3.5, -0.0, 0.0, 1e-300, 2.25, 1e300, float("inf")]
    keys = [L.nsa_max_key_encode(v) for v in vals]
    assert keys == sorted(keys)
    for v, k in zip(vals, keys):
        assert L.nsa_max_key_decode(k) == v
    assert L.nsa_max_key_decode(L.nsa_max_key_init()) == -float("inf")


def test_bad_arguments_return_errors_without_a_gpu():
    L = lib.load()
    hdr = (ctypes.c_uint32 * 32)()
    rc = L.nsa_flow_run(hdr, None, 0, 0, 10, None, None, None, None, 0, 0, None)
    assert rc < 0 and b"NULL" in L.nsa_last_error()
    rc = L.nsa_populate_weights(-1, None, None, 0.0, None, None, None)
    assert rc < 0
    assert L.nsa_populate_accept_workspace(1_000_000) >= 1_000_000
