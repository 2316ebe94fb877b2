"""ctypes binding of ``libnessai_amd.so`` (C ABI: ``include/nessai_amd.h``).

The library is built in-tree by ``nessai_amd/csrc/Makefile`` (see
``__graft_entry__.build``).  There is no fallback: if the shared object is
missing or a call fails, a ``RuntimeError`` is raised.
"""

import ctypes
import os

_LIB = None
# NESSAI_AMD_LIBRARY points at another build of the same library (A/B timing of kernel variants)
_LIB_PATH = os.environ.get("NESSAI_AMD_LIBRARY") or os.path.join(
    os.path.dirname(os.path.abspath(__file__)), "libnessai_amd.so"
)

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_ll = ctypes.c_longlong
c_ull = ctypes.c_ulonglong
c_double = ctypes.c_double
c_size_t = ctypes.c_size_t
c_float = ctypes.c_float



class PostRules(ctypes.Structure):
    """``nsa_post_rules`` (include/nessai_amd.h): the truncation rules that act after the flow."""

    _fields_ = [("min_log_q", c_double), ("ll_kind", c_int), ("ll_mu_dev", c_void_p), ("ll_sigma_dev", c_void_p),
                ("ll_const", c_double), ("log_l_threshold", c_double)]


SIGNATURES = {
    "nsa_version": (c_int, []),
    "nsa_last_error": (ctypes.c_char_p, []),
    "nsa_flow_supported": (c_int, [c_int, c_int, c_int]),
    "nsa_last_geometry": (None, [c_void_p, c_void_p, c_void_p]),
    "nsa_last_row_tiles": (ctypes.c_int, []),
    "nsa_last_resident": (ctypes.c_int, []),
    "nsa_last_team": (ctypes.c_int, []),
    "nsa_last_ring": (ctypes.c_int, []),
    "nsa_flow_generic_bins": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "nsa_flow_run": (
        c_int,
        [c_void_p, c_void_p, c_int, c_int, c_ll, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    ),
    "nsa_flow_run_ctx": (
        c_int,
        [c_void_p, c_void_p, c_int, c_int, c_ll, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
         c_void_p],
    ),
    "nsa_flow_run_many": (
        c_int,
        [c_void_p, c_void_p, c_int, c_int, c_int, c_ll, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
         c_void_p],
    ),
    "nsa_populate_draw": (
        c_int,
        [c_void_p, c_void_p, c_ll, c_void_p, c_ull, c_ull, c_double, c_double, c_void_p, c_void_p, c_void_p,
         c_void_p, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
         c_void_p, c_int, c_int, c_void_p],
    ),
    "nsa_populate_passes": (
        c_int,
        [c_void_p, c_void_p, c_ll, c_int, c_ull, c_ull, c_double, c_double, c_void_p, c_void_p, c_void_p, c_void_p,
         c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_void_p, c_void_p,
         c_void_p, c_ll, c_void_p, c_void_p, c_void_p, ctypes.c_size_t, c_void_p, c_int, c_int, c_void_p,
         ctypes.POINTER(PostRules)],
    ),
    "nsa_populate_draw_weights": (
        c_int,
        [c_void_p, c_void_p, c_ll, c_ull, c_ull, c_double, c_double, c_void_p, c_void_p, c_void_p, c_void_p,
         c_double, c_void_p, c_void_p, c_void_p, c_double, c_double, c_void_p, c_void_p, c_void_p, c_void_p,
         c_void_p, c_int, c_int, c_void_p],
    ),
    "nsa_populate_passes_lean": (
        c_int,
        [c_void_p, c_void_p, c_ll, c_int, c_ull, c_ull, c_double, c_double, c_void_p, c_void_p, c_void_p, c_void_p,
         c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_ll,
         c_void_p, c_void_p, c_void_p, ctypes.c_size_t, c_void_p, c_int, c_int, c_void_p, ctypes.POINTER(PostRules)],
    ),
    "nsa_populate_replay": (
        c_int,
        [c_void_p, c_void_p, c_ll, c_void_p, c_ull, c_double, c_double, c_void_p, c_void_p, c_void_p, c_void_p,
         c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
         c_void_p],
    ),
    "nsa_populate_train_heads_lean": (
        c_int,
        [c_void_p, c_void_p, c_ll, c_int, c_ull, c_ull, c_ull, c_double, c_double, c_void_p, c_void_p, c_void_p,
         c_void_p, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_void_p,
         c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, ctypes.POINTER(PostRules)],
    ),
    "nsa_populate_train_tails_lean": (
        c_int,
        [c_ll, c_int, c_int, c_void_p, c_void_p, c_ull, c_ull, c_ull, c_ll, c_void_p, c_void_p, c_void_p,
         c_void_p, c_void_p, ctypes.c_size_t, c_void_p],
    ),
    "nsa_populate_passes_small": (
        c_int,
        [c_void_p, c_void_p, c_ll, c_int, c_ull, c_ull, c_double, c_double, c_void_p, c_void_p, c_void_p, c_void_p,
         c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_void_p, c_ll, c_void_p,
         c_void_p, c_ll, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, ctypes.POINTER(PostRules)],
    ),
    "nsa_populate_pass_head": (
        c_int,
        [c_void_p, c_void_p, c_ll, c_ull, c_ull, c_double, c_double, c_void_p, c_void_p, c_void_p, c_void_p,
         c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_void_p, c_void_p,
         c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    ),
    "nsa_populate_pass_tail": (
        c_int,
        [c_ll, c_int, c_void_p, c_void_p, c_ull, c_ull, c_void_p, c_ll, c_void_p, c_void_p, c_void_p, c_size_t,
         c_void_p],
    ),
    "nsa_populate_pass_commit": (c_int, [c_void_p, c_ll, c_void_p, c_void_p, c_void_p]),
    "nsa_populate_train_heads": (
        c_int,
        [c_void_p, c_void_p, c_ll, c_int, c_ull, c_ull, c_ull, c_double, c_double, c_void_p, c_void_p, c_void_p,
         c_void_p, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_void_p,
         c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, ctypes.POINTER(PostRules)],
    ),
    "nsa_populate_train_tails": (
        c_int,
        [c_ll, c_int, c_int, c_void_p, c_void_p, c_ull, c_ull, c_ull, c_void_p, c_ll, c_void_p, c_void_p, c_void_p,
         c_void_p, c_void_p, c_size_t, c_void_p],
    ),
    "nsa_populate_train_commit": (c_int, [c_void_p, c_ll, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nsa_populate_sumexp": (c_int, [c_ll, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nsa_repack_image": (c_int, [c_ll, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "nsa_probe_compare": (c_int, [c_ll, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nsa_probe_nudge": (c_int, [c_ll, c_void_p, c_void_p, c_void_p]),
    "nsa_probe_range": (c_int, [c_ll, c_void_p, c_float, c_void_p, c_void_p]),
    "nsa_importance_draw": (
        c_int,
        [c_void_p, c_void_p, c_ll, c_void_p, c_ull, c_ull, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
         c_int, c_void_p],
    ),
    "nsa_importance_to_prime": (c_int, [c_ll, c_int, c_void_p, c_double, c_void_p, c_void_p, c_void_p]),
    "nsa_importance_mix_packed": (c_int, [c_ll, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                          c_void_p]),
    "nsa_importance_mix_points": (c_int, [c_ll, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                          c_void_p, c_double, c_void_p, c_void_p, c_void_p]),
    "nsa_importance_mix": (c_int, [c_ll, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nsa_populate_truncate": (c_int, [c_ll, c_void_p, c_double, c_void_p, c_double, c_void_p]),
    "nsa_populate_weights": (c_int, [c_ll, c_void_p, c_void_p, c_double, c_void_p, c_void_p, c_void_p]),
    "nsa_max_key_init": (c_ll, []),
    "nsa_max_key_encode": (c_ll, [c_double]),
    "nsa_max_key_decode": (c_double, [c_ll]),
    "nsa_populate_accept_workspace": (c_size_t, [c_ll]),
    "nsa_train_workspace_floats": (c_size_t, [c_void_p, c_ll]),
    "nsa_train_context_offset": (c_size_t, [c_void_p, c_ll]),
    "nsa_log_likelihood": (c_int, [c_int, c_ll, c_int, c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_void_p]),
    "nsa_train_step_dev": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_ll, c_void_p,
         c_void_p, c_float, c_float, c_float, c_float, c_float, c_int, c_void_p, c_void_p],
    ),
    "nsa_train_step": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_ll,
         c_void_p, c_float, c_float, c_float, c_float, c_float, c_float, c_ll, c_int, c_void_p, c_void_p],
    ),
    "nsa_train_eval": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_ll, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "nsa_train_forward": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_ll, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
         c_void_p],
    ),
    "nsa_train_backward": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_ll, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
         c_void_p, c_void_p],
    ),
    "nsa_populate_accept": (
        c_int,
        [c_ll, c_int, c_void_p, c_void_p, c_ull, c_ull, c_void_p, c_void_p, c_ll, c_void_p, c_void_p,
         c_void_p, c_void_p, c_size_t, c_void_p],
    ),
}


def library_path():
    return _LIB_PATH


def load():
    """Load the shared library (once) and declare the prototypes."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            f"nessai_amd HIP library not found at {_LIB_PATH}. Build it with "
            "`make -C nessai_amd/csrc` (or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "There is no CPU fallback."
        )
    lib = ctypes.CDLL(_LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _check_instance_matrix(lib)
    _LIB = lib
    return lib


def _check_instance_matrix(lib):
    """The packer lays a spline flow out for the kernel instance ``pack.kernel_bins`` names - the EXACT layout for
    4 / 8 / 10 / 16 bins, the padded 16-bin one otherwise - while the library picks the instance it will launch by
    itself (``nsa_find``: the exact one, else the generic one).  A library built with a reduced ``TRANSFORMS`` list
    (csrc/Makefile) could therefore run an exact-layout image on the generic instance: refuse such a build here."""
    from .pack import EXACT_BINS

    rq = 1  # NSA_T_RQ (csrc/nsa_format.h)
    for ht in (1, 2, 4, 8):
        for bins in EXACT_BINS:
            if lib.nsa_flow_supported(ht, rq, bins) and lib.nsa_flow_generic_bins(ht, bins):
                raise RuntimeError(
                    f"{_LIB_PATH} has no exact spline instance for {bins} bins at {ht} hidden tile(s) but would run "
                    "such a flow on the generic instance, whose parameter layout differs from what nessai_amd.pack "
                    "writes for this bin count: rebuild with the full TRANSFORMS list of nessai_amd/csrc/Makefile"
                )


def check(rc, what):
    if rc != 0:
        msg = load().nsa_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def ptr(tensor):
    """Device/host address of a tensor or ``None``."""
    if tensor is None:
        return None
    return ctypes.c_void_p(tensor.data_ptr())
