"""CPU oracle for the normalising-flow proposal path — TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (eager PyTorch fp32 for the flow arithmetic,
numpy float64 for the proposal glue) of the algorithm the reference runs on
its hot path.  It exists so that the HIP kernels in
``nessai_amd/csrc`` can be checked against an independent implementation.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` — as the checker / the timed CPU
baseline, never as the thing shipped.  Nothing under ``nessai_amd/`` imports
this package; the product path raises if the HIP library is missing.

PARITY STATUS — "parity unpinned" at the glasflow boundary
----------------------------------------------------------
The coupling / spline / ResidualNet / LU / BatchNorm arithmetic of the
reference is *not* in the reference tree: it is imported from the third-party
package ``glasflow`` (which vendors a fork of ``nflows`` as
``glasflow.nflows``), un-pinned in ``pyproject.toml:25`` and not installed in
the build container (no network).  ``oracle/nflows_restated`` therefore
restates the *published* nflows algorithm (Durkan et al., "Neural Spline
Flows", NeurIPS 2019, and the public ``nflows`` 0.14 API) from its
specification; the reference's own tests hold **no numeric golden vectors**
for these transforms, so the arithmetic cannot be pinned
against the real dependency here.  What *is* pinned:

* the reference's own property tests for this path (invertibility,
  ``log_J == -log_J_inv``, ``sample_and_log_prob == log_prob``,
  ``forward_and_log_prob == forward + log_prob``, Gaussian base log-prob vs
  scipy, constant-volume radius, BatchNorm reset constants) — see
  ``tests/test_oracle_properties.py``;
* analytic known-answer tests derived from the spline definition (knots map
  to knots, identity tails with zero log-det, zero-parameter spline);
* the rational-quadratic spline itself, **bit for bit**, against an independent
  third-party port of the nflows spline functions that IS installed here:
  HuggingFace ``transformers`` 5.15.0, ``models/vits/modeling_vits.py``
  ``_unconstrained_rational_quadratic_spline`` (forward and inverse, K = 4, 8, 10, 16,
  inputs inside / on / beyond the tail bound): ``tests/golden/make_spline_goldens.py``
  -> ``tests/golden/spline_hf_goldens.npz`` ->
  ``tests/test_oracle_properties.py::test_spline_matches_the_independent_huggingface_port_of_nflows``.
  This is not the reference's pinned dependency (there is none), so the header keeps the
  words "parity unpinned"; it does remove the risk of a mis-restated spline;
* the *assembly* of the flows and the whole proposal glue, against the real
  reference code: ``tests/golden/make_golden.py`` imports ``nessai`` from
  ``/root/reference`` in the build container with ``glasflow`` replaced by
  this restatement, drives the reference's own ``RealNVP`` /
  ``NeuralSplineFlow`` / ``FlowModel`` / ``FlowProposal`` and stores inputs
  and outputs as fixtures under ``tests/golden/``.
"""
