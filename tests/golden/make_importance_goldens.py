#!/usr/bin/env python
"""Golden vectors G6 for the importance-sampler glue, from the REFERENCE's code.

Run in the build container only (needs /root/reference):

    python tests/golden/make_importance_goldens.py

Calls the reference's ``ImportanceFlowProposal.to_prime / from_prime / compute_log_Q /
compute_meta_proposal_from_log_q`` as unbound methods on a stand-in ``self`` that carries fixed
weights and a fixed ``log_prob_all`` matrix (no flow arithmetic involved), and stores inputs and
outputs in ``importance_goldens.npz``.
"""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import reference_harness  # noqa: E402

reference_harness.install()

from nessai.proposal.importance import ImportanceFlowProposal as Ref  # noqa: E402


def main():
    rng = np.random.default_rng(20260930)
    n, d, n_flows = 211, 5, 3
    x_unit = rng.random((n, d))
    x_unit[0, :] = [0.0, 1.0, 1e-12, 1 - 1e-12, 0.5]  # clamped to [eps, 1 - eps] by the logit
    me = types.SimpleNamespace(reparameterisation="logit")
    x_prime, log_j = Ref.to_prime(me, x_unit)
    x_back, log_j_inv = Ref.from_prime(me, x_prime)

    log_prob_all = rng.normal(-8.0, 3.0, size=(n, n_flows))
    log_prob_all[3, 1] = -np.inf
    weights = {-1: 0.1, 0: 0.2, 1: 0.3, 2: 0.4}
    flow = types.SimpleNamespace(
        n_models=n_flows,
        models=[types.SimpleNamespace(training=False) for _ in range(n_flows)],
        log_prob_all=lambda x: log_prob_all,
    )
    me = types.SimpleNamespace(
        weights=weights, weights_array=np.fromiter(weights.values(), float), n_proposals=n_flows + 1, flow=flow
    )
    log_Q, log_q_all = Ref.compute_log_Q(me, x_prime, log_j=log_j)
    log_Q_from_q = Ref.compute_meta_proposal_from_log_q(me, log_q_all)

    out = dict(x_unit=x_unit, x_prime=x_prime, log_j=log_j, x_back=x_back, log_j_inv=log_j_inv,
               log_prob_all=log_prob_all, weights=np.fromiter(weights.values(), float), log_Q=log_Q,
               log_q_all=log_q_all, log_Q_from_q=log_Q_from_q)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "importance_goldens.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
