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
    # -- draw / update_log_q: the reference's own loop on a stand-in flow with fixed outputs ----
    names = [f"x_{i}" for i in range(d)]
    batches = []
    for k in range(3):
        xp = rng.normal(0.0, 2.5, size=(int(1.01 * 150), d))
        if k == 0:
            xp[1, 2] = np.nan       # non-finite flow sample
            xp[2, 0] = 800.0        # sigmoid saturates at 1: log|J_inv| = -inf
            xp[4, 1] = -800.0       # ... at 0
            xp[5, 3] = np.inf
        batches.append(xp)
    lp_tables = [rng.normal(-9.0, 2.0, size=(len(b), n_flows)) for b in batches]
    lp_tables[0][7, :] = np.inf     # every flow +inf: rejected by the second filter
    lp_tables[0][8, 0] = -np.inf
    state = dict(k=0)

    class Model:
        names = None

        def in_unit_hypercube(self, x):
            return ~np.any([(x[nm] < 0) | (x[nm] > 1) for nm in names], axis=0)

        def batch_evaluate_log_prior(self, x, unit_hypercube=False):
            out = np.full(x.size, -3.25)
            out[x[names[0]] > 0.97] = -np.inf   # prior rejects a corner
            return out

        def batch_evaluate_log_prior_unit_hypercube(self, x):
            return np.log(self.in_unit_hypercube(x), dtype=float)

    model = Model()
    model.names = names

    class FlowStub:
        n_models = n_flows
        models = [types.SimpleNamespace(training=False) for _ in range(n_flows)]

        def sample_ith(self, i, N):
            assert N == len(batches[state["k"]])
            return batches[state["k"]].copy()

        def log_prob_all(self, x):
            table = lp_tables[state["k"]]
            keep = np.array([np.flatnonzero(np.all((batches[state["k"]] == row) | (np.isnan(batches[state["k"]]) & np.isnan(row)), axis=1))[0] for row in x])
            state["k"] += 1
            return table[keep]

        def log_prob_ith(self, x, i):
            return fixed_ith

    from nessai.livepoint import get_dtype

    me = types.SimpleNamespace(
        level_count=n_flows - 1, dtype=None, model=model, clip=False,
        reparameterisation="logit", weights=weights, weights_array=np.fromiter(weights.values(), float),
        n_proposals=n_flows + 1, flow=FlowStub(),
    )
    from nessai.livepoint import add_extra_parameters_to_live_points

    add_extra_parameters_to_live_points(["logW", "logQ", "logU"])  # samplers/importancesampler.py:649
    me.dtype = get_dtype(names)
    for name in ("to_prime", "from_prime", "rescale", "inverse_rescale", "compute_log_Q", "get_proposal_log_prob"):
        setattr(me, name, types.MethodType(getattr(Ref, name), me))
    drawn, drawn_log_q = Ref.draw(me, 150)
    out.update(draw_names=np.array(names), draw_batches=np.stack(batches), draw_lp_tables=np.stack(lp_tables),
               draw_batches_used=np.array(state["k"]),
               draw_x=np.stack([drawn[nm] for nm in names], axis=1), draw_logQ=drawn["logQ"], draw_logW=drawn["logW"],
               draw_logP=drawn["logP"], draw_logU=drawn["logU"], draw_log_q=drawn_log_q)
    # update_log_q: a (n, n_flows) table gains the column of the newest flow
    fixed_ith = rng.normal(-7.0, 1.0, size=drawn.size)
    me.level_count = n_flows - 1
    upd = Ref.update_log_q(me, drawn, drawn_log_q[:, :-1])
    out.update(update_ith=fixed_ith, update_log_q=upd)

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "importance_goldens.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
