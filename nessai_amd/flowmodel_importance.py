"""``ImportanceFlowModel``: a list of flows for the importance nested sampler.

Counterpart of the reference's ``nessai.flowmodel.importance.ImportanceFlowModel``
(``src/nessai/flowmodel/importance.py:22``): every level of the importance sampler adds a
flow; ``log_prob_all`` evaluates every stored flow on the same batch (the dominant cost of
config 4) and ``sample_ith`` draws from one of them.  Here each flow is a
``HipFlow``: the batch is uploaded once, all flows are evaluated by ONE grouped launch
(``nsa_flow_run_many``: gridDim.y = flow index, so a 5000-row batch under 8 flows fills the chip)
writing an ``(n_flows, N)`` device matrix, and the matrix comes back in one copy.
"""

import copy
import glob
import logging
import os

import numpy as np
import torch

from .flowmodel import FlowModel
from .flows import configure_model
from .flows.base import log_prob_many
from .utils import eval_mode

logger = logging.getLogger(__name__)


class ImportanceFlowModel(FlowModel):
    models = None
    _resume_n_models = None

    def __init__(self, flow_config=None, training_config=None, output=None, rng=None, precision="auto"):
        """``precision``: arithmetic of the flows' conditioner GEMMs; ``"auto"`` applies the policy of
        the populate kernel (``nessai_amd.flows.base.auto_candidates``: split-f16, then split-bf16 inside its
        validated envelope, e.g. the 64-D / 128-hidden flows of config 4; exact float32 otherwise)."""
        super().__init__(flow_config=flow_config, training_config=training_config, output=output, rng=rng)
        if precision not in ("auto", "f32", "bf16x3", "f16x3"):
            raise ValueError("precision must be 'auto', 'f32', 'bf16x3' or 'f16x3'")
        self.precision = precision
        self.weights_files = []
        self.models = torch.nn.ModuleList()

    @property
    def model(self):
        """The newest flow, or ``None`` (``importance.py:41-50``)."""
        if self.models:
            return self.models[-1]
        return None

    @model.setter
    def model(self, model):
        if model is not None:
            self.models.append(model)

    @property
    def n_models(self):
        return len(self.models) if self.models else 0

    def initialise(self):
        self.device = torch.device(self.training_config.get("device_tag", "cuda"))
        if self.device.type != "cuda":
            raise RuntimeError("nessai_amd needs a ROCm device; there is no CPU fallback")
        self.inference_device = self.device
        self.initialised = True

    def add_new_flow(self, reset=False):
        """Append a flow: fresh, or a copy of the newest one (``importance.py:80-98``)."""
        if not self.initialised:
            self.initialise()
        if reset or not self.models:
            new_flow = configure_model(self.flow_config)
        else:
            new_flow = copy.deepcopy(self.model)
            new_flow._image_key = None
        new_flow.to(self.device)
        self._set_precision(new_flow)
        self.models.eval()
        self.models.append(new_flow)
        self._trainer = None  # the optimiser follows the current model (reset_optimiser)

    def _set_precision(self, flow):
        flow.precision = self.precision  # "auto" is resolved by the flow per set of weights (HipFlow._pack_auto)

    def _tensor(self, x):
        return torch.from_numpy(np.ascontiguousarray(x)).type(torch.float32).to(self.device)

    def log_prob_ith(self, x, i):
        """log-density of the i-th flow (``importance.py:100-112``)."""
        eval_mode(self.models[i])
        with torch.inference_mode():
            return self._to_numpy(self.models[i].log_prob(self._tensor(x)))

    def log_prob_all(self, x):
        """``(N, n_flows)`` log-densities, every flow on the same batch (``importance.py:114-129``)."""
        xt = self._tensor(x)
        eval_mode(self.models)
        out = log_prob_many(self.models, xt)
        return self._to_numpy(out.t())

    def sample_ith(self, i, N=1):
        """Draw from the i-th flow (``importance.py:131-142``)."""
        if self.models is None:
            raise RuntimeError("Models are not initialised yet!")
        eval_mode(self.models[i])
        with torch.inference_mode():
            return self._to_numpy(self.models[i].sample(int(N)))

    def save_weights(self, weights_file):
        super().save_weights(weights_file)
        self.weights_files.append(self.weights_file)

    def load_all_weights(self):
        """(``importance.py:149-166``)."""
        self.models = torch.nn.ModuleList()
        if not self.initialised:
            self.initialise()
        for wf in self.weights_files:
            new_flow = configure_model(self.flow_config)
            new_flow.load_state_dict(torch.load(wf, map_location="cpu", weights_only=True))
            new_flow.to(self.device)
            self._set_precision(new_flow)
            self.models.append(new_flow)
        self.models.eval()

    def update_weights_path(self, weights_path, n=None):
        """(``importance.py:168-207``)."""
        all_weights_files = glob.glob(os.path.join(weights_path, "", "level_*", "model.pt"))
        if n is None:
            if self.n_models:
                n = self.n_models
            else:
                raise RuntimeError("n is None and no models are defined, cannot update weights path.")
        if len(all_weights_files) < n:
            raise RuntimeError(f"Cannot use weights from: {weights_path}. Not enough files.")
        self.weights_files = [os.path.join(weights_path, f"level_{i}", "model.pt") for i in range(n)]

    def resume(self, flow_config, training_config=None, weights_path=None):
        from .flowmodel import update_config

        self.flow_config, tc = update_config(flow_config, training_config or self.training_config)
        if training_config is not None:
            self.training_config = tc
        if weights_path is None:
            weights_path = self.output
        self.update_weights_path(weights_path, n=self._resume_n_models)
        self.load_all_weights()
        self.initialise()

    def __getstate__(self):
        d = self.__dict__
        exclude = {"models", "_trainer", "flow_config"}
        state = {k: d[k] for k in d.keys() - exclude}
        state["initialised"] = False
        state["models"] = None
        state["_trainer"] = None
        state["_resume_n_models"] = len(d["models"])
        return state
