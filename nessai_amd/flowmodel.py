"""``FlowModel``: numpy-float64-in / numpy-float64-out wrapper around a HIP flow.

Counterpart of the reference's ``nessai.flowmodel.FlowModel``
(``src/nessai/flowmodel/base.py:25``): same constructor, same method names and
return conventions, so ``BaseFlowProposal._FlowModelClass`` could point at it
(SURVEY.md §8b surface B2).  Host arrays are converted to float32 and moved to
the ROCm device (``numpy_array_to_tensor``, ``base.py:774-780``); results come
back as float64 numpy arrays (``base.py:809-810``).
"""

import logging
import os

import numpy as np
import torch

from .flows import configure_model, reset_permutations, reset_weights
from .flows.utils import get_n_neurons
from .utils import eval_mode

logger = logging.getLogger(__name__)

DEFAULT_FLOW_CONFIG = dict(  # src/nessai/flowmodel/config.py:13-28
    n_inputs=None,
    n_neurons=None,
    n_blocks=4,
    n_layers=2,
    ftype="realnvp",
)

DEFAULT_TRAINING_CONFIG = dict(  # src/nessai/flowmodel/config.py:31-46
    device_tag="cuda",
    lr=1e-3,
    annealing=False,
    clip_grad_norm=5.0,
    batch_size=1000,
    val_size=0.1,
    max_epochs=500,
    patience=20,
    noise_type=None,
    noise_scale=None,
    use_dataloader=False,
    optimiser="adamw",
    optimiser_kwargs=None,
    # not in the reference: bit-reproducible fits.  The fused coupling kernels add the weight gradients
    # of a workgroup's rows with float atomics (run-to-run differences in the last bits); True selects
    # the layer-wise kernels, whose reductions run in a fixed order (about 2x slower per step).
    deterministic=False,
)


def update_config(flow_config, training_config):
    """Merge user dictionaries over the defaults (``flowmodel/utils.py:12-87``)."""
    fc = dict(DEFAULT_FLOW_CONFIG)
    fc.update(flow_config or {})
    if fc.get("n_inputs") is not None:
        fc["n_neurons"] = get_n_neurons(fc.get("n_neurons"), n_inputs=fc["n_inputs"])
    tc = dict(DEFAULT_TRAINING_CONFIG)
    tc.update(training_config or {})
    if tc["noise_type"] and not tc["noise_scale"]:
        raise RuntimeError("`noise_scale` must be specified when `noise_type` is given.")
    return fc, tc


class FlowModel:
    model = None

    def __init__(self, flow_config=None, training_config=None, output=None, rng=None):
        if output is None:
            output = os.getcwd()
        self.model = None
        self.rng = rng if rng is not None else np.random.default_rng()
        self.initialised = False
        self.output = output
        os.makedirs(self.output, exist_ok=True)
        self.flow_config, self.training_config = update_config(flow_config, training_config)
        self.weights_file = None
        self._batch_size = None
        self._trainer = None

    # -- set-up ----------------------------------------------------------------

    def initialise(self):
        """Build the flow and place it on the ROCm device (``base.py:147-176``)."""
        self.model = configure_model(self.flow_config)
        self.device = torch.device(self.training_config.get("device_tag", "cuda"))
        if self.device.type != "cuda":
            raise RuntimeError(
                "nessai_amd.FlowModel needs a ROCm device (device_tag='cuda'); "
                "there is no CPU fallback"
            )
        self.model.to(self.device)
        self.inference_device = self.device
        eval_mode(self.model)
        self._trainer = None
        self.initialised = True

    def move_to(self, device, update_default=False):
        device = torch.device(device)
        self.model.to(device)
        if update_default:
            self.device = device

    @staticmethod
    def check_batch_size(x, batch_size, min_fraction=0.1):
        """Largest batch size <= ``batch_size`` whose last batch is either empty or holds at least
        ``min_fraction`` of the REQUESTED size (same outcome as ``base.py:195-236``; pinned by
        ``tests/test_golden.py``).  Sizes at or below that floor are accepted with a warning as soon
        as they leave more than one sample over."""
        if batch_size == 1:
            raise ValueError("Cannot use a batch size of 1!")
        n = len(x)
        floor = int(min_fraction * batch_size)

        def tail_ok(size):
            rest = n % size
            return rest == 0 or rest >= floor

        if tail_ok(batch_size):
            return batch_size
        for size in range(batch_size - 1, 1, -1):
            if tail_ok(size):
                return size
            if size <= floor and n % size > 1:
                logger.warning(f"Batch size is less than {floor} but valid. Setting batch size to: {size}")
                return size
        raise RuntimeError("Could not find a valid batch size")

    # -- numpy <-> device ------------------------------------------------------

    def numpy_array_to_tensor(self, array):
        return (
            torch.from_numpy(np.ascontiguousarray(array))
            .type(torch.float32)
            .to(self.model.device)
        )

    @staticmethod
    def _to_numpy(t):
        return t.detach().cpu().numpy().astype(np.float64)

    def _context(self, conditional):
        """``conditional`` (numpy, one row per sample) -> the flow's ``context`` tensor, ``None`` passed through."""
        return None if conditional is None else self.numpy_array_to_tensor(conditional)

    def forward_and_log_prob(self, x, conditional=None):
        """x' -> (z, log_prob) (``base.py:782-811``)."""
        eval_mode(self.model)
        with torch.inference_mode():
            z, log_prob = self.model.forward_and_log_prob(self.numpy_array_to_tensor(x),
                                                          context=self._context(conditional))
        return self._to_numpy(z), self._to_numpy(log_prob)

    def inverse(self, z, conditional=None):
        """z -> (x', log|det J|) (``base.py:813-840``)."""
        eval_mode(self.model)
        with torch.inference_mode():
            x, log_j = self.model.inverse(self.numpy_array_to_tensor(z), context=self._context(conditional))
        return self._to_numpy(x), self._to_numpy(log_j)

    def log_prob(self, x, conditional=None):
        """(``base.py:842-864``)."""
        eval_mode(self.model)
        with torch.inference_mode():
            return self._to_numpy(self.model.log_prob(self.numpy_array_to_tensor(x), context=self._context(conditional)))

    def sample(self, n=1, conditional=None):
        """(``base.py:866-887``)."""
        eval_mode(self.model)
        with torch.inference_mode():
            return self._to_numpy(self.model.sample(int(n), context=self._context(conditional)))

    def sample_latent_distribution(self, n=1):
        """(``base.py:889-904``)."""
        return self._to_numpy(self.model.sample_latent_distribution(int(n)))

    def sample_and_log_prob(self, N=1, z=None, conditional=None):
        """(``base.py:906-948``)."""
        if self.model is None:
            raise RuntimeError("Model is not initialised yet!")
        eval_mode(self.model)
        context = self._context(conditional)
        with torch.inference_mode():
            if z is None:
                x, log_prob = self.model.sample_and_log_prob(int(N), context=context)
            else:
                if isinstance(z, np.ndarray):
                    z = self.numpy_array_to_tensor(z)
                log_prob = self.model.base_distribution_log_prob(z)
                x, log_J = self.model.inverse(z, context=context)
                log_prob = log_prob - log_J
        return self._to_numpy(x), self._to_numpy(log_prob)

    # -- training --------------------------------------------------------------

    def prep_data(self, samples, val_size, batch_size, weights=None, conditional=None):
        """Shuffle, split train/validation, fix the batch size (``base.py:238-352``).

        With ``weights`` returns ``((x_train, w_train), (x_val, w_val), batch_size)``, with ``conditional``
        (one row of context per sample) ``((x_train, c_train), (x_val, c_val), batch_size)``."""
        if not self.initialised:
            self.initialise()
        if not np.isfinite(samples).all():
            raise ValueError("Cannot train with non-finite samples!")
        if weights is not None and conditional is not None:
            raise RuntimeError("weights and conditional inputs not supported")
        idx = self.rng.permutation(samples.shape[0])
        samples = samples[idx]
        if conditional is not None:
            conditional = np.asarray(conditional, dtype=np.float64)
            if conditional.ndim == 1:
                conditional = conditional[:, None]
            conditional = conditional[idx]
        if weights is not None:
            if not np.isfinite(weights).all():
                raise ValueError("Weights contain non-finite values!")
            weights = np.asarray(weights, dtype=np.float64)[idx]
        if val_size is None:
            val_size = 0
        n = int((1 - val_size) * samples.shape[0])
        x_train, x_val = samples[:n], samples[n:]
        if isinstance(batch_size, bool) or not isinstance(batch_size, int):
            if batch_size == "all" or batch_size is None:
                batch_size = x_train.shape[0]
            else:
                raise RuntimeError(f"Unknown batch size: {batch_size}")
        batch_size = self.check_batch_size(x_train, batch_size)
        self._batch_size = batch_size
        if weights is not None:
            return (
                (self.numpy_array_to_tensor(x_train), self.numpy_array_to_tensor(weights[:n])),
                (self.numpy_array_to_tensor(x_val), self.numpy_array_to_tensor(weights[n:])),
                batch_size,
            )
        if conditional is not None:
            return (
                (self.numpy_array_to_tensor(x_train), self.numpy_array_to_tensor(conditional[:n])),
                (self.numpy_array_to_tensor(x_val), self.numpy_array_to_tensor(conditional[n:])),
                batch_size,
            )
        return (
            self.numpy_array_to_tensor(x_train),
            self.numpy_array_to_tensor(x_val),
            batch_size,
        )

    def train(self, samples, weights=None, conditional=None, max_epochs=None, patience=None,
              output=None, val_size=None, plot=False):
        """Maximum-likelihood fit of the flow (``base.py:530-696``).

        Loss ``-mean(log_prob)``, AdamW, gradient-norm clipping, early stopping
        on the validation loss with best-weights restore.  The forward, backward
        and optimiser arithmetic run in the gfx950 kernels (``nessai_amd.train``).
        """
        from .train import train_flow

        if not self.initialised:
            self.initialise()
        if not np.isfinite(samples).all():
            raise ValueError("Training data is not finite")
        if output is None:
            output = self.output
        else:
            os.makedirs(output, exist_ok=True)
        if val_size is None:
            val_size = self.training_config["val_size"]
        history = train_flow(self, samples, max_epochs=max_epochs, patience=patience, val_size=val_size,
                             weights=weights, conditional=conditional)
        current_weights_file = os.path.join(output, "model.pt")
        self.save_weights(current_weights_file)
        eval_mode(self.model)
        return history

    # -- weights I/O -----------------------------------------------------------

    def save_weights(self, weights_file):
        """``torch.save(state_dict)`` with a ``.old`` backup (``base.py:698-712``)."""
        if os.path.exists(weights_file):
            os.replace(weights_file, weights_file + ".old")
        torch.save(self.model.state_dict(), weights_file)
        self.weights_file = weights_file

    def load_weights(self, weights_file):
        """(``base.py:714-731``)."""
        if not self.initialised:
            self.initialise()
        self.model.load_state_dict(
            torch.load(weights_file, map_location=self.device, weights_only=True)
        )
        eval_mode(self.model)
        self.weights_file = weights_file

    def reload_weights(self, weights_file):
        if weights_file is None:
            weights_file = self.weights_file
        self.load_weights(weights_file)

    def reset_model(self, weights=True, permutations=False):
        """(``base.py:748-772``)."""
        if not any([weights, permutations]):
            return
        if weights and permutations:
            self.model = configure_model(self.flow_config)
            self.model.to(self.device)
            eval_mode(self.model)
        elif weights:
            self.model.apply(reset_weights)
        elif permutations:
            self.model.apply(reset_permutations)
        self._trainer = None

    def __getstate__(self):
        state = self.__dict__.copy()
        state["initialised"] = False
        state["_trainer"] = None
        del state["model"]
        del state["flow_config"]
        return state
