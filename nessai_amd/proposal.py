"""``FlowProposal`` with the populate loop fused on the MI355X.

Counterpart of the reference's ``nessai.proposal.FlowProposal``
(``src/nessai/proposal/flowproposal/flowproposal.py:29`` and its base
``proposal/flowproposal/base.py:40``): per-parameter reparameterisations (z-score by
default, ``reparameterisations/rescale.py:119-312``; rescale-to-bounds / logit / null through
``nessai_amd.reparam``), latent-radius truncation
(``truncation.py:201-365``, constant-volume by default, ``truncation.py:15-21``)
and rejection sampling against the prior.  The sampler-facing protocol is the
reference's (SURVEY.md §8b surface B3): ``initialise()``, ``train(live_points)``,
``draw(worst_point)``, ``populate(...)``, ``populated``, ``indices``,
``samples``, ``x``, ``population_acceptance``, ``population_time``.

One pass of the populate loop is three device steps and one small copy:

1. ``nsa_populate_draw``  — Philox latent draw, radius truncation, inverse
   flow, inverse reparameterisation, prior bounds, ``log_q`` (one fused kernel);
2. ``nsa_populate_weights`` — ``log_w = log_prior - log_q`` and its maximum
   (all-reduced over ranks when the draw batch is sharded);
3. ``nsa_populate_accept`` — ``log_w - max > log(u)`` and order-preserving
   compaction;
4. the accepted rows only are copied to the host and packed into live points.
"""

import datetime
import math
import logging
import os
import time

import numpy as np
import torch

from . import lib
from .flowmodel import FlowModel
from .livepoint import (
    array_to_live_points_fast,
    copy_points,
    save_live_points,
    get_dtype,
    live_points_to_array,
)
from .reparam import ReparameterisationTable
from .truncation import build_scheme, resolve_truncation
from .utils import compute_radius

logger = logging.getLogger(__name__)


class FlowProposal:
    """Draw live points from a trained flow by rejection sampling.

    Parameters mirror the reference where they exist there: ``model``, ``rng``,
    ``flow_config``, ``training_config``, ``output``, ``poolsize``,
    ``drawsize``, ``volume_fraction``, ``fixed_radius``, ``radius_mode``
    ({'constant_volume', 'fixed', 'adaptive'}), ``min_radius``, ``max_radius``,
    ``accumulate_weights``, ``max_poolsize_scale``, ``update_poolsize``.

    Additional parameters
    ---------------------
    device_seed : int, optional
        Seed of the on-device Philox streams (default: drawn from ``rng``).
    prior : {'auto', 'device', 'host'}
        ``'device'`` evaluates the prior on the GPU: a constant for a
        uniform-in-bounds prior (``model.uniform_prior``) or the separable
        analytic densities declared in ``model.device_prior``; ``'host'`` copies
        every candidate to the host and calls ``model.log_prior`` (any prior);
        ``'auto'`` picks ``'device'`` when the model declares either.
    likelihood : {'auto', 'device', 'host'}
        ``'device'`` evaluates the log-likelihood of the accepted rows on the GPU right after
        compaction for models that declare ``device_likelihood`` (the analytic benchmark models);
        ``'host'`` calls ``model.batch_evaluate_log_likelihood`` as the reference does
        (``flowproposal.py:521``); ``'auto'`` picks ``'device'`` when declared.
    shard : bool
        When ``torch.distributed`` is initialised, split every draw batch over
        the ranks (one all-reduce(max) + one all-gather per pass).
    pass_trains : bool
        Keep the loop control of ``populate`` on the device (``nsa_populate_passes``): several
        passes are enqueued per host synchronisation and surplus ones switch themselves off.
        Results are identical to ``False`` (one pass per synchronisation).
    small_trains : bool
        Passes of at most 16384 rows (the reference's default ``drawsize = nlive``): run a whole train
        of them as one draw launch plus two tail launches (``nsa_populate_passes_small``) instead of
        three launches per pass.  Results are identical to ``False``.
    lean : bool
        Big passes of the device-side loops keep NO candidate rows (``nsa_populate_passes_lean``): a pass
        writes ``log_q`` alone (8 bytes per row instead of ``8 (D + 1)``), the accept step lists the accepted
        rows by their global Philox rows and ``nsa_populate_replay`` computes just those again at the end of
        ``populate`` - only accepted rows are ever used (``flowproposal.py:431-498``).  Results are identical
        to ``False`` bit for bit.  Not used with a likelihood-threshold rule, ``accumulate_weights``,
        injected latents or a host prior, which need every candidate.
    precision : {'auto', 'f32', 'bf16x3', 'f16x3'}
        Arithmetic of the conditioner GEMMs inside the fused draw kernel.
        ``'auto'`` uses the fastest 16-bit matrix-core mode (split-bf16, then split-f16) that is
        MEASURED to agree with exact float32 on the flow's current weights to half the 1e-5
        relative parity bar (``nessai_amd.flows.base.HipFlow._pack_auto``) and exact float32
        otherwise; a split-f16 population that meets a value beyond the f16 range is detected and
        drawn again in float32.
    """

    _FlowModelClass = FlowModel
    #: class default of ``use_default_reparameterisations`` (``proposal/flowproposal/base.py:95``; a subclass with
    #: defaults of its own - the reference's GW proposal - sets it and overrides ``add_default_reparameterisations``)
    use_default_reparameterisations = False

    def __init__(
        self,
        model,
        rng=None,
        flow_config=None,
        training_config=None,
        output=None,
        poolsize=None,
        drawsize=None,
        latent_temperature=None,
        truncation_method=None,
        truncation_methods=None,
        truncation_kwargs=None,
        truncate_log_q=False,
        enforce_likelihood_threshold=False,
        constant_volume_mode=None,
        volume_fraction=None,
        fuzz=None,
        expansion_fraction=None,
        fixed_radius=None,
        radius_mode=None,
        min_radius=None,
        max_radius=None,
        compute_radius_with_all=None,
        accumulate_weights=False,
        update_poolsize=True,
        max_poolsize_scale=10,
        check_acceptance=False,
        device_seed=None,
        prior="auto",
        likelihood="auto",
        shard=False,
        replicate_training=False,
        precision="auto",
        pass_trains=True,
        small_trains=True,
        lean=True,
        reparameterisations=None,
        fallback_reparameterisation="z-score",
        save_training_data=False,
        use_default_reparameterisations=None,
        reverse_reparameterisations=False,
        map_to_unit_hypercube=False,
        plot=False,
        **kwargs,
    ):
        if kwargs:
            raise TypeError(f"Unknown FlowProposal arguments: {sorted(kwargs)}")
        # the rest of BaseFlowProposal's keyword surface (proposal/flowproposal/base.py:116-121, 140-158)
        self.save_training_data = bool(save_training_data)
        if use_default_reparameterisations is not None:
            self.use_default_reparameterisations = bool(use_default_reparameterisations)
        self.reverse_reparameterisations = bool(reverse_reparameterisations)
        self.map_to_unit_hypercube = bool(map_to_unit_hypercube)
        self._prior_bounds = None
        if latent_temperature is not None:  # flowproposal.py:263-272
            if isinstance(latent_temperature, bool) or not isinstance(latent_temperature, (int, float)):
                raise TypeError("latent_temperature must be a positive float")
            if not latent_temperature > 0.0:
                raise ValueError("latent_temperature must be a positive float")
        self.model = model
        self.rng = rng if rng is not None else np.random.default_rng()
        self.output = output if output is not None else os.getcwd()
        self.flow_config = dict(flow_config or {})
        self.training_config = dict(training_config or {})
        if poolsize is None:
            raise RuntimeError("Must specify a poolsize!")
        self._poolsize = int(poolsize)
        self._poolsize_scale = 1.0
        self.update_poolsize = update_poolsize
        self.max_poolsize_scale = max_poolsize_scale
        self.drawsize = int(drawsize) if drawsize is not None else self._poolsize
        self.latent_temperature = None if latent_temperature is None else float(latent_temperature)
        self.accumulate_weights = accumulate_weights
        self.check_acceptance = check_acceptance
        # truncation rules (flowproposal.py:246-331): the new arguments plus the deprecated
        # latent-radius spellings, resolved exactly as the reference resolves them
        legacy = dict(constant_volume_mode=constant_volume_mode, volume_fraction=volume_fraction, fuzz=fuzz,
                      expansion_fraction=expansion_fraction, fixed_radius=fixed_radius, radius_mode=radius_mode,
                      min_radius=min_radius, max_radius=max_radius, compute_radius_with_all=compute_radius_with_all)
        names, rule_kwargs = resolve_truncation(
            truncation_method=truncation_method, truncation_methods=truncation_methods,
            truncation_kwargs=truncation_kwargs, truncate_log_q=truncate_log_q,
            enforce_likelihood_threshold=enforce_likelihood_threshold, legacy=legacy,
        )
        self.truncation = build_scheme(names, rule_kwargs)
        if prior not in ("auto", "device", "host"):
            raise ValueError("prior must be 'auto', 'device' or 'host'")
        self._prior_mode = prior
        if likelihood not in ("auto", "device", "host"):
            raise ValueError("likelihood must be 'auto', 'device' or 'host'")
        self._likelihood_mode = likelihood
        self.shard = shard
        #: sharded runs: every rank repeats the fit (True) or rank 0 fits and the others receive the flow (False)
        self.replicate_training = bool(replicate_training)
        self.small_trains = bool(small_trains)
        self.lean = bool(lean)
        #: True when the last ``populate`` ran lean passes + one replay launch
        self.last_populate_lean = False
        self._status_hold = False
        self.timing = None  # NESSAI_AMD_TIMING=1: seconds per phase of the last populate() (FlowProposal._mark)
        self._t_last = 0.0
        #: which loop the last ``populate`` ran: "trains" (device-side loop control), "trains_sharded",
        #: "accumulate" (device arena), "accumulate_host" or "passes" (one synchronisation per pass)
        self.last_populate_path = None
        #: what the last sharded populate did: trains, passes, collectives (bench.py reports it for N > 1)
        self.sharded_stats = None
        if precision not in ("auto", "f32", "bf16x3", "f16x3"):
            raise ValueError("precision must be 'auto', 'f32', 'bf16x3' or 'f16x3'")
        self.precision = precision
        self.pass_trains = bool(pass_trains)
        self.reparameterisations = reparameterisations
        self.fallback_reparameterisation = fallback_reparameterisation
        self.device_seed = (
            int(device_seed) if device_seed is not None else int(self.rng.integers(0, 2**63 - 1))
        )

        self.flow = None
        self.initialised = False
        self.populated = False
        self.populating = False
        self.indices = []
        self.samples = None
        self.x = None
        self.training_data = None
        self.training_count = 0
        self.populated_count = 0
        self.population_acceptance = None
        self.population_time = datetime.timedelta()
        self.acceptance = []
        self.ns_acceptance = 1.0
        self.r = np.nan
        self._checked_population = True
        self._draw_counter = 0  # rows drawn so far: Philox row offset
        self._buffers = None
        self.n_proposed = 0
        self.n_accepted = 0

    # -- configuration ---------------------------------------------------------

    @property
    def poolsize(self):
        return int(self._poolsize_scale * self._poolsize)

    # The columns the device works on.  Without a reparameterisation that changes their number these are the model's
    # parameters; an angle (reparameterisations/angle.py) adds the column of its auxiliary radius right behind it -
    # the flow then has `dims` = D' inputs, the draw kernel writes D' output columns, and `x` carries the auxiliary
    # parameters as the reference's x does (its dtype is built from the reparameterisations' parameters).
    @property
    def names(self):
        table = getattr(self, "_reparam", None)
        return list(table.device_names) if table is not None else list(self.model.names)

    @property
    def dims(self):
        return len(self.names)

    @property
    def prime_parameters(self):
        table = getattr(self, "_reparam", None)
        return list(table.prime_names) if table is not None else [n + "_prime" for n in self.model.names]

    @property
    def prime_dims(self):
        return len(self.prime_parameters)

    # the latent-radius rule's settings under their historical names
    def _latent_rule(self):
        return self.truncation.get_rule("latent_radius")

    def _latent_get(self, name, default):
        rule = self._latent_rule()
        return default if rule is None else getattr(rule, name)

    def _latent_set(self, name, value):
        rule = self._latent_rule()
        if rule is None:
            raise AttributeError(f"{name}: the proposal has no latent-radius truncation rule")
        setattr(rule, name, value)

    radius_mode = property(lambda self: self._latent_get("radius_mode", None),
                           lambda self, v: self._latent_set("radius_mode", v))
    fixed_radius = property(lambda self: self._latent_get("fixed_radius", False),
                            lambda self, v: self._latent_set("fixed_radius", v))
    min_radius = property(lambda self: self._latent_get("min_radius", False),
                          lambda self, v: self._latent_set("min_radius", v))
    max_radius = property(lambda self: self._latent_get("max_radius", False),
                          lambda self, v: self._latent_set("max_radius", v))
    fuzz = property(lambda self: self._latent_get("fuzz", 1.0), lambda self, v: self._latent_set("fuzz", v))
    volume_fraction = property(lambda self: self._latent_get("volume_fraction", None))

    @property
    def x_dtype(self):
        return get_dtype(self.names)

    def update_poolsize_scale(self, acceptance):
        """Pool-size multiplier 1 / acceptance, kept inside [1, ``max_poolsize_scale``]; the upper
        limit when nothing was accepted (``proposal/flowproposal/base.py:416-435``)."""
        wanted = 1.0 / acceptance if acceptance else self.max_poolsize_scale
        self._poolsize_scale = max(min(wanted, self.max_poolsize_scale), 1.0)

    def initialise(self, resumed=False):
        """Set up the rescaling and the flow (``base.py:358-391``, ``:578-653``).

        ``resumed``: called from ``resume`` on an unpickled proposal - the reparameterisations
        (with their fitted constants) and the truncation rules are kept as they were."""
        os.makedirs(self.output, exist_ok=True)
        if not resumed or getattr(self, "_reparam", None) is None:
            config = self.reparameterisations
            if getattr(self.model, "reparameterisations", None) is not None:  # base.py:639-640
                config = self.model.reparameterisations
            self._reparam = ReparameterisationTable(
                list(self.model.names), self.prior_bounds, config, fallback=self.fallback_reparameterisation,
                rng=self.rng, reverse_order=self.reverse_reparameterisations,
            )
            if self.use_default_reparameterisations:  # base.py:602-603
                self.add_default_reparameterisations()
            self.truncation.configure(self)  # truncation.py:292-308 (needs the dimensions)
        config = dict(self.flow_config)
        config["n_inputs"] = self.dims
        self.flow = self._FlowModelClass(
            flow_config=config, training_config=self.training_config, output=self.output,
            rng=self.rng,
        )
        self.flow.initialise()
        self.device = self.flow.device
        # the proposal's precision policy applies to every call it makes through the flow - the forward pass of
        # the adaptive radius, forward_pass / backward_pass - from the start, not from the first populate on
        # (a resumed proposal must reproduce the numbers of the one that was pickled)
        self.flow.model.precision = self.precision
        mode = self._prior_mode
        has_device_prior = getattr(self.model, "uniform_prior", False) or (
            getattr(self.model, "device_prior", None) is not None
        )
        if self.map_to_unit_hypercube:
            # In the unit hypercube the prior is the model's log_prior_unit_hypercube (base.py:1053-1067, 1089-1090):
            # 0 in [0, 1)^D and -inf outside unless the model says otherwise (model.py:593-601) - that default is the
            # "uniform box" of the draw kernel (bounds [0, 1), constant 0); a model with a cube prior of its own is
            # evaluated on the host.
            from .model import Model as _OwnModel

            has_device_prior = (
                getattr(type(self.model), "batch_evaluate_log_prior_unit_hypercube", None)
                is _OwnModel.batch_evaluate_log_prior_unit_hypercube
            )
        if mode == "auto":
            mode = "device" if has_device_prior else "host"
        if mode == "device" and not has_device_prior:
            raise ValueError("prior='device' needs model.uniform_prior or model.device_prior")
        self.prior_mode = mode
        self._prior_kind_dev = self._prior_par_dev = None
        aux = self._reparam.has_aux
        if aux and self.map_to_unit_hypercube:
            raise NotImplementedError("angle reparameterisations together with map_to_unit_hypercube")
        if mode == "device" and not self.map_to_unit_hypercube and (
                aux or not getattr(self.model, "uniform_prior", False)):
            # per device column: the model's separable term (a uniform box: -log(width) each), and for the auxiliary
            # radius of an angle its chi(2) density - the reparameterisation's own prior, which the reference adds
            # to the model's (base.py:1046-1049; angle.py:181-183)
            if getattr(self.model, "uniform_prior", False):
                kinds = np.zeros(len(self.model.names), dtype=np.int32)
                pars = np.array([[0.0, 0.0, -math.log(self.model.bounds[n][1] - self.model.bounds[n][0])]
                                 for n in self.model.names], dtype=np.float64)
            else:
                kinds, pars = self.model.device_prior_tables()
            if aux:
                col = {n: i for i, n in enumerate(self.model.names)}
                k2, p2 = [], []
                for n in self.names:
                    if n in col:
                        k2.append(kinds[col[n]])
                        p2.append(pars[col[n]])
                    else:  # chi with k degrees of freedom: (k - 1) log r - r^2 / 2 - (k / 2 - 1) log 2 - lgamma(k / 2)
                        k = self._reparam.aux_dof[n]
                        k2.append(5)
                        p2.append([k - 1.0, 0.0, -(0.5 * k - 1.0) * math.log(2.0) - math.lgamma(0.5 * k)])
                kinds, pars = np.asarray(k2, dtype=np.int32), np.asarray(p2, dtype=np.float64)
            self._prior_kind_dev = torch.from_numpy(np.ascontiguousarray(kinds)).to(self.device)
            self._prior_par_dev = torch.from_numpy(np.ascontiguousarray(pars)).to(self.device)
        self._upload_constants()
        spec = getattr(self.model, "device_likelihood", None)
        if self._likelihood_mode == "device" and spec is None:
            raise ValueError("likelihood='device' needs model.device_likelihood")
        self._ll_spec = None
        if self.map_to_unit_hypercube or aux:
            # the rows of a population are points of the unit hypercube until convert_to_samples maps them back
            # (a user map, on the host) / carry auxiliary columns: the device likelihood would see the wrong coordinates
            if self._likelihood_mode == "device":
                raise ValueError("likelihood='device' is not available with map_to_unit_hypercube=True or with "
                                 "reparameterisations that add auxiliary parameters (angles)")
            spec = None
        if spec is not None and self._likelihood_mode != "host" and 2 <= self.dims <= 129:
            f64 = dict(dtype=torch.float64, device=self.device)
            if spec[0] == "rosenbrock":
                self._ll_spec = (0, None, None, 0.0)
            elif spec[0] == "gaussian":
                mu = None if spec[1] is None else torch.tensor(np.asarray(spec[1], dtype=np.float64), **f64)
                sg = None if spec[2] is None else torch.tensor(np.asarray(spec[2], dtype=np.float64), **f64)
                self._ll_spec = (1, mu, sg, float(spec[3]))
            else:
                raise ValueError(f"Unknown device likelihood {spec[0]!r}")
        self.populated = False
        self.initialised = True
        # One-time device set-up, here rather than inside the first populate() of a run: the flow's kernel
        # images (host pack), the maps for rebuilding them on the device after every later fit, and a first
        # run of the precision probe's kernels.  A sampler then pays per training only the device-side
        # rebuild + probe (~1.5 ms for config 3; profiles/r04_ns_run_c3.json).
        # (the flag lives on the proposal: FlowModel.reset_model(weights=True, permutations=True) builds a NEW
        # flow object, and _prepare_flow hands the flag - like the precision - to whatever flow is current)
        self._repack_eager = True
        try:
            with torch.no_grad():
                self._prepare_flow()
        except RuntimeError as exc:
            if torch.cuda.is_available():
                raise
            raise RuntimeError(
                "FlowProposal.initialise() sets up the flow's kernel images on the GPU and found no ROCm device "
                "(there is no CPU fallback; a proposal can be constructed and pickled without a device, not "
                "initialised)") from exc

    # -- reparameterisation (nessai_amd.reparam) -----------------------------------

    @property
    def scale(self):
        """sigma of the scale-and-shift (z-score) parameters."""
        return {p: g.scale[p] for g in self._reparam.groups if g.kind == "scale_shift" for p in g.parameters}

    @property
    def shift(self):
        """mu of the scale-and-shift (z-score) parameters."""
        return {p: (g.shift[p] if g.shift else 0.0) for g in self._reparam.groups if g.kind == "scale_shift"
                for p in g.parameters}

    @property
    def prior_bounds(self):
        """The bounds the reparameterisations and the bounds test work in (``base.py:277-292``): the model's, or the
        unit hypercube's with ``map_to_unit_hypercube``."""
        if self._prior_bounds is None:
            if self.map_to_unit_hypercube:
                self._prior_bounds = {n: np.array([0.0, 1.0]) for n in self.model.names}
            else:
                self._prior_bounds = getattr(self.model, "bounds", None)
        return self._prior_bounds

    def add_default_reparameterisations(self):
        """Hook of ``use_default_reparameterisations`` (``base.py:437-439``: none by default)."""
        logger.debug("No default reparameterisations")

    def _to_cube(self, x):
        """``model.to_unit_hypercube`` when the proposal works in the unit hypercube (``base.py:744-745, 796-797``)."""
        return self.model.to_unit_hypercube(x) if self.map_to_unit_hypercube else x

    def _from_cube(self, x):
        """... and back (``base.py:781-782, 1122-1123``)."""
        return self.model.from_unit_hypercube(x) if self.map_to_unit_hypercube else x

    def unit_hypercube_log_prior(self, x):
        """Log-prior of points of the unit hypercube (``base.py:1053-1067``; the one-to-one reparameterisations of this
        package carry no prior of their own)."""
        return np.asarray(self.model.batch_evaluate_log_prior_unit_hypercube(x), dtype=np.float64)

    def check_state(self, x):
        """Update the data-dependent constants from the live points (``base.py:850-868``):
        z-score mean / std (``rescale.py:293-304``), rescale-to-bounds limits (``:720-741``)."""
        if self._reparam.should_update:
            self._reparam.update(self._to_cube(x))
        self._upload_constants()

    def set_rescaling_constants(self, shift, scale):
        """Set mu / sigma of the z-score parameters directly (arrays over ``model.names``)."""
        col = {n: i for i, n in enumerate(self.model.names)}
        for g in self._reparam.groups:
            if g.kind != "scale_shift":
                continue
            if g.shift is None:
                g.shift = {}
            for p in g.parameters:
                g.shift[p] = float(shift[col[p]])
                g.scale[p] = float(scale[col[p]])
        self._upload_constants()

    @torch.inference_mode(False)
    def _upload_constants(self):
        dev = self.device
        f64 = dict(dtype=torch.float64, device=dev)
        kind, par, scale, shift, log_j = self._reparam.device_tables()
        self._scale_dev = torch.tensor(scale, **f64)
        self._shift_dev = torch.tensor(shift, **f64)
        if self.map_to_unit_hypercube:
            # the cube prior's support [0, 1) (model.py:593-601: -inf for x < 0 or x >= 1): the kernel's test is
            # lower <= x <= upper, so the upper bound is the largest float64 below 1
            self._lower_dev = torch.zeros(self.dims, **f64)
            self._upper_dev = torch.full((self.dims,), float(np.nextafter(1.0, 0.0)), **f64)
        else:
            lo, hi = self._reparam.device_bounds(self.model.lower_bounds, self.model.upper_bounds)
            self._lower_dev = torch.tensor(lo, **f64)
            self._upper_dev = torch.tensor(hi, **f64)
        self._log_j_rescale = log_j
        if self._reparam.all_scale_shift:
            self._rp_kind_dev = self._rp_par_dev = None
        else:
            self._rp_kind_dev = torch.from_numpy(kind).to(dev)
            self._rp_par_dev = torch.from_numpy(par).to(dev)

    def rescale(self, x, compute_radius=False, test=None, **kwargs):
        """x -> (x', log|J|)  (``base.py:716-753``).  ``compute_radius`` (the reference's flag for the radius
        of the latent contour) makes every boundary inversion duplicate its rows; a boundary inversion that
        has just detected its edges changes the draw kernel's per-column table."""
        known = self._reparam.has_inversion and any(
            e is None for g in self._reparam.groups if g.kind == "to_bounds" and g.inversion for e in g.edges.values())
        out = self._reparam.rescale(self._to_cube(x), compute_radius=compute_radius, test=test)
        if known:  # edges were open before this call: they are decided now
            self._upload_constants()
        return out

    def inverse_rescale(self, x_prime, return_unit_hypercube=False):
        """x' -> (x, log|J|)  (``base.py:755-784``); with ``map_to_unit_hypercube`` the points are mapped on to the
        physical space unless ``return_unit_hypercube``."""
        x, log_j = self._reparam.inverse_rescale(x_prime)
        if not return_unit_hypercube:
            x = self._from_cube(x)
        return x, log_j

    # -- training ----------------------------------------------------------------

    def train(self, x, plot=False):
        """Fit the flow to the live points (``base.py:870-925``)."""
        if not self.initialised:
            raise RuntimeError("FlowProposal is not initialised.")
        self.training_data = x.copy()
        self.check_state(self.training_data)
        x_prime, _ = self.rescale(x)
        x_prime_array = live_points_to_array(x_prime, self.prime_parameters)
        rank, world = self._rank_world()
        # every rank writes its own weights file (no race on a shared output directory)
        output = self.output if rank == 0 else os.path.join(self.output, f"rank_{rank}")
        if self.save_training_data:  # base.py:887-900: one directory per training, the live points as JSON
            output = os.path.join(output, "training", f"block_{self.training_count}", "")
            os.makedirs(output, exist_ok=True)
            save_live_points(x, os.path.join(output, "training_data.json"))
        fit_error = None
        if rank == 0 or self.replicate_training:
            try:
                self.flow.train(x_prime_array, output=output, plot=False)
            except Exception as exc:  # (re-raised below, on every rank of a sharded run)
                if world == 1:
                    raise
                fit_error = exc
        else:
            # Training does not shard and N identical fits are N times the energy for
            # one flow: rank 0 fits, the others wait for its weights, buffers and generator states
            # (_replicate_flow) - they only need to stand where a fit would have left them.
            if not np.isfinite(x_prime_array).all():
                raise ValueError("Training data is not finite")
            os.makedirs(output, exist_ok=True)
            self.flow.weights_file = os.path.join(output, "model.pt")
            self.flow.model.eval()
        if world > 1:
            # A fit that failed on one rank (a HIP error, an I/O error in save_weights ...) must not leave the
            # others waiting in the collectives of _replicate_flow until the process group times out: the ranks
            # agree on success first, and every rank raises.  With replicate_training=False only rank 0 can
            # fail here; its message travels with the flag.
            import torch.distributed as dist

            report = [None if fit_error is None else f"rank {rank}: {type(fit_error).__name__}: {fit_error}"]
            gathered = [None] * world
            dist.all_gather_object(gathered, report[0])
            failed = [g for g in gathered if g is not None]
            if failed:
                if fit_error is not None:
                    raise fit_error
                raise RuntimeError("sharded FlowProposal.train: the fit failed on another rank (" + "; ".join(failed) + ")")
            self._replicate_flow()
        self.populated = False
        self.training_count += 1

    def _rank_world(self):
        """(rank, world size) of a sharded run; (0, 1) otherwise."""
        if not self.shard:
            return 0, 1
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()):
            return 0, 1
        return dist.get_rank(), dist.get_world_size()

    def _replicate_flow(self):
        """Make rank 0's flow THE flow of a sharded run (training does not
        shard - replicas only).

        Rank 0 alone runs the fit (``replicate_training=True``: every rank runs it, the round-3
        behaviour - the fused trainer sums weight gradients with float atomics, so even identical
        fits differ in the last bits from rank to rank, and a populate sharded over different flows
        would mix draws from different proposals).  Rank 0's parameters and buffers (batch-norm
        statistics, permutations) replace everyone's, and so do the states of the generators a fit
        advances - the proposal's numpy generator (validation split, epoch bookkeeping) and torch's
        global CPU generator (epoch shuffles, weight resets) - so that the replicated samplers stay in
        lock step; the rescaling constants, which are exact functions of the live points, are checked
        to agree."""
        import torch.distributed as dist

        model = self.flow.model
        with torch.no_grad():
            trainer = model.trainer()
            dist.broadcast(trainer.flat, src=0)  # every Parameter and running statistic is a view of it
            for name, buf in model.named_buffers():
                if buf.numel() and buf.data_ptr() not in trainer.flat_range():
                    dist.broadcast(buf, src=0)
            for what, t in (("scale", self._scale_dev), ("shift", self._shift_dev)):
                ref = t.clone()
                dist.broadcast(ref, src=0)
                if not torch.equal(ref, t):
                    raise RuntimeError(
                        f"sharded FlowProposal: the rescaling {what} differs between ranks - "
                        "the replicated samplers were not given the same live points"
                    )
        # early stopping may have ended the fits at different epochs: the host generator (epoch
        # shuffles) continues from rank 0's state everywhere
        state = [self.rng.bit_generator.state, torch.get_rng_state()]
        dist.broadcast_object_list(state, src=0)
        self.rng.bit_generator.state = state[0]
        torch.set_rng_state(state[1])
        model._image_key = None  # repack the kernel image from the received weights
        if self.flow.weights_file is not None:
            self.flow.save_weights(self.flow.weights_file)

    # -- passes through the flow ---------------------------------------------------

    def forward_pass(self, x, rescale=True, **kwargs):
        """Live points -> (z, log_prob incl. rescaling Jacobian) (``base.py:961-994``).

        Extra keyword arguments go to ``rescale`` as in the reference (``compute_radius=True``: boundary
        inversions duplicate their rows, so there are more latents than points)."""
        log_J = 0
        if rescale:
            x, log_J = self.rescale(x, **kwargs)
        arr = live_points_to_array(x, self.prime_parameters)
        z, log_prob = self.flow.forward_and_log_prob(arr)
        return z, log_prob + log_J

    def backward_pass(self, z, return_z=False, return_unit_hypercube=False):
        """Latents -> (x, log_q[, z]) for the kept rows (``flowproposal.py:345-389``).

        Array-level entry point of the fused draw kernel with injected latents.  With ``map_to_unit_hypercube`` the
        kernel's bounds test is the unit hypercube's (the image of the prior's support) and the kept rows are mapped
        on to the physical space unless ``return_unit_hypercube``.
        """
        z = np.ascontiguousarray(z, dtype=np.float64)
        n = z.shape[0]
        dev = self.device
        z_dev = torch.from_numpy(z).to(torch.float32).to(dev)
        x64 = torch.empty((n, self.dims), dtype=torch.float64, device=dev)
        log_q = torch.empty(n, dtype=torch.float64, device=dev)
        self._launch_draw(n, z_dev, 0, np.inf, x64, log_q, None)
        if n and self._f16_range_left():  # split-f16 met a value beyond its range: again in float32
            self.flow.model.note_f16_overflow()
            self._launch_draw(n, z_dev, 0, np.inf, x64, log_q, None)
        log_q_h = log_q.cpu().numpy()
        keep = ~np.isnan(log_q_h)  # NaN marks a dropped row; +inf (saturated logit) is kept as in the reference
        x = array_to_live_points_fast(x64.cpu().numpy()[keep], self.names)
        if not return_unit_hypercube:
            x = self._from_cube(x)
        if return_z:
            return x, log_q_h[keep], z[keep]
        return x, log_q_h[keep]

    def log_prior(self, x):
        """The model's log-prior plus the reparameterisations' own (``base.py:1032-1051``: the auxiliary radius of an
        angle)."""
        log_p = self.model.batch_evaluate_log_prior(x)
        if getattr(self, "_reparam", None) is not None and self._reparam.has_aux:
            log_p = log_p + self._reparam.log_prior(x)
        return log_p

    def _population_log_prior(self, x):
        """Log-prior of candidate rows in the space the population lives in (``base.py:1089-1092``)."""
        return self.unit_hypercube_log_prior(x) if self.map_to_unit_hypercube else self.log_prior(x)

    def compute_weights(self, x, log_q, return_log_prior=False):
        """``log_prior(x) - log_q`` (``base.py:1069-1098``): the unit-hypercube prior for the rows of a population
        kept in the unit hypercube."""
        log_p = self._population_log_prior(x)
        log_w = log_p - log_q
        return (log_w, log_p) if return_log_prior else log_w

    def convert_to_samples(self, x, plot=False):
        """Map back from the unit hypercube if the population lives there, attach ``logP`` (``base.py:1100-1128``)."""
        x = copy_points(x)  # (one memcpy of the records; numpy copies a structured array field by field)
        x = self._from_cube(x)
        if getattr(self, "_reparam", None) is not None and self._reparam.has_aux:
            # auxiliary parameters stay behind (base.py:1124-1126: x[model.names + non-sampling parameters])
            kept = np.empty(x.size, dtype=get_dtype(list(self.model.names)))
            for f in kept.dtype.names:
                kept[f] = x[f]
            x = kept
        x["logP"] = self.model.batch_evaluate_log_prior(x)
        return x

    # -- device plumbing --------------------------------------------------------------

    def _prepare_flow(self):
        """(flow, header, image, status) for the draw kernel: eval mode, precision policy and the
        packed device image.  Costs ~0.3 ms of host time (it walks the module tree), so
        ``populate`` does it once per call, not once per pass."""
        flow = self.flow.model
        if flow.training:
            flow.eval()
        if getattr(self, "_repack_eager", False) and not flow.__dict__.get("_repack_eager"):
            flow.__dict__["_repack_eager"] = True  # (a flow rebuilt by reset_model: see initialise)
        # "auto": the flow resolves it per set of weights - split-bf16 inside the static envelope AND
        # measured to agree with float32 on a probe (HipFlow._pack_auto)
        if flow.precision != self.precision:
            flow.precision = self.precision
        header, image = flow.device_image()
        return flow, header, image, flow._status

    def _launch_draw(self, n, z_in, row_offset, radius, x64, log_q, z_out, log_prior=None, prepared=None):
        flow, header, image, status = prepared if prepared is not None else self._prepare_flow()
        if not self._status_hold:  # inside populate() the flags accumulate over the passes
            status.zero_()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = lib.load().nsa_populate_draw(
            header, lib.ptr(image), n, lib.ptr(z_in), self.device_seed, int(row_offset),
            float(radius), float(self.latent_temperature or 1.0), lib.ptr(self._scale_dev),
            lib.ptr(self._shift_dev), lib.ptr(self._lower_dev), lib.ptr(self._upper_dev),
            float(self._log_j_rescale), lib.ptr(x64), lib.ptr(log_q), lib.ptr(z_out),
            lib.ptr(self._prior_kind_dev if log_prior is not None else None),
            lib.ptr(self._prior_par_dev if log_prior is not None else None), lib.ptr(log_prior),
            lib.ptr(self._rp_kind_dev), lib.ptr(self._rp_par_dev), lib.ptr(status), flow.waves_per_group,
            flow.max_groups, lib.c_void_p(stream),
        )
        lib.check(rc, "nsa_populate_draw")

    def _launch_draw_weights(self, n, row_offset, radius, log_w, max_key, log_prior=None, min_log_q=-math.inf,
                             prepared=None):
        """The head of one lean pass (``nsa_populate_draw_weights``): ``log_w = log_prior - log_q`` of ``n`` drawn
        rows and the launch's largest weight, nothing else written."""
        flow, header, image, status = prepared if prepared is not None else self._prepare_flow()
        if not self._status_hold:
            status.zero_()
        separable = self._prior_kind_dev is not None
        stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = lib.load().nsa_populate_draw_weights(
            header, lib.ptr(image), n, self.device_seed, int(row_offset), float(radius),
            float(self.latent_temperature or 1.0), lib.ptr(self._scale_dev), lib.ptr(self._shift_dev),
            lib.ptr(self._lower_dev), lib.ptr(self._upper_dev), float(self._log_j_rescale),
            lib.ptr(self._prior_kind_dev), lib.ptr(self._prior_par_dev), lib.ptr(log_prior if separable else None),
            0.0 if separable else self._uniform_log_prior_constant(), float(min_log_q),
            lib.ptr(self._rp_kind_dev), lib.ptr(self._rp_par_dev), lib.ptr(log_w), lib.ptr(max_key), lib.ptr(status),
            flow.waves_per_group, flow.max_groups, lib.c_void_p(stream),
        )
        lib.check(rc, "nsa_populate_draw_weights")

    def _device_log_likelihood(self, rows):
        """log L of device rows ``(m, D)`` float64 -> ``(m,)`` float64 device tensor."""
        kind, mu, sigma, const = self._ll_spec
        out = torch.empty(rows.shape[0], dtype=torch.float64, device=rows.device)
        stream = lib.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        rc = lib.load().nsa_log_likelihood(kind, rows.shape[0], self.dims, lib.ptr(rows), lib.ptr(mu),
                                           lib.ptr(sigma), const, lib.ptr(out), stream)
        lib.check(rc, "nsa_log_likelihood")
        return out

    @torch.inference_mode(False)  # cached across calls: never inference tensors
    def _get_buffers(self, n, capacity):
        key = (n, capacity, self.dims)
        if self._buffers is not None and self._buffers["key"] == key:
            return self._buffers
        dev = self.device
        L = lib.load()
        ws_bytes = int(L.nsa_populate_accept_workspace(n))
        # log_q / log_p also receive the per-row outputs of the lean replay, which writes one element per
        # replayed row: up to ``capacity`` (= n_samples) of them, which exceeds a pass when poolsize > drawsize
        # or when a sharded rank accepts more than its share drawsize / world of the batch
        n_rows = max(n, capacity)
        self._buffers = dict(
            key=key,
            x64=torch.empty((n, self.dims), dtype=torch.float64, device=dev),
            log_q=torch.empty(n_rows, dtype=torch.float64, device=dev),
            log_w=torch.empty(n, dtype=torch.float64, device=dev),
            max_key=torch.empty(1, dtype=torch.int64, device=dev),
            log_p=torch.empty(n_rows, dtype=torch.float64, device=dev),
            x_acc=torch.empty((capacity, self.dims), dtype=torch.float64, device=dev),
            n_acc=torch.zeros(1, dtype=torch.int64, device=dev),
            workspace=torch.empty(ws_bytes, dtype=torch.uint8, device=dev),
            ws_bytes=ws_bytes,
        )
        return self._buffers

    def _weights_and_max(self, buf, n, reset_max=True):
        L = lib.load()
        stream = lib.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        if reset_max:
            buf["max_key"].fill_(int(L.nsa_max_key_init()))
        if self.prior_mode == "device" and self._prior_kind_dev is not None:
            # separable analytic prior, evaluated by the draw kernel into buf["log_p"]
            rc = L.nsa_populate_weights(
                n, lib.ptr(buf["log_q"]), lib.ptr(buf["log_p"]), 0.0, lib.ptr(buf["log_w"]),
                lib.ptr(buf["max_key"]), stream,
            )
        elif self.prior_mode == "device":
            log_prior_const = self._uniform_log_prior_constant()
            rc = L.nsa_populate_weights(
                n, lib.ptr(buf["log_q"]), None, log_prior_const, lib.ptr(buf["log_w"]),
                lib.ptr(buf["max_key"]), stream,
            )
        else:
            # general prior: evaluate the user's numpy log_prior on the host
            x_host = array_to_live_points_fast(buf["x64"][:n].cpu().numpy(), self.names)
            with np.errstate(divide="ignore", invalid="ignore"):
                log_p = np.asarray(self._population_log_prior(x_host), dtype=np.float64)
            log_p_dev = torch.from_numpy(log_p).to(self.device)
            rc = L.nsa_populate_weights(
                n, lib.ptr(buf["log_q"]), lib.ptr(log_p_dev), 0.0, lib.ptr(buf["log_w"]),
                lib.ptr(buf["max_key"]), stream,
            )
        lib.check(rc, "nsa_populate_weights")

    def _uniform_log_prior_constant(self):
        """-sum log(width), subtracted bound by bound like ``examples/rosenbrock.py:28-31``."""
        if self.map_to_unit_hypercube:
            return 0.0  # log 1: the unit hypercube's own prior (model.py:593-601)
        log_p = np.float64(0.0)
        for n in self.model.names:
            lo, hi = self.model.bounds[n]
            log_p -= np.log(hi - lo)
        return float(log_p)

    def _accept(self, buf, n, capacity, row_offset, u_in=None, log_w=None, x64=None):
        L = lib.load()
        stream = lib.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        rc = L.nsa_populate_accept(
            n, self.dims, lib.ptr(buf["log_w"] if log_w is None else log_w), lib.ptr(buf["max_key"]),
            self.device_seed, int(row_offset), lib.ptr(u_in),
            lib.ptr(buf["x64"] if x64 is None else x64), capacity, lib.ptr(buf["x_acc"]), None,
            lib.ptr(buf["n_acc"]), lib.ptr(buf["workspace"]), buf["ws_bytes"], stream,
        )
        lib.check(rc, "nsa_populate_accept")

    # -- population ----------------------------------------------------------------

    def compute_population_radius(self, worst_point, r=None):
        """Radius of the latent-radius rule for one population (``truncation.py:323-356``)."""
        rule = self._latent_rule()
        if rule is None:
            return math.nan if r is None else float(r)
        return rule._population_radius(self, worst_point, r)

    def _truncate_after_flow(self, buf, n, min_log_q, log_l_threshold):
        """The rules that act after the flow, as one device predicate on ``log_q``
        (``nsa_populate_truncate``): ``log_q > min_log_q`` and ``logL > threshold``.  The
        likelihood of every candidate comes from the device likelihood when the model declares
        one, else from ``model.batch_evaluate_log_likelihood`` on the host (the rule is expensive
        by construction: the reference evaluates every candidate too, ``flowproposal.py:455-466``)."""
        if not (min_log_q > -math.inf or log_l_threshold is not None):
            return
        log_l_dev = None
        if log_l_threshold is not None:
            if self._ll_spec is not None:
                log_l_dev = self._device_log_likelihood(buf["x64"][:n])
            else:
                alive = torch.isfinite(buf["log_q"][:n])
                idx = torch.nonzero(alive).reshape(-1)
                pts = self._from_cube(array_to_live_points_fast(buf["x64"][:n][idx].cpu().numpy(), self.names))
                values = np.asarray(self.model.batch_evaluate_log_likelihood(pts), dtype=np.float64)
                log_l_dev = torch.full((n,), -math.inf, dtype=torch.float64, device=self.device)
                log_l_dev[idx] = torch.from_numpy(values).to(self.device)
        stream = lib.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        rc = lib.load().nsa_populate_truncate(
            n, lib.ptr(buf["log_q"]), float(min_log_q), lib.ptr(log_l_dev),
            float(log_l_threshold) if log_l_threshold is not None else 0.0, stream,
        )
        lib.check(rc, "nsa_populate_truncate")

    def _device_rules(self):
        """``nsa_post_rules`` of the current population (``None``: no rule acts after the flow): the
        ``min_log_q`` / likelihood-threshold numbers ``truncation.prepare`` computed, with the model's
        built-in device likelihood."""
        min_log_q, log_l_threshold = self._post_limits
        if not (min_log_q > -math.inf or log_l_threshold is not None):
            return None
        rules = lib.PostRules(float(min_log_q), -1, None, None, 0.0, 0.0)
        if log_l_threshold is not None:
            kind, mu, sigma, const = self._ll_spec
            rules.ll_kind = int(kind)
            rules.ll_mu_dev = mu.data_ptr() if mu is not None else None
            rules.ll_sigma_dev = sigma.data_ptr() if sigma is not None else None
            rules.ll_const = float(const)
            rules.log_l_threshold = float(log_l_threshold)
        return rules

    def _lean_ok(self, n, rules):
        """Lean passes (``lean=True``) apply to big passes of the device-side loops whose rules after the flow do
        not need the candidate rows (``min_log_q`` alone is a predicate on ``log_q``)."""
        if not self.lean or os.environ.get("NESSAI_AMD_LEAN") == "0":
            return False
        if n <= 16384:  # small passes: the fused small-train kernels keep their staging blocks
            return False
        return rules is None or rules.ll_kind < 0

    def _replay(self, idx, m, radius, x64, log_q, log_prior, prepared):
        """``x64[:m]`` = the rows whose global Philox rows are ``idx[:m]`` (``nsa_populate_replay``): the accepted
        rows of lean passes, computed again by the kernel instance that accepted them."""
        if m <= 0:
            return
        flow, header, image, status = prepared
        stream = lib.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        rc = lib.load().nsa_populate_replay(
            header, lib.ptr(image), int(m), lib.ptr(idx), self.device_seed, float(radius),
            float(self.latent_temperature or 1.0), lib.ptr(self._scale_dev), lib.ptr(self._shift_dev),
            lib.ptr(self._lower_dev), lib.ptr(self._upper_dev), float(self._log_j_rescale), lib.ptr(x64),
            lib.ptr(log_q), lib.ptr(self._prior_kind_dev if log_prior is not None else None),
            lib.ptr(self._prior_par_dev if log_prior is not None else None), lib.ptr(log_prior),
            lib.ptr(self._rp_kind_dev), lib.ptr(self._rp_par_dev), lib.ptr(status), flow.waves_per_group,
            flow.max_groups, stream,
        )
        lib.check(rc, "nsa_populate_replay")

    def _f16_range_left(self, flag_host=None):
        """True when the draw kernel ran in split-f16 and reported a value beyond the f16 range
        (``status[1]``; agreed over the ranks of a sharded proposal).  One 16-byte read-back, or none when
        the caller already brought the flag to the host with its own copies (``flag_host``)."""
        flow = self.flow.model
        if flow.effective_precision != "f16x3" or flow._status is None:
            return False
        sharded = self.shard and torch.distributed.is_available() and torch.distributed.is_initialized()
        if flag_host is not None and not sharded:
            return int(flag_host) != 0
        flag = flow._status[1:2].clone()
        if sharded:
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
        return int(flag.item()) != 0

    def populate(self, worst_point=None, n_samples=10000, plot=False, r=None,
                 max_samples=1_000_000, injected=None):
        """Fill the pool with ``n_samples`` accepted draws (``flowproposal.py:391-534``).

        ``injected`` (parity tests) is an iterator of ``(z, u)`` host arrays used
        in place of the on-device draws, the analogue of the reference test's
        mocked ``rng.random`` / latent draws.

        The draw kernel's split-f16 mode is exact-float32 class only while every operand of its
        matrix products stays inside the f16 range; a population that left it (reported by the
        kernel, checked once per call) is discarded and drawn again with the mode switched off for
        the current weights (``precision="auto"``) or raises (``precision="f16x3"``)."""
        if self.initialised:
            if self.flow.model.precision != self.precision:  # (a flow rebuilt by reset_model)
                self.flow.model.precision = self.precision
            if self.flow.model._status is not None:
                self.flow.model._status.zero_()
        self._status_hold = True
        self._f16_discarded = False
        # NESSAI_AMD_TIMING=1: where this call's wall time goes (synchronised phase marks; profiling only)
        self.timing = {} if os.environ.get("NESSAI_AMD_TIMING") == "1" else None
        self._t_last = time.perf_counter()
        try:
            self._populate(worst_point, n_samples, plot, r, max_samples, injected)
            if self._f16_discarded:  # seen by _finish_population BEFORE it touched the pool
                self._f16_discarded = False
                self.flow.model.note_f16_overflow()  # raises F16RangeError unless "auto"
                if injected is not None:
                    raise RuntimeError("injected latents left the f16 range of the split-f16 kernels")
                logger.warning("split-f16 draw kernel left the f16 range: population redrawn in float32")
                self.flow.model._status.zero_()
                self._populate(worst_point, n_samples, plot, r, max_samples, injected)
        finally:
            self._status_hold = False

    def _mark(self, phase):
        """Profiling (``NESSAI_AMD_TIMING=1``): time since the last mark, device drained, added to ``timing[phase]``."""
        if self.timing is None:
            return
        torch.cuda.synchronize(self.device)
        now = time.perf_counter()
        self.timing[phase] = self.timing.get(phase, 0.0) + (now - self._t_last)
        self._t_last = now

    def _populate(self, worst_point, n_samples, plot, r, max_samples, injected):
        st = datetime.datetime.now()
        if not self.initialised:
            raise RuntimeError(
                "Proposal has not been initialised. Try calling `initialise()` first."
            )
        # every rule computes its numbers for this population (truncation.py:483-486); the latent
        # radius is compared inside the draw kernel, the other two by nsa_populate_truncate
        self.truncation.prepare(self, worst_point, radius=r)
        threshold, min_log_q, log_l_threshold = self.truncation.device_limits()
        latent = self._latent_rule()
        self.r = latent.radius if latent is not None else np.nan
        self._post_limits = (min_log_q, log_l_threshold)
        self.last_populate_lean = False
        self.indices = []
        n_samples = int(n_samples)

        world, rank = 1, 0
        sharded = self.shard and torch.distributed.is_available() and torch.distributed.is_initialized()
        if sharded:
            world = torch.distributed.get_world_size()
            rank = torch.distributed.get_rank()

        n_local = self.drawsize if injected is None else None
        if world > 1:
            n_local = (self.drawsize + world - 1) // world
        n_proposed = 0
        n_accepted = 0
        rows = []  # accepted blocks (host, float64)
        # the rules that act after the flow stay on the device-side loop unless the likelihood rule needs
        # the user's host likelihood (no built-in device likelihood declared by the model)
        host_rule = log_l_threshold is not None and self._ll_spec is None
        if self.accumulate_weights:
            if injected is None and self.prior_mode == "device" and not host_rule:
                return self._populate_accumulate(st, threshold, n_samples, max_samples, n_local, world, rank)
            if sharded:
                raise RuntimeError("a sharded accumulate_weights population needs the device prior "
                                   "(and, for a likelihood rule, the model's device likelihood)")
            return self._populate_accumulate_host(st, threshold, n_samples, max_samples, injected)

        if injected is None and self.prior_mode == "device" and self.pass_trains and not host_rule:
            if sharded:
                return self._populate_trains_sharded(st, threshold, n_samples, max_samples, n_local, world, rank)
            return self._populate_trains(st, threshold, n_samples, max_samples)

        self.last_populate_path = "passes"
        prepared = self._prepare_flow()
        while n_accepted < n_samples:
            z_in = u_in = None
            if injected is not None:
                z_np, u_np = next(injected)
                n_local = int(len(z_np))
                z_in = torch.from_numpy(np.ascontiguousarray(z_np)).to(torch.float32).to(self.device)
                u_in = torch.from_numpy(np.ascontiguousarray(u_np, dtype=np.float64)).to(self.device)
            buf = self._get_buffers(n_local, n_samples)
            row_offset = self._draw_counter + rank * n_local
            self._launch_draw(n_local, z_in, row_offset, threshold, buf["x64"], buf["log_q"], None,
                              log_prior=buf["log_p"] if self._prior_kind_dev is not None else None,
                              prepared=prepared)
            self._truncate_after_flow(buf, n_local, min_log_q, log_l_threshold)
            self._weights_and_max(buf, n_local)
            if sharded:
                torch.distributed.all_reduce(buf["max_key"], op=torch.distributed.ReduceOp.MAX)
            self._accept(buf, n_local, n_samples, row_offset, u_in=u_in)
            if sharded:
                # counts of all ranks in one collective + one device->host copy (also this rank's)
                n_acc_batch, block = _gather_accepted(buf["x_acc"], buf["n_acc"], n_samples, self.dims)
            else:
                n_acc_batch = int(buf["n_acc"].item())  # the pass's only device->host sync
                block = buf["x_acc"][: min(n_acc_batch, n_samples)]
            n_proposed += n_local * world
            self._draw_counter += n_local * world
            m = min(n_samples - n_accepted, n_acc_batch)
            if m > 0:
                kept = block[:m].contiguous()
                if self._ll_spec is not None:  # log L travels with the rows as one more column
                    kept = torch.cat([kept, self._device_log_likelihood(kept)[:, None]], dim=1)
                rows.append(kept.cpu().numpy())
            n_accepted += n_acc_batch
            if n_proposed > max_samples:
                logger.warning("Reached max samples (%s)", max_samples)
                break

        width = self.dims + (1 if self._ll_spec is not None else 0)
        accepted = np.concatenate(rows, axis=0) if rows else np.empty((0, width))
        accepted = accepted[: min(n_accepted, n_samples)]
        self.x = array_to_live_points_fast(np.ascontiguousarray(accepted[:, : self.dims]), self.names)
        log_l = accepted[:, self.dims].copy() if self._ll_spec is not None else None
        self._finish_population(st, n_accepted, n_proposed, log_l=log_l)

    @torch.inference_mode(False)
    def _populate_trains(self, st, threshold, n_samples, max_samples):
        """The populate loop with its control on the device (``nsa_populate_passes``): trains of
        passes are enqueued without host round trips; a pass whose predecessors already reached
        ``n_samples`` (or the ``max_samples`` exit) does nothing, so rows and counters equal those
        of the one-pass-at-a-time loop.  The train length follows the acceptance seen so far."""
        L = lib.load()
        n = self.drawsize
        self.last_populate_path = "trains"
        self._mark("rules, radius")
        flow, header, image, status = self._prepare_flow()
        self._mark("flow image, precision probe")
        buf = self._get_buffers(n, n_samples)
        self._mark("buffers")
        dev = self.device
        max_train = 64
        if buf.get("state") is None:
            buf["state"] = torch.zeros(8, dtype=torch.int64, device=dev)
            buf["max_keys"] = torch.empty(max_train, dtype=torch.int64, device=dev)
            buf["state_host"] = torch.zeros(8, dtype=torch.int64).pin_memory()
        state_h = buf["state_host"]
        state_h.zero_()
        state_h[2] = n_samples
        state_h[5] = max_samples // n + 1  # first pass count with n_proposed > max_samples
        buf["state"].copy_(state_h, non_blocking=True)
        key_init = int(L.nsa_max_key_init())
        separable = self._prior_kind_dev is not None
        log_prior_const = 0.0 if separable else self._uniform_log_prior_constant()
        stream = lib.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        # expected accepted rows per pass: last populate call's acceptance, else unknown (probe 1 pass)
        per_pass = (self.population_acceptance or 0.0) * n
        n_accepted = passes = 0
        rules = self._device_rules()
        small = n <= 16384 and self.small_trains
        # lean passes: no candidate rows, accepted rows listed by their Philox rows and computed again at the end
        lean = self._lean_ok(n, rules)
        self.last_populate_lean = lean
        if lean and buf.get("idx_acc") is None:
            buf["idx_acc"] = torch.empty(n_samples, dtype=torch.int64, device=dev)
        while True:
            remaining = n_samples - n_accepted
            train = 1 if per_pass <= 0 else int(np.ceil(1.15 * remaining / per_pass))
            train = max(1, min(train, max_train, int(state_h[5]) - passes))
            if lean:
                buf["max_keys"][:train].fill_(key_init)
                rc = L.nsa_populate_passes_lean(
                    header, lib.ptr(image), n, train, self.device_seed, int(self._draw_counter + passes * n),
                    float(threshold), float(self.latent_temperature or 1.0), lib.ptr(self._scale_dev),
                    lib.ptr(self._shift_dev), lib.ptr(self._lower_dev), lib.ptr(self._upper_dev),
                    float(self._log_j_rescale), lib.ptr(buf["log_q"]),
                    lib.ptr(self._prior_kind_dev), lib.ptr(self._prior_par_dev),
                    lib.ptr(buf["log_p"] if separable else None), float(log_prior_const),
                    lib.ptr(self._rp_kind_dev), lib.ptr(self._rp_par_dev), lib.ptr(buf["log_w"]),
                    lib.ptr(buf["max_keys"]), n_samples, lib.ptr(buf["idx_acc"]), lib.ptr(buf["state"]),
                    lib.ptr(buf["workspace"]), buf["ws_bytes"], lib.ptr(status), flow.waves_per_group,
                    flow.max_groups, stream, rules,
                )
                lib.check(rc, "nsa_populate_passes_lean")
            elif small:
                # small passes (the reference's default drawsize = nlive): the whole train as ONE draw launch over
                # train * n rows + two tail launches - a pass of 2000 rows is a ~40 us dependent chain through the
                # flow however few rows it has, and sixty of them in a row were the whole populate call
                if buf.get("small_passes", 0) < train:
                    stage_rows = min(n, n_samples)
                    buf["x64_small"] = torch.empty((train * n, self.dims), dtype=torch.float64, device=dev)
                    buf["log_q_small"] = torch.empty(train * n, dtype=torch.float64, device=dev)
                    buf["log_p_small"] = torch.empty(train * n if separable else 1, dtype=torch.float64, device=dev)
                    buf["stage_small"] = torch.empty((train, stage_rows, self.dims), dtype=torch.float64, device=dev)
                    buf["counts_small"] = torch.zeros(train, dtype=torch.int64, device=dev)
                    buf["small_passes"] = train
                rc = L.nsa_populate_passes_small(
                    header, lib.ptr(image), n, train, self.device_seed, int(self._draw_counter + passes * n),
                    float(threshold), float(self.latent_temperature or 1.0), lib.ptr(self._scale_dev),
                    lib.ptr(self._shift_dev), lib.ptr(self._lower_dev), lib.ptr(self._upper_dev),
                    float(self._log_j_rescale), lib.ptr(buf["x64_small"]), lib.ptr(buf["log_q_small"]),
                    lib.ptr(self._prior_kind_dev), lib.ptr(self._prior_par_dev),
                    lib.ptr(buf["log_p_small"] if separable else None), float(log_prior_const),
                    lib.ptr(self._rp_kind_dev), lib.ptr(self._rp_par_dev), buf["stage_small"].shape[1],
                    lib.ptr(buf["stage_small"]), lib.ptr(buf["counts_small"]), n_samples, lib.ptr(buf["x_acc"]),
                    lib.ptr(buf["state"]), lib.ptr(status), flow.waves_per_group, flow.max_groups, stream, rules,
                )
                lib.check(rc, "nsa_populate_passes_small")
            else:
                buf["max_keys"][:train].fill_(key_init)
                rc = L.nsa_populate_passes(
                    header, lib.ptr(image), n, train, self.device_seed, int(self._draw_counter + passes * n),
                    float(threshold), float(self.latent_temperature or 1.0), lib.ptr(self._scale_dev),
                    lib.ptr(self._shift_dev), lib.ptr(self._lower_dev), lib.ptr(self._upper_dev),
                    float(self._log_j_rescale), lib.ptr(buf["x64"]), lib.ptr(buf["log_q"]),
                    lib.ptr(self._prior_kind_dev), lib.ptr(self._prior_par_dev),
                    lib.ptr(buf["log_p"] if separable else None), float(log_prior_const),
                    lib.ptr(self._rp_kind_dev), lib.ptr(self._rp_par_dev), lib.ptr(buf["log_w"]),
                    lib.ptr(buf["max_keys"]), n_samples, lib.ptr(buf["x_acc"]), lib.ptr(buf["state"]),
                    lib.ptr(buf["workspace"]), buf["ws_bytes"], lib.ptr(status), flow.waves_per_group,
                    flow.max_groups, stream, rules,
                )
                lib.check(rc, "nsa_populate_passes")
            state_h.copy_(buf["state"])  # the train's one synchronisation
            n_accepted, passes = int(state_h[0]), int(state_h[1])
            if n_accepted >= n_samples:
                break
            if passes >= int(state_h[5]):
                logger.warning("Reached max samples (%s)", max_samples)
                break
            per_pass = n_accepted / passes if n_accepted else per_pass * 0.5
        n_proposed = passes * n
        self._draw_counter += n_proposed
        self._mark("passes")
        # accepted rows, their log-likelihood and the kernel's range flag come to the host through pinned
        # buffers behind ONE synchronisation (three pageable .cpu() copies, each its own round trip, before)
        m = min(n_accepted, n_samples)
        if lean:
            self._replay(buf["idx_acc"], m, threshold, buf["x_acc"], buf["log_q"], buf["log_p"] if separable else None,
                         (flow, header, image, status))
            self._mark("replay")
        kept = buf["x_acc"][:m]
        if buf.get("x_host") is None or buf["x_host"].shape[0] < n_samples:
            buf["x_host"] = torch.empty((n_samples, self.dims), dtype=torch.float64).pin_memory()
            buf["ll_host"] = torch.empty(n_samples, dtype=torch.float64).pin_memory()
            buf["status_host"] = torch.zeros(4, dtype=torch.int32).pin_memory()
        buf["x_host"][:m].copy_(kept, non_blocking=True)
        with_ll = self._ll_spec is not None and m > 0
        if with_ll:
            buf["ll_host"][:m].copy_(self._device_log_likelihood(kept), non_blocking=True)
        buf["status_host"].copy_(status, non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()
        log_l = buf["ll_host"][:m].numpy().copy() if with_ll else None
        self.x = array_to_live_points_fast(buf["x_host"][:m].numpy(), self.names)
        self._mark("read-back")
        self._finish_population(st, n_accepted, n_proposed, log_l=log_l, f16_flag=int(buf["status_host"][1]))
        self._mark("pool (log prior, permutation)")

    #: bytes of candidate rows a sharded train may keep on the device (its passes are independent until
    #: the stop test, so their rows wait for ONE exchange of the maxima and ONE of the counts)
    TRAIN_ARENA_BYTES = 4 << 30

    # -- the three device steps of a train of sharded passes (overridden by the CPU emulation of
    #    tests/test_distributed_gloo.py, which drives the loop below on 8 gloo ranks) ------------------------

    def _sharded_setup(self, n, n_samples):
        """Per-call device context of the sharded loop: prepared flow, buffers, rules, lean or not."""
        prepared = self._prepare_flow()
        buf = self._get_buffers(n, n_samples)
        rules = self._device_rules()
        lean = self._lean_ok(n, rules)
        separable = self._prior_kind_dev is not None
        return dict(prepared=prepared, buf=buf, rules=rules, lean=lean, separable=separable,
                    log_prior_const=0.0 if separable else self._uniform_log_prior_constant(),
                    key_init=int(lib.load().nsa_max_key_init()),
                    stream=lib.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))

    def _sharded_heads(self, ctx, n, train, off, stride, threshold, keys, state):
        """K heads: pass i draws rows [off + i stride, + n), leaves its weights in the arena and its maximum in
        ``keys[i]`` (``nsa_populate_train_heads`` / ``_lean``)."""
        L = lib.load()
        flow, header, image, status = ctx["prepared"]
        buf, separable = ctx["buf"], ctx["separable"]
        common = (header, lib.ptr(image), n, train, self.device_seed, off, stride, float(threshold),
                  float(self.latent_temperature or 1.0), lib.ptr(self._scale_dev), lib.ptr(self._shift_dev),
                  lib.ptr(self._lower_dev), lib.ptr(self._upper_dev), float(self._log_j_rescale))
        rest = (lib.ptr(buf["log_q"]), lib.ptr(self._prior_kind_dev), lib.ptr(self._prior_par_dev),
                lib.ptr(buf["log_p"] if separable else None), float(ctx["log_prior_const"]), lib.ptr(self._rp_kind_dev),
                lib.ptr(self._rp_par_dev), lib.ptr(buf["log_w_many"]), lib.ptr(keys), lib.ptr(state), lib.ptr(status),
                flow.waves_per_group, flow.max_groups, ctx["stream"], ctx["rules"])
        if ctx["lean"]:
            lib.check(L.nsa_populate_train_heads_lean(*common, *rest), "nsa_populate_train_heads_lean")
        else:
            lib.check(L.nsa_populate_train_heads(*common, lib.ptr(buf["x64_many"]), *rest), "nsa_populate_train_heads")

    def _sharded_tails(self, ctx, n, train, off, stride, n_samples, keys, state, lcount):
        """K accept + append steps against the GLOBAL maxima; ``lcount[i]`` = this rank's accepted rows of pass i
        (``nsa_populate_train_tails`` / ``_lean``)."""
        L = lib.load()
        buf = ctx["buf"]
        if ctx["lean"]:
            rc = L.nsa_populate_train_tails_lean(
                n, self.dims, train, lib.ptr(buf["log_w_many"]), lib.ptr(keys), self.device_seed, off, stride,
                n_samples, lib.ptr(buf["idx_acc"]), lib.ptr(state), lib.ptr(buf["scratch"]), lib.ptr(lcount),
                lib.ptr(buf["workspace"]), buf["ws_bytes"], ctx["stream"],
            )
            lib.check(rc, "nsa_populate_train_tails_lean")
        else:
            rc = L.nsa_populate_train_tails(
                n, self.dims, train, lib.ptr(buf["log_w_many"]), lib.ptr(keys), self.device_seed, off, stride,
                lib.ptr(buf["x64_many"]), n_samples, lib.ptr(buf["x_acc"]), lib.ptr(state), lib.ptr(buf["scratch"]),
                lib.ptr(lcount), lib.ptr(buf["workspace"]), buf["ws_bytes"], ctx["stream"],
            )
            lib.check(rc, "nsa_populate_train_tails")

    def _sharded_commit(self, ctx, train, n_samples, state, lcount, gcount, history):
        """Walk the K passes in order against the real loop state with the GLOBAL counts (``nsa_populate_train_commit``)."""
        rc = lib.load().nsa_populate_train_commit(lib.ptr(state), n_samples, train, lib.ptr(lcount), lib.ptr(gcount),
                                                  lib.ptr(history), ctx["stream"])
        lib.check(rc, "nsa_populate_train_commit")

    @torch.inference_mode(False)
    def _populate_trains_sharded(self, st, threshold, n_samples, max_samples, n, world, rank):
        """``_populate_trains`` for a draw batch sharded over ranks.  A pass needs two exchanges - the
        batch-global maximum of the weights before the accept step (``flowproposal.py:492``) and the
        global accepted count for the stop test - and at n = drawsize / world rows a pass is shorter
        than two collectives.  The passes of a train are independent until the stop test, so a train is
        ``K heads -> ONE all-reduce(max) of K maxima -> K tails -> ONE all-reduce(sum) of K counts ->
        commit`` (``nsa_populate_train_*``), all stream ordered: the commit replays the passes in order
        against the loop state, rows that surplus passes appended are never read.  Rows are appended
        locally and interleaved (pass, rank, draw order) once at the end - the order of a per-pass
        gather.  The train length is the number of passes the acceptance seen so far predicts,
        rounded down: a short train costs one more round of collectives, a long one whole draw
        kernels.  ``self.sharded_stats`` keeps what the call did: trains, passes, collectives."""
        import torch.distributed as dist

        self.last_populate_path = "trains_sharded"
        ctx = self._sharded_setup(n, n_samples)
        buf, lean, separable = ctx["buf"], ctx["lean"], ctx["separable"]
        dev = self.device
        # lean heads keep log_w alone: the arena of a train is 8 bytes per row instead of 8 (D + 1)
        self.last_populate_lean = lean
        if lean and buf.get("idx_acc") is None:
            buf["idx_acc"] = torch.empty(n_samples, dtype=torch.int64, device=dev)
        max_train = int(max(1, min(64, self.TRAIN_ARENA_BYTES // (8 * n * (1 if lean else self.dims + 1)))))
        if buf.get("state") is None:
            buf["state"] = torch.zeros(8, dtype=torch.int64, device=dev)
            buf["max_keys"] = torch.empty(64, dtype=torch.int64, device=dev)
            buf["state_host"] = _pinned(torch.zeros(8, dtype=torch.int64))
        if buf.get("gcount") is None:
            buf["gcount"] = torch.zeros(64, dtype=torch.int64, device=dev)
            buf["lcount"] = torch.zeros(64, dtype=torch.int64, device=dev)
            buf["scratch"] = torch.zeros(8, dtype=torch.int64, device=dev)
            buf["history"] = torch.zeros(256, dtype=torch.int64, device=dev)
        state, state_h = buf["state"], buf["state_host"]
        state_h.zero_()
        state_h[2] = n_samples
        state_h[5] = max_samples // (n * world) + 1
        state.copy_(state_h, non_blocking=True)
        per_pass = (self.population_acceptance or 0.0) * n * world
        n_accepted = passes = 0
        stride = n * world
        stats = dict(trains=0, passes=0, all_reduce=0, all_gather=0, train_lengths=[])
        while True:
            remaining = n_samples - n_accepted
            train = 1 if per_pass <= 0 else int(remaining / per_pass)
            train = max(1, min(train, max_train, int(state_h[5]) - passes))
            if buf.get("arena_passes", 0) < train or (not lean and buf.get("x64_many") is None):
                if not lean:
                    buf["x64_many"] = torch.empty((train, n, self.dims), dtype=torch.float64, device=dev)
                else:
                    buf["x64_many"] = None
                buf["log_w_many"] = torch.empty((train, n), dtype=torch.float64, device=dev)
                buf["arena_passes"] = train
            if buf["history"].numel() < passes + train:
                grown = torch.zeros(2 * (passes + train), dtype=torch.int64, device=dev)
                grown[: buf["history"].numel()] = buf["history"]
                buf["history"] = grown
            keys, lcount, gcount = buf["max_keys"][:train], buf["lcount"][:train], buf["gcount"][:train]
            keys.fill_(ctx["key_init"])
            off = int(self._draw_counter + rank * n + passes * stride)
            self._sharded_heads(ctx, n, train, off, stride, threshold, keys, state)
            dist.all_reduce(keys, op=dist.ReduceOp.MAX)
            self._sharded_tails(ctx, n, train, off, stride, n_samples, keys, state, lcount)
            gcount.copy_(lcount)
            dist.all_reduce(gcount, op=dist.ReduceOp.SUM)
            self._sharded_commit(ctx, train, n_samples, state, lcount, gcount, buf["history"])
            state_h.copy_(state)  # the train's one synchronisation
            stats["trains"] += 1
            stats["all_reduce"] += 2
            stats["train_lengths"].append(train)
            n_accepted, passes = int(state_h[0]), int(state_h[1])
            if n_accepted >= n_samples:
                break
            if passes >= int(state_h[5]):
                logger.warning("Reached max samples (%s)", max_samples)
                break
            per_pass = n_accepted / passes if n_accepted else per_pass * 0.5
        n_proposed = passes * n * world
        self._draw_counter += n_proposed
        stats["passes"] = passes
        stats["all_gather"] = 2 if world > 1 else 0  # per-pass counts + accepted rows (_assemble_sharded_pool)
        self.sharded_stats = stats
        if lean:  # this rank's accepted rows, computed again from their Philox rows
            self._replay(buf["idx_acc"], int(state_h[3]), threshold, buf["x_acc"], buf["log_q"],
                         buf["log_p"] if separable else None, ctx["prepared"])
        accepted = self._assemble_sharded_pool(buf["history"][:passes], buf["x_acc"], n_samples, world)
        self.x = array_to_live_points_fast(np.ascontiguousarray(accepted), self.names)
        log_l = None
        if self._ll_spec is not None and accepted.shape[0]:
            dev_rows = torch.from_numpy(np.ascontiguousarray(accepted)).to(dev)
            log_l = self._device_log_likelihood(dev_rows).cpu().numpy()
        self._finish_population(st, n_accepted, n_proposed, log_l=log_l)

    def _assemble_sharded_pool(self, counts_dev, x_acc, n_samples, world):
        """First ``n_samples`` accepted rows in (pass, rank, draw) order - the order of a per-pass
        gather - from every rank's per-pass counts and its locally appended rows: one exchange of
        the counts, one of the rows."""
        import torch.distributed as dist

        dev = self.device
        passes = int(counts_dev.numel())
        if world > 1:
            hist_all = torch.empty(world * passes, dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(hist_all, counts_dev.contiguous())
            hist = hist_all.cpu().numpy().reshape(world, passes)
        else:
            hist = counts_dev.cpu().numpy().reshape(1, passes)
        written = np.minimum(hist.sum(axis=1), n_samples)
        pad = max(int(written.max()), 1)
        if world > 1:
            rows_all = torch.empty((world * pad, self.dims), dtype=torch.float64, device=dev)
            dist.all_gather_into_tensor(rows_all, x_acc[:pad].contiguous())
            rows_all = rows_all.cpu().numpy().reshape(world, pad, self.dims)
        else:
            rows_all = x_acc[:pad].cpu().numpy().reshape(1, pad, self.dims)
        starts = np.concatenate([np.zeros((world, 1), dtype=np.int64), np.cumsum(hist, axis=1)], axis=1)
        pieces, have = [], 0
        for p in range(passes):
            for r in range(world):
                lo, hi = int(starts[r, p]), int(min(starts[r, p + 1], written[r]))
                if hi > lo and have < n_samples:
                    pieces.append(rows_all[r, lo:hi])
                    have += hi - lo
        return np.concatenate(pieces, axis=0)[:n_samples] if pieces else np.empty((0, self.dims))

    @torch.inference_mode(False)
    def _populate_accumulate(self, st, threshold, n_samples, max_samples, n, world, rank):
        """``accumulate_weights=True`` (``flowproposal.py:471-490, 504-510``) on the device, for one rank or
        a draw batch sharded over ranks.

        Every draw of the population is kept: pass ``p`` lives in slot ``p`` of an arena on the device
        (``x`` as float64 and ``log_w``; a dropped row keeps weight -inf).  The reference's test after
        each batch - ``logsumexp(log_weights - log_constant) >= log(n_samples)`` with the running
        ``log_constant = max(log_w)`` - needs, per pass, the maximum of the pass and
        ``sum exp(log_w - max)`` (``nsa_populate_sumexp``); sharded, the ranks exchange them with ONE
        all-reduce(max) and ONE all-reduce(sum) per train of passes ("the running
        log_constant all-reduced each pass") and every rank folds the passes into the same running
        constant.  The accept step over all rows so far (``log_w - log_constant > log u``, ``u`` from
        the Philox stream of the GLOBAL row, so the rows do not depend on the sharding) appends to the
        rank's pool; the counts are summed over the ranks and the loop goes on while fewer than
        ``n_samples`` rows were accepted, exactly as the reference redraws its uniforms over the
        grown set.  Passes a train drew beyond the one that met the test are kept for the next round
        of the loop (never counted unless used)."""
        import torch.distributed as dist

        L = lib.load()
        self.last_populate_path = "accumulate"
        flow, header, image, status = self._prepare_flow()
        buf = self._get_buffers(n, n_samples)
        dev = self.device
        D = self.dims
        if buf.get("state") is None:
            buf["state"] = torch.zeros(8, dtype=torch.int64, device=dev)
            buf["max_keys"] = torch.empty(64, dtype=torch.int64, device=dev)
            buf["state_host"] = torch.zeros(8, dtype=torch.int64).pin_memory()
        if buf.get("scratch") is None:
            buf["scratch"] = torch.zeros(8, dtype=torch.int64, device=dev)
        state, state_h = buf["state"], buf["state_host"]
        state_h.zero_()
        state_h[2] = 2**62  # the loop is steered from the host here: heads never stop by themselves
        state_h[5] = 2**62
        state.copy_(state_h, non_blocking=True)
        key_init = int(L.nsa_max_key_init())
        separable = self._prior_kind_dev is not None
        log_prior_const = 0.0 if separable else self._uniform_log_prior_constant()
        stream = lib.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        rules = self._device_rules()
        stride = n * world
        first = int(self._draw_counter + rank * n)
        log_n = np.log(n_samples)

        cap = 0  # arena slots
        x_many = lw_many = keys = sums = lcount = gcount = None
        pass_max, pass_sum = [], []  # per pass, global: max log_w and sum exp(log_w - max)
        have = used = 0
        log_constant, total = -np.inf, 0.0  # running max and sum exp(log_w - log_constant) over the used passes
        n_accepted = 0
        accepted_at = -1  # number of passes the last accept step covered
        hint = int(getattr(self, "_accumulate_passes_hint", 0))

        def grow(slots):
            nonlocal cap, x_many, lw_many, keys, sums, lcount, gcount
            if slots <= cap:
                return
            slots = max(slots, 2 * cap)
            nx = torch.empty((slots, n, D), dtype=torch.float64, device=dev)
            nw = torch.empty((slots, n), dtype=torch.float64, device=dev)
            if have:
                nx[:have] = x_many[:have]
                nw[:have] = lw_many[:have]
            x_many, lw_many, cap = nx, nw, slots
            keys = torch.empty(slots, dtype=torch.int64, device=dev)
            sums = torch.empty(slots, dtype=torch.float64, device=dev)
            lcount = torch.zeros(slots, dtype=torch.int64, device=dev)
            gcount = torch.zeros(slots, dtype=torch.int64, device=dev)

        def draw(k):
            """k more passes into the arena; their global maxima and sums onto the host lists."""
            nonlocal have
            grow(have + k)
            kk, ss = keys[have : have + k], sums[have : have + k]
            kk.fill_(key_init)
            rc = L.nsa_populate_train_heads(
                header, lib.ptr(image), n, k, self.device_seed, first + have * stride, stride, float(threshold),
                float(self.latent_temperature or 1.0), lib.ptr(self._scale_dev), lib.ptr(self._shift_dev),
                lib.ptr(self._lower_dev), lib.ptr(self._upper_dev), float(self._log_j_rescale),
                lib.ptr(x_many[have:]), lib.ptr(buf["log_q"]), lib.ptr(self._prior_kind_dev),
                lib.ptr(self._prior_par_dev), lib.ptr(buf["log_p"] if separable else None),
                float(log_prior_const), lib.ptr(self._rp_kind_dev), lib.ptr(self._rp_par_dev),
                lib.ptr(lw_many[have:]), lib.ptr(kk), lib.ptr(state), lib.ptr(status), flow.waves_per_group,
                flow.max_groups, stream, rules,
            )
            lib.check(rc, "nsa_populate_train_heads")
            if world > 1:
                dist.all_reduce(kk, op=dist.ReduceOp.MAX)
            rc = L.nsa_populate_sumexp(n, k, lib.ptr(lw_many[have:]), lib.ptr(kk), lib.ptr(ss), stream)
            lib.check(rc, "nsa_populate_sumexp")
            if world > 1:
                dist.all_reduce(ss, op=dist.ReduceOp.SUM)
            kk_h, ss_h = kk.cpu().numpy(), ss.cpu().numpy()  # the train's synchronisation
            for i in range(k):
                pass_max.append(float(L.nsa_max_key_decode(int(kk_h[i]))))
                pass_sum.append(float(ss_h[i]))
            have += k

        def accept(m):
            """Accept step over the first m passes with the current constant; global accepted count."""
            ckeys = torch.full((m,), int(L.nsa_max_key_encode(float(log_constant))), dtype=torch.int64, device=dev)
            rc = L.nsa_populate_train_tails(
                n, D, m, lib.ptr(lw_many), lib.ptr(ckeys), self.device_seed, first, stride, lib.ptr(x_many),
                n_samples, lib.ptr(buf["x_acc"]), lib.ptr(state), lib.ptr(buf["scratch"]), lib.ptr(lcount),
                lib.ptr(buf["workspace"]), buf["ws_bytes"], stream,
            )
            lib.check(rc, "nsa_populate_train_tails")
            g = gcount[:m]
            g.copy_(lcount[:m])
            if world > 1:
                dist.all_reduce(g, op=dist.ReduceOp.SUM)
            return int(g.sum().item())

        n_proposed = 0
        while n_accepted < n_samples:
            if used == have:
                draw(max(1, hint - used - 1) if hint > used else 1)
            m, s = pass_max[used], pass_sum[used]
            used += 1
            n_proposed = used * stride
            if m > log_constant:  # a new constant rescales what was summed so far
                total = total * np.exp(log_constant - m) if total > 0.0 else 0.0
                log_constant = m
            if s > 0.0:
                total += s * np.exp(m - log_constant)
            if total > 0.0 and np.log(total) >= log_n:
                n_accepted = accept(used)
                accepted_at = used
            if n_proposed > max_samples:
                logger.warning("Reached max samples (%s)", max_samples)
                break
        if accepted_at != used:
            n_accepted = accept(used) if np.isfinite(log_constant) else 0
        self._accumulate_passes_hint = used
        self._draw_counter += n_proposed
        if n_accepted:
            accepted = self._assemble_sharded_pool(lcount[:used], buf["x_acc"], n_samples, world)
        else:
            accepted = np.empty((0, D))
        self.x = array_to_live_points_fast(np.ascontiguousarray(accepted), self.names)
        log_l = None
        if self._ll_spec is not None and accepted.shape[0]:
            dev_rows = torch.from_numpy(np.ascontiguousarray(accepted)).to(dev)
            log_l = self._device_log_likelihood(dev_rows).cpu().numpy()
        self._finish_population(st, n_accepted, n_proposed, log_l=log_l)

    def _populate_accumulate_host(self, st, threshold, n_samples, max_samples, injected):
        """``accumulate_weights=True`` (``flowproposal.py:471-490, 504-510``) one pass per synchronisation:
        injected latents / uniforms (parity tests: the uniforms belong to the KEPT rows, as the
        reference's ``rng.random(len(log_weights))``), a prior evaluated on the host, or a likelihood rule
        on the user's host likelihood.  One rank only."""
        L = lib.load()
        self.last_populate_path = "accumulate_host"
        log_n = np.log(n_samples)
        xs, lws = [], []
        first_row = self._draw_counter
        n_proposed = 0
        n_accepted = 0
        accept_done = False
        buf = None
        u_all = []
        prepared = self._prepare_flow()
        while n_accepted < n_samples:
            z_in = None
            n_local = self.drawsize
            if injected is not None:
                z_np, u_np = next(injected)
                n_local = int(len(z_np))
                z_in = torch.from_numpy(np.ascontiguousarray(z_np)).to(torch.float32).to(self.device)
            fresh = buf is None
            buf = self._get_buffers(n_local, n_samples)
            self._launch_draw(n_local, z_in, self._draw_counter, threshold, buf["x64"], buf["log_q"], None,
                              log_prior=buf["log_p"] if self._prior_kind_dev is not None else None,
                              prepared=prepared)
            self._truncate_after_flow(buf, n_local, *self._post_limits)
            self._weights_and_max(buf, n_local, reset_max=fresh)
            n_proposed += n_local
            self._draw_counter += n_local
            valid = ~torch.isnan(buf["log_q"][:n_local])  # every row backward_pass returns, weight -inf included
            if injected is not None:  # uniforms belong to draws: keep those of the kept rows
                u_all.append(np.asarray(u_np, dtype=np.float64)[valid.cpu().numpy()])
            xs.append(buf["x64"][:n_local][valid].clone())
            lws.append(buf["log_w"][:n_local][valid].clone())
            log_weights = torch.cat(lws)
            log_constant = float(L.nsa_max_key_decode(int(buf["max_key"].item())))
            log_n_expected = float(torch.logsumexp(log_weights - log_constant, dim=0).item())
            if log_n_expected >= log_n:
                n_accepted, block = self._accept_accumulated(
                    buf, torch.cat(xs), log_weights, n_samples, first_row, u_all, injected
                )
                accept_done = True
            if n_proposed > max_samples:
                logger.warning("Reached max samples (%s)", max_samples)
                break
        if not accept_done:
            if lws:
                n_accepted, block = self._accept_accumulated(
                    buf, torch.cat(xs), torch.cat(lws), n_samples, first_row, u_all, injected
                )
            else:
                n_accepted, block = 0, np.empty((0, self.dims))
        self.x = array_to_live_points_fast(block[:n_samples], self.names)
        self._finish_population(st, n_accepted, n_proposed)

    def _accept_accumulated(self, buf, x_all, log_w_all, n_samples, first_row, u_all, injected):
        n = int(log_w_all.shape[0])
        L = lib.load()
        ws_bytes = int(L.nsa_populate_accept_workspace(n))
        if ws_bytes > buf["ws_bytes"]:
            buf["workspace"] = torch.empty(ws_bytes, dtype=torch.uint8, device=self.device)
            buf["ws_bytes"] = ws_bytes
        u_in = None
        if injected is not None:
            u_in = torch.from_numpy(np.concatenate(u_all)[:n].copy()).to(self.device)
        self._accept(buf, n, n_samples, first_row, u_in=u_in, log_w=log_w_all.contiguous(),
                     x64=x_all.contiguous())
        n_acc = int(buf["n_acc"].item())
        return n_acc, buf["x_acc"][: min(n_acc, n_samples)].cpu().numpy()

    def _finish_population(self, st, n_accepted, n_proposed, log_l=None, f16_flag=None):
        # A population whose draw kernel left the f16 range is discarded HERE: no HOST likelihood call on its
        # (in-bounds but invalid) rows, no permutation drawn from the host generator, pool, counters and
        # acceptance untouched; populate() redraws it in float32.  What it does leave behind: its Philox
        # offsets (`_draw_counter` has advanced - the redraw uses fresh rows, so a run that fell back is not
        # row-for-row the run that started in float32) and, in the device-side loops, a device likelihood
        # evaluation of the discarded rows, whose result is dropped here.
        if self._f16_range_left(f16_flag):
            self._f16_discarded = True
            self.populated = False
            self.x = None
            return
        self.samples = self.convert_to_samples(self.x)
        self.population_time += datetime.datetime.now() - st
        if log_l is not None:  # evaluated on the device with the rows (nsa_log_likelihood)
            self.samples["logL"] = log_l
        else:
            self.samples["logL"] = self.model.batch_evaluate_log_likelihood(self.samples)
        self.indices = self.rng.permutation(self.samples.size).tolist()
        self.n_accepted = int(n_accepted)
        self.n_proposed = int(n_proposed)
        self.population_acceptance = n_accepted / n_proposed if n_proposed else 0.0
        self.populated_count += 1
        self.populated = True
        self._checked_population = False

    def _refill(self, worst_point):
        """Populate until the pool holds samples again (the sampler's acceptance sets its size)."""
        self.populating = True
        try:
            if self.update_poolsize:
                self.update_poolsize_scale(self.ns_acceptance)
            while not self.populated:
                self.populate(worst_point, n_samples=self.poolsize)
        finally:
            self.populating = False

    def draw(self, worst_point):
        """Next pooled sample; an empty pool is refilled first (``base.py:1152-1183``)."""
        if not self.populated:
            self._refill(worst_point)
        sample = self.samples[self.indices.pop()]
        self.populated = bool(self.indices)
        return sample

    def reset_model_weights(self, **kwargs):
        self.flow.reset_model(**kwargs)

    # -- checkpoint / resume (base.py:1237-1309, proposal/base.py:64-80) -------------------

    def update_output(self, output):
        """Move the proposal (and the weights files it writes from now on) to another directory."""
        self.output = output
        os.makedirs(self.output, exist_ok=True)
        if self.flow is not None:
            self.flow.output = output

    def resume(self, model, flow_config, weights_file=None):
        """Bring an unpickled proposal back to life (``base.py:1237-1271``): the model and the flow
        configuration are not pickled and must be given again; the reparameterisations keep their
        fitted constants; the flow is rebuilt and its last weights are loaded (``weights_file``, or
        the file the proposal last saved).  The pool is not kept: the sampler repopulates."""
        self.model = model
        self.flow_config = dict(flow_config or {})
        self.initialise(resumed=True)
        if weights_file is None:
            weights_file = self.weights_file
        if weights_file is not None and os.path.exists(weights_file):
            self.flow.reload_weights(weights_file)
        else:
            logger.warning("Could not reload weights for flow")

    def reset(self):
        """(``base.py:1273-1284``)"""
        self.indices = []
        self.samples = None
        self.x = None
        self.populated = False
        self.populated_count = 0
        self.population_acceptance = None
        self._poolsize_scale = 1.0
        self._checked_population = True
        self.acceptance = []
        self.truncation.reset()

    def __getstate__(self):
        state = self.__dict__.copy()
        state["initialised"] = False
        state["_buffers"] = None
        for k in ("_scale_dev", "_shift_dev", "_lower_dev", "_upper_dev", "_rp_kind_dev", "_rp_par_dev",
                  "_prior_kind_dev", "_prior_par_dev", "_ll_spec", "device"):
            state.pop(k, None)
        state["weights_file"] = getattr(self.flow, "weights_file", None) if self.flow else None
        state["resume_populated"] = bool(state.get("populated") and state.get("indices"))
        # the user provides the model and the flow configuration again on resume
        state["model"] = None
        state["flow_config"] = None
        state["flow"] = None
        return state


def _pinned(t):
    """``t`` in page-locked memory where a device is present (a CPU-only process - the gloo tests - keeps it pageable)."""
    return t.pin_memory() if torch.cuda.is_available() else t


def _gather_accepted(x_acc, n_acc_dev, capacity, dims):
    """All-gather the accepted rows of every rank, rank-major then draw order.

    ``x_acc``: this rank's compaction buffer ``(capacity, dims)``; ``n_acc_dev``: its accepted
    count, still on the device.  One collective for the counts (whose host copy is the pass's
    first synchronisation), one for the rows padded to the largest count;
    payloads are a few tens of KB, i.e. latency-bound on xGMI.
    """
    import torch.distributed as dist

    world = dist.get_world_size()
    counts = torch.empty(world, dtype=torch.int64, device=x_acc.device)
    dist.all_gather_into_tensor(counts, n_acc_dev.reshape(1))
    counts_h = counts.cpu().numpy()
    kept = np.minimum(counts_h, capacity)
    pad = max(int(kept.max()), 1)
    send = x_acc[:pad].contiguous()  # rows beyond this rank's count are ignored by the receivers
    recv = torch.empty((world * pad, dims), dtype=torch.float64, device=x_acc.device)  # concatenated form
    dist.all_gather_into_tensor(recv, send)
    parts = [recv[r * pad : r * pad + int(kept[r])] for r in range(world)]
    return int(counts_h.sum()), torch.cat(parts, dim=0)
