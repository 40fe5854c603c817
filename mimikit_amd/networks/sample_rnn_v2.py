"""SampleRNN behind the ARM protocol, generating on the MI355X.

Config, tier wiring and ``state_dict`` layout follow the reference
(``mimikit/networks/sample_rnn_v2.py``: ``SampleRNNTier`` :35-119, ``SampleRNN``
:122-311).  The tiered per-step schedule of ``generate_step`` (:236-260) and the
shifted warm-up of ``before_generate`` (:226-234) run inside
``csrc/srnn_plan.hip``; recurrent state lives in HBM between steps.
``eval()`` paths run only on the HIP device; training-mode ``forward`` is the
stock differentiable torch graph (out of scope, kept for the trainer).
"""
import dataclasses as dtc
from enum import auto
from typing import Dict, Iterable, List, Optional, Set, Tuple, Union

import torch
import torch.nn as nn

from .. import native
from ..features.functionals import Discrete
from ..features.item_spec import ItemSpec
from ..io_spec import IOSpec
from ..modules.io import FramedConv1dIO, FramedLinearIO, ZipMode, ZipReduceVariables
from ..modules.mlp import MLP
from ..modules.resamplers import LinearResampler
from ..modules.targets import OutputWrapper, per_row_temperature
from ..utils import AutoStrEnum
from .arm import fold_weight_norm, weight_norm_leaves, ARMWithHidden, NetworkConfig

__all__ = ["SampleRNN", "SampleRNNTier"]

T = torch.Tensor


class RNNType(AutoStrEnum):
    lstm = auto()
    rnn = auto()
    gru = auto()
    none = auto()


class H0Init(AutoStrEnum):
    zeros = auto()
    ones = auto()
    randn = auto()


class SampleRNNTier(nn.Module):
    """input projection (+ upper tier vector) -> RNN -> linear up-sampler; parameters under the
    reference's names, differentiable forward for training"""

    def __init__(self, *, input_module: nn.Module = nn.Identity(), hidden_dim: int = 256,
                 rnn_class: RNNType = "lstm", n_rnn: int = 1, rnn_dropout: float = 0., rnn_bias: bool = True,
                 h0_init: H0Init = "zeros", weight_norm: bool = False, up_sampling: Optional[int] = None):
        super().__init__()
        self.input_module = input_module
        self.hidden_dim, self.rnn_class, self.n_rnn = hidden_dim, str(rnn_class), n_rnn
        self.rnn_dropout, self.rnn_bias, self.h0_init = rnn_dropout, rnn_bias, str(h0_init)
        self.weight_norm, self.up_sampling = weight_norm, up_sampling
        self.hidden = None
        self.has_rnn = self.rnn_class != "none"
        self.has_up_sampling = up_sampling is not None
        if self.has_rnn:
            make = getattr(nn, self.rnn_class.upper())
            self.rnn = make(hidden_dim, hidden_dim, num_layers=n_rnn, batch_first=True, dropout=rnn_dropout,
                            bias=rnn_bias)
        if self.has_up_sampling:
            self.up_sampler = LinearResampler(hidden_dim, t_factor=up_sampling, d_factor=1)
        if weight_norm:
            # the reference re-parametrises EVERY parameter of the RNN, the up-sampler and the leaves of the input module,
            # biases included (:67-81): ``name`` becomes ``name_g`` / ``name_v`` in the state_dict
            if self.has_rnn:
                for name in dict(self.rnn.named_parameters()):
                    nn.utils.weight_norm(self.rnn, name)
            if self.has_up_sampling:
                for module in self.up_sampler.children():
                    for name in dict(module.named_parameters()):
                        nn.utils.weight_norm(module, name)
            weight_norm_leaves(self.input_module)

    def _fresh(self, batch: int, device):
        return getattr(torch, self.h0_init)(self.n_rnn, batch, self.hidden_dim).to(device)

    def _reset_hidden(self, x: T, hidden):
        is_lstm = self.rnn_class == "lstm"
        current = hidden[0] if (is_lstm and hidden is not None) else hidden
        if current is None or x.size(0) != current.size(1):
            if is_lstm:
                return self._fresh(x.size(0), x.device), self._fresh(x.size(0), x.device)
            return self._fresh(x.size(0), x.device)
        return tuple(h.detach() for h in hidden) if is_lstm else hidden.detach()

    def forward(self, inputs: Tuple[Tuple[T, ...], Optional[T]]) -> T:
        x, x_upper = inputs
        x = self.input_module(x)
        if x_upper is not None:
            x = x + x_upper
        if self.has_rnn:
            self.hidden = self._reset_hidden(x, self.hidden)
            x, self.hidden = self.rnn(x, self.hidden)
        if self.has_up_sampling:
            x = self.up_sampler(x)
        return x


class SampleRNN(ARMWithHidden, nn.Module):
    @dtc.dataclass
    class Config(NetworkConfig):
        frame_sizes: Tuple[int, ...] = (16, 8, 8)
        hidden_dim: int = 256
        rnn_class: RNNType = "lstm"
        n_rnn: int = 1
        rnn_dropout: float = 0.
        rnn_bias: bool = True
        h0_init: H0Init = "zeros"
        weight_norm: bool = False
        inputs_mode: ZipMode = "sum"
        io_spec: IOSpec = None

    @classmethod
    def from_config(cls, config: "SampleRNN.Config") -> "SampleRNN":
        h, fs = config.hidden_dim, config.frame_sizes
        tiers = []
        for i, size in enumerate(fs[:-1]):
            heads = tuple(spec.module.copy().set(frame_size=size, hop_length=size, out_dim=h).module()
                          for spec in config.io_spec.inputs)
            tiers.append(SampleRNNTier(
                input_module=ZipReduceVariables(mode=config.inputs_mode, modules=heads), hidden_dim=h,
                rnn_class=config.rnn_class, n_rnn=config.n_rnn, rnn_dropout=config.rnn_dropout,
                rnn_bias=config.rnn_bias, h0_init=config.h0_init, weight_norm=config.weight_norm,
                # every RNN tier up-samples to the next tier's rate, the last one to the sample rate
                up_sampling=size // (fs[i + 1] if i < len(fs) - 2 else 1)))
        heads = []
        for spec in config.io_spec.inputs:
            if isinstance(spec.elem_type, Discrete) and not isinstance(spec.module, FramedLinearIO):
                # (the reference builds an EmbeddingConv1d bottom tier here, :161-167, but cannot run the resulting network: its
                #  upper tiers then embed samples, not frames, and forward fails on mismatched lengths - DESIGN.md section 8)
                raise NotImplementedError("a SampleRNN on discrete inputs needs FramedLinearIO input modules (IOSpec.mulaw_io's default): "
                                          "the reference's own network fails in forward with any other")
            params = dict(class_size=spec.elem_type.size) if isinstance(spec.elem_type, Discrete) else {}
            heads.append(FramedConv1dIO().set(**params, frame_size=fs[-1], hop_length=1, out_dim=h).module())
        tiers.append(SampleRNNTier(input_module=ZipReduceVariables(mode=config.inputs_mode, modules=heads),
                                   hidden_dim=h, rnn_class="none", up_sampling=None))
        output_module = [spec.module.copy().set(in_dim=h).module() for spec in config.io_spec.targets]
        return cls(config=config, tiers=tiers, output_module=output_module)

    def __init__(self, *, config: "SampleRNN.Config", tiers: Iterable[nn.Module], output_module: List[nn.Module]):
        super().__init__()
        self._config = config
        self.frame_sizes = config.frame_sizes
        self.tiers: List[SampleRNNTier] = nn.ModuleList(tiers)
        self.output_modules = nn.ModuleList(output_module)
        if config.weight_norm:
            weight_norm_leaves(self.output_modules)
        self.outputs = []
        self.prompt_length = 0
        self._plan: Optional[native.SrnnPlan] = None
        self._plan_batch = 0
        self._weights = native.WeightsTracker()
        self.exec_tuning = {}   # execution switches of THIS network's plans ({"MMK_...": "0"}: include/mmk.h `tuning`); merged over native.PLAN_TUNING
        self._plan_tuning = None            # the tuning text the plan at hand was built with
        self._state_batch = 0
        self._next_t: Optional[int] = None

    # -- ARM properties -----------------------------------------------------------
    @property
    def config(self):
        return self._config

    @property
    def rf(self):
        return self.frame_sizes[0]

    def train_batch(self, item_spec: ItemSpec):
        fs0 = self.frame_sizes[0]
        return tuple(
            spec.to_batch_item(ItemSpec(shift=0, length=fs0, unit=spec.unit) + item_spec)
            for spec in self.config.io_spec.inputs
        ), tuple(
            spec.to_batch_item(ItemSpec(shift=fs0, unit=spec.unit) + item_spec)
            for spec in self.config.io_spec.targets
        )

    def test_batch(self, item_spec: ItemSpec):
        fs0 = self.frame_sizes[0]
        return tuple(
            spec.to_batch_item(item_spec.to(spec.unit)) for spec in self.config.io_spec.inputs
        ), tuple(
            spec.to_batch_item(ItemSpec(shift=fs0, length=-fs0, unit=spec.unit) + item_spec)
            for spec in self.config.io_spec.targets
        )

    @property
    def generate_params(self) -> Set[str]:
        return {p for m in self.output_modules for p in getattr(m, "sampling_params", {})}

    # -- differentiable forward (training only) --------------------------------------
    def forward(self, inputs: Tuple):
        # mode-independent, as in the reference (:188-199): the teacher-forced graph over whole frames (training steps and
        # the trainer's validation); generation goes through before_generate / generate_step / generate_block (HIP)
        fs0, prev = self.frame_sizes[0], None
        for tier, fs in zip(self.tiers[:-1], self.frame_sizes[:-1]):
            prev = tier((tuple(x[:, fs0 - fs:-fs] for x in inputs), prev))
        fs = self.frame_sizes[-1]
        prev = self.tiers[-1]((tuple(x[:, fs0 - fs:-1] for x in inputs), prev))
        return tuple(mod(prev) for mod in self.output_modules)

    # -- HIP plan ---------------------------------------------------------------------
    _exec_mode = 0          # 1 while a batch is being redone with the kernels in turns (mmk_srnn_config.exec_mode)

    def _describe(self, max_batch: int) -> native.SrnnConfig:
        cfg, io = self._config, self._config.io_spec
        unsupported = []
        n_in, n_tgt = len(io.inputs), len(io.targets)
        if n_in > native.MAX_STREAMS or n_tgt > n_in:
            unsupported.append(f"more than {native.MAX_STREAMS} inputs, or more targets than inputs (the loop writes output k into "
                               "input k, loops/generate.py:213-218)")
        if not 1 <= cfg.n_rnn <= 8:
            unsupported.append("n_rnn outside [1, 8]")
        if cfg.rnn_dropout:
            unsupported.append("rnn_dropout")
        if str(cfg.h0_init) == "randn":
            unsupported.append("h0_init='randn'")
        if str(cfg.rnn_class) not in ("lstm", "gru", "rnn"):
            unsupported.append(f"rnn_class='{cfg.rnn_class}'")
        if len(cfg.frame_sizes) > native.MAX_TIERS:
            unsupported.append("too many tiers")
        head = self.output_modules[0]
        c = native.SrnnConfig()
        if isinstance(head, OutputWrapper) and native.only_mlp(head.estimator):
            mlp: MLP = head.estimator[0]
            if native.mlp_head_problem(mlp, self.training):
                unsupported.append(native.mlp_head_problem(mlp, self.training))
            else:
                c.mlp_act = native.mlp_act(mlp)
            if mlp.n_hidden_layers > 4:
                unsupported.append("n_mlp_layers > 4")
            c.mlp_hidden, c.mlp_n_hidden, c.learn_temp = mlp.hidden_dim, mlp.n_hidden_layers, int(mlp.learn_temperature)
            c.q_levels = mlp.out_dim - c.learn_temp
            c.min_temp = float(mlp.min_temp) if mlp.learn_temperature else 0.
        else:
            unsupported.append(f"output module of type {type(head).__name__}")
        # several inputs (every tier's ZipReduceVariables, :141-145, :160-173) and targets (:181-182)
        c.n_inputs, c.n_targets = n_in, n_tgt
        c.inputs_mode = {"sum": 0, "mean": 1, "static_mix": 2}[str(cfg.inputs_mode)]
        for m, spec in enumerate(io.inputs[:native.MAX_STREAMS]):
            c.in_class[m] = spec.elem_type.size
        for k in range(1, min(n_tgt, native.MAX_STREAMS)):
            hk = self.output_modules[k]
            if not (isinstance(hk, OutputWrapper) and native.only_mlp(hk.estimator)):
                unsupported.append(f"output module {k} of type {type(hk).__name__}")
                continue
            mlp = hk.estimator[0]
            if native.mlp_head_problem(mlp, self.training) or mlp.n_hidden_layers > 4 or native.mlp_act(mlp) != c.mlp_act:
                unsupported.append(f"target {k}: {native.mlp_head_problem(mlp, self.training) or 'more than 4 hidden layers, or another activation than target 0'}")
            c.x_mlp_hidden[k], c.x_mlp_n_hidden[k], c.x_learn_temp[k] = mlp.hidden_dim, mlp.n_hidden_layers, int(mlp.learn_temperature)
            c.x_q_levels[k] = mlp.out_dim - c.x_learn_temp[k]
            c.x_min_temp[k] = float(mlp.min_temp) if mlp.learn_temperature else 0.
        for k in range(min(n_tgt, n_in, native.MAX_STREAMS)):
            if (c.q_levels if k == 0 else c.x_q_levels[k]) > c.in_class[k]:
                unsupported.append(f"target {k} draws classes that input {k} cannot take")
        if unsupported:
            raise NotImplementedError("the HIP generate path does not cover: " + "; ".join(unsupported))
        c.n_tiers = len(cfg.frame_sizes)
        for i, fs in enumerate(cfg.frame_sizes):
            c.frame_size[i] = fs
        c.hidden_dim = cfg.hidden_dim
        c.rnn_kind = {"lstm": 0, "gru": 1, "rnn": 2}[str(cfg.rnn_class)]
        c.rnn_bias = int(cfg.rnn_bias)
        c.n_rnn = int(cfg.n_rnn)
        c.exec_mode = int(self._exec_mode)
        c.tuning = native.tuning_text(native.PLAN_TUNING, self.exec_tuning)       # execution switches of this plan (never the environment)
        c.h0_ones = int(str(cfg.h0_init) == "ones")
        c.max_batch = max_batch
        return c

    def _ensure_plan(self, batch: int, refresh_weights: bool):
        device = self.device
        if device.type != "cuda":
            raise RuntimeError("SampleRNN generates on the MI355X only: move the network to the HIP device ('cuda'); "
                               "there is no CPU implementation in this package")
        rebuilt = False
        tuning = native.tuning_text(native.PLAN_TUNING, self.exec_tuning)
        if self._plan is None or self._plan_tuning != tuning or self._plan_batch < batch or self._plan.device != device:
            self._plan = native.SrnnPlan(self._describe(max(batch, 1)), device)
            self._plan_batch = max(batch, 1)
            self._plan_tuning = tuning
            self._resident_seen = 0                          # (the new plan's resident-block counter starts over)
            rebuilt = True
        if rebuilt or refresh_weights:
            if rebuilt or self._weights.changed(self, content=True):      # re-pack only when a parameter changed since the last commit
                sd = self.state_dict()
                for k, head in enumerate(self.output_modules):      # (a head with dropout modules between its Linears: the plans know `fc.{2 i}`)
                    est = getattr(head, "estimator", None)
                    if native.only_mlp(est):
                        sd = native.mlp_linear_keys(sd, f"output_modules.{k}.estimator.0.", est[0])
                self._plan.bind_state_dict(fold_weight_norm(sd) if self._config.weight_norm else sd)
                self._plan.commit()
                self._weights.committed(self)
            else:
                self._plan.reset()                           # hidden states back to h0, same packed weights
            self._next_t = None

    def _sampling(self, batch: int, n_steps: int, parameters: Dict):
        temperature = parameters.get("temperature", None)
        if temperature is None:
            return None, None
        n_tgt = len(self.output_modules)
        return (per_row_temperature(temperature, batch, self.device),
                torch.rand((batch, n_steps) if n_tgt == 1 else (n_tgt, batch, n_steps), device=self.device, dtype=torch.float32))

    # -- ARM generation protocol ------------------------------------------------------
    def reset_hidden(self) -> None:
        for t in self.tiers:
            t.hidden = None
        if self._plan is not None and self._plan.workspace is not None:
            self._plan.reset()
        self._next_t = None

    def before_generate(self, prompts: Tuple[torch.Tensor, ...], batch_index) -> None:
        prompts = tuple(prompts)
        native.require_device(*prompts)
        idx = prompts[0]
        batch, length = idx.size(0), idx.size(1)
        self._blocks = []                       # the generate_block calls since this warm-up (replayed after a reported timeout)
        self._ensure_plan(batch, refresh_weights=True)   # also resets the hidden state
        offset = length % self.rf
        self.prompt_length = length - offset
        if length < self.rf:
            raise RuntimeError(f"prompt of {length} steps is shorter than frame_sizes[0]={self.rf}")
        streams = tuple((x if x.stride(1) == 1 else x.contiguous()) for x in prompts[:len(self.tiers[0].input_module.heads)])
        self._plan.warmup(tuple(x.long() if x.dtype != torch.int64 else x for x in streams), length)
        self._next_t, self._state_batch = length, batch

    def generate_step(self, inputs: Tuple[torch.Tensor, ...], *, t: int = 0, **parameters):
        inputs = tuple(inputs)
        native.require_device(*inputs)
        window = inputs[0]
        batch, rf = window.size(0), self.rf
        if window.size(1) < rf:
            raise RuntimeError(f"window of {window.size(1)} steps is shorter than frame_sizes[0]={rf}")
        if self._plan is None or self._state_batch != batch or self._plan.workspace is None:
            self._ensure_plan(batch, refresh_weights=True)
            self._state_batch = batch
        if t < self.prompt_length:
            return ()     # steps inside the prompt only advance the tiers (:254-255): before_generate did that on the device
        bufs = []
        for x in inputs:               # scratch copies of the windows with one free column for the produced step
            buf = torch.cat([x[:, -rf:], torch.zeros_like(x[:, :1])], dim=1)
            bufs.append((buf if buf.dtype == torch.int64 else buf.long()).contiguous())
        temp, uni = self._sampling(batch, 1, parameters)
        self._plan.generate(tuple(bufs), t, 1, temp, uni, t_first=t - rf)
        self._next_t = t + 1
        return tuple(bufs[k][:, rf:rf + 1] for k in range(len(self.output_modules)))

    def generate_block(self, tensors: Tuple[torch.Tensor, ...], t0: int, n_steps: int, **parameters):
        tensors = tuple(tensors)
        native.require_device(*tensors)
        idx = tensors[0]
        batch = idx.size(0)
        if any(x.dtype != torch.int64 for x in tensors):
            raise TypeError("generate_block writes in place: the tensors must be int64 class indices")
        if self._plan is None or self._next_t != t0 or self._state_batch != batch:
            self.before_generate(tuple(x[:, :t0] for x in tensors), None)
        temp, uni = self._sampling(batch, n_steps, parameters)
        self._plan.generate(tensors, t0, n_steps, temp, uni, t_first=0)
        self._next_t = t0 + n_steps
        getattr(self, "_blocks", []).append((tensors, t0, n_steps, dict(parameters)))
        return True

    def after_generate(self, final_outputs: Tuple[torch.Tensor, ...], batch_index) -> None:
        self.outputs = []
        if self._plan is not None and self._plan.resident_blocks() != getattr(self, "_resident_seen", 0):   # (a new plan counts from 0)
            # resident mode relies on the tier kernels and the bottom kernel running side by side; a wait that timed out (the
            # CUs were held by something else) leaves invalid samples: regenerate the batch once with the kernels in turns
            self._resident_seen = self._plan.resident_blocks()
            try:
                self._plan.sync_status()
            except native.NativeError as err:
                self._redo_in_turns(err)
        self.reset_hidden()

    def _redo_in_turns(self, err):
        import warnings
        blocks, self._blocks = getattr(self, "_blocks", []), []
        if not blocks or self._exec_mode == 1:
            raise err
        warnings.warn(f"{err}; regenerating this batch with the tier and bottom kernels in turns")
        self._exec_mode = 1                      # the next plan is created with exec_mode = 1: no resident mode
        try:
            self._plan = None
            first_tensors, first_t0 = blocks[0][0], blocks[0][1]
            self.before_generate(tuple(x[:, :first_t0] for x in first_tensors), None)
            for tensors, t0, n_steps, params in blocks:
                self.generate_block(tensors, t0, n_steps, **params)
            torch.cuda.synchronize(self.device)
        finally:
            self._exec_mode = 0
            self._blocks = []
            self._plan = None                    # the next generation may run resident again
            self._next_t = None
