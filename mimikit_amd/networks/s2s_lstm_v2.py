"""Seq2Seq bi-LSTM frame predictor behind the ARM protocol, generating on the MI355X.

Config, module wiring and ``state_dict`` layout follow the reference
(``mimikit/networks/s2s_lstm_v2.py``: ``EncoderLSTM`` :53-116, ``DecoderLSTM``
:119-182, ``Seq2SeqLSTMNetwork`` :185-303).  One ``generate_step`` maps the last
``hop`` frames to the next ``hop`` frames; on the HIP device it runs as
``csrc/s2s_plan.hip`` (GEMM-shaped: fp32 matrix cores).  Covered option space:
continuous (magspec) frames or class indices (embedding in, MLP head + argmax out), every ``enc_downsampling`` /
``dec_upsampling``, up to 8 LSTMs per side, residuals, weight norm (DESIGN.md section 8 lists what is refused).
"""
import dataclasses as dtc
from enum import auto
from typing import Dict, Set, Tuple

import torch
import torch.nn as nn

from .. import native
from ..features.functionals import Continuous
from ..features.item_spec import ItemSpec
from ..io_spec import IOSpec
from ..modules.io import ZipReduceVariables
from ..modules.misc import Chunk
from ..modules.mlp import MLP
from ..modules.targets import CategoricalSampler, per_row_temperature
from ..modules.resamplers import LinearResampler
from ..utils import AutoStrEnum
from .arm import ARMWithHidden, NetworkConfig, fold_weight_norm, weight_norm_leaves

__all__ = ["EncoderLSTM", "DecoderLSTM", "Seq2SeqLSTMNetwork"]


class DownSampling(AutoStrEnum):
    edge_sum = auto()
    edge_mean = auto()
    sum = auto()
    mean = auto()
    linear_resample = auto()


class UpSampling(AutoStrEnum):
    repeat = auto()
    interp = auto()
    linear_resample = auto()


def _bi_lstms(in_dim, dim, n):
    return nn.ModuleList([nn.LSTM(in_dim if i == 0 else dim, dim, batch_first=True, bidirectional=True)
                          for i in range(n)])


def _fold_directions(y, dim):
    """the reference's ``y.view(..., dim, 2).sum(-1)`` (:100, :174): sums ADJACENT channel pairs of the
    [forward | backward] concatenation"""
    return y.view(*y.size()[:-1], dim, 2).sum(dim=-1)


class EncoderLSTM(nn.Module):
    def __init__(self, downsampling: str, input_dim: int = 512, output_dim: int = 512, num_layers: int = 1,
                 hop: int = 4, apply_residuals: bool = False, weight_norm: bool = False):
        super().__init__()
        self.downsampling, self.input_dim, self.output_dim = str(downsampling), input_dim, output_dim
        self.num_layers, self.hop, self.apply_residuals = num_layers, hop, apply_residuals
        self.lstm = _bi_lstms(input_dim, output_dim, num_layers)
        if self.downsampling == "linear_resample":
            self.fc = LinearResampler(output_dim, 1 / hop, 1)
        self.fc_out = nn.Linear(output_dim, output_dim, bias=False)
        self.hidden = [None] * num_layers
        if weight_norm:
            weight_norm_leaves(self)      # every parameter of every leaf becomes a (g, v) pair (:86-91)

    def forward(self, x):
        assert x.size(1) == self.hop
        for n, lstm in enumerate(self.lstm):
            y, self.hidden[n] = lstm(x)   # fresh zero state on every call
            y = _fold_directions(y, self.output_dim)
            x = x + y if (n > 0 and self.apply_residuals) else y
        if self.downsampling == "linear_resample":
            return self.fc_out(self.fc(x)), self.hidden[-1]
        x = x.unfold(1, self.hop, self.hop)
        if "edge" in self.downsampling:
            x = x[..., [0, -1]]
        pooled = x.sum(dim=-1) if "sum" in self.downsampling else x.mean(dim=-1)
        return self.fc_out(pooled), self.hidden[-1]


class DecoderLSTM(nn.Module):
    def __init__(self, upsampling: str, model_dim: int = 512, num_layers: int = 1, hop: int = 4,
                 apply_residuals: bool = False, weight_norm: bool = False):
        super().__init__()
        self.upsampling = str(upsampling)
        self.output_dim = self.model_dim = model_dim
        self.num_layers, self.hop, self.apply_residuals = num_layers, hop, apply_residuals
        self.lstm = _bi_lstms(model_dim, model_dim, num_layers)
        if self.upsampling == "linear_resample":
            self.fc = LinearResampler(model_dim, hop, 1)
        self.hidden = [None] * num_layers
        if weight_norm:
            weight_norm_leaves(self)      # (:148-153)

    def forward(self, x, hidden=None):
        assert x.size(1) == 1
        if self.upsampling == "linear_resample":
            x = self.fc(x)
        elif self.upsampling == "repeat":
            x = x.repeat_interleave(self.hop, 1)
        elif self.upsampling == "interp":
            # the encoder's two final states (forward, reverse) spread over the hop frames, nearest neighbour (:162-165)
            interp = nn.functional.interpolate(hidden[0].permute(1, 2, 0), (self.hop,)).permute(0, 2, 1)
            x = x.expand(-1, self.hop, -1) + interp
        else:
            raise ValueError(f"unknown dec_upsampling '{self.upsampling}'")
        self.hidden[0] = hidden
        for n, lstm in enumerate(self.lstm):
            y, self.hidden[n] = lstm(x, hidden)   # every layer is seeded with the encoder state (:171)
            y = _fold_directions(y, self.model_dim)
            x = x + y if self.apply_residuals else y
        return x


class Seq2SeqLSTMNetwork(ARMWithHidden, nn.Module):
    @dtc.dataclass
    class Config(NetworkConfig):
        io_spec: IOSpec = None
        model_dim: int = 1024
        enc_downsampling: DownSampling = "edge_sum"
        enc_n_lstm: int = 1
        enc_apply_residuals: bool = False
        enc_weight_norm: bool = False
        dec_upsampling: UpSampling = "linear_resample"
        dec_n_lstm: int = 1
        dec_apply_residuals: bool = False
        dec_weight_norm: bool = False
        hop: int = 8

    @classmethod
    def from_config(cls, cfg: "Seq2SeqLSTMNetwork.Config"):
        first = cfg.io_spec.inputs[0]
        if isinstance(first.elem_type, Continuous):
            input_dim, input_module = first.elem_type.size, sum   # continuous inputs are just added up (:202-204)
        else:
            input_dim = cfg.model_dim
            input_module = ZipReduceVariables(mode="sum", modules=[
                spec.module.copy().set(out_dim=cfg.model_dim).module() for spec in cfg.io_spec.inputs])
        enc = EncoderLSTM(downsampling=cfg.enc_downsampling, input_dim=input_dim, output_dim=cfg.model_dim,
                          num_layers=cfg.enc_n_lstm, weight_norm=cfg.enc_weight_norm, hop=cfg.hop,
                          apply_residuals=cfg.enc_apply_residuals)
        # (the reference passes dec_apply_residuals as the decoder's weight_norm flag, :221)
        dec = DecoderLSTM(upsampling=cfg.dec_upsampling, model_dim=cfg.model_dim, num_layers=cfg.dec_n_lstm,
                          hop=cfg.hop, apply_residuals=cfg.dec_apply_residuals, weight_norm=cfg.dec_apply_residuals)
        heads = [spec.module.copy().set(in_dim=cfg.model_dim).module() for spec in cfg.io_spec.targets]
        return cls(cfg, input_module=input_module, output_module=ZipReduceVariables(mode="sum", modules=heads),
                   encoder=enc, decoder=dec)

    def __init__(self, config, input_module, output_module, encoder: EncoderLSTM, decoder: DecoderLSTM):
        super().__init__()
        self._config = config
        self.input_module = input_module
        self.enc = encoder
        self.dec = decoder
        self.output_module = output_module
        self.output_length = lambda n: n
        self._plan = None
        self._plan_batch = 0
        self._weights = native.WeightsTracker()
        self.exec_tuning = {}   # execution switches of THIS network's plans ({"MMK_...": "0"}: include/mmk.h `tuning`); merged over native.PLAN_TUNING
        self._plan_tuning = None            # the tuning text the plan at hand was built with

    # -- ARM properties -----------------------------------------------------------
    @property
    def config(self) -> NetworkConfig:
        return self._config

    @property
    def rf(self):
        return self._config.hop

    @property
    def generate_params(self) -> Set[str]:
        return {p for m in getattr(self.output_module, "heads", []) for p in getattr(m, "sampling_params", {})}

    def train_batch(self, item_spec: ItemSpec):
        hop = self._config.hop
        return tuple(
            spec.to_batch_item(ItemSpec(shift=0, length=hop, unit=item_spec.unit)) for spec in self.config.io_spec.inputs
        ), tuple(
            spec.to_batch_item(ItemSpec(shift=hop, length=hop, unit=item_spec.unit)) for spec in self.config.io_spec.targets
        )

    def test_batch(self, item_spec: ItemSpec):
        return tuple(spec.to_batch_item(item_spec) for spec in self.config.io_spec.inputs), ()

    # -- forward ------------------------------------------------------------------------
    def _forward_autograd(self, x: Tuple, temperature=None):
        x = self.input_module(x)
        coded, (h_enc, c_enc) = self.enc(x)
        out = self.dec(coded, (h_enc, c_enc))
        return self.output_module((out,), *((temperature,) if temperature is not None else ()))

    def forward(self, x: Tuple, temperature=None):
        if self.training:
            return self._forward_autograd(x, temperature)
        return self._device_step(tuple(x), temperature)

    # -- HIP plan ---------------------------------------------------------------------
    def _describe(self, max_batch: int) -> native.S2SConfig:
        cfg = self._config
        unsupported = []
        c = native.S2SConfig()
        discrete = self.input_module is not sum
        if discrete:
            # class indices through nn.Embedding under ZipReduceVariables (:205-210); the loop feeds the head's classes back
            first = list(self.input_module.heads)[0] if len(self.input_module.heads) == 1 else None
            table = first[0] if isinstance(first, nn.Sequential) and len(first) == 1 else first
            if not isinstance(table, nn.Embedding) or table.padding_idx is not None or table.max_norm is not None:
                unsupported.append("discrete inputs through a module other than one plain nn.Embedding")
            else:
                c.in_classes = table.num_embeddings
        pooling = {"edge_sum": 0, "edge_mean": 1, "sum": 2, "mean": 3, "linear_resample": 4}
        upsampling = {"linear_resample": 0, "repeat": 1, "interp": 2}
        if str(cfg.enc_downsampling) not in pooling:
            unsupported.append(f"enc_downsampling='{cfg.enc_downsampling}'")
        if str(cfg.dec_upsampling) not in upsampling:
            unsupported.append(f"dec_upsampling='{cfg.dec_upsampling}'")
        if not (1 <= cfg.enc_n_lstm <= 8 and 1 <= cfg.dec_n_lstm <= 8):
            unsupported.append("more than 8 LSTMs per side")
        heads = list(self.output_module.heads)
        if len(heads) != 1 or (discrete and len(cfg.io_spec.inputs) != 1):
            # (several continuous inputs are just added up, `input_module = sum`, :202-204: done in front of the plan)
            unsupported.append("more than one target, or more than one discrete input")
        elif discrete:
            head = heads[0]
            est = getattr(head, "estimator", None)
            mlp = est[0] if native.only_mlp(est) else None
            if not isinstance(mlp, MLP) or not isinstance(getattr(head, "sampler", None), CategoricalSampler):
                unsupported.append("discrete inputs with a head other than MLPIO + CategoricalSampler")
            elif native.mlp_head_problem(mlp, self.training) or mlp.n_hidden_layers > 4:
                unsupported.append(native.mlp_head_problem(mlp, self.training) or "MLP head with more than 4 hidden blocks")
            else:
                c.head_kind, c.mlp_hidden, c.mlp_n_hidden, c.mlp_act = 1, mlp.hidden_dim, mlp.n_hidden_layers, native.mlp_act(mlp)
                c.learn_temp = int(mlp.learn_temperature)
                c.min_temp = float(mlp.min_temp) if mlp.learn_temperature else 0.
                c.out_dim = mlp.out_dim - int(mlp.learn_temperature)
        else:
            head = heads[0]
            lin = head[0] if isinstance(head, nn.Sequential) else None
            tail = [m for m in list(head)[1:] if not (isinstance(m, Chunk) and m.chunks == 1)] if lin is not None else []
            kinds = [type(m).__name__ for m in tail]
            if not isinstance(lin, nn.Linear) or lin.bias is None or kinds not in ([], ["Abs"]):
                unsupported.append("output module other than (Chunked)LinearIO [+ Abs]")
            else:
                c.out_dim, c.out_abs = lin.out_features, int(kinds == ["Abs"])
        if unsupported:
            raise NotImplementedError("the HIP generate path does not cover: " + "; ".join(unsupported))
        c.in_dim = self.enc.input_dim
        c.model_dim, c.hop = cfg.model_dim, cfg.hop
        c.enc_n_lstm, c.dec_n_lstm = cfg.enc_n_lstm, cfg.dec_n_lstm
        c.enc_downsampling, c.dec_upsampling = pooling[str(cfg.enc_downsampling)], upsampling[str(cfg.dec_upsampling)]
        c.enc_apply_residuals, c.dec_apply_residuals = int(cfg.enc_apply_residuals), int(cfg.dec_apply_residuals)
        c.max_batch = max_batch
        c.exec_mode = int(self._exec_mode)
        c.tuning = native.tuning_text(native.PLAN_TUNING, self.exec_tuning)       # execution switches of this plan (never the environment)
        return c

    _blocks = ()            # generate_block calls since before_generate: (tensor, t0, n_steps)
    _exec_mode = 0          # 1 while a call is being redone with one launch per frame (mmk_s2s_config.exec_mode)
    _resident_seen = 0
    _plan_stale = False     # the plan is the one-launch-per-frame plan of a repeated call: replaced at the next _ensure_plan

    def _ensure_plan(self, batch: int, refresh_weights: bool):
        device = self.device
        if device.type != "cuda":
            raise RuntimeError("Seq2SeqLSTMNetwork generates on the MI355X only: move the network to the HIP device "
                               "('cuda'); there is no CPU implementation in this package")
        rebuilt = False
        tuning = native.tuning_text(native.PLAN_TUNING, self.exec_tuning)
        if self._plan is None or self._plan_tuning != tuning or self._plan_batch < batch or self._plan.device != device or (self._plan_stale and self._exec_mode == 0):
            self._plan_stale = False
            self._plan = native.S2SPlan(self._describe(max(batch, 1)), device)
            self._plan_batch = max(batch, 1)
            self._plan_tuning = tuning
            self._resident_seen = 0
            rebuilt = True
        # every call: a step keeps no state between calls, but the plan holds a re-packed copy of the weights, and eval
        # forward / generate_step may follow training steps or a load_state_dict at any time (per-epoch validation)
        # (the content fingerprint - a device reduction and a read-back - only where a generation starts; the steps of one compare
        # the host-side identity, which training steps and load_state_dict change)
        if rebuilt or self._weights.changed(self, content=refresh_weights):
            sd = self.state_dict()
            for k, head in enumerate(getattr(self.output_module, "heads", [])):      # (a head with dropout modules between its Linears: the plan knows `fc.{2 i}`)
                est = getattr(head, "estimator", None)
                if native.only_mlp(est):
                    sd = native.mlp_linear_keys(sd, f"output_module.heads.{k}.estimator.0.", est[0])
            self._plan.bind_state_dict(fold_weight_norm(sd) if any(k.endswith("_g") for k in sd) else sd)
            self._plan.commit()
            self._weights.committed(self)

    def _checked(self, run, rerun=None):
        """``run()`` on the plan (``rerun`` if what has to be repeated differs); when a wait inside the resident bi-LSTM kernel (csrc/lstm_seq.hip: one launch per layer, its
        workgroups wait for each other) timed out - the CUs were held by something else - once more with one launch per frame"""
        out = run()
        if self._plan.resident_launches() == self._resident_seen:
            return out
        self._resident_seen = self._plan.resident_launches()
        try:
            self._plan.sync_status()
            return out
        except native.NativeError as err:
            if self._exec_mode == 1:
                raise
            import warnings
            warnings.warn(f"{err}; repeating the call with one bi-LSTM launch per frame")
            self._exec_mode = 1
            try:
                self._plan = None
                self._ensure_plan(self._plan_batch, refresh_weights=False)
                out = (rerun or run)()
                torch.cuda.synchronize(self.device)
                return out
            finally:
                self._exec_mode = 0
                # the one-launch-per-frame plan stays until the caller has read what the repeated call left in it (last_logits of a
                # sampled step); the next _ensure_plan replaces it, so that the next call runs resident again
                self._plan_stale = True

    def _device_step(self, inputs: Tuple[torch.Tensor, ...], temperature=None):
        native.require_device(*inputs)
        if self.input_module is not sum:
            x = inputs[0]
            if x.size(1) != self._config.hop:
                raise AssertionError(f"expected {self._config.hop} input classes, got {x.size(1)}")
            self._ensure_plan(x.size(0), refresh_weights=False)
            y = self._checked(lambda: self._plan.step_classes(x.long()))
            if temperature is not None:
                # an eval-mode forward with a temperature (decode hands it to the sampler, :250-253; generate_step never does): one draw
                # per (clip, position) from softmax(logits / T) - the reference's torch.multinomial stream cannot be reproduced, the HIP
                # sampler inverts the CDF of the same distribution at uniforms drawn here
                c = self._plan.cfg
                raw = self._plan.last_logits(x.size(0))
                rows = x.size(0) * c.hop
                t = per_row_temperature(temperature, x.size(0), x.device).repeat_interleave(c.hop)
                u = torch.rand(rows, device=x.device, dtype=torch.float32)
                y = native.categorical_sample(raw, c.out_dim, bool(c.learn_temp), float(c.min_temp), t, u).reshape(x.size(0), c.hop)
            # the reference returns classes * w with w the float weight of ZipReduceVariables: a float tensor (modules/io.py:310)
            return y.to(torch.float32)
        x = inputs[0] if len(inputs) == 1 else sum(inputs)
        if x.size(1) != self._config.hop:
            raise AssertionError(f"expected {self._config.hop} input frames, got {x.size(1)}")
        self._ensure_plan(x.size(0), refresh_weights=False)
        x = x.float() if x.dtype != torch.float32 else x
        return self._checked(lambda: self._plan.step(x if x.stride(2) == 1 else x.contiguous()))

    # -- ARM generation protocol ------------------------------------------------------
    def reset_hidden(self):
        self.enc.hidden = [None] * self._config.enc_n_lstm
        self.dec.hidden = [None] * self._config.dec_n_lstm

    def before_generate(self, prompts: Tuple[torch.Tensor, ...], batch_index) -> None:
        self.reset_hidden()
        native.require_device(*tuple(prompts))
        self._blocks = []
        self._ensure_plan(prompts[0].size(0), refresh_weights=True)

    def generate_step(self, inputs: Tuple[torch.Tensor, ...], *, t: int = 0, **parameters):
        return self._device_step(tuple(inputs))

    def generate_block(self, tensors: Tuple[torch.Tensor, ...], t0: int, n_steps: int, **parameters):
        tensors = tuple(tensors)
        native.require_device(*tensors)
        if len(tensors) != 1:
            return None
        frames = tensors[0]
        if self.input_module is not sum:
            if frames.dtype != torch.int64 or frames.dim() != 2:
                return None
            self._ensure_plan(frames.size(0), refresh_weights=False)
            self._plan.generate_classes(frames, t0, n_steps)
            self._blocks = list(self._blocks) + [(frames, t0, n_steps)]
            return True
        if frames.dtype != torch.float32:
            return None
        self._ensure_plan(frames.size(0), refresh_weights=False)
        self._plan.generate(frames, t0, n_steps)
        self._blocks = list(self._blocks) + [(frames, t0, n_steps)]
        return True

    def after_generate(self, final_outputs: Tuple[torch.Tensor, ...], batch_index) -> None:
        blocks, self._blocks = self._blocks, []
        if blocks and self._plan is not None:
            # the blocks of this generation wrote their frames in place, each from the frames before its t0: after a reported
            # time-out they are run again, in order, with one launch per frame
            def again():
                for frames, t0, n_steps in blocks:
                    (self._plan.generate_classes if frames.dtype == torch.int64 else self._plan.generate)(frames, t0, n_steps)
            self._checked(lambda: None, again)
        self.reset_hidden()
