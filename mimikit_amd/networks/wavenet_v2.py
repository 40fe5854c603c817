"""WaveNet behind the ARM protocol, generating on the MI355X.

Config fields, layer wiring and ``state_dict`` layout follow the reference
(``mimikit/networks/wavenet_v2.py``: ``WNLayer`` :32-182, ``WaveNet`` :185-469).

Generation differs in HOW, not WHAT: the reference's ``generate_step`` is a full
forward over the rf-long window (its "fast generate" hooks are dead code,
SURVEY.md section 0), i.e. O(rf * L) layer evaluations per sample.  Here every layer keeps
a dilation queue of its past inputs in HBM (``csrc/wavenet_plan.hip``) and a step
costs L fused layer kernels; results are the same arithmetic on the same
operands.  ``eval()`` forward / ``generate_step`` / ``generate_block`` run only on
the HIP device through ``libmmk_hip.so``; training-mode ``forward`` is the stock
differentiable torch graph (training is out of this package's scope and is
kept only so the network still drops into the reference's trainer).
"""
import dataclasses as dtc
import operator
from itertools import accumulate, chain
from typing import Dict, Iterable, List, Optional, Set, Tuple

import torch
import torch.nn as nn

from .. import native
from ..features.item_spec import ItemSpec, Step
from ..io_spec import IOSpec
from ..modules.activations import ActivationConfig, ActivationEnum
from ..modules.misc import CausalPad, Chunk, Transpose
from ..modules.mlp import MLP
from ..modules.targets import OutputWrapper, per_row_temperature
from .arm import ARM, NetworkConfig

__all__ = ["WNLayer", "WaveNet"]


class ParametrizedLinear(nn.Module):
    """x_hat * a + b of one (1x1) convolution / Linear to three times the width (networks/parametrized.py:34-47); the
    parameter lives in ``params`` as in the reference (state_dict key ``...aff_res.params.weight``)."""

    def __init__(self, in_dim, out_dim, bias=True, as_1x1_conv=False):
        super().__init__()
        self.params = nn.Conv1d(in_dim, out_dim * 3, 1, bias=bias) if as_1x1_conv else nn.Linear(in_dim, out_dim * 3, bias=bias)
        self.chunk_dim = -2 if as_1x1_conv else -1

    def forward(self, x):
        x_hat, a, b = torch.chunk(self.params(x), 3, dim=self.chunk_dim)
        return x_hat.mul(a).add(b)


class WNLayer(nn.Module):
    """One dilated causal conv layer with gated units, 1x1 conditioning, skip and residual
    connections.  Holds the parameters (reference names) and the differentiable forward."""

    def __init__(self, input_dim: Optional[int] = None, dims_dilated: Tuple[int] = (128,),
                 dims_1x1: Tuple[int] = tuple(), residuals_dim: Optional[int] = None,
                 apply_residuals: bool = False, skips_dim: Optional[int] = None, kernel_size: int = 2,
                 groups: int = 1, act_f: nn.Module = nn.Tanh(), act_g: Optional[nn.Module] = nn.Sigmoid(),
                 pad_side: int = 1, stride: int = 1, bias: bool = True, dilation: int = 1,
                 with_affine_residuals: bool = False):
        super().__init__()
        self.input_dim, self.dims_dilated, self.dims_1x1 = input_dim, dims_dilated, dims_1x1
        self.residuals_dim, self.apply_residuals, self.skips_dim = residuals_dim, apply_residuals, skips_dim
        self.kernel_size, self.groups, self.act_f, self.act_g = kernel_size, groups, act_f, act_g
        self.pad_side, self.stride, self.bias, self.dilation = pad_side, stride, bias, dilation

        self.cause = (kernel_size - 1) * dilation
        self.needs_padding = pad_side != 0
        self.has_gated_units = act_g is not None
        self.has_skips = skips_dim is not None
        self.has_residuals = residuals_dim is not None and (input_dim is None or input_dim == residuals_dim)
        self.has_affine_residuals = with_affine_residuals

        inner = dims_dilated[0]
        outer = inner if residuals_dim is None else residuals_dim
        in_dim = outer if input_dim is None else input_dim
        kw_dil = dict(kernel_size=(kernel_size,), dilation=dilation, stride=stride, bias=bias, groups=groups)
        kw_1x1 = dict(kernel_size=(1,), stride=stride, bias=bias)
        if self.needs_padding:
            self.pad = CausalPad((0, 0, pad_side * self.cause))
        if self.has_gated_units:
            def gated(i, o, **kw):
                return nn.Sequential(nn.Conv1d(i, o * 2, **kw), Chunk(2, dim=1, sum_outputs=False))

            self.conv_dil = nn.ModuleList([gated(in_dim, d, **kw_dil) for d in dims_dilated])
            self.conv_1x1 = nn.ModuleList([gated(d, inner, **kw_1x1) for d in dims_1x1])
        else:
            self.conv_dil = nn.ModuleList([nn.Conv1d(in_dim, d, **kw_dil) for d in dims_dilated])
            self.conv_1x1 = nn.ModuleList([nn.Conv1d(d, inner, **kw_1x1) for d in dims_1x1])
        if self.has_skips:
            self.conv_skip = nn.Conv1d(inner, skips_dim, **kw_1x1)
        if self.has_residuals:
            self.conv_res = nn.Conv1d(inner, outer, **kw_1x1)
        if self.has_affine_residuals:      # (wavenet_v2.py:121-122)
            self.aff_res = ParametrizedLinear(in_dim, in_dim, as_1x1_conv=True)

    def trim_cause(self, x):
        cs = self.cause
        return x[:, :, cs:] if self.pad_side >= 0 else x[:, :, :-cs]

    def forward(self, inputs_dilated, inputs_1x1, skips=None):
        x = inputs_dilated[0]
        if self.needs_padding:
            x = self.pad(x)
        conds = [c if self.needs_padding else self.trim_cause(c) for c in inputs_1x1]
        if self.has_affine_residuals:      # the dilated convolution AND the residual sum see aff(x) (:148-149, :160-161, :174)
            if not self.has_gated_units:
                conds = [self.aff_res(c) + c for c in conds]     # (:157-158)
            x = self.aff_res(x)
        if self.has_gated_units:
            z_f, z_g = self.conv_dil[0](x)
            for conv, c in zip(self.conv_1x1, conds):
                c_f, c_g = conv(c)
                z_f, z_g = z_f + c_f, z_g + c_g
            y = self.act_f(z_f) * self.act_g(z_g)
        else:
            z = self.conv_dil[0](x)
            for conv, c in zip(self.conv_1x1, conds):
                z = z + conv(c)
            y = self.act_f(z)
        if self.has_skips:
            if skips is not None and not self.needs_padding:
                skips = self.trim_cause(skips)
            skips = self.conv_skip(y) if skips is None else self.conv_skip(y) + skips
        if self.has_residuals:
            y = self.trim_cause(x) + self.conv_res(y)
        return y, skips


class WaveNet(ARM, nn.Module):
    @dtc.dataclass
    class Config(NetworkConfig):
        io_spec: IOSpec = None
        kernel_sizes: Tuple[int, ...] = (2,)
        blocks: Tuple[int, ...] = (4,)
        dims_dilated: Tuple[int, ...] = (128,)
        dims_1x1: Tuple[int, ...] = ()
        residuals_dim: Optional[int] = None
        apply_residuals: bool = False
        skips_dim: Optional[int] = None
        with_affine_residuals: bool = False
        groups: int = 1
        act_f: ActivationEnum = "Tanh"
        act_g: Optional[ActivationEnum] = "Sigmoid"
        pad_side: int = 0
        stride: int = 1
        bias: bool = True
        use_fast_generate: bool = False
        tie_io_weights: bool = False
        layerwise_inputs: bool = False
        reverse_layer_order: bool = False

    # -- construction -----------------------------------------------------------
    @classmethod
    def get_kernels_and_dilation(cls, kernel_sizes, blocks):
        """(kernel size, dilation) per layer from the (kernel_sizes, blocks) shorthand
        (reference :295-327): one kernel + n blocks, one block pattern repeated, or explicit."""
        kernel_sizes, blocks = tuple(kernel_sizes), tuple(blocks)
        if not blocks:
            return list(kernel_sizes), list(accumulate([1, *kernel_sizes], operator.mul))
        if len(set(blocks)) == 1 and blocks[0] == len(kernel_sizes):
            one = list(accumulate([1, *kernel_sizes[:-1]], operator.mul))
            return list(kernel_sizes) * len(blocks), one * len(blocks)
        if len(kernel_sizes) == sum(blocks):
            dil, start = [], 0
            for b in blocks:
                dil += list(accumulate([1, *kernel_sizes[start:start + b - 1]], operator.mul))
                start += b
            return list(kernel_sizes), dil
        if len(kernel_sizes) == 1:
            k = kernel_sizes[0]
            return [k] * sum(blocks), [k ** i for b in blocks for i in range(b)]
        raise ValueError("number of layers and number of kernel sizes not compatible."
                         f" Got kernel_sizes={kernel_sizes} ; blocks={blocks}")

    @classmethod
    def get_layers(cls, config: "WaveNet.Config") -> List[WNLayer]:
        ks, ds = cls.get_kernels_and_dilation(config.kernel_sizes, config.blocks)
        ks, ds = list(ks), list(ds)
        n_layers = sum(config.blocks) if config.blocks else len(ks)
        layers = []
        for n, (k, d) in enumerate(zip(ks, ds)):
            layers.append(WNLayer(
                input_dim=config.dims_dilated[0], dims_dilated=config.dims_dilated, dims_1x1=config.dims_1x1,
                residuals_dim=config.residuals_dim if n != n_layers - 1 else None,  # no residuals for last layer
                apply_residuals=config.apply_residuals and n != 0, skips_dim=config.skips_dim, kernel_size=k,
                groups=config.groups, act_f=ActivationConfig(str(config.act_f)).get(),
                act_g=ActivationConfig(str(config.act_g)).get() if config.act_g is not None else None,
                pad_side=config.pad_side, stride=config.stride, bias=config.bias, dilation=d,
                with_affine_residuals=config.with_affine_residuals))
        return layers

    @classmethod
    def from_config(cls, config: "WaveNet.Config") -> "WaveNet":
        layers = cls.get_layers(config)
        hidden = [*config.dims_dilated, *config.dims_1x1]
        input_modules = [spec.module.copy().set(out_dim=h).module() for spec, h in zip(config.io_spec.inputs, hidden)]
        head_in = config.skips_dim if config.skips_dim is not None else hidden[0]
        output_modules = [spec.module.copy().set(in_dim=head_in).module() for spec in config.io_spec.targets]
        if config.tie_io_weights:
            # reference :240-249: every Linear of an input module hands its transposed weight to the same-named Linear of
            # the output module (an initialisation: the new Parameter is a copy, the state_dict layout does not change)
            for i_mod, o_mod in zip(input_modules, output_modules):
                for name, m in i_mod.named_modules():
                    if isinstance(m, nn.Linear):
                        try:
                            o_mod.get_submodule(name).weight = nn.Parameter(m.weight.transpose(0, 1))
                        except AttributeError:
                            continue
        return cls(config=config, layers=layers, input_modules=input_modules, output_modules=output_modules)

    def __init__(self, config: "WaveNet.Config", layers: List[WNLayer], input_modules: List[nn.Module],
                 output_modules: List[nn.Module]):
        super().__init__()
        self._config = config
        self.input_modules = nn.ModuleList(input_modules)
        self.transpose = Transpose(1, 2)
        self.layers: Iterable[WNLayer] = nn.ModuleList(reversed(layers) if config.reverse_layer_order else layers)
        self.has_skips = config.skips_dim is not None
        self.output_modules = nn.ModuleList(output_modules)
        self.eval_slice = slice(-1, None) if config.pad_side == 1 else slice(0, 1)
        self._plan: Optional[native.WaveNetPlan] = None
        self._plan_batch = 0
        self._weights = native.WeightsTracker()
        self.exec_tuning = {}   # execution switches of THIS network's plans ({"MMK_...": "0"}: include/mmk.h `tuning`); merged over native.PLAN_TUNING
        self._plan_tuning = None            # the tuning text the plan at hand was built with
        self._next_t: Optional[int] = None   # absolute time the queues are ready to produce
        self._state_batch = 0

    # -- ARM properties -----------------------------------------------------------
    @property
    def config(self) -> Config:
        return self._config

    @property
    def shift(self) -> int:
        return 1 if self.config.pad_side == 1 else self.rf

    @property
    def rf(self) -> int:
        return sum(layer.cause for layer in self.layers) + 1

    def output_length(self, n_input_steps: int) -> int:
        return n_input_steps if self.config.pad_side != 0 else n_input_steps - self.shift + 1

    @property
    def use_fast_generate(self):
        return self._config.use_fast_generate

    def train_batch(self, item_spec: ItemSpec):
        inputs = tuple(spec.to_batch_item(item_spec) for spec in self.config.io_spec.inputs)
        shifted = item_spec + ItemSpec(self.shift, self.output_length(0), unit=Step())
        return inputs, tuple(spec.to_batch_item(shifted) for spec in self.config.io_spec.targets)

    def test_batch(self, item_spec: ItemSpec):
        return self.train_batch(item_spec)

    @property
    def generate_params(self) -> Set[str]:
        # Reproduced quirk: the reference asks the ModuleList itself (wavenet_v2.py:364-366), which
        # has no `sampling_params`, so GenerateLoopV2 filters every parameter out and a WaveNet
        # always decodes greedily inside the loop.  `generate_step` / `generate_block` called
        # directly still honour `temperature=`, as in the reference's tests.
        return set(getattr(self.output_modules, "sampling_params", {}))

    # -- differentiable forward (training only) --------------------------------------
    def _forward_autograd(self, inputs: Tuple, **parameters):
        feats = tuple(self.transpose(mod(x)) for mod, x in zip(self.input_modules, inputs))
        dilated, in_1x1, skips = feats[0], feats[1:], None
        for layer in self.layers:
            dilated, skips = layer(inputs_dilated=(dilated,), inputs_1x1=in_1x1, skips=skips)
            if self._config.layerwise_inputs:
                dilated = dilated + feats[0][..., -dilated.size(-1):]
            if not layer.needs_padding:
                in_1x1 = tuple(layer.trim_cause(x) for x in in_1x1)
        y = self.transpose(skips if self.has_skips else dilated)
        return tuple(mod(y, **parameters) for mod in self.output_modules)

    def forward(self, inputs: Tuple, **parameters):
        if self.training:
            return self._forward_autograd(inputs, **parameters)
        # eval: one output step computed from the first rf positions (eval_slice, reference :273, :291-292)
        inputs = tuple(inputs)
        native.require_device(*inputs)
        rf = self.rf
        if self._config.pad_side == 1:
            # every layer pads its cause on the left and eval keeps the LAST position (reference :87-88, :273): with at
            # least rf steps the padding is never reached and that is the step computed from the last rf positions
            if inputs[0].size(1) < rf:
                raise NotImplementedError("pad_side=1 on a window shorter than the receptive field (zero-padded hidden "
                                          "states) is outside the HIP generate path")
            return self._window_step(tuple(x[:, -rf:] for x in inputs), t=inputs[0].size(1), **parameters)
        if inputs[0].size(1) < rf:
            raise RuntimeError(f"Calculated output size is too small: window of {inputs[0].size(1)} steps "
                               f"for a receptive field of {rf}")
        return self._window_step(tuple(x[:, :rf] for x in inputs), t=rf, **parameters)

    # -- HIP plan ---------------------------------------------------------------------
    _exec_mode = 0          # 1 while a batch is being redone on the per-layer launch path (mmk_wavenet_config.exec_mode)

    def _describe(self, max_batch: int) -> native.WaveNetConfig:
        cfg, io = self._config, self._config.io_spec
        unsupported = []
        if cfg.pad_side not in (0, 1):
            unsupported.append("pad_side other than 0 / 1")
        if cfg.groups < 1 or cfg.dims_dilated[0] % cfg.groups:
            unsupported.append(f"groups={cfg.groups} does not divide the dilated width")
        if cfg.stride != 1:
            unsupported.append("stride != 1")
        if cfg.with_affine_residuals and (cfg.pad_side != 0 or cfg.layerwise_inputs or (cfg.act_g is None and cfg.dims_1x1)):
            unsupported.append("with_affine_residuals together with pad_side, layerwise_inputs, or conditioning inputs of an ungated network")
        if str(cfg.act_f) not in native.ACT or (cfg.act_g is not None and str(cfg.act_g) not in native.ACT):
            unsupported.append("act_f / act_g other than " + " / ".join(k for k in native.ACT if isinstance(k, str) and k != "none"))
        if len(cfg.dims_dilated) != 1:
            unsupported.append("more than one dilated path")
        n_tgt = len(io.targets)
        if n_tgt > min(len(io.inputs), native.MAX_STREAMS):
            unsupported.append("more targets than inputs (the loop writes output k into input k, loops/generate.py:213-218) or than "
                               f"{native.MAX_STREAMS}")
        if len(cfg.dims_1x1) > native.MAX_COND or len(self.layers) > native.MAX_LAYERS:
            unsupported.append("too many conditioning inputs / layers")
        c = native.WaveNetConfig()
        c.n_layers = len(self.layers)
        c.res_explicit = 1           # per layer in RUN order (reverse_layer_order moves the residual-free layer to the front)
        for i, layer in enumerate(self.layers):
            c.kernel_size[i], c.dilation[i] = layer.kernel_size, layer.dilation
            c.layer_has_res[i] = int(layer.has_residuals)
        c.exec_mode = int(self._exec_mode)
        c.tuning = native.tuning_text(native.PLAN_TUNING, self.exec_tuning)       # execution switches of this plan (never the environment)
        c.layerwise_inputs = int(cfg.layerwise_inputs)
        c.with_affine_residuals = int(cfg.with_affine_residuals)
        first = self.input_modules[0][0]
        if isinstance(first, nn.Embedding) and len(self.input_modules[0]) == 1:
            c.q_levels, c.in_dim = first.num_embeddings, 0
        elif isinstance(first, nn.Linear) and first.bias is not None and all(
                isinstance(m, Chunk) and m.chunks == 1 for m in list(self.input_modules[0])[1:]):
            c.q_levels, c.in_dim = 0, first.in_features
        else:
            unsupported.append(f"input module 0 of type {type(first).__name__}")
        c.dim_dilated = cfg.dims_dilated[0]
        c.residuals_dim = cfg.residuals_dim or 0
        c.skips_dim = cfg.skips_dim or 0
        c.n_cond = len(cfg.dims_1x1)
        for j, mod in enumerate(self.input_modules[1:]):
            lin = mod[0]
            if isinstance(lin, nn.Embedding) and len(mod) == 1:       # a class stream through an EmbeddingIO: its row of the table per position
                c.cond_q_levels[j], c.cond_dim[j] = lin.num_embeddings, lin.embedding_dim
                continue
            if not isinstance(lin, nn.Linear) or lin.bias is None or not all(
                    isinstance(m, Chunk) and m.chunks == 1 for m in list(mod)[1:]):
                unsupported.append(f"conditioning input module {j + 1} is not a plain (Chunked)LinearIO")
                continue
            c.cond_in_dim[j], c.cond_dim[j] = lin.in_features, lin.out_features
        c.bias, c.gated = int(cfg.bias), int(cfg.act_g is not None)
        c.act_f, c.act_g = native.ACT.get(str(cfg.act_f), 0), native.ACT.get(str(cfg.act_g), 0) if cfg.act_g is not None else 0
        head = self.output_modules[0]
        if isinstance(head, OutputWrapper) and native.only_mlp(head.estimator):
            mlp: MLP = head.estimator[0]
            if native.mlp_head_problem(mlp, self.training):
                unsupported.append(native.mlp_head_problem(mlp, self.training))
            else:
                c.mlp_act = native.mlp_act(mlp)
            c.head_kind, c.mlp_hidden, c.mlp_n_hidden = 0, mlp.hidden_dim, mlp.n_hidden_layers
            c.learn_temp = int(mlp.learn_temperature)
            c.out_dim = mlp.out_dim - c.learn_temp
            c.min_temp = float(mlp.min_temp) if mlp.learn_temperature else 0.
            if mlp.n_hidden_layers > 4:
                unsupported.append("n_mlp_layers > 4")
        elif isinstance(head, nn.Sequential) and isinstance(head[0], nn.Linear) and head[0].bias is not None:
            tail = [m for m in list(head)[1:] if not (isinstance(m, Chunk) and m.chunks == 1)]
            kinds = [type(m).__name__ for m in tail]
            if kinds == ["Abs"]:
                c.head_kind = 1
            elif not kinds:
                c.head_kind = 2
            else:
                unsupported.append(f"linear head followed by {kinds}")
            c.out_dim = head[0].out_features
        else:
            unsupported.append(f"output module of type {type(head).__name__}")
        # further targets (:293: one output module per target on the same vector): MLP heads whose class goes into input k
        c.n_targets = n_tgt
        for k in range(1, min(n_tgt, native.MAX_STREAMS)):
            hk = self.output_modules[k]
            if not (isinstance(hk, OutputWrapper) and native.only_mlp(hk.estimator)) or c.head_kind != 0:
                unsupported.append(f"target {k}: several targets need MLP heads with samplers")
                continue
            mlp = hk.estimator[0]
            if native.mlp_head_problem(mlp, self.training) or mlp.n_hidden_layers > 4 or native.mlp_act(mlp) != c.mlp_act:
                unsupported.append(f"target {k}: {native.mlp_head_problem(mlp, self.training) or 'more than 4 hidden layers, or another activation than target 0'}")
            c.x_mlp_hidden[k], c.x_mlp_n_hidden[k], c.x_learn_temp[k] = mlp.hidden_dim, mlp.n_hidden_layers, int(mlp.learn_temperature)
            c.x_out_dim[k] = mlp.out_dim - c.x_learn_temp[k]
            c.x_min_temp[k] = float(mlp.min_temp) if mlp.learn_temperature else 0.
            if k - 1 < len(cfg.dims_1x1) and c.cond_q_levels[k - 1] < c.x_out_dim[k]:
                unsupported.append(f"target {k} draws classes that input {k} (not a class stream of as many classes) cannot take")
        c.max_batch = max_batch
        if unsupported:
            raise NotImplementedError("the HIP generate path does not cover: " + "; ".join(unsupported))
        return c

    def _ensure_plan(self, batch: int, refresh_weights: bool):
        device = self.device
        if device.type != "cuda":
            raise RuntimeError("WaveNet generates on the MI355X only: move the network to the HIP device ('cuda'); "
                               "there is no CPU implementation in this package")
        rebuilt = False
        tuning = native.tuning_text(native.PLAN_TUNING, self.exec_tuning)
        stale = self._plan is None or self._plan_tuning != tuning or self._plan_batch < batch or self._plan.device != device
        # the step kernel is chosen for the batch a plan is made for: one made for more than 128 clips serves groups of 16 clips per visit (~107 us per step
        # whatever the batch), which a later call of fewer clips must not pay - it gets a plan of its own size (the ring: ~1.1 us per clip, 0.8 from 60 clips on)
        if not stale and not native.wn_bpipe_by_default(batch) and getattr(self._plan, "batch_pipelined", False) and b"MMK_WN_BPIPE=1" not in tuning:
            stale = True
        if stale:
            self._plan = native.make_wavenet_plan(self._describe, max(batch, 1), device)
            self._plan_batch = max(batch, 1)
            self._plan_tuning = tuning
            rebuilt = True
        if rebuilt or refresh_weights:
            # the plan holds a re-packed copy of the weights: redo it only when a parameter changed since (optimiser step,
            # load_state_dict, .to(device)); the reference's before_generate never touches the weights either.  The queues
            # need no clearing: every slot a step reads is rewritten by the warm-up that precedes it.
            if rebuilt or self._weights.changed(self, content=True):
                self._plan.bind_state_dict(self._plan_tensors())
                self._plan.commit()
                self._weights.committed(self)
            self._next_t = None

    def _plan_tensors(self):
        """``state_dict`` as the plan binds it.  A grouped dilated convolution (``groups`` > 1, :93 of the reference) is
        handed over as the block-diagonal dense matrix it is: the kernels multiply whole channel tiles, and a zero
        weight adds exactly 0 to a sum, so the result is that of the grouped convolution."""
        sd = self.state_dict()
        for k, head in enumerate(self.output_modules):      # (a head with dropout modules between its Linears: the plans know `fc.{2 i}`)
            est = getattr(head, "estimator", None)
            if native.only_mlp(est):
                sd = native.mlp_linear_keys(sd, f"output_modules.{k}.estimator.0.", est[0])
        groups = self._config.groups
        if groups == 1:
            return sd
        out = {}
        for key, w in sd.items():
            if ".conv_dil." in key and key.endswith("weight"):
                n_out, in_g, k = w.shape
                dense = w.new_zeros(n_out, in_g * groups, k)
                og = n_out // groups
                for gi in range(groups):
                    dense[gi * og:(gi + 1) * og, gi * in_g:(gi + 1) * in_g] = w[gi * og:(gi + 1) * og]
                w = dense
            out[key] = w
        return out

    def _sampling(self, batch: int, n_steps: int, parameters: Dict):
        temperature = parameters.get("temperature", None)
        if temperature is None:
            return None, None
        if self._plan.cfg.head_kind != 0:
            return None, None
        t = per_row_temperature(temperature, batch, self.device)
        n_tgt = max(int(self._plan.cfg.n_targets), 1)
        u = torch.rand((batch, n_steps) if n_tgt == 1 else (n_tgt, batch, n_steps), device=self.device, dtype=torch.float32)
        return t, u

    @staticmethod
    def _time_major(x: torch.Tensor) -> torch.Tensor:
        """(batch, T[, dim]) with unit stride on dim and dim-stride on T; the batch stride is free,
        so windows of a longer tensor are used in place"""
        ok = x.stride(-1) == 1 and (x.dim() == 2 or x.stride(1) == x.shape[2])
        return x if ok else x.contiguous()

    def _prepare(self, tensors: Tuple[torch.Tensor, ...]):
        in0, cond = tensors[0], tuple(tensors[1:])
        if self._plan.cfg.q_levels == 0 and in0.dtype != torch.float32:
            in0 = in0.float()
        classes = self._plan.cfg.cond_q_levels
        cond = tuple(self._time_major((c if c.dtype == torch.int64 else c.long()) if classes[j] > 0 else
                                      (c if c.dtype == torch.float32 else c.float())) for j, c in enumerate(cond))
        return self._time_major(in0), cond

    def _with_blank(self, in0, cond):
        """scratch copies with one free column behind the window for the streams a target is written to"""
        n_tgt = max(int(self._plan.cfg.n_targets), 1)
        blank = lambda x: torch.cat([x, torch.zeros_like(x[:, :1])], dim=1).contiguous()
        return blank(in0), tuple(blank(c) if j + 1 < n_tgt else c for j, c in enumerate(cond))

    def _outputs(self, buf, cond, col: int):
        n_tgt = max(int(self._plan.cfg.n_targets), 1)
        return (buf[:, col:col + 1],) + tuple(cond[k - 1][:, col:col + 1] for k in range(1, n_tgt))

    def _window_step(self, window: Tuple[torch.Tensor, ...], t: int, **parameters):
        """rebuild the queues from an rf-long window ending at absolute time t, then produce step t"""
        batch, rf = window[0].size(0), self.rf
        self._ensure_plan(batch, refresh_weights=True)
        in0, cond = self._prepare(window)
        # scratch copy of the window with one free column for the produced step
        buf, cond = self._with_blank(in0, cond)
        t_first = t - rf
        self._plan.warmup(buf, cond, t_first, t - 1, t_first=t_first)
        temp, uni = self._sampling(batch, 1, parameters)
        self._plan.generate(buf, cond, t, 1, temp, uni, t_first=t_first)
        self._next_t, self._state_batch = t + 1, batch
        return self._outputs(buf, cond, rf)

    # -- ARM generation protocol ------------------------------------------------------
    def before_generate(self, prompts: Tuple[torch.Tensor, ...], batch_index) -> None:
        prompts = tuple(prompts)
        native.require_device(*prompts)
        batch, length, rf = prompts[0].size(0), prompts[0].size(1), self.rf
        self._blocks = []                       # the generate_block calls since this warm-up (replayed if the kernel reports a timeout)
        self._ensure_plan(batch, refresh_weights=True)
        if length < rf:
            # the reference fails at its first step on such a prompt (negative window start)
            self._next_t = None
            return
        in0, cond = self._prepare(prompts)
        # positions [P - rf, P - 1) fill the queues; position P - 1 is consumed by the first step
        self._plan.warmup(in0, cond, length - rf, length - 1, t_first=0)
        self._next_t, self._state_batch = length, batch

    def generate_step(self, inputs: Tuple[torch.Tensor, ...], *, t: int = 0, **parameters):
        inputs = tuple(inputs)
        native.require_device(*inputs)
        batch, rf = inputs[0].size(0), self.rf
        if inputs[0].size(1) < rf:
            raise RuntimeError(f"Calculated output size is too small: window of {inputs[0].size(1)} steps "
                               f"for a receptive field of {rf}")
        if self._plan is None or self._next_t != t or self._state_batch != batch:
            return self._window_step(tuple(x[:, -rf:] for x in inputs), t=t, **parameters)
        # queues are in sync: only the newest position (t - 1) is consumed
        in0, cond = self._prepare(tuple(x[:, -1:] for x in inputs))
        buf, cond = self._with_blank(in0, cond)
        temp, uni = self._sampling(batch, 1, parameters)
        self._plan.generate(buf, cond, t, 1, temp, uni, t_first=t - 1)
        self._next_t = t + 1
        return self._outputs(buf, cond, 1)

    def generate_block(self, tensors: Tuple[torch.Tensor, ...], t0: int, n_steps: int, **parameters):
        """all steps of one batch in one device call; ``tensors`` are the loop's (batch, prior+steps[, dim])
        tensors, target 0 is written in place into ``tensors[0]``"""
        tensors = tuple(tensors)
        native.require_device(*tensors)
        batch = tensors[0].size(0)
        if self._plan is None or self._next_t != t0 or self._state_batch != batch:
            self.before_generate(tuple(x[:, :t0] for x in tensors), None)
            if self._next_t != t0:
                raise RuntimeError(f"prompt of {t0} steps is shorter than the receptive field ({self.rf})")
        in0, cond = self._prepare(tensors)
        if in0.data_ptr() != tensors[0].data_ptr():
            raise TypeError("generate_block writes in place: tensors[0] must already have the network's input dtype")
        for k in range(1, max(int(self._plan.cfg.n_targets), 1)):
            if cond[k - 1].data_ptr() != tensors[k].data_ptr():
                raise TypeError(f"generate_block writes in place: tensors[{k}] must be int64 class indices, contiguous along time")
        temp, uni = self._sampling(batch, n_steps, parameters)
        self._plan.generate(in0, cond, t0, n_steps, temp, uni, t_first=0)
        self._next_t = t0 + n_steps
        if self._plan.persistent:
            getattr(self, "_blocks", []).append((tensors, t0, n_steps, dict(parameters)))
        return True

    def after_generate(self, final_outputs: Tuple[torch.Tensor, ...], batch_index) -> None:
        self._next_t = None
        if self._plan is not None and self._plan.persistent:
            # the loop reads the outputs right after this call anyway: a hand-off timeout of the persistent kernel (its
            # workgroups were not all resident - another kernel held CUs) must not return blanks.  The generation is redone
            # once on the per-layer launch path, which needs no co-residency; if that is not possible the error is raised.
            try:
                self._plan.sync_status()
                self._blocks = []               # (nothing to redo: drop the references to the loop's tensors)
            except native.NativeError as err:
                self._redo_on_launch_path(err)

    def _redo_on_launch_path(self, err):
        import warnings
        blocks, self._blocks = getattr(self, "_blocks", []), []
        if not blocks or self._exec_mode == 1:
            raise err
        warnings.warn(f"{err}; regenerating this batch on the per-layer launch path")
        self._exec_mode = 1                     # the next plan is created with exec_mode = 1: one fused kernel per layer half
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
            self._plan = None                   # the next generation gets a persistent plan again
            self._next_t = None
