"""Auto-regressive generation driver.

Surface and semantics follow the reference's ``mimikit/loops/generate.py``
(``GenerateLoopV2`` :85-252, ``fill`` :50-73, ``prepare_prompt`` :26-39): prompts
are extended with ``n_steps`` blank positions, the network is asked for one step
at a time over windows ``tensor[:, t-rf:t]`` and every output is written in place
at ``t``.  Differences in HOW:

* if the network implements ``generate_block`` (all networks of this package do)
  the whole ``for t`` loop of a batch is handed to the device in ONE call
  (hipGraph-replayed step kernels, no per-step host work); networks that only
  implement ``generate_step`` are driven exactly as in the reference;
* the dataloader may be any iterable of ``[prompt_idx, *prompts]`` (the form the
  reference uses in ``models/ensemble_generator.py:132-139``); ``get_dataloader``
  serves prompts from in-memory arrays instead of an h5mapper file and applies the
  feature transforms on the device.
"""
import dataclasses as dtc
from typing import Any, Callable, Dict, Iterable, Optional, Tuple, Union

import numpy as np
import torch
from typing_extensions import Literal

from ..config import Config
from ..features.item_spec import Frame, ItemSpec, Sample, Second, convert
from ..networks.arm import ARM
from ..utils import default_device

__all__ = ["GenerateLoopV2", "prepare_prompt", "fill", "process_batch"]


def process_batch(batch, test: Callable[[Any], bool], func: Callable[[Any], Any]):
    """apply ``func`` to every leaf of a nested tuple/list/dict for which ``test`` holds"""
    if test(batch):
        return func(batch)
    if isinstance(batch, (tuple, list)):
        return type(batch)(process_batch(b, test, func) for b in batch)
    if isinstance(batch, dict):
        return {k: process_batch(v, test, func) for k, v in batch.items()}
    return batch


def prepare_prompt(device, prompt, n_blanks, at_least_nd=2):
    def one(p):
        if isinstance(p, np.ndarray):
            p = torch.from_numpy(p)
        while p.dim() < at_least_nd:
            p = p.unsqueeze(0)
        p = p.to(device)
        if n_blanks > 0:
            blank = torch.zeros(p.size(0), n_blanks, *p.size()[2:]).to(p)
            return torch.cat((p, blank), dim=1)
        return p

    return process_batch(prompt, lambda x: isinstance(x, (np.ndarray, torch.Tensor)), one)


FillType = Union[Literal["blank", "data"], torch.Tensor]


def fill(x: Optional[torch.Tensor], prior_t: Tuple[FillType, int], n_steps: Tuple[FillType, int]):
    """concatenate, along time, the data and the requested fills: ``"data"`` adds nothing,
    ``"blank"`` adds zeros, a (B,) tensor adds its value repeated"""
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    if x is None:
        parts, dtype, device, batch, feat = [], torch.float32, "cpu", 1, (1,)
    else:
        parts, dtype, device, batch, feat = [x], x.dtype, x.device, x.size(0), tuple(x.shape[2:])
    for kind, n in (prior_t, n_steps):
        if isinstance(kind, torch.Tensor):
            assert kind.shape == (batch,)
            parts.append(kind.expand(batch, n, 1))
        elif kind == "blank":
            parts.append(torch.zeros(batch, n, *feat, dtype=dtype, device=device))
    return torch.cat(parts, dim=1)


class _ArrayPromptLoader:
    """Serves ``[prompt_idx, *prompts]`` batches cut from in-memory feature arrays."""

    def __init__(self, dataset, items, positions, max_i, batch_size, device, downsampling):
        self.dataset, self.items, self.positions = dataset, items, positions
        self.max_i, self.batch_size, self.device, self.downsampling = max_i, batch_size, device, downsampling
        self.sampler = None

    def _source(self, name):
        if isinstance(self.dataset, dict):
            return self.dataset[name]
        return getattr(self.dataset, name)

    def __iter__(self):
        # reference get_dataloader (:113-139): fixed positions as given, None = a random position on the dataset's
        # down-sampling stride, drawn again for every pass (IndicesSampler, loops/samplers.py:50-81)
        if self.sampler is None:
            from .samplers import IndicesSampler
            self.sampler = IndicesSampler(N=len(self.positions), indices=tuple(self.positions), max_i=max(self.max_i, 1),
                                          redraw=True, sampling_stride=self.downsampling)
        where = [int(i) for i in self.sampler]
        for start in range(0, len(where), self.batch_size):
            chunk = where[start:start + self.batch_size]
            feats = []
            for item in self.items:
                data = self._source(item.data)
                rows = []
                for i in chunk:
                    lo = i + item.shift
                    rows.append(np.asarray(data[lo:lo + item.length:item.downsampling]))
                x = torch.from_numpy(np.stack(rows)).to(self.device)
                feats.append(item.transform(x) if item.transform is not None else x)
            yield [np.asarray(chunk, dtype=np.int32), *feats]


def _same_device(a, b) -> bool:
    a, b = torch.device(a), torch.device(b)
    if a.type != b.type:
        return False
    if a.type != "cuda":
        return True
    cur = torch.cuda.current_device()
    return (cur if a.index is None else a.index) == (cur if b.index is None else b.index)


class GenerateLoopV2:
    @dtc.dataclass
    class Config(Config):
        output_duration_sec: float = 1.
        prompts_length_sec: float = 1.
        prompts_position_sec: Tuple[Optional[float], ...] = (None,)  # random if None
        parameters: Optional[Dict[str, Any]] = None
        batch_size: int = 1
        downsampling: int = 1
        output_name_template: Optional[str] = None
        display_waveform: bool = True
        write_waveform: bool = False
        yield_inversed_outputs: bool = True
        callback: Optional[Callable[[Tuple[torch.Tensor, ...]], None]] = None

    @classmethod
    def get_n_steps(cls, config: "GenerateLoopV2.Config", network: ARM) -> int:
        io_spec = network.config.io_spec
        n_samples = int(io_spec.sr * config.output_duration_sec)
        unit = io_spec.unit
        if isinstance(unit, Frame):
            return convert(n_samples, Sample(1), unit, as_length=True) + 1
        return n_samples

    @classmethod
    def get_dataloader(cls, config: "GenerateLoopV2.Config", dataset, network: ARM):
        """``dataset``: mapping or object exposing one array per extractor name (e.g. ``.signal``)"""
        sr = network.config.io_spec.sr
        prompt_n_samples = int(sr * config.prompts_length_sec)
        signal = dataset["signal"] if isinstance(dataset, dict) else dataset.signal
        max_i = signal.shape[0] - prompt_n_samples
        prompt_items, _ = network.test_batch(ItemSpec(0, length=config.prompts_length_sec, unit=Second(sr)))
        positions = tuple(int(p * sr) if p is not None else None for p in config.prompts_position_sec)
        return _ArrayPromptLoader(dataset, prompt_items, positions, max_i, config.batch_size, default_device(),
                                  config.downsampling)

    @classmethod
    def from_config(cls, config: "GenerateLoopV2.Config", dataset, network: ARM, logger=None):
        return cls(config, network, cls.get_n_steps(config, network), cls.get_dataloader(config, dataset, network),
                   logger)

    def __init__(self, config: "GenerateLoopV2.Config", network: ARM, n_steps: int, dataloader: Iterable,
                 logger=None):
        self.config = config
        self.network = network
        self.n_steps = n_steps
        self.dataloader = dataloader
        self.logger = logger
        self._initial_device = None
        self._was_training = False
        self.device = None
        self.template_vars = {}

    def setup(self):
        net = self.network
        self._initial_device = net.device
        self._was_training = net.training
        net.eval()
        self.device = default_device()
        self._moved = not _same_device(self._initial_device, self.device)
        if self._moved:
            net.to(self.device)     # (an unconditional .to() re-flattens nn.RNN weights into new storage every time)
        torch.set_grad_enabled(False)

    def teardown(self):
        if self._moved:
            self.network.to(self._initial_device)
        if self._was_training:
            self.network.train()
        torch.set_grad_enabled(True)

    def run(self):
        self.setup()
        net = self.network
        for batch in self.dataloader:
            prompt_idx, batch = batch[0], batch[1:]
            batch = tuple((torch.from_numpy(x) if isinstance(x, np.ndarray) else x).to(self.device) for x in batch)
            net.before_generate(batch, prompt_idx)
            rf, prior_t, n_steps = net.rf, batch[0].size(1), self.n_steps
            tensors = tuple(fill(x, prior_t=("data", prior_t), n_steps=("blank", n_steps)) for x in batch)
            params = self.config.parameters or {}
            params = {k: v for k, v in params.items() if k in net.generate_params}

            handled = net.generate_block(tensors, prior_t, n_steps, **params) if hasattr(net, "generate_block") else None
            if not handled:
                until = 0
                for t in range(prior_t, prior_t + n_steps):
                    if t < until:
                        continue   # covered by a multi-step output (e.g. Seq2Seq hop)
                    outputs = net.generate_step(tuple(x[:, t - rf:t] for x in tensors), t=t, **params)
                    if not isinstance(outputs, tuple):
                        outputs = (outputs,)
                    for tensor, out in zip(tensors, outputs):
                        if out is not None:   # a net may skip a step
                            n_out = min(out.size(1), tensor.size(1) - t)
                            tensor.data[:, t:t + n_out] = out[:, :n_out]
                            until = t + n_out

            final_outputs = tuple(x.data for x in tensors)
            net.after_generate(final_outputs, prompt_idx)
            final_outputs = self.process_outputs(final_outputs, prompt_idx, **self.template_vars)
            yield final_outputs
            if self.config.callback is not None:
                self.config.callback(final_outputs)
        self.teardown()

    def process_outputs(self, final_outputs: Tuple[torch.Tensor, ...], prompt_idx, **template_vars):
        wants_audio = self.logger is not None and (self.config.write_waveform or self.config.display_waveform)
        if not wants_audio and not self.config.yield_inversed_outputs:
            return final_outputs
        features = self.network.config.io_spec.targets
        outputs = tuple(feature.inv(out) for feature, out in zip(features, final_outputs))
        if wants_audio:
            for output in outputs:
                for example, idx in zip(output, prompt_idx):
                    idx = idx.item() if hasattr(idx, "item") else idx
                    if self.config.write_waveform:
                        self.logger.write(example, prompt_idx=idx, **template_vars)
                    if self.config.display_waveform:
                        self.logger.display(example, prompt_idx=idx, **template_vars)
        return outputs if self.config.yield_inversed_outputs else final_outputs
