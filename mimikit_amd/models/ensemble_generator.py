"""Long-form generation by a sequence of networks.

Behaviour of the reference's ``EnsembleGenerator`` (``mimikit/models/ensemble_generator.py:54-163``): a clip of
``max_seconds`` at ``base_sr`` starts with the caller's prompt; a stream of events ``{generator, seconds, temperature}``
extends it.  Every event sees the most recent ``len(prompt)`` samples of the clip, brought to its network's sample rate,
generates ``seconds`` of audio and hands it back at ``base_sr``.  The first event that does not fit before ``max_seconds``
ends the clip: what is left stays silent.  An exhausted stream raises ``StopIteration`` like the reference's ``next()``.

Here the whole event runs on the device: ``Resample`` is the HIP polyphase kernel (``csrc/features.hip``), the network's
input transforms and ``GenerateLoopV2`` are the package's own, and an event may name a network or a ``Checkpoint`` of this
package (``.npz`` weights + the reference's YAML config, see ``mimikit_amd/checkpoint.py``).
"""
import dataclasses as dtc
from typing import Dict, Iterator, Optional, Tuple, Union

import torch

from ..checkpoint import Checkpoint
from ..features.functionals import Resample
from ..features.item_spec import Sample, convert
from ..loops.generate import GenerateLoopV2
from ..networks.arm import ARM
from ..utils import default_device

__all__ = ["Event", "EnsembleGenerator"]


@dtc.dataclass
class Event:
    generator: Union[ARM, Checkpoint]
    seconds: float
    temperature: Optional[float] = None

    def network(self) -> ARM:
        g = self.generator
        if isinstance(g, Checkpoint):
            return g.network
        if isinstance(g, ARM):
            return g
        raise TypeError(f"event generator type '{type(g)}' not supported")

    def loop_parameters(self) -> Dict:
        return {} if self.temperature is None else {"temperature": self.temperature}


class EnsembleGenerator:
    def __init__(self, prompt: torch.Tensor, max_seconds: float = 10., base_sr: int = 22050, stream: Iterator[dict] = (),
                 print_events: bool = False, device=None):
        self.device = default_device() if device is None else device
        self.prompt = prompt.to(self.device)
        self.max_seconds = max_seconds
        self.base_sr = base_sr
        self.stream = stream
        self.print_events = print_events

    # -- the clip as a timeline ---------------------------------------------------------------------
    @property
    def total_samples(self) -> int:
        return int(self.max_seconds * self.base_sr)

    def run(self) -> torch.Tensor:
        context = self.prompt.size(-1)
        clip = self.prompt.new_zeros(self.prompt.size(0), self.total_samples)
        clip[:, :context] = self.prompt
        cursor = context
        while cursor < self.total_samples:
            piece = self.generate_step(cursor, clip[:, cursor - context:cursor])
            if piece is None:
                break
            room = self.total_samples - cursor
            clip[:, cursor:cursor + min(room, piece.size(1))] = piece[:, :room]
            cursor += piece.size(1)
        return clip

    def generate_step(self, t: int, inputs: torch.Tensor) -> Optional[torch.Tensor]:
        """the audio (at ``base_sr``) that follows position ``t``: one event's output, or silence up to the end of the clip
        when the next event would overrun it; ``None`` past the end"""
        if t >= self.total_samples:
            return None
        event, net, n_steps, params = self.next_event()
        if t / self.base_sr + event.seconds >= self.max_seconds:
            return inputs.new_zeros(inputs.size(0), self.total_samples - t)
        if self.print_events:
            print({"generator": type(net).__name__, "seconds": event.seconds, "temperature": event.temperature,
                   "start": t / self.base_sr})
        return self.run_event(inputs, net.to(self.device), n_steps, params)

    def next_event(self) -> Tuple[Event, ARM, int, Dict]:
        event = Event(**next(self.stream))
        net = event.network()
        n_steps = GenerateLoopV2.get_n_steps(GenerateLoopV2.Config(output_duration_sec=event.seconds), net)
        return event, net, n_steps, event.loop_parameters()

    # -- one event ------------------------------------------------------------------------------------
    def run_event(self, inputs: torch.Tensor, net: ARM, n_steps: int, params: Dict) -> torch.Tensor:
        spec = net.config.io_spec
        to_net, to_base = Resample(self.base_sr, spec.sr), Resample(spec.sr, self.base_sr)
        at_net_rate = to_net(inputs)
        prompts = [feature.transform(at_net_rate) for feature in spec.inputs]
        # a framed input (an STFT) covers fewer samples than it was given: what the loop returns in front of the new audio
        # is the prompt as the TARGET unit counts it
        kept = convert(prompts[0].shape[1], spec.targets[0].unit, Sample(sr=spec.sr), True)
        loop = GenerateLoopV2(GenerateLoopV2.Config(parameters=params, display_waveform=False, write_waveform=False,
                                                    yield_inversed_outputs=True),
                              network=net, n_steps=n_steps, logger=None,
                              dataloader=[[torch.ones(1), *prompts]])      # (first entry: the prompt's index, unused without a logger)
        outputs = next(iter(loop.run()))
        return to_base(outputs[0][:, kept:])
