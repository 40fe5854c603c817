"""The callers either side of the generate path on the HIP device: prompt serving through
GenerateLoopV2.get_dataloader / from_config (IndicesSampler positions, features computed on the device), GenerateCallback,
chunked long-form generation, EnsembleGenerator with the on-device Resample between networks."""
import numpy as np
import pytest
import torch

import mimikit_amd as mmk
from oracle import torch_ref as O
from tests import helpers as H

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def test_resample_kernel_vs_oracle(device):
    """Resample (HIP polyphase FIR) against the oracle's restatement of torchaudio 2.0.1's resample - PARITY UNPINNED for the
    reference itself (torchaudio is not installed in the build container); fp32 tolerance 2e-6 of the largest sample"""
    gen = torch.Generator().manual_seed(5)
    for o_sr, n_sr, n in ((22050, 16000, 22050), (16000, 22050, 7001), (44100, 16000, 5000), (16000, 8000, 1234), (16000, 16000, 100)):
        x = torch.randn(3, n, generator=gen)
        got = mmk.Resample(o_sr, n_sr)(x.to(device)).cpu()
        want = O.resample(x, o_sr, n_sr)
        assert got.shape == want.shape
        assert float((got - want).abs().max()) <= 2e-6 * float(want.abs().max()) + 1e-7
    # leading dimensions and a strided view
    x = torch.randn(2, 3, 4000, generator=gen)
    got = mmk.Resample(22050, 16000)(x.to(device)[..., 100:3100]).cpu()
    want = O.resample(x[..., 100:3100], 22050, 16000)
    assert got.shape == want.shape and float((got - want).abs().max()) <= 2e-6 * float(want.abs().max())
    assert mmk.Resample(22050, 16000).inv == mmk.Resample(16000, 22050)


def _srnn_and_signal():
    net, sd, arch = H.srnn("gru")
    gen = torch.Generator().manual_seed(2)
    signal = (torch.rand(16000 * 3, generator=gen) * 2 - 1).numpy().astype(np.float32)
    return net, sd, arch, signal


def test_loop_from_config_serves_prompts_from_a_dataset(device):
    """GenerateLoopV2.from_config over an in-memory dataset: positions in seconds -> sample indices, None -> drawn on the
    down-sampling stride (IndicesSampler), prompt features (mu-law classes) computed on the device; the generated clips equal
    the oracle's for the prompts the loader cut"""
    net, sd, arch, signal = _srnn_and_signal()
    cfg = mmk.GenerateLoopV2.Config(output_duration_sec=0.004, prompts_length_sec=0.008, prompts_position_sec=(0.5, None, 1.25, None),
                                    batch_size=2, downsampling=16, display_waveform=False, yield_inversed_outputs=False)
    torch.manual_seed(99)
    loop = mmk.GenerateLoopV2.from_config(cfg, {"signal": signal}, net, logger=None)
    assert loop.n_steps == 64
    outs = list(loop.run())
    torch.set_grad_enabled(False)
    assert len(outs) == 2                                  # 4 prompts in batches of 2
    # re-derive what the loader served: positions 0.5 s and 1.25 s are fixed, the others lie on the stride
    torch.manual_seed(99)
    s = mmk.IndicesSampler(N=4, indices=(8000, None, 20000, None), max_i=len(signal) - 128, redraw=True, sampling_stride=16)
    pos = [int(i) for i in s]
    assert pos[0] == 8000 and pos[2] == 20000 and pos[1] % 16 == 0 and pos[3] % 16 == 0
    for b, out in enumerate(outs):
        for r in range(2):
            p = pos[2 * b + r]
            prompt = O.mulaw_compress(torch.from_numpy(signal[p:p + 128])[None])
            want = O.SampleRNNOracle(sd, **arch).generate(prompt, 64)
            assert torch.equal(out[0][r].cpu(), want[0])


def test_generate_callback_drains_the_loop(device):
    net, sd, arch, signal = _srnn_and_signal()
    cfg = mmk.GenerateLoopV2.Config(output_duration_sec=0.002, prompts_length_sec=0.004, prompts_position_sec=(0.1, 0.2),
                                    batch_size=1, display_waveform=False)
    logged = []

    class Logger:
        def write(self, audio, **kw):
            logged.append(("write", kw))

        def display(self, audio, **kw):
            logged.append(("display", dict(kw)))

    loop = mmk.GenerateLoopV2.from_config(cfg, {"signal": signal}, net, logger=None)
    calls = []
    loop.config.callback = lambda outs: calls.append(tuple(o.shape for o in outs))

    class Trainer:
        current_epoch = 9

    mmk.GenerateCallback(loop, every_n_epochs=10).on_train_epoch_end(Trainer(), None)
    torch.set_grad_enabled(False)
    assert loop.template_vars == {"epoch": 10}
    assert len(calls) == 2 and calls[0][0] == (1, 64 + 32)          # both prompts generated: 64 prompt + 32 new samples


def test_generate_chunks_chain(device):
    """every chunk is prompted with the tail of the one before: three chunks equal the oracle run chunk by chunk"""
    net, sd, arch = H.srnn("lstm")
    prompt = torch.randint(0, 256, (2, 48), generator=torch.Generator().manual_seed(4))
    cfg = mmk.GenerateLoopV2.Config(output_duration_sec=40 / 16000, display_waveform=False, yield_inversed_outputs=False)
    chunks = list(mmk.generate_chunks(cfg, net, prompt.to(device), 3))
    torch.set_grad_enabled(False)
    assert len(chunks) == 3 and all(c.shape == (2, 40) for c in chunks)
    p = prompt
    for c in chunks:
        want = O.SampleRNNOracle(sd, **arch).generate(p, 40)
        assert torch.equal(c.cpu(), want[:, 48:])
        p = want[:, -48:]
    with pytest.raises(ValueError):
        next(mmk.generate_chunks(mmk.GenerateLoopV2.Config(display_waveform=False), net, prompt.to(device), 1))


def test_ensemble_generator_chains_networks_across_sample_rates(device):
    """two events at a base rate of 22.05 kHz: a 16 kHz SampleRNN, then a 16 kHz WaveNet - each event resamples the running
    clip to the network's rate on the device, generates, and resamples back (ensemble_generator.py:113-144).  Checked:
    the prompt is kept, every event writes its share, the run is deterministic (greedy), and the first event equals the
    oracle's composition resample -> mu-law -> generate -> expand -> resample within fp32 resampling tolerance."""
    srnn, sd, arch = H.srnn("gru")
    wn, _, _ = H.wavenet_a()
    gen = torch.Generator().manual_seed(6)
    prompt = (torch.rand(2, 2205, generator=gen) * 2 - 1) * 0.5

    def stream():
        yield dict(generator=srnn, seconds=0.02)
        yield dict(generator=wn, seconds=0.01)
        while True:
            yield dict(generator=srnn, seconds=1.0)          # never fits: the tail stays blank

    def run():
        eg = mmk.EnsembleGenerator(prompt, max_seconds=0.14, base_sr=22050, stream=stream(), device=device)
        out = eg.run()
        torch.set_grad_enabled(False)
        return out.cpu()

    out = run()
    assert out.shape == (2, int(0.14 * 22050)) and bool(torch.isfinite(out).all())
    assert torch.equal(out[:, :2205], prompt)
    assert torch.equal(run(), out)
    # 0.02 s at 16 kHz = 320 samples -> 441 at 22.05 kHz; 0.01 s = 160 -> 221
    seg1 = out[:, 2205:2205 + 441]
    assert float(seg1.abs().max()) > 0
    assert float(out[:, 2205 + 441 + 221 + 5:].abs().max()) == 0.0      # nothing after the two events
    # oracle composition of event 1
    x16 = O.resample(prompt, 22050, 16000)
    codes = O.mulaw_compress(x16)
    full = O.SampleRNNOracle(sd, **arch).generate(codes, 320)
    want = O.resample(O.mulaw_expand(full)[:, codes.shape[1]:], 16000, 22050)
    assert want.shape[1] == 441
    got_codes = mmk.MuLawCompress()(mmk.Resample(22050, 16000)(prompt.to(device))).cpu()
    if torch.equal(got_codes, codes):                       # (a sample within fp32 rounding of a mu-law bin edge may flip a code)
        assert float((seg1 - want).abs().max()) <= 1e-5
