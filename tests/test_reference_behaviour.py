"""The behaviours the reference's own test-suite asserts for the modules on the generate path, asserted of this build.

Each test names the reference test it restates (`tests/...` in the reference tree); nothing here imports the reference - the
expectations are the ones its tests write down (shapes, lengths, alignments, receptive fields).  CPU tests use the networks'
training-mode forward (the stock torch graph); the spectral ones go through the HIP functionals and need the GPU.
"""
import itertools

import pytest
import torch
from torch import nn

import mimikit_amd as mmk
from mimikit_amd.networks.wavenet_v2 import WNLayer


# ---------------------------------------------------------------------------- tests/test_wavenet.py
GRAPHS = list(itertools.product((True, False), (True, False), (None, 7), (0, 1), (None, 5, 7), (None, 34),
                                ((), (3,), (8, 2), (4, 9, 64)), ((16,), (32,), (8,))))


@pytest.mark.parametrize("case", range(0, len(GRAPHS), 7))          # every 7th point of the reference's 1152-point grid
def test_layer_should_support_various_graphs(case):
    """tests/test_wavenet.py:23-112: how input_dim / residuals_dim / skips_dim / pad_side shape a layer's two outputs"""
    with_gate, feed_skips, input_dim_cfg, pad, residuals, skips_dim, dims_1x1, dims_dil = GRAPHS[case]
    layer = WNLayer(input_dim=input_dim_cfg, dims_dilated=dims_dil, dims_1x1=dims_1x1, skips_dim=skips_dim, residuals_dim=residuals,
                    pad_side=pad, act_g=nn.Sigmoid() if with_gate else None)
    B, T = 1, 8
    in_dim = input_dim_cfg if input_dim_cfg is not None else (dims_dil[0] if residuals is None else residuals)
    skips = torch.randn(B, skips_dim, T) if feed_skips and skips_dim is not None else None
    out = layer((torch.randn(B, in_dim, T),), tuple(torch.randn(B, d, T) for d in dims_1x1), skips)
    assert isinstance(out, tuple) and len(out) == 2
    # the residual sum only exists when the layer's input has the residual width
    want_dim = residuals if residuals is not None and (input_dim_cfg is None or input_dim_cfg == residuals) else dims_dil[0]
    assert out[0].size(1) == want_dim
    if skips_dim is not None:
        assert out[1].size(1) == skips_dim and out[1].size(-1) == out[0].size(-1)
    if pad:
        assert out[0].size(-1) == T
    else:
        assert out[0].size(-1) < T


def test_wavenet_should_instantiate_from_default_config():
    """tests/test_wavenet.py:115-123"""
    net = mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=mmk.IOSpec.mulaw_io(mmk.IOSpec.MuLawIOConfig(input_module_type="embedding"))))
    assert isinstance(net, mmk.WaveNet) and len(net.layers) == 4 and net.rf == 16     # blocks=(4,), kernel 2


@pytest.mark.parametrize("blocks", [(3,), (1, 1, 1, 1, 1, 1, 1), (2, 2, 1), (1, 2, 2), (1, 1, 1, 1, 2)])
def test_rf_should_be_correct(blocks):
    """tests/test_wavenet.py:251-270: every one of these stacks has a receptive field of 8; a window of rf positions yields one
    output position, one more input position one more"""
    io = mmk.IOSpec.magspec_io(mmk.IOSpec.MagSpecIOConfig())
    net = mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=io, blocks=blocks)).train()
    assert net.rf == 8
    bins = io.inputs[0].elem_type.size
    assert net((torch.randn(2, 8, bins),))[0].size(1) == 1
    assert net((torch.randn(2, 9, bins),))[0].size(1) == 2


def test_wavenet_should_support_multiple_io():
    """tests/test_wavenet.py:168-212: two real-valued inputs (the second one conditions every layer) and two targets: a tuple of
    equally shaped outputs (the training graph; the HIP generate path refuses a second target)"""
    ext = mmk.Extractor("signal", mmk.FileToSignal(16000))

    def an_input():
        return mmk.InputSpec(extractor_name=ext.name, transform=mmk.Normalize(), module=mmk.LinearIO()).bind_to(ext)

    def a_target():
        return mmk.TargetSpec(extractor_name=ext.name, transform=mmk.Normalize(), module=mmk.LinearIO(),
                              objective=mmk.Objective("reconstruction")).bind_to(ext)

    io = mmk.IOSpec(inputs=(an_input(), an_input()), targets=(a_target(), a_target()))
    net = mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=io, dims_dilated=(128,), dims_1x1=(44,))).train()
    out = net.forward((torch.randn(1, 32, 1), torch.randn(1, 32, 1)))
    assert isinstance(out, tuple) and len(out) == 2 and out[0].size() == out[1].size()


# ---------------------------------------------------------------------------- tests/test_sample_rnn.py
def test_sample_rnn_should_instantiate_from_default_config():
    """tests/test_sample_rnn.py:16-24"""
    net = mmk.SampleRNN.from_config(mmk.SampleRNN.Config(io_spec=mmk.IOSpec.mulaw_io(mmk.IOSpec.MuLawIOConfig())))
    assert isinstance(net, mmk.SampleRNN) and len(net.tiers) == 3 and net.rf == 16


def test_sample_rnn_should_take_n_unfolded_inputs():
    """tests/test_sample_rnn.py:27-45: a training batch of T classes per row gives T - frame_sizes[0] predictions over the classes"""
    io = mmk.IOSpec.mulaw_io(mmk.IOSpec.MuLawIOConfig())
    fs = (16, 4, 2)
    net = mmk.SampleRNN.from_config(mmk.SampleRNN.Config(io_spec=io, frame_sizes=fs, inputs_mode="sum")).train()
    x = torch.arange(128).reshape(2, 64)
    out = net((x,))
    assert isinstance(out, tuple)
    assert out[0].shape == (2, x.size(1) - fs[0], io.inputs[0].elem_type.size)


# ---------------------------------------------------------------------------- tests/test_seq2seq.py
@pytest.mark.parametrize("downsampling", ["edge_sum", "edge_mean", "sum", "mean", "linear_resample"])
@pytest.mark.parametrize("n_lstm,residuals", [(1, False), (3, True)])
def test_encoder_decoder_forward_shapes(downsampling, n_lstm, residuals):
    """tests/test_seq2seq.py:18-110: the encoder folds hop frames into one coded frame (and hands on its final state), the decoder
    unfolds one coded frame into hop frames, whatever the pooling / stack depth"""
    io = mmk.IOSpec.magspec_io(mmk.IOSpec.MagSpecIOConfig(n_fft=64, hop_length=16))
    net = mmk.Seq2SeqLSTMNetwork.from_config(mmk.Seq2SeqLSTMNetwork.Config(
        io_spec=io, model_dim=32, hop=4, enc_n_lstm=n_lstm, dec_n_lstm=n_lstm, enc_apply_residuals=residuals, dec_apply_residuals=residuals,
        enc_downsampling=downsampling)).train()
    x = torch.randn(3, 4, 33)
    out = net((x,))
    out = out[0] if isinstance(out, tuple) else out
    assert out.shape == (3, 4, 33)


def test_seq2seq_takes_class_indices():
    """tests/test_seq2seq.py:149-154 builds (and trains) the network on IOSpec.mulaw_io with an embedding input: the training graph
    maps (batch, hop) classes to (batch, hop, q_levels) logits; generating with it is tests/test_gpu_networks.py's
    test_seq2seq_on_class_indices_*"""
    io = mmk.IOSpec.mulaw_io(mmk.IOSpec.MuLawIOConfig(input_module_type="embedding"))
    net = mmk.Seq2SeqLSTMNetwork.from_config(mmk.Seq2SeqLSTMNetwork.Config(io_spec=io, model_dim=32, hop=2)).train()
    out = net((torch.randint(0, 256, (3, 2)),))
    out = out[0] if isinstance(out, tuple) else out
    assert out.shape == (3, 2, 256)
    assert net.generate_params == {"temperature"}


def test_sample_rnn_on_embedding_inputs_is_refused_when_built():
    """sample_rnn_v2.py:161-167 would build an EmbeddingConv1d bottom tier; the reference's forward then fails on mismatched tier
    lengths, so there is nothing to reproduce: refused with the reason"""
    io = mmk.IOSpec.mulaw_io(mmk.IOSpec.MuLawIOConfig(input_module_type="embedding"))
    with pytest.raises(NotImplementedError, match="FramedLinearIO"):
        mmk.SampleRNN.from_config(mmk.SampleRNN.Config(io_spec=io))


# ---------------------------------------------------------------------------- tests/test_fft_alignment.py
def _signal(n, device):
    x = torch.randn(n, generator=torch.Generator().manual_seed(n))
    return (x / x.abs().max()).to(device)


# (the reference runs these on numpy arrays, i.e. through librosa; its torch functionals - the ones on the generate loop and the
#  ones this build implements - invert with torch.istft's defaults whatever `center` says (functionals.py:553-564, reproduced), so
#  only the centred cases hold for them)
@pytest.mark.gpu
@pytest.mark.parametrize("center,extra", [(True, 104)])
def test_convert_should_match_inverse(device, center, extra):
    """tests/test_fft_alignment.py:49-66: the frame count of the transform and the sample count `convert` gives for it agree with
    what the inverse transform returns"""
    n_fft, hop, n_frames = 2048, 512, 8
    fft = mmk.STFT(n_fft, hop, center=center, alignment="end")
    n = (n_frames - 1) * hop + extra if center else (n_fft - hop) + n_frames * hop + extra
    S = fft(_signal(n, device))
    assert S.shape[0] == n_frames
    y = fft.inv(S)
    assert mmk.convert(S.shape[0], fft.unit, mmk.Sample(sr=1), as_length=True) == y.shape[0]


@pytest.mark.gpu
@pytest.mark.parametrize("alignment,center,extra", [("end", True, 104), ("start", True, 87)])
def test_should_align(device, alignment, center, extra):
    """tests/test_fft_alignment.py:71-87, :116-132: with alignment "end" the inverse returns the LAST samples of the signal, with
    "start" the first ones"""
    n_fft, hop, n_frames = 2048, 512, 8
    fft = mmk.STFT(n_fft, hop, center=center, alignment=alignment, window="hann")
    n = (n_frames - 1) * hop + extra if center else (n_fft - hop) + n_frames * hop + extra
    x = _signal(n, device)
    S = fft(x)
    assert S.shape[0] == n_frames
    y = fft.inv(S)
    ref = x[-y.shape[0]:] if alignment == "end" else x[:y.shape[0]]
    skip = 0 if center else 1
    assert torch.allclose(ref[skip:], y[skip:], atol=2e-5)


@pytest.mark.gpu
def test_should_fail_with_magspec(device):
    """tests/test_fft_alignment.py:9-25: magnitudes alone do not invert to the signal (Griffin-Lim finds A signal with them)"""
    fft = mmk.MagSpec(2048, 512, center=True, alignment="end")
    x = _signal(7 * 512 + 104, device)
    S = fft(x)
    assert S.shape[0] == 8
    y = fft.inv(S)
    assert not torch.allclose(x[-y.shape[0]:], y, atol=1e-3)
