"""CPU-side tests: host logic mirrors the reference (facts captured from the reference's own code in
tests/golden/reference_facts.json), the C-ABI library loads and exports what include/mmk.h declares,
and the product path refuses to run anywhere but on the HIP device."""
import dataclasses as dtc
import os
import re

import numpy as np
import pytest
import torch

import mimikit_amd as mmk
from mimikit_amd import native
from mimikit_amd.features.functionals import mulaw_edges, mulaw_table
from oracle import torch_ref as O
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def shapes(net):
    return {k: list(v.shape) for k, v in net.state_dict().items()}


def cfg4_io():
    io = H.mu_emb()
    ext = mmk.Extractor("signal", mmk.FileToSignal(16000))
    return mmk.IOSpec(inputs=(io.inputs[0], mmk.InputSpec("signal", mmk.MagSpec(1024, 256, center=False), mmk.LinearIO()).bind_to(ext)),
                      targets=io.targets)


# ------------------------------------------------------------------ state_dict layout == reference
def test_state_dict_layout_matches_reference():
    f = H.facts()
    assert shapes(mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=H.mu_emb()))) == f["wavenet_default"]
    assert shapes(mmk.WaveNet.from_config(mmk.WaveNet.Config(
        io_spec=H.mu_emb(), blocks=(10,), dims_dilated=(64,), residuals_dim=64, skips_dim=64))) == f["wavenet_cfg2"]
    assert shapes(mmk.WaveNet.from_config(mmk.WaveNet.Config(
        io_spec=cfg4_io(), blocks=(10, 10, 10), dims_dilated=(256,), dims_1x1=(256,), residuals_dim=256,
        skips_dim=256))) == f["wavenet_cfg4"]
    assert shapes(mmk.SampleRNN.from_config(mmk.SampleRNN.Config(io_spec=H.mu_lin()))) == f["srnn_cfg1"]
    assert shapes(mmk.SampleRNN.from_config(mmk.SampleRNN.Config(
        io_spec=H.mu_lin(), frame_sizes=(16, 4, 1), hidden_dim=512, rnn_class="gru"))) == f["srnn_cfg3"]
    assert shapes(mmk.Seq2SeqLSTMNetwork.from_config(mmk.Seq2SeqLSTMNetwork.Config(
        io_spec=mmk.IOSpec.magspec_io(mmk.IOSpec.MagSpecIOConfig(sr=22050, n_fft=1024, hop_length=256))))) == f["s2s_cfg5"]


def test_rf_n_steps_and_unit_conversions_match_reference():
    host = H.facts()["host"]
    for key, rf in host["rf"].items():
        blocks, ks = (eval(p) for p in key.split("|"))
        net = mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=H.mu_emb(), blocks=blocks, kernel_sizes=ks, dims_dilated=(4,)))
        assert net.rf == rf, key
        assert O.wavenet_rf(*mmk.WaveNet.get_kernels_and_dilation(ks, blocks)) == rf
    s2s = mmk.Seq2SeqLSTMNetwork.from_config(mmk.Seq2SeqLSTMNetwork.Config(
        io_spec=mmk.IOSpec.magspec_io(mmk.IOSpec.MagSpecIOConfig(sr=22050, n_fft=1024, hop_length=256)), model_dim=8))
    wn = mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=H.mu_emb(), dims_dilated=(4,)))
    for key, n in host["n_steps"].items():
        name, dur = key.split("|")
        net = s2s if name == "s2s" else wn
        assert mmk.GenerateLoopV2.get_n_steps(mmk.GenerateLoopV2.Config(output_duration_sec=float(dur)), net) == n, key
    for c in host["convert"]:
        fr = mmk.Frame(c["n_fft"], c["hop"], padding=c["pad"])
        assert mmk.convert(c["n"], mmk.Sample(1), fr, True) == c["s2f_len"]
        assert mmk.convert(c["n"], mmk.Sample(1), fr, False) == c["s2f_pos"]
        assert mmk.convert(c["n"] // c["hop"], fr, mmk.Sample(1), True) == c["f2s_len"]
        assert mmk.convert(c["n"] // c["hop"], fr, mmk.Sample(1), False) == c["f2s_pos"]
        if not c["pad"] and c["n"] >= c["n_fft"]:
            keep = mmk.STFT(c["n_fft"], c["hop"], "mag", center=False).fixed_length(c["n"])
            assert keep == O.stft_fixed_length(c["n"], c["n_fft"], c["hop"], False)
            assert native.load_library().mmk_stft_n_frames(keep, c["n_fft"], c["hop"], 0) == c["s2f_len"]


def test_wavenet_rf_examples_of_the_reference_tests():
    """tests/test_wavenet.py:251-262 of the reference: all these block layouts have rf == 8"""
    io = mmk.IOSpec.magspec_io(mmk.IOSpec.MagSpecIOConfig())
    for blocks in [(3,), (1, 1, 1, 1, 1, 1, 1), (2, 2, 1), (1, 2, 2), (1, 1, 1, 1, 2)]:
        assert mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=io, blocks=blocks)).rf == 8
    with pytest.raises(ValueError):
        mmk.WaveNet.get_kernels_and_dilation((2, 3), (3,))


# ------------------------------------------------------------------ training-mode graphs (shapes)
def test_training_forward_shapes():
    wn = mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=mmk.IOSpec.magspec_io(mmk.IOSpec.MagSpecIOConfig()), blocks=(3,)))
    x = torch.randn(2, 9, 1025)
    assert wn((x,))[0].shape == (2, 2, 1025)          # T = rf + 1 -> 2 outputs
    with pytest.raises(RuntimeError):
        wn((x[:, :7],))                                  # shorter than rf
    srnn = mmk.SampleRNN.from_config(mmk.SampleRNN.Config(frame_sizes=(16, 4, 2), io_spec=H.mu_lin()))
    out = srnn((torch.arange(128).reshape(2, 64),))
    assert type(out) is tuple and out[0].shape == (2, 48, 256)
    s2s = mmk.Seq2SeqLSTMNetwork.from_config(mmk.Seq2SeqLSTMNetwork.Config(
        io_spec=mmk.IOSpec.magspec_io(mmk.IOSpec.MagSpecIOConfig(n_fft=128, hop_length=32)), model_dim=16, hop=4))
    y = s2s((torch.randn(4, 4, 65),))
    assert isinstance(y, torch.Tensor) and y.shape == (4, 4, 65)
    assert isinstance(wn, mmk.ARM) and isinstance(srnn, mmk.ARMWithHidden) and isinstance(s2s, mmk.ARM)
    assert len(mmk.SampleRNN.from_config(mmk.SampleRNN.Config(io_spec=H.mu_lin())).tiers) == 3
    assert len(mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=H.mu_emb())).layers) == 4


def test_generate_params():
    assert mmk.SampleRNN.from_config(mmk.SampleRNN.Config(io_spec=H.mu_lin())).generate_params == {"temperature"}
    # reproduced reference quirk: WaveNet exposes no sampling parameter to the loop
    assert mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=H.mu_emb())).generate_params == set()


# ------------------------------------------------------------------ error behaviour of the config layer
def test_io_module_wiring_errors():
    m = mmk.LinearIO()
    with pytest.raises(AttributeError):
        m.set(nope=1)
    m.set(in_dim=3)
    with pytest.raises(RuntimeError):
        m.set(in_dim=4)
    with pytest.raises(ValueError):
        m.module()                                       # out_dim missing
    with pytest.raises(ValueError):
        mmk.IOSpec.mulaw_io(mmk.IOSpec.MuLawIOConfig(input_module_type="nope"))
    with pytest.raises(ValueError):
        # WaveNet needs the embedding input; framed_linear lacks a frame size there
        mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=H.mu_lin()))
    io = cfg4_io()
    with pytest.raises(RuntimeError):
        io.unit                                          # Sample and Frame units mixed
    assert io.sr == 16000


def test_config_round_trip_and_type_tags():
    cfg = mmk.GenerateLoopV2.Config(output_duration_sec=2., batch_size=3, parameters={"temperature": 0.5})
    again = mmk.Config.deserialize(cfg.serialize())
    assert isinstance(again, mmk.GenerateLoopV2.Config)
    assert again.output_duration_sec == 2. and again.batch_size == 3 and again.parameters == {"temperature": 0.5}
    assert mmk.WaveNet.Config().type == "WaveNet.Config"
    assert "in_dim" not in mmk.LinearIO().serialize()     # runtime wiring is not serialised


def test_fill_and_prepare_prompt():
    x = torch.arange(6).reshape(2, 3)
    y = mmk.fill(x, ("data", 3), ("blank", 2))
    assert y.shape == (2, 5) and y.dtype == x.dtype and bool((y[:, 3:] == 0).all())
    z = mmk.fill(torch.ones(2, 3, 4), ("data", 3), ("blank", 5))
    assert z.shape == (2, 8, 4)
    with pytest.raises(AssertionError):
        mmk.fill(x, (torch.zeros(3), 1), ("blank", 1))
    p = mmk.prepare_prompt("cpu", (np.zeros(4, dtype=np.float32), torch.ones(1, 4)), 2)
    assert p[0].shape == (1, 6) and p[1].shape == (1, 6)


# ------------------------------------------------------------------ mu-law tables (host side of the kernel)
@pytest.mark.parametrize("q,comp", [(256, 1.0), (256, 0.5), (64, 1.0)])
def test_mulaw_edge_table_reproduces_the_formula(q, comp):
    e = mulaw_edges(q, comp)
    assert e.shape == (q - 1,) and bool((e[1:] > e[:-1]).all())
    x = torch.rand(500000, generator=torch.Generator().manual_seed(0)) * 2 - 1
    assert torch.equal(torch.searchsorted(e, x, right=True), O.mulaw_compress(x, q, comp))
    assert torch.equal(mulaw_table(q, comp), O.mulaw_expand(torch.arange(q), q, comp))
    g = H.golden("mulaw_c1.npz")
    if (q, comp) == (256, 1.0):
        xin = H.T(g["x"])[:-4]
        assert torch.equal(torch.searchsorted(e, xin, right=True), H.T(g["codes"])[:-4])


def test_numpy_twins():
    x = np.linspace(-1, 1, 1001).astype(np.float32)
    codes = mmk.MuLawCompress()(x)
    assert codes.dtype == np.int64 and codes.min() == 0 and codes.max() == 255
    assert np.abs(mmk.MuLawExpand()(codes) - x).max() < 0.05
    assert isinstance(mmk.MuLawCompress().inv, mmk.MuLawExpand) and isinstance(mmk.MagSpec().inv, mmk.GLA)


# ------------------------------------------------------------------ C ABI
def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "mmk.h")).read()
    declared = set(re.findall(r"\b(mmk_[a-z0-9_]+)\s*\(", header))
    declared -= {"mmk_stream_t"}
    lib = native.load_library()
    assert lib.mmk_abi_version() == native.ABI_VERSION == int(re.search(r"#define MMK_ABI_VERSION (\d+)", header).group(1))
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} is declared in include/mmk.h but not exported"
    assert declared == set(native.EXPORTED_SYMBOLS), declared ^ set(native.EXPORTED_SYMBOLS)


def _cfg4_plan_mode(tuning: bytes, clips: int = 32) -> int:
    """create (not commit) a plan of BASELINE config 4's geometry and say which step path it chose (mmk_wavenet_mode)"""
    lib = native.load_library()
    cfg = native.WaveNetConfig()
    cfg.n_layers, cfg.dim_dilated, cfg.residuals_dim, cfg.skips_dim, cfg.max_batch, cfg.q_levels = 30, 256, 256, 256, clips, 256
    for l in range(30):
        cfg.kernel_size[l], cfg.dilation[l] = 2, 2 ** (l % 10)
    cfg.n_cond, cfg.cond_in_dim[0], cfg.cond_dim[0] = 1, 513, 256
    cfg.mlp_hidden, cfg.out_dim, cfg.learn_temp, cfg.gated, cfg.bias = 128, 256, 1, 1, 1
    cfg.act_f, cfg.act_g, cfg.mlp_act = native.ACT['Tanh'], native.ACT['Sigmoid'], native.ACT['Mish']
    cfg.tuning = tuning
    handle = native.vp()
    assert lib.mmk_wavenet_plan_create(native.C.byref(cfg), native.C.byref(handle)) == 0, lib.mmk_last_error()
    mode = lib.mmk_wavenet_mode(handle)
    lib.mmk_wavenet_plan_destroy(handle)
    return mode


def test_execution_switches_travel_in_the_config_not_in_the_environment(monkeypatch):
    """which kernel a plan gets is decided by its config's `tuning` text (include/mmk.h) - an environment variable of the same name
    changes nothing (round 3's library read ~35 of them at plan creation)"""
    assert _cfg4_plan_mode(b"") == 5                                  # the stage pipeline
    assert _cfg4_plan_mode(b"MMK_WN_SPIPE=0") in (1, 2)               # the two- or one-hand-off persistent kernel
    assert _cfg4_plan_mode(b"MMK_WN_PERSISTENT=0") == 0               # the per-layer launch path
    monkeypatch.setenv("MMK_WN_SPIPE", "0")
    monkeypatch.setenv("MMK_WN_PERSISTENT", "0")
    assert _cfg4_plan_mode(b"") == 5
    # many clips per GPU: the same stages with groups of 16 clips per visit on the matrix pipe (wavenet_bpipe.hip), up to 512 clips a launch
    assert _cfg4_plan_mode(b"", clips=104) == 5 and _cfg4_plan_mode(b"", clips=105) == 6 and _cfg4_plan_mode(b"", clips=512) == 6
    # ... except where the ring takes the clips two per visit (an even count, up to its 128 clips): round 6
    assert _cfg4_plan_mode(b"", clips=106) == 5 and _cfg4_plan_mode(b"", clips=128) == 5 and _cfg4_plan_mode(b"", clips=129) == 6 and _cfg4_plan_mode(b"", clips=130) == 6
    for clips in (1, 32, 64, 104, 105, 106, 107, 108, 127, 128, 129, 132, 256):      # (what wavenet_v2._ensure_plan re-plans a smaller batch by)
        assert (_cfg4_plan_mode(b"", clips=clips) == 6) == native.wn_bpipe_by_default(clips), clips
    assert _cfg4_plan_mode(b"MMK_WN_BPIPE=0", clips=128) == 5 and _cfg4_plan_mode(b"MMK_WN_BPIPE=1", clips=8) == 6
    assert _cfg4_plan_mode(b"MMK_WN_BPIPE=0", clips=256) not in (5, 6) and _cfg4_plan_mode(b"", clips=513) not in (5, 6)
    assert native.tuning_text({"A": "1"}, {"B": "0", "A": "2"}) == b"A=2;B=0"
    with pytest.raises(ValueError):
        native.tuning_text({f"MMK_SWITCH_{i}": "1" for i in range(40)})
    header = open(os.path.join(ROOT, "include", "mmk.h")).read()
    assert int(re.search(r"#define MMK_TUNING_CHARS (\d+)", header).group(1)) == native.TUNING_CHARS
    for src in os.listdir(os.path.join(ROOT, "mimikit_amd", "csrc")):
        text = open(os.path.join(ROOT, "mimikit_amd", "csrc", src)).read()
        assert text.count("getenv(") == (1 if src == "plan_util.h" else 0), f"{src} reads the environment"


def test_abi_argument_validation_needs_no_gpu():
    lib = native.load_library()
    cfg = native.WaveNetConfig()
    handle = native.vp()
    assert lib.mmk_wavenet_plan_create(native.C.byref(cfg), native.C.byref(handle)) == -1   # n_layers = 0
    assert b"n_layers" in lib.mmk_last_error()
    cfg.n_layers, cfg.dim_dilated, cfg.max_batch, cfg.q_levels = 2, 16, 2, 256
    cfg.kernel_size[0] = cfg.kernel_size[1] = 2
    cfg.dilation[0], cfg.dilation[1] = 1, 2
    cfg.mlp_hidden, cfg.out_dim, cfg.learn_temp, cfg.gated, cfg.bias = 8, 256, 1, 1, 1
    cfg.act_f, cfg.act_g, cfg.mlp_act = native.ACT['Tanh'], native.ACT['Sigmoid'], native.ACT['Mish']
    assert lib.mmk_wavenet_plan_create(native.C.byref(cfg), native.C.byref(handle)) == 0
    assert lib.mmk_wavenet_receptive_field(handle) == 4
    assert lib.mmk_wavenet_workspace_bytes(handle) > 0
    # generate before commit is a state error, not a crash
    assert lib.mmk_wavenet_generate(handle, 1, 1, 8, None, None, 4, 1, None, None, None) == -5
    lib.mmk_wavenet_plan_destroy(handle)
    assert lib.mmk_stft_n_frames(22050, 1024, 256, 0) == 83
    assert lib.mmk_packed_weight_floats(17, 33) == 32 * 48


# ------------------------------------------------------------------ no CPU fallback
def test_generate_path_refuses_cpu():
    wn = mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=H.mu_emb(), blocks=(2,), dims_dilated=(8,))).eval()
    x = torch.zeros(1, 8, dtype=torch.int64)
    for call in (lambda: wn((x,)), lambda: wn.generate_step((x,), t=8), lambda: wn.before_generate((x,), 0),
                 lambda: mmk.MuLawCompress()(torch.zeros(4)), lambda: mmk.MagSpec()(torch.zeros(4096)),
                 lambda: mmk.CategoricalSampler().eval()(torch.zeros(2, 256))):
        with pytest.raises(RuntimeError, match="HIP device|MI355X"):
            call()
    srnn = mmk.SampleRNN.from_config(mmk.SampleRNN.Config(io_spec=H.mu_lin())).eval()
    with pytest.raises(RuntimeError):
        srnn.before_generate((torch.zeros(1, 32, dtype=torch.int64),), 0)
    with pytest.raises(RuntimeError):
        srnn((torch.zeros(1, 64, dtype=torch.int64),))    # eval forward is not the training graph
    with pytest.raises(native.NativeError, match="no CPU fallback"):
        native.load_library("/nonexistent/libmmk_hip.so")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "mimikit_amd")
    for dirpath, _, files in os.walk(pkg):
        for name in files:
            if name.endswith(".py"):
                src = open(os.path.join(dirpath, name)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f"{name} imports the oracle"


# ---------------------------------------------------------------------------- prompt serving / callers
def test_indices_sampler_matches_reference_draws():
    """IndicesSampler under a fixed torch seed against the draws of the reference's own class (reference_facts.json): fixed
    positions stay, None entries are drawn on the sampling stride and redrawn after every pass; list form = N plain draws"""
    import torch
    f = H.facts()["host"]["indices_sampler"]
    torch.manual_seed(f["seed"])
    s = mmk.IndicesSampler(N=4, indices=tuple(f["indices"]), max_i=f["max_i"], redraw=True, sampling_stride=f["stride"])
    assert [[int(i) for i in s] for _ in range(3)] == f["passes"]
    assert all(i % f["stride"] == 0 for p in f["passes"] for k, i in enumerate(p) if k != 1)
    g = H.facts()["host"]["indices_sampler_n"]
    torch.manual_seed(g["seed"])
    s = mmk.IndicesSampler(N=g["N"], indices=[], min_i=g["min_i"], max_i=g["max_i"], redraw=False)
    assert [[int(i) for i in s] for _ in range(2)] == g["passes"]
    assert mmk.PromptIndices(10)(7).tolist() == [7] and mmk.PromptIndices(10)(7).dtype == np.int32


def test_generate_callback_contract():
    """loops/callbacks.py:155-169: every `every_n_epochs` epochs the callback sets loop.template_vars = {epoch} and drains
    loop.run() to the end"""
    class Loop:
        def __init__(self):
            self.template_vars, self.drained = {}, 0

        def run(self):
            for i in range(3):
                yield i
            self.drained += 1

    class Trainer:
        current_epoch = 0

    loop, tr = Loop(), Trainer()
    cb = mmk.GenerateCallback(loop, every_n_epochs=2)
    for epoch in range(4):
        tr.current_epoch = epoch
        cb.on_train_epoch_end(tr, None)
    assert loop.drained == 2 and loop.template_vars == {"epoch": 4}


def test_resample_filter_bank_and_oracle_properties():
    """the polyphase filter bank the HIP kernel takes is the one the oracle's restatement of torchaudio's resample builds:
    evaluating output sample n * new + j as sum_k table[j][k] x[n * orig + k - width] reproduces the oracle; a constant
    stays a constant away from the edges (unit DC gain), lengths are ceil(new * T / orig)"""
    import torch
    from mimikit_amd.features.functionals import resample_filter_bank
    for o_sr, n_sr in ((22050, 16000), (16000, 22050), (44100, 16000), (8000, 16000)):
        orig, new, width, table = resample_filter_bank(o_sr, n_sr)
        assert table.shape == (new, 2 * width + orig)
        x = torch.randn(2, 3000, generator=torch.Generator().manual_seed(o_sr))
        y = O.resample(x, o_sr, n_sr)
        assert y.shape == (2, -(-new * 3000 // orig))
        xp = torch.nn.functional.pad(x, (width, width + orig)).double()
        for idx in (0, 1, new + 3, y.shape[1] - 1, y.shape[1] // 2):
            n, j = divmod(idx, new)
            want = (table[j].double() * xp[:, n * orig:n * orig + 2 * width + orig]).sum(-1)
            assert torch.allclose(want.float(), y[:, idx], rtol=1e-4, atol=1e-5)
        ones = O.resample(torch.ones(1, 4000), o_sr, n_sr)
        mid = ones[0, ones.shape[1] // 4: 3 * ones.shape[1] // 4]
        assert float((mid - 1).abs().max()) < 2e-3


# ---------------------------------------------------------------------------- checkpoint interchange
REFERENCE_LAYOUT_YAML = """
type: SampleRNN.Config
io_spec:
  inputs:
  - extractor_name: signal
    transform:
      type: MuLawCompress
      q_levels: 256
      compression: 0.5
    module:
      type: FramedLinearIO
      activation: null
      dropout: 0.0
      dropout1d: 0.0
      frame_size: null
      hop_length: null
      bias: true
  targets:
  - extractor_name: signal
    transform:
      type: MuLawCompress
      q_levels: 256
      compression: 0.5
    module:
      type: MLPIO
      activation:
        act: Mish
        scaled: false
        static: false
        with_rate: false
        params: {}
      dropout: 0.0
      dropout1d: 0.0
      hidden_dim: 64
      n_hidden_layers: 1
      bias: true
      min_temperature: 0.0001
    objective:
      objective_type: categorical_dist
      params: {}
      weight: 1.0
    extra_loss_terms: []
frame_sizes:
- 8
- 2
- 2
hidden_dim: 48
rnn_class: gru
n_rnn: 2
rnn_dropout: 0.0
rnn_bias: true
h0_init: zeros
weight_norm: false
inputs_mode: sum
"""

REFERENCE_LAYOUT_DATASET = """
sources: []
filename: unknown
extractors:
- name: signal
  functional:
    type: Compose
    functionals:
    - type: FileToSignal
      sr: 16000
      offset: 0.0
      duration: null
    - type: Normalize
      p: .inf
      dim: -1
    - type: RemoveDC
  merge_files_labels: false
  consolidate_labels: false
  derived_from: null
"""


def test_checkpoint_round_trip_and_reference_yaml_layout(tmp_path):
    """Checkpoint.create / .network (reference checkpoint.py:96-173): config YAML -> io_spec.bind_to(dataset config) ->
    from_config -> load_state_dict(strict=True); and a config written in the REFERENCE's YAML layout (mappings under
    io_spec / inputs / targets / objective / activation / extractors carry no `type` tag: the key types them,
    config.py:33-42) deserialises into a network with the reference's state_dict keys.  (The layout is restated from the
    reference's config.py - OmegaConf is not installed here, so no reference-written YAML exists to pin it.)"""
    import warnings
    warnings.filterwarnings("ignore")
    for build in (H.wavenet_b, lambda: H.srnn("lstm", weight_norm=True), H.s2s_tiny, lambda: H.wavenet_option("mlp2"),
                  lambda: H.freqnet("g4"), lambda: H.srnn_option("gru_n2")):
        net = build()[0]
        ck = mmk.Checkpoint("run", 7, str(tmp_path)).create(net)
        assert ck.os_path.endswith("run/epoch=7.ckpt") and mmk.Checkpoint.get_id_and_epoch(ck.os_path) == ("run", 7)
        back = mmk.Checkpoint.from_path(ck.os_path)
        net2 = back.network
        a, b = net.state_dict(), net2.state_dict()
        assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)
        assert net2.config.serialize() == net.config.serialize()
        assert back.dataset_config.schema.keys() == {"signal"}
        ck.delete()
    cfg = mmk.Config.deserialize(REFERENCE_LAYOUT_YAML)
    ds = mmk.Config.deserialize(REFERENCE_LAYOUT_DATASET, as_type=mmk.DatasetConfig)
    assert isinstance(cfg, mmk.SampleRNN.Config) and isinstance(cfg.io_spec, mmk.IOSpec)
    assert isinstance(cfg.io_spec.targets[0], mmk.TargetSpec) and str(cfg.io_spec.targets[0].objective.objective_type) == "categorical_dist"
    assert isinstance(ds.extractors[0], mmk.Extractor) and ds.extractors[0].functional.functionals[0].sr == 16000
    cfg.io_spec.bind_to(ds)
    net = mmk.SampleRNN.from_config(cfg)
    keys = set(net.state_dict())
    assert {"tiers.0.rnn.weight_ih_l1", "tiers.1.up_sampler.fc.weight", "output_modules.0.estimator.0.fc.2.weight",
            "tiers.2.input_module.heads.0.2.2.cv.weight"} <= keys
    assert net.config.io_spec.sr == 16000 and net.config.io_spec.inputs[0].transform.compression == 0.5
    # our own emitter writes that layout: no `type` under the statically typed keys
    y = net.config.serialize()
    assert "type: IOSpec" not in y and "type: InputSpec" not in y and "type: Objective" not in y and "type: SampleRNN.Config" in y


def test_library_in_the_tree_was_built_from_these_sources():
    """the library carries a digest of the sources it was compiled from (mimikit_amd/build.py, `mmk_build_digest`): a prebuilt
    libmmk_hip.so that travelled with the tree is used only if that digest is the tree's - by content, not by modification time"""
    from mimikit_amd import build as hip_build
    assert hip_build.library_digest() == hip_build.source_digest() and len(hip_build.source_digest()) == 32
    lib = native.load_library()
    assert lib.mmk_build_digest().decode() == hip_build.source_digest()
    assert not hip_build._stale()


def _wavenet_plan_mode(channels, layers, clips, mlp_hidden, classes, cond_dims=(), tuning=b""):
    lib = native.load_library()
    cfg = native.WaveNetConfig()
    cfg.n_layers, cfg.dim_dilated, cfg.residuals_dim, cfg.skips_dim, cfg.max_batch, cfg.q_levels = layers, channels, channels, channels, clips, classes
    for l in range(layers):
        cfg.kernel_size[l], cfg.dilation[l] = 2, 2 ** (l % 10)
    cfg.n_cond = len(cond_dims)
    for j, d in enumerate(cond_dims):
        cfg.cond_in_dim[j], cfg.cond_dim[j] = 513, d
    cfg.mlp_hidden, cfg.out_dim, cfg.learn_temp, cfg.gated, cfg.bias = mlp_hidden, classes, 1, 1, 1
    cfg.act_f, cfg.act_g, cfg.mlp_act = native.ACT['Tanh'], native.ACT['Sigmoid'], native.ACT['Mish']
    cfg.tuning = tuning
    handle = native.vp()
    assert lib.mmk_wavenet_plan_create(native.C.byref(cfg), native.C.byref(handle)) == 0, lib.mmk_last_error()
    mode = lib.mmk_wavenet_mode(handle)
    lib.mmk_wavenet_plan_destroy(handle)
    return mode


def test_flagship_step_kernels_are_not_cut_to_the_baseline_head():
    """which step path a plan takes for heads and conditioning other than BASELINE's (the plan pads a narrower head to the kernels' 128 x 256
    and lays two conditioning inputs side by side - DESIGN 5.6): decided at plan creation, which needs no GPU"""
    SPIPE, LPIPE = 5, 4
    assert _wavenet_plan_mode(256, 30, 32, 128, 256, (256,)) == SPIPE                     # BASELINE config 4
    assert _wavenet_plan_mode(256, 30, 32, 40, 64, (16,)) == SPIPE                        # 40 hidden units (not a multiple of 16), 64 classes
    assert _wavenet_plan_mode(256, 30, 32, 128, 256, (128, 128)) == SPIPE                 # two conditioning inputs, 256 channels together
    assert _wavenet_plan_mode(256, 30, 32, 128, 256, (256, 16)) != SPIPE                  # ... more than the helpers' 256 K slots
    assert _wavenet_plan_mode(256, 30, 32, 256, 256, ()) != SPIPE                         # a head wider than the kernel's
    assert _wavenet_plan_mode(64, 10, 8, 128, 256) == LPIPE                               # BASELINE config 2
    assert _wavenet_plan_mode(64, 10, 8, 64, 128) == LPIPE                                # a narrower head in whole tiles of 16
    assert _wavenet_plan_mode(64, 10, 8, 40, 128) != LPIPE
    assert _wavenet_plan_mode(64, 10, 8, 128, 256, (16,)) == LPIPE                        # one conditioning input: its products come from the plan's GEMM
    assert _wavenet_plan_mode(64, 10, 8, 128, 256, (16, 32)) == LPIPE                     # two: side by side
    assert _wavenet_plan_mode(64, 10, 8, 128, 256, (16, 32), b"MMK_WN_LPIPE=0") == 0      # ... which only the layer pipeline takes: else the launch path
    assert _wavenet_plan_mode(128, 10, 8, 128, 256, (16, 32)) == 0


def test_several_inputs_and_targets_are_described_or_refused_by_name():
    """the plan description of networks of several inputs / targets (host logic, no GPU): class sizes per input, the ZipReduceVariables mode,
    one head per target; what the HIP path cannot take is refused with the option's name (more targets than inputs - the loop's
    `zip(tensors, outputs)` would drop them -, a further target on a stream that is not a class stream)"""
    mu = lambda q, kind="framed_linear", **kw: mmk.IOSpec.mulaw_io(mmk.IOSpec.MuLawIOConfig(q_levels=q, input_module_type=kind, **kw))
    a, b = mu(256, mlp_dim=32), mu(64, mlp_dim=48, n_mlp_layers=1)
    io = mmk.IOSpec(inputs=(a.inputs[0], b.inputs[0]), targets=(a.targets[0], b.targets[0]))
    net = mmk.SampleRNN.from_config(mmk.SampleRNN.Config(io_spec=io, frame_sizes=(16, 4, 1), hidden_dim=32, rnn_class="gru", inputs_mode="static_mix")).eval()
    c = net._describe(4)
    assert (c.n_inputs, c.n_targets, c.inputs_mode) == (2, 2, 2) and list(c.in_class)[:2] == [256, 64]
    assert (c.q_levels, c.x_q_levels[1], c.x_mlp_hidden[1], c.x_mlp_n_hidden[1]) == (256, 64, 48, 1)
    io_bad = mmk.IOSpec(inputs=(a.inputs[0],), targets=(a.targets[0], b.targets[0]))
    net = mmk.SampleRNN.from_config(mmk.SampleRNN.Config(io_spec=io_bad, frame_sizes=(16, 4, 1), hidden_dim=32, rnn_class="gru")).eval()
    with pytest.raises(NotImplementedError, match="more targets than inputs"):
        net._describe(4)
    # WaveNet: a class stream as conditioning input (EmbeddingIO) that a second target feeds
    a, b = mu(256, "embedding", mlp_dim=32), mu(64, "embedding", mlp_dim=16)
    io = mmk.IOSpec(inputs=(a.inputs[0], b.inputs[0]), targets=(a.targets[0], b.targets[0]))
    wn = mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=io, blocks=(3,), dims_dilated=(32,), dims_1x1=(16,), residuals_dim=32, skips_dim=32)).eval()
    c = wn._describe(4)
    assert (c.n_cond, c.cond_q_levels[0], c.cond_dim[0], c.n_targets, c.x_out_dim[1], c.x_mlp_hidden[1]) == (1, 64, 16, 2, 64, 16)
    # ... and a second target whose input is NOT a class stream
    ext = mmk.Extractor("signal", mmk.FileToSignal(16000))
    io_bad = mmk.IOSpec(inputs=(a.inputs[0], mmk.InputSpec("signal", mmk.MagSpec(22, 4, center=False), mmk.LinearIO()).bind_to(ext)),
                        targets=(a.targets[0], b.targets[0]))
    wn = mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=io_bad, blocks=(3,), dims_dilated=(32,), dims_1x1=(16,), residuals_dim=32, skips_dim=32)).eval()
    with pytest.raises(NotImplementedError, match="target 1"):
        wn._describe(4)
