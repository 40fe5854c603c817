"""Headline benchmark: generated audio samples / second on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...      (no launcher around it: bench.py starts its own N ranks, see launch_ranks)

Workload (BASELINE.json `metric`: "WaveNet 256-ch mu-law, 16 kHz" = configs[3]): WaveNet
blocks=(10,10,10), 256 dilated/residual/skip channels, one STFT-magnitude conditioning input
(513 bins -> LinearIO 256 -> per-layer 1x1), mu-law-256 MLP head, 32 clips per GPU (256 clips
over 8 GPUs), prompt 3072 samples, 1 s = 16 000 generated samples per clip, greedy decode.
One "step" = one full generate pass over the local batch: before_generate (queue warm-up over the
prompt) + 16 000 auto-regressive steps + mu-law expansion of the result, inputs resident in HBM.
Clips shard over ranks with one weight broadcast and no other collective (weak scaling).

The JSON line also carries
  roofline     : HBM roofline of the dominant kernel, from HIP start/stop events on its launches
  cpu_baseline : the reference ALGORITHM (oracle/torch_ref.py: naive full-window forward per step,
                 Python loop) timed on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling
MFMA_F32_PEAK_TFLOPS = 157.3   # dense fp32 MFMA (v_mfma_f32_16x16x4_f32), MI355X_MICROARCH.md
CPU_BASELINE_THREADS = 16      # the oracle's small ops get SLOWER on all 128 cores of the GPU box (intra-op pool overhead)


class cpu_threads:
    """time the CPU baseline with a thread count at which the reference algorithm actually scales"""

    def __enter__(self):
        self.old = torch.get_num_threads()
        torch.set_num_threads(max(1, min(CPU_BASELINE_THREADS, os.cpu_count() or 1)))

    def __exit__(self, *a):
        torch.set_num_threads(self.old)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="wavenet_cfg4",
                    choices=["wavenet_cfg4", "wavenet_cfg2", "srnn_cfg3", "s2s_cfg5", "mulaw", "stft", "istft", "gla", "stub"])
    ap.add_argument("--clips", type=int, default=0, help="clips per GPU (0 = the workload's BASELINE value)")
    ap.add_argument("--seconds", type=float, default=1.0, help="generated audio per clip")
    ap.add_argument("--temperature", type=float, default=0.0,
                    help="> 0: sampled decode at this temperature (throughput only: the CPU check needs greedy decode)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-strong-leg", action="store_true", help="wavenet_cfg4: skip the strong-scaling leg (256 clips / N ranks) after the timed region")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline sample")
    ap.add_argument("--no-others", action="store_true", help="wavenet_cfg4 at N = 1: skip the short passes of the other BASELINE workloads (`other_workloads` key)")
    ap.add_argument("--force-dist", action="store_true", help="with --gpus 1: initialise the RCCL process group of ONE rank all the same and run the path's "
                    "collectives (weight broadcast, fenced max-over-ranks clock) on it - what an N-GPU run does, on a 1-GPU box")
    ap.add_argument("--tuning", default="", help="execution switches of the plans of this run, NAME=VALUE[;NAME=VALUE...] (include/mmk.h `tuning`): "
                    "the library reads no environment variable - this is the only way to A/B a kernel choice from the command line")
    return ap.parse_args()


def apply_tuning(text):
    """--tuning -> mimikit_amd.native.PLAN_TUNING, before any plan is built; an MMK_* variable in the environment is refused: it would be ignored
    by the product library and an A/B run would silently compare two identical configurations"""
    # (only the names that look like the library's execution switches: MMK_REFERENCE_ROOT and the like are other programs' business; a diagnostic
    # build - MMK_DIAG_LIB set - does read its MMK_* variables)
    switches = ("MMK_WN_", "MMK_SRNN_", "MMK_S2S_", "MMK_BP_", "MMK_SP_", "MMK_LSTM_", "MMK_FEAT_")
    stray = sorted(k for k in os.environ if k.startswith(switches) and not os.environ.get("MMK_DIAG_LIB"))
    if stray:
        raise SystemExit(f"bench.py: {', '.join(stray)} set in the environment - the library does not read it; use --tuning NAME=VALUE")
    if not text:
        return
    from mimikit_amd import native
    for item in text.split(";"):
        if not item:
            continue
        name, eq, value = item.partition("=")
        if not eq or not name.startswith("MMK_"):
            raise SystemExit(f"bench.py: --tuning item {item!r} is not MMK_NAME=VALUE")
        native.PLAN_TUNING[name] = value


# ----------------------------------------------------------------------------- workloads
def build_wavenet(cfg_name):
    import mimikit_amd as mmk
    io = mmk.IOSpec.mulaw_io(mmk.IOSpec.MuLawIOConfig(sr=16000, q_levels=256, input_module_type="embedding"))
    if cfg_name == "wavenet_cfg4":
        ext = mmk.Extractor("signal", mmk.FileToSignal(16000))
        cond = mmk.InputSpec("signal", mmk.MagSpec(1024, 256, center=False), mmk.LinearIO()).bind_to(ext)
        io = mmk.IOSpec(inputs=(io.inputs[0], cond), targets=io.targets)
        cfg = mmk.WaveNet.Config(io_spec=io, blocks=(10, 10, 10), dims_dilated=(256,), dims_1x1=(256,),
                                 residuals_dim=256, skips_dim=256)
        clips, cond_dim = 32, 513
    else:
        cfg = mmk.WaveNet.Config(io_spec=io, blocks=(10,), dims_dilated=(64,), residuals_dim=64, skips_dim=64)
        clips, cond_dim = 8, 0
    torch.manual_seed(1234)
    return mmk.WaveNet.from_config(cfg).eval(), clips, cond_dim


class WaveNetJob:
    unit = "audio samples/s"

    def __init__(self, args, device, rank):
        import mimikit_amd as mmk
        self.mmk = mmk
        self.net, clips, cond_dim = build_wavenet(args.workload)
        self.clips = args.clips or clips
        self.device = device
        self.rf = self.net.rf
        self.prompt_len = -(-self.rf // 16) * 16          # rf rounded up (cfg2: 1024, cfg4: 3072)
        self.n_steps = int(16000 * args.seconds)
        gen = torch.Generator().manual_seed(1234 + rank)
        audio = torch.rand(self.clips, self.prompt_len, generator=gen) * 2 - 1
        total = self.prompt_len + self.n_steps
        # (device_data: the strong-scaling leg - its inputs are drawn on the device, nothing of it is checked on the CPU)
        self.device_data = bool(getattr(args, "device_data", False))
        self.cond_dim, self.total = cond_dim, total
        self.cond_cpu = torch.rand(self.clips, total, cond_dim, generator=gen) if (cond_dim and not self.device_data) else None
        self.audio_cpu = audio
        self.expand = mmk.MuLawExpand(256)
        self.name = args.workload
        self.dtype = "f32"
        self.params = dict(temperature=(args.temperature,)) if getattr(args, "temperature", 0) > 0 else {}
        self.decode = f"sampled, temperature {args.temperature}" if self.params else "greedy"

    def to_device(self):
        self.net.to(self.device)
        prompt = self.mmk.MuLawCompress(256)(self.audio_cpu.to(self.device))
        self.idx = torch.cat([prompt, torch.zeros(self.clips, self.n_steps, dtype=torch.int64, device=self.device)], 1)
        self.cond = (self.cond_cpu.to(self.device),) if self.cond_cpu is not None else ()
        if self.device_data and self.cond_dim:
            self.cond = (torch.rand(self.clips, self.total, self.cond_dim, device=self.device),)
        self.prompt_cpu = prompt.cpu()

    def one_pass(self):
        p = self.prompt_len
        net = self.net
        net.before_generate((self.idx[:, :p], *[c[:, :p] for c in self.cond]), None)
        net.generate_block((self.idx, *self.cond), p, self.n_steps, **self.params)
        net.after_generate((self.idx,), None)
        self.audio_out = self.expand(self.idx)

    def units_per_pass(self):
        return self.clips * self.n_steps

    def config(self, world):
        c = self.net.config
        return {"workload": f"{self.name}: WaveNet blocks={tuple(c.blocks)} x {c.dims_dilated[0]} ch, "
                            f"{len(c.dims_1x1)} cond input(s), mu-law-256, 16 kHz",
                "clips_per_gpu": self.clips, "global_clips": self.clips * world, "prompt_samples": self.prompt_len,
                "generated_samples_per_clip": self.n_steps, "decode": self.decode, "parallelism": f"clip-shard x{world}"}

    # HBM roofline of the dominant kernel, measured with HIP events on its launches
    def roofline(self):
        p = self.prompt_len
        net, plan = self.net, self.net._plan
        net.before_generate((self.idx[:, :p], *[c[:, :p] for c in self.cond]), None)
        if plan.persistent:
            # ONE kernel runs every step of a block: HIP events around its launches on the launch stream
            n = min(self.n_steps, 1024)
            torch.cuda.synchronize()
            start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
            net.generate_block((self.idx, *self.cond), p, n)
            stop.record()
            torch.cuda.synchronize()
            net.after_generate((self.idx,), None)
            us = start.elapsed_time(stop) * 1e3
            nbytes = self.step_bytes() * n
            achieved = nbytes / (us * 1e-6) / 1e9
            kshort = ("wavenet_bpipe_kernel" if getattr(plan, "batch_pipelined", False) else "wavenet_spipe_pair_kernel" if getattr(plan, "pair_visits", False)
                      else "wavenet_spipe_kernel" if plan.stage_pipelined
                      else "wavenet_lpipe_kernel" if plan.layer_pipelined else "wavenet_chain_kernel" if plan.chain else "wavenet_persist_kernel")
            traffic, traffic_source = None, None
            try:  # PMC-derived HBM bytes of one 1024-step launch: NOT measured in this run (counters need their own rocprofv3
                  # --pmc passes) - taken from profiles/traffic.json, and only if that entry was collected on THIS kernel
                with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                    entries = json.load(f).get(self.name, {})
                    entry = entries.get(f"persistent_clips{int(self.clips)}") or entries.get("persistent", {})
                # ... and at THIS clip count (an entry without one was collected at the BASELINE share of 32 clips)
                if entry.get("bytes_per_step") and str(entry.get("kernel", "")).startswith(kshort) and int(entry.get("clips", 32)) == int(self.clips):
                    traffic = int(entry["bytes_per_step"] * n)
                    traffic_source = (f"profiles/traffic.json, build {entry.get('build')} (commit {entry.get('commit')}): rocprofv3 --pmc "
                                      f"FETCH_SIZE / WRITE_SIZE passes of {entry.get('kernel')}, scaled to {n} steps; not re-measured in this run")
            except (OSError, ValueError):
                pass
            kname = ("wavenet_bpipe_kernel (one layer per stage of 8 CUs, weights in registers as MFMA A operands, clips in groups of 16 per visit)" if getattr(plan, "batch_pipelined", False)
                     else "wavenet_spipe_pair_kernel (one layer per stage of 8 CUs, weights in registers, clips streamed through two at a time)" if getattr(plan, "pair_visits", False)
                     else "wavenet_spipe_kernel (one layer per stage of 8 CUs, weights in registers, clips streamed through one at a time)" if plan.stage_pipelined
                     else "wavenet_lpipe_kernel (four workgroups per clip that own whole layers, weights in registers)" if plan.layer_pipelined
                     else "wavenet_chain_kernel (one hand-off per layer)" if plan.chain else "wavenet_persist_kernel")
            tflops = self.step_flops() * n / (us * 1e-6) / 1e12
            if getattr(plan, "batch_pipelined", False):
                # 16 clips per visit on the matrix pipe: the weights never leave the registers and the arithmetic is what the launch does - the fp32
                # matrix peak bounds it (the HBM figure of the one-clip kernels stays in the line as hbm_nominal_frac)
                return {"bound": "mfma", "kernel": kname + ": all layers + head of every step of a block",
                        "achieved": round(tflops, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tflops / MFMA_F32_PEAK_TFLOPS, 5),
                        "traffic": traffic, "traffic_source": traffic_source, "algorithmic_flops_per_launch": self.step_flops() * n,
                        "algorithmic_bytes_per_launch": nbytes, "hbm_nominal_frac": round(achieved / HBM_PEAK_GBS, 5), "avg_launch_us": round(us, 1),
                        "launches_timed": 1, "steps_per_launch": n, "us_per_step_in_kernel": round(us / n, 2)}
            return {"bound": "hbm", "kernel": kname + ": all layers + head of every step of a block",
                    "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source,
                    "algorithmic_bytes_per_launch": nbytes, "avg_launch_us": round(us, 1), "launches_timed": 1,
                    "steps_per_launch": n, "us_per_step_in_kernel": round(us / n, 2), "fp32_tflops": round(tflops, 2)}
        stats = plan.profile_steps(self.idx, self.cond, p, 48)
        net.after_generate((self.idx,), None)
        c, B = plan.cfg, self.clips
        C, k = c.dim_dilated, c.kernel_size[0]
        k_a = k * C + sum(c.cond_dim[j] for j in range(c.n_cond))
        n_a = 2 * C if c.gated else C
        bytes_a = 4 * (n_a * k_a + n_a) + 4 * B * (k_a + C)
        n_b = (C if c.residuals_dim else 0) + c.skips_dim
        bytes_b = 4 * (n_b * C + n_b) + 4 * B * (C + 2 * n_b)
        per = {"layer_a": bytes_a, "layer_b": bytes_b}
        name = max(("layer_a", "layer_b"), key=lambda n: stats[n][0])
        ms, launches = stats[name]
        dur_us = 1e3 * ms / max(launches, 1)
        achieved = per[name] / (dur_us * 1e-6) / 1e9
        traffic = None
        try:  # PMC-derived HBM bytes per launch, collected in separate rocprofv3 --pmc passes (profiles/)
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                traffic = json.load(f).get(self.name, {}).get(name, {}).get("bytes")
        except (OSError, ValueError):
            pass
        return {"bound": "hbm", "kernel": f"linear_kernel ({name}: " +
                ("dilated taps + 1x1 cond + gate" if name == "layer_a" else "residual + skip 1x1") + ")",
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "algorithmic_bytes_per_launch": per[name], "avg_launch_us": round(dur_us, 3),
                "launches_timed": int(launches),
                "per_class_avg_us": {n: round(1e3 * stats[n][0] / max(stats[n][1], 1), 3) for n in stats}}

    def step_bytes(self):
        """algorithmic bytes of one auto-regressive step of the local batch (weights once + state)"""
        c = self.net._plan.cfg
        w = sum(p.numel() for n, p in self.net.named_parameters() if not n.startswith("input_modules.0."))
        w += c.dim_dilated                    # one embedding row per clip is negligible; count one
        state = self.clips * (2 * c.dim_dilated * 4 * c.n_layers + sum(c.cond_in_dim[j] for j in range(c.n_cond)) * 4)
        return 4 * w + state

    def step_flops(self):
        """algorithmic fp32 FLOPs of one auto-regressive step of the local batch: every weight of the network meets every clip once (2 per multiply-add) -
        the layers' dilated, 1x1, residual and skip convolutions, the conditioning LinearIO, the MLP head (wavenet_v2.py:131-182, mlp.py:58-63)"""
        w = sum(p.numel() for n, p in self.net.named_parameters() if p.dim() >= 2 and not n.startswith("input_modules.0."))
        return 2 * w * self.clips

    def cpu_baseline(self, budget_s):
        from oracle import torch_ref as O
        sd = {k: v.detach().cpu() for k, v in self.net.state_dict().items()}
        c = self.net._plan.cfg
        arch = dict(kernels=[c.kernel_size[i] for i in range(c.n_layers)],
                    dilations=[c.dilation[i] for i in range(c.n_layers)], has_skips=bool(c.skips_dim), residuals=True)
        b = min(self.clips, 2)
        prompt = self.prompt_cpu[:b]
        cond = [self.cond_cpu[:b]] if self.cond_cpu is not None else []
        all_cores = torch.get_num_threads()
        t0 = time.perf_counter()
        O.wavenet_generate(sd, prompt, [x[:, :self.prompt_len + 1] for x in cond], 1, **arch)
        one_all = time.perf_counter() - t0
        with cpu_threads():
            cores = torch.get_num_threads()
            t0 = time.perf_counter()
            O.wavenet_generate(sd, prompt, [x[:, :self.prompt_len + 1] for x in cond], 1, **arch)
            one = time.perf_counter() - t0
            n = max(2, min(64, int(budget_s / max(one, 1e-3))))
            t0 = time.perf_counter()
            out = O.wavenet_generate(sd, prompt, [x[:, :self.prompt_len + n] for x in cond], n, **arch)
            dt = time.perf_counter() - t0
        # the GPU run and the CPU run must agree on what they generated
        agree = bool((out[:, self.prompt_len:] == self.idx[:b, self.prompt_len:self.prompt_len + n].cpu()).all())
        return {"value": round(b * n / dt, 3), "unit": self.unit, "cores": cores, "kind": "port",
                "sample": f"{b} clips x {n} steps of the same network/prompt, naive full-window forward per step "
                          f"(reference algorithm), torch CPU fp32", "matches_gpu_output": agree,
                "all_cores": {"cores": all_cores, "value": round(b / one_all, 3), "sample": f"{b} clips x 1 step"}}


class SrnnJob:
    unit = "audio samples/s"

    def __init__(self, args, device, rank):
        import mimikit_amd as mmk
        self.mmk = mmk
        torch.manual_seed(1234)
        io = mmk.IOSpec.mulaw_io(mmk.IOSpec.MuLawIOConfig(sr=16000))
        self.net = mmk.SampleRNN.from_config(mmk.SampleRNN.Config(io_spec=io, frame_sizes=(16, 4, 1), hidden_dim=512,
                                                                  rnn_class="gru")).eval()
        self.clips, self.device = args.clips or 64, device
        self.prompt_len, self.n_steps = 512, int(16000 * args.seconds)
        gen = torch.Generator().manual_seed(1234 + rank)
        self.audio_cpu = torch.rand(self.clips, self.prompt_len, generator=gen) * 2 - 1
        self.expand = mmk.MuLawExpand(256)
        self.name, self.dtype = args.workload, "f32"
        self.params = dict(temperature=(args.temperature,)) if getattr(args, "temperature", 0) > 0 else {}
        self.decode = f"sampled, temperature {args.temperature}" if self.params else "greedy"

    def to_device(self):
        self.net.to(self.device)
        prompt = self.mmk.MuLawCompress(256)(self.audio_cpu.to(self.device))
        self.idx = torch.cat([prompt, torch.zeros(self.clips, self.n_steps, dtype=torch.int64, device=self.device)], 1)
        self.prompt_cpu = prompt.cpu()

    def one_pass(self):
        self.net.before_generate((self.idx[:, :self.prompt_len],), None)
        self.net.generate_block((self.idx,), self.prompt_len, self.n_steps, **self.params)
        self.net.after_generate((self.idx,), None)
        self.audio_out = self.expand(self.idx)

    def units_per_pass(self):
        return self.clips * self.n_steps

    def config(self, world):
        return {"workload": "srnn_cfg3: SampleRNN frame_sizes=(16,4,1), GRU hidden 512, mu-law-256, 16 kHz",
                "clips_per_gpu": self.clips, "global_clips": self.clips * world, "prompt_samples": self.prompt_len,
                "generated_samples_per_clip": self.n_steps, "decode": self.decode, "parallelism": f"clip-shard x{world}"}

    def step_bytes(self):
        """algorithmic bytes of one auto-regressive step of the local batch, amortised over the tier clocks: every
        tier's weights once per update (tier i fires every frame_sizes[i] steps, the bottom tier and the MLP every
        step) + the per-clip state it touches"""
        fs = (16, 4, 1)
        total = 0.0
        for n, p in self.net.named_parameters():
            if n.startswith("tiers."):
                i = int(n.split(".")[1])
                total += 4 * p.numel() / fs[i]
            else:
                total += 4 * p.numel()
        H = 512
        state = self.clips * 4 * (2 * H / 16 + 2 * H / 4 + 4 * H / 16 + 4 * H / 4 + H + 8)   # h in/out, up-sampled outputs, x
        return int(total + state)

    def roofline(self):
        """one generate block is ONE launch (srnn_resident_kernel: every tier, the bottom tier and the head resident, csrc/srnn_resident.hip) plus the
        class-ring fill in front of it and one up-sampler GEMM per tier behind it: HIP events around the whole block on the launch stream"""
        n = min(self.n_steps, 1600)
        self.net.before_generate((self.idx[:, :self.prompt_len],), None)
        torch.cuda.synchronize()
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        self.net.generate_block((self.idx,), self.prompt_len, n)
        stop.record()
        torch.cuda.synchronize()
        resident = self.net._plan.resident_blocks() > 0
        self.net.after_generate((self.idx,), None)
        us = start.elapsed_time(stop) * 1e3
        nbytes = self.step_bytes() * n
        achieved = nbytes / (us * 1e-6) / 1e9
        traffic, traffic_source = None, None
        try:  # PMC-derived HBM bytes per step (separate rocprofv3 --pmc passes over the same 1600-step block, profiles/): not measured in this run
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                entry = json.load(f).get("srnn_cfg3", {}).get("resident" if resident else "step_chain", {})
            if entry.get("bytes_per_step") and entry.get("clips", self.clips) == self.clips:
                traffic = int(entry["bytes_per_step"] * n)
                traffic_source = (f"profiles/traffic.json, build {entry.get('build')} (commit {entry.get('commit')}): rocprofv3 --pmc FETCH_SIZE / "
                                  f"WRITE_SIZE passes of {entry.get('kernel')}, {entry.get('clips', self.clips)} clips, scaled to {n} steps; not re-measured in this run")
        except (OSError, ValueError):
            pass
        kernel = ("srnn_resident_kernel<32, false> (all tiers + bottom tier + head, one launch per generate block)" if resident else
                  "SampleRNN step chain (srnn_bottom_kernel + srnn_gru_kernel, the kernels in turns), one generate block")
        return {"bound": "hbm", "kernel": kernel,
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes_per_launch": nbytes, "avg_launch_us": round(us, 1), "launches_timed": 1,
                "steps_per_launch": n, "us_per_step": round(us / n, 2)}

    def cpu_baseline(self, budget_s):
        from oracle import torch_ref as O
        sd = {k: v.detach().cpu() for k, v in self.net.state_dict().items()}
        o = O.SampleRNNOracle(sd, (16, 4, 1), 512, "gru")
        n = 400
        with cpu_threads():
            cores = torch.get_num_threads()
            t0 = time.perf_counter()
            out = o.generate(self.prompt_cpu, n)
            dt = time.perf_counter() - t0
        agree = bool((out[:, self.prompt_len:] == self.idx[:, self.prompt_len:self.prompt_len + n].cpu()).all())
        return {"value": round(self.clips * n / dt, 3), "unit": self.unit, "cores": cores,
                "kind": "port", "sample": f"{self.clips} clips x {n} steps (+ prompt warm-up), reference algorithm, torch CPU fp32",
                "matches_gpu_output": agree}


class S2SJob:
    unit = "audio samples/s"

    def __init__(self, args, device, rank):
        import mimikit_amd as mmk
        self.mmk = mmk
        torch.manual_seed(1234)
        io = mmk.IOSpec.magspec_io(mmk.IOSpec.MagSpecIOConfig(sr=22050, n_fft=1024, hop_length=256))
        self.net = mmk.Seq2SeqLSTMNetwork.from_config(mmk.Seq2SeqLSTMNetwork.Config(io_spec=io)).eval()
        self.clips, self.device = args.clips or 64, device
        self.n_steps = mmk.GenerateLoopV2.get_n_steps(mmk.GenerateLoopV2.Config(output_duration_sec=args.seconds), self.net)
        self.prompt_frames = 16
        gen = torch.Generator().manual_seed(1234 + rank)
        self.audio_cpu = torch.rand(self.clips, 1024 + 256 * (self.prompt_frames - 1), generator=gen) * 2 - 1
        self.name, self.dtype = args.workload, "f32"
        self.samples_per_frame = 256

    def to_device(self):
        self.net.to(self.device)
        prompt = self.mmk.MagSpec(1024, 256, center=False)(self.audio_cpu.to(self.device))
        assert prompt.shape[1] == self.prompt_frames
        self.frames = torch.cat([prompt, torch.zeros(self.clips, self.n_steps, 513, device=self.device)], 1).contiguous()

    def one_pass(self):
        self.net.before_generate((self.frames[:, :self.prompt_frames],), None)
        self.net.generate_block((self.frames,), self.prompt_frames, self.n_steps)
        self.net.after_generate((self.frames,), None)

    def units_per_pass(self):
        return self.clips * self.n_steps * self.samples_per_frame

    def config(self, world):
        return {"workload": "s2s_cfg5: Seq2Seq bi-LSTM D=1024 hop=8 on 1024-pt STFT magnitudes @22.05 kHz",
                "clips_per_gpu": self.clips, "global_clips": self.clips * world, "prompt_frames": self.prompt_frames,
                "generated_frames_per_clip": self.n_steps, "parallelism": f"clip-shard x{world}"}

    def step_flops(self):
        """2 FLOP per weight of the matrices a generate_step multiplies, per clip and per frame they are
        applied to: the four gate matrices of both directions of the two bi-LSTMs on all hop frames, the encoder output
        projection and the decoder up-sampler once, the output projection on all hop frames (497 MFLOP per clip-step)"""
        c = self.net.config
        D, hop, nb = c.model_dim, c.hop, 513
        macs = hop * 2 * 4 * D * (nb + D) + hop * 2 * 4 * D * (D + D) + D * D + D * hop * D + hop * D * nb
        return 2 * macs

    def roofline(self):
        """the path is dense fp32 contraction (AI ~ 200 FLOP/B at 64 clips): fp32 MFMA roofline over one generate block,
        HIP events on the launch stream"""
        self.net.before_generate((self.frames[:, :self.prompt_frames],), None)
        torch.cuda.synchronize()
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        self.net.generate_block((self.frames,), self.prompt_frames, self.n_steps)
        stop.record()
        torch.cuda.synchronize()
        self.net.after_generate((self.frames,), None)
        us = start.elapsed_time(stop) * 1e3
        calls = -(-self.n_steps // self.net.config.hop)
        flops = self.step_flops() * self.clips * calls
        achieved = flops / (us * 1e-6) / 1e12
        traffic, traffic_source = None, None
        try:   # HBM-side bytes of one generate_step from the PMC passes (profiles/traffic.json), times the block's generate_steps -
               # only if that entry was collected with the resident bi-LSTM kernel this build runs
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                entry = json.load(f).get(self.name, {}).get("generate_step", {})
            per_call = entry.get("bytes")
            if per_call and any("lstm_seq_kernel" in k for k in entry.get("kib_by_kernel", {})) and self.net._plan.resident_launches() > 0:
                traffic = int(per_call * calls)
                traffic_source = (f"profiles/traffic.json, build {entry.get('build')} (commit {entry.get('commit')}): rocprofv3 --pmc FETCH_SIZE / "
                                  "WRITE_SIZE passes over all kernels of a generate_step, times the block's generate_steps; not re-measured in this run")
        except (OSError, ValueError, AttributeError):
            traffic, traffic_source = None, None
        return {"bound": "mfma", "kernel": "Seq2Seq generate block: lstm_inproj_kernel (input half of a bi-LSTM layer, W_ih in registers) + "
                                           "lstm_seq_kernel (all frames of a layer in one launch, W_hh in registers) + skinny_linear_kernel / "
                                           "gemm_bias_act_kernel (up-sampler, output projection), all generate_steps of one block",
                "achieved": round(achieved, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / MFMA_F32_PEAK_TFLOPS, 5), "traffic": traffic, "traffic_source": traffic_source,
                "algorithmic_flops_per_launch": flops, "avg_launch_us": round(us, 1), "launches_timed": 1,
                "generate_steps_per_launch": calls, "us_per_generate_step": round(us / calls, 2)}

    def cpu_baseline(self, budget_s):
        from oracle import torch_ref as O
        sd = {k: v.detach().cpu() for k, v in self.net.state_dict().items()}
        b = 4
        x_dev = self.frames[:b, self.prompt_frames - 8:self.prompt_frames].contiguous()
        got = self.net.generate_step((x_dev,), t=self.prompt_frames).cpu()
        x = x_dev.cpu()
        want = O.s2s_step(sd, x, 8)
        agree = bool(float((got - want).abs().max()) <= 1e-4 * float(want.abs().max()))
        with cpu_threads():
            t0 = time.perf_counter()
            n = 0
            while time.perf_counter() - t0 < min(budget_s, 10.0):
                O.s2s_step(sd, x, 8)
                n += 1
            dt = time.perf_counter() - t0
            cores = torch.get_num_threads()
        return {"value": round(b * n * 8 * 256 / dt, 3), "unit": self.unit, "cores": cores,
                "kind": "port", "sample": f"{b} clips x {n} generate_steps (8 frames each), torch CPU fp32",
                "matches_gpu_output": agree}


class FeatureJob:
    """the feature functionals either side of the networks: mu-law, MagSpec, ISTFT, Griffin-Lim"""

    def __init__(self, args, device, rank):
        import mimikit_amd as mmk
        self.mmk, self.device, self.name, self.dtype = mmk, device, args.workload, "f32"
        gen = torch.Generator().manual_seed(1234 + rank)
        w = args.workload
        if w == "mulaw":
            self.x_cpu = torch.rand(64, 16000 * 60, generator=gen) * 2 - 1
            self.unit, self.what = "audio samples/s", "MuLawCompress(256) on (64, 960000) fp32"
        elif w == "stft":
            self.x_cpu = torch.randn(64, 22050 * 10, generator=gen)
            self.unit, self.what = "frames/s", "MagSpec(1024, 256) on (64, 220500) fp32"
        elif w == "istft":      # 10 s of 22.05 kHz audio per clip as (abs, angle) frames
            self.x_cpu = torch.stack((torch.rand(64, 862, 513, generator=gen), (torch.rand(64, 862, 513, generator=gen) * 2 - 1) * math.pi), -1)
            self.unit, self.what = "frames/s", "ISTFT(1024, 256, 'pol') on (64, 862, 513, 2) fp32"
        else:                   # gla: magnitudes of 10 s clips, torchaudio defaults (32 iterations, momentum 0.99)
            self.x_cpu = torch.rand(64, 862, 513, generator=gen)
            self.unit, self.what = "frames/s", "GLA(1024, 256): 32 Griffin-Lim iterations on (64, 862, 513) fp32 magnitudes"

    def to_device(self):
        mmk = self.mmk
        self.x = self.x_cpu.to(self.device)
        self.f = {"mulaw": mmk.MuLawCompress(256), "stft": mmk.MagSpec(1024, 256, center=False), "istft": mmk.ISTFT(1024, 256, "pol"),
                  "gla": mmk.GLA(1024, 256)}[self.name]

    def one_pass(self):
        self.out = self.f(self.x)

    def units_per_pass(self):
        if self.name == "mulaw":
            return self.x.numel()
        if self.name == "stft":
            return self.out.shape[0] * self.out.shape[1]
        return self.x.shape[0] * self.x.shape[1]

    def config(self, world):
        return {"workload": f"{self.name}: {self.what}", "parallelism": f"replicas x{world}"}

    def roofline(self):
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20 if self.name != "gla" else 3
        start.record()
        for _ in range(n):
            self.one_pass()
        stop.record()
        torch.cuda.synchronize()
        us = start.elapsed_time(stop) * 1e3 / n
        frames = self.units_per_pass()
        # algorithmic bytes: what any implementation must read and write
        if self.name == "mulaw":
            nbytes, kernel = self.x.numel() * 12, "mulaw_compress_stream_kernel"
        elif self.name == "stft":
            nbytes, kernel = frames * (4 * 256 + 4 * 513), "stft1024_kernel"
        elif self.name == "istft":      # a frame's complex bins in, hop samples out
            nbytes, kernel = frames * (8 * 513 + 4 * 256), "istft1024q_kernel"
        else:   # per iteration: magnitudes, previous spectrum and waveform in; spectrum and waveform out (the phase estimates
            # themselves never need to exist in HBM); plus the first inverse transform of mag x initial phases
            it = 32
            nbytes = frames * (it * (20 * 513 + 2 * 4 * 256) + 12 * 513 + 4 * 256)
            kernel = "Griffin-Lim chain: istft1024q_kernel + 32 x gla1024q_iter_kernel (stft -> phase update -> istft per launch)"
        achieved = nbytes / (us * 1e-6) / 1e9
        traffic = None
        try:  # PMC-derived HBM bytes per launch of this workload's shapes (separate rocprofv3 --pmc passes, profiles/)
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                traffic = json.load(f).get(self.name, {}).get("launch", {}).get("bytes")
        except (OSError, ValueError):
            pass
        return {"bound": "hbm", "kernel": kernel,
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic, "algorithmic_bytes_per_launch": nbytes, "avg_launch_us": round(us, 2)}

    def cpu_baseline(self, budget_s):
        from oracle import torch_ref as O
        x = self.x_cpu[:8]
        fn = {"mulaw": lambda: O.mulaw_compress(x), "stft": lambda: O.magspec(x, 1024, 256, False),
              "istft": lambda: O.istft(x, 1024, 256, "pol"),
              "gla": lambda: O.griffin_lim(x, 1024, 256, 32, 0.99, torch.rand(x.shape, dtype=torch.complex64))}[self.name]
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < min(budget_s, 5.0):
            out = fn()
            n += 1
        dt = time.perf_counter() - t0
        units = x.numel() if self.name == "mulaw" else (out.shape[0] * out.shape[1] if self.name == "stft" else x.shape[0] * x.shape[1])
        return {"value": round(units * n / dt, 1), "unit": self.unit, "cores": torch.get_num_threads(), "kind": "port",
                "sample": f"8 of the 64 rows, {n} repetitions, torch CPU fp32"}


class StubJob:
    """No device work: a tiny module to broadcast and a pass that sleeps longer on higher ranks.  Lets the multi-rank plumbing of
    this file (own launcher, rendezvous, ONE broadcast, fenced max-over-ranks clock, rank 0's line) run under gloo on a box
    without GPUs - tests/test_shard_gloo.py drives `bench.py --workload stub --gpus 2`."""
    unit = "stub units/s"

    def __init__(self, args, device, rank):
        torch.manual_seed(900 + rank)                 # ranks start from different weights
        self.net = torch.nn.Linear(8, 8)
        self.clips, self.rank, self.dtype = args.clips or 4, rank, "f32"
        self.pass_s = 0.02 * (1 + rank)

    def to_device(self):
        pass

    def one_pass(self):
        time.sleep(self.pass_s)

    def units_per_pass(self):
        return self.clips

    def config(self, world):
        import torch.distributed as dist
        w = torch.cat([p.detach().reshape(-1) for p in self.net.parameters()]).double().sum().reshape(1)
        lo, hi = w.clone(), w.clone()
        if world > 1:                                 # (every rank builds the line: a collective here is matched)
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        return {"workload": "stub (no device work)", "clips_per_gpu": self.clips, "global_clips": self.clips * world,
                "parallelism": f"clip-shard x{world}", "weights_checksum": float(w), "weights_checksum_spread": float(hi - lo),
                "broadcasts": getattr(self, "broadcasts", 0), "slowest_pass_s": self.pass_s}

    def roofline(self):
        return None


JOBS = {"stub": StubJob, "wavenet_cfg4": WaveNetJob, "wavenet_cfg2": WaveNetJob, "srnn_cfg3": SrnnJob, "s2s_cfg5": S2SJob,
        "mulaw": FeatureJob, "stft": FeatureJob, "istft": FeatureJob, "gla": FeatureJob}


# ----------------------------------------------------------------------------- main
def visible_gpus():
    """GPUs of this node WITHOUT touching HIP (the launcher must not initialise the GPU it hands to its ranks): the KFD topology nodes with
    SIMDs, cut down by a ROCR / HIP visible-devices list; 0 where there is no KFD topology (no AMD GPU driver on this node)"""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if os.environ.get(var, "") != "":
            return len([d for d in os.environ[var].split(",") if d.strip() != ""])
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        count = 0
        for node in os.listdir(root):
            with open(os.path.join(root, node, "properties")) as f:
                props = dict(line.split(None, 1) for line in f.read().splitlines() if " " in line)
            if int(props.get("simd_count", "0")) > 0:
                count += 1
        return count
    except (OSError, ValueError):
        return 0


def launch_ranks(args) -> int:
    """`bench.py --gpus N` without a launcher around it (WORLD_SIZE unset): start N ranks of this file, one per GPU, and pass rank 0's
    JSON line on.  The parent makes no HIP call at all (the GPUs are counted from sysfs) and never re-execs; fewer than N visible
    devices, or any rank failing, is a non-zero exit - never a silent single-GPU run.  Every rank is polled: the first one that fails
    ends the others (a rank waiting in a collective for a dead peer would otherwise sit there until the process group times out)."""
    import socket
    import subprocess
    import threading
    import time
    n = args.gpus
    stub = args.workload == "stub"
    if not stub:
        have = visible_gpus()
        if have < n:
            print(f"bench.py: --gpus {n} but only {have} GPU(s) are visible on this node", file=sys.stderr)
            return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # (dmabuf IPC: what RCCL needs on this driver)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    codes = [None] * n
    while any(c is None for c in codes):
        for r, pr in enumerate(procs):
            if codes[r] is None:
                codes[r] = pr.poll()
        if any(c not in (None, 0) for c in codes):
            for r, pr in enumerate(procs):
                if codes[r] is None:
                    pr.kill()
                    codes[r] = pr.wait()
            break
        time.sleep(0.05)
    reader.join(timeout=5)
    if any(codes):
        print(f"bench.py: rank exit codes {codes}", file=sys.stderr)
        return 1
    sys.stdout.write("".join(out0))
    sys.stdout.flush()
    return 0


STRONG_GLOBAL_CLIPS = 256      # BASELINE config 4's whole job (256 clips, partitioned over the ranks)


def strong_scaling_leg(args, device, rank, world, sync):
    """The SAME total job on any number of GPUs: BASELINE config 4's 256 clips divided over the ranks (R = 1: 256 clips on this GPU - groups of 16
    on the matrix pipe; R = 8: 32 per GPU, the weak-scaling line's share).  After the timed region, same fences and --seconds of audio, 1 + 2 passes;
    every rank runs it, rank 0 reports it as an extra key."""
    import copy
    from mimikit_amd.shard import timed_passes, clip_slice
    a2 = copy.copy(args)
    lo, hi = clip_slice(STRONG_GLOBAL_CLIPS, rank, world)
    a2.clips, a2.device_data = hi - lo, True
    job = WaveNetJob(a2, device, rank)
    job.to_device()
    steps, warmup = 2, 1
    elapsed = timed_passes(job.one_pass, steps, warmup, sync)
    plan = job.net._plan
    kernel = ("wavenet_bpipe_kernel" if getattr(plan, "batch_pipelined", False) else "wavenet_spipe_pair_kernel" if getattr(plan, "pair_visits", False)
              else "wavenet_spipe_kernel" if plan.stage_pipelined else "other")
    out = {"global_clips": STRONG_GLOBAL_CLIPS, "n_gpus": world, "clips_on_rank_0": hi - lo, "generated_samples_per_clip": job.n_steps,
           "steps": steps, "warmup": warmup, "value": round(STRONG_GLOBAL_CLIPS * job.n_steps * steps / elapsed, 1), "unit": job.unit,
           "us_per_ar_step": round(1e6 * elapsed / (steps * job.n_steps), 2), "kernel": kernel, "decode": job.decode}
    del job
    torch.cuda.empty_cache()
    return out


OTHER_WORKLOADS = (("wavenet_cfg2", 2, 1), ("srnn_cfg3", 3, 1), ("s2s_cfg5", 5, 2), ("mulaw", 10, 2), ("stft", 10, 2), ("istft", 10, 2), ("gla", 2, 1))


def other_workloads(args, device, sync):
    """Every other BASELINE config and the feature kernels on the SAME clock as the headline: after the timed region of the default run (N = 1), one
    short fenced run of each - the same Job classes `--workload NAME` runs, the same fences (shard.timed_passes), its dominant kernel's roofline from
    HIP events, a small CPU sample of the oracle - so that the figures DESIGN.md quotes for them are lines the driver saw.  ~40 s in all."""
    import copy
    from mimikit_amd.shard import timed_passes
    out = {}
    for name, steps, warmup in OTHER_WORKLOADS:
        a2 = copy.copy(args)
        a2.workload, a2.clips, a2.seconds, a2.temperature = name, 0, 1.0, 0.0
        t_start = time.perf_counter()
        try:
            job = JOBS[name](a2, device, 0)
            job.to_device()
            elapsed = timed_passes(job.one_pass, steps, warmup, sync)
            entry = {"value": round(job.units_per_pass() * steps / elapsed, 1), "unit": job.unit, "steps": steps, "warmup": warmup,
                     "ms_per_step": round(1e3 * elapsed / steps, 3), "config": job.config(1)}
            if hasattr(job, "clips") and hasattr(job, "n_steps") and name != "s2s_cfg5":
                entry["us_per_ar_step"] = round(1e6 * elapsed / (steps * job.n_steps), 3)
            roof = job.roofline()
            entry["roofline"] = {k: roof[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us") if k in roof}
            for k in ("us_per_step_in_kernel", "us_per_step", "us_per_generate_step"):
                if k in roof:
                    entry["roofline"][k] = roof[k]
            cpu = job.cpu_baseline(3.0)
            entry["cpu_baseline"] = {k: cpu[k] for k in ("value", "unit", "cores", "kind", "sample", "matches_gpu_output") if k in cpu}
        except Exception as e:      # (a failing side workload must not take the headline line with it: it is reported as what it is)
            entry = {"error": f"{type(e).__name__}: {e}"[:300]}
        entry["wall_s"] = round(time.perf_counter() - t_start, 2)
        out[name] = entry
        job = None
        torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist
    stub = args.workload == "stub"
    if stub:
        device, sync = torch.device("cpu"), (lambda: None)
        if world > 1:
            dist.init_process_group(backend="gloo")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs the MI355X: the generate path has no CPU implementation")
        if torch.cuda.device_count() <= local_rank:
            raise SystemExit(f"rank {rank}: no GPU {local_rank} on this node ({torch.cuda.device_count()} visible)")
        torch.cuda.set_device(local_rank)
        device, sync = torch.device("cuda", local_rank), torch.cuda.synchronize
        if world > 1 or args.force_dist:
            if world == 1:                                                  # (--force-dist: a process group of one rank, on a free local port)
                import socket
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                if "MASTER_PORT" not in os.environ:
                    with socket.socket() as sock:
                        sock.bind(("127.0.0.1", 0))
                        os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
                os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group(backend="nccl", device_id=device, rank=rank, world_size=world)     # RCCL over xGMI
            if args.force_dist:
                from mimikit_amd import shard
                shard.FORCE_COLLECTIVES = True
    torch.set_grad_enabled(False)
    apply_tuning(args.tuning)

    job = JOBS[args.workload](args, device, rank)
    job.to_device()
    if (world > 1 or args.force_dist) and hasattr(job, "net"):
        from mimikit_amd.shard import broadcast_weights
        broadcast_weights(job.net, src=0)          # the path's only collective
        job.broadcasts = 1

    from mimikit_amd.shard import timed_passes
    elapsed = timed_passes(job.one_pass, args.steps, args.warmup, sync)   # barrier + sync, MAX over ranks

    units = job.units_per_pass() * args.steps * world
    value = units / elapsed
    strong = strong_scaling_leg(args, device, rank, world, sync) if (args.workload == "wavenet_cfg4" and not args.no_strong_leg) else None
    line = {
        "metric": "audio samples/sec generated (WaveNet 256-ch mu-law, 16kHz)" if args.workload == "wavenet_cfg4"
        else f"{job.unit} ({args.workload})",
        "value": round(value, 1), "unit": job.unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": job.dtype, "data": "synthetic", "config": job.config(world),
    }
    if args.force_dist and not stub:
        line["collectives"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "weight_broadcasts": getattr(job, "broadcasts", 0),
                               "clock": "barrier + all_reduce(MAX) on the device"}
    if rank == 0:
        if hasattr(job, "step_bytes"):
            steps_per_s = value / (job.clips * world)
            roof = job.clips * HBM_PEAK_GBS * 1e9 / job.step_bytes()
            line["ar_steps_per_s_per_gpu"] = round(steps_per_s, 1)
            line["us_per_ar_step"] = round(1e6 / steps_per_s, 2)
            line["whole_step_hbm_roofline"] = {"samples_per_s_per_gpu_at_peak": round(roof, 1),
                                               "algorithmic_bytes_per_step": job.step_bytes(),
                                               "frac": round(value / world / roof, 5)}
        roof = job.roofline()
        if roof is not None:
            line["roofline"] = roof
        if strong is not None:
            line["strong_scaling"] = strong
        if world == 1 and not args.no_cpu_baseline and not args.temperature > 0:    # N = 1 only; the CPU leg re-checks the GREEDY samples
            line["cpu_baseline"] = job.cpu_baseline(args.cpu_seconds)
        if world == 1 and args.workload == "wavenet_cfg4" and not args.no_others and not args.clips and not args.temperature > 0 and not args.tuning:
            line["other_workloads"] = other_workloads(args, device, sync)
        print(json.dumps(line), flush=True)
    if world > 1 or (args.force_dist and dist.is_initialized()):
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
