"""Diagnostic: kernel times of the spectral functionals per n_fft (64 clips x 10 s at 22.05 kHz, hop = n_fft / 4)."""
import math, sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mimikit_amd as mmk
from mimikit_amd import native
dev = torch.device("cuda", 0)
x = torch.randn(64, 220500, device=dev)

def t(f, n=10):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n

for n_fft in (512, 1024, 2048, 4096):
    hop = n_fft // 4
    mag = mmk.MagSpec(n_fft, hop, center=True)(x)
    pol = mmk.STFT(n_fft, hop, "pol", center=True)(x)
    us_stft = t(lambda: mmk.MagSpec(n_fft, hop, center=True)(x))
    us_istft = t(lambda: mmk.ISTFT(n_fft, hop, "pol")(pol))
    us_gla = t(lambda: mmk.GLA(n_fft, hop)(mag), n=2)
    print(f"n_fft {n_fft}: frames {mag.shape[1]}  MagSpec {us_stft:.0f} us  ISTFT {us_istft:.0f} us  GLA(32) {us_gla / 1e3:.1f} ms")
