import sys, os, time, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import helpers as H, mimikit_amd as mmk
from oracle.weights import load_recipe
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
for blocks in [(4,), (6,), (8,), (10,), (5, 6)]:
    net = mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=H.mu_emb(mlp_dim=128), blocks=blocks, dims_dilated=(64,), residuals_dim=64, skips_dim=64)).eval()
    load_recipe(net, seed=3, gain=2.0)
    net = net.to(dev)
    B, P, n = 8, net.rf + 3, 4096
    idx = torch.cat([torch.randint(0, 256, (B, P)), torch.zeros(B, n, dtype=torch.int64)], 1).to(dev)
    net.generate_block((idx,), P, 1024)          # warm
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    net.generate_block((idx,), P + 1024, 3072)
    b.record(); torch.cuda.synchronize()
    net.after_generate((idx,), None)
    print(blocks, sum(blocks), "layers:", round(a.elapsed_time(b) * 1e3 / 3072, 2), "us per step", net._plan.layer_pipelined)
