import sys, os, time, torch
sys.path.insert(0, os.getcwd())
import bench
torch.set_grad_enabled(False)
class A: workload="wavenet_cfg2"; clips=0; seconds=1.0; temperature=0.0
job = bench.WaveNetJob(A, torch.device("cuda",0), 0)
job.to_device()
job.one_pass(); torch.cuda.synchronize()
net, p = job.net, job.prompt_len
for rep in range(2):
    torch.cuda.synchronize(); t0=time.perf_counter()
    net.before_generate((job.idx[:, :p],), None); torch.cuda.synchronize(); t1=time.perf_counter()
    net.generate_block((job.idx,), p, job.n_steps); torch.cuda.synchronize(); t2=time.perf_counter()
    net.after_generate((job.idx,), None); torch.cuda.synchronize(); t3=time.perf_counter()
    job.one_pass(); torch.cuda.synchronize(); t4=time.perf_counter()
    print(f"before_generate {1e3*(t1-t0):.2f} ms, generate_block {1e3*(t2-t1):.2f} ms, after_generate {1e3*(t3-t2):.2f} ms, one_pass {1e3*(t4-t3):.2f} ms")
