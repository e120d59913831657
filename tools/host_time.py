import time, torch, sys
sys.path.insert(0, '.')
from vbnn_amd.engine import FusedMLP
opt = dict(var_init=1e-3, mu_init=1, B=1e6, S=1, mode="lrt", dtype="f32", seed=1, input_size=784, hidden=[400, 400], n_classes=10, type="vb", fuse_kl=True)
eng = FusedMLP(opt)
N = 256
x = torch.randn(N, 784, device="cuda"); t = (torch.arange(N, device="cuda") % 10).to(torch.int32)
def step():
    eng.resetGradients(); eng.sample(); eng.run(x, t); eng.finish()
for _ in range(50): step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(2000): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"issue {1e6*(t1-t0)/2000:.1f} us/step, issue+drain {1e6*(t2-t0)/2000:.1f} us/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
