import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from prosstt_amd import device
from tests.test_gpu_sampler import synthetic
ctx = device.get_context()
rows, G, N = 400, 20000, 50000
means, roc, sc, al, be = synthetic(1, rows, G, N)
dm = ctx.tensor(means, torch.float32); dr = ctx.tensor(roc, torch.int32); ds = ctx.tensor(sc, torch.float64)
da = ctx.tensor(al, torch.float64); db = ctx.tensor(be, torch.float64)
out = torch.empty((N, G), dtype=torch.int32, device='cuda')
mu, p, r, path = ctx.nb_params(dm[:, :], dr[:2000], ds[:2000], da, db)
print('heavy frac', float((path == 2).float().mean()), 'mu median', float(mu.median()), 'mean', float(mu.mean()))
del mu, p, r, path
for i in range(3):
    ctx.sample_counts(dm, dr, ds, da, db, seed=i, out=out, check_domain=False, time_kernel=True)
    ms = ctx.last_kernel_ms()
    print(f'kernel {ms:.3f} ms  {N*G/ms/1e6:.1f} G samples/s  {N*G*4/ms/1e9*1e3/8e12*100:.2f}% of 8TB/s')
print('zeros frac', float((out[:2000] == 0).float().mean()))
