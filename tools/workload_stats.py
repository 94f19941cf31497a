import sys; sys.path.insert(0, '.')
import numpy as np, torch
from prosstt_amd import device, workloads
ctx = device.get_context()
w = workloads.build(sys.argv[1] if len(sys.argv) > 1 else "C3")
pt, br, sc, rows = w.plan(2000)
mu, p, r, path = ctx.nb_params(w.tree.device_means(), rows, sc, w.alpha, w.beta)
mu = mu.flatten(); path = path.flatten(); theta = (p/(1-p)).flatten()
print("paths: light %.4f heavy %.4f invalid %.4f" % tuple(float((path==k).float().mean()) for k in (1,2,0)))
ms = mu[::7].float()
print("mu: median %.3f mean %.3f p90 %.3f p99 %.3f max %.1f" % (float(ms.median()), float(mu.mean()), float(ms.quantile(0.9)), float(ms.quantile(0.99)), float(mu.max())))
X = ctx.sample_counts(w.tree.device_means(), rows, sc, w.alpha, w.beta, seed=1).flatten()
light = path==1
print("zeros overall %.4f; among light: k=0 %.4f k<=1 %.4f k<=2 %.4f mean k %.3f; mean k | k>=1 %.3f" % (
  float((X==0).float().mean()), float((X[light]==0).float().mean()), float((X[light]<=1).float().mean()), float((X[light]<=2).float().mean()),
  float(X[light].float().mean()), float(X[light][X[light]>=1].float().mean())))
print("heavy: r<1 frac %.4f, lambda>=10 approx (mu>=10) %.4f" % (float((r.flatten()[path==2]<1).float().mean()), float((mu[path==2]>=10).float().mean())))
Xl = X[light]; nz = Xl[Xl >= 1]
print("light nonzero: " + "  ".join("k<=%d %.3f" % (k, float((nz <= k).float().mean())) for k in (1, 2, 3, 4, 6, 8, 12, 16)))
print("stage-3 passes per survivor at s steps per pass (after settling k<=c in stage 2):")
for c in (0, 1, 2):
    rest = nz[nz > c].float()
    for st in (2, 3, 4):
        print("  settle k<=%d, %d steps/pass: to stage 3 %.3f of survivors, passes per stage-3 entry %.2f, per survivor %.3f" % (
            c, st, rest.numel() / nz.numel(), float(torch.ceil((rest - c) / st).mean()), float(torch.ceil((rest - c) / st).sum()) / nz.numel()))
