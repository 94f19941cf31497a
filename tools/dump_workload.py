"""Dump a benchmark workload (mean tensor subsample, plan, NB parameters) for offline sampler design.
   python tools/dump_workload.py C3 gpurun_out/c3_dump.npz [genes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from prosstt_amd import workloads
cfg, path = sys.argv[1], sys.argv[2]
ng = int(sys.argv[3]) if len(sys.argv) > 3 else 6000
w = workloads.build(cfg, verbose=True)
pt, br, sc, rows = w.plan()
M = w.tree.device_means().cpu().numpy()
np.savez_compressed(path, means=M[:, :ng].astype(np.float32), rows=rows.astype(np.int32), sc=sc, alpha=w.alpha[:ng], beta=w.beta[:ng],
                    gene_mean_all=M.mean(axis=0), gene_max_all=M.max(axis=0), alpha_all=w.alpha, beta_all=w.beta)
print("dumped", M.shape, "->", path, os.path.getsize(path) / 1e6, "MB", w.info)
