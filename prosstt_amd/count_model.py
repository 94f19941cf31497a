"""
The UMI count model, mirroring ``prosstt.count_model``
(reference: /root/reference/prosstt/count_model.py).

On the hot path the mean/variance -> (p, r) conversion is fused into the HIP
count sampler; ``get_pr_umi`` below is the reference's stand-alone formula for
users who want the parameters themselves.  ``sample_counts`` is NEW: the name
BASELINE.json gives the fused sampler (the reference spells the same operation
``simulation.draw_counts``).  The reference's unused analytic pmf classes
(count_model.py:51-128, 164-228; dead code there) are not restated.
"""
import numpy as np
from numpy import random

from . import device as _device
from .device import HOST_OUTS as _HOST_OUTS, OUT_CHOICES as _OUT_CHOICES, host_return as _host_return


def generate_negbin_params(tree, mean_alpha=0.2, mean_beta=2, a_scale=1.5, b_scale=1.5):
    """Per-gene (alpha, beta) (count_model.py:14-48).  ``log(scale)`` is used as a
    standard deviation, as in the reference (:43, :45); same two normal draws of G."""
    alphas = np.exp(random.standard_normal(tree.G) * np.log(a_scale) + np.log(mean_alpha))
    betas = np.exp(random.standard_normal(tree.G) * np.log(b_scale) + np.log(mean_beta)) + 1
    return alphas, betas


def get_pr_umi(a, b, m):
    """Negative-binomial (p, r) from mean m and variance a*m^2 + b*m (count_model.py:131-161)."""
    a, b, m = np.asarray(a, dtype=float), np.asarray(b, dtype=float), np.asarray(m, dtype=float)
    with np.errstate(divide="ignore", invalid="ignore"):
        s2 = a * m ** 2 + b * m
        p = np.array((s2 - m) / s2, dtype=float)
        r = np.array((m ** 2) / (s2 - m), dtype=float)
    p[s2 <= 0] = 0
    r[s2 <= 0] = 0
    return p, r


def sample_counts(mu, alpha, beta, *, seed=None, out="numpy", strict=True):
    """Counts X[n, g] ~ NB(mean mu[n, g], variance alpha[g]*mu^2 + beta[g]*mu) on the device.

    ``mu`` is the (N, G) matrix of per-cell means (what simulation.py:633-640 builds);
    the fused sampler is run with one mean row per cell and unit scalings.
    """
    mu = np.ascontiguousarray(mu, dtype=np.float32)
    if mu.ndim != 2:
        raise ValueError("mu must be (cells, genes)")
    N, G = mu.shape
    if seed is None:
        lo, hi = random.randint(0, 2 ** 32, size=2, dtype=np.uint64)
        seed = int(lo) | (int(hi) << 32)
    ctx = _device.get_context()
    counts = ctx.sample_counts(mu, np.arange(N, dtype=np.int32), np.ones(N),
                               np.broadcast_to(np.asarray(alpha, dtype=np.float64), (G,)),
                               np.broadcast_to(np.asarray(beta, dtype=np.float64), (G,)),
                               seed=seed, check_domain=strict)
    if out == "torch":
        return counts
    if out not in _HOST_OUTS:
        raise ValueError(_OUT_CHOICES)
    return _host_return(counts, out)
