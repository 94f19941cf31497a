"""
-m gpu: the law where binary32 is weakest -- tiny means.  P(X = 0) is a binary32 number compared with a 24-bit-rounded
uniform (absolute error <= 2e-7); for means of 1e-4 .. 1e-2 that is 0.002 % .. 0.3 % of P(X >= 1).
Over 1e10 draws per parameter set, #(X >= 1) and #(X >= 2) are held against the binary64 pmf of the reference's law
(count_model.get_pr_umi, /root/reference/prosstt/count_model.py:156-160: NB(n = r, p = 1 - p), theta = a m + b - 1): within
5 sigma of the sampling error PLUS the definition's stated resolution -- every threshold of the walk is compared with a
uniform that carries 24 significant bits, 2^-24 of probability near w = 2^32, where the thresholds of a tiny mean lie.
Measured (round 6, 1.015e10 draws per set): #(X >= 1) within 1.5 sigma for every set (3.8e-4 relative at m = 1e-4);
#(X >= 2) within 2 sigma except at the Poisson limit with m = 1e-4, where P(X >= 2) = 5.0e-9 comes out as 3.1e-8: an
absolute 2.6e-8, inside the 2^-24 = 6.0e-8 the definition states (a walk on the complement of the uniform with a polynomial
expm1 would remove it for ~10 vector instructions per stage-2 pass, 1.5 % of the kernel: not taken -- DESIGN.md section 4).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SETS = [(1e-4, 0.2, 2.0), (1e-3, 0.2, 2.0), (1e-2, 0.2, 2.0),
        (1e-4, 0.0, 1.0 + 1e-8), (1e-3, 0.0, 1.0 + 1e-8), (1e-2, 0.0, 1.0 + 1e-8)]      # a = 0, b = 1 + 1e-8: the Poisson limit


def exact_tail(m, a, b):
    """P(X >= 1), P(X >= 2) of NB(mean m, theta = a m + b - 1) in binary64."""
    theta = a * m + (b - 1.0)
    log_p0 = -(m / theta) * np.log1p(theta)
    p_ge1 = -np.expm1(log_p0)
    p1 = np.exp(log_p0) * m / (1.0 + theta)
    return p_ge1, p_ge1 - p1


def test_counts_of_tiny_means_against_the_binary64_pmf(capsys):
    import torch
    from prosstt_amd import device
    ctx = device.get_context()
    N, per = 50000, 3328
    G = per * len(SETS)
    means = np.concatenate([np.full(per, m, np.float32) for m, _, _ in SETS])[None, :]
    alpha = np.concatenate([np.full(per, a) for _, a, _ in SETS])
    beta = np.concatenate([np.full(per, b) for _, _, b in SETS])
    d_means = ctx.tensor(means, torch.float32)
    d_rows = ctx.tensor(np.zeros(N, np.int32), torch.int32)
    d_sc = ctx.tensor(np.ones(N), torch.float64)
    d_al, d_be = ctx.tensor(alpha, torch.float64), ctx.tensor(beta, torch.float64)
    out = torch.empty((N, G), dtype=torch.int32, device=ctx.torch_device)
    launches = 61                                            # 61 x 50 000 x 3 328 = 1.015e10 draws per set
    ge1 = torch.zeros(len(SETS), dtype=torch.int64, device=out.device)
    ge2 = torch.zeros_like(ge1)
    for i in range(launches):
        ctx.sample_counts(d_means, d_rows, d_sc, d_al, d_be, seed=20260000 + i, cell_offset=i * N, out=out, check_domain=False)
        blocks = out.view(N, len(SETS), per)
        ge1 += (blocks >= 1).sum(dim=(0, 2))
        ge2 += (blocks >= 2).sum(dim=(0, 2))
    n = launches * N * per
    ge1, ge2 = ge1.cpu().numpy(), ge2.cpu().numpy()
    lines, worst = [], 0.0
    grain = n * 2.0 ** -24                                   # the stated resolution of a threshold, in draws
    for k, (m, a, b) in enumerate(SETS):
        p1, p2 = exact_tail(m, a, b)
        s1, s2 = np.sqrt(n * p1 * (1 - p1)), np.sqrt(n * p2 * (1 - p2))
        z1, z2 = (ge1[k] - n * p1) / s1, (ge2[k] - n * p2) / s2
        worst = max(worst, abs(ge1[k] - n * p1) / (5.0 * s1 + grain), abs(ge2[k] - n * p2) / (5.0 * s2 + grain))
        lines.append("m = %g, a = %g, b - 1 = %g: #(X>=1) %d (expected %.1f, %+.2f sigma, %+.2e relative), #(X>=2) %d (expected %.1f, %+.2f sigma)"
                     % (m, a, b - 1, ge1[k], n * p1, z1, ge1[k] / (n * p1) - 1, ge2[k], n * p2, z2))
    with capsys.disabled():
        print("\n[tiny means] %.3g draws per set\n  " % n + "\n  ".join(lines))
    assert worst < 1.0, lines                                 # |observed - expected| <= 5 sigma + n 2^-24, every set, both tails
