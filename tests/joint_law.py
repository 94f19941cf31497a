"""
Statistics of INDEPENDENCE of a count matrix (test helper).

The reference draws its N x G counts independently given the parameters (one ``rvs()`` over the flattened arrays,
/root/reference/prosstt/simulation.py:647-648).  The device sampler keys a counter-based generator by (cell, gene
quad): marginal law tests cannot see a dependence between neighbouring counters, words of one Philox call, gene
tiles or cells.  Here every count is standardised, z = (X - mu) / sqrt(var) with the reference's mean and variance
(mu = M * s, var = alpha * mu^2 + beta * mu: count_model.py:156-158), and products of NEIGHBOURS are summed: under
independence E[z_i z_j] = 0 and Var[z_i z_j] = 1 exactly, so  T = sum(z_i z_j) / sqrt(#pairs)  is a standard normal
score whatever the shapes of the marginals.  Row and column sums get a variance-ratio score: the sum of (X - mu)
over a row has variance sum(var) under independence (a positive dependence inflates it by the number of partners).

Works on torch tensors (device or CPU); accumulates over chunks of cells.
"""
import math

import torch

MU_MIN = 0.05       # samples under this mean are all but constant: left out of the products


class JointLaw:
    """Accumulator over chunks of consecutive cells of one matrix."""

    GENE_CLASSES = ("genes_lag1_inside_quad", "genes_lag1_across_quads", "genes_lag1_across_tiles",
                    "genes_lag2_inside_quad")

    def __init__(self, G, device):
        self.G, self.device = G, device
        g = torch.arange(G - 1, device=device)
        self.cls = {
            "genes_lag1_inside_quad": (g % 4 != 3),
            "genes_lag1_across_quads": (g % 4 == 3) & (g % 256 != 255),
            "genes_lag1_across_tiles": (g % 256 == 255),
        }
        g2 = torch.arange(G - 2, device=device)
        self.cls2 = (g2 % 4 < 2)
        self.sums = {k: [0.0, 0.0] for k in self.GENE_CLASSES + ("cells_lag1", "paired_matrix")}
        self.col_dev = torch.zeros(G, dtype=torch.float64, device=device)
        self.col_var = torch.zeros(G, dtype=torch.float64, device=device)
        self.row_ratio = []

    @staticmethod
    def standardise(X, mu, var):
        w = mu >= MU_MIN
        z = torch.where(w, (X.to(mu.dtype) - mu) * torch.rsqrt(var), torch.zeros_like(mu))
        return z, w

    def _acc(self, key, prod, pairs):
        self.sums[key][0] += float(prod.sum(dtype=torch.float64))
        self.sums[key][1] += float(pairs.sum(dtype=torch.float64))

    def add(self, X, mu, var, X_pair=None):
        """One chunk of consecutive cells: counts, means and variances (n, G); ``X_pair``: a second matrix drawn with the
        SAME parameters under other counters (e.g. cell ids 2^32 higher), correlated sample by sample."""
        z, w = self.standardise(X, mu, var)
        p1, w1 = z[:, :-1] * z[:, 1:], w[:, :-1] & w[:, 1:]
        for key, m in self.cls.items():
            self._acc(key, p1[:, m], w1[:, m])
        p2, w2 = z[:, :-2] * z[:, 2:], w[:, :-2] & w[:, 2:]
        self._acc("genes_lag2_inside_quad", p2[:, self.cls2], w2[:, self.cls2])
        self._acc("cells_lag1", z[:-1] * z[1:], w[:-1] & w[1:])
        if X_pair is not None:
            z2, _ = self.standardise(X_pair, mu, var)
            self._acc("paired_matrix", z * z2, w)
        dev = X.to(torch.float64) - mu.to(torch.float64)
        v64 = var.to(torch.float64)
        self.col_dev += dev.sum(0)
        self.col_var += v64.sum(0)
        self.row_ratio.append(dev.sum(1) ** 2 / v64.sum(1))

    def scores(self):
        """name -> (score, detail).  Every score is ~N(0, 1) under independence."""
        out = {}
        for key, (s, n) in self.sums.items():
            if n > 0:
                out[key] = (s / math.sqrt(n), "%d pairs, mean product %.3e" % (n, s / n))
        # variance ratios: mean over rows (columns) of (sum of deviations)^2 / (sum of variances), expectation 1; scored
        # against its own spread over the rows (columns)
        for key, ratio in (("row_sums_variance", torch.cat(self.row_ratio)),
                           ("column_sums_variance", self.col_dev ** 2 / self.col_var)):
            n = ratio.numel()
            mean, sd = float(ratio.mean()), float(ratio.std())
            out[key] = ((mean - 1.0) / (sd / math.sqrt(n)), "%d sums, variance ratio %.5f" % (n, mean))
        return out


def moments(means, rows, sc, alpha, beta, lo, hi):
    """mu and var (float32, (hi - lo, G)) of cells lo..hi as the sampler forms them: m = M[row] * s in binary32."""
    mu = means[rows[lo:hi].long()] * sc[lo:hi].to(torch.float32)[:, None]
    var = alpha.to(torch.float32)[None, :] * mu * mu + beta.to(torch.float32)[None, :] * mu
    return mu, var


def report(scores, title):
    lines = ["[joint law] %s" % title]
    for key, (t, detail) in scores.items():
        lines.append("   %-28s %+6.2f sigma   (%s)" % (key, t, detail))
    return "\n".join(lines)
