/*
 * ORACLE -- test infrastructure only.  NOT part of the product path.
 *
 * C restatement of the THIRD-PARTY arithmetic behind the reference's count draw
 *   scipy.stats.nbinom(n=r, p=1-p).rvs()      /root/reference/prosstt/simulation.py:647-648
 * which forwards to numpy's global legacy RandomState.negative_binomial.  numpy
 * and scipy are un-vendored, unpinned dependencies of the reference (setup.py:12;
 * doc/installation.rst:9-12 lists numpy 1.14 / scipy 1.0.0 as "tested"); numpy's
 * legacy stream is frozen by NEP 19, so the published algorithm restated here
 *   MT19937 (Matsumoto & Nishimura 1998) -> 53-bit doubles,
 *   legacy polar Gaussian with a one-value cache,
 *   legacy standard gamma (Ahrens-Dieter style for shape<1, Marsaglia-Tsang above),
 *   Poisson: multiplication method below 10, PTRS (Hoermann 1993) from 10,
 *   negative_binomial(n,p) = Poisson(Gamma(n) * (1-p)/p)
 * is what every numpy >= 1.17 computes.  tests/test_numpy_legacy.py pins this
 * file against numpy itself (bit-identical streams), and bench.py may time it
 * as a numpy-free single-core CPU baseline.
 */
#include <stdint.h>
#include <math.h>

#define NPL_EXPORT __attribute__((visibility("default")))

typedef struct {
    uint32_t key[624];
    int pos;
    int has_gauss;
    double gauss;
} npl_state;

NPL_EXPORT int npl_state_size(void) { return (int)sizeof(npl_state); }

/* np.random.seed(int): Knuth's linear initialiser */
NPL_EXPORT void npl_seed(npl_state* st, uint32_t seed)
{
    for (int i = 0; i < 624; ++i) {
        st->key[i] = seed;
        seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)i + 1u;
    }
    st->pos = 624;
    st->has_gauss = 0;
    st->gauss = 0.0;
}

static void mt_refill(npl_state* st)
{
    uint32_t* k = st->key;
    int i;
    for (i = 0; i < 624 - 397; ++i) {
        uint32_t y = (k[i] & 0x80000000u) | (k[i + 1] & 0x7fffffffu);
        k[i] = k[i + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    for (; i < 623; ++i) {
        uint32_t y = (k[i] & 0x80000000u) | (k[i + 1] & 0x7fffffffu);
        k[i] = k[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    uint32_t y = (k[623] & 0x80000000u) | (k[0] & 0x7fffffffu);
    k[623] = k[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    st->pos = 0;
}

static inline uint32_t mt_next(npl_state* st)
{
    if (st->pos == 624) mt_refill(st);
    uint32_t y = st->key[st->pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

static inline double next_double(npl_state* st)
{
    int32_t a = (int32_t)(mt_next(st) >> 5), b = (int32_t)(mt_next(st) >> 6);
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

static double legacy_gauss(npl_state* st)
{
    if (st->has_gauss) {
        double t = st->gauss;
        st->has_gauss = 0;
        st->gauss = 0.0;
        return t;
    }
    double f, x1, x2, r2;
    do {
        x1 = 2.0 * next_double(st) - 1.0;
        x2 = 2.0 * next_double(st) - 1.0;
        r2 = x1 * x1 + x2 * x2;
    } while (r2 >= 1.0 || r2 == 0.0);
    f = sqrt(-2.0 * log(r2) / r2);
    st->gauss = f * x1;
    st->has_gauss = 1;
    return f * x2;
}

static inline double legacy_exponential(npl_state* st) { return -log(1.0 - next_double(st)); }

static double legacy_standard_gamma(npl_state* st, double shape)
{
    double b, c, U, V, X, Y;
    if (shape == 1.0) return legacy_exponential(st);
    if (shape == 0.0) return 0.0;
    if (shape < 1.0) {
        for (;;) {
            U = next_double(st);
            V = legacy_exponential(st);
            if (U <= 1.0 - shape) {
                X = pow(U, 1. / shape);
                if (X <= V) return X;
            } else {
                Y = -log((1 - U) / shape);
                X = pow(1.0 - shape + shape * Y, 1. / shape);
                if (X <= (V + Y)) return X;
            }
        }
    }
    b = shape - 1. / 3.;
    c = 1. / sqrt(9 * b);
    for (;;) {
        do {
            X = legacy_gauss(st);
            V = 1.0 + c * X;
        } while (V <= 0.0);
        V = V * V * V;
        U = next_double(st);
        if (U < 1.0 - 0.0331 * (X * X) * (X * X)) return (b * V);
        if (log(U) < 0.5 * X * X + b * (1. - V + log(V))) return (b * V);
    }
}

/* numpy's own log-gamma (Zhang & Jin), used only inside PTRS */
static double loggam(double x)
{
    static const double a[10] = {8.333333333333333e-02, -2.777777777777778e-03,
                                 7.936507936507937e-04, -5.952380952380952e-04,
                                 8.417508417508418e-04, -1.917526917526918e-03,
                                 6.410256410256410e-03, -2.955065359477124e-02,
                                 1.796443723688307e-01, -1.39243221690590e+00};
    double x0 = x, x2, gl, gl0;
    long n = 0;
    if (x == 1.0 || x == 2.0) return 0.0;
    if (x <= 7.0) {
        n = (long)(7 - x);
        x0 = x + n;
    }
    x2 = 1.0 / (x0 * x0);
    gl0 = a[9];
    for (int k = 8; k >= 0; --k) {
        gl0 *= x2;
        gl0 += a[k];
    }
    gl = gl0 / x0 + 0.5 * log(2.0 * M_PI) + (x0 - 0.5) * log(x0) - x0;
    if (x <= 7.0) {
        for (long k = 1; k <= n; ++k) {
            gl -= log(x0 - 1.0);
            x0 -= 1.0;
        }
    }
    return gl;
}

static int64_t poisson_mult(npl_state* st, double lam)
{
    double enlam = exp(-lam), prod = 1.0;
    int64_t X = 0;
    for (;;) {
        prod *= next_double(st);
        if (prod > enlam) X += 1;
        else return X;
    }
}

static int64_t poisson_ptrs(npl_state* st, double lam)
{
    double slam = sqrt(lam), loglam = log(lam);
    double b = 0.931 + 2.53 * slam;
    double a = -0.059 + 0.02483 * b;
    double invalpha = 1.1239 + 1.1328 / (b - 3.4);
    double vr = 0.9277 - 3.6224 / (b - 2);
    for (;;) {
        double U = next_double(st) - 0.5;
        double V = next_double(st);
        double us = 0.5 - fabs(U);
        int64_t k = (int64_t)floor((2 * a / us + b) * U + lam + 0.43);
        if (us >= 0.07 && V <= vr) return k;
        if (k < 0 || (us < 0.013 && V > us)) continue;
        if ((log(V) + log(invalpha) - log(a / (us * us) + b)) <= (-lam + k * loglam - loggam(k + 1)))
            return k;
    }
}

static inline int64_t legacy_poisson(npl_state* st, double lam)
{
    if (lam >= 10) return poisson_ptrs(st, lam);
    if (lam == 0) return 0;
    return poisson_mult(st, lam);
}

/* ---- exported probes ------------------------------------------------------ */

NPL_EXPORT void npl_random_sample(npl_state* st, double* out, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) out[i] = next_double(st);
}

NPL_EXPORT void npl_standard_normal(npl_state* st, double* out, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) out[i] = legacy_gauss(st);
}

NPL_EXPORT void npl_standard_gamma(npl_state* st, double shape, double* out, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) out[i] = legacy_standard_gamma(st, shape);
}

NPL_EXPORT void npl_poisson(npl_state* st, const double* lam, int64_t* out, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) out[i] = legacy_poisson(st, lam[i]);
}

/* RandomState.negative_binomial(n[i], p[i]) element by element, C order */
NPL_EXPORT void npl_negative_binomial(npl_state* st, const double* n, const double* p,
                                      int64_t* out, int64_t count)
{
    for (int64_t i = 0; i < count; ++i) {
        double Y = legacy_standard_gamma(st, n[i]) * ((1 - p[i]) / p[i]);
        out[i] = legacy_poisson(st, Y);
    }
}

/*
 * The whole of draw_counts (simulation.py:633-651) for one chunk of cells, in C:
 * mu = means[row]*scaling; (p, r) = get_pr_umi (count_model.py:156-158); NB draw.
 * Returns the number of domain errors scipy's argument check would have raised on.
 */
NPL_EXPORT int64_t npl_draw_counts(npl_state* st, const double* means, int64_t G,
                                   const int64_t* row_of_cell, const double* scaling,
                                   const double* alpha, const double* beta, int64_t N, int64_t* out)
{
    int64_t bad = 0;
    for (int64_t c = 0; c < N; ++c) {
        const double* mrow = means + row_of_cell[c] * G;
        for (int64_t g = 0; g < G; ++g) {
            double m = mrow[g] * scaling[c];
            double s2 = alpha[g] * (m * m) + beta[g] * m;
            double p = (s2 - m) / s2, r = (m * m) / (s2 - m);
            if (s2 <= 0) { p = 0; r = 0; }
            double pp = 1 - p;
            if (!(r > 0) || !(pp > 0) || !(pp <= 1)) { bad++; out[c * G + g] = 0; continue; }
            double Y = legacy_standard_gamma(st, r) * ((1 - pp) / pp);
            out[c * G + g] = legacy_poisson(st, Y);
        }
    }
    return bad;
}
