/*
 * ORACLE -- test infrastructure only.  NOT part of the product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; prosstt_amd never does.
 *
 * Scalar C model of the *device* count sampler ("PRNB-6"): the
 * fused  gather * scale -> get_pr_umi -> negative-binomial draw  that replaces
 *   simulation.draw_counts           /root/reference/prosstt/simulation.py:602-651
 *   count_model.get_pr_umi           /root/reference/prosstt/count_model.py:131-161
 *   scipy.stats.nbinom(n=r,p=1-p).rvs() = RandomState.negative_binomial  (simulation.py:647-648)
 *
 * The reference draws from numpy's sequential MT19937 stream, which no
 * parallel device can reproduce (SURVEY.md section 0 "RNG reality check"); the count
 * law, not the stream, is the contract.  PRNB-6 is a counter-based sampler of
 * the SAME law  NB(n = r, p = 1-p)  with  theta = a*m + b - 1,  r = m/theta,
 * p = theta/(1+theta)  (the algebraic form of get_pr_umi), defined so that
 * every sample is a pure function of (M, s, a, b, seed, cell, gene):
 *   - randomness: Philox4x32-7 (Salmon et al. 2011: the fewest rounds that pass BigCrush), key = seed,
 *     counter = (cell_lo, cell_hi, gene-or-quad, domain);
 *   - arithmetic: IEEE binary32 add/mul/fma/sqrt, the polynomial log/exp/cos and the Newton
 *     reciprocal below -- no libm -- and, for P(X = 0) of the inversion class only, three functions of
 *     the gfx950 hardware: HW_RCP (v_rcp_f32), HW_LOG2 (v_log_f32), HW_EXP2 (v_exp_f32).  This model
 *     does not re-implement those: it reads their values from TABLES that the product's three-line probe
 *     kernel (prosstt_amd_hw_math; hw_math_kernel in prosstt_amd.hip: y = v_rcp_f32(x) etc. over a range
 *     of bit patterns) writes on the device under test -- prnb_set_hw_tables, oracle/nb_model.py:
 *     install_hw_tables.  With the tables installed this C model and the HIP kernels agree BIT FOR BIT.
 *     Without them (no GPU: the CPU-only tests) the three functions fall back to libm's
 *     1/x, log2f, exp2f -- each within an ulp or two of the hardware's values -- which is NOT the
 *     device's definition bit for bit but the same law to 1e-6: that mode serves the law tests that
 *     run without a GPU (prnb_hw_mode() tells which is active).
 * The law itself is pinned against scipy/numpy (tests/test_nb_model.py, fixture g7; on the GPU the same
 * tests run with the tables) and the kernel is pinned against this model (tests/test_gpu_*.py).
 *
 * Build: see oracle/Makefile  (-O2 -ffp-contract=off; never -ffast-math).
 */
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#define PRNB_EXPORT __attribute__((visibility("default")))
#if defined(__x86_64__)
#define PRNB_CLONES __attribute__((target_clones("avx2,fma", "default")))
#else
#define PRNB_CLONES
#endif

/* ---- sampler constants (part of the PRNB-6 definition) ------------------ */
#define PRNB_LIGHT_T2     27.4112f     /* inversion iff theta <= 24 and t2 = -log2 P(X=0) < 27.4112 (19/ln 2): P0 * 2^32 >= 24 */
#define PRNB_LIGHT_THETA  24.0f        /* tail ratio theta/(1+theta) <= 24/25 (PRNB-6; 16 in PRNB-5) */
#define PRNB_THETA_MIN    1.1920929e-7f  /* 2^-23: 1 + theta > 1 in binary32; below this NB == Poisson to 1e-7 of the variance */
#define PRNB_THETA_MAX    1.0e18f
#define PRNB_R_MIN        9.094947e-13f  /* 2^-40: P(X>0) < 2^-32, return 0 */
#define PRNB_WALK_END     1024         /* the group k = 1021..1024 is a walk's last: P(X > 1024) < 1e-12 in the inversion class */
#define PRNB_KTAB         1032         /* 1/k for 1 <= k < KTAB */
#define PRNB_POIS_INV     10.0f        /* Poisson: inversion below, PTRS above */
#define PRNB_LAM_BIG      4194304.0f   /* 2^22: rounded normal above */
#define PRNB_MAX_TRIES    64

static inline uint32_t f2u(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
static inline float u2f(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }
#define FMA(a, b, c) __builtin_fmaf((a), (b), (c))

/* ---- the three hardware functions of the definition: values from the device's own tables --------
 *   rcp_mant[i]  = v_rcp_f32(x) for the 2^23 floats x in [1, 2) (bit pattern 0x3F800000 + i); for any other
 *                  normal x = 2^e * x1 the value is rcp_mant(x1) * 2^-e  (exact scaling: checked on the
 *                  device over the whole range the sampler presents, tests/test_gpu_hw_math.py)
 *   log2_tab[i]  = v_log_f32(x) for bit patterns log2_first + i   (the sampler presents 1 < x <= 17)
 *   exp2_tab[i]  = v_exp_f32(-x) for bit patterns exp2_first + i  (0 <= x < 27.4112 presented; below the table's
 *                  first entry, 2^-24, the value is 1.0f: checked on the device for every such x)
 * A lookup outside a table aborts: never a silent substitute. */
static const float* g_rcp_mant = 0;
static const float* g_log2_tab = 0;
static const float* g_exp2_tab = 0;
static uint32_t g_log2_first = 0, g_log2_count = 0, g_exp2_first = 0, g_exp2_count = 0;

static void hw_fail(const char* what, float x)
{
    fprintf(stderr, "oracle/nb_model.c: %s(%a) is outside the installed hardware table\n", what, (double)x);
    abort();
}

static inline float hw_rcp(float x)
{
    if (!g_rcp_mant) return 1.0f / x;
    const uint32_t b = f2u(x);
    const int32_t e = (int32_t)(b >> 23) - 127;
    if ((b >> 31) || e < -100 || e > 100) hw_fail("HW_RCP", x);
    return u2f(f2u(g_rcp_mant[b & 0x7FFFFFu]) - ((uint32_t)e << 23));
}
static inline float hw_log2(float x)
{
    if (!g_log2_tab) return log2f(x);
    const uint32_t i = f2u(x) - g_log2_first;
    if (i >= g_log2_count) hw_fail("HW_LOG2", x);
    return g_log2_tab[i];
}
/* v_exp_f32(-x), x >= 0 */
static inline float hw_exp2neg(float x)
{
    if (!g_exp2_tab) return exp2f(-x);
    if (x < 5.9604645e-8f) return 1.0f;                    /* 2^-24 */
    const uint32_t i = f2u(x) - g_exp2_first;
    if (i >= g_exp2_count) hw_fail("HW_EXP2", -x);
    return g_exp2_tab[i];
}

/* ---- Philox4x32-R (Salmon, Moraes, Dror, Shaw 2011) ----------------------- */
static inline void philox4x32_r(int rounds, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                uint32_t k0, uint32_t k1, uint32_t out[4])
{
    for (int round = 0; round < rounds; ++round) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
/* the library default (known-answer vectors; the device-mode lineage walk PRLW-1) */
static inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                 uint32_t k0, uint32_t k1, uint32_t out[4])
{
    philox4x32_r(10, c0, c1, c2, c3, k0, k1, out);
}
/* the count sampler: Philox4x32-7, the fewest rounds that pass BigCrush (Salmon et al. 2011, table 2) */
#define PRNB_COUNT_ROUNDS 7
/* Rounds the sampler's draws use.  PRNB_COUNT_ROUNDS is the definition; prnb_set_count_rounds exists for ONE purpose:
 * the joint-law tests draw the same matrix with 10 rounds (the library default of Random123) and hold its statistics of
 * independence beside those of the 7-round definition (tests/test_joint_law.py). */
static int g_count_rounds = PRNB_COUNT_ROUNDS;
static inline void philox_count(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                uint32_t k0, uint32_t k1, uint32_t out[4])
{
    philox4x32_r(g_count_rounds, c0, c1, c2, c3, k0, k1, out);
}

/* ---- deterministic binary32 math ----------------------------------------- */

/* 1/x for normal x>0: integer seed + 3 Newton steps (rel. err < 2^-23). */
static inline float det_rcp(float x)
{
    float y = u2f(0x7EF311C7u - f2u(x));
    float e = FMA(-x, y, 1.0f); y = FMA(y, e, y);
    e = FMA(-x, y, 1.0f); y = FMA(y, e, y);
    e = FMA(-x, y, 1.0f); y = FMA(y, e, y);
    return y;
}

/* log(1+f) - f + f*f/2  =  f^3 * P(f)  on  f in [sqrt(1/2)-1, sqrt(2)-1]  (Cephes logf) */
static inline float log_tail(float f)
{
    float p = 7.0376836292e-2f;
    p = FMA(p, f, -1.1514610310e-1f);
    p = FMA(p, f, 1.1676998740e-1f);
    p = FMA(p, f, -1.2420140846e-1f);
    p = FMA(p, f, 1.4249322787e-1f);
    p = FMA(p, f, -1.6668057665e-1f);
    p = FMA(p, f, 2.0000714765e-1f);
    p = FMA(p, f, -2.4999993993e-1f);
    p = FMA(p, f, 3.3333331174e-1f);
    return p * f * (f * f);
}

/* log(x + c) for normal x>0 and a small correction c (|c| <= ulp(x)). */
static inline float det_log_c(float x, float c)
{
    uint32_t ix = f2u(x);
    int32_t e = (int32_t)(ix - 0x3F3504F3u) >> 23;           /* x = 2^e * mant, mant in [0.7071, 1.4142) */
    float mant = u2f(ix - ((uint32_t)e << 23));
    float scale = u2f((uint32_t)(127 - e) << 23);            /* 2^-e */
    float f = (mant - 1.0f) + c * scale;
    float fe = (float)e;
    float y = log_tail(f);
    y = FMA(fe, -2.12194440e-4f, y);
    y = FMA(-0.5f, f * f, y);
    return FMA(fe, 0.693359375f, f + y);
}
static inline float det_log(float x) { return det_log_c(x, 0.0f); }

/* log(1+t), t >= 0 : exact two-sum of 1+t, residual folded into the mantissa. */
static inline float det_log1p(float t)
{
    float u = 1.0f + t;
    float c = (t >= 1.0f) ? (1.0f - (u - t)) : (t - (u - 1.0f));
    return det_log_c(u, c);
}

/* log(1+d) - d for d > -1, without cancellation near 0. */
static inline float det_log1pmx(float d, float rho /* = 1+d computed by the caller */)
{
    if (d >= -0.29289323f && d < 0.41421354f)
        return FMA(-0.5f, d * d, log_tail(d));
    return det_log(rho) - d;
}

/* exp(x) for x <= 0 (Cephes expf); 0 below -87. */
static inline float det_exp(float x)
{
    if (!(x > -87.0f)) return 0.0f;
    float z = floorf(FMA(x, 1.44269504088896341f, 0.5f));
    float r = FMA(z, -0.693359375f, x);
    r = FMA(z, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = FMA(p, r, 1.3981999507e-3f);
    p = FMA(p, r, 8.3334519073e-3f);
    p = FMA(p, r, 4.1665795894e-2f);
    p = FMA(p, r, 1.6666665459e-1f);
    p = FMA(p, r, 5.0000001201e-1f);
    float y = FMA(p, r * r, r) + 1.0f;
    return y * u2f((uint32_t)((int32_t)z + 127) << 23);
}

/* cos(2*pi*w/2^32): octant from the top 3 bits, Cephes sinf/cosf kernels. */
static inline float det_cos2pi(uint32_t w)
{
    uint32_t j = w >> 29;
    float f = (float)(w & 0x1FFFFFFFu) * 1.862645149230957e-9f;   /* 2^-29 */
    if (j & 1u) f = f - 1.0f;
    uint32_t q = ((j + 1u) >> 1) & 3u;
    float y = f * 0.78539816339744830962f;
    float z = y * y;
    float s = FMA(-1.9515295891e-4f, z, 8.3321608736e-3f);
    s = FMA(s, z, -1.6666654611e-1f);
    s = FMA(s * z, y, y);
    float c = FMA(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    c = FMA(c, z, 4.166664568298827e-2f);
    c = FMA(c * z, z, FMA(-0.5f, z, 1.0f));
    float v = (q & 1u) ? s : c;
    return (q == 1u || q == 2u) ? -v : v;
}

/* uniform in (0,1] from 32 bits, full resolution near 0 */
static inline float unif(uint32_t w) { return ((float)w + 0.5f) * 2.3283064365386963e-10f; }

static float g_inv_k[PRNB_KTAB];       /* 1/k, correctly rounded */
static const float LOGFACT[10] = {0.0f, 0.0f, 0.69314718f, 1.7917595f, 3.1780538f, 4.7874917f,
                                  6.5792512f, 8.5251614f, 10.604603f, 12.801827f};

__attribute__((constructor)) static void prnb_init(void)
{
    g_inv_k[0] = 0.0f;
    for (int k = 1; k < PRNB_KTAB; ++k) g_inv_k[k] = 1.0f / (float)k;
}

/*
 * Inversion by chop-down on a binary32 remainder.  pmf recurrence
 *   P(k+1) = P(k) * (q + (mp - q)/(k+1))      (= P(k) * (mp + k*q)/(k+1);  k = 0: P(0) * mp)
 * (NB: mp = m/(1+theta), q = theta/(1+theta);  Poisson: mp = lambda, q = 0), carried scaled by 2^32:
 * `ps` enters as P(0) * 2^32.  `w` is the 32-bit uniform; the remainder starts as (float)w (round to
 * nearest: 24 significant bits, full resolution below 2^24) and every term is subtracted from it in
 * binary32; the draw is the first k whose subtraction leaves the remainder negative.  Terms come in the
 * groups the device walks in (k = 0..4 -- stage 2 of the streaming kernel --, then four at a time: PRNB-7; PRNB-6 had
 * k = 0..2 first, and a third more walks in the kernel's third stage); when a group ends without a negative
 * remainder and its LAST term is below 1 -- the pmf has fallen under 2^-32: mass lost to rounding, < 1e-6 --
 * the draw is that group's last k, as it is when the group is the last one (k = PRNB_WALK_END).
 * The ratio of a term comes by ONE fma from the table's 1/(k+1), the term by ONE multiplication.  The
 * cancellation in q + (mp - q)/(k+1) when mp << q costs at most ulp(q) of the ratio, 6e-8 * q/mp of the
 * terms k >= 2 relative -- times P(X >= 2) that is under 2e-7 absolute.
 */
static inline int32_t chop_down(uint32_t w, float ps, float mp, float q)
{
    const float d = mp - q;
    float rem = (float)w;
    for (int k = 0; ; ) {
        rem = rem - ps;
        if (rem < 0.0f) return k;
        if (k >= 4 && (k & 3) == 0 && (ps < 1.0f || k >= PRNB_WALK_END)) return k;     /* k = 4, 8, 12, ...: a group's last term */
        ps = (k == 0) ? ps * mp : ps * FMA(d, g_inv_k[k + 1], q);
        ++k;
    }
}

/* ---- the gamma-Poisson class: hardware transcendentals by QUERY ------------------------------------------------
 * PRNB-7 draws this class with the gfx950 instructions v_log_f32, v_exp_f32, v_sqrt_f32, v_rsq_f32, v_cos_f32 (and
 * v_rcp_f32: the mantissa table above) over arguments no table can enumerate.  The model therefore ASKS the device:
 * a sample's evaluation runs until it needs values it does not have, registers the (op, x) pairs, and stops; the
 * resolver below collects the questions of all waiting samples, has the product's probe answer them in one call per
 * round (prosstt_amd_hw_math_at: y[i] = op(x[i]) on the device under test), appends the answers to each sample's tape
 * and runs the sample again from the start -- it replays its tape and gets further.  An evaluation is a pure function
 * of (sample, tape), so the replay asks the same questions in the same order (checked: the tape keeps x beside y).
 * Without a query function (no GPU) the ops are libm stand-ins: the same law, not the same bits. */
enum { HWOP_RCP = 0, HWOP_LOG2 = 1, HWOP_EXP2NEG = 2, HWOP_SQRT = 3, HWOP_RSQ = 4, HWOP_COS = 5, HWOP_COUNT = 6 };
#define HWQ_MAX_ASK 4            /* questions a sample may leave open at once */
typedef struct { float x, y; } hwq_pair;
typedef struct {
    const hwq_pair* tape; int len, pos;          /* answered so far / replay position */
    int npend; int32_t pend_op[HWQ_MAX_ASK]; float pend_x[HWQ_MAX_ASK];
} hwq;

typedef void (*prnb_hw_query_fn)(int64_t n, const int32_t* op, const float* x, float* y);
static prnb_hw_query_fn g_query = 0;

static float hw_standin(int op, float x)
{
    switch (op) {
    case HWOP_RCP: return 1.0f / x;
    case HWOP_LOG2: return log2f(x);
    case HWOP_EXP2NEG: return exp2f(-x);
    case HWOP_SQRT: return sqrtf(x);
    case HWOP_RSQ: return (float)(1.0 / sqrt((double)x));
    case HWOP_COS: return (float)cos(6.283185307179586476925 * (double)x);      /* v_cos_f32 takes revolutions */
    default: return 0.0f;
    }
}

/* ask for op(x): *y is valid after the next HWQ_SYNC that does not leave the function */
static inline void hwq_ask(hwq* Q, int op, float x, float* y)
{
    if (!Q) { *y = hw_standin(op, x); return; }
    if (Q->pos < Q->len) {
        if (f2u(Q->tape[Q->pos].x) != f2u(x)) {
            fprintf(stderr, "oracle/nb_model.c: replay diverged (op %d: %a asked, %a on the tape)\n", op, (double)x, (double)Q->tape[Q->pos].x);
            abort();
        }
        *y = Q->tape[Q->pos++].y;
        return;
    }
    if (Q->npend >= HWQ_MAX_ASK) { fprintf(stderr, "oracle/nb_model.c: too many open questions\n"); abort(); }
    Q->pend_op[Q->npend] = op;
    Q->pend_x[Q->npend] = x;
    Q->npend += 1;
    *y = 0.0f;
}
#define HWQ_SYNC(Q) do { if ((Q) && (Q)->npend) return 0; } while (0)

#define LN2_F     0.69314718f
#define LOG2E_F   1.44269504f
#define M2LN2_F   (-1.3862944f)
#define TWO_M32_F 2.3283064365386963e-10f

/* log(1+d) - d: the polynomial where the difference cancels, the hardware's log2 of rho = 1 + d elsewhere */
static inline int hq_log1pmx(hwq* Q, float d, float rho, float* out)
{
    if (d >= -0.29289323f && d < 0.41421354f) { *out = FMA(-0.5f, d * d, log_tail(d)); return 1; }
    float l;
    hwq_ask(Q, HWOP_LOG2, rho, &l);
    HWQ_SYNC(Q);
    *out = FMA(LN2_F, l, -d);
    return 1;
}

/* Poisson(lam) on counter domain 0x80000000+j of (cell, gene).  Returns 1 when *out is final, 0 when it waits. */
static inline int poisson_draw(hwq* Q, float lam, uint32_t c0, uint32_t c1, uint32_t gene,
                               uint32_t k0, uint32_t k1, int32_t* out)
{
    uint32_t w[4];
    *out = 0;
    if (!(lam > 0.0f)) return 1;
    if (lam < PRNB_POIS_INV) {
        float p0;
        hwq_ask(Q, HWOP_EXP2NEG, lam * LOG2E_F, &p0);
        HWQ_SYNC(Q);
        philox_count(c0, c1, gene, 0x80000000u, k0, k1, w);
        *out = chop_down(w[0], fminf(p0, 0.99999994f) * 4294967296.0f, lam, 0.0f);
        return 1;
    }
    float slam;
    hwq_ask(Q, HWOP_SQRT, lam, &slam);
    if (!(lam < PRNB_LAM_BIG)) {               /* rounded normal; never reached with abs_max=5000 */
        float l, c, s;
        philox_count(c0, c1, gene, 0x80000000u, k0, k1, w);
        hwq_ask(Q, HWOP_LOG2, unif(w[0]), &l);
        hwq_ask(Q, HWOP_COS, (float)w[1] * TWO_M32_F, &c);
        HWQ_SYNC(Q);
        hwq_ask(Q, HWOP_SQRT, M2LN2_F * l, &s);
        HWQ_SYNC(Q);
        float kf = floorf(FMA(slam, s * c, lam) + 0.5f);
        *out = (int32_t)fminf(fmaxf(kf, 0.0f), 2147483520.0f);
        return 1;
    }
    HWQ_SYNC(Q);
    /* PTRS, Hoermann 1993 (the algorithm behind numpy's random_poisson for lam >= 10) */
    float bb = FMA(2.53f, slam, 0.931f);
    float aa = FMA(0.02483f, bb, -0.059f);
    float invalpha = FMA(1.1328f, hw_rcp(bb - 3.4f), 1.1239f);
    float vr = FMA(-3.6224f, hw_rcp(bb - 2.0f), 0.9277f);
    float kf = floorf(lam);
    for (int j = 0; j < 2 * PRNB_MAX_TRIES; ++j) {
        if ((j & 1) == 0) philox_count(c0, c1, gene, 0x80000000u + (uint32_t)(j >> 1), k0, k1, w);
        float U = unif(w[(j & 1) * 2]) - 0.5f;
        float V = unif(w[(j & 1) * 2 + 1]);
        float us = fmaxf(0.5f - fabsf(U), 5.8207661e-11f);       /* 2^-34 */
        float rus = hw_rcp(us);
        kf = floorf(FMA(FMA(2.0f * aa, rus, bb), U, lam + 0.43f));
        if (us >= 0.07f && V <= vr) break;
        if (kf < 0.0f || (us < 0.013f && V > us)) { kf = floorf(lam); continue; }
        float l1, l2, rhs;
        hwq_ask(Q, HWOP_LOG2, (V * invalpha) * hw_rcp(FMA(aa * rus, rus, bb)), &l1);
        if (kf < 10.0f) {
            hwq_ask(Q, HWOP_LOG2, lam, &l2);
            HWQ_SYNC(Q);
            rhs = FMA(kf, LN2_F * l2, -lam) - LOGFACT[(int)kf];
        } else {
            float rk = hw_rcp(kf);
            float d = (lam - kf) * rk;
            float lp;
            hwq_ask(Q, HWOP_LOG2, 6.2831855f * kf, &l2);
            if (!hq_log1pmx(Q, d, lam * rk, &lp)) return 0;
            HWQ_SYNC(Q);
            float st = rk * FMA(-0.0027777778f, rk * rk, 0.083333336f);
            rhs = FMA(kf, lp, FMA(-0.5f, LN2_F * l2, -st));
        }
        if (LN2_F * l1 <= rhs) break;
        kf = floorf(lam);
    }
    *out = (int32_t)fminf(fmaxf(kf, 0.0f), 2147483520.0f);
    return 1;
}

/* Gamma(r) * theta on counter domain 1+i of (cell, gene): Marsaglia-Tsang 2000,
 * Box-Muller normal, U^(1/r) boost below r = 1.  Returns 1 when *lam is final, 0 when it waits. */
static inline int gamma_scaled(hwq* Q, float r, float theta, uint32_t c0, uint32_t c1, uint32_t gene,
                               uint32_t k0, uint32_t k1, float* lam)
{
    uint32_t w[4];
    int boost = r < 1.0f;
    float rr = boost ? r + 1.0f : r;
    float dd = rr - 0.33333334f;
    float cc;
    hwq_ask(Q, HWOP_RSQ, 9.0f * dd, &cc);
    float v = 1.0f;
    for (int i = 0; i < PRNB_MAX_TRIES; ++i) {
        /* every attempt is a pure function of (i, parameters); attempt MAX_TRIES-1 is final */
        const int last = (i == PRNB_MAX_TRIES - 1);
        float l, c, s;
        philox_count(c0, c1, gene, 1u + (uint32_t)i, k0, k1, w);
        hwq_ask(Q, HWOP_LOG2, unif(w[0]), &l);
        hwq_ask(Q, HWOP_COS, (float)w[1] * TWO_M32_F, &c);
        HWQ_SYNC(Q);
        hwq_ask(Q, HWOP_SQRT, M2LN2_F * l, &s);
        HWQ_SYNC(Q);
        float x = s * c;
        float t = cc * x;
        float v1 = 1.0f + t;
        if (!(v1 > 0.0f)) {
            v = 1.0f;
            if (last) break;
            continue;
        }
        v = v1 * v1 * v1;
        if (last) break;
        float u = unif(w[2]);
        float x2 = x * x;
        if (u < FMA(-0.0331f, x2 * x2, 1.0f)) break;
        /* log u < x^2/2 + d*(1 - v + log v),  1 - v + log v = 3*log1pmx(t) - 3t^2 - t^3 */
        float t2 = t * t, lu, lp;
        hwq_ask(Q, HWOP_LOG2, u, &lu);
        if (!hq_log1pmx(Q, t, v1, &lp)) return 0;
        HWQ_SYNC(Q);
        float h = FMA(3.0f, lp, FMA(-t2, t, -3.0f * t2));
        if (LN2_F * lu < FMA(dd, h, 0.5f * x2)) break;
    }
    float g = dd * v;
    if (boost) {
        float lb, e;
        hwq_ask(Q, HWOP_LOG2, unif(w[3]), &lb);
        HWQ_SYNC(Q);
        hwq_ask(Q, HWOP_EXP2NEG, -(lb * hw_rcp(r)), &e);       /* u^(1/r) */
        HWQ_SYNC(Q);
        g = g * e;
    }
    *lam = theta * g;
    return 1;
}

typedef struct { float m, theta, p, r; int32_t path; } prnb_detail;

/* One count.  path: 0 = degenerate (returns 0), 1 = NB inversion, 2 = gamma-Poisson.  Returns 1 when *count is final,
 * 0 when the sample (of the gamma-Poisson class, in query mode) waits for hardware values.  draw = 0: parameters and path only. */
static inline int prnb_one_q(float M, float s, float a, float bm1, uint32_t k0, uint32_t k1,
                             uint64_t cell, uint32_t gene, prnb_detail* det, hwq* Q, int draw, int32_t* count)
{
    float m = M * s;
    float theta = FMA(a, m, bm1);
    uint32_t c0 = (uint32_t)cell, c1 = (uint32_t)(cell >> 32);
    *count = 0;
    if (det) { det->m = m; det->theta = theta; det->p = 0.0f; det->r = 0.0f; det->path = 0; }
    if (!(m > 0.0f) || !(theta > 0.0f)) return 1;
    theta = fminf(fmaxf(theta, PRNB_THETA_MIN), PRNB_THETA_MAX);
    const float u1 = 1.0f + theta;                       /* > 1 */
    const float iu = hw_rcp(u1);
    const float q = theta * iu;
    const float r = m * hw_rcp(theta);
    if (det) { det->p = q; det->r = r; }
    if (theta <= PRNB_LIGHT_THETA) {
        /* t2 = -log2 P(X = 0) = m * log2(1 + theta)/theta, the quotient taken at the theta' = u1 - 1 that u1 stands for */
        const float t2 = m * (hw_log2(u1) * hw_rcp(u1 - 1.0f));
        if (t2 < PRNB_LIGHT_T2) {
            uint32_t w[4];
            if (det) det->path = 1;
            if (!draw) return 1;
            philox_count(c0, c1, gene >> 2, 0u, k0, k1, w);
            *count = chop_down(w[gene & 3u], hw_exp2neg(t2) * 4294967296.0f, m * iu, q);
            return 1;
        }
    }
    if (det) det->path = 2;
    if (!draw || !(r >= PRNB_R_MIN)) return 1;
    float lam;
    if (!gamma_scaled(Q, r, theta, c0, c1, gene, k0, k1, &lam)) return 0;
    return poisson_draw(Q, lam, c0, c1, gene, k0, k1, count);
}

/* ---- the resolver of waiting samples (query mode) ------------------------------------------------------------- */
typedef struct { int64_t idx; float M, s, a, bm1; uint64_t cell; uint32_t gene; } prnb_waiting;
typedef struct { prnb_waiting* v; int64_t n, cap; } prnb_wait_list;

static void wait_push(prnb_wait_list* L, prnb_waiting w)
{
    if (L->n == L->cap) {
        L->cap = L->cap ? 2 * L->cap : 1024;
        L->v = (prnb_waiting*)realloc(L->v, (size_t)L->cap * sizeof(prnb_waiting));
        if (!L->v) { fprintf(stderr, "oracle/nb_model.c: out of memory\n"); abort(); }
    }
    L->v[L->n++] = w;
}

static int64_t g_query_rounds = 0, g_query_values = 0;     /* statistics of the last resolve (tests) */

/* out[W[i].idx] = the count of waiting sample i, for all n of them */
static void resolve_waiting(const prnb_waiting* W, int64_t n, uint32_t k0, uint32_t k1, int32_t* out)
{
    g_query_rounds = 0;
    g_query_values = 0;
    if (n == 0) return;
    if (!g_query) { fprintf(stderr, "oracle/nb_model.c: samples wait for hardware values but no query function is installed\n"); abort(); }
    typedef struct { hwq_pair* t; int len, cap; } tape_t;
    tape_t* tapes = (tape_t*)calloc((size_t)n, sizeof(tape_t));
    int64_t* active = (int64_t*)malloc((size_t)n * sizeof(int64_t));
    int32_t* pn = (int32_t*)malloc((size_t)n * sizeof(int32_t));
    int32_t* pop = (int32_t*)malloc((size_t)n * HWQ_MAX_ASK * sizeof(int32_t));
    float* px = (float*)malloc((size_t)n * HWQ_MAX_ASK * sizeof(float));
    float* py = (float*)malloc((size_t)n * HWQ_MAX_ASK * sizeof(float));
    if (!tapes || !active || !pn || !pop || !px || !py) { fprintf(stderr, "oracle/nb_model.c: out of memory\n"); abort(); }
    for (int64_t i = 0; i < n; ++i) active[i] = i;
    int64_t na = n;
    while (na > 0) {
#pragma omp parallel for schedule(dynamic, 256)
        for (int64_t j = 0; j < na; ++j) {
            const prnb_waiting* w = &W[active[j]];
            tape_t* T = &tapes[active[j]];
            hwq Q;
            Q.tape = T->t; Q.len = T->len; Q.pos = 0; Q.npend = 0;
            int32_t c;
            if (prnb_one_q(w->M, w->s, w->a, w->bm1, k0, k1, w->cell, w->gene, 0, &Q, 1, &c)) {
                out[w->idx] = c;
                pn[j] = 0;
            } else {
                pn[j] = Q.npend;
                for (int q = 0; q < Q.npend; ++q) { pop[j * HWQ_MAX_ASK + q] = Q.pend_op[q]; px[j * HWQ_MAX_ASK + q] = Q.pend_x[q]; }
            }
        }
        /* compact the questions (in place: the write position never passes the read position), ask, hand out the answers */
        int64_t nq = 0, nw = 0;
        for (int64_t j = 0; j < na; ++j)
            for (int q = 0; q < pn[j]; ++q, ++nq) { pop[nq] = pop[j * HWQ_MAX_ASK + q]; px[nq] = px[j * HWQ_MAX_ASK + q]; }
        if (nq) g_query(nq, pop, px, py);
        g_query_rounds += 1;
        g_query_values += nq;
        nq = 0;
        for (int64_t j = 0; j < na; ++j) {
            if (!pn[j]) { free(tapes[active[j]].t); tapes[active[j]].t = 0; continue; }
            tape_t* T = &tapes[active[j]];
            if (T->len + pn[j] > T->cap) {
                T->cap = T->cap ? 2 * T->cap : 16;
                T->t = (hwq_pair*)realloc(T->t, (size_t)T->cap * sizeof(hwq_pair));
                if (!T->t) { fprintf(stderr, "oracle/nb_model.c: out of memory\n"); abort(); }
            }
            for (int q = 0; q < pn[j]; ++q, ++nq) { T->t[T->len].x = px[nq]; T->t[T->len].y = py[nq]; T->len += 1; }
            active[nw++] = active[j];
        }
        na = nw;
    }
    free(tapes); free(active); free(pn); free(pop); free(px); free(py);
}

/* ---- exported entry points (ctypes) --------------------------------------- */

/* Per-gene parameters reach the sampler as binary32:  a = (float)alpha  and
 * bm1 = (float)(beta - 1.0)  with the subtraction in binary64 -- the reference's
 * own examples use beta = 1 + 1e-8 (examples/linear.ipynb), which binary32 beta
 * cannot hold.  Scalings are rounded to binary32. */
#define GENE_A(alpha, g)   ((float)(alpha)[g])
#define GENE_BM1(beta, g)  ((float)((beta)[g] - 1.0))

PRNB_EXPORT void prnb_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
}

PRNB_EXPORT void prnb_philox_rounds(int rounds, const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    philox4x32_r(rounds, ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
}

PRNB_EXPORT int prnb_count_rounds(void) { return PRNB_COUNT_ROUNDS; }
/* test-only (see g_count_rounds): 0 restores the definition's round count */
PRNB_EXPORT void prnb_set_count_rounds(int rounds) { g_count_rounds = rounds > 0 ? rounds : PRNB_COUNT_ROUNDS; }

/* Install (or, with NULL pointers, remove) the tables of the three hardware functions -- see the comment at hw_rcp.
 * The arrays stay the caller's and must outlive their use. */
PRNB_EXPORT void prnb_set_hw_tables(const float* rcp_mant, const float* log2_tab, uint32_t log2_first, uint32_t log2_count,
                                    const float* exp2_tab, uint32_t exp2_first, uint32_t exp2_count)
{
    g_rcp_mant = rcp_mant;
    g_log2_tab = log2_tab; g_log2_first = log2_first; g_log2_count = log2_tab ? log2_count : 0;
    g_exp2_tab = exp2_tab; g_exp2_first = exp2_first; g_exp2_count = exp2_tab ? exp2_count : 0;
}
/* Install (NULL: remove) the function that answers the gamma-Poisson class's questions: y[i] = op[i](x[i]) for n
 * (op, x) pairs, ops as in the HWOP_ enum (= the op codes of prosstt_amd_hw_math_at).  Called from one thread. */
PRNB_EXPORT void prnb_set_hw_query(prnb_hw_query_fn fn) { g_query = fn; }
/* a query function made of the libm stand-ins: the query machinery must then reproduce the direct evaluation bit for
 * bit (tests/test_nb_model.py: the replay mechanism checked without a GPU) */
static void query_standins(int64_t n, const int32_t* op, const float* x, float* y)
{
    for (int64_t i = 0; i < n; ++i) y[i] = hw_standin(op[i], x[i]);
}
PRNB_EXPORT void prnb_set_hw_query_standins(void) { g_query = query_standins; }
PRNB_EXPORT void prnb_query_stats(int64_t* rounds, int64_t* values) { *rounds = g_query_rounds; *values = g_query_values; }
/* 1: the device's tables and query function are installed (bit-exact mode); 0: the libm stand-ins (law tests without a GPU) */
PRNB_EXPORT int prnb_hw_mode(void) { return g_rcp_mant && g_log2_tab && g_exp2_tab && g_query && g_query != query_standins; }

/* the model's view of the hardware functions (tests of the tables): which = 0 HW_RCP(x), 1 HW_LOG2(x), 2 HW_EXP2(-x) */
PRNB_EXPORT void prnb_hw_math(int which, const float* x, float* y, int64_t n)
{
    for (int64_t i = 0; i < n; ++i)
        y[i] = which == 0 ? hw_rcp(x[i]) : (which == 1 ? hw_log2(x[i]) : hw_exp2neg(x[i]));
}

/* elementwise math probes: which = 0 rcp, 1 log, 2 log1p, 3 exp, 4 cos2pi(bits of x), 5 unif(bits) */
PRNB_EXPORT PRNB_CLONES void prnb_math(int which, const float* x, float* y, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) {
        switch (which) {
        case 0: y[i] = det_rcp(x[i]); break;
        case 1: y[i] = det_log(x[i]); break;
        case 2: y[i] = det_log1p(x[i]); break;
        case 3: y[i] = det_exp(x[i]); break;
        case 4: y[i] = det_cos2pi(f2u(x[i])); break;
        case 5: y[i] = unif(f2u(x[i])); break;
        case 6: y[i] = det_log1pmx(x[i], 1.0f + x[i]); break;
        default: y[i] = 0.0f;
        }
    }
}

/*
 * Same signature as the product's prosstt_amd_sample_counts (include/prosstt_amd.h)
 * minus flags/stream: counts[n*ld + g] for cell n (global index cell_offset+n), gene g.
 * In query mode the samples of the gamma-Poisson class are collected and resolved together (resolve_waiting).
 */
PRNB_EXPORT PRNB_CLONES void prnb_sample_counts(const float* means, int64_t rows, int32_t G,
                                                const int32_t* row_of_cell, const double* scaling,
                                                const double* alpha, const double* beta, int64_t N,
                                                uint64_t seed, uint64_t cell_offset,
                                                const int64_t* cell_index, int32_t* out, int64_t ld)
{
    (void)rows;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    prnb_wait_list all = {0, 0, 0};
#pragma omp parallel
    {
        prnb_wait_list mine = {0, 0, 0};
        hwq Q;
#pragma omp for schedule(static) nowait
        for (int64_t n = 0; n < N; ++n) {
            const float* mrow = means + (int64_t)row_of_cell[n] * G;
            float s = (float)scaling[n];
            const uint64_t cell = cell_index ? (uint64_t)cell_index[n] : cell_offset + (uint64_t)n;
            for (int32_t g = 0; g < G; ++g) {
                Q.tape = 0; Q.len = 0; Q.pos = 0; Q.npend = 0;
                if (!prnb_one_q(mrow[g], s, GENE_A(alpha, g), GENE_BM1(beta, g), k0, k1, cell, (uint32_t)g, 0,
                                g_query ? &Q : 0, 1, &out[n * ld + g])) {
                    prnb_waiting w = {n * ld + g, mrow[g], s, GENE_A(alpha, g), GENE_BM1(beta, g), cell, (uint32_t)g};
                    wait_push(&mine, w);
                }
            }
        }
#pragma omp critical
        {
            for (int64_t i = 0; i < mine.n; ++i) wait_push(&all, mine.v[i]);
        }
        free(mine.v);
    }
    resolve_waiting(all.v, all.n, k0, k1, out);
    free(all.v);
}

/* Deterministic intermediates (mu, p, r) exactly as the kernel forms them, plus the path taken. */
PRNB_EXPORT PRNB_CLONES void prnb_nb_params(const float* means, int64_t rows, int32_t G,
                                            const int32_t* row_of_cell, const double* scaling,
                                            const double* alpha, const double* beta, int64_t N,
                                            float* mu, float* p, float* r, int32_t* path)
{
    (void)rows;
    for (int64_t n = 0; n < N; ++n) {
        const float* mrow = means + (int64_t)row_of_cell[n] * G;
        for (int32_t g = 0; g < G; ++g) {
            prnb_detail d;
            int32_t c;
            prnb_one_q(mrow[g], (float)scaling[n], GENE_A(alpha, g), GENE_BM1(beta, g), 0u, 0u, 0u, (uint32_t)g, &d, 0, 0, &c);
            mu[n * (int64_t)G + g] = d.m; p[n * (int64_t)G + g] = d.p; r[n * (int64_t)G + g] = d.r;
            if (path) path[n * (int64_t)G + g] = d.path;
        }
    }
}

/* path (0 degenerate / 1 inversion / 2 gamma-Poisson) and count of selected samples: sample i is cell n = cells[i] of
 * the call's arrays, gene genes[i] (tests of the list the streaming kernel leaves to its second kernel). */
PRNB_EXPORT PRNB_CLONES void prnb_sample_selected(const float* means, int64_t rows, int32_t G, const int32_t* row_of_cell,
                                                  const double* scaling, const double* alpha, const double* beta,
                                                  uint64_t seed, uint64_t cell_offset, const int64_t* cell_index,
                                                  const int64_t* cells, const int32_t* genes, int64_t count,
                                                  int32_t* out_path, int32_t* out_count)
{
    (void)rows;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    prnb_wait_list all = {0, 0, 0};
    for (int64_t i = 0; i < count; ++i) {
        const int64_t n = cells[i];
        const int32_t g = genes[i];
        const uint64_t cell = cell_index ? (uint64_t)cell_index[n] : cell_offset + (uint64_t)n;
        prnb_detail d;
        hwq Q;
        Q.tape = 0; Q.len = 0; Q.pos = 0; Q.npend = 0;
        const float M = means[(int64_t)row_of_cell[n] * G + g];
        if (!prnb_one_q(M, (float)scaling[n], GENE_A(alpha, g), GENE_BM1(beta, g), k0, k1, cell, (uint32_t)g, &d,
                        g_query ? &Q : 0, 1, &out_count[i])) {
            prnb_waiting w = {i, M, (float)scaling[n], GENE_A(alpha, g), GENE_BM1(beta, g), cell, (uint32_t)g};
            wait_push(&all, w);
        }
        out_path[i] = d.path;
    }
    resolve_waiting(all.v, all.n, k0, k1, out_count);
    free(all.v);
}

/*
 * Device-mode expression programs ("PRLW-1"): the walk of simulation.diffusion
 * (/root/reference/prosstt/simulation.py:89-124) with counter-based variates instead of numpy's
 * stream.  Program k of walk stream `sid` draws from Philox4x32-10 with key
 * (seed_lo ^ 0x57414C4B, seed_hi) and counter (k, j, sid_lo, sid_hi):
 *   call j = 0:  w0 -> start = log(1.5*U),  (w1, w2) -> vel0 = 0.2*N,  w3 -> eta = U
 *   call j >= 1: (w0, w1) -> N for step 2(j-1),  (w2, w3) -> N for step 2(j-1)+1,  eps = (2/T)*N
 * U = unif(w), N = sqrt(-2 log U) cos(2 pi w') in the deterministic binary32 math above; the
 * recurrence  walk[t+1] = walk[t] + vel[t],  vel[t+1] = eta*vel[t] + eps[t]  runs in binary64.
 * out[t*K + k] = walk_k[t]  (the (T, K) layout sim_expr_branch returns).
 */
PRNB_EXPORT PRNB_CLONES void prnb_lineage_walk(uint64_t seed, uint64_t sid, int32_t T, int32_t K, double* out)
{
    const uint32_t k0 = (uint32_t)seed ^ 0x57414C4Bu, k1 = (uint32_t)(seed >> 32);
    for (int32_t k = 0; k < K; ++k) {
        uint32_t w[4];
        philox4x32_10((uint32_t)k, 0u, (uint32_t)sid, (uint32_t)(sid >> 32), k0, k1, w);
        double walk = (double)det_log(1.5f * unif(w[0]));
        double vel = 0.2 * (double)(sqrtf(-2.0f * det_log(unif(w[1]))) * det_cos2pi(w[2]));
        const double eta = (double)unif(w[3]);
        const double s_eps = 2.0 / (double)T;
        out[k] = walk;
        for (int32_t t = 0; t + 1 < T; ++t) {
            if ((t & 1) == 0) philox4x32_10((uint32_t)k, 1u + (uint32_t)(t >> 1), (uint32_t)sid, (uint32_t)(sid >> 32), k0, k1, w);
            const uint32_t wa = (t & 1) ? w[2] : w[0], wb = (t & 1) ? w[3] : w[1];
            const double eps = s_eps * (double)(sqrtf(-2.0f * det_log(unif(wa))) * det_cos2pi(wb));
            walk = walk + vel;
            vel = eta * vel + eps;
            out[(int64_t)(t + 1) * K + k] = walk;
        }
    }
}

/* n draws from one parameter set (law tests): cell = first_cell + i, gene fixed. */
PRNB_EXPORT PRNB_CLONES void prnb_sample_iid(float m, double a, double b, uint64_t seed,
                                             uint64_t first_cell, uint32_t gene, int64_t n,
                                             int32_t* out)
{
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const float af = (float)a, bm1 = (float)(b - 1.0);
    if (!g_query) {
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i)
            prnb_one_q(m, 1.0f, af, bm1, k0, k1, first_cell + (uint64_t)i, gene, 0, 0, 1, &out[i]);
        return;
    }
    prnb_wait_list all = {0, 0, 0};
    for (int64_t i = 0; i < n; ++i) {
        hwq Q;
        Q.tape = 0; Q.len = 0; Q.pos = 0; Q.npend = 0;
        if (!prnb_one_q(m, 1.0f, af, bm1, k0, k1, first_cell + (uint64_t)i, gene, 0, &Q, 1, &out[i])) {
            prnb_waiting w = {i, m, 1.0f, af, bm1, first_cell + (uint64_t)i, gene};
            wait_push(&all, w);
        }
    }
    resolve_waiting(all.v, all.n, k0, k1, out);
    free(all.v);
}
