// PRNB-7 count sampler, device side (gfx950).  DESIGN.md section 4 defines it; this header holds
// its building blocks -- the counter generator, the functions of the definition, the
// parameters of a sample, the inversion walk -- that the kernels (k3_stream.h, k3_heavy.h,
// nb_params_kernel) are made of.  Replaces, per (cell, gene):
//   count_model.get_pr_umi                 /root/reference/prosstt/count_model.py:131-161
//   scipy.stats.nbinom(n=r,p=1-p).rvs()    /root/reference/prosstt/simulation.py:647-648
//
// Arithmetic of the definition: IEEE binary32 add / mul / fma and integer ops (the translation
// unit is compiled with -ffp-contract=off and every fused multiply-add below is spelled out), plus
// six functions of the gfx950 hardware, one instruction each, deterministic on the chip: HW_RCP = v_rcp_f32,
// HW_LOG2 = v_log_f32, HW_EXP2 = v_exp_f32 (both classes), HW_SQRT = v_sqrt_f32, HW_RSQ = v_rsq_f32,
// HW_COS = v_cos_f32 (argument in revolutions) -- the gamma-Poisson class only (PRNB-7; PRNB-6 drew that class in
// polynomial arithmetic: 665 vector lane-instructions per sample, the kernel furthest below its roof).  The scalar
// model that checks the kernels (a test-side C file) takes the values of rcp / log2 / exp2 of the inversion class
// from tables that a probe kernel (hw_math_kernel, prosstt_amd.hip) writes on the device under test, and ASKS the
// device for the gamma-Poisson class's values argument by argument (hw_math_at_kernel: y[i] = op(x[i])); everything
// else of the model is its own C.  So counts still compare bit for bit.
// (PRNB-4 defined P(X = 0) by polynomial log/exp and a Newton reciprocal and let the streaming kernel
// use the hardware functions inside error margins, with a give-up list for samples near a threshold:
// a fifth of the kernel's instructions; profiles/r04_ablation.txt.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace prnb {

constexpr float kLightT2 = 27.4112f;   // inversion iff theta <= 24 and t2 = -log2 P(X=0) < 27.4112 (= 19 / ln 2): P0 * 2^32 >= 24
constexpr float kLightTheta = 24.0f;   // tail ratio <= 24/25 (PRNB-6; 16 until round 5: the samples between are cheaper as walks than as
                                       // gamma-Poisson draws -- K3h -35 % on a 32-branch tree for +0.4 % of the stream kernel, profiles/r05_ablation.txt section 7)
constexpr float kThetaMin = 1.1920929e-7f;   // 2^-23: 1 + theta > 1 in binary32 (below this NB == Poisson to 1e-7 of the variance)
constexpr float kThetaMax = 1.0e18f;
constexpr float kRMin = 9.094947e-13f;       // 2^-40
// The inversion walk ends at k = kWalkEnd at the latest: the group k = 1021..1024 is the last one, and when it ends
// without a negative remainder the count is 1024 (in the inversion class the mean is below 142 and the tail ratio at
// most 24/25: P(X > 1022) < 1e-12, scipy's nbinom.sf at the class corner).  The 1/k table holds 1/k for 1 <= k < kKTab (K3h's walk pass reads four groups).
// Groups: k = 0..4 (stage 2 of the streaming kernel), then four at a time from k = 5 -- so a 16-byte read of 1/(k+1)..1/(k+4)
// is aligned when the table's entry 6 is (kTabShift).
constexpr int kWalkEnd = 1024;
constexpr int kKTab = 1048;            // (K3h's walk pass reads 1/(k+1) .. 1/(k+16) from k = 1021 at the latest)
constexpr int kTabShift = 2;           // inv_k = (16-byte aligned store) + kTabShift: &inv_k[6 + 4 j] is 16-byte aligned
constexpr float kPoisInv = 10.0f;
constexpr float kLamBig = 4194304.0f;        // 2^22
constexpr int kMaxTries = 64;

#define PRNB_FMA(a, b, c) __builtin_fmaf((a), (b), (c))

__device__ __forceinline__ uint32_t f2u(float x) { return __float_as_uint(x); }
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }

struct Words { uint32_t w[4]; };

// The three hardware functions of the definition (one instruction each; the model reads their values from tables
// written by hw_math_kernel on the same device).
__device__ __forceinline__ float hw_rcp(float x) { return __builtin_amdgcn_rcpf(x); }     // v_rcp_f32
__device__ __forceinline__ float hw_log2(float x) { return __builtin_amdgcn_logf(x); }    // v_log_f32
__device__ __forceinline__ float hw_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32
// ... and the three more of the gamma-Poisson class (the model asks the device for their values: hw_math_at_kernel)
__device__ __forceinline__ float hw_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }   // v_sqrt_f32
__device__ __forceinline__ float hw_rsq(float x) { return __builtin_amdgcn_rsqf(x); }     // v_rsq_f32
__device__ __forceinline__ float hw_cos(float x) { return __builtin_amdgcn_cosf(x); }     // v_cos_f32: cos(2 pi x)

// Philox4x32-R (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11).  Rounds
// from kXor3From on spell the two 3-input xors as one v_bitop3_b32 each (gfx950); callers whose counter is
// partly wave-uniform leave the first two rounds to the compiler, which moves their uniform halves to the
// scalar unit.
template <int kRounds, int kXor3From>
__device__ __forceinline__ Words philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                            uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int round = 0; round < kRounds; ++round) {
        // one 32x32->64 product per multiplier (v_mad_u64_u32) instead of separate hi and lo multiplies
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        if (round >= kXor3From) {
            c0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
            c2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
        } else {
            c0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
            c2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        }
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    Words r;
    r.w[0] = c0; r.w[1] = c1; r.w[2] = c2; r.w[3] = c3;
    return r;
}

// The count sampler draws from Philox4x32-7: the fewest rounds of Philox4x32 that pass BigCrush
// (Salmon et al. 2011, section 5 and table 2: "Crush-resistant"; 10 rounds are the library default for
// margin).  Every counter is used once, a sample's uniforms are never compared with a neighbour's, and
// the statistical tests of tests/ run on this generator; the three rounds are 3 % of the whole kernel.
// The device-mode lineage walk (PRLW-1, not a hot path) keeps the 10-round default.
constexpr int kCountRounds = 7;
template <int kXor3From = kCountRounds>
__device__ __forceinline__ Words philox_count(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
    return philox4x32<kCountRounds, kXor3From>(c0, c1, c2, c3, k0, k1);
}
__device__ __forceinline__ Words philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
    return philox4x32<10, 10>(c0, c1, c2, c3, k0, k1);
}

// 1/x, x > 0 normal: integer seed (5 % error) + 3 Newton steps
__device__ __forceinline__ float det_rcp(float x)
{
    float y = u2f(0x7EF311C7u - f2u(x));
    float e = PRNB_FMA(-x, y, 1.0f); y = PRNB_FMA(y, e, y);
    e = PRNB_FMA(-x, y, 1.0f); y = PRNB_FMA(y, e, y);
    e = PRNB_FMA(-x, y, 1.0f); y = PRNB_FMA(y, e, y);
    return y;
}

// log(1+f) - f + f^2/2 on [sqrt(1/2)-1, sqrt(2)-1]
__device__ __forceinline__ float log_tail(float f)
{
    float p = 7.0376836292e-2f;
    p = PRNB_FMA(p, f, -1.1514610310e-1f);
    p = PRNB_FMA(p, f, 1.1676998740e-1f);
    p = PRNB_FMA(p, f, -1.2420140846e-1f);
    p = PRNB_FMA(p, f, 1.4249322787e-1f);
    p = PRNB_FMA(p, f, -1.6668057665e-1f);
    p = PRNB_FMA(p, f, 2.0000714765e-1f);
    p = PRNB_FMA(p, f, -2.4999993993e-1f);
    p = PRNB_FMA(p, f, 3.3333331174e-1f);
    return (p * f) * (f * f);
}

__device__ __forceinline__ float det_log_c(float x, float c)
{
    const uint32_t ix = f2u(x);
    const int32_t e = (int32_t)(ix - 0x3F3504F3u) >> 23;
    const float mant = u2f(ix - ((uint32_t)e << 23));
    const float scale = u2f((uint32_t)(127 - e) << 23);
    const float f = (mant - 1.0f) + c * scale;
    const float fe = (float)e;
    float y = log_tail(f);
    y = PRNB_FMA(fe, -2.12194440e-4f, y);
    y = PRNB_FMA(-0.5f, f * f, y);
    return PRNB_FMA(fe, 0.693359375f, f + y);
}
__device__ __forceinline__ float det_log(float x) { return det_log_c(x, 0.0f); }

__device__ __forceinline__ float det_log1p(float t)
{
    const float u = 1.0f + t;
    const float c = (t >= 1.0f) ? (1.0f - (u - t)) : (t - (u - 1.0f));
    return det_log_c(u, c);
}

__device__ __forceinline__ float det_log1pmx(float d, float rho)
{
    if (d >= -0.29289323f && d < 0.41421354f)
        return PRNB_FMA(-0.5f, d * d, log_tail(d));
    return det_log(rho) - d;
}

// det_exp without the underflow guard: the same value for every x > -87
__device__ __forceinline__ float det_exp_small(float x)
{
    const float z = __builtin_floorf(PRNB_FMA(x, 1.44269504088896341f, 0.5f));
    float r = PRNB_FMA(z, -0.693359375f, x);
    r = PRNB_FMA(z, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = PRNB_FMA(p, r, 1.3981999507e-3f);
    p = PRNB_FMA(p, r, 8.3334519073e-3f);
    p = PRNB_FMA(p, r, 4.1665795894e-2f);
    p = PRNB_FMA(p, r, 1.6666665459e-1f);
    p = PRNB_FMA(p, r, 5.0000001201e-1f);
    const float y = PRNB_FMA(p, r * r, r) + 1.0f;
    return y * u2f((uint32_t)((int32_t)z + 127) << 23);
}

__device__ __forceinline__ float det_exp(float x)
{
    if (!(x > -87.0f)) return 0.0f;
    return det_exp_small(x);
}
__device__ __forceinline__ float det_cos2pi(uint32_t w)
{
    const uint32_t j = w >> 29;
    float f = (float)(w & 0x1FFFFFFFu) * 1.862645149230957e-9f;
    if (j & 1u) f = f - 1.0f;
    const uint32_t q = ((j + 1u) >> 1) & 3u;
    const float y = f * 0.78539816339744830962f;
    const float z = y * y;
    float s = PRNB_FMA(-1.9515295891e-4f, z, 8.3321608736e-3f);
    s = PRNB_FMA(s, z, -1.6666654611e-1f);
    s = PRNB_FMA(s * z, y, y);
    float c = PRNB_FMA(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    c = PRNB_FMA(c, z, 4.166664568298827e-2f);
    c = PRNB_FMA(c * z, z, PRNB_FMA(-0.5f, z, 1.0f));
    const float v = (q & 1u) ? s : c;
    return (q == 1u || q == 2u) ? -v : v;
}

__device__ __forceinline__ float unif(uint32_t w) { return ((float)w + 0.5f) * 2.3283064365386963e-10f; }

// ---- the gamma-Poisson class on the hardware's transcendentals (PRNB-7) ----
constexpr float kLn2 = 0.69314718f, kLog2e = 1.44269504f, kMinus2Ln2 = -1.3862944f, kTwoM32 = 2.3283064365386963e-10f;
// a standard normal by Box-Muller: sqrt(-2 ln u) cos(2 pi w / 2^32) in three hardware instructions
__device__ __forceinline__ float hw_normal(uint32_t wa, uint32_t wb)
{
    return hw_sqrt(kMinus2Ln2 * hw_log2(unif(wa))) * hw_cos((float)wb * kTwoM32);
}
// log(1+d) - d: the polynomial where the difference cancels, the hardware's log2 of rho = 1 + d elsewhere
__device__ __forceinline__ float hw_log1pmx(float d, float rho)
{
    if (d >= -0.29289323f && d < 0.41421354f)
        return PRNB_FMA(-0.5f, d * d, log_tail(d));
    return PRNB_FMA(kLn2, hw_log2(rho), -d);
}

__device__ __forceinline__ float det_sqrt(float x) { return __builtin_sqrtf(x); }  // IEEE (see Makefile)

// Inversion by chop-down on a binary32 remainder; inv_k = LDS table of 1/k
// (kKTab entries, &inv_k[6] 16-byte aligned: kTabShift).
// P(k+1) = P(k) * (q + (mp - q)/(k+1)) -- the ratio (mp + k*q)/(k+1) by ONE fma from the table's 1/(k+1); k = 0:
// P(0) * mp --, carried scaled by 2^32.  The remainder starts as
// (float)w and every term is subtracted from it; the draw is the first k whose subtraction leaves it
// negative.  Terms come in groups (k = 0..4, then four at a time: the streaming kernel's passes); when
// a group ends without a negative remainder and its last term is below 1 (the pmf has fallen under
// 2^-32 before w is used up: mass lost to rounding, < 1e-6) the draw is that group's last k, as it is
// when the group is the last one (k = kWalkEnd).
//
// For a whole wave at once (K3h's Poisson draws below lambda = 10: q = 0): every lane carries its own walk, all
// lanes are at the same k.  A group of four terms is evaluated without a branch per term (terms behind a lane's
// draw are computed and ignored), and the reciprocals of two groups come from one wave-uniform LDS round trip.
// Must be called by all lanes of the wave (`active` = this lane has a walk); returns when no lane is walking any more.
__device__ __forceinline__ int32_t chop_down_wave(bool active, uint32_t w, float p0, float mp, float q,
                                                  const float* inv_k)
{
    float ps = __builtin_fminf(p0, 0.99999994f) * 4294967296.0f;
    float rem = (float)w;
    const float d = mp - q;
    // k = 0 .. 4
    const float r0 = rem - ps;
    const float p1 = ps * mp;
    const float r1 = r0 - p1;
    const float p2 = p1 * PRNB_FMA(d, inv_k[2], q);
    const float r2 = r1 - p2;
    const float p3 = p2 * PRNB_FMA(d, inv_k[3], q);
    const float r3 = r2 - p3;
    const float p4 = p3 * PRNB_FMA(d, inv_k[4], q);
    const float r4 = r3 - p4;
    int32_t res = (r0 < 0.0f) ? 0 : ((r1 < 0.0f) ? 1 : ((r2 < 0.0f) ? 2 : ((r3 < 0.0f) ? 3 : 4)));
    bool busy = active && !((r0 < 0.0f) || (r1 < 0.0f) || (r2 < 0.0f) || (r3 < 0.0f) || (r4 < 0.0f) || (p4 < 1.0f));
    ps = p4 * PRNB_FMA(d, inv_k[5], q);
    rem = r4;
    const float4* tab = reinterpret_cast<const float4*>(__builtin_assume_aligned(inv_k + 6, 16));
    int k = 5;
    auto group = [&](const float4 inv) {
        const float a1 = rem - ps;
        const float q2 = ps * PRNB_FMA(d, inv.x, q);
        const float a2 = a1 - q2;
        const float q3 = q2 * PRNB_FMA(d, inv.y, q);
        const float a3 = a2 - q3;
        const float q4 = q3 * PRNB_FMA(d, inv.z, q);
        const float a4 = a3 - q4;
        const bool end = (a1 < 0.0f) || (a2 < 0.0f) || (a3 < 0.0f) || (a4 < 0.0f) || (q4 < 1.0f) || (k + 3 >= kWalkEnd);
        const int32_t at = (a1 < 0.0f) ? k : ((a2 < 0.0f) ? k + 1 : ((a3 < 0.0f) ? k + 2 : k + 3));
        if (busy && end) { res = at; busy = false; }
        ps = q4 * PRNB_FMA(d, inv.w, q);
        k += 4;
        rem = a4;
    };
    // two groups per LDS round trip (k + 3 reaches kWalkEnd in a first group: the second one then runs with
    // busy = false everywhere and reads 1/k up to k = kWalkEnd + 5 < kKTab)
    while (__builtin_amdgcn_ballot_w64(busy) != 0ull) {
        const float4 ia = tab[(k - 5) >> 2], ib = tab[((k - 5) >> 2) + 1];
        group(ia);
        group(ib);
    }
    return res;
}

__device__ __forceinline__ float logfact_small(int k)
{
    // log(k!) for k < 10, binary32-rounded
    switch (k) {
    case 0: case 1: return 0.0f;
    case 2: return 0.69314718f;
    case 3: return 1.7917595f;
    case 4: return 3.1780538f;
    case 5: return 4.7874917f;
    case 6: return 6.5792512f;
    case 7: return 8.5251614f;
    case 8: return 10.604603f;
    default: return 12.801827f;
    }
}

struct Params {
    float m, theta;
    float iu;      // HW_RCP(1 + theta):  mp = m * iu,  q = theta * iu
    float t2;      // -log2 P(X = 0) = m * (HW_LOG2(1 + theta) * HW_RCP((1 + theta) - 1))
    float inv_th;  // HW_RCP(theta): r = m * inv_th (gamma-Poisson class)
    bool valid;    // m > 0 and a*m + b - 1 > 0
    bool light;    // inversion class: theta <= kLightTheta and t2 < kLightT2
};

// Per-gene factor of the streaming kernel's zero test.  With theta = a*m + b - 1 >= b - 1 =: c
// (a >= 0) and f(theta) = log1p(theta)/theta decreasing, -log P(X = 0) = m*f(theta) <= m*f(c):
// x = m * phi, phi = f(c), bounds it from above; phi = 1 (f <= 1 for every theta > 0) when that
// argument does not apply.  A sample of the gamma-Poisson class must never pass the zero test
// (it draws from other counter domains): t > 19 implies x > 18.9, where the test's cubic is
// negative; theta > 24 implies the same once the gene's theta stays under 24 for every x <= 1.7 --
// a gene for which it does not gets a huge phi, so that all of its samples take the exact path.
__device__ __forceinline__ float zero_test_factor(float a, float bm1)
{
    float phi = 1.0f;
    if (a >= 0.0f && bm1 >= 2.44140625e-4f) phi = det_log1p(bm1) * det_rcp(bm1);
    const float th_edge = PRNB_FMA(a, 1.7f * det_rcp(phi), bm1);      // theta at x = 1.7
    if (!(__builtin_fmaxf(th_edge, bm1) <= kLightTheta - 0.1f)) phi = 3.0e38f;      // also NaN
    return phi;
}

// theta >= 2^-23 as a plain v_max_f32 (the builtin puts a canonicalising copy in front)
__device__ __forceinline__ float clamp_theta_min(float theta_raw)
{
    float theta;
    asm("v_max_f32 %0, 0x34000000, %1" : "=v"(theta) : "v"(theta_raw));
    return theta;
}

// P(X = 0) of the inversion class by the hardware functions: u1 = fl(1 + theta) > 1; log2(u1)/(u1 - 1) is
// log2(1 + th')/th' of the th' = u1 - 1 that u1 stands for exactly (the rounding of the sum cancels), within
// 6e-8 of the same quotient at theta.
struct HwP0 { float iu, t2; };
__device__ __forceinline__ HwP0 hw_p0(float m, float theta)
{
    HwP0 h;
    const float u1 = 1.0f + theta;
    h.iu = hw_rcp(u1);
    h.t2 = m * (hw_log2(u1) * hw_rcp(u1 - 1.0f));
    return h;
}

// the parameters of a sample from its scaled mean m = M * s
__device__ __forceinline__ Params make_params_m(float m, float a, float bm1)
{
    Params P;
    P.m = m;
    const float theta_raw = PRNB_FMA(a, m, bm1);
    P.valid = (m > 0.0f) && (theta_raw > 0.0f);
    const float theta = __builtin_fminf(clamp_theta_min(theta_raw), kThetaMax);
    const HwP0 h = hw_p0(m, theta);
    P.theta = theta;
    P.iu = h.iu;
    P.t2 = h.t2;
    P.inv_th = hw_rcp(theta);
    P.light = (theta <= kLightTheta) && (h.t2 < kLightT2);
    return P;
}
__device__ __forceinline__ Params make_params(float M, float s, float a, float bm1)
{
    return make_params_m(M * s, a, bm1);
}

}  // namespace prnb
