// K3h: the samples the streaming kernel (k3_stream.h) listed instead of drawing them: the
// gamma-Poisson class of PRNB-7 (prnb_device.h; theta above 24 or -log2 P(X = 0) of 27.4 or more, one to
// three in a thousand of a typical workload), the walks that were still running when their strip was done
// and the walks that ran past term 252 (seventy per 10^9 samples): both continued here from the state the streaming
// kernel hands over.  Only the samples of a region whose list overflowed start at k = 0 here.
// Both halves of the gamma-Poisson path are rejection samplers; run lane-per-sample
// they would make every wave repeat each half until its unluckiest lane is accepted.  Here
// every ATTEMPT is a stack entry: a gamma pass runs one Marsaglia-Tsang attempt for 64
// entries of HG (accepted -> the Poisson stack HP, rejected -> back on HG with the next
// attempt number), a Poisson pass one PTRS attempt for 64 entries of HP.  Attempts are pure
// functions of (parameters, seed, cell, gene, attempt), so the order of evaluation cannot
// change a result, and every pass runs with all lanes doing the same thing.  The walks
// keep their state in registers across passes, and idle lanes take the next entries (below).
//
// PRNB-7 (round 6): the class's transcendentals are the hardware's -- v_log_f32, v_exp_f32, v_sqrt_f32, v_rsq_f32,
// v_cos_f32, v_rcp_f32, one instruction each (PRNB-6: polynomial log 25, sqrt 17, cos 29, Newton reciprocal 7
// instructions; ~665 vector lane-instructions per listed sample) -- and a list entry carries its sample's scaled mean,
// so that nothing is gathered here but the two per-gene parameters.
#pragma once
#include "prnb_device.h"
#include "k3_stream.h"

namespace k3 {

constexpr int kHeavyBlock = 256;
constexpr int kHCap = 128;     // < 64 left over + 64 pushed (new entries, or re-pushed ones after 64 were popped)
constexpr uint32_t kWalkerBlocks = 6u;   // of every 16 blocks, the ones that walk (2: 81 / 93 us at C3 / T32, 3: 57 / 71, 4: 50 / 68, 5: 42 / 68, 6: 41 / 68, 8: 38 / 71)
constexpr int kWCap = 96;      // walks waiting for a lane: < 32 left over + 64 pushed by one step of the list traversal (five blocks per CU: 30 496 B)

struct HGEntry { int32_t n, g, attempt; float m; };
struct HPEntry { int32_t n, g; float lam; int32_t attempt; };
// a walk waiting for a lane: its state {the next term, d = mp - q, q, the remainder}, where it goes, the next term's k
struct HWId { int32_t n, g; };

struct HeavyLds {
    HGEntry hg[kHCap];
    HPEntry hp[kHCap];
    f32x4_t wst[kWCap];
    HWId wid[kWCap];
    uint16_t wk[kWCap];
};

// heavy: what the streaming kernel's waves listed (k3::HeavyList): per region (of the streaming kernel's block -> (gene
// tile, strip group) map) the first kDense entries and the walk states, kSegs dense segments of {cell, gene, m} with what
// the regions of a hot gene tile list beyond that (dealt out to the waves 64 at a time), and the regions whose list or
// segment was too small: those are redone here sample by sample instead -- slow, but any parameter set stays correct.
__global__ __launch_bounds__(kHeavyBlock) void sample_counts_heavy_kernel(
    const HeavyList* __restrict__ heavy_ptr, uint32_t regions, int32_t strips, int32_t strip_cells,
    const float* __restrict__ means, int64_t rows,
    int32_t G, const int32_t* __restrict__ row_of_cell, const float* __restrict__ scal,
    const float* __restrict__ ga, const float* __restrict__ gbm1, int64_t N, uint32_t k0, uint32_t k1,
    uint64_t cell_offset, const int64_t* __restrict__ cell_index, int32_t* __restrict__ out, int64_t ld,
    int64_t* __restrict__ check_words, uint32_t check_request, uint32_t check_bad_row, uint32_t check_verdict)
{
    __shared__ __attribute__((aligned(16))) float inv_k_store[prnb::kTabShift + prnb::kKTab];       // 1/k, k < kKTab (a walk ends at kWalkEnd)
    float* const inv_k = inv_k_store + prnb::kTabShift;                      // &inv_k[6 + 4 j] is 16-byte aligned
    __shared__ HeavyLds lds_all[kHeavyBlock / 64];
    const HeavyList heavy = *heavy_ptr;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    HeavyLds& L = lds_all[wv];
    for (int k = tid - prnb::kTabShift; k < prnb::kKTab; k += kHeavyBlock) inv_k[k] = k > 0 ? 1.0f / (float)k : 0.0f;
    // The segments' fill, in chunks of 64: the exclusive prefix over the kSegs segments
    __shared__ uint32_t seg_n[kSegs], seg_first[kSegs + 1];
    if (wv == 0) {
        uint32_t n = heavy.seg_cnt[lane];
        n = n < heavy.ent_cap ? n : heavy.ent_cap;
        uint32_t x = (n + 63u) >> 6;                        // chunks of this segment
        const uint32_t mine = x;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)x, d, 64);
            if (lane >= d) x += t;
        }
        seg_n[lane] = n;
        seg_first[lane] = x - mine;
        if (lane == 63) seg_first[kSegs] = x;
    }
    __syncthreads();
    const int32_t groups = (strips + 3) / 4;
    const uint32_t waves = gridDim.x * (kHeavyBlock / 64);
    // chunk -> (segment, first entry): the last segment whose first chunk is not behind it (wave-uniform)
    auto locate = [&](uint32_t chunk, uint32_t& seg, uint32_t& first) __attribute__((always_inline)) {
        uint32_t lo = 0u;
#pragma unroll
        for (uint32_t step = kSegs / 2; step != 0u; step >>= 1)
            if (seg_first[lo + step] <= chunk) lo += step;
        seg = lo;
        first = (chunk - seg_first[lo]) << 6;
    };
    // blk / groups for blk < 2^32: the estimate by the rounded-down reciprocal is the quotient or one below it (an integer
    // division is ~30 instructions on this chip, and the kernel is bound by what it issues)
    const uint32_t groups_rcp = groups > 1 ? (uint32_t)(0x100000000ull / (uint64_t)(uint32_t)groups) : 0u;
    auto tile_of = [&](uint32_t blk) __attribute__((always_inline)) -> int32_t {
        if (groups <= 1) return (int32_t)blk;
        uint32_t q = __umulhi(blk, groups_rcp);
        q += (blk - q * (uint32_t)groups >= (uint32_t)groups) ? 1u : 0u;
        return (int32_t)q;
    };

    int hg_top = 0, hp_top = 0, hw_top = 0;      // wave-uniform

    auto cell_id = [&](int32_t n) __attribute__((always_inline)) -> uint64_t {
        return cell_index ? (uint64_t)cell_index[n] : cell_offset + (uint64_t)n;
    };

    // ---- one PTRS attempt (or the whole of the rare small/huge-lambda branches) ---------------
    // (every lambda of this kernel is inlined by force: left to its heuristics at -O2 the compiler keeps some as calls
    // and their captured state in scratch memory)
    auto poisson_pass = [&]() __attribute__((always_inline)) {
        const int cnt = hp_top < 64 ? hp_top : 64;
        bool again = false, small = false;
        HPEntry e;
        e.n = 0; e.g = 0; e.lam = 0.0f; e.attempt = 0;
        int32_t x = 0;
        if (lane < cnt) {
            e = L.hp[hp_top - 1 - lane];
            const uint64_t cell = cell_id(e.n);
            const uint32_t c0 = (uint32_t)cell, c1 = (uint32_t)(cell >> 32);
            const float lam = e.lam;
            if (!(lam > 0.0f)) {
                x = 0;
            } else if (lam < prnb::kPoisInv) {
                small = true;                      // walked below, all such lanes side by side
            } else if (!(lam < prnb::kLamBig)) {
                const prnb::Words w = prnb::philox_count<0>(c0, c1, (uint32_t)e.g, 0x80000000u, k0, k1);
                const float z = prnb::hw_normal(w.w[0], w.w[1]);
                const float kf = __builtin_floorf(PRNB_FMA(prnb::hw_sqrt(lam), z, lam) + 0.5f);
                x = (int32_t)__builtin_fminf(__builtin_fmaxf(kf, 0.0f), 2147483520.0f);
            } else {
                const int j = e.attempt;
                const prnb::Words w =
                    prnb::philox_count<0>(c0, c1, (uint32_t)e.g, 0x80000000u + (uint32_t)(j >> 1), k0, k1);
                const float slam = prnb::hw_sqrt(lam);
                const float bb = PRNB_FMA(2.53f, slam, 0.931f);
                const float aa = PRNB_FMA(0.02483f, bb, -0.059f);
                const float U = prnb::unif((j & 1) ? w.w[2] : w.w[0]) - 0.5f;
                const float V = prnb::unif((j & 1) ? w.w[3] : w.w[1]);
                const float us = __builtin_fmaxf(0.5f - __builtin_fabsf(U), 5.8207661e-11f);
                const float rus = prnb::hw_rcp(us);
                float kf = __builtin_floorf(PRNB_FMA(PRNB_FMA(2.0f * aa, rus, bb), U, lam + 0.43f));
                const float vr = PRNB_FMA(-3.6224f, prnb::hw_rcp(bb - 2.0f), 0.9277f);
                bool accept = (us >= 0.07f) && (V <= vr);
                if (!accept && !(kf < 0.0f || (us < 0.013f && V > us))) {
                    const float invalpha = PRNB_FMA(1.1328f, prnb::hw_rcp(bb - 3.4f), 1.1239f);
                    const float lhs = prnb::kLn2 * prnb::hw_log2((V * invalpha) * prnb::hw_rcp(PRNB_FMA(aa * rus, rus, bb)));
                    float rhs;
                    if (kf < 10.0f) {
                        rhs = PRNB_FMA(kf, prnb::kLn2 * prnb::hw_log2(lam), -lam) - prnb::logfact_small((int)kf);
                    } else {
                        const float rk = prnb::hw_rcp(kf);
                        const float d = (lam - kf) * rk;
                        const float lp = prnb::hw_log1pmx(d, lam * rk);
                        const float st = rk * PRNB_FMA(-0.0027777778f, rk * rk, 0.083333336f);
                        rhs = PRNB_FMA(kf, lp, PRNB_FMA(-0.5f, prnb::kLn2 * prnb::hw_log2(6.2831855f * kf), -st));
                    }
                    accept = lhs <= rhs;
                }
                if (!accept) {
                    kf = __builtin_floorf(lam);
                    again = j + 1 < 2 * prnb::kMaxTries;
                }
                x = (int32_t)__builtin_fminf(__builtin_fmaxf(kf, 0.0f), 2147483520.0f);
            }
        }
        if (__builtin_amdgcn_ballot_w64(small) != 0ull) {
            // lambda under 10: inversion (the chop-down of prnb_device.h with q = 0)
            const uint64_t cell = cell_id(e.n);
            const prnb::Words w = prnb::philox_count<0>((uint32_t)cell, (uint32_t)(cell >> 32), (uint32_t)e.g, 0x80000000u, k0, k1);
            const float lam = small ? e.lam : 1.0f;
            const int32_t xs = prnb::chop_down_wave(small, w.w[0], prnb::hw_exp2(-(lam * prnb::kLog2e)), lam, 0.0f, inv_k);
            if (small) x = xs;
        }
        if (lane < cnt && !again && x != 0) out[(int64_t)e.n * ld + e.g] = x;
        hp_top -= cnt;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(again);
        if (again) {
            e.attempt += 1;
            L.hp[hp_top + lane_rank(m)] = e;
        }
        hp_top += __popcll(m);
    };

    // ---- one Marsaglia-Tsang attempt ------------------------------------------------------------
    auto gamma_pass = [&]() __attribute__((always_inline)) {
        const int cnt = hg_top < 64 ? hg_top : 64;
        bool again = false, accepted = false;
        HGEntry e;
        e.n = 0; e.g = 0; e.attempt = 0; e.m = 0.0f;
        float lam = 0.0f;
        if (lane < cnt) {
            e = L.hg[hg_top - 1 - lane];
            const prnb::Params P = prnb::make_params_m(e.m, ga[e.g], gbm1[e.g]);
            const float r = P.m * P.inv_th;
            // (a flagged sample is valid: m > 0, theta > 0); r under 2^-40 is a count of 0 (P(X > 0) < 2^-32)
            if (P.valid && r >= prnb::kRMin) {
                const uint64_t cell = cell_id(e.n);
                const bool boost = r < 1.0f;
                const float rr = boost ? r + 1.0f : r;
                const float dd = rr - 0.33333334f;
                const float cc = prnb::hw_rsq(9.0f * dd);
                const int i = e.attempt;
                const bool last = (i == prnb::kMaxTries - 1);
                const prnb::Words w = prnb::philox_count<0>((uint32_t)cell, (uint32_t)(cell >> 32), (uint32_t)e.g,
                                                          1u + (uint32_t)i, k0, k1);
                const float x = prnb::hw_normal(w.w[0], w.w[1]);
                const float t = cc * x;
                const float v1 = 1.0f + t;
                float v = 1.0f;
                bool ok;
                if (!(v1 > 0.0f)) {
                    ok = last;
                } else {
                    v = (v1 * v1) * v1;
                    ok = last;
                    if (!ok) {
                        const float u = prnb::unif(w.w[2]);
                        const float x2 = x * x;
                        ok = u < PRNB_FMA(-0.0331f, x2 * x2, 1.0f);
                        if (!ok) {
                            const float t2 = t * t;
                            const float h = PRNB_FMA(3.0f, prnb::hw_log1pmx(t, v1), PRNB_FMA(-t2, t, -3.0f * t2));
                            ok = prnb::kLn2 * prnb::hw_log2(u) < PRNB_FMA(dd, h, 0.5f * x2);
                        }
                    }
                }
                if (ok) {
                    float g = dd * v;
                    if (boost) g = g * prnb::hw_exp2(prnb::hw_log2(prnb::unif(w.w[3])) * prnb::hw_rcp(r));
                    lam = P.theta * g;
                    accepted = true;
                } else {
                    again = true;
                }
            }
        }
        hg_top -= cnt;
        const unsigned long long ma = __builtin_amdgcn_ballot_w64(again);
        if (again) {
            e.attempt += 1;
            L.hg[hg_top + lane_rank(ma)] = e;
        }
        hg_top += __popcll(ma);
        const unsigned long long mp = __builtin_amdgcn_ballot_w64(accepted);
        if (accepted) {
            HPEntry p;
            p.n = e.n; p.g = e.g; p.lam = lam; p.attempt = 0;
            L.hp[hp_top + lane_rank(mp)] = p;
        }
        hp_top += __popcll(mp);
        while (hp_top >= 64) poisson_pass();
    };

    // ---- inversion walks: the ones the streaming kernel had not finished when their strip ended arrive with their
    // state and go on from it; only the samples of an overflowed region (and a fifth walk past term 252 of one strip) start at k = 0 (from_start).
    // Every lane walks its own pmf sixteen terms per pass; walks differ in length by two orders of magnitude, so a lane's
    // walk lives in registers across passes and an idle lane takes the next entry of the stack: a pass runs with more
    // than half of the lanes walking, and new walks start at least 32 at a time.
    int32_t wk = -1;                                  // next term of this lane's walk; -1: idle
    float wps = 0.0f, wrem = 0.0f, wd = 0.0f, wq = 0.0f;         // the next term, the remainder, mp - q, q
    int32_t wn = 0, wg = 0;
    auto walk_take = [&]() __attribute__((always_inline)) {
        const unsigned long long idle_m = __builtin_amdgcn_ballot_w64(wk < 0);
        const int rank = lane_rank(idle_m);
        if (wk < 0 && rank < hw_top) {
            const int at = hw_top - 1 - rank;
            const f32x4_t s = L.wst[at];
            const HWId id = L.wid[at];
            wk = (int32_t)L.wk[at];
            wps = s.x; wd = s.y; wq = s.z; wrem = s.w;
            wn = id.n; wg = id.g;
        }
        const int idle = __popcll(idle_m);
        hw_top -= idle < hw_top ? idle : hw_top;
    };
    auto walk_pass = [&]() __attribute__((always_inline)) {
        // FOUR groups of four terms for every walking lane (an idle lane computes on zeros): the longest walk of a launch is a
        // chain of passes one after the other -- its length is K3h's floor --, and a pass of 16 terms pays the LDS round trip
        // for the reciprocals, the tests and the loop once per 16 terms (8 terms per pass: the chain 17 us at C3).  Most
        // passes of the longest walks end nowhere: one wave-level test then skips everything but the arithmetic.
        const bool busy = wk >= 0;
        const float* tab = &inv_k[busy ? wk + 1 : 6];
        float4 iv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) iv[j] = *reinterpret_cast<const float4*>(__builtin_assume_aligned(tab + 4 * j, 16));
        // term t[i] is the pmf at k = wk + i (t[0] = wps), r[i] the remainder behind it
        float t = wps, r = wrem;
        float rg[4][4], tl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            rg[j][0] = r - t;
            t = t * PRNB_FMA(wd, iv[j].x, wq);
            rg[j][1] = rg[j][0] - t;
            t = t * PRNB_FMA(wd, iv[j].y, wq);
            rg[j][2] = rg[j][1] - t;
            t = t * PRNB_FMA(wd, iv[j].z, wq);
            rg[j][3] = rg[j][2] - t;
            tl[j] = t;                                   // the group's last term (k = wk + 4 j + 3)
            r = rg[j][3];
            t = t * PRNB_FMA(wd, iv[j].w, wq);           // the next group's first term
        }
        // a group ends the walk when its last remainder is negative (the remainders only fall: one of its four is negative iff
        // the last one is), when its last term is under 1, or when it is the walk's last group; the first such group decides
        bool end[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) end[j] = (rg[j][3] < 0.0f) || (tl[j] < 1.0f) || (wk + 4 * j + 3 >= prnb::kWalkEnd);
        const bool ended = busy && (end[0] || end[1] || end[2] || end[3]);
        if (__builtin_amdgcn_ballot_w64(ended) != 0ull) {
            // the count: the group's last k less one for each of its first three remainders that is negative
            int32_t at[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                at[j] = ((wk + 4 * j + 3) + ((int32_t)prnb::f2u(rg[j][0]) >> 31)) + (((int32_t)prnb::f2u(rg[j][1]) >> 31) + ((int32_t)prnb::f2u(rg[j][2]) >> 31));
            if (ended) {
                out[(int64_t)wn * ld + wg] = end[0] ? at[0] : (end[1] ? at[1] : (end[2] ? at[2] : at[3]));         // (>= 5)
                wk = -1;
                wps = 0.0f;
            }
        }
        if (wk >= 0) {
            wps = t;
            wk += 16;
            wrem = r;
        }
    };
    auto walk_service = [&](bool drain) __attribute__((always_inline)) {
        for (;;) {
            const int busy = __popcll(__builtin_amdgcn_ballot_w64(wk >= 0));
            if (busy <= 32 && (drain ? hw_top > 0 : hw_top >= 32)) walk_take();
            else if (drain ? busy > 0 : busy > 32) walk_pass();
            else break;
        }
    };
    // push walk states of the lanes in `push` (wave-level call)
    auto walk_push = [&](bool push, int32_t n, int32_t g, int32_t k, f32x4_t s) __attribute__((always_inline)) {
        const unsigned long long mw = __builtin_amdgcn_ballot_w64(push);
        if (push) {
            const int at = hw_top + lane_rank(mw);
            L.wst[at] = s;
            HWId id;
            id.n = n; id.g = g;
            L.wid[at] = id;
            L.wk[at] = (uint16_t)k;
        }
        hw_top += __popcll(mw);
        if (hw_top >= 32) walk_service(false);
    };

    // ---- a sample from its start (rare: any sample of an overflowed region, a fifth walk past term 252 of one strip): the mean is
    // gathered, the class decided; the gamma-Poisson class goes on HG, an inversion walk is taken through its first
    // group (k = 0 .. 4) and, when that does not decide it, pushed as a walk state at k = 5
    auto from_start = [&](bool has, int32_t n, int32_t g) __attribute__((always_inline)) {
        bool heavy_c = false, light = false;
        prnb::Params P;
        P.m = 1.0f; P.theta = 1.0f; P.iu = 0.5f; P.t2 = 1.0f; P.inv_th = 1.0f; P.valid = false; P.light = false;
        if (has) {
            int32_t row = row_of_cell[n];
            row = row < 0 ? 0 : (row >= rows ? (int32_t)(rows - 1) : row);     // as the preparation kernel: never a wild read
            P = prnb::make_params(means[(int64_t)row * G + g], scal[n], ga[g], gbm1[g]);
            light = P.valid && P.light;
            heavy_c = P.valid && !P.light;
        }
        const unsigned long long mh = __builtin_amdgcn_ballot_w64(heavy_c);
        if (heavy_c) {
            HGEntry e;
            e.n = n; e.g = g; e.attempt = 0; e.m = P.m;
            L.hg[hg_top + lane_rank(mh)] = e;
        }
        hg_top += __popcll(mh);
        bool push = false;
        f32x4_t s = {0.0f, 0.0f, 0.0f, 0.0f};
        if (__builtin_amdgcn_ballot_w64(light) != 0ull) {
            const uint64_t cell = cell_id(has ? n : 0);
            const prnb::Words w = prnb::philox_count<0>((uint32_t)cell, (uint32_t)(cell >> 32), (uint32_t)g >> 2, 0u, k0, k1);
            const uint32_t sel = (uint32_t)g & 3u;
            const uint32_t wj = sel == 0u ? w.w[0] : (sel == 1u ? w.w[1] : (sel == 2u ? w.w[2] : w.w[3]));
            const float mp = P.m * P.iu, q = P.theta * P.iu, d = mp - q;
            const float ps = prnb::hw_exp2(-P.t2) * 4294967296.0f;
            const float r0 = (float)wj - ps;
            const float p1 = ps * mp;
            const float r1 = r0 - p1;
            const float p2 = p1 * PRNB_FMA(d, inv_k[2], q);
            const float r2 = r1 - p2;
            const float p3 = p2 * PRNB_FMA(d, inv_k[3], q);
            const float r3 = r2 - p3;
            const float p4 = p3 * PRNB_FMA(d, inv_k[4], q);
            const float r4 = r3 - p4;
            const bool done = (r0 < 0.0f) || (r1 < 0.0f) || (r2 < 0.0f) || (r3 < 0.0f) || (r4 < 0.0f) || (p4 < 1.0f);
            const int32_t x = (r0 < 0.0f) ? 0 : ((r1 < 0.0f) ? 1 : ((r2 < 0.0f) ? 2 : ((r3 < 0.0f) ? 3 : 4)));
            if (light && done && x != 0) out[(int64_t)n * ld + g] = x;
            push = light && !done;
            s.x = p4 * PRNB_FMA(d, inv_k[5], q);
            s.y = d; s.z = q; s.w = r4;
        }
        walk_push(push, n, g, 5, s);
        while (hg_top >= 64) gamma_pass();
    };

    // ---- the list -------------------------------------------------------------------------------------
    const uint32_t wave_id = blockIdx.x * (kHeavyBlock / 64) + (uint32_t)wv;
    // Six blocks in sixteen (kWalkerBlocks) only walk, the others only draw: the walks differ in length by two orders of magnitude (the
    // longest of a launch runs twenty passes of sixteen terms, one after the other), so they start at once, on waves that
    // have nothing else to do, and a walker with a few hundred walks keeps its lanes busy.  (With every wave walking the
    // fifty walks of its own regions, a wave ran as many passes as its longest walk needs, mostly for
    // a handful of lanes.)
    const bool walker = (blockIdx.x & 15u) < kWalkerBlocks;
    const uint32_t walker_blocks = (gridDim.x >> 4) * kWalkerBlocks + ((gridDim.x & 15u) < kWalkerBlocks ? (gridDim.x & 15u) : kWalkerBlocks);
    const uint32_t n_redo = heavy.seg_cnt[kSegs + 1] < (uint32_t)kRedoCap ? heavy.seg_cnt[kSegs + 1] : (uint32_t)kRedoCap;
    auto push_entries = [&](bool has, int32_t n, int32_t g, float m) __attribute__((always_inline)) {
        const unsigned long long mh = __builtin_amdgcn_ballot_w64(has);
        if (has) {
            HGEntry h;
            h.n = n; h.g = g; h.attempt = 0; h.m = m;
            L.hg[hg_top + lane_rank(mh)] = h;
        }
        hg_top += __popcll(mh);
        while (hg_top >= 64) gamma_pass();
    };
    if (!walker) {
        const uint32_t blk_rank = blockIdx.x - ((blockIdx.x >> 4) * kWalkerBlocks + ((blockIdx.x & 15u) < kWalkerBlocks ? (blockIdx.x & 15u) : kWalkerBlocks));
        const uint32_t rank = blk_rank * (kHeavyBlock / 64) + (uint32_t)wv;       // among the drawing waves
        const uint32_t drawers = (gridDim.x - walker_blocks) * (kHeavyBlock / 64);
        // The listed samples (every one of the gamma-Poisson class: the streaming kernel decided that).  First the regions'
        // first kDense entries: four regions per wave and step, 16 lanes each, two entries per lane; a step's loads go out
        // one step ahead of their use.  A wave's regions lie `drawers` apart: the listed samples cluster in a few gene tiles
        // (regions are laid out tile by tile).
        {
            const int sub = lane >> 4, sl = lane & 15;
            const uint4* const dense4 = reinterpret_cast<const uint4*>(heavy.dense);
            struct Step { uint32_t cnt; uint4 d4; };
            auto load_step = [&](uint64_t r0) __attribute__((always_inline)) -> Step {
                Step S;
                S.cnt = 0u; S.d4 = make_uint4(0u, 0u, 0u, 0u);
                const uint64_t r = r0 + (uint64_t)sub * drawers;
                if (r < (uint64_t)regions) {
                    S.cnt = heavy.count[r] & 0xffffu;
                    S.d4 = dense4[r * (kDense / 2) + (uint32_t)sl];
                }
                return S;
            };
            Step nxt = load_step(rank);
            for (uint64_t r0 = rank; r0 < (uint64_t)regions; r0 += 4ull * drawers) {
                const uint32_t r = (uint32_t)(r0 + (uint64_t)sub * drawers);
                const Step cur = nxt;
                nxt = load_step(r0 + 4ull * drawers);
                const uint32_t cnt = cur.cnt > heavy.cap ? 0u : cur.cnt;       // (above the cap: the region is redone as a whole)
                const int32_t blk = (int32_t)(r >> 2);
                const int32_t tile_g = tile_of((uint32_t)blk);
                const int32_t n0 = ((blk - tile_g * groups) * 4 + (int32_t)(r & 3u)) * strip_cells;
                push_entries(2u * (uint32_t)sl < cnt, n0 + (int32_t)(cur.d4.x >> 8), tile_g * kTileG + (int32_t)(cur.d4.x & 255u), __uint_as_float(cur.d4.y));
                push_entries(2u * (uint32_t)sl + 1u < cnt, n0 + (int32_t)(cur.d4.z >> 8), tile_g * kTileG + (int32_t)(cur.d4.z & 255u), __uint_as_float(cur.d4.w));
            }
        }
        // ... then what the regions of a hot gene tile listed beyond that: chunk c of the segments goes to drawing wave c mod
        // (drawing waves) -- 64 entries per coalesced load, the next chunk's entries requested before this chunk's are looked at.
        {
            const uint32_t chunks = seg_first[kSegs];
            HeavyEntry nx;
            nx.n = 0; nx.g = 0; nx.m = 0.0f; nx.pad = 0u;
            bool nx_has = false;
            auto request = [&](uint32_t chunk) __attribute__((always_inline)) {
                uint32_t sg, first;
                locate(chunk, sg, first);
                nx_has = first + (uint32_t)lane < seg_n[sg];
                if (nx_has) nx = heavy.ent[(uint64_t)sg * heavy.ent_cap + first + (uint32_t)lane];
            };
            if (rank < chunks) request(rank);
            for (uint32_t chunk = rank; chunk < chunks; chunk += drawers) {
                const HeavyEntry e = nx;
                const bool has = nx_has;
                if (chunk + drawers < chunks) request(chunk + drawers);
                push_entries(has, e.n, e.g, e.m);
            }
        }
    } else {
        // The walks: first the ones to redo from their start (a strip kept more than kLongSlots walks past term 252: rare),
        // then the regions' walk states: a walker takes a run of neighbouring regions, 16 per step, four lanes each.
        const uint32_t wrank = ((blockIdx.x >> 4) * kWalkerBlocks + (blockIdx.x & 15u)) * (kHeavyBlock / 64) + (uint32_t)wv;
        const uint32_t walkers = walker_blocks * (kHeavyBlock / 64);
        for (uint32_t i = wrank * 64u; i < n_redo; i += walkers * 64u) {
            const bool has = i + (uint32_t)lane < n_redo;
            const int2 ng = has ? heavy.redo[i + (uint32_t)lane] : make_int2(0, 0);
            from_start(has, ng.x, ng.y);
        }
        const uint32_t per_walker = ((regions + walkers - 1u) / walkers + 15u) & ~15u;
        const uint32_t r_end = (wrank + 1u) * per_walker < regions ? (wrank + 1u) * per_walker : regions;
        const uint32_t rl = (uint32_t)lane >> 2, s0 = (uint32_t)lane & 3u;
        static_assert(kWalkSlots <= 12, "three slots per lane");
        struct WStep { uint32_t n_walks; f32x4_t st[3]; uint32_t id[3]; };
        auto load_wstep = [&](uint32_t rb) __attribute__((always_inline)) -> WStep {
            WStep S;
            S.n_walks = 0u;
            const uint32_t r = rb + rl;
#pragma unroll
            for (int j = 0; j < 3; ++j) { S.st[j] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f}; S.id[j] = 0u; }
            if (r < r_end) {
                S.n_walks = heavy.count[r] >> 16;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const uint32_t slot = s0 + 4u * (uint32_t)j;
                    if (slot < (uint32_t)kWalkSlots) {
                        S.st[j] = heavy.wst[(uint64_t)r * kWalkSlots + slot];
                        S.id[j] = heavy.wid[(uint64_t)r * kWalkSlots + slot];
                    }
                }
            }
            return S;
        };
        WStep wn = load_wstep(wrank * per_walker);
        for (uint32_t rb = wrank * per_walker; rb < r_end; rb += 16u) {
            const WStep cur = wn;
            wn = load_wstep(rb + 16u);
            const uint32_t r = rb + rl;
            const int32_t blk = (int32_t)(r >> 2);
            const int32_t tile_g = tile_of((uint32_t)blk);
            const int32_t n0 = ((blk - tile_g * groups) * 4 + (int32_t)(r & 3u)) * strip_cells;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const uint32_t slot = s0 + 4u * (uint32_t)j;
                walk_push(slot < cur.n_walks, n0 + (int32_t)((cur.id[j] & 0xffffu) >> 8), tile_g * kTileG + (int32_t)(cur.id[j] & 255u),
                          (int32_t)(cur.id[j] >> 16) - 3, cur.st[j]);
            }
        }
    }
    // Regions whose list or segment was too small: every sample of the region (strip_cells x 256 of the matrix) goes
    // through the classification here, 64 at a time: slow (the streaming kernel's own results are recomputed), but any
    // parameter set stays correct and only those regions pay
    {
        const uint32_t n_ovf = heavy.seg_cnt[kSegs];
        for (uint32_t i = wave_id; i < n_ovf; i += waves) {
            const uint32_t r = heavy.ovf_regions[i];
            const int32_t blk = (int32_t)(r >> 2);
            const int32_t tg = tile_of((uint32_t)blk);
            const int64_t nb = (int64_t)((blk - tg * groups) * 4 + (int32_t)(r & 3u)) * strip_cells;
            for (int32_t c = 0; c < strip_cells && nb + c < N; ++c)
                for (int j = 0; j < 4; ++j) {
                    const int32_t g = tg * kTileG + j * 64 + lane;
                    from_start(g < G, (int32_t)(nb + c), g);
                }
        }
    }
    walk_service(true);
    while (hg_top > 0) gamma_pass();
    while (hp_top > 0) poisson_pass();

    // The rest of the reference's argument check (scipy behind simulation.py:647-648), for the calls that ask for it
    // (check_words: the ctx's flag words; NULL: an unchecked call, or the caller has verified alpha >= 0 and beta >= 1).
    // The preparation kernel has done the per-cell and per-gene parts and raised check_words[check_request] iff some gene
    // has alpha < 0 or beta < 1: only then can alpha*m + beta < 1 happen with every mean positive, and only then is the
    // whole matrix looked at here -- m = M*s <= 0 (or NaN) or alpha*m + beta - 1 < 0 anywhere sets the verdict word.  It
    // rides in this kernel so that a checked call launches nothing more than an unchecked one.  (A block takes whole
    // cells and walks their genes: no division per sample.)
    if (check_words && check_words[check_request] != 0 && check_words[check_bad_row] == 0) {
        bool bad = false;
        for (int64_t n = blockIdx.x; n < N; n += gridDim.x) {
            const float s = scal[n];
            const float* mrow = means + (int64_t)row_of_cell[n] * G;
            for (int32_t g = tid; g < G; g += kHeavyBlock) {
                const float m = mrow[g] * s;
                bad = bad || !(m > 0.0f) || (PRNB_FMA(ga[g], m, gbm1[g]) < 0.0f);
            }
        }
        if (bad) check_words[check_verdict] = 1;
    }
}

}  // namespace k3
