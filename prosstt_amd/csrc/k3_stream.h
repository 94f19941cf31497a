// K3, streaming form: the fused count sampler as a three-stage pipeline run by each wave
// on its own strip of the count matrix, with LDS stacks between the stages so that every
// stage executes with (nearly) all 64 lanes busy.
//
// The scalar algorithm (PRNB-4, prnb_device.h) has very different costs per sample:
//   ~65 % of the samples of the headline workload are zeros that a 6-instruction bound
//         settles (exp(-m * phi_gene) <= P(X = 0), prnb::zero_test_factor);
//   the rest need P(X = 0) exactly (reciprocal + log1p + exp, ~100 VALU), and
//   ~33 % then walk the pmf for k >= 1 (data-dependent length, half of them end at k <= 2);
//   ~0.1 % need gamma-Poisson.
// Run lane-per-sample, every wave pays for its slowest lane in every one of these.  Here
//   stage 1 (uniform)   one count-matrix row segment per wave pass: 16-B mean load, one
//                       Philox call per lane, the bound test as a compare mask; survivors are
//                       pushed on stack S1 under exec = mask;
//   stage 2 (64 of S1)  P(X = 0) and the class test, then the terms k = 0, 1, 2; what is still
//                       undecided is pushed on S2 with the pmf state at k = 3;
//   stage 3 (lanes pull from S2)  four pmf steps per lane per pass; a lane that finishes
//                       writes its count into the LDS row ring and pulls the next entry;
//   output              the last kRing rows of the strip live in LDS, 8 bits per count (a count above
//                       255 -- 1 in 10^4 -- is left to K3h); a row
//                       is stored (one coalesced 1 KiB store per wave) kRing cells after stage 1
//                       started it.  The few counts that arrive later than that (long walks)
//                       are collected in LDS and written in bursts of 4-B stores -- a store per
//                       late count would sit in front of every wait for the next mean load
//                       (loads and stores retire in order on one counter);
//   samples of the gamma-Poisson class are only LISTED here (a lane that meets one keeps it in
//   a register; when a lane meets its second, and at the end of the strip, the wave writes what
//   its lanes hold to its own region of a global list -- no atomics: one counter for all waves
//   serialises them); sample_counts_heavy_kernel (k3_heavy.h) draws them afterwards.
//
// Stages 2 and 3 evaluate P(X = 0) with the hardware's v_rcp/v_log/v_exp (each within 1.2e-7 of
// the true value on gfx950, tools/hwmath_probe.hip) instead of PRNB-4's deterministic binary32
// arithmetic -- a third of the instructions.  The result must still be the model's, bit for bit:
// a walk's answer is the first k whose subtraction leaves the remainder negative, so it can
// only differ from the exact evaluation's when a remainder lies within the two evaluations' distance
// of 0.  Every lane therefore tracks how close its remainders came to 0 and, when that is
// within a margin of about three times the worst-case distance (2^-20 + t*2^-19 + k*2^-21 of 2^32,
// t = -log2 P0), gives the sample up: it goes on the same list, and K3h redoes it with the
// exact arithmetic (about 1 in 10^3 of the walks).  So does a sample whose class the
// approximate t cannot decide, and a walk whose end-of-pmf test (a term against 1) is too close to call.
// Results are pure functions of (sample parameters, seed, global cell id, gene), so the
// order in which the stacks are drained cannot change them.
//
// Written against the issue model measured on gfx950 (tools/microbench5.hip, microbench6.hip;
// DESIGN.md section 6): a SIMD issues one vector instruction per ~2.4 cycles; binary32
// mul/add/sub/fma run beside everything else, the other kinds (compare, select, convert, min/max,
// left shift, mbcnt, 3-operand integer forms, 64-bit multiply) occupy a second unit for ~4.3 cycles
// each (simple integer add/xor/and/or/right shift ~2.4 there, transcendentals ~8.3), and scalar
// instructions issue beside both.  The kernel is bound by that second unit, so the walk is pure
// binary32 arithmetic on a binary32 remainder (PRNB-4), hits are counted from sign bits, wave-level
// tests are lane masks formed by ONE compare each, and per-cell values arrive by scalar loads.
#pragma once
#include "prnb_device.h"

namespace k3 {

constexpr int kBlock = 256;        // 4 waves
constexpr int kTileG = 256;        // genes per wave pass: 64 lanes x 4
constexpr int kStripCells = 128;   // most cells per wave (pos keeps the cell in 7 bits); the host launches 64
constexpr int kS1Cap = 320;        // < 64 left over + 256 pushed by one pass
constexpr int kS2Cap = 96;         // < 32 left over + 64 pushed by one stage-2 pass
constexpr int kS2Run = 32;         // stage 3 runs while S2 holds at least this many entries (8, 16: 3-5 % slower)
static_assert(kS2Cap >= kS2Run - 1 + 64 && kS1Cap >= 63 + 256, "a stack must take one more pass of pushes");
constexpr int kRing = 8;           // rows of the strip kept in LDS (8 bits per count) before they are stored; a power of 2
// Threshold margins of the hardware-math evaluation, in units of 2^-32 (see the header): about
// three times the worst-case distance between the two evaluations of a running pmf sum C_k --
//   P0:  |t_exact - t_hw| <= (6.1e-7 + 3.5e-7) * t  (det_log1p 2.5e-7, det_rcp 1.2e-7, four roundings |
//        log2(u1)*rcp(u1-1) 2.9e-7 measured over (0, 16], one rounding), exp 2.0e-7 + 0.9e-7;
//   ratios: none (mp - q and q are formed by PRNB-4's own arithmetic here, so the ratio q + (mp - q)/(k+1) of a term
//        is the same number in both paths): a term adds only the rounding difference of its one multiplication,
//        2^-24 at most.
//   the remainder itself: both paths subtract their terms from a binary32 remainder below 2^32, so
//        each subtraction can round differently by up to ulp(2^32)/2 = 256 units -- far less for the
//        small remainders of a walk that is about to end, but the margin does not rely on that.
constexpr float kMargin0 = 4096.0f;          // 2^-20        (worst case 2.9e-7 = 1245 units, + 256)
constexpr float kMarginPerT2 = 8192.0f;      // 2^-19 per unit of t2 = t / ln 2   (worst case 9.6e-7 * ln 2 = 2858 units)
constexpr float kMarginPerTerm = 2048.0f;    // 2^-21 per term                   (worst case 1.2e-7 = 515 units, + 256)
constexpr float kTailBand = 9.765625e-4f;    // a group's last term within 2^-10 of 1: the end-of-pmf test is K3h's
constexpr float kT2Sure = 27.41120f * (1.0f - 1.53e-5f);   // 19 / ln 2, less 2^-16: surely t <= 19
constexpr int kBail = 6;           // walks left to K3h when a strip has nothing else to do (see the drain)
constexpr int kLateCap = 128;      // results that missed their row wait here for one burst of stores (< 64 left + 64)

// What stage 1 needs to know about a cell, packed by the preparation kernel so that one scalar load
// fetches it: byte offset of the cell's row in the mean tensor, library-size factor, (cell index
// within its strip) << 8, and the cell's share of the Philox call -- the counter is (cell_lo, cell_hi,
// gene quad, 0), so the halves of rounds 1 and 2 that do not depend on the gene are the same for a whole
// row of the count matrix: ph[0] = cell_hi ^ k0, ph[1] = hi(M1 * (hi(M0 * cell_lo) ^ k1)) ^ (k0 + W0),
// ph[2] = lo(M0 * cell_lo) ^ (k1 + W1), ph[3] = lo(M1 * (hi(M0 * cell_lo) ^ k1))  (philox_cell_part).
// The array holds N + 4 entries (the last cell repeated) so that prefetches need no clamp.
struct CellInfo { uint64_t row_bytes; float s; uint32_t pos_base; uint32_t ph[4]; };
static_assert(sizeof(CellInfo) == 32, "one s_load_dwordx8");

constexpr uint32_t kPhiloxM0 = 0xD2511F53u, kPhiloxM1 = 0xCD9E8D57u, kPhiloxW0 = 0x9E3779B9u, kPhiloxW1 = 0xBB67AE85u;

// the cell-only part of Philox4x32 on counter (cell_lo, cell_hi, x, 0), key (k0, k1): see CellInfo
__host__ __device__ inline void philox_cell_part(uint32_t cell_lo, uint32_t cell_hi, uint32_t k0, uint32_t k1, uint32_t ph[4])
{
    const uint64_t p0 = (uint64_t)kPhiloxM0 * cell_lo;                       // round 1, first product
    const uint64_t p1 = (uint64_t)kPhiloxM1 * ((uint32_t)(p0 >> 32) ^ k1);   // round 2, second product: c2' = hi(p0) ^ c3 ^ k1, c3 = 0
    ph[0] = cell_hi ^ k0;
    ph[1] = (uint32_t)(p1 >> 32) ^ (k0 + kPhiloxW0);
    ph[2] = (uint32_t)p0 ^ (k1 + kPhiloxW1);
    ph[3] = (uint32_t)p1;
}

// Philox4x32-7 of stage 1: rounds 1 and 2 from the cell's part (scalar registers) and the gene quad's
// part (hi1, lo1 = the halves of M1 * quad, constant per lane over a strip), rounds 3..7 as usual.
// Same words as prnb::philox_count(cell_lo, cell_hi, quad, 0, k0, k1).
__device__ __forceinline__ prnb::Words philox_count_row(const uint32_t ph[4], uint32_t hi1, uint32_t lo1, uint32_t k0, uint32_t k1)
{
    static_assert(prnb::kCountRounds >= 3, "two rounds are spelled out here");
    // round 1: c0' = hi1 ^ cell_hi ^ k0, c1' = lo1, c2' = hi(M0*cell_lo) ^ k1 (scalar), c3' = lo(M0*cell_lo) (scalar)
    const uint32_t c0a = hi1 ^ ph[0];
    // round 2: p0 = M0 * c0', p1 = M1 * c2' (scalar, in ph)
    const uint64_t p0 = (uint64_t)kPhiloxM0 * c0a;
    const uint32_t c0b = lo1 ^ ph[1];                 // hi(p1) ^ c1' ^ (k0 + W0)
    const uint32_t c1b = ph[3];                       // lo(p1)
    const uint32_t c2b = (uint32_t)(p0 >> 32) ^ ph[2];  // hi(p0) ^ c3' ^ (k1 + W1)
    const uint32_t c3b = (uint32_t)p0;
    return prnb::philox4x32<prnb::kCountRounds - 2, 0>(c0b, c1b, c2b, c3b, k0 + 2u * kPhiloxW0, k1 + 2u * kPhiloxW1);
}

struct S1Entry { float m, theta, wf; uint32_t pos; };  // theta = a*m + b - 1, not yet clamped; wf = (float)(32-bit uniform)
struct S2Entry { float ps, mp, q, rem; };             // pmf (x 2^32) at k = 3 and what is left of wf; `mp` holds mp - q: the ratio of step k is q + (mp - q)/(k+1)
// S2 meta word: the threshold margin of the terms k = 3..6 (binary32, rounded up to 16 significant bits
// -- it is a bound, and the model knows nothing of it) with pos in the 16 bits that frees

struct WaveLds {
    S1Entry s1_null;               // m = 0: what a lane of stage 2 reads when the stack holds fewer than 64 entries
    S1Entry s1[kS1Cap];
    S2Entry s2[kS2Cap];
    uint32_t s2m[kS2Cap];
    uint8_t ring[kRing * 256];     // [row slot][gene-in-tile]; a count of 256 or more is left to K3h
    uint32_t late[kLateCap];       // (pos << 16) | count of results whose row has left the ring already
};

// The list of samples left to K3h: wave w of block b owns region r = 4*b + w, entries
// list[r * cap .. + min(count[r], cap)) = pos (cell-in-strip << 8 | gene-in-tile); a region that was too
// small (count[r] > cap: more than one sample in 16 listed) is redone by K3h sample by sample, and
// overflow[0] != 0 says that there was one.
static_assert(offsetof(WaveLds, s1) == offsetof(WaveLds, s1_null) + sizeof(S1Entry), "s1[-1] must be the null entry");

struct HeavyList { uint32_t* count; uint32_t* list; uint32_t* overflow; uint32_t cap; };

__device__ __forceinline__ uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

// rank of this lane among the lanes whose bit is set in `mask`
__device__ __forceinline__ int lane_rank(unsigned long long mask)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                          __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

template <bool VEC>
__global__ __launch_bounds__(kBlock, 4) void sample_counts_stream_kernel(
    const float* __restrict__ means, int32_t G, const CellInfo* __restrict__ cellinfo,
    const float* __restrict__ ga, const float* __restrict__ gbm1, const float* __restrict__ gphi,
    int64_t N, uint32_t k0, uint32_t k1, int32_t* __restrict__ out, int64_t ld, int32_t strips,
    int32_t strip_cells, HeavyList heavy)
{
    // 1/k for k = -4 .. KTAB+7: 0 below k = 1 (an idle stage-3 lane reads there) and from the sentinel (k = KTAB-1) on
    __shared__ __attribute__((aligned(16))) float inv_k_store[4 + prnb::kKTab + 8];
    float* const inv_k = inv_k_store + 4;
    __shared__ WaveLds lds_all[kBlock / 64];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    WaveLds& L = lds_all[wv];

    for (int k = tid - 4; k < prnb::kKTab + 8; k += kBlock) inv_k[k] = (k > 0 && k < prnb::kKTab - 1) ? 1.0f / (float)k : 0.0f;
    __syncthreads();

    // block -> (gene tile, group of 4 strips); adjacent blocks share the gene tile (L2 reuse of
    // the mean tensor's column slice and of the per-gene parameters)
    const int32_t groups = (strips + 3) / 4;
    const int32_t tile_g = blockIdx.x / groups;
    const int32_t strip = (blockIdx.x - tile_g * groups) * 4 + wv;
    const int32_t gbase = tile_g * kTileG;
    const int32_t g0 = gbase + lane * 4;
    const int64_t n0 = (int64_t)strip * strip_cells;
    const uint32_t region = blockIdx.x * 4u + (uint32_t)wv;
    if (strip >= strips || n0 >= N) {                // whole wave leaves together (no barrier below)
        if (lane == 0) heavy.count[region] = 0u;
        return;
    }
    const int cells = (int)((N - n0 < strip_cells) ? (N - n0) : strip_cells);

    // lanes beyond G read some valid mean (see load_seg); a = b - 1 = 0 makes theta = 0, which
    // stage 2 drops before the class test
    float a[4], bm1[4], phi[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool in = g0 + j < G;
        a[j] = in ? ga[g0 + j] : 0.0f;
        bm1[j] = in ? gbm1[g0 + j] : 0.0f;
        phi[j] = in ? gphi[g0 + j] : 1.0f;
    }

    int s1_top = 0, s2_top = 0;                      // wave-uniform
    constexpr uint32_t kNoHeavy = 0xffffffffu;
    uint32_t hpend = kNoHeavy;                       // pos of the sample this lane holds for the next append to K3h's list
    // stage-3 lane state
    float ps = 0.0f, mp = 0.0f, q = 0.0f;              // of a busy lane: the next term, mp - q, q
    float rem = 0.0f, dl = 0.0f;                     // what is left of wf; this lane's threshold margin (grows with k)
    uint32_t pos = 0u;
    constexpr int kIdle = -5;        // k + 1 = 0 mod 4 (the aligned read of four reciprocals), k + 3 < 0 (no result)
    int k = kIdle;
    float4 inv = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // 1/(k+1) .. 1/(k+4): read one pass ahead

    // S1/S2 entries carry pos = (cell-in-strip << 8) | gene-in-tile
    int32_t* const strip_out = out + n0 * ld + gbase;
    const uint32_t ld32 = (uint32_t)ld;              // strip rows * ld < 2^32, checked by the host
    // pos of the last sample whose row has already left the ring (wave-uniform; -1: none)
    int32_t flushed_pos = -1;
    for (int i = lane; i < kRing * 64; i += 64) reinterpret_cast<uint32_t*>(L.ring)[i] = 0u;
    if (lane < 4) reinterpret_cast<uint32_t*>(&L.s1_null)[lane] = 0u;

    // store row `cl` of the strip from ring slot cl % kRing and clear the slot
    auto flush_row = [&](int cl, int32_t* row_ptr) {
        uint32_t* slot = reinterpret_cast<uint32_t*>(L.ring + (cl & (kRing - 1)) * 256) + lane;
        const uint32_t packed = *slot;
        *slot = 0u;
        const int32_t v[4] = {(int32_t)(packed & 0xffu), (int32_t)((packed >> 8) & 0xffu),
                              (int32_t)((packed >> 16) & 0xffu), (int32_t)(packed >> 24)};
        int32_t* dst = row_ptr + lane * 4;
        if (g0 < G) {
            if (VEC) {
                // non-temporal: the matrix is written once; keeping it out of L2 leaves that to the mean tensor (-3.8 % on C3)
                typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
                const i32x4 row4 = {v[0], v[1], v[2], v[3]};
                __builtin_nontemporal_store(row4, reinterpret_cast<i32x4*>(dst));
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (g0 + j < G) dst[j] = v[j];
            }
        }
        flushed_pos = (cl << 8) | 255;
    };

    // a finished count (> 0) goes into the row ring while its row is still there, else (rare) on
    // the late list; a burst of 4-B stores empties the list when it is full and when the strip ends,
    // always after the rows' own stores.  That a late 4-B store lands on top of its row's 16-B store rests
    // on the gfx9 rule that a wave's global_* operations are performed in issue order (it is what makes
    // vmcnt a counter: MI355X_MICROARCH.md, "s_waitcnt vmcnt"; flat_* would not be) -- tests/test_kernel_isa.py
    // checks that this kernel has no flat_ instruction, the whole-matrix tests that no late count is lost.
    // Every lane of the wave calls this (res = 0: nothing to deliver).
    int late_top = 0;                                // wave-uniform
    auto flush_late = [&]() {
        for (int i = lane; i < late_top; i += 64) {
            const uint32_t e = L.late[i];
            const uint32_t p = e >> 16;
            strip_out[(p >> 8) * ld32 + (p & 255u)] = (int32_t)(e & 0xffffu);
        }
        late_top = 0;
    };
    // (Wave-level tests are written on 64-bit lane masks, each the SGPR result of ONE compare taken
    // outside divergent control flow; masks are combined on the scalar unit and applied as exec.)
#define K3_MASK(x) __builtin_amdgcn_ballot_w64(x)
    const uint32_t ring_lds = (uint32_t)(uintptr_t)&L.ring[0];
    const uint32_t late_lds = (uint32_t)(uintptr_t)&L.late[0];
    // write the samples the lanes hold for K3h to this wave's region of the list (about ten entries
    // each time on the headline workload: the first lane to meet its second sample triggers it)
    uint32_t* const my_list = heavy.list + (uint64_t)region * heavy.cap;
    uint32_t h_cnt = 0u;                              // wave-uniform
    auto flush_heavy = [&]() {
        const unsigned long long mp_ = K3_MASK(hpend != kNoHeavy);
        if (hpend != kNoHeavy) {
            const uint32_t slot = h_cnt + (uint32_t)lane_rank(mp_);
            if (slot < heavy.cap) my_list[slot] = hpend;
        }
        h_cnt += (uint32_t)__popcll(mp_);
        hpend = kNoHeavy;
    };
    // Every lane of the wave calls this at the end of a stage-2 or stage-3 pass.  The lanes of `ok_m`
    // deliver count `res` (1..255) of sample `p`: into the row ring while the row is still there, else
    // (rare) onto the late list; the lanes of `give_m` (rare as well) leave sample `p` to K3h.  One test
    // covers both rare paths.
    auto deliver = [&](unsigned long long ok_m, unsigned long long give_m, uint32_t p, uint32_t res) {
        const unsigned long long late_m = K3_MASK((int32_t)p <= flushed_pos);
        // slot = cell % kRing, gene-in-tile
        asm volatile("s_mov_b64 exec, %0\n\tds_write_b8 %1, %2\n\ts_mov_b64 exec, -1"
                     :: "s"(ok_m & ~late_m), "v"(ring_lds + (p & (uint32_t)(kRing * 256 - 1))), "v"(res) : "memory");
        const unsigned long long ml = ok_m & late_m;
        if ((ml | give_m) != 0ull) {
            if (ml != 0ull) {
                const int cnt = __popcll(ml);
                if (late_top + cnt > kLateCap) flush_late();
                asm volatile("s_mov_b64 exec, %0\n\tds_write_b32 %1, %2\n\ts_mov_b64 exec, -1"
                             :: "s"(ml), "v"(late_lds + (uint32_t)((late_top + lane_rank(ml)) << 2)), "v"((p << 16) | res) : "memory");
                late_top += cnt;
            }
            if (give_m != 0ull) {
                if ((give_m & K3_MASK(hpend != kNoHeavy)) != 0ull) flush_heavy();
                asm volatile("s_mov_b64 exec, %1\n\tv_mov_b32 %0, %2\n\ts_mov_b64 exec, -1" : "+v"(hpend) : "s"(give_m), "v"(p));
            }
        }
    };
    // ---- stage 3: four pmf steps for every busy lane; idle lanes pull from S2 ------------------
    // A lane is idle iff k == kIdle, and an idle lane keeps ps = 0: its "count" k + 3 is negative,
    // which keeps it out of every mask below, so the arithmetic never asks which lanes are busy.
    // PRNB-4's walk (the chop-down of prnb_device.h): the terms are subtracted from a binary32 remainder, the count
    // is the first k whose subtraction leaves it negative; when a group of four ends without that
    // and its last term is under 1 (the pmf has fallen under 2^-32) the count is the group's last k.
    // A walk enters at k = 3 and advances by 4: the four reciprocals 1/(k+1)..1/(k+4) are one aligned
    // 16-byte LDS read.  The remainders only fall, so there is a hit iff the last one is negative, and
    // the hit is at term 4 - (number of negative remainders): sign bits, no compares.
    // Margin (see the header): a remainder within dl of 0, or the last term within 2^-10 of 1, may come
    // out differently in the exact arithmetic: the lane stops and leaves its sample to K3h.
    auto stage3_pass = [&]() {
        unsigned long long idle_m;
        asm("v_cmp_eq_u32 %0, -5, %1" : "=s"(idle_m) : "v"(k));
        static_assert(kIdle == -5, "the inline constant above");
        if (idle_m != 0ull && s2_top > 0) {
            const int rank = lane_rank(idle_m);
            if (k == kIdle && rank < s2_top) {
                const int idx = s2_top - 1 - rank;
                const S2Entry e = L.s2[idx];
                ps = e.ps; mp = e.mp; q = e.q; rem = e.rem;
                const uint32_t md = L.s2m[idx];
                pos = md & 0xffffu;
                dl = prnb::u2f(md);                   // pos rides in the low bits: the margin only grows by it
                k = 3;
                inv = *reinterpret_cast<const float4*>(&inv_k[4]);
            }
            const int left = s2_top - __popcll(idle_m);
            s2_top = left > 0 ? left : 0;
        }
        const float r1 = rem - ps;
        const float ps1 = ps * PRNB_FMA(mp, inv.x, q);          // (`mp` holds mp - q here: PRNB-4's ratio q + (mp - q)/(k+1))
        const float r2 = r1 - ps1;
        const float ps2 = ps1 * PRNB_FMA(mp, inv.y, q);
        const float r3 = r2 - ps2;
        const float ps3 = ps2 * PRNB_FMA(mp, inv.z, q);
        const float r4 = r3 - ps3;
        const unsigned long long hit_m = K3_MASK(r4 < 0.0f);
        const unsigned long long tail_m = K3_MASK(ps3 < 1.0f);
        const float near = __builtin_fminf(__builtin_fminf(__builtin_fminf(__builtin_fabsf(r1), __builtin_fabsf(r2)),
                                                           __builtin_fabsf(r3)), __builtin_fabsf(r4));
        const unsigned long long close_m = K3_MASK(near < dl) | K3_MASK(__builtin_fabsf(ps3 - 1.0f) < kTailBand);
        const uint32_t nneg = (prnb::f2u(r1) >> 31) + (prnb::f2u(r2) >> 31) + (prnb::f2u(r3) >> 31) + (prnb::f2u(r4) >> 31);
        const int k4 = k + 4;
        // a hit at term 4 - nneg: k + 4 - nneg; no hit: the group's last k = k + 3 (taken when the tail test ends the walk)
        const int32_t res_k = k4 - (int32_t)(nneg > 1u ? nneg : 1u);
        const unsigned long long busy_m = K3_MASK(res_k > 0);          // an idle lane's is -2
        const unsigned long long big_m = K3_MASK(res_k > 255);         // does not fit the ring's 8 bits (1 in 10^4): K3h's as well
        const unsigned long long done_m = hit_m | tail_m | close_m;    // (idle lanes: their ps3 = 0 is under 1)
        const unsigned long long give_m = (close_m | (big_m & (hit_m | tail_m))) & busy_m;
        deliver(done_m & busy_m & ~close_m & ~big_m, give_m, pos, (uint32_t)res_k);
        rem = r4;
        const float ps4 = ps3 * PRNB_FMA(mp, inv.w, q);
        dl = dl + 4.0f * kMarginPerTerm;
        // done lanes go idle (ps = 0, k = kIdle)
        asm("v_cndmask_b32 %0, %1, 0, %2" : "=v"(ps) : "v"(ps4), "s"(done_m));
        asm("v_cndmask_b32 %0, %1, -5, %2" : "=v"(k) : "v"(k4), "s"(done_m));
        inv = *reinterpret_cast<const float4*>(__builtin_assume_aligned(&inv_k[k + 1], 16));   // k + 1 = 0 mod 4
    };

    // ---- stage 2: P(X = 0), class test, then the terms k = 0, 1, 2, for up to 64 entries of S1 -------
    // prnb::make_params with the hardware's log2, reciprocal and exp2 for P(X = 0):
    // log1p(theta)/theta = log(u1)/(u1 - 1) (u1 = fl(1 + theta): the rounding of the sum cancels),
    // P0 = 2^-t2; mp and q by PRNB-4's own arithmetic (they multiply
    // into every term of a walk).  What the approximation cannot decide -- the class of a sample with t within
    // 2^-16 of 19, a remainder within the margin -- goes to K3h's list.
    const uint32_t s2_lds = (uint32_t)(uintptr_t)&L.s2[0];
    const uint32_t s2m_lds = (uint32_t)(uintptr_t)&L.s2m[0];
    auto stage2_pass = [&]() {
        // Straight-line for every lane (a lane beyond the entries reads a null entry, which is invalid):
        // every wave-level test below is a lane mask formed outside divergent control flow.
        const int cnt = s1_top < 64 ? s1_top : 64;
        const int at = s1_top - 1 - lane;
        const S1Entry e = (&L.s1[0])[at >= 0 ? at : -1];      // s1[-1] is s1_null: invalid, so no mask of the lanes that hold an entry
        const uint32_t p2 = e.pos;
        // (m <= 0 or theta <= 0: the count is 0 by definition; stage 1 does not test that)
        const unsigned long long valid_m = K3_MASK(e.m > 0.0f) & K3_MASK(e.theta > 0.0f);
        float theta, thetaq;                          // (plain v_max_f32: the builtin puts a canonicalising copy in front)
        asm("v_max_f32 %0, 0x21800000, %1" : "=v"(theta) : "v"(e.theta));      // prnb::kThetaMin = 2^-60
        asm("v_max_f32 %0, 0x34000000, %1" : "=v"(thetaq) : "v"(e.theta));     // 2^-23
        const float u1 = 1.0f + theta;
        const float inv_u1 = prnb::det_rcp(theta * u1) * theta;
        const float qq = theta * inv_u1;
        const float mpp = e.m * inv_u1;
        // t2 = -log2 P(X = 0) = m * log2(1+theta)/theta; below 2^-23, where 1 + theta is 1 in binary32, the
        // quotient is taken at 2^-23 (it is log2(e) * (1 - theta/2 + ...): 6e-8 off, well inside the margins)
        const float u1q = 1.0f + thetaq;
        const float t2 = e.m * (__builtin_amdgcn_logf(u1q) * __builtin_amdgcn_rcpf(u1q - 1.0f));
        // inversion class for sure: theta <= 16 and t = t2 * ln 2 below 19 by more than the two
        // evaluations can differ (NaN: not); every other valid sample is K3h's
        const unsigned long long light_m = K3_MASK(theta <= prnb::kLightTheta) & K3_MASK(t2 < kT2Sure);
        const float ps0 = __builtin_amdgcn_exp2f(-t2) * 4294967296.0f;       // pmf scaled by 2^32 (exact scaling)
        // threshold margin of this sample at k = 2 (in units of 2^-32)
        const float d2 = PRNB_FMA(t2, kMarginPerT2, kMargin0 + 2.0f * kMarginPerTerm);
        // PRNB-4's terms: P(k+1) = P(k) * (q + (mp - q)/(k+1)), the ratio by one fma from the 1/k table (k = 0: P(0) * mp)
        const float dd = mpp - qq;
        const float r0 = e.wf - ps0;
        const float ps1 = ps0 * mpp;
        const float r1 = r0 - ps1;
        const float ps2 = ps1 * PRNB_FMA(dd, 0.5f, qq);
        const float r2 = r1 - ps2;
        const unsigned long long hit_m = K3_MASK(r2 < 0.0f);
        const unsigned long long tail_m = K3_MASK(ps2 < 1.0f);
        const float near = __builtin_fminf(__builtin_fminf(__builtin_fabsf(r0), __builtin_fabsf(r1)), __builtin_fabsf(r2));
        const unsigned long long close_m = K3_MASK(near < d2) | K3_MASK(__builtin_fabsf(ps2 - 1.0f) < kTailBand);
        const uint32_t nneg = (prnb::f2u(r0) >> 31) + (prnb::f2u(r1) >> 31) + (prnb::f2u(r2) >> 31);
        // a hit at term 3 - nneg; no hit: 2 if the tail test ends the walk here
        const uint32_t res = 3u - (nneg > 1u ? nneg : 1u);
        const unsigned long long walk_m = valid_m & light_m & ~close_m;       // decided by this kernel
        const unsigned long long nz_m = K3_MASK(res != 0u);
        deliver(walk_m & (hit_m | tail_m) & nz_m, valid_m & ~walk_m, p2, res);
        s1_top -= cnt;
        const unsigned long long push_m = walk_m & ~(hit_m | tail_m);
        {
            const uint32_t slot = (uint32_t)s2_top + (uint32_t)lane_rank(push_m);
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            f32x4 e2;
            e2.x = ps2 * PRNB_FMA(dd, 0.33333334f, qq);      // pmf at k = 3 (the 1/k table's 1/3)
            e2.y = dd;
            e2.z = qq;
            e2.w = r2;
            // margin of the terms k = 3..6, rounded up to 16 significant bits, | pos
            const uint32_t m2 = ((prnb::f2u(d2 + 4.0f * kMarginPerTerm) + 0xffffu) & 0xffff0000u) | p2;
            asm volatile("s_mov_b64 exec, %0\n\tds_write_b128 %1, %2\n\tds_write_b32 %3, %4\n\ts_mov_b64 exec, -1"
                         :: "s"(push_m), "v"(s2_lds + (slot << 4)), "v"(e2), "v"(s2m_lds + (slot << 2)), "v"(m2) : "memory");
        }
        s2_top += __popcll(push_m);
    };
#undef K3_MASK

    // ---- stage 1 over the strip ----------------------------------------------------------------
    // What a pass needs per cell is wave-uniform and arrives by scalar loads issued one pass (the
    // row offset: three passes) earlier; the mean segments are loaded two cells ahead.  Nothing
    // is waited for inside the pass that needs it.
    static_assert(kStripCells == 128, "pos keeps the cell in 7 bits above the gene's 8");
    const CellInfo* cinfo = cellinfo + n0;                   // wave-uniform running pointers
    int32_t* flush_ptr = strip_out;                          // row cl - kRing of the strip
    const uint32_t lane4 = (uint32_t)lane * 4u;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t s1_lds = (uint32_t)(uintptr_t)&L.s1[0];     // LDS byte address of the stack
    const int32_t gload = VEC ? (g0 < G ? g0 : G - 4) - gbase : 0;
    const float* const mcol = means + gbase;
    struct Seg { float M[4]; };
    auto load_seg = [&](uint64_t row_bytes) -> Seg {
        Seg r;
        const float* rowp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(mcol) + row_bytes);
        if (VEC) {
            const float4 v = *reinterpret_cast<const float4*>(rowp + gload);
            r.M[0] = v.x; r.M[1] = v.y; r.M[2] = v.z; r.M[3] = v.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) r.M[j] = (g0 + j < G) ? rowp[lane4 + j] : 0.0f;
        }
        return r;
    };
    Seg cur = load_seg(cinfo[0].row_bytes), nxt = load_seg(cinfo[1].row_bytes);
    uint64_t row2 = cinfo[2].row_bytes;
    float s = cinfo[0].s;
    uint32_t posbase = cinfo[0].pos_base;
    uint32_t ph[4] = {cinfo[0].ph[0], cinfo[0].ph[1], cinfo[0].ph[2], cinfo[0].ph[3]};
    const uint64_t quad_p1 = (uint64_t)kPhiloxM1 * ((uint32_t)g0 >> 2);     // the gene quad's part of round 1
    const uint32_t quad_hi = (uint32_t)(quad_p1 >> 32), quad_lo = (uint32_t)quad_p1;
    // Nothing may be pending on the vector-memory counter when the loop is entered: the compiler's wait
    // for the first pass's mean segment would otherwise sit in the loop body, behind the row store.
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0)
#pragma unroll 1
    for (int cl = 0; cl < cells; ++cl) {
        // every lane runs the whole pass: the stack tops must stay wave-uniform, so no ballot
        // may sit under a divergent branch
        const float M[4] = {cur.M[0], cur.M[1], cur.M[2], cur.M[3]};
        if (cl >= kRing) {
            flush_row(cl - kRing, flush_ptr);
            flush_ptr += ld;
        }
        // The scalar loads go out only now, behind the flush's LDS read: scalar and LDS returns
        // share one counter that can only be waited down to zero, and the next LDS read is a
        // whole Philox call away.  (The mean load behind the row store, not in front of it: measured
        // 4 % faster, although the end-of-pass wait for the load then includes the store -- hipcc waits
        // for vmcnt(0) whenever a load and a store are both pending; a hand-counted vmcnt(1) behind a
        // load issued first bought nothing.)
        __builtin_amdgcn_sched_barrier(0);
        const Seg nn = load_seg(row2);
        const uint64_t row3 = cinfo[3].row_bytes;
        const float s_next = cinfo[1].s;
        const uint32_t posbase_next = cinfo[1].pos_base;
        const uint32_t ph_next[4] = {cinfo[1].ph[0], cinfo[1].ph[1], cinfo[1].ph[2], cinfo[1].ph[3]};
        ++cinfo;
        __builtin_amdgcn_sched_barrier(0);
        const prnb::Words W = philox_count_row(ph, quad_hi, quad_lo, k0, k1);
        uint32_t s1_at = s1_lds + ((uint32_t)s1_top << 4);     // LDS byte address of the stack's top (wave-uniform)
        // P(X=0) = exp(-m*log1p(theta)/theta) >= exp(-x) >= 1 - x + x^2/2 - x^3/6, x = m * phi
        // (prnb::zero_test_factor).  The polynomial is evaluated times 2^32 with 1e-5 taken off
        // the constant term: far more than every rounding of the exact evaluation, so a sample
        // settled here is one the exact path would also call 0 (and a sample with theta <= 0 is
        // 0 by definition); it is negative from x = 1.6 on, which keeps every sample of the
        // gamma-Poisson class out.
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float m = M[j] * s;
            // (packed binary32 instructions for two bounds at a time were tried: fewer instructions, 2 % slower)
            const float x = m * phi[j];
            const float bound32 = PRNB_FMA(PRNB_FMA(PRNB_FMA(-715827882.7f, x, 2147483648.0f), x, -4294967296.0f),
                                           x, 4294924346.0f);
            // The compare mask goes straight into an SGPR pair (every lane is active here, so it is
            // the ballot), the push is one LDS store under exec = mask: no branch, no exec save.
            // (The stage-1 loop must stay wave-uniform: the asm below ends with exec = -1.)
            const float wf = (float)W.w[j];                    // PRNB-4's remainder starts as this
            unsigned long long push_m;
            asm("v_cmp_nlt_f32 %0, %1, %2" : "=s"(push_m) : "v"(wf), "v"(bound32));   // not settled as 0 (or NaN)
            u32x4 e;                                           // S1Entry {m, theta, wf, pos}
            e.x = __float_as_uint(m);
            e.y = __float_as_uint(PRNB_FMA(a[j], m, bm1[j]));
            e.z = __float_as_uint(wf);
            e.w = posbase | (lane4 + j);
            uint32_t slot;                                     // top of the stack + 16 * rank among the pushing lanes
            asm("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(slot) : "v"(lane_rank(push_m)), "s"(s1_at));
            asm volatile("s_mov_b64 exec, %0\n\tds_write_b128 %1, %2\n\ts_mov_b64 exec, -1"
                         :: "s"(push_m), "v"(slot), "v"(e) : "memory");
            asm("s_lshl4_add_u32 %0, %1, %0" : "+s"(s1_at) : "s"(__popcll(push_m)) : "scc");
        }
        s1_top = (int)((s1_at - s1_lds) >> 4);
        while (s1_top >= 64) {
            stage2_pass();
            while (s2_top >= kS2Run) stage3_pass();
        }
        cur = nxt;
        nxt = nn;
        row2 = row3;
        s = s_next;
        ph[0] = ph_next[0]; ph[1] = ph_next[1]; ph[2] = ph_next[2]; ph[3] = ph_next[3];
        posbase = posbase_next;
    }

    // ---- drain ------------------------------------------------------------------------------------
    while (s1_top > 0) {
        stage2_pass();
        while (s2_top >= kS2Run) stage3_pass();
    }
    // The last walks of a strip would run with a handful of busy lanes (13 % of the lanes over a quarter of a
    // pass per cell on the headline workload): once nothing waits on S2 and at most kBail lanes still walk,
    // their samples go on K3h's list instead, which redoes them from the start.
    for (;;) {
        const unsigned long long busy_m = __builtin_amdgcn_ballot_w64(k != kIdle);
        if (s2_top == 0) {
            if (busy_m == 0ull) break;
            if (__popcll(busy_m) <= kBail) {
                if ((busy_m & __builtin_amdgcn_ballot_w64(hpend != kNoHeavy)) != 0ull) flush_heavy();
                if (k != kIdle) hpend = pos;
                break;
            }
        }
        stage3_pass();
    }
    for (int cl = (cells > kRing ? cells - kRing : 0); cl < cells; ++cl) flush_row(cl, strip_out + (int64_t)cl * ld);
    flush_late();
    if (__builtin_amdgcn_ballot_w64(hpend != kNoHeavy) != 0ull) flush_heavy();
    if (lane == 0) {
        heavy.count[region] = h_cnt;                 // above cap: K3h redoes the whole region
        if (h_cnt > heavy.cap) heavy.overflow[0] = 1u;   // (for prosstt_amd_last_list)
    }
}

}  // namespace k3
