// K3, streaming form: the fused count sampler as a three-stage pipeline run by each wave
// on its own strip of the count matrix, with LDS stacks between the stages so that every
// stage executes with (nearly) all 64 lanes busy.
//
// The scalar algorithm (PRNB-1, prnb_device.h) has three very different costs per sample:
//   ~67 % of the samples of the headline workload are zeros that a 5-instruction bound
//         settles (exp(-m) <= P(X = 0));
//   the rest need P(X = 0) exactly (reciprocal + log1p + exp, ~60 VALU), and
//   ~30 % then walk the pmf for k >= 1 (data-dependent length), ~2 % need gamma-Poisson.
// Run lane-per-sample, every wave pays for its slowest lane in every one of these.  Here
//   stage 1 (uniform)   one count-matrix row segment per wave pass: 16-B mean load, one
//                       Philox call per lane, bound test; the row is stored as zeros (one
//                       coalesced 1 KiB store); survivors are pushed on stack S1;
//   stage 2 (64 of S1)  exact P(X = 0); survivors (k >= 1) are pushed on S2 with the pmf
//                       state at k = 1;
//   stage 3 (lanes pull from S2)  two pmf steps per lane per pass; a lane that finishes writes
//                       its count into the LDS row ring and pulls the next entry;
//   output              the last kRing rows of the strip live in LDS, 16 bits per count; a row
//                       is stored (one coalesced 1 KiB store per wave) kRing cells after stage 1
//                       started it.  The few counts that arrive later than that are written
//                       directly (4-B store, after the row's own store);
//   samples of the gamma-Poisson path are only FLAGGED here (4 bits per lane and pass);
//   sample_counts_heavy_kernel (k3_heavy.h) draws them afterwards.
// Results are pure functions of (sample parameters, seed, global cell id, gene), so the
// order in which the stacks are drained cannot change them.
#pragma once
#include "prnb_device.h"

#ifndef K3_ABLATE
#define K3_ABLATE 0   // timing-only experiments (tools/ablate.sh); 0 in every shipped build
#endif

namespace k3 {

constexpr int kBlock = 256;        // 4 waves
constexpr int kTileG = 256;        // genes per wave pass: 64 lanes x 4
constexpr int kStripCells = 128;   // most cells per wave: long strips amortise the drain of stage 3
constexpr int kS1Cap = 320;        // < 64 left over + 256 pushed by one pass
constexpr int kS2Cap = 96;         // < 32 left over + 64 pushed by one stage-2 pass
constexpr int kS2Run = 32;
constexpr int kRing = 4;           // rows of the strip kept in LDS (16 bits per count) before they are stored; a power of 2         // stage 3 runs while S2 holds at least this many entries

struct S1Entry { float m, theta; uint32_t w, pos; };   // theta = a*m + b - 1, not yet clamped
struct S2Entry { float ps, num, q; uint32_t rem; };   // pmf (x 2^32) and numerator at k = 1

struct WaveLds {
    S1Entry s1[kS1Cap];
    S2Entry s2[kS2Cap];
    uint32_t s2pos[kS2Cap];
    uint16_t ring[kRing * 256];    // [row slot][gene-in-tile]: a walk ends below the 1/k table's 1023 entries
};

// rank of this lane among the lanes whose bit is set in `mask`
__device__ __forceinline__ int lane_rank(unsigned long long mask)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                          __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

template <bool VEC>
__global__ __launch_bounds__(kBlock) void sample_counts_stream_kernel(
    const float* __restrict__ means, int32_t G, const int32_t* __restrict__ row_of_cell,
    const float* __restrict__ scal, const float* __restrict__ ga, const float* __restrict__ gbm1,
    int64_t N, uint32_t k0, uint32_t k1, uint64_t cell_offset, const int64_t* __restrict__ cell_index,
    int32_t* __restrict__ out, int64_t ld, int32_t strips, int32_t strip_cells,
    uint8_t* __restrict__ heavy_flags, int32_t tiles_g)
{
    __shared__ float inv_k[prnb::kKTab + 4];          // 0 from the sentinel (k = KTAB-1) on
    __shared__ WaveLds lds_all[kBlock / 64];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    WaveLds& L = lds_all[wv];

    for (int k = tid; k < prnb::kKTab + 4; k += kBlock) inv_k[k] = (k && k < prnb::kKTab - 1) ? 1.0f / (float)k : 0.0f;
    __syncthreads();

    // block -> (gene tile, group of 4 strips); adjacent blocks share the gene tile (L2 reuse of
    // the mean tensor's column slice and of the per-gene parameters)
    const int32_t groups = (strips + 3) / 4;
    const int32_t tile_g = blockIdx.x / groups;
    const int32_t strip = (blockIdx.x - tile_g * groups) * 4 + wv;
    const int32_t gbase = tile_g * kTileG;
    const int32_t g0 = gbase + lane * 4;
    const int64_t n0 = (int64_t)strip * strip_cells;
    if (strip >= strips || n0 >= N) return;          // whole wave leaves together (no barrier below)
    const int cells = (int)((N - n0 < strip_cells) ? (N - n0) : strip_cells);

    float a[4], bm1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool in = g0 + j < G;
        a[j] = in ? ga[g0 + j] : 0.0f;
        bm1[j] = in ? gbm1[g0 + j] : 0.0f;
    }

    int s1_top = 0, s2_top = 0;                      // wave-uniform
    // stage-3 lane state
    bool active = false;
    float ps = 0.0f, num = 0.0f, q = 0.0f;
    uint32_t rem = 0u, pos = 0u;
    int k = 0;
    float inv1 = 0.0f, inv2 = 0.0f;                  // 1/(k+1), 1/(k+2): read one pass ahead

    // S1/S2 entries carry pos = (cell-in-strip << 8) | gene-in-tile
    int32_t* const strip_out = out + n0 * ld + gbase;
    const uint32_t ld32 = (uint32_t)ld;              // strip rows * ld < 2^32, checked by the host
    // pos of the last sample whose row has already left the ring (wave-uniform; -1: none)
    int32_t flushed_pos = -1;
    for (int i = lane; i < kRing * 128; i += 64) reinterpret_cast<uint32_t*>(L.ring)[i] = 0u;

    // a finished count: into the row ring while its row is still there, else straight to memory
    auto deliver = [&](uint32_t p, int32_t res) {
        if ((int32_t)p > flushed_pos) {
            L.ring[p & (kRing * 256 - 1)] = (uint16_t)res;       // slot = cell % kRing, then gene-in-tile
        } else {
            strip_out[(p >> 8) * ld32 + (p & 255u)] = res;
        }
    };
    // store row `cl` of the strip from ring slot cl % kRing and clear the slot
    auto flush_row = [&](int cl) {
        uint2* slot = reinterpret_cast<uint2*>(L.ring + (cl & (kRing - 1)) * 256) + lane;
        const uint2 packed = *slot;
        *slot = make_uint2(0u, 0u);
        const int32_t v[4] = {(int32_t)(packed.x & 0xffffu), (int32_t)(packed.x >> 16),
                              (int32_t)(packed.y & 0xffffu), (int32_t)(packed.y >> 16)};
        int32_t* dst = strip_out + (int64_t)cl * ld + lane * 4;
        if (g0 < G) {
            if (VEC) {
                *reinterpret_cast<int4*>(dst) = make_int4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (g0 + j < G) dst[j] = v[j];
            }
        }
        flushed_pos = (cl << 8) | 255;
    };

    // ---- stage 3: one pmf step for every busy lane; idle lanes pull from S2 -------------------
    auto stage3_pass = [&]() {
        const unsigned long long want = __builtin_amdgcn_ballot_w64(!active);
        if (want != 0ull && s2_top > 0) {
            const int rank = lane_rank(want);
            if (!active && rank < s2_top) {
                const int idx = s2_top - 1 - rank;
                const S2Entry e = L.s2[idx];
                ps = e.ps; num = e.num; q = e.q; rem = e.rem;
                pos = L.s2pos[idx];
                k = 1;
                inv1 = 0.5f;                           // 1/2
                inv2 = 0.33333334f;                    // 1/3 (binary32-rounded, = inv_k[3])
                active = true;
            }
            const int taken = __popcll(want);
            s2_top = (taken < s2_top) ? s2_top - taken : 0;
        }
        // two pmf steps per pass, straight-line: lanes that are idle or finish at the first
        // step compute garbage that nothing reads
        const uint32_t pfa = (uint32_t)ps;
        const bool hit_a = rem < pfa;
        const bool end_a = hit_a || pfa == 0u;          // pmf under 2^-32 (or the table's end): 0
        const uint32_t rem_b = rem - pfa;
        const float ps_b = (ps * num) * inv1;
        const float num_b = num + q;
        const uint32_t pfb = (uint32_t)ps_b;
        const bool hit_b = rem_b < pfb;
        const bool end_b = hit_b || pfb == 0u;
        const bool done = active && (end_a || end_b);
        if (done) {
            const int32_t res = end_a ? (hit_a ? k : 0) : (hit_b ? k + 1 : 0);
            if (res != 0) deliver(pos, res);
            active = false;
        }
        rem = rem_b - pfb;
        ps = (ps_b * num_b) * inv2;
        num = num_b + q;
        k = active ? k + 2 : 0;                        // idle lanes must not walk off the table
        inv1 = inv_k[k + 1];
        inv2 = inv_k[k + 2];
    };

    // ---- stage 2: exact P(X = 0) for up to 64 entries of S1 ----------------------------------
    auto stage2_pass = [&]() {
        const int cnt = s1_top < 64 ? s1_top : 64;
        const bool mine = lane < cnt;
        bool push = false;
        S2Entry e2;
        uint32_t p2 = 0u;
        e2.ps = 0.0f; e2.num = 0.0f; e2.q = 0.0f; e2.rem = 0u;
        if (mine) {
            const S1Entry e = L.s1[s1_top - 1 - lane];
            const float theta = __builtin_fminf(__builtin_fmaxf(e.theta, prnb::kThetaMin), prnb::kThetaMax);
            const float u1 = 1.0f + theta;
            const float d = prnb::det_rcp(theta * u1);
            const float inv_th = d * u1, inv_u1 = d * theta;
            const float qq = theta * inv_u1;
            const float mpp = e.m * inv_u1;
            const float t = e.m * (prnb::det_log1p(theta) * inv_th);
            const float p0 = __builtin_fminf(prnb::det_exp(-t), 0.99999994f);
            const float ps0 = p0 * 4294967296.0f;
            const uint32_t pf = (uint32_t)ps0;
            if (e.w >= pf) {                      // k >= 1 (pf > 0 on the light path: P0 >= e^-12)
                e2.rem = e.w - pf;
                e2.num = mpp + qq;                // numerator of the step 1 -> 2
                e2.q = qq;
                e2.ps = (ps0 * mpp) * inv_k[1];   // pmf at k = 1, scaled by 2^32 (exact scaling)
                p2 = e.pos;
                push = true;
            }
        }
        s1_top -= cnt;
        const unsigned long long m2 = __builtin_amdgcn_ballot_w64(push);
        if (push) {
            const int slot = s2_top + lane_rank(m2);
            L.s2[slot] = e2;
            L.s2pos[slot] = p2;
        }
        s2_top += __popcll(m2);
    };

    // ---- stage 1 over the strip ----------------------------------------------------------------
    // Row index, scaling and global id of all 128 cells of the strip are fetched once (lane l
    // holds cells l and l+64) and handed out by v_readlane; the mean segments are loaded two
    // cells ahead.  Nothing a pass needs is waited for inside the pass.
    static_assert(kStripCells == 128, "two cells per lane");   // strip_cells <= kStripCells
    int32_t rowv[2];
    float sv[2];
    uint64_t cellv[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c = lane + 64 * h;
        const int64_t n = n0 + (c < cells ? c : 0);
        rowv[h] = row_of_cell[n];
        sv[h] = scal[n];
        cellv[h] = cell_index ? (uint64_t)cell_index[n] : cell_offset + (uint64_t)n;
    }
    auto cell_row = [&](int cl) -> int64_t {
        const int src = cl & 63;
        return (int64_t)((cl < 64) ? __builtin_amdgcn_readlane(rowv[0], src) : __builtin_amdgcn_readlane(rowv[1], src));
    };
    struct Seg { float M[4]; };
    auto load_seg = [&](int cl) -> Seg {
        Seg r;
        r.M[0] = r.M[1] = r.M[2] = r.M[3] = 0.0f;
        if (cl < cells && g0 < G) {
            const int64_t row = cell_row(cl);
            if (VEC) {
                const float4 v = *reinterpret_cast<const float4*>(means + row * G + g0);
                r.M[0] = v.x; r.M[1] = v.y; r.M[2] = v.z; r.M[3] = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (g0 + j < G) r.M[j] = means[row * G + g0 + j];
            }
        }
        return r;
    };
    Seg cur = load_seg(0), nxt = load_seg(1);
#pragma unroll 1
    for (int cl = 0; cl < cells; ++cl) {
        const Seg nn = load_seg(cl + 2);
        const int src = cl & 63;
        const float s = __uint_as_float((cl < 64) ? __builtin_amdgcn_readlane(__float_as_uint(sv[0]), src)
                                                  : __builtin_amdgcn_readlane(__float_as_uint(sv[1]), src));
        const uint32_t c_lo = (cl < 64) ? __builtin_amdgcn_readlane((uint32_t)cellv[0], src)
                                        : __builtin_amdgcn_readlane((uint32_t)cellv[1], src);
        const uint32_t c_hi = (cl < 64) ? __builtin_amdgcn_readlane((uint32_t)(cellv[0] >> 32), src)
                                        : __builtin_amdgcn_readlane((uint32_t)(cellv[1] >> 32), src);
        const uint64_t cell = ((uint64_t)c_hi << 32) | c_lo;
        // every lane runs the whole pass (lanes beyond G just never qualify): the stack tops
        // must stay wave-uniform, so no ballot may sit under a divergent branch
        const float M[4] = {cur.M[0], cur.M[1], cur.M[2], cur.M[3]};
        if (cl >= kRing) flush_row(cl - kRing);
#if K3_ABLATE == 3     // no Philox: a 2-instruction hash stands in
        prnb::Words W;
        W.w[0] = ((uint32_t)cell * 2654435761u) ^ ((uint32_t)g0 * 40503u); W.w[1] = W.w[0] * 3u + k0;
        W.w[2] = W.w[1] ^ 0x9E3779B9u; W.w[3] = W.w[2] + W.w[0];
#else
        const prnb::Words W =
            prnb::philox4x32_10((uint32_t)cell, (uint32_t)(cell >> 32), (uint32_t)g0 >> 2, 0u, k0, k1);
#endif
        uint32_t hflag = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // genes beyond G carry M = 0 and so never qualify; no lane-dependent branch here
            const float m = M[j] * s;
            const float theta = PRNB_FMA(a[j], m, bm1[j]);
            const bool valid = (m > 0.0f) && (theta > 0.0f);
            const bool light = (m <= prnb::kLightM) && (theta <= prnb::kLightTheta);
            // P(X=0) = exp(-m*log1p(theta)/theta) >= exp(-m) >= 1 - m + m^2/2 - m^3/6.  The
            // polynomial is evaluated times 2^32 with 1e-5 taken off the constant term: far more
            // than every rounding of the exact evaluation, so a sample settled here is one the
            // exact path would also call 0.
            const float bound32 = PRNB_FMA(PRNB_FMA(PRNB_FMA(-715827882.7f, m, 2147483648.0f), m, -4294967296.0f),
                                           m, 4294924346.0f);
            const bool zero = (float)W.w[j] < bound32;
            const bool to_s1 = valid && light && !zero;
            const unsigned long long m1 = __builtin_amdgcn_ballot_w64(to_s1);
            if (to_s1) {
                S1Entry e;
                e.m = m; e.theta = theta; e.w = W.w[j]; e.pos = ((uint32_t)cl << 8) | (uint32_t)(lane * 4 + j);
                L.s1[s1_top + lane_rank(m1)] = e;
            }
            s1_top += __popcll(m1);
            hflag |= (valid && !light) ? (1u << j) : 0u;
        }
        // gamma-Poisson samples are only flagged: 4 bits per lane, one byte per (cell, tile, lane)
        heavy_flags[(((n0 + cl) * tiles_g + tile_g) << 6) + lane] = (uint8_t)hflag;
#if K3_ABLATE == 2      // stage 1 only
        s1_top = 0;
#elif K3_ABLATE == 1    // no stage 3
        while (s1_top >= 64) { stage2_pass(); s2_top = 0; }
#else
        while (s1_top >= 64) {
            stage2_pass();
            while (s2_top >= kS2Run) stage3_pass();
        }
#endif
        cur = nxt;
        nxt = nn;
    }

    // ---- drain ------------------------------------------------------------------------------------
    while (s1_top > 0) {
        stage2_pass();
        while (s2_top >= kS2Run) stage3_pass();
    }
    while (s2_top > 0 || __builtin_amdgcn_ballot_w64(active) != 0ull) stage3_pass();
    for (int cl = (cells > kRing ? cells - kRing : 0); cl < cells; ++cl) flush_row(cl);
}

}  // namespace k3
