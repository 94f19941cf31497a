// K3, streaming form: the fused count sampler as a three-stage pipeline run by each wave
// on its own strip of the count matrix, with LDS stacks between the stages so that every
// stage executes with (nearly) all 64 lanes busy.
//
// The scalar algorithm (PRNB-7, prnb_device.h) has very different costs per sample:
//   ~65 % of the samples of the headline workload are zeros that a 6-instruction bound
//         settles (exp(-m * phi_gene) <= P(X = 0), prnb::zero_test_factor);
//   the rest need P(X = 0) itself (two reciprocals, a log2 and an exp2 of the hardware), and
//   ~33 % then walk the pmf for k >= 1 (data-dependent length, half of them end at k <= 2);
//   ~0.1 % need gamma-Poisson.
// Run lane-per-sample, every wave pays for its slowest lane in every one of these.  Here
//   stage 1 (uniform)   one count-matrix row segment per wave pass: 16-B mean load, one
//                       Philox call per lane, the bound test as a compare mask; survivors are
//                       pushed on stack S1 under exec = mask;
//   stage 2 (64 of S1)  P(X = 0) and the class test, then the terms k = 0 .. 4 (PRNB-7; 0 .. 2 until round 5: a quarter of
//                       the walks that reached stage 3 ended within two more terms); what is still
//                       undecided is pushed on S2 with the pmf state at k = 5;
//   stage 3 (lanes pull from S2)  two groups of four pmf steps per lane per pass; a lane that finishes
//                       writes its count into the LDS row ring and pulls the next entry;
//   output              the last kRing rows of the strip live in LDS, 8 bits per count (a walk past
//                       term 252 -- 70 per 10^9 samples of the headline workload -- leaves for K3h with its state); a row
//                       is stored (one coalesced 1 KiB store per wave) kRing cells after stage 1
//                       started it.  The few counts that arrive later than that (long walks, and the
//                       entries at the bottom of the two LIFO stacks, which wait for the drain)
//                       are collected in LDS and written in bursts of 4-B stores -- a store per
//                       late count would sit in front of every wait for the next mean load
//                       (loads and stores retire in order on one counter);
//   samples of the gamma-Poisson class are only LISTED here (a lane that meets one keeps it and its scaled
//   mean in two registers; when a lane meets its second, and at the end of the strip, the wave writes what
//   its lanes hold to its own region of K3h's lists -- no atomics on that path: one counter for all waves
//   serialises them); the walks still running when the strip has nothing else to do travel there with their
//   state; sample_counts_heavy_kernel (k3_heavy.h) draws and continues them afterwards.
//
// P(X = 0): the hardware's v_rcp/v_log/v_exp ARE the definition (prnb_device.h: hw_p0), so
// stage 2 needs no error margins and no sample is given up because it came close to a threshold (PRNB-4
// defined P(X = 0) in polynomial arithmetic and paid for margins, a give-up list and K3h redo walks: -5 % of the
// kernel and -0.03 ms of K3h when removed, profiles/r04_ablation.txt).
// Results are pure functions of (sample parameters, seed, global cell id, gene), so the
// order in which the stacks are drained cannot change them.
//
// Written against the issue model measured on gfx950 (tools/microbench5.hip, microbench6.hip;
// DESIGN.md section 6): a SIMD issues one instruction per ~2.3 cycles at five waves per SIMD, whatever its kind;
// binary32 mul/add/sub/fma run beside the other vector kinds (compare, select, convert, min/max, left shift, mbcnt,
// 3-operand integer forms, 64-bit multiply: ~4.3 cycles each on a second unit; transcendentals ~8.3), and scalar
// instructions weigh little beside both.  The kernel is bound by the number of (vector) instructions it issues and by
// how many waves a CU holds (five blocks: the LDS budget below), so the walk is pure binary32 arithmetic on a binary32
// remainder, counts come from arithmetic shifts of sign bits, wave-level tests are lane masks formed by ONE compare each,
// and per-cell values arrive by scalar loads.
#pragma once
#include <type_traits>
#include "prnb_device.h"

namespace k3 {

constexpr int kBlock = 256;        // 4 waves
constexpr int kTileG = 256;        // genes per wave pass: 64 lanes x 4
constexpr int kStripCells = 128;   // most cells per wave (pos keeps the cell in 7 bits); the host launches 64
constexpr int kS1Cap = 192;        // < 64 left over + 128 pushed by half a pass (two samples per lane)
constexpr int kS2Cap = 104;        // < 40 left over + 64 pushed by one stage-2 pass
constexpr int kS2Run = 40;         // stage 3 runs while S2 holds at least this many entries (with eight terms per pass: 32 +0.9 %, 24 +4 %, 16 +7 %, 44 and 48 +0.7 %: profiles/r05_ablation.txt)
static_assert(kS2Cap >= kS2Run - 1 + 64 && kS1Cap >= 63 + 128, "a stack must take one more round of pushes");
constexpr int kRing = 8;           // rows of the strip kept in LDS (8 bits per count) before they are stored; a power of 2
constexpr int kBail = 6;           // walks handed to K3h WITH THEIR STATE when a strip has nothing else to do (see the drain)

constexpr int kLateCap = 64;       // results that missed their row wait here for one burst of stores (a pass delivers at most 64)
constexpr int kRingMaxK3 = 248;    // the last pass of a walk in this kernel is k3 = 248 (terms 245 .. 252: counts fit the ring's 8 bits); a walk still
                                   // undecided then is redone by K3h
constexpr int kLongSlots = 4;      // walks past term 252 a strip can keep for the hand-over (seventy per 10^9 samples: a fifth one is listed to be redone)
constexpr int kListTail = 16;      // uint2 words behind a region's staging list: kLongSlots states of 3 words ({term, d}, {q, remainder}, {pos | k3 << 16, -})
constexpr int kInvTab = 272;       // 1/k for k < 272: the reciprocals a pass at k3 <= 254 reads ahead (k3 + 6 .. k3 + 13)

// What stage 1 needs to know about a cell, packed by the preparation kernel so that one scalar load
// fetches it: byte offset of the cell's row in the mean tensor, library-size factor, (cell index
// within its strip) << 8, and the cell's share of the Philox call -- the counter is (cell_lo, cell_hi,
// gene quad, 0), so the halves of rounds 1 and 2 that do not depend on the gene are the same for a whole
// row of the count matrix: ph[0] = cell_hi ^ k0, ph[1] = hi(M1 * (hi(M0 * cell_lo) ^ k1)) ^ (k0 + W0),
// ph[2] = lo(M0 * cell_lo) ^ (k1 + W1), ph[3] = lo(M1 * (hi(M0 * cell_lo) ^ k1))  (philox_cell_part).
// The array holds N + 4 entries (the last cell repeated) so that prefetches need no clamp.
struct CellInfo { uint64_t row_bytes; float s; uint32_t reserved; uint32_t ph[4]; };
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
struct S2Entry { float ps, d, q, rem; };              // pmf (x 2^32) at k = 3, d = mp - q (the ratio of step k is q + d/(k+1)), q, what is left of wf

struct WaveLds {
    S1Entry s1_null;               // m = 0: what a lane of stage 2 reads when the stack holds fewer than 64 entries
    S1Entry s1[kS1Cap];
    S2Entry s2[kS2Cap];
    uint32_t s2p[kS2Cap];          // pos of the S2 entries
    uint32_t late[kLateCap];       // (pos << 16) | count of results whose row has left the ring already
};

// The samples left to K3h.  Wave w of block b owns region r = 4*b + w of every per-region array below:
//   dense[r][kDense]     the first kDense samples it lists, {pos (cell-in-strip << 8 | gene-in-tile), the sample's scaled mean m};
//   list[r][cap + tail]  the ones behind them, staged: when the strip is done a wave that listed more than kDense reserves room
//                        in one of kSegs dense segments (ONE atomic add, by the few waves of a hot gene tile only) and copies
//                        them there as {cell, gene, m}: K3h deals the segments out 64 entries at a time, so the hundreds of
//                        entries of a hot region spread over all its waves (K3h used to walk them region by region: the waves
//                        that owned a hot region or two ran three times as long as the median wave);
//   wst / wid[r][kWalkSlots]  the walks the wave had not finished when its strip ended, WITH THEIR STATE: {the next term, d, q,
//                        the remainder} and pos | k3 << 16 (the next term is k = k3 - 3), and behind them the walks that ran
//                        past term 252 during the strip (kept in the tail of `list` until then): K3h continues them instead of
//                        redoing them from k = 0 (until round 5: 2.8e5 walks of mean length 64-75 per C3 launch);
//   count[r]             listed | walks << 16.
// A region whose list was too small (more than one sample in 16 listed), or whose segment is full, is put on ovf_regions and
// redone by K3h sample by sample; overflow[0] != 0 says that there was one.
static_assert(offsetof(WaveLds, s1) == offsetof(WaveLds, s1_null) + sizeof(S1Entry), "s1[-1] must be the null entry");

typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int kSegs = 64;
constexpr int kDense = 32;
constexpr int kWalkSlots = kBail + kLongSlots;
struct HeavyEntry { int32_t n, g; float m; uint32_t pad; };
struct HeavyList {
    uint2* list; uint32_t cap;
    uint2* dense;
    uint32_t* count;
    f32x4_t* wst; uint32_t* wid;
    uint32_t* overflow;
    uint32_t* seg_cnt;               // [0, kSegs): entries per segment; [kSegs]: overflowed regions; [kSegs + 1]: walks to redo
    HeavyEntry* ent; uint32_t ent_cap;          // segment s: ent[s * ent_cap ..]
    uint32_t* ovf_regions;
    int2* redo;                      // {cell, gene} of the walks to redo from their start (kRedoCap)
};
constexpr int kRedoCap = 65536;

__device__ __forceinline__ uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

// rank of this lane among the lanes whose bit is set in `mask`
__device__ __forceinline__ int lane_rank(unsigned long long mask)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                          __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// VEC: 16-byte mean loads and row stores (G, ld multiples of 4, both bases aligned).  BIG: the two settings that pay on a
// problem of full-length strips whose output does not fit the last-level cache and cost 6 % each on a small one (C2:
// 8-cell strips, 100 MB of counts) -- the raised issue priority of stages 2 and 3, and system scope on the row stores.
#define K3_GLOBAL(T) __attribute__((address_space(1))) T*
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef int32_t i32x2_t __attribute__((ext_vector_type(2)));
static_assert(sizeof(HeavyEntry) == sizeof(u32x4_t) && sizeof(int2) == sizeof(i32x2_t), "stored as plain vectors");
template <bool VEC, bool BIG>
__global__ __launch_bounds__(kBlock, 5) void sample_counts_stream_kernel(
    const float* __restrict__ means, int32_t G, const CellInfo* __restrict__ cellinfo,
    const float* __restrict__ ga, const float* __restrict__ gbm1, const float* __restrict__ gphi,
    int64_t N, uint32_t k0, uint32_t k1, int32_t* __restrict__ out, int64_t ld, int32_t strips,
    int32_t strip_cells, uint2* __restrict__ heavy_list, uint32_t heavy_cap, uint2* __restrict__ heavy_dense,
    const HeavyList* __restrict__ heavy_ptr)
{
    // (K3h's lists are described by a record in device memory, read where it is needed -- in two rare branches and at the
    // end of the strip --, not by kernel arguments: twenty scalar registers held across the strip loop for nothing spill)
    // (the pointers of the record are global memory: said so, or the compiler emits flat_* operations, which are not
    // ordered with this wave's global_* / buffer_* operations -- see flush_late)
    struct HeavyGlobal {
        K3_GLOBAL(uint32_t) count; K3_GLOBAL(f32x4_t) wst; K3_GLOBAL(uint32_t) wid; K3_GLOBAL(uint32_t) overflow;
        K3_GLOBAL(uint32_t) seg_cnt; K3_GLOBAL(u32x4_t) ent; uint32_t ent_cap; K3_GLOBAL(uint32_t) ovf_regions; K3_GLOBAL(i32x2_t) redo;
    };
    auto heavy_late = [&]() -> HeavyGlobal {
        const HeavyList* p = heavy_ptr;
        asm volatile("" : "+s"(p));
        const __attribute__((address_space(4))) HeavyList* h = (const __attribute__((address_space(4))) HeavyList*)p;      // constant memory: scalar loads
        HeavyGlobal g;
        g.count = (K3_GLOBAL(uint32_t))h->count; g.wst = (K3_GLOBAL(f32x4_t))h->wst; g.wid = (K3_GLOBAL(uint32_t))h->wid;
        g.overflow = (K3_GLOBAL(uint32_t))h->overflow; g.seg_cnt = (K3_GLOBAL(uint32_t))h->seg_cnt; g.ent = (K3_GLOBAL(u32x4_t))h->ent;
        g.ent_cap = h->ent_cap; g.ovf_regions = (K3_GLOBAL(uint32_t))h->ovf_regions; g.redo = (K3_GLOBAL(i32x2_t))h->redo;
        return g;
    };
    // LDS of a block: 30 304 B.  gfx950 hands it out in 1 280-byte granules, so the five blocks per CU that the kernel's
    // speed rests on (the fifth: -11 %) fit as long as a block stays at or under 32 000 B (measured: 32 352 B gives four).
    // 1/k for k = -2 .. kInvTab-1: 0 below k = 1 (never used below 1: an idle stage-3 lane rests at k3 = 8); a pass reads
    // 1/(k3-2) .. 1/(k3+5) with k3 = 8, 16, ...: entry 6 is 16-byte aligned (prnb::kTabShift)
    __shared__ __attribute__((aligned(16))) float inv_k_store[prnb::kTabShift + kInvTab + 2];
    float* const inv_k = inv_k_store + prnb::kTabShift;
    __shared__ WaveLds lds_all[kBlock / 64];
    // the row rings, [wave][row slot][gene-in-tile] (8 bits per count), each aligned to its own size: a delivery's LDS
    // address is then (pos AND (size - 1)) OR base -- one instruction
    __shared__ __attribute__((aligned(kRing * 256))) uint8_t ring_all[kBlock / 64][kRing * 256];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    WaveLds& L = lds_all[wv];

    for (int k = tid - prnb::kTabShift; k < kInvTab + 2; k += kBlock) inv_k[k] = k > 0 ? 1.0f / (float)k : 0.0f;
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
        if (lane == 0) heavy_late().count[region] = 0u;
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

    // The stacks' tops are LDS byte addresses (wave-uniform): s1_at points behind the last S1 entry.
    const uint32_t s1_lds = (uint32_t)(uintptr_t)&L.s1[0];
    uint32_t s1_at = s1_lds;
    int s2_top = 0;                                  // wave-uniform
    constexpr uint32_t kNoHeavy = 0xffffffffu;
    uint32_t hpend = kNoHeavy;                       // pos of the sample this lane holds for the next append to K3h's list
    float hpend_m = 0.0f;                            // ... and its scaled mean
    // stage-3 lane state: st = {the next term (k = k3 - 3), d = mp - q, q, what is left of wf}; pos; k3 = the first of the
    // pass's two groups ends at k3 (6, 14, 22, ...).  WHICH lanes walk is wave-level state on the scalar unit (idle_s: bit = the lane
    // holds no walk): a pass takes every mask it forms AND NOT idle_s, so an idle lane may compute on whatever its
    // registers hold, and no vector instruction is spent on asking who is idle.
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 st = {0.0f, 0.0f, 0.0f, 0.0f};
    uint32_t pos = 0u;
    int k3 = 8;
    f32x4 inv = *reinterpret_cast<const f32x4*>(__builtin_assume_aligned(&inv_k[6], 16));   // 1/(k+1) .. 1/(k+4), k = 5: read one pass ahead
    f32x4 inv2 = *reinterpret_cast<const f32x4*>(__builtin_assume_aligned(&inv_k[10], 16));  // 1/(k+5) .. 1/(k+8)
    unsigned long long idle_s = ~0ull;                   // wave-uniform

    // S1/S2 entries carry pos = (cell-in-strip << 8) | gene-in-tile, under the bits of 2^23 (kPosMagic, below)
    constexpr uint32_t kPosMagic = 0x4B000000u;
    int32_t* const strip_out = out + n0 * ld + gbase;
    const uint32_t ld32 = (uint32_t)ld;              // strip rows * ld * 4 < 2^31, checked by the host
    // pos of the last sample whose row has already left the ring (wave-uniform; -1: none)
    int32_t flushed_pos = -1;
    uint8_t* const ring = ring_all[wv];
    for (int i = lane; i < kRing * 64; i += 64) reinterpret_cast<uint32_t*>(ring)[i] = 0u;
    if (lane < 4) reinterpret_cast<uint32_t*>(&L.s1_null)[lane] = 0u;

    // Rows are stored through a buffer resource over the strip's part of the matrix: the row is the scalar offset,
    // and a lane whose genes lie beyond G carries a vector offset of 2^31 = num_records, which the hardware's range
    // check drops (gfx950 checks soffset + voffset: tools/buffer_probe.hip) -- no exec switch around the store.
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(strip_out, 0, 0x80000000u, 0x00020000);
    const uint32_t store_voff = g0 < G ? (uint32_t)lane * 16u : 0x80000000u;
    uint32_t flush_off = 0u;                         // byte offset of the next row to store (wave-uniform)

    // store row `cl` of the strip from ring slot cl % kRing and clear the slot
    auto flush_row = [&](int cl) {
        uint32_t* slot = reinterpret_cast<uint32_t*>(ring + (cl & (kRing - 1)) * 256) + lane;
        const uint32_t packed = __hip_atomic_exchange(slot, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);   // read and clear: one ds_wrxchg_rtn_b32
        const int32_t v[4] = {(int32_t)(packed & 0xffu), (int32_t)((packed >> 8) & 0xffu),
                              (int32_t)((packed >> 16) & 0xffu), (int32_t)(packed >> 24)};
        if (VEC) {
            // non-temporal (and system scope when BIG): the matrix is written once; keeping it out of L2 leaves that to the
            // mean tensor (nt: -3.8 % on C3 against a plain store; sc1 on top: -0.4 % on C3, +6 % on C2, whose 100 MB of counts
            // the last-level cache takes)
            typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
            const u32x4_ row4 = {(uint32_t)v[0], (uint32_t)v[1], (uint32_t)v[2], (uint32_t)v[3]};
            if (BIG) __builtin_amdgcn_raw_buffer_store_b128(row4, out_rsrc, store_voff, flush_off, 18 /* nt | sc1 */);
            else __builtin_amdgcn_raw_buffer_store_b128(row4, out_rsrc, store_voff, flush_off, 2 /* nt */);
        } else {
            int32_t* dst = strip_out + (int64_t)cl * ld + lane * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (g0 + j < G) dst[j] = v[j];
        }
        flush_off += ld32 * 4u;
        flushed_pos = (int32_t)kPosMagic | (cl << 8) | 255;
    };

    // a finished count (> 0) goes into the row ring while its row is still there, else (rare) on
    // the late list; a burst of 4-B stores empties the list when it is full and when the strip ends,
    // always after the rows' own stores.  That a late 4-B store lands on top of its row's 16-B store rests
    // on the gfx9 rule that a wave's global_* / buffer_* operations are performed in issue order (it is what makes
    // vmcnt a counter: MI355X_MICROARCH.md, "s_waitcnt vmcnt"; flat_* would not be) -- tests/test_kernel_isa.py
    // checks that this kernel has no flat_ instruction, the whole-matrix tests that no late count is lost.
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
    const uint32_t ring_lds = (uint32_t)(uintptr_t)&ring[0];
    uint32_t ring_mask = (uint32_t)(kRing * 256 - 1);
    asm volatile("" : "+v"(ring_mask));            // (a register: v_and_or_b32 takes one scalar operand, the base)
    const uint32_t late_lds = (uint32_t)(uintptr_t)&L.late[0];
    // write the samples the lanes hold for K3h to this wave's region of the list (about ten entries
    // each time on the headline workload: the first lane to meet its second sample triggers it)
    uint2* const my_list = heavy_list + (uint64_t)region * (heavy_cap + (uint32_t)kListTail);
    uint2* const my_long = my_list + heavy_cap;      // (global memory, not LDS: the block is at its LDS budget, and this is touched seventy times per 10^9 samples)
    int n_long = 0;                                   // walks kept behind the staging list (wave-uniform; -1: one found every list full)
    uint32_t h_cnt = 0u;                              // wave-uniform
    auto flush_heavy = [&]() {
        const unsigned long long mp_ = K3_MASK(hpend != kNoHeavy);
        if (hpend != kNoHeavy) {
            const uint32_t slot = h_cnt + (uint32_t)lane_rank(mp_);
            if (slot < heavy_cap) {
                uint2* const dst = slot < (uint32_t)kDense ? heavy_dense + ((uint64_t)region * kDense + slot) : my_list + slot;
                *dst = make_uint2(hpend & 0xffffu, __float_as_uint(hpend_m));
            }
        }
        h_cnt += (uint32_t)__popcll(mp_);
        hpend = kNoHeavy;
    };
    // Every lane of the wave calls this at the end of a stage-2 or stage-3 pass.  The lanes of `ok_m`
    // deliver count `res` (1..255) of sample `p`: into the row ring while the row is still there, else
    // (rare) onto the late list; the lanes of `give_m` (rare as well) leave sample `p` of scaled mean `gm` to K3h.  One
    // test covers both rare paths.
    auto deliver = [&](unsigned long long ok_m, unsigned long long give_m, uint32_t p, uint32_t res, float gm) {
        const unsigned long long late_m = K3_MASK((int32_t)p <= flushed_pos);
        // slot = cell % kRing, gene-in-tile
        uint32_t ring_at;
        asm volatile("v_and_or_b32 %0, %2, %3, %4\n\ts_mov_b64 exec, %1\n\tds_write_b8 %0, %5\n\ts_mov_b64 exec, -1"
                     : "=&v"(ring_at) : "s"(ok_m & ~late_m), "v"(p), "v"(ring_mask), "s"(ring_lds), "v"(res) : "memory");
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
                asm volatile("s_mov_b64 exec, %2\n\tv_mov_b32 %0, %3\n\tv_mov_b32 %1, %4\n\ts_mov_b64 exec, -1"
                             : "+v"(hpend), "+v"(hpend_m) : "s"(give_m), "v"(p), "v"(gm));
            }
        }
    };
    // ---- stage 3: four pmf steps for every busy lane; idle lanes pull from S2 ------------------
    // PRNB-6's walk (the chop-down of prnb_device.h): the terms are subtracted from a binary32 remainder, the count
    // is the first k whose subtraction leaves it negative; when a group of four ends without that
    // and its last term is under 1 (the pmf has fallen under 2^-32) the count is the group's last k.
    // A walk enters at k = 3 and advances by 4: the four reciprocals 1/(k+1)..1/(k+4) are one aligned
    // 16-byte LDS read.  The remainders only fall, so there is a hit iff the last one is negative, and
    // the count is k3 less the number of negative remainders among the first three: arithmetic shifts of the
    // sign bits, no compares.  An idle lane runs the same arithmetic on stale registers; the three lane masks of a pass
    // are cleared of idle_s on the scalar unit.
    const uint32_t s2_lds = (uint32_t)(uintptr_t)&L.s2[0];
    const uint32_t s2p_lds = (uint32_t)(uintptr_t)&L.s2p[0];
    auto stage3_pass = [&]() {
        if (BIG) __builtin_amdgcn_s_setprio(2);           // (see stage2_pass)
        if (idle_s != 0ull && s2_top > 0) {
            // the idle lanes take the top entries of S2 (the lane of rank r the entry left + r), under exec = (idle and an
            // entry left): loads only.  (The stack's part of every address is formed on the scalar unit.)
            const int left_ = s2_top - __popcll(idle_s);
            const int left = left_ > 0 ? left_ : 0;
            const int rank = lane_rank(idle_s);
            const unsigned long long take_m = idle_s & K3_MASK(rank < s2_top - left);
            asm volatile("s_mov_b64 exec, %[tm]\n\t"
                         "ds_read_b128 %[st], %[ea]\n\t"
                         "ds_read_b32 %[pos], %[pa]\n\t"
                         "s_mov_b64 exec, -1\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : [st] "+v"(st), [pos] "+v"(pos)
                         : [tm] "s"(take_m), [ea] "v"(((uint32_t)rank << 4) + (s2_lds + ((uint32_t)left << 4))),
                           [pa] "v"(((uint32_t)rank << 2) + (s2p_lds + ((uint32_t)left << 2)))
                         : "memory");
            s2_top = left;
            idle_s &= ~take_m;
        }
        // Two groups of four terms per pass (A: k3 - 3 .. k3, B: k3 + 1 .. k3 + 4).  The definition decides group by group:
        // A ends the walk if it holds a negative remainder or its last term is under 1, else -- at k3 = kRingMaxK3 -- the
        // walk is K3h's, else B decides the same way.  (One group per pass, round 4's form, pays the pass's fixed cost --
        // the pull, the masks, the delivery -- per four terms: profiles/r05_ablation.txt.)
        const float d = st.y, q = st.z;
        const float r1 = st.w - st.x;
        const float ps1 = st.x * PRNB_FMA(d, inv.x, q);
        const float r2 = r1 - ps1;
        const float ps2 = ps1 * PRNB_FMA(d, inv.y, q);
        const float r3 = r2 - ps2;
        const float ps3 = ps2 * PRNB_FMA(d, inv.z, q);
        const float r4 = r3 - ps3;
        const float ps4 = ps3 * PRNB_FMA(d, inv.w, q);
        const float r5 = r4 - ps4;
        const float ps5 = ps4 * PRNB_FMA(d, inv2.x, q);
        const float r6 = r5 - ps5;
        const float ps6 = ps5 * PRNB_FMA(d, inv2.y, q);
        const float r7 = r6 - ps6;
        const float ps7 = ps6 * PRNB_FMA(d, inv2.z, q);
        const float r8 = r7 - ps7;
        const unsigned long long hit_a = K3_MASK(r4 < 0.0f), tail_a = K3_MASK(ps3 < 1.0f);
        const unsigned long long hit_b = K3_MASK(r8 < 0.0f), tail_b = K3_MASK(ps7 < 1.0f);
        // the count: the group's last k less one for each of its first three remainders that is negative (no hit: none is)
        const int32_t res_a = (k3 + ((int32_t)prnb::f2u(r1) >> 31) + ((int32_t)prnb::f2u(r2) >> 31)) + ((int32_t)prnb::f2u(r3) >> 31);
        const int32_t res_b = ((k3 + 4) + ((int32_t)prnb::f2u(r5) >> 31) + ((int32_t)prnb::f2u(r6) >> 31)) + ((int32_t)prnb::f2u(r7) >> 31);
        const unsigned long long end_a = (hit_a | tail_a) & ~idle_s;
        // a walk still undecided behind term 252 (k3 = 248: seventy per 10^9 samples of the headline workload) leaves this
        // kernel (the ring holds 8 bits per count, the 1/k table ends at 272): its state waits in LDS for the hand-over
        // at the end of the strip (below)
        const unsigned long long big_m = K3_MASK(k3 >= kRingMaxK3) & ~idle_s;
        const unsigned long long end_b = (hit_b | tail_b) & ~(idle_s | end_a);
        const unsigned long long long_m = big_m & ~(end_a | end_b);
        int32_t res_k;
        asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(res_k) : "v"(res_b), "v"(res_a), "s"(end_a));
        deliver(end_a | end_b, 0ull, pos, (uint32_t)res_k, 0.0f);
        idle_s |= end_a | end_b | big_m;                  // done lanes go idle
        st.x = ps7 * PRNB_FMA(d, inv2.w, q);
        // an idle lane rests at k3 = 6 with the reciprocals of a walk's first two groups (the reads below fetch them again
        // every pass): a pull then brings only the entry
        const int k3n = k3 + 8;
        st.w = r8;
        if (long_m != 0ull) {
            // (rare) keep the state -- the next term is k = k3n - 3 = 253 -- behind the wave's staging list; should a strip have
            // more than kLongSlots such walks, the others go onto K3h's list of walks to redo from their start; should that be
            // full as well, the whole region is redone (count -1)
            const int n_have = n_long < 0 ? kLongSlots : n_long;
            const int slot = n_have + lane_rank(long_m);
            const bool mine = (long_m >> lane) & 1ull;
            bool full = false;
            if (mine && slot < kLongSlots) {
                my_long[3 * slot] = make_uint2(__float_as_uint(st.x), __float_as_uint(st.y));
                my_long[3 * slot + 1] = make_uint2(__float_as_uint(st.z), __float_as_uint(st.w));
                my_long[3 * slot + 2] = make_uint2((pos & 0xffffu) | ((uint32_t)k3n << 16), 0u);
            }
            if (mine && slot >= kLongSlots) {
                const HeavyGlobal heavy = heavy_late();
                const uint32_t at = __hip_atomic_fetch_add(heavy.seg_cnt + kSegs + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (at < (uint32_t)kRedoCap) heavy.redo[at] = i32x2_t{(int32_t)(n0 + ((pos & 0xffffu) >> 8)), gbase + (int32_t)(pos & 255u)};
                else full = true;
            }
            const bool any_full = __builtin_amdgcn_ballot_w64(full) != 0ull;
            const int n_new = n_have + __popcll(long_m);
            n_long = (n_long < 0 || any_full) ? -1 : (n_new < kLongSlots ? n_new : kLongSlots);
        }
        asm("v_cndmask_b32 %0, %1, 8, %2" : "=v"(k3) : "v"(k3n), "s"(idle_s));
        inv = *reinterpret_cast<const f32x4*>(__builtin_assume_aligned(&inv_k[k3 - 2], 16));    // 1/(k+1..k+4), k = k3 - 3
        inv2 = *reinterpret_cast<const f32x4*>(__builtin_assume_aligned(&inv_k[k3 + 2], 16));   // 1/(k+5..k+8)
        if (BIG) __builtin_amdgcn_s_setprio(0);
    };

    // ---- stage 2: P(X = 0), class test, then the terms k = 0, 1, 2, for up to 64 entries of S1 -------
    // PRNB-6's parameters (prnb::hw_p0) and first group.  Samples of the gamma-Poisson class go to K3h's list.
    uint32_t lane16p = (uint32_t)lane * 16u + 16u;
    asm volatile("" : "+v"(lane16p));          // (kept as one register: top - lane16p is one instruction)
    // full = the stack holds at least 64 entries (every pass of the strip loop; the drain's may not)
    auto stage2_pass = [&](auto full_tag) {
        constexpr bool kFull = decltype(full_tag)::value;
        // Straight-line for every lane (a lane beyond the entries reads a null entry, which is invalid):
        // every wave-level test below is a lane mask formed outside divergent control flow.
        // The passes of stages 2 and 3 run at a raised issue priority: they are chains of dependent instructions behind
        // LDS reads, and a wave that gets through them sooner is back in stage 1, where the vector pipe is fed
        // (-0.8 % of the call on C3; the same priority for stage 1 instead: +1.6 %; on C2's 8-cell strips +6 %, hence BIG --
        // profiles/r04_ablation.txt section 7)
        if (BIG) __builtin_amdgcn_s_setprio(2);
        const uint32_t top = s1_at;
        int32_t at_c = (int32_t)(top - lane16p);                               // byte address of this lane's entry
        if (!kFull) at_c = at_c > (int32_t)(s1_lds - 16u) ? at_c : (int32_t)(s1_lds - 16u);      // s1[-1] is s1_null: invalid
        typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
        const u32x4_ raw = *reinterpret_cast<const __attribute__((address_space(3))) u32x4_*>((uint32_t)at_c);
        const float m = prnb::u2f(raw.x), theta_raw = prnb::u2f(raw.y), wf = prnb::u2f(raw.z);
        const uint32_t p2 = raw.w;
        // (m <= 0 or theta <= 0: the count is 0 by definition; stage 1 does not test that)
        const unsigned long long valid_m = K3_MASK(m > 0.0f) & K3_MASK(theta_raw > 0.0f);
        const float theta = prnb::clamp_theta_min(theta_raw);
        const prnb::HwP0 h = prnb::hw_p0(m, theta);
        const float qq = theta * h.iu;
        const float mpp = m * h.iu;
        // inversion class: theta <= 24 and t2 < 19 / ln 2 (NaN: not); every other valid sample is K3h's
        const unsigned long long light_m = K3_MASK(theta <= prnb::kLightTheta) & K3_MASK(h.t2 < prnb::kLightT2);
        const float ps0 = prnb::hw_exp2(-h.t2) * 4294967296.0f;       // pmf scaled by 2^32 (exact scaling)
        // the terms: P(k+1) = P(k) * (q + (mp - q)/(k+1)), the ratio by one fma from the 1/k table (k = 0: P(0) * mp)
        const float dd = mpp - qq;
        const float r0 = wf - ps0;
        const float ps1 = ps0 * mpp;
        const float r1 = r0 - ps1;
        const float ps2 = ps1 * PRNB_FMA(dd, 0.5f, qq);
        const float r2 = r1 - ps2;
        const float ps3 = ps2 * PRNB_FMA(dd, 0.33333334f, qq);      // (the 1/k table's values)
        const float r3 = r2 - ps3;
        const float ps4 = ps3 * PRNB_FMA(dd, 0.25f, qq);
        const float r4 = r3 - ps4;
        const unsigned long long hit_m = K3_MASK(r4 < 0.0f);
        const unsigned long long tail_m = K3_MASK(ps4 < 1.0f);
        // a hit: 4 less one for each of r0 .. r3 that is negative; no hit: 4 if the tail test ends the walk here
        const uint32_t res = (uint32_t)(((4 + ((int32_t)prnb::f2u(r0) >> 31)) + ((int32_t)prnb::f2u(r1) >> 31)) +
                                        (((int32_t)prnb::f2u(r2) >> 31) + ((int32_t)prnb::f2u(r3) >> 31)));
        const unsigned long long walk_m = valid_m & light_m;                  // decided by this kernel
        const unsigned long long end_m = hit_m | tail_m;
        deliver(walk_m & end_m, valid_m & ~light_m, p2, res, m);     // (a count of 0 is written as well: the ring slot holds 0 anyway)
        const uint32_t taken = kFull ? 1024u : (top - s1_lds < 1024u ? top - s1_lds : 1024u);
        s1_at = top - taken;
        const unsigned long long push_m = walk_m & ~end_m;
        {
            const uint32_t rank = (uint32_t)lane_rank(push_m);
            f32x4 e2;
            e2.x = ps4 * PRNB_FMA(dd, 0.2f, qq);             // pmf at k = 5 (the 1/k table's 1/5)
            e2.y = dd;
            e2.z = qq;
            e2.w = r4;
            asm volatile("s_mov_b64 exec, %0\n\tds_write_b128 %1, %2\n\tds_write_b32 %3, %4\n\ts_mov_b64 exec, -1"
                         :: "s"(push_m), "v"((rank << 4) + (s2_lds + ((uint32_t)s2_top << 4))), "v"(e2),
                            "v"((rank << 2) + (s2p_lds + ((uint32_t)s2_top << 2))), "v"(p2) : "memory");
        }
        s2_top += __popcll(push_m);
        if (BIG) __builtin_amdgcn_s_setprio(0);
    };
#undef K3_MASK

    // ---- stage 1 over the strip ----------------------------------------------------------------
    // What a pass needs per cell is wave-uniform and arrives by scalar loads issued one pass (the
    // row offset: three passes) earlier; the mean segments are loaded two cells ahead.  Nothing
    // is waited for inside the pass that needs it.
    static_assert(kStripCells == 128, "pos keeps the cell in 7 bits above the gene's 8");
    const CellInfo* cinfo = cellinfo + n0;                   // wave-uniform running pointers
    const uint32_t lane4 = (uint32_t)lane * 4u;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t gload_b = VEC ? (uint32_t)((g0 < G ? g0 : G - 4) - gbase) * 4u : 0u;
    const float* const mcol = means + gbase;
    typedef float Seg __attribute__((ext_vector_type(4)));      // (a vector: the two-deep rotation below is then four 64-bit moves)
    auto load_seg = [&](uint64_t row_bytes) -> Seg {
        Seg r;
        const float* rowp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(mcol) + row_bytes);
        if (VEC) {
            // a buffer load: the row address is the resource's base (scalar arithmetic), the lane's byte offset a
            // loop-invariant register -- no vector add per pass (hipcc forms base + lane offset in a VGPR pair otherwise)
            typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(rowp), 0, 0x7fffffff, 0x00020000);
            const u32x4_ raw = __builtin_amdgcn_raw_buffer_load_b128(rs, gload_b, 0, 0);
            r.x = __uint_as_float(raw.x); r.y = __uint_as_float(raw.y); r.z = __uint_as_float(raw.z); r.w = __uint_as_float(raw.w);
        } else {
            r.x = (g0 + 0 < G) ? rowp[lane4 + 0] : 0.0f;
            r.y = (g0 + 1 < G) ? rowp[lane4 + 1] : 0.0f;
            r.z = (g0 + 2 < G) ? rowp[lane4 + 2] : 0.0f;
            r.w = (g0 + 3 < G) ? rowp[lane4 + 3] : 0.0f;
        }
        return r;
    };
    Seg cur = load_seg(cinfo[0].row_bytes), nxt = load_seg(cinfo[1].row_bytes);
    uint64_t row2 = cinfo[2].row_bytes;
    float s = cinfo[0].s;
    uint32_t ph[4] = {cinfo[0].ph[0], cinfo[0].ph[1], cinfo[0].ph[2], cinfo[0].ph[3]};
    const uint64_t quad_p1 = (uint64_t)kPhiloxM1 * ((uint32_t)g0 >> 2);     // the gene quad's part of round 1
    const uint32_t quad_hi = (uint32_t)(quad_p1 >> 32), quad_lo = (uint32_t)quad_p1;
    // Nothing may be pending on the vector-memory counter when the loop is entered: the compiler's wait
    // for the first pass's mean segment would otherwise sit in the loop body, behind the row store.
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0)
    // pos travels as the binary32 number 2^23 + pos, whose low mantissa bits ARE pos (pos < 2^15): the step to the next
    // cell and the gene's offset within the quad are then binary32 additions, which issue beside the integer work of
    // the pass (-0.6 %: profiles/r04_ablation.txt).  Every reader masks (the ring address, the late list's shift) or
    // compares against a bound that carries the same 2^23 (flushed_pos); the list for K3h gets the low 16 bits.
    float posf = __uint_as_float(kPosMagic | lane4);
#pragma unroll 1
    for (int cl = 0; cl < cells; ++cl) {
        // every lane runs the whole pass: the stack tops must stay wave-uniform, so no ballot
        // may sit under a divergent branch
        const float m4[4] = {cur.x * s, cur.y * s, cur.z * s, cur.w * s};
        if (cl >= kRing) flush_row(cl - kRing);
        // The scalar loads go out only now, behind the flush's LDS read: scalar and LDS returns
        // share one counter that can only be waited down to zero, and the next LDS read is a
        // whole Philox call away.  (The mean load behind the row store, not in front of it: measured
        // 4 % faster, although the end-of-pass wait for the load then includes the store -- hipcc waits
        // for vmcnt(0) whenever a load and a store are both pending; a hand-counted vmcnt(1) behind a
        // load issued first bought nothing.  Two cells ahead: loading the next cell's segment and record in
        // place, into the registers this pass has just finished with, saves the ten moves of the rotation
        // below and is 4-5 % slower -- profiles/r04_ablation.txt, inplace1.)
        __builtin_amdgcn_sched_barrier(0);
        const Seg nn = load_seg(row2);
        const uint64_t row3 = cinfo[3].row_bytes;
        const float s_next = cinfo[1].s;
        const uint32_t ph_next[4] = {cinfo[1].ph[0], cinfo[1].ph[1], cinfo[1].ph[2], cinfo[1].ph[3]};
        ++cinfo;
        __builtin_amdgcn_sched_barrier(0);
        const prnb::Words W = philox_count_row(ph, quad_hi, quad_lo, k0, k1);
        // P(X=0) = exp(-m*log1p(theta)/theta) >= exp(-x) >= 1 - x + x^2/2 - x^3/6, x = m * phi
        // (prnb::zero_test_factor).  The polynomial is evaluated times 2^32 with 1e-5 taken off
        // the constant term: far more than every rounding and the hardware functions' error of the exact
        // evaluation, so a sample settled here is one the exact path would also call 0 (and a sample with
        // theta <= 0 is 0 by definition); it is negative from x = 1.6 on, which keeps every sample of the
        // gamma-Poisson class out.
        u32x4 e[4];                                            // S1Entry {m, theta, wf, pos}
        unsigned long long push_m[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float m = m4[j];
            // (packed binary32 instructions for two bounds at a time were tried: fewer instructions, 2 % slower)
            const float x = m * phi[j];
            const float bound32 = PRNB_FMA(PRNB_FMA(PRNB_FMA(-715827882.7f, x, 2147483648.0f), x, -4294967296.0f),
                                           x, 4294924346.0f);
            const float wf = (float)W.w[j];                    // the walk's remainder starts as this
            // The compare mask goes straight into an SGPR pair (every lane is active here, so it is the ballot)
            asm("v_cmp_nlt_f32 %0, %1, %2" : "=s"(push_m[j]) : "v"(wf), "v"(bound32));   // not settled as 0 (or NaN)
            e[j].x = __float_as_uint(m);
            e[j].y = __float_as_uint(PRNB_FMA(a[j], m, bm1[j]));
            e[j].z = __float_as_uint(wf);
            e[j].w = __float_as_uint(posf + (float)j);
        }
        // The pushes: one LDS store each under exec = its mask, at the stack's top + 16 * (rank among the pushing
        // lanes) -- no branch, no exec save, exec restored once per pair.  (v_mbcnt counts the bits of the mask it is GIVEN
        // below the lane, so it runs under the mask as well.)  Two samples of the quad, then the stages behind, then the
        // other two: S1 never holds more than 63 + 128 entries, which is what lets five blocks share a CU's LDS.
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint32_t t0, t1, c_;
            asm volatile(
                "s_mov_b64 exec, %[m0]\n\t"
                "v_mbcnt_lo_u32_b32 %[t0], exec_lo, 0\n\t"
                "v_mbcnt_hi_u32_b32 %[t0], exec_hi, %[t0]\n\t"
                "v_lshl_add_u32 %[t0], %[t0], 4, %[at]\n\t"
                "ds_write_b128 %[t0], %[e0]\n\t"
                "s_bcnt1_i32_b64 %[c], exec\n\t"
                "s_lshl4_add_u32 %[at], %[c], %[at]\n\t"
                "s_mov_b64 exec, %[m1]\n\t"
                "v_mbcnt_lo_u32_b32 %[t1], exec_lo, 0\n\t"
                "v_mbcnt_hi_u32_b32 %[t1], exec_hi, %[t1]\n\t"
                "v_lshl_add_u32 %[t1], %[t1], 4, %[at]\n\t"
                "ds_write_b128 %[t1], %[e1]\n\t"
                "s_bcnt1_i32_b64 %[c], exec\n\t"
                "s_lshl4_add_u32 %[at], %[c], %[at]\n\t"
                "s_mov_b64 exec, -1"
                : [at] "+s"(s1_at), [t0] "=&v"(t0), [t1] "=&v"(t1), [c] "=&s"(c_)
                : [m0] "s"(push_m[2 * h]), [m1] "s"(push_m[2 * h + 1]), [e0] "v"(e[2 * h]), [e1] "v"(e[2 * h + 1])
                : "memory", "scc");
            while (s1_at - s1_lds >= 64u * 16u) {
                stage2_pass(std::true_type{});
                while (s2_top >= kS2Run) stage3_pass();
            }
        }
        cur = nxt;
        nxt = nn;
        row2 = row3;
        s = s_next;
        ph[0] = ph_next[0]; ph[1] = ph_next[1]; ph[2] = ph_next[2]; ph[3] = ph_next[3];
        posf += 256.0f;
    }

    // ---- drain ------------------------------------------------------------------------------------
    while (s1_at != s1_lds) {
        stage2_pass(std::false_type{});
        while (s2_top >= kS2Run) stage3_pass();
    }
    // The last walks of a strip would run with a handful of busy lanes (13 % of the lanes over a quarter of a
    // pass per cell on the headline workload): once nothing waits on S2 and at most kBail lanes still walk,
    // their walks go to K3h with their state -- the next term, d, q, the remainder, k3 --, which continues them.
    unsigned long long bail_m = 0ull;                // the lanes whose walks travel (wave-uniform)
    for (;;) {
        const unsigned long long busy_m = ~idle_s;
        if (s2_top == 0) {
            if (busy_m == 0ull) break;
            if (__popcll(busy_m) <= kBail) { bail_m = busy_m; break; }
        }
        stage3_pass();
    }
    if (__builtin_amdgcn_ballot_w64(hpend != kNoHeavy) != 0ull) flush_heavy();
    for (int cl = (cells > kRing ? cells - kRing : 0); cl < cells; ++cl) flush_row(cl);
    flush_late();
    // ---- the hand-over to K3h ----------------------------------------------------------------------------
    const HeavyGlobal heavy = heavy_late();
    auto staged = [&](uint32_t i) -> unsigned long long {
        return __hip_atomic_load(reinterpret_cast<unsigned long long*>(my_list + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    const int n_kept = n_long < 0 ? kLongSlots : n_long;
    const uint32_t n_still = (uint32_t)__popcll(bail_m);
    // the walks, with their state: the ones still running, then the ones that ran past term 252 (this wave's own stores
    // behind its staging list have arrived once vmcnt is 0, and the loads read the device-level cache)
    if ((bail_m >> lane) & 1ull) {
        const uint64_t at = (uint64_t)region * kWalkSlots + (uint32_t)lane_rank(bail_m);
        heavy.wst[at] = st;
        heavy.wid[at] = (pos & 0xffffu) | ((uint32_t)k3 << 16);
    }
    if (n_kept != 0) {
        __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0)
        if (lane < n_kept) {
            const uint64_t at = (uint64_t)region * kWalkSlots + n_still + (uint32_t)lane;
            const unsigned long long w0 = staged(heavy_cap + 3u * (uint32_t)lane), w1 = staged(heavy_cap + 3u * (uint32_t)lane + 1u);
            const f32x4_t stl = {__uint_as_float((uint32_t)w0), __uint_as_float((uint32_t)(w0 >> 32)), __uint_as_float((uint32_t)w1), __uint_as_float((uint32_t)(w1 >> 32))};
            heavy.wst[at] = stl;
            heavy.wid[at] = (uint32_t)staged(heavy_cap + 3u * (uint32_t)lane + 2u);
        }
    }
    const bool too_many = h_cnt > heavy_cap || n_long < 0;
    if (lane == 0) heavy.count[region] = (h_cnt < 0xffffu ? h_cnt : 0xffffu) | ((n_still + (uint32_t)n_kept) << 16);
    // what was listed behind the first kDense entries (a wave of a hot gene tile): from the staging list to a segment, as
    // {cell, gene, m}
    bool spills = too_many;
    if (!too_many && h_cnt > (uint32_t)kDense) {
        const uint32_t n_over = h_cnt - (uint32_t)kDense;
        const uint32_t seg = region & (uint32_t)(kSegs - 1);
        uint32_t base_e = 0u;
        if (lane == 0) base_e = __hip_atomic_fetch_add(heavy.seg_cnt + seg, n_over, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0): the atomic's answer, and this wave's stores to its staging list
        base_e = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)base_e);
        spills = base_e + n_over > heavy.ent_cap;
        for (uint32_t i = (uint32_t)lane; i < n_over; i += 64u) {
            if (base_e + i < heavy.ent_cap) {
                const unsigned long long raw = staged((uint32_t)kDense + i);
                const uint32_t px = (uint32_t)raw;
                const u32x4_t e = {(uint32_t)(n0 + (px >> 8)), (uint32_t)(gbase + (int32_t)(px & 255u)), (uint32_t)(raw >> 32), 0u};       // HeavyEntry {n, g, m, -}
                heavy.ent[(uint64_t)seg * heavy.ent_cap + base_e + i] = e;
            }
        }
    }
    if (spills && lane == 0) {
        // redone by K3h as a whole (what did reach K3h's lists is drawn twice, to the same counts)
        const uint32_t at = __hip_atomic_fetch_add(heavy.seg_cnt + kSegs, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        heavy.ovf_regions[at] = region;
        heavy.overflow[0] = 1u;                      // (for prosstt_amd_last_list)
    }
}

}  // namespace k3
