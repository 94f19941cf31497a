// libprosstt_amd.so -- HIP kernels (gfx950) and the C ABI of include/prosstt_amd.h.
//
// Kernels
//   prep_kernel             binary64 scaling/alpha/beta -> binary32 sampler parameters, per-cell records, flag words,
//                           the per-cell / per-gene part of the domain check
//   k3::sample_counts_stream_kernel + k3::sample_counts_heavy_kernel (k3_stream.h, k3_heavy.h)
//                           K3: fused gather * scale -> get_pr_umi -> NB draw
//                           (simulation.py:602-651, count_model.py:131-161)
//   row_flags_kernel        the per-row part of scipy's argument check (simulation.py:647-648); the per-sample part rides in K3h
//   nb_params_kernel        the deterministic intermediates of the same path
//   hw_math_kernel          the probe of the three hardware functions of the sampler's definition
//   lineage_attempt_lds_kernel / lineage_attempt_kernel
//                           K2a: max(programs@H) and per-sibling anticorrelated-gene
//                           counts without materialising (T,G)   (simulation.py:269-272)
//   lineage_commit_kernel   K2b: rel = programs@H in binary64 + per-gene max
//   lineage_walk_kernel     K1: device-mode expression programs (simulation.py:89-124)
//   means_from_rel_kernel   Tree.add_genes: exp(rel)*base -> binary32 mean tensor (tree.py:181-182)
// Host only: prosstt_amd_numpy_programs (numpy_stream.h): numpy's legacy stream for a batch of attempts.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <limits>
#include <new>
#include <vector>

#include <hip/hip_ext.h>
#include "../../include/prosstt_amd.h"
#include "prnb_device.h"
#include "k3_stream.h"
#include "k3_heavy.h"
#include "numpy_stream.h"

#define PA_EXPORT extern "C" __attribute__((visibility("default")))
// Nothing throws across the C ABI: every exported function is a function-try-block that ends in PA_CATCH (the host side
// uses std::vector: an allocation failure becomes PROSSTT_AMD_ENOMEM, not an exception in a ctypes caller).
#define PA_CATCH                                                                                       \
    catch (const std::bad_alloc&) { return fail(PROSSTT_AMD_ENOMEM, "out of host memory"); }          \
    catch (const std::exception& e) { return fail(PROSSTT_AMD_EINVAL, "unexpected exception: %s", e.what()); } \
    catch (...) { return fail(PROSSTT_AMD_EINVAL, "unexpected exception"); }

// ------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(e_ == hipErrorOutOfMemory ? PROSSTT_AMD_ENOMEM : PROSSTT_AMD_EHIP,     \
                        "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);   \
    } while (0)

// ------------------------------------------------------------------ context
struct prosstt_amd_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    void* ws = nullptr;          // grow-only device workspace
    size_t ws_bytes = 0;
    int64_t* scratch = nullptr;  // device: [0] domain flag, [1] max bits, [2..] counters
    int64_t* h_scratch = nullptr;  // pinned mirror
    std::vector<hipEvent_t> events;  // (start, stop) pairs of kernels launched with TIME_KERNEL
    size_t events_used = 0;
    // the K3h list of the last sample_counts call (inside `ws`; read by prosstt_amd_last_list)
    k3::HeavyList list{};        // (list.seg_cnt == nullptr: none)
    uint64_t list_regions = 0;
    int64_t list_groups = 0, list_strip_cells = 0;
    int heavy_grid = 1280;       // blocks of K3h that are resident at once on this device (5 per CU: its 30 496 B of LDS)
    // domain check: one byte per row of the mean tensor last scanned ("has an entry that is not > 0"), and which tensor that was
    uint8_t* row_bad = nullptr;
    size_t row_bad_cap = 0;
    const float* row_bad_means = nullptr;
    int64_t row_bad_rows = 0;
    int32_t row_bad_G = 0;
    uint32_t call_parity = 0;    // the full-test request word of a call is scratch[6 + parity]; a call clears the other one
};

// next (start, stop) event pair of the ctx's pool
static int next_event_pair(prosstt_amd_ctx* c, hipEvent_t* a, hipEvent_t* b)
{
    if (c->events_used + 2 > c->events.size()) {
        hipEvent_t e0, e1;
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        c->events.push_back(e0);
        c->events.push_back(e1);
    }
    *a = c->events[c->events_used];
    *b = c->events[c->events_used + 1];
    c->events_used += 2;
    return 0;
}
// scratch words: [3] list overflow of the last sample_counts call; [4] sticky "domain error", [5] sticky "row index outside
// the tensor" (both set by checked calls, read and cleared by prosstt_amd_domain_status); [6], [7] "some gene has alpha < 0 or
// beta < 1: run the full test" of calls of even / odd parity
constexpr int kScratchWords = 128 + 32 * 16;
constexpr int kStickyDomain = 4, kStickyRow = 5, kFullReq = 6;

static int ws_reserve(prosstt_amd_ctx* c, size_t bytes)
{
    // every user of the workspace overwrites what the last sample_counts call left there: its list is gone
    c->list = k3::HeavyList{};
    if (bytes <= c->ws_bytes) return 0;
    if (c->ws) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipFree(c->ws));
        c->ws = nullptr;
        c->ws_bytes = 0;
    }
    bytes = (bytes + (1u << 20)) & ~((size_t)(1u << 20) - 1);
    HIP_TRY(hipMalloc(&c->ws, bytes));
    c->ws_bytes = bytes;
    return 0;
}

// Temporary device copies of host arrays (PROSSTT_AMD_HOST_INPUTS / _OUTPUT).
struct Staging {
    std::vector<void*> bufs;
    ~Staging() { for (void* p : bufs) (void)hipFree(p); }
    int alloc(void** p, size_t bytes)
    {
        HIP_TRY(hipMalloc(p, bytes ? bytes : 1));
        bufs.push_back(*p);
        return 0;
    }
    int upload(const void* host, size_t bytes, const void** dev, hipStream_t s)
    {
        void* p = nullptr;
        int rc = alloc(&p, bytes);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(p, host, bytes, hipMemcpyHostToDevice, s));
        *dev = p;
        return 0;
    }
};

// ------------------------------------------------------------------ kernels

// One launch prepares a sample_counts call: binary64 scaling/alpha/beta -> the binary32 sampler
// parameters (and the zero-test factor), the per-cell records of the streaming kernel (k3::CellInfo;
// N + 4 entries, the last cell repeated; skipped when `info` is NULL), and the call's flag words.
// With `row_bad` (a checked call) it is also the per-cell and per-gene part of the domain check: scipy's argument
// check in the reference fails iff some mean m = M*s is <= 0 (or NaN) or some theta = a*m + b - 1 is < 0.  With every
// scaling > 0 the first holds iff a USED row of the mean tensor has an entry that is not > 0 (row_bad, from
// row_flags_kernel); with every a >= 0 and b >= 1 the second cannot happen -- only when a gene has a < 0 or b < 1
// is the full N x G test needed (flags[kFullReq + parity], read at the end of K3h).
__global__ void prep_kernel(const double* __restrict__ scaling, int64_t N,
                            const double* __restrict__ alpha, const double* __restrict__ beta, int32_t G,
                            float* __restrict__ scal_f, float* __restrict__ a_f,
                            float* __restrict__ bm1_f, float* __restrict__ phi_f,
                            const int32_t* __restrict__ row_of_cell, int64_t rows, uint64_t cell_offset,
                            const int64_t* __restrict__ cell_index, int32_t strip_cells,
                            uint32_t k0, uint32_t k1, k3::CellInfo* __restrict__ info, int64_t* __restrict__ flags,
                            const uint8_t* __restrict__ row_bad, uint32_t parity, k3::HeavyList heavy,
                            k3::HeavyList* __restrict__ heavy_rec)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (heavy_rec && i < k3::kSegs + 2) heavy.seg_cnt[i] = 0u;      // the fill of the dense lists K3h reads
    if (heavy_rec && i == 0) *heavy_rec = heavy;                       // ... and their description, for the two kernels behind
    if (flags && i == 0) {
        flags[3] = 0;                       // list overflow of this call (set by the streaming kernel)
        flags[kFullReq + (parity ^ 1u)] = 0;  // the NEXT call's full-test request (this call's was cleared by the previous one)
    }
    if (i < N) {
        const float s = (float)scaling[i];
        scal_f[i] = s;
        if (row_bad) {
            const int64_t r = row_of_cell[i];
            if (r < 0 || r >= rows) flags[kStickyRow] = 1;     // reported as EINVAL, never read through
            else if (row_bad[r] || !(s > 0.0f)) flags[kStickyDomain] = 1;
        }
    }
    if (i < G) {
        const float a = (float)alpha[i];
        const float bm1 = (float)(beta[i] - 1.0);   // binary64 subtraction: beta = 1 + 1e-8 must survive
        a_f[i] = a;
        bm1_f[i] = bm1;
        phi_f[i] = prnb::zero_test_factor(a, bm1);
        if (row_bad && (!(a >= 0.0f) || !(bm1 >= 0.0f))) flags[kFullReq + parity] = 1;
    }
    if (info && i < N + 4) {
        const int64_t n = i < N ? i : N - 1;
        const uint64_t cell = cell_index ? (uint64_t)cell_index[n] : cell_offset + (uint64_t)n;
        k3::CellInfo c;
        // an index outside the tensor (the caller's bug; the checked mode reports it) must not become a wild read
        const int64_t row = row_of_cell[n] < 0 ? 0 : (row_of_cell[n] >= rows ? rows - 1 : row_of_cell[n]);
        c.row_bytes = (uint64_t)row * (uint64_t)G * 4u;
        c.s = (float)scaling[n];
        c.reserved = 0u;
        k3::philox_cell_part((uint32_t)cell, (uint32_t)(cell >> 32), k0, k1, c.ph);
        info[i] = c;
    }
}

constexpr int kTileG = 256;   // genes per tile of the streaming kernel = 64 lanes x 4

__global__ void nb_params_kernel(const float* __restrict__ means, int32_t G,
                                 const int32_t* __restrict__ row_of_cell,
                                 const float* __restrict__ scal, const float* __restrict__ ga,
                                 const float* __restrict__ gbm1, int64_t N, int64_t rows, float* __restrict__ mu,
                                 float* __restrict__ p, float* __restrict__ r,
                                 int32_t* __restrict__ path)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * G) return;
    const int64_t n = i / G;
    const int32_t g = (int32_t)(i - n * G);
    int64_t row = row_of_cell[n];
    row = row < 0 ? 0 : (row >= rows ? rows - 1 : row);      // device pointers are unchecked: never a wild read
    const prnb::Params P = prnb::make_params(means[row * G + g], scal[n], ga[g], gbm1[g]);
    if (mu) mu[i] = P.m;
    if (p) p[i] = P.valid ? P.theta * P.iu : 0.0f;
    if (r) r[i] = P.valid ? P.m * P.inv_th : 0.0f;
    if (path) path[i] = !P.valid ? 0 : (P.light ? 1 : 2);
}

// The probe of the three hardware functions of the inversion class (prnb_device.h: hw_rcp, hw_log2, hw_exp2): their values over a
// range of binary32 bit patterns, written by the device itself.  The scalar model that checks the kernels reads these
// tables instead of re-implementing the functions.  op: 0 v_rcp_f32(x), 1 v_log_f32(x), 2 v_exp_f32(-x).
__global__ void hw_math_kernel(int32_t op, uint32_t first_bits, uint64_t count, float* __restrict__ y)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float(first_bits + (uint32_t)i);
        y[i] = op == 0 ? prnb::hw_rcp(x) : (op == 1 ? prnb::hw_log2(x) : prnb::hw_exp2(-x));
    }
}

// ... and the gather form: y[i] = op(x[i]) for the caller's arguments -- what the scalar model ASKS while it evaluates the
// gamma-Poisson class (PRNB-7), whose transcendentals take arguments no table can enumerate.  Ops 0..2 as above, 3
// v_sqrt_f32(x), 4 v_rsq_f32(x), 5 v_cos_f32(x) (x in revolutions).
__global__ void hw_math_at_kernel(int32_t op, const float* __restrict__ x, uint64_t count, float* __restrict__ y)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        float r;
        switch (op) {
        case 0: r = prnb::hw_rcp(v); break;
        case 1: r = prnb::hw_log2(v); break;
        case 2: r = prnb::hw_exp2(-v); break;
        case 3: r = prnb::hw_sqrt(v); break;
        case 4: r = prnb::hw_rsq(v); break;
        default: r = prnb::hw_cos(v); break;
        }
        y[i] = r;
    }
}

// order-preserving map double -> uint64 so that max() can be an integer atomic
__device__ __forceinline__ unsigned long long ordered_bits(double x)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
static double from_ordered_bits(unsigned long long o)
{
    const unsigned long long b = (o >> 63) ? (o & 0x7fffffffffffffffull) : ~o;
    double d;
    memcpy(&d, &b, 8);
    return d;
}

// progs: [0] current programs raw [T][K]; then for each sibling j two blocks, both
// truncated to common_j = min(T, T_j) steps and centred over them:
// current [common_j][K], sibling [common_j][K].   meta: n_sib, then common_j.
// Block = 64 genes x 4 time chunks: wave c covers steps [c*T/4, (c+1)*T/4) of every series for
// the block's 64 genes (x_t = sum_k P[t][k] H[k][g]; program reads are wave-uniform, the H
// column of a gene sits in registers when K <= 32); the chunks' partial sums meet in LDS.
constexpr int kLinChunks = 4;

template <bool HREG>
__global__ __launch_bounds__(256) void lineage_attempt_kernel(
    const double* __restrict__ progs, const int32_t* __restrict__ meta, int32_t T, int32_t K,
    const double* __restrict__ H, int64_t G, unsigned long long* __restrict__ max_bits,
    unsigned long long* __restrict__ anticorr, int64_t attempt_stride, int32_t result_stride)
{
    __shared__ double part[kLinChunks][64][3];
    // attempt blockIdx.y of a batch: its own block of programs and its own result words
    progs += (int64_t)blockIdx.y * attempt_stride;
    max_bits += (int64_t)blockIdx.y * result_stride;
    anticorr += (int64_t)blockIdx.y * result_stride;
    const int gl = threadIdx.x & 63, c = threadIdx.x >> 6;
    const int64_t g = (int64_t)blockIdx.x * 64 + gl;
    const bool live = g < G;
    const double* h = H + (live ? g : 0);
    const int n_sib = meta[0];
    double hreg[HREG ? 32 : 1];
    if (HREG) {
#pragma unroll
        for (int k = 0; k < 32; ++k) hreg[k] = k < K ? h[(int64_t)k * G] : 0.0;
    }
    auto dot = [&](const double* row) -> double {
        double acc = 0.0;
        if (HREG) {
#pragma unroll
            for (int k = 0; k < 32; ++k)
                if (k < K) acc = fma(row[k], hreg[k], acc);
        } else {
            for (int k = 0; k < K; ++k) acc = fma(row[k], h[(int64_t)k * G], acc);
        }
        return acc;
    };

    double mx = -std::numeric_limits<double>::infinity();
    for (int t = (c * T) / kLinChunks; t < ((c + 1) * T) / kLinChunks; ++t) mx = fmax(mx, dot(progs + (int64_t)t * K));
    if (!live) mx = -std::numeric_limits<double>::infinity();
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
    if (gl == 0) atomicMax(max_bits, ordered_bits(mx));

    const double* blk = progs + (int64_t)T * K;
    for (int j = 0; j < n_sib; ++j) {
        const int common = meta[1 + j];
        const double* pc = blk;
        const double* ps = blk + (int64_t)common * K;
        blk += 2 * (int64_t)common * K;
        double cov = 0.0, vx = 0.0, vy = 0.0;
        for (int t = (c * common) / kLinChunks; t < ((c + 1) * common) / kLinChunks; ++t) {
            const double x = dot(pc + (int64_t)t * K), y = dot(ps + (int64_t)t * K);
            cov = fma(x, y, cov);
            vx = fma(x, x, vx);
            vy = fma(y, y, vy);
        }
        part[c][gl][0] = cov; part[c][gl][1] = vx; part[c][gl][2] = vy;
        __syncthreads();
        if (c == 0) {
            for (int o = 1; o < kLinChunks; ++o) { cov += part[o][gl][0]; vx += part[o][gl][1]; vy += part[o][gl][2]; }
            // Pearson r < 0  <=>  cov < 0 with both series non-constant (scipy returns NaN otherwise)
            const bool neg = live && cov < 0.0 && vx > 0.0 && vy > 0.0;
            const unsigned long long votes = __ballot(neg);
            if (gl == 0 && votes) atomicAdd(&anticorr[j], (unsigned long long)__popcll(votes));
        }
        __syncthreads();
    }
}

// The same for program matrices that fit the block's LDS (every benchmark configuration: K <= 32 and
// max(T, 2 * common_j) * K <= kAttLdsDoubles): the attempt's programs are staged in LDS once per block (the kernel
// above re-reads every program row from global memory for every dot product and runs at a tenth of the binary64
// FMA rate), and every thread carries TWO genes, so that one broadcast LDS read of a program entry feeds two FMAs --
// the ratio at which the LDS (256 B/clk/CU) keeps up with the four SIMDs' binary64 pipes.  Same operations in the same
// order per gene as the kernel above: identical maxima and counts.
constexpr int kAttLdsDoubles = 6144;      // 48 KB
constexpr int kAttGenes = 128;            // genes per block: 64 lanes x 2

__global__ __launch_bounds__(256) void lineage_attempt_lds_kernel(
    const double* __restrict__ progs, const int32_t* __restrict__ meta, int32_t T, int32_t K,
    const double* __restrict__ H, int64_t G, unsigned long long* __restrict__ max_bits,
    unsigned long long* __restrict__ anticorr, int64_t attempt_stride, int32_t result_stride)
{
    extern __shared__ double buf[];                // max(T, 2 * common_j) * K doubles: sized by the launch
    __shared__ double part[kLinChunks][kAttGenes][3];
    progs += (int64_t)blockIdx.y * attempt_stride;
    max_bits += (int64_t)blockIdx.y * result_stride;
    anticorr += (int64_t)blockIdx.y * result_stride;
    const int gl = threadIdx.x & 63, c = threadIdx.x >> 6;
    const int64_t g0 = (int64_t)blockIdx.x * kAttGenes + gl, g1 = g0 + 64;
    const bool live0 = g0 < G, live1 = g1 < G;
    const double* h0 = H + (live0 ? g0 : 0);
    const double* h1 = H + (live1 ? g1 : 0);
    const int n_sib = meta[0];
    double a0[32], a1[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        a0[k] = k < K ? h0[(int64_t)k * G] : 0.0;
        a1[k] = k < K ? h1[(int64_t)k * G] : 0.0;
    }
    auto dot2 = [&](const double* row, double& x0, double& x1) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k)
            if (k < K) {
                const double p = row[k];
                s0 = fma(p, a0[k], s0);
                s1 = fma(p, a1[k], s1);
            }
        x0 = s0;
        x1 = s1;
    };
    auto stage = [&](const double* src, int doubles) {
        __syncthreads();                               // (the previous phase has finished with buf)
        for (int i = threadIdx.x; i < doubles; i += 256) buf[i] = src[i];
        __syncthreads();
    };

    stage(progs, T * K);
    double mx = -std::numeric_limits<double>::infinity();
    for (int t = (c * T) / kLinChunks; t < ((c + 1) * T) / kLinChunks; ++t) {
        double x0, x1;
        dot2(buf + t * K, x0, x1);
        if (live0) mx = fmax(mx, x0);
        if (live1) mx = fmax(mx, x1);
    }
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
    if (gl == 0) atomicMax(max_bits, ordered_bits(mx));

    const double* blk = progs + (int64_t)T * K;
    for (int j = 0; j < n_sib; ++j) {
        const int common = meta[1 + j];
        stage(blk, 2 * common * K);
        blk += 2 * (int64_t)common * K;
        const double* pc = buf;
        const double* ps = buf + common * K;
        double cov0 = 0.0, vx0 = 0.0, vy0 = 0.0, cov1 = 0.0, vx1 = 0.0, vy1 = 0.0;
        for (int t = (c * common) / kLinChunks; t < ((c + 1) * common) / kLinChunks; ++t) {
            double x0, x1, y0, y1;
            dot2(pc + t * K, x0, x1);
            dot2(ps + t * K, y0, y1);
            cov0 = fma(x0, y0, cov0); vx0 = fma(x0, x0, vx0); vy0 = fma(y0, y0, vy0);
            cov1 = fma(x1, y1, cov1); vx1 = fma(x1, x1, vx1); vy1 = fma(y1, y1, vy1);
        }
        part[c][gl][0] = cov0; part[c][gl][1] = vx0; part[c][gl][2] = vy0;
        part[c][gl + 64][0] = cov1; part[c][gl + 64][1] = vx1; part[c][gl + 64][2] = vy1;
        __syncthreads();
        if (c < 2) {                                   // wave 0 sums up the block's first 64 genes, wave 1 the other 64
            const int gi = gl + 64 * c;
            double cov = part[0][gi][0], vx = part[0][gi][1], vy = part[0][gi][2];
            for (int o = 1; o < kLinChunks; ++o) { cov += part[o][gi][0]; vx += part[o][gi][1]; vy += part[o][gi][2]; }
            // Pearson r < 0  <=>  cov < 0 with both series non-constant (scipy returns NaN otherwise)
            const bool neg = (c == 0 ? live0 : live1) && cov < 0.0 && vx > 0.0 && vy > 0.0;
            const unsigned long long votes = __ballot(neg);
            if (gl == 0 && votes) atomicAdd(&anticorr[j], (unsigned long long)__popcll(votes));
        }
    }
}

// K1: one lane per expression program, T sequential steps (walk definition PRLW-1)
__global__ void lineage_walk_kernel(uint32_t k0, uint32_t k1, uint32_t sid_lo, uint32_t sid_hi, int32_t T,
                                    int32_t K, double* out)
{
    const int32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    // walk stream blockIdx.y of a batch: consecutive stream ids, consecutive (T, K) blocks of the output
    const uint64_t sid = (((uint64_t)sid_hi << 32) | sid_lo) + blockIdx.y;
    sid_lo = (uint32_t)sid;
    sid_hi = (uint32_t)(sid >> 32);
    out += (int64_t)blockIdx.y * T * K;
    prnb::Words w = prnb::philox4x32_10((uint32_t)k, 0u, sid_lo, sid_hi, k0, k1);
    double walk = (double)prnb::det_log(1.5f * prnb::unif(w.w[0]));
    double vel = 0.2 * (double)(prnb::det_sqrt(-2.0f * prnb::det_log(prnb::unif(w.w[1]))) * prnb::det_cos2pi(w.w[2]));
    const double eta = (double)prnb::unif(w.w[3]);
    const double s_eps = 2.0 / (double)T;
    out[k] = walk;
    for (int32_t t = 0; t + 1 < T; ++t) {
        if ((t & 1) == 0) w = prnb::philox4x32_10((uint32_t)k, 1u + (uint32_t)(t >> 1), sid_lo, sid_hi, k0, k1);
        const uint32_t wa = (t & 1) ? w.w[2] : w.w[0], wb = (t & 1) ? w.w[3] : w.w[1];
        const double eps = s_eps * (double)(prnb::det_sqrt(-2.0f * prnb::det_log(prnb::unif(wa))) * prnb::det_cos2pi(wb));
        walk = walk + vel;
        vel = eta * vel + eps;
        out[(int64_t)(t + 1) * K + k] = walk;
    }
}

// max into a binary64 cell shared by a few blocks (compare-and-swap on the bits; NaN never wins)
__device__ __forceinline__ void atomic_max_f64(double* cell, double v)
{
    unsigned long long* bits = reinterpret_cast<unsigned long long*>(cell);
    unsigned long long seen = *bits;
    while (v > __longlong_as_double((long long)seen)) {
        const unsigned long long was = atomicCAS(bits, seen, (unsigned long long)__double_as_longlong(v));
        if (was == seen) break;
        seen = was;
    }
}

// K2b: rel[t][g] = sum_k progs[t][k] * H[k][g] for an accepted branch, and its per-gene maximum folded into gene_max.
// Block = 64 genes x 4 groups of time steps, blockIdx.y = one of gridDim.y ranges of the T steps (the host picks them
// so that the launch has >= 1024 blocks); the range's programs sit in LDS (wave-uniform reads), a gene's H column in
// registers when K <= 32.  The dot product runs over k in ascending order by fma, as the attempt kernel's does.
constexpr int kCommitLdsDoubles = 2048;     // programs of one block's time range: steps * K <= this (the host sizes gridDim.y)

template <bool HREG>
__global__ __launch_bounds__(256) void lineage_commit_kernel(const double* __restrict__ progs,
                                                             int32_t T, int32_t K,
                                                             const double* __restrict__ H, int64_t G,
                                                             double* __restrict__ rel_out,
                                                             double* __restrict__ gene_max)
{
    __shared__ double pl[kCommitLdsDoubles];
    __shared__ double part[4][64];
    const int gl = threadIdx.x & 63, c = threadIdx.x >> 6;
    const int64_t g = (int64_t)blockIdx.x * 64 + gl;
    const bool live = g < G;
    const int t0 = (int)(((int64_t)blockIdx.y * T) / gridDim.y), t1 = (int)(((int64_t)(blockIdx.y + 1) * T) / gridDim.y);
    for (int i = threadIdx.x; i < (t1 - t0) * K; i += 256) pl[i] = progs[(int64_t)t0 * K + i];
    const double* h = H + (live ? g : 0);
    double hreg[HREG ? 32 : 1];
    if (HREG) {
#pragma unroll
        for (int k = 0; k < 32; ++k) hreg[k] = k < K ? h[(int64_t)k * G] : 0.0;
    }
    __syncthreads();
    double mx = -std::numeric_limits<double>::infinity();
    for (int t = t0 + c; t < t1; t += 4) {
        const double* row = pl + (t - t0) * K;
        double acc = 0.0;
        if (HREG) {
#pragma unroll
            for (int k = 0; k < 32; ++k)
                if (k < K) acc = fma(row[k], hreg[k], acc);
        } else {
            for (int k = 0; k < K; ++k) acc = fma(row[k], h[(int64_t)k * G], acc);
        }
        if (live && rel_out) rel_out[(int64_t)t * G + g] = acc;
        mx = fmax(mx, acc);
    }
    if (!gene_max) return;
    part[c][gl] = mx;
    __syncthreads();
    if (c == 0 && live) atomic_max_f64(&gene_max[g], fmax(fmax(part[0][gl], part[1][gl]), fmax(part[2][gl], part[3][gl])));
}

// gene_max[g] = max(gene_max[g], max over rows of rel[row][g]): block = 64 genes x 4 groups of rows, blockIdx.y = a range of rows
__global__ __launch_bounds__(256) void gene_max_kernel(const double* __restrict__ rel, int64_t rows,
                                                       int64_t G, double* __restrict__ gene_max)
{
    __shared__ double part[4][64];
    const int gl = threadIdx.x & 63, c = threadIdx.x >> 6;
    const int64_t g = (int64_t)blockIdx.x * 64 + gl;
    const int64_t r0 = ((int64_t)blockIdx.y * rows) / gridDim.y, r1 = ((int64_t)(blockIdx.y + 1) * rows) / gridDim.y;
    double mx = -std::numeric_limits<double>::infinity();
    if (g < G)
        for (int64_t t = r0 + c; t < r1; t += 4) mx = fmax(mx, rel[t * G + g]);
    part[c][gl] = mx;
    __syncthreads();
    if (c == 0 && g < G) atomic_max_f64(&gene_max[g], fmax(fmax(part[0][gl], part[1][gl]), fmax(part[2][gl], part[3][gl])));
}

__global__ __launch_bounds__(256) void means_from_rel_kernel(const double* __restrict__ rel,
                                                             const double* __restrict__ base,
                                                             int64_t rows, int64_t G,
                                                             float* __restrict__ out)
{
    const int64_t total = rows * G;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t g = i % G;
        const double v = exp(rel[i]) * base[g];
        // a positive binary64 mean must stay positive in binary32: the reference never sees the zero
        // (and the ValueError that goes with it) that an underflow below 1e-38 would fake
        out[i] = (v > 0.0 && v < 1.17549435e-38) ? 1.17549435e-38f : (float)v;
    }
}

// ---- domain check of the streaming path (PROSSTT_AMD_CHECK_DOMAIN / _CHECK_DEFERRED; see prep_kernel) ----------
// row_bad[r] = 1 iff row r of the mean tensor has an entry that is not > 0 (zero, negative, NaN): one block per row.
// Scanned once per mean tensor (the ctx remembers which tensor its flags belong to).
__global__ __launch_bounds__(256) void row_flags_kernel(const float* __restrict__ means, int64_t rows, int64_t G,
                                                        uint8_t* __restrict__ row_bad)
{
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const float* row = means + r * G;
        int bad = 0;
        for (int64_t g = threadIdx.x; g < G; g += blockDim.x) bad |= !(row[g] > 0.0f);
        bad = __syncthreads_or(bad);
        if (threadIdx.x == 0) row_bad[r] = (uint8_t)(bad != 0);
    }
}

// ------------------------------------------------------------------ ABI

PA_EXPORT int prosstt_amd_version(void) { return PROSSTT_AMD_VERSION; }
PA_EXPORT const char* prosstt_amd_last_error(void) { return g_err; }

PA_EXPORT int prosstt_amd_device_count(int* count) try
{
    if (!count) return fail(PROSSTT_AMD_EINVAL, "count is NULL");
    *count = 0;
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) {
        *count = 0;
        return fail(PROSSTT_AMD_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    return 0;
}
PA_CATCH

PA_EXPORT int prosstt_amd_ctx_create(int device, void* stream, prosstt_amd_ctx** out) try
{
    if (!out) return fail(PROSSTT_AMD_EINVAL, "out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(PROSSTT_AMD_ENODEV, "no HIP device visible: the prosstt_amd path has no CPU fallback");
    if (device < 0 || device >= n) return fail(PROSSTT_AMD_EINVAL, "device %d out of range [0,%d)", device, n);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(PROSSTT_AMD_ENODEV, "device %d is %s; this library is built for gfx950 only", device,
                    prop.gcnArchName);
    HIP_TRY(hipSetDevice(device));
    prosstt_amd_ctx* c = new (std::nothrow) prosstt_amd_ctx();
    if (!c) return fail(PROSSTT_AMD_ENOMEM, "out of host memory");
    c->device = device;
    c->stream = (hipStream_t)stream;   // NULL = the device's default stream
    if (hipMalloc((void**)&c->scratch, kScratchWords * 8) != hipSuccess ||
        hipHostMalloc((void**)&c->h_scratch, kScratchWords * 8) != hipSuccess) {
        prosstt_amd_ctx_destroy(c);
        return fail(PROSSTT_AMD_EHIP, "ctx allocation failed");
    }
    if (hipMemset(c->scratch, 0, kScratchWords * 8) != hipSuccess) {
        prosstt_amd_ctx_destroy(c);
        return fail(PROSSTT_AMD_EHIP, "ctx initialisation failed");
    }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k3::sample_counts_heavy_kernel, k3::kHeavyBlock, 0) == hipSuccess &&
        per_cu > 0 && prop.multiProcessorCount > 0)
        c->heavy_grid = per_cu * prop.multiProcessorCount;
    *out = c;
    return 0;
}
PA_CATCH

PA_EXPORT int prosstt_amd_ctx_destroy(prosstt_amd_ctx* c) try
{
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->ws) (void)hipFree(c->ws);
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->row_bad) (void)hipFree(c->row_bad);
    if (c->h_scratch) (void)hipHostFree(c->h_scratch);
    for (hipEvent_t e : c->events) (void)hipEventDestroy(e);
    delete c;
    return 0;
}
PA_CATCH

PA_EXPORT int prosstt_amd_ctx_synchronize(prosstt_amd_ctx* c) try
{
    if (!c) return fail(PROSSTT_AMD_EINVAL, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}
PA_CATCH

PA_EXPORT int prosstt_amd_last_kernel_ms(prosstt_amd_ctx* c, float* ms) try
{
    if (!c || !ms) return fail(PROSSTT_AMD_EINVAL, "NULL argument");
    if (c->events_used == 0)
        return fail(PROSSTT_AMD_EINVAL, "no kernel was launched with PROSSTT_AMD_TIME_KERNEL since the last call");
    HIP_TRY(hipEventSynchronize(c->events[c->events_used - 1]));
    double total = 0.0;
    for (size_t i = 0; i < c->events_used; i += 2) {
        float one = 0.0f;
        HIP_TRY(hipEventElapsedTime(&one, c->events[i], c->events[i + 1]));
        total += one;
    }
    *ms = (float)(total / (double)(c->events_used / 2));
    c->events_used = 0;
    return 0;
}
PA_CATCH

// Common front end of sample_counts / nb_params: validates, stages host inputs,
// converts the binary64 per-cell / per-gene parameters into the workspace.
struct SamplerArgs {
    const float* means; const int32_t* row_of_cell;
    const int64_t* cell_index;   // device copy (NULL: cells are numbered from cell_offset)
    float *scal, *ga, *gbm1, *gphi;
    void* extra;     // `extra_bytes` of workspace behind the parameter vectors (256-B aligned)
    k3::HeavyList heavy;           // K3h's lists (a sample_counts call) ...
    k3::HeavyList* heavy_rec;      // ... and where the preparation kernel puts their description for the kernels behind it
    k3::CellInfo* cellinfo;
};

// Geometry of a streaming-kernel launch (k3_stream.h): strips of 64 cells per wave (the kernel takes up
// to 128; shorter ones when the problem is too small to give every SIMD of the chip a few waves), four
// strips and one 256-gene tile per block.
struct StreamGeometry {
    int64_t tiles_g, strip_cells, strips, groups;
    uint64_t regions;       // one region of the K3h list per wave
    uint32_t region_cap;    // room for one in 16 of a wave's samples (beyond that K3h redoes the region itself)
    uint32_t ent_cap;          // entries per segment of the dense lists K3h reads (k3::HeavyList)
    size_t list_bytes, cnt_bytes, dense_bytes, count_bytes, ent_bytes, wst_bytes, wid_bytes, ovf_bytes, redo_bytes, info_bytes;
    size_t total() const { return list_bytes + cnt_bytes + dense_bytes + count_bytes + ent_bytes + wst_bytes + wid_bytes + ovf_bytes + redo_bytes + info_bytes + 256; }
};

static StreamGeometry stream_geometry(int64_t N, int32_t G, int64_t rows)
{
    StreamGeometry g;
    g.tiles_g = ((int64_t)(G > 0 ? G : 0) + kTileG - 1) / kTileG;
    g.strip_cells = k3::kStripCells / 2;    // 64: measured best on C3 (128: +1.7 %, 32: +2.7 %)
    const int64_t n = N > 0 ? N : 0;
    while (g.strip_cells > 8 && ((n + g.strip_cells - 1) / g.strip_cells) * g.tiles_g < 4 * 5 * 1024) g.strip_cells /= 2;
    g.strips = (n + g.strip_cells - 1) / g.strip_cells;
    g.groups = (g.strips + 3) / 4;
    g.regions = (uint64_t)(g.groups * g.tiles_g) * 4u;
    g.region_cap = (uint32_t)g.strip_cells * (kTileG / 16);
    if (g.region_cap > 1024u) g.region_cap = 1024u;             // K3h's second phase takes kDense + 16 chunks of 64
    auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
    g.list_bytes = pad(g.regions * ((size_t)g.region_cap + k3::kListTail) * sizeof(uint2));      // staging: {pos, scaled mean} per region (+ its tail)
    g.cnt_bytes = pad(768 + sizeof(k3::HeavyList));      // the segments' counters, and (at 768) the lists' description
    static_assert((k3::kSegs + 2) * 4 <= 768, "the counters end where the record starts");
    g.dense_bytes = pad(g.regions * (size_t)k3::kDense * sizeof(uint2));      // a region's first entries
    g.count_bytes = pad(g.regions * 4u);
    // the segments (what regions list beyond their first kDense entries) take one sample in 64 of the matrix between them
    // (typical workloads list one to three in a thousand); a region that finds its segment full is redone as a whole, like
    // one whose own list was too small
    const uint64_t per_seg = ((uint64_t)n * (uint64_t)(G > 0 ? G : 0) / 64u + k3::kSegs - 1) / k3::kSegs;
    g.ent_cap = (uint32_t)(per_seg < 4u * g.region_cap ? 4u * g.region_cap : (per_seg > 0x7fffffffull ? 0x7fffffffull : per_seg));
    g.ent_bytes = pad((size_t)k3::kSegs * g.ent_cap * sizeof(k3::HeavyEntry));
    g.wst_bytes = pad(g.regions * (size_t)k3::kWalkSlots * 16u);            // walk states handed over
    g.wid_bytes = pad(g.regions * (size_t)k3::kWalkSlots * 4u);
    g.ovf_bytes = pad(g.regions * 4u);
    g.redo_bytes = pad((size_t)k3::kRedoCap * sizeof(int2));
    g.info_bytes = ((size_t)n + 4) * sizeof(k3::CellInfo);
    return g;
}

static int sampler_setup(prosstt_amd_ctx* c, Staging& st, const float* means, int64_t rows, int32_t G,
                         const int32_t* row_of_cell, const double* scaling, const double* alpha,
                         const double* beta, int64_t N, uint32_t flags, SamplerArgs* A,
                         const StreamGeometry* geo, uint64_t cell_offset, const int64_t* cell_index, uint64_t seed)
{
    if (!c) return fail(PROSSTT_AMD_EINVAL, "ctx is NULL");
    if (N < 0 || G < 0 || rows < 0) return fail(PROSSTT_AMD_EINVAL, "negative size");
    if (N > 0 && G > 0 && (!means || !row_of_cell || !scaling || !alpha || !beta))
        return fail(PROSSTT_AMD_EINVAL, "NULL input array");
    if (N > 0 && G > 0 && rows == 0) return fail(PROSSTT_AMD_EINVAL, "the mean tensor has no rows");
    if ((int64_t)G * 4 > (int64_t)1 << 33) return fail(PROSSTT_AMD_EINVAL, "G too large");
    HIP_TRY(hipSetDevice(c->device));
    if (N == 0 || G == 0) return 0;
    if (flags & PROSSTT_AMD_HOST_INPUTS) {
        // host-side bounds check is free here; device pointers are the caller's contract
        for (int64_t n = 0; n < N; ++n)
            if (row_of_cell[n] < 0 || row_of_cell[n] >= rows)
                return fail(PROSSTT_AMD_EINVAL, "row_of_cell[%lld] = %d outside [0,%lld)", (long long)n,
                            row_of_cell[n], (long long)rows);
        int rc;
        const void* d;
        if ((rc = st.upload(means, (size_t)rows * G * 4, &d, c->stream))) return rc;
        means = (const float*)d;
        if ((rc = st.upload(row_of_cell, (size_t)N * 4, &d, c->stream))) return rc;
        row_of_cell = (const int32_t*)d;
        if ((rc = st.upload(scaling, (size_t)N * 8, &d, c->stream))) return rc;
        scaling = (const double*)d;
        if ((rc = st.upload(alpha, (size_t)G * 8, &d, c->stream))) return rc;
        alpha = (const double*)d;
        if ((rc = st.upload(beta, (size_t)G * 8, &d, c->stream))) return rc;
        beta = (const double*)d;
        if (cell_index) {
            if ((rc = st.upload(cell_index, (size_t)N * 8, &d, c->stream))) return rc;
            cell_index = (const int64_t*)d;
        }
    }
    const int64_t n_pad = (N + 15) & ~(int64_t)15;
    const size_t vec_bytes = ((((size_t)n_pad + 3 * (size_t)G) * sizeof(float)) + 255) & ~(size_t)255;
    int rc = ws_reserve(c, vec_bytes + (geo ? geo->total() : 0));
    if (rc) return rc;
    A->means = means;
    A->row_of_cell = row_of_cell;
    A->cell_index = cell_index;
    A->scal = (float*)c->ws;
    A->ga = A->scal + n_pad;
    A->gbm1 = A->ga + G;
    A->gphi = A->gbm1 + G;
    A->extra = (char*)c->ws + vec_bytes;
    k3::CellInfo* info = nullptr;
    A->heavy = k3::HeavyList{};
    A->heavy_rec = nullptr;
    if (geo) {
        char* at = (char*)A->extra;
        k3::HeavyList& heavy = A->heavy;
        heavy.list = (uint2*)at; at += geo->list_bytes;
        heavy.seg_cnt = (uint32_t*)at;                                // zeroed by the preparation kernel
        A->heavy_rec = (k3::HeavyList*)(at + 768); at += geo->cnt_bytes;
        heavy.dense = (uint2*)at; at += geo->dense_bytes;
        heavy.count = (uint32_t*)at; at += geo->count_bytes;
        heavy.ent = (k3::HeavyEntry*)at; at += geo->ent_bytes;
        heavy.wst = (k3::f32x4_t*)at; at += geo->wst_bytes;
        heavy.wid = (uint32_t*)at; at += geo->wid_bytes;
        heavy.ovf_regions = (uint32_t*)at; at += geo->ovf_bytes;
        heavy.redo = (int2*)at; at += geo->redo_bytes;
        info = (k3::CellInfo*)at;
        heavy.cap = geo->region_cap;
        heavy.ent_cap = geo->ent_cap;
        heavy.overflow = (uint32_t*)(c->scratch + 3);           // zeroed by the preparation kernel
    }
    A->cellinfo = info;
    // a checked call needs the per-row flags of THIS mean tensor: scanned now unless the caller vouches that the tensor
    // the ctx last scanned (same pointer, same shape) has not changed since
    const uint8_t* row_bad = nullptr;
    if (geo && (flags & (PROSSTT_AMD_CHECK_DOMAIN | PROSSTT_AMD_CHECK_DEFERRED))) {
        const bool cached = (flags & PROSSTT_AMD_MEANS_CACHED) && !(flags & PROSSTT_AMD_HOST_INPUTS) && c->row_bad &&
                            c->row_bad_means == means && c->row_bad_rows == rows && c->row_bad_G == G;
        if (!cached) {
            if ((size_t)rows > c->row_bad_cap) {
                HIP_TRY(hipStreamSynchronize(c->stream));
                if (c->row_bad) HIP_TRY(hipFree(c->row_bad));
                c->row_bad = nullptr;
                c->row_bad_cap = 0;
                c->row_bad_means = nullptr;
                HIP_TRY(hipMalloc((void**)&c->row_bad, ((size_t)rows + 4095) & ~(size_t)4095));
                c->row_bad_cap = ((size_t)rows + 4095) & ~(size_t)4095;
            }
            row_flags_kernel<<<dim3((unsigned)(rows < 65536 ? rows : 65536)), dim3(256), 0, c->stream>>>(means, rows, G, c->row_bad);
            HIP_TRY(hipGetLastError());
            // (a staged copy of host inputs dies with the call: its flags are never reused)
            c->row_bad_means = (flags & PROSSTT_AMD_HOST_INPUTS) ? nullptr : means;
            c->row_bad_rows = rows;
            c->row_bad_G = G;
        }
        row_bad = c->row_bad;
    }
    const int64_t span = (N + 4 > G ? N + 4 : G);
    prep_kernel<<<dim3((unsigned)((span + 255) / 256)), dim3(256), 0, c->stream>>>(
        scaling, N, alpha, beta, G, A->scal, A->ga, A->gbm1, A->gphi, row_of_cell, rows, cell_offset, cell_index,
        geo ? (int32_t)geo->strip_cells : 1, (uint32_t)seed, (uint32_t)(seed >> 32), info, geo ? c->scratch : nullptr,
        row_bad, c->call_parity, A->heavy, A->heavy_rec);
    HIP_TRY(hipGetLastError());
    return 0;
}

// Reads and clears the sticky verdict of the checked calls since the last time (synchronises the stream).
static int domain_verdict(prosstt_amd_ctx* c, int64_t rows, int* verdict)
{
    HIP_TRY(hipMemcpyAsync(c->h_scratch + kStickyDomain, c->scratch + kStickyDomain, 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemsetAsync(c->scratch + kStickyDomain, 0, 16, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *verdict = 0;
    if (c->h_scratch[kStickyRow]) {
        *verdict = PROSSTT_AMD_EINVAL;
        if (rows >= 0) return fail(PROSSTT_AMD_EINVAL, "row_of_cell holds an index outside [0,%lld)", (long long)rows);
        return fail(PROSSTT_AMD_EINVAL, "row_of_cell of a checked call held an index outside the mean tensor");
    }
    if (c->h_scratch[kStickyDomain]) {
        *verdict = PROSSTT_AMD_EDOMAIN;
        return fail(PROSSTT_AMD_EDOMAIN, "Domain error in arguments: a mean <= 0 or alpha*m + beta < 1");
    }
    return 0;
}

PA_EXPORT int prosstt_amd_domain_status(prosstt_amd_ctx* c, int32_t* status) try
{
    if (!c) return fail(PROSSTT_AMD_EINVAL, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    int verdict = 0;
    const int rc = domain_verdict(c, -1, &verdict);
    if (status) *status = verdict;
    return (rc == PROSSTT_AMD_EHIP) ? rc : 0;     // the verdict travels in *status; the message is in last_error
}
PA_CATCH

PA_EXPORT int prosstt_amd_plan_order(const int32_t* row_of_cell, int64_t N, int64_t rows, int32_t* order) try
{
    if (N < 0 || rows < 0 || N > 0x7fffffffll) return fail(PROSSTT_AMD_EINVAL, "bad size");
    if (N > 0 && (!row_of_cell || !order)) return fail(PROSSTT_AMD_EINVAL, "NULL argument");
    std::vector<int64_t> first((size_t)rows + 1, 0);
    for (int64_t n = 0; n < N; ++n) {
        if (row_of_cell[n] < 0 || row_of_cell[n] >= rows)
            return fail(PROSSTT_AMD_EINVAL, "row_of_cell[%lld] = %d outside [0,%lld)", (long long)n, row_of_cell[n], (long long)rows);
        first[(size_t)row_of_cell[n] + 1] += 1;
    }
    for (int64_t r = 0; r < rows; ++r) first[(size_t)r + 1] += first[(size_t)r];
    for (int64_t n = 0; n < N; ++n) order[first[(size_t)row_of_cell[n]]++] = (int32_t)n;
    return 0;
}
PA_CATCH

PA_EXPORT int prosstt_amd_sample_counts(prosstt_amd_ctx* c, const float* means, int64_t rows, int32_t G,
                                        const int32_t* row_of_cell, const double* scaling,
                                        const double* alpha, const double* beta, int64_t N,
                                        uint64_t seed, uint64_t cell_offset, const int64_t* cell_index,
                                        int32_t* out, int64_t ld_out, uint32_t flags) try
{
    Staging st;
    SamplerArgs A{};
    // everything that can be refused is refused before the workspace grows
    if (N > 0 && G > 0) {
        if (!out) return fail(PROSSTT_AMD_EINVAL, "out is NULL");
        if (ld_out < G) return fail(PROSSTT_AMD_EINVAL, "ld_out %lld < G %d", (long long)ld_out, G);
        if (ld_out * k3::kStripCells >= ((int64_t)1 << 29))
            return fail(PROSSTT_AMD_EINVAL, "ld_out too large: a strip's rows must span less than 2 GiB");
        if (N > 0x7fffffffll) return fail(PROSSTT_AMD_EINVAL, "too many cells; chunk them");
        if ((uint64_t)(rows > 0 ? rows : 0) * (uint64_t)G >= ((uint64_t)1 << 61)) return fail(PROSSTT_AMD_EINVAL, "mean tensor too large");
    }
    const StreamGeometry geo = stream_geometry(N, G, rows);
    if (geo.groups * geo.tiles_g > 0x1fffffffll) return fail(PROSSTT_AMD_EINVAL, "too many tiles; chunk the cells");
    int rc = sampler_setup(c, st, means, rows, G, row_of_cell, scaling, alpha, beta, N, flags, &A, &geo, cell_offset,
                           cell_index, seed);
    if (rc) return rc;
    if (N == 0 || G == 0) return 0;
    const k3::HeavyList heavy = A.heavy;
    k3::CellInfo* cellinfo = A.cellinfo;
    const int64_t* d_cell_index = A.cell_index;
    int32_t* d_out = out;
    if (flags & PROSSTT_AMD_HOST_OUTPUT) {
        void* p = nullptr;
        if ((rc = st.alloc(&p, (size_t)N * ld_out * 4))) return rc;
        d_out = (int32_t*)p;
    }

    const bool vec = (G % 4 == 0) && (ld_out % 4 == 0) && (((uintptr_t)A.means & 15) == 0) &&
                     (((uintptr_t)d_out & 15) == 0);
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const bool checked = (flags & (PROSSTT_AMD_CHECK_DOMAIN | PROSSTT_AMD_CHECK_DEFERRED)) != 0;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    if (flags & PROSSTT_AMD_TIME_KERNEL) {
        if ((rc = next_event_pair(c, &ev_start, &ev_stop))) return rc;
    }
    const dim3 grid((unsigned)(geo.groups * geo.tiles_g)), block(k3::kBlock);
    // full-length strips and more counts than the last-level cache (256 MB) takes: k3_stream.h, BIG
    const bool big = geo.strip_cells >= k3::kStripCells / 2 && (double)N * (double)ld_out * 4.0 >= 1073741824.0;
    // The dominant kernel is timed alone (PROSSTT_AMD_TIME_KERNEL): the two events ride on the kernel's OWN dispatch packet
    // (hipExtLaunchKernelGGL: its start and end timestamps), not as records in front of and behind it -- two more packets
    // on the queue, each a barrier: 6 us of gap on either side of the kernel (tools/gap_trace.py), which would be in
    // every timed step of bench.py.
#define K3_LAUNCH(V, B)                                                                                          \
    hipExtLaunchKernelGGL((k3::sample_counts_stream_kernel<V, B>), grid, block, 0, c->stream, ev_start, ev_stop, 0,  \
        A.means, G, (const k3::CellInfo*)cellinfo, (const float*)A.ga, (const float*)A.gbm1, (const float*)A.gphi, N, k0, k1, \
        d_out, ld_out, (int32_t)geo.strips, (int32_t)geo.strip_cells, heavy.list, heavy.cap, heavy.dense, (const k3::HeavyList*)A.heavy_rec)
    if (vec && big) K3_LAUNCH(true, true);
    else if (vec) K3_LAUNCH(true, false);
    else if (big) K3_LAUNCH(false, true);
    else K3_LAUNCH(false, false);
#undef K3_LAUNCH
    HIP_TRY(hipGetLastError());
    // every wave takes whole regions of the list, four per step; as many blocks as the device holds at once (6 per CU: 1536 --
    // 1024, 2048 and 3072 blocks are 7-18 us slower at C3), fewer when the list has fewer than a step of regions per wave
    // (C2's 12 500 regions: 782 blocks, -3 % of the call)
    const uint64_t region_blocks = (geo.regions + 15u) / 16u;
    const unsigned heavy_blocks = (unsigned)(region_blocks < 256u ? 256u : (region_blocks < (uint64_t)c->heavy_grid ? region_blocks : (uint64_t)c->heavy_grid));
    k3::sample_counts_heavy_kernel<<<dim3(heavy_blocks), dim3(k3::kHeavyBlock), 0, c->stream>>>(
        (const k3::HeavyList*)A.heavy_rec, (uint32_t)geo.regions, (int32_t)geo.strips, (int32_t)geo.strip_cells, A.means, rows, G, A.row_of_cell,
        A.scal, A.ga, A.gbm1, N, k0, k1, cell_offset, d_cell_index, d_out, ld_out,
        // the per-sample part of a checked call's domain test rides at the end of K3h (it leaves at once unless a gene has
        // alpha < 0 or beta < 1: prep_kernel's request word of this call's parity)
        (checked && !(flags & PROSSTT_AMD_PARAMS_NONNEG)) ? c->scratch : nullptr, (uint32_t)kFullReq + c->call_parity,
        (uint32_t)kStickyRow, (uint32_t)kStickyDomain);
    HIP_TRY(hipGetLastError());
    c->list = heavy;
    c->list_regions = geo.regions;
    c->list_groups = geo.groups;
    c->list_strip_cells = geo.strip_cells;
    c->call_parity ^= 1u;
    if (flags & PROSSTT_AMD_HOST_OUTPUT)   // G columns of every row; the caller's padding beyond G is left alone
        HIP_TRY(hipMemcpy2DAsync(out, (size_t)ld_out * 4, d_out, (size_t)ld_out * 4, (size_t)G * 4, (size_t)N,
                                 hipMemcpyDeviceToHost, c->stream));
    if (flags & PROSSTT_AMD_CHECK_DOMAIN) {
        int verdict = 0;
        rc = domain_verdict(c, rows, &verdict);
        if (rc) return rc;
    } else if (flags & (PROSSTT_AMD_HOST_OUTPUT | PROSSTT_AMD_HOST_INPUTS)) {
        HIP_TRY(hipStreamSynchronize(c->stream));   // staging buffers die with `st`
    }
    return 0;
}
PA_CATCH

// The samples the streaming kernel of the LAST sample_counts call on this ctx left to K3h (the gamma-Poisson
// class, walks too close to a threshold, counts above 255), decoded to (cell, gene) pairs: `cells[i]` is the
// cell's index in that call's arrays.  At most `cap` pairs are written; `*total` receives the number listed
// (of a region that was too small: the entries that fitted) and `*overflowed` whether there was such a region
// (K3h redoes those regions sample by sample).  Valid until the next call on the ctx grows its workspace.
PA_EXPORT int prosstt_amd_last_list(prosstt_amd_ctx* c, int64_t* cells, int32_t* genes, int64_t cap,
                                    int64_t* total, int32_t* overflowed) try
{
    if (!c || !total) return fail(PROSSTT_AMD_EINVAL, "NULL argument");
    if (cap > 0 && (!cells || !genes)) return fail(PROSSTT_AMD_EINVAL, "NULL output array");
    *total = 0;
    if (overflowed) *overflowed = 0;
    if (!c->list.seg_cnt) return 0;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    int64_t flagw[4];
    HIP_TRY(hipMemcpy(flagw, c->scratch, sizeof(flagw), hipMemcpyDeviceToHost));
    if (overflowed) *overflowed = (int32_t)((uint32_t)flagw[3] != 0u);
    int64_t written = 0;
    // per region: the first entries and the walks
    std::vector<uint32_t> counts(c->list_regions);
    HIP_TRY(hipMemcpy(counts.data(), c->list.count, counts.size() * 4, hipMemcpyDeviceToHost));
    uint2 dense[k3::kDense];
    uint32_t wid[k3::kWalkSlots];
    for (uint64_t r = 0; r < c->list_regions; ++r) {
        const uint32_t listed = counts[r] & 0xffffu, walks = counts[r] >> 16;
        const uint32_t nd = listed > c->list.cap ? 0u : (listed < (uint32_t)k3::kDense ? listed : (uint32_t)k3::kDense);
        *total += nd + walks;
        if (nd + walks == 0 || written >= cap) continue;
        if (nd) HIP_TRY(hipMemcpy(dense, c->list.dense + r * k3::kDense, (size_t)nd * sizeof(uint2), hipMemcpyDeviceToHost));
        if (walks) HIP_TRY(hipMemcpy(wid, c->list.wid + r * k3::kWalkSlots, (size_t)walks * 4, hipMemcpyDeviceToHost));
        const int64_t blk = (int64_t)(r >> 2), tile_g = blk / c->list_groups;
        const int64_t n0 = ((blk - tile_g * c->list_groups) * 4 + (int64_t)(r & 3)) * c->list_strip_cells;
        for (uint32_t i = 0; i < nd + walks && written < cap; ++i, ++written) {
            const uint32_t pos = i < nd ? dense[i].x : (wid[i - nd] & 0xffffu);
            cells[written] = n0 + (pos >> 8);
            genes[written] = (int32_t)(tile_g * kTileG + (pos & 255u));
        }
    }
    // the segments: what regions listed beyond their first entries
    uint32_t fill[k3::kSegs];
    HIP_TRY(hipMemcpy(fill, c->list.seg_cnt, sizeof(fill), hipMemcpyDeviceToHost));
    std::vector<k3::HeavyEntry> ent;
    for (int sg = 0; sg < k3::kSegs; ++sg) {
        const uint32_t ne = fill[sg] < c->list.ent_cap ? fill[sg] : c->list.ent_cap;
        *total += (int64_t)ne;
        if (written >= cap || ne == 0) continue;
        ent.resize(ne);
        HIP_TRY(hipMemcpy(ent.data(), c->list.ent + (size_t)sg * c->list.ent_cap, (size_t)ne * sizeof(k3::HeavyEntry), hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < ne && written < cap; ++i, ++written) {
            cells[written] = ent[i].n;
            genes[written] = ent[i].g;
        }
    }
    return 0;
}
PA_CATCH

PA_EXPORT int prosstt_amd_nb_params(prosstt_amd_ctx* c, const float* means, int64_t rows, int32_t G,
                                    const int32_t* row_of_cell, const double* scaling,
                                    const double* alpha, const double* beta, int64_t N, float* mu,
                                    float* p, float* r, int32_t* path, uint32_t flags) try
{
    Staging st;
    SamplerArgs A{};
    int rc = sampler_setup(c, st, means, rows, G, row_of_cell, scaling, alpha, beta, N, flags, &A, nullptr, 0, nullptr, 0);
    if (rc) return rc;
    if (N == 0 || G == 0) return 0;
    const size_t bytes = (size_t)N * G * 4;
    void* d[4] = {mu, p, r, path};
    void* h[4] = {mu, p, r, path};
    if (flags & PROSSTT_AMD_HOST_OUTPUT)
        for (int i = 0; i < 4; ++i)
            if (h[i] && (rc = st.alloc(&d[i], bytes))) return rc;
    const int64_t total = N * (int64_t)G;
    nb_params_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream>>>(
        A.means, G, A.row_of_cell, A.scal, A.ga, A.gbm1, N, rows, (float*)d[0], (float*)d[1], (float*)d[2],
        (int32_t*)d[3]);
    HIP_TRY(hipGetLastError());
    if (flags & PROSSTT_AMD_HOST_OUTPUT)
        for (int i = 0; i < 4; ++i)
            if (h[i]) HIP_TRY(hipMemcpyAsync(h[i], d[i], bytes, hipMemcpyDeviceToHost, c->stream));
    if (flags & (PROSSTT_AMD_HOST_OUTPUT | PROSSTT_AMD_HOST_INPUTS)) HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}
PA_CATCH

PA_EXPORT int prosstt_amd_hw_math(prosstt_amd_ctx* c, int32_t op, uint32_t first_bits, uint64_t count, float* out,
                                  uint32_t flags) try
{
    if (!c || !out) return fail(PROSSTT_AMD_EINVAL, "NULL argument");
    if (op < 0 || op > 2) return fail(PROSSTT_AMD_EINVAL, "op must be 0 (rcp), 1 (log2) or 2 (exp2 of -x)");
    if ((uint64_t)first_bits + count > ((uint64_t)1 << 32)) return fail(PROSSTT_AMD_EINVAL, "the range leaves the 32-bit patterns");
    if (count == 0) return 0;
    HIP_TRY(hipSetDevice(c->device));
    float* d = out;
    Staging st;
    if (flags & PROSSTT_AMD_HOST_OUTPUT) {
        void* p = nullptr;
        int rc = st.alloc(&p, count * 4);
        if (rc) return rc;
        d = (float*)p;
    }
    const uint64_t blocks = (count + 255) / 256;
    hw_math_kernel<<<dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, c->stream>>>(op, first_bits, count, d);
    HIP_TRY(hipGetLastError());
    if (flags & PROSSTT_AMD_HOST_OUTPUT) {
        HIP_TRY(hipMemcpyAsync(out, d, count * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return 0;
}
PA_CATCH

PA_EXPORT int prosstt_amd_hw_math_at(prosstt_amd_ctx* c, int32_t op, const float* x, uint64_t count, float* out,
                                     uint32_t flags) try
{
    if (!c || !out || !x) return fail(PROSSTT_AMD_EINVAL, "NULL argument");
    if (op < 0 || op > 5) return fail(PROSSTT_AMD_EINVAL, "op must be 0 (rcp), 1 (log2), 2 (exp2 of -x), 3 (sqrt), 4 (rsq) or 5 (cos of x revolutions)");
    if (count == 0) return 0;
    HIP_TRY(hipSetDevice(c->device));
    Staging st;
    const float* dx = x;
    float* dy = out;
    int rc;
    if (flags & PROSSTT_AMD_HOST_INPUTS) {
        const void* p = nullptr;
        if ((rc = st.upload(x, count * 4, &p, c->stream))) return rc;
        dx = (const float*)p;
    }
    if (flags & PROSSTT_AMD_HOST_OUTPUT) {
        void* p = nullptr;
        if ((rc = st.alloc(&p, count * 4))) return rc;
        dy = (float*)p;
    }
    const uint64_t blocks = (count + 255) / 256;
    hw_math_at_kernel<<<dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, c->stream>>>(op, dx, count, dy);
    HIP_TRY(hipGetLastError());
    if (flags & PROSSTT_AMD_HOST_OUTPUT) HIP_TRY(hipMemcpyAsync(out, dy, count * 4, hipMemcpyDeviceToHost, c->stream));
    if (flags & (PROSSTT_AMD_HOST_OUTPUT | PROSSTT_AMD_HOST_INPUTS)) HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}
PA_CATCH

// ------------------------------------------------------------------ the exchange on RCCL (include/prosstt_amd.h)
// RCCL is opened at first use (dlopen: a process that never calls these never loads it, and one that has torch loaded
// gets torch's copy by its soname).
#include <dlfcn.h>
namespace rccl {
typedef struct { char internal[128]; } UniqueId;
typedef void* Comm;
enum { kInt32 = 2, kUint8 = 1 };
static int (*GetUniqueId)(UniqueId*) = nullptr;
static int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
static int (*CommDestroy)(Comm) = nullptr;
static int (*Send)(const void*, size_t, int, int, Comm, hipStream_t) = nullptr;
static int (*Recv)(void*, size_t, int, int, Comm, hipStream_t) = nullptr;
static int (*GroupStart)() = nullptr;
static int (*GroupEnd)() = nullptr;
static const char* (*GetErrorString)(int) = nullptr;
static bool open()
{
    if (GroupEnd) return true;
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
        if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return false;
    auto sym = [&](const char* n) { return dlsym(h, n); };
    GetUniqueId = (int (*)(UniqueId*))sym("ncclGetUniqueId");
    CommInitRank = (int (*)(Comm*, int, UniqueId, int))sym("ncclCommInitRank");
    CommDestroy = (int (*)(Comm))sym("ncclCommDestroy");
    Send = (int (*)(const void*, size_t, int, int, Comm, hipStream_t))sym("ncclSend");
    Recv = (int (*)(void*, size_t, int, int, Comm, hipStream_t))sym("ncclRecv");
    GroupStart = (int (*)())sym("ncclGroupStart");
    GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
    auto ge = (int (*)())sym("ncclGroupEnd");
    if (!GetUniqueId || !CommInitRank || !CommDestroy || !Send || !Recv || !GroupStart || !ge) return false;
    GroupEnd = ge;
    return true;
}
}  // namespace rccl

struct prosstt_amd_comm { rccl::Comm comm = nullptr; int32_t rank = 0, world = 1; };

#define RCCL_TRY(expr)                                                                                   \
    do {                                                                                                 \
        int r_ = (expr);                                                                                 \
        if (r_ != 0) return fail(PROSSTT_AMD_ERCCL, "%s: %s", #expr, rccl::GetErrorString ? rccl::GetErrorString(r_) : "RCCL error"); \
    } while (0)

PA_EXPORT int prosstt_amd_comm_unique_id(void* id_out) try
{
    if (!id_out) return fail(PROSSTT_AMD_EINVAL, "NULL argument");
    if (!rccl::open()) return fail(PROSSTT_AMD_ERCCL, "librccl.so.1 could not be opened: %s", dlerror());
    rccl::UniqueId id;
    RCCL_TRY(rccl::GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return 0;
}
PA_CATCH

PA_EXPORT int prosstt_amd_comm_init(prosstt_amd_ctx* c, const void* id, int32_t rank, int32_t world, prosstt_amd_comm** out) try
{
    if (!c || !id || !out) return fail(PROSSTT_AMD_EINVAL, "NULL argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(PROSSTT_AMD_EINVAL, "rank %d of %d", rank, world);
    if (!rccl::open()) return fail(PROSSTT_AMD_ERCCL, "librccl.so.1 could not be opened: %s", dlerror());
    HIP_TRY(hipSetDevice(c->device));
    rccl::UniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    prosstt_amd_comm* k = new prosstt_amd_comm;
    k->rank = rank;
    k->world = world;
    const int r = rccl::CommInitRank(&k->comm, world, uid, rank);
    if (r != 0) {
        delete k;
        return fail(PROSSTT_AMD_ERCCL, "ncclCommInitRank: %s", rccl::GetErrorString ? rccl::GetErrorString(r) : "RCCL error");
    }
    *out = k;
    return 0;
}
PA_CATCH

PA_EXPORT int prosstt_amd_comm_destroy(prosstt_amd_comm* k) try
{
    if (!k) return 0;
    if (k->comm && rccl::CommDestroy) (void)rccl::CommDestroy(k->comm);
    delete k;
    return 0;
}
PA_CATCH

PA_EXPORT int prosstt_amd_gather_counts(prosstt_amd_ctx* c, prosstt_amd_comm* k, const int32_t* local_rows,
                                        const int64_t* rows_of_rank, int32_t G, int32_t root, int32_t* dst) try
{
    if (!c || !k || !rows_of_rank) return fail(PROSSTT_AMD_EINVAL, "NULL argument");
    if (G < 0 || root < 0 || root >= k->world) return fail(PROSSTT_AMD_EINVAL, "bad G or root");
    for (int32_t r = 0; r < k->world; ++r)
        if (rows_of_rank[r] < 0) return fail(PROSSTT_AMD_EINVAL, "rows_of_rank[%d] is negative", r);
    const int64_t mine = rows_of_rank[k->rank];
    if (mine > 0 && G > 0 && !local_rows) return fail(PROSSTT_AMD_EINVAL, "local_rows is NULL");
    HIP_TRY(hipSetDevice(c->device));
    if (k->rank != root) {
        if (mine > 0 && G > 0) RCCL_TRY(rccl::Send(local_rows, (size_t)mine * G, rccl::kInt32, root, k->comm, c->stream));
        return 0;
    }
    if (!dst) return fail(PROSSTT_AMD_EINVAL, "dst is NULL on the root");
    // every sender at once: one group, so that every link of the root carries data; the root's own rows are a local copy
    RCCL_TRY(rccl::GroupStart());
    int64_t first = 0;
    int err = 0;
    for (int32_t r = 0; r < k->world && !err; ++r) {
        if (r != root && rows_of_rank[r] > 0 && G > 0)
            err = rccl::Recv(dst + first * G, (size_t)rows_of_rank[r] * G, rccl::kInt32, r, k->comm, c->stream);
        first += rows_of_rank[r];
    }
    const int end = rccl::GroupEnd();
    if (err) RCCL_TRY(err);
    RCCL_TRY(end);
    first = 0;
    for (int32_t r = 0; r < root; ++r) first += rows_of_rank[r];
    if (mine > 0 && G > 0)
        HIP_TRY(hipMemcpyAsync(dst + first * G, local_rows, (size_t)mine * G * 4, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}
PA_CATCH

PA_EXPORT int prosstt_amd_comm_selftest(prosstt_amd_ctx* c, prosstt_amd_comm* k, uint64_t bytes) try
{
    if (!c || !k) return fail(PROSSTT_AMD_EINVAL, "NULL argument");
    if (bytes == 0 || bytes > ((uint64_t)1 << 30)) return fail(PROSSTT_AMD_EINVAL, "1 .. 2^30 bytes");
    HIP_TRY(hipSetDevice(c->device));
    Staging st;
    void *src = nullptr, *dstb = nullptr;
    int rc;
    if ((rc = st.alloc(&src, bytes)) || (rc = st.alloc(&dstb, bytes))) return rc;
    std::vector<unsigned char> pattern(bytes), back(bytes, 0);
    for (uint64_t i = 0; i < bytes; ++i) pattern[i] = (unsigned char)((i * 2654435761u) >> 13);
    HIP_TRY(hipMemcpyAsync(src, pattern.data(), bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(dstb, 0, bytes, c->stream));
    RCCL_TRY(rccl::GroupStart());
    const int e1 = rccl::Send(src, bytes, rccl::kUint8, k->rank, k->comm, c->stream);
    const int e2 = rccl::Recv(dstb, bytes, rccl::kUint8, k->rank, k->comm, c->stream);
    const int e3 = rccl::GroupEnd();
    RCCL_TRY(e1);
    RCCL_TRY(e2);
    RCCL_TRY(e3);
    HIP_TRY(hipMemcpyAsync(back.data(), dstb, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (memcmp(back.data(), pattern.data(), bytes) != 0) return fail(PROSSTT_AMD_ERCCL, "the bytes RCCL delivered differ from the ones sent");
    return 0;
}
PA_CATCH

PA_EXPORT int prosstt_amd_numpy_programs(uint32_t* mt_words, int32_t* mt_next, int32_t* has_gauss, double* gauss,
                                         int32_t attempts, int32_t T, int32_t K, double* start, double* vel0, double* eta,
                                         double* noise, uint32_t* after_words, int32_t* after_next,
                                         int32_t* after_has_gauss, double* after_gauss) try
{
    if (!mt_words || !mt_next || !has_gauss || !gauss || !start || !vel0 || !eta || !after_words || !after_next ||
        !after_has_gauss || !after_gauss || (T > 1 && !noise))
        return fail(PROSSTT_AMD_EINVAL, "NULL argument");
    if (attempts <= 0 || T <= 0 || K <= 0) return fail(PROSSTT_AMD_EINVAL, "bad size");
    if (*mt_next < 0 || *mt_next > npstream::kWords) return fail(PROSSTT_AMD_EINVAL, "generator position %d outside [0,624]", *mt_next);
    npstream::Generator g;
    memcpy(g.word, mt_words, sizeof(g.word));
    g.next = *mt_next;
    g.has_spare = *has_gauss != 0;
    g.spare = *gauss;
    npstream::draw_programs(g, attempts, T, K, start, vel0, eta, noise, after_words, after_next, after_has_gauss, after_gauss);
    memcpy(mt_words, g.word, sizeof(g.word));
    *mt_next = g.next;
    *has_gauss = g.has_spare;
    *gauss = g.spare;
    return 0;
}
PA_CATCH

// centre the first `steps` rows of a [*][K] program matrix over time (scipy.stats.pearsonr's xm = x - mean)
static void centred(const double* src, int steps, int K, double* dst)
{
    for (int k = 0; k < K; ++k) {
        double mean = 0.0;
        for (int t = 0; t < steps; ++t) mean += src[t * K + k];
        mean /= steps;
        for (int t = 0; t < steps; ++t) dst[t * K + k] = src[t * K + k] - mean;
    }
}

// One launch for B attempts of a branch.  Device layout per attempt (stride = `per` doubles):
// raw programs [T][K], then per sibling the centred current and sibling series (see the kernel);
// results: per attempt (1 + n_sib) words {max bits, anticorrelated-gene counts}.
PA_EXPORT int prosstt_amd_lineage_attempt_batch(prosstt_amd_ctx* c, const double* programs, int32_t B, int32_t T,
                                                int32_t K, const double* H, int64_t G, int32_t n_sib,
                                                const double* const* sib_programs, const int32_t* sib_T,
                                                double* out_max, int64_t* out_anticorr) try
{
    if (!c || !programs || !H || !out_max) return fail(PROSSTT_AMD_EINVAL, "NULL argument");
    if (B <= 0 || T <= 0 || K <= 0 || G <= 0 || n_sib < 0) return fail(PROSSTT_AMD_EINVAL, "bad size");
    if (B > 65535) return fail(PROSSTT_AMD_EINVAL, "more than 65535 attempts in a batch");
    if (n_sib > 64) return fail(PROSSTT_AMD_EINVAL, "more than 64 siblings");
    if (n_sib && (!sib_programs || !sib_T || !out_anticorr)) return fail(PROSSTT_AMD_EINVAL, "NULL sibling argument");
    HIP_TRY(hipSetDevice(c->device));

    std::vector<int32_t> meta(1 + n_sib);
    meta[0] = n_sib;
    size_t per = (size_t)T * K;
    for (int j = 0; j < n_sib; ++j) {
        if (sib_T[j] <= 0 || !sib_programs[j]) return fail(PROSSTT_AMD_EINVAL, "bad sibling %d", j);
        meta[1 + j] = T < sib_T[j] ? T : sib_T[j];
        per += 2 * (size_t)meta[1 + j] * K;
    }
    std::vector<double> host(per * (size_t)B);
    for (int b = 0; b < B; ++b) {
        const double* cur = programs + (size_t)b * T * K;
        double* base = host.data() + per * (size_t)b;
        memcpy(base, cur, sizeof(double) * T * K);
        size_t at = (size_t)T * K;                                   // offset inside an attempt's block
        for (int j = 0; j < n_sib; ++j) {
            const size_t len = (size_t)meta[1 + j] * K;
            centred(cur, meta[1 + j], K, base + at);
            if (b == 0) centred(sib_programs[j], meta[1 + j], K, base + at + len);
            else memcpy(base + at + len, host.data() + at + len, sizeof(double) * len);   // the same for every attempt
            at += 2 * len;
        }
    }
    const int32_t words = 1 + n_sib;
    const size_t prog_bytes = host.size() * 8, meta_bytes = (meta.size() * 4 + 7) & ~(size_t)7;
    const size_t res_bytes = (size_t)B * words * 8;
    int rc = ws_reserve(c, prog_bytes + meta_bytes + res_bytes);
    if (rc) return rc;
    double* d_prog = (double*)c->ws;
    int32_t* d_meta = (int32_t*)((char*)c->ws + prog_bytes);
    unsigned long long* d_res = (unsigned long long*)((char*)c->ws + prog_bytes + meta_bytes);
    HIP_TRY(hipMemcpyAsync(d_prog, host.data(), prog_bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_meta, meta.data(), meta.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(d_res, 0, res_bytes, c->stream));   // ordered_bits(x) > 0 for every x
    const dim3 grid((unsigned)((G + 63) / 64), (unsigned)B);
    int64_t lds_need = (int64_t)T * K;
    for (int j = 0; j < n_sib; ++j) lds_need = lds_need > 2 * (int64_t)meta[1 + j] * K ? lds_need : 2 * (int64_t)meta[1 + j] * K;
    if (K <= 32 && lds_need <= kAttLdsDoubles)
        lineage_attempt_lds_kernel<<<dim3((unsigned)((G + kAttGenes - 1) / kAttGenes), (unsigned)B), dim3(256),
                                     (size_t)lds_need * 8, c->stream>>>(
            d_prog, d_meta, T, K, H, G, d_res, d_res + 1, (int64_t)per, words);
    else if (K <= 32)
        lineage_attempt_kernel<true><<<grid, dim3(256), 0, c->stream>>>(d_prog, d_meta, T, K, H, G, d_res, d_res + 1,
                                                                       (int64_t)per, words);
    else
        lineage_attempt_kernel<false><<<grid, dim3(256), 0, c->stream>>>(d_prog, d_meta, T, K, H, G, d_res, d_res + 1,
                                                                        (int64_t)per, words);
    HIP_TRY(hipGetLastError());
    std::vector<unsigned long long> res((size_t)B * words);
    HIP_TRY(hipMemcpyAsync(res.data(), d_res, res_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));   // also keeps `host` alive until the H2D copies are done
    for (int b = 0; b < B; ++b) {
        out_max[b] = from_ordered_bits(res[(size_t)b * words]);
        for (int j = 0; j < n_sib; ++j) out_anticorr[(size_t)b * n_sib + j] = (int64_t)res[(size_t)b * words + 1 + j];
    }
    return 0;
}
PA_CATCH

PA_EXPORT int prosstt_amd_lineage_attempt(prosstt_amd_ctx* c, const double* programs, int32_t T, int32_t K,
                                          const double* H, int64_t G, int32_t n_sib,
                                          const double* const* sib_programs, const int32_t* sib_T,
                                          double* out_max, int64_t* out_anticorr) try
{
    return prosstt_amd_lineage_attempt_batch(c, programs, 1, T, K, H, G, n_sib, sib_programs, sib_T, out_max,
                                             out_anticorr);
}
PA_CATCH

PA_EXPORT int prosstt_amd_lineage_walk_batch(prosstt_amd_ctx* c, uint64_t seed, uint64_t first_stream_id, int32_t B,
                                             int32_t T, int32_t K, double* programs_out) try
{
    if (!c || !programs_out) return fail(PROSSTT_AMD_EINVAL, "NULL argument");
    if (B <= 0 || B > 65535 || T <= 0 || K <= 0) return fail(PROSSTT_AMD_EINVAL, "bad size");
    HIP_TRY(hipSetDevice(c->device));
    const size_t bytes = (size_t)B * T * K * 8;
    int rc = ws_reserve(c, bytes);
    if (rc) return rc;
    lineage_walk_kernel<<<dim3((unsigned)((K + 63) / 64), (unsigned)B), dim3(64), 0, c->stream>>>(
        (uint32_t)seed ^ 0x57414C4Bu, (uint32_t)(seed >> 32), (uint32_t)first_stream_id, (uint32_t)(first_stream_id >> 32), T, K,
        (double*)c->ws);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(programs_out, c->ws, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}
PA_CATCH

PA_EXPORT int prosstt_amd_lineage_walk(prosstt_amd_ctx* c, uint64_t seed, uint64_t stream_id, int32_t T,
                                       int32_t K, double* programs_out) try
{
    return prosstt_amd_lineage_walk_batch(c, seed, stream_id, 1, T, K, programs_out);
}
PA_CATCH

PA_EXPORT int prosstt_amd_lineage_commit(prosstt_amd_ctx* c, const double* programs, int32_t T, int32_t K,
                                         const double* H, int64_t G, double* rel_out, double* gene_max) try
{
    if (!c || !programs || !H) return fail(PROSSTT_AMD_EINVAL, "NULL argument");
    if (T <= 0 || K <= 0 || G <= 0) return fail(PROSSTT_AMD_EINVAL, "bad size");
    HIP_TRY(hipSetDevice(c->device));
    const size_t bytes = (size_t)T * K * 8;
    int rc = ws_reserve(c, bytes);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c->ws, programs, bytes, hipMemcpyHostToDevice, c->stream));
    // ranges of time steps: enough blocks to fill the chip (>= 1024 when T allows), few enough steps per block for its LDS
    const int64_t gene_blocks = (G + 63) / 64;
    if ((int64_t)K * 2 > kCommitLdsDoubles) return fail(PROSSTT_AMD_EINVAL, "more than %d expression programs", kCommitLdsDoubles / 2);
    int64_t ranges = (1024 + gene_blocks - 1) / gene_blocks;
    while (ranges < T && (int64_t)((T + ranges - 1) / ranges + 1) * K > kCommitLdsDoubles) ++ranges;
    if (ranges > T) ranges = T;
    const dim3 grid((unsigned)gene_blocks, (unsigned)ranges);
    if (K <= 32)
        lineage_commit_kernel<true><<<grid, dim3(256), 0, c->stream>>>((const double*)c->ws, T, K, H, G, rel_out, gene_max);
    else
        lineage_commit_kernel<false><<<grid, dim3(256), 0, c->stream>>>((const double*)c->ws, T, K, H, G, rel_out, gene_max);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));   // `programs` is a pageable host buffer the caller may reuse
    return 0;
}
PA_CATCH

PA_EXPORT int prosstt_amd_gene_max(prosstt_amd_ctx* c, const double* rel, int64_t rows, int64_t G,
                                   double* gene_max) try
{
    if (!c || !rel || !gene_max) return fail(PROSSTT_AMD_EINVAL, "NULL argument");
    if (rows < 0 || G < 0) return fail(PROSSTT_AMD_EINVAL, "negative size");
    if (rows == 0 || G == 0) return 0;
    HIP_TRY(hipSetDevice(c->device));
    const int64_t gene_blocks = (G + 63) / 64;
    int64_t ranges = (2048 + gene_blocks - 1) / gene_blocks;
    if (ranges > (rows + 15) / 16) ranges = (rows + 15) / 16;
    gene_max_kernel<<<dim3((unsigned)gene_blocks, (unsigned)(ranges > 0 ? ranges : 1)), dim3(256), 0, c->stream>>>(rel, rows, G, gene_max);
    HIP_TRY(hipGetLastError());
    return 0;
}
PA_CATCH

PA_EXPORT int prosstt_amd_means_from_rel(prosstt_amd_ctx* c, const double* rel, const double* base,
                                         int64_t rows, int64_t G, float* means_out) try
{
    if (!c || !rel || !base || !means_out) return fail(PROSSTT_AMD_EINVAL, "NULL argument");
    if (rows < 0 || G < 0) return fail(PROSSTT_AMD_EINVAL, "negative size");
    if (rows == 0 || G == 0) return 0;
    HIP_TRY(hipSetDevice(c->device));
    int64_t blocks = (rows * G + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    means_from_rel_kernel<<<dim3((unsigned)blocks), dim3(256), 0, c->stream>>>(rel, base, rows, G, means_out);
    HIP_TRY(hipGetLastError());
    return 0;
}
PA_CATCH
