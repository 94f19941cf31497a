/*
 * prosstt_amd -- C ABI of the MI355X-native PROSSTT hot path (libprosstt_amd.so).
 *
 * The reference (soedinglab/prosstt v1.2.0) is pure Python and has no FFI or
 * plugin interface; its boundary is a set of Python call signatures.  Each entry
 * point below replaces the numeric body of the reference functions it cites
 * (file:line under prosstt/), and is what a ctypes stub inside those functions
 * would bind (INTEGRATION.md shows the stubs).
 *
 * Conventions
 *  - extern "C", plain pointers and sizes; never throws, never calls back.
 *  - return 0 on success, a negative PROSSTT_AMD_E* code otherwise; the message
 *    is in the thread-local prosstt_amd_last_error().
 *  - One ctx per (device, stream).  Calls on one ctx are serialised by the caller.
 *    All work is enqueued on the ctx's HIP stream; entry points that hand results
 *    back through HOST pointers synchronise that stream before returning.
 *  - Array arguments are DEVICE pointers unless a flag or the comment says host.
 *  - Results are a pure function of (inputs, seed, cell_offset): independent of
 *    launch geometry, chunking over cells and the number of GPUs.
 */
#ifndef PROSSTT_AMD_H
#define PROSSTT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PROSSTT_AMD_VERSION 600 /* 0.6.0: sampler definition PRNB-7 (counts for a given seed differ from 0.5.0's PRNB-6); prosstt_amd_hw_math_at.
                                  The version is the reproducibility key: it changes whenever the sampler's definition does. */

enum {
    PROSSTT_AMD_OK = 0,
    PROSSTT_AMD_EINVAL = -1,  /* bad argument */
    PROSSTT_AMD_EDOMAIN = -2, /* a mean <= 0 or a*m+b < 1 (the reference raises ValueError there) */
    PROSSTT_AMD_EHIP = -3,    /* HIP runtime error */
    PROSSTT_AMD_ENOMEM = -4,
    PROSSTT_AMD_ENODEV = -5,  /* no gfx950 device visible */
    PROSSTT_AMD_ERCCL = -6    /* RCCL could not be opened, or reported an error (prosstt_amd_comm_*, prosstt_amd_gather_counts) */
};

/* flags of prosstt_amd_sample_counts / prosstt_amd_nb_params */
#define PROSSTT_AMD_HOST_INPUTS  1u /* every input array is a host pointer (staged by the library) */
#define PROSSTT_AMD_HOST_OUTPUT  2u /* every output array is a host pointer */
#define PROSSTT_AMD_CHECK_DOMAIN 4u /* synchronise and return EDOMAIN like scipy's argument check */
#define PROSSTT_AMD_TIME_KERNEL  8u /* bracket the main kernel with HIP events (last_kernel_ms) */
#define PROSSTT_AMD_CHECK_DEFERRED 16u /* the same check without synchronising: the verdict stays in the ctx until
                                          prosstt_amd_domain_status reads it (no kernel is added to the call: the per-cell and
                                          per-gene tests ride in the preparation kernel) */
#define PROSSTT_AMD_PARAMS_NONNEG 64u /* with a checked call: the caller has verified alpha >= 0 and beta >= 1 for every gene, so
                                        alpha*m + beta < 1 cannot happen and the per-sample pass for it is not enqueued */
#define PROSSTT_AMD_MEANS_CACHED 32u /* with a checked call: the mean tensor (same pointer, rows, G) has not changed since the
                                        previous checked call on this ctx -- its per-row flags are reused, not rescanned */

typedef struct prosstt_amd_ctx prosstt_amd_ctx;

int prosstt_amd_version(void);
const char* prosstt_amd_last_error(void);
int prosstt_amd_device_count(int* count);

/* stream: a hipStream_t owned by the caller (e.g. torch's current stream); NULL is
 * the device's default stream.  The ctx never creates or destroys a stream. */
int prosstt_amd_ctx_create(int device, void* stream, prosstt_amd_ctx** out);
int prosstt_amd_ctx_destroy(prosstt_amd_ctx* ctx);
int prosstt_amd_ctx_synchronize(prosstt_amd_ctx* ctx);
/* mean duration (ms) of the main kernels launched with PROSSTT_AMD_TIME_KERNEL since the
 * previous call (HIP events on the ctx stream); synchronises on the last of them */
int prosstt_amd_last_kernel_ms(prosstt_amd_ctx* ctx, float* ms);

/*
 * Fused count sampler.  Replaces the body of
 *   simulation.draw_counts                      simulation.py:602-651
 *   count_model.get_pr_umi (per cell)           count_model.py:131-161
 *   scipy.stats.nbinom(n=r, p=1-p).rvs()        simulation.py:647-648
 * out[n*ld_out + g] ~ NB(mean m, variance alpha*m^2 + beta*m),
 *   m = means[row_of_cell[n]*G + g] * scaling[n],
 * drawn by the PRNB-6 counter-based sampler keyed by
 * (seed, global cell id, g); the global id of cell n is cell_index[n] when
 * cell_index is given, cell_offset + n otherwise.
 *   means        [rows][G] binary32, row-major: the (branch, time, gene) mean tensor
 *   row_of_cell  [N] row of every cell = row offset of its branch + time inside the branch
 *   scaling      [N] library-size factor (sim_utils.calc_scalings, sim_utils.py:473-498)
 *   alpha, beta  [G] variance hyper-parameters
 *   cell_index   [N] global id of every cell, or NULL (a shard of a larger plan passes
 *                the positions of its cells in that plan: results do not depend on sharding)
 *   out          [N][ld_out] int32 counts (the reference returns int64)
 * Limits, refused with PROSSTT_AMD_EINVAL before anything is allocated: N < 2^31, ld_out >= G,
 * ld_out * 128 < 2^29, ceil(N/64)/4 * ceil(G/256) < 2^29 (chunk the cells beyond that), rows > 0.
 * Workspace: the ctx grows its device workspace to about 0.8 * N*G + 36*N + 12*G bytes for the call (per 64 x 256
 * block of the matrix one region of the lists the second kernel draws from: N*G/2 for the staging lists, N*G/4 for the
 * segments, 400 B of dense entries and walk states; 0.76 GB for the 4 GB matrix of 50 000 x 20 000).
 */
int prosstt_amd_sample_counts(prosstt_amd_ctx* ctx, const float* means, int64_t rows, int32_t G,
                              const int32_t* row_of_cell, const double* scaling,
                              const double* alpha, const double* beta, int64_t N, uint64_t seed,
                              uint64_t cell_offset, const int64_t* cell_index, int32_t* out,
                              int64_t ld_out, uint32_t flags);

/*
 * In which ORDER to present the cells (host helper, no device work).  Every count is a function of the cell's global id
 * (cell_index / cell_offset) and the gene alone, so a caller may present its cells to prosstt_amd_sample_counts in any order
 * -- row n of `out` is the cell presented n-th -- and the time depends on it: the kernel walks 64 presented cells at a time
 * against one 256-gene slice of the mean tensor, and cells presented grouped by row_of_cell find the slice's rows in cache
 * instead of fetching them once per cell: 2 % of the kernel on the 8-branch tree, 5 % on the 32-branch tree at 50 000 x
 * 20 000 (profiles/r05_ablation.txt).  order[i] = the cell to present i-th: a stable counting sort of row_of_cell (HOST
 * arrays; rows outside [0, rows) are refused).  Present row_of_cell[order[i]], scaling[order[i]] and cell_index[i] = the global
 * id of cell order[i]; out row i then belongs to cell order[i].  The host layer's simulation.draw_counts
 * (simulation.py:602-651) does this and puts the rows back in plan order inside its copy to the host.
 */
int prosstt_amd_plan_order(const int32_t* row_of_cell, int64_t N, int64_t rows, int32_t* order);

/*
 * The verdict of the PROSSTT_AMD_CHECK_DEFERRED calls since the last time, read and cleared (synchronises the stream):
 * *status = 0, PROSSTT_AMD_EDOMAIN (a mean <= 0 or alpha*m + beta < 1: where scipy's argument check behind
 * simulation.py:647-648 raises ValueError) or PROSSTT_AMD_EINVAL (a row index outside the mean tensor); the message is in
 * prosstt_amd_last_error().  Returns 0 unless the HIP runtime failed.
 */
int prosstt_amd_domain_status(prosstt_amd_ctx* ctx, int32_t* status);

/*
 * The samples that the streaming kernel of the LAST prosstt_amd_sample_counts call on this ctx left
 * to its second kernel (K3h): the gamma-Poisson class, the walks still running when their strip of
 * 64 cells ended and the walks past term 252 (both handed over with their state).  Decoded to (cell, gene)
 * pairs, cells[i] indexing that call's arrays; at most `cap` pairs are written, *total receives the
 * number listed, *overflowed whether a wave's region of the list was too small (more than one in 16 of
 * its 64 x 256 samples listed; K3h then redoes that region sample by sample, and the list holds the
 * entries that fitted).  For tests and diagnostics: it synchronises and copies.
 * Valid until the next call on the ctx.
 */
int prosstt_amd_last_list(prosstt_amd_ctx* ctx, int64_t* cells, int32_t* genes, int64_t cap,
                          int64_t* total, int32_t* overflowed);

/*
 * The deterministic intermediates of the same path (simulation.py:633-645,
 * count_model.py:156-158), as the sampler forms them:  mu = m,  p = theta/(1+theta),
 * r = m/theta  with theta = alpha*m + beta - 1;  path = 0 degenerate / 1 inversion /
 * 2 gamma-Poisson.  Each [N][G]; any output pointer may be NULL.
 */
int prosstt_amd_nb_params(prosstt_amd_ctx* ctx, const float* means, int64_t rows, int32_t G,
                          const int32_t* row_of_cell, const double* scaling, const double* alpha,
                          const double* beta, int64_t N, float* mu, float* p, float* r,
                          int32_t* path, uint32_t flags);

/*
 * The three hardware functions that the sampler's definition (PRNB-6) takes from gfx950,
 * tabulated by the device itself over a range of binary32 bit patterns:
 *   out[i] = f(as_float(first_bits + i)),  i < count;   op 0: f = v_rcp_f32(x), 1: v_log_f32(x), 2: v_exp_f32(-x)
 * -- the side input of the scalar model that checks the sampler bit for bit (the test-side model reads these values
 * instead of re-implementing the hardware).  Replaces nothing of the reference: its scipy.stats.nbinom draws
 * (simulation.py:647-648) evaluate log/exp in libm.  `out`: DEVICE, or HOST with PROSSTT_AMD_HOST_OUTPUT.
 */
int prosstt_amd_hw_math(prosstt_amd_ctx* ctx, int32_t op, uint32_t first_bits, uint64_t count, float* out,
                        uint32_t flags);

/*
 * The gather form of the same probe:  out[i] = f(x[i]),  i < count;  op 0..2 as above, 3: v_sqrt_f32(x),
 * 4: v_rsq_f32(x), 5: v_cos_f32(x) (x in revolutions: cos(2 pi x)).  The gamma-Poisson class of the sampler (PRNB-7)
 * evaluates its logarithms, square roots, cosines and exponentials by these instructions over arguments no table can
 * enumerate; the checking model asks the device for them argument by argument, level by level of its evaluation.
 * Replaces nothing of the reference (numpy's legacy gamma / Poisson behind simulation.py:647-648 call libm).
 * `x`: DEVICE, or HOST with PROSSTT_AMD_HOST_INPUTS; `out`: DEVICE, or HOST with PROSSTT_AMD_HOST_OUTPUT.
 */
int prosstt_amd_hw_math_at(prosstt_amd_ctx* ctx, int32_t op, const float* x, uint64_t count, float* out,
                           uint32_t flags);

/*
 * Multi-GPU at the C boundary: the ONE exchange of the path (SURVEY.md section 8 b/e) -- count rows of every rank's shard to
 * one root, point-to-point over xGMI -- on RCCL directly, for host bindings that do not go through torch.distributed
 * (prosstt_amd/parallel.py is the Python form of the same exchange).  One process per GPU.  The reference has no
 * multi-device path at all: its draw_counts (simulation.py:602-651) fills one (N, G) matrix in one process.
 *
 *   prosstt_amd_comm_unique_id   rank 0 makes the 128-byte id; the caller carries it to the other ranks by whatever it
 *                                has (MPI, a file, a socket): the library has no bootstrap of its own.
 *   prosstt_amd_comm_init        collective over the `world` ranks; the communicator works on ctx's device and stream.
 *   prosstt_amd_gather_counts    every rank: `local_rows` = its n_local x G int32 counts (DEVICE, contiguous rows) and
 *                                rows_of_rank[world] (HOST: how many rows every rank holds -- each rank derives that from
 *                                the plan all ranks share).  On `root`, dst (DEVICE, (sum of rows) x G) receives rank r's
 *                                rows at row offset rows_of_rank[0] + ... + rows_of_rank[r-1] -- shard order, as
 *                                parallel.sample_and_gather(order="shard"); every sender is received at once (one RCCL
 *                                group: every xGMI link of the root carries data).  dst is ignored on the other ranks.
 *                                Enqueued on ctx's stream, not waited for.
 *   prosstt_amd_comm_selftest    one send / receive of `bytes` bytes from this rank to itself through RCCL, compared on the
 *                                host: the communicator moves data on this machine (works with world = 1).
 * RCCL is opened when the first of these is called (librccl.so.1 of the process, i.e. torch's when torch is loaded);
 * PROSSTT_AMD_ERCCL when it cannot be opened or reports an error.
 */
typedef struct prosstt_amd_comm prosstt_amd_comm;
#define PROSSTT_AMD_COMM_ID_BYTES 128
int prosstt_amd_comm_unique_id(void* id_out);
int prosstt_amd_comm_init(prosstt_amd_ctx* ctx, const void* id, int32_t rank, int32_t world, prosstt_amd_comm** out);
int prosstt_amd_comm_destroy(prosstt_amd_comm* comm);
int prosstt_amd_gather_counts(prosstt_amd_ctx* ctx, prosstt_amd_comm* comm, const int32_t* local_rows,
                              const int64_t* rows_of_rank, int32_t G, int32_t root, int32_t* dst);
int prosstt_amd_comm_selftest(prosstt_amd_ctx* ctx, prosstt_amd_comm* comm, uint64_t bytes);

/*
 * Host only (no device, no ctx): the variates of `attempts` consecutive simulation.sim_expr_branch(T, K) calls
 * (simulation.py:21-86; per walk, simulation.diffusion's draws, simulation.py:104-113: uniform(0, 1.5),
 * normal(0, 0.2), uniform(0, 1), normal(0, 2/T) x (T-1)) taken from numpy's legacy global stream (MT19937,
 * 53-bit doubles, polar normals with the cached second value) exactly as those calls take them.
 *   mt_words[624], *mt_next, *has_gauss, *gauss   in: the fields of np.random.get_state(); out: the state behind
 *                                                 the last attempt
 *   start, vel0, eta   [attempts][K];   noise  [attempts][K][T-1]
 *   after_words [attempts][624], after_next, after_has_gauss, after_gauss [attempts]: the state behind EVERY attempt
 *                                                 (the caller rewinds numpy to the attempt it accepts)
 * Replaces the Python-level loop of 4 numpy calls per walk that bounds the lineage stage on large trees.
 */
int prosstt_amd_numpy_programs(uint32_t* mt_words, int32_t* mt_next, int32_t* has_gauss, double* gauss,
                               int32_t attempts, int32_t T, int32_t K, double* start, double* vel0, double* eta,
                               double* noise, uint32_t* after_words, int32_t* after_next, int32_t* after_has_gauss,
                               double* after_gauss);

/*
 * One attempt of the accept/reject loop of simulation.simulate_lineage
 * (simulation.py:264-282) for one branch, without materialising (T,G):
 *   rel = programs @ H                                   simulation.py:269
 *   *out_max = max(rel)                                  simulation.py:270
 *   out_anticorr[j] = #genes with Pearson r < 0 between rel and sibling j's
 *     rel over the first min(T, sib_T[j]) steps          sim_utils.py:145-168, 249-250
 *   programs      HOST [T][K] (already adjusted to the parent, sim_utils.py:611-640)
 *   H             DEVICE [K][G] coefficients (simulation.py:192-212)
 *   sib_programs  HOST array of n_sib HOST pointers, each [sib_T[j]][K]
 *   out_max, out_anticorr  HOST
 */
int prosstt_amd_lineage_attempt(prosstt_amd_ctx* ctx, const double* programs, int32_t T, int32_t K,
                                const double* H, int64_t G, int32_t n_sib,
                                const double* const* sib_programs, const int32_t* sib_T,
                                double* out_max, int64_t* out_anticorr);

/*
 * B attempts of the same branch in one launch (the loop of simulation.py:264-282 evaluated a
 * batch at a time: the host draws the B candidate programs in the reference's stream order,
 * takes the first accepted one and rewinds its generator to just behind it).
 *   programs      HOST [B][T][K]
 *   out_max       HOST [B];   out_anticorr  HOST [B][n_sib]
 * Everything else as prosstt_amd_lineage_attempt.
 */
int prosstt_amd_lineage_attempt_batch(prosstt_amd_ctx* ctx, const double* programs, int32_t B, int32_t T,
                                      int32_t K, const double* H, int64_t G, int32_t n_sib,
                                      const double* const* sib_programs, const int32_t* sib_T,
                                      double* out_max, int64_t* out_anticorr);

/*
 * Device-mode expression programs (K1): the K random walks of one branch attempt,
 * simulation.sim_expr_branch / diffusion (simulation.py:21-124), drawn on the device from
 * Philox streams instead of numpy's global stream (walk definition PRLW-1).
 * One lane per program, T sequential steps in binary64.  programs_out: HOST [T][K].
 * stream_id distinguishes walks drawn with the same seed (branch and attempt number).
 */
int prosstt_amd_lineage_walk(prosstt_amd_ctx* ctx, uint64_t seed, uint64_t stream_id, int32_t T,
                             int32_t K, double* programs_out);
/* B attempts in one launch and one copy back: walk streams first_stream_id .. first_stream_id + B - 1;
 * programs_out: HOST [B][T][K]. */
int prosstt_amd_lineage_walk_batch(prosstt_amd_ctx* ctx, uint64_t seed, uint64_t first_stream_id, int32_t B,
                                   int32_t T, int32_t K, double* programs_out);

/*
 * Materialise an accepted branch: rel_out[t][g] = sum_k programs[t][k]*H[k][g]
 * (binary64; simulation.py:269) and fold max_t rel into gene_max[g] (the log of
 * sim_utils.max_relat_exp, sim_utils.py:406-426).
 *   programs HOST [T][K];  H, rel_out ([T][G], may be NULL), gene_max ([G], may be NULL) DEVICE.
 *   gene_max must be initialised to -inf by the caller before the first branch.
 */
int prosstt_amd_lineage_commit(prosstt_amd_ctx* ctx, const double* programs, int32_t T, int32_t K,
                               const double* H, int64_t G, double* rel_out, double* gene_max);

/*
 * gene_max[g] = max(gene_max[g], max_row rel[row][g]) for a DEVICE [rows][G] binary64
 * matrix: the reduction inside sim_utils.max_relat_exp (sim_utils.py:423-425) in
 * log space (exp is monotone).  gene_max is DEVICE [G], initialised by the caller.
 */
int prosstt_amd_gene_max(prosstt_amd_ctx* ctx, const double* rel, int64_t rows, int64_t G,
                         double* gene_max);

/*
 * Tree.add_genes (tree.py:166-183): means[row][g] = exp(rel[row][g]) * base[g],
 * evaluated in binary64 and stored as binary32.  All DEVICE; rows = sum of T_b.
 */
int prosstt_amd_means_from_rel(prosstt_amd_ctx* ctx, const double* rel, const double* base,
                               int64_t rows, int64_t G, float* means_out);

#ifdef __cplusplus
}
#endif
#endif /* PROSSTT_AMD_H */
