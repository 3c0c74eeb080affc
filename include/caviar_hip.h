/*
 * caviar_hip.h -- C ABI of libcaviar_hip.so: MI355X (gfx950) implementation of circuitmap's CAVIaR
 * coordinate-ascent inner loop.
 *
 * Drop-in boundary: this library replaces the callable
 *     circuitmap.optimise.caviar(y_psc, I, mu_prior, beta_prior, shape_prior, rate_prior, phi_prior,
 *                                phi_cov_prior, **fit_options)        reference circuitmap/optimise/caviar.py:20-23
 * whose 17-tuple return (caviar.py:100) is consumed by Model._fit_caviar (reference circuitmap/model.py:107-136).
 * The reference has no FFI of its own (it is pure Python on jax); a maintainer binds these entry points with
 * ctypes -- see INTEGRATION.md and circuitmap_amd/_lib.py.
 *
 * Conventions: plain pointers and sizes only; every array is float64 unless noted, row-major; the library never
 * frees caller memory; all functions return CM_OK (0) or a negative cm_status, with a message available from
 * cm_last_error() (thread-local); no exceptions cross the ABI.  One cm_ctx = one GPU + one HIP stream; calls on
 * one ctx must be serialised by the caller, distinct ctxs are independent (one process per GPU, or several ctxs
 * per process).
 *
 * Limits (also in INTEGRATION.md): N >= 2; at most CM_MAX_VALS distinct stimulus values; active sets of more than 18 000 neurons
 * are refused (CM_ERR_INVALID: the row accumulator of the precision build lives in LDS); at most 1024 circuits per lock-step group.
 */
#ifndef CAVIAR_HIP_H
#define CAVIAR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum cm_status {
  CM_OK = 0,
  CM_ERR_INVALID = -1,        /* bad argument (NULL, N < 2, K < 1, ...)                         */
  CM_ERR_HIP = -2,            /* a HIP runtime call failed (message has the HIP error string)    */
  CM_ERR_NO_DEVICE = -3,      /* no gfx950 device visible: the library has NO CPU fallback       */
  CM_ERR_TOO_MANY_POWERS = -4,/* more than CM_MAX_VALS distinct stimulus values                  */
  CM_ERR_NOT_SPD = -5,        /* weight-posterior precision matrix not positive definite         */
  CM_ERR_STATE = -6           /* call order violated (e.g. fetch before run)                     */
} cm_status;

#define CM_MAX_VALS 64        /* distinct values of the stimulus matrix (0 and the laser powers) */

/* ABI handshake.  Bumped whenever the layout of a struct of this header changes (history: 1 = round 1; 2 = cm_timings re-laid out
 * in round 2; 3 = this field, cm_timings.fact_tri_tiles / fact_products; 4 = cm_options.interleave).  cm_options_default() writes it into cm_options.abi_version and every entry point
 * that takes a cm_options refuses (CM_ERR_INVALID) a value other than the library's own: a caller compiled against another header
 * gets an error code instead of silently mis-read fields or an overrun cm_timings.  cm_abi_version() returns the library's value. */
#define CM_ABI_VERSION 4

typedef enum cm_mem_kind { CM_MEM_HOST = 0, CM_MEM_DEVICE = 1 } cm_mem_kind;
typedef enum cm_rng_mode {
  CM_RNG_THREEFRY = 0,        /* counter-based Threefry-2x32-20 keyed by `seed`, jax.random conventions
                                 (replaces jax.random.PRNGKey/split/uniform/choice, caviar.py:76,196,209-210,304) */
  CM_RNG_EXPLICIT = 1         /* host-supplied streams: update_orders[iters*N], uniforms[iters*N*S*2]              */
} cm_rng_mode;

typedef struct cm_ctx cm_ctx;           /* device context: stream + workspaces                    */
typedef struct cm_problem cm_problem;   /* one circuit's encoded inputs + CAVIaR state, in HBM    */

/* Replaces the reference's positional prior arguments (caviar.py:20; defaults circuitmap/model.py:24-31). Host memory. */
typedef struct cm_priors {
  const double* mu;        /* [N]      */
  const double* beta;      /* [N]  prior std-dev of the weights */
  double shape;
  double rate;
  const double* phi;       /* [N*2]    */
  const double* phi_cov;   /* [N*2*2]  */
} cm_priors;

/* Mirrors every keyword of caviar() (caviar.py:21-23), same names and defaults (see cm_options_default). */
typedef struct cm_options {
  int32_t abi_version;             /* CM_ABI_VERSION of the header the caller was compiled against (set by cm_options_default) */
  int32_t iters;
  int32_t num_mc_samples;
  uint64_t seed;
  double y_xcorr_thresh;
  double minimum_spike_count;
  int32_t delay_spont_est;
  double msrmp;
  double scale_factor;
  double penalty;
  int32_t save_histories;
  int32_t max_backtrack_iters;
  double tol;
  double spont_orthogonality;
  int32_t fn_scan;
  /* additions of this library */
  int32_t rng_mode;                /* cm_rng_mode */
  const int32_t* update_orders;    /* CM_RNG_EXPLICIT: [iters*N] host, sweep order per iteration               */
  const double* uniforms;          /* CM_RNG_EXPLICIT: [iters*N*S*2] host, U[0,1) indexed by sweep step         */
  int32_t verbose;                 /* >= 1: the reference's progress messages of the false-negative scan (stderr); >= 3: one line per iteration */
  int32_t timing;                  /* cm_timings detail.  0: fit totals only (4 events per fit, the default: per-kernel events cost ~2 us
                                      of launch boundary each, ~8 % of a fit); 1: + events around the chain kernel and the mid-iteration
                                      kernel of every iteration (sweep_seq_ms, mid_ms); 2: + all phase stamps                                  */
  int32_t interleave;              /* cm_caviar_run_batch / cm_caviar_fit_batch.  0 (default): two lock-step batches at a time -- a group of
                                      >= CM_INTERLEAVE_MIN circuits is split in two halves, the second half (or every other group) runs on a
                                      second stream of the context from a helper thread, so that the dependency-bound phases of one batch
                                      (Gauss-Seidel chains, Newton steps, last Cholesky panels, false-negative scan) overlap the wide phases
                                      of the other.  1: one lock-step batch at a time on the context's
                                      stream (kernel-level profiling; the per-kernel cm_timings are then free of cross-stream contention).
                                      Results are identical either way.                                                                          */
} cm_options;
#define CM_INTERLEAVE_MIN 32       /* smallest group that is split in two halves */

/* Caller-allocated host output buffers (any pointer may be NULL = not wanted).  Replaces the 17-tuple of caviar.py:100. */
typedef struct cm_result {
  double* mu;          /* [N]                                                                      */
  double* beta;        /* [N]   posterior VARIANCE as returned by the reference (caviar.py:172)    */
  double* lam;         /* [N*K] dense                                                              */
  double shape;
  double rate;
  double* phi;         /* [N*2]  */
  double* phi_cov;     /* [N*4]  */
  double* z;           /* [K]    */
  /* histories, only filled when options.save_histories != 0 (caviar.py:53-73,90-92) */
  double* mu_hist;       /* [iters*N]   */
  double* beta_hist;     /* [iters*N]   */
  double* lam_hist;      /* [iters*N*K] */
  double* shape_hist;    /* [iters*K] (scalar broadcast, as the reference does) */
  double* rate_hist;     /* [iters*K]   */
  double* phi_hist;      /* [iters*N*2] */
  double* phi_cov_hist;  /* [iters*N*4] */
  double* z_hist;        /* [iters*K]   */
} cm_result;

/* Per-run device timings in milliseconds, from HIP events recorded on the ctx stream (no extra host synchronisation; elapsed
 * times are read after the run's final sync).  The three totals are always filled; the per-kernel sums need
 * cm_options.timing >= 1 (sweep_seq_ms, mid_ms) or >= 2 (the rest) and are 0 otherwise -- every event is a launch boundary. */
typedef struct cm_timings {
  double total_ms;       /* whole run: state init + iterations + fn_scan                      */
  double iters_ms;       /* the `iters` CAVIaR iterations only                                */
  double fn_scan_ms;     /* reconnect_spont_cells (host) + final update_phi                   */
  double solve_ms;       /* timing >= 2, sum over iterations: host wait + block_update_mu of large active sets (build + Cholesky + inverse) */
  double sweep_indep_ms; /* timing >= 2, sum: k_presweep + k_seq_pack launches                */
  double sweep_seq_ms;   /* timing >= 1, sum: chain-kernel launches (k_sweep_par / k_sweep_seq: the Gauss-Seidel chain) */
  double mid_ms;         /* timing >= 1, sum: k_iter_mid launches (sigma, z, small solve || phi + next MC logit), each incl. one launch boundary */
  double post_ms;        /* timing >= 2, sum: k_stats + k_iter_mid                            */
  double mc_dead_ms;     /* timing >= 1, batches only, sum: k_mc_dead launches (the all-zero rows' Monte-Carlo logit; for one circuit those rows are inside k_iter_mid) */
  double fact_ms;        /* timing >= 1, sum over the tiled solves of the run: first to last launch of the Cholesky factorisation (event pair per solve) */
  int32_t iters;
  int32_t batch;         /* circuits that shared the launches of this run (1 for cm_caviar_run): the *_ms fields are those of the whole set */
  int32_t last_active;   /* size of the active (lam row != 0) set in the last iteration       */
  int64_t sum_active;    /* sum over iterations of the active-set size                        */
  int64_t sum_seq_rows;  /* sum over iterations: rows in the sequential chain (mu != 0)        */
  int64_t sum_seq_records; /* sum over iterations: 32-byte records streamed by k_sweep_seq (rows + their trials) */
  int64_t fact_launches; /* trailing-update passes inside fact_ms: k_chol_panel launches (one panel each), or panel-pair steps
                            (k_chol_trsm, thin k_chol_bulk, k_chol_trsm, rank-64 k_chol_bulk) for batches and above 48 tile rows   */
  int64_t fact_tile_updates; /* 32 x 32 fp64 tiles those passes read AND rewrote (16 KB each), summed over the circuits that took part */
  int64_t fact_panel_tiles;  /* further 8 KB tile transfers the same passes cannot avoid, each counted once per launch: panel tiles read,
                                solved panel tiles written, the diagonal tile's inverse                                               */
  int64_t fact_tri_tiles;    /* sum over the tiled solves of the run and the circuits that took part: 32 x 32 tiles of the active set's lower
                                triangle, nb (nb + 1) / 2 -- the COMPULSORY traffic of a solve is 3 x 8 KB per such tile (precision matrix read once,
                                factor and inverse written once), whatever the blocking                                                 */
  int64_t fact_products;     /* 32 x 32 x 32 tile products (65 536 flop each, on the fp64 matrix cores) inside fact_ms: panel solves + trailing updates */
  int64_t inv_products;      /* tile products of the recursive-doubling inverse of the factor (k_inv_T, k_inv_X21) of the same solves          */
  double tiled_ms;           /* timing >= 1, sum over the tiled solves of the run: precision-matrix build, factorisation, inverse, mu / beta
                                reductions -- first to last launch of each block_update_mu that took the tiled path (event pair per solve)   */
  int64_t tiled_solves;      /* iterations of the run whose block_update_mu took the tiled path (for at least one circuit of the set)              */
} cm_timings;

const char* cm_last_error(void);
const char* cm_version(void);
int cm_abi_version(void);                 /* CM_ABI_VERSION the library was built with */
int cm_device_count(void);

void cm_options_default(cm_options* o);   /* defaults of caviar.py:21-23 */

/* hip_stream: a hipStream_t to run on (e.g. torch's current stream) or NULL for a private stream. */
int cm_ctx_create(int device_id, void* hip_stream, cm_ctx** out);
int cm_ctx_destroy(cm_ctx* ctx);
int cm_ctx_synchronize(cm_ctx* ctx);
/* A context keeps the device memory of destroyed problems (their two slabs and the random tables of their runs) in a pool and hands it
 * to the next problem of that size class instead of returning it to the driver: hipMalloc / hipFree cost more than encoding a circuit.
 * Memory only, never contents.  Idle bytes are capped (environment CAVIAR_HIP_POOL_GB, default 64; 0 = no pooling) and returned by
 * cm_ctx_destroy; cm_ctx_release_cache returns them now. */
int cm_ctx_release_cache(cm_ctx* ctx);

/* Device buffers for hosts that have no other way to hold HBM (a ctypes caller without torch): lets the output of one
 * entry point (e.g. cm_nwd_demix, nwd_hip.h) feed the next (cm_problem_create with CM_MEM_DEVICE) without a host round
 * trip.  cm_mem_copy: kind 1 = host->device, 2 = device->host, 3 = device->device; ordered on the ctx stream and complete
 * on return. */
int cm_mem_alloc(cm_ctx* ctx, uint64_t bytes, void** out);
int cm_mem_free(cm_ctx* ctx, void* ptr);
int cm_mem_copy(cm_ctx* ctx, void* dst, const void* src, uint64_t bytes, int kind);

/* Uploads (CM_MEM_HOST) or adopts by copy (CM_MEM_DEVICE) psc[K*T] and stim[N*K], then runs the one-off prologue
 * on the GPU: y = trapz(psc), sum(psc^2) (caviar.py:28-30) and the compressed encoding of the stimulus matrix
 * (distinct powers, per-row and per-column index of targeted trials; caviar.py:32-35,42). */
int cm_problem_create(cm_ctx* ctx, int32_t N, int32_t K, int32_t T, const double* psc, const double* stim,
                      int mem_kind, cm_problem** out);
/* The same for n circuits at once (any mix of shapes): every kernel of the prologue is launched once for the group (blockIdx.y = circuit) and the
 * host waits twice per group instead of twice per circuit -- 0.03 instead of 0.22 ms per circuit at the headline shape.  psc[c] / stim[c]:
 * the arrays of circuit c (all host or all device, mem_kind); out[n] receives the problems (on failure none is left allocated). */
int cm_problem_create_batch(cm_ctx* ctx, int32_t n, const int32_t* N, const int32_t* K, const int32_t* T, const double* const* psc,
                            const double* const* stim, int mem_kind, cm_problem** out);
/* Compact inputs.  The reference's callers hold float arrays with K x T recorded samples and an N x K stimulus matrix that is zero in all but
 * H of N entries per trial (scripts/run_circuitmap_main.py:22-32); as dense float64 that is 30 MB per circuit at N = 1000 / K = 2000 / T = 900, and
 * moving it over PCIe was 5x the fit of 256 circuits.  cm_problem_create_batch_ex takes, per circuit, psc as float64 or float32 and the stimulus
 * matrix dense (float64 / float32) or as its non-zero entries; the library expands them on the device to the float64 arrays of
 * cm_problem_create_batch, so the circuit -- and every result -- is bitwise the one built from those arrays (float32 values enter as their exact
 * float64 equals; entries of the stimulus matrix that no triple names are 0; a (trial, neuron) pair named twice takes either power: list each once).
 * A triple outside the N x K matrix fails the call with CM_ERR_INVALID. */
enum { CM_DTYPE_F64 = 0, CM_DTYPE_F32 = 1 };
enum { CM_STIM_DENSE_F64 = 0, CM_STIM_DENSE_F32 = 1, CM_STIM_TRIPLES = 2 };
typedef struct cm_stim_triple { int32_t trial, neuron; float power; } cm_stim_triple;   /* stim_matrix[neuron][trial] = power */
typedef struct cm_inputs {
  int32_t psc_dtype;      /* CM_DTYPE_*: element type of psc [K * T], row-major */
  int32_t stim_layout;    /* CM_STIM_*: stim is [N * K] float64 / float32 row-major, or n_triples cm_stim_triple */
  const void* psc;
  const void* stim;
  int64_t n_triples;
} cm_inputs;
int cm_problem_create_batch_ex(cm_ctx* ctx, int32_t n, const int32_t* N, const int32_t* K, const int32_t* T, const cm_inputs* inputs, int mem_kind,
                               cm_problem** out);
int cm_problem_destroy(cm_problem* p);
int cm_problem_nnz(const cm_problem* p, int64_t* nnz, int32_t* n_vals);

/* Runs caviar(): `iters` iterations (mu/beta -> lam -> shape/rate -> phi -> z, caviar.py:79-92) and, if
 * options.fn_scan, reconnect_spont_cells + the final update_phi (caviar.py:94-98).  Results stay in HBM. */
int cm_caviar_run(cm_ctx* ctx, cm_problem* p, const cm_priors* priors, const cm_options* options, cm_timings* timings);

/* Copies the results of the last cm_caviar_run to host buffers. */
int cm_caviar_fetch(cm_ctx* ctx, cm_problem* p, cm_result* result);
/* cm_caviar_fetch for n problems of one context with one device-to-host copy: mu, beta, phi, phi_cov, z, shape, rate of every circuit are packed
 * on the device first.  results[c] as for cm_caviar_fetch; a result that asks for lam or histories is fetched on its own. */
int cm_caviar_fetch_batch(cm_ctx* ctx, int32_t n, cm_problem* const* problems, cm_result* results);

/* Convenience: create + run + fetch + destroy for host inputs (what a ctypes drop-in of caviar() calls). */
int cm_caviar_fit(cm_ctx* ctx, int32_t N, int32_t K, int32_t T, const double* psc, const double* stim,
                  const cm_priors* priors, const cm_options* options, cm_result* result, cm_timings* timings);

/* Lock-step batch on ONE GPU: fits `n_circuits` problems (all created on `ctx`, distinct) with the same options.  Circuits with the same
 * sizes form a group whose every kernel launch carries the circuit as its outer grid dimension (one sweep workgroup per circuit, the
 * per-neuron phases of all circuits side by side): a single fit is a latency chain that occupies a few percent of an MI355X, a batch
 * fills it.  Each circuit's results are bit-identical to its own cm_caviar_run (same kernels bodies, same summation orders; the
 * random streams depend only on (seed, N, num_mc_samples) and are generated once per group).  priors[n_circuits]; timings[n_circuits]
 * or NULL (device times are those of the circuit's group).  Replaces the reference's only scale-out, one process per circuit
 * (reference scripts/generate_loho_cv_slurm_scripts.py:8-36).  Results stay in HBM: cm_caviar_fetch per problem. */
int cm_caviar_run_batch(cm_ctx* ctx, int32_t n_circuits, cm_problem* const* problems, const cm_priors* priors,
                        const cm_options* options, cm_timings* timings);

/* Convenience over several GPUs of this process: circuit c (host arrays psc[c] = K[c] x T[c], stim[c] = N[c] x K[c]) is created on
 * device_ids[c % n_devices]; every device runs its circuits as one cm_caviar_run_batch from its own host thread; results[n_circuits]
 * hold caller-allocated buffers as for cm_caviar_fetch.  device_ids == NULL: device 0. */
int cm_caviar_fit_batch(int32_t n_circuits, const int32_t* N, const int32_t* K, const int32_t* T, const double* const* psc,
                        const double* const* stim, const cm_priors* priors, const cm_options* options, const int32_t* device_ids,
                        int32_t n_devices, cm_result* results, cm_timings* timings);
/* cm_caviar_fit_batch on compact inputs (cm_inputs: float32 psc, the stimulus matrix as float32 or as its non-zero entries): 7.4 instead of 30 MB
 * per circuit over PCIe at the headline shape; results bitwise those of the dense float64 call on the same values. */
int cm_caviar_fit_batch_ex(int32_t n_circuits, const int32_t* N, const int32_t* K, const int32_t* T, const cm_inputs* inputs, const cm_priors* priors,
                           const cm_options* options, const int32_t* device_ids, int32_t n_devices, cm_result* results, cm_timings* timings);

/* cm_caviar_fit_batch* keep one context per device between calls, its memory pool warm (creating the pool for 256 circuits and handing its ~20 GB
 * back to the driver cost more than the fit).  This destroys them; CAVIAR_HIP_FIT_BATCH_NO_CACHE in the environment keeps none. */
int cm_caviar_fit_batch_release(void);

/* ---- per-phase entry points (parity tests exercise each kernel in isolation; host buffers in/out) ---- */

/* caviar.py:28-30: y[K] = trapz(psc[k,:]), ss[K] = sum_t psc[k,t]^2 */
int cm_k_prologue(cm_ctx* ctx, int32_t K, int32_t T, const double* psc, double* y, double* ss);
/* pava.py:86-88: last element of the isotonic (non-decreasing) regression of y[n] (host-side helper, P <= CM_MAX_VALS) */
int cm_pava_last(const double* y, int32_t n, double* out);
/* caviar.py:166-172 on a dense lam[N*K] */
int cm_k_block_update_mu(cm_ctx* ctx, int32_t N, int32_t K, const double* stim, const double* lam, const double* y,
                         double shape, double rate, const double* mu_prior, const double* beta_prior,
                         double* mu, double* beta);
/* caviar.py:246-316 on dense lam/stim */
int cm_k_update_phi(cm_ctx* ctx, int32_t N, int32_t K, const double* stim, const double* lam,
                    const double* phi_prior, const double* phi_cov_prior, double* phi, double* phi_cov);
/* caviar.py:238-244 update_sigma on dense lam: shape = shape_prior + K/2, rate = rate_prior + (|y - lam^T mu|^2 - ... )/2 */
int cm_k_sigma(cm_ctx* ctx, int32_t N, int32_t K, const double* stim, const double* lam, const double* y, const double* mu,
               const double* beta, double shape_prior, double rate_prior, double* shape, double* rate);
/* caviar.py:147-163 estimate_spont_act_soft_thresh as caviar() calls it at OUTER iteration `it` (:86: loop counter = it, err = sum(y)),
 * then spont_rate = mean(z != 0) (:88).  mask[K] = lam_mask (bytes); z[K] in: previous z (kept when the loop does not run), out: new z */
int cm_k_spont(cm_ctx* ctx, int32_t N, int32_t K, const double* stim, const double* lam, const double* y, const uint8_t* mask,
               const double* mu, int32_t it, int32_t max_backtrack_iters, double penalty, double scale_factor, double tol,
               double spont_orthogonality, double* z, double* spont_rate);
/* caviar.py:190-231 update_lam on dense lam[N*K] (in/out).  key[2]: the jax key passed to update_lam, replaced by the key it returns;
 * minimax_spk_prob = msrmp + spont_rate as at the call site (:81-83); beta = what block_update_mu returned (a variance, quirk 1) */
int cm_k_lam_sweep(cm_ctx* ctx, int32_t N, int32_t K, const double* stim, double* lam, const double* y, const uint8_t* mask,
                   const double* mu, const double* beta, double shape, double rate, const double* phi, const double* phi_cov,
                   uint32_t* key, int32_t num_mc_samples, double minimum_spike_count, double minimax_spk_prob, int32_t it,
                   int32_t delay_spont_est);
/* caviar.py:102-144 reconnect_spont_cells on dense lam (in/out: mu, beta, lam[N*K], z are updated in place) */
int cm_k_reconnect(cm_ctx* ctx, int32_t N, int32_t K, const double* stim, double* lam, double* mu, double* beta, double* z,
                   double minimax_spk_prob, double minimum_spike_count);
/* Threefry streams as the device generates them (for comparison with oracle/threefry.py): order[N], u[N*S*2] of iteration `it` */
int cm_k_rng_streams(cm_ctx* ctx, uint64_t seed, int32_t it, int32_t N, int32_t S, int32_t* order, double* u);
/* the standard normal quantile the Monte-Carlo draws use (ndtri of reference caviar.py:212; Wichura AS 241 on the device) on n host values */
int cm_k_ndtri(cm_ctx* ctx, int32_t n, const double* p, double* out);

/* Measurement aid: cm_timings brackets a kernel with two HIP events recorded on the stream, so each figure contains one
 * dependent-launch boundary besides the kernel itself.  This returns that boundary, measured the same way around a kernel that
 * does nothing (average over `reps` event pairs behind a busy stream), so that bench.py can report kernel durations comparable
 * with a profiler's kernel trace. */
int cm_k_event_pair_overhead(cm_ctx* ctx, int32_t reps, int32_t blocks, int32_t threads, float* ms_per_pair);

/* Measurement aid for the tiled solve: B synthetic SPD matrices of n rows go through the Cholesky factorisation and the inverse of the
 * factor exactly as a lock-step batch of B circuits with n active rows each launches them (B = 1: the single-circuit launches); average
 * milliseconds over `reps` repetitions after one warm-up (tools/solve_bench.py). */
int cm_k_solve_bench(cm_ctx* ctx, int32_t n, int32_t B, int32_t reps, float* ms_chol, float* ms_inv);
/* measurement aid: the fp64 matrix-core rate (TFLOP/s) the device sustains with operands in registers and no memory traffic; `iters` MFMA
   rounds per wavefront (200000 runs ~0.3 s) */
int cm_k_mfma_peak(cm_ctx* ctx, int32_t iters, double* tflops);

/* Test hook for the library's dynamic-LDS bookkeeping (no GPU needed): hipFuncAttributeMaxDynamicSharedMemorySize REPLACES the value in
 * force for a (kernel, device), so the library keeps one process-wide table and only ever raises it.  Returns the value the runtime
 * would be called with for this request (0 = the value in force already covers it).  `kernel_key` is an arbitrary test key. */
int64_t cm_test_lds_request(int32_t device, uint64_t kernel_key, uint64_t bytes, uint64_t floor_bytes);
/* The same bookkeeping under contention: n_threads host threads issue interleaved, growing requests for one (device, kernel) against a
 * recording stand-in for the runtime call, which the library invokes while holding the table's lock.  out[0] = value the stand-in holds at
 * the end, out[1] = largest request issued, out[2] = calls made, out[3] = value applied by a repeat of out[0] (must be 0).  fail_above > 0:
 * the stand-in refuses larger values and the table must not record them. */
int cm_test_lds_race(uint64_t kernel_key, int32_t n_threads, int32_t reps, uint64_t base, uint64_t step, uint64_t fail_above, uint64_t* out);

#ifdef __cplusplus
}
#endif
#endif /* CAVIAR_HIP_H */
