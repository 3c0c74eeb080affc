// Internal declarations shared by the HIP translation units of libcaviar_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <map>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/caviar_hip.h"
#include "cm_host.h"

#define CM_WAVE 64
#define CM_TILE 32          // tile edge of the blocked SPD solve
#ifndef CM_LD_PAD
#define CM_LD_PAD 0
#endif
//                  CM_LD_PAD: doubles added to the row stride of the solve matrices (8 was measured: no gain -- the trailing update is not channel-conflict bound)
#define CM_EV_PER_IT 12
#define CM_WSTAT 6             // doubles per row of cm_pdev::wstat
#define CM_MAX_BATCH 1024      // circuits in one lock-step group
#define CM_MAILBOX_BATCH0 32   // 64-bit word index of the first per-circuit mailbox of a batch
#define CM_STATUS_OFF (2 * (CM_MAILBOX_BATCH0 + CM_MAX_BATCH))   // int32 index of the end-of-run status words: 8 per circuit (n_act[0..3], chain rows, chain records, -, -)
#define CM_MAILBOX_WORDS (CM_STATUS_OFF + 8 * CM_MAX_BATCH)  // int32 words of host-mapped pinned memory per context: [0..1] single-circuit mailbox, [8..11] scratch, then one 64-bit word per circuit of a batch

void cm_set_error(const char* fmt, ...);

#define CM_HIP(call)                                                                          \
  do {                                                                                        \
    hipError_t _e = (call);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      cm_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(_e));      \
      return CM_ERR_HIP;                                                                      \
    }                                                                                         \
  } while (0)

#define CM_TRY(expr)              \
  do {                            \
    int _s = (expr);              \
    if (_s != CM_OK) return _s;   \
  } while (0)

// Device-memory pool of a context: the slabs of destroyed problems and the random tables of their runs are kept for the next problem of
// that size class instead of going back to the driver -- hipMalloc / hipFree are 50-200 us of driver time each (hipFree also waits for the
// device), four to six of them per circuit, which made creating and destroying the 256 circuits of a batch cost as much as fitting them.
// Only the MEMORY is reused, never contents.  Size classes are 12.5 % apart; idle bytes are capped (CAVIAR_HIP_POOL_GB, default 64).
struct cm_pool {
  std::mutex m;
  std::unordered_map<void*, size_t> live;          // handed out: pointer -> class size
  std::multimap<size_t, void*> idle;               // class size -> pointer
  size_t idle_bytes = 0, cap_bytes = (size_t)64 << 30;
};
int cm_pool_alloc(cm_ctx* ctx, void** out, size_t bytes);
void cm_pool_free(cm_ctx* ctx, void* ptr);        // (the caller has made sure no queued work still uses it)
void cm_pool_trim(cm_ctx* ctx, size_t keep_bytes);

// a host thread that runs one task at a time (cm_ctx::helper)
struct cm_helper {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::function<void()> task;
  bool has = false, quit = false, idle = true;
  cm_helper() {
    th = std::thread([this] {
      std::unique_lock<std::mutex> lk(m);
      for (;;) {
        cv.wait(lk, [this] { return has || quit; });
        if (quit) return;
        has = false;
        lk.unlock();
        task();
        lk.lock();
        idle = true;
        cv.notify_all();
      }
    });
  }
  void start(std::function<void()> f) { std::lock_guard<std::mutex> lk(m); task = std::move(f); has = true; idle = false; cv.notify_all(); }
  void wait() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [this] { return idle; }); }
  ~cm_helper() { { std::lock_guard<std::mutex> lk(m); quit = true; cv.notify_all(); } th.join(); }
};

struct cm_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t side = nullptr;     // side stream: update_phi(it) -> MC logit(it + 1), overlapped with sigma/z and the next solve
  bool own_stream = false;
  hipEvent_t ev[10] = {};
  int32_t* h_pinned = nullptr;   // host-mapped pinned memory: words [0..1] = 64-bit mailbox (seq | flag | active-set size); [8..11] scratch for copies
  int32_t* d_mailbox = nullptr;  // device address of h_pinned
  int mail_seq = 0;
  int mb_n = 0, mb_flag = 0;     // last values read from the mailbox (active-set size, not-SPD flag)
  bool use_mailbox = true;
  int solve_group = 0;           // circuits per group of the tiled solve of a batch (0: the whole batch in lock-step); CAVIAR_HIP_SOLVE_GROUP
  int xcd_mode = 3;              // cm_xcd_remap mode of the bandwidth-bound solve kernels; CAVIAR_HIP_XCD_MODE
  int chol_family = 1;           // tiled Cholesky above large_min tile rows: 0 = panel pairs, 1 = grouped (four panels per pass, 64 x 64 super-tiles) for batches and pairs for one circuit, 2 = grouped everywhere; CAVIAR_HIP_CHOL_FAMILY (bitwise the same factor)
  int num_cus = 256;
  void* h_stage = nullptr;        // pinned staging buffer (priors upload, fn_scan round trip), grown on demand
  size_t h_stage_bytes = 0;
  size_t lds_attr[80] = {};       // dynamic-LDS caps already raised for this context's device, by kernel slot (cm_raise_lds)
  long long* dbg_block0 = nullptr; // diagnostic buffer (CAVIAR_HIP_TRACE_BLOCK0): 64 launches x 8 device-clock stamps
  void* d_pdev = nullptr;         // descriptors of the current batch (device) and their pinned source
  void* h_pdev = nullptr;
  size_t pdev_cap = 0;
  std::vector<hipEvent_t> pool;   // per-iteration events (CM_EV_PER_IT per iteration): phase stamps + cross-stream edges
  // two lock-step batches at a time (cm_caviar_run_batch, cm_options.interleave): the second one runs on `twin` -- a context of its own on the
  // same device (stream, mailboxes, events, scratch) driven by `helper`, a host thread that lives as long as the twin (a fresh thread per call
  // would pay HIP's per-thread initialisation every time)
  cm_ctx* twin = nullptr;
  cm_helper* helper = nullptr;
  cm_ctx* root = nullptr;         // the context whose pool this one allocates from (itself; a twin: its parent)
  cm_pool mem;
};

// One circuit: encoded stimulus + observations + CAVIaR state.  All pointers are device memory.
struct cm_problem {
  cm_ctx* ctx = nullptr;
  int32_t N = 0, K = 0, T = 0;
  int64_t nnz = 0;       // number of targeted (stim > 0) entries
  int32_t V = 0;         // number of distinct stimulus values (sorted ascending; powers = vals[1:])
  int32_t ld = 0;        // leading dimension of the solve matrices (multiple of CM_TILE), sized for N
  int32_t max_row_len = 0; // longest CSR row (targeted trials of one neuron)
  int32_t max_col_len = 0; // most neurons targeted in one trial
  int32_t small_cap = 0;   // largest active set solved inside the fused mid-iteration kernel
  // --- static encoding (built once by cm_problem_create) ---
  double* y = nullptr;        // [K]  trapz(psc)
  double* ss = nullptr;       // [K]  sum psc^2
  double* sum_y = nullptr;    // [2]  sum(y), sum(y^2)
  double* vals = nullptr;     // [CM_MAX_VALS] sorted distinct stimulus values
  int32_t* row_ptr = nullptr; // [N+1] CSR over targeted entries
  int32_t* col = nullptr;     // [nnz] trial index
  uint8_t* grp = nullptr;     // [nnz] index into vals
  int32_t* rowcnt = nullptr;  // [N*CM_MAX_VALS] number of trials of row n with stim == vals[g] (targeted or not)
  int32_t* col_ptr = nullptr; // [K+1] CSC
  int32_t* csc_row = nullptr; // [nnz] neuron index
  int32_t* csc_pos = nullptr; // [nnz] position of the entry in CSR order
  uint8_t* csc_grp = nullptr; // [nnz] power code of the entry (grp[csc_pos[ce]])
  // host mirror of the CSR (for reconnect_spont_cells and dense expansion)
  bool have_mirror = false;       // the vectors below are filled (cm_host_mirror)
  void *slabA = nullptr, *slabB = nullptr;   // the two allocations behind every static / state pointer below (sizes from N, K; from nnz)
  unsigned long long* enc_table = nullptr; int32_t *enc_flags = nullptr, *enc_row_nnz = nullptr, *enc_col_cnt = nullptr, *enc_cursor = nullptr;   // scratch of the encoding (slab A)
  std::vector<int32_t> h_row_ptr, h_col, h_col_ptr, h_csc_row;
  std::vector<uint8_t> h_grp;
  std::vector<double> h_vals;
  std::vector<uint32_t> h_keys;   // host copy of the threefry key schedule (source of asynchronous uploads)
  // --- CAVIaR state ---
  uint8_t* mask = nullptr;    // [K] lam_mask
  double* ye = nullptr;       // [nnz] y[col[e]]: per-entry copy of the trial's response (y is constant during a run) -- the row kernels
  uint8_t* turn8 = nullptr;   // [nnz] the entry's ticket in the multi-wave chain (earlier chain rows on its trial), left by the pre-sweep's column role while every row is a chain row
  uint8_t* me = nullptr;      // [nnz] mask[col[e]]                 read it contiguously instead of gathering 8 (1) bytes per 64-byte line
  int32_t* qcnt = nullptr;    // [N*8] per power: unmasked trials << 16 | trials (compact copy for the closed-form gate of all-zero rows)
  int32_t* mcnt = nullptr;    // [N*CM_MAX_VALS] unmasked trials of row n at each stimulus value (per run: the mask depends on y_xcorr_thresh)
  uint8_t* touch = nullptr;   // [N] the pre-sweep wrote this (previously all-zero) row in the current iteration: k_stats must look at it
  double* slab = nullptr;     // one allocation: mu | beta | z | lam (so the false-negative scan moves them in one copy each way)
  double* pslab = nullptr;    // one allocation: mu_prior | beta_prior | phi_prior | phi_cov_prior | scal
  double* lam = nullptr;      // [nnz] CSR order
  double* mu = nullptr;       // [N]
  double* beta = nullptr;     // [N] posterior variance (reference's "beta")
  double* mu_prior = nullptr; // [N]
  double* beta_prior = nullptr; // [N]
  double* phi = nullptr;      // [N*2]
  double* phi_cov = nullptr;  // [N*4]
  int32_t* phi_rep = nullptr;  // [N] representatives of k_phi_dead (cm_pdev)
  double* phi_dead = nullptr; // [N*8] update_phi result of neuron n if its lam row is identically zero: phi [N*2] | phi_cov [N*4] | ndtr(-mean/sdev) [N*2]
  double* phi_prior = nullptr;
  double* phi_cov_prior = nullptr;
  double* scal = nullptr;     // [16] device scalars, see CM_S_*
  double* R = nullptr;        // [K]  sum_n mu_n lam_nk
  uint8_t* colany = nullptr;  // [K]  any_n lam_nk >= spont_orthogonality
  double* z = nullptr;        // [K]
  double* mcE = nullptr;      // [N*CM_MAX_VALS] Monte-Carlo logit per (neuron, stimulus value)
  double* rs = nullptr;       // [N*4] row stats: sum lam, sum lam^2, sum y*lam, (unused)
  int32_t* act_idx = nullptr; // [N] active neurons (compact list)
  int32_t* act_inv = nullptr; // [N] neuron -> compact index or -1
  int32_t* n_act = nullptr;   // [4] device: N_a, flags
  int32_t* alive = nullptr;   // [1 + N] count + list of rows with lam != 0 (batches: work list of the per-neuron role of k_iter_mid)
  int32_t *tstamp = nullptr, *tlist_s = nullptr, *tlist_r = nullptr, *tcnt = nullptr;   // listed trials of the column sums (cm_pdev)
  int32_t* order = nullptr;   // sweep order of the current iteration (points into orders_all)
  int32_t* orders_all = nullptr; // [iters*N] sweep orders of every iteration (device generated, or the explicit stream)
  int32_t* orders_tmp = nullptr; // [iters*N] ping-pong buffer of the shuffle rounds
  int32_t* inv_all = nullptr;    // [iters*N] inverse of every sweep order: position of neuron n in iteration it
  double* wtab = nullptr;        // [iters*N*2S*2] (ndtri(u), u) for every uniform of the run (data independent)
  size_t wtab_cap = 0;
  double* wstat = nullptr;        // statistics of the table rows (allocated with wtab)
  size_t wstat_rows = 0;          // rows (iterations x N) wstat is sized for
  int32_t orders_cap = 0;     // iterations the two buffers are sized for
  int32_t* seq_n = nullptr;   // [N]   neurons with mu != 0 in sweep order
  int32_t* seq_off = nullptr; // [N+1] record offset of each such row in `packed`
  int32_t* seq_pos = nullptr; // [N]   position of neuron n in the chain, -1 if mu_n == 0
  int32_t* chunk_first = nullptr; // [n_chunk_cap] first row of each LDS chunk
  int32_t* seq_meta = nullptr;// [4]   nseq, total records, nchunks
  void* packed = nullptr;     // [(N + nnz) * 32 B] packed row records of the sequential chain
  uint32_t* keys = nullptr;   // [iters*(N+3)*2] threefry key schedule (host computed)
  double* uexp = nullptr;     // explicit uniforms (device copy) or nullptr
  double* A = nullptr;        // [ld*ld] precision matrix / Cholesky factor
  double* X = nullptr;        // [ld*ld] inverse factor
  double* Lf = nullptr;       // [ld*ld] Cholesky factor (panel tiles below the diagonal)
  double* Dinv = nullptr;     // [ld*CM_TILE] inverses of the diagonal tiles
  double* rhs = nullptr;      // [ld*2]
  // histories
  int32_t hist_iters = 0;
  double *h_mu = nullptr, *h_beta = nullptr, *h_lam = nullptr, *h_shape = nullptr, *h_rate = nullptr,
         *h_phi = nullptr, *h_phi_cov = nullptr, *h_z = nullptr;
  bool ran = false;
  int32_t keys_iters = 0;
};

enum {
  CM_S_SHAPE = 0, CM_S_RATE = 1, CM_S_SPONT_RATE = 2, CM_S_SHAPE_PRIOR = 3, CM_S_RATE_PRIOR = 4, CM_S_ERR = 5,
  CM_S_COUNT = 16
};

// ------------------------------------------------------------------------------------------------
// Device view of one circuit: every pointer and size the per-iteration kernels need.  A kernel receives it either BY VALUE in its
// arguments (one circuit: nothing sits between the launch and the first load, as before) or as an ARRAY in HBM indexed by
// blockIdx.y (a lock-step batch of circuits: the circuit is the outer grid dimension of every launch).  Quantities that change
// from iteration to iteration travel as ordinary kernel arguments and are common to all circuits of a batch.
// ------------------------------------------------------------------------------------------------
struct cm_pdev {
  int N, K, V, small_cap, max_row_len, ld_alloc;
  int mce_ld;                    // row stride of mcE: the smallest power of two >= V (the logits of a row sit in one cache line for V <= 8)
  long long nnz;
  const double *y, *ss, *sum_y, *vals;
  const int *row_ptr, *col;
  const uint8_t* grp;
  const int *rowcnt, *col_ptr, *csc_row, *csc_pos;
  const uint8_t* csc_grp;
  uint8_t *mask, *colany;
  const double* ye; const uint8_t* me;      // per-entry copies of y and mask (refreshed by k_init_state / cm_launch_entry_copies)
  uint8_t* turn8;                           // per-entry tickets of the multi-wave chain (cm_col_rank16x2)
  const int* mcnt; uint8_t* touch;          // unmasked trials per (row, value); rows the pre-sweep revived this iteration
  const int* qcnt;                          // [N*8] power q of row n: unmasked trials << 16 | trials (quiescent_gate; valid for <= 8 powers, rows < 65536 trials)
  double *lam, *mu, *beta, *phi, *phi_cov, *scal, *R, *z, *mcE, *rs;
  double *phi_dead, *phi_cov_dead, *phi_c_dead;   // update_phi of an all-zero row, per neuron, and ndtr(-mean / sdev) of it: constants of a run (k_phi_dead)
  int* phi_rep;                  // [N] first neuron with the same trial counts and phi priors (itself: a representative whose constants are computed; the others copy)
  const double *pslab, *mu_prior, *beta_prior, *phi_prior, *phi_cov_prior;
  int *act_idx, *act_inv, *n_act, *seq_n, *seq_off, *seq_pos, *chunk_first, *seq_meta;
  int* alive;                    // [1 + N] batches: number and list of the rows with lam != 0 after the sweep (k_stats)
  // Trials the column sums R = lam^T mu / colany have to visit (quiescent regime): only trials of rows with lam != 0 can contribute, and
  // k_stats walks exactly those rows.  Double-buffered by iteration parity q = it & 1: tstamp[q K + k] = 2 it if k_stats(it - 1) listed
  // trial k for iteration it (tlist_s[q K + ..], tcnt[q] entries), 2 it + 1 if the pre-sweep of iteration it added it for a row it
  // revived (tlist_r[q K + ..], tcnt[2 + q] entries), anything else: not listed -- its sums are zero.
  int *tstamp, *tlist_s, *tlist_r, *tcnt;
  void* packed;
  double *A, *X, *Lf, *Dinv, *rhs;
  int* mailbox;                  // this circuit's 64-bit word in host-mapped pinned memory (nullptr: no mailbox)
  // random tables: data independent, shared by all circuits of a batch (same seed, N, S -> same streams)
  const int *orders_all, *inv_all;
  const double *wtab, *uexp;
  const double* wstat;           // [iters x N x CM_WSTAT] per table row (stored by neuron): mean and max of the phi_0 quantiles, mean and min of the phi_1 quantiles, smallest phi_0 uniform (k_ndtri_table)
  const uint32_t* keys;
};

#ifdef __HIPCC__
template <bool BATCH> struct cm_parg;
template <> struct cm_parg<false> {
  cm_pdev d;
  __device__ __forceinline__ const cm_pdev& get() const { return d; }
  __device__ __forceinline__ const cm_pdev& get(int) const { return d; }
};
template <> struct cm_parg<true> {
  const cm_pdev* __restrict__ arr;
  __device__ __forceinline__ const cm_pdev& get() const { return arr[blockIdx.y]; }
  __device__ __forceinline__ const cm_pdev& get(int c) const { return arr[c]; }
};
// XCD-aware workgroup -> (tile, circuit) map for the bandwidth-bound kernels of a batch.  Workgroups are dealt to the 8 XCDs (each
// with its own 4 MB L2) round-robin by their linear index; with the plain (blockIdx.x, blockIdx.y) = (tile, circuit) reading,
// the tiles of one circuit land on all eight and every panel tile they share is fetched into eight L2s.  Remapped, circuit c runs
// entirely on XCD c mod 8 (needs gridDim.y % 8 == 0, otherwise the plain map).  Placement is a speed heuristic only: any map is correct.
// mode 0: plain; 1: the circuits of an XCD interleaved (workgroup by workgroup); 2: circuit-major (an XCD works through one of its
// circuits after the other: the panel tiles in its L2 are those of one or two circuits instead of all of them); 3: circuit-major after
// the block 0 of every circuit.
__device__ __forceinline__ void cm_xcd_remap(int& bx, int& c, int mode) {
  // (32-bit unsigned arithmetic throughout: a 64-bit division is ~150 scalar instructions in front of the workgroup's first load)
  const unsigned nx = gridDim.x, B = gridDim.y;
  bx = (int)blockIdx.x; c = (int)blockIdx.y;
  if ((B & 7u) || mode == 0) return;
  const unsigned L = blockIdx.y * nx + blockIdx.x;
  const unsigned xcd = L & 7u, per = B >> 3, q = L >> 3;
  if (mode == 1) { const unsigned d = q / per; c = (int)(xcd + 8u * (q - d * per)); bx = (int)d; }
  else if (mode == 2 || nx < 2u) { const unsigned d = q / nx; c = (int)(xcd + 8u * d); bx = (int)(q - d * nx); }
  else if (q < per) { c = (int)(xcd + 8u * q); bx = 0; }   // mode 3: every circuit's block 0 (the serial diagonal-tile factorisation) first,
  else { const unsigned q2 = q - per, d = q2 / (nx - 1u); c = (int)(xcd + 8u * d); bx = 1 + (int)(q2 - d * (nx - 1u)); }   // then circuit-major
}
#endif

// LDS footprint of the one-workgroup solver (solve_small_body, cm_bodies.h) for n active rows with tot_cap staged entries
#define CM_SMALL_MAX 128
__host__ __device__ static inline size_t small_lds_bytes(int n, int tot_cap) {
  return ((size_t)n * (n + 1) + 3 * CM_SMALL_MAX + tot_cap) * sizeof(double) + ((size_t)tot_cap + 2 * CM_SMALL_MAX + 1) * sizeof(int);
}

// Which solver handles block_update_mu for an active set of `na` rows (host and device agree by construction):
//   0  nothing to do here: the fused mid-iteration kernel has already solved it in LDS (`skip_solved`)
//   1  the one-workgroup LDS solver (sweep-operator inverse, ~10 + 2 n microseconds: small sets only)
//   2  the tiled factorisation (~50 microseconds for one tile, + ~8 per further tile row)
// `small_cap` is the circuit's largest LDS-solved set (cm_small_cap): the measured crossover of the two, bounded by the LDS budget.
#define CM_FUSED_SOLVE_MAX 24
__host__ __device__ static inline int cm_solve_kind(int na, int small_cap, bool skip_solved) {
  if (na <= small_cap) return skip_solved ? 0 : 1;
  return 2;
}

// Chain rows (cm_sweep.hip) of up to this many trials travel as packed records: half a record buffer of `cap` records, so that a chunk of the
// chain still holds at least as many records again.  Which kernel can walk the chain: 1 / 2 = the multi-wave ticketed kernel with 1024-record
// buffers + 32-bit tickets (K up to ~5600 trials, rows up to 511 trials) / 512-record buffers + 8-bit tickets (K up to ~10900, rows up to 255);
// 0 = the single-consumer kernel (anything else: longer rows, more than 255 neurons per trial, K beyond the LDS budget).
static inline int cm_pack_limit(int cap) { return cap / 2 - 1; }
// A DENSE dependency graph leaves the tickets nothing to overlap: two chain rows share on average (trials per row) x (targets per trial - 1) /
// (N - 1) trials -- 0.18 at N = 1000 / K = 2000 / H = 10 (critical path ~130 of 1000 rows), 8 at the reference's demo shape N = 200 / K = 3600
// (every row waits for its predecessor: measured 143 us per sweep on the multi-wave kernel against 119 us on the single-consumer kernel).  Above
// 4 shared trials the single-consumer kernel walks the chain.  (Any choice is correct: the kernels give bitwise the same sweep.)
static inline int cm_par_class(int N, int K, int max_row_len, int max_col_len, long long nnz) {
  if (max_col_len > 255 || nnz >= (1 << 24)) return 0;
  if (N > 1 && K > 0 && ((double)nnz / N) * ((double)nnz / K - 1.0) / (double)(N - 1) > 4.0) return 0;
  if ((size_t)K * 12 <= 68 * 1024 && max_row_len <= cm_pack_limit(1024)) return 1;
  if ((size_t)K * 9 + 16 <= 96 * 1024 && max_row_len <= cm_pack_limit(512)) return 2;
  return 0;
}

// The circuits one launch serves: a single problem, or a lock-step group with identical launch configuration.
struct cm_set {
  int B = 1;
  cm_problem* lead = nullptr;            // sizes and random tables (identical over the group)
  std::vector<cm_problem*> ps;
  cm_pdev h0;                            // B == 1: the descriptor, passed by value
  const cm_pdev* d_arr = nullptr;        // B > 1: descriptors in HBM
  int small_cap_max = 0, max_row_len_max = 0, max_col_len_max = 0;
  size_t small_lds_max = 0;              // largest LDS need of the fused in-LDS solve over the circuits of the set
  long long nnz_max = 0, nnz_min = 0;
  long long fact_launches = 0, fact_tile_updates = 0, fact_panel_tiles = 0, fact_tri_tiles = 0, fact_products = 0, inv_products = 0;   // counters of the tiled solves (cm_timings)
  bool last_fact = false;                // the last cm_solve_launch ran a factorisation (and recorded its event pair)
  bool lists_ok = false;                 // every prior mean of the set is zero: a row outside the active set has mu = 0, only rows with lam != 0 feed R (run_set)
  int tlist_it = -1;                     // iteration for which k_stats listed the trials of the rows with lam != 0 (-1: no list)
  int tlist_sweep_it = -1;               // iteration whose pre-sweep ran on that list (and appended the trials of revived rows)
  int scan_it = -2;                      // iteration whose chain tables block 0 of the last k_iter_mid leaves in every circuit that takes its in-LDS solve (-2: none)
  std::vector<int> na;                   // last active-set sizes read from the mailboxes
  int na_max = 0;
};
void cm_fill_pdev(const cm_problem* p, const cm_problem* rng_owner, int* mailbox, cm_pdev* d);
int cm_set_single(cm_ctx* ctx, cm_problem* p, cm_set* s);
bool cm_same_launch_config(const cm_problem* a, const cm_problem* b);

// launch helpers: KERN is a kernel template whose FIRST template argument is BATCH; CM_LAUNCH_SET_T passes further template
// arguments in parentheses, e.g. CM_LAUNCH_SET_T(k_iter_mid, (4, true), set, ...)
#define CM_UNPAREN(...) __VA_ARGS__
#define CM_LAUNCH_SET(KERN, set, gridx, block, lds, st, ...)                                                                               \
  do {                                                                                                                                    \
    if ((set).B == 1) hipLaunchKernelGGL((KERN<false>), dim3(gridx), dim3(block), lds, st, cm_parg<false>{(set).h0}, ##__VA_ARGS__);       \
    else hipLaunchKernelGGL((KERN<true>), dim3(gridx, (set).B), dim3(block), lds, st, cm_parg<true>{(set).d_arr}, ##__VA_ARGS__);          \
  } while (0)
#define CM_LAUNCH_SET_T(KERN, TARGS, set, gridx, block, lds, st, ...)                                                                      \
  do {                                                                                                                                    \
    if ((set).B == 1) hipLaunchKernelGGL((KERN<false, CM_UNPAREN TARGS>), dim3(gridx), dim3(block), lds, st, cm_parg<false>{(set).h0}, ##__VA_ARGS__); \
    else hipLaunchKernelGGL((KERN<true, CM_UNPAREN TARGS>), dim3(gridx, (set).B), dim3(block), lds, st, cm_parg<true>{(set).d_arr}, ##__VA_ARGS__);    \
  } while (0)

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
#ifdef __HIPCC__
// ---- DPP cross-lane adds (VALU data path; no LDS crossbar).  All 64 lanes must be active at the call site.
#define CM_DPP_XOR1 0xB1          // quad_perm [1,0,3,2]
#define CM_DPP_XOR2 0x4E          // quad_perm [2,3,0,1]
#define CM_DPP_HALF_MIRROR 0x141  // lane i <-> 7 - i within 8
#define CM_DPP_MIRROR 0x140       // lane i <-> 15 - i within 16
template <int CTRL>
__device__ __forceinline__ double cm_dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double cm_readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
// sum over aligned segments of SEG lanes (SEG = 1,2,4,8,16 by DPP; 32, 64 add LDS-crossbar steps); every lane of a
// segment receives the segment's sum; fixed association -> deterministic
template <int SEG>
__device__ __forceinline__ double cm_seg_sum(double v) {
  if (SEG >= 2) v += cm_dpp_f64<CM_DPP_XOR1>(v);
  if (SEG >= 4) v += cm_dpp_f64<CM_DPP_XOR2>(v);
  if (SEG >= 8) v += cm_dpp_f64<CM_DPP_HALF_MIRROR>(v);
  if (SEG >= 16) v += cm_dpp_f64<CM_DPP_MIRROR>(v);
  if (SEG >= 32) v += __shfl_xor(v, 16, CM_WAVE);
  if (SEG >= 64) v += __shfl_xor(v, 32, CM_WAVE);
  return v;
}
// full-wave sum, result uniform: four DPP steps give every 16-lane row its sum, the four row sums are read to scalars
__device__ __forceinline__ double cm_wave_sum(double v) {
  v = cm_seg_sum<16>(v);
  const double r0 = cm_readlane_f64(v, 0), r1 = cm_readlane_f64(v, 16), r2 = cm_readlane_f64(v, 32), r3 = cm_readlane_f64(v, 48);
  return (r0 + r1) + (r2 + r3);
}
// Sum over each half, every lane receiving its half's sum.  Bitwise what cm_wave_sum gives the same row alone in a wavefront (there
// the upper two 16-lane partial sums are zeros and (r0 + r1) + (0 + 0) == r0 + r1).
__device__ __forceinline__ double cm_half_sum(double v) {
  v = cm_seg_sum<16>(v);
  const double r0 = cm_readlane_f64(v, 0), r1 = cm_readlane_f64(v, 16), r2 = cm_readlane_f64(v, 32), r3 = cm_readlane_f64(v, 48);
  return (threadIdx.x & 32) ? (r2 + r3) : (r0 + r1);
}
__device__ __forceinline__ int cm_wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, CM_WAVE);
  return v;
}
__device__ __forceinline__ double cm_wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, CM_WAVE));
  return v;
}

// jax.nn.sigmoid / scipy expit: 1 / (1 + exp(-x))
__device__ __forceinline__ double cm_sigmoid(double x) { return 1.0 / (1.0 + exp(-x)); }

// numpy nan_to_num for float64: nan -> 0, +-inf -> +-DBL_MAX
__device__ __forceinline__ double cm_nan_to_num(double x) {
  if (isnan(x)) return 0.0;
  if (isinf(x)) return x > 0 ? 1.7976931348623157e308 : -1.7976931348623157e308;
  return x;
}

// Threefry-2x32, 20 rounds (Salmon et al. 2011), the block function behind jax.random in the 0.3 series.
__host__ __device__ __forceinline__ void cm_threefry2x32(uint32_t k0, uint32_t k1, uint32_t& x0, uint32_t& x1) {
  const uint32_t ks0 = k0, ks1 = k1, ks2 = k0 ^ k1 ^ 0x1BD11BDAu;
#define CM_TF_R(r)                      \
  x0 += x1;                             \
  x1 = (x1 << (r)) | (x1 >> (32 - (r))); \
  x1 ^= x0;
  x0 += ks0; x1 += ks1;
  CM_TF_R(13) CM_TF_R(15) CM_TF_R(26) CM_TF_R(6)
  x0 += ks1; x1 += ks2 + 1u;
  CM_TF_R(17) CM_TF_R(29) CM_TF_R(16) CM_TF_R(24)
  x0 += ks2; x1 += ks0 + 2u;
  CM_TF_R(13) CM_TF_R(15) CM_TF_R(26) CM_TF_R(6)
  x0 += ks0; x1 += ks1 + 3u;
  CM_TF_R(17) CM_TF_R(29) CM_TF_R(16) CM_TF_R(24)
  x0 += ks1; x1 += ks2 + 4u;
  CM_TF_R(13) CM_TF_R(15) CM_TF_R(26) CM_TF_R(6)
  x0 += ks2; x1 += ks0 + 5u;
#undef CM_TF_R
}
#else
inline void cm_threefry2x32(uint32_t k0, uint32_t k1, uint32_t& x0, uint32_t& x1) {
  const uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
  static const int R[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
  x0 += ks[0]; x1 += ks[1];
  for (int g = 0; g < 5; ++g) {
    for (int i = 0; i < 4; ++i) { int r = R[g % 2][i]; x0 += x1; x1 = (x1 << r) | (x1 >> (32 - r)); x1 ^= x0; }
    x0 += ks[(g + 1) % 3]; x1 += ks[(g + 2) % 3] + (uint32_t)(g + 1);
  }
}
#endif

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, device), and setting it REPLACES the previous value (it is
// not a running maximum).  Several contexts can live on one device (cm_batch, fit_batch workers), so the value in force is kept
// in ONE process-wide table keyed (device, kernel) under a mutex and is only ever raised; the per-context array is a cache of
// lower bounds of that table (valid because the table never decreases), so the common launch pays no lock and no runtime call.
enum cm_lds_slot { CM_LDS_ITER = 0 /* 7 x 2 x 2 template variants */, CM_LDS_SOLVE_SMALL = 28, CM_LDS_SWEEP_PAR, CM_LDS_SWEEP_PAR2, CM_LDS_SWEEP_SEQ,
                   CM_LDS_NWD, CM_LDS_SIM_SORT, CM_LDS_BUILD_PREC, CM_LDS_SWEEP_PAR_B, CM_LDS_SWEEP_PAR2_B, CM_LDS_SWEEP_SEQ_B,
                   CM_LDS_SOLVE_SMALL_B, CM_LDS_BUILD_PREC_B, CM_LDS_ITER_LISTED /* 7 SEG variants */, CM_LDS_MC_DEAD = 47, CM_LDS_MC_DEAD_B, CM_LDS_ITER_OVERFLOW /* 7 SEG variants */, CM_LDS_CHOL_DIAG = 56, CM_LDS_CHOL_GPANEL, CM_LDS_CHOL_WIDE,
                   CM_LDS_CHOL_DIAG_B, CM_LDS_CHOL_GPANEL_B, CM_LDS_CHOL_WIDE_B, CM_LDS_FN_SCAN, CM_LDS_FN_SCAN_B, CM_LDS_FN_SCAN_G, CM_LDS_FN_SCAN_GB, CM_LDS_PHI_REP, CM_LDS_PHI_REP_B, CM_LDS_ITER_FLAT /* 7 SEG variants */ = 68, CM_LDS_SLOTS = 75 };
static_assert(CM_LDS_SLOTS <= 80, "cm_ctx::lds_attr");
// Table update AND runtime call under ONE lock: `set(fn, bytes, user)` is invoked while the table's mutex is held, and the table takes
// the new value only after the setter succeeded -- two host threads (contexts on one GPU) can therefore never apply their values in
// the opposite order of their table updates (the smaller request landing last would leave the attribute below what the table and both
// per-context caches claim, for the rest of the process).  Returns CM_OK, or the setter's error with the table unchanged.
// *applied (optional): the value the setter was called with, 0 if the value in force already covered the request.
typedef int (*cm_lds_setter)(const void* fn, size_t bytes, void* user);
int cm_lds_table_raise(int device, const void* fn, size_t bytes, size_t floor_bytes, cm_lds_setter set, void* user, size_t* applied);
int cm_lds_set_attribute(const void* fn, size_t bytes, void*);      // the real setter: hipFuncSetAttribute(MaxDynamicSharedMemorySize)
inline int cm_raise_lds(cm_ctx* ctx, int slot, const void* fn, size_t bytes, size_t floor_bytes) {
  if (bytes <= floor_bytes || bytes <= ctx->lds_attr[slot]) return CM_OK;
  CM_TRY(cm_lds_table_raise(ctx->device, fn, bytes, floor_bytes, cm_lds_set_attribute, nullptr, nullptr));
  ctx->lds_attr[slot] = bytes;
  return CM_OK;
}

// ------------------------------------------------------------------------------------------------
// phase launchers (each enqueues kernels on ctx->stream; no host sync unless stated)
// ------------------------------------------------------------------------------------------------
int cm_launch_psc_prologue(cm_ctx* ctx, int K, int T, const double* d_psc, double* d_y, double* d_ss);
// one circuit of a group being encoded (cm_problem_create_batch): inputs, and where the pieces of its two slabs are
struct cm_enc {
  const double *psc, *stim;
  int N, K, T;
  long long nnz;                                   // filled by the host between the stages
  double *y, *ss, *vals, *mcE;
  unsigned long long* table;
  int *flags, *rowcnt, *row_nnz, *row_ptr, *col_cnt, *cursor, *col_ptr, *col, *csc_row, *csc_pos, *alive;
  uint8_t *grp, *csc_grp;
};
int cm_encode_stage1(cm_ctx* ctx, const cm_enc* d_arr, int n, int N_max, int K_max, int* d_report);                       // enqueues; the caller copies the report and waits
int cm_encode_stage2(cm_ctx* ctx, const cm_enc* d_arr, int n, int N_max, int K_max, long long nnz_max, int* d_report);    // enqueues the nnz-sized part
int cm_host_mirror(cm_ctx* ctx, cm_problem* p);                                      // host copy of the compressed rows, on first use
int cm_launch_entry_copies(cm_ctx* ctx, cm_set& s);                                  // ye / me from the current y / mask (after a test hook uploaded either)
int cm_launch_init_state(cm_ctx* ctx, cm_set& s, double y_xcorr_thresh);             // state from the uploaded priors, lam0, mask, z, counters
int cm_launch_scatter_dense_lam(cm_ctx* ctx, cm_problem* p, const double* d_lam_dense);
int cm_launch_expand_lam(cm_ctx* ctx, cm_problem* p, const double* d_lam_sparse, double* d_dense);

int cm_solve_mu(cm_ctx* ctx, cm_set& s, int* n_active_out);                          // syncs once (active-set size)
int cm_solve_launch(cm_ctx* ctx, cm_set& s, int skip_mode, hipEvent_t fact_begin = nullptr, hipEvent_t fact_end = nullptr);                          // sizes from s.na; skip_mode 0: circuits the fused kernel solved are left alone, -1: solve all
int cm_small_cap(const cm_problem* p);
int cm_launch_iter_mid(cm_ctx* ctx, cm_set& s, int it, int iters, int S, int rng_mode, const cm_options* o, int do_sigma, int do_phi, int roles = 3,
                       hipEvent_t between = nullptr);   // between: recorded between k_iter_mid and k_mc_dead (batches)
int cm_launch_ndtri_table(cm_ctx* ctx, cm_problem* p, int it0, int n_it, int S, int rng_mode, hipStream_t st);
int cm_launch_rng(cm_ctx* ctx, cm_problem* p, int it0, int n_it, hipStream_t st);
int cm_launch_sweep(cm_ctx* ctx, cm_set& s, int it, const cm_options* o, hipEvent_t mid, int table_it = -1, int fused_pack = 0);   // table_it: iteration whose sweep order is used (default: it); fused_pack: the chain tables exist (cm_set::scan_it), the pre-sweep packs
void cm_sweep_chunking(const cm_set& s, int par, int* cap, int* pack_max, int* gran);
int cm_par_min_rows();
int cm_launch_sum_y(cm_ctx* ctx, cm_set& s);
int cm_sweep_variant(const cm_set& s, int rows_hint);
int cm_launch_stats(cm_ctx* ctx, cm_set& s, double spont_orthogonality, int quiescent = 0, int it = -1);   // quiescent: cm_sweep_quiescent() of the sweep just launched
int cm_sweep_quiescent(const cm_ctx* ctx, const cm_set& s, int it, const cm_options* o);
int cm_launch_post_single(cm_ctx* ctx, cm_set& s, int it, const cm_options* o, int do_sigma);
int cm_wait_active(cm_ctx* ctx, cm_set& s);
int cm_launch_update_phi(cm_ctx* ctx, cm_set& s, hipStream_t st, int use_dead = 0, int screen = 0);   // screen: rs holds the statistics of the last sweep (final update of a run)
int cm_launch_phi_dead(cm_ctx* ctx, cm_set& s, hipStream_t st);
bool cm_fn_scan_fits(const cm_set& s);
int cm_launch_fn_scan(cm_ctx* ctx, cm_set& s, double minimax_spk_prob, double minimum_spike_count);   // reconnect_spont_cells on the device (cm_fnscan.hip)

