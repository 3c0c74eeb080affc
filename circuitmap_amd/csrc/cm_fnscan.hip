// reconnect_spont_cells (reference circuitmap/optimise/caviar.py:102-144) on the device: ONE workgroup per circuit, straight on the
// circuit's state in HBM -- mu, beta, z are edited in place and the reconnected cells' lam entries set to 1 -- so a run no longer stops
// for a host scan between the last iteration and the final update_phi (it was 1.3 ms per 128-circuit step with the GPU idle, a host
// thread pool and a stream synchronisation).  The host implementation (cm_host.cpp: cm_host_reconnect) remains the reference for
// the sanitizer build and serves verbose runs (the reference's progress messages).
//
// The reference loop: while cells with mu == 0 remain and more than minimum_spike_count spontaneous events (z != 0) exist, take the cell
// with the most events on its targeted trials (first maximum in ascending cell order), compute per power the rate of events on its
// trials, and reconnect it if the isotonic last value reaches the threshold and it has minimum_spike_count events:
// mu = mean(z[locs]), beta = sem(z[locs]) (ddof = 1: NaN for one event), lam[cell, locs] = 1, z[locs] = 0; the cell leaves the list either
// way.  A cell below minimum_spike_count events can never reconnect and counts only ever fall, so such cells are dropped without being
// examined (identical outcome, as in cm_host_reconnect).  Event counts per candidate live in LDS, built from the events' side (a
// few percent of the trials carry one; each lists its targeted cells through the CSC) with integer atomics -- exact, order-free.
// Every floating-point sum is taken by one thread in the row's CSR order, as the host takes it: the results are bit-identical.
#include <stdlib.h>

#include "cm_bodies.h"

#define CM_FN_THREADS 256

// Layout of the loop.  What the reference does one cell at a time -- examine the cell with the most events, reject or reconnect it, drop it
// from the list -- only changes state when a cell RECONNECTS (z loses events, other cells' counts fall).  Between two reconnections
// every examined cell sees the same z, so all candidates are evaluated at once (one thread each: per power, events on the cell's
// trials / trials of that power, isotonic last value, the two tests), the first one in the reference's order (most events, lowest cell
// on ties) that passes is the next reconnection, and the candidates ahead of it are dropped as rejected -- exactly the cells the reference
// would have examined and rejected before reaching it.  One round per reconnection (a handful per circuit) instead of one per candidate
// (dozens, each a chain of dependent loads).  The per-power event counts come from the events' side, like the totals: LDS counters
// hits[cell][power], integer atomics; the trials of each power are the static rowcnt.
#define CM_FN_MAXP 8              // powers the kernel handles (per-thread isotonic stacks in LDS); more: the host path

// A round costs what its dependent global loads cost, so the CACHED instantiation keeps in LDS everything a round reads more than once --
// z, the row / column pointers, the trials per (cell, power) -- and the per-cell verdict "would reconnect" (acc), which only changes
// for the cells a reconnection touches (they are re-examined from a dirty list); a round is then: best accepting key (LDS scan), drop the
// cells ahead of it, one trip for the cell's trial indices, one for the column entries of its events, one for their power codes.
// Shapes that do not fit (N = 5000 / K = 10000: 270 KB) run the same code on the global arrays with every verdict re-taken per round.
//
// dynamic LDS (8-byte aligned blocks): cnt [N] int | list [N] int | hits [N * P] u16 | row trial [cap] int | row z [cap] f64 |
//   amplitudes [cap] f64 | stacks [CM_FN_THREADS][2][CM_FN_MAXP] f64 | CACHED: acc [N] u8 | rcnt [N * P] u16 | rptr [N + 1] int | cptr [K + 1] int | zl [K] f64
__host__ __device__ static inline size_t fn_r8(size_t b) { return (b + 7) & ~(size_t)7; }
__host__ __device__ static inline size_t fn_scan_lds(int N, int K, int P, int max_row_len, bool cached) {
  size_t b = fn_r8((size_t)N * 4) * 2 + fn_r8((size_t)N * P * 2) + fn_r8((size_t)max_row_len * 4) + (size_t)max_row_len * 16 +
             (size_t)CM_FN_THREADS * 2 * CM_FN_MAXP * 8;
  if (cached) b += fn_r8((size_t)N) + fn_r8((size_t)N * P * 2) + fn_r8(((size_t)N + 1) * 4) + fn_r8(((size_t)K + 1) * 4) + (size_t)K * 8;
  return b + 16;
}

template <bool BATCH, bool CACHED>
__global__ __launch_bounds__(CM_FN_THREADS) void k_fn_scan(cm_parg<BATCH> pa, double minimax_spk_prob, double min_count, int row_cap, int maxcol) {
  const cm_pdev& p = pa.get();
  extern __shared__ double fn_sm[];
  const int N = p.N, K = p.K, P = p.V - 1, tid = threadIdx.x;
  char* base = reinterpret_cast<char*>(fn_sm);
  int* cnt = reinterpret_cast<int*>(base); base += fn_r8((size_t)N * 4);     // events on the cell's trials; -1: not a candidate (connected, dropped, examined)
  int* list = reinterpret_cast<int*>(base); base += fn_r8((size_t)N * 4);    // candidates of the round / cells to re-examine
  unsigned short* hits = reinterpret_cast<unsigned short*>(base); base += fn_r8((size_t)N * P * 2);   // [N][P] events on the cell's trials of power i + 1
  int* rk = reinterpret_cast<int*>(base); base += fn_r8((size_t)row_cap * 4);
  double* rz = reinterpret_cast<double*>(base); base += (size_t)row_cap * 8;
  double* amp = reinterpret_cast<double*>(base); base += (size_t)row_cap * 8;
  double* stk = reinterpret_cast<double*>(base); base += (size_t)CM_FN_THREADS * 2 * CM_FN_MAXP * 8;
  uint8_t* acc = reinterpret_cast<uint8_t*>(base); if (CACHED) base += fn_r8((size_t)N);              // bit 0: would reconnect; bit 1: on the dirty list
  unsigned short* rcnt = reinterpret_cast<unsigned short*>(base); if (CACHED) base += fn_r8((size_t)N * P * 2);
  int* rptr = reinterpret_cast<int*>(base); if (CACHED) base += fn_r8(((size_t)N + 1) * 4);
  int* cptr = reinterpret_cast<int*>(base); if (CACHED) base += fn_r8(((size_t)K + 1) * 4);
  double* zl = reinterpret_cast<double*>(base);
  __shared__ int nz, nlist;
  __shared__ unsigned long long best_sh;
  const double* __restrict__ mu = p.mu;
  double* __restrict__ z = p.z;
  const int* __restrict__ row_ptr = p.row_ptr;
  const int* __restrict__ col = p.col;
  const uint8_t* __restrict__ grp = p.grp;
  const int* __restrict__ rowcnt = p.rowcnt;
  const int* __restrict__ col_ptr = p.col_ptr;
  const int* __restrict__ csc_row = p.csc_row;
  const uint8_t* __restrict__ csc_grp = p.csc_grp;
  auto hit_word = [&](int n, int g, unsigned& inc) -> unsigned* {             // 16-bit counters, two per LDS word
    const size_t i = (size_t)n * P + (g - 1);
    inc = (i & 1) ? 0x10000u : 1u;
    return reinterpret_cast<unsigned*>(hits) + (i >> 1);
  };
  // the reference's examination of one cell against the current z (:121-130): srates straight into the isotonic walk (pava.py:9-88,
  // gamma = 1, strict '>' merge, last value only), then the two tests
  auto would_reconnect = [&](int n) -> bool {
    double* sv = stk + (size_t)tid * 2 * CM_FN_MAXP;
    double* sw = sv + CM_FN_MAXP;
    int top = -1;
    double spike_count = 0.0;
    for (int q = 0; q < P; ++q) {
      const int trials = CACHED ? (int)rcnt[(size_t)n * P + q] : rowcnt[(size_t)n * CM_MAX_VALS + q + 1];
      const int hit = hits[(size_t)n * P + q];
      const double y = trials > 0 ? (double)hit / (double)trials : 0.0;
      if (trials > 0) spike_count += (double)hit;
      ++top;
      sv[top] = y; sw[top] = 1.0;
      while (top > 0 && (sv[top - 1] / sw[top - 1]) > (sv[top] / sw[top])) { --top; sv[top] += sv[top + 1]; sw[top] += sw[top + 1]; }
    }
    const double pava = P > 0 ? sv[top] / sw[top] : 0.0;                       // :128
    return pava >= minimax_spk_prob && spike_count >= min_count;              // :130
  };
  auto key_of = [](int c, int n) -> unsigned long long {                      // most events first, lowest cell on ties (:117 first maximum)
    return ((unsigned long long)(unsigned)(c + 1) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)n);
  };
  if (tid == 0) nz = 0;
  for (int n = tid; n < N; n += CM_FN_THREADS) cnt[n] = (mu[n] == 0.0) ? 0 : -1;           // :103 exact zero test (quirk 7)
  for (int i = tid; i < N * P; i += CM_FN_THREADS) hits[i] = 0;
  if (CACHED) {
    for (int n = tid; n <= N; n += CM_FN_THREADS) rptr[n] = row_ptr[n];
    for (int k = tid; k <= K; k += CM_FN_THREADS) cptr[k] = col_ptr[k];
    for (int k = tid; k < K; k += CM_FN_THREADS) zl[k] = z[k];
    for (int i = tid; i < N * P; i += CM_FN_THREADS) rcnt[i] = (unsigned short)rowcnt[(size_t)(i / P) * CM_MAX_VALS + (i % P) + 1];
  }
  __syncthreads();
  // events -> counts, flattened over (trial, position in the trial's column): every item's loads are independent of every other's
  for (int t0 = tid; t0 < K * maxcol; t0 += 4 * CM_FN_THREADS) {
    int nn[4], gg[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = t0 + u * CM_FN_THREADS;
      nn[u] = -1; gg[u] = 0;
      if (t >= K * maxcol) continue;
      const int k = t / maxcol, j = t % maxcol;
      if ((CACHED ? zl[k] : z[k]) == 0.0) continue;
      if (j == 0) atomicAdd(&nz, 1);
      const int c0 = CACHED ? cptr[k] : col_ptr[k], c1 = CACHED ? cptr[k + 1] : col_ptr[k + 1];
      if (c0 + j < c1) { nn[u] = csc_row[c0 + j]; gg[u] = csc_grp[c0 + j]; }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int n = nn[u];
      if (n < 0 || cnt[n] < 0) continue;
      atomicAdd(&cnt[n], 1);
      if (gg[u] < 1) continue;                                                // value code 0 = the smallest value, which is no power (:104, quirk 8)
      unsigned inc;
      unsigned* wd = hit_word(n, gg[u], inc);
      atomicAdd(wd, inc);
    }
  }
  __syncthreads();
  const bool filter = min_count > 0.0;
  if (CACHED) {                                                               // first verdicts: every candidate
    for (int n = tid; n < N; n += CM_FN_THREADS) {
      const int c = cnt[n];
      acc[n] = (c >= 0 && (!filter || (double)c >= min_count) && would_reconnect(n)) ? 1 : 0;
    }
    __syncthreads();
  }
  for (;;) {
    if (!((double)nz > min_count)) break;                                     // :111,141 (uniform: nz is only written between barriers)
    if (tid == 0) { nlist = 0; best_sh = 0ull; }
    __syncthreads();
    // the first cell in the reference's order that reconnects (cells that fell below minimum_spike_count can never reconnect --
    // counts only fall: dropped without a look)
    for (int n = tid; n < N; n += CM_FN_THREADS) {
      const int c = cnt[n];
      if (c < 0) continue;
      if (filter && !((double)c >= min_count)) { cnt[n] = -1; continue; }
      if (CACHED ? (acc[n] & 1) : would_reconnect(n)) atomicMax(&best_sh, key_of(c, n));
    }
    __syncthreads();
    const unsigned long long best = best_sh;
    // the cells the reference examines before it reaches that one (all of them, if none reconnects) leave the list rejected (:139)
    for (int n = tid; n < N; n += CM_FN_THREADS) {
      const int c = cnt[n];
      if (c >= 0 && key_of(c, n) >= best) cnt[n] = -1;
    }
    if (best == 0ull) break;                                                  // nobody reconnects: done (uniform)
    const int focus = (int)(0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFull));
    // the cell's row: trial and z of the trial, staged by all threads; the sums are taken by one thread in row order
    const int e0 = CACHED ? rptr[focus] : row_ptr[focus], len = (CACHED ? rptr[focus + 1] : row_ptr[focus + 1]) - e0;
    for (int i = tid; i < len; i += CM_FN_THREADS) { const int k = col[e0 + i]; rk[i] = k; rz[i] = CACHED ? zl[k] : z[k]; }
    __syncthreads();
    if (tid == 0) {
      // np.mean and scipy.stats.sem of the event amplitudes, summed as NumPy sums them (cm_np_sum), in row order
      int m = 0;
      for (int q = 0; q < len; ++q) if (rz[q] != 0.0) amp[m++] = rz[q];
      const double mean = cm_np_sum(amp, m) / (double)m;
      for (int q = 0; q < m; ++q) { const double d = amp[q] - mean; amp[q] = d * d; }
      const double ssd = cm_np_sum(amp, m);
      p.mu[focus] = mean;                                                      // :134
      p.beta[focus] = m > 1 ? sqrt(ssd / (double)(m - 1)) / sqrt((double)m) : NAN;   // :135 scipy.stats.sem, ddof = 1
      nz -= m;
    }
    for (int t = tid; t < len * maxcol; t += CM_FN_THREADS) {                   // :136-137, flattened over (event, position in its trial's column)
      const int i = t / maxcol, j = t % maxcol;
      if (rz[i] == 0.0) continue;
      const int k = rk[i];
      if (j == 0) {
        p.lam[e0 + i] = 1.0;
        z[k] = 0.0;
      }
      // the event leaves the counts of the other candidates targeted in that trial (their counts include it: never below zero)
      const int c0 = CACHED ? cptr[k] : col_ptr[k], c1 = CACHED ? cptr[k + 1] : col_ptr[k + 1];
      if (c0 + j >= c1) continue;
      const int mrow = csc_row[c0 + j], g = csc_grp[c0 + j];
      if (cnt[mrow] <= 0) continue;
      atomicSub(&cnt[mrow], 1);
      if (g >= 1) { unsigned inc; unsigned* wd = hit_word(mrow, g, inc); atomicSub(wd, inc); }
      if (CACHED) {                                                           // its verdict may have changed: once on the dirty list
        unsigned* aw = reinterpret_cast<unsigned*>(acc) + (mrow >> 2);
        const unsigned bit = 2u << (8 * (mrow & 3));
        if (!(atomicOr(aw, bit) & bit)) list[atomicAdd(&nlist, 1)] = mrow;
      }
    }
    __syncthreads();
    if (CACHED) {
      for (int i = tid; i < len; i += CM_FN_THREADS) if (rz[i] != 0.0) zl[rk[i]] = 0.0;   // (every thread has read this round's events)
      const int nd = nlist;
      for (int i = tid; i < nd; i += CM_FN_THREADS) {
        const int n = list[i], c = cnt[n];
        acc[n] = (c >= 0 && (!filter || (double)c >= min_count) && would_reconnect(n)) ? 1 : 0;      // (clears the dirty bit)
      }
      __syncthreads();
    }
  }
}

// the circuits of the set fit the kernel (LDS budget, number of powers); otherwise the caller takes the host path
static int fn_scan_shape(const cm_set& s, int* vmax_out, int* kmax_out) {
  int vmax = 0, kmax = 0;
  for (const cm_problem* q : s.ps) { vmax = std::max(vmax, q->V); kmax = std::max(kmax, q->K); }
  *vmax_out = vmax; *kmax_out = kmax;
  return std::max(1, vmax - 1);
}
// Where reconnect_spont_cells runs: 2 = the LDS-resident kernel, 1 = the same kernel on the global arrays, 0 = on the host.
// On the global arrays every round is a chain of dependent loads: for ONE large circuit (N = 5000 / K = 10000: 7.7 ms measured) the host scan
// (0.6 ms incl. its copies) is the faster path.  The kernel takes all circuits of a batch side by side, though, where the host scans them one
// after the other behind a stream synchronisation (35 ms for 64 such circuits: 7 % of the step): from eight circuits on the batch stays on the
// device.  CAVIAR_HIP_FN_SCAN_GLOBAL=1 (test hook) takes the global-array form wherever its own state fits.
static int cm_fn_scan_mode(const cm_set& s) {
  static const bool force_global = getenv("CAVIAR_HIP_FN_SCAN_GLOBAL") != nullptr;
  int vmax, kmax;
  const int P = fn_scan_shape(s, &vmax, &kmax);
  if (vmax - 1 > CM_FN_MAXP || s.max_row_len_max >= 65536) return 0;
  const int row_cap = std::max(1, s.max_row_len_max);
  if (!force_global && fn_scan_lds(s.lead->N, kmax, P, row_cap, true) <= 150 * 1024) return 2;
  if ((force_global || s.B >= 8) && fn_scan_lds(s.lead->N, kmax, P, row_cap, false) <= 150 * 1024) return 1;
  return 0;
}
bool cm_fn_scan_fits(const cm_set& s) { return cm_fn_scan_mode(s) != 0; }
int cm_launch_fn_scan(cm_ctx* ctx, cm_set& s, double minimax_spk_prob, double minimum_spike_count) {
  const int row_cap = std::max(1, s.max_row_len_max);
  int vmax, kmax;
  const int P = fn_scan_shape(s, &vmax, &kmax);
  const int mode = cm_fn_scan_mode(s);
  if (mode == 0) { cm_set_error("cm_launch_fn_scan: the circuit does not fit the scan kernel (N = %d, %d powers)", s.lead->N, vmax - 1); return CM_ERR_INVALID; }
  // the LDS-resident form needs 16-bit trial counts per (cell, power) and the whole state in 150 KB
  const bool cached = mode == 2;
  const size_t lds = fn_scan_lds(s.lead->N, kmax, P, row_cap, cached);
  if (cached) {
    if (s.B == 1) CM_TRY(cm_raise_lds(ctx, CM_LDS_FN_SCAN, reinterpret_cast<const void*>(k_fn_scan<false, true>), lds, 48 * 1024));
    else CM_TRY(cm_raise_lds(ctx, CM_LDS_FN_SCAN_B, reinterpret_cast<const void*>(k_fn_scan<true, true>), lds, 48 * 1024));
    CM_LAUNCH_SET_T(k_fn_scan, (true), s, 1, CM_FN_THREADS, lds, ctx->stream, minimax_spk_prob, minimum_spike_count, row_cap, std::max(1, s.max_col_len_max));
  } else {
    if (s.B == 1) CM_TRY(cm_raise_lds(ctx, CM_LDS_FN_SCAN_G, reinterpret_cast<const void*>(k_fn_scan<false, false>), lds, 48 * 1024));
    else CM_TRY(cm_raise_lds(ctx, CM_LDS_FN_SCAN_GB, reinterpret_cast<const void*>(k_fn_scan<true, false>), lds, 48 * 1024));
    CM_LAUNCH_SET_T(k_fn_scan, (false), s, 1, CM_FN_THREADS, lds, ctx->stream, minimax_spk_prob, minimum_spike_count, row_cap, std::max(1, s.max_col_len_max));
  }
  CM_HIP(hipGetLastError());
  return CM_OK;
}
