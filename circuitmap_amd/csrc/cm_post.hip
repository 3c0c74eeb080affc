// Post-sweep phases of one CAVIaR iteration:
//   update_sigma                      reference circuitmap/optimise/caviar.py:238-244
//   update_phi / _laplace_approx      caviar.py:246-316
//   estimate_spont_act_soft_thresh    caviar.py:147-163  (+ spont_rate, caviar.py:88)
#include <algorithm>

#include "cm_bodies.h"

// ------------------------------------------------------------------------------------------------
// Statistics of the new lam, one launch, two roles (256-thread blocks):
//   blocks [0, nbR): one wave per row  -> rs[n] = {sum lam, sum lam^2, sum y*lam, sum lam(1-lam) or -1 if the row is all zero}
//   blocks [nbR, ..): 16 lanes per trial -> R_k = sum_n mu_n lam_nk (= lam.T @ mu),
//                     colany_k = any_n lam_nk >= spont_orthogonality
// ------------------------------------------------------------------------------------------------
// statistics of one row by one wavefront: rs[n] = {sum lam, sum lam^2, sum y lam, sum lam (1 - lam) or -1 for an all-zero row}
// mark_it >= 0: a row with lam != 0 also puts its trials on the list of iteration mark_it (cm_pdev::tstamp: first writer of a stamp appends)
template <bool BATCH>
__device__ __forceinline__ void stats_row(const cm_pdev& p, const double* __restrict__ lam, int n, int lane, int mark_it = -1) {
    const int* __restrict__ row_ptr = p.row_ptr;
  const double* __restrict__ ye = p.ye;               // y[col[e]], entry by entry: contiguous like lam
  double s1 = 0.0, s2 = 0.0, sy = 0.0, d = 0.0;
  int any = 0;
  for (int e = row_ptr[n] + lane; e < row_ptr[n + 1]; e += CM_WAVE) {
    const double l = lam[e];
    s1 += l;
    s2 = fma(l, l, s2);
    sy = fma(ye[e], l, sy);
    d = fma(l, 1.0 - l, d);
    any |= (l != 0.0);
  }
  s1 = cm_wave_sum(s1); s2 = cm_wave_sum(s2); sy = cm_wave_sum(sy); d = cm_wave_sum(d);
  any = __any(any);
  if (lane == 0) {
    double* __restrict__ rs = p.rs;
    rs[(size_t)n * 4 + 0] = s1;
    rs[(size_t)n * 4 + 1] = s2;
    rs[(size_t)n * 4 + 2] = sy;
    rs[(size_t)n * 4 + 3] = any ? d : -1.0;   // sum lam(1-lam) for active rows, -1 marks an all-zero row
    // batches: list of the rows with lam != 0 (order irrelevant) for the per-neuron role of k_iter_mid; one atomic per such row
    // (a few dozen per circuit in the steady state), counter zeroed by k_init_state / k_presweep
    if (BATCH && any) { const int pos = atomicAdd(&p.alive[0], 1); if (pos < p.N) p.alive[1 + pos] = n; }
  }
  if (mark_it >= 0 && any) {
    const int q = mark_it & 1, K = p.K, stamp = 2 * mark_it;
    for (int e = row_ptr[n] + lane; e < row_ptr[n + 1]; e += CM_WAVE) {
      const int k = p.col[e];
      if (atomicExch(&p.tstamp[q * K + k], stamp) != stamp) p.tlist_s[q * K + atomicAdd(&p.tcnt[q], 1)] = k;
    }
  }
}

// `quiescent` (set with the pre-sweep's, cm_launch_sweep): rows are screened one per THREAD -- a row that was all zero before the sweep, was
// not on the chain (mu == 0) and was not revived by the pre-sweep is still all zero and keeps its statistics {0, 0, 0, -1}; only the
// others (a few dozen per circuit in the steady state) are re-read, one wavefront each.
// rows screened per workgroup in the quiescent regime: few enough that each of the four wavefronts re-reads (and lists the trials of) about one row --
// with 256 rows per workgroup a wavefront walked three or four rows one after the other, five dependent round trips each
#ifndef CM_STATS_QROWS
#define CM_STATS_QROWS 64
#endif
template <bool BATCH>
__global__ __launch_bounds__(256) void k_stats(cm_parg<BATCH> pa, int nbR, double orth, int quiescent, int nbL, int list_it, int mark_it) {
  const cm_pdev& p = pa.get();
  const double* __restrict__ lam = p.lam;
  if (quiescent && (int)blockIdx.x < nbR) {
    __shared__ int qlist[256];
    __shared__ int qn;
    if (threadIdx.x == 0) qn = 0;
    __syncthreads();
    const int n = (int)blockIdx.x * CM_STATS_QROWS + (int)threadIdx.x;
    if ((int)threadIdx.x < CM_STATS_QROWS && n < p.N) {
      const bool revived = p.touch[n] != 0;
      if (revived) p.touch[n] = 0;
      if (revived || p.rs[(size_t)n * 4 + 3] >= 0.0 || p.seq_pos[n] >= 0) qlist[atomicAdd(&qn, 1)] = n;
    }
    __syncthreads();
    const int cnt = qn;
    for (int i = threadIdx.x / CM_WAVE; i < cnt; i += 4) stats_row<BATCH>(p, lam, qlist[i], threadIdx.x % CM_WAVE, mark_it);
    return;
  }
  if (BATCH && (int)blockIdx.x < nbR) {
    // batches: FOUR rows per wavefront -- two pairs (one row per 32-lane half), the load levels of both pairs in flight together: the
    // launch is latency-bound at full occupancy, so its duration is its number of wavefronts times the length of their dependent
    // load chain; cm_half_sum gives each row bitwise the sums cm_wave_sum gives it alone in a wavefront
    const int lane = threadIdx.x % CM_WAVE, sub = lane & 31, N = p.N;
    const int npairs = (N + 1) >> 1, W = (npairs + 1) >> 1;          // W wavefronts: pair w and pair w + W
    const int w = (int)((blockIdx.x * blockDim.x + threadIdx.x) / CM_WAVE);
    if (w >= W) return;
    const int* __restrict__ row_ptr = p.row_ptr;
    const int nA = 2 * w + (lane >> 5), nB = 2 * (w + W) + (lane >> 5);
    const bool onA = nA < N, onB = nB < N;
    const int begA = onA ? row_ptr[nA] : 0, lenA = onA ? row_ptr[nA + 1] - begA : 0;
    const int begB = onB ? row_ptr[nB] : 0, lenB = onB ? row_ptr[nB + 1] - begB : 0;
    if (!__any(lenA > 32 || lenB > 32)) {
      const double* __restrict__ ye = p.ye;           // y[col[e]], entry by entry: contiguous like lam -- no gather level
      const bool inA = sub < lenA, inB = sub < lenB;
      const double lA = inA ? lam[begA + sub] : 0.0, lB = inB ? lam[begB + sub] : 0.0;
      const double yA = inA ? ye[begA + sub] : 0.0, yB = inB ? ye[begB + sub] : 0.0;
      double* __restrict__ rs = p.rs;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const bool in = h ? inB : inA, row_on = h ? onB : onA;
        const double l = h ? lB : lA, yy = h ? yB : yA;
        const int n = h ? nB : nA;
        double s1 = 0.0, s2 = 0.0, sy = 0.0, d = 0.0;
        bool nz = false;
        if (in) {
          s1 += l;
          s2 = fma(l, l, s2);
          sy = fma(yy, l, sy);
          d = fma(l, 1.0 - l, d);
          nz = l != 0.0;
        }
        s1 = cm_half_sum(s1); s2 = cm_half_sum(s2); sy = cm_half_sum(sy); d = cm_half_sum(d);
        const bool any = ((__ballot(nz) >> (lane & 32)) & 0xFFFFFFFFull) != 0ull;
        if (sub == 0 && row_on) {
          rs[(size_t)n * 4 + 0] = s1;
          rs[(size_t)n * 4 + 1] = s2;
          rs[(size_t)n * 4 + 2] = sy;
          rs[(size_t)n * 4 + 3] = any ? d : -1.0;
          if (any) { const int pos = atomicAdd(&p.alive[0], 1); if (pos < N) p.alive[1 + pos] = n; }
        }
      }
      return;
    }
    // a long row among the four: one row after the other, a full wavefront each
    for (int r = 0; r < 2; ++r) {
      if (2 * w + r < N) stats_row<BATCH>(p, lam, 2 * w + r, lane);
      if (2 * (w + W) + r < N) stats_row<BATCH>(p, lam, 2 * (w + W) + r, lane);
    }
  } else if ((int)blockIdx.x < nbR) {
    const int n = (blockIdx.x * blockDim.x + threadIdx.x) / CM_WAVE;
    const int lane = threadIdx.x % CM_WAVE;
    if (n >= p.N) return;
    stats_row<BATCH>(p, lam, n, lane);
  } else if (list_it >= 0) {
    // listed trials (the pre-sweep of this iteration ran on the list too and added the trials of the rows it revived): blocks [nbR, nbR + nbL)
    // walk the two lists, the blocks behind them zero every trial that is on neither
    const int q = list_it & 1, K = p.K, b = (int)blockIdx.x - nbR;
    if (b < nbL)
      cm_col_dot16_listed(b * (int)(blockDim.x >> 4) + (int)(threadIdx.x >> 4), nbL * (int)(blockDim.x >> 4), threadIdx.x & 15, p.tlist_s + q * K, p.tcnt[q],
                          p.tlist_r + q * K, p.tcnt[2 + q], K, p.mu, lam, p.col_ptr, p.csc_row, p.csc_pos, orth, p.R, p.colany);
    else
      cm_col_zero_unlisted((b - nbL) * (int)blockDim.x + (int)threadIdx.x, K, p.tstamp + q * K, 2 * list_it, 2 * list_it + 1, p.R, p.colany);
  } else {
    const int t = ((int)blockIdx.x - nbR) * blockDim.x + threadIdx.x;
    if (BATCH) cm_col_dot16x2(t, p.K, p.mu, lam, p.col_ptr, p.csc_row, p.csc_pos, orth, p.R, p.colany);
    else cm_col_dot16(t, p.K, p.mu, lam, p.col_ptr, p.csc_row, p.csc_pos, orth, p.R, p.colany);
  }
}

// stand-alone single-workgroup phases (per-phase test hooks): 256 threads like block 0 of the fused kernel -> the same sums, bit for bit
__global__ __launch_bounds__(CM_POST_THREADS) void k_post_single(cm_parg<false> pa, int it, int max_iters, double penalty, double scale_factor,
                                                                  double tol, int do_sigma, int seq) {
  const cm_pdev& p = pa.get();
  post_single_body(p.y, p.R, p.colany, p.mask, p.mu, p.beta, p.rs, p.sum_y, p.N, p.K, it, max_iters, penalty, scale_factor, tol, do_sigma, seq,
                   p.act_idx, p.act_inv, p.n_act, p.mailbox, p.z, p.scal);
}

// Row + column statistics of the current lam.
int cm_launch_stats(cm_ctx* ctx, cm_set& s, double spont_orthogonality, int quiescent, int it) {
  const cm_problem* p = s.lead;
  const int nbR = quiescent ? (p->N + CM_STATS_QROWS - 1) / CM_STATS_QROWS : (s.B > 1 ? (p->N + 15) / 16 : (p->N + 3) / 4);      // one wave per row (batches: per four rows); screened: CM_STATS_QROWS rows per block
  int nbC = ((s.B > 1 ? (p->K + 1) / 2 : p->K) * 16 + 255) / 256;   // 16 lanes per trial column (batches: per two columns)
  // Listed trials (cm_pdev::tstamp): in the quiescent regime the row role walks exactly the rows with lam != 0 and leaves the list of their trials
  // for the next iteration (mark_it); the column role uses this iteration's list if the pre-sweep ran on it (and completed it with the trials of
  // the rows it revived).  Grid for the listed walk: the trials the rows of the largest active set can touch, with slack for revived rows --
  // a stride loop takes whatever exceeds it.
  static const bool no_lists = getenv("CAVIAR_HIP_NO_TRIAL_LISTS") != nullptr;   // diagnostic: every column, every iteration
  long long na_sum = 0;
  for (int v : s.na) na_sum += v;
  const long long rows_mean = s.na.empty() ? p->N : (na_sum + (long long)s.na.size() - 1) / (long long)s.na.size();
  const bool few = ((s.nnz_max + p->N - 1) / p->N) * rows_mean * 2 <= p->K || getenv("CAVIAR_HIP_TEST_TRIAL_LISTS") != nullptr;     // the rows with lam != 0 touch at most half of the trials
  const int mark_it = (quiescent && it >= 0 && !no_lists && s.lists_ok && few) ? it + 1 : -1;
  const int list_it = (quiescent && it >= 0 && s.tlist_it == it && s.tlist_sweep_it == it) ? it : -1;
  int nbL = 0;
  if (list_it >= 0) {
    const long long per_row = (s.nnz_max + p->N - 1) / p->N;
    const long long est = std::min<long long>(p->K, per_row * (long long)s.na_max * 3 / 2 + 64);
    nbL = (int)((((est + 1) / 2) * 16 + 255) / 256);                 // two listed trials per 16-lane segment
    nbC = nbL + (p->K + 255) / 256;
  }
  CM_LAUNCH_SET(k_stats, s, nbR + nbC, 256, 0, ctx->stream, nbR, spont_orthogonality, quiescent, nbL, list_it, mark_it);
  CM_HIP(hipGetLastError());
  s.tlist_it = mark_it;
  return CM_OK;
}

// Compaction (+ mailbox post) and, if do_sigma, update_sigma / z / spont_rate.  `it` < 0 means "state init".  Single circuit only.
int cm_launch_post_single(cm_ctx* ctx, cm_set& s, int it, const cm_options* o, int do_sigma) {
  hipStream_t st = ctx->stream;
  const int seq = ++ctx->mail_seq;
  hipLaunchKernelGGL(k_post_single, dim3(1), dim3(CM_POST_THREADS), 0, st, cm_parg<false>{s.h0}, it, o ? o->max_backtrack_iters : 0,
                     o ? o->penalty : 0.0, o ? o->scale_factor : 0.0, o ? o->tol : 0.0, do_sigma, seq);
  CM_HIP(hipGetLastError());
  if (!ctx->use_mailbox) {
    CM_HIP(hipMemcpyAsync(ctx->h_pinned + 8, s.lead->n_act, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
    CM_HIP(hipEventRecord(ctx->ev[7], st));
  }
  return CM_OK;
}

// Host side of the mailboxes: wait until the kernel launched by the last post has published the active-set size of every circuit
// of the set; the decoded values land in s.na / s.na_max (and ctx->mb_n / ctx->mb_flag: largest size, OR of the not-SPD flags).
int cm_wait_active(cm_ctx* ctx, cm_set& s) {
  s.na.resize(s.B);
  if (!ctx->use_mailbox) {                             // single circuit only (cm_set_single refuses batches without mailbox)
    CM_HIP(hipEventSynchronize(ctx->ev[7]));
    ctx->mb_n = ctx->h_pinned[8]; ctx->mb_flag = ctx->h_pinned[9];
    s.na[0] = s.na_max = ctx->mb_n;
    return CM_OK;
  }
  const unsigned seq = (unsigned)ctx->mail_seq;
  int nmax = 0, flag = 0;
  for (int c = 0; c < s.B; ++c) {
    volatile unsigned long long* mb = (volatile unsigned long long*)ctx->h_pinned + (s.B == 1 ? 0 : CM_MAILBOX_BATCH0 + c);
    unsigned long long spins = 0, w;
    while ((unsigned)((w = __atomic_load_n(mb, __ATOMIC_ACQUIRE)) >> 32) != seq) {
      if ((++spins & 0xFFFFull) == 0) {                      // every 64k polls make sure the stream is still alive
        hipError_t q = hipStreamQuery(ctx->stream);
        if (q != hipErrorNotReady && q != hipSuccess) { cm_set_error("stream failed while waiting for the active set: %s", hipGetErrorString(q)); return CM_ERR_HIP; }
        if (q == hipSuccess && (unsigned)(__atomic_load_n(mb, __ATOMIC_ACQUIRE) >> 32) != seq) {
          cm_set_error("active-set mailbox of circuit %d was not posted (seq %u)", c, seq);
          return CM_ERR_STATE;
        }
      }
    }
    s.na[c] = (int)(w & 0x7FFFFFFFull);
    nmax = std::max(nmax, s.na[c]);
    flag |= (int)((w >> 31) & 1ull);
  }
  s.na_max = nmax;
  ctx->mb_n = nmax;
  ctx->mb_flag = flag;
  return CM_OK;
}

// update_phi stand-alone (final update after the false-negative scan, and the per-phase test hook): one wave per neuron
// use_dead: rows found to be identically zero take the per-run constants (valid once k_phi_dead has run for the current priors)
// screen (batches, final update of a run: rs still holds the statistics of the last sweep): 64 rows per workgroup, one per THREAD first -- a row
// that was all zero after the last sweep and has mu == 0 was not reconnected by the false-negative scan (a reconnected cell gets mu = mean of
// its events > 0) and takes its constants without a wavefront; only the others (a few dozen per circuit) run the Newton iteration.  One
// wavefront per row for all N rows, 95 % of which only found out that they were all zero, was 0.44 ms per step.
#define CM_PHI_QROWS 64
template <bool BATCH, int SEG>
__global__ __launch_bounds__(256) void k_update_phi(cm_parg<BATCH> pa, int use_dead, int screen) {
  __shared__ double etab[64], ltab[128];
  __shared__ int qlist[CM_PHI_QROWS], qn;
  cm_exptab_init(etab);
  cm_logtab_init(ltab);
  if (threadIdx.x == 0) qn = 0;
  __syncthreads();                                   // (before any wavefront leaves)
  const cm_pdev& p = pa.get();
  double out[4];
  if (BATCH && screen) {
    const int n = (int)blockIdx.x * CM_PHI_QROWS + (int)threadIdx.x;
    if ((int)threadIdx.x < CM_PHI_QROWS && n < p.N) {
      if (p.rs[(size_t)n * 4 + 3] < 0.0 && p.mu[n] == 0.0) {
        p.phi[2 * n] = p.phi_dead[2 * n]; p.phi[2 * n + 1] = p.phi_dead[2 * n + 1];
#pragma unroll
        for (int i = 0; i < 4; ++i) p.phi_cov[4 * n + i] = p.phi_cov_dead[4 * n + i];
      } else qlist[atomicAdd(&qn, 1)] = n;
    }
    __syncthreads();
    const int cnt = qn;
    for (int i0 = 2 * (threadIdx.x / CM_WAVE); i0 < cnt; i0 += 8) {          // two rows per wavefront, one per half (phi_pair)
      const int i = i0 + (int)((threadIdx.x % CM_WAVE) >> 5);
      phi_pair<SEG>(i < cnt ? qlist[i] : -1, threadIdx.x % CM_WAVE, p.lam, p.row_ptr, p.grp, p.rowcnt, p.vals, p.V, p.phi_prior, p.phi_cov_prior, p.phi, p.phi_cov, out,
                    etab, ltab, p.phi_dead, p.phi_cov_dead);
    }
    return;
  }
  const int n = 2 * (int)((blockIdx.x * blockDim.x + threadIdx.x) / CM_WAVE) + (int)((threadIdx.x % CM_WAVE) >> 5);
  if (!__any(n < p.N)) return;
  phi_pair<SEG>(n < p.N ? n : -1, threadIdx.x % CM_WAVE, p.lam, p.row_ptr, p.grp, p.rowcnt, p.vals, p.V, p.phi_prior, p.phi_cov_prior, p.phi, p.phi_cov, out,
                etab, ltab, use_dead ? p.phi_dead : nullptr, use_dead ? p.phi_cov_dead : nullptr);
}

// the per-run constants: update_phi of neuron n for an all-zero lam row -> phi_dead / phi_cov_dead.
// That result is a function of the neuron's trial counts per stimulus value and of its phi priors alone, and neurons share those: at the benchmark
// shape a circuit's 1000 neurons have ~200 distinct (counts, priors) tuples.  So the Newton iteration (as expensive as that of a live row: the
// launch took ~1 ms of every 256-circuit step on the side stream) runs for one REPRESENTATIVE per tuple -- the first neuron that carries it, found
// by an exact comparison behind a hash -- and the other neurons copy its six numbers: the same function of the same arguments, bit for bit.
#define CM_REP_MAXN 8192           // hashes of all neurons in LDS; larger circuits: every neuron its own representative
__device__ __forceinline__ unsigned rep_hash(const cm_pdev& p, int n) {
  unsigned long long h = 0x9E3779B97F4A7C15ull;
  for (int g = 0; g < p.V; ++g) h = (h ^ (unsigned long long)(unsigned)p.rowcnt[(size_t)n * CM_MAX_VALS + g]) * 0x100000001B3ull;
  for (int i = 0; i < 2; ++i) h = (h ^ (unsigned long long)__double_as_longlong(p.phi_prior[2 * (size_t)n + i])) * 0x100000001B3ull;
  for (int i = 0; i < 4; ++i) h = (h ^ (unsigned long long)__double_as_longlong(p.phi_cov_prior[4 * (size_t)n + i])) * 0x100000001B3ull;
  return (unsigned)(h ^ (h >> 32));
}
__device__ __forceinline__ bool rep_same(const cm_pdev& p, int a, int b) {
  for (int g = 0; g < p.V; ++g) if (p.rowcnt[(size_t)a * CM_MAX_VALS + g] != p.rowcnt[(size_t)b * CM_MAX_VALS + g]) return false;
  for (int i = 0; i < 2; ++i) if (__double_as_longlong(p.phi_prior[2 * (size_t)a + i]) != __double_as_longlong(p.phi_prior[2 * (size_t)b + i])) return false;
  for (int i = 0; i < 4; ++i) if (__double_as_longlong(p.phi_cov_prior[4 * (size_t)a + i]) != __double_as_longlong(p.phi_cov_prior[4 * (size_t)b + i])) return false;
  return true;
}
template <bool BATCH>
__global__ __launch_bounds__(256) void k_phi_rep(cm_parg<BATCH> pa) {
  const cm_pdev& p = pa.get();
  const int N = p.N;
  const int n = (int)blockIdx.x * (int)blockDim.x + (int)threadIdx.x;
  if (N > CM_REP_MAXN) { if (n < N) p.phi_rep[n] = n; return; }
  extern __shared__ unsigned rep_h[];              // [N]
  const int hi = min(N, ((int)blockIdx.x + 1) * (int)blockDim.x);     // this workgroup's neurons look at 0 .. hi - 1
  for (int i = threadIdx.x; i < hi; i += blockDim.x) rep_h[i] = rep_hash(p, i);
  __syncthreads();
  if (n >= N) return;
  const unsigned mine = rep_h[n];
  int rep = n;
  for (int m = 0; m < n; ++m)
    if (rep_h[m] == mine && rep_same(p, m, n)) { rep = m; break; }
  p.phi_rep[n] = rep;
}
template <bool BATCH, int SEG>
__global__ __launch_bounds__(256) void k_phi_dead(cm_parg<BATCH> pa) {
  __shared__ double etab[64], ltab[128];
  cm_exptab_init(etab);
  cm_logtab_init(ltab);
  __syncthreads();                                   // (before any wavefront leaves)
  const cm_pdev& p = pa.get();
  int n = 2 * (int)((blockIdx.x * blockDim.x + threadIdx.x) / CM_WAVE) + (int)((threadIdx.x % CM_WAVE) >> 5);   // two neurons per wavefront, one per half (phi_pair)
  if (n >= p.N || p.phi_rep[n < p.N ? n : 0] != n) n = -1;     // (not a representative: k_phi_copy fills it in from its representative)
  if (!__any(n >= 0)) return;
  double out[4];
  phi_pair<SEG, true>(n, threadIdx.x % CM_WAVE, p.lam, p.row_ptr, p.grp, p.rowcnt, p.vals, p.V, p.phi_prior, p.phi_cov_prior, p.phi_dead,
                      p.phi_cov_dead, out, etab, ltab);
  // the lower truncation points of the Monte-Carlo draws from this phi (caviar.py:212): constants of the run as well
  if (n >= 0 && (threadIdx.x & 31) == 0) {
    p.phi_c_dead[2 * n] = normcdf(-out[0] / out[2]);
    p.phi_c_dead[2 * n + 1] = normcdf(-out[1] / out[3]);
  }
}
template <bool BATCH>
__global__ __launch_bounds__(256) void k_phi_copy(cm_parg<BATCH> pa) {
  const cm_pdev& p = pa.get();
  const int n = (int)blockIdx.x * (int)blockDim.x + (int)threadIdx.x;
  if (n >= p.N) return;
  const int r = p.phi_rep[n];
  if (r == n) return;
  p.phi_dead[2 * n] = p.phi_dead[2 * r]; p.phi_dead[2 * n + 1] = p.phi_dead[2 * r + 1];
#pragma unroll
  for (int i = 0; i < 4; ++i) p.phi_cov_dead[4 * n + i] = p.phi_cov_dead[4 * r + i];
  p.phi_c_dead[2 * n] = p.phi_c_dead[2 * r]; p.phi_c_dead[2 * n + 1] = p.phi_c_dead[2 * r + 1];
}
template <int SEG>
static void launch_phi_dead(cm_set& s, hipStream_t st) {
  CM_LAUNCH_SET_T(k_phi_dead, (SEG), s, (unsigned)((s.lead->N + 7) / 8), 256, 0, st);
}
int cm_launch_phi_dead(cm_ctx* ctx, cm_set& s, hipStream_t st) {
  const int V = s.lead->V, N = s.lead->N;
  {
    const size_t lds = N <= CM_REP_MAXN ? (size_t)N * sizeof(unsigned) : 0;
    if (s.B == 1) CM_TRY(cm_raise_lds(ctx, CM_LDS_PHI_REP, reinterpret_cast<const void*>(k_phi_rep<false>), lds, 48 * 1024));
    else CM_TRY(cm_raise_lds(ctx, CM_LDS_PHI_REP_B, reinterpret_cast<const void*>(k_phi_rep<true>), lds, 48 * 1024));
    CM_LAUNCH_SET(k_phi_rep, s, (unsigned)((N + 255) / 256), 256, lds, st);
  }
  if (V <= 1) launch_phi_dead<1>(s, st);
  else if (V <= 2) launch_phi_dead<2>(s, st);
  else if (V <= 4) launch_phi_dead<4>(s, st);
  else if (V <= 8) launch_phi_dead<8>(s, st);
  else if (V <= 16) launch_phi_dead<16>(s, st);
  else if (V <= 32) launch_phi_dead<32>(s, st);
  else launch_phi_dead<64>(s, st);
  CM_LAUNCH_SET(k_phi_copy, s, (unsigned)((N + 255) / 256), 256, 0, st);
  CM_HIP(hipGetLastError());
  return CM_OK;
}

template <int SEG>
static void launch_phi(cm_set& s, hipStream_t st, int use_dead, int screen) {
  const int N = s.lead->N;
  CM_LAUNCH_SET_T(k_update_phi, (SEG), s, (unsigned)(screen ? (N + CM_PHI_QROWS - 1) / CM_PHI_QROWS : (N + 7) / 8), 256, 0, st, use_dead, screen);
}

int cm_launch_update_phi(cm_ctx* ctx, cm_set& s, hipStream_t st, int use_dead, int screen) {
  (void)ctx;
  screen = (screen && use_dead && s.B > 1) ? 1 : 0;
  const int V = s.lead->V;                                  // the group shares the lane-segment class of V
  if (V <= 1) launch_phi<1>(s, st, use_dead, screen);
  else if (V <= 2) launch_phi<2>(s, st, use_dead, screen);
  else if (V <= 4) launch_phi<4>(s, st, use_dead, screen);
  else if (V <= 8) launch_phi<8>(s, st, use_dead, screen);
  else if (V <= 16) launch_phi<16>(s, st, use_dead, screen);
  else if (V <= 32) launch_phi<32>(s, st, use_dead, screen);
  else launch_phi<64>(s, st, use_dead, screen);
  CM_HIP(hipGetLastError());
  return CM_OK;
}
