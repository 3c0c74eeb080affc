// Device bodies shared by the stand-alone kernels and the fused mid-iteration kernel (cm_iter.hip).
#pragma once
#include "cm_internal.h"


// R_k = sum_n mu_n lam_nk (and colany_k = any lam_nk >= orth) with 16 lanes per trial column: lane j takes the column's j-th
// entry (three dependent load levels in total: col_ptr -> indices -> operands), the 16 products are summed by a DPP segment
// reduction (fixed tree -> deterministic); columns with more than 16 entries add their 16-entry partial sums in order.
// `t` is the global thread index; every lane of the wave must call this (DPP), out-of-range columns just do not store.
__device__ __forceinline__ void cm_col_dot16(int t, int K, const double* __restrict__ mu, const double* __restrict__ lam,
                                             const int* __restrict__ col_ptr, const int* __restrict__ csc_row,
                                             const int* __restrict__ csc_pos, double orth, double* __restrict__ R,
                                             uint8_t* __restrict__ colany) {
  const int k = t >> 4, j = t & 15;
  const bool ok = k < K;
  const int beg = ok ? col_ptr[k] : 0, end = ok ? col_ptr[k + 1] : 0;
  int len = end - beg;
  // the loop trip count must be uniform over the 16 lanes of a segment (it is: same column) but may differ between segments
  // of a wave; DPP sources always lie inside the segment, so partial execution masks are safe
  double s = 0.0;
  int any = 0;
  for (int c0 = 0; c0 < len; c0 += 16) {
    const int ce = beg + c0 + j;
    double v = 0.0;
    if (ce < end) {
      // without the "any" output only rows with mu != 0 contribute (mu * lam = +0 exactly otherwise: lam is finite and >= 0), and those are
      // a few percent of the entries in the steady state: lam is gathered for them alone
      const double m = mu[csc_row[ce]];
      if (colany || m != 0.0) {
        const double l = lam[csc_pos[ce]];
        v = m * l;
        any |= (l >= orth);
      }
    }
    s += cm_seg_sum<16>(v);
  }
  // OR over the segment
  any |= __builtin_amdgcn_update_dpp(0, any, CM_DPP_XOR1, 0xf, 0xf, true);
  any |= __builtin_amdgcn_update_dpp(0, any, CM_DPP_XOR2, 0xf, 0xf, true);
  any |= __builtin_amdgcn_update_dpp(0, any, CM_DPP_HALF_MIRROR, 0xf, 0xf, true);
  any |= __builtin_amdgcn_update_dpp(0, any, CM_DPP_MIRROR, 0xf, 0xf, true);
  if (ok && j == 0) { R[k] = s; if (colany) colany[k] = (uint8_t)(any != 0); }
}

// Two trial columns per 16-lane segment (batches: the launch is latency-bound at full occupancy -- half the wavefronts, each with the
// three load levels of both of its columns in flight together).  Column k and column k + ceil(K / 2); every column is summed exactly
// as cm_col_dot16 sums it (same entries per lane, same segment reduction, same 0.0 + x), so R is bit for bit the same.
// (columns k0 and k1 given explicitly; ok0 / ok1: the column exists -- uniform over the segment)
__device__ __forceinline__ void cm_col_dot16x2_k(int k0, bool ok0, int k1, bool ok1, int j, int K, const double* __restrict__ mu, const double* __restrict__ lam,
                                                 const int* __restrict__ col_ptr, const int* __restrict__ csc_row,
                                                 const int* __restrict__ csc_pos, double orth, double* __restrict__ R,
                                                 uint8_t* __restrict__ colany) {
  const int beg0 = ok0 ? col_ptr[k0] : 0, end0 = ok0 ? col_ptr[k0 + 1] : 0;
  const int beg1 = ok1 ? col_ptr[k1] : 0, end1 = ok1 ? col_ptr[k1 + 1] : 0;
  if (end0 - beg0 > 16 || end1 - beg1 > 16) {        // (uniform over the segment) long columns: one after the other, the plain way
    if (ok0) cm_col_dot16((k0 << 4) | j, K, mu, lam, col_ptr, csc_row, csc_pos, orth, R, colany);
    if (ok1) cm_col_dot16((k1 << 4) | j, K, mu, lam, col_ptr, csc_row, csc_pos, orth, R, colany);   // (DPP sources lie inside the segment)
    return;
  }
  const int ce0 = beg0 + j, ce1 = beg1 + j;
  const bool in0 = ce0 < end0, in1 = ce1 < end1;
  const int p0 = in0 ? csc_pos[ce0] : 0, r0 = in0 ? csc_row[ce0] : 0;
  const int p1 = in1 ? csc_pos[ce1] : 0, r1 = in1 ? csc_row[ce1] : 0;
  const double m0 = in0 ? mu[r0] : 0.0, m1 = in1 ? mu[r1] : 0.0;
  // (without the "any" output lam is gathered only where mu != 0: mu * lam = +0 exactly otherwise -- see cm_col_dot16)
  const double l0 = (in0 && (colany || m0 != 0.0)) ? lam[p0] : 0.0;
  const double l1 = (in1 && (colany || m1 != 0.0)) ? lam[p1] : 0.0;
  double s0 = 0.0, s1 = 0.0;
  if (end0 > beg0) s0 += cm_seg_sum<16>(in0 ? m0 * l0 : 0.0);          // (a column without entries adds nothing: s stays +0.0)
  if (end1 > beg1) s1 += cm_seg_sum<16>(in1 ? m1 * l1 : 0.0);
  int any0 = in0 && (l0 >= orth), any1 = in1 && (l1 >= orth);
  int any = any0 | (any1 << 1);
  any |= __builtin_amdgcn_update_dpp(0, any, CM_DPP_XOR1, 0xf, 0xf, true);
  any |= __builtin_amdgcn_update_dpp(0, any, CM_DPP_XOR2, 0xf, 0xf, true);
  any |= __builtin_amdgcn_update_dpp(0, any, CM_DPP_HALF_MIRROR, 0xf, 0xf, true);
  any |= __builtin_amdgcn_update_dpp(0, any, CM_DPP_MIRROR, 0xf, 0xf, true);
  if (j == 0) {
    if (ok0) { R[k0] = s0; if (colany) colany[k0] = (uint8_t)((any & 1) != 0); }
    if (ok1) { R[k1] = s1; if (colany) colany[k1] = (uint8_t)((any & 2) != 0); }
  }
}
__device__ __forceinline__ void cm_col_dot16x2(int t, int K, const double* __restrict__ mu, const double* __restrict__ lam,
                                               const int* __restrict__ col_ptr, const int* __restrict__ csc_row,
                                               const int* __restrict__ csc_pos, double orth, double* __restrict__ R,
                                               uint8_t* __restrict__ colany) {
  const int Kh = (K + 1) >> 1;
  const int k0 = t >> 4, k1 = k0 + Kh;
  const bool ok0 = k0 < Kh, ok1 = ok0 && k1 < K;
  cm_col_dot16x2_k(k0, ok0, k1, ok1, t & 15, K, mu, lam, col_ptr, csc_row, csc_pos, orth, R, colany);
}

// The tickets of the multi-wave chain (k_sweep_par), one trial column per 16-lane segment half as in cm_col_dot16x2: an entry's ticket is the
// number of EARLIER chain rows (mu != 0, by sweep position pos[]) that touch its trial.  While every row is a chain row (the first iterations)
// k_seq_pack spent most of its time finding them entry by entry (the column walked once per entry: 2 x 10 dependent gathers each); here the
// column is gathered once for all its entries, next to the column sum that gathers the same indices.  turn8[CSR slot] <- ticket (chain rows only).
template <int N_>
__device__ __forceinline__ int cm_row_ror(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x120 + N_, 0xf, 0xf, true); }   // lane i <- lane (i - N_) mod 16 of its row
__device__ __forceinline__ int cm_rank16(int key) {     // number of lanes of the 16-lane row whose key is below this lane's (INT_MAX: not a chain row)
  int r = 0;
#define CM_RK(n) r += (cm_row_ror<n>(key) < key);
  CM_RK(1) CM_RK(2) CM_RK(3) CM_RK(4) CM_RK(5) CM_RK(6) CM_RK(7) CM_RK(8) CM_RK(9) CM_RK(10) CM_RK(11) CM_RK(12) CM_RK(13) CM_RK(14) CM_RK(15)
#undef CM_RK
  return r;
}
// a column of more than 16 entries: every entry counts over the whole column (rare: more than 16 targets per trial)
__device__ __forceinline__ void cm_col_rank_long(int k, int j, const double* __restrict__ mu, const int* __restrict__ pos, const int* __restrict__ col_ptr,
                                                 const int* __restrict__ csc_row, const int* __restrict__ csc_pos, uint8_t* __restrict__ turn8) {
  const int beg = col_ptr[k], end = col_ptr[k + 1];
  for (int ce = beg + j; ce < end; ce += 16) {
    const int r = csc_row[ce];
    if (mu[r] == 0.0) continue;
    const int mine = pos[r];
    int t = 0;
    for (int c2 = beg; c2 < end; ++c2) { const int r2 = csc_row[c2]; t += (mu[r2] != 0.0 && pos[r2] < mine); }
    turn8[csc_pos[ce]] = (uint8_t)t;
  }
}
__device__ __forceinline__ void cm_col_rank16x2(int t, int K, const double* __restrict__ mu, const int* __restrict__ pos, const int* __restrict__ col_ptr,
                                                const int* __restrict__ csc_row, const int* __restrict__ csc_pos, uint8_t* __restrict__ turn8) {
  const int Kh = (K + 1) >> 1;
  const int k0 = t >> 4, k1 = k0 + Kh, j = t & 15;
  const bool ok0 = k0 < Kh, ok1 = ok0 && k1 < K;
  const int beg0 = ok0 ? col_ptr[k0] : 0, end0 = ok0 ? col_ptr[k0 + 1] : 0;
  const int beg1 = ok1 ? col_ptr[k1] : 0, end1 = ok1 ? col_ptr[k1 + 1] : 0;
  if (end0 - beg0 > 16 || end1 - beg1 > 16) {        // (uniform over the segment)
    if (ok0) cm_col_rank_long(k0, j, mu, pos, col_ptr, csc_row, csc_pos, turn8);
    if (ok1) cm_col_rank_long(k1, j, mu, pos, col_ptr, csc_row, csc_pos, turn8);
    return;
  }
  const int ce0 = beg0 + j, ce1 = beg1 + j;
  const bool in0 = ce0 < end0, in1 = ce1 < end1;
  const int p0 = in0 ? csc_pos[ce0] : 0, r0 = in0 ? csc_row[ce0] : 0;
  const int p1 = in1 ? csc_pos[ce1] : 0, r1 = in1 ? csc_row[ce1] : 0;
  const bool c0 = in0 && mu[r0] != 0.0, c1 = in1 && mu[r1] != 0.0;
  const int key0 = c0 ? pos[r0] : 0x7fffffff, key1 = c1 ? pos[r1] : 0x7fffffff;
  const int t0 = cm_rank16(key0), t1 = cm_rank16(key1);
  if (c0) turn8[p0] = (uint8_t)t0;
  if (c1) turn8[p1] = (uint8_t)t1;
}
// (one circuit: a column per 16-lane segment, cm_col_dot16's layout)
__device__ __forceinline__ void cm_col_rank16(int t, int K, const double* __restrict__ mu, const int* __restrict__ pos, const int* __restrict__ col_ptr,
                                              const int* __restrict__ csc_row, const int* __restrict__ csc_pos, uint8_t* __restrict__ turn8) {
  const int k = t >> 4, j = t & 15;
  const bool ok = k < K;
  const int beg = ok ? col_ptr[k] : 0, end = ok ? col_ptr[k + 1] : 0;
  if (end - beg > 16) { cm_col_rank_long(k, j, mu, pos, col_ptr, csc_row, csc_pos, turn8); return; }
  const int ce = beg + j;
  const bool in = ce < end;
  const int pp = in ? csc_pos[ce] : 0, r = in ? csc_row[ce] : 0;
  const bool c = in && mu[r] != 0.0;
  const int key = c ? pos[r] : 0x7fffffff;
  const int tt = cm_rank16(key);
  if (c) turn8[pp] = (uint8_t)tt;
}

// LISTED trials (cm_pdev::tstamp ...): the column sums of the trials on the list(s) -- every listed column exactly as cm_col_dot16 sums it -- in a
// stride loop over 16-lane segments; `seg` = this segment's index among `nseg` (both uniform over the segment).  ls / ns: the list k_stats left,
// lr / nr: the trials of the rows the pre-sweep revived (nr = 0: none).
__device__ __forceinline__ void cm_col_dot16_listed(int seg, int nseg, int j, const int* __restrict__ ls, int ns, const int* __restrict__ lr, int nr,
                                                    int K, const double* __restrict__ mu, const double* __restrict__ lam,
                                                    const int* __restrict__ col_ptr, const int* __restrict__ csc_row, const int* __restrict__ csc_pos,
                                                    double orth, double* __restrict__ R, uint8_t* __restrict__ colany) {
  const int cnt = ns + nr, half = (cnt + 1) >> 1;   // two listed trials per segment (entries i and i + half), their load levels in flight together
  for (int i = seg; i < half; i += nseg) {
    const int i1 = i + half;
    const int k0 = i < ns ? ls[i] : lr[i - ns];
    const bool ok1 = i1 < cnt;
    const int k1 = ok1 ? (i1 < ns ? ls[i1] : lr[i1 - ns]) : 0;
    cm_col_dot16x2_k(k0, true, k1, ok1, j, K, mu, lam, col_ptr, csc_row, csc_pos, orth, R, colany);
  }
}
// ... and zero for every trial that is not listed: one thread per trial.  stamp_lo .. stamp_hi: the stamps that mean "listed".
__device__ __forceinline__ void cm_col_zero_unlisted(int k, int K, const int* __restrict__ stamp, int stamp_lo, int stamp_hi,
                                                     double* __restrict__ R, uint8_t* __restrict__ colany) {
  if (k >= K) return;
  const int st = stamp[k];
  if (st >= stamp_lo && st <= stamp_hi) return;
  R[k] = 0.0;
  if (colany) colany[k] = 0;
}

// The single-workgroup phases run on the first CM_POST_THREADS threads of their workgroup, whatever its launch size (the other
// waves have retired; a retired wave no longer counts at s_barrier): the strided partial sums and their combination order are
// then the same in every kernel that embeds these bodies, so a batched launch reproduces the single-circuit results bit for bit.
#define CM_POST_THREADS 256
#define CM_Z_SPEC 8            // soft-threshold passes evaluated per barrier pair (post_single_body)

// sum over the first `nthr` threads of the workgroup, result valid in every one of them
__device__ __forceinline__ double block_sum(double v, double* sh, int nthr) {
  v = cm_wave_sum(v);
  __syncthreads();
  if (threadIdx.x % CM_WAVE == 0) sh[threadIdx.x / CM_WAVE] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < nthr / CM_WAVE; ++w) t += sh[w];
  return t;
}

// ------------------------------------------------------------------------------------------------
// Everything of an iteration that is a single-workgroup reduction (the first CM_POST_THREADS threads of the workgroup):
//   1. ordered compaction of the active rows (lam row != 0) for the next block_update_mu, and the size of that set
//      posted straight into host-mapped pinned memory (mailbox) so the host can size the next solve while 2-3 still run;
//   2. update_sigma: rate = rate0 + 1/2 ( sum_k (y_k - R_k)^2 - sum_n mu_n^2 sum_k lam_nk^2 + sum_n (mu_n^2 + beta_n^2) sum_k lam_nk )
//   3. z: <= (max_iters - it) penalty steps pen_j = penalty * scale^j while err > tol (the loop counter is seeded with
//      the OUTER iteration index and err with sum(y), caviar.py:86; quirk 4: nothing happens once it >= max_iters),
//      then spont_rate = mean(z != 0) (caviar.py:88).
// Returns the size of the active set (same value in every thread).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int post_single_body(const double* __restrict__ y, const double* __restrict__ R,
                                                 const uint8_t* __restrict__ colany, const uint8_t* __restrict__ mask,
                                                 const double* __restrict__ mu, const double* __restrict__ beta,
                                                 const double* __restrict__ rs, const double* __restrict__ sum_y, int N, int K,
                                                 int it, int max_iters, double penalty, double scale_factor, double tol,
                                                 int do_sigma, int seq, int* __restrict__ act_idx, int* __restrict__ act_inv,
                                                 int* __restrict__ n_act, int* mailbox, double* __restrict__ z,
                                                 double* __restrict__ scal, double* lds_scratch = nullptr, long long* dbg = nullptr,
                                                 int parts = 3) {
  // parts: 1 = the active set and update_sigma, 2 = the soft threshold and spont_rate.  The two halves share no data (rs, mu,
  // beta vs y - R, colany, mask), so the fused mid-iteration kernel runs them in two workgroups side by side; each half sums in
  // the same order whether it runs alone or after the other.
  __shared__ double sh[16];
  __shared__ int wave_cnt[16];
  __shared__ int base;
  constexpr int nthr_ = CM_POST_THREADS;
  int n_active = 0;
  if (parts & 1) {
  // ---- 1. active set
  if (dbg && threadIdx.x == 0) dbg[0] = wall_clock64();
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  {
    const int lane = threadIdx.x % CM_WAVE, w = threadIdx.x / CM_WAVE;
    const int nthr = nthr_, nwv = nthr_ / CM_WAVE;
    // (the rounds' flags are loaded together: four dependent round trips at N = 1000 were most of this phase's 4 us)
    constexpr int PRE = 8;
    bool pre[PRE];
#pragma unroll
    for (int q = 0; q < PRE; ++q) { const int n = q * nthr + (int)threadIdx.x; pre[q] = n < N && rs[(size_t)n * 4 + 3] >= 0.0; }
    for (int n0 = 0, q = 0; n0 < N; n0 += nthr, ++q) {
      const int n = n0 + threadIdx.x;
      bool a;
      switch (q) {      // (static register indices)
        case 0: a = pre[0]; break; case 1: a = pre[1]; break; case 2: a = pre[2]; break; case 3: a = pre[3]; break;
        case 4: a = pre[4]; break; case 5: a = pre[5]; break; case 6: a = pre[6]; break; case 7: a = pre[7]; break;
        default: a = n < N && rs[(size_t)n * 4 + 3] >= 0.0;
      }
      const unsigned long long m = __ballot(a);
      if (lane == 0) wave_cnt[w] = __popcll(m);
      __syncthreads();
      int off = base;
      for (int i = 0; i < w; ++i) off += wave_cnt[i];
      if (n < N) {
        if (a) {
          const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
          act_idx[pos] = n;
          act_inv[n] = pos;
        } else {
          act_inv[n] = -1;
        }
      }
      __syncthreads();
      if (threadIdx.x == 0) { int t = 0; for (int i = 0; i < nwv; ++i) t += wave_cnt[i]; base += t; }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      n_act[0] = base;
      if (mailbox) {
        // host-mapped (fine-grained) pinned memory.  Everything the host needs travels in ONE naturally aligned 64-bit word
        // [seq : 32 | not-SPD flag : 1 | active-set size : 31], stored relaxed at system scope: single-copy atomic, so no
        // release fence (= no L2 write-back, which costs tens of microseconds right after kernels that dirtied a lot of L2).
        const unsigned long long word = ((unsigned long long)(unsigned)seq << 32) | ((unsigned long long)(n_act[1] != 0) << 31) |
                                        (unsigned long long)(unsigned)base;
        __hip_atomic_store((unsigned long long*)mailbox, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  __syncthreads();
  n_active = base;
  if (dbg && threadIdx.x == 0) dbg[1] = wall_clock64();
  if (!do_sigma) return n_active;
  // ---- 2. update_sigma
  double a = 0.0;
  // (unrolled for the loads' sake: the sums keep their order)
#pragma unroll 8
  for (int k = threadIdx.x; k < K; k += nthr_) { const double r = y[k] - R[k]; a = fma(r, r, a); }
  double b = 0.0, c = 0.0;
#pragma unroll 4
  for (int n = threadIdx.x; n < N; n += nthr_) {
    const double m = mu[n], be = beta[n];
    b = fma(m * m, rs[(size_t)n * 4 + 1], b);
    c = fma(m * m + be * be, rs[(size_t)n * 4 + 0], c);
  }
  a = block_sum(a, sh, nthr_); b = block_sum(b, sh, nthr_); c = block_sum(c, sh, nthr_);
  if (threadIdx.x == 0) {
    scal[CM_S_SHAPE] = scal[CM_S_SHAPE_PRIOR] + (double)K / 2.0;
    scal[CM_S_RATE] = scal[CM_S_RATE_PRIOR] + 0.5 * (a - b + c);
  }
  if (dbg && threadIdx.x == 0) dbg[2] = wall_clock64();
  }   // parts & 1
  if (!(parts & 2) || !do_sigma) return n_active;
  if (dbg && threadIdx.x == 0) dbg[7] = wall_clock64();
  // ---- 3. estimate_spont_act_soft_thresh.  The residual and the "blocked" flag (column has a confident spike, or trial
  // masked out) do not change across the penalty steps: with LDS scratch they are staged once and the up-to-20 steps run
  // out of LDS instead of re-reading four global arrays each.
  double err = sum_y[0];
  const double ysq = sum_y[1] + 1e-5;
  double pen = penalty;
  int itc = it;
  const bool staged = lds_scratch != nullptr && itc < max_iters && err > tol;
  double* rl = lds_scratch;
  uint8_t* bl = (uint8_t*)(lds_scratch + K);
  if (staged) {
    for (int k = threadIdx.x; k < K; k += nthr_) { rl[k] = y[k] - R[k]; bl[k] = (uint8_t)(colany[k] || !mask[k]); }
    __syncthreads();
  }
  // z itself is only needed from the LAST pass: the passes accumulate the error norm, and z is written once afterwards with the
  // penalty of that last pass (a store to z in every pass puts a global-store round trip inside each of up to 20 barriers).
  // The passes form a chain only through the loop test (the penalties pen_j = penalty * scale^j do not depend on the data): up to
  // CM_Z_SPEC consecutive passes are evaluated together -- every thread keeps one error accumulator per pass over its trials, one
  // barrier pair reduces them all -- and the sequential loop is then replayed on the results.  Each pass's error is summed
  // exactly as a pass on its own would sum it (same strided order per thread, same wave and block trees), so z, spont_rate and
  // the pass count are bit for bit those of the one-pass-at-a-time loop, at 1/8 of its barriers.
  bool ran = false;
  double last_pen = pen;
  __shared__ double zsh[CM_Z_SPEC][CM_POST_THREADS / CM_WAVE];
  while (itc < max_iters && err > tol) {
    const int np = (max_iters - itc) < CM_Z_SPEC ? (max_iters - itc) : CM_Z_SPEC;
    double pens[CM_Z_SPEC], e[CM_Z_SPEC];
    {
      double pp = pen;
#pragma unroll
      for (int j = 0; j < CM_Z_SPEC; ++j) { pens[j] = pp; pp *= scale_factor; e[j] = 0.0; }
    }
    for (int k = threadIdx.x; k < K; k += nthr_) {
      const double resid = staged ? rl[k] : y[k] - R[k];
      const bool blocked = staged ? (bl[k] != 0) : (colany[k] || !mask[k]);
#pragma unroll
      for (int j = 0; j < CM_Z_SPEC; ++j) {
        double zz = resid < pens[j] ? 0.0 : resid - pens[j];               // :154
        zz = zz < 0.0 ? 0.0 : zz;                                          // :155
        zz = blocked ? 0.0 : zz;                                           // :156-157
        const double d = resid - zz;
        e[j] = fma(d, d, e[j]);                                            // :158
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < CM_Z_SPEC; ++j) {
      const double v = cm_wave_sum(e[j]);
      if (threadIdx.x % CM_WAVE == 0) zsh[j][threadIdx.x / CM_WAVE] = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < CM_Z_SPEC; ++j) {
      if (j < np && (j == 0 || err > tol)) {                               // pass j runs iff the loop test held after pass j - 1
        double t = 0.0;
        for (int w = 0; w < nthr_ / CM_WAVE; ++w) t += zsh[j][w];
        err = t / ysq;
        ++itc;
        last_pen = pens[j];
        ran = true;
        pen = pens[j] * scale_factor;
      }
    }
  }
  double cnt = 0.0;
  if (ran) {
    for (int k = threadIdx.x; k < K; k += nthr_) {
      const double resid = staged ? rl[k] : y[k] - R[k];
      const bool blocked = staged ? (bl[k] != 0) : (colany[k] || !mask[k]);
      double zz = resid < last_pen ? 0.0 : resid - last_pen;
      zz = zz < 0.0 ? 0.0 : zz;
      zz = blocked ? 0.0 : zz;
      z[k] = zz;
      cnt += (zz != 0.0) ? 1.0 : 0.0;
    }
  } else {
    for (int k = threadIdx.x; k < K; k += nthr_) cnt += (z[k] != 0.0) ? 1.0 : 0.0;   // z is frozen (caviar.py:86-87 with it >= 20)
  }
  cnt = block_sum(cnt, sh, nthr_);
  if (threadIdx.x == 0) scal[CM_S_SPONT_RATE] = cnt / (double)K;
  if (dbg && threadIdx.x == 0) { dbg[3] = wall_clock64(); dbg[5] = itc; }
  return n_active;
}

// ------------------------------------------------------------------------------------------------
// exp(x) for the Monte-Carlo logit stream (|x| < 20: no overflow / underflow / NaN handling), table-driven:
//   x = (64 m + j) ln2 / 64 + r,  |r| <= ln2 / 128:   exp(x) = 2^m * 2^(j/64) * (1 + r + r^2/2 + ... + r^5/120)
// 16 instructions against ~30 of the library routine (whose degree-11 polynomial serves |r| <= ln2 / 2); 1.3 ulp worst case measured
// over 2e6 arguments (library: 0.7) -- far inside what the consumer can see (see cm_logit_of_sigmoid: the error of e enters the logit as an
// ABSOLUTE 1e-16).  `tab` = the 64 correctly rounded values 2^(j/64) in LDS (cm_exptab_init).
static __device__ const double cm_exp2_64[64] = {
    0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
    0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,
    0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
    0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
    0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
    0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
    0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
    0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
    0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
    0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,
    0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
    0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
    0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
    0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
    0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0};
__device__ __forceinline__ void cm_exptab_init(double* tab) {       // first 64 threads of the workgroup; the caller synchronises
  if (threadIdx.x < 64) tab[threadIdx.x] = cm_exp2_64[threadIdx.x];
}
__device__ __forceinline__ double cm_exp_tab(double x, const double* tab) {
  // k = rint(64 x / ln2) by the 1.5 * 2^52 shift: the sum's low word is k in two's complement, the difference is k as a double.
  // (v_rndne_f64, v_cvt_i32_f64 and v_ldexp_f64 issue at a quarter of the fma rate: three of them cost more than the polynomial.)
  const double y = fma(x, 0x1.71547652b82fep+6, 0x1.8p52);     // 64 / ln2
  const double kf = y - 0x1.8p52;
  const int k = __double2loint(y);
  double r = fma(kf, -0x1.62e42fefa3000p-7, x);                // ln2 / 64, 41 significant bits: the product is exact
  r = fma(kf, -0x1.3de6af278ece6p-48, r);
  double q = fma(r, 1.0 / 120.0, 1.0 / 24.0);
  q = fma(r, q, 1.0 / 6.0);
  q = fma(r, q, 0.5);
  q = fma(r, q, 1.0);
  const double t = tab[k & 63];
  const double v = fma(t, r * q, t);                           // in [1, 2): scaling by 2^(k >> 6) is an add to the exponent field
  return __hiloint2double(__double2hiint(v) + ((k >> 6) << 20), __double2loint(v));
}

// ------------------------------------------------------------------------------------------------
// Table-driven exp / log for the Newton objective of update_phi (the instruction stream of that 10-step chain is what bounds the
// per-neuron role; the library routines are ~30 / ~70 fp64 instructions, these 16 / 25).  Accuracy: exp 1.3 ulp (cm_exp_tab, below);
// log <= 6 ulp in general and <= 1.2e-16 ABSOLUTE for arguments in [0.5, 1] (where the result is small and the library's relative
// accuracy buys nothing: the values are summed into an objective of order 1 - 100).  Arguments outside the fast range -- zero,
// negative, NaN, infinite, subnormal, |x| >= 700 for exp -- take the library routine, so the nan_to_num classes of the reference's objective
// (quirk 11) see exactly the special values they saw before.
// log(x), x positive normal: x = 2^e m, m in [1, 2), table index i = top six mantissa bits, c_i the interval's centre:
//   r = m / c_i - 1 (one fma with the tabulated 1 / c_i), |r| <= 2^-7;  log x = e' ln2 + log c_i' + log1p(r), e' = e + (i >= 27), c_i' = c_i / 2 there
// (so that arguments just below 1 meet a small table value: no cancellation against ln 2).
static __device__ const double cm_log_tab128[128] = {   // [0..63] 1 / c_i, [64..127] log(c_i) (- ln 2 from i = 27 on)
    0x1.fc07f01fc07f0p-1, 0x1.f44659e4a4271p-1, 0x1.ecc07b301ecc0p-1, 0x1.e573ac901e574p-1,
    0x1.de5d6e3f8868ap-1, 0x1.d77b654b82c34p-1, 0x1.d0cb58f6ec074p-1, 0x1.ca4b3055ee191p-1,
    0x1.c3f8f01c3f8f0p-1, 0x1.bdd2b899406f7p-1, 0x1.b7d6c3dda338bp-1, 0x1.b2036406c80d9p-1,
    0x1.ac5701ac5701bp-1, 0x1.a6d01a6d01a6dp-1, 0x1.a16d3f97a4b02p-1, 0x1.9c2d14ee4a102p-1,
    0x1.970e4f80cb872p-1, 0x1.920fb49d0e229p-1, 0x1.8d3018d3018d3p-1, 0x1.886e5f0abb04ap-1,
    0x1.83c977ab2beddp-1, 0x1.7f405fd017f40p-1, 0x1.7ad2208e0ecc3p-1, 0x1.767dce434a9b1p-1,
    0x1.724287f46debcp-1, 0x1.6e1f76b4337c7p-1, 0x1.6a13cd1537290p-1, 0x1.661ec6a5122f9p-1,
    0x1.623fa77016240p-1, 0x1.5e75bb8d015e7p-1, 0x1.5ac056b015ac0p-1, 0x1.571ed3c506b3ap-1,
    0x1.5390948f40febp-1, 0x1.5015015015015p-1, 0x1.4cab88725af6ep-1, 0x1.49539e3b2d067p-1,
    0x1.460cbc7f5cf9ap-1, 0x1.42d6625d51f87p-1, 0x1.3fb013fb013fbp-1, 0x1.3c995a47babe7p-1,
    0x1.3991c2c187f63p-1, 0x1.3698df3de0748p-1, 0x1.33ae45b57bcb2p-1, 0x1.30d190130d190p-1,
    0x1.2e025c04b8097p-1, 0x1.2b404ad012b40p-1, 0x1.288b01288b013p-1, 0x1.25e22708092f1p-1,
    0x1.23456789abcdfp-1, 0x1.20b470c67c0d9p-1, 0x1.1e2ef3b3fb874p-1, 0x1.1bb4a4046ed29p-1,
    0x1.19453808ca29cp-1, 0x1.16e0689427379p-1, 0x1.1485f0e0acd3bp-1, 0x1.12358e75d3033p-1,
    0x1.0fef010fef011p-1, 0x1.0db20a88f4696p-1, 0x1.0b7e6ec259dc8p-1, 0x1.0953f39010954p-1,
    0x1.073260a47f7c6p-1, 0x1.05197f7d73404p-1, 0x1.03091b51f5e1ap-1, 0x1.0101010101010p-1,
    0x1.fe02a6b106799p-8, 0x1.7b91b07d5b126p-6, 0x1.39e87b9febd68p-5, 0x1.b42dd711971b9p-5,
    0x1.16536eea37ae3p-4, 0x1.51b073f06183cp-4, 0x1.8c345d6319b23p-4, 0x1.c5e548f5bc743p-4,
    0x1.fec9131dbeabcp-4, 0x1.1b72ad52f67a2p-3, 0x1.371fc201e8f75p-3, 0x1.526e5e3a1b438p-3,
    0x1.6d60fe719d21bp-3, 0x1.87fa06520c911p-3, 0x1.a23bc1fe2b561p-3, 0x1.bc286742d8cd4p-3,
    0x1.d5c216b4fbb94p-3, 0x1.ef0adcbdc5935p-3, 0x1.0402594b4d041p-2, 0x1.1058bf9ae4ad4p-2,
    0x1.1c898c16999fbp-2, 0x1.2895a13de86a4p-2, 0x1.347dd9a987d56p-2, 0x1.404308686a7e4p-2,
    0x1.4be5f957778a1p-2, 0x1.5767717455a6cp-2, 0x1.62c82f2b9c796p-2, -0x1.57bf753c8d1fbp-2,
    -0x1.4c9e09e172c3dp-2, -0x1.419b423d5e8c6p-2, -0x1.36b6776be1116p-2, -0x1.2bef07cdc9355p-2,
    -0x1.214456d0eb8d5p-2, -0x1.16b5ccbacfb73p-2, -0x1.0c42d676162e2p-2, -0x1.01eae5626c691p-2,
    -0x1.ef5ade4dcffe5p-3, -0x1.db13db0d48941p-3, -0x1.c6ffbc6f00f71p-3, -0x1.b31d8575bce3bp-3,
    -0x1.9f6c407089663p-3, -0x1.8beafeb38fe8fp-3, -0x1.7898d85444c74p-3, -0x1.6574ebe8c1339p-3,
    -0x1.527e5e4a1b58dp-3, -0x1.3fb45a59928cap-3, -0x1.2d1610c86813dp-3, -0x1.1aa2b7e23f729p-3,
    -0x1.08598b59e3a07p-3, -0x1.ec739830a1126p-4, -0x1.c885801bc4b20p-4, -0x1.a4e7640b1bc38p-4,
    -0x1.8197e2f40e3f0p-4, -0x1.5e95a4d9791cdp-4, -0x1.3bdf5a7d1ee5ep-4, -0x1.1973bd1465561p-4,
    -0x1.eea31c006b87cp-5, -0x1.aaef2d0fb1108p-5, -0x1.67c94f2d4bb65p-5, -0x1.252f32f8d1840p-5,
    -0x1.c63d2ec14aad7p-6, -0x1.432a925980cbcp-6, -0x1.82448a388a283p-7, -0x1.0080559588b25p-8,
};
__device__ __forceinline__ void cm_logtab_init(double* tab) {       // first 128 threads of the workgroup; the caller synchronises
  if (threadIdx.x < 128) tab[threadIdx.x] = cm_log_tab128[threadIdx.x];
}
__device__ __forceinline__ double cm_log_tab(double x, const double* tab) {
  if (!(x >= 0x1p-1022 && x < INFINITY)) return log(x);
  const int hi = __double2hiint(x);
  const int i = (hi >> 14) & 63;
  const double m = __hiloint2double((hi & 0x000FFFFF) | 0x3FF00000, __double2loint(x));
  const double r = fma(m, tab[i], -1.0);
  const double ef = (double)(((hi >> 20) & 0x7FF) - 1023 + (i >= 27));
  double q = fma(r, 1.0 / 7.0, -1.0 / 6.0);
  q = fma(r, q, 0.2);
  q = fma(r, q, -0.25);
  q = fma(r, q, 1.0 / 3.0);
  q = fma(r, q, -0.5);
  const double pl = fma(r * r, q, r);
  return fma(ef, 0x1.62e42fee00000p-1, tab[64 + i] + pl) + ef * 0x1.a39ef35793c76p-33;
}

// ------------------------------------------------------------------------------------------------
// update_phi (caviar.py:246-316).  The K-sums of the Newton iteration collapse onto per-stimulus-value sufficient
// statistics because I[n,:] takes <= V distinct values.  nan_to_num is applied per ELEMENT in the reference objective
// (quirk 11) so elements are classed by y == 0, y == 1, 0 < y < 1.
// ------------------------------------------------------------------------------------------------

// full-precision reciprocal from the hardware estimate + two Newton steps (normal-range operands)
__device__ __forceinline__ double cm_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
}

__device__ __forceinline__ void inv2x2(double h11, double h12, double h21, double h22, double o[4]) {
  // LU with partial pivoting on the 2x2 (the structure of LAPACK getrf/getri behind jnp.linalg.inv); the two pivots are
  // inverted once each (reciprocal + Newton) instead of dividing by them repeatedly -- this sits on the Newton chain
  double a = h11, b = h12, c = h21, d = h22;
  const bool sw = fabs(c) > fabs(a);
  if (sw) { double t = a; a = c; c = t; t = b; b = d; d = t; }
  const double ra = cm_rcp(a);
  const double l = c * ra;
  const double ru = cm_rcp(d - l * b);
  // columns of the inverse: solve for P e0 and P e1
  double r0 = sw ? 0.0 : 1.0, r1 = sw ? 1.0 : 0.0;
  double x1 = (r1 - l * r0) * ru;
  o[0] = (r0 - b * x1) * ra; o[2] = x1;
  r0 = sw ? 1.0 : 0.0; r1 = sw ? 0.0 : 1.0;
  x1 = (r1 - l * r0) * ru;
  o[1] = (r0 - b * x1) * ra; o[3] = x1;
}

// negloglik_with_barrier (caviar.py:312-316) on the sufficient statistics of one stimulus value per lane.  Instruction count is
// what bounds this path (it sits on the 10-step Newton chain of every neuron), so the four logarithms of the literal form
// are reduced to two, with the per-element nan_to_num classes of the reference kept exact:
//   * log(s) and log(1 - s), s = sigmoid(x), come from ONE logarithm: for x > 0  ls = log(s), l1s = ls - x; otherwise
//     l1s = log(1 - s), ls = l1s + x (log(1 - s) = log(s) - x identically; each branch takes the logarithm of the operand that is
//     not small, so the result is at least as accurate as the literal pair).  s == 1 and s == 0 -- the saturated sigmoid, where
//     the literal form produces log(0) = -inf and nan_to_num turns the element into -DBL_MAX (quirk 11) -- are restored by hand;
//   * the barrier's log(phi_0) + log(phi_1) is evaluated by two lanes of the segment in one instruction stream (lane 0: phi_0,
//     lane 1: phi_1) and enters the segment sum; negative phi gives NaN as in the literal form.
// sigmoid with the table-driven exp (the library's beyond |x| = 700 and for NaN); phi_body's gradient / Hessian and the objective
// must evaluate the SAME function (the accepted candidate's sigmoid is reused by the next Newton step)
__device__ __forceinline__ double cm_sigmoid_tab(double x, const double* etab) {
  const double e = fabs(x) < 700.0 ? cm_exp_tab(-x, etab) : exp(-x);
  return 1.0 / (1.0 + e);
}
template <int SEG>
__device__ __forceinline__ double phi_nll(double p0, double p1, double val, double n0, double n1, double smid, double nmid,
                                          bool lane_on, int g, double pr0, double pr1, const double prec[4], double t, double* sig_out,
                                          const double* etab, const double* ltab) {
  double tot = 0.0;
  *sig_out = 0.0;
  if (lane_on) {
    const double x = p0 * val - p1;
    const double s = cm_sigmoid_tab(x, etab);
    *sig_out = s;
    const bool pos = x > 0.0;
    const double la = cm_log_tab(pos ? s : 1.0 - s, ltab);
    double ls = pos ? la : la + x, l1s = pos ? la - x : la;
    if (s == 1.0) l1s = -INFINITY;                     // log(1 - s) = log(0)
    if (s == 0.0) ls = -INFINITY;                      // log(s) = log(0)
    if (n0 > 0.0) tot += n0 * cm_nan_to_num(0.0 * ls + 1.0 * l1s);
    if (n1 > 0.0) tot += n1 * cm_nan_to_num(1.0 * ls + 0.0 * l1s);
    if (nmid > 0.0) {
      if (isfinite(ls) && isfinite(l1s)) tot += smid * ls + (nmid - smid) * l1s;
      else tot += nmid * cm_nan_to_num(0.5 * ls + 0.5 * l1s);
    }
  }
  double barrier;
  if (SEG >= 2) {
    tot += cm_log_tab(g == 0 ? p0 : (g == 1 ? p1 : 1.0), ltab) / t;  // lanes 0, 1 of the segment: log(phi_0) / t, log(phi_1) / t; the others add log(1) = 0
    tot = cm_seg_sum<SEG>(tot);
    barrier = 0.0;
  } else {
    tot = cm_seg_sum<SEG>(tot);
    barrier = (cm_log_tab(p0, ltab) + cm_log_tab(p1, ltab)) / t;
  }
  const double d0 = p0 - pr0, d1 = p1 - pr1;
  const double quad = d0 * (prec[0] * d0 + prec[1] * d1) + d1 * (prec[2] * d0 + prec[3] * d1);
  return -tot - barrier + 0.5 * quad;
}

// One wavefront per neuron.  SEG = lanes per evaluation point (power of two >= V; lane g of a segment owns stimulus
// value g) and E = 64 / SEG evaluation points per wave: in one pass segment 0 evaluates the objective at phi (the
// right-hand side of the Armijo test) while segments 1..E-1 evaluate it at phi + 2^-b v for consecutive b, so the
// reference's sequential backtracking (caviar.py:259-268: accept the first b whose test passes, at most 40 halvings)
// collapses into one parallel evaluation per Newton step in the common case.
// DEAD = true: the row is taken to be identically zero (no scan of lam): the result then depends only on the trial counts and the
// priors of the neuron, i.e. it is a per-run constant -- computed once (k_phi_dead) and copied by rows that are all zero.
template <int SEG, bool DEAD = false>
__device__ __forceinline__ void phi_body(int n, int lane, const double* __restrict__ lam, const int* __restrict__ row_ptr,
                                         const uint8_t* __restrict__ grp, const int* __restrict__ rowcnt,
                                         const double* __restrict__ vals, int V,
                                         const double* __restrict__ phi_prior, const double* __restrict__ phi_cov_prior,
                                         double* __restrict__ phi, double* __restrict__ phi_cov, double out[4],
                                         const double* etab, const double* ltab,
                                         const double* __restrict__ dead_phi = nullptr, const double* __restrict__ dead_cov = nullptr) {
  constexpr int E = CM_WAVE / SEG;
  const int g = lane % SEG, ev = lane / SEG;
  // sufficient statistics of value g: segment `ev` scans entries ev, ev + E, ...; partial sums are then combined
  // across the segments (lanes at stride SEG)
  const bool on = g < V;
  const double val = on ? vals[g] : 0.0;
  const double cnt = on ? (double)rowcnt[(size_t)n * CM_MAX_VALS + g] : 0.0;
  double n1 = 0.0, nmid = 0.0, smid = 0.0;
  if (!DEAD && on && val > 0.0) {              // untargeted values carry lam == 0 on every trial
    for (int e = row_ptr[n] + ev; e < row_ptr[n + 1]; e += E) {
      if (grp[e] != g) continue;
      const double l = lam[e];
      if (l == 1.0) n1 += 1.0;
      else if (l > 0.0 && l < 1.0) { nmid += 1.0; smid += l; }
    }
  }
#pragma unroll
  for (int o = SEG; o < CM_WAVE; o <<= 1) {
    n1 += __shfl_xor(n1, o, CM_WAVE); nmid += __shfl_xor(nmid, o, CM_WAVE); smid += __shfl_xor(smid, o, CM_WAVE);
  }
  const double n0 = cnt - n1 - nmid;
  const double sy = n1 + smid;
  if (!DEAD && dead_phi != nullptr && __all(n1 == 0.0 && nmid == 0.0)) {
    // the row turned out to be identically zero: its result is the per-run constant of this neuron (k_phi_dead ran this very
    // function on those statistics)
    const double p0c = dead_phi[2 * n], p1c = dead_phi[2 * n + 1];
    double cv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cv[i] = dead_cov[4 * n + i];
    if (lane == 0) {
      phi[2 * n] = p0c; phi[2 * n + 1] = p1c;
      phi_cov[4 * n] = cv[0]; phi_cov[4 * n + 1] = cv[1]; phi_cov[4 * n + 2] = cv[2]; phi_cov[4 * n + 3] = cv[3];
    }
    out[0] = p0c; out[1] = p1c; out[2] = cv[0]; out[3] = cv[3];
    return;
  }

  const double pr0 = phi_prior[2 * n], pr1 = phi_prior[2 * n + 1];
  double prec[4];
  inv2x2(phi_cov_prior[4 * n], phi_cov_prior[4 * n + 1], phi_cov_prior[4 * n + 2], phi_cov_prior[4 * n + 3], prec);   // caviar.py:306
  const double t = 1e1, alpha = 0.25;
  double p0 = pr0, p1 = pr1;                     // caviar.py:305: restart from the prior every call (quirk 3)
  double cov[4] = {0, 0, 0, 0};
  double fcur = 0.0;                              // sigmoid(p0 * val - p1) of this lane's value at the current phi
  bool have_f = false;
  for (int stepi = 0; stepi < 10; ++stepi) {     // caviar.py:308
    double j1 = 0.0, j2 = 0.0, h11 = 0.0, h12 = 0.0, h22 = 0.0;
    if (on) {
      // the accepted candidate of the previous step evaluated exactly this sigmoid (same expression, same operands)
      const double f = have_f ? fcur : cm_sigmoid_tab(p0 * val - p1, etab);
      const double r = sy - cnt * f;
      j1 = -val * r; j2 = r;
      const double ff = cnt * f * (1.0 - f);
      h11 = val * val * ff; h12 = -val * ff; h22 = ff;
    }
    j1 = cm_seg_sum<SEG>(j1); j2 = cm_seg_sum<SEG>(j2);
    h11 = cm_seg_sum<SEG>(h11); h12 = cm_seg_sum<SEG>(h12); h22 = cm_seg_sum<SEG>(h22);
    const double d0 = p0 - pr0, d1 = p1 - pr1;
    const double J0 = j1 + (prec[0] * d0 + prec[1] * d1) - 1.0 / (t * p0);      // caviar.py:278-280
    const double J1 = j2 + (prec[2] * d0 + prec[3] * d1) - 1.0 / (t * p1);
    inv2x2(h11 + prec[0] + 1.0 / (t * p0 * p0), h12 + prec[1], h12 + prec[2], h22 + prec[3] + 1.0 / (t * p1 * p1), cov);   // :283-289
    const double v0 = -(cov[0] * J0 + cov[1] * J1), v1 = -(cov[2] * J0 + cov[3] * J1);
    const double Jv = J0 * v0 + J1 * v1;
    // candidates b = 0..40 (step 2^-b); b <= 39 needs its Armijo test to pass, b = 40 is taken unconditionally
    int accepted = -1;
    double f0 = 0.0;
    int base = 0;                                 // first candidate of this round
    bool first = true;
    while (accepted < 0 && base < 40) {
      const int b = first ? base + ev - 1 : base + ev;       // first round: segment 0 evaluates phi itself
      const bool is_f0 = first && ev == 0;
      const double st = is_f0 ? 0.0 : ldexp(1.0, -b);
      const double q0 = is_f0 ? p0 : p0 + st * v0, q1 = is_f0 ? p1 : p1 + st * v1;
      double sg;
      const double fv = phi_nll<SEG>(q0, q1, val, n0, n1, smid, nmid, on, g, pr0, pr1, prec, t, &sg, etab, ltab);
      if (first) f0 = cm_readlane_f64(fv, 0);
      const double rhs = f0 + alpha * st * Jv;
      const bool ok = !is_f0 && b <= 39 && !(isnan(fv) || fv > rhs);
      const unsigned long long m = __ballot(ok);
      if (m) {
        const int seg = (int)(__ffsll((long long)m) - 1) / SEG;          // first accepting segment of this round
        accepted = (first ? base - 1 : base) + seg;
        fcur = __shfl(sg, seg * SEG + g, CM_WAVE);                        // its sigmoid for this lane's value
      }
      base += first ? E - 1 : E;
      first = false;
    }
    have_f = accepted >= 0;                        // b = 40 is taken without an evaluation: recompute next step
    if (accepted < 0) accepted = 40;
    const double step = ldexp(1.0, -accepted);
    const double q0 = p0 + step * v0, q1 = p1 + step * v1;                        // caviar.py:301
    // A step that leaves phi bit-for-bit unchanged is a fixed point of the loop body: every later step would recompute the same
    // J, H, covariance and (rejected) candidates and leave phi unchanged again, so the remaining steps are skipped -- exactly
    // the reference's result.  Converged neurons are the expensive ones otherwise: their Armijo test fails for all 40 candidates.
    const bool fixed = (q0 == p0) && (q1 == p1);
    p0 = q0; p1 = q1;
    if (fixed) break;
  }
  if (lane == 0) {
    phi[2 * n] = p0; phi[2 * n + 1] = p1;
    phi_cov[4 * n] = cov[0]; phi_cov[4 * n + 1] = cov[1]; phi_cov[4 * n + 2] = cov[2]; phi_cov[4 * n + 3] = cov[3];
  }
  out[0] = p0; out[1] = p1; out[2] = cov[0]; out[3] = cov[3];     // mean and variance diagonal, for the MC logit that follows
}


// TWO rows per wavefront, one per 32-lane half: lanes 0-31 carry row `n` of the lower half, lanes 32-63 that of the upper half (n < 0: the
// half has no row).  The Newton iteration is a chain of ~260-instruction steps whose time is its instruction stream: a ~20-entry row with
// four stimulus values left most of a 64-lane wavefront's evaluation points unused (16 of them, of which the first candidate is accepted in
// nearly every step); with a row per half the same stream serves two rows, each with E = 32 / SEG evaluation points (SEG = 4: the objective
// at phi and seven step candidates per round).  Each half runs exactly the single-row iteration -- the same sufficient statistics, the same
// sums within a segment, the first accepted candidate in the reference's order (caviar.py:259-268) -- so a row's result does not depend on
// its partner, on which half it sits in or on the kernel that evaluates it; a half that is finished (fixed point, or no row) idles through
// its partner's remaining steps.  SEG <= 16 (two evaluation points per half at least); larger SEG: phi_pair below falls back to phi_body.
template <int SEG, bool DEAD = false>
__device__ __forceinline__ void phi_body2(int n, int lane, const double* __restrict__ lam, const int* __restrict__ row_ptr,
                                          const uint8_t* __restrict__ grp, const int* __restrict__ rowcnt,
                                          const double* __restrict__ vals, int V,
                                          const double* __restrict__ phi_prior, const double* __restrict__ phi_cov_prior,
                                          double* __restrict__ phi, double* __restrict__ phi_cov, double out[4],
                                          const double* etab, const double* ltab,
                                          const double* __restrict__ dead_phi = nullptr, const double* __restrict__ dead_cov = nullptr) {
  static_assert(SEG <= 16, "two evaluation points per half");
  constexpr int E = 32 / SEG;
  const int hl = lane & 31, hb = lane & 32;
  const int g = hl % SEG, ev = hl / SEG;
  const bool row = n >= 0;
  const int nn = row ? n : 0;
  // sufficient statistics of value g: segment `ev` of the half scans entries ev, ev + E, ...; partial sums are then combined across the
  // half's segments (lanes at stride SEG)
  const bool on = row && g < V;
  const double val = on ? vals[g] : 0.0;
  const double cnt = on ? (double)rowcnt[(size_t)nn * CM_MAX_VALS + g] : 0.0;
  double n1 = 0.0, nmid = 0.0, smid = 0.0;
  if (!DEAD && on && val > 0.0) {              // untargeted values carry lam == 0 on every trial
    for (int e = row_ptr[nn] + ev; e < row_ptr[nn + 1]; e += E) {
      if (grp[e] != g) continue;
      const double l = lam[e];
      if (l == 1.0) n1 += 1.0;
      else if (l > 0.0 && l < 1.0) { nmid += 1.0; smid += l; }
    }
  }
#pragma unroll
  for (int o = SEG; o < 32; o <<= 1) {
    n1 += __shfl_xor(n1, o, CM_WAVE); nmid += __shfl_xor(nmid, o, CM_WAVE); smid += __shfl_xor(smid, o, CM_WAVE);
  }
  const double n0 = cnt - n1 - nmid;
  const double sy = n1 + smid;
  bool done = !row;
  out[0] = out[1] = out[2] = out[3] = 0.0;
  if (!DEAD && dead_phi != nullptr) {
    const unsigned zero_h = (unsigned)(__ballot(n1 == 0.0 && nmid == 0.0) >> hb);
    if (row && zero_h == 0xFFFFFFFFu) {
      // the row turned out to be identically zero: its result is the per-run constant of this neuron (k_phi_dead ran this very
      // function on those statistics)
      const double p0c = dead_phi[2 * nn], p1c = dead_phi[2 * nn + 1];
      double cv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) cv[i] = dead_cov[4 * nn + i];
      if (hl == 0) {
        phi[2 * nn] = p0c; phi[2 * nn + 1] = p1c;
        phi_cov[4 * nn] = cv[0]; phi_cov[4 * nn + 1] = cv[1]; phi_cov[4 * nn + 2] = cv[2]; phi_cov[4 * nn + 3] = cv[3];
      }
      out[0] = p0c; out[1] = p1c; out[2] = cv[0]; out[3] = cv[3];
      done = true;
    }
  }
  if (__all(done)) return;
  const bool live = !done;                        // this half runs the Newton iteration (and writes its result)
  const double pr0 = phi_prior[2 * nn], pr1 = phi_prior[2 * nn + 1];
  double prec[4];
  inv2x2(phi_cov_prior[4 * nn], phi_cov_prior[4 * nn + 1], phi_cov_prior[4 * nn + 2], phi_cov_prior[4 * nn + 3], prec);   // caviar.py:306
  const double t = 1e1, alpha = 0.25;
  double p0 = pr0, p1 = pr1;                     // caviar.py:305: restart from the prior every call (quirk 3)
  double cov[4] = {0, 0, 0, 0};
  double fcur = 0.0;                              // sigmoid(p0 * val - p1) of this lane's value at the current phi
  bool have_f = false;
  for (int stepi = 0; stepi < 10; ++stepi) {     // caviar.py:308
    if (__all(done)) break;
    double j1 = 0.0, j2 = 0.0, h11 = 0.0, h12 = 0.0, h22 = 0.0;
    if (on) {
      // the accepted candidate of the previous step evaluated exactly this sigmoid (same expression, same operands)
      const double f = have_f ? fcur : cm_sigmoid_tab(p0 * val - p1, etab);
      const double r = sy - cnt * f;
      j1 = -val * r; j2 = r;
      const double ff = cnt * f * (1.0 - f);
      h11 = val * val * ff; h12 = -val * ff; h22 = ff;
    }
    j1 = cm_seg_sum<SEG>(j1); j2 = cm_seg_sum<SEG>(j2);
    h11 = cm_seg_sum<SEG>(h11); h12 = cm_seg_sum<SEG>(h12); h22 = cm_seg_sum<SEG>(h22);
    const double d0 = p0 - pr0, d1 = p1 - pr1;
    const double J0 = j1 + (prec[0] * d0 + prec[1] * d1) - 1.0 / (t * p0);      // caviar.py:278-280
    const double J1 = j2 + (prec[2] * d0 + prec[3] * d1) - 1.0 / (t * p1);
    double covn[4];
    inv2x2(h11 + prec[0] + 1.0 / (t * p0 * p0), h12 + prec[1], h12 + prec[2], h22 + prec[3] + 1.0 / (t * p1 * p1), covn);   // :283-289
    const double v0 = -(covn[0] * J0 + covn[1] * J1), v1 = -(covn[2] * J0 + covn[3] * J1);
    const double Jv = J0 * v0 + J1 * v1;
    // candidates b = 0..40 (step 2^-b); b <= 39 needs its Armijo test to pass, b = 40 is taken unconditionally
    int accepted = -1;
    double f0 = 0.0;
    int base = 0;                                 // first candidate of this round (the two halves walk their rounds in step)
    bool first = true;
    while (base < 40) {
      const bool need = !done && accepted < 0;
      if (!__any(need)) break;
      const int b = first ? base + ev - 1 : base + ev;       // first round: segment 0 evaluates phi itself
      const bool is_f0 = first && ev == 0;
      const double st = is_f0 ? 0.0 : ldexp(1.0, -b);
      const double q0 = is_f0 ? p0 : p0 + st * v0, q1 = is_f0 ? p1 : p1 + st * v1;
      double sg;
      const double fv = phi_nll<SEG>(q0, q1, val, n0, n1, smid, nmid, on, g, pr0, pr1, prec, t, &sg, etab, ltab);
      if (first) f0 = __shfl(fv, hb, CM_WAVE);               // (segment 0 of the half: the objective at phi)
      const double rhs = f0 + alpha * st * Jv;
      const bool ok = need && !is_f0 && b <= 39 && !(isnan(fv) || fv > rhs);
      const unsigned mh = (unsigned)(__ballot(ok) >> hb);    // this half's accepting lanes
      const int seg = mh ? (__ffs((int)mh) - 1) / SEG : 0;   // first accepting segment of this round
      const double fsel = __shfl(sg, hb + seg * SEG + g, CM_WAVE);   // its sigmoid for this lane's value
      if (need && mh) { accepted = (first ? base - 1 : base) + seg; fcur = fsel; }
      base += first ? E - 1 : E;
      first = false;
    }
    if (!done) {
      have_f = accepted >= 0;                      // b = 40 is taken without an evaluation: recompute next step
      if (accepted < 0) accepted = 40;
      const double step = ldexp(1.0, -accepted);
      const double q0 = p0 + step * v0, q1 = p1 + step * v1;                        // caviar.py:301
      // A step that leaves phi bit-for-bit unchanged is a fixed point of the loop body: every later step would recompute the same
      // J, H, covariance and (rejected) candidates and leave phi unchanged again, so the remaining steps are skipped -- exactly
      // the reference's result.  Converged neurons are the expensive ones otherwise: their Armijo test fails for all 40 candidates.
      const bool fixed = (q0 == p0) && (q1 == p1);
      p0 = q0; p1 = q1;
#pragma unroll
      for (int i = 0; i < 4; ++i) cov[i] = covn[i];
      if (fixed) done = true;
    }
  }
  if (live) {
    if (hl == 0) {
      phi[2 * nn] = p0; phi[2 * nn + 1] = p1;
      phi_cov[4 * nn] = cov[0]; phi_cov[4 * nn + 1] = cov[1]; phi_cov[4 * nn + 2] = cov[2]; phi_cov[4 * nn + 3] = cov[3];
    }
    out[0] = p0; out[1] = p1; out[2] = cov[0]; out[3] = cov[3];     // mean and variance diagonal, for the MC logit that follows
  }
}

// update_phi of the rows of a wavefront's two halves (n < 0: none): the pair iteration for SEG <= 16, one row after the other through the
// single-row body beyond (more than 16 stimulus values).  EVERY kernel evaluates update_phi through this function, so a row's result is the
// same bits wherever it is computed.  out[]: (phi_0, phi_1, cov_00, cov_11) of the lane's half.
template <int SEG, bool DEAD = false>
__device__ __forceinline__ void phi_pair(int n, int lane, const double* __restrict__ lam, const int* __restrict__ row_ptr,
                                         const uint8_t* __restrict__ grp, const int* __restrict__ rowcnt,
                                         const double* __restrict__ vals, int V,
                                         const double* __restrict__ phi_prior, const double* __restrict__ phi_cov_prior,
                                         double* __restrict__ phi, double* __restrict__ phi_cov, double out[4],
                                         const double* etab, const double* ltab,
                                         const double* __restrict__ dead_phi = nullptr, const double* __restrict__ dead_cov = nullptr) {
  if constexpr (SEG <= 16) {
    phi_body2<SEG, DEAD>(n, lane, lam, row_ptr, grp, rowcnt, vals, V, phi_prior, phi_cov_prior, phi, phi_cov, out, etab, ltab, dead_phi, dead_cov);
  } else {
    out[0] = out[1] = out[2] = out[3] = 0.0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int rown = __builtin_amdgcn_readlane(n, 32 * h);
      if (rown < 0) continue;
      double o[4];
      phi_body<SEG, DEAD>(rown, lane, lam, row_ptr, grp, rowcnt, vals, V, phi_prior, phi_cov_prior, phi, phi_cov, o, etab, ltab, dead_phi, dead_cov);
      if ((lane >> 5) == h) { out[0] = o[0]; out[1] = o[1]; out[2] = o[2]; out[3] = o[3]; }
    }
  }
}


// ------------------------------------------------------------------------------------------------
// Small active sets (the steady state of a fit): the whole block_update_mu (caviar.py:166-172) in ONE workgroup, in LDS.
//   1. stage the active rows' (trial, lam) lists;  2. A_ab = sigma * sum_k lam_ak lam_bk by merge-intersection of the two
//   sorted trial lists (ascending trial order -> deterministic);  3. A^-1 in place by the sweep operator (n pivots, two
//   barriers each; pivots of an SPD matrix are positive, no pivoting needed);  4. mu = A^-1 rhs, beta = diag(A^-1).
// n == 0 only fills the priors.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void solve_small_body(double* sm, const double* __restrict__ lam, const int* __restrict__ row_ptr,
                                                 const int* __restrict__ col, const int* __restrict__ act_idx,
                                                 const double* __restrict__ rs, const double* __restrict__ scal,
                                                 const double* __restrict__ mu_prior, const double* __restrict__ beta_prior,
                                                 int n, int N, int tot_cap, double* __restrict__ mu, double* __restrict__ beta,
                                                 int* __restrict__ flag, int nthr) {
  const int ldA = n + 1;
  double* A = sm;                              // [n][n+1]
  double* rowk = A + (size_t)n * ldA;          // [CM_SMALL_MAX]
  double* colk = rowk + CM_SMALL_MAX;
  double* rhs = colk + CM_SMALL_MAX;
  double* lams = rhs + CM_SMALL_MAX;           // [tot_cap]
  int* cols = (int*)(lams + tot_cap);          // [tot_cap]
  int* rp = cols + tot_cap;                    // [CM_SMALL_MAX + 1]
  int* nidx = rp + CM_SMALL_MAX + 1;           // [CM_SMALL_MAX]
  __shared__ int bad;
  const int tid = threadIdx.x;
  // inactive rows: mu = mu0, beta = beta0^2 exactly (their rows of the precision matrix are diagonal)
#pragma unroll 4
  for (int i = tid; i < N; i += nthr) { mu[i] = mu_prior[i]; const double b0 = beta_prior[i]; beta[i] = b0 * b0; }
  if (tid == 0) bad = 0;
  // ---- 1. stage
  if (tid < n) { const int nn = act_idx[tid]; nidx[tid] = nn; rp[tid + 1] = row_ptr[nn + 1] - row_ptr[nn]; }
  if (tid == 0) rp[0] = 0;
  __syncthreads();
  for (int o = 1; o < n; o <<= 1) {            // inclusive scan of the row lengths
    int v = 0;
    if (tid < n && tid >= o) v = rp[tid + 1 - o];
    __syncthreads();
    if (tid < n && tid >= o) rp[tid + 1] += v;
    __syncthreads();
  }
  const int wave = tid / CM_WAVE, lane = tid % CM_WAVE, nw = nthr / CM_WAVE;
  for (int a = wave; a < n; a += nw) {
    const int g0 = row_ptr[nidx[a]], len = rp[a + 1] - rp[a], o0 = rp[a];
    for (int e = lane; e < len; e += CM_WAVE) { cols[o0 + e] = col[g0 + e]; lams[o0 + e] = lam[g0 + e]; }
  }
  __syncthreads();
  // ---- 2. build (lower triangle pairs, mirrored)
  const double sig = scal[CM_S_SHAPE] / scal[CM_S_RATE];
  const int npairs = n * (n + 1) / 2;
  for (int q = tid; q < npairs; q += nthr) {
    int a = (int)((sqrt(8.0 * (double)q + 1.0) - 1.0) * 0.5);
    while ((a + 1) * (a + 2) / 2 <= q) ++a;
    while (a * (a + 1) / 2 > q) --a;
    const int b = q - a * (a + 1) / 2;
    int ia = rp[a], ib = rp[b];
    const int ea = rp[a + 1], eb = rp[b + 1];
    double acc = 0.0;
    while (ia < ea && ib < eb) {
      const int ka = cols[ia], kb = cols[ib];
      if (ka == kb) { acc = fma(lams[ia], lams[ib], acc); ++ia; ++ib; }
      else if (ka < kb) ++ia; else ++ib;
    }
    double v = sig * acc;
    if (a == b) {
      const int nn = nidx[a];
      const double b0 = beta_prior[nn], prec0 = 1.0 / (b0 * b0);
      v = sig * (rs[(size_t)nn * 4 + 3] + acc) + prec0;
      rhs[a] = sig * rs[(size_t)nn * 4 + 2] + prec0 * mu_prior[nn];
    }
    A[a * ldA + b] = v;
    A[b * ldA + a] = v;
  }
  // ---- 3. sweep-operator inverse
  for (int k = 0; k < n; ++k) {
    __syncthreads();
    const double d = A[k * ldA + k];
    const double inv = 1.0 / d;
    if (tid == 0 && !(d > 0.0)) bad = 1;
    if (tid < n) { rowk[tid] = A[k * ldA + tid] * inv; colk[tid] = A[tid * ldA + k]; }
    __syncthreads();
    for (int idx = tid; idx < n * n; idx += nthr) {
      const int i = idx / n, j = idx % n;
      double v;
      if (i == k) v = (j == k) ? inv : rowk[j];
      else if (j == k) v = -colk[i] * inv;
      else v = fma(-colk[i], rowk[j], A[i * ldA + j]);
      A[i * ldA + j] = v;
    }
  }
  __syncthreads();
  // ---- 4. mu = A^-1 rhs (one wave per row), beta = diag(A^-1)
  for (int a = wave; a < n; a += nw) {
    double sacc = 0.0;
    for (int b = lane; b < n; b += CM_WAVE) sacc = fma(A[a * ldA + b], rhs[b], sacc);
    sacc = cm_wave_sum(sacc);
    if (lane == 0) { mu[nidx[a]] = sacc; beta[nidx[a]] = A[a * ldA + a]; }
  }
  if (tid == 0 && bad) *flag = 1;
}


// The closed-form gate of a row that is identically zero with mu == 0 (cm_sweep.hip: quiescent_gate), from the P mean logits `mce` of its powers:
// est_p = sigmoid(mce_p - sigma beta^2 / 2) on its unmasked trials, rate_p = m_p est_p / trials_p, total = sum_p m_p est_p, isotonic last value
// max_j mean(rate[j:]), the two tests of caviar.py:220.  qrow[q] = unmasked trials << 16 | trials of power q.  Monotone in every mce_p.
__device__ __forceinline__ bool cm_quiescent_gate_core(int P, double arg, double thresh, double min_count, const int* __restrict__ qrow, const double (&mce)[8]) {
  double sr[8], total = 0.0;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    sr[q] = 0.0;
    if (q < P) {
      const double est = cm_sigmoid(mce[q] - 0.5 * arg);
      const int packed = qrow[q], cnt = packed & 0xFFFF;
      const double num = (double)(packed >> 16) * est;
      total += num;
      sr[q] = num / ((double)cnt + 1e-4 * (double)(cnt == 0));                  // caviar.py:183
    }
  }
  double S = 0.0, best = -1.0;
  int c = 0;
#pragma unroll
  for (int j = 7; j >= 0; --j) {
    if (j < P) { S += sr[j]; ++c; const double m = S / (double)c; best = m > best ? m : best; }
  }
  const double iso = P > 0 ? best : 0.0;
  return iso >= thresh && total >= min_count;                                   // caviar.py:220
}

// ------------------------------------------------------------------------------------------------
// Monte-Carlo logit pieces shared by cm_sweep.hip (table kernel) and cm_iter.hip (fused kernel)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double cm_uniform64(uint32_t k0, uint32_t k1, int j, int S) {
  uint32_t x0 = (uint32_t)j, x1 = (uint32_t)(2 * S + j);   // 64-bit draw j: hi = word j, lo = word 2S + j
  cm_threefry2x32(k0, k1, x0, x1);
  const unsigned long long bits = ((unsigned long long)x0 << 32) | (unsigned long long)x1;
  return __longlong_as_double((long long)((bits >> 12) | 0x3FF0000000000000ull)) - 1.0;
}


// _eval_lam_update_monte_carlo (caviar.py:233-236): log(f / (1 - f)), f = sigmoid(x) -- x itself in exact arithmetic, x plus the
// rounding of exp, the two divisions and the logarithm as the reference computes it, +-inf once the sigmoid saturates (quirk a5).
// N x S x P of these per iteration make this the widest transcendental stream of the path, so the final logarithm (~100 fp64
// instructions) is not taken literally where that is exactly recoverable: with e = exp(-x), f = 1 / (1 + e), r = f / (1 - f) as the
// reference forms them, log(r) = x + log1p(r e - 1) up to the last-ulp error of e; r e - 1 is the rounding residue of those three
// operations, a single fma away, and tiny for moderate |x| (|r e - 1| < 2e-7 below |x| = 20), where two terms of log1p are exact to
// double precision.  Beyond, and for non-finite x, the literal form (with its saturation to +-inf) is evaluated.
// saturating arguments (and non-finite ones): the literal form, out of line -- rare, and its logarithm and divisions would otherwise
// set the register budget of every copy of the evaluation loop
__device__ __noinline__ double cm_logit_literal(double x) {
  const double f = cm_sigmoid(x);
  return log(f / (1.0 - f));
}
// Where the rounding residue is evaluated at all.  rho(x) = log(f / (1 - f)) - x is the reference's own rounding noise: exp(-x) enters it as
// an absolute 1e-16 and the rounding of f = 1 / (1 + e) as 2^-53 / (1 - f) = 2^-53 (1 + e^x), so |rho(x)| <= 2^-52 for x <= 0 and
// <= 2^-53 (1 + e^x) above: 3.3e-13 at x = 8.  Which way f rounds depends on the last bit of e, i.e. on the exp implementation (the
// library's, this table's and XLA's differ in a fraction of the arguments), so below that size rho is not reproducible against the
// reference in the first place -- for -700 < x < CM_LOGIT_RESIDUE_FROM the logit is x itself, and the ~40 instructions of the residue
// are only spent where it can matter (large positive x; near the overflow of exp(-x), x < -709.78, the literal form decides).
// -DCM_LOGIT_RESIDUE_FROM=-1e300 evaluates the residue everywhere (tests/test_gpu_parity.py::test_mc_logit_residue_threshold
// compares the two builds' trajectories).
#ifndef CM_LOGIT_RESIDUE_FROM
#define CM_LOGIT_RESIDUE_FROM 8.0
#endif
__device__ __forceinline__ double cm_logit_of_sigmoid(double x, const double* etab) {
  if (x < CM_LOGIT_RESIDUE_FROM && x > -700.0) return x;
  if (fabs(x) < 20.0) {
    const double e = cm_exp_tab(-x, etab);
    const double f = cm_rcp(1.0 + e);                  // 1 + e in [1, 5e8]: normal range
    const double r = f * cm_rcp(1.0 - f);              // 1 - f in [2e-9, 1)
    const double t = fma(r, e, -1.0);
    return x + (t - 0.5 * t * t);
  }
  return cm_logit_literal(x);
}

// ndtri (the standard normal quantile of caviar.py:212; jax.scipy.special.ndtri in the reference, a third-party routine): Wichura's
// algorithm AS 241, PPND16 (Applied Statistics 37, 1988) -- a degree-7 / degree-7 rational function of r = 0.180625 - (p - 1/2)^2 in the
// centre |p - 1/2| <= 0.425 and of sqrt(-log(min(p, 1 - p))) in the tails, relative error about 1e-16 (measured against SciPy's ndtri over
// 2.4e6 arguments incl. tails down to 1e-300: 1.1e-15).  ~30 instructions in the centre, ~110 in a tail, where the device library's
// normcdfinv is a 1290-instruction routine (several hundred executed): every live row draws 2 S of these per iteration, which made the
// quantile most of the early iterations' k_iter_mid.  p = 0 -> -inf, p = 1 -> +inf, outside [0, 1] or NaN -> NaN.
__device__ __forceinline__ double cm_div_fast(double a, double b) {      // a / b for normal-range b: reciprocal (two Newton steps) and one residual correction
  double r = __builtin_amdgcn_rcp(b);
  r = fma(fma(-b, r, 1.0), r, r);
  r = fma(fma(-b, r, 1.0), r, r);
  const double q = a * r;
  return fma(fma(-b, q, a), r, q);
}
__device__ __forceinline__ double cm_ndtri(double p) {
  const double q = p - 0.5;
  if (fabs(q) <= 0.425) {
    const double r = 0.180625 - q * q;
    double n = 2.5090809287301226727e3, d = 5.2264952788528545610e3;
    n = fma(n, r, 3.3430575583588128105e4); d = fma(d, r, 2.8729085735721942674e4);
    n = fma(n, r, 6.7265770927008700853e4); d = fma(d, r, 3.9307895800092710610e4);
    n = fma(n, r, 4.5921953931549871457e4); d = fma(d, r, 2.1213794301586595867e4);
    n = fma(n, r, 1.3731693765509461125e4); d = fma(d, r, 5.3941960214247511077e3);
    n = fma(n, r, 1.9715909503065514427e3); d = fma(d, r, 6.8718700749205790830e2);
    n = fma(n, r, 1.3314166789178437745e2); d = fma(d, r, 4.2313330701600911252e1);
    n = fma(n, r, 3.3871328727963666080e0); d = fma(d, r, 1.0);
    return q * cm_div_fast(n, d);
  }
  double r = q < 0.0 ? p : 1.0 - p;
  if (!(r > 0.0)) return r == 0.0 ? (q < 0.0 ? -INFINITY : INFINITY) : NAN;      // p = 0, p = 1; p outside [0, 1] or NaN
  r = sqrt(-log(r));
  double n, d;
  if (r <= 5.0) {
    r -= 1.6;
    n = 7.74545014278341407640e-4; d = 1.05075007164441684324e-9;
    n = fma(n, r, 2.27238449892691845833e-2); d = fma(d, r, 5.47593808499534494600e-4);
    n = fma(n, r, 2.41780725177450611770e-1); d = fma(d, r, 1.51986665636164571966e-2);
    n = fma(n, r, 1.27045825245236838258e0); d = fma(d, r, 1.48103976427480074590e-1);
    n = fma(n, r, 3.64784832476320460504e0); d = fma(d, r, 6.89767334985100004550e-1);
    n = fma(n, r, 5.76949722146069140550e0); d = fma(d, r, 1.67638483018380384940e0);
    n = fma(n, r, 4.63033784615654529590e0); d = fma(d, r, 2.05319162663775882187e0);
    n = fma(n, r, 1.42343711074968357734e0); d = fma(d, r, 1.0);
  } else {
    r -= 5.0;
    n = 2.01033439929228813265e-7; d = 2.04426310338993978564e-15;
    n = fma(n, r, 2.71155556874348757815e-5); d = fma(d, r, 1.42151175831644588870e-7);
    n = fma(n, r, 1.24266094738807843860e-3); d = fma(d, r, 1.84631831751005468180e-5);
    n = fma(n, r, 2.65321895265761230930e-2); d = fma(d, r, 7.86869131145613259100e-4);
    n = fma(n, r, 2.96560571828504891230e-1); d = fma(d, r, 1.48753612908506148525e-2);
    n = fma(n, r, 1.78482653991729133580e0); d = fma(d, r, 1.36929880922735805310e-1);
    n = fma(n, r, 5.46378491116411436990e0); d = fma(d, r, 5.99832206555887937690e-1);
    n = fma(n, r, 6.65790464350110377720e0); d = fma(d, r, 1.0);
  }
  const double v = cm_div_fast(n, d);
  return q < 0.0 ? -v : v;
}
// rare path of the steady-state kernel, kept out of line so that its registers do not burden the common path
__device__ __noinline__ double cm_quantile_slow(double arg) { return cm_ndtri(arg); }

// one wavefront; mc is the wave's LDS scratch [2*S]; (mean0, mean1, sd0, sd1) = phi[n], diag(phi_cov[n]); wu[2*S] = the (ndtri(u), u)
// pairs of this sweep step (k_ndtri_table: the uniforms are data independent, so the 20 threefry rounds and the quantile of
// every draw are taken off the per-iteration path).
// arg = c + u (1 - c) equals u bit for bit whenever c is below half an ulp of u (c = ndtr(-mean/sdev) underflows once mean/sdev is
// large), and then ndtri(arg) is the tabulated ndtri(u).  FULL = true inlines the quantile function (iteration 0, where phi is the
// prior and c is not small); FULL = false calls it out of line on a miss.  The draws are walked component by component (all
// phi_0 draws, then all phi_1 draws): c is then uniform over the wavefront and the miss path is only entered for a component
// that needs it (phi_1 of rows whose posterior is wide, e.g. all-zero rows: c ~ 1e-7 there).
// HAVE_C: c0 / c1 = ndtr(-mean / sdev) are supplied (per-run constants of an all-zero row, k_phi_dead) instead of evaluated here.
// PRE: the first two rounds of the table row (samples lane, lane + 64; both components) were fetched by the caller (`pre`).
template <bool FULL, bool HAVE_C = false, bool PRE = false>
__device__ __forceinline__ void mc_logit_body(int n, int lane, double mean0, double mean1, double sd0, double sd1,
                                              const double2* __restrict__ wu, const double* __restrict__ vals, int V, int S,
                                              double* mc, double* __restrict__ mcE, const double* etab, double c0_in = 0.0, double c1_in = 0.0,
                                              const double2* pre = nullptr, int mld = CM_MAX_VALS) {
  const double c0 = HAVE_C ? c0_in : normcdf(-mean0 / sd0), c1 = HAVE_C ? c1_in : normcdf(-mean1 / sd1);
#pragma unroll
  for (int comp = 0; comp < 2; ++comp) {
    const double c = comp ? c1 : c0, sd = comp ? sd1 : sd0, mean = comp ? mean1 : mean0;
    const bool tiny = !FULL && c < 0x1p-106;           // (wave-uniform)
    for (int s_ = lane, rnd = 0; s_ < S; s_ += CM_WAVE, ++rnd) {
      const int j = 2 * s_ + comp;                     // u has shape [S, 2] (caviar.py:210)
      const double2 t = (PRE && rnd < 2) ? (rnd == 0 ? pre[comp] : pre[2 + comp]) : wu[j];
      double q;
      if (FULL) {
        q = cm_ndtri(c + t.y * (1.0 - c));             // caviar.py:212
      } else if (tiny) {
        // c is below half an ulp of every u >= 2^-52, the smallest non-zero uniform: arg = c + u (1 - c) is u bit for bit and
        // ndtri(arg) is the tabulated value.  The one exception, u == 0 (arg = c), shows in the table itself: ndtri(0) = -inf.
        q = (t.x == -INFINITY && c != 0.0) ? cm_quantile_slow(c) : t.x;
      } else {
        const double arg = c + t.y * (1.0 - c);
        q = (arg == t.y) ? t.x : cm_quantile_slow(arg);
      }
      mc[j] = q * sd + mean;
    }
  }
#ifdef CM_MC_AFFINE
  // MEASUREMENT VARIANT (tools/build_variant.sh affine -DCM_MC_AFFINE; never the default build): the mean logit of a power from the means of
  // the draws, mean_s(mc0_s v - mc1_s) = v mean(mc0) - mean(mc1), wherever no sample of the row can saturate the sigmoid
  // (max |mc0| v_max + max |mc1| < 36); rows that can take the literal path below.
  {
    double a0 = 0.0, a1 = 0.0, m0 = 0.0, m1 = 0.0;
    for (int s_ = lane; s_ < S; s_ += CM_WAVE) {
      const double x0 = mc[2 * s_], x1 = mc[2 * s_ + 1];       // (this lane's own draws: written above by this lane)
      a0 += x0; a1 += x1; m0 = fmax(m0, fabs(x0)); m1 = fmax(m1, fabs(x1));
    }
    double vmax = 0.0;
    for (int g = 0; g < V; ++g) vmax = fmax(vmax, vals[g]);
    m0 = cm_wave_max(m0); m1 = cm_wave_max(m1);
    if (m0 * vmax + m1 < 36.0) {
      a0 = cm_wave_sum(a0); a1 = cm_wave_sum(a1);
      const double e0 = a0 / (double)S, e1 = a1 / (double)S;
      if (lane < V && vals[lane] > 0.0) mcE[(size_t)n * mld + lane] = e0 * vals[lane] - e1;
      return;
    }
  }
#endif
  // same-wave LDS write -> read: the LDS pipe is in order for one wave; the (free) wave barrier keeps the compiler from moving the
  // reads above the writes of other lanes
  __builtin_amdgcn_wave_barrier();
  // The S x P evaluations.  Lane l sums the samples l, l + 64, ... of a power (in that order), then the lanes are reduced.  With
  // S = 100 the second round of every power fills 36 of the 64 lanes, so the tails of up to four powers are PACKED into shared
  // rounds (item = power * tail + i), their values handed to lane i through the draw scratch (dead by then) and added where the
  // plain loop adds them: the same sums in the same order, in ceil(P tail / 64) tail rounds instead of P.
  const int nfull = S / CM_WAVE, Tn = S - nfull * CM_WAVE;
  int gi[4] = {0, 0, 0, 0}, P = 0;
  for (int g = 0; g < V; ++g)
    if (vals[g] > 0.0) { if (P < 4) gi[P] = g; ++P; }
  const int tail_items = P * Tn;
  if (P >= 2 && P <= 4 && Tn > 0 && tail_items <= 2 * CM_WAVE && tail_items <= 2 * S && (tail_items + CM_WAVE - 1) / CM_WAVE < P) {
    double acc[4] = {0.0, 0.0, 0.0, 0.0}, vk[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      vk[k] = vals[gi[k]];
      if (k < P)
        for (int r = 0; r < nfull; ++r) { const int s = r * CM_WAVE + lane; acc[k] += cm_logit_of_sigmoid(mc[2 * s] * vk[k] - mc[2 * s + 1], etab); }
    }
    double vt[2] = {0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int item = q * CM_WAVE + lane;
      if (item < tail_items) {
        const int k = (item >= Tn) + (item >= 2 * Tn) + (item >= 3 * Tn);
        const int s = nfull * CM_WAVE + (item - k * Tn);
        const double v = k == 0 ? vk[0] : (k == 1 ? vk[1] : (k == 2 ? vk[2] : vk[3]));
        vt[q] = cm_logit_of_sigmoid(mc[2 * s] * v - mc[2 * s + 1], etab);
      }
    }
    __builtin_amdgcn_wave_barrier();                 // every read of the draws is done: the scratch carries the tail values
#pragma unroll
    for (int q = 0; q < 2; ++q) { const int item = q * CM_WAVE + lane; if (item < tail_items) mc[item] = vt[q]; }
    __builtin_amdgcn_wave_barrier();
    double tot = 0.0;                                // lane k: the sum of power k -- ONE division serves the (up to four) means
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < P) {
        double a = acc[k];
        if (lane < Tn) a += mc[k * Tn + lane];
        a = cm_wave_sum(a);
        if (lane == k) tot = a;
      }
    }
    const double mean = tot / (double)S;
    if (lane < P) mcE[(size_t)n * mld + (lane == 0 ? gi[0] : (lane == 1 ? gi[1] : (lane == 2 ? gi[2] : gi[3])))] = mean;
    __builtin_amdgcn_wave_barrier();                 // (a caller's next row rewrites the scratch)
    return;
  }
  for (int g = 0; g < V; ++g) {
    const double v = vals[g];
    if (!(v > 0.0)) continue;
    double acc = 0.0;
    for (int s = lane; s < S; s += CM_WAVE) acc += cm_logit_of_sigmoid(mc[2 * s] * v - mc[2 * s + 1], etab);
    acc = cm_wave_sum(acc);
    if (lane == 0) mcE[(size_t)n * mld + g] = acc / (double)S;
  }
}

// The chain of a sweep: ordered compaction of the rows with mu != 0 (in sweep order) + record offsets + chunk table, by the first NT threads of
// ONE workgroup (NT = 1024: the last block of k_presweep; NT = CM_POST_THREADS: block 0 of k_iter_mid behind its in-LDS solve, which leaves the
// tables for the next sweep so that the pre-sweep packs the records itself -- cm_launch_sweep).  The per-run totals are added by the packing.
template <int NT>
__device__ __forceinline__ void seq_scan_body(const int* __restrict__ order, const double* __restrict__ mu,
                                              const int* __restrict__ row_ptr, int N, int* __restrict__ seq_n,
                                              int* __restrict__ seq_off, int* __restrict__ chunk_first, int* __restrict__ meta,
                                              int* __restrict__ seq_pos, int gran, int pack_max, int (*wsum)[NT / CM_WAVE], int* carry) {
  constexpr int E = 4;                           // consecutive sweep positions per thread: their loads in flight together, one scan round per E * NT rows
  if (threadIdx.x < 2) carry[threadIdx.x] = 0;
  __syncthreads();
  const int lane = threadIdx.x % CM_WAVE, w = threadIdx.x / CM_WAVE;
  for (int m0 = 0; m0 < N; m0 += E * NT) {
    const int mb = m0 + E * (int)threadIdx.x;
    int n[E], f[E], recs[E];
#pragma unroll
    for (int e = 0; e < E; ++e) n[e] = (mb + e < N) ? order[mb + e] : -1;
    int tf = 0, tr = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      f[e] = 0; recs[e] = 0;
      if (n[e] >= 0) {
        f[e] = mu[n[e]] != 0.0;
        const int len = row_ptr[n[e] + 1] - row_ptr[n[e]];
        recs[e] = f[e] ? 1 + (len <= pack_max ? len : 0) : 0;     // (a row beyond pack_max keeps its header only: the single-consumer kernel walks it from the CSR arrays)
      }
      tf += f[e]; tr += recs[e];
    }
    int sf = tf, sr = tr;                      // inclusive wave scans of the threads' totals
    for (int o = 1; o < CM_WAVE; o <<= 1) {
      const int uf = __shfl_up(sf, o, CM_WAVE), ur = __shfl_up(sr, o, CM_WAVE);
      if (lane >= o) { sf += uf; sr += ur; }
    }
    if (lane == CM_WAVE - 1) { wsum[0][w] = sf; wsum[1][w] = sr; }
    __syncthreads();
    int bf = carry[0], br = carry[1];
    for (int i = 0; i < w; ++i) { bf += wsum[0][i]; br += wsum[1][i]; }
    int j = bf + sf - tf, off = br + sr - tr;    // exclusive positions of the thread's first row
#pragma unroll
    for (int e = 0; e < E; ++e) {
      if (f[e]) {
        seq_n[j] = n[e];
        seq_off[j] = off;
        // chunk of this row = floor((end offset - 1) / B); first row of each chunk registers itself
        const int c = (off + recs[e] - 1) / gran;
        const int cprev = off == 0 ? -1 : (off - 1) / gran;       // previous row ends at `off`
        if (c != cprev) chunk_first[c] = j;
      }
      if (n[e] >= 0) seq_pos[n[e]] = f[e] ? j : -1;               // position in the chain, for the per-trial tickets
      j += f[e]; off += recs[e];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int uf = 0, ur = 0;
      for (int i = 0; i < NT / CM_WAVE; ++i) { uf += wsum[0][i]; ur += wsum[1][i]; }
      carry[0] += uf; carry[1] += ur;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int nseq = carry[0], total = carry[1];
    const int nchunks = total > 0 ? (total - 1) / gran + 1 : 0;
    seq_off[nseq] = total;
    chunk_first[nchunks] = nseq;
    meta[0] = nseq; meta[1] = total; meta[2] = nchunks;
  }
}
