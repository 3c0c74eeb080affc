// Spike posterior sweep: replaces update_lam / body_fun / _eval_spike_rates / _eval_lam_update_monte_carlo,
// reference circuitmap/optimise/caviar.py:174-236, and the jax.random calls it makes (caviar.py:196,209-210).
//
// MI355X design.  The reference walks the N neurons strictly in sequence (Gauss-Seidel, caviar.py:229) and each step
// re-sums (N-1) x K products.  Three observations remove almost all of that while keeping the maths identical:
//   1. the only cross-neuron state is the K-vector R = sum_m mu_m lam_m; leave-one-out = R - mu_n lam_n(old)
//       -> O(nnz) per sweep, R lives in LDS;
//   2. lam is supported on the targeted trials only (caviar.py:216), so a step touches the ~K*H/N entries of one
//      CSR row: one wavefront per neuron, lanes over its trials, shuffle reductions for the per-power rates;
//   3. the Monte-Carlo logit depends on the trial only through its stimulus value, and the truncated-normal
//      samples only on phi of the PREVIOUS iteration, so all N x S x P transcendental evaluations run as one
//      embarrassingly parallel kernel before the sweep; a neuron with mu_n == 0 neither reads nor changes R, so
//      those rows are swept in parallel and only the mu_n != 0 rows form the sequential chain (one workgroup).
#include <stdlib.h>

#include <algorithm>

#include "cm_bodies.h"

// ------------------------------------------------------------------------------------------------
// RNG.  Key schedule (host, cm_host_key_schedule): per iteration (3 + N) threefry keys:
//   [0..2]  sort-key subkeys of the permutation rounds (jax _shuffle), [3 + m] the uniform key of sweep step m.
// ------------------------------------------------------------------------------------------------
// 32-bit sort keys: jax threefry_2x32(key, iota(n)) layout (odd n padded with one zero counter)
__device__ __forceinline__ uint32_t cm_bits32(uint32_t k0, uint32_t k1, int i, int n) {
  const int odd = n & 1, half = (n + odd) >> 1;
  uint32_t x0, x1;
  if (i < half) {
    x0 = (uint32_t)i;
    x1 = (half + i < n) ? (uint32_t)(half + i) : 0u;
    cm_threefry2x32(k0, k1, x0, x1);
    return x0;
  }
  x0 = (uint32_t)(i - half);
  x1 = (uint32_t)i;
  cm_threefry2x32(k0, k1, x0, x1);
  return x1;
}

// One round of jax's sort-based shuffle for ALL iterations at once (the streams are data independent): stable sort of
// `in` by fresh 32-bit keys, done as rank-by-counting.  grid = (ceil(N / 64), iters); a block ranks 64 elements, its
// four waves each scanning a quarter of every 1024-key tile (same-address LDS reads broadcast).
__global__ __launch_bounds__(256) void k_perm_batch(const uint32_t* __restrict__ keys_all, int key_stride, int round,
                                                     const int* __restrict__ in_all, int* __restrict__ out_all, int* __restrict__ inv_all,
                                                     int N) {
  __shared__ uint32_t tile[1024];
  __shared__ int part_rank[4][64];
  const int it = blockIdx.y;
  const uint32_t* keys = keys_all + (size_t)it * key_stride;
  const uint32_t k0 = keys[2 * round], k1 = keys[2 * round + 1];
  const int li = threadIdx.x % 64, part = threadIdx.x / 64;
  const int i = blockIdx.x * 64 + li;
  const uint32_t mine = i < N ? cm_bits32(k0, k1, i, N) : 0u;
  int rank = 0;
  for (int j0 = 0; j0 < N; j0 += 1024) {
    __syncthreads();
    for (int q = threadIdx.x; q < 1024; q += 256) tile[q] = (j0 + q < N) ? cm_bits32(k0, k1, j0 + q, N) : 0xFFFFFFFFu;
    __syncthreads();
    const int lim = min(1024, N - j0);
    const int jb = part * 256, je = min(jb + 256, lim);
    for (int j = jb; j < je; ++j) {
      const uint32_t o = tile[j];
      rank += (o < mine) || (o == mine && (j0 + j) < i);
    }
  }
  part_rank[part][li] = rank;
  __syncthreads();
  if (part == 0 && i < N) {
    const int r = part_rank[0][li] + part_rank[1][li] + part_rank[2][li] + part_rank[3][li];
    const int v = in_all ? in_all[(size_t)it * N + i] : i;
    out_all[(size_t)it * N + r] = v;
    if (inv_all) inv_all[(size_t)it * N + v] = r;          // last round: position of neuron v in the sweep
  }
}

__global__ __launch_bounds__(256) void k_invert_orders(const int* __restrict__ orders, int* __restrict__ inv, int N, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) { const long long it = i / N; inv[it * N + orders[i]] = (int)(i - it * N); }
}

// ------------------------------------------------------------------------------------------------
// Monte-Carlo logit  (caviar.py:209-215, 233-236)
//   u ~ U[0,1)^{S x 2}   (threefry 64-bit draws, or host supplied)
//   mc = ndtri(c + u (1 - c)) * sdev + mean,  c = ndtr(-mean/sdev),  mean = phi[n],  sdev = diag(phi_cov[n])  (a VARIANCE, quirk 2)
//   mcE[n][g] = mean_s log(f / (1 - f)),  f = sigmoid(mc_s0 * vals[g] - mc_s1)
// The uniforms do not depend on the data, so w = ndtri(u) is tabulated once per run (k_ndtri_table) for every
// (iteration, sweep step, sample); when c underflows to exactly 0 -- the normal case, mean/sdev >> 1 -- the sample is
// w * sdev + mean bit for bit and the quantile function leaves the per-iteration critical path.
// ------------------------------------------------------------------------------------------------
// grid (N, n_it): block (m, it) tabulates ndtri(u) of sweep step m of iteration it0 + it
// The table rows are stored BY NEURON (row of iteration `it`, neuron orders_all[it][m] <- the draws of sweep step m): a consumer that works on
// neuron n reads its row without first looking up the neuron's sweep position (one dependent memory round trip less in kernels that are
// latency-bound).  orders_all == nullptr (cm_k_rng_streams): rows by sweep step.
__global__ __launch_bounds__(64) void k_ndtri_table(const uint32_t* __restrict__ keys_all, const double* __restrict__ uexp_all, const int* __restrict__ orders_all,
                                                     int it0, int N, int S, double* __restrict__ wtab, double* __restrict__ u_out, double* __restrict__ wstat) {
  const int m = blockIdx.x, it = it0 + blockIdx.y;
  const size_t base = ((size_t)it * N + m) * 2 * S;
  const size_t wrow = (size_t)it * N + (orders_all ? orders_all[(size_t)it * N + m] : m);
  const size_t wbase = wrow * 2 * S;
  uint32_t k0 = 0, k1 = 0;
  if (!uexp_all) { const uint32_t* keys = keys_all + (size_t)it * (N + 3) * 2; k0 = keys[2 * (3 + m)]; k1 = keys[2 * (3 + m) + 1]; }
  // Row statistics for k_mc_dead's screening of all-zero rows: mean and maximum of the phi_0 quantiles (draws j = 2 s), mean and minimum of
  // the phi_1 quantiles (j = 2 s + 1), smallest phi_0 uniform.  A row with a non-finite quantile (u == 0) is marked by max = +inf: never screened.
  double s0 = 0.0, s1 = 0.0, mx0 = -INFINITY, mn1 = INFINITY, mnu0 = INFINITY;
  bool fin = true;
  for (int j = threadIdx.x; j < 2 * S; j += CM_WAVE) {
    const double u = uexp_all ? uexp_all[base + j] : cm_uniform64(k0, k1, j, S);
    if (wtab) {
      const double q = cm_ndtri(u);
      wtab[2 * (wbase + j)] = q; wtab[2 * (wbase + j) + 1] = u;   // (ndtri(u), u) pairs
      fin = fin && isfinite(q);
      if (j & 1) { s1 += q; mn1 = fmin(mn1, q); } else { s0 += q; mx0 = fmax(mx0, q); mnu0 = fmin(mnu0, u); }
    }
    if (u_out) u_out[(size_t)m * 2 * S + j] = u;
  }
  if (wtab && wstat) {
    s0 = cm_wave_sum(s0); s1 = cm_wave_sum(s1);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mx0 = fmax(mx0, __shfl_xor(mx0, o, CM_WAVE)); mn1 = fmin(mn1, __shfl_xor(mn1, o, CM_WAVE)); mnu0 = fmin(mnu0, __shfl_xor(mnu0, o, CM_WAVE)); }
    const bool all_fin = __all(fin);
    if (threadIdx.x == 0) {
      double* w = wstat + wrow * CM_WSTAT;
      w[0] = s0 / (double)S; w[1] = all_fin ? mx0 : INFINITY; w[2] = s1 / (double)S; w[3] = mn1; w[4] = mnu0; w[5] = 0.0;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// one sweep step for neuron n (caviar.py:203-225)
// ------------------------------------------------------------------------------------------------
struct SweepArgs {
  const int* qcnt; const double* rs; uint8_t* touch; int quiescent_ok;   // closed-form gate of all-zero rows (quiescent_gate)
  int *lst_stamp, *lst_r, *lst_cnt; int lst_it;                          // listed trials (cm_pdev::tstamp), this iteration's parity; lst_it < 0: no list
  int mld;                                                                // row stride of mcE
  const int* row_ptr; const int* col; const uint8_t* grp; const int* rowcnt;
  const double* y; const uint8_t* mask; const double* ye; const uint8_t* me; const double* mu; const double* beta; const double* mcE; const double* scal;
  double* lam;
  int V, gated;
  double msrmp, min_count;
};

__device__ __forceinline__ SweepArgs sweep_args(const cm_pdev& p, int gated, double msrmp, double min_count) {
  SweepArgs a;
  a.row_ptr = p.row_ptr; a.col = p.col; a.grp = p.grp; a.rowcnt = p.rowcnt;
  a.y = p.y; a.mask = p.mask; a.ye = p.ye; a.me = p.me; a.mu = p.mu; a.beta = p.beta; a.mcE = p.mcE; a.scal = p.scal;
  a.lam = p.lam; a.V = p.V; a.gated = gated; a.msrmp = msrmp; a.min_count = min_count;
  a.qcnt = p.qcnt; a.rs = p.rs; a.touch = p.touch; a.mld = p.mce_ld;
  a.lst_stamp = nullptr; a.lst_r = nullptr; a.lst_cnt = nullptr; a.lst_it = -1;
  a.quiescent_ok = gated && p.V - 1 <= 8 && p.V >= 1 && p.max_row_len < 65536 && p.vals[0] == 0.0;   // (a stimulus matrix without zeros keeps value-0 entries in the rows: the plain path)
  return a;
}

// The gate of a row that is identically zero and has mu == 0 (95 % of the rows in the steady state), WITHOUT touching its entries.  Such
// a row's estimate takes one value per power, est_p = sigmoid(mcE[n][p] - sigma beta^2 / 2) on its unmasked trials (caviar.py:205-216 with
// mu = 0), so the per-power sums of caviar.py:179-184 are counts times values: rate_p = m_p est_p / trials_p (m_p unmasked trials of
// power p: a per-run constant, k_mask_counts), total = sum_p m_p est_p; then the isotonic last value max_j mean(rate[j:]) and the two
// tests of :220.  A row that fails stays zero -- nothing to write, nothing for the statistics to re-read; a row that passes is
// written by the plain path with the gate already decided.  (Sums of equal terms taken as count x term: the decision can differ from
// the entry-by-entry sum only in an exact tie of a threshold; single fits and batches take the same path.)  One thread evaluates it.
__device__ __forceinline__ bool quiescent_gate(const SweepArgs& a, int n, double sig, double thresh) {
  const int P = a.V - 1;
  const double be = a.beta[n];
  const double arg = sig * (be * be);
  double mce[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) mce[q] = q < P ? a.mcE[(size_t)n * a.mld + q + 1] : 0.0;
  return cm_quiescent_gate_core(P, arg, thresh, a.min_count, a.qcnt + (size_t)n * 8, mce);
}

// pava.py:9-88 with gamma = 1, last element only; literal pool walk (strict '>' merge test, pava.py:42)
__device__ __forceinline__ double cm_pava_last_dev(const double* y, int n, double* v, double* w) {
  int i = 0;
  v[0] = y[0]; w[0] = 1.0;
  for (int t = 1; t < n; ++t) {
    ++i;
    v[i] = y[t]; w[i] = 1.0;
    while (i > 0 && (v[i - 1] / w[i - 1]) > (v[i] / w[i])) {
      --i;
      v[i] += v[i + 1];
      w[i] += w[i + 1];
    }
  }
  return v[i] / w[i];
}

// Gate of a row held one entry per lane (<= 64 targeted trials): per-power mean rates by wave reductions, isotonic
// tail, minimum spike count (caviar.py:219-221).  For P <= 8 powers everything stays in registers and the isotonic
// tail uses its closed form max_j mean(srates[j:]) (== the pool walk's last value; tests/test_oracle_golden.py).
// PS is the compile-time number of powers (straight-line code: the P + 1 reductions interleave); PS = 0 -> runtime P.
template <int PS>
__device__ __forceinline__ double row_gate_small(bool valid, int g, double est, int Prt, double thresh, double min_count) {
  const int P = PS > 0 ? PS : Prt;
  constexpr int PM = PS > 0 ? PS : 8;
  const double total = cm_wave_sum(est);
  // the P rate divisions run in P different lanes at once, and so do the P suffix-mean divisions
  const int lane = (int)(threadIdx.x % CM_WAVE);
  double num = 0.0, den = 1.0;
#pragma unroll
  for (int p = 0; p < PM; ++p) {
    if (p < P) {
      const bool in = valid && g == p + 1;
      const double s = cm_wave_sum(in ? est : 0.0);
      const int cnt = __popcll(__ballot(in));
      if (lane == p) { num = s; den = (double)cnt + 1e-4 * (double)(cnt == 0); }   // caviar.py:183
    }
  }
  const double sr_l = num / den;                 // lane p: mean rate at power p
  double S = 0.0, numj = 0.0, denj = 1.0;
  int c = 0;
#pragma unroll
  for (int j = PM - 1; j >= 0; --j) {
    if (j < P) {
      S += cm_readlane_f64(sr_l, j); ++c;
      if (lane == j) { numj = S; denj = (double)c; }
    }
  }
  const double m_l = numj / denj;                // lane j: mean(srates[j:])
  double best = -1.0;
#pragma unroll
  for (int j = 0; j < PM; ++j) {
    if (j < P) { const double m = cm_readlane_f64(m_l, j); best = m > best ? m : best; }
  }
  const double iso = P > 0 ? best : 0.0;
  return (iso >= thresh && total >= min_count) ? 1.0 : 0.0;                // caviar.py:220
}

// LEAN: the caller guarantees P <= 8 (no literal pool walk in the instantiation: its registers would set the wavefronts per SIMD)
template <bool LEAN = false>
__device__ __forceinline__ double row_gate(bool valid, int g, double est, int P, double thresh, double min_count, double* scr) {
  switch (P) {
    case 1: return row_gate_small<1>(valid, g, est, P, thresh, min_count);
    case 2: return row_gate_small<2>(valid, g, est, P, thresh, min_count);
    case 3: return row_gate_small<3>(valid, g, est, P, thresh, min_count);
    case 4: return row_gate_small<4>(valid, g, est, P, thresh, min_count);
    default: break;
  }
  if (LEAN || P <= 8) return row_gate_small<0>(valid, g, est, P, thresh, min_count);
  // many powers: rates and pool stacks in the wave's LDS scratch (3 x 64 doubles), literal pool walk
  const double total = cm_wave_sum(est);
  double* sr = scr;
  for (int p = 0; p < P; ++p) {
    const bool in = valid && g == p + 1;
    const double s = cm_wave_sum(in ? est : 0.0);
    const int cnt = __popcll(__ballot(in));
    sr[p] = s / ((double)cnt + 1e-4 * (double)(cnt == 0));     // every lane writes the same value
  }
  const double iso = cm_pava_last_dev(sr, P, scr + CM_MAX_VALS, scr + 2 * CM_MAX_VALS);
  return (iso >= thresh && total >= min_count) ? 1.0 : 0.0;                // caviar.py:220
}

// generic (any row length) step straight from the CSR arrays: two passes over the row in chunks of 64
template <bool USE_R, typename RPtr>
__device__ __forceinline__ void sweep_step_long(const SweepArgs& a, int n, RPtr R, double sig, double thresh, int lane, double* scr) {
  const int beg = a.row_ptr[n], end = a.row_ptr[n + 1];
  const double mu = a.mu[n], be = a.beta[n];
  const double t3 = sig * (mu * mu + be * be);      // caviar.py:206 (beta squared again, quirk 1)
  const int P = a.V - 1;                            // powers = vals[1:]
  double pava = 1.0;
  if (a.gated) {
    double total = 0.0;
    double* srate = scr;
    for (int p0 = 0; p0 < (P > 0 ? P : 1); p0 += 8) {
      double psum[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) psum[j] = 0.0;
      for (int e0 = beg; e0 < end; e0 += CM_WAVE) {
        const int e = e0 + lane;
        double est = 0.0;
        int g = -1;
        if (e < end) {
          const int k = a.col[e];
          g = a.grp[e];
          if (a.mask[k]) {
            const double lo = a.lam[e];
            const double loo = USE_R ? sig * (R[k] - mu * lo) : 0.0;
            const double arg = -2.0 * sig * a.y[k] * mu + 2.0 * mu * loo + t3;          // caviar.py:205-206
            est = cm_sigmoid(a.mcE[(size_t)n * a.mld + g] - 0.5 * arg);            // caviar.py:216
          }
        }
        if (p0 == 0) total += est;
#pragma unroll
        for (int j = 0; j < 8; ++j) psum[j] += (g == p0 + j + 1) ? est : 0.0;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (p0 + j < P) {
          const double s = cm_wave_sum(psum[j]);
          const int cnt = a.rowcnt[(size_t)n * CM_MAX_VALS + p0 + j + 1];
          srate[p0 + j] = s / ((double)cnt + 1e-4 * (double)(cnt == 0));                // caviar.py:183
        }
      }
    }
    total = cm_wave_sum(total);
    const double iso = P > 0 ? cm_pava_last_dev(srate, P, scr + CM_MAX_VALS, scr + 2 * CM_MAX_VALS) : 0.0;
    pava = (iso >= thresh && total >= a.min_count) ? 1.0 : 0.0;                          // caviar.py:220-221
  }
  for (int e = beg + lane; e < end; e += CM_WAVE) {
    const int k = a.col[e];
    double est = 0.0;
    const double lo = a.lam[e];
    if (a.mask[k]) {
      const double loo = USE_R ? sig * (R[k] - mu * lo) : 0.0;
      const double arg = -2.0 * sig * a.y[k] * mu + 2.0 * mu * loo + t3;
      est = cm_sigmoid(a.mcE[(size_t)n * a.mld + a.grp[e]] - 0.5 * arg);
    }
    const double nl = est * pava;                                                         // caviar.py:224
    a.lam[e] = nl;
    if (USE_R) R[k] += mu * (nl - lo);
  }
}

// ---- two rows per wavefront (one per 32-lane half): rows with <= 32 targeted trials, <= 8 powers.
// row_gate_small for two rows at once: the same operations per row, in the same order (the per-power divisions in lanes 0..P-1 of
// the row's half, the suffix means likewise), so each row's gate is bit for bit that of row_gate_small
template <int PS>
__device__ __forceinline__ double row_gate_half(bool valid, int g, double est, int Prt, double thresh, double min_count) {
  const int P = PS > 0 ? PS : Prt;
  constexpr int PM = PS > 0 ? PS : 8;
  const int lane = (int)(threadIdx.x % CM_WAVE), sub = lane & 31, hb = lane & 32;
  const double total = cm_half_sum(est);
  double num = 0.0, den = 1.0;
#pragma unroll
  for (int p = 0; p < PM; ++p) {
    if (p < P) {
      const bool in = valid && g == p + 1;
      const double s_ = cm_half_sum(in ? est : 0.0);
      const int cnt = __popc((unsigned)((__ballot(in) >> hb) & 0xFFFFFFFFull));
      if (sub == p) { num = s_; den = (double)cnt + 1e-4 * (double)(cnt == 0); }   // caviar.py:183
    }
  }
  const double sr_l = num / den;                 // lane hb + p: mean rate of this half's row at power p
  double S = 0.0, numj = 0.0, denj = 1.0;
  int c = 0;
#pragma unroll
  for (int j = PM - 1; j >= 0; --j) {
    if (j < P) {
      S += __shfl(sr_l, hb + j, CM_WAVE); ++c;
      if (sub == j) { numj = S; denj = (double)c; }
    }
  }
  const double m_l = numj / denj;                // lane hb + j: mean(srates[j:])
  double best = -1.0;
#pragma unroll
  for (int j = 0; j < PM; ++j) {
    if (j < P) { const double m = __shfl(m_l, hb + j, CM_WAVE); best = m > best ? m : best; }
  }
  const double iso = P > 0 ? best : 0.0;
  return (iso >= thresh && total >= min_count) ? 1.0 : 0.0;                // caviar.py:220
}

// rows with mu == 0: arg = sigma * beta^2 does not involve R and mu * lam contributes nothing to R, so these rows are
// independent of the chain and of each other -> one wave each over the whole chip
// LEAN: the caller guarantees rows of at most 64 trials and at most 8 powers
template <bool LEAN = false>
__device__ __forceinline__ void sweep_indep_row(const SweepArgs& a, int n, int lane, double* scr) {
  if (a.mu[n] != 0.0) return;
  const double sig = a.scal[CM_S_SHAPE] / a.scal[CM_S_RATE];
  const double thresh = a.msrmp + a.scal[CM_S_SPONT_RATE];
  bool decided = false;                              // the row is all zero and its gate was taken in closed form (and passed)
  if (a.quiescent_ok && a.rs[(size_t)n * 4 + 3] < 0.0) {
    if (!quiescent_gate(a, n, sig, thresh)) return;  // stays zero: nothing to write
    decided = true;
    if (lane == 0) a.touch[n] = 1;
    if (a.lst_it >= 0) {                             // the column statistics after the sweep must visit this row's trials: list those not listed yet
      const int lo = 2 * a.lst_it;
      for (int e = a.row_ptr[n] + lane; e < a.row_ptr[n + 1]; e += CM_WAVE) {
        const int k = a.col[e], old = a.lst_stamp[k];
        if (old < lo && atomicCAS(&a.lst_stamp[k], old, lo + 1) == old) a.lst_r[atomicAdd(a.lst_cnt, 1)] = k;
      }
    }
  }
  const int beg = a.row_ptr[n], len = a.row_ptr[n + 1] - beg;
  if (!LEAN && len > CM_WAVE) {
    if (decided) {                                   // revived long row: the estimates, ungated
      SweepArgs b = a; b.gated = 0;
      sweep_step_long<false>(b, n, (double*)nullptr, sig, thresh, lane, scr);
    } else sweep_step_long<false>(a, n, (double*)nullptr, sig, thresh, lane, scr);
    return;
  }
  const bool valid = lane < len;
  const double be = a.beta[n];
  const double arg = sig * (be * be);                // mu == 0: caviar.py:205-206 reduces to sigma * beta^2
  int g = -1;
  double est = 0.0;
  if (valid) {
    const int e = beg + lane;
    g = a.grp[e];
    if (a.me[e]) est = cm_sigmoid(a.mcE[(size_t)n * a.mld + g] - 0.5 * arg);     // me[e] = mask[col[e]]
  }
  double pava = 1.0;
  if (a.gated && !decided) pava = row_gate<LEAN>(valid, g, est, a.V - 1, thresh, a.min_count, scr);
  if (valid) a.lam[beg + lane] = est * pava;
}

// Rows n0 and n0 + 1 in one wavefront (batches: half the wavefronts of the one-row-per-wave launch, whose duration is set by their
// number).  Falls back to one row after the other when a row is longer than 32 trials or there are more than 8 powers.
template <bool LEAN>
__device__ __forceinline__ void sweep_indep_pair(const SweepArgs& a, int n0, int N, int lane, double* scr) {
  const int sub = lane & 31, n = n0 + (lane >> 5);
  bool row_on = n < N && a.mu[n < N ? n : 0] == 0.0;
  const double sig = a.scal[CM_S_SHAPE] / a.scal[CM_S_RATE];
  const double thresh = a.msrmp + a.scal[CM_S_SPONT_RATE];
  // all-zero rows: the gate in closed form, per half (every lane of a half evaluates its row's); a row that fails is done
  bool decided = false;
  if (row_on && a.quiescent_ok && a.rs[(size_t)n * 4 + 3] < 0.0) {
    if (!quiescent_gate(a, n, sig, thresh)) row_on = false;
    else { decided = true; if (sub == 0) a.touch[n] = 1; }
  }
  if (!__any(row_on)) return;
  const int beg = row_on ? a.row_ptr[n] : 0, len = row_on ? a.row_ptr[n + 1] - beg : 0;
  const int P = a.V - 1;
  if (__any(len > 32) || (!LEAN && P > 8)) {         // (sweep_indep_row takes the same decisions again, row by row)
    sweep_indep_row<LEAN>(a, n0, lane, scr);
    if (n0 + 1 < N) sweep_indep_row<LEAN>(a, n0 + 1, lane, scr);
    return;
  }
  const bool valid = sub < len;
  int g = -1;
  double est = 0.0;
  if (valid) {
    const double be = a.beta[n];
    const double arg = sig * (be * be);              // mu == 0: caviar.py:205-206 reduces to sigma * beta^2
    const int e = beg + sub;
    g = a.grp[e];
    if (a.me[e]) est = cm_sigmoid(a.mcE[(size_t)n * a.mld + g] - 0.5 * arg);     // me[e] = mask[col[e]]
  }
  double pava = 1.0;
  if (a.gated) {
    switch (P) {
      case 1: pava = row_gate_half<1>(valid, g, est, P, thresh, a.min_count); break;
      case 2: pava = row_gate_half<2>(valid, g, est, P, thresh, a.min_count); break;
      case 3: pava = row_gate_half<3>(valid, g, est, P, thresh, a.min_count); break;
      case 4: pava = row_gate_half<4>(valid, g, est, P, thresh, a.min_count); break;
      default: pava = row_gate_half<0>(valid, g, est, P, thresh, a.min_count); break;
    }
  }
  if (decided) pava = 1.0;                           // (a revived row: its gate was the closed form)
  if (valid) a.lam[beg + sub] = est * pava;
}

// ------------------------------------------------------------------------------------------------
// The sequential chain (rows with mu != 0, in sweep order).
//   k_seq_scan : ordered compaction of those rows + record offsets + chunk table      (one workgroup)
//   k_seq_pack : per row, one 32-byte header + one 32-byte record per targeted trial holding everything the step
//                needs (trial, power code, mask, CSR slot, lam_old, MC logit, -2 sigma y mu)   (whole chip)
//   k_sweep_seq: one workgroup.  R lives in LDS; waves 1..15 stream the packed records of the NEXT chunk into an LDS
//                double buffer while wave 0 walks the rows of the current chunk (software-pipelined LDS reads,
//                DPP reductions), so no global-memory latency sits on the chain.
// ------------------------------------------------------------------------------------------------
struct __align__(16) SeqRec {
  uint32_t a; int32_t b; double x, y, z;
  // header: a = neuron, b = row length, x = mu, y = sigma (mu^2 + beta^2), z unused
  // entry : a = trial | power code << 24 | mask << 31, b = CSR slot, x = lam_old, y = MC logit, z = -2 sigma y_k mu
};
#ifndef CM_SEQ_CAP
#define CM_SEQ_CAP 1024
#endif
//                       // records per LDS buffer
#define CM_SEQ_B (CM_SEQ_CAP - CM_WAVE - 1)   // chunking granule: a chunk spans < B + 65 <= CAP records
#define CM_SEQ_LDS_CHUNKS 512

// `turns` != 0: also store in bits 24..31 of the slot word how many EARLIER chain rows touch the entry's trial (its ticket
// in the multi-wave chain kernel); requires nnz < 2^24 and at most 255 targeted neurons per trial.
// Wavefront `w` of `nw` takes the chain rows w, w + nw, ...  (k_seq_pack: the whole grid; k_presweep: the blocks of its fourth role.)
__device__ __forceinline__ void seq_pack_body(const cm_pdev& p, const SweepArgs& a, int turns, int pack_max, int w, int nw, int lane) {
  const int* __restrict__ seq_n = p.seq_n;
  const int* __restrict__ seq_off = p.seq_off;
  const int* __restrict__ seq_pos = p.seq_pos;
  const int* __restrict__ col_ptr = p.col_ptr;
  const int* __restrict__ csc_row = p.csc_row;
  SeqRec* __restrict__ packed = (SeqRec*)p.packed;
  const int nseq = p.seq_meta[0];
  if (w == 0 && lane == 0) { p.seq_meta[4] += nseq; p.seq_meta[5] += p.seq_meta[1]; }   // per-run totals (reported in cm_timings): once per sweep
  const double sig = a.scal[CM_S_SHAPE] / a.scal[CM_S_RATE];
  // The grid is sized by the host for twice the largest active set of the batch (the chain rows, mu != 0, are normally among the rows the last
  // solve covered: a wavefront then takes one row); whatever lies beyond it -- a prior mean that is not zero makes every row a chain row -- is
  // reached by the stride.  A grid of N / 4 workgroups per circuit, of which 95 % found nothing to do, was most of this kernel's time in a batch.
  for (int j = w; j < nseq; j += nw) {
    const int n = seq_n[j], off = seq_off[j];
    const int beg = a.row_ptr[n], len = a.row_ptr[n + 1] - beg;
    const double mu = a.mu[n], be = a.beta[n];
    if (lane == 0) {
      SeqRec h;
      h.a = (uint32_t)n; h.b = len; h.x = mu; h.y = sig * (mu * mu + be * be); h.z = 0.0;   // caviar.py:206
      packed[off] = h;
    }
    if (len > pack_max) continue;                     // (header only: see seq_scan_body)
    for (int l = lane; l < len; l += CM_WAVE) {       // a row of more than 64 trials: ceil(len / 64) segments of records, in trial order
      const int e = beg + l;
      const int k = a.col[e], g = a.grp[e];
      SeqRec r;
      r.a = (uint32_t)k | ((uint32_t)g << 24) | ((uint32_t)(a.me[e] != 0) << 31);
      int turn = 0;
      if (turns & 4) turn = p.turn8[e];                // (the pre-sweep's column role ranked the whole column: cm_col_rank16x2)
      else if (turns)     // (gathering four neurons at a time was measured slower than this plain loop: 28.3 vs 26.8 us at 128 circuits, 6.6 vs 5.4 us for one)
        for (int ce = col_ptr[k]; ce < col_ptr[k + 1]; ++ce) { const int q = seq_pos[csc_row[ce]]; turn += (q >= 0 && q < j); }
      if ((turns & 3) == 2 && j == 1 && l == 0) turn = (turn + 1) & 0xFF;   // fault injection (CAVIAR_HIP_TEST_BAD_TICKET): a turn that never comes
      r.b = e | (turn << 24);
      r.x = a.lam[e];
      r.y = a.mcE[(size_t)n * a.mld + g];
      r.z = -2.0 * sig * a.ye[e] * mu;                                                      // caviar.py:205, first term (ye[e] = y[k])
      packed[off + 1 + l] = r;
    }
  }
}
// One launch, three independent roles (1024-thread blocks):
//   blocks [0, nbI)        rows with mu == 0, one wave per row (they neither read nor change R)
//   blocks [nbI, nbI+nbC)  R_k = sum_n mu_n lam_nk per trial (= lam.T @ mu), 16 lanes per trial.  Rows handled by the
//                          first role have mu == 0, so whichever of their old/new lam is read contributes exactly 0.
//   last block             ordered compaction of the mu != 0 rows + record offsets + chunk table of the chain
// LEAN (batches whose rows hold at most 64 trials and whose stimuli take at most 8 powers -- the host checks): the instantiation has
// neither the long-row walk nor the literal pool walk, so that its register count allows two workgroups per CU; the launch is
// latency-bound and as long as its wavefronts take turns.
// `quiescent` (the host sets it once fewer than half of the rows are active and the gate is on): the first role screens ONE ROW PER THREAD
// -- rows with mu == 0 that are all zero take their gate in closed form (quiescent_gate) and, failing it, are done without a
// wavefront, a row pointer or an entry of theirs being touched -- and only the rows left over (rows with lam != 0 but mu == 0, revived
// rows: a handful) are walked by the workgroup's wavefronts.  1 workgroup per 1024 rows instead of 32 or 64.
template <bool BATCH, bool LEAN>
__global__ __launch_bounds__(1024, (BATCH && LEAN) ? 8 : 4) void k_presweep(cm_parg<BATCH> pa, int it, int gated, double msrmp, double min_count, int nbI, int nbC,
                                                    int roles, int gran, int quiescent, int nbL, int list_it, int pack_max, int scan_done, int turns) {
  __shared__ double scr_all[16][3 * CM_MAX_VALS];
  __shared__ int wsum[2][16];
  __shared__ int carry[2];
  const cm_pdev& p = pa.get();
  SweepArgs a = sweep_args(p, gated, msrmp, min_count);
  if (list_it >= 0) { const int q = list_it & 1; a.lst_stamp = p.tstamp + q * p.K; a.lst_r = p.tlist_r + q * p.K; a.lst_cnt = p.tcnt + 2 + q; a.lst_it = list_it; }
  const int N = p.N;
  const int b = blockIdx.x;
  if (b < nbI) {
    if (!(roles & 1)) return;
    if (quiescent) {
      __shared__ int qlist[1024];
      __shared__ int qn;
      if (threadIdx.x == 0) qn = 0;
      __syncthreads();
      const int n = b * 1024 + (int)threadIdx.x;
      if (n < N && a.mu[n] == 0.0) {
        bool need = true;
        if (a.quiescent_ok && a.rs[(size_t)n * 4 + 3] < 0.0)
          need = quiescent_gate(a, n, a.scal[CM_S_SHAPE] / a.scal[CM_S_RATE], a.msrmp + a.scal[CM_S_SPONT_RATE]);
        if (need) qlist[atomicAdd(&qn, 1)] = n;
      }
      __syncthreads();
      const int cnt = qn;
      for (int i = threadIdx.x / CM_WAVE; i < cnt; i += 16) sweep_indep_row<LEAN>(a, qlist[i], threadIdx.x % CM_WAVE, scr_all[threadIdx.x / CM_WAVE]);
      return;
    }
    if (BATCH) {                                     // two rows per wavefront (four were measured slower here, unlike in k_stats)
      const int n0 = 2 * (b * 16 + threadIdx.x / CM_WAVE);
      if (n0 < N) sweep_indep_pair<LEAN>(a, n0, N, threadIdx.x % CM_WAVE, scr_all[threadIdx.x / CM_WAVE]);
    } else {
      const int n = b * 16 + threadIdx.x / CM_WAVE;
      if (n < N) sweep_indep_row<LEAN>(a, n, threadIdx.x % CM_WAVE, scr_all[threadIdx.x / CM_WAVE]);
    }
  } else if (b < nbI + nbC) {
    if (!(roles & 2)) return;
    if (list_it >= 0) {
      // listed trials: only the trials of rows with lam != 0 (k_stats of the last iteration listed them) can carry a row with mu != 0; the
      // blocks behind the walk zero every other trial.  (A trial a revived row adds during this launch has mu = 0 on all its rows: zero too.)
      const int q = list_it & 1, K = p.K, bb = b - nbI;
      if (bb < nbL)
        cm_col_dot16_listed(bb * 64 + (int)(threadIdx.x >> 4), nbL * 64, threadIdx.x & 15, p.tlist_s + q * K, p.tcnt[q], nullptr, 0, K, a.mu, a.lam,
                            p.col_ptr, p.csc_row, p.csc_pos, 0.0, p.R, nullptr);
      else
        cm_col_zero_unlisted((bb - nbL) * 1024 + (int)threadIdx.x, K, p.tstamp + q * K, 2 * list_it, 2 * list_it, p.R, nullptr);
      return;
    }
    if (BATCH) cm_col_dot16x2((b - nbI) * 1024 + threadIdx.x, p.K, a.mu, a.lam, p.col_ptr, p.csc_row, p.csc_pos, 0.0, p.R, nullptr);
    else cm_col_dot16((b - nbI) * 1024 + threadIdx.x, p.K, a.mu, a.lam, p.col_ptr, p.csc_row, p.csc_pos, 0.0, p.R, nullptr);
    if (turns & 4) {       // ... and the chain's tickets, column by column (k_seq_pack reads them: turns & 4 there too)
      const int* __restrict__ pos = p.inv_all + (size_t)it * N;
      if (BATCH) cm_col_rank16x2((b - nbI) * 1024 + threadIdx.x, p.K, a.mu, pos, p.col_ptr, p.csc_row, p.csc_pos, p.turn8);
      else cm_col_rank16((b - nbI) * 1024 + threadIdx.x, p.K, a.mu, pos, p.col_ptr, p.csc_row, p.csc_pos, p.turn8);
    }
  } else if (b == nbI + nbC) {
    if (!(roles & 4)) return;
    if (BATCH && threadIdx.x == 0) p.alive[0] = 0;     // list of rows with lam != 0 that k_stats rebuilds after the sweep
    if (threadIdx.x == 0) { const int qn = (it + 1) & 1; p.tcnt[qn] = 0; p.tcnt[2 + qn] = 0; }   // the trial lists of the next iteration start empty
    // (scan_done: block 0 of the k_iter_mid that produced this iteration's mu has left the chain's tables already -- every circuit of the set
    // took the in-LDS solve -- and the blocks behind this one pack the records in this launch)
    if (!scan_done) seq_scan_body<1024>(p.orders_all + (size_t)it * N, a.mu, a.row_ptr, N, p.seq_n, p.seq_off, p.chunk_first, p.seq_meta, p.seq_pos, gran, pack_max, wsum, carry);
  } else {
    // the chain's records (k_seq_pack's work as a fourth role: a short chain -- at most small_cap rows per circuit -- is not worth a launch)
    if (!(roles & 4)) return;
    const int nw = (int)(gridDim.x - (nbI + nbC + 1)) * 16;
    seq_pack_body(p, a, turns, pack_max, (b - (nbI + nbC + 1)) * 16 + (int)threadIdx.x / CM_WAVE, nw, threadIdx.x % CM_WAVE);
  }
}

template <bool BATCH>
__global__ __launch_bounds__(256) void k_seq_pack(cm_parg<BATCH> pa, int gated, double msrmp, double min_count, int turns, int pack_max) {
  const cm_pdev& p = pa.get();
  const SweepArgs a = sweep_args(p, gated, msrmp, min_count);
  seq_pack_body(p, a, turns, pack_max, (int)(blockIdx.x * blockDim.x + threadIdx.x) / CM_WAVE, (int)(gridDim.x * blockDim.x) / CM_WAVE, threadIdx.x % CM_WAVE);
}

// A chain row of MORE than 64 trials from its packed records (LDS; `rec` = the row's first entry record, `len` entries in trial order), as
// ceil(len / 64) segments of one entry per lane.  Shared by the multi-wave and the single-consumer chain kernels: the same operations in the
// same order, so the two give bitwise the same row.  The structure is sweep_step_long's (the CSR walk): pass 1 evaluates every entry's
// estimate -- kept in the record's `y` field, which is dead once the estimate exists; a chunk's records belong to the wavefront that walks
// them -- and the gate's sums, per lane over the segments in trial order, then across the lanes; literal pool walk; pass 2 writes lam and R.
// The caller has made sure every trial of the row shows its turn (multi-wave kernel) and publishes the tickets afterwards.
template <typename RT>
__device__ __forceinline__ void sweep_row_packed_long(const SweepArgs& a, SeqRec* rec, int len, double mu, double t3, RT* Rl, double sig, double thresh,
                                                      int P, int lane, double* scr) {
  double total = 0.0;
  double psum[8];
  int pcnt[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { psum[j] = 0.0; pcnt[j] = 0; }
  for (int i0 = 0; i0 < len; i0 += CM_WAVE) {
    const int i = i0 + lane;
    double est = 0.0;
    int g = -1;
    if (i < len) {
      const SeqRec r = rec[i];
      g = (int)((r.a >> 24) & 0x7Fu);
      if ((r.a >> 31) != 0u) {
        const double rk = Rl[r.a & 0xFFFFFFu];
        const double loo = sig * (rk - mu * r.x);
        const double arg = r.z + 2.0 * mu * loo + t3;                                // caviar.py:205-206
        est = cm_sigmoid(r.y - 0.5 * arg);                                           // caviar.py:216
      }
      rec[i].y = est;
    }
    total += est;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const bool in = g == j + 1; psum[j] += in ? est : 0.0; pcnt[j] += __popcll(__ballot(in)); }
  }
  double pava = 1.0;
  if (a.gated && P <= 8) {
    // at most eight powers: rates and the isotonic tail in registers (closed form max_j mean(srates[j:]), the P divisions in P lanes at once),
    // as row_gate_small does for a row of one segment
    total = cm_wave_sum(total);
    double num = 0.0, den = 1.0;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < P) { const double s = cm_wave_sum(psum[j]); if (lane == j) { num = s; den = (double)pcnt[j] + 1e-4 * (double)(pcnt[j] == 0); } }   // caviar.py:183
    const double sr_l = num / den;
    double S = 0.0, numj = 0.0, denj = 1.0;
    int c = 0;
#pragma unroll
    for (int j = 7; j >= 0; --j)
      if (j < P) { S += cm_readlane_f64(sr_l, j); ++c; if (lane == j) { numj = S; denj = (double)c; } }
    const double m_l = numj / denj;
    double best = -1.0;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < P) { const double m = cm_readlane_f64(m_l, j); best = m > best ? m : best; }
    const double iso = P > 0 ? best : 0.0;
    pava = (iso >= thresh && total >= a.min_count) ? 1.0 : 0.0;                       // caviar.py:220-221
  } else if (a.gated) {
    double* srate = scr;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < P) { const double s = cm_wave_sum(psum[j]); srate[j] = s / ((double)pcnt[j] + 1e-4 * (double)(pcnt[j] == 0)); }   // caviar.py:183
    for (int p0 = 8; p0 < P; p0 += 8) {                 // more than eight powers: further passes over the stored estimates
#pragma unroll
      for (int j = 0; j < 8; ++j) { psum[j] = 0.0; pcnt[j] = 0; }
      __builtin_amdgcn_wave_barrier();
      for (int i0 = 0; i0 < len; i0 += CM_WAVE) {
        const int i = i0 + lane;
        double est = 0.0;
        int g = -1;
        if (i < len) { const SeqRec r = rec[i]; g = (int)((r.a >> 24) & 0x7Fu); est = r.y; }
#pragma unroll
        for (int j = 0; j < 8; ++j) { const bool in = g == p0 + j + 1; psum[j] += in ? est : 0.0; pcnt[j] += __popcll(__ballot(in)); }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (p0 + j < P) { const double s = cm_wave_sum(psum[j]); srate[p0 + j] = s / ((double)pcnt[j] + 1e-4 * (double)(pcnt[j] == 0)); }
    }
    total = cm_wave_sum(total);
    __builtin_amdgcn_wave_barrier();
    const double iso = P > 0 ? cm_pava_last_dev(srate, P, scr + CM_MAX_VALS, scr + 2 * CM_MAX_VALS) : 0.0;
    pava = (iso >= thresh && total >= a.min_count) ? 1.0 : 0.0;                       // caviar.py:220-221
  }
  __builtin_amdgcn_wave_barrier();                      // (the estimates written above are read back by the lanes that wrote them)
  for (int i = lane; i < len; i += CM_WAVE) {
    const SeqRec r = rec[i];
    const int k = (int)(r.a & 0xFFFFFFu);
    const double nl = r.y * pava;                                                     // caviar.py:224
    a.lam[r.b & 0xFFFFFF] = nl;
    Rl[k] = Rl[k] + mu * (nl - r.x);
  }
}

template <bool BATCH, bool R_IN_LDS>
__global__ __launch_bounds__(1024) void k_sweep_seq(cm_parg<BATCH> pa, int gated, double msrmp, double min_count, int packed_cap, int nchunks_cap, int pack_max) {
  const cm_pdev& p = pa.get();
  const SweepArgs a = sweep_args(p, gated, msrmp, min_count);
  const SeqRec* __restrict__ packed = (const SeqRec*)p.packed;
  const int* __restrict__ seq_off = p.seq_off;
  const int* __restrict__ chunk_first = p.chunk_first;
  const int* __restrict__ meta = p.seq_meta;
  const int K = p.K;
  double* __restrict__ Rg = p.R;
  __shared__ SeqRec buf[2][CM_SEQ_CAP];
  __shared__ int cfo[CM_SEQ_LDS_CHUNKS][2];      // per chunk: first row, first record offset
  __shared__ double scr[3 * CM_MAX_VALS];
  extern __shared__ double Rl[];
#define R_AT(k) (R_IN_LDS ? Rl[k] : Rg[k])
  // Start-up in as few dependent global round trips as possible: chunk 0 always starts at record 0, so its records are
  // requested speculatively (up to a full buffer, clamped to the allocation) together with meta, the statically addressed
  // chunk-table entries and R; only the record offsets of later chunks need a second round trip.
  {
    const int i = threadIdx.x;                                   // blockDim.x == CM_SEQ_CAP == 1024
    if (i < packed_cap) buf[0][i] = packed[i];
  }
  const int nseq = meta[0], nchunks = meta[2];
  for (int i = threadIdx.x; i < CM_SEQ_LDS_CHUNKS; i += blockDim.x) cfo[i][0] = (i <= nchunks_cap) ? chunk_first[i] : 0;
  if (R_IN_LDS)
    for (int k = threadIdx.x; k < K; k += blockDim.x) Rl[k] = Rg[k];
  if (nseq == 0) return;
  __syncthreads();
  for (int i = threadIdx.x; i <= nchunks && i < CM_SEQ_LDS_CHUNKS; i += blockDim.x) cfo[i][1] = (i == 0) ? 0 : seq_off[cfo[i][0]];
  __syncthreads();
  auto first_row = [&](int c) { return c < CM_SEQ_LDS_CHUNKS ? cfo[c][0] : chunk_first[c]; };
  auto first_off = [&](int c) { return c < CM_SEQ_LDS_CHUNKS ? cfo[c][1] : seq_off[chunk_first[c]]; };
  const int wave = threadIdx.x / CM_WAVE, lane = threadIdx.x % CM_WAVE;
  const double sig = a.scal[CM_S_SHAPE] / a.scal[CM_S_RATE];
  const double thresh = a.msrmp + a.scal[CM_S_SPONT_RATE];
  const int P = a.V - 1;
  int cur = 0;
  for (int c = 0; c < nchunks; ++c) {
    if (wave > 0) {
      if (c + 1 < nchunks) {                                   // producers: stream chunk c + 1 into the other buffer
        const int o0 = first_off(c + 1), o1 = first_off(c + 2);
        for (int i = threadIdx.x - CM_WAVE; i < o1 - o0; i += blockDim.x - CM_WAVE) buf[cur ^ 1][i] = packed[o0 + i];
      }
    } else {                                                   // consumer: the Gauss-Seidel chain
      const int rows = first_row(c + 1) - first_row(c);
      int pos = 0;
      SeqRec h = buf[cur][0];
      SeqRec r = buf[cur][min(1 + lane, CM_SEQ_CAP - 1)];
      for (int j = 0; j < rows; ++j) {
        const int len = h.b;
        const bool lng = len > CM_WAVE, lng_packed = lng && len <= pack_max;
        const int npos = pos + 1 + ((lng && !lng_packed) ? 0 : len);
        // software pipeline: the next row's header and records are requested before this row is computed
        const SeqRec hn = buf[cur][min(npos, CM_SEQ_CAP - 1)];
        const SeqRec rn = buf[cur][min(npos + 1 + lane, CM_SEQ_CAP - 1)];
        if (lng_packed) {
          if (R_IN_LDS) sweep_row_packed_long(a, &buf[cur][pos + 1], len, h.x, h.y, Rl, sig, thresh, P, lane, scr);
          else sweep_row_packed_long(a, &buf[cur][pos + 1], len, h.x, h.y, Rg, sig, thresh, P, lane, scr);
        } else if (lng) {
          if (R_IN_LDS) sweep_step_long<true>(a, (int)h.a, Rl, sig, thresh, lane, scr);
          else sweep_step_long<true>(a, (int)h.a, Rg, sig, thresh, lane, scr);
        } else {
          const bool valid = lane < len;
          const double mu = h.x, t3 = h.y;
          const int k = (int)(r.a & 0xFFFFFFu), g = (int)((r.a >> 24) & 0x7Fu);
          const bool msk = (r.a >> 31) != 0u;
          double est = 0.0, rk = 0.0;
          if (valid) {
            rk = R_AT(k);
            if (msk) {
              const double loo = sig * (rk - mu * r.x);
              const double arg = r.z + 2.0 * mu * loo + t3;                              // caviar.py:205-206
              est = cm_sigmoid(r.y - 0.5 * arg);                                         // caviar.py:216
            }
          }
          double pava = 1.0;
          if (a.gated) pava = row_gate(valid, g, est, P, thresh, a.min_count, scr);      // caviar.py:219-221
          if (valid) {
            const double nl = est * pava;                                                // caviar.py:224
            a.lam[r.b] = nl;
            const double rnew = rk + mu * (nl - r.x);
            if (R_IN_LDS) Rl[k] = rnew; else Rg[k] = rnew;
          }
        }
        h = hn; r = rn; pos = npos;
      }
    }
    __syncthreads();
    cur ^= 1;
  }
#undef R_AT
}

// ------------------------------------------------------------------------------------------------
// Multi-wave chain.  Two chain rows conflict only if they share a trial (through R_k), and with H targets per trial the
// dependency graph of a sweep is sparse: at the benchmark shape its critical path is ~130 of 1000 rows.  CM_PAR_W consumer
// waves take the rows of a chunk round-robin, each in increasing order; a record carries its trial's "turn" (number of earlier
// chain rows touching that trial, from k_seq_pack) and a per-trial LDS counter is the ticket: a row starts when every one of
// its trials shows its turn, and publishes turn + 1 after updating R.  Static in-order assignment makes this deadlock free
// (the smallest unfinished row never waits); results are bit-identical to the one-wave walk because every R_k still sees
// its updates in sweep order.  Requires rows <= 64 trials, <= 255 neurons per trial, nnz < 2^24 (host checks).
// ------------------------------------------------------------------------------------------------
#ifndef CM_PAR_W
#define CM_PAR_W 12   // consumer waves (the other 4 stream records); 8 and 4 measured 1.5 % and 4 % slower
#endif
#ifndef CM_PAR_MIN_ROWS
#define CM_PAR_MIN_ROWS 0    // measured: the multi-wave kernel also wins for the ~30-row steady-state chains
#endif
// CAP = records per LDS buffer, TK = ticket type: (1024, int) for K up to ~5600 trials, (512, uint8_t) up to ~10900.
template <bool BATCH, int CAP, typename TK>
__global__ __launch_bounds__(1024) void k_sweep_par(cm_parg<BATCH> pa, int gated, double msrmp, double min_count, int packed_cap, int nchunks_cap) {
  const cm_pdev& p = pa.get();
  const SweepArgs a = sweep_args(p, gated, msrmp, min_count);
  const SeqRec* __restrict__ packed = (const SeqRec*)p.packed;
  const int* __restrict__ seq_off = p.seq_off;
  const int* __restrict__ chunk_first = p.chunk_first;
  const int* __restrict__ meta = p.seq_meta;
  const int K = p.K;
  const double* __restrict__ Rg = p.R;
  int* __restrict__ err_flag = p.n_act + 2;
  __shared__ SeqRec buf[2][CAP];
  __shared__ int cfo[CM_SEQ_LDS_CHUNKS][2];
  __shared__ int roff[2][CAP + 1];            // record offset of every row of the chunk, relative to the chunk start
  __shared__ double scr[CM_PAR_W][3 * CM_MAX_VALS];
  extern __shared__ double dyn[];
  double* Rl = dyn;                                  // [K]
  volatile TK* cnt = (volatile TK*)(dyn + K);        // [K] tickets
  {
    const int i = threadIdx.x;
    if (i < packed_cap && i < CAP) buf[0][i] = packed[i];
  }
  const int nseq = meta[0], nchunks = meta[2];
  for (int i = threadIdx.x; i < CM_SEQ_LDS_CHUNKS; i += blockDim.x) cfo[i][0] = (i <= nchunks_cap) ? chunk_first[i] : 0;
  for (int k = threadIdx.x; k < K; k += blockDim.x) { Rl[k] = Rg[k]; cnt[k] = 0; }
  if (nseq == 0) return;
  __syncthreads();
  for (int i = threadIdx.x; i <= nchunks && i < CM_SEQ_LDS_CHUNKS; i += blockDim.x) cfo[i][1] = (i == 0) ? 0 : seq_off[cfo[i][0]];
  __syncthreads();
  auto first_row = [&](int c) { return c < CM_SEQ_LDS_CHUNKS ? cfo[c][0] : chunk_first[c]; };
  auto first_off = [&](int c) { return c < CM_SEQ_LDS_CHUNKS ? cfo[c][1] : seq_off[chunk_first[c]]; };
  {                                                  // row offsets of chunk 0
    const int r0 = first_row(0), rows0 = first_row(1) - r0;
    for (int j = threadIdx.x; j <= rows0; j += blockDim.x) roff[0][j] = seq_off[r0 + j];
  }
  __syncthreads();
  const int wave = threadIdx.x / CM_WAVE, lane = threadIdx.x % CM_WAVE;
  const double sig = a.scal[CM_S_SHAPE] / a.scal[CM_S_RATE];
  const double thresh = a.msrmp + a.scal[CM_S_SPONT_RATE];
  const int P = a.V - 1;
  int cur = 0;
  for (int c = 0; c < nchunks; ++c) {
    const int rows = first_row(c + 1) - first_row(c);
    if (wave >= CM_PAR_W) {
      if (c + 1 < nchunks) {                         // producers: records and row offsets of chunk c + 1
        const int o0 = first_off(c + 1), o1 = first_off(c + 2);
        const int nth = blockDim.x - CM_PAR_W * CM_WAVE, t = threadIdx.x - CM_PAR_W * CM_WAVE;
        for (int i = t; i < o1 - o0; i += nth) buf[cur ^ 1][i] = packed[o0 + i];
        const int r1 = first_row(c + 1), rows1 = first_row(c + 2) - r1;
        for (int j = t; j <= rows1; j += nth) roff[cur ^ 1][j] = seq_off[r1 + j] - o0;
      }
    } else {
      int j = wave;
      SeqRec h, r;
      if (j < rows) { const int pos = roff[cur][j]; h = buf[cur][pos]; r = buf[cur][min(pos + 1 + lane, CAP - 1)]; }
      for (; j < rows; j += CM_PAR_W) {
        SeqRec hn = h, rn = r;
        if (j + CM_PAR_W < rows) {                   // this wave's next row: requested before waiting on the tickets
          const int pn = roff[cur][j + CM_PAR_W];
          hn = buf[cur][pn]; rn = buf[cur][min(pn + 1 + lane, CAP - 1)];
        }
        const int len = h.b;
        if (len > CM_WAVE) {
          // a row of more than 64 trials: its records as segments of 64 -- wait for the tickets of all of them, walk the row
          // (sweep_row_packed_long: the function the single-consumer kernel runs on the same records), publish all of them
          SeqRec* rec = &buf[cur][roff[cur][j] + 1];
          unsigned spins = 0;
          while (true) {
            bool ok = true;
            for (int i = lane; i < len; i += CM_WAVE) { const SeqRec q = rec[i]; ok = ok && ((int)cnt[q.a & 0xFFFFFFu] == (int)((unsigned)q.b >> 24)); }
            if (__all(ok)) break;
            if (++spins > (1u << 18)) { if (lane == 0) *err_flag = 1; break; }
            __builtin_amdgcn_s_sleep(1);
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          sweep_row_packed_long(a, rec, len, h.x, h.y, Rl, sig, thresh, P, lane, scr[wave]);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          for (int i = lane; i < len; i += CM_WAVE) { const SeqRec q = rec[i]; cnt[q.a & 0xFFFFFFu] = (TK)(((unsigned)q.b >> 24) + 1u); }
          h = hn; r = rn;
          continue;
        }
        const bool valid = lane < len;
        const double mu = h.x, t3 = h.y;
        const int k = valid ? (int)(r.a & 0xFFFFFFu) : 0, g = (int)((r.a >> 24) & 0x7Fu);
        const bool msk = (r.a >> 31) != 0u;
        const int slot = r.b & 0xFFFFFF, turn = (int)((unsigned)r.b >> 24);
        // ---- wait until every trial of this row shows its turn (bounded: a logic error must not hang the GPU)
        unsigned spins = 0;
        while (true) {
          const bool ok = !valid || ((int)cnt[k] == turn);
          if (__all(ok)) break;
          if (++spins > (1u << 18)) { if (lane == 0) *err_flag = 1; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        double est = 0.0, rk = 0.0;
        if (valid) {
          rk = Rl[k];
          if (msk) {
            const double loo = sig * (rk - mu * r.x);
            const double arg = r.z + 2.0 * mu * loo + t3;                              // caviar.py:205-206
            est = cm_sigmoid(r.y - 0.5 * arg);                                         // caviar.py:216
          }
        }
        double pava = 1.0;
        if (a.gated) pava = row_gate(valid, g, est, P, thresh, a.min_count, scr[wave]);  // caviar.py:219-221
        if (valid) {
          const double nl = est * pava;                                                // caviar.py:224
          a.lam[slot] = nl;
          Rl[k] = rk + mu * (nl - r.x);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (valid) cnt[k] = (TK)(turn + 1);
        h = hn; r = rn;
      }
    }
    __syncthreads();
    cur ^= 1;
  }
}

// Sweep orders of iterations [it0, it0 + n_it) into p->orders_all (caviar.py:196); threefry mode only.
int cm_launch_rng(cm_ctx* ctx, cm_problem* p, int it0, int n_it, hipStream_t st) {
  (void)ctx;
  const int N = p->N;
  const int rounds = cm_perm_rounds(N);           // >= 1 for N >= 2
  const uint32_t* keys = p->keys + (size_t)it0 * (N + 3) * 2;
  int* fin = p->orders_all + (size_t)it0 * N;
  int* tmp = p->orders_tmp + (size_t)it0 * N;
  const int* in = nullptr;                        // nullptr = identity (jnp.arange(N))
  for (int r = 0; r < rounds; ++r) {
    int* out = ((rounds - 1 - r) % 2 == 0) ? fin : tmp;    // ping-pong so the last round lands in orders_all
    int* inv = (r == rounds - 1) ? p->inv_all + (size_t)it0 * N : (int*)nullptr;
    hipLaunchKernelGGL(k_perm_batch, dim3((N + 63) / 64, n_it), dim3(256), 0, st, keys, (N + 3) * 2, r, in, out, inv, N);
    in = out;
  }
  CM_HIP(hipGetLastError());
  return CM_OK;
}

// w = ndtri(u) for every (iteration, sweep step, sample) of the run; explicit streams are tabulated from the uploaded uniforms
int cm_launch_ndtri_table(cm_ctx* ctx, cm_problem* p, int it0, int n_it, int S, int rng_mode, hipStream_t st) {
  (void)ctx;
  if (n_it <= 0) return CM_OK;
  const double* uexp = (rng_mode == CM_RNG_EXPLICIT) ? p->uexp : nullptr;
  hipLaunchKernelGGL(k_ndtri_table, dim3(p->N, n_it), dim3(64), 0, st, p->keys, uexp, (const int*)p->orders_all, it0, p->N, S, p->wtab, (double*)nullptr, p->wstat);
  if (rng_mode == CM_RNG_EXPLICIT) {
    const long long total = (long long)n_it * p->N;
    hipLaunchKernelGGL(k_invert_orders, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p->orders_all + (size_t)it0 * p->N,
                       p->inv_all + (size_t)it0 * p->N, p->N, total);
  }
  CM_HIP(hipGetLastError());
  return CM_OK;
}

// per-phase test hook: the device's standard normal quantile (cm_ndtri, caviar.py:212) on n arguments
__global__ __launch_bounds__(256) void k_ndtri_hook(const double* __restrict__ p, double* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = cm_ndtri(p[i]);
}
int cm_k_ndtri_impl(cm_ctx* ctx, int n, const double* d_p, double* d_out) {
  hipLaunchKernelGGL(k_ndtri_hook, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_p, d_out, n);
  CM_HIP(hipGetLastError());
  return CM_OK;
}

int cm_k_rng_streams_impl(cm_ctx* ctx, cm_problem* p, int it, int S, double* d_u) {
  hipLaunchKernelGGL(k_ndtri_table, dim3(p->N, 1), dim3(64), 0, ctx->stream, p->keys, (const double*)nullptr, (const int*)nullptr, it, p->N, S, (double*)nullptr, d_u, (double*)nullptr);
  CM_HIP(hipGetLastError());
  return CM_OK;
}

// the sweep of iteration `it` screens its mu == 0 rows one per thread (and the statistics after it skip the rows left untouched): once the
// gate is on and fewer than half of the rows are active.  Any choice is correct; both launches of an iteration must take the same one.
int cm_sweep_quiescent(const cm_ctx* ctx, const cm_set& s, int it, const cm_options* o) {
  (void)ctx;
  return (it > o->delay_spont_est && 2 * s.na_max < s.lead->N) ? 1 : 0;
}

// Chunking of the chain's records for sweep variant `par` (cm_sweep_variant): chain rows of up to pack_max trials travel as packed records (rows
// of more than 64 as several 64-entry segments); a longer row keeps its header only and is walked from the CSR arrays by the single-consumer kernel
// (the multi-wave kernel requires every row packed).  Chunking granule: a chunk spans < gran + 1 + pack_max <= cap records.
void cm_sweep_chunking(const cm_set& s, int par, int* cap_out, int* pack_max_out, int* gran_out) {
  const int cap = (par == 2) ? 512 : CM_SEQ_CAP;
  const int pack_max = std::max(CM_WAVE, std::min(s.max_row_len_max, cm_pack_limit(cap)));
  if (cap_out) *cap_out = cap;
  if (pack_max_out) *pack_max_out = pack_max;
  if (gran_out) *gran_out = cap - pack_max - 1;
}
int cm_par_min_rows() { return CM_PAR_MIN_ROWS; }

int cm_launch_sweep(cm_ctx* ctx, cm_set& s, int it, const cm_options* o, hipEvent_t mid, int table_it, int fused_pack) {
  const cm_problem* p = s.lead;
  const int gated = it > o->delay_spont_est;
  const double msrmp = o->msrmp, min_count = o->minimum_spike_count;
  hipStream_t st = ctx->stream;
  const int N = p->N;
  const int nbI = s.B > 1 ? (N + 31) / 32 : (N + 15) / 16;          // 16 rows per block (batches: 32, two rows per wavefront)
  const int nbC = ((s.B > 1 ? (p->K + 1) / 2 : p->K) * 16 + 1023) / 1024;   // 16 lanes per trial column (batches: per two columns)
  // multi-wave chain when the ticket encoding fits (see k_sweep_par): variant 1 = 1024-record buffers + 32-bit tickets,
  // variant 2 = 512-record buffers + 8-bit tickets (larger K); 0 = single-consumer kernel.  Whichever kernel walks the chain,
  // every R_k sees its updates in sweep order: the variants give bit-identical results.
  const int par = cm_sweep_variant(s, s.B > 1 ? 1 : ctx->mb_n);
  int cap, pack_max, gran;
  cm_sweep_chunking(s, par, &cap, &pack_max, &gran);
  const size_t lds_par = (par == 2) ? (size_t)p->K * 9 + 16 : (size_t)p->K * 12;
  static const bool split = getenv("CAVIAR_HIP_SPLIT_PRESWEEP") != nullptr;     // diagnostic: one launch per role (profiling only)
  int v_max = 0;
  for (const cm_problem* q : s.ps) v_max = std::max(v_max, q->V);
  const bool lean = s.max_row_len_max <= CM_WAVE && v_max - 1 <= 8;     // every circuit of the set: rows of at most 64 trials, at most 8 powers
  // one row per THREAD in the first role once the gate is on and fewer than half of the rows are active (any choice is correct)
  const int quiescent = table_it < 0 ? cm_sweep_quiescent(ctx, s, it, o) : 0;
  const int nbIq = quiescent ? (N + 1023) / 1024 : nbI;
  // listed trials (cm_pdev::tstamp): k_stats of the last iteration left the trials of the rows with lam != 0 -- the column sums walk those (at most
  // the trials the rows of the active set can touch; a stride loop takes what exceeds the grid) and zero the rest
  const int list_it = (quiescent && s.tlist_it == it && it >= 0) ? it : -1;
  int nbL = 0, nbCk = nbC;
  if (list_it >= 0) {
    const long long per_row = (s.nnz_max + N - 1) / N;
    const long long est = std::min<long long>(p->K, per_row * (long long)s.na_max * 5 / 4 + 64);
    nbL = (int)(((est + 1) / 2 + 63) / 64);                         // two listed trials per 16-lane segment
    nbCk = nbL + (p->K + 1023) / 1024;
  }
  static const bool bad_ticket = getenv("CAVIAR_HIP_TEST_BAD_TICKET") != nullptr;   // test hook: exercises the bounded wait + error report of the ticketed chain
  static const bool no_fused_pack = getenv("CAVIAR_HIP_NO_FUSED_PACK") != nullptr;  // diagnostic: the scan in the pre-sweep and k_seq_pack as its own launch, always
  const int turns = par ? (bad_ticket ? 2 : 1) : 0;
  // fused_pack: the k_iter_mid before this sweep left the chain's tables (every circuit took its in-LDS solve: chains of at most small_cap rows) --
  // the records are packed by a few more blocks of the pre-sweep (16 rows each) instead of a launch of their own
  const bool fused = fused_pack && !split && !no_fused_pack && table_it < 0;
  // the tickets from the pre-sweep's column role (one gather of a column for all its entries) when that role walks every column and the chain is long:
  // the iterations with a tiled solve
  static const bool no_col_rank = getenv("CAVIAR_HIP_NO_COLUMN_RANKS") != nullptr;   // diagnostic: k_seq_pack finds every ticket itself
  const int turns_p = (turns && !fused && list_it < 0 && !split && !no_col_rank && table_it < 0) ? (turns | 4) : turns;
  const int nbP = fused ? (std::min(N, std::max(1, s.na_max)) + 15) / 16 : 0;       // (a wavefront per chain row of the largest active set; the stride takes what a non-zero prior mean adds)
  for (int pass = 0; pass < (split ? 3 : 1); ++pass) {
    if (lean) CM_LAUNCH_SET_T(k_presweep, (true), s, nbIq + nbCk + 1 + nbP, 1024, 0, st, (table_it < 0 ? it : table_it), gated, msrmp, min_count, nbIq, nbCk, split ? (1 << pass) : 7, gran, quiescent, nbL, list_it, pack_max, fused ? 1 : 0, turns_p);
    else CM_LAUNCH_SET_T(k_presweep, (false), s, nbIq + nbCk + 1 + nbP, 1024, 0, st, (table_it < 0 ? it : table_it), gated, msrmp, min_count, nbIq, nbCk, split ? (1 << pass) : 7, gran, quiescent, nbL, list_it, pack_max, fused ? 1 : 0, turns_p);
  }
  s.tlist_sweep_it = list_it;
  if (!fused) {
    const int pack_rows = s.B > 1 ? std::min(N, std::max(64, 2 * s.na_max)) : N;      // (one circuit: the N / 4 workgroups are not what its 5 us are made of)
    CM_LAUNCH_SET(k_seq_pack, s, (pack_rows + 3) / 4, 256, 0, st, gated, msrmp, min_count, turns_p, pack_max);
  }
  if (mid) CM_HIP(hipEventRecord(mid, st));
  const size_t lds = (size_t)p->K * sizeof(double);
  const long long nrec = (long long)p->N + s.nnz_min;          // smallest allocation of `packed` (records) and of the chunk table in the set
  const int packed_cap = (int)std::min<long long>(cap, nrec);
  const int nchunks_cap = (int)std::min<long long>(CM_SEQ_LDS_CHUNKS - 1, nrec / 256 + 7);
  if (par == 1) {
    if (s.B == 1) CM_TRY(cm_raise_lds(ctx, CM_LDS_SWEEP_PAR, reinterpret_cast<const void*>(k_sweep_par<false, CM_SEQ_CAP, int>), lds_par, 0));
    else CM_TRY(cm_raise_lds(ctx, CM_LDS_SWEEP_PAR_B, reinterpret_cast<const void*>(k_sweep_par<true, CM_SEQ_CAP, int>), lds_par, 0));
    CM_LAUNCH_SET_T(k_sweep_par, (CM_SEQ_CAP, int), s, 1, 1024, lds_par, st, gated, msrmp, min_count, packed_cap, nchunks_cap);
  } else if (par == 2) {
    if (s.B == 1) CM_TRY(cm_raise_lds(ctx, CM_LDS_SWEEP_PAR2, reinterpret_cast<const void*>(k_sweep_par<false, 512, uint8_t>), lds_par, 0));
    else CM_TRY(cm_raise_lds(ctx, CM_LDS_SWEEP_PAR2_B, reinterpret_cast<const void*>(k_sweep_par<true, 512, uint8_t>), lds_par, 0));
    CM_LAUNCH_SET_T(k_sweep_par, (512, uint8_t), s, 1, 1024, lds_par, st, gated, msrmp, min_count, packed_cap, nchunks_cap);
  } else if (lds <= 88 * 1024) {     // 64 KiB of record buffers + R must fit the CU's 160 KiB
    if (s.B == 1) CM_TRY(cm_raise_lds(ctx, CM_LDS_SWEEP_SEQ, reinterpret_cast<const void*>(k_sweep_seq<false, true>), lds, 0));
    else CM_TRY(cm_raise_lds(ctx, CM_LDS_SWEEP_SEQ_B, reinterpret_cast<const void*>(k_sweep_seq<true, true>), lds, 0));
    CM_LAUNCH_SET_T(k_sweep_seq, (true), s, 1, 1024, lds, st, gated, msrmp, min_count, packed_cap, nchunks_cap, pack_max);
  } else {
    CM_LAUNCH_SET_T(k_sweep_seq, (false), s, 1, 1024, 0, st, gated, msrmp, min_count, packed_cap, nchunks_cap, pack_max);
  }
  CM_HIP(hipGetLastError());
  return CM_OK;
}

// which kernel walks the chain: 1 / 2 = multi-wave (32-bit / 8-bit tickets), 0 = single consumer.  `rows_hint` = 0 (empty active
// set) keeps the single-consumer kernel for one circuit, as before.
int cm_sweep_variant(const cm_set& s, int rows_hint) {
  static const bool no_par = getenv("CAVIAR_HIP_NO_PAR_SWEEP") != nullptr;
  static const bool force_par = getenv("CAVIAR_HIP_FORCE_PAR_SWEEP") != nullptr;   // (test hook: the multi-wave kernel whatever the density of the chain)
  const cm_problem* p = s.lead;
  int par = 0;
  if (!no_par && rows_hint > CM_PAR_MIN_ROWS) par = cm_par_class(force_par ? 0 : p->N, p->K, s.max_row_len_max, s.max_col_len_max, s.nnz_max);
  return par;
}
