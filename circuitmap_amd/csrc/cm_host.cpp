// Host-only, sequential pieces of the CAVIaR driver (see cm_host.h): the jax key chain of caviar() (threefry splits are a serial
// dependency chain), the literal pool-adjacent-violators walk, and reconnect_spont_cells.
#include "cm_host.h"

#include <math.h>
#include <stdio.h>

#include <queue>

// ------------------------------------------------------------------------------------------------
// RNG.  Key schedule: per iteration (3 + N) threefry keys:
//   [0..2]  sort-key subkeys of the permutation rounds (jax _shuffle), [3 + m] the uniform key of sweep step m.
// ------------------------------------------------------------------------------------------------
int cm_perm_rounds(int N) {
  const double r = 3.0 * log((double)(N > 1 ? N : 1)) / log(4294967295.0);
  int k = (int)ceil(r);
  return k < 0 ? 0 : k;
}

static void host_split(const uint32_t key[2], uint32_t out0[2], uint32_t out1[2]) {
  uint32_t a0 = 0, a1 = 2, b0 = 1, b1 = 3;
  cm_threefry2x32_host(key[0], key[1], a0, a1);
  cm_threefry2x32_host(key[0], key[1], b0, b1);
  out0[0] = a0; out0[1] = b0;
  out1[0] = a1; out1[1] = b1;
}

// jax key handling of caviar(): see oracle/threefry.py JaxLikeStream (caviar.py:76,196,209-210,227,251,304)
// Resumable: `key` is the carried jax key (PRNGKey(seed) before iteration 0), updated in place; writes the (3 + N) keys of
// iterations [it0, it0 + n_it) at their place in `keys` (which holds all iterations).
void cm_host_key_schedule_range(uint32_t key[2], int it0, int n_it, int N, uint32_t* keys) {
  const size_t per = (size_t)(N + 3) * 2;
  const int rounds = cm_perm_rounds(N);
  for (int it = it0; it < it0 + n_it; ++it) {
    uint32_t* out = keys + per * it;
    uint32_t k[2] = {key[0], key[1]}, s0[2], s1[2];
    for (int r = 0; r < rounds && r < 3; ++r) {       // key, subkey = split(key)
      host_split(k, s0, s1);
      out[2 * r] = s1[0]; out[2 * r + 1] = s1[1];
      k[0] = s0[0]; k[1] = s0[1];
    }
    k[0] = key[0]; k[1] = key[1];
    for (int m = 0; m < N; ++m) {                      // key, key_next = split(key); uniform(key); carry key_next
      host_split(k, s0, s1);
      out[2 * (3 + m)] = s0[0]; out[2 * (3 + m) + 1] = s0[1];
      k[0] = s1[0]; k[1] = s1[1];
    }
    host_split(k, s0, s1);                             // update_phi returns split(key)[1]
    key[0] = s1[0]; key[1] = s1[1];
  }
}

void cm_host_key_schedule(uint64_t seed, int iters, int N, std::vector<uint32_t>& keys) {
  keys.assign((size_t)(N + 3) * 2 * iters, 0u);
  uint32_t key[2] = {(uint32_t)(seed >> 32), (uint32_t)(seed & 0xFFFFFFFFull)};
  cm_host_key_schedule_range(key, 0, iters, N, keys.data());
}


double cm_host_pava_last(const double* y, int n) {   // pava.py:9-88, gamma = 1, literal pool walk
  if (n <= 0) return 0.0;
  std::vector<double> v(n), w(n);
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


// reconnect_spont_cells, caviar.py:102-144.  Host C++ on the compressed rows (one-off, after the last iteration).
void cm_host_reconnect(const cm_host_rows& p, double* mu, double* beta, double* lam, double* z, double minimax_spk_prob,
                       double minimum_spike_count, int verbose, std::vector<int32_t>* ones) {
  const int N = p.N, K = p.K, V = p.V;
  const int P = V - 1;
  std::vector<int> disc;
  for (int n = 0; n < N; ++n) if (mu[n] == 0.0) disc.push_back(n);           // :103 exact zero test (quirk 7)
  const size_t init = disc.size();
  // np.where(stim_matrix[n]) selects NON-ZERO entries; targeted entries (> 0) are the CSR row.  Negative stimulus values
  // are outside the reference's contract (powers are non-negative) and are treated as untargeted here.
  // The reference re-scans all remaining cells in every pass (:109-117): it takes the cell with the most spontaneous events on
  // its targeted trials (first maximum in ascending cell order), tests it, and removes it.  Counts only ever decrease (a
  // reconnection clears events from z), so a max-heap keyed (count desc, cell asc) with lazy refresh of stale entries visits
  // the cells in exactly that order; a reconnection updates the counts of the cells sharing the cleared trials (host CSC).
  // Spontaneous events on a cell's targeted trials, counted from the events' side: the trials with z != 0 are few (a few percent of K),
  // each lists its targeted cells (CSC) -- K + nz x H steps instead of one pass over every candidate's row.
  long nz = 0;
  std::vector<long> cnt(N, -1);                         // -1: not a candidate (connected, or already visited)
  for (int n : disc) cnt[n] = 0;
  for (int k = 0; k < K; ++k) {
    if (z[k] == 0.0) continue;
    ++nz;
    for (int ce = p.col_ptr[k]; ce < p.col_ptr[k + 1]; ++ce) { const int n = p.csc_row[ce]; if (cnt[n] >= 0) ++cnt[n]; }
  }
  struct Ent { long c; int n; };
  auto worse = [](const Ent& a, const Ent& b) { return a.c < b.c || (a.c == b.c && a.n > b.n); };
  std::priority_queue<Ent, std::vector<Ent>, decltype(worse)> heap(worse);
  for (int n : disc) {
    const long c = cnt[n];
    // A cell can only reconnect with spike_count >= minimum_spike_count (:130), spike_count is exactly this count (every targeted
    // entry belongs to one power), and counts only decrease: cells below the threshold are examined by the reference to no
    // effect, so they never enter the heap (identical outcome; most cells have 0 or 1 spontaneous events on their trials).
    if (!(minimum_spike_count > 0.0) || (double)c >= minimum_spike_count) heap.push({c, n});
    else cnt[n] = -1;
  }
  std::vector<double> srates(P > 0 ? P : 0);
  std::vector<int> locs;
  std::vector<double> amp;
  while (!heap.empty()) {
    if (!((double)nz > minimum_spike_count)) break;                           // :111,141
    const Ent top = heap.top();
    heap.pop();
    if (cnt[top.n] < 0) continue;                                             // already visited
    if (top.c != cnt[top.n]) {                                                // stale: re-queue with the current count
      if (!(minimum_spike_count > 0.0) || (double)cnt[top.n] >= minimum_spike_count) heap.push({cnt[top.n], top.n});
      else cnt[top.n] = -1;
      continue;
    }
    const int focus = top.n;
    double spike_count = 0.0;
    for (int i = 0; i < P; ++i) {                                             // :121-127
      long c = 0, hit = 0;
      for (int e = p.row_ptr[focus]; e < p.row_ptr[focus + 1]; ++e)
        if (p.grp[e] == i + 1) { ++c; hit += (z[p.col[e]] != 0.0); }
      srates[i] = c > 0 ? (double)hit / (double)c : 0.0;
      if (c > 0) spike_count += (double)hit;
    }
    const double pava = P > 0 ? cm_host_pava_last(srates.data(), P) : 0.0;    // :128
    cnt[focus] = -1;                                                          // :139 (removed whether or not it reconnects)
    if (pava >= minimax_spk_prob && spike_count >= minimum_spike_count) {     // :130
      locs.clear();
      for (int e = p.row_ptr[focus]; e < p.row_ptr[focus + 1]; ++e) if (z[p.col[e]] != 0.0) locs.push_back(e);
      const size_t m = locs.size();
      amp.resize(m);
      for (size_t i = 0; i < m; ++i) amp[i] = z[p.col[locs[i]]];
      const double mean = cm_np_sum(amp.data(), (int)m) / (double)m;           // np.mean: NumPy's pairwise sum
      for (size_t i = 0; i < m; ++i) { const double d = amp[i] - mean; amp[i] = d * d; }
      const double ssd = cm_np_sum(amp.data(), (int)m);                        // np.std(ddof = 1) inside scipy.stats.sem
      mu[focus] = mean;                                                        // :134
      beta[focus] = m > 1 ? sqrt(ssd / (double)(m - 1)) / sqrt((double)m) : NAN;   // :135 scipy.stats.sem, ddof = 1
      for (int e : locs) {                                                     // :136-137
        const int k = p.col[e];
        if (lam) lam[e] = 1.0;
        if (ones) ones->push_back(e);
        z[k] = 0.0;
        for (int ce = p.col_ptr[k]; ce < p.col_ptr[k + 1]; ++ce) { const int mrow = p.csc_row[ce]; if (cnt[mrow] > 0) --cnt[mrow]; }
      }
      nz -= (long)m;
      if (verbose) fprintf(stderr, "Reconnecting cell %d with maximal pava spike rate %.2f\n", focus, pava);
    }
  }
  if (verbose) fprintf(stderr, "Examined %zu/%d cells for false negatives. Cell reconnection complete.\n", init, N);
}

