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
#include "cm_bodies.h"

#define CM_FN_THREADS 256

// pava.py:9-88 with gamma = 1, last element only; literal pool walk (strict '>' merge test, pava.py:42)
__device__ __forceinline__ double fn_pava_last(const double* y, int n, double* v, double* w) {
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

// dynamic LDS: cnt [N] int | row trial [max_row_len] int | row z [max_row_len] double | amplitudes [max_row_len] double | row power [max_row_len] uint8
__host__ __device__ static inline size_t fn_scan_lds(int N, int max_row_len) {
  const size_t a = (((size_t)N * 4 + 7) / 8) * 8, b = (((size_t)max_row_len * 4 + 7) / 8) * 8;
  return a + b + (size_t)max_row_len * 16 + (size_t)max_row_len + 16;
}

template <bool BATCH>
__global__ __launch_bounds__(CM_FN_THREADS) void k_fn_scan(cm_parg<BATCH> pa, double minimax_spk_prob, double min_count, int row_cap) {
  const cm_pdev& p = pa.get();
  extern __shared__ double fn_sm[];
  const int N = p.N, K = p.K, P = p.V - 1, tid = threadIdx.x;
  int* cnt = reinterpret_cast<int*>(fn_sm);                                   // -1: not a candidate (connected, dropped or already examined)
  int* rk = cnt + ((N + 1) & ~1);
  double* rz = reinterpret_cast<double*>(rk + ((row_cap + 1) & ~1));
  double* amp = rz + row_cap;
  uint8_t* rg = reinterpret_cast<uint8_t*>(amp + row_cap);
  __shared__ int nz, focus_sh, do_rec, m_sh;
  __shared__ unsigned long long best_w[CM_FN_THREADS / CM_WAVE];
  __shared__ double srates[CM_MAX_VALS], pv[CM_MAX_VALS], pw[CM_MAX_VALS];
  const double* __restrict__ mu = p.mu;
  double* __restrict__ z = p.z;
  const int* __restrict__ row_ptr = p.row_ptr;
  const int* __restrict__ col = p.col;
  const uint8_t* __restrict__ grp = p.grp;
  const int* __restrict__ col_ptr = p.col_ptr;
  const int* __restrict__ csc_row = p.csc_row;
  if (tid == 0) nz = 0;
  for (int n = tid; n < N; n += CM_FN_THREADS) cnt[n] = (mu[n] == 0.0) ? 0 : -1;           // :103 exact zero test (quirk 7)
  __syncthreads();
  for (int k = tid; k < K; k += CM_FN_THREADS) {
    if (z[k] == 0.0) continue;
    atomicAdd(&nz, 1);
    for (int ce = col_ptr[k]; ce < col_ptr[k + 1]; ++ce) { const int n = csc_row[ce]; if (cnt[n] >= 0) atomicAdd(&cnt[n], 1); }
  }
  __syncthreads();
  const bool filter = min_count > 0.0;
  for (;;) {
    if (!((double)nz > min_count)) break;                                     // :111,141 (uniform: nz is only written between barriers)
    // the candidate with the most events, the lowest cell on ties; candidates that fell below minimum_spike_count are dropped on the way
    unsigned long long best = 0ull;                                           // (count + 1) << 32 | ~cell: 0 = none
    for (int n = tid; n < N; n += CM_FN_THREADS) {
      const int c = cnt[n];
      if (c < 0) continue;
      if (filter && !((double)c >= min_count)) { cnt[n] = -1; continue; }
      const unsigned long long key = ((unsigned long long)(unsigned)(c + 1) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)n);
      best = key > best ? key : best;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long other = __shfl_xor(best, o, CM_WAVE); best = other > best ? other : best; }
    if (tid % CM_WAVE == 0) best_w[tid / CM_WAVE] = best;
    __syncthreads();
    if (tid == 0) {
      unsigned long long b = 0ull;
      for (int w = 0; w < CM_FN_THREADS / CM_WAVE; ++w) b = best_w[w] > b ? best_w[w] : b;
      focus_sh = b ? (int)(0xFFFFFFFFu - (unsigned)(b & 0xFFFFFFFFull)) : -1;
    }
    __syncthreads();
    const int focus = focus_sh;
    if (focus < 0) break;
    // the cell's row: trial, z of the trial, power code -- staged by all threads in one round trip, examined by one
    const int e0 = row_ptr[focus], len = row_ptr[focus + 1] - e0;
    for (int i = tid; i < len; i += CM_FN_THREADS) { const int k = col[e0 + i]; rk[i] = k; rz[i] = z[k]; rg[i] = grp[e0 + i]; }
    __syncthreads();
    if (tid == 0) {
      double spike_count = 0.0;
      for (int i = 0; i < P; ++i) {                                           // :121-127
        long c = 0, hit = 0;
        for (int q = 0; q < len; ++q)
          if (rg[q] == i + 1) { ++c; hit += (rz[q] != 0.0); }
        srates[i] = c > 0 ? (double)hit / (double)c : 0.0;
        if (c > 0) spike_count += (double)hit;
      }
      const double pava = P > 0 ? fn_pava_last(srates, P, pv, pw) : 0.0;      // :128
      cnt[focus] = -1;                                                        // :139 (removed whether or not it reconnects)
      int rec = 0, m = 0;
      if (pava >= minimax_spk_prob && spike_count >= min_count) {             // :130
        // np.mean and scipy.stats.sem of the event amplitudes, summed as NumPy sums them (cm_np_sum): the amplitudes are compacted, in row
        // order, into `amp` (the row's trial indices are consumed below from rk, which amp does not touch)
        for (int q = 0; q < len; ++q) if (rz[q] != 0.0) amp[m++] = rz[q];
        const double mean = cm_np_sum(amp, m) / (double)m;
        for (int q = 0; q < m; ++q) { const double d = amp[q] - mean; amp[q] = d * d; }
        const double ssd = cm_np_sum(amp, m);
        p.mu[focus] = mean;                                                    // :134
        p.beta[focus] = m > 1 ? sqrt(ssd / (double)(m - 1)) / sqrt((double)m) : NAN;   // :135 scipy.stats.sem, ddof = 1
        rec = 1;
        nz -= m;
      }
      do_rec = rec; m_sh = m;
    }
    __syncthreads();
    if (do_rec) {
      for (int i = tid; i < len; i += CM_FN_THREADS) {                          // :136-137
        if (rz[i] == 0.0) continue;
        const int k = rk[i];
        p.lam[e0 + i] = 1.0;
        z[k] = 0.0;
        // the event leaves the counts of the other candidates targeted in that trial (their counts include it: never below zero)
        for (int ce = col_ptr[k]; ce < col_ptr[k + 1]; ++ce) { const int mrow = csc_row[ce]; if (cnt[mrow] > 0) atomicSub(&cnt[mrow], 1); }
      }
    }
    __syncthreads();
  }
}

// enqueues the scan for every circuit of the set; CM_ERR_INVALID if a circuit does not fit the kernel's LDS (the caller then takes the host path)
bool cm_fn_scan_fits(const cm_set& s) { return fn_scan_lds(s.lead->N, std::max(1, s.max_row_len_max)) <= 150 * 1024; }
int cm_launch_fn_scan(cm_ctx* ctx, cm_set& s, double minimax_spk_prob, double minimum_spike_count) {
  const int row_cap = std::max(1, s.max_row_len_max);
  const size_t lds = fn_scan_lds(s.lead->N, row_cap);
  if (lds > 150 * 1024) { cm_set_error("cm_launch_fn_scan: N = %d does not fit the scan's LDS", s.lead->N); return CM_ERR_INVALID; }
  if (s.B == 1) CM_TRY(cm_raise_lds(ctx, CM_LDS_FN_SCAN, reinterpret_cast<const void*>(k_fn_scan<false>), lds, 48 * 1024));
  else CM_TRY(cm_raise_lds(ctx, CM_LDS_FN_SCAN_B, reinterpret_cast<const void*>(k_fn_scan<true>), lds, 48 * 1024));
  CM_LAUNCH_SET(k_fn_scan, s, 1, CM_FN_THREADS, lds, ctx->stream, minimax_spk_prob, minimum_spike_count, row_cap);
  CM_HIP(hipGetLastError());
  return CM_OK;
}
