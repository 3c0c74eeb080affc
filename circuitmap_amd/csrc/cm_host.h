// Host-only pieces of the CAVIaR driver: plain C++17, no HIP.  They are compiled into libcaviar_hip.so by hipcc AND, on their own,
// into a CPU test program with -fsanitize=address,undefined (tests/test_host_sanitizers.py; SURVEY section 5: sanitizers run on the
// CPU build only).
#pragma once
#include <stdint.h>

#include <vector>

// host mirror of a circuit's compressed stimulus matrix (CSR over targeted trials + CSC)
struct cm_host_rows {
  int N = 0, K = 0, V = 0;            // V distinct stimulus values (vals[0] = untargeted), powers = vals[1:]
  const int32_t* row_ptr = nullptr;   // [N + 1]
  const int32_t* col = nullptr;       // [nnz] trial of each targeted entry
  const uint8_t* grp = nullptr;       // [nnz] index of its stimulus value
  const int32_t* col_ptr = nullptr;   // [K + 1]
  const int32_t* csc_row = nullptr;   // [nnz] neuron of each entry, by trial
};

// Threefry-2x32, 20 rounds (Salmon et al. 2011)
inline void cm_threefry2x32_host(uint32_t k0, uint32_t k1, uint32_t& x0, uint32_t& x1) {
  const uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
  static const int R[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
  x0 += ks[0]; x1 += ks[1];
  for (int g = 0; g < 5; ++g) {
    for (int i = 0; i < 4; ++i) { const int r = R[g % 2][i]; x0 += x1; x1 = (x1 << r) | (x1 >> (32 - r)); x1 ^= x0; }
    x0 += ks[(g + 1) % 3]; x1 += ks[(g + 2) % 3] + (uint32_t)(g + 1);
  }
}

// np.add.reduce over a contiguous float64 array as NumPy evaluates it (pairwise summation, numpy/core/src/umath/loops_utils.h.src:
// below 8 elements a plain loop; up to 128 eight strided partial sums combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus the
// remainder in order; above, halves split at a multiple of 8).  reconnect_spont_cells takes np.mean and scipy.stats.sem of the
// event amplitudes with it (caviar.py:134-135), so a cell reconnected on 8 or more events gets bitwise NumPy's mean only in this order
// (checked against numpy.sum for 1000 random arrays of 1..1000 elements when this was written).
#ifdef __HIPCC__
#define CM_HD __host__ __device__
#else
#define CM_HD
#endif
CM_HD static inline double cm_np_sum_block(const double* a, int n) {        // n <= 128
  if (n < 8) {
    double r = 0.0;
    for (int i = 0; i < n; ++i) r += a[i];
    return r;
  }
  double r[8];
  for (int j = 0; j < 8; ++j) r[j] = a[j];
  int i = 8;
  for (; i < n - (n % 8); i += 8)
    for (int j = 0; j < 8; ++j) r[j] += a[i + j];
  double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) res += a[i];
  return res;
}
CM_HD static inline double cm_np_sum(const double* a, int n) {
  if (n <= 128) return cm_np_sum_block(a, n);
  // post-order walk of the split tree with an explicit stack (depth <= log2(n / 64))
  int off[40], len[40], phase[40];
  double left[40];
  int sp = 0;
  off[0] = 0; len[0] = n; phase[0] = 0;
  double ret = 0.0;
  while (sp >= 0) {
    if (len[sp] <= 128) { ret = cm_np_sum_block(a + off[sp], len[sp]); --sp; continue; }
    int n2 = len[sp] / 2;
    n2 -= n2 % 8;
    if (phase[sp] == 0) { phase[sp] = 1; off[sp + 1] = off[sp]; len[sp + 1] = n2; phase[sp + 1] = 0; ++sp; }
    else if (phase[sp] == 1) { left[sp] = ret; phase[sp] = 2; off[sp + 1] = off[sp] + n2; len[sp + 1] = len[sp] - n2; phase[sp + 1] = 0; ++sp; }
    else { ret = left[sp] + ret; --sp; }
  }
  return ret;
}

int cm_perm_rounds(int N);
void cm_host_key_schedule_range(uint32_t key[2], int it0, int n_it, int N, uint32_t* keys);
void cm_host_key_schedule(uint64_t seed, int iters, int N, std::vector<uint32_t>& keys);
double cm_host_pava_last(const double* y, int n);
// reconnect_spont_cells (caviar.py:102-144) on the compressed rows; mu, beta [N], z [K] are edited in place.  lam [nnz, CSR order] is
// only ever WRITTEN (entries set to 1, :136): pass the array, or NULL and collect the entries in `ones` (the driver then leaves lam
// in HBM and sets them there).
void cm_host_reconnect(const cm_host_rows& p, double* mu, double* beta, double* lam, double* z, double minimax_spk_prob,
                       double minimum_spike_count, int verbose, std::vector<int32_t>* ones = nullptr);
