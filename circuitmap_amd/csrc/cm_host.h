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

int cm_perm_rounds(int N);
void cm_host_key_schedule_range(uint32_t key[2], int it0, int n_it, int N, uint32_t* keys);
void cm_host_key_schedule(uint64_t seed, int iters, int N, std::vector<uint32_t>& keys);
double cm_host_pava_last(const double* y, int n);
// reconnect_spont_cells (caviar.py:102-144) on the compressed rows; mu, beta [N], z [K] are edited in place.  lam [nnz, CSR order] is
// only ever WRITTEN (entries set to 1, :136): pass the array, or NULL and collect the entries in `ones` (the driver then leaves lam
// in HBM and sets them there).
void cm_host_reconnect(const cm_host_rows& p, double* mu, double* beta, double* lam, double* z, double minimax_spk_prob,
                       double minimum_spike_count, int verbose, std::vector<int32_t>* ones = nullptr);
