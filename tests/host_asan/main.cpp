// CPU-only driver of the library's host-side code (circuitmap_amd/csrc/cm_host.cpp), built by tests/test_host_sanitizers.py with
// g++ -fsanitize=address,undefined.  Commands on stdin, results on stdout; doubles travel as C99 hex floats (exact).
//   pava n y0 .. y(n-1)                                      -> last element of the isotonic regression
//   keys seed iters N                                        -> (N + 3) * 2 * iters uint32 of the key schedule
//   reconnect N K V nnz row_ptr[N+1] col[nnz] grp[nnz] col_ptr[K+1] csc_row[nnz] mu[N] beta[N] lam[nnz] z[K] minimax min_count
//                                                            -> mu[N] beta[N] lam[nnz] z[K]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../circuitmap_amd/csrc/cm_host.h"

static double rd() { char buf[64]; if (scanf("%63s", buf) != 1) exit(2); return strtod(buf, nullptr); }
static long ri() { long v; if (scanf("%ld", &v) != 1) exit(2); return v; }
static void pd(double v) { printf("%a ", v); }

int main() {
  char cmd[32];
  while (scanf("%31s", cmd) == 1) {
    if (!strcmp(cmd, "pava")) {
      const int n = (int)ri();
      std::vector<double> y(n);
      for (auto& v : y) v = rd();
      pd(cm_host_pava_last(y.data(), n));
      printf("\n");
    } else if (!strcmp(cmd, "keys")) {
      const unsigned long long seed = (unsigned long long)ri();
      const int iters = (int)ri(), N = (int)ri();
      std::vector<uint32_t> keys;
      cm_host_key_schedule(seed, iters, N, keys);
      // the same schedule produced in two resumed chunks must agree (this is how the driver extends it behind the GPU)
      std::vector<uint32_t> k2(keys.size(), 0u);
      uint32_t chain[2] = {(uint32_t)(seed >> 32), (uint32_t)(seed & 0xFFFFFFFFull)};
      const int first = iters / 2;
      cm_host_key_schedule_range(chain, 0, first, N, k2.data());
      cm_host_key_schedule_range(chain, first, iters - first, N, k2.data());
      if (k2 != keys) { fprintf(stderr, "chunked key schedule differs\n"); return 3; }
      printf("%d ", cm_perm_rounds(N));
      for (uint32_t k : keys) printf("%u ", k);
      printf("\n");
    } else if (!strcmp(cmd, "reconnect")) {
      cm_host_rows r;
      r.N = (int)ri(); r.K = (int)ri(); r.V = (int)ri();
      const long nnz = ri();
      std::vector<int32_t> row_ptr(r.N + 1), col(nnz), col_ptr(r.K + 1), csc_row(nnz);
      std::vector<uint8_t> grp(nnz);
      for (auto& v : row_ptr) v = (int32_t)ri();
      for (auto& v : col) v = (int32_t)ri();
      for (auto& v : grp) v = (uint8_t)ri();
      for (auto& v : col_ptr) v = (int32_t)ri();
      for (auto& v : csc_row) v = (int32_t)ri();
      std::vector<double> mu(r.N), beta(r.N), lam(nnz), z(r.K);
      for (auto& v : mu) v = rd();
      for (auto& v : beta) v = rd();
      for (auto& v : lam) v = rd();
      for (auto& v : z) v = rd();
      const double minimax = rd(), min_count = rd();
      r.row_ptr = row_ptr.data(); r.col = col.data(); r.grp = grp.data(); r.col_ptr = col_ptr.data(); r.csc_row = csc_row.data();
      cm_host_reconnect(r, mu.data(), beta.data(), lam.data(), z.data(), minimax, min_count, 0);
      for (double v : mu) pd(v);
      for (double v : beta) pd(v);
      for (double v : lam) pd(v);
      for (double v : z) pd(v);
      printf("\n");
    } else {
      fprintf(stderr, "unknown command %s\n", cmd);
      return 2;
    }
  }
  return 0;
}
