// TEST INFRASTRUCTURE ONLY -- compiled (C++17 / OpenMP) CPU restatement of the reference's CAVIaR path, running-residual formulation.
//
// What it is for: bench.py's `cpu_baseline` leg (kind "port": the reference's own jitted JAX-CPU path cannot run in this image -- jax is a
// third-party dependency absent from /root/reference and from the image) and tests/test_caviar_port.py, which holds
// it to the NumPy oracle (oracle/caviar_oracle.py, literal=False -- itself pinned bit for bit, in its literal form, by the fixtures recorded
// from the reference's source).  Follows circuitmap/optimise/caviar.py and pava.py phase by phase (citations are file:line relative to
// /root/reference), with the same algebraic restructurings as the oracle's fast formulation: running residual R = sum_m mu_m lam_m instead of the
// O(N^2 K) leave-one-out sum (caviar.py:204-206), active-set Cholesky for block_update_mu (:166-172), per-power sufficient statistics for the
// Newton sums of update_phi (:274-291, :312-316).  Third-party arithmetic the reference reaches through jax is restated here:
// Threefry-2x32-20 + the jax-0.3 key conventions (as oracle/threefry.py), ndtr = erfc / 2, ndtri = Wichura AS 241 (1e-15 of SciPy's).
// Threads: the Monte-Carlo logits, the trailing update / inverse of the weight solve and the per-neuron Newton iterations run under OpenMP;
// the Gauss-Seidel sweep is sequential as in the reference (caviar.py:229).
//
// NOT part of the product: nothing under circuitmap_amd/ includes, links or loads this file (the product has no CPU path at all).
// Build (by __graft_entry__.build()):  g++ -O3 -march=x86-64-v3 -fopenmp -shared -fPIC oracle/caviar_port.cpp -o oracle/libcaviar_port.so
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <limits>
#include <numeric>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ---------------------------------------------------------------------------------------------------------------- PRNG (jax.random, 0.3 series)
inline void threefry2x32(uint32_t k0, uint32_t k1, uint32_t& x0, uint32_t& x1) {
  const uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
  static const int R[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
  x0 += ks[0]; x1 += ks[1];
  for (int g = 0; g < 5; ++g) {
    for (int i = 0; i < 4; ++i) { const int r = R[g % 2][i]; x0 += x1; x1 = (x1 << r) | (x1 >> (32 - r)); x1 ^= x0; }
    x0 += ks[(g + 1) % 3]; x1 += ks[(g + 2) % 3] + (uint32_t)(g + 1);
  }
}
// n 32-bit words from counters iota(n) (odd n padded with one zero counter): first half -> x0 lane, second half -> x1 lane
void bits32(const uint32_t key[2], int n, uint32_t* out) {
  const int odd = n & 1, half = (n + odd) / 2;
  for (int i = 0; i < half; ++i) {
    uint32_t x0 = (uint32_t)i, x1 = (half + i < n) ? (uint32_t)(half + i) : 0u;
    threefry2x32(key[0], key[1], x0, x1);
    out[i] = x0;
    if (half + i < n) out[half + i] = x1;
  }
}
void split2(const uint32_t key[2], uint32_t a[2], uint32_t b[2]) {          // split(key, 2)
  uint32_t w[4];
  bits32(key, 4, w);
  a[0] = w[0]; a[1] = w[1]; b[0] = w[2]; b[1] = w[3];
}
void uniform_f64(const uint32_t key[2], int size, double* out) {            // 2 size words: hi = first half, lo = second half
  std::vector<uint32_t> w(2 * (size_t)size);
  bits32(key, 2 * size, w.data());
  for (int i = 0; i < size; ++i) {
    const uint64_t bits = ((uint64_t)w[i] << 32) | (uint64_t)w[size + i];
    const uint64_t fb = (bits >> 12) | 0x3FF0000000000000ull;
    double d;
    memcpy(&d, &fb, 8);
    out[i] = d - 1.0;
  }
}
void permutation(const uint32_t key_in[2], int n, int* x) {                  // choice(key, n, [n], replace=False)
  std::iota(x, x + n, 0);
  const int rounds = (int)ceil(3.0 * log((double)std::max(1, n)) / log(4294967295.0));
  uint32_t key[2] = {key_in[0], key_in[1]};
  std::vector<uint32_t> sk(n);
  std::vector<int> idx(n), y(n);
  for (int r = 0; r < rounds; ++r) {
    uint32_t nk[2], sub[2];
    split2(key, nk, sub);
    key[0] = nk[0]; key[1] = nk[1];
    bits32(sub, n, sk.data());
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return sk[a] < sk[b]; });
    for (int i = 0; i < n; ++i) y[i] = x[idx[i]];
    std::copy(y.begin(), y.end(), x);
  }
}

// ---------------------------------------------------------------------------------------------------------------- special functions
inline double expit(double x) { return 1.0 / (1.0 + exp(-x)); }
inline double ndtr(double x) { return 0.5 * erfc(-x * 0.70710678118654752440); }
double ndtri(double p) {                                                     // Wichura (1988), AS 241, PPND16
  const double q = p - 0.5;
  if (fabs(q) <= 0.425) {
    const double r = 0.180625 - q * q;
    const double n = (((((((2.5090809287301226727e3 * r + 3.3430575583588128105e4) * r + 6.7265770927008700853e4) * r + 4.5921953931549871457e4) * r +
                         1.3731693765509461125e4) * r + 1.9715909503065514427e3) * r + 1.3314166789178437745e2) * r + 3.3871328727963666080e0);
    const double d = (((((((5.2264952788528545610e3 * r + 2.8729085735721942674e4) * r + 3.9307895800092710610e4) * r + 2.1213794301586595867e4) * r +
                         5.3941960214247511077e3) * r + 6.8718700749205790830e2) * r + 4.2313330701600911252e1) * r + 1.0);
    return q * n / d;
  }
  double r = q < 0.0 ? p : 1.0 - p;
  if (!(r > 0.0)) return r == 0.0 ? (q < 0.0 ? -INFINITY : INFINITY) : NAN;
  r = sqrt(-log(r));
  double v;
  if (r <= 5.0) {
    r -= 1.6;
    v = (((((((7.74545014278341407640e-4 * r + 2.27238449892691845833e-2) * r + 2.41780725177450611770e-1) * r + 1.27045825245236838258e0) * r +
            3.64784832476320460504e0) * r + 5.76949722146069140550e0) * r + 4.63033784615654529590e0) * r + 1.42343711074968357734e0) /
        (((((((1.05075007164441684324e-9 * r + 5.47593808499534494600e-4) * r + 1.51986665636164571966e-2) * r + 1.48103976427480074590e-1) * r +
            6.89767334985100004550e-1) * r + 1.67638483018380384940e0) * r + 2.05319162663775882187e0) * r + 1.0);
  } else {
    r -= 5.0;
    v = (((((((2.01033439929228813265e-7 * r + 2.71155556874348757815e-5) * r + 1.24266094738807843860e-3) * r + 2.65321895265761230930e-2) * r +
            2.96560571828504891230e-1) * r + 1.78482653991729133580e0) * r + 5.46378491116411436990e0) * r + 6.65790464350110377720e0) /
        (((((((2.04426310338993978564e-15 * r + 1.42151175831644588870e-7) * r + 1.84631831751005468180e-5) * r + 7.86869131145613259100e-4) * r +
            1.48753612908506148525e-2) * r + 1.36929880922735805310e-1) * r + 5.99832206555887937690e-1) * r + 1.0);
  }
  return q < 0.0 ? -v : v;
}
inline double nan_to_num(double x) {
  if (std::isnan(x)) return 0.0;
  if (std::isinf(x)) return x > 0 ? std::numeric_limits<double>::max() : -std::numeric_limits<double>::max();
  return x;
}
double pava_last(const double* y, int n) {                                   // isotonic_regression(y)[-1] == max_j mean(y[j:])  (pava.py:9-88)
  double best = -INFINITY, s = 0.0;
  for (int j = n - 1; j >= 0; --j) { s += y[j]; const double m = s / (double)(n - j); if (m > best) best = m; }
  return best;
}
double isotonic_last_literal(const double* y, int n) {                       // the pool walk itself (reconnect_spont_cells uses it: caviar.py:128)
  std::vector<double> v(n), w(n);
  int i = 0;
  v[0] = y[0]; w[0] = 1.0;
  for (int t = 1; t < n; ++t) {
    ++i; v[i] = y[t]; w[i] = 1.0;
    while (i > 0 && (v[i - 1] / w[i - 1]) > (v[i] / w[i])) { --i; v[i] += v[i + 1]; w[i] += w[i + 1]; }
  }
  return v[i] / w[i];
}
void inv2x2(const double* m, double* o) {                                    // LU with partial pivoting (what LAPACK getrf / getri do on a 2 x 2)
  double a = m[0], b = m[1], c = m[2], d = m[3];
  const bool sw = fabs(c) > fabs(a);
  if (sw) { std::swap(a, c); std::swap(b, d); }
  const double l = c / a, u = d - l * b;
  double r0 = sw ? 0.0 : 1.0, r1 = sw ? 1.0 : 0.0;
  double x1 = (r1 - l * r0) / u;
  o[0] = (r0 - b * x1) / a; o[2] = x1;
  r0 = sw ? 1.0 : 0.0; r1 = sw ? 0.0 : 1.0;
  x1 = (r1 - l * r0) / u;
  o[1] = (r0 - b * x1) / a; o[3] = x1;
}

// ---------------------------------------------------------------------------------------------------------------- update_phi on sufficient statistics
struct Suff { double cnt, n0, n1, smid, nmid; };
double phi_nll(const double phi[2], int V, const double* vals, const Suff* st, const double pr[2], const double prec[4], double t) {
  double tot = 0.0;                                                          // caviar.py:312-316, nan_to_num per ELEMENT (quirk 11)
  for (int g = 0; g < V; ++g) {
    const double s = expit(phi[0] * vals[g] - phi[1]);
    const double ls = log(s), l1s = log(1.0 - s);
    if (st[g].n0 > 0) tot += st[g].n0 * nan_to_num(0.0 * ls + 1.0 * l1s);
    if (st[g].n1 > 0) tot += st[g].n1 * nan_to_num(1.0 * ls + 0.0 * l1s);
    if (st[g].nmid > 0) {
      if (std::isfinite(ls) && std::isfinite(l1s)) tot += st[g].smid * ls + (st[g].nmid - st[g].smid) * l1s;
      else tot += st[g].nmid * nan_to_num(0.5 * ls + 0.5 * l1s);
    }
  }
  const double d0 = phi[0] - pr[0], d1 = phi[1] - pr[1];
  return -tot - (log(phi[0]) + log(phi[1])) / t + 0.5 * (d0 * (prec[0] * d0 + prec[1] * d1) + d1 * (prec[2] * d0 + prec[3] * d1));
}
void laplace_one(int V, const double* vals, const Suff* st, const double pr[2], const double cov_prior[4], double phi_out[2], double cov_out[4]) {
  const double t = 10.0, alpha = 0.25, beta = 0.5;                           // caviar.py:253-308
  double prec[4];
  inv2x2(cov_prior, prec);                                                   // :306
  double phi[2] = {pr[0], pr[1]}, cov[4] = {0, 0, 0, 0};                     // :305 restart from the prior (quirk 3)
  for (int step_i = 0; step_i < 10; ++step_i) {                              // :308
    double j1 = 0, j2 = 0, h11 = 0, h12 = 0, h22 = 0;
    for (int g = 0; g < V; ++g) {
      const double f = expit(phi[0] * vals[g] - phi[1]);
      const double sy = st[g].n1 + st[g].smid, r = sy - st[g].cnt * f, ff = st[g].cnt * f * (1.0 - f);
      j1 -= vals[g] * r; j2 += r;
      h11 += vals[g] * vals[g] * ff; h12 -= vals[g] * ff; h22 += ff;
    }
    const double d0 = phi[0] - pr[0], d1 = phi[1] - pr[1];
    const double J0 = j1 + (prec[0] * d0 + prec[1] * d1) - 1.0 / (t * phi[0]), J1 = j2 + (prec[2] * d0 + prec[3] * d1) - 1.0 / (t * phi[1]);   // :278-280
    const double H[4] = {h11 + prec[0] + 1.0 / (t * phi[0] * phi[0]), h12 + prec[1], h12 + prec[2], h22 + prec[3] + 1.0 / (t * phi[1] * phi[1])};
    inv2x2(H, cov);                                                          // :289, :295: the covariance BEFORE the update is what is returned
    const double v0 = -(cov[0] * J0 + cov[1] * J1), v1 = -(cov[2] * J0 + cov[3] * J1);
    const double Jv = J0 * v0 + J1 * v1;
    const double f0 = phi_nll(phi, V, vals, st, pr, prec, t);
    double step = 1.0;
    double cand[2] = {phi[0] + step * v0, phi[1] + step * v1};
    double lhs = phi_nll(cand, V, vals, st, pr, prec, t), rhs = f0 + alpha * step * Jv;
    int bt = 0;
    while (bt < 40 && (std::isnan(lhs) || lhs > rhs)) {                      // :259-268
      ++bt; step *= beta;
      cand[0] = phi[0] + step * v0; cand[1] = phi[1] + step * v1;
      lhs = phi_nll(cand, V, vals, st, pr, prec, t); rhs = f0 + alpha * step * Jv;
    }
    phi[0] += step * v0; phi[1] += step * v1;                                // :301
  }
  phi_out[0] = phi[0]; phi_out[1] = phi[1];
  memcpy(cov_out, cov, sizeof(cov));
}

// ---------------------------------------------------------------------------------------------------------------- SPD solve on the active set
// mean = A^-1 b, var = diag(A^-1) through L (blocked right-looking Cholesky, trailing update and triangular inverse under OpenMP)
bool spd_mean_var(std::vector<double>& A, int n, const std::vector<double>& b, std::vector<double>& mean, std::vector<double>& var) {
  const int NB = 64;
  bool ok = true;
  for (int kb = 0; kb < n; kb += NB) {
    const int ke = std::min(n, kb + NB);
    for (int i = kb; i < ke; ++i) {                                          // diagonal block, row by row
      for (int j = kb; j <= i; ++j) {
        double s = A[(size_t)i * n + j];
        for (int k = kb; k < j; ++k) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
        if (i == j) { if (!(s > 0.0)) ok = false; A[(size_t)i * n + i] = sqrt(s); }
        else A[(size_t)i * n + j] = s / A[(size_t)j * n + j];
      }
    }
#pragma omp parallel for schedule(static)
    for (int i = ke; i < n; ++i) {                                           // panel: L[i][kb:ke] = A[i][kb:ke] L_kk^-T
      for (int j = kb; j < ke; ++j) {
        double s = A[(size_t)i * n + j];
        for (int k = kb; k < j; ++k) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
        A[(size_t)i * n + j] = s / A[(size_t)j * n + j];
      }
    }
#pragma omp parallel for schedule(dynamic, 8)
    for (int i = ke; i < n; ++i) {                                           // trailing update, lower triangle
      const double* li = &A[(size_t)i * n + kb];
      for (int j = ke; j <= i; ++j) {
        const double* lj = &A[(size_t)j * n + kb];
        double s = 0.0;
#pragma omp simd reduction(+ : s)
        for (int k = 0; k < ke - kb; ++k) s += li[k] * lj[k];
        A[(size_t)i * n + j] -= s;
      }
    }
  }
  // X = L^-1, column by column (forward substitution on unit vectors), stored transposed: Xt[c][r] = X[r][c]
  std::vector<double> Xt((size_t)n * n, 0.0);
#pragma omp parallel for schedule(dynamic, 4)
  for (int c = 0; c < n; ++c) {
    double* x = &Xt[(size_t)c * n];
    x[c] = 1.0 / A[(size_t)c * n + c];
    for (int r = c + 1; r < n; ++r) {
      const double* lr = &A[(size_t)r * n];
      double s = 0.0;
#pragma omp simd reduction(+ : s)
      for (int k = c; k < r; ++k) s += lr[k] * x[k];
      x[r] = -s / lr[r];
    }
  }
  std::vector<double> w(n, 0.0);                                             // w = X b
#pragma omp parallel for schedule(static)
  for (int r = 0; r < n; ++r) {
    double s = 0.0;
    for (int c = 0; c <= r; ++c) s += Xt[(size_t)c * n + r] * b[c];
    w[r] = s;
  }
  mean.assign(n, 0.0); var.assign(n, 0.0);
#pragma omp parallel for schedule(static)
  for (int c = 0; c < n; ++c) {                                              // mean = X^T w, var = colsum(X^2)
    const double* x = &Xt[(size_t)c * n];
    double sm = 0.0, sv = 0.0;
    for (int r = c; r < n; ++r) { sm += x[r] * w[r]; sv += x[r] * x[r]; }
    mean[c] = sm; var[c] = sv;
  }
  return ok;
}

}  // namespace

// caviar(): reference circuitmap/optimise/caviar.py:20-100.  Dense row-major inputs (psc [K x T], stim [N x K]), priors as in Model
// (mu, beta [N], shape, rate, phi [N x 2], phi_cov [N x 2 x 2]); outputs mu, beta [N], lam [N x K] dense, shape_rate [2], phi, phi_cov, z [K].
// Returns 0, or -1 (bad argument / more than 64 distinct stimulus values), -2 (precision matrix not positive definite).
extern "C" int caviar_port(int N, int K, int T, const double* psc, const double* stim, const double* mu_prior, const double* beta_prior, double shape_prior,
                           double rate_prior, const double* phi_prior, const double* phi_cov_prior, int iters, int S, uint64_t seed, double y_xcorr_thresh,
                           double minimum_spike_count, int delay_spont_est, double msrmp, double scale_factor, double penalty, int max_backtrack_iters,
                           double tol, double spont_orthogonality, int fn_scan, int n_threads, double* mu_out, double* beta_out, double* lam_out,
                           double* shape_rate_out, double* phi_out, double* phi_cov_out, double* z_out) {
  if (N < 2 || K < 1 || T < 1 || !psc || !stim) return -1;
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
  // ---- prologue (caviar.py:28-51)
  std::vector<double> y(K);
  std::vector<char> mask(K);
#pragma omp parallel for schedule(static)
  for (int k = 0; k < K; ++k) {
    const double* p = psc + (size_t)k * T;
    double s = 0.0, ss = 0.0;
    for (int t = 0; t < T; ++t) { s += p[t]; ss += p[t] * p[t]; }
    y[k] = T > 1 ? s - 0.5 * (p[0] + p[T - 1]) : 0.0;                         // np.trapz, unit spacing (:28)
    mask[k] = ss > y_xcorr_thresh;                                           // :30
  }
  std::vector<double> uniq(stim, stim + (size_t)N * K);
  std::sort(uniq.begin(), uniq.end());
  uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
  if (uniq.size() > 64 || uniq.empty() || uniq[0] != 0.0) return -1;       // (a stimulus matrix without zeros -- SURVEY quirk 8 -- is left to the NumPy oracle)
  const std::vector<double> powers(uniq.begin() + 1, uniq.end());            // :42 np.unique(I)[1:]
  const int P = (int)powers.size();
  // compressed rows over the targeted (I > 0) entries; group 0 of update_phi's statistics is I == 0 (lam == 0 there, :216)
  std::vector<int> row_ptr(N + 1, 0), col, grp;
  std::vector<double> ival, lam;
  std::vector<int> cnt((size_t)N * std::max(P, 1), 0), cnt0(N, 0);
  for (int n = 0; n < N; ++n) {
    for (int k = 0; k < K; ++k) {
      const double v = stim[(size_t)n * K + k];
      int g = -1;
      for (int p = 0; p < P; ++p) if (v == powers[p]) g = p;                 // exact float equality (:181)
      if (g >= 0) ++cnt[(size_t)n * P + g];
      if (v == 0.0) ++cnt0[n];
      if (v > 0) { col.push_back(k); grp.push_back(g); ival.push_back(v); lam.push_back(mask[k] ? 0.95 : 0.0); }   // :32-34
    }
    row_ptr[n + 1] = (int)col.size();
  }
  const int nnz = (int)col.size();
  std::vector<int> col_ptr(K + 1, 0), csc_row(nnz), csc_pos(nnz);
  for (int e = 0; e < nnz; ++e) ++col_ptr[col[e] + 1];
  for (int k = 0; k < K; ++k) col_ptr[k + 1] += col_ptr[k];
  {
    std::vector<int> cur(col_ptr.begin(), col_ptr.end() - 1);
    for (int n = 0; n < N; ++n)
      for (int e = row_ptr[n]; e < row_ptr[n + 1]; ++e) { const int q = cur[col[e]]++; csc_row[q] = n; csc_pos[q] = e; }
  }
  std::vector<double> mu(mu_prior, mu_prior + N), beta(beta_prior, beta_prior + N), phi(phi_prior, phi_prior + 2 * (size_t)N),
      phi_cov(phi_cov_prior, phi_cov_prior + 4 * (size_t)N), z(K, 0.0), R(K, 0.0);
  double shape = shape_prior, rate = rate_prior, spont_rate = 0.0;
  const double sum_y2 = std::inner_product(y.begin(), y.end(), y.begin(), 0.0), sum_y = std::accumulate(y.begin(), y.end(), 0.0);
  uint32_t key[2] = {(uint32_t)(seed >> 32), (uint32_t)(seed & 0xFFFFFFFFull)};   // PRNGKey(seed) (:76)
  // update_phi with a per-row cache: a row whose statistics did not change keeps its result (all-zero rows repeat the same problem)
  const int V = P + 1;
  std::vector<double> vals(V, 0.0);
  for (int p = 0; p < P; ++p) vals[p + 1] = powers[p];
  std::vector<Suff> last_st((size_t)N * V);
  std::vector<char> have_last(N, 0);
  auto update_phi = [&]() {
#pragma omp parallel for schedule(dynamic, 16)
    for (int n = 0; n < N; ++n) {
      Suff st[65];
      for (int g = 0; g < V; ++g) st[g] = Suff{g == 0 ? (double)cnt0[n] : (double)cnt[(size_t)n * P + g - 1], 0, 0, 0, 0};
      for (int e = row_ptr[n]; e < row_ptr[n + 1]; ++e) {
        const int g = grp[e] + 1;
        if (g < 1) continue;
        const double l = lam[e];
        if (l == 1.0) st[g].n1 += 1;
        else if (l > 0.0 && l < 1.0) { st[g].nmid += 1; st[g].smid += l; }
      }
      for (int g = 0; g < V; ++g) st[g].n0 = st[g].cnt - st[g].n1 - st[g].nmid;
      if (have_last[n] && memcmp(st, &last_st[(size_t)n * V], sizeof(Suff) * V) == 0) continue;
      laplace_one(V, vals.data(), st, phi_prior + 2 * (size_t)n, phi_cov_prior + 4 * (size_t)n, &phi[2 * (size_t)n], &phi_cov[4 * (size_t)n]);
      memcpy(&last_st[(size_t)n * V], st, sizeof(Suff) * V);
      have_last[n] = 1;
    }
  };
  std::vector<int> order(N), act_inv(N);
  std::vector<double> mcE((size_t)N * std::max(P, 1)), u_all;
  std::vector<uint32_t> step_keys(2 * (size_t)N);
  for (int it = 0; it < iters; ++it) {                                       // caviar.py:79
    // ---- block_update_mu (:166-172) on the active set; rows with lam == 0 decouple exactly: mean = mu0, var = beta0^2
    const double sig = shape / rate;
    std::vector<int> act;
    for (int n = 0; n < N; ++n) {
      bool any = false;
      for (int e = row_ptr[n]; e < row_ptr[n + 1] && !any; ++e) any = lam[e] != 0.0;
      act_inv[n] = any ? (int)act.size() : -1;
      if (any) act.push_back(n);
      mu[n] = mu_prior[n]; beta[n] = beta_prior[n] * beta_prior[n];
    }
    const int na = (int)act.size();
    if (na > 0) {
      std::vector<double> A((size_t)na * na, 0.0), b(na), mean, var;
      for (int k = 0; k < K; ++k)                                            // L = lam lam^T from the co-targeted pairs of every trial
        for (int q1 = col_ptr[k]; q1 < col_ptr[k + 1]; ++q1) {
          const int a = act_inv[csc_row[q1]];
          const double la = lam[csc_pos[q1]];
          if (a < 0 || la == 0.0) continue;
          for (int q2 = col_ptr[k]; q2 < col_ptr[k + 1]; ++q2) {
            const int bb = act_inv[csc_row[q2]];
            if (bb >= 0 && bb <= a) A[(size_t)a * na + bb] += la * lam[csc_pos[q2]];
          }
        }
      for (int a = 0; a < na; ++a) {
        const int n = act[a];
        double d = 0.0, ly = 0.0;
        for (int e = row_ptr[n]; e < row_ptr[n + 1]; ++e) { d += lam[e] * (1.0 - lam[e]); ly += lam[e] * y[col[e]]; }
        for (int bb = 0; bb <= a; ++bb) A[(size_t)a * na + bb] *= sig;
        const double pr = 1.0 / (beta_prior[n] * beta_prior[n]);
        A[(size_t)a * na + a] += sig * d + pr;
        b[a] = sig * ly + mu_prior[n] * pr;
      }
      if (!spd_mean_var(A, na, b, mean, var)) return -2;
      for (int a = 0; a < na; ++a) { mu[act[a]] = mean[a]; beta[act[a]] = var[a]; }
    }
    // ---- update_lam (:190-231): RNG keys of the sweep (sequential chain), Monte-Carlo logits of all neurons (parallel), then the sweep
    permutation(key, N, order.data());                                       // :196 choice(key) uses the carried key itself
    for (int m = 0; m < N; ++m) {                                            // :209: key, key_next = split(key)
      uint32_t a[2], b2[2];
      split2(key, a, b2);
      step_keys[2 * (size_t)m] = a[0]; step_keys[2 * (size_t)m + 1] = a[1];
      key[0] = b2[0]; key[1] = b2[1];
    }
#pragma omp parallel
    {
      std::vector<double> u(2 * (size_t)S), mc(2 * (size_t)S);
#pragma omp for schedule(dynamic, 8)
      for (int m = 0; m < N; ++m) {
        const int n = order[m];
        uniform_f64(&step_keys[2 * (size_t)m], 2 * S, u.data());             // :210 uniform(key, [S, 2])
        const double mean0 = phi[2 * (size_t)n], mean1 = phi[2 * (size_t)n + 1], sd0 = phi_cov[4 * (size_t)n], sd1 = phi_cov[4 * (size_t)n + 3];   // :211 (variance as sdev, quirk 2)
        const double c0 = ndtr(-mean0 / sd0), c1 = ndtr(-mean1 / sd1);
        for (int s = 0; s < S; ++s) {                                        // :212
          mc[2 * s] = ndtri(c0 + u[2 * s] * (1.0 - c0)) * sd0 + mean0;
          mc[2 * s + 1] = ndtri(c1 + u[2 * s + 1] * (1.0 - c1)) * sd1 + mean1;
        }
        for (int p = 0; p < P; ++p) {                                        // :215, :233-236
          double acc = 0.0;
          for (int s = 0; s < S; ++s) { const double f = expit(mc[2 * s] * powers[p] - mc[2 * s + 1]); acc += log(f / (1.0 - f)); }
          mcE[(size_t)n * P + p] = acc / (double)S;
        }
      }
    }
    std::fill(R.begin(), R.end(), 0.0);
    for (int n = 0; n < N; ++n)
      if (mu[n] != 0.0)
        for (int e = row_ptr[n]; e < row_ptr[n + 1]; ++e) R[col[e]] += mu[n] * lam[e];
    const double thresh = msrmp + spont_rate;                                // :82
    std::vector<double> est, srate(std::max(P, 1));
    for (int m = 0; m < N; ++m) {                                            // :229 strictly sequential
      const int n = order[m];
      const int beg = row_ptr[n], len = row_ptr[n + 1] - beg;
      est.assign(len, 0.0);
      std::fill(srate.begin(), srate.end(), 0.0);
      double total = 0.0;
      for (int i = 0; i < len; ++i) {
        const int e = beg + i, k = col[e];
        if (!mask[k]) continue;
        const double loo = sig * (R[k] - mu[n] * lam[e]);
        const double arg = -2.0 * sig * y[k] * mu[n] + 2.0 * mu[n] * loo + sig * (mu[n] * mu[n] + beta[n] * beta[n]);   // :205-206 (quirk 1)
        est[i] = expit(mcE[(size_t)n * P + grp[e]] - 0.5 * arg);             // :216
        total += est[i];
        srate[grp[e]] += est[i];
      }
      double pava = 1.0;
      if (it > delay_spont_est) {                                            // :221
        for (int p = 0; p < P; ++p) { const int c = cnt[(size_t)n * P + p]; srate[p] /= ((double)c + 1e-4 * (double)(c == 0)); }   // :183
        const double iso = P > 0 ? pava_last(srate.data(), P) : 0.0;
        pava = (iso >= thresh && total >= minimum_spike_count) ? 1.0 : 0.0;  // :220
      }
      for (int i = 0; i < len; ++i) {
        const int e = beg + i;
        const double nl = est[i] * pava;                                     // :224
        R[col[e]] += mu[n] * (nl - lam[e]);
        lam[e] = nl;
      }
    }
    // ---- update_sigma (:238-244)
    {
      double s1 = 0.0, s2 = 0.0, s3 = 0.0;
      for (int k = 0; k < K; ++k) { const double d = y[k] - R[k]; s1 += d * d; }
      for (int n = 0; n < N; ++n) {
        double l1 = 0.0, l2 = 0.0;
        for (int e = row_ptr[n]; e < row_ptr[n + 1]; ++e) { l1 += lam[e]; l2 += lam[e] * lam[e]; }
        s2 += mu[n] * mu[n] * l2; s3 += (mu[n] * mu[n] + beta[n] * beta[n]) * l1;
      }
      shape = shape_prior + 0.5 * (double)K;
      rate = rate_prior + 0.5 * (s1 - s2 + s3);
    }
    update_phi();                                                            // :85
    { uint32_t a[2], b2[2]; split2(key, a, b2); key[0] = b2[0]; key[1] = b2[1]; }   // :251, :304: update_phi returns split(key)[1]
    // ---- estimate_spont_act_soft_thresh (:147-163): loop counter seeded with the OUTER iteration (quirk 4)
    {
      double err = sum_y, pen = penalty;
      std::vector<char> colany(K, 0);
      for (int e = 0; e < nnz; ++e) if (lam[e] >= spont_orthogonality) colany[col[e]] = 1;
      for (int cnt_it = it; cnt_it < max_backtrack_iters && err > tol; ++cnt_it) {
        double e2 = 0.0;
        for (int k = 0; k < K; ++k) {
          const double resid = y[k] - R[k];                                  // :153
          double zz = resid < pen ? 0.0 : resid - pen;                       // :154
          if (zz < 0.0) zz = 0.0;                                            // :155
          if (colany[k]) zz = 0.0;                                           // :156
          zz *= mask[k] ? 1.0 : 0.0;                                         // :157
          z[k] = zz;
          e2 += (resid - zz) * (resid - zz);
        }
        err = e2 / (sum_y2 + 1e-5);                                          // :158
        pen *= scale_factor;
      }
      int nz = 0;
      for (int k = 0; k < K; ++k) nz += z[k] != 0.0;
      spont_rate = (double)nz / (double)K;                                   // :88
    }
  }
  // ---- reconnect_spont_cells (:102-144) + final update_phi (:94-98)
  if (fn_scan) {
    std::vector<int> disc;
    for (int n = 0; n < N; ++n) if (mu[n] == 0.0) disc.push_back(n);         // :103
    while (!disc.empty()) {
      int nzz = 0;
      for (int k = 0; k < K; ++k) nzz += z[k] != 0.0;
      if (!(nzz > minimum_spike_count)) break;                               // :111, :141
      int best = -1, best_cnt = -1;
      for (size_t i = 0; i < disc.size(); ++i) {                             // :112-117 (stim_matrix[n] != 0: every entry of the compressed row)
        int c = 0;
        for (int e = row_ptr[disc[i]]; e < row_ptr[disc[i] + 1]; ++e) c += z[col[e]] != 0.0;
        if (c > best_cnt) { best_cnt = c; best = (int)i; }
      }
      const int focus = disc[best];
      std::vector<double> sr(std::max(P, 1), 0.0);
      double spike_count = 0.0;
      for (int p = 0; p < P; ++p) {                                          // :121-127
        int tot = 0, ev = 0;
        for (int e = row_ptr[focus]; e < row_ptr[focus + 1]; ++e) if (grp[e] == p) { ++tot; ev += z[col[e]] != 0.0; }
        if (tot > 0) { sr[p] = (double)ev / (double)tot; spike_count += ev; }
      }
      const double pv = P > 0 ? isotonic_last_literal(sr.data(), P) : 0.0;   // :128
      if (pv >= msrmp && spike_count >= minimum_spike_count) {               // :130
        std::vector<double> zs;
        for (int e = row_ptr[focus]; e < row_ptr[focus + 1]; ++e) if (z[col[e]] != 0.0) zs.push_back(z[col[e]]);
        const double m = std::accumulate(zs.begin(), zs.end(), 0.0) / (double)zs.size();
        double sq = 0.0;
        for (double v : zs) sq += (v - m) * (v - m);
        mu[focus] = m;                                                       // :134
        beta[focus] = zs.size() > 1 ? sqrt(sq / (double)(zs.size() - 1)) / sqrt((double)zs.size()) : NAN;   // :135 scipy.stats.sem, ddof = 1
        for (int e = row_ptr[focus]; e < row_ptr[focus + 1]; ++e) if (z[col[e]] != 0.0) { lam[e] = 1.0; z[col[e]] = 0.0; }   // :136-137
      }
      disc.erase(disc.begin() + best);                                       // :139
    }
    update_phi();
  }
  memcpy(mu_out, mu.data(), sizeof(double) * N);
  memcpy(beta_out, beta.data(), sizeof(double) * N);
  if (lam_out) {
    memset(lam_out, 0, sizeof(double) * (size_t)N * K);
    for (int n = 0; n < N; ++n)
      for (int e = row_ptr[n]; e < row_ptr[n + 1]; ++e) lam_out[(size_t)n * K + col[e]] = lam[e];
  }
  shape_rate_out[0] = shape; shape_rate_out[1] = rate;
  memcpy(phi_out, phi.data(), sizeof(double) * 2 * N);
  memcpy(phi_cov_out, phi_cov.data(), sizeof(double) * 4 * N);
  memcpy(z_out, z.data(), sizeof(double) * K);
  return 0;
}
