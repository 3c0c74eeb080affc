// Post-sweep phases of one CAVIaR iteration:
//   update_sigma                      reference circuitmap/optimise/caviar.py:238-244
//   update_phi / _laplace_approx      caviar.py:246-316
//   estimate_spont_act_soft_thresh    caviar.py:147-163  (+ spont_rate, caviar.py:88)
#include "cm_internal.h"

// ------------------------------------------------------------------------------------------------
// Statistics of the new lam, one launch, two roles (256-thread blocks):
//   blocks [0, nbR): one wave per row  -> rs[n] = {sum lam, sum lam^2, sum y*lam, sum lam(1-lam) or -1 if the row is all zero}
//   blocks [nbR, ..): one thread per trial -> R_k = sum_n mu_n lam_nk (= lam.T @ mu, ascending neuron order),
//                     colany_k = any_n lam_nk >= spont_orthogonality
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_stats(const double* __restrict__ lam, const int* __restrict__ row_ptr, const int* __restrict__ col,
                                                const double* __restrict__ y, const double* __restrict__ mu, const int* __restrict__ col_ptr,
                                                const int* __restrict__ csc_row, const int* __restrict__ csc_pos, int N, int K, int nbR,
                                                double orth, double* __restrict__ rs, double* __restrict__ R, uint8_t* __restrict__ colany) {
  if ((int)blockIdx.x < nbR) {
    const int n = (blockIdx.x * blockDim.x + threadIdx.x) / CM_WAVE;
    const int lane = threadIdx.x % CM_WAVE;
    if (n >= N) return;
    double s1 = 0.0, s2 = 0.0, sy = 0.0, d = 0.0;
    int any = 0;
    for (int e = row_ptr[n] + lane; e < row_ptr[n + 1]; e += CM_WAVE) {
      const double l = lam[e];
      s1 += l;
      s2 = fma(l, l, s2);
      sy = fma(y[col[e]], l, sy);
      d = fma(l, 1.0 - l, d);
      any |= (l != 0.0);
    }
    s1 = cm_wave_sum(s1); s2 = cm_wave_sum(s2); sy = cm_wave_sum(sy); d = cm_wave_sum(d);
    any = __any(any);
    if (lane == 0) {
      rs[(size_t)n * 4 + 0] = s1;
      rs[(size_t)n * 4 + 1] = s2;
      rs[(size_t)n * 4 + 2] = sy;
      rs[(size_t)n * 4 + 3] = any ? d : -1.0;   // sum lam(1-lam) for active rows, -1 marks an all-zero row
    }
  } else {
    const int k = ((int)blockIdx.x - nbR) * blockDim.x + threadIdx.x;
    if (k >= K) return;
    double s = 0.0;
    int any = 0;
    for (int ce = col_ptr[k]; ce < col_ptr[k + 1]; ++ce) {
      const double l = lam[csc_pos[ce]];
      s = fma(mu[csc_row[ce]], l, s);
      any |= (l >= orth);
    }
    R[k] = s;
    colany[k] = (uint8_t)any;
  }
}

// block-wide sum (1024 threads), result valid in every thread
__device__ __forceinline__ double block_sum(double v, double* sh) {
  v = cm_wave_sum(v);
  __syncthreads();
  if (threadIdx.x % CM_WAVE == 0) sh[threadIdx.x / CM_WAVE] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < (int)(blockDim.x / CM_WAVE); ++w) t += sh[w];
  return t;
}

// ------------------------------------------------------------------------------------------------
// Everything of an iteration that is a single-workgroup reduction, in one kernel (1024 threads):
//   1. ordered compaction of the active rows (lam row != 0) for the next block_update_mu, and the size of that set
//      posted straight into host-mapped pinned memory (mailbox) so the host can size the next solve while 2-3 still run;
//   2. update_sigma: rate = rate0 + 1/2 ( sum_k (y_k - R_k)^2 - sum_n mu_n^2 sum_k lam_nk^2 + sum_n (mu_n^2 + beta_n^2) sum_k lam_nk )
//   3. z: <= (max_iters - it) penalty steps pen_j = penalty * scale^j while err > tol (the loop counter is seeded with
//      the OUTER iteration index and err with sum(y), caviar.py:86; quirk 4: nothing happens once it >= max_iters),
//      then spont_rate = mean(z != 0) (caviar.py:88).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_post_single(const double* __restrict__ y, const double* __restrict__ R,
                                                       const uint8_t* __restrict__ colany, const uint8_t* __restrict__ mask,
                                                       const double* __restrict__ mu, const double* __restrict__ beta,
                                                       const double* __restrict__ rs, const double* __restrict__ sum_y, int N, int K,
                                                       int it, int max_iters, double penalty, double scale_factor, double tol,
                                                       int do_sigma, int seq, int* __restrict__ act_idx, int* __restrict__ act_inv,
                                                       int* __restrict__ n_act, int* mailbox, double* __restrict__ z,
                                                       double* __restrict__ scal) {
  __shared__ double sh[16];
  __shared__ int wave_cnt[16];
  __shared__ int base;
  // ---- 1. active set
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  {
    const int lane = threadIdx.x % CM_WAVE, w = threadIdx.x / CM_WAVE;
    for (int n0 = 0; n0 < N; n0 += 1024) {
      const int n = n0 + threadIdx.x;
      const bool a = n < N && rs[(size_t)n * 4 + 3] >= 0.0;
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
      if (threadIdx.x == 0) { int t = 0; for (int i = 0; i < 16; ++i) t += wave_cnt[i]; base += t; }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      n_act[0] = base;
      if (mailbox) {                                  // host-mapped pinned memory; system-scope release on the sequence word
        mailbox[0] = base;
        mailbox[1] = n_act[1];
        __hip_atomic_store(&mailbox[2], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  if (!do_sigma) return;
  // ---- 2. update_sigma
  double a = 0.0;
  for (int k = threadIdx.x; k < K; k += blockDim.x) { const double r = y[k] - R[k]; a = fma(r, r, a); }
  double b = 0.0, c = 0.0;
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    const double m = mu[n], be = beta[n];
    b = fma(m * m, rs[(size_t)n * 4 + 1], b);
    c = fma(m * m + be * be, rs[(size_t)n * 4 + 0], c);
  }
  a = block_sum(a, sh); b = block_sum(b, sh); c = block_sum(c, sh);
  if (threadIdx.x == 0) {
    scal[CM_S_SHAPE] = scal[CM_S_SHAPE_PRIOR] + (double)K / 2.0;
    scal[CM_S_RATE] = scal[CM_S_RATE_PRIOR] + 0.5 * (a - b + c);
  }
  // ---- 3. estimate_spont_act_soft_thresh
  double err = sum_y[0];
  const double ysq = sum_y[1] + 1e-5;
  double pen = penalty;
  int itc = it;
  while (itc < max_iters && err > tol) {
    double e = 0.0;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
      const double resid = y[k] - R[k];
      double zz = resid < pen ? 0.0 : resid - pen;
      zz = zz < 0.0 ? 0.0 : zz;
      zz = colany[k] ? 0.0 : zz;
      zz *= (double)mask[k];
      z[k] = zz;
      const double d = resid - zz;
      e = fma(d, d, e);
    }
    err = block_sum(e, sh) / ysq;
    ++itc;
    pen *= scale_factor;
  }
  __syncthreads();
  double cnt = 0.0;
  for (int k = threadIdx.x; k < K; k += blockDim.x) cnt += (z[k] != 0.0) ? 1.0 : 0.0;
  cnt = block_sum(cnt, sh);
  if (threadIdx.x == 0) scal[CM_S_SPONT_RATE] = cnt / (double)K;
}

// Row + column statistics of the current lam.  The caller follows with cm_launch_post_single().
int cm_launch_stats(cm_ctx* ctx, cm_problem* p, double spont_orthogonality) {
  const int nbR = (p->N + 3) / 4, nbC = (p->K + 255) / 256;
  hipLaunchKernelGGL(k_stats, dim3(nbR + nbC), dim3(256), 0, ctx->stream, p->lam, p->row_ptr, p->col, p->y, p->mu, p->col_ptr, p->csc_row,
                     p->csc_pos, p->N, p->K, nbR, spont_orthogonality, p->rs, p->R, p->colany);
  CM_HIP(hipGetLastError());
  return CM_OK;
}

// Compaction (+ mailbox post) and, if do_sigma, update_sigma / z / spont_rate.  `it` < 0 means "state init".
int cm_launch_post_single(cm_ctx* ctx, cm_problem* p, int it, const cm_options* o, int do_sigma) {
  hipStream_t st = ctx->stream;
  const int seq = ++ctx->mail_seq;
  hipLaunchKernelGGL(k_post_single, dim3(1), dim3(1024), 0, st, p->y, p->R, p->colany, p->mask, p->mu, p->beta, p->rs, p->sum_y, p->N, p->K,
                     it, o ? o->max_backtrack_iters : 0, o ? o->penalty : 0.0, o ? o->scale_factor : 0.0, o ? o->tol : 0.0, do_sigma, seq,
                     p->act_idx, p->act_inv, p->n_act, ctx->use_mailbox ? ctx->d_mailbox : (int*)nullptr, p->z, p->scal);
  CM_HIP(hipGetLastError());
  if (!ctx->use_mailbox) {
    CM_HIP(hipMemcpyAsync(ctx->h_pinned, p->n_act, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
    CM_HIP(hipEventRecord(ctx->ev[7], st));
  }
  return CM_OK;
}

// Host side of the mailbox: wait until the kernel launched by the last cm_launch_post_single() has posted the active-set size.
int cm_wait_active(cm_ctx* ctx, cm_problem* p) {
  (void)p;
  if (!ctx->use_mailbox) { CM_HIP(hipEventSynchronize(ctx->ev[7])); return CM_OK; }
  volatile int32_t* mb = ctx->h_pinned;
  const int seq = ctx->mail_seq;
  unsigned long long spins = 0;
  while (__atomic_load_n(&mb[2], __ATOMIC_ACQUIRE) != seq) {
    if ((++spins & 0xFFFFull) == 0) {                      // every 64k polls make sure the stream is still alive
      hipError_t q = hipStreamQuery(ctx->stream);
      if (q != hipErrorNotReady && q != hipSuccess) { cm_set_error("stream failed while waiting for the active set: %s", hipGetErrorString(q)); return CM_ERR_HIP; }
      if (q == hipSuccess && __atomic_load_n(&mb[2], __ATOMIC_ACQUIRE) != seq) {
        cm_set_error("active-set mailbox was not posted (seq %d, have %d)", seq, (int)mb[2]);
        return CM_ERR_STATE;
      }
    }
  }
  return CM_OK;
}

// ------------------------------------------------------------------------------------------------
// update_phi: one wavefront per neuron.  The K-sums of the Newton iteration collapse onto per-stimulus-value
// sufficient statistics because I[n,:] takes <= V distinct values; lane g owns value g.  nan_to_num is applied per
// ELEMENT in the reference objective (quirk 11) so elements are classed by y == 0, y == 1, 0 < y < 1.
// ------------------------------------------------------------------------------------------------
struct Phi2 { double a, b; };

__device__ __forceinline__ void inv2x2(double h11, double h12, double h21, double h22, double o[4]) {
  // LU with partial pivoting on the 2x2 (what LAPACK getrf/getri do for jnp.linalg.inv)
  double a = h11, b = h12, c = h21, d = h22;
  bool sw = fabs(c) > fabs(a);
  if (sw) { double t = a; a = c; c = t; t = b; b = d; d = t; }
  const double l = c / a;
  const double u22 = d - l * b;
  // solve for unit vectors (with row swap applied to the right-hand side)
  // column 0 of inverse: rhs = P e0
  double r0 = sw ? 0.0 : 1.0, r1 = sw ? 1.0 : 0.0;
  double y1 = r1 - l * r0;
  double x1 = y1 / u22;
  double x0 = (r0 - b * x1) / a;
  o[0] = x0; o[2] = x1;
  r0 = sw ? 1.0 : 0.0; r1 = sw ? 0.0 : 1.0;
  y1 = r1 - l * r0;
  x1 = y1 / u22;
  x0 = (r0 - b * x1) / a;
  o[1] = x0; o[3] = x1;
}

template <int SEG>
__device__ __forceinline__ double phi_nll(double p0, double p1, double val, double n0, double n1, double smid, double nmid,
                                          bool lane_on, double pr0, double pr1, const double prec[4], double t) {
  double tot = 0.0;
  if (lane_on) {
    const double s = cm_sigmoid(p0 * val - p1);
    const double ls = log(s), l1s = log(1.0 - s);
    if (n0 > 0.0) tot += n0 * cm_nan_to_num(0.0 * ls + 1.0 * l1s);
    if (n1 > 0.0) tot += n1 * cm_nan_to_num(1.0 * ls + 0.0 * l1s);
    if (nmid > 0.0) {
      if (isfinite(ls) && isfinite(l1s)) tot += smid * ls + (nmid - smid) * l1s;
      else tot += nmid * cm_nan_to_num(0.5 * ls + 0.5 * l1s);
    }
  }
  tot = cm_seg_sum<SEG>(tot);
  const double d0 = p0 - pr0, d1 = p1 - pr1;
  const double quad = d0 * (prec[0] * d0 + prec[1] * d1) + d1 * (prec[2] * d0 + prec[3] * d1);
  return -tot - (log(p0) + log(p1)) / t + 0.5 * quad;
}

// One wavefront per neuron.  SEG = lanes per evaluation point (power of two >= V; lane g of a segment owns stimulus
// value g) and E = 64 / SEG evaluation points per wave: in one pass segment 0 evaluates the objective at phi (the
// right-hand side of the Armijo test) while segments 1..E-1 evaluate it at phi + 2^-b v for consecutive b, so the
// reference's sequential backtracking (caviar.py:259-268: accept the first b whose test passes, at most 40 halvings)
// collapses into one parallel evaluation per Newton step in the common case.
template <int SEG>
__global__ __launch_bounds__(256) void k_update_phi(const double* __restrict__ lam, const int* __restrict__ row_ptr,
                                                     const uint8_t* __restrict__ grp, const int* __restrict__ rowcnt,
                                                     const double* __restrict__ vals, int V, int N,
                                                     const double* __restrict__ phi_prior, const double* __restrict__ phi_cov_prior,
                                                     double* __restrict__ phi, double* __restrict__ phi_cov) {
  constexpr int E = CM_WAVE / SEG;
  const int n = (blockIdx.x * blockDim.x + threadIdx.x) / CM_WAVE;
  const int lane = threadIdx.x % CM_WAVE;
  if (n >= N) return;
  const int g = lane % SEG, ev = lane / SEG;
  // sufficient statistics of value g: segment `ev` scans entries ev, ev + E, ...; partial sums are then combined
  // across the segments (lanes at stride SEG)
  const bool on = g < V;
  const double val = on ? vals[g] : 0.0;
  const double cnt = on ? (double)rowcnt[(size_t)n * CM_MAX_VALS + g] : 0.0;
  double n1 = 0.0, nmid = 0.0, smid = 0.0;
  if (on && val > 0.0) {                       // untargeted values carry lam == 0 on every trial
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

  const double pr0 = phi_prior[2 * n], pr1 = phi_prior[2 * n + 1];
  double prec[4];
  inv2x2(phi_cov_prior[4 * n], phi_cov_prior[4 * n + 1], phi_cov_prior[4 * n + 2], phi_cov_prior[4 * n + 3], prec);   // caviar.py:306
  const double t = 1e1, alpha = 0.25;
  double p0 = pr0, p1 = pr1;                     // caviar.py:305: restart from the prior every call (quirk 3)
  double cov[4] = {0, 0, 0, 0};
  for (int stepi = 0; stepi < 10; ++stepi) {     // caviar.py:308
    double j1 = 0.0, j2 = 0.0, h11 = 0.0, h12 = 0.0, h22 = 0.0;
    if (on) {
      const double f = cm_sigmoid(p0 * val - p1);
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
      const double fv = phi_nll<SEG>(q0, q1, val, n0, n1, smid, nmid, on, pr0, pr1, prec, t);
      if (first) f0 = cm_readlane_f64(fv, 0);
      const double rhs = f0 + alpha * st * Jv;
      const bool ok = !is_f0 && b <= 39 && !(isnan(fv) || fv > rhs);
      const unsigned long long m = __ballot(ok);
      if (m) accepted = (first ? base - 1 : base) + (int)(__ffsll((long long)m) - 1) / SEG;
      base += first ? E - 1 : E;
      first = false;
    }
    if (accepted < 0) accepted = 40;
    const double step = ldexp(1.0, -accepted);
    p0 += step * v0; p1 += step * v1;                                             // caviar.py:301
  }
  if (lane == 0) {
    phi[2 * n] = p0; phi[2 * n + 1] = p1;
    phi_cov[4 * n] = cov[0]; phi_cov[4 * n + 1] = cov[1]; phi_cov[4 * n + 2] = cov[2]; phi_cov[4 * n + 3] = cov[3];
  }
}

template <int SEG>
static void launch_phi(cm_problem* p, hipStream_t st) {
  hipLaunchKernelGGL(k_update_phi<SEG>, dim3((unsigned)((p->N + 3) / 4)), dim3(256), 0, st, p->lam, p->row_ptr, p->grp,
                     p->rowcnt, p->vals, p->V, p->N, p->phi_prior, p->phi_cov_prior, p->phi, p->phi_cov);
}

int cm_launch_update_phi(cm_ctx* ctx, cm_problem* p, hipStream_t st) {
  (void)ctx;
  const int V = p->V;
  if (V <= 1) launch_phi<1>(p, st);
  else if (V <= 2) launch_phi<2>(p, st);
  else if (V <= 4) launch_phi<4>(p, st);
  else if (V <= 8) launch_phi<8>(p, st);
  else if (V <= 16) launch_phi<16>(p, st);
  else if (V <= 32) launch_phi<32>(p, st);
  else launch_phi<64>(p, st);
  CM_HIP(hipGetLastError());
  return CM_OK;
}

