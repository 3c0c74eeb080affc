// Post-sweep phases of one CAVIaR iteration:
//   update_sigma                      reference circuitmap/optimise/caviar.py:238-244
//   update_phi / _laplace_approx      caviar.py:246-316
//   estimate_spont_act_soft_thresh    caviar.py:147-163  (+ spont_rate, caviar.py:88)
#include "cm_bodies.h"

// ------------------------------------------------------------------------------------------------
// Statistics of the new lam, one launch, two roles (256-thread blocks):
//   blocks [0, nbR): one wave per row  -> rs[n] = {sum lam, sum lam^2, sum y*lam, sum lam(1-lam) or -1 if the row is all zero}
//   blocks [nbR, ..): 16 lanes per trial -> R_k = sum_n mu_n lam_nk (= lam.T @ mu),
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
    cm_col_dot16(((int)blockIdx.x - nbR) * blockDim.x + threadIdx.x, K, mu, lam, col_ptr, csc_row, csc_pos, orth, R, colany);
  }
}

__global__ __launch_bounds__(1024) void k_post_single(const double* __restrict__ y, const double* __restrict__ R,
                                                       const uint8_t* __restrict__ colany, const uint8_t* __restrict__ mask,
                                                       const double* __restrict__ mu, const double* __restrict__ beta,
                                                       const double* __restrict__ rs, const double* __restrict__ sum_y, int N, int K,
                                                       int it, int max_iters, double penalty, double scale_factor, double tol,
                                                       int do_sigma, int seq, int* __restrict__ act_idx, int* __restrict__ act_inv,
                                                       int* __restrict__ n_act, int* mailbox, double* __restrict__ z,
                                                       double* __restrict__ scal) {
  post_single_body(y, R, colany, mask, mu, beta, rs, sum_y, N, K, it, max_iters, penalty, scale_factor, tol, do_sigma, seq, act_idx,
                   act_inv, n_act, mailbox, z, scal);
}

// Row + column statistics of the current lam.  The caller follows with cm_launch_post_single().
int cm_launch_stats(cm_ctx* ctx, cm_problem* p, double spont_orthogonality) {
  const int nbR = (p->N + 3) / 4, nbC = (p->K * 16 + 255) / 256;     // one wave per row; 16 lanes per trial column
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
    CM_HIP(hipMemcpyAsync(ctx->h_pinned + 8, p->n_act, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
    CM_HIP(hipEventRecord(ctx->ev[7], st));
  }
  return CM_OK;
}

// Host side of the mailbox: wait until the kernel launched by the last post has published the active-set size; the decoded
// values land in ctx->mb_n / ctx->mb_flag.
int cm_wait_active(cm_ctx* ctx, cm_problem* p) {
  (void)p;
  if (!ctx->use_mailbox) {
    CM_HIP(hipEventSynchronize(ctx->ev[7]));
    ctx->mb_n = ctx->h_pinned[8]; ctx->mb_flag = ctx->h_pinned[9];
    return CM_OK;
  }
  volatile unsigned long long* mb = (volatile unsigned long long*)ctx->h_pinned;
  const unsigned seq = (unsigned)ctx->mail_seq;
  unsigned long long spins = 0, w;
  while ((unsigned)((w = __atomic_load_n(mb, __ATOMIC_ACQUIRE)) >> 32) != seq) {
    if ((++spins & 0xFFFFull) == 0) {                      // every 64k polls make sure the stream is still alive
      hipError_t q = hipStreamQuery(ctx->stream);
      if (q != hipErrorNotReady && q != hipSuccess) { cm_set_error("stream failed while waiting for the active set: %s", hipGetErrorString(q)); return CM_ERR_HIP; }
      if (q == hipSuccess && (unsigned)(__atomic_load_n(mb, __ATOMIC_ACQUIRE) >> 32) != seq) {
        cm_set_error("active-set mailbox was not posted (seq %u)", seq);
        return CM_ERR_STATE;
      }
    }
  }
  ctx->mb_n = (int)(w & 0x7FFFFFFFull);
  ctx->mb_flag = (int)((w >> 31) & 1ull);
  return CM_OK;
}

// update_phi stand-alone (final update after the false-negative scan, and the per-phase test hook): one wave per neuron
template <int SEG>
__global__ __launch_bounds__(256) void k_update_phi(const double* __restrict__ lam, const int* __restrict__ row_ptr,
                                                     const uint8_t* __restrict__ grp, const int* __restrict__ rowcnt,
                                                     const double* __restrict__ vals, int V, int N,
                                                     const double* __restrict__ phi_prior, const double* __restrict__ phi_cov_prior,
                                                     double* __restrict__ phi, double* __restrict__ phi_cov) {
  const int n = (blockIdx.x * blockDim.x + threadIdx.x) / CM_WAVE;
  if (n >= N) return;
  double out[4];
  phi_body<SEG>(n, threadIdx.x % CM_WAVE, lam, row_ptr, grp, rowcnt, vals, V, phi_prior, phi_cov_prior, phi, phi_cov, out);
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

