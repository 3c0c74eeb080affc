// The middle of a CAVIaR iteration as ONE launch on ONE stream (256-thread workgroups, two roles):
//
//   block 0       everything that is a single-workgroup reduction, back to back:
//                 active-set compaction -> size posted to the host mailbox -> update_sigma (caviar.py:238-244) ->
//                 soft threshold + spont_rate (:147-163, :88) -> block_update_mu of the NEXT iteration (:166-172) when the
//                 active set fits the in-LDS solver (otherwise the host, which has the size from the mailbox, launches
//                 the tiled factorisation after this kernel);
//   blocks 1..    one wavefront per neuron: update_phi (:246-316) immediately followed by the Monte-Carlo logit of the
//                 NEXT iteration for that neuron (:209-215), which needs only that neuron's new phi.
//
// Neither role reads what the other writes (sigma/z/mu/beta vs phi/mcE), so they overlap inside the launch without
// streams or events; the launch is as long as its longer role (~phi + MC logit).
#include <stdlib.h>

#include "cm_bodies.h"

#define CM_ITER_THREADS 256                     // 4 waves = one per SIMD; with <= ~50 KB of LDS the blocks spread one per CU
#define CM_ITER_WAVES (CM_ITER_THREADS / CM_WAVE)

struct IterArgs {
  // block 0: post
  const double* y; const double* R; const uint8_t* colany; const uint8_t* mask; const double* rs; const double* sum_y;
  int N, K, it, max_iters; double penalty, scale_factor, tol; int do_sigma, seq;
  int* act_idx; int* act_inv; int* n_act; int* mailbox; double* z; double* scal;
  // block 0: next block_update_mu
  const double* lam; const int* row_ptr; const int* col; const double* mu_prior; const double* beta_prior;
  int small_cap, max_row_len, do_solve, stage_ok; double* mu; double* beta;
  // blocks 1..: update_phi + next MC logit
  const uint8_t* grp; const int* rowcnt; const double* vals; int V; const double* phi_prior; const double* phi_cov_prior;
  double* phi; double* phi_cov; int do_phi, do_mc, S, roles;
  const uint32_t* keys_next; const double* uexp_next; const double* wtab_next; const int* inv_next; double* mcE;
};

template <int SEG, bool FULLMC>
__global__ __launch_bounds__(CM_ITER_THREADS) void k_iter_mid(IterArgs a) {
  extern __shared__ double sm[];
  if (blockIdx.x == 0) {
    if (!(a.roles & 1)) return;
    const int n = post_single_body(a.y, a.R, a.colany, a.mask, a.mu, a.beta, a.rs, a.sum_y, a.N, a.K, a.it, a.max_iters, a.penalty,
                                   a.scale_factor, a.tol, a.do_sigma, a.seq, a.act_idx, a.act_inv, a.n_act, a.mailbox, a.z, a.scal,
                                   a.stage_ok ? sm : nullptr);
    if (a.do_solve && n <= a.small_cap) {
      __syncthreads();
      solve_small_body(sm, a.lam, a.row_ptr, a.col, a.act_idx, a.rs, a.scal, a.mu_prior, a.beta_prior, n, a.N, n * a.max_row_len, a.mu,
                       a.beta, a.n_act + 1);
    }
    return;
  }
  if (!(a.roles & 2)) return;
  const int wave = threadIdx.x / CM_WAVE, lane = threadIdx.x % CM_WAVE;
  const int n = ((int)blockIdx.x - 1) * CM_ITER_WAVES + wave;
  if (n >= a.N) return;
  double pc[4];
  if (a.do_phi) {
    phi_body<SEG>(n, lane, a.lam, a.row_ptr, a.grp, a.rowcnt, a.vals, a.V, a.phi_prior, a.phi_cov_prior, a.phi, a.phi_cov, pc);
  } else {
    pc[0] = a.phi[2 * n]; pc[1] = a.phi[2 * n + 1]; pc[2] = a.phi_cov[4 * n]; pc[3] = a.phi_cov[4 * n + 3];
  }
  if (a.do_mc) {
    const int m = a.inv_next[n];
    const size_t off = (size_t)m * 2 * a.S;
    mc_logit_body<FULLMC>(n, m, lane, pc[0], pc[1], pc[2], pc[3], a.keys_next, a.uexp_next ? a.uexp_next + off : nullptr, a.wtab_next + off,
                  a.vals, a.V, a.S, sm + (size_t)wave * 2 * a.S, a.mcE);
  }
}

// largest active set the fused kernel may solve in LDS (0 = never): A (n x (n+1)) + staged rows must fit beside the statics
int cm_small_cap(const cm_problem* p) {
  int cap = 0;
  for (int n = 1; n <= 64; ++n)                 // 64: steady-state active sets fit, and the LDS footprint stays ~50 KB per block
    if (small_lds_bytes(n, n * p->max_row_len) <= 56 * 1024) cap = n; else break;
  return cap;
}

template <int SEG, bool FULLMC>
static int launch_iter2(cm_ctx* ctx, const IterArgs& a, size_t lds, int blocks) {
  const int slot = CM_LDS_ITER + 2 * (31 - __builtin_clz((unsigned)SEG)) + (FULLMC ? 1 : 0);          // SEG = 1, 2, 4 .. 64
  CM_TRY(cm_raise_lds(ctx, slot, reinterpret_cast<const void*>(k_iter_mid<SEG, FULLMC>), lds, 48 * 1024));
  hipLaunchKernelGGL((k_iter_mid<SEG, FULLMC>), dim3(blocks), dim3(CM_ITER_THREADS), lds, ctx->stream, a);
  CM_HIP(hipGetLastError());
  return CM_OK;
}
template <int SEG>
static int launch_iter(cm_ctx* ctx, const IterArgs& a, size_t lds, int blocks) {
  return a.it < 0 ? launch_iter2<SEG, true>(ctx, a, lds, blocks) : launch_iter2<SEG, false>(ctx, a, lds, blocks);
}

// it = -1: state init (no sigma / phi; MC logit and solve for iteration 0).  Otherwise the middle of iteration `it`.
int cm_launch_iter_mid(cm_ctx* ctx, cm_problem* p, int it, int iters, int S, int rng_mode, const cm_options* o, int do_sigma, int do_phi) {
  IterArgs a;
  const int N = p->N, nxt = it + 1;
  a.y = p->y; a.R = p->R; a.colany = p->colany; a.mask = p->mask; a.rs = p->rs; a.sum_y = p->sum_y;
  a.N = N; a.K = p->K; a.it = it; a.max_iters = o->max_backtrack_iters; a.penalty = o->penalty; a.scale_factor = o->scale_factor;
  a.tol = o->tol; a.do_sigma = do_sigma; a.seq = ++ctx->mail_seq;
  a.act_idx = p->act_idx; a.act_inv = p->act_inv; a.n_act = p->n_act; a.mailbox = ctx->d_mailbox; a.z = p->z; a.scal = p->scal;
  a.lam = p->lam; a.row_ptr = p->row_ptr; a.col = p->col; a.mu_prior = p->mu_prior; a.beta_prior = p->beta_prior;
  a.small_cap = p->small_cap; a.max_row_len = p->max_row_len; a.do_solve = nxt < iters; a.mu = p->mu; a.beta = p->beta;
  a.grp = p->grp; a.rowcnt = p->rowcnt; a.vals = p->vals; a.V = p->V; a.phi_prior = p->phi_prior; a.phi_cov_prior = p->phi_cov_prior;
  a.phi = p->phi; a.phi_cov = p->phi_cov; a.do_phi = do_phi; a.do_mc = nxt < iters; a.S = S;
  const int nx = a.do_mc ? nxt : 0;
  a.keys_next = (rng_mode == CM_RNG_EXPLICIT) ? nullptr : p->keys + (size_t)nx * (N + 3) * 2;
  a.uexp_next = (rng_mode == CM_RNG_EXPLICIT) ? p->uexp + (size_t)nx * N * 2 * S : nullptr;
  a.wtab_next = p->wtab + (size_t)nx * N * 2 * S;
  a.inv_next = p->inv_all + (size_t)nx * N;
  a.mcE = p->mcE;
  size_t lds = (size_t)CM_ITER_WAVES * 2 * S * sizeof(double);
  if (p->small_cap > 0) lds = std::max(lds, small_lds_bytes(p->small_cap, p->small_cap * p->max_row_len));
  a.stage_ok = 0;
  if ((size_t)p->K * 9 + 16 <= 56 * 1024) { lds = std::max(lds, (size_t)p->K * 9 + 16); a.stage_ok = 1; }   // residual + flag per trial
  const int blocks = 1 + ((do_phi || a.do_mc) ? (N + CM_ITER_WAVES - 1) / CM_ITER_WAVES : 0);
  const int V = p->V;
  static const bool split = getenv("CAVIAR_HIP_SPLIT_ITER") != nullptr;         // diagnostic: one launch per role (profiling only)
  a.roles = 3;
  if (split && it >= 0) {
    a.roles = 2;
    if (V <= 4) { CM_TRY(launch_iter<4>(ctx, a, lds, blocks)); } else { CM_TRY(launch_iter<64>(ctx, a, lds, blocks)); }
    a.roles = 1;
  }
  if (V <= 1) return launch_iter<1>(ctx, a, lds, blocks);
  if (V <= 2) return launch_iter<2>(ctx, a, lds, blocks);
  if (V <= 4) return launch_iter<4>(ctx, a, lds, blocks);
  if (V <= 8) return launch_iter<8>(ctx, a, lds, blocks);
  if (V <= 16) return launch_iter<16>(ctx, a, lds, blocks);
  if (V <= 32) return launch_iter<32>(ctx, a, lds, blocks);
  return launch_iter<64>(ctx, a, lds, blocks);
}
