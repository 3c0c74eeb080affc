// The middle of a CAVIaR iteration as ONE launch on ONE stream, two roles:
//
//   block 0       single-workgroup reductions, back to back (on its first CM_POST_THREADS threads):
//                 active-set compaction -> size posted to the host mailbox -> update_sigma (caviar.py:238-244) ->
//                 block_update_mu of the NEXT iteration (:166-172) when the active set fits the in-LDS solver (otherwise the
//                 host, which has the size from the mailbox, launches the tiled factorisation after this kernel);
//   block 1       soft threshold + spont_rate (:147-163, :88): shares no data with block 0, so it runs beside it;
//   blocks 2..    one wavefront per neuron: update_phi (:246-316) immediately followed by the Monte-Carlo logit of the
//                 NEXT iteration for that neuron (:209-215), which needs only that neuron's new phi.
//
// Neither role reads what the other writes (sigma/z/mu/beta vs phi/mcE), so they overlap inside the launch without
// streams or events; the launch is as long as its longer role (~phi + MC logit).
//
// 256-thread workgroups (4 waves = one per SIMD; for one circuit the blocks spread one per CU: shortest latency).  A batch of
// circuits (blockIdx.y) keeps that shape: measured 256 = 512 > 1024 threads per workgroup (a workgroup holds its CU slot until its
// slowest wave -- a row with a full Newton iteration -- is done; the dynamic LDS, ~18 KB, does not limit occupancy).
#include <stdlib.h>

#include "cm_bodies.h"

#define CM_ITER_THREADS 256
#define CM_ITER_THREADS_BATCH 256

struct IterDyn {     // what changes from iteration to iteration (common to all circuits of a batch)
  int it, max_iters; double penalty, scale_factor, tol; int do_sigma, seq, do_solve, stage_ok, do_phi, do_mc, S, roles, nx, explicit_rng, skip_dead;
  int rpw;             // rows per wavefront of the per-neuron role (phi_pair): 2 = a row per 32-lane half (throughput: the early iterations of a batch), 1 = lower half only (latency: one circuit, listed rows)
  int mc_screen, dead_sep; double msrmp, min_count;   // k_mc_dead: the next sweep is gated -> all-zero rows whose gate cannot pass need no Monte-Carlo logit
  long long* dbg;      // diagnostic (CAVIAR_HIP_TRACE_BLOCK0): device clock at the phase boundaries of block 0, 8 words per launch
  // block 0, behind its in-LDS solve: the chain tables of the NEXT sweep (seq_scan_body) -- granule and packed-row limit of that sweep's
  // chunking for the single-consumer kernel [0] and for the multi-wave class of the set [1] (one circuit: chosen by the active-set size as
  // cm_launch_sweep chooses, cm_sweep_variant)
  int scan_on, scan_gran[2], scan_pack[2], scan_min_rows;
};

// an all-zero row takes the per-run constants of its neuron: phi, phi_cov (written by lane 0) and the truncation points c0, c1
__device__ __forceinline__ void dead_phi_copy(const cm_pdev& p, int n, int lane, double pc[4], double* c0, double* c1) {
  const double* __restrict__ pd = p.phi_dead + 2 * (size_t)n;
  const double* __restrict__ cd = p.phi_cov_dead + 4 * (size_t)n;
  const double* __restrict__ cc = p.phi_c_dead + 2 * (size_t)n;
  pc[0] = pd[0]; pc[1] = pd[1];
  const double c01 = cd[1], c10 = cd[2];
  pc[2] = cd[0]; pc[3] = cd[3];
  *c0 = cc[0]; *c1 = cc[1];
  if (lane == 0) {
    p.phi[2 * n] = pc[0]; p.phi[2 * n + 1] = pc[1];
    p.phi_cov[4 * n] = pc[2]; p.phi_cov[4 * n + 1] = c01; p.phi_cov[4 * n + 2] = c10; p.phi_cov[4 * n + 3] = pc[3];
  }
}

// The exact Monte-Carlo logit (caviar.py:209-215) of TWO rows, one per 32-lane half (`need`, n, the row's phi = (mean0, mean1), variance diagonal
// (sd0, sd1: used as standard deviations, quirk 2) and truncation points c = ndtr(-mean / sdev) are per half): the per-row overhead (constants,
// address arithmetic, the reductions' instruction streams, the division) is issued once for two rows.  Every row's sums are bitwise those of
// mc_logit_body (one row per wavefront): there lane l accumulates the samples l, l + 64 and cm_wave_sum reduces the 64 lanes as four 16-lane
// rows, (r0 + r1) + (r2 + r3); here physical lane j of a half carries the two virtual lanes j (samples j, j + 64) and j + 32 (samples j + 32,
// j + 96) in separate accumulators, the 16-lane reductions run on both sets, and the four row sums are combined in the same order.  A lane's
// samples never leave its registers: no LDS.  At most four powers (gi / vk: their value codes and values) and 128 samples: the callers take the
// general body otherwise.
__device__ __forceinline__ void mc_logit_pair(const cm_pdev& p, bool need, int n, double mean0, double mean1, double sd0, double sd1, double c0, double c1,
                                              int nx, int S, int P, const int (&gi)[4], const double (&vk)[4], const double* etab, int lane) {
  const int j = lane & 31;
  const int N = p.N;
  {
    const double2* __restrict__ wu = (const double2*)p.wtab + ((size_t)nx * N + (need ? n : 0)) * 2 * S;
    double2 t[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int s_ = j + 32 * (q >> 1) + 64 * (q & 1);               // slots: j, j + 64 (virtual lane j), j + 32, j + 96 (virtual lane j + 32)
#pragma unroll
      for (int comp = 0; comp < 2; ++comp) t[q][comp] = (need && s_ < S) ? wu[2 * s_ + comp] : make_double2(0.0, 0.0);
    }
    const bool tiny0 = c0 < 0x1p-106, tiny1 = c1 < 0x1p-106;
    double acc[4][2] = {{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int s_ = j + 32 * (q >> 1) + 64 * (q & 1);
      if (need && s_ < S) {
        // the draw of mc_logit_body<false> (caviar.py:212): the tabulated quantile whenever arg = c + u (1 - c) is u bit for bit
        double qq[2];
#pragma unroll
        for (int comp = 0; comp < 2; ++comp) {
          const double c = comp ? c1 : c0;
          const bool tiny = comp ? tiny1 : tiny0;
          const double2 tt = t[q][comp];
          if (tiny) qq[comp] = (tt.x == -INFINITY && c != 0.0) ? cm_quantile_slow(c) : tt.x;
          else { const double arg = c + tt.y * (1.0 - c); qq[comp] = (arg == tt.y) ? tt.x : cm_quantile_slow(arg); }
        }
        const double m0 = qq[0] * sd0 + mean0, m1 = qq[1] * sd1 + mean1;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (k < P) acc[k][q >> 1] += cm_logit_of_sigmoid(m0 * vk[k] - m1, etab);
      }
    }
    double tot = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < P) {
        double r01 = cm_seg_sum<16>(acc[k][0]), r23 = cm_seg_sum<16>(acc[k][1]);   // r0 | r1 and r2 | r3 in the two 16-lane rows of the half
        r01 += __shfl_xor(r01, 16, CM_WAVE);                                          // r0 + r1 (the other row: r1 + r0, the same bits)
        r23 += __shfl_xor(r23, 16, CM_WAVE);
        const double sum = r01 + r23;
        if (j == k) tot = sum;
      }
    }
    const double mean = tot / (double)S;                 // lane k of the half: the mean of power k -- one division serves all of them
    if (need && j < P) p.mcE[(size_t)n * p.mce_ld + (j == 0 ? gi[0] : (j == 1 ? gi[1] : (j == 2 ? gi[2] : gi[3])))] = mean;
  }
}

// the next Monte-Carlo logit of the rows of a wavefront's two halves (n < 0: none; pc = phi_0, phi_1, cov_00, cov_11 of the half's row; dead: the
// row is all zero and (dc0, dc1) are its tabulated truncation points): both rows at once (mc_logit_pair) where its shape limits hold and two rows
// are there, one after the other on the whole wavefront otherwise.  Either way every row's sums are those of mc_logit_body.
template <bool FULLMC>
__device__ __forceinline__ void mc_logit_rows(const cm_pdev& p, const IterDyn& a, int n, const double (&pc)[4], bool dead, double dc0, double dc1, int lane, int wave,
                                              double* sm, const double* etab) {
  const int N = p.N;
  int gi[4] = {0, 0, 0, 0}, P = 0;
  for (int g = 0; g < p.V; ++g)
    if (p.vals[g] > 0.0) { if (P < 4) gi[P] = g; ++P; }
  if (!FULLMC && a.rpw == 2 && P >= 1 && P <= 4 && a.S <= 2 * CM_WAVE) {
    double vk[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) vk[k] = p.vals[gi[k]];
    const bool need = n >= 0;
    const double c0 = dead ? dc0 : normcdf(-pc[0] / pc[2]), c1 = dead ? dc1 : normcdf(-pc[1] / pc[3]);   // (mc_logit_body evaluates the same expressions)
    mc_logit_pair(p, need, n, pc[0], pc[1], pc[2], pc[3], c0, c1, a.nx, a.S, P, gi, vk, etab, lane);
    return;
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {                        // one row after the other on the whole wavefront
    const int rown = __builtin_amdgcn_readlane(n, 32 * h);
    if (rown < 0) continue;
    const size_t off = ((size_t)a.nx * N + rown) * 2 * a.S;      // random tables of iteration nx (the next one), stored by neuron
    const double2* wu = (const double2*)p.wtab + off;
    const double m0 = cm_readlane_f64(pc[0], 32 * h), m1 = cm_readlane_f64(pc[1], 32 * h), s0 = cm_readlane_f64(pc[2], 32 * h), s1 = cm_readlane_f64(pc[3], 32 * h);
    if (!FULLMC && __builtin_amdgcn_readlane((int)dead, 32 * h))
      mc_logit_body<false, true>(rown, lane, m0, m1, s0, s1, wu, p.vals, p.V, a.S, sm + (size_t)wave * 2 * a.S, p.mcE, etab, cm_readlane_f64(dc0, 32 * h), cm_readlane_f64(dc1, 32 * h), nullptr, p.mce_ld);
    else
      mc_logit_body<FULLMC>(rown, lane, m0, m1, s0, s1, wu, p.vals, p.V, a.S, sm + (size_t)wave * 2 * a.S, p.mcE, etab, 0.0, 0.0, nullptr, p.mce_ld);
  }
}

// Batches: the all-zero rows of every circuit (95 % of the rows in the steady state) in a kernel of their own -- copy of the
// constants + the next Monte-Carlo logit, nothing else, so that no workgroup waits for a Newton iteration.  Rows with lam != 0 are not
// touched (k_iter_mid takes them).
// PHASE 1, one row per THREAD (rows_per_wg = 256 rows per workgroup while the next sweep applies the gate): copy of the constants, then the
// SCREENING.  The logit of an all-zero row is only ever read by the gate of the next sweep (mu = 0 there as long as the prior mean is zero: the
// row is outside the active set of the next solve), and for nearly all of these rows that gate fails by a wide margin -- which a bound shows
// without the S x P evaluations: with q0_s, q1_s the row's tabulated quantiles, every draw of phi_1 is at least mean1 + sd1 q1_s (its
// truncation point only moves a quantile up), so while no sample reaches the residue range (x < 8: the logit is x itself) the mean logit of
// power v is at most v (mean0 + sd0 mean(q0)) - (mean1 + sd1 mean(q1)) -- two row statistics the table kernel leaves (wstat).  If the gate
// fails even on these upper bounds (it is monotone in every logit; sigma, the spontaneous rate and the row's beta = prior are final by now), the
// row stays zero whatever its exact logits are: -inf is stored and the evaluations are skipped.  Rows the bound cannot decide are LISTED in
// LDS and take the exact path (PHASE 2).  Decisions, and with them every output, are those of the unscreened run.
template <bool BATCH>
__global__ __launch_bounds__(256) void k_mc_dead(cm_parg<BATCH> pa, IterDyn a, int rows_per_wg) {
  extern __shared__ double sm[];
  __shared__ double etab[64];
  __shared__ int list[256], nlist;
  cm_exptab_init(etab);
  if (threadIdx.x == 0) nlist = 0;
  __syncthreads();                                   // (before any wavefront leaves)
  const cm_pdev& p = pa.get();
  const int wave = threadIdx.x / CM_WAVE, lane = threadIdx.x % CM_WAVE, h = lane >> 5, j = lane & 31;
  const int N = p.N, S = a.S;
  {
    const int n = (int)blockIdx.x * rows_per_wg + (int)threadIdx.x;
    const bool in = (int)threadIdx.x < rows_per_wg && n < N;
    const int nn = in ? n : N - 1;
    const bool dead = in && p.rs[(size_t)nn * 4 + 3] < 0.0;
    bool need = false;
    if (dead) {
      const double* __restrict__ pd = p.phi_dead + 2 * (size_t)n;
      const double* __restrict__ cd = p.phi_cov_dead + 4 * (size_t)n;
      const double mean0 = pd[0], mean1 = pd[1], sd0 = cd[0], c01 = cd[1], c10 = cd[2], sd1 = cd[3];
      p.phi[2 * n] = mean0; p.phi[2 * n + 1] = mean1;
      p.phi_cov[4 * n] = sd0; p.phi_cov[4 * n + 1] = c01; p.phi_cov[4 * n + 2] = c10; p.phi_cov[4 * n + 3] = sd1;
      need = a.do_mc != 0;
      const int Pq = p.V - 1;
      if (need && a.mc_screen && Pq >= 1 && Pq <= 8 && p.max_row_len < 65536 && p.vals[0] == 0.0) {   // (the conditions of the closed-form gate: sweep_args)
        const double* __restrict__ ws = p.wstat + ((size_t)a.nx * N + n) * CM_WSTAT;
        const double qbar0 = ws[0], qmax0 = ws[1], qbar1 = ws[2], qmin1 = ws[3], umin0 = ws[4];
        const double c0 = p.phi_c_dead[2 * (size_t)n];
        // c0 below half an ulp of the row's smallest phi_0 uniform: arg = c0 + u (1 - c0) is u bit for bit for every sample, the phi_0 draws ARE
        // the tabulated quantiles (the test mc_logit_body makes sample by sample)
        bool ok = p.mu_prior[n] == 0.0 && c0 >= 0.0 && c0 < 0x1p-54 * umin0 && sd0 >= 0.0 && sd1 >= 0.0;
        const double a0 = mean0 + sd0 * qbar0, a0max = mean0 + sd0 * qmax0, b1 = mean1 + sd1 * qbar1, b1min = mean1 + sd1 * qmin1;
        double ub[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          ub[q] = 0.0;
          if (q < Pq) {
            const double v = p.vals[q + 1];
            ok = ok && (a0max * v - b1min < CM_LOGIT_RESIDUE_FROM - 1e-6);                          // every sample of this power below the residue range
            ub[q] = (a0 * v - b1) + 1e-9 * (1.0 + fabs(a0 * v) + fabs(b1));                         // (summation order, the rounding of the two means)
          }
        }
        const double b0 = p.beta_prior[n], be = b0 * b0;                                            // beta of a row outside the active set (k_fill_prior)
        const double sig = p.scal[CM_S_SHAPE] / p.scal[CM_S_RATE], thresh = a.msrmp + p.scal[CM_S_SPONT_RATE];
        if (ok && !cm_quiescent_gate_core(Pq, sig * (be * be), thresh, a.min_count, p.qcnt + (size_t)n * 8, ub)) {
          for (int q = 0; q < Pq; ++q) p.mcE[(size_t)n * p.mce_ld + q + 1] = -INFINITY;
          need = false;
        }
      }
    }
    const unsigned long long m = __ballot(need);     // ordered append of the wavefront's rows
    int base = 0;
    if (lane == 0 && m) base = atomicAdd(&nlist, __popcll(m));
    base = __builtin_amdgcn_readfirstlane(base);
    if (need) list[base + __popcll(m & ((1ull << lane) - 1ull))] = n;
  }
  __syncthreads();
  const int nl = nlist;
  if (nl == 0 || !a.do_mc) return;
  // PHASE 2: the exact Monte-Carlo logit of the listed rows, TWO rows per wavefront, one per 32-lane half: the per-row overhead (constants,
  // address arithmetic, the reductions' instruction streams, the division) is issued once for two rows.  Every row's sums are bitwise those
  // of mc_logit_body (one row per wavefront): there lane l accumulates the samples l, l + 64 and cm_wave_sum reduces the 64 lanes as four
  // 16-lane rows, (r0 + r1) + (r2 + r3); here physical lane j of a half carries the two virtual lanes j (samples j, j + 64) and j + 32
  // (samples j + 32, j + 96) in separate accumulators, the 16-lane reductions run on both sets, and the four row sums are combined in the
  // same order.  A lane's samples never leave its registers: no LDS.  More than four powers or more than 128 samples: the general body.
  int gi[4] = {0, 0, 0, 0}, P = 0;
  for (int g = 0; g < p.V; ++g)
    if (p.vals[g] > 0.0) { if (P < 4) gi[P] = g; ++P; }
  double vk[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) vk[k] = p.vals[gi[k]];
  const int nwaves = (int)blockDim.x / CM_WAVE;
  for (int i0 = 2 * wave; i0 < nl; i0 += 2 * nwaves) {
    const bool need = i0 + h < nl;
    const int n = list[need ? i0 + h : i0];
    const double* __restrict__ pd = p.phi_dead + 2 * (size_t)n;
    const double* __restrict__ cd = p.phi_cov_dead + 4 * (size_t)n;
    const double* __restrict__ cc = p.phi_c_dead + 2 * (size_t)n;
    const double mean0 = pd[0], mean1 = pd[1], sd0 = cd[0], sd1 = cd[3], c0 = cc[0], c1 = cc[1];
    if (P > 4 || S > 2 * CM_WAVE) {                  // rare shapes: the general body, one row after the other on the whole wavefront
      for (int hh = 0; hh < 2; ++hh) {
        const int src = 32 * hh;
        if (!__builtin_amdgcn_readlane((int)need, src)) continue;
        const int row = __builtin_amdgcn_readlane(n, src);
        mc_logit_body<false, true>(row, lane, cm_readlane_f64(mean0, src), cm_readlane_f64(mean1, src), cm_readlane_f64(sd0, src), cm_readlane_f64(sd1, src),
                                   (const double2*)p.wtab + ((size_t)a.nx * N + row) * 2 * S, p.vals, p.V, S, sm + (size_t)wave * 2 * S, p.mcE, etab,
                                   cm_readlane_f64(c0, src), cm_readlane_f64(c1, src), nullptr, p.mce_ld);
      }
      continue;
    }
    mc_logit_pair(p, need, n, mean0, mean1, sd0, sd1, c0, c1, a.nx, S, P, gi, vk, etab, lane);
  }
}

// LISTED (batches, steady state): the per-neuron role walks the list of rows with lam != 0 in a stride loop -- its own instantiation,
// because the loop costs the role ~100 registers (232 against 126: two wavefronts per SIMD instead of three), which the early
// iterations, where every row is alive and the role is one wavefront per row, must not pay.
#ifndef CM_ITER_MINW_PLAIN
#define CM_ITER_MINW_PLAIN 4       // wavefronts per SIMD the batch instantiations are compiled for (their VGPR budget: 512 / n)
#endif
#ifndef CM_ITER_MINW_LISTED
#define CM_ITER_MINW_LISTED 4
#endif
// The body of k_iter_mid for workgroup `bx` of circuit `p` (first: the first circuit of its launch, which carries the diagnostic stamps).  Two
// launch shapes call it: k_iter_mid (grid (blocks, circuits)) and, for the listed rows of a batch, k_iter_listed_flat (a flat grid sized per circuit).
template <bool BATCH, int SEG, bool FULLMC, bool LISTED>
__device__ __forceinline__ void iter_mid_body(const cm_pdev& p, IterDyn a, const int bx, const bool first, double* sm) {
  if (bx <= 1) {
    if (!(a.roles & 1)) return;
    if (threadIdx.x >= CM_POST_THREADS) return;
    if (BATCH && !first) a.dbg = nullptr;              // (diagnostic stamps: circuit 0 only)
    if (bx == 1) {                                       // soft threshold + spont_rate: its own workgroup, beside block 0
      if (a.do_sigma)
        post_single_body(p.y, p.R, p.colany, p.mask, p.mu, p.beta, p.rs, p.sum_y, p.N, p.K, a.it, a.max_iters, a.penalty, a.scale_factor, a.tol,
                         a.do_sigma, a.seq, p.act_idx, p.act_inv, p.n_act, p.mailbox, p.z, p.scal, a.stage_ok ? sm : nullptr, a.dbg, 2);
      return;
    }
    const int n = post_single_body(p.y, p.R, p.colany, p.mask, p.mu, p.beta, p.rs, p.sum_y, p.N, p.K, a.it, a.max_iters, a.penalty,
                                   a.scale_factor, a.tol, a.do_sigma, a.seq, p.act_idx, p.act_inv, p.n_act, p.mailbox, p.z, p.scal,
                                   nullptr, a.dbg, 1);
    if (a.do_solve && n <= p.small_cap) {
      __syncthreads();
      solve_small_body(sm, p.lam, p.row_ptr, p.col, p.act_idx, p.rs, p.scal, p.mu_prior, p.beta_prior, n, p.N, n * p.max_row_len, p.mu,
                       p.beta, p.n_act + 1, CM_POST_THREADS);
      if (a.scan_on) {
        // mu of the next sweep is final: its chain tables now, beside the Newton chains of the other blocks -- the pre-sweep then packs the
        // (few) chain rows itself and the sweep is one launch shorter
        __shared__ int wsum[2][CM_POST_THREADS / CM_WAVE];
        __shared__ int carry[2];
        __threadfence_block();
        __syncthreads();
        const int sel = (BATCH || n > a.scan_min_rows) ? 1 : 0;
        seq_scan_body<CM_POST_THREADS>(p.orders_all + (size_t)(a.it + 1) * p.N, p.mu, p.row_ptr, p.N, p.seq_n, p.seq_off, p.chunk_first, p.seq_meta, p.seq_pos,
                                       a.scan_gran[sel], a.scan_pack[sel], wsum, carry);
      }
    }
    if (a.dbg && threadIdx.x == 0) { a.dbg[4] = wall_clock64(); a.dbg[6] = n; }
    return;
  }
  if (!(a.roles & 2)) return;
  __shared__ double etab[64], ltab[128];
  cm_exptab_init(etab);
  cm_logtab_init(ltab);
  __syncthreads();                                   // (before any wavefront leaves)
  const int wave = threadIdx.x / CM_WAVE, lane = threadIdx.x % CM_WAVE;
  const int N = p.N;
  if (BATCH && LISTED) {
    // Batches: the rows with lam != 0 -- listed by k_stats, a few percent of the rows in the steady state -- are spread over a grid
    // sized for them (the host knows the previous active-set sizes; the stride loop covers whatever else turns up), instead of
    // N wavefronts per circuit of which most leave at once: that launch was bound by the number of wavefronts, not by its work.
    // (no loop here: one listed row per wavefront -- a stride loop around this body costs the kernel 100 registers, i.e. two
    // wavefronts per SIMD instead of four; rows beyond the grid, if the list ever outgrows it, are taken by k_iter_overflow)
    const int n_alive = min(p.alive[0], N);
    const int i = a.rpw * ((bx - 2) * (int)(blockDim.x / CM_WAVE) + wave) + (lane >> 5);   // rpw listed rows per wavefront, one per half (phi_pair)
    const bool mine = i < n_alive && (lane >> 5) < a.rpw;
    if (__any(mine)) {
      const int n = mine ? p.alive[1 + i] : -1;
      double pc[4];
      phi_pair<SEG>(n, lane, p.lam, p.row_ptr, p.grp, p.rowcnt, p.vals, p.V, p.phi_prior, p.phi_cov_prior, p.phi, p.phi_cov, pc, etab, ltab);
      if (a.do_mc) mc_logit_rows<FULLMC>(p, a, n, pc, false, 0.0, 0.0, lane, wave, sm, etab);
    }
    return;
  }
  // two rows per wavefront, one per 32-lane half (phi_pair): rows 2 w and 2 w + 1
  const int nrow = a.rpw * ((bx - 2) * (int)(blockDim.x / CM_WAVE) + wave) + (lane >> 5);
  const bool mine = nrow < N && (lane >> 5) < a.rpw;
  if (!__any(mine)) return;
  int n = mine ? nrow : -1;                            // this half's row
  double pc[4] = {0.0, 0.0, 0.0, 0.0};
  bool dead_row = false;
  double dc0 = 0.0, dc1 = 0.0;
  if (a.do_phi) {
    if (n >= 0 && p.rs[(size_t)n * 4 + 3] < 0.0) {
      // the row is identically zero (k_stats of this iteration): its update_phi is the per-run constant of this neuron
      if (BATCH && a.dead_sep) n = -1;                 // ... and k_mc_dead, launched behind this kernel, takes it (screening, one row per thread)
      else dead_row = true;
    }
    if (!__any(n >= 0)) return;
    double pl[4];
    phi_pair<SEG>(dead_row ? -1 : n, lane, p.lam, p.row_ptr, p.grp, p.rowcnt, p.vals, p.V, p.phi_prior, p.phi_cov_prior, p.phi, p.phi_cov, pl, etab, ltab);
    if (dead_row) dead_phi_copy(p, n, lane & 31, pc, &dc0, &dc1);
    else { pc[0] = pl[0]; pc[1] = pl[1]; pc[2] = pl[2]; pc[3] = pl[3]; }
  } else if (n >= 0) {
    pc[0] = p.phi[2 * n]; pc[1] = p.phi[2 * n + 1]; pc[2] = p.phi_cov[4 * n]; pc[3] = p.phi_cov[4 * n + 3];
  }
  if (a.do_mc) mc_logit_rows<FULLMC>(p, a, n, pc, dead_row, dc0, dc1, lane, wave, sm, etab);
}

template <bool BATCH, int SEG, bool FULLMC, bool LISTED = false>
__global__ __launch_bounds__(CM_ITER_THREADS, (BATCH && !FULLMC) ? (LISTED ? CM_ITER_MINW_LISTED : CM_ITER_MINW_PLAIN) : 1) void k_iter_mid(cm_parg<BATCH> pa, IterDyn a) {
  extern __shared__ double sm[];
  iter_mid_body<BATCH, SEG, FULLMC, LISTED>(pa.get(), a, (int)blockIdx.x, !BATCH || blockIdx.y == 0, sm);
}

// The listed launch of a batch as a FLAT grid: workgroups [0, 2 B) are the two single-workgroup roles of every circuit, the rest the row blocks of
// the circuits one after the other, as many per circuit as ITS active set asks for (pre: exclusive prefix of the per-circuit counts, made by the
// host from the mailbox values) -- the (blocks, circuits) grid is sized for the largest active set of the batch, and at 512 circuits most of its
// workgroups found nothing to do while occupying the slots of those that did.
struct IterFlat { int B; unsigned short pre[CM_MAX_BATCH + 1]; };
template <int SEG>
__global__ __launch_bounds__(CM_ITER_THREADS, CM_ITER_MINW_LISTED) void k_iter_listed_flat(cm_parg<true> pa, IterDyn a, IterFlat f) {
  extern __shared__ double sm[];
  const int g = (int)blockIdx.x;
  int c, bx;
  if (g < 2 * f.B) { c = g >> 1; bx = g & 1; }
  else {
    const int r = g - 2 * f.B;
    int lo = 0, hi = f.B;                              // pre[lo] <= r < pre[hi]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int)f.pre[mid] <= r) lo = mid; else hi = mid; }
    c = lo; bx = 2 + r - (int)f.pre[lo];
  }
  iter_mid_body<true, SEG, false, true>(pa.get(c), a, bx, c == 0, sm);
}

// Listed rows beyond the grid of the LISTED k_iter_mid (`first` = the rows that launch covers): a stride loop over whatever is left.
// The list holds the rows with lam != 0 after the sweep; the host sizes the main grid for twice the active set the iteration began
// with, so this kernel normally finds nothing and leaves at once -- it exists so that no growth of the list can lose a row.
template <int SEG>
__global__ __launch_bounds__(CM_ITER_THREADS) void k_iter_overflow(cm_parg<true> pa, IterDyn a, int first_arg, IterFlat f) {
  extern __shared__ double sm[];
  __shared__ double etab[64], ltab[128];
  cm_exptab_init(etab);
  cm_logtab_init(ltab);
  __syncthreads();
  const cm_pdev& p = pa.get();
  const int wave = threadIdx.x / CM_WAVE, lane = threadIdx.x % CM_WAVE;
  const int N = p.N;
  const int n_alive = min(p.alive[0], N);
  const int total = (int)gridDim.x * (int)(blockDim.x / CM_WAVE);
  // (flat listed launch: the rows the circuit's own row blocks covered)
  const int first = f.B > 0 ? ((int)f.pre[blockIdx.y + 1] - (int)f.pre[blockIdx.y]) * a.rpw * (int)(blockDim.x / CM_WAVE) : first_arg;
  for (int i0 = first + a.rpw * ((int)blockIdx.x * (int)(blockDim.x / CM_WAVE) + wave); i0 < n_alive; i0 += a.rpw * total) {
    const int i = i0 + (lane >> 5);                    // rpw listed rows per wavefront, one per half (phi_pair)
    const int n = (i < n_alive && (lane >> 5) < a.rpw) ? p.alive[1 + i] : -1;
    double pc[4];
    phi_pair<SEG>(n, lane, p.lam, p.row_ptr, p.grp, p.rowcnt, p.vals, p.V, p.phi_prior, p.phi_cov_prior, p.phi, p.phi_cov, pc, etab, ltab);
    if (a.do_mc) mc_logit_rows<false>(p, a, n, pc, false, 0.0, 0.0, lane, wave, sm, etab);
  }
}

// largest active set the fused kernel may solve in LDS (0 = never): A (n x (n+1)) + staged rows must fit beside the statics
int cm_small_cap(const cm_problem* p) {
  int cap = 0;
  for (int n = 1; n <= CM_FUSED_SOLVE_MAX; ++n)   // steady-state active sets fit; beyond ~24 rows the tiled solve is faster than n sequential pivots
    if (small_lds_bytes(n, n * p->max_row_len) <= 56 * 1024) cap = n; else break;
  return cap;
}

template <int SEG, bool FULLMC>
static int launch_iter2(cm_ctx* ctx, cm_set& s, const IterDyn& a0, size_t lds, int blocks, size_t lds_waves, int threads, hipEvent_t between) {
  const int slot = CM_LDS_ITER + 4 * (31 - __builtin_clz((unsigned)SEG)) + (FULLMC ? 1 : 0) + (s.B > 1 ? 2 : 0);          // SEG = 1, 2, 4 .. 64
  if (s.B == 1) CM_TRY(cm_raise_lds(ctx, slot, reinterpret_cast<const void*>(k_iter_mid<false, SEG, FULLMC>), lds, 48 * 1024));
  else CM_TRY(cm_raise_lds(ctx, slot, reinterpret_cast<const void*>(k_iter_mid<true, SEG, FULLMC>), lds, 48 * 1024));
  static const int batch_split = getenv("CAVIAR_HIP_BATCH_SPLIT") ? atoi(getenv("CAVIAR_HIP_BATCH_SPLIT")) : 0;
  if (s.B > 1 && batch_split > 0 && a0.roles == 3 && !FULLMC) {
    // diagnostic: the two roles of a batch as two launches; the per-neuron role in workgroups of `batch_split` threads with only its own LDS
    IterDyn a = a0;
    const int thr = batch_split, waves = thr / CM_WAVE, N = s.lead->N;
    a.roles = 1;
    CM_LAUNCH_SET_T(k_iter_mid, (SEG, FULLMC), s, 2, CM_ITER_THREADS_BATCH, lds, ctx->stream, a);
    a.roles = 2;
    CM_LAUNCH_SET_T(k_iter_mid, (SEG, FULLMC), s, 2 + (N + waves - 1) / waves, thr, (size_t)waves * lds_waves, ctx->stream, a);
    CM_HIP(hipGetLastError());
    return CM_OK;
  }
  if (a0.skip_dead && !FULLMC && s.B > 1) {
    // batches, steady state: the rows with lam != 0 from the list; the all-zero rows in their lean kernel (after the mailboxes are on their way)
    const int slot_l = CM_LDS_ITER_LISTED + (31 - __builtin_clz((unsigned)SEG));
    CM_TRY(cm_raise_lds(ctx, slot_l, reinterpret_cast<const void*>(k_iter_mid<true, SEG, false, true>), lds, 48 * 1024));
    // row blocks per circuit: twice ITS active set (rows can be revived; whatever outgrows it is k_iter_overflow's), as a flat grid when the
    // prefix fits its 16-bit words; else -- and with the test hook's fixed grid -- the (blocks, circuits) grid sized for the largest active set
    static const bool no_flat = getenv("CAVIAR_HIP_NO_FLAT_LISTED") != nullptr || getenv("CAVIAR_HIP_TEST_LISTED_ROWS") != nullptr;
    IterFlat* f = nullptr;
    thread_local IterFlat flat_tl;                       // (2 KB, filled per launch: one per driving thread -- the context's and its twin's)
    const int waves = threads / CM_WAVE, N = s.lead->N, per_wg = waves * a0.rpw;
    if (!no_flat && (int)s.na.size() == s.B) {
      long long tot = 0;
      flat_tl.B = s.B;
      for (int c = 0; c < s.B; ++c) {
        flat_tl.pre[c] = (unsigned short)tot;
        const int want = std::min(N, std::max(16, 2 * s.na[c]));
        tot += (want + per_wg - 1) / per_wg;
        if (tot > 65535) break;
      }
      if (tot <= 65535) { flat_tl.pre[s.B] = (unsigned short)tot; f = &flat_tl; }
    }
    if (f) {
      const int slot_f = CM_LDS_ITER_FLAT + (31 - __builtin_clz((unsigned)SEG));
      CM_TRY(cm_raise_lds(ctx, slot_f, reinterpret_cast<const void*>(k_iter_listed_flat<SEG>), lds, 48 * 1024));
      hipLaunchKernelGGL((k_iter_listed_flat<SEG>), dim3(2 * s.B + (unsigned)f->pre[s.B]), dim3(threads), lds, ctx->stream, cm_parg<true>{s.d_arr}, a0, *f);
      CM_TRY(cm_raise_lds(ctx, CM_LDS_ITER_OVERFLOW + (31 - __builtin_clz((unsigned)SEG)), reinterpret_cast<const void*>(k_iter_overflow<SEG>), (size_t)waves * lds_waves, 48 * 1024));
      hipLaunchKernelGGL((k_iter_overflow<SEG>), dim3(8, s.B), dim3(threads), (size_t)waves * lds_waves, ctx->stream, cm_parg<true>{s.d_arr}, a0, 0, *f);
    } else {
      hipLaunchKernelGGL((k_iter_mid<true, SEG, false, true>), dim3(blocks, s.B), dim3(threads), lds, ctx->stream, cm_parg<true>{s.d_arr}, a0);
      const int covered = (blocks - 2) * waves * a0.rpw;
      if (covered < s.lead->N) {                         // (a grid that already covers N rows cannot be outgrown)
        flat_tl.B = 0;
        CM_TRY(cm_raise_lds(ctx, CM_LDS_ITER_OVERFLOW + (31 - __builtin_clz((unsigned)SEG)), reinterpret_cast<const void*>(k_iter_overflow<SEG>), (size_t)waves * lds_waves, 48 * 1024));
        hipLaunchKernelGGL((k_iter_overflow<SEG>), dim3(8, s.B), dim3(threads), (size_t)waves * lds_waves, ctx->stream, cm_parg<true>{s.d_arr}, a0, covered, flat_tl);
      }
    }
    if (between) CM_HIP(hipEventRecord(between, ctx->stream));
    CM_TRY(cm_raise_lds(ctx, CM_LDS_MC_DEAD_B, reinterpret_cast<const void*>(k_mc_dead<true>), 4 * lds_waves, 48 * 1024));   // (num_mc_samples > 768)
    {
      // screening on: 256 rows per workgroup (one per thread; the few rows the bound leaves are evaluated two per wavefront); off: 8 rows per
      // workgroup, i.e. every wavefront has its two rows at once
      const int rows_per_wg = a0.mc_screen ? 256 : 8;
      CM_LAUNCH_SET(k_mc_dead, s, (s.lead->N + rows_per_wg - 1) / rows_per_wg, 256, 4 * lds_waves, ctx->stream, a0, rows_per_wg);
    }
  } else {
    CM_LAUNCH_SET_T(k_iter_mid, (SEG, FULLMC), s, blocks, threads, lds, ctx->stream, a0);
    if (between) CM_HIP(hipEventRecord(between, ctx->stream));
    if (a0.dead_sep && !FULLMC && s.B > 1) {
      CM_TRY(cm_raise_lds(ctx, CM_LDS_MC_DEAD_B, reinterpret_cast<const void*>(k_mc_dead<true>), 4 * lds_waves, 48 * 1024));
      CM_LAUNCH_SET(k_mc_dead, s, (s.lead->N + 255) / 256, 256, 4 * lds_waves, ctx->stream, a0, 256);
    }
  }
  CM_HIP(hipGetLastError());
  return CM_OK;
}
template <int SEG>
static int launch_iter(cm_ctx* ctx, cm_set& s, const IterDyn& a, size_t lds, int blocks, int threads, hipEvent_t between) {
  const size_t lw = 2 * (size_t)a.S * sizeof(double);
  // state init (it < 0): the quantile of every draw is needed (phi is the prior: c is not small).  One circuit: the instantiation with
  // the quantile function inlined (shortest path).  Batches: the out-of-line call -- inlined it makes the kernel a 424-register one-
  // wavefront-per-SIMD launch, and at 128 circuits that launch is throughput-bound.  Same function, same arguments: same logits.
  return (a.it < 0 && s.B == 1) ? launch_iter2<SEG, true>(ctx, s, a, lds, blocks, lw, threads, between) : launch_iter2<SEG, false>(ctx, s, a, lds, blocks, lw, threads, between);
}

// it = -1: state init (no sigma / phi; MC logit and solve for iteration 0).  Otherwise the middle of iteration `it`.
// roles: 1 = the single-workgroup role, 2 = the per-neuron role, 3 = both.
int cm_launch_iter_mid(cm_ctx* ctx, cm_set& s, int it, int iters, int S, int rng_mode, const cm_options* o, int do_sigma, int do_phi, int roles,
                       hipEvent_t between) {
  IterDyn a;
  const cm_problem* p = s.lead;
  const int N = p->N, nxt = it + 1;
  a.it = it; a.max_iters = o->max_backtrack_iters; a.penalty = o->penalty; a.scale_factor = o->scale_factor;
  a.tol = o->tol; a.do_sigma = do_sigma; a.seq = ++ctx->mail_seq;
  a.do_solve = nxt < iters;
  a.do_phi = do_phi; a.do_mc = nxt < iters; a.S = S;
  a.nx = a.do_mc ? nxt : 0;
  a.explicit_rng = rng_mode == CM_RNG_EXPLICIT;
  // batches: listed rows + k_mc_dead once fewer than half of the rows were active when the iteration began (any choice is correct: the
  // plain form handles all-zero rows itself); the early iterations, with every row alive, run one wavefront per row
  a.skip_dead = (s.B > 1 && do_phi && it >= 0 && roles == 3 && 2 * s.na_max < N) ? 1 : 0;
  static const bool no_screen = getenv("CAVIAR_HIP_NO_MC_SCREEN") != nullptr;     // diagnostic: every all-zero row through the exact Monte-Carlo logit
  a.mc_screen = (!no_screen && nxt > o->delay_spont_est) ? 1 : 0;                   // (the sweep of iteration nxt applies the gate: cm_launch_sweep)
  a.msrmp = o->msrmp; a.min_count = o->minimum_spike_count;
  // batches, the gated iterations before the listed regime: every row still has its wavefront (the rows with lam != 0 are too many for a list),
  // but the all-zero ones -- most rows from the first gated sweep on -- go to k_mc_dead, where nearly all of them are screened
  a.dead_sep = (s.B > 1 && do_phi && it >= 0 && roles == 3 && a.mc_screen && !a.skip_dead && nxt < iters) ? 1 : 0;
  a.dbg = (ctx->dbg_block0 && it >= 0 && it < 64) ? ctx->dbg_block0 + 8 * it : nullptr;
  // the chain tables of the next sweep from block 0 (cm_launch_sweep packs in the pre-sweep when every circuit of the set got them)
  a.scan_on = (a.do_solve && (roles & 1)) ? 1 : 0;
  {
    const int par = cm_sweep_variant(s, 1 << 30);
    cm_sweep_chunking(s, 0, nullptr, &a.scan_pack[0], &a.scan_gran[0]);
    cm_sweep_chunking(s, par, nullptr, &a.scan_pack[1], &a.scan_gran[1]);
    a.scan_min_rows = cm_par_min_rows();
    if (par == 0) { a.scan_pack[1] = a.scan_pack[0]; a.scan_gran[1] = a.scan_gran[0]; }
  }
  s.scan_it = a.scan_on ? nxt : -2;
  const int threads = s.B == 1 ? CM_ITER_THREADS : CM_ITER_THREADS_BATCH;
  const int waves = threads / CM_WAVE;
  size_t lds = (size_t)waves * 2 * S * sizeof(double);
  lds = std::max(lds, s.small_lds_max);
  a.stage_ok = 0;
  if ((size_t)p->K * 9 + 16 <= 56 * 1024) { lds = std::max(lds, (size_t)p->K * 9 + 16); a.stage_ok = 1; }   // residual + flag per trial
  // rows per wavefront: batches put a row in each half (measured at 256 circuits: early iterations 1.99 -> 1.66 ms per launch, listed rows 8.9
  // -> 8.2 ms per step); one circuit keeps one row per wavefront -- its launch is as long as one wavefront's chain, which the second row's
  // Monte-Carlo logit would only lengthen (26.9 against 30.3 us per launch).  The results do not depend on the choice (phi_pair).
  a.rpw = s.B > 1 ? 2 : 1;
  int blocks = 2 + ((do_phi || a.do_mc) ? (N + a.rpw * waves - 1) / (a.rpw * waves) : 0);
  if (a.skip_dead) {
    // rows with lam != 0 after this sweep: at most a few more than the active set the iteration started with (rows can be revived);
    // twice that, at least 64 wavefronts per circuit -- the kernel's stride loop takes care of the rest
    int want = std::min(N, std::max(64, 2 * s.na_max));
    if (const char* e = getenv("CAVIAR_HIP_TEST_LISTED_ROWS")) want = std::min(N, std::max(1, atoi(e)));   // test hook: a grid the list outgrows (k_iter_overflow)
    blocks = 2 + (want + a.rpw * waves - 1) / (a.rpw * waves);
  }
  const int V = p->V;
  static const bool split = getenv("CAVIAR_HIP_SPLIT_ITER") != nullptr;         // diagnostic: one launch per role (profiling only)
  a.roles = roles;
  if (split && it >= 0 && roles == 3) {
    a.roles = 2;
    if (V <= 4) { CM_TRY(launch_iter<4>(ctx, s, a, lds, blocks, threads, between)); } else { CM_TRY(launch_iter<64>(ctx, s, a, lds, blocks, threads, between)); }
    a.roles = 1;
  }
  if (V <= 1) return launch_iter<1>(ctx, s, a, lds, blocks, threads, between);
  if (V <= 2) return launch_iter<2>(ctx, s, a, lds, blocks, threads, between);
  if (V <= 4) return launch_iter<4>(ctx, s, a, lds, blocks, threads, between);
  if (V <= 8) return launch_iter<8>(ctx, s, a, lds, blocks, threads, between);
  if (V <= 16) return launch_iter<16>(ctx, s, a, lds, blocks, threads, between);
  if (V <= 32) return launch_iter<32>(ctx, s, a, lds, blocks, threads, between);
  return launch_iter<64>(ctx, s, a, lds, blocks, threads, between);
}
