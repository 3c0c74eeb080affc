/*
 * sim_hip.h -- C ABI of the device-side synthetic experiment generator in libcaviar_hip.so (MI355X, gfx950).
 *
 * Generative model of the reference's `circuitmap.simulation.simulate`
 * (reference circuitmap/simulation.py:25-195): blockwise H-spot hologram design (:44-63) or random design (:65-77), per-neuron PSC kernels and
 * power curves (:81-92), Bernoulli spikes padded to a minimum rate at maximal power (:93-108), gamma latencies
 * (:110-114,197-198), sparse strong/weak connectivity (:116-127), evoked PSCs (:17-23,141-155), spontaneous PSCs
 * (:157-170), correlated + iid noise (:172-176).  Same argument names and defaults (cm_sim_options_default).
 *
 * The reference draws from NumPy's global Mersenne-Twister stream in program order; a parallel generator cannot
 * reproduce those draws.  This one is counter-based (Threefry-2x32-20 keyed by seed and stream): same distributions
 * and structure, different draws, reproducible from `seed` on any launch geometry.  Its CPU oracle is
 * oracle/sim_oracle.py, which documents the two distribution-preserving substitutions (sort-by-random-key subsets,
 * filtered-white-noise GP).  Outputs feed cm_problem_create(..., CM_MEM_DEVICE) without touching the host.
 */
#ifndef SIM_HIP_H
#define SIM_HIP_H

#include "caviar_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define CM_SIM_MAX_POWERS 16
#define CM_SIM_MAX_DIM 16384          /* N and trials: the permutations are sorted inside one workgroup's LDS */

/* keyword arguments of simulate() (simulation.py:25-30), same names; `trials` = K */
typedef struct cm_sim_options {
  int32_t N, T, H, trials, nreps;
  int32_t n_powers;
  double powers[CM_SIM_MAX_POWERS];
  double connection_prob, min_latency, gamma_beta, sigma, frac_strongly_connected;
  double strong_weight_lower, strong_weight_upper, weak_exp_mean, min_weight;
  double phi_0_lower, phi_0_upper, phi_1_lower, phi_1_upper, mult_noise_log_var;
  double tau_r_min, tau_r_max, tau_delta_min, tau_delta_max;
  double gp_scale, gp_lengthscale, spont_prob, max_power_min_spike_rate;
  uint64_t seed;
} cm_sim_options;

/* caller-allocated outputs, all float64, all in `mem` (cm_mem_kind); psc and stim_matrix are required, the rest may be NULL.
 * Keys of the reference's `sim` dict (simulation.py:179-193). */
typedef struct cm_sim_result {
  double* psc;          /* [trials, T]  */
  double* stim_matrix;  /* [N, trials]  */
  double* weights;      /* [N] */
  double* phi_0;        /* [N] */
  double* phi_1;        /* [N] */
  double* spks;         /* [N, trials] */
  double* spk_times;    /* [N, trials] */
  double* I;            /* [trials]  power of each trial */
  double* gp_noise;     /* [trials, T] */
  double* spont_pscs;   /* [trials, T] */
  double* tau_r;        /* [N] */
  double* tau_d;        /* [N] */
} cm_sim_result;

typedef enum cm_sim_design { CM_SIM_BLOCKWISE = 0, CM_SIM_RANDOM = 1 } cm_sim_design;   /* `design` keyword (simulation.py:32,44,65) */

void cm_sim_options_default(cm_sim_options* o);      /* N=300, T=900, H=10, trials=1000, powers {45,55,65}, ... seed=0 */
int cm_simulate(cm_ctx* ctx, const cm_sim_options* o, int design, const cm_sim_result* out, int mem, float* device_ms);

#ifdef __cplusplus
}
#endif
#endif
