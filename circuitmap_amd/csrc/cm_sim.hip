// cm_sim.hip -- device-side synthetic experiment generator (include/sim_hip.h): the generative model of the reference's
// simulate(design='blockwise') (circuitmap/simulation.py:25-195) on a counter-based generator.  Every variate is a pure function
// of (seed, stream, index) through Threefry-2x32-20, so each kernel draws what it needs where it needs it; oracle/sim_oracle.py
// restates the same functions in NumPy (stream ids, index conventions and formulas are kept identical on both sides).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <vector>

#include "../../include/sim_hip.h"
#include "cm_internal.h"

namespace {
enum : uint32_t { S_ORDER, S_SHUFFLE, S_CONN, S_TAU_R, S_TAU_D, S_PHI0, S_PHI1, S_WEIGHT, S_SPIKE, S_PAD, S_GAMMA_N, S_GAMMA_U, S_MULT,
                  S_SPONT, S_SP_TR, S_SP_TD, S_SP_T0, S_SP_W, S_GP, S_IID, S_GAMMA_B, S_TARGET };
constexpr int GAMMA_TRIES = 16;
constexpr int GP_HALF_WIDTHS = 4;
constexpr int SIM_MAX_H = 64;              // cells per hologram the trace kernel stages
constexpr int PAD_CAP = 8192;              // silent maximal-power trials of one neuron the padding step can rank

__device__ __forceinline__ uint64_t sim_bits(uint64_t seed, uint32_t stream, uint64_t idx) {
  uint32_t x0 = (uint32_t)idx, x1 = (uint32_t)(idx >> 32);
  cm_threefry2x32((uint32_t)seed, (uint32_t)(seed >> 32) ^ (0x5EED0000u + stream), x0, x1);
  return ((uint64_t)x0 << 32) | x1;
}
__device__ __forceinline__ double sim_uniform(uint64_t seed, uint32_t stream, uint64_t idx) {
  return (double)(sim_bits(seed, stream, idx) >> 11) * 0x1.0p-53;
}
__device__ __forceinline__ double sim_normal(uint64_t seed, uint32_t stream, uint64_t idx) {      // Box-Muller on draws 2 idx, 2 idx + 1
  const double u1 = sim_uniform(seed, stream, 2 * idx), u2 = sim_uniform(seed, stream, 2 * idx + 1);
  return sqrt(-2.0 * log(1.0 - u1)) * cos(2.0 * M_PI * u2);
}
// Marsaglia-Tsang with a fixed budget of tries; shape < 1 by the U^(1/a) boost (oracle/sim_oracle.py: gamma)
__device__ double sim_gamma(uint64_t seed, uint64_t idx, double shape, double scale) {
  const bool boost = shape < 1.0;
  const double a = boost ? shape + 1.0 : shape;
  const double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
  double out = d;
  for (int j = 0; j < GAMMA_TRIES; ++j) {
    const double x = sim_normal(seed, S_GAMMA_N, idx * GAMMA_TRIES + j);
    const double u = sim_uniform(seed, S_GAMMA_U, idx * GAMMA_TRIES + j);
    const double v = 1.0 + c * x, v3 = v * v * v;
    if (v > 0.0 && log(1.0 - u) < 0.5 * x * x + d - d * v3 + d * log(v3)) { out = d * v3; break; }
  }
  if (boost) out *= pow(1.0 - sim_uniform(seed, S_GAMMA_B, idx), 1.0 / shape);
  return out * scale;
}
__device__ __forceinline__ double sim_geom(double tau, double m) { return (1.0 - exp(-m / tau)) / (1.0 - exp(-1.0 / tau)); }

struct SimDev {
  cm_sim_options o;
  double powers_desc[CM_SIM_MAX_POWERS];
  int random_design;     // 0 = blockwise, 1 = random
  int K, nholo, tpr, R, n_conn, n_strong, J, W;
  int32_t* orders;       // [R][N]
  int32_t* reorder;      // [K]
  int32_t* conn;         // [N]
  int32_t* rank;         // [N]
  int32_t* cells;        // [K][H]
  uint8_t* spkb;         // [N][K]
  double* area;          // [N]
  double* scal;          // [2] min / max connected weight
  double* hfilt;         // [2 J + 1]
  int32_t* err;          // [1]
  double *psc, *stim, *weights, *phi_0, *phi_1, *spks, *spk_times, *I, *gp_noise, *spont_pscs, *tau_r, *tau_d;
};

// ---- random permutations: sort (50 random bits | index) inside one workgroup's LDS.  Block b < R: neuron order of design
// block b; b == R: trial shuffle; b == R + 1: connectivity order.
__global__ __launch_bounds__(1024) void k_sim_sort(SimDev s) {
  extern __shared__ uint64_t keys[];
  const int b = blockIdx.x;
  const int n = (b == s.R) ? s.K : s.o.N;
  const uint32_t stream = (b < s.R) ? S_ORDER : (b == s.R ? S_SHUFFLE : S_CONN);
  const uint64_t base = (b < s.R) ? (uint64_t)b * s.o.N : 0;
  int32_t* out = (b < s.R) ? s.orders + (size_t)b * s.o.N : (b == s.R ? s.reorder : s.conn);
  int npad = 1;
  while (npad < n) npad <<= 1;
  for (int i = threadIdx.x; i < npad; i += blockDim.x)
    keys[i] = (i < n) ? ((sim_bits(s.o.seed, stream, base + i) >> 14 << 14) | (uint64_t)i) : ~0ull;
  __syncthreads();
  for (int k = 2; k <= npad; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < npad; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const uint64_t a = keys[i], c = keys[ixj];
          const bool up = (i & k) == 0;
          if ((a > c) == up) { keys[i] = c; keys[ixj] = a; }
        }
      }
      __syncthreads();
    }
  for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] = (int32_t)(keys[i] & 0x3FFF);
}

// ---- per-neuron parameters (one workgroup: also reduces the range of the connected weights)
__global__ __launch_bounds__(1024) void k_sim_neurons(SimDev s) {
  __shared__ double rmin[16], rmax[16];
  const cm_sim_options& o = s.o;
  const int N = o.N, T = o.T;
  for (int i = threadIdx.x; i < N; i += blockDim.x) s.rank[s.conn[i]] = i;
  __threadfence_block();
  __syncthreads();
  double lo = INFINITY, hi = -INFINITY;
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    const double tr = o.tau_r_min + (o.tau_r_max - o.tau_r_min) * sim_uniform(o.seed, S_TAU_R, n);
    const double td = tr + o.tau_delta_min + (o.tau_delta_max - o.tau_delta_min) * sim_uniform(o.seed, S_TAU_D, n);
    const double p0 = o.phi_0_lower + (o.phi_0_upper - o.phi_0_lower) * sim_uniform(o.seed, S_PHI0, n);
    const double p1 = o.phi_1_lower + (o.phi_1_upper - o.phi_1_lower) * sim_uniform(o.seed, S_PHI1, n);
    // trapz over [0, T) of exp(-j / tau_d) - exp(-j / tau_r), + 1e-5   (simulation.py:285-289; the first sample is 0)
    s.area[n] = (sim_geom(td, T) - sim_geom(tr, T)) - 0.5 * (exp(-(T - 1) / td) - exp(-(T - 1) / tr)) + 1e-5;
    const int rk = s.rank[n];
    const double uw = sim_uniform(o.seed, S_WEIGHT, n);
    double w = 0.0;
    if (rk < s.n_strong) w = o.strong_weight_lower + (o.strong_weight_upper - o.strong_weight_lower) * uw;
    else if (rk < s.n_conn) w = o.min_weight - o.weak_exp_mean * log(1.0 - uw);
    if (rk < s.n_conn) { lo = w < lo ? w : lo; hi = w > hi ? w : hi; }
    s.tau_r[n] = tr; s.tau_d[n] = td; s.phi_0[n] = p0; s.phi_1[n] = p1; s.weights[n] = w;
  }
  for (int off = 32; off; off >>= 1) {
    const double a = __shfl_xor(lo, off), b = __shfl_xor(hi, off);
    lo = a < lo ? a : lo; hi = b > hi ? b : hi;
  }
  if ((threadIdx.x & 63) == 0) { rmin[threadIdx.x >> 6] = lo; rmax[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) { lo = rmin[i] < lo ? rmin[i] : lo; hi = rmax[i] > hi ? rmax[i] : hi; }
    s.scal[0] = s.n_conn ? lo : 0.0;
    s.scal[1] = s.n_conn ? hi : 0.0;
  }
}

// ---- the design: trial k is entry reorder[k] of the block-ordered list (power-major, hologram, repetition)   (:44-63)
__global__ __launch_bounds__(256) void k_sim_design(SimDev s) {
  const int k = blockIdx.x * (blockDim.x / CM_WAVE) + threadIdx.x / CM_WAVE, lane = threadIdx.x % CM_WAVE;
  if (k >= s.K) return;
  const int N = s.o.N, H = s.o.H;
  const int t = s.reorder[k];
  double power;
  if (s.random_design) {
    // 'random' (:65-77): entry t of the equal split of the powers (caller's order: np.array_split sizes), H targets with replacement
    const int P = s.o.n_powers, q = s.K / P, rmd = s.K % P;                // the first `rmd` groups have q + 1 entries
    const int pi = (t < rmd * (q + 1)) ? t / (q + 1) : rmd + (t - rmd * (q + 1)) / (q > 0 ? q : 1);
    power = s.o.powers[pi];
    for (int i = lane; i < H; i += CM_WAVE) {
      const int n = (int)floor(sim_uniform(s.o.seed, S_TARGET, (uint64_t)k * H + i) * N);
      bool dup = false;
      for (int j = 0; j < i; ++j) dup |= ((int)floor(sim_uniform(s.o.seed, S_TARGET, (uint64_t)k * H + j) * N) == n);
      s.stim[(size_t)n * s.K + k] = power;
      s.cells[(size_t)k * H + i] = dup ? -1 : n;
    }
  } else {
    const int r = t / s.tpr, rem = t - r * s.tpr;
    const int pi = rem / (s.nholo * s.o.nreps), h = (rem / s.o.nreps) % s.nholo;
    power = s.powers_desc[pi];
    for (int i = lane; i < H; i += CM_WAVE) {
      const int idx = h * H + i;
      int n = -1;
      if (idx < N) { n = s.orders[(size_t)r * N + idx]; s.stim[(size_t)n * s.K + k] = power; }
      s.cells[(size_t)k * H + i] = n;
    }
  }
  if (lane == 0 && s.I) s.I[k] = power;
}

// ---- spikes of one neuron (one workgroup per neuron), padded to the minimum rate at maximal power   (:93-108)
__global__ __launch_bounds__(256) void k_sim_spikes(SimDev s) {
  __shared__ int zero_k[PAD_CAP];
  __shared__ int n_loc, n_spk, n_zero;
  const cm_sim_options& o = s.o;
  const int n = blockIdx.x, K = s.K;
  if (threadIdx.x == 0) { n_loc = 0; n_spk = 0; n_zero = 0; }
  __syncthreads();
  const double p0 = s.phi_0[n], p1 = s.phi_1[n], pmax = s.powers_desc[0];
  const double* row = s.stim + (size_t)n * K;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    const double st = row[k];
    uint8_t sp = 0;
    if (st > 0.0) {
      const double fr = 1.0 / (1.0 + exp(-(p0 * st - p1)));
      sp = sim_uniform(o.seed, S_SPIKE, (uint64_t)n * K + k) <= fr;
      if (st == pmax) {
        atomicAdd(&n_loc, 1);
        if (sp) atomicAdd(&n_spk, 1);
        else {
          const int q = atomicAdd(&n_zero, 1);
          if (q < PAD_CAP) zero_k[q] = k; else *s.err = 1;
        }
      }
    }
    s.spkb[(size_t)n * K + k] = sp;
  }
  __syncthreads();
  if (n_loc == 0) return;
  const double diff = o.max_power_min_spike_rate - (double)n_spk / (double)n_loc;
  if (!(diff > 0.0)) return;
  const int nz = n_zero < PAD_CAP ? n_zero : PAD_CAP;
  int req = (int)ceil(diff * (double)n_loc);
  req = req < nz ? req : nz;
  // the `req` silent trials with the smallest random keys fire (= a uniformly random subset)
  for (int e = threadIdx.x; e < nz; e += blockDim.x) {
    const int ke = zero_k[e];
    const uint64_t key = (sim_bits(o.seed, S_PAD, (uint64_t)n * K + ke) >> 14 << 14) | (uint64_t)ke;
    int rk = 0;
    for (int f = 0; f < nz; ++f) {
      const int kf = zero_k[f];
      const uint64_t kq = (sim_bits(o.seed, S_PAD, (uint64_t)n * K + kf) >> 14 << 14) | (uint64_t)kf;
      rk += kq < key;
    }
    if (rk < req) s.spkb[(size_t)n * K + ke] = 1;
  }
}

// ---- dense views of the spike matrix (optional outputs)
__global__ void k_sim_expand_spks(SimDev s) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)s.o.N * s.K) { if (s.spks) s.spks[i] = (double)s.spkb[i]; if (s.spk_times) s.spk_times[i] = 0.0; }
}

// ---- one trace per workgroup: evoked + spontaneous PSCs + correlated noise + iid noise   (:141-176)
__global__ __launch_bounds__(1024) void k_sim_traces(SimDev s) {
  extern __shared__ double zs[];                 // [W] white noise of this trial, then [2 J + 1] filter taps
  __shared__ double e_td[SIM_MAX_H], e_tr[SIM_MAX_H], e_amp[SIM_MAX_H], red[16];
  __shared__ int e_s[SIM_MAX_H];
  const cm_sim_options& o = s.o;
  const int k = blockIdx.x, K = s.K, T = o.T, H = o.H, J = s.J, W = s.W;
  double* hf = zs + W;
  for (int q = threadIdx.x; q < W; q += blockDim.x) zs[q] = sim_normal(o.seed, S_GP, (uint64_t)k * W + q);
  for (int q = threadIdx.x; q < 2 * J + 1; q += blockDim.x) hf[q] = s.hfilt[q];
  if ((int)threadIdx.x < H) {
    const int i = threadIdx.x, n = s.cells[(size_t)k * H + i];
    int sidx = -1;
    double amp = 0.0, td = 1.0, tr = 1.0;
    if (n >= 0 && s.spkb[(size_t)n * K + k]) {
      const uint64_t e = (uint64_t)n * K + k;
      const double power = s.stim[e];
      const double st = o.min_latency + sim_gamma(o.seed, e, 1e4 / (power * power), o.gamma_beta);
      if (s.spk_times) s.spk_times[e] = st;
      const double m = exp(o.mult_noise_log_var * sim_normal(o.seed, S_MULT, e));
      const double w = s.weights[n];
      const int si = (int)st;
      if (w != 0.0 && si < T) {
        td = s.tau_d[n]; tr = s.tau_r[n];
        const double area = s.area[n];
        const double tot = (sim_geom(td, T - si) - sim_geom(tr, T - si)) / area;      // area of the shifted, truncated unit kernel
        amp = (m * w) / ((tot + 1e-5) * area);
        sidx = si;
      }
    }
    e_s[i] = sidx; e_td[i] = td; e_tr[i] = tr; e_amp[i] = amp;
  }
  // spontaneous event of this trial (:157-170)
  const bool spont = sim_uniform(o.seed, S_SPONT, k) <= o.spont_prob;
  const double sp_tr = o.tau_r_min + (o.tau_r_max - o.tau_r_min) * sim_uniform(o.seed, S_SP_TR, k);
  const double sp_td = sp_tr + o.tau_delta_min + (o.tau_delta_max - o.tau_delta_min) * sim_uniform(o.seed, S_SP_TD, k);
  const double sp_t0 = 1.0 + floor(sim_uniform(o.seed, S_SP_T0, k) * (double)(T - 1));
  const double sp_w = s.scal[0] + (s.scal[1] - s.scal[0]) * sim_uniform(o.seed, S_SP_W, k);
  __syncthreads();
  double ksum = 0.0, klast = 0.0;
  if (spont) {                                    // trapz(kern) = sum - (first + last) / 2; the first sample is 0 (t0 >= 1)
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
      const double j = (double)t - sp_t0;
      const double v = j > 0.0 ? exp(-j / sp_td) - exp(-j / sp_tr) : 0.0;
      ksum += v;
      if (t == T - 1) klast = v;
    }
    for (int off = 32; off; off >>= 1) { ksum += __shfl_xor(ksum, off); klast += __shfl_xor(klast, off); }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ksum - 0.5 * klast;
  }
  __syncthreads();
  double trap = 0.0;
  if (spont) for (int i = 0; i < (int)(blockDim.x >> 6); ++i) trap += red[i];
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    double ev = 0.0;
    for (int i = 0; i < H; ++i) {
      const int si = e_s[i];
      if (si >= 0 && t >= si) {
        const double j = (double)(t - si);
        ev += (exp(-j / e_td[i]) - exp(-j / e_tr[i])) * e_amp[i];
      }
    }
    double sp = 0.0;
    if (spont) {
      const double j = (double)t - sp_t0;
      if (j > 0.0) sp = sp_w * (exp(-j / sp_td) - exp(-j / sp_tr)) / (trap + 1e-5);
    }
    double gp = 0.0;
    for (int q = 0; q < 2 * J + 1; ++q) gp = fma(hf[q], zs[t + q], gp);
    gp *= o.gp_scale;
    const double iid = o.sigma * sim_normal(o.seed, S_IID, (uint64_t)k * T + t);
    s.psc[(size_t)k * T + t] = ev + sp + gp + iid;
    if (s.gp_noise) s.gp_noise[(size_t)k * T + t] = gp;
    if (s.spont_pscs) s.spont_pscs[(size_t)k * T + t] = sp;
  }
}
}  // namespace

extern "C" void cm_sim_options_default(cm_sim_options* o) {
  if (!o) return;
  *o = cm_sim_options{};
  o->N = 300; o->T = 900; o->H = 10; o->trials = 1000; o->nreps = 1;
  o->n_powers = 3; o->powers[0] = 45; o->powers[1] = 55; o->powers[2] = 65;
  o->connection_prob = 0.05; o->min_latency = 160; o->gamma_beta = 1.5e1; o->sigma = 6e-4; o->frac_strongly_connected = 0.2;
  o->strong_weight_lower = 20; o->strong_weight_upper = 40; o->weak_exp_mean = 4; o->min_weight = 9;
  o->phi_0_lower = 0.2; o->phi_0_upper = 0.25; o->phi_1_lower = 10; o->phi_1_upper = 15; o->mult_noise_log_var = 0.01;
  o->tau_r_min = 25; o->tau_r_max = 60; o->tau_delta_min = 75; o->tau_delta_max = 250;
  o->gp_scale = 4e-3; o->gp_lengthscale = 50; o->spont_prob = 0.05; o->max_power_min_spike_rate = 0.4;
  o->seed = 0;
}

extern "C" int cm_simulate(cm_ctx* ctx, const cm_sim_options* o, int design, const cm_sim_result* out, int mem, float* device_ms) {
  if (!ctx || !o || !out || !out->psc || !out->stim_matrix) { cm_set_error("cm_simulate: NULL argument (psc and stim_matrix are required)"); return CM_ERR_INVALID; }
  const int N = o->N, T = o->T, H = o->H, K = o->trials;
  if (design != CM_SIM_BLOCKWISE && design != CM_SIM_RANDOM) { cm_set_error("cm_simulate: design must be CM_SIM_BLOCKWISE or CM_SIM_RANDOM"); return CM_ERR_INVALID; }
  if (N < 2 || K < 1 || T < 2 || H < 1 || H > SIM_MAX_H || o->nreps < 1 || o->n_powers < 1 || o->n_powers > CM_SIM_MAX_POWERS ||
      N > CM_SIM_MAX_DIM || K > CM_SIM_MAX_DIM || !(o->gp_lengthscale > 0.0) || o->gp_lengthscale > 200.0) {
    cm_set_error("cm_simulate: need 2 <= N <= %d, 1 <= trials <= %d, T >= 2, 1 <= H <= %d, 1 <= n_powers <= %d, 0 < gp_lengthscale <= 200",
                 CM_SIM_MAX_DIM, CM_SIM_MAX_DIM, SIM_MAX_H, CM_SIM_MAX_POWERS);
    return CM_ERR_INVALID;
  }
  for (int i = 0; i < o->n_powers; ++i)
    if (!(o->powers[i] > 0.0)) { cm_set_error("cm_simulate: powers must be positive"); return CM_ERR_INVALID; }
  CM_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  SimDev s = {};
  s.o = *o;
  std::vector<double> pw(o->powers, o->powers + o->n_powers);
  std::sort(pw.begin(), pw.end(), [](double a, double b) { return a > b; });            // highest power first (:49)
  for (int i = 0; i < o->n_powers; ++i) s.powers_desc[i] = pw[i];
  s.K = K;
  s.nholo = (N + H - 1) / H;
  s.tpr = o->n_powers * s.nholo * o->nreps;
  s.R = (K + s.tpr - 1) / s.tpr;
  s.random_design = design;
  if (design) s.R = 0;
  s.n_conn = (int)(o->connection_prob * N);
  s.n_strong = (int)std::ceil(o->frac_strongly_connected * s.n_conn);
  s.J = (int)(GP_HALF_WIDTHS * o->gp_lengthscale);
  s.W = T + 2 * s.J;
  if ((size_t)(s.W + 2 * s.J + 1) * sizeof(double) > 60 * 1024) { cm_set_error("cm_simulate: T + 16 * gp_lengthscale too large for the trace kernel's LDS window"); return CM_ERR_INVALID; }
  std::vector<double> hf(2 * s.J + 1);
  {
    double ss = 0.0;
    for (int q = 0; q <= 2 * s.J; ++q) { const double j = q - s.J; hf[q] = std::exp(-(j * j) / (o->gp_lengthscale * o->gp_lengthscale)); ss += hf[q] * hf[q]; }
    for (double& v : hf) v /= std::sqrt(ss);
  }
  // one slab for every temporary and for the outputs the caller did not ask for / asked for on the host
  const size_t NK = (size_t)N * K, KT = (size_t)K * T;
  const bool host = (mem == CM_MEM_HOST);
  size_t nd = 0;                                                   // doubles
  auto take = [&](size_t n) { const size_t o0 = nd; nd += (n + 1) & ~(size_t)1; return o0; };
  const size_t o_area = take(N), o_scal = take(2), o_h = take(hf.size());
  const size_t o_w = take(N), o_p0 = take(N), o_p1 = take(N), o_tr = take(N), o_td = take(N);
  const size_t o_psc = host ? take(KT) : 0, o_stim = host ? take(NK) : 0;
  const size_t o_spks = (host && out->spks) ? take(NK) : 0, o_spt = (host && out->spk_times) ? take(NK) : 0;
  const size_t o_I = (host && out->I) ? take(K) : 0, o_gp = (host && out->gp_noise) ? take(KT) : 0, o_sp = (host && out->spont_pscs) ? take(KT) : 0;
  size_t ni = 0;                                                   // int32s
  auto takei = [&](size_t n) { const size_t o0 = ni; ni += (n + 1) & ~(size_t)1; return o0; };
  const size_t i_orders = takei((size_t)s.R * N), i_reorder = takei(K), i_conn = takei(N), i_rank = takei(N), i_cells = takei((size_t)K * H), i_err = takei(2);
  const size_t bytes = nd * sizeof(double) + ni * sizeof(int32_t) + NK;
  char* slab = nullptr;
  CM_HIP(hipMalloc(&slab, bytes));
  double* dd = reinterpret_cast<double*>(slab);
  int32_t* di = reinterpret_cast<int32_t*>(slab + nd * sizeof(double));
  s.spkb = reinterpret_cast<uint8_t*>(slab + nd * sizeof(double) + ni * sizeof(int32_t));
  s.area = dd + o_area; s.scal = dd + o_scal; s.hfilt = dd + o_h;
  s.orders = di + i_orders; s.reorder = di + i_reorder; s.conn = di + i_conn; s.rank = di + i_rank; s.cells = di + i_cells; s.err = di + i_err;
  // per-neuron vectors are always computed on the device slab (the kernels read them), then copied out if requested
  s.weights = dd + o_w; s.phi_0 = dd + o_p0; s.phi_1 = dd + o_p1; s.tau_r = dd + o_tr; s.tau_d = dd + o_td;
  s.psc = host ? dd + o_psc : out->psc;
  s.stim = host ? dd + o_stim : out->stim_matrix;
  s.spks = out->spks ? (host ? dd + o_spks : out->spks) : nullptr;
  s.spk_times = out->spk_times ? (host ? dd + o_spt : out->spk_times) : nullptr;
  s.I = out->I ? (host ? dd + o_I : out->I) : nullptr;
  s.gp_noise = out->gp_noise ? (host ? dd + o_gp : out->gp_noise) : nullptr;
  s.spont_pscs = out->spont_pscs ? (host ? dd + o_sp : out->spont_pscs) : nullptr;

  int rc = CM_OK;
#define SIM_HIP(call) do { hipError_t _e = (call); if (_e != hipSuccess) { cm_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(_e)); hipFree(slab); return CM_ERR_HIP; } } while (0)
  SIM_HIP(hipMemcpyAsync(s.hfilt, hf.data(), hf.size() * sizeof(double), hipMemcpyHostToDevice, st));
  SIM_HIP(hipMemsetAsync(s.stim, 0, NK * sizeof(double), st));
  SIM_HIP(hipMemsetAsync(s.err, 0, 2 * sizeof(int32_t), st));
  SIM_HIP(hipEventRecord(ctx->ev[0], st));
  int npad = 1;
  while (npad < std::max(N, K)) npad <<= 1;
  if (cm_raise_lds(ctx, CM_LDS_SIM_SORT, reinterpret_cast<const void*>(k_sim_sort), CM_SIM_MAX_DIM * 8, 0) != CM_OK) { hipFree(slab); return CM_ERR_HIP; }
  hipLaunchKernelGGL(k_sim_sort, dim3(s.R + 2), dim3(1024), (size_t)npad * sizeof(uint64_t), st, s);
  hipLaunchKernelGGL(k_sim_neurons, dim3(1), dim3(1024), 0, st, s);
  hipLaunchKernelGGL(k_sim_design, dim3((K + 3) / 4), dim3(256), 0, st, s);
  hipLaunchKernelGGL(k_sim_spikes, dim3(N), dim3(256), 0, st, s);
  if (s.spks || s.spk_times) hipLaunchKernelGGL(k_sim_expand_spks, dim3((unsigned)((NK + 255) / 256)), dim3(256), 0, st, s);
  hipLaunchKernelGGL(k_sim_traces, dim3(K), dim3(1024), (size_t)(s.W + 2 * s.J + 1) * sizeof(double), st, s);
  SIM_HIP(hipGetLastError());
  SIM_HIP(hipEventRecord(ctx->ev[1], st));
  int32_t h_err[2] = {0, 0};
  SIM_HIP(hipMemcpyAsync(h_err, s.err, sizeof(h_err), hipMemcpyDeviceToHost, st));
  auto give = [&](double* dst, const double* src, size_t n) -> hipError_t {
    if (!dst) return hipSuccess;
    return hipMemcpyAsync(dst, src, n * sizeof(double), host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, st);
  };
  SIM_HIP(give(out->weights, s.weights, N)); SIM_HIP(give(out->phi_0, s.phi_0, N)); SIM_HIP(give(out->phi_1, s.phi_1, N));
  SIM_HIP(give(out->tau_r, s.tau_r, N)); SIM_HIP(give(out->tau_d, s.tau_d, N));
  if (host) {
    SIM_HIP(give(out->psc, s.psc, KT)); SIM_HIP(give(out->stim_matrix, s.stim, NK));
    SIM_HIP(give(out->spks, s.spks, NK)); SIM_HIP(give(out->spk_times, s.spk_times, NK)); SIM_HIP(give(out->I, s.I, K));
    SIM_HIP(give(out->gp_noise, s.gp_noise, KT)); SIM_HIP(give(out->spont_pscs, s.spont_pscs, KT));
  }
  SIM_HIP(hipStreamSynchronize(st));
  if (device_ms) SIM_HIP(hipEventElapsedTime(device_ms, ctx->ev[0], ctx->ev[1]));
#undef SIM_HIP
  hipFree(slab);
  if (h_err[0]) { cm_set_error("cm_simulate: a neuron has more than %d silent maximal-power trials (padding step)", PAD_CAP); rc = CM_ERR_INVALID; }
  return rc;
}
