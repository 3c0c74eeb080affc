// Internal declarations shared by the HIP translation units of libcaviar_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/caviar_hip.h"

#define CM_WAVE 64
#define CM_TILE 32          // tile edge of the blocked SPD solve
#define CM_EV_PER_IT 10
#define CM_MAILBOX_WORDS 64  // int32 words of host-mapped pinned memory per context: [0..1] mailbox, [8..11] scratch

void cm_set_error(const char* fmt, ...);

#define CM_HIP(call)                                                                          \
  do {                                                                                        \
    hipError_t _e = (call);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      cm_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(_e));      \
      return CM_ERR_HIP;                                                                      \
    }                                                                                         \
  } while (0)

#define CM_TRY(expr)              \
  do {                            \
    int _s = (expr);              \
    if (_s != CM_OK) return _s;   \
  } while (0)

struct cm_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t side = nullptr;     // side stream: update_phi(it) -> MC logit(it + 1), overlapped with sigma/z and the next solve
  bool own_stream = false;
  hipEvent_t ev[8] = {};
  int32_t* h_pinned = nullptr;   // host-mapped pinned memory: words [0..1] = 64-bit mailbox (seq | flag | active-set size); [8..11] scratch for copies
  int32_t* d_mailbox = nullptr;  // device address of h_pinned
  int mail_seq = 0;
  int mb_n = 0, mb_flag = 0;     // last values read from the mailbox (active-set size, not-SPD flag)
  bool use_mailbox = true;
  int num_cus = 256;
  void* h_stage = nullptr;        // pinned staging buffer (priors upload, fn_scan round trip), grown on demand
  size_t h_stage_bytes = 0;
  size_t lds_attr[40] = {};       // dynamic-LDS caps already raised for this context's device, by kernel slot (cm_raise_lds)
  std::vector<hipEvent_t> pool;   // per-iteration events (CM_EV_PER_IT per iteration): phase stamps + cross-stream edges
};

// One circuit: encoded stimulus + observations + CAVIaR state.  All pointers are device memory.
struct cm_problem {
  cm_ctx* ctx = nullptr;
  int32_t N = 0, K = 0, T = 0;
  int64_t nnz = 0;       // number of targeted (stim > 0) entries
  int32_t V = 0;         // number of distinct stimulus values (sorted ascending; powers = vals[1:])
  int32_t ld = 0;        // leading dimension of the solve matrices (multiple of CM_TILE), sized for N
  int32_t max_row_len = 0; // longest CSR row (targeted trials of one neuron)
  int32_t max_col_len = 0; // most neurons targeted in one trial
  int32_t small_cap = 0;   // largest active set solved inside the fused mid-iteration kernel
  // --- static encoding (built once by cm_problem_create) ---
  double* y = nullptr;        // [K]  trapz(psc)
  double* ss = nullptr;       // [K]  sum psc^2
  double* sum_y = nullptr;    // [2]  sum(y), sum(y^2)
  double* vals = nullptr;     // [CM_MAX_VALS] sorted distinct stimulus values
  int32_t* row_ptr = nullptr; // [N+1] CSR over targeted entries
  int32_t* col = nullptr;     // [nnz] trial index
  uint8_t* grp = nullptr;     // [nnz] index into vals
  int32_t* rowcnt = nullptr;  // [N*CM_MAX_VALS] number of trials of row n with stim == vals[g] (targeted or not)
  int32_t* col_ptr = nullptr; // [K+1] CSC
  int32_t* csc_row = nullptr; // [nnz] neuron index
  int32_t* csc_pos = nullptr; // [nnz] position of the entry in CSR order
  // host mirror of the CSR (for reconnect_spont_cells and dense expansion)
  std::vector<int32_t> h_row_ptr, h_col, h_col_ptr, h_csc_row;
  std::vector<uint8_t> h_grp;
  std::vector<double> h_vals;
  std::vector<uint32_t> h_keys;   // host copy of the threefry key schedule (source of asynchronous uploads)
  // --- CAVIaR state ---
  uint8_t* mask = nullptr;    // [K] lam_mask
  double* slab = nullptr;     // one allocation: mu | beta | z | lam (so the false-negative scan moves them in one copy each way)
  double* pslab = nullptr;    // one allocation: mu_prior | beta_prior | phi_prior | phi_cov_prior | scal
  double* lam = nullptr;      // [nnz] CSR order
  double* mu = nullptr;       // [N]
  double* beta = nullptr;     // [N] posterior variance (reference's "beta")
  double* mu_prior = nullptr; // [N]
  double* beta_prior = nullptr; // [N]
  double* phi = nullptr;      // [N*2]
  double* phi_cov = nullptr;  // [N*4]
  double* phi_prior = nullptr;
  double* phi_cov_prior = nullptr;
  double* scal = nullptr;     // [16] device scalars, see CM_S_*
  double* R = nullptr;        // [K]  sum_n mu_n lam_nk
  uint8_t* colany = nullptr;  // [K]  any_n lam_nk >= spont_orthogonality
  double* z = nullptr;        // [K]
  double* mcE = nullptr;      // [N*CM_MAX_VALS] Monte-Carlo logit per (neuron, stimulus value)
  double* rs = nullptr;       // [N*4] row stats: sum lam, sum lam^2, sum y*lam, (unused)
  int32_t* act_idx = nullptr; // [N] active neurons (compact list)
  int32_t* act_inv = nullptr; // [N] neuron -> compact index or -1
  int32_t* n_act = nullptr;   // [4] device: N_a, flags
  int32_t* order = nullptr;   // sweep order of the current iteration (points into orders_all)
  int32_t* orders_all = nullptr; // [iters*N] sweep orders of every iteration (device generated, or the explicit stream)
  int32_t* orders_tmp = nullptr; // [iters*N] ping-pong buffer of the shuffle rounds
  int32_t* inv_all = nullptr;    // [iters*N] inverse of every sweep order: position of neuron n in iteration it
  double* wtab = nullptr;        // [iters*N*2S] ndtri(u) of every uniform of the run (data independent)
  size_t wtab_cap = 0;
  int32_t orders_cap = 0;     // iterations the two buffers are sized for
  int32_t* seq_n = nullptr;   // [N]   neurons with mu != 0 in sweep order
  int32_t* seq_off = nullptr; // [N+1] record offset of each such row in `packed`
  int32_t* seq_pos = nullptr; // [N]   position of neuron n in the chain, -1 if mu_n == 0
  int32_t* chunk_first = nullptr; // [n_chunk_cap] first row of each LDS chunk
  int32_t* seq_meta = nullptr;// [4]   nseq, total records, nchunks
  void* packed = nullptr;     // [(N + nnz) * 32 B] packed row records of the sequential chain
  uint32_t* keys = nullptr;   // [iters*(N+3)*2] threefry key schedule (host computed)
  double* uexp = nullptr;     // explicit uniforms (device copy) or nullptr
  double* A = nullptr;        // [ld*ld] precision matrix / Cholesky factor
  double* X = nullptr;        // [ld*ld] inverse factor
  double* Lf = nullptr;       // [ld*ld] Cholesky factor (panel tiles below the diagonal)
  double* Dinv = nullptr;     // [ld*CM_TILE] inverses of the diagonal tiles
  double* rhs = nullptr;      // [ld*2]
  // histories
  int32_t hist_iters = 0;
  double *h_mu = nullptr, *h_beta = nullptr, *h_lam = nullptr, *h_shape = nullptr, *h_rate = nullptr,
         *h_phi = nullptr, *h_phi_cov = nullptr, *h_z = nullptr;
  bool ran = false;
  int32_t keys_iters = 0;
};

enum {
  CM_S_SHAPE = 0, CM_S_RATE = 1, CM_S_SPONT_RATE = 2, CM_S_SHAPE_PRIOR = 3, CM_S_RATE_PRIOR = 4, CM_S_ERR = 5,
  CM_S_COUNT = 16
};

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
#ifdef __HIPCC__
// ---- DPP cross-lane adds (VALU data path; no LDS crossbar).  All 64 lanes must be active at the call site.
#define CM_DPP_XOR1 0xB1          // quad_perm [1,0,3,2]
#define CM_DPP_XOR2 0x4E          // quad_perm [2,3,0,1]
#define CM_DPP_HALF_MIRROR 0x141  // lane i <-> 7 - i within 8
#define CM_DPP_MIRROR 0x140       // lane i <-> 15 - i within 16
template <int CTRL>
__device__ __forceinline__ double cm_dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double cm_readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
// sum over aligned segments of SEG lanes (SEG = 1,2,4,8,16 by DPP; 32, 64 add LDS-crossbar steps); every lane of a
// segment receives the segment's sum; fixed association -> deterministic
template <int SEG>
__device__ __forceinline__ double cm_seg_sum(double v) {
  if (SEG >= 2) v += cm_dpp_f64<CM_DPP_XOR1>(v);
  if (SEG >= 4) v += cm_dpp_f64<CM_DPP_XOR2>(v);
  if (SEG >= 8) v += cm_dpp_f64<CM_DPP_HALF_MIRROR>(v);
  if (SEG >= 16) v += cm_dpp_f64<CM_DPP_MIRROR>(v);
  if (SEG >= 32) v += __shfl_xor(v, 16, CM_WAVE);
  if (SEG >= 64) v += __shfl_xor(v, 32, CM_WAVE);
  return v;
}
// full-wave sum, result uniform: four DPP steps give every 16-lane row its sum, the four row sums are read to scalars
__device__ __forceinline__ double cm_wave_sum(double v) {
  v = cm_seg_sum<16>(v);
  const double r0 = cm_readlane_f64(v, 0), r1 = cm_readlane_f64(v, 16), r2 = cm_readlane_f64(v, 32), r3 = cm_readlane_f64(v, 48);
  return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ int cm_wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, CM_WAVE);
  return v;
}
__device__ __forceinline__ double cm_wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, CM_WAVE));
  return v;
}

// jax.nn.sigmoid / scipy expit: 1 / (1 + exp(-x))
__device__ __forceinline__ double cm_sigmoid(double x) { return 1.0 / (1.0 + exp(-x)); }

// numpy nan_to_num for float64: nan -> 0, +-inf -> +-DBL_MAX
__device__ __forceinline__ double cm_nan_to_num(double x) {
  if (isnan(x)) return 0.0;
  if (isinf(x)) return x > 0 ? 1.7976931348623157e308 : -1.7976931348623157e308;
  return x;
}

// Threefry-2x32, 20 rounds (Salmon et al. 2011), the block function behind jax.random in the 0.3 series.
__host__ __device__ __forceinline__ void cm_threefry2x32(uint32_t k0, uint32_t k1, uint32_t& x0, uint32_t& x1) {
  const uint32_t ks0 = k0, ks1 = k1, ks2 = k0 ^ k1 ^ 0x1BD11BDAu;
#define CM_TF_R(r)                      \
  x0 += x1;                             \
  x1 = (x1 << (r)) | (x1 >> (32 - (r))); \
  x1 ^= x0;
  x0 += ks0; x1 += ks1;
  CM_TF_R(13) CM_TF_R(15) CM_TF_R(26) CM_TF_R(6)
  x0 += ks1; x1 += ks2 + 1u;
  CM_TF_R(17) CM_TF_R(29) CM_TF_R(16) CM_TF_R(24)
  x0 += ks2; x1 += ks0 + 2u;
  CM_TF_R(13) CM_TF_R(15) CM_TF_R(26) CM_TF_R(6)
  x0 += ks0; x1 += ks1 + 3u;
  CM_TF_R(17) CM_TF_R(29) CM_TF_R(16) CM_TF_R(24)
  x0 += ks1; x1 += ks2 + 4u;
  CM_TF_R(13) CM_TF_R(15) CM_TF_R(26) CM_TF_R(6)
  x0 += ks2; x1 += ks0 + 5u;
#undef CM_TF_R
}
#else
inline void cm_threefry2x32(uint32_t k0, uint32_t k1, uint32_t& x0, uint32_t& x1) {
  const uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
  static const int R[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
  x0 += ks[0]; x1 += ks[1];
  for (int g = 0; g < 5; ++g) {
    for (int i = 0; i < 4; ++i) { int r = R[g % 2][i]; x0 += x1; x1 = (x1 << r) | (x1 >> (32 - r)); x1 ^= x0; }
    x0 += ks[(g + 1) % 3]; x1 += ks[(g + 2) % 3] + (uint32_t)(g + 1);
  }
}
#endif

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, device), and setting it REPLACES the previous value (it is
// not a running maximum).  Several contexts can live on one device (cm_batch, fit_batch workers), so the value in force is kept
// in ONE process-wide table keyed (device, kernel) under a mutex and is only ever raised; the per-context array is a cache of
// lower bounds of that table (valid because the table never decreases), so the common launch pays no lock and no runtime call.
enum cm_lds_slot { CM_LDS_ITER = 0 /* 7 x 2 x 2 template variants */, CM_LDS_SOLVE_SMALL = 28, CM_LDS_SWEEP_PAR, CM_LDS_SWEEP_PAR2, CM_LDS_SWEEP_SEQ,
                   CM_LDS_NWD, CM_LDS_SIM_SORT, CM_LDS_BUILD_PREC, CM_LDS_SLOTS = 40 };
// table update: returns the value hipFuncSetAttribute has to be called with (0 = the value in force already covers `bytes`)
size_t cm_lds_table_request(int device, const void* fn, size_t bytes, size_t floor_bytes);
inline int cm_raise_lds(cm_ctx* ctx, int slot, const void* fn, size_t bytes, size_t floor_bytes) {
  if (bytes <= floor_bytes || bytes <= ctx->lds_attr[slot]) return CM_OK;
  const size_t set = cm_lds_table_request(ctx->device, fn, bytes, floor_bytes);
  if (set) CM_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)set));
  ctx->lds_attr[slot] = bytes;
  return CM_OK;
}

// ------------------------------------------------------------------------------------------------
// phase launchers (each enqueues kernels on ctx->stream; no host sync unless stated)
// ------------------------------------------------------------------------------------------------
int cm_launch_psc_prologue(cm_ctx* ctx, int K, int T, const double* d_psc, double* d_y, double* d_ss);
int cm_encode_stim(cm_ctx* ctx, cm_problem* p, const double* d_stim);               // syncs (sizes nnz)
int cm_launch_init_state(cm_ctx* ctx, cm_problem* p, double y_xcorr_thresh);
int cm_launch_scatter_dense_lam(cm_ctx* ctx, cm_problem* p, const double* d_lam_dense);
int cm_launch_expand_lam(cm_ctx* ctx, cm_problem* p, const double* d_lam_sparse, double* d_dense);

int cm_solve_mu(cm_ctx* ctx, cm_problem* p, int* n_active_out);                      // syncs once (active-set size)
int cm_solve_launch(cm_ctx* ctx, cm_problem* p, int na);
int cm_small_cap(const cm_problem* p);
int cm_launch_iter_mid(cm_ctx* ctx, cm_problem* p, int it, int iters, int S, int rng_mode, const cm_options* o, int do_sigma, int do_phi);
int cm_launch_ndtri_table(cm_ctx* ctx, cm_problem* p, int it0, int n_it, int S, int rng_mode, hipStream_t st);
void cm_host_key_schedule_range(uint32_t key[2], int it0, int n_it, int N, uint32_t* keys);
int cm_launch_rng(cm_ctx* ctx, cm_problem* p, int it0, int n_it, hipStream_t st);
int cm_launch_mc_logit(cm_ctx* ctx, cm_problem* p, int it, int S, int rng_mode, hipStream_t st);
int cm_launch_sweep(cm_ctx* ctx, cm_problem* p, int it, const cm_options* o, hipEvent_t mid);
int cm_launch_stats(cm_ctx* ctx, cm_problem* p, double spont_orthogonality);
int cm_launch_post_single(cm_ctx* ctx, cm_problem* p, int it, const cm_options* o, int do_sigma);
int cm_wait_active(cm_ctx* ctx, cm_problem* p);
int cm_launch_update_phi(cm_ctx* ctx, cm_problem* p, hipStream_t st);
int cm_launch_col_stats(cm_ctx* ctx, cm_problem* p, double spont_orthogonality);

void cm_host_key_schedule(uint64_t seed, int iters, int N, std::vector<uint32_t>& keys);
int cm_perm_rounds(int N);
double cm_host_pava_last(const double* y, int n);
