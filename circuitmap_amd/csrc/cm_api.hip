// C ABI of libcaviar_hip.so (include/caviar_hip.h): context / problem lifecycle, the caviar() driver loop
// (reference circuitmap/optimise/caviar.py:20-100) and the host-side false-negative scan (caviar.py:102-144).
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <mutex>
#include <queue>
#include <utility>
#include <vector>

#include "cm_internal.h"

int cm_k_rng_streams_impl(cm_ctx* ctx, cm_problem* p, int it, int S, double* d_u);

static thread_local char g_err[512] = "";

void cm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* cm_last_error(void) { return g_err; }

// dynamic-LDS caps in force, per (device, kernel): only ever raised (see cm_raise_lds, cm_internal.h)
static std::mutex g_lds_mu;
static std::map<std::pair<int, const void*>, size_t> g_lds_table;
size_t cm_lds_table_request(int device, const void* fn, size_t bytes, size_t floor_bytes) {
  std::lock_guard<std::mutex> lk(g_lds_mu);
  size_t& slot = g_lds_table[std::make_pair(device, fn)];
  const size_t cur = slot > floor_bytes ? slot : floor_bytes;
  if (bytes <= cur) return 0;
  slot = bytes;
  return bytes;
}
extern "C" int64_t cm_test_lds_request(int32_t device, uint64_t kernel_key, uint64_t bytes, uint64_t floor_bytes) {
  // keys of this hook live in their own range of the table: odd addresses, which no kernel has
  return (int64_t)cm_lds_table_request(device, reinterpret_cast<const void*>((uintptr_t)(2 * kernel_key + 1)), (size_t)bytes, (size_t)floor_bytes);
}
extern "C" const char* cm_version(void) { return "caviar_hip 0.1 (gfx950)"; }

extern "C" int cm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" void cm_options_default(cm_options* o) {   // caviar.py:21-23
  memset(o, 0, sizeof(*o));
  o->iters = 50; o->num_mc_samples = 100; o->seed = 0; o->y_xcorr_thresh = 1e-2; o->minimum_spike_count = 3;
  o->delay_spont_est = 1; o->msrmp = 0.3; o->scale_factor = 0.75; o->penalty = 5e0; o->save_histories = 0;
  o->max_backtrack_iters = 20; o->tol = 0.05; o->spont_orthogonality = 0.1; o->fn_scan = 1;
  o->rng_mode = CM_RNG_THREEFRY; o->update_orders = nullptr; o->uniforms = nullptr; o->verbose = 0; o->timing = 0;
}

extern "C" int cm_ctx_destroy(cm_ctx* c);
static int ctx_init(cm_ctx* c, int device_id, void* hip_stream) {
  if (hip_stream) { c->stream = (hipStream_t)hip_stream; c->own_stream = false; }
  else { CM_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
  CM_HIP(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
  for (auto& e : c->ev) CM_HIP(hipEventCreate(&e));
  CM_HIP(hipHostMalloc((void**)&c->h_pinned, CM_MAILBOX_WORDS * sizeof(int32_t), hipHostMallocMapped | hipHostMallocCoherent));
  memset(c->h_pinned, 0, CM_MAILBOX_WORDS * sizeof(int32_t));
  CM_HIP(hipHostGetDevicePointer((void**)&c->d_mailbox, c->h_pinned, 0));
  c->use_mailbox = getenv("CAVIAR_HIP_NO_MAILBOX") == nullptr;
  hipDeviceProp_t prop;
  CM_HIP(hipGetDeviceProperties(&prop, device_id));
  c->num_cus = prop.multiProcessorCount;
  return CM_OK;
}

extern "C" int cm_ctx_create(int device_id, void* hip_stream, cm_ctx** out) {
  if (!out) { cm_set_error("cm_ctx_create: out is NULL"); return CM_ERR_INVALID; }
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
    cm_set_error("no HIP device visible: libcaviar_hip has no CPU fallback");
    return CM_ERR_NO_DEVICE;
  }
  if (device_id < 0 || device_id >= n) { cm_set_error("device_id %d out of range (%d devices)", device_id, n); return CM_ERR_INVALID; }
  CM_HIP(hipSetDevice(device_id));
  cm_ctx* c = new cm_ctx();
  c->device = device_id;
  const int rc = ctx_init(c, device_id, hip_stream);
  if (rc != CM_OK) { cm_ctx_destroy(c); return rc; }      // a half-built context is released, not leaked
  *out = c;
  return CM_OK;
}

extern "C" int cm_ctx_destroy(cm_ctx* c) {
  if (!c) return CM_OK;
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  if (c->side) { hipStreamSynchronize(c->side); hipStreamDestroy(c->side); }
  for (auto& e : c->ev) if (e) hipEventDestroy(e);
  for (auto& e : c->pool) hipEventDestroy(e);
  if (c->h_pinned) hipHostFree(c->h_pinned);
  if (c->h_stage) hipHostFree(c->h_stage);
  if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
  delete c;
  return CM_OK;
}

extern "C" int cm_ctx_synchronize(cm_ctx* c) {
  if (!c) return CM_ERR_INVALID;
  CM_HIP(hipStreamSynchronize(c->stream));
  return CM_OK;
}

extern "C" int cm_mem_alloc(cm_ctx* c, uint64_t bytes, void** out) {
  if (!c || !out) { cm_set_error("cm_mem_alloc: NULL argument"); return CM_ERR_INVALID; }
  CM_HIP(hipSetDevice(c->device));
  *out = nullptr;
  if (bytes == 0) return CM_OK;
  CM_HIP(hipMalloc(out, (size_t)bytes));
  return CM_OK;
}
extern "C" int cm_mem_free(cm_ctx* c, void* ptr) {
  if (!c) { cm_set_error("cm_mem_free: NULL context"); return CM_ERR_INVALID; }
  if (!ptr) return CM_OK;
  CM_HIP(hipSetDevice(c->device));
  CM_HIP(hipStreamSynchronize(c->stream));
  CM_HIP(hipFree(ptr));
  return CM_OK;
}
extern "C" int cm_mem_copy(cm_ctx* c, void* dst, const void* src, uint64_t bytes, int kind) {
  if (!c || (bytes && (!dst || !src))) { cm_set_error("cm_mem_copy: NULL argument"); return CM_ERR_INVALID; }
  if (kind < 1 || kind > 3) { cm_set_error("cm_mem_copy: kind must be 1 (H2D), 2 (D2H) or 3 (D2D)"); return CM_ERR_INVALID; }
  if (bytes == 0) return CM_OK;
  CM_HIP(hipSetDevice(c->device));
  const hipMemcpyKind k = kind == 1 ? hipMemcpyHostToDevice : (kind == 2 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice);
  CM_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, k, c->stream));
  CM_HIP(hipStreamSynchronize(c->stream));
  return CM_OK;
}

template <typename T>
static int dmalloc(T** p, size_t n) {
  CM_HIP(hipMalloc((void**)p, (n ? n : 1) * sizeof(T)));
  return CM_OK;
}

static void free_hist(cm_problem* p) {
  double** h[] = {&p->h_mu, &p->h_beta, &p->h_lam, &p->h_shape, &p->h_rate, &p->h_phi, &p->h_phi_cov, &p->h_z};
  for (auto q : h) { if (*q) hipFree(*q); *q = nullptr; }
  p->hist_iters = 0;
}

extern "C" int cm_problem_destroy(cm_problem* p) {
  if (!p) return CM_OK;
  hipSetDevice(p->ctx->device);
  hipStreamSynchronize(p->ctx->stream);
  void* ptrs[] = {p->y, p->ss, p->sum_y, p->vals, p->row_ptr, p->col, p->grp, p->rowcnt, p->col_ptr, p->csc_row, p->csc_pos,
                  p->mask, p->slab, p->pslab, p->phi, p->phi_cov,
                  p->R, p->colany, p->mcE, p->rs, p->act_idx, p->act_inv, p->n_act, p->orders_all, p->orders_tmp, p->inv_all, p->wtab,
                  p->keys, p->uexp, p->A, p->X, p->Lf, p->Dinv, p->rhs, p->seq_n, p->seq_off, p->seq_pos, p->chunk_first,
                  p->seq_meta, p->packed};
  for (void* q : ptrs) if (q) hipFree(q);
  free_hist(p);
  delete p;
  return CM_OK;
}

extern "C" int cm_problem_create(cm_ctx* ctx, int32_t N, int32_t K, int32_t T, const double* psc, const double* stim, int mem_kind,
                                 cm_problem** out) {
  if (!ctx || !psc || !stim || !out) { cm_set_error("cm_problem_create: NULL argument"); return CM_ERR_INVALID; }
  if (N < 2 || K < 1 || T < 1) { cm_set_error("cm_problem_create: need N >= 2, K >= 1, T >= 1 (got %d, %d, %d)", N, K, T); return CM_ERR_INVALID; }
  CM_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  cm_problem* p = new cm_problem();
  p->ctx = ctx; p->N = N; p->K = K; p->T = T;
  int rc = CM_OK;
  double *d_psc = nullptr, *d_stim = nullptr;
  auto fail = [&](int code) { if (mem_kind == CM_MEM_HOST) { if (d_psc) hipFree(d_psc); if (d_stim) hipFree(d_stim); } cm_problem_destroy(p); return code; };
  if (mem_kind == CM_MEM_HOST) {
    if ((rc = dmalloc(&d_psc, (size_t)K * T)) || (rc = dmalloc(&d_stim, (size_t)N * K))) return fail(rc);
    if (hipMemcpyAsync(d_psc, psc, (size_t)K * T * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(d_stim, stim, (size_t)N * K * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess) {
      cm_set_error("upload of psc/stim failed");
      return fail(CM_ERR_HIP);
    }
  } else {
    d_psc = const_cast<double*>(psc);
    d_stim = const_cast<double*>(stim);
  }
  if ((rc = dmalloc(&p->y, K)) || (rc = dmalloc(&p->ss, K)) || (rc = dmalloc(&p->sum_y, 2))) return fail(rc);
  if ((rc = cm_launch_psc_prologue(ctx, K, T, d_psc, p->y, p->ss))) return fail(rc);
  if ((rc = cm_encode_stim(ctx, p, d_stim))) return fail(rc);
  if (mem_kind == CM_MEM_HOST) { hipFree(d_psc); hipFree(d_stim); d_psc = d_stim = nullptr; }

  p->ld = ((N + CM_TILE - 1) / CM_TILE) * CM_TILE;
  const size_t ld = p->ld;
  if ((rc = dmalloc(&p->mask, K)) || (rc = dmalloc(&p->slab, 2 * (size_t)N + K + (size_t)p->nnz)) ||
      (rc = dmalloc(&p->pslab, 8 * (size_t)N + CM_S_COUNT)) || (rc = dmalloc(&p->phi, 2 * (size_t)N)) ||
      (rc = dmalloc(&p->phi_cov, 4 * (size_t)N)) || (rc = dmalloc(&p->R, K)) ||
      (rc = dmalloc(&p->colany, K)) || (rc = dmalloc(&p->mcE, (size_t)N * CM_MAX_VALS)) ||
      (rc = dmalloc(&p->rs, 4 * (size_t)N)) || (rc = dmalloc(&p->act_idx, N)) || (rc = dmalloc(&p->act_inv, N)) ||
      (rc = dmalloc(&p->n_act, 4)) || (rc = dmalloc(&p->A, ld * ld)) || (rc = dmalloc(&p->X, ld * ld)) || (rc = dmalloc(&p->Lf, ld * ld)) ||
      (rc = dmalloc(&p->Dinv, ld * CM_TILE)) || (rc = dmalloc(&p->rhs, 2 * ld)) || (rc = dmalloc(&p->seq_n, N)) ||
      (rc = dmalloc(&p->seq_off, (size_t)N + 1)) || (rc = dmalloc(&p->seq_pos, N)) || (rc = dmalloc(&p->chunk_first, ((size_t)N + p->nnz) / 256 + 8)) ||
      (rc = dmalloc(&p->seq_meta, 8)) || (rc = dmalloc((char**)&p->packed, ((size_t)N + p->nnz) * 32)))
    return fail(rc);
  p->mu = p->slab; p->beta = p->slab + N; p->z = p->slab + 2 * (size_t)N; p->lam = p->slab + 2 * (size_t)N + K;
  p->mu_prior = p->pslab; p->beta_prior = p->pslab + N; p->phi_prior = p->pslab + 2 * (size_t)N;
  p->phi_cov_prior = p->pslab + 4 * (size_t)N; p->scal = p->pslab + 8 * (size_t)N;
  if (hipMemsetAsync(p->mcE, 0, (size_t)N * CM_MAX_VALS * sizeof(double), st) != hipSuccess) return fail(CM_ERR_HIP);
  if (hipStreamSynchronize(st) != hipSuccess) { cm_set_error("prologue failed: %s", hipGetErrorString(hipGetLastError())); return fail(CM_ERR_HIP); }
  *out = p;
  return CM_OK;
}

extern "C" int cm_problem_nnz(const cm_problem* p, int64_t* nnz, int32_t* n_vals) {
  if (!p) return CM_ERR_INVALID;
  if (nnz) *nnz = p->nnz;
  if (n_vals) *n_vals = p->V;
  return CM_OK;
}

// ------------------------------------------------------------------------------------------------
// host side of the driver
// ------------------------------------------------------------------------------------------------
double cm_host_pava_last(const double* y, int n) {   // pava.py:9-88, gamma = 1, literal pool walk
  if (n <= 0) return 0.0;
  std::vector<double> v(n), w(n);
  int i = 0;
  v[0] = y[0]; w[0] = 1.0;
  for (int t = 1; t < n; ++t) {
    ++i;
    v[i] = y[t]; w[i] = 1.0;
    while (i > 0 && (v[i - 1] / w[i - 1]) > (v[i] / w[i])) {
      --i;
      v[i] += v[i + 1];
      w[i] += w[i + 1];
    }
  }
  return v[i] / w[i];
}

extern "C" int cm_pava_last(const double* y, int32_t n, double* out) {
  if (!y || !out || n < 1) { cm_set_error("cm_pava_last: bad argument"); return CM_ERR_INVALID; }
  *out = cm_host_pava_last(y, n);
  return CM_OK;
}

static int stage_reserve(cm_ctx* ctx, size_t bytes) {
  if (ctx->h_stage_bytes >= bytes) return CM_OK;
  if (ctx->h_stage) hipHostFree(ctx->h_stage);
  ctx->h_stage = nullptr; ctx->h_stage_bytes = 0;
  CM_HIP(hipHostMalloc(&ctx->h_stage, bytes, hipHostMallocDefault));
  ctx->h_stage_bytes = bytes;
  return CM_OK;
}

// priors -> device in ONE pinned transfer (mu_prior | beta_prior | phi_prior | phi_cov_prior | scalars), then the state is
// initialised from them on the device (caviar.py:44-49)
__global__ __launch_bounds__(256) void k_state_from_priors(const double* __restrict__ pslab, int N, double* __restrict__ mu,
                                                            double* __restrict__ beta, double* __restrict__ phi, double* __restrict__ phi_cov) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) { mu[i] = pslab[i]; beta[i] = pslab[N + i]; }
  if (i < 2 * N) phi[i] = pslab[2 * (size_t)N + i];
  if (i < 4 * N) phi_cov[i] = pslab[4 * (size_t)N + i];
}

static int upload_priors(cm_ctx* ctx, cm_problem* p, const cm_priors* pr) {
  hipStream_t st = ctx->stream;
  const size_t N = p->N, n = 8 * N + CM_S_COUNT;
  CM_HIP(hipStreamSynchronize(st));                 // the staging buffer may still feed an earlier transfer of this context
  CM_TRY(stage_reserve(ctx, std::max(n, 2 * N + (size_t)p->K + (size_t)p->nnz) * sizeof(double)));
  double* h = (double*)ctx->h_stage;
  memcpy(h, pr->mu, N * 8); memcpy(h + N, pr->beta, N * 8); memcpy(h + 2 * N, pr->phi, 2 * N * 8); memcpy(h + 4 * N, pr->phi_cov, 4 * N * 8);
  double* sc = h + 8 * N;
  for (int i = 0; i < CM_S_COUNT; ++i) sc[i] = 0.0;
  sc[CM_S_SHAPE] = pr->shape; sc[CM_S_RATE] = pr->rate; sc[CM_S_SHAPE_PRIOR] = pr->shape; sc[CM_S_RATE_PRIOR] = pr->rate;
  CM_HIP(hipMemcpyAsync(p->pslab, h, n * sizeof(double), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_state_from_priors, dim3((unsigned)((4 * N + 255) / 256)), dim3(256), 0, st, p->pslab, (int)N, p->mu, p->beta, p->phi, p->phi_cov);
  CM_HIP(hipGetLastError());
  return CM_OK;
}

// reconnect_spont_cells, caviar.py:102-144.  Host C++ on the compressed rows (one-off, after the last iteration).
static void host_reconnect(const cm_problem* p, double* mu, double* beta, double* lam, double* z, double minimax_spk_prob,
                           double minimum_spike_count, int verbose) {
  const int N = p->N, K = p->K, V = p->V;
  const int P = V - 1;
  std::vector<int> disc;
  for (int n = 0; n < N; ++n) if (mu[n] == 0.0) disc.push_back(n);           // :103 exact zero test (quirk 7)
  const size_t init = disc.size();
  // np.where(stim_matrix[n]) selects NON-ZERO entries; targeted entries (> 0) are the CSR row.  Negative stimulus values
  // are outside the reference's contract (powers are non-negative) and are treated as untargeted here.
  // The reference re-scans all remaining cells in every pass (:109-117): it takes the cell with the most spontaneous events on
  // its targeted trials (first maximum in ascending cell order), tests it, and removes it.  Counts only ever decrease (a
  // reconnection clears events from z), so a max-heap keyed (count desc, cell asc) with lazy refresh of stale entries visits
  // the cells in exactly that order; a reconnection updates the counts of the cells sharing the cleared trials (host CSC).
  long nz = 0;
  for (int k = 0; k < K; ++k) nz += (z[k] != 0.0);
  std::vector<long> cnt(N, -1);                         // -1: not a candidate (connected, or already visited)
  struct Ent { long c; int n; };
  auto worse = [](const Ent& a, const Ent& b) { return a.c < b.c || (a.c == b.c && a.n > b.n); };
  std::priority_queue<Ent, std::vector<Ent>, decltype(worse)> heap(worse);
  for (int n : disc) {
    long c = 0;
    for (int e = p->h_row_ptr[n]; e < p->h_row_ptr[n + 1]; ++e) c += (z[p->h_col[e]] != 0.0);
    cnt[n] = c;
    // A cell can only reconnect with spike_count >= minimum_spike_count (:130), spike_count is exactly this count (every targeted
    // entry belongs to one power), and counts only decrease: cells below the threshold are examined by the reference to no
    // effect, so they never enter the heap (identical outcome; most cells have 0 or 1 spontaneous events on their trials).
    if (!(minimum_spike_count > 0.0) || (double)c >= minimum_spike_count) heap.push({c, n});
    else cnt[n] = -1;
  }
  std::vector<double> srates(P > 0 ? P : 0);
  std::vector<int> locs;
  while (!heap.empty()) {
    if (!((double)nz > minimum_spike_count)) break;                           // :111,141
    const Ent top = heap.top();
    heap.pop();
    if (cnt[top.n] < 0) continue;                                             // already visited
    if (top.c != cnt[top.n]) {                                                // stale: re-queue with the current count
      if (!(minimum_spike_count > 0.0) || (double)cnt[top.n] >= minimum_spike_count) heap.push({cnt[top.n], top.n});
      else cnt[top.n] = -1;
      continue;
    }
    const int focus = top.n;
    double spike_count = 0.0;
    for (int i = 0; i < P; ++i) {                                             // :121-127
      long c = 0, hit = 0;
      for (int e = p->h_row_ptr[focus]; e < p->h_row_ptr[focus + 1]; ++e)
        if (p->h_grp[e] == i + 1) { ++c; hit += (z[p->h_col[e]] != 0.0); }
      srates[i] = c > 0 ? (double)hit / (double)c : 0.0;
      if (c > 0) spike_count += (double)hit;
    }
    const double pava = P > 0 ? cm_host_pava_last(srates.data(), P) : 0.0;    // :128
    cnt[focus] = -1;                                                          // :139 (removed whether or not it reconnects)
    if (pava >= minimax_spk_prob && spike_count >= minimum_spike_count) {     // :130
      locs.clear();
      for (int e = p->h_row_ptr[focus]; e < p->h_row_ptr[focus + 1]; ++e) if (z[p->h_col[e]] != 0.0) locs.push_back(e);
      const size_t m = locs.size();
      double s = 0.0;
      for (int e : locs) s += z[p->h_col[e]];
      const double mean = s / (double)m;
      double ssd = 0.0;
      for (int e : locs) { const double d = z[p->h_col[e]] - mean; ssd += d * d; }
      mu[focus] = mean;                                                        // :134
      beta[focus] = m > 1 ? sqrt(ssd / (double)(m - 1)) / sqrt((double)m) : NAN;   // :135 scipy.stats.sem, ddof = 1
      for (int e : locs) {                                                     // :136-137
        const int k = p->h_col[e];
        lam[e] = 1.0; z[k] = 0.0;
        for (int ce = p->h_col_ptr[k]; ce < p->h_col_ptr[k + 1]; ++ce) { const int mrow = p->h_csc_row[ce]; if (cnt[mrow] > 0) --cnt[mrow]; }
      }
      nz -= (long)m;
      if (verbose) fprintf(stderr, "Reconnecting cell %d with maximal pava spike rate %.2f\n", focus, pava);
    }
  }
  if (verbose) fprintf(stderr, "Examined %zu/%d cells for false negatives. Cell reconnection complete.\n", init, N);
}

static int alloc_hist(cm_problem* p, int iters) {
  free_hist(p);
  const size_t it = iters, N = p->N, K = p->K;
  CM_TRY(dmalloc(&p->h_mu, it * N)); CM_TRY(dmalloc(&p->h_beta, it * N)); CM_TRY(dmalloc(&p->h_lam, it * (size_t)p->nnz));
  CM_TRY(dmalloc(&p->h_shape, it)); CM_TRY(dmalloc(&p->h_rate, it)); CM_TRY(dmalloc(&p->h_phi, it * 2 * N));
  CM_TRY(dmalloc(&p->h_phi_cov, it * 4 * N)); CM_TRY(dmalloc(&p->h_z, it * K));
  p->hist_iters = iters;
  return CM_OK;
}

static int record_hist(cm_ctx* ctx, cm_problem* p, int it, int part) {
  hipStream_t st = ctx->stream;
  const size_t N = p->N, K = p->K, nnz = p->nnz;
  const hipMemcpyKind k = hipMemcpyDeviceToDevice;
  if (part == 0) {     // before the fused mid-iteration kernel overwrites mu / beta with the next iteration's solve
    CM_HIP(hipMemcpyAsync(p->h_mu + it * N, p->mu, N * 8, k, st));
    CM_HIP(hipMemcpyAsync(p->h_beta + it * N, p->beta, N * 8, k, st));
    if (nnz) CM_HIP(hipMemcpyAsync(p->h_lam + it * nnz, p->lam, nnz * 8, k, st));
    return CM_OK;
  }
  CM_HIP(hipMemcpyAsync(p->h_shape + it, p->scal + CM_S_SHAPE, 8, k, st));
  CM_HIP(hipMemcpyAsync(p->h_rate + it, p->scal + CM_S_RATE, 8, k, st));
  CM_HIP(hipMemcpyAsync(p->h_phi + it * 2 * N, p->phi, 2 * N * 8, k, st));
  CM_HIP(hipMemcpyAsync(p->h_phi_cov + it * 4 * N, p->phi_cov, 4 * N * 8, k, st));
  CM_HIP(hipMemcpyAsync(p->h_z + it * K, p->z, K * 8, k, st));
  return CM_OK;
}

static int caviar_run_impl(cm_ctx* ctx, cm_problem* p, const cm_priors* pr, const cm_options* o, cm_timings* tm);

// Single exit path: an error return (bad argument, HIP failure, not-SPD, ...) may leave kernels and asynchronous copies queued on
// both streams that still read host arrays of the CALLER (explicit RNG streams) or of this problem (key schedule); both streams
// are drained before the error is handed back, so that nothing outlives the buffers it reads.
extern "C" int cm_caviar_run(cm_ctx* ctx, cm_problem* p, const cm_priors* pr, const cm_options* o, cm_timings* tm) {
  const int rc = caviar_run_impl(ctx, p, pr, o, tm);
  if (rc != CM_OK && ctx) {
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    if (ctx->side) hipStreamSynchronize(ctx->side);
  }
  return rc;
}

static int caviar_run_impl(cm_ctx* ctx, cm_problem* p, const cm_priors* pr, const cm_options* o, cm_timings* tm) {
  if (!ctx || !p || !pr || !o) { cm_set_error("cm_caviar_run: NULL argument"); return CM_ERR_INVALID; }
  if (!pr->mu || !pr->beta || !pr->phi || !pr->phi_cov) { cm_set_error("cm_caviar_run: NULL prior array"); return CM_ERR_INVALID; }
  if (o->iters < 0 || o->num_mc_samples < 1) { cm_set_error("cm_caviar_run: iters >= 0 and num_mc_samples >= 1 required"); return CM_ERR_INVALID; }
  if (o->rng_mode == CM_RNG_EXPLICIT && (!o->update_orders || !o->uniforms)) {
    cm_set_error("cm_caviar_run: CM_RNG_EXPLICIT needs update_orders and uniforms"); return CM_ERR_INVALID;
  }
  CM_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int N = p->N, K = p->K, S = o->num_mc_samples, iters = o->iters;
  p->ran = false;
  CM_HIP(hipEventRecord(ctx->ev[0], st));

  // RNG streams: sweep orders of all iterations live in orders_all (uploaded, or generated below in one batched launch)
  if (p->orders_cap < iters) {
    if (p->orders_all) hipFree(p->orders_all);
    if (p->orders_tmp) hipFree(p->orders_tmp);
    if (p->inv_all) hipFree(p->inv_all);
    p->orders_all = p->orders_tmp = p->inv_all = nullptr;
    CM_TRY(dmalloc(&p->orders_all, (size_t)iters * N));
    CM_TRY(dmalloc(&p->orders_tmp, (size_t)iters * N));
    CM_TRY(dmalloc(&p->inv_all, (size_t)iters * N));
    p->orders_cap = iters;
  }
  if (p->wtab_cap < (size_t)iters * N * S * 2) {
    if (p->wtab) hipFree(p->wtab);
    p->wtab = nullptr;
    CM_TRY(dmalloc(&p->wtab, (size_t)iters * N * S * 2));
    p->wtab_cap = (size_t)iters * N * S * 2;
  }
  if (o->rng_mode == CM_RNG_EXPLICIT) {
    if (p->uexp) { hipFree(p->uexp); p->uexp = nullptr; }
    CM_TRY(dmalloc(&p->uexp, (size_t)iters * N * S * 2));
    CM_HIP(hipMemcpyAsync(p->uexp, o->uniforms, (size_t)iters * N * S * 2 * sizeof(double), hipMemcpyHostToDevice, st));
    CM_HIP(hipMemcpyAsync(p->orders_all, o->update_orders, (size_t)iters * N * sizeof(int32_t), hipMemcpyHostToDevice, st));
  }
  CM_TRY(upload_priors(ctx, p, pr));
  CM_TRY(cm_launch_init_state(ctx, p, o->y_xcorr_thresh));
  CM_HIP(hipMemsetAsync(p->n_act, 0, 4 * sizeof(int), st));
  CM_HIP(hipMemsetAsync(p->seq_meta, 0, 8 * sizeof(int), st));
  // Threefry key chain (host, sequential by nature): only iteration 0's keys are produced here; the rest is computed while
  // the GPU works on iteration 0 and turned into permutations / ndtri tables on the side stream (see the loop below).
  std::vector<uint32_t>& keys = p->h_keys;           // lives in the problem: asynchronous uploads read it after this function returns
  uint32_t chain[2] = {(uint32_t)(o->seed >> 32), (uint32_t)(o->seed & 0xFFFFFFFFull)};
  const size_t key_per = (size_t)(N + 3) * 2;
  bool rng_deferred = false;
  hipEvent_t pending_ev[2] = {nullptr, nullptr};   // long RNG chunks whose event the main stream has not waited on yet
  int pending_from[2] = {0, 0}, n_pending = 0;     // first iteration each of them serves
  int rng_done = 1;                              // iterations whose keys / orders / tables exist (iteration 0 is done up front)
  if (o->rng_mode == CM_RNG_THREEFRY && iters > 0) {
    keys.assign(key_per * iters, 0u);
    if (p->keys_iters < iters) {
      if (p->keys) hipFree(p->keys);
      p->keys = nullptr;
      CM_TRY(dmalloc(&p->keys, keys.size()));
      p->keys_iters = iters;
    }
    cm_host_key_schedule_range(chain, 0, 1, N, keys.data());
    CM_HIP(hipMemcpyAsync(p->keys, keys.data(), key_per * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    CM_TRY(cm_launch_rng(ctx, p, 0, 1, st));                       // :196 update order of iteration 0
    CM_TRY(cm_launch_ndtri_table(ctx, p, 0, 1, S, o->rng_mode, st));
    rng_deferred = iters > 1;
  } else {
    CM_TRY(cm_launch_ndtri_table(ctx, p, 0, iters, S, o->rng_mode, st));   // explicit streams: tabulate the uploaded uniforms
  }
  if (o->save_histories && iters > 0) CM_TRY(alloc_hist(p, iters)); else free_hist(p);

  static const bool trace_iters = getenv("CAVIAR_HIP_TRACE_ITERS") != nullptr;   // diagnostic: implies the per-kernel stamps
  static const bool trace_host = getenv("CAVIAR_HIP_TRACE_HOST") != nullptr;     // diagnostic: host clock at every mailbox arrival (no events)
  std::vector<std::chrono::steady_clock::time_point> host_stamps;
  const bool timed = tm != nullptr && (o->timing >= 1 || trace_iters);   // stamps around the chain kernel (events 3, 4) and k_iter_mid (8, 9)
  const bool phases = tm != nullptr && o->timing >= 2;  // all phase stamps (more events and host API calls per iteration)
  while (ctx->pool.size() < (size_t)(iters + 1) * CM_EV_PER_IT) { hipEvent_t e; CM_HIP(hipEventCreate(&e)); ctx->pool.push_back(e); }
  auto EV = [&](int it, int j) { return ctx->pool[(size_t)it * CM_EV_PER_IT + j]; };
  auto post_wait_fallback = [&]() -> int {       // CAVIAR_HIP_NO_MAILBOX=1: classic copy + event instead of the mapped mailbox
    if (!ctx->use_mailbox) {
      CM_HIP(hipMemcpyAsync(ctx->h_pinned + 8, p->n_act, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
      CM_HIP(hipEventRecord(ctx->ev[7], st));
    }
    return CM_OK;
  };
  CM_HIP(hipEventRecord(ctx->ev[1], st));
  CM_TRY(cm_launch_stats(ctx, p, o->spont_orthogonality));           // row sums of lam0
  if (iters > 0) {                                                   // active set + block_update_mu + MC logit of iteration 0
    CM_TRY(cm_launch_iter_mid(ctx, p, -1, iters, S, o->rng_mode, o, 0, 0));
    CM_TRY(post_wait_fallback());
  }
  int last_active = 0;
  long long sum_active = 0;
  for (int it = 0; it < iters; ++it) {                               // caviar.py:79
    p->order = p->orders_all + (size_t)it * N;
    if (phases) CM_HIP(hipEventRecord(EV(it, 0), st));
    CM_TRY(cm_wait_active(ctx, p));                                  // the one host wait of an iteration: size of the active set
    if (trace_host) host_stamps.push_back(std::chrono::steady_clock::now());
    last_active = ctx->mb_n;
    sum_active += last_active;
    if (ctx->mb_flag) { cm_set_error("iteration %d: weight precision matrix not positive definite", it); return CM_ERR_NOT_SPD; }
    if (last_active > p->small_cap) CM_TRY(cm_solve_launch(ctx, p, last_active));   // :80 block_update_mu, large active set
    if (phases) CM_HIP(hipEventRecord(EV(it, 1), st));
    CM_TRY(cm_launch_sweep(ctx, p, it, o, timed ? EV(it, 3) : nullptr));  // :81  update_lam
    if (timed) CM_HIP(hipEventRecord(EV(it, 4), st));
    CM_TRY(cm_launch_stats(ctx, p, o->spont_orthogonality));         // row + column statistics of the new lam
    if (rng_deferred) {
      // The work of this iteration is enqueued and the GPU is busy with it: extend the (sequential, host-side) threefry key
      // chain (~17 us per iteration at N = 1000) and turn it into sweep orders / ndtri tables on the side stream.  Three chunks,
      // each sized to the slack the host has while the GPU works through a full-size solve (the first three iterations keep
      // it busy for 0.6-0.9 ms each, of which the ~55 launches take 0.25 ms of host time): iterations 1-12, 13-28, the rest.
      // Only the first chunk is consumed by the very next kernel (the MC logit of iteration 1); the main stream waits for the
      // other two where their first table is read, a dozen iterations later.
      const int upto = (it == 0) ? std::min(iters, 13) : (it == 1) ? std::min(iters, 29) : iters;
      if (upto > rng_done) {
        const int n_new = upto - rng_done;
        cm_host_key_schedule_range(chain, rng_done, n_new, N, keys.data());
        CM_HIP(hipStreamWaitEvent(ctx->side, ctx->ev[1], 0));        // (ev[1]: first part of the keys is uploaded)
        CM_HIP(hipMemcpyAsync(p->keys + key_per * rng_done, keys.data() + key_per * rng_done, key_per * n_new * sizeof(uint32_t),
                              hipMemcpyHostToDevice, ctx->side));
        CM_TRY(cm_launch_rng(ctx, p, rng_done, n_new, ctx->side));   // orders of the new iterations (:196)
        CM_TRY(cm_launch_ndtri_table(ctx, p, rng_done, n_new, S, o->rng_mode, ctx->side));
        hipEvent_t chunk_ev = (it == 0) ? ctx->ev[6] : (it == 1 ? ctx->ev[4] : ctx->ev[5]);
        CM_HIP(hipEventRecord(chunk_ev, ctx->side));
        if (it == 0) CM_HIP(hipStreamWaitEvent(st, chunk_ev, 0));     // the MC logit of iteration 1 (next kernel) needs it
        else { pending_ev[n_pending] = chunk_ev; pending_from[n_pending] = rng_done; ++n_pending; }
        rng_done = upto;
      }
      rng_deferred = rng_done < iters;
    }
    for (int q = 0; q < n_pending; ++q)                               // k_iter_mid below draws the MC logit of iteration it + 1
      if (pending_ev[q] && it + 1 >= pending_from[q]) { CM_HIP(hipStreamWaitEvent(st, pending_ev[q], 0)); pending_ev[q] = nullptr; }
    if (o->save_histories) CM_TRY(record_hist(ctx, p, it, 0));       // mu, beta, lam of this iteration (:90-92)
    // :84-88 update_sigma, update_phi, z, spont_rate; then block_update_mu (small sets) and the MC logit of iteration it + 1
    if (timed) CM_HIP(hipEventRecord(EV(it, 8), st));
    CM_TRY(cm_launch_iter_mid(ctx, p, it, iters, S, o->rng_mode, o, 1, 1));
    if (timed) CM_HIP(hipEventRecord(EV(it, 9), st));
    CM_TRY(post_wait_fallback());
    if (o->save_histories) CM_TRY(record_hist(ctx, p, it, 1));       // shape, rate, phi, phi_cov, z
    if (phases) CM_HIP(hipEventRecord(EV(it, 5), st));
    if (o->verbose >= 3) {
      CM_HIP(hipStreamSynchronize(st));
      fprintf(stderr, "[caviar_hip] it %d active %d\n", it, last_active);
    }
  }
  CM_HIP(hipEventRecord(ctx->ev[2], st));

  if (o->fn_scan) {                                                  // caviar.py:94-98
    // mu | beta | z | lam are one device slab: one pinned copy down, the host scan, one copy back, final update_phi
    const size_t nslab = 2 * (size_t)N + K + (size_t)p->nnz;
    CM_TRY(stage_reserve(ctx, nslab * sizeof(double)));
    double* h = (double*)ctx->h_stage;
    CM_HIP(hipMemcpyAsync(h, p->slab, nslab * sizeof(double), hipMemcpyDeviceToHost, st));
    CM_HIP(hipStreamSynchronize(st));
    host_reconnect(p, h, h + N, h + 2 * (size_t)N + K, h + 2 * (size_t)N, o->msrmp, o->minimum_spike_count, o->verbose);
    CM_HIP(hipMemcpyAsync(p->slab, h, nslab * sizeof(double), hipMemcpyHostToDevice, st));
    CM_TRY(cm_launch_update_phi(ctx, p, st));
  }
  CM_HIP(hipEventRecord(ctx->ev[3], st));
  CM_HIP(hipStreamSynchronize(st));
  if (iters > 0) {
    CM_HIP(hipMemcpy(ctx->h_pinned + 8, p->n_act, 4 * sizeof(int), hipMemcpyDeviceToHost));   // [8..11]: outside the mailbox words
    if (ctx->h_pinned[9]) { cm_set_error("weight precision matrix not positive definite"); return CM_ERR_NOT_SPD; }
    if (ctx->h_pinned[10]) { cm_set_error("sweep ticket wait timed out (internal error)"); return CM_ERR_STATE; }
  }
  if (trace_host)
    for (size_t i = 1; i < host_stamps.size(); ++i)
      fprintf(stderr, "mailbox %2zu -> %2zu: %7.1f us\n", i - 1, i, std::chrono::duration<double, std::micro>(host_stamps[i] - host_stamps[i - 1]).count());
  if (tm) {
    float a = 0, b = 0, c = 0;
    hipEventElapsedTime(&a, ctx->ev[0], ctx->ev[3]);
    hipEventElapsedTime(&b, ctx->ev[1], ctx->ev[2]);
    hipEventElapsedTime(&c, ctx->ev[2], ctx->ev[3]);
    memset(tm, 0, sizeof(*tm));
    tm->total_ms = a; tm->iters_ms = b; tm->fn_scan_ms = c; tm->iters = iters; tm->last_active = last_active;
    tm->sum_active = sum_active;
    int meta[8] = {0};
    if (hipMemcpy(meta, p->seq_meta, sizeof(meta), hipMemcpyDeviceToHost) == hipSuccess) { tm->sum_seq_rows = meta[4]; tm->sum_seq_records = meta[5]; }
    for (int it = 0; timed && it < iters; ++it) {
      float d = 0;
      hipEventElapsedTime(&d, EV(it, 3), EV(it, 4));
      tm->sweep_seq_ms += d;
      hipEventElapsedTime(&d, EV(it, 8), EV(it, 9));
      tm->mid_ms += d;
      if (trace_iters && it > 0) {                                            // diagnostic: wall time of every iteration
        float e = 0;
        hipEventElapsedTime(&e, EV(it - 1, 9), EV(it, 9));
        fprintf(stderr, "it %2d %8.1f us (k_iter_mid incl. its launch gap %.1f)\n", it, e * 1e3, d * 1e3);
      }
      if (!phases) continue;
      hipEventElapsedTime(&d, EV(it, 0), EV(it, 1)); tm->solve_ms += d;        // host wait + large-set solve (small sets are fused)
      hipEventElapsedTime(&d, EV(it, 1), EV(it, 3)); tm->sweep_indep_ms += d;  // pre-sweep + pack
      hipEventElapsedTime(&d, EV(it, 4), EV(it, 5)); tm->post_ms += d;         // stats + fused mid-iteration kernel
    }
  }
  p->ran = true;
  return CM_OK;
}

extern "C" int cm_caviar_fetch(cm_ctx* ctx, cm_problem* p, cm_result* r) {
  if (!ctx || !p || !r) { cm_set_error("cm_caviar_fetch: NULL argument"); return CM_ERR_INVALID; }
  if (!p->ran) { cm_set_error("cm_caviar_fetch: no completed cm_caviar_run on this problem"); return CM_ERR_STATE; }
  CM_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t N = p->N, K = p->K, nnz = p->nnz;
  const hipMemcpyKind k = hipMemcpyDeviceToHost;
  if (r->mu) CM_HIP(hipMemcpyAsync(r->mu, p->mu, N * 8, k, st));
  if (r->beta) CM_HIP(hipMemcpyAsync(r->beta, p->beta, N * 8, k, st));
  if (r->phi) CM_HIP(hipMemcpyAsync(r->phi, p->phi, 2 * N * 8, k, st));
  if (r->phi_cov) CM_HIP(hipMemcpyAsync(r->phi_cov, p->phi_cov, 4 * N * 8, k, st));
  if (r->z) CM_HIP(hipMemcpyAsync(r->z, p->z, K * 8, k, st));
  double sc[CM_S_COUNT];
  CM_HIP(hipMemcpyAsync(sc, p->scal, sizeof(sc), k, st));
  std::vector<double> sp;
  if (r->lam) {   // compressed rows over PCIe, dense expansion on the host (lam is zero off the targeted trials)
    sp.resize(nnz ? nnz : 1);
    if (nnz) CM_HIP(hipMemcpyAsync(sp.data(), p->lam, nnz * 8, k, st));
  }
  CM_HIP(hipStreamSynchronize(st));
  r->shape = sc[CM_S_SHAPE];
  r->rate = sc[CM_S_RATE];
  auto expand = [&](const double* src, double* dst) {
    memset(dst, 0, N * K * sizeof(double));
    for (size_t n = 0; n < N; ++n)
      for (int e = p->h_row_ptr[n]; e < p->h_row_ptr[n + 1]; ++e) dst[n * K + p->h_col[e]] = src[e];
  };
  if (r->lam) expand(sp.data(), r->lam);
  if (p->hist_iters > 0) {
    const size_t H = p->hist_iters;
    if (r->mu_hist) CM_HIP(hipMemcpy(r->mu_hist, p->h_mu, H * N * 8, k));
    if (r->beta_hist) CM_HIP(hipMemcpy(r->beta_hist, p->h_beta, H * N * 8, k));
    if (r->phi_hist) CM_HIP(hipMemcpy(r->phi_hist, p->h_phi, H * 2 * N * 8, k));
    if (r->phi_cov_hist) CM_HIP(hipMemcpy(r->phi_cov_hist, p->h_phi_cov, H * 4 * N * 8, k));
    if (r->z_hist) CM_HIP(hipMemcpy(r->z_hist, p->h_z, H * K * 8, k));
    if (r->shape_hist || r->rate_hist) {                                  // scalars broadcast to [iters, K] (caviar.py:60-61)
      std::vector<double> a(H), b(H);
      CM_HIP(hipMemcpy(a.data(), p->h_shape, H * 8, k));
      CM_HIP(hipMemcpy(b.data(), p->h_rate, H * 8, k));
      for (size_t i = 0; i < H; ++i)
        for (size_t j = 0; j < K; ++j) {
          if (r->shape_hist) r->shape_hist[i * K + j] = a[i];
          if (r->rate_hist) r->rate_hist[i * K + j] = b[i];
        }
    }
    if (r->lam_hist) {
      std::vector<double> row(nnz ? nnz : 1);
      for (size_t i = 0; i < H; ++i) {
        if (nnz) CM_HIP(hipMemcpy(row.data(), p->h_lam + i * nnz, nnz * 8, k));
        expand(row.data(), r->lam_hist + i * N * K);
      }
    }
  }
  return CM_OK;
}

extern "C" int cm_caviar_fit(cm_ctx* ctx, int32_t N, int32_t K, int32_t T, const double* psc, const double* stim, const cm_priors* pr,
                             const cm_options* o, cm_result* r, cm_timings* tm) {
  cm_problem* p = nullptr;
  CM_TRY(cm_problem_create(ctx, N, K, T, psc, stim, CM_MEM_HOST, &p));
  int rc = cm_caviar_run(ctx, p, pr, o, tm);
  if (rc == CM_OK && r) rc = cm_caviar_fetch(ctx, p, r);
  cm_problem_destroy(p);
  return rc;
}

// ------------------------------------------------------------------------------------------------
// per-phase entry points for the parity tests
// ------------------------------------------------------------------------------------------------
extern "C" int cm_k_prologue(cm_ctx* ctx, int32_t K, int32_t T, const double* psc, double* y, double* ss) {
  if (!ctx || !psc || !y || !ss || K < 1 || T < 1) { cm_set_error("cm_k_prologue: bad argument"); return CM_ERR_INVALID; }
  CM_HIP(hipSetDevice(ctx->device));
  double *d_psc = nullptr, *d_y = nullptr, *d_ss = nullptr;
  CM_TRY(dmalloc(&d_psc, (size_t)K * T)); CM_TRY(dmalloc(&d_y, K)); CM_TRY(dmalloc(&d_ss, K));
  CM_HIP(hipMemcpyAsync(d_psc, psc, (size_t)K * T * 8, hipMemcpyHostToDevice, ctx->stream));
  CM_TRY(cm_launch_psc_prologue(ctx, K, T, d_psc, d_y, d_ss));
  CM_HIP(hipMemcpyAsync(y, d_y, (size_t)K * 8, hipMemcpyDeviceToHost, ctx->stream));
  CM_HIP(hipMemcpyAsync(ss, d_ss, (size_t)K * 8, hipMemcpyDeviceToHost, ctx->stream));
  CM_HIP(hipStreamSynchronize(ctx->stream));
  hipFree(d_psc); hipFree(d_y); hipFree(d_ss);
  return CM_OK;
}

// builds a problem around a stimulus matrix and a dense lam (psc is not needed by the phase under test)
static int problem_from_dense(cm_ctx* ctx, int N, int K, const double* stim, const double* lam, const double* y, cm_problem** out) {
  std::vector<double> psc((size_t)K, 0.0);   // T = 1 placeholder
  CM_TRY(cm_problem_create(ctx, N, K, 1, psc.data(), stim, CM_MEM_HOST, out));
  cm_problem* p = *out;
  double* d_dense = nullptr;
  CM_TRY(dmalloc(&d_dense, (size_t)N * K));
  CM_HIP(hipMemcpyAsync(d_dense, lam, (size_t)N * K * 8, hipMemcpyHostToDevice, ctx->stream));
  CM_TRY(cm_launch_scatter_dense_lam(ctx, p, d_dense));
  if (y) CM_HIP(hipMemcpyAsync(p->y, y, (size_t)K * 8, hipMemcpyHostToDevice, ctx->stream));
  CM_HIP(hipStreamSynchronize(ctx->stream));
  hipFree(d_dense);
  return CM_OK;
}

extern "C" int cm_k_block_update_mu(cm_ctx* ctx, int32_t N, int32_t K, const double* stim, const double* lam, const double* y,
                                    double shape, double rate, const double* mu_prior, const double* beta_prior, double* mu, double* beta) {
  if (!ctx || !stim || !lam || !y || !mu_prior || !beta_prior || !mu || !beta) { cm_set_error("cm_k_block_update_mu: NULL argument"); return CM_ERR_INVALID; }
  cm_problem* p = nullptr;
  CM_TRY(problem_from_dense(ctx, N, K, stim, lam, y, &p));
  hipStream_t st = ctx->stream;
  double h[CM_S_COUNT] = {0};
  h[CM_S_SHAPE] = shape; h[CM_S_RATE] = rate;
  int rc = CM_OK;
  if (hipMemcpy(p->scal, h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(p->mu_prior, mu_prior, (size_t)N * 8, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(p->beta_prior, beta_prior, (size_t)N * 8, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemset(p->n_act, 0, 4 * sizeof(int)) != hipSuccess) { cm_set_error("upload failed"); rc = CM_ERR_HIP; }
  int na = 0;
  if (!rc) rc = cm_launch_stats(ctx, p, 0.1);
  if (!rc) rc = cm_launch_post_single(ctx, p, -1, nullptr, 0);
  if (!rc) rc = cm_solve_mu(ctx, p, &na);
  if (!rc && (hipMemcpyAsync(mu, p->mu, (size_t)N * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
              hipMemcpyAsync(beta, p->beta, (size_t)N * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
              hipMemcpyAsync(ctx->h_pinned + 8, p->n_act, 16, hipMemcpyDeviceToHost, st) != hipSuccess ||
              hipStreamSynchronize(st) != hipSuccess)) { cm_set_error("download failed: %s", hipGetErrorString(hipGetLastError())); rc = CM_ERR_HIP; }
  if (!rc && ctx->h_pinned[9]) { cm_set_error("precision matrix not positive definite"); rc = CM_ERR_NOT_SPD; }
  cm_problem_destroy(p);
  return rc;
}

extern "C" int cm_k_update_phi(cm_ctx* ctx, int32_t N, int32_t K, const double* stim, const double* lam, const double* phi_prior,
                               const double* phi_cov_prior, double* phi, double* phi_cov) {
  if (!ctx || !stim || !lam || !phi_prior || !phi_cov_prior || !phi || !phi_cov) { cm_set_error("cm_k_update_phi: NULL argument"); return CM_ERR_INVALID; }
  cm_problem* p = nullptr;
  CM_TRY(problem_from_dense(ctx, N, K, stim, lam, nullptr, &p));
  hipStream_t st = ctx->stream;
  int rc = CM_OK;
  if (hipMemcpy(p->phi_prior, phi_prior, (size_t)N * 16, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(p->phi_cov_prior, phi_cov_prior, (size_t)N * 32, hipMemcpyHostToDevice) != hipSuccess) { cm_set_error("upload failed"); rc = CM_ERR_HIP; }
  if (!rc) rc = cm_launch_update_phi(ctx, p, st);
  if (!rc && (hipMemcpyAsync(phi, p->phi, (size_t)N * 16, hipMemcpyDeviceToHost, st) != hipSuccess ||
              hipMemcpyAsync(phi_cov, p->phi_cov, (size_t)N * 32, hipMemcpyDeviceToHost, st) != hipSuccess ||
              hipStreamSynchronize(st) != hipSuccess)) { cm_set_error("download failed: %s", hipGetErrorString(hipGetLastError())); rc = CM_ERR_HIP; }
  cm_problem_destroy(p);
  return rc;
}

extern "C" int cm_k_reconnect(cm_ctx* ctx, int32_t N, int32_t K, const double* stim, double* lam, double* mu, double* beta, double* z,
                              double minimax_spk_prob, double minimum_spike_count) {
  if (!ctx || !stim || !lam || !mu || !beta || !z) { cm_set_error("cm_k_reconnect: NULL argument"); return CM_ERR_INVALID; }
  cm_problem* p = nullptr;
  CM_TRY(problem_from_dense(ctx, N, K, stim, lam, nullptr, &p));     // builds the compressed rows (device) + host mirror
  const size_t nnz = p->nnz;
  std::vector<double> vm(mu, mu + N), vb(beta, beta + N), vz(z, z + K), vl(nnz ? nnz : 1);
  for (int n = 0; n < N; ++n)
    for (int e = p->h_row_ptr[n]; e < p->h_row_ptr[n + 1]; ++e) vl[e] = lam[(size_t)n * K + p->h_col[e]];
  host_reconnect(p, vm.data(), vb.data(), vl.data(), vz.data(), minimax_spk_prob, minimum_spike_count, 0);
  for (int n = 0; n < N; ++n)
    for (int e = p->h_row_ptr[n]; e < p->h_row_ptr[n + 1]; ++e) lam[(size_t)n * K + p->h_col[e]] = vl[e];
  std::copy(vm.begin(), vm.end(), mu); std::copy(vb.begin(), vb.end(), beta); std::copy(vz.begin(), vz.end(), z);
  cm_problem_destroy(p);
  return CM_OK;
}

__global__ void k_nop(int* sink) { if (sink && threadIdx.x == 0xFFFFFF) *sink = 0; }

extern "C" int cm_k_event_pair_overhead(cm_ctx* ctx, int32_t reps, int32_t blocks, int32_t threads, float* ms_per_pair) {
  if (!ctx || !ms_per_pair || reps < 1 || blocks < 1 || threads < 1 || threads > 1024) { cm_set_error("cm_k_event_pair_overhead: bad argument"); return CM_ERR_INVALID; }
  CM_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  double sum = 0.0;
  for (int r = 0; r < reps; ++r) {
    hipLaunchKernelGGL(k_nop, dim3(blocks), dim3(threads), 0, st, (int*)nullptr);     // predecessor: the pair starts behind a running kernel
    CM_HIP(hipEventRecord(ctx->ev[0], st));
    hipLaunchKernelGGL(k_nop, dim3(blocks), dim3(threads), 0, st, (int*)nullptr);
    CM_HIP(hipEventRecord(ctx->ev[1], st));
    CM_HIP(hipStreamSynchronize(st));
    float d = 0.0f;
    CM_HIP(hipEventElapsedTime(&d, ctx->ev[0], ctx->ev[1]));
    sum += d;
  }
  *ms_per_pair = (float)(sum / reps);
  return CM_OK;
}

extern "C" int cm_k_rng_streams(cm_ctx* ctx, uint64_t seed, int32_t it, int32_t N, int32_t S, int32_t* order, double* u) {
  if (!ctx || !order || !u || N < 2 || S < 1 || it < 0) { cm_set_error("cm_k_rng_streams: bad argument"); return CM_ERR_INVALID; }
  CM_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  cm_problem q;                         // only the RNG buffers are used
  q.ctx = ctx; q.N = N;
  std::vector<uint32_t> keys;
  cm_host_key_schedule(seed, it + 1, N, keys);
  double* d_u = nullptr;
  int rc = CM_OK;
  if ((rc = dmalloc(&q.keys, keys.size())) || (rc = dmalloc(&q.orders_all, (size_t)(it + 1) * N)) ||
      (rc = dmalloc(&q.orders_tmp, (size_t)(it + 1) * N)) || (rc = dmalloc(&q.phi, 2 * (size_t)N)) || (rc = dmalloc(&q.phi_cov, 4 * (size_t)N)) ||
      (rc = dmalloc(&q.mcE, 1)) || (rc = dmalloc(&q.vals, CM_MAX_VALS)) || (rc = dmalloc(&d_u, (size_t)N * S * 2))) {
  } else if (hipMemcpyAsync(q.keys, keys.data(), keys.size() * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
             hipMemsetAsync(q.phi, 0, (size_t)N * 16, st) != hipSuccess || hipMemsetAsync(q.phi_cov, 0, (size_t)N * 32, st) != hipSuccess) {
    cm_set_error("rng upload failed"); rc = CM_ERR_HIP;
  }
  if (!rc) rc = cm_launch_rng(ctx, &q, 0, it + 1, st);
  q.order = q.orders_all + (size_t)it * N;
  if (!rc) rc = cm_k_rng_streams_impl(ctx, &q, it, S, d_u);
  if (!rc && (hipMemcpyAsync(order, q.order, (size_t)N * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
              hipMemcpyAsync(u, d_u, (size_t)N * S * 16, hipMemcpyDeviceToHost, st) != hipSuccess ||
              hipStreamSynchronize(st) != hipSuccess)) { cm_set_error("rng download failed: %s", hipGetErrorString(hipGetLastError())); rc = CM_ERR_HIP; }
  void* ptrs[] = {q.keys, q.orders_all, q.orders_tmp, q.phi, q.phi_cov, q.mcE, q.vals, d_u};
  for (void* x : ptrs) if (x) hipFree(x);
  return rc;
}
