// C ABI of libcaviar_hip.so (include/caviar_hip.h): context / problem lifecycle, the caviar() driver loop
// (reference circuitmap/optimise/caviar.py:20-100) and the host-side false-negative scan (caviar.py:102-144).
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include <queue>
#include <thread>
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
int cm_lds_table_raise(int device, const void* fn, size_t bytes, size_t floor_bytes, cm_lds_setter set, void* user, size_t* applied) {
  std::lock_guard<std::mutex> lk(g_lds_mu);
  size_t& slot = g_lds_table[std::make_pair(device, fn)];
  const size_t cur = slot > floor_bytes ? slot : floor_bytes;
  if (applied) *applied = 0;
  if (bytes <= cur) return CM_OK;
  const int rc = set(fn, bytes, user);                   // the runtime call happens under the lock ...
  if (rc != CM_OK) return rc;
  slot = bytes;                                          // ... and the table only records what the runtime accepted
  if (applied) *applied = bytes;
  return CM_OK;
}
int cm_lds_set_attribute(const void* fn, size_t bytes, void*) {
  CM_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return CM_OK;
}
// test hooks (no GPU needed): a recording setter stands in for the runtime.  Keys of these hooks live in their own range of the table:
// odd addresses, which no kernel has.
struct lds_fake { std::atomic<size_t> in_force{0}; std::atomic<int> calls{0}; int fail_above = 0; int delay_us = 0; };
static int lds_fake_set(const void*, size_t bytes, void* user) {
  lds_fake* f = (lds_fake*)user;
  if (f->fail_above && bytes > (size_t)f->fail_above) return CM_ERR_HIP;
  if (f->delay_us) std::this_thread::sleep_for(std::chrono::microseconds(f->delay_us));   // widens the window a race would need
  f->in_force.store(bytes);                              // like the runtime: the new value REPLACES the old one
  f->calls.fetch_add(1);
  return CM_OK;
}
extern "C" int64_t cm_test_lds_request(int32_t device, uint64_t kernel_key, uint64_t bytes, uint64_t floor_bytes) {
  lds_fake f;
  size_t applied = 0;
  if (cm_lds_table_raise(device, reinterpret_cast<const void*>((uintptr_t)(2 * kernel_key + 1)), (size_t)bytes, (size_t)floor_bytes, lds_fake_set, &f, &applied) != CM_OK) return -1;
  return (int64_t)applied;
}
// n_threads host threads issue `reps` requests each for ONE (device, kernel) with interleaved sizes (thread t asks for base + (r * n_threads + t)
// * step); out[0] = value the fake runtime holds at the end, out[1] = largest request, out[2] = setter calls.  A correct table
// leaves out[0] == out[1].  fail_above > 0: requests above it are refused by the setter (the table must not record them): then out[0] is
// the largest accepted request.
extern "C" int cm_test_lds_race(uint64_t kernel_key, int32_t n_threads, int32_t reps, uint64_t base, uint64_t step, uint64_t fail_above, uint64_t* out) {
  if (n_threads < 1 || reps < 1 || !out) return CM_ERR_INVALID;
  lds_fake f;
  f.fail_above = (int)fail_above; f.delay_us = 20;
  const void* fn = reinterpret_cast<const void*>((uintptr_t)(2 * kernel_key + 1));
  std::vector<std::thread> th;
  std::atomic<int> go{0};
  for (int t = 0; t < n_threads; ++t)
    th.emplace_back([&, t]() {
      while (!go.load()) std::this_thread::yield();
      for (int r = 0; r < reps; ++r) cm_lds_table_raise(7, fn, (size_t)(base + (uint64_t)(r * n_threads + t) * step), 0, lds_fake_set, &f, nullptr);
    });
  go.store(1);
  for (auto& x : th) x.join();
  out[0] = f.in_force.load(); out[1] = base + (uint64_t)(reps * n_threads - 1) * step; out[2] = (uint64_t)f.calls.load();
  // what the table believes is in force: a further request for exactly that much must be a no-op, one byte more must call the setter
  size_t applied = 0;
  cm_lds_table_raise(7, fn, (size_t)out[0], 0, lds_fake_set, &f, &applied);
  out[3] = applied;
  return CM_OK;
}
extern "C" const char* cm_version(void) { return "caviar_hip 0.3 (gfx950, ABI 4)"; }
extern "C" int cm_abi_version(void) { return CM_ABI_VERSION; }

extern "C" int cm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" void cm_options_default(cm_options* o) {   // caviar.py:21-23
  memset(o, 0, sizeof(*o));
  o->abi_version = CM_ABI_VERSION;
  o->iters = 50; o->num_mc_samples = 100; o->seed = 0; o->y_xcorr_thresh = 1e-2; o->minimum_spike_count = 3;
  o->delay_spont_est = 1; o->msrmp = 0.3; o->scale_factor = 0.75; o->penalty = 5e0; o->save_histories = 0;
  o->max_backtrack_iters = 20; o->tol = 0.05; o->spont_orthogonality = 0.1; o->fn_scan = 1;
  o->rng_mode = CM_RNG_THREEFRY; o->update_orders = nullptr; o->uniforms = nullptr; o->verbose = 0; o->timing = 0; o->interleave = 0;
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
  if (const char* e = getenv("CAVIAR_HIP_SOLVE_GROUP")) c->solve_group = atoi(e);
  if (const char* e = getenv("CAVIAR_HIP_XCD_MODE")) c->xcd_mode = atoi(e);
  if (const char* e = getenv("CAVIAR_HIP_CHOL_FAMILY")) c->chol_family = atoi(e);
  if (getenv("CAVIAR_HIP_TRACE_BLOCK0")) { CM_HIP(hipMalloc((void**)&c->dbg_block0, 64 * 8 * sizeof(long long))); CM_HIP(hipMemset(c->dbg_block0, 0, 64 * 8 * sizeof(long long))); }
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
  c->root = c;
  if (const char* e = getenv("CAVIAR_HIP_POOL_GB")) c->mem.cap_bytes = (size_t)(atof(e) * (double)((size_t)1 << 30));
  const int rc = ctx_init(c, device_id, hip_stream);
  if (rc != CM_OK) { cm_ctx_destroy(c); return rc; }      // a half-built context is released, not leaked
  *out = c;
  return CM_OK;
}

extern "C" int cm_ctx_destroy(cm_ctx* c) {
  if (!c) return CM_OK;
  if (c->helper) { delete c->helper; c->helper = nullptr; }          // (idle: cm_caviar_run_batch waits for its task before it returns)
  if (c->twin) { cm_ctx_destroy(c->twin); c->twin = nullptr; }
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  if (c->side) { hipStreamSynchronize(c->side); hipStreamDestroy(c->side); }
  for (auto& e : c->ev) if (e) hipEventDestroy(e);
  for (auto& e : c->pool) hipEventDestroy(e);
  if (c->h_pinned) hipHostFree(c->h_pinned);
  if (c->h_stage) hipHostFree(c->h_stage);
  if (c->h_pdev) hipHostFree(c->h_pdev);
  if (c->d_pdev) hipFree(c->d_pdev);
  if (c->root == c) cm_pool_trim(c, 0);                               // the idle slabs of the pool (a twin allocates from its parent's)
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

// ---- device-memory pool (cm_internal.h)
static size_t pool_class(size_t b) {                // next size of the form m * 2^e, m in 8..15 (classes 12.5 % apart), at least 64 KB
  if (b <= ((size_t)64 << 10)) return (size_t)64 << 10;
  int e = 0;
  while ((b >> e) > 15) ++e;
  size_t c = ((b + (((size_t)1 << e) - 1)) >> e) << e;
  return c;
}
int cm_pool_alloc(cm_ctx* ctx, void** out, size_t bytes) {
  cm_pool& pl = ctx->root->mem;
  const size_t c = pool_class(bytes ? bytes : 1);
  {
    std::lock_guard<std::mutex> lk(pl.m);
    auto it = pl.idle.find(c);
    if (it != pl.idle.end()) {
      *out = it->second;
      pl.idle.erase(it);
      pl.idle_bytes -= c;
      pl.live[*out] = c;
      return CM_OK;
    }
  }
  hipError_t e = hipMalloc(out, c);
  if (e != hipSuccess) {                              // out of memory with idle slabs around: give them back and try once more
    (void)hipGetLastError();
    cm_pool_trim(ctx, 0);
    e = hipMalloc(out, c);
  }
  if (e != hipSuccess) { *out = nullptr; cm_set_error("out of device memory (%zu bytes): %s", c, hipGetErrorString(e)); return CM_ERR_HIP; }
  std::lock_guard<std::mutex> lk(pl.m);
  pl.live[*out] = c;
  return CM_OK;
}
void cm_pool_free(cm_ctx* ctx, void* ptr) {
  if (!ptr) return;
  cm_pool& pl = ctx->root->mem;
  size_t c = 0;
  {
    std::lock_guard<std::mutex> lk(pl.m);
    auto it = pl.live.find(ptr);
    if (it == pl.live.end()) { hipFree(ptr); return; }      // not from the pool
    c = it->second;
    pl.live.erase(it);
    if (pl.idle_bytes + c <= pl.cap_bytes) { pl.idle.emplace(c, ptr); pl.idle_bytes += c; return; }
  }
  hipFree(ptr);
}
void cm_pool_trim(cm_ctx* ctx, size_t keep_bytes) {
  cm_pool& pl = ctx->root->mem;
  std::vector<void*> drop;
  {
    std::lock_guard<std::mutex> lk(pl.m);
    while (pl.idle_bytes > keep_bytes && !pl.idle.empty()) {
      auto it = std::prev(pl.idle.end());            // largest first
      drop.push_back(it->second);
      pl.idle_bytes -= it->first;
      pl.idle.erase(it);
    }
  }
  for (void* q : drop) hipFree(q);
}
template <typename T>
static int pmalloc(cm_ctx* ctx, T** p, size_t n) { return cm_pool_alloc(ctx, (void**)p, (n ? n : 1) * sizeof(T)); }
extern "C" int cm_ctx_release_cache(cm_ctx* c) {
  if (!c) { cm_set_error("cm_ctx_release_cache: NULL context"); return CM_ERR_INVALID; }
  CM_HIP(hipSetDevice(c->device));
  cm_pool_trim(c, 0);
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
  // the two slabs hold every static / state buffer; the random tables of a run (sized by iters, S) are allocated by the run
  void* ptrs[] = {p->slabA, p->slabB, p->orders_all, p->orders_tmp, p->inv_all, p->wtab, p->wstat, p->keys, p->uexp};
  for (void* q : ptrs) cm_pool_free(p->ctx, q);                      // (the stream is idle: nothing queued still reads them)
  free_hist(p);
  delete p;
  return CM_OK;
}

// carves aligned pieces out of one allocation: first pass (base == nullptr) only sizes it
struct cm_carver {
  char* base = nullptr; size_t off = 0;
  template <typename T> void take(T** out, size_t n) { if (base) *out = reinterpret_cast<T*>(base + off); off = (off + (n ? n : 1) * sizeof(T) + 255) & ~(size_t)255; }
};
static void carve_a(cm_problem* p, cm_carver& c) {           // sizes known from N, K
  const size_t N = p->N, K = p->K, ld = p->ld;
  c.take(&p->y, K); c.take(&p->ss, K); c.take(&p->sum_y, 2); c.take(&p->vals, CM_MAX_VALS); c.take(&p->rowcnt, N * CM_MAX_VALS); c.take(&p->mcnt, N * CM_MAX_VALS); c.take(&p->qcnt, N * 8); c.take(&p->touch, N);
  c.take(&p->row_ptr, N + 1); c.take(&p->col_ptr, K + 1); c.take(&p->mask, K); c.take(&p->pslab, 8 * N + CM_S_COUNT); c.take(&p->phi, 2 * N);
  c.take(&p->phi_cov, 4 * N); c.take(&p->phi_dead, 8 * N); c.take(&p->phi_rep, N); c.take(&p->R, K); c.take(&p->colany, K); c.take(&p->mcE, N * CM_MAX_VALS);
  c.take(&p->rs, 4 * N); c.take(&p->act_idx, N); c.take(&p->act_inv, N); c.take(&p->n_act, 4); c.take(&p->alive, N + 1);
  c.take(&p->tstamp, 2 * K); c.take(&p->tlist_s, 2 * K); c.take(&p->tlist_r, 2 * K); c.take(&p->tcnt, 8);
  c.take(&p->A, ld * (ld + CM_LD_PAD)); c.take(&p->X, ld * (ld + CM_LD_PAD)); c.take(&p->Lf, ld * (ld + CM_LD_PAD)); c.take(&p->Dinv, ld * CM_TILE);
  c.take(&p->rhs, 2 * ld); c.take(&p->seq_n, N); c.take(&p->seq_off, N + 1); c.take(&p->seq_pos, N); c.take(&p->seq_meta, 8);
  c.take(&p->enc_table, CM_MAX_VALS); c.take(&p->enc_flags, 8); c.take(&p->enc_row_nnz, N); c.take(&p->enc_col_cnt, K); c.take(&p->enc_cursor, K);
}
static void carve_b(cm_problem* p, cm_carver& c) {           // sizes that need nnz
  const size_t N = p->N, K = p->K, nnz = p->nnz > 0 ? (size_t)p->nnz : 1;
  c.take(&p->col, nnz); c.take(&p->grp, nnz); c.take(&p->csc_row, nnz); c.take(&p->csc_pos, nnz); c.take(&p->csc_grp, nnz);
  c.take(&p->ye, nnz + 1); c.take(&p->me, nnz + 1); c.take(&p->turn8, nnz + 1); c.take(&p->slab, 2 * N + K + nnz);
  c.take(&p->chunk_first, (N + nnz) / 256 + 8);
  char* pk = nullptr; c.take(&pk, (N + nnz) * 32); if (c.base) p->packed = pk;
}

static int stage_reserve(cm_ctx* ctx, size_t bytes);
// A group of circuits at once: every kernel of the encoding is launched once for the group (cm_encode_stage1 / 2), the host waits twice per
// group (sizes of the nnz-dependent slabs; completion) instead of twice per circuit, and the memory comes from the context's pool.  Host
// inputs are staged through pooled device buffers, at most CM_CREATE_CHUNK circuits at a time (30 MB per circuit at the headline shape).
#define CM_CREATE_CHUNK 64
// compact inputs (cm_inputs): expanded on the device into the dense float64 arrays the encoding reads -- the circuit is, bit for bit, the one
// cm_problem_create would build from those arrays
__global__ __launch_bounds__(256) void k_in_f32_to_f64(const float* __restrict__ src, double* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (double)src[i];
}
__global__ __launch_bounds__(256) void k_in_triples(const cm_stim_triple* __restrict__ t, long long n, int N, int K, double* __restrict__ dense, int* __restrict__ bad) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const cm_stim_triple e = t[i];
  if (e.trial < 0 || e.trial >= K || e.neuron < 0 || e.neuron >= N) { *bad = 1; return; }
  dense[(size_t)e.neuron * K + e.trial] = (double)e.power;
}

static int create_chunk(cm_ctx* ctx, int n, const int32_t* N, const int32_t* K, const int32_t* T, const cm_inputs* in,
                        int mem_kind, cm_problem** out) {
  hipStream_t st = ctx->stream;
  std::vector<cm_problem*> ps(n, nullptr);
  std::vector<void*> tmp;                                             // staged host inputs + descriptor / report buffers (pool)
  int rc = CM_OK;
  auto cleanup = [&](int code) {
    hipStreamSynchronize(st);
    for (void* q : tmp) cm_pool_free(ctx, q);
    if (code != CM_OK) for (cm_problem* p : ps) if (p) cm_problem_destroy(p);
    return code;
  };
  std::vector<cm_enc> enc(n);
  int N_max = 0, K_max = 0;
  int* d_bad = nullptr;                                                // set by k_in_triples: an entry outside the matrix
  for (int c = 0; c < n; ++c) {
    cm_problem* p = ps[c] = new cm_problem();
    p->ctx = ctx; p->N = N[c]; p->K = K[c]; p->T = T[c];
    p->ld = ((N[c] + CM_TILE - 1) / CM_TILE) * CM_TILE;
    cm_carver sz;
    carve_a(p, sz);
    if (cm_pool_alloc(ctx, &p->slabA, sz.off) != CM_OK) return cleanup(CM_ERR_HIP);
    cm_carver cv; cv.base = (char*)p->slabA;
    carve_a(p, cv);
    // the dense float64 arrays the encoding reads: the caller's own (device, float64 dense), or pooled staging filled by an upload and / or
    // an expansion kernel (float32 -> float64; (trial, neuron, power) triples scattered into a zeroed matrix)
    const cm_inputs& ic = in[c];
    const size_t n_psc = (size_t)K[c] * T[c], n_stim = (size_t)N[c] * K[c];
    auto pooled = [&](size_t bytes) -> void* { void* q = nullptr; if (cm_pool_alloc(ctx, &q, bytes ? bytes : 8) != CM_OK) return nullptr; tmp.push_back(q); return q; };
    auto raw_on_device = [&](const void* src, size_t bytes) -> const void* {       // the caller's bytes where a kernel can read them
      if (mem_kind != CM_MEM_HOST) return src;
      void* q = pooled(bytes);
      if (!q) return nullptr;
      if (hipMemcpyAsync(q, src, bytes, hipMemcpyHostToDevice, st) != hipSuccess) { cm_set_error("upload of psc/stim failed"); return nullptr; }
      return q;
    };
    const double *d_psc = nullptr, *d_stim = nullptr;
    if (ic.psc_dtype == CM_DTYPE_F64) {
      d_psc = (const double*)raw_on_device(ic.psc, n_psc * sizeof(double));
      if (!d_psc) return cleanup(CM_ERR_HIP);
    } else {
      const float* r = (const float*)raw_on_device(ic.psc, n_psc * sizeof(float));
      double* a = (double*)pooled(n_psc * sizeof(double));
      if (!r || !a) return cleanup(CM_ERR_HIP);
      hipLaunchKernelGGL(k_in_f32_to_f64, dim3((unsigned)std::min<size_t>((n_psc + 255) / 256, 4096)), dim3(256), 0, st, r, a, n_psc);
      d_psc = a;
    }
    if (ic.stim_layout == CM_STIM_DENSE_F64) {
      d_stim = (const double*)raw_on_device(ic.stim, n_stim * sizeof(double));
      if (!d_stim) return cleanup(CM_ERR_HIP);
    } else if (ic.stim_layout == CM_STIM_DENSE_F32) {
      const float* r = (const float*)raw_on_device(ic.stim, n_stim * sizeof(float));
      double* b = (double*)pooled(n_stim * sizeof(double));
      if (!r || !b) return cleanup(CM_ERR_HIP);
      hipLaunchKernelGGL(k_in_f32_to_f64, dim3((unsigned)std::min<size_t>((n_stim + 255) / 256, 4096)), dim3(256), 0, st, r, b, n_stim);
      d_stim = b;
    } else {
      const cm_stim_triple* r = ic.n_triples > 0 ? (const cm_stim_triple*)raw_on_device(ic.stim, (size_t)ic.n_triples * sizeof(cm_stim_triple)) : nullptr;
      double* b = (double*)pooled(n_stim * sizeof(double));
      if ((ic.n_triples > 0 && !r) || !b) return cleanup(CM_ERR_HIP);
      if (!d_bad) { d_bad = (int*)pooled(sizeof(int)); if (!d_bad || hipMemsetAsync(d_bad, 0, sizeof(int), st) != hipSuccess) return cleanup(CM_ERR_HIP); }
      if (hipMemsetAsync(b, 0, n_stim * sizeof(double), st) != hipSuccess) return cleanup(CM_ERR_HIP);
      if (ic.n_triples > 0)
        hipLaunchKernelGGL(k_in_triples, dim3((unsigned)((ic.n_triples + 255) / 256)), dim3(256), 0, st, r, (long long)ic.n_triples, (int)N[c], (int)K[c], b, d_bad);
      d_stim = b;
    }
    cm_enc& e = enc[c];
    memset(&e, 0, sizeof(e));
    e.psc = d_psc; e.stim = d_stim; e.N = N[c]; e.K = K[c]; e.T = T[c];
    e.y = p->y; e.ss = p->ss; e.vals = p->vals; e.mcE = p->mcE; e.table = p->enc_table; e.flags = p->enc_flags; e.rowcnt = p->rowcnt;
    e.row_nnz = p->enc_row_nnz; e.row_ptr = p->row_ptr; e.col_cnt = p->enc_col_cnt; e.cursor = p->enc_cursor; e.col_ptr = p->col_ptr; e.alive = p->alive;
    N_max = std::max(N_max, (int)N[c]); K_max = std::max(K_max, (int)K[c]);
  }
  // descriptors and the report live in pooled device memory; their host side in the context's pinned staging buffer
  const size_t enc_bytes = (size_t)n * sizeof(cm_enc), rep_bytes = (size_t)n * 8 * sizeof(int);
  void *d_enc = nullptr, *d_rep = nullptr;
  if (cm_pool_alloc(ctx, &d_enc, enc_bytes) != CM_OK) return cleanup(CM_ERR_HIP);
  tmp.push_back(d_enc);
  if (cm_pool_alloc(ctx, &d_rep, rep_bytes) != CM_OK) return cleanup(CM_ERR_HIP);
  tmp.push_back(d_rep);
  if ((rc = stage_reserve(ctx, enc_bytes + rep_bytes)) != CM_OK) return cleanup(rc);
  cm_enc* h_enc = (cm_enc*)ctx->h_stage;
  int* h_rep = (int*)((char*)ctx->h_stage + enc_bytes);
  memcpy(h_enc, enc.data(), enc_bytes);
  if (hipMemcpyAsync(d_enc, h_enc, enc_bytes, hipMemcpyHostToDevice, st) != hipSuccess) return cleanup(CM_ERR_HIP);
  if ((rc = cm_encode_stage1(ctx, (const cm_enc*)d_enc, n, N_max, K_max, (int*)d_rep)) != CM_OK) return cleanup(rc);
  int h_bad = 0;
  if (d_bad && hipMemcpyAsync(&h_bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess) return cleanup(CM_ERR_HIP);
  if (hipMemcpyAsync(h_rep, d_rep, rep_bytes, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
    cm_set_error("encoding (stage 1) failed: %s", hipGetErrorString(hipGetLastError()));
    return cleanup(CM_ERR_HIP);
  }
  if (h_bad) { cm_set_error("stimulus triples: a (trial, neuron) index lies outside the N x K matrix"); return cleanup(CM_ERR_INVALID); }
  long long nnz_max = 0;
  for (int c = 0; c < n; ++c) {                                       // (too many values?, V, longest row, -, nnz)
    cm_problem* p = ps[c];
    const int* h = h_rep + 8 * c;
    if (h[0]) { cm_set_error("stimulus matrix has more than %d distinct values", CM_MAX_VALS); return cleanup(CM_ERR_TOO_MANY_POWERS); }
    p->V = h[1]; p->max_row_len = h[2]; p->nnz = h[4];
    p->small_cap = cm_small_cap(p);
    nnz_max = std::max<long long>(nnz_max, p->nnz);
    cm_carver sz;
    carve_b(p, sz);
    if (cm_pool_alloc(ctx, &p->slabB, sz.off) != CM_OK) return cleanup(CM_ERR_HIP);
    cm_carver cv; cv.base = (char*)p->slabB;
    carve_b(p, cv);
    cm_enc& e = h_enc[c];                                             // (the copy of stage 1 has completed: the staging buffer is free again)
    e.nnz = p->nnz; e.col = p->col; e.grp = p->grp; e.csc_row = p->csc_row; e.csc_pos = p->csc_pos; e.csc_grp = p->csc_grp;
    const size_t Nz = p->N;
    p->mu = p->slab; p->beta = p->slab + Nz; p->z = p->slab + 2 * Nz; p->lam = p->slab + 2 * Nz + p->K;
    p->mu_prior = p->pslab; p->beta_prior = p->pslab + Nz; p->phi_prior = p->pslab + 2 * Nz;
    p->phi_cov_prior = p->pslab + 4 * Nz; p->scal = p->pslab + 8 * Nz;
  }
  if (hipMemcpyAsync(d_enc, h_enc, enc_bytes, hipMemcpyHostToDevice, st) != hipSuccess) return cleanup(CM_ERR_HIP);
  if ((rc = cm_encode_stage2(ctx, (const cm_enc*)d_enc, n, N_max, K_max, nnz_max, (int*)d_rep)) != CM_OK) return cleanup(rc);
  if (hipMemcpyAsync(h_rep, d_rep, rep_bytes, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
    cm_set_error("encoding (stage 2) failed: %s", hipGetErrorString(hipGetLastError()));
    return cleanup(CM_ERR_HIP);
  }
  for (int c = 0; c < n; ++c) { ps[c]->max_col_len = h_rep[8 * c + 3]; out[c] = ps[c]; }
  return cleanup(CM_OK);
}

// ---- host inputs: pageable memory staged through pinned slots by a few copy threads ----
// hipMemcpyAsync from pageable memory pins the caller's pages on first touch: 13.6 GB/s for the 1.9 GB of a 256-circuit batch (tools/micro/h2d.hip).
// CM_UPLOAD_THREADS threads copying 4 MB pieces into pinned slots (two per thread, one being filled while the other is on the DMA engine) reach
// 45-55 GB/s, the rate of a pinned source, whatever the caller's allocation is.  Slots, streams and events are made once per device and process.
#define CM_UPLOAD_THREADS 4
#define CM_UPLOAD_SLOT ((size_t)4 << 20)
struct cm_upload_piece { char* dst; const char* src; size_t bytes; };
struct cm_upload_run {
  std::vector<cm_upload_piece> pieces;
  std::atomic<size_t> next{0};
  std::atomic<int> err{0};
  bool started = false;
  void add(void* dst, const void* src, size_t bytes) {
    for (size_t o = 0; o < bytes; o += CM_UPLOAD_SLOT) pieces.push_back({(char*)dst + o, (const char*)src + o, std::min(CM_UPLOAD_SLOT, bytes - o)});
  }
};
struct cm_upload_res {
  int device = 0;
  hipStream_t st[CM_UPLOAD_THREADS] = {};
  char* slot[CM_UPLOAD_THREADS][2] = {};
  hipEvent_t ev[CM_UPLOAD_THREADS][2] = {};
  bool used[CM_UPLOAD_THREADS][2] = {};
  std::mutex use;                                                      // one problem_create_batch at a time per device
  // the copy threads live as long as the process (a fresh thread per upload pays HIP's per-thread initialisation: 0.4 ms on a 6 ms Model.fit):
  // they sleep on `wake`, take the pieces of the posted run from its shared counter, and the last one to finish signals `done`
  std::mutex jm;
  std::condition_variable wake, done;
  cm_upload_run* job = nullptr;
  unsigned gen = 0;
  int active = 0;
};
static void upload_worker(cm_upload_res* r, int t) {
  const bool dev_ok = hipSetDevice(r->device) == hipSuccess;
  unsigned seen = 0;
  for (;;) {
    cm_upload_run* u;
    {
      std::unique_lock<std::mutex> lk(r->jm);
      r->wake.wait(lk, [&] { return r->gen != seen; });
      seen = r->gen;
      u = r->job;
    }
    if (!dev_ok) u->err = 1;
    for (int k = 0; dev_ok; ++k) {
      const size_t i = u->next.fetch_add(1);
      if (i >= u->pieces.size()) break;
      const cm_upload_piece& pc = u->pieces[i];
      const int s = k & 1;
      if (r->used[t][s] && hipEventSynchronize(r->ev[t][s]) != hipSuccess) { u->err = 1; break; }   // the slot's last copy has left it
      memcpy(r->slot[t][s], pc.src, pc.bytes);
      if (hipMemcpyAsync(pc.dst, r->slot[t][s], pc.bytes, hipMemcpyHostToDevice, r->st[t]) != hipSuccess || hipEventRecord(r->ev[t][s], r->st[t]) != hipSuccess) { u->err = 1; break; }
      r->used[t][s] = true;
    }
    if (dev_ok && hipStreamSynchronize(r->st[t]) != hipSuccess) u->err = 1;
    {
      std::lock_guard<std::mutex> lk(r->jm);
      if (--r->active == 0) r->done.notify_all();
    }
  }
}
static cm_upload_res* upload_res(int device) {
  static std::mutex m;
  static std::map<int, cm_upload_res*> all;                            // never freed: a few pinned MB and four sleeping threads per device for the life of the process
  std::lock_guard<std::mutex> lk(m);
  auto it = all.find(device);
  if (it != all.end()) return it->second;
  cm_upload_res* r = new cm_upload_res();
  r->device = device;
  bool ok = true;
  for (int t = 0; ok && t < CM_UPLOAD_THREADS; ++t) {
    ok = hipStreamCreateWithFlags(&r->st[t], hipStreamNonBlocking) == hipSuccess;
    for (int s = 0; ok && s < 2; ++s)
      ok = hipHostMalloc((void**)&r->slot[t][s], CM_UPLOAD_SLOT, hipHostMallocDefault) == hipSuccess &&
           hipEventCreateWithFlags(&r->ev[t][s], hipEventDisableTiming) == hipSuccess;
  }
  if (!ok) { cm_set_error("pinned staging for host inputs: %s", hipGetErrorString(hipGetLastError())); return nullptr; }   // (partial allocations are left to the process)
  for (int t = 0; t < CM_UPLOAD_THREADS; ++t) std::thread(upload_worker, r, t).detach();
  return all[device] = r;
}
static void upload_start(cm_upload_res* r, int device, cm_upload_run& u) {
  (void)device;
  u.next = 0; u.err = 0;
  if (u.pieces.empty()) return;
  std::lock_guard<std::mutex> lk(r->jm);
  r->job = &u;
  r->active = CM_UPLOAD_THREADS;
  ++r->gen;
  u.started = true;
  r->wake.notify_all();
}
static int upload_wait(cm_upload_res* r, cm_upload_run& u) {
  if (u.started) {
    std::unique_lock<std::mutex> lk(r->jm);
    r->done.wait(lk, [&] { return r->active == 0; });
    u.started = false;
  }
  u.pieces.clear();
  if (u.err) { cm_set_error("upload of psc / stim failed: %s", hipGetErrorString(hipGetLastError())); return CM_ERR_HIP; }
  return CM_OK;
}

extern "C" int cm_problem_create_batch_ex(cm_ctx* ctx, int32_t n, const int32_t* N, const int32_t* K, const int32_t* T, const cm_inputs* in, int mem_kind,
                                          cm_problem** out) {
  if (!ctx || !N || !K || !T || !in || !out || n < 1) { cm_set_error("cm_problem_create_batch_ex: NULL argument or n < 1"); return CM_ERR_INVALID; }
  for (int c = 0; c < n; ++c) {
    out[c] = nullptr;
    if (!in[c].psc || (!in[c].stim && !(in[c].stim_layout == CM_STIM_TRIPLES && in[c].n_triples == 0))) { cm_set_error("cm_problem_create_batch: NULL psc / stim (circuit %d)", c); return CM_ERR_INVALID; }
    if (N[c] < 2 || K[c] < 1 || T[c] < 1) { cm_set_error("cm_problem_create_batch: need N >= 2, K >= 1, T >= 1 (circuit %d: %d, %d, %d)", c, N[c], K[c], T[c]); return CM_ERR_INVALID; }
    if ((in[c].psc_dtype != CM_DTYPE_F64 && in[c].psc_dtype != CM_DTYPE_F32) || in[c].stim_layout < CM_STIM_DENSE_F64 || in[c].stim_layout > CM_STIM_TRIPLES ||
        (in[c].stim_layout == CM_STIM_TRIPLES && in[c].n_triples < 0)) { cm_set_error("cm_problem_create_batch_ex: unknown dtype / layout (circuit %d)", c); return CM_ERR_INVALID; }
  }
  CM_HIP(hipSetDevice(ctx->device));
  auto undo = [&](int c0) { for (int c = 0; c < c0; ++c) { cm_problem_destroy(out[c]); out[c] = nullptr; } };
  static const bool plain_upload = getenv("CAVIAR_HIP_PLAIN_UPLOAD") != nullptr;      // diagnostic: hipMemcpyAsync straight from the caller's pages
  size_t host_bytes = 0;                                                              // (a few MB are not worth four threads: one small circuit)
  if (mem_kind == CM_MEM_HOST)
    for (int c = 0; c < n; ++c) {
      const size_t n_stim = (size_t)N[c] * K[c];
      host_bytes += (size_t)K[c] * T[c] * (in[c].psc_dtype == CM_DTYPE_F64 ? 8 : 4) +
                    (in[c].stim_layout == CM_STIM_DENSE_F64 ? n_stim * 8 : in[c].stim_layout == CM_STIM_DENSE_F32 ? n_stim * 4 : (size_t)in[c].n_triples * sizeof(cm_stim_triple));
    }
  if (mem_kind != CM_MEM_HOST || plain_upload || host_bytes < 2 * CM_UPLOAD_SLOT) {
    for (int c0 = 0; c0 < n; c0 += CM_CREATE_CHUNK) {
      const int m = std::min(CM_CREATE_CHUNK, n - c0);
      const int rc = create_chunk(ctx, m, N + c0, K + c0, T + c0, in + c0, mem_kind, out + c0);
      if (rc != CM_OK) { undo(c0); return rc; }
    }
    return CM_OK;
  }
  // Host inputs: the raw bytes of group g + 1 travel (copy threads -> pinned slots -> DMA) while the encoding kernels of group g run; the
  // group is then created from device-resident inputs.
  cm_upload_res* R = upload_res(ctx->device);
  if (!R) return CM_ERR_HIP;
  std::lock_guard<std::mutex> lk(R->use);
  CM_HIP(hipStreamSynchronize(ctx->stream));                            // the copy streams write pool memory: nothing of this context may still be reading it
  struct staged { std::vector<cm_inputs> in; std::vector<void*> bufs; cm_upload_run run; } S[2];
  auto release = [&](staged& g) { for (void* q : g.bufs) cm_pool_free(ctx, q); g.bufs.clear(); };
  auto stage = [&](int c0, staged& g) -> int {
    const int m = std::min(CM_CREATE_CHUNK, n - c0);
    g.in.assign(in + c0, in + c0 + m);
    for (int c = 0; c < m; ++c) {
      cm_inputs& ic = g.in[c];
      const size_t n_psc = (size_t)K[c0 + c] * T[c0 + c], n_stim = (size_t)N[c0 + c] * K[c0 + c];
      const size_t b_psc = n_psc * (ic.psc_dtype == CM_DTYPE_F64 ? 8 : 4);
      const size_t b_stim = ic.stim_layout == CM_STIM_DENSE_F64 ? n_stim * 8 : ic.stim_layout == CM_STIM_DENSE_F32 ? n_stim * 4 : (size_t)ic.n_triples * sizeof(cm_stim_triple);
      void *q = nullptr, *r = nullptr;
      if (cm_pool_alloc(ctx, &q, b_psc) != CM_OK) return CM_ERR_HIP;
      g.bufs.push_back(q);
      g.run.add(q, ic.psc, b_psc); ic.psc = q;
      if (b_stim) {
        if (cm_pool_alloc(ctx, &r, b_stim) != CM_OK) return CM_ERR_HIP;
        g.bufs.push_back(r);
        g.run.add(r, ic.stim, b_stim); ic.stim = r;
      }
    }
    upload_start(R, ctx->device, g.run);
    return CM_OK;
  };
  int rc = stage(0, S[0]);
  for (int c0 = 0, g = 0; c0 < n; c0 += CM_CREATE_CHUNK, ++g) {
    staged& cur = S[g & 1];
    staged& nxt = S[(g + 1) & 1];
    const int rc_up = upload_wait(R, cur.run);
    if (rc == CM_OK) rc = rc_up;
    const bool more = c0 + CM_CREATE_CHUNK < n;
    if (rc == CM_OK && more) rc = stage(c0 + CM_CREATE_CHUNK, nxt);
    if (rc == CM_OK) rc = create_chunk(ctx, std::min(CM_CREATE_CHUNK, n - c0), N + c0, K + c0, T + c0, cur.in.data(), CM_MEM_DEVICE, out + c0);
    if (rc != CM_OK) {
      upload_wait(R, nxt.run);                                          // (waits for whatever was started)
      hipStreamSynchronize(ctx->stream);
      release(cur); release(nxt);
      undo(c0);
      return rc;
    }
    release(cur);                                                       // (create_chunk returns with its stream drained)
  }
  return CM_OK;
}

extern "C" int cm_problem_create_batch(cm_ctx* ctx, int32_t n, const int32_t* N, const int32_t* K, const int32_t* T, const double* const* psc,
                                       const double* const* stim, int mem_kind, cm_problem** out) {
  if (!ctx || !N || !K || !T || !psc || !stim || !out || n < 1) { cm_set_error("cm_problem_create_batch: NULL argument or n < 1"); return CM_ERR_INVALID; }
  std::vector<cm_inputs> in((size_t)n);
  for (int c = 0; c < n; ++c) {
    out[c] = nullptr;
    if (!psc[c] || !stim[c]) { cm_set_error("cm_problem_create_batch: NULL psc / stim (circuit %d)", c); return CM_ERR_INVALID; }
    in[c].psc_dtype = CM_DTYPE_F64; in[c].stim_layout = CM_STIM_DENSE_F64; in[c].psc = psc[c]; in[c].stim = stim[c]; in[c].n_triples = 0;
  }
  return cm_problem_create_batch_ex(ctx, n, N, K, T, in.data(), mem_kind, out);
}

extern "C" int cm_problem_create(cm_ctx* ctx, int32_t N, int32_t K, int32_t T, const double* psc, const double* stim, int mem_kind,
                                 cm_problem** out) {
  if (!ctx || !psc || !stim || !out) { cm_set_error("cm_problem_create: NULL argument"); return CM_ERR_INVALID; }
  if (N < 2 || K < 1 || T < 1) { cm_set_error("cm_problem_create: need N >= 2, K >= 1, T >= 1 (got %d, %d, %d)", N, K, T); return CM_ERR_INVALID; }
  return cm_problem_create_batch(ctx, 1, &N, &K, &T, &psc, &stim, mem_kind, out);
}

extern "C" int cm_problem_nnz(const cm_problem* p, int64_t* nnz, int32_t* n_vals) {
  if (!p) return CM_ERR_INVALID;
  if (nnz) *nnz = p->nnz;
  if (n_vals) *n_vals = p->V;
  return CM_OK;
}

// ------------------------------------------------------------------------------------------------
// host side of the driver (the sequential host-only pieces live in cm_host.cpp, plain C++: also built with sanitizers on the CPU)
// ------------------------------------------------------------------------------------------------
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

// ------------------------------------------------------------------------------------------------
// launch sets
// ------------------------------------------------------------------------------------------------
void cm_fill_pdev(const cm_problem* p, const cm_problem* r, int* mailbox, cm_pdev* d) {
  memset(d, 0, sizeof(*d));
  d->N = p->N; d->K = p->K; d->V = p->V; d->small_cap = p->small_cap; d->max_row_len = p->max_row_len; d->ld_alloc = p->ld; d->nnz = p->nnz;
  d->y = p->y; d->ss = p->ss; d->sum_y = p->sum_y; d->vals = p->vals; d->row_ptr = p->row_ptr; d->col = p->col; d->grp = p->grp;
  d->rowcnt = p->rowcnt; d->col_ptr = p->col_ptr; d->csc_row = p->csc_row; d->csc_pos = p->csc_pos; d->csc_grp = p->csc_grp; d->mask = p->mask; d->colany = p->colany; d->ye = p->ye; d->me = p->me; d->mcnt = p->mcnt; d->touch = p->touch; d->qcnt = p->qcnt;
  { int l = 1; while (l < p->V) l <<= 1; d->mce_ld = l; }
  d->lam = p->lam; d->mu = p->mu; d->beta = p->beta; d->phi = p->phi; d->phi_cov = p->phi_cov; d->scal = p->scal; d->R = p->R; d->z = p->z;
  d->mcE = p->mcE; d->rs = p->rs; d->phi_rep = p->phi_rep; d->phi_dead = p->phi_dead; d->phi_cov_dead = p->phi_dead ? p->phi_dead + 2 * (size_t)p->N : nullptr; d->phi_c_dead = p->phi_dead ? p->phi_dead + 6 * (size_t)p->N : nullptr; d->pslab = p->pslab; d->mu_prior = p->mu_prior; d->beta_prior = p->beta_prior; d->phi_prior = p->phi_prior;
  d->phi_cov_prior = p->phi_cov_prior; d->act_idx = p->act_idx; d->act_inv = p->act_inv; d->n_act = p->n_act; d->alive = p->alive; d->tstamp = p->tstamp; d->tlist_s = p->tlist_s; d->tlist_r = p->tlist_r; d->tcnt = p->tcnt; d->seq_n = p->seq_n;
  d->turn8 = p->turn8; d->seq_off = p->seq_off; d->seq_pos = p->seq_pos; d->chunk_first = p->chunk_first; d->seq_meta = p->seq_meta; d->packed = p->packed;
  d->A = p->A; d->X = p->X; d->Lf = p->Lf; d->Dinv = p->Dinv; d->rhs = p->rhs; d->mailbox = mailbox;
  d->orders_all = r->orders_all; d->inv_all = r->inv_all; d->wtab = r->wtab; d->wstat = r->wstat; d->uexp = r->uexp; d->keys = r->keys;
}

static void set_limits(cm_set* s) {
  s->small_cap_max = s->max_row_len_max = s->max_col_len_max = 0;
  s->nnz_max = 0; s->nnz_min = s->ps.empty() ? 0 : s->ps[0]->nnz;
  s->small_lds_max = 0;
  for (const cm_problem* q : s->ps) {
    if (q->small_cap > 0) s->small_lds_max = std::max(s->small_lds_max, small_lds_bytes(q->small_cap, q->small_cap * q->max_row_len));
    s->small_cap_max = std::max(s->small_cap_max, q->small_cap);
    s->max_row_len_max = std::max(s->max_row_len_max, q->max_row_len);
    s->max_col_len_max = std::max(s->max_col_len_max, q->max_col_len);
    s->nnz_max = std::max<long long>(s->nnz_max, q->nnz);
    s->nnz_min = std::min<long long>(s->nnz_min, q->nnz);
  }
}

// (re)builds the descriptors of a set; the random tables of every circuit are those of the leader
static int set_build(cm_ctx* ctx, cm_set* s) {
  s->B = (int)s->ps.size();
  s->lead = s->ps[0];
  set_limits(s);
  s->na.assign(s->B, 0);
  s->na_max = 0;
  if (s->B == 1) {
    cm_fill_pdev(s->lead, s->lead, ctx->use_mailbox ? ctx->d_mailbox : nullptr, &s->h0);
    s->d_arr = nullptr;
    return CM_OK;
  }
  if (s->B > CM_MAX_BATCH) { cm_set_error("batch of %d circuits exceeds CM_MAX_BATCH = %d", s->B, CM_MAX_BATCH); return CM_ERR_INVALID; }
  if (!ctx->use_mailbox) { cm_set_error("CAVIAR_HIP_NO_MAILBOX is a single-circuit diagnostic: batches need the mailbox"); return CM_ERR_INVALID; }
  const size_t bytes = (size_t)s->B * sizeof(cm_pdev);
  if (ctx->pdev_cap < bytes) {
    if (ctx->d_pdev) hipFree(ctx->d_pdev);
    if (ctx->h_pdev) hipHostFree(ctx->h_pdev);
    ctx->d_pdev = nullptr; ctx->h_pdev = nullptr; ctx->pdev_cap = 0;
    CM_HIP(hipMalloc(&ctx->d_pdev, bytes));
    CM_HIP(hipHostMalloc(&ctx->h_pdev, bytes, hipHostMallocDefault));
    ctx->pdev_cap = bytes;
  }
  CM_HIP(hipStreamSynchronize(ctx->stream));           // the pinned copy may still feed the previous run's upload
  cm_pdev* h = (cm_pdev*)ctx->h_pdev;
  for (int c = 0; c < s->B; ++c)
    cm_fill_pdev(s->ps[c], s->lead, ctx->d_mailbox + 2 * (CM_MAILBOX_BATCH0 + c), &h[c]);
  CM_HIP(hipMemcpyAsync(ctx->d_pdev, h, bytes, hipMemcpyHostToDevice, ctx->stream));
  s->h0 = h[0];
  s->d_arr = (const cm_pdev*)ctx->d_pdev;
  return CM_OK;
}

int cm_set_single(cm_ctx* ctx, cm_problem* p, cm_set* s) {
  s->ps.assign(1, p);
  return set_build(ctx, s);
}

__global__ __launch_bounds__(256) void k_priors_scatter(cm_parg<true> pa, const double* __restrict__ stage, size_t per, const int* __restrict__ slot) {
  const cm_pdev& p = pa.get();
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < per) const_cast<double*>(p.pslab)[i] = stage[(size_t)slot[blockIdx.y] * per + i];
}

// priors -> device, one pinned transfer per circuit (mu_prior | beta_prior | phi_prior | phi_cov_prior | scalars); the state is
// then initialised from them on the device (k_init_state, caviar.py:44-49).  Circuits that are handed the SAME prior arrays (one priors
// dict for a whole batch: the usual case) share one staged copy -- staging 64 KB per circuit was a 16 MB host copy and a 16 MB read over
// PCIe in front of every 256-circuit run.
static int upload_priors(cm_ctx* ctx, cm_set& s, const cm_priors* prs) {
  hipStream_t st = ctx->stream;
  const size_t N = s.lead->N, n = 8 * N + CM_S_COUNT;
  CM_HIP(hipStreamSynchronize(st));                 // the staging buffer may still feed an earlier transfer of this context
  size_t slab_tot = 0;
  for (const cm_problem* q : s.ps) slab_tot += 2 * N + (size_t)q->K + (size_t)q->nnz;
  std::vector<int> slot(s.B), first;                // first[k]: the circuit whose priors slot k holds
  for (int c = 0; c < s.B; ++c) {
    const cm_priors& a = prs[c];
    int k = 0;
    for (; k < (int)first.size(); ++k) {
      const cm_priors& b = prs[first[k]];
      if (a.mu == b.mu && a.beta == b.beta && a.phi == b.phi && a.phi_cov == b.phi_cov && a.shape == b.shape && a.rate == b.rate) break;
    }
    if (k == (int)first.size()) first.push_back(c);
    slot[c] = k;
  }
  const size_t n_slots = s.B == 1 ? 1 : first.size();
  CM_TRY(stage_reserve(ctx, std::max(n * n_slots + (size_t)s.B, slab_tot) * sizeof(double)));
  for (size_t k = 0; k < n_slots; ++k) {
    const cm_priors* pr = &prs[first[k]];
    double* h = (double*)ctx->h_stage + n * k;
    memcpy(h, pr->mu, N * 8); memcpy(h + N, pr->beta, N * 8); memcpy(h + 2 * N, pr->phi, 2 * N * 8); memcpy(h + 4 * N, pr->phi_cov, 4 * N * 8);
    double* sc = h + 8 * N;
    for (int i = 0; i < CM_S_COUNT; ++i) sc[i] = 0.0;
    sc[CM_S_SHAPE] = pr->shape; sc[CM_S_RATE] = pr->rate; sc[CM_S_SHAPE_PRIOR] = pr->shape; sc[CM_S_RATE_PRIOR] = pr->rate;
    if (s.B == 1) CM_HIP(hipMemcpyAsync(s.ps[0]->pslab, h, n * sizeof(double), hipMemcpyHostToDevice, st));
  }
  if (s.B > 1) {                                     // one kernel reads the pinned staging buffer for all circuits (instead of B copies)
    int* hslot = (int*)((double*)ctx->h_stage + n * n_slots);
    memcpy(hslot, slot.data(), (size_t)s.B * sizeof(int));
    double* dbase = nullptr;
    CM_HIP(hipHostGetDevicePointer((void**)&dbase, ctx->h_stage, 0));
    hipLaunchKernelGGL(k_priors_scatter, dim3((unsigned)((n + 255) / 256), s.B), dim3(256), 0, st, cm_parg<true>{s.d_arr}, (const double*)dbase, n,
                       (const int*)(dbase + n * n_slots));
    CM_HIP(hipGetLastError());
  }
  return CM_OK;
}

static cm_host_rows host_rows(const cm_problem* p) {
  cm_host_rows r;
  r.N = p->N; r.K = p->K; r.V = p->V;
  r.row_ptr = p->h_row_ptr.data(); r.col = p->h_col.data(); r.grp = p->h_grp.data();
  r.col_ptr = p->h_col_ptr.data(); r.csc_row = p->h_csc_row.data();
  return r;
}

// end of a run: every circuit's flags and counters into host-mapped pinned memory (read after the final synchronisation: no copies)
template <bool BATCH>
__global__ __launch_bounds__(64) void k_collect_status(cm_parg<BATCH> pa, int* __restrict__ out) {
  const cm_pdev& p = pa.get();
  const int i = threadIdx.x, c = BATCH ? (int)blockIdx.y : 0;
  if (i < 4) out[8 * c + i] = p.n_act[i];
  else if (i < 6) out[8 * c + i] = p.seq_meta[i];
}

// false-negative scan, device side: mu | beta | z of every circuit to / from host-mapped pinned memory (`stage`: `per` doubles per
// circuit), and on the way back the lam entries the scan set to 1 (`hdr`: per circuit (count, start), then the index lists)
template <bool BATCH>
__global__ __launch_bounds__(256) void k_fn_gather(cm_parg<BATCH> pa, double* __restrict__ stage, size_t per) {
  const cm_pdev& p = pa.get();
  const size_t c = BATCH ? blockIdx.y : 0, i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < per) stage[c * per + i] = p.mu[i];                   // the slab: mu [N] | beta [N] | z [K] (| lam, which stays)
}
template <bool BATCH>
__global__ __launch_bounds__(256) void k_fn_scatter(cm_parg<BATCH> pa, const double* __restrict__ stage, size_t per, const int32_t* __restrict__ hdr) {
  const cm_pdev& p = pa.get();
  const size_t c = BATCH ? blockIdx.y : 0, i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < per) p.mu[i] = stage[c * per + i];
  const int B = BATCH ? (int)gridDim.y : 1;
  const int cnt = hdr[2 * c], start = hdr[2 * c + 1];
  const int32_t* __restrict__ idx = hdr + 2 * (size_t)B + start;
  for (size_t e = i; e < (size_t)cnt; e += (size_t)gridDim.x * blockDim.x) p.lam[idx[e]] = 1.0;
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

static int run_set(cm_ctx* ctx, cm_set& s, const cm_priors* prs, const cm_options* o, cm_timings* tm);

// Single exit path: an error return (bad argument, HIP failure, not-SPD, ...) may leave kernels and asynchronous copies queued on
// both streams that still read host arrays of the CALLER (explicit RNG streams) or of the problem (key schedule); both streams
// are drained before the error is handed back, so that nothing outlives the buffers it reads.
static int run_guarded(cm_ctx* ctx, cm_set& s, const cm_priors* prs, const cm_options* o, cm_timings* tm) {
  const int rc = run_set(ctx, s, prs, o, tm);
  if (rc != CM_OK) {
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    if (ctx->side) hipStreamSynchronize(ctx->side);
  }
  return rc;
}

static int check_run_args(const char* who, cm_ctx* ctx, const cm_priors* pr, int n, const cm_options* o) {
  if (!ctx || !pr || !o) { cm_set_error("%s: NULL argument", who); return CM_ERR_INVALID; }
  if (o->abi_version != CM_ABI_VERSION) {
    cm_set_error("%s: cm_options.abi_version is %d, this library is ABI %d (build the caller against this library's caviar_hip.h and fill cm_options with cm_options_default)", who, o->abi_version, CM_ABI_VERSION);
    return CM_ERR_INVALID;
  }
  for (int c = 0; c < n; ++c)
    if (!pr[c].mu || !pr[c].beta || !pr[c].phi || !pr[c].phi_cov) { cm_set_error("%s: NULL prior array (circuit %d)", who, c); return CM_ERR_INVALID; }
  if (o->iters < 0 || o->num_mc_samples < 1) { cm_set_error("%s: iters >= 0 and num_mc_samples >= 1 required", who); return CM_ERR_INVALID; }
  if (o->rng_mode == CM_RNG_EXPLICIT && (!o->update_orders || !o->uniforms)) {
    cm_set_error("%s: CM_RNG_EXPLICIT needs update_orders and uniforms", who); return CM_ERR_INVALID;
  }
  return CM_OK;
}

extern "C" int cm_caviar_run(cm_ctx* ctx, cm_problem* p, const cm_priors* pr, const cm_options* o, cm_timings* tm) {
  if (!p) { cm_set_error("cm_caviar_run: NULL argument"); return CM_ERR_INVALID; }
  CM_TRY(check_run_args("cm_caviar_run", ctx, pr, 1, o));
  if (p->ctx != ctx) { cm_set_error("cm_caviar_run: the problem belongs to another context"); return CM_ERR_INVALID; }
  CM_HIP(hipSetDevice(ctx->device));
  cm_set s;
  s.ps.assign(1, p);
  return run_guarded(ctx, s, pr, o, tm);
}

// the second context of a user-visible one (same device: its own stream, side stream, mailboxes, events; the parent's pool) and the thread that drives it
static int ensure_twin(cm_ctx* ctx) {
  if (!ctx->twin) {
    CM_TRY(cm_ctx_create(ctx->device, nullptr, &ctx->twin));
    ctx->twin->solve_group = ctx->solve_group; ctx->twin->xcd_mode = ctx->xcd_mode; ctx->twin->chol_family = ctx->chol_family;
    ctx->twin->use_mailbox = ctx->use_mailbox;
    ctx->twin->root = ctx;                                         // one pool per user-visible context
  }
  if (!ctx->helper) ctx->helper = new cm_helper();
  return CM_OK;
}

// Lock-step batch (include/caviar_hip.h): circuits with the same launch configuration form a group and share every launch (the
// circuit is the outer grid dimension); groups run one after the other.
extern "C" int cm_caviar_run_batch(cm_ctx* ctx, int32_t n, cm_problem* const* problems, const cm_priors* priors, const cm_options* o,
                                   cm_timings* timings) {
  if (!problems || n < 1) { cm_set_error("cm_caviar_run_batch: need n_circuits >= 1 and a problem array"); return CM_ERR_INVALID; }
  CM_TRY(check_run_args("cm_caviar_run_batch", ctx, priors, n, o));
  for (int c = 0; c < n; ++c) {
    if (!problems[c]) { cm_set_error("cm_caviar_run_batch: problem %d is NULL", c); return CM_ERR_INVALID; }
    if (problems[c]->ctx != ctx) { cm_set_error("cm_caviar_run_batch: problem %d belongs to another context", c); return CM_ERR_INVALID; }
    for (int d = 0; d < c; ++d)
      if (problems[d] == problems[c]) { cm_set_error("cm_caviar_run_batch: problem %d appears twice", c); return CM_ERR_INVALID; }
  }
  CM_HIP(hipSetDevice(ctx->device));
  struct job { cm_set s; std::vector<int> idx; std::vector<cm_priors> prs; std::vector<cm_timings> tms; };
  std::vector<job> jobs;
  std::vector<char> done(n, 0);
  for (int c0 = 0; c0 < n; ++c0) {                                  // groups of circuits that can share launches
    if (done[c0]) continue;
    jobs.emplace_back();
    job& j = jobs.back();
    for (int c = c0; c < n && (int)j.idx.size() < CM_MAX_BATCH; ++c)
      if (!done[c] && cm_same_launch_config(problems[c0], problems[c])) { j.idx.push_back(c); done[c] = 1; }
  }
  // Two lock-step batches at a time: a batch leaves most of the GPU idle in its dependency-bound phases (one Gauss-Seidel workgroup per
  // circuit, the Newton chains of k_iter_mid, the last panels of a factorisation, the false-negative scan), which a second, independent
  // batch on another stream fills.  A single group is split in two halves; results do not depend on the grouping (every circuit is
  // bit-identical to its own single fit).
  static const bool env_off = getenv("CAVIAR_HIP_NO_INTERLEAVE") != nullptr;
  const bool inter = o->interleave == 0 && !env_off && !o->verbose;
  if (inter && jobs.size() == 1 && (int)jobs[0].idx.size() >= CM_INTERLEAVE_MIN) {
    const size_t h = (jobs[0].idx.size() + 1) / 2;
    jobs.emplace_back();
    jobs[1].idx.assign(jobs[0].idx.begin() + h, jobs[0].idx.end());
    jobs[0].idx.resize(h);
  }
  for (job& j : jobs) {
    j.prs.resize(j.idx.size());
    j.tms.resize(j.idx.size());
    for (size_t i = 0; i < j.idx.size(); ++i) { j.prs[i] = priors[j.idx[i]]; j.s.ps.push_back(problems[j.idx[i]]); }
  }
  int rc = CM_OK;
  if (!inter || jobs.size() < 2) {
    for (job& j : jobs)
      if ((rc = run_guarded(ctx, j.s, j.prs.data(), o, timings ? j.tms.data() : nullptr)) != CM_OK) return rc;
  } else {
    CM_TRY(ensure_twin(ctx));
    cm_ctx* tw = ctx->twin;
    CM_HIP(hipEventRecord(ctx->ev[9], ctx->stream));                 // whatever the caller queued on the context's stream comes first on both
    CM_HIP(hipStreamWaitEvent(tw->stream, ctx->ev[9], 0));
    std::atomic<int> next(0), failed(CM_OK);
    std::string twin_msg;
    auto worker = [&](cm_ctx* c, std::string* msg) {
      if (hipSetDevice(c->device) != hipSuccess) { failed = CM_ERR_HIP; if (msg) *msg = "hipSetDevice failed in the helper thread"; return; }
      for (;;) {
        const int ji = next.fetch_add(1);
        if (ji >= (int)jobs.size() || failed.load() != CM_OK) return;
        job& j = jobs[ji];
        const int r = run_guarded(c, j.s, j.prs.data(), o, timings ? j.tms.data() : nullptr);
        if (r != CM_OK) { failed = r; if (msg) *msg = cm_last_error(); return; }
      }
    };
    ctx->helper->start([&] { worker(tw, &twin_msg); });
    worker(ctx, nullptr);
    ctx->helper->wait();
    hipStreamSynchronize(tw->stream);                                // (a finished run has synchronised already; an abandoned one has not)
    if (tw->side) hipStreamSynchronize(tw->side);
    rc = failed.load();
    if (rc != CM_OK) {
      if (!twin_msg.empty()) cm_set_error("%s", twin_msg.c_str());
      return rc;
    }
  }
  if (timings)
    for (job& j : jobs)
      for (size_t i = 0; i < j.idx.size(); ++i) timings[j.idx[i]] = j.tms[i];
  return CM_OK;
}

// circuits that can share launches: same sizes (grids, table layouts), same lane-segment class of V (kernel template), and every
// static property the kernel variants are chosen by
static int seg_class(int V) { int c = 1; while (c < V) c <<= 1; return c; }
bool cm_same_launch_config(const cm_problem* a, const cm_problem* b) {
  if (a->N != b->N || a->K != b->K || seg_class(a->V) != seg_class(b->V)) return false;
  return cm_par_class(0, a->K, a->max_row_len, a->max_col_len, a->nnz) == cm_par_class(0, b->K, b->max_row_len, b->max_col_len, b->nnz);   // (static limits only)
}

static int run_set(cm_ctx* ctx, cm_set& s, const cm_priors* prs, const cm_options* o, cm_timings* tm) {
  hipStream_t st = ctx->stream;
  static const bool trace_run = getenv("CAVIAR_HIP_TRACE_RUN") != nullptr;      // diagnostic: host wall clock of the phases of a run
  const auto t_run0 = std::chrono::steady_clock::now();
  auto t_mark = t_run0;
  auto lap = [&](const char* what) {
    if (!trace_run) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[run B=%d] %-12s %8.1f us\n", (int)s.ps.size(), what, std::chrono::duration<double, std::micro>(now - t_mark).count());
    t_mark = now;
  };
  cm_problem* p = s.ps[0];                           // the leader: sizes and random tables of the whole set
  const int B = (int)s.ps.size();
  const int N = p->N, S = o->num_mc_samples, iters = o->iters;
  for (cm_problem* q : s.ps) q->ran = false;
  s.tlist_it = s.tlist_sweep_it = -1;
  // (the listed column sums, cm_pdev::tstamp, rely on mu = 0 outside the active set; they pay when there are many trial columns to skip: at
  // K = 1000 the marking and the extra roles cost more than the shorter walk saves -- 502 k against 528 k it/s at N = 300 -- at K = 2000 they gain 3.5 %)
  s.lists_ok = p->K >= 1536 || getenv("CAVIAR_HIP_TEST_TRIAL_LISTS") != nullptr;      // (test hook: the listed walk on small shapes)
  for (int c = 0; c < B && s.lists_ok; ++c) {
    if (c > 0 && prs[c].mu == prs[c - 1].mu) continue;          // (the same array as the last circuit's: checked)
    for (int i = 0; i < s.ps[c]->N; ++i)
      if (prs[c].mu[i] != 0.0) { s.lists_ok = false; break; }
  }
  CM_HIP(hipEventRecord(ctx->ev[0], st));

  // RNG streams: sweep orders of all iterations live in orders_all (uploaded, or generated below in one batched launch).  They are
  // data independent: the circuits of a batch share the leader's tables (same seed, N, S).
  if (p->orders_cap < iters) {
    cm_pool_free(ctx, p->orders_all);                 // (tables of an earlier, completed run)
    cm_pool_free(ctx, p->orders_tmp);
    cm_pool_free(ctx, p->inv_all);
    p->orders_all = p->orders_tmp = p->inv_all = nullptr;
    p->orders_cap = 0;
    CM_TRY(pmalloc(ctx, &p->orders_all, (size_t)iters * N));
    CM_TRY(pmalloc(ctx, &p->orders_tmp, (size_t)iters * N));
    CM_TRY(pmalloc(ctx, &p->inv_all, (size_t)iters * N));
    p->orders_cap = iters;
  }
  // (ndtri(u), u) for each of the iters x N x S x 2 uniforms, and the row statistics (iters x N rows).  The two buffers have their own
  // capacities: a re-run with more iterations and fewer samples -- (50, 100) then (100, 50) -- keeps the product but needs twice the rows.
  if (p->wtab_cap < (size_t)iters * N * S * 4 || p->wstat_rows < (size_t)iters * N) {
    cm_pool_free(ctx, p->wtab);
    cm_pool_free(ctx, p->wstat);
    p->wtab = p->wstat = nullptr; p->wtab_cap = 0; p->wstat_rows = 0;
    CM_TRY(pmalloc(ctx, &p->wtab, (size_t)iters * N * S * 4));
    CM_TRY(pmalloc(ctx, &p->wstat, (size_t)iters * N * CM_WSTAT));
    p->wtab_cap = (size_t)iters * N * S * 4;
    p->wstat_rows = (size_t)iters * N;
  }
  if (o->rng_mode == CM_RNG_EXPLICIT) {
    cm_pool_free(ctx, p->uexp); p->uexp = nullptr;
    CM_TRY(pmalloc(ctx, &p->uexp, (size_t)iters * N * S * 2));
    CM_HIP(hipMemcpyAsync(p->uexp, o->uniforms, (size_t)iters * N * S * 2 * sizeof(double), hipMemcpyHostToDevice, st));
    CM_HIP(hipMemcpyAsync(p->orders_all, o->update_orders, (size_t)iters * N * sizeof(int32_t), hipMemcpyHostToDevice, st));
  }
  // Threefry key chain (host, sequential by nature): only iteration 0's keys are produced here; the rest is computed while
  // the GPU works on iteration 0 and turned into permutations / ndtri tables on the side stream (see the loop below).
  std::vector<uint32_t>& keys = p->h_keys;           // lives in the problem: asynchronous uploads read it after this function returns
  uint32_t chain[2] = {(uint32_t)(o->seed >> 32), (uint32_t)(o->seed & 0xFFFFFFFFull)};
  const size_t key_per = (size_t)(N + 3) * 2;
  if (o->rng_mode == CM_RNG_THREEFRY && iters > 0) {
    keys.assign(key_per * iters, 0u);
    if (p->keys_iters < iters) {
      cm_pool_free(ctx, p->keys);
      p->keys = nullptr; p->keys_iters = 0;
      CM_TRY(pmalloc(ctx, &p->keys, keys.size()));
      p->keys_iters = iters;
    }
  }
  lap("tables");
  CM_TRY(set_build(ctx, &s));                        // descriptors (after the table allocations they point to)
  lap("descriptors");
  CM_TRY(upload_priors(ctx, s, prs));
  lap("priors");
  CM_TRY(cm_launch_init_state(ctx, s, o->y_xcorr_thresh));
  // update_phi of an all-zero row depends only on the neuron's trial counts and priors: a per-run constant, produced once on the side
  // stream (first needed by the k_iter_mid of iteration 0, half a millisecond from here) and copied by dead rows in every iteration
  if (iters > 0) {
    CM_HIP(hipEventRecord(ctx->ev[8], st));                          // priors are uploaded
    CM_HIP(hipStreamWaitEvent(ctx->side, ctx->ev[8], 0));
    CM_TRY(cm_launch_phi_dead(ctx, s, ctx->side));
    CM_HIP(hipEventRecord(ctx->ev[9], ctx->side));
  }
  bool rng_deferred = false;
  hipEvent_t pending_ev[2] = {nullptr, nullptr};   // long RNG chunks whose event the main stream has not waited on yet
  int pending_from[2] = {0, 0}, n_pending = 0;     // first iteration each of them serves
  int rng_done = 1;                              // iterations whose keys / orders / tables exist (iteration 0 is done up front)
  if (o->rng_mode == CM_RNG_THREEFRY && iters > 0) {
    cm_host_key_schedule_range(chain, 0, 1, N, keys.data());
    CM_HIP(hipMemcpyAsync(p->keys, keys.data(), key_per * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    CM_TRY(cm_launch_rng(ctx, p, 0, 1, st));                       // :196 update order of iteration 0
    CM_TRY(cm_launch_ndtri_table(ctx, p, 0, 1, S, o->rng_mode, st));
    rng_deferred = iters > 1;
  } else {
    CM_TRY(cm_launch_ndtri_table(ctx, p, 0, iters, S, o->rng_mode, st));   // explicit streams: tabulate the uploaded uniforms
  }
  for (cm_problem* q : s.ps) { if (o->save_histories && iters > 0) CM_TRY(alloc_hist(q, iters)); else free_hist(q); }

  static const bool trace_iters = getenv("CAVIAR_HIP_TRACE_ITERS") != nullptr;   // diagnostic: implies the per-kernel stamps
  static const bool trace_host = getenv("CAVIAR_HIP_TRACE_HOST") != nullptr;     // diagnostic: host clock at every mailbox arrival (no events)
  std::vector<std::chrono::steady_clock::time_point> host_stamps;
  const bool timed = tm != nullptr && (o->timing >= 1 || trace_iters);   // stamps around the chain kernel (events 3, 4) and k_iter_mid (8, 9)
  const bool phases = tm != nullptr && o->timing >= 2;  // all phase stamps (more events and host API calls per iteration)
  while (ctx->pool.size() < (size_t)(iters + 1) * CM_EV_PER_IT) { hipEvent_t e; CM_HIP(hipEventCreate(&e)); ctx->pool.push_back(e); }
  auto EV = [&](int it, int j) { return ctx->pool[(size_t)it * CM_EV_PER_IT + j]; };
  auto post_wait_fallback = [&]() -> int {       // CAVIAR_HIP_NO_MAILBOX=1 (one circuit): classic copy + event instead of the mapped mailbox
    if (!ctx->use_mailbox) {
      CM_HIP(hipMemcpyAsync(ctx->h_pinned + 8, p->n_act, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
      CM_HIP(hipEventRecord(ctx->ev[7], st));
    }
    return CM_OK;
  };
  CM_HIP(hipEventRecord(ctx->ev[1], st));
  CM_TRY(cm_launch_stats(ctx, s, o->spont_orthogonality));           // row sums of lam0
  if (iters > 0) {                                                   // active set + block_update_mu + MC logit of iteration 0
    CM_TRY(cm_launch_iter_mid(ctx, s, -1, iters, S, o->rng_mode, o, 0, 0));
    CM_TRY(post_wait_fallback());
  }
  int last_active = 0;
  std::vector<long long> sum_active(B, 0);
  std::vector<int> last_act(B, 0);
  std::vector<char> fact_it(iters > 0 ? iters : 1, 0);      // iterations that ran a tiled solve
  for (int it = 0; it < iters; ++it) {                               // caviar.py:79
    p->order = p->orders_all + (size_t)it * N;
    if (phases) CM_HIP(hipEventRecord(EV(it, 0), st));
    CM_TRY(cm_wait_active(ctx, s));                                  // the one host wait of an iteration: sizes of the active sets
    if (trace_host) host_stamps.push_back(std::chrono::steady_clock::now());
    last_active = s.na_max;
    bool any_big = false;
    for (int c = 0; c < B; ++c) { sum_active[c] += s.na[c]; last_act[c] = s.na[c]; any_big |= s.na[c] > s.ps[c]->small_cap; }
    if (ctx->mb_flag) { cm_set_error("iteration %d: weight precision matrix not positive definite", it); return CM_ERR_NOT_SPD; }
    if (any_big && timed) CM_HIP(hipEventRecord(EV(it, 10), st));
    if (any_big) CM_TRY(cm_solve_launch(ctx, s, 0, timed ? EV(it, 6) : nullptr, timed ? EV(it, 7) : nullptr));   // :80 block_update_mu, large active sets (small ones: fused kernel)
    if (any_big && timed) CM_HIP(hipEventRecord(EV(it, 11), st));
    fact_it[it] = any_big && s.last_fact;
    if (phases) CM_HIP(hipEventRecord(EV(it, 1), st));
    // (every circuit took the in-LDS solve of the last k_iter_mid: block 0 left the chain tables of this sweep, the pre-sweep packs the records)
    CM_TRY(cm_launch_sweep(ctx, s, it, o, timed ? EV(it, 3) : nullptr, -1, (!any_big && s.scan_it == it) ? 1 : 0));  // :81  update_lam
    if (timed) CM_HIP(hipEventRecord(EV(it, 4), st));
    CM_TRY(cm_launch_stats(ctx, s, o->spont_orthogonality, cm_sweep_quiescent(ctx, s, it, o), it));   // row + column statistics of the new lam
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
    if (o->save_histories) for (cm_problem* q : s.ps) CM_TRY(record_hist(ctx, q, it, 0));   // mu, beta, lam of this iteration (:90-92)
    // :84-88 update_sigma, update_phi, z, spont_rate; then block_update_mu (small sets) and the MC logit of iteration it + 1
    if (it == 0) CM_HIP(hipStreamWaitEvent(st, ctx->ev[9], 0));      // the dead-row constants (long done)
    if (timed) CM_HIP(hipEventRecord(EV(it, 8), st));
    CM_TRY(cm_launch_iter_mid(ctx, s, it, iters, S, o->rng_mode, o, 1, 1, 3, (timed && B > 1) ? EV(it, 2) : nullptr));
    if (timed) CM_HIP(hipEventRecord(EV(it, 9), st));
    CM_TRY(post_wait_fallback());
    if (o->save_histories) for (cm_problem* q : s.ps) CM_TRY(record_hist(ctx, q, it, 1));   // shape, rate, phi, phi_cov, z
    if (phases) CM_HIP(hipEventRecord(EV(it, 5), st));
    if (o->verbose >= 3) {
      CM_HIP(hipStreamSynchronize(st));
      fprintf(stderr, "[caviar_hip] it %d largest active set %d (%d circuit%s)\n", it, last_active, B, B == 1 ? "" : "s");
    }
  }
  CM_HIP(hipEventRecord(ctx->ev[2], st));
  lap("iterations");

  if (o->fn_scan && o->verbose == 0 && cm_fn_scan_fits(s)) {         // caviar.py:94-98
    // reconnect_spont_cells on the device, one workgroup per circuit, straight on mu | beta | z | lam in HBM (cm_fnscan.hip): no host
    // round trip, no synchronisation; the final update_phi follows on the stream
    CM_TRY(cm_launch_fn_scan(ctx, s, o->msrmp, o->minimum_spike_count));
    CM_TRY(cm_launch_update_phi(ctx, s, st, iters > 0 ? 1 : 0, 1));  // (iters == 0: the per-run constants were not produced)
  } else if (o->fn_scan) {
    // Host path (verbose runs: the reference's progress messages; circuits too large for the scan kernel's LDS).
    // mu | beta | z | lam are one device slab per circuit.  The scan reads mu and z and only ever WRITES lam (entries set to 1), so
    // only mu | beta | z travel and lam stays in HBM; the few entries a reconnection sets go back as an index list.  Both directions
    // are ONE kernel for the whole set, reading / writing host-mapped pinned memory directly.
    const size_t per = 2 * (size_t)N + p->K;                           // doubles per circuit (N, K are those of the group)
    size_t nnz_tot = 0;
    for (int c = 0; c < B; ++c) nnz_tot += (size_t)s.ps[c]->nnz;
    CM_TRY(stage_reserve(ctx, (size_t)B * per * sizeof(double) + ((size_t)2 * B + nnz_tot) * sizeof(int32_t)));
    double* hbase = (double*)ctx->h_stage;
    int32_t* hhdr = (int32_t*)(hbase + (size_t)B * per);               // per circuit: (count, start) of its index list, then the lists
    int32_t* hidx = hhdr + 2 * (size_t)B;
    double* dbase = nullptr;
    CM_HIP(hipHostGetDevicePointer((void**)&dbase, ctx->h_stage, 0));
    CM_LAUNCH_SET(k_fn_gather, s, (unsigned)((per + 255) / 256), 256, 0, st, dbase, per);
    CM_HIP(hipStreamSynchronize(st));
    std::vector<std::vector<int32_t>> ones(B);
    for (int c = 0; c < B; ++c) CM_TRY(cm_host_mirror(ctx, s.ps[c]));
    for (int c = 0; c < B; ++c) {
      const cm_problem* q = s.ps[c];
      double* h = hbase + (size_t)c * per;
      cm_host_reconnect(host_rows(q), h, h + N, nullptr, h + 2 * (size_t)N, o->msrmp, o->minimum_spike_count, o->verbose, &ones[c]);
    }
    size_t start = 0;
    for (int c = 0; c < B; ++c) {                                      // lam[e] = 1 for the reconnected cells' events (:136)
      hhdr[2 * c] = (int32_t)ones[c].size();
      hhdr[2 * c + 1] = (int32_t)start;
      if (!ones[c].empty()) memcpy(hidx + start, ones[c].data(), ones[c].size() * sizeof(int32_t));
      start += ones[c].size();
    }
    CM_LAUNCH_SET(k_fn_scatter, s, (unsigned)((per + 255) / 256), 256, 0, st, (const double*)dbase, per, (const int32_t*)(dbase + (size_t)B * per));
    CM_TRY(cm_launch_update_phi(ctx, s, st, iters > 0 ? 1 : 0, 1));  // (iters == 0: the per-run constants were not produced)
  }
  CM_HIP(hipEventRecord(ctx->ev[3], st));
  lap("fn_scan");
  CM_LAUNCH_SET(k_collect_status, s, 1, 64, 0, st, ctx->d_mailbox + CM_STATUS_OFF);
  CM_HIP(hipStreamSynchronize(st));
  lap("final sync");
  const int32_t* status = ctx->h_pinned + CM_STATUS_OFF;
  if (iters > 0) {
    for (int c = 0; c < B; ++c) {
      if (status[8 * c + 1]) { cm_set_error("weight precision matrix not positive definite (circuit %d of the set)", c); return CM_ERR_NOT_SPD; }
      if (status[8 * c + 2]) { cm_set_error("sweep ticket wait timed out (internal error, circuit %d of the set)", c); return CM_ERR_STATE; }
    }
  }
  if (ctx->dbg_block0) {
    long long h[64 * 8];
    if (hipMemcpy(h, ctx->dbg_block0, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess)
      for (int it = 0; it < iters && it < 64; ++it) {
        const long long* d = h + 8 * it;
        fprintf(stderr, "it %2d: block 0: compaction %6.1f sigma %6.1f solve %6.1f (n = %lld) | block 1: z-loop %6.1f (passes to %lld), ends %+6.1f us after block 0\n",
                it, (d[1] - d[0]) * 0.01, (d[2] - d[1]) * 0.01, d[4] ? (d[4] - d[2]) * 0.01 : 0.0, d[6], (d[3] - d[7]) * 0.01, d[5], d[4] ? (d[3] - d[4]) * 0.01 : 0.0);
      }
  }
  if (trace_host)
    for (size_t i = 1; i < host_stamps.size(); ++i)
      fprintf(stderr, "mailbox %2zu -> %2zu: %7.1f us\n", i - 1, i, std::chrono::duration<double, std::micro>(host_stamps[i] - host_stamps[i - 1]).count());
  if (tm) {
    float a = 0, b = 0, c = 0;
    hipEventElapsedTime(&a, ctx->ev[0], ctx->ev[3]);
    hipEventElapsedTime(&b, ctx->ev[1], ctx->ev[2]);
    hipEventElapsedTime(&c, ctx->ev[2], ctx->ev[3]);
    cm_timings t0;
    memset(&t0, 0, sizeof(t0));
    t0.total_ms = a; t0.iters_ms = b; t0.fn_scan_ms = c; t0.iters = iters; t0.batch = B;
    t0.fact_launches = s.fact_launches; t0.fact_tile_updates = s.fact_tile_updates; t0.fact_panel_tiles = s.fact_panel_tiles;
    t0.fact_tri_tiles = s.fact_tri_tiles; t0.fact_products = s.fact_products; t0.inv_products = s.inv_products;
    for (int it = 0; timed && it < iters; ++it) {
      float d = 0;
      hipEventElapsedTime(&d, EV(it, 3), EV(it, 4));
      t0.sweep_seq_ms += d;
      if (B > 1) {                                                            // k_iter_mid | k_mc_dead
        hipEventElapsedTime(&d, EV(it, 8), EV(it, 2)); t0.mid_ms += d;
        hipEventElapsedTime(&d, EV(it, 2), EV(it, 9)); t0.mc_dead_ms += d;
      } else {
        hipEventElapsedTime(&d, EV(it, 8), EV(it, 9));
        t0.mid_ms += d;
      }
      if (fact_it[it] && hipEventElapsedTime(&d, EV(it, 6), EV(it, 7)) == hipSuccess) t0.fact_ms += d;
      if (fact_it[it] && hipEventElapsedTime(&d, EV(it, 10), EV(it, 11)) == hipSuccess) t0.tiled_ms += d;
      if (fact_it[it]) ++t0.tiled_solves;
      if (trace_iters && it > 0) {                                            // diagnostic: wall time of every iteration
        float e = 0;
        hipEventElapsedTime(&e, EV(it - 1, 9), EV(it, 9));
        fprintf(stderr, "it %2d %8.1f us (k_iter_mid incl. its launch gap %.1f)\n", it, e * 1e3, d * 1e3);
      }
      if (!phases) continue;
      hipEventElapsedTime(&d, EV(it, 0), EV(it, 1)); t0.solve_ms += d;        // host wait + large-set solve (small sets are fused)
      hipEventElapsedTime(&d, EV(it, 1), EV(it, 3)); t0.sweep_indep_ms += d;  // pre-sweep + pack
      hipEventElapsedTime(&d, EV(it, 4), EV(it, 5)); t0.post_ms += d;         // stats + fused mid-iteration kernel
    }
    for (int ci = 0; ci < B; ++ci) {                                          // the device times are those of the whole set
      tm[ci] = t0;
      tm[ci].last_active = last_act[ci];
      tm[ci].sum_active = sum_active[ci];
      tm[ci].sum_seq_rows = status[8 * ci + 4]; tm[ci].sum_seq_records = status[8 * ci + 5];
    }
  }
  for (cm_problem* q : s.ps) q->ran = true;
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
  if (r->lam || (p->hist_iters > 0 && r->lam_hist)) CM_TRY(cm_host_mirror(ctx, p));
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

// Results of a group of circuits with ONE device-to-host copy: a kernel packs mu, beta, phi, phi_cov, z, shape, rate of every circuit
// into one buffer (per-circuit fetches are six small copies and a wait each: 0.1 ms per circuit, half the time of fitting it in a batch).
// Circuits whose result asks for lam or histories go through cm_caviar_fetch.
struct cm_fetch_desc { const double *mu, *beta, *phi, *phi_cov, *z, *scal; int N, K; long long off; };
__global__ __launch_bounds__(256) void k_pack_results(const cm_fetch_desc* __restrict__ arr, double* __restrict__ out) {
  const cm_fetch_desc& d = arr[blockIdx.y];
  const long long N = d.N, K = d.K, L = 8 * N + K + 2;
  double* o = out + d.off;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < L; i += (long long)gridDim.x * blockDim.x) {
    double v;
    if (i < N) v = d.mu[i];
    else if (i < 2 * N) v = d.beta[i - N];
    else if (i < 4 * N) v = d.phi[i - 2 * N];
    else if (i < 8 * N) v = d.phi_cov[i - 4 * N];
    else if (i < 8 * N + K) v = d.z[i - 8 * N];
    else v = d.scal[i - 8 * N - K == 0 ? CM_S_SHAPE : CM_S_RATE];
    o[i] = v;
  }
}
extern "C" int cm_caviar_fetch_batch(cm_ctx* ctx, int32_t n, cm_problem* const* problems, cm_result* results) {
  if (!ctx || !problems || !results || n < 1) { cm_set_error("cm_caviar_fetch_batch: NULL argument or n < 1"); return CM_ERR_INVALID; }
  CM_HIP(hipSetDevice(ctx->device));
  std::vector<int> packed;
  std::vector<cm_fetch_desc> desc;
  long long total = 0, L_max = 0;
  for (int c = 0; c < n; ++c) {
    cm_problem* p = problems[c];
    cm_result& r = results[c];
    if (!p) { cm_set_error("cm_caviar_fetch_batch: problem %d is NULL", c); return CM_ERR_INVALID; }
    if (p->ctx != ctx) { cm_set_error("cm_caviar_fetch_batch: problem %d belongs to another context (its results are ordered on that context's stream)", c); return CM_ERR_INVALID; }
    if (!p->ran) { cm_set_error("cm_caviar_fetch_batch: no completed run on problem %d", c); return CM_ERR_STATE; }
    const bool hist = p->hist_iters > 0 && (r.mu_hist || r.beta_hist || r.lam_hist || r.shape_hist || r.rate_hist || r.phi_hist || r.phi_cov_hist || r.z_hist);
    if (r.lam || hist) continue;                                      // below, one by one
    const long long L = 8ll * p->N + p->K + 2;
    desc.push_back({p->mu, p->beta, p->phi, p->phi_cov, p->z, p->scal, p->N, p->K, total});
    packed.push_back(c);
    total += L; L_max = std::max(L_max, L);
  }
  if (!packed.empty()) {
    hipStream_t st = ctx->stream;
    const size_t dbytes = desc.size() * sizeof(cm_fetch_desc), obytes = (size_t)total * sizeof(double);
    void *d_desc = nullptr, *d_out = nullptr;
    CM_TRY(cm_pool_alloc(ctx, &d_desc, dbytes));
    if (cm_pool_alloc(ctx, &d_out, obytes) != CM_OK) { cm_pool_free(ctx, d_desc); return CM_ERR_HIP; }
    auto done = [&](int code) { hipStreamSynchronize(st); cm_pool_free(ctx, d_desc); cm_pool_free(ctx, d_out); return code; };
    int rc = stage_reserve(ctx, dbytes + obytes);
    if (rc != CM_OK) return done(rc);
    char* h = (char*)ctx->h_stage;
    memcpy(h, desc.data(), dbytes);
    if (hipMemcpyAsync(d_desc, h, dbytes, hipMemcpyHostToDevice, st) != hipSuccess) return done(CM_ERR_HIP);
    hipLaunchKernelGGL(k_pack_results, dim3((unsigned)std::min<long long>((L_max + 255) / 256, 64), (unsigned)desc.size()), dim3(256), 0, st,
                       (const cm_fetch_desc*)d_desc, (double*)d_out);
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(h + dbytes, d_out, obytes, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
      cm_set_error("cm_caviar_fetch_batch: %s", hipGetErrorString(hipGetLastError()));
      return done(CM_ERR_HIP);
    }
    const double* ho = (const double*)(h + dbytes);
    for (size_t i = 0; i < packed.size(); ++i) {
      cm_result& r = results[packed[i]];
      const cm_fetch_desc& d = desc[i];
      const size_t N = d.N, K = d.K;
      const double* o = ho + d.off;
      if (r.mu) memcpy(r.mu, o, N * 8);
      if (r.beta) memcpy(r.beta, o + N, N * 8);
      if (r.phi) memcpy(r.phi, o + 2 * N, 2 * N * 8);
      if (r.phi_cov) memcpy(r.phi_cov, o + 4 * N, 4 * N * 8);
      if (r.z) memcpy(r.z, o + 8 * N, K * 8);
      r.shape = o[8 * N + K]; r.rate = o[8 * N + K + 1];
    }
    done(CM_OK);
  }
  size_t pi = 0;
  for (int c = 0; c < n; ++c) {
    if (pi < packed.size() && packed[pi] == c) { ++pi; continue; }
    CM_TRY(cm_caviar_fetch(ctx, problems[c], &results[c]));
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

// Batch of circuits over the GPUs of this process: circuit c goes to device_ids[c % n_devices]; every device gets one
// host thread with its own context, which creates its circuits, runs them as ONE lock-step batch, fetches and destroys them.
static int fit_batch_impl(int32_t n, const int32_t* N, const int32_t* K, const int32_t* T, const cm_inputs* in, const cm_priors* priors, const cm_options* o,
                          const int32_t* device_ids, int32_t n_devices, cm_result* results, cm_timings* timings);
extern "C" int cm_caviar_fit_batch(int32_t n, const int32_t* N, const int32_t* K, const int32_t* T, const double* const* psc,
                                   const double* const* stim, const cm_priors* priors, const cm_options* o, const int32_t* device_ids,
                                   int32_t n_devices, cm_result* results, cm_timings* timings) {
  if (n < 1 || !N || !K || !T || !psc || !stim || !priors || !o || !results) { cm_set_error("cm_caviar_fit_batch: NULL argument or n_circuits < 1"); return CM_ERR_INVALID; }
  std::vector<cm_inputs> in((size_t)n);
  for (int c = 0; c < n; ++c) { in[c].psc_dtype = CM_DTYPE_F64; in[c].stim_layout = CM_STIM_DENSE_F64; in[c].psc = psc[c]; in[c].stim = stim[c]; in[c].n_triples = 0; }
  return fit_batch_impl(n, N, K, T, in.data(), priors, o, device_ids, n_devices, results, timings);
}
extern "C" int cm_caviar_fit_batch_ex(int32_t n, const int32_t* N, const int32_t* K, const int32_t* T, const cm_inputs* inputs, const cm_priors* priors,
                                      const cm_options* o, const int32_t* device_ids, int32_t n_devices, cm_result* results, cm_timings* timings) {
  if (n < 1 || !N || !K || !T || !inputs || !priors || !o || !results) { cm_set_error("cm_caviar_fit_batch_ex: NULL argument or n_circuits < 1"); return CM_ERR_INVALID; }
  return fit_batch_impl(n, N, K, T, inputs, priors, o, device_ids, n_devices, results, timings);
}
// The contexts of cm_caviar_fit_batch* stay with the process, one per device, their memory pools warm: building a context's pool for 256 circuits
// and handing ~20 GB back to the driver cost 30 + 115 ms per call, more than the fit (CAVIAR_HIP_FIT_BATCH_TIMES).  cm_caviar_fit_batch_release
// destroys them; CAVIAR_HIP_FIT_BATCH_NO_CACHE keeps none.
static std::mutex g_fb_m;
static std::map<int, cm_ctx*> g_fb_ctx;
static cm_ctx* fb_take(int device) {
  std::lock_guard<std::mutex> lk(g_fb_m);
  auto it = g_fb_ctx.find(device);
  if (it == g_fb_ctx.end()) return nullptr;
  cm_ctx* c = it->second;
  g_fb_ctx.erase(it);
  return c;
}
static void fb_give(int device, cm_ctx* c) {
  static const bool no_cache = getenv("CAVIAR_HIP_FIT_BATCH_NO_CACHE") != nullptr;
  if (!no_cache) {
    std::lock_guard<std::mutex> lk(g_fb_m);
    if (!g_fb_ctx.count(device)) { g_fb_ctx[device] = c; return; }
  }
  cm_ctx_destroy(c);                                                   // (a concurrent call on the same device already left its own)
}
extern "C" int cm_caviar_fit_batch_release(void) {
  std::vector<cm_ctx*> all;
  {
    std::lock_guard<std::mutex> lk(g_fb_m);
    for (auto& kv : g_fb_ctx) all.push_back(kv.second);
    g_fb_ctx.clear();
  }
  int rc = CM_OK;
  for (cm_ctx* c : all) { const int r = cm_ctx_destroy(c); if (r != CM_OK) rc = r; }
  return rc;
}
static int fit_batch_impl(int32_t n, const int32_t* N, const int32_t* K, const int32_t* T, const cm_inputs* in, const cm_priors* priors, const cm_options* o,
                          const int32_t* device_ids, int32_t n_devices, cm_result* results, cm_timings* timings) {
  const int32_t dev0 = 0;
  if (!device_ids || n_devices < 1) { device_ids = &dev0; n_devices = 1; }
  std::vector<int> rcs(n_devices, CM_OK);
  std::vector<std::string> msgs(n_devices);
  auto work = [&](int d) {
    auto fail = [&](int rc) { rcs[d] = rc; msgs[d] = cm_last_error(); };
    std::vector<int> mine;
    for (int c = d; c < n; c += n_devices) mine.push_back(c);
    if (mine.empty()) return;
    cm_ctx* ctx = nullptr;
    static const bool show = getenv("CAVIAR_HIP_FIT_BATCH_TIMES") != nullptr;        // diagnostic: where the call's time goes (stderr)
    double tq[8] = {}; int nq = 0;
    auto stamp = [&] { if (nq < 8) tq[nq++] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    stamp();
    int rc = CM_OK;
    if (!(ctx = fb_take(device_ids[d]))) rc = cm_ctx_create(device_ids[d], nullptr, &ctx);
    if (rc) return fail(rc);
    stamp();
    std::vector<cm_problem*> ps(mine.size(), nullptr);
    std::vector<cm_priors> prs;
    std::vector<int32_t> n_(mine.size()), k_(mine.size()), t_(mine.size());
    std::vector<cm_inputs> in_(mine.size());
    std::vector<cm_result> res_(mine.size());
    for (size_t i = 0; i < mine.size(); ++i) {
      const int c = mine[i];
      n_[i] = N[c]; k_[i] = K[c]; t_[i] = T[c]; in_[i] = in[c];
      prs.push_back(priors[c]);
      res_[i] = results[c];
    }
    std::vector<cm_timings> tms(mine.size());
    // From 2 x 64 circuits on, the call is a two-stage pipeline: the first half is created (upload + encoding) on the context and fitted by the
    // context's helper thread while this thread creates the second half on the twin context -- its bytes cross PCIe under the first half's fit --
    // and fits it there: the two halves are the two lock-step batches cm_caviar_run_batch would have in flight anyway (cm_options.interleave).
    static const bool no_pipe = getenv("CAVIAR_HIP_FIT_BATCH_NO_PIPELINE") != nullptr;
    const bool pipe = !no_pipe && mine.size() >= 2 * (size_t)CM_CREATE_CHUNK && o->verbose == 0;
    if (!pipe) {
      rc = cm_problem_create_batch_ex(ctx, (int)mine.size(), n_.data(), k_.data(), t_.data(), in_.data(), CM_MEM_HOST, ps.data());
      stamp();
      if (!rc) rc = cm_caviar_run_batch(ctx, (int)ps.size(), ps.data(), prs.data(), o, timings ? tms.data() : nullptr);
      stamp();
      if (!rc) rc = cm_caviar_fetch_batch(ctx, (int)ps.size(), ps.data(), res_.data());
      stamp();
    } else {
      const int nA = (int)(((mine.size() / 2 + CM_CREATE_CHUNK - 1) / CM_CREATE_CHUNK) * CM_CREATE_CHUNK), nB = (int)mine.size() - nA;
      cm_options oh = *o;
      oh.interleave = 1;                                               // each half is one lock-step batch on its context's stream
      std::string msgA;
      int rcA = CM_OK, rcB = CM_OK;
      bool started = false;
      rc = ensure_twin(ctx);
      if (!rc) rc = cm_problem_create_batch_ex(ctx, nA, n_.data(), k_.data(), t_.data(), in_.data(), CM_MEM_HOST, ps.data());
      stamp();
      if (!rc) {
        cm_ctx* c0 = ctx;
        ctx->helper->start([&, c0] {
          if (hipSetDevice(c0->device) != hipSuccess) { rcA = CM_ERR_HIP; msgA = "hipSetDevice failed in the helper thread"; return; }
          rcA = cm_caviar_run_batch(c0, nA, ps.data(), prs.data(), &oh, timings ? tms.data() : nullptr);
          if (!rcA) rcA = cm_caviar_fetch_batch(c0, nA, ps.data(), res_.data());
          if (rcA) msgA = cm_last_error();
        });
        started = true;
        rcB = cm_problem_create_batch_ex(ctx->twin, nB, n_.data() + nA, k_.data() + nA, t_.data() + nA, in_.data() + nA, CM_MEM_HOST, ps.data() + nA);
        stamp();
        if (!rcB) rcB = cm_caviar_run_batch(ctx->twin, nB, ps.data() + nA, prs.data() + nA, &oh, timings ? tms.data() + nA : nullptr);
        if (!rcB) rcB = cm_caviar_fetch_batch(ctx->twin, nB, ps.data() + nA, res_.data() + nA);
      }
      if (started) ctx->helper->wait();
      stamp();
      if (!rc && rcA) { rc = rcA; cm_set_error("%s", msgA.c_str()); }
      else if (!rc && rcB) rc = rcB;                                   // (this thread's last error is the second half's)
      stamp();
    }
    for (size_t i = 0; !rc && i < mine.size(); ++i) {
      results[mine[i]].shape = res_[i].shape; results[mine[i]].rate = res_[i].rate;
      if (timings) timings[mine[i]] = tms[i];
    }
    if (rc) fail(rc);
    for (cm_problem* q : ps) if (q) cm_problem_destroy(q);
    stamp();
    if (rc) cm_ctx_destroy(ctx); else fb_give(device_ids[d], ctx);
    stamp();
    if (show && !pipe) fprintf(stderr, "cm_caviar_fit_batch device %d: context %.1f ms, create (upload + encoding) %.1f, fit %.1f, fetch %.1f, destroy %.1f, context kept / destroyed %.1f\n",
                               device_ids[d], tq[1] - tq[0], tq[2] - tq[1], tq[3] - tq[2], tq[4] - tq[3], tq[5] - tq[4], tq[6] - tq[5]);
    if (show && pipe) fprintf(stderr, "cm_caviar_fit_batch device %d (two halves): context %.1f ms, create of the first half %.1f, of the second (under the first's fit) %.1f, "
                                      "until both are fitted and fetched %.1f, destroy %.1f, context kept / destroyed %.1f\n",
                              device_ids[d], tq[1] - tq[0], tq[2] - tq[1], tq[3] - tq[2], tq[4] - tq[3], tq[6] - tq[5], tq[7] - tq[6]);
  };
  if (n_devices == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (int d = 0; d < n_devices; ++d) th.emplace_back(work, d);
    for (auto& t : th) t.join();
  }
  for (int d = 0; d < n_devices; ++d)
    if (rcs[d] != CM_OK) { cm_set_error("device %d: %s", device_ids[d], msgs[d].c_str()); return rcs[d]; }
  return CM_OK;
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
  cm_set set;
  if (!rc) rc = cm_set_single(ctx, p, &set);
  if (!rc) rc = cm_launch_entry_copies(ctx, set);              // per-entry y / mask from what the hook uploaded
  if (!rc) rc = cm_launch_stats(ctx, set, 0.1);
  if (!rc) rc = cm_launch_post_single(ctx, set, -1, nullptr, 0);
  if (!rc) rc = cm_solve_mu(ctx, set, &na);
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
  cm_set set;
  if (!rc) rc = cm_set_single(ctx, p, &set);
  if (!rc) rc = cm_launch_entry_copies(ctx, set);              // per-entry y / mask from what the hook uploaded
  if (!rc) rc = cm_launch_update_phi(ctx, set, st);
  if (!rc && (hipMemcpyAsync(phi, p->phi, (size_t)N * 16, hipMemcpyDeviceToHost, st) != hipSuccess ||
              hipMemcpyAsync(phi_cov, p->phi_cov, (size_t)N * 32, hipMemcpyDeviceToHost, st) != hipSuccess ||
              hipStreamSynchronize(st) != hipSuccess)) { cm_set_error("download failed: %s", hipGetErrorString(hipGetLastError())); rc = CM_ERR_HIP; }
  cm_problem_destroy(p);
  return rc;
}

// update_sigma (caviar.py:238-244) and estimate_spont_act_soft_thresh + spont_rate (caviar.py:147-163, :88) on a dense lam: the
// single-workgroup phases exactly as block 0 of the fused mid-iteration kernel runs them
static int post_hook(cm_ctx* ctx, int N, int K, const double* stim, const double* lam, const double* y, const uint8_t* mask, const double* mu,
                     const double* beta, double shape_prior, double rate_prior, int it, const cm_options* o, double* z, double* scal_out) {
  cm_problem* p = nullptr;
  CM_TRY(problem_from_dense(ctx, N, K, stim, lam, y, &p));
  hipStream_t st = ctx->stream;
  double h[CM_S_COUNT] = {0};
  h[CM_S_SHAPE] = h[CM_S_SHAPE_PRIOR] = shape_prior; h[CM_S_RATE] = h[CM_S_RATE_PRIOR] = rate_prior;
  std::vector<uint8_t> ones((size_t)K, 1);
  std::vector<double> zeros((size_t)std::max(N, K), 0.0);
  int rc = CM_OK;
  if (hipMemcpy(p->scal, h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(p->mu, mu, (size_t)N * 8, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(p->beta, beta ? beta : zeros.data(), (size_t)N * 8, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(p->mask, mask ? mask : ones.data(), (size_t)K, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(p->z, z ? z : zeros.data(), (size_t)K * 8, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemset(p->n_act, 0, 4 * sizeof(int)) != hipSuccess) { cm_set_error("upload failed"); rc = CM_ERR_HIP; }
  cm_set set;
  if (!rc) rc = cm_set_single(ctx, p, &set);
  if (!rc) rc = cm_launch_entry_copies(ctx, set);              // per-entry y / mask from what the hook uploaded
  if (!rc) rc = cm_launch_sum_y(ctx, set);
  if (!rc) rc = cm_launch_stats(ctx, set, o->spont_orthogonality);
  if (!rc) rc = cm_launch_post_single(ctx, set, it, o, 1);
  if (!rc && ((z && hipMemcpyAsync(z, p->z, (size_t)K * 8, hipMemcpyDeviceToHost, st) != hipSuccess) ||
              hipMemcpyAsync(scal_out, p->scal, CM_S_COUNT * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
              hipStreamSynchronize(st) != hipSuccess)) { cm_set_error("download failed: %s", hipGetErrorString(hipGetLastError())); rc = CM_ERR_HIP; }
  cm_problem_destroy(p);
  return rc;
}

extern "C" int cm_k_sigma(cm_ctx* ctx, int32_t N, int32_t K, const double* stim, const double* lam, const double* y, const double* mu,
                          const double* beta, double shape_prior, double rate_prior, double* shape, double* rate) {
  if (!ctx || !stim || !lam || !y || !mu || !beta || !shape || !rate) { cm_set_error("cm_k_sigma: NULL argument"); return CM_ERR_INVALID; }
  CM_HIP(hipSetDevice(ctx->device));
  cm_options o;
  cm_options_default(&o);
  o.max_backtrack_iters = 0;                        // no soft-threshold passes: only update_sigma is wanted
  double sc[CM_S_COUNT];
  CM_TRY(post_hook(ctx, N, K, stim, lam, y, nullptr, mu, beta, shape_prior, rate_prior, 0, &o, nullptr, sc));
  *shape = sc[CM_S_SHAPE]; *rate = sc[CM_S_RATE];
  return CM_OK;
}

extern "C" int cm_k_spont(cm_ctx* ctx, int32_t N, int32_t K, const double* stim, const double* lam, const double* y, const uint8_t* mask,
                          const double* mu, int32_t it, int32_t max_backtrack_iters, double penalty, double scale_factor, double tol,
                          double spont_orthogonality, double* z, double* spont_rate) {
  if (!ctx || !stim || !lam || !y || !mask || !mu || !z) { cm_set_error("cm_k_spont: NULL argument"); return CM_ERR_INVALID; }
  CM_HIP(hipSetDevice(ctx->device));
  cm_options o;
  cm_options_default(&o);
  o.max_backtrack_iters = max_backtrack_iters; o.penalty = penalty; o.scale_factor = scale_factor; o.tol = tol; o.spont_orthogonality = spont_orthogonality;
  double sc[CM_S_COUNT];
  CM_TRY(post_hook(ctx, N, K, stim, lam, y, mask, mu, nullptr, 1.0, 1.0, it, &o, z, sc));
  if (spont_rate) *spont_rate = sc[CM_S_SPONT_RATE];
  return CM_OK;
}

// update_lam (caviar.py:190-231): one Gauss-Seidel sweep over the neurons in the order jax.random.choice(key) gives, the
// Monte-Carlo logit from phi / phi_cov, the isotonic gate; `key` is the carried jax key, replaced by the key update_lam returns
extern "C" int cm_k_lam_sweep(cm_ctx* ctx, int32_t N, int32_t K, const double* stim, double* lam, const double* y, const uint8_t* mask,
                              const double* mu, const double* beta, double shape, double rate, const double* phi, const double* phi_cov,
                              uint32_t* key, int32_t num_mc_samples, double minimum_spike_count, double minimax_spk_prob, int32_t it,
                              int32_t delay_spont_est) {
  if (!ctx || !stim || !lam || !y || !mask || !mu || !beta || !phi || !phi_cov || !key || num_mc_samples < 1) { cm_set_error("cm_k_lam_sweep: bad argument"); return CM_ERR_INVALID; }
  CM_HIP(hipSetDevice(ctx->device));
  cm_problem* p = nullptr;
  CM_TRY(problem_from_dense(ctx, N, K, stim, lam, y, &p));
  hipStream_t st = ctx->stream;
  const int S = num_mc_samples;
  double h[CM_S_COUNT] = {0};
  h[CM_S_SHAPE] = shape; h[CM_S_RATE] = rate;                       // spont_rate = 0: the threshold is minimax_spk_prob itself
  cm_options o;
  cm_options_default(&o);
  o.iters = 1; o.num_mc_samples = S; o.msrmp = minimax_spk_prob; o.minimum_spike_count = minimum_spike_count; o.delay_spont_est = delay_spont_est;
  std::vector<uint32_t>& keys = p->h_keys;
  keys.assign((size_t)(N + 3) * 2, 0u);
  uint32_t chain[2] = {key[0], key[1]};
  cm_host_key_schedule_range(chain, 0, 1, N, keys.data());
  uint32_t kout[2] = {key[0], key[1]};                              // the key update_lam returns: the carry after N splits
  for (int m = 0; m < N; ++m) { uint32_t a0 = 0, a1 = 2, b0 = 1, b1 = 3; cm_threefry2x32(kout[0], kout[1], a0, a1); cm_threefry2x32(kout[0], kout[1], b0, b1); kout[0] = a1; kout[1] = b1; }
  int rc = CM_OK;
  if ((rc = dmalloc(&p->keys, keys.size())) || (rc = dmalloc(&p->orders_all, (size_t)N)) || (rc = dmalloc(&p->orders_tmp, (size_t)N)) ||
      (rc = dmalloc(&p->inv_all, (size_t)N)) || (rc = dmalloc(&p->wtab, (size_t)N * 4 * S))) {
  } else if (hipMemcpy(p->keys, keys.data(), keys.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
             hipMemcpy(p->scal, h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess ||
             hipMemcpy(p->mu, mu, (size_t)N * 8, hipMemcpyHostToDevice) != hipSuccess ||
             hipMemcpy(p->beta, beta, (size_t)N * 8, hipMemcpyHostToDevice) != hipSuccess ||
             hipMemcpy(p->mask, mask, (size_t)K, hipMemcpyHostToDevice) != hipSuccess ||
             hipMemcpy(p->phi, phi, (size_t)N * 16, hipMemcpyHostToDevice) != hipSuccess ||
             hipMemcpy(p->phi_cov, phi_cov, (size_t)N * 32, hipMemcpyHostToDevice) != hipSuccess ||
             hipMemset(p->n_act, 0, 4 * sizeof(int)) != hipSuccess || hipMemset(p->seq_meta, 0, 8 * sizeof(int)) != hipSuccess) {
    cm_set_error("upload failed"); rc = CM_ERR_HIP;
  }
  p->keys_iters = 1; p->orders_cap = 1; p->wtab_cap = (size_t)N * 4 * S;
  cm_set set;
  if (!rc) rc = cm_launch_rng(ctx, p, 0, 1, st);
  if (!rc) rc = cm_launch_ndtri_table(ctx, p, 0, 1, S, CM_RNG_THREEFRY, st);
  if (!rc) rc = cm_set_single(ctx, p, &set);
  if (!rc) rc = cm_launch_entry_copies(ctx, set);              // per-entry y / mask from what the hook uploaded
  if (!rc) rc = cm_launch_iter_mid(ctx, set, -1, 1, S, CM_RNG_THREEFRY, &o, 0, 0, 2);   // per-neuron role only: MC logit from phi / phi_cov
  ctx->mb_n = N;
  if (!rc) rc = cm_launch_sweep(ctx, set, it, &o, nullptr, 0);
  double* d_dense = nullptr;
  if (!rc) rc = dmalloc(&d_dense, (size_t)N * K);
  if (!rc) rc = cm_launch_expand_lam(ctx, p, p->lam, d_dense);
  if (!rc && (hipMemcpyAsync(lam, d_dense, (size_t)N * K * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
              hipMemcpyAsync(ctx->h_pinned + 8, p->n_act, 16, hipMemcpyDeviceToHost, st) != hipSuccess ||
              hipStreamSynchronize(st) != hipSuccess)) { cm_set_error("download failed: %s", hipGetErrorString(hipGetLastError())); rc = CM_ERR_HIP; }
  if (!rc && ctx->h_pinned[10]) { cm_set_error("sweep ticket wait timed out (internal error)"); rc = CM_ERR_STATE; }
  if (d_dense) hipFree(d_dense);
  cm_problem_destroy(p);
  if (!rc) { key[0] = kout[0]; key[1] = kout[1]; }
  return rc;
}

extern "C" int cm_k_reconnect(cm_ctx* ctx, int32_t N, int32_t K, const double* stim, double* lam, double* mu, double* beta, double* z,
                              double minimax_spk_prob, double minimum_spike_count) {
  if (!ctx || !stim || !lam || !mu || !beta || !z) { cm_set_error("cm_k_reconnect: NULL argument"); return CM_ERR_INVALID; }
  CM_HIP(hipSetDevice(ctx->device));
  cm_problem* p = nullptr;
  CM_TRY(problem_from_dense(ctx, N, K, stim, lam, nullptr, &p));     // compressed rows + the dense lam scattered onto them
  hipStream_t st = ctx->stream;
  int rc = CM_OK;
  if (hipMemcpy(p->mu, mu, (size_t)N * 8, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(p->beta, beta, (size_t)N * 8, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(p->z, z, (size_t)K * 8, hipMemcpyHostToDevice) != hipSuccess) { cm_set_error("upload failed"); rc = CM_ERR_HIP; }
  cm_set set;
  if (!rc) rc = cm_set_single(ctx, p, &set);
  if (!rc) rc = cm_launch_fn_scan(ctx, set, minimax_spk_prob, minimum_spike_count);      // the device scan (cm_fnscan.hip), as a run launches it
  double* d_dense = nullptr;
  if (!rc) rc = dmalloc(&d_dense, (size_t)N * K);
  if (!rc) rc = cm_launch_expand_lam(ctx, p, p->lam, d_dense);
  if (!rc && (hipMemcpyAsync(lam, d_dense, (size_t)N * K * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
              hipMemcpyAsync(mu, p->mu, (size_t)N * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
              hipMemcpyAsync(beta, p->beta, (size_t)N * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
              hipMemcpyAsync(z, p->z, (size_t)K * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
              hipStreamSynchronize(st) != hipSuccess)) { cm_set_error("download failed: %s", hipGetErrorString(hipGetLastError())); rc = CM_ERR_HIP; }
  if (d_dense) hipFree(d_dense);
  cm_problem_destroy(p);
  return rc;
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

int cm_k_ndtri_impl(cm_ctx* ctx, int n, const double* d_p, double* d_out);
extern "C" int cm_k_ndtri(cm_ctx* ctx, int32_t n, const double* p, double* out) {
  if (!ctx || !p || !out || n < 1) { cm_set_error("cm_k_ndtri: bad argument"); return CM_ERR_INVALID; }
  CM_HIP(hipSetDevice(ctx->device));
  double *d_p = nullptr, *d_o = nullptr;
  int rc = CM_OK;
  if ((rc = dmalloc(&d_p, (size_t)n)) || (rc = dmalloc(&d_o, (size_t)n))) {
  } else if (hipMemcpyAsync(d_p, p, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { cm_set_error("cm_k_ndtri: upload failed"); rc = CM_ERR_HIP; }
  if (!rc) rc = cm_k_ndtri_impl(ctx, n, d_p, d_o);
  if (!rc && (hipMemcpyAsync(out, d_o, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)) {
    cm_set_error("cm_k_ndtri: download failed: %s", hipGetErrorString(hipGetLastError())); rc = CM_ERR_HIP;
  }
  hipStreamSynchronize(ctx->stream);
  if (d_p) hipFree(d_p);
  if (d_o) hipFree(d_o);
  return rc;
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
