// One-off prologue kernels: PSC integrals and the compressed encoding of the stimulus matrix.
// Replaces reference circuitmap/optimise/caviar.py:28-51 (prologue of caviar()).
#include <algorithm>

#include "cm_internal.h"

// ------------------------------------------------------------------------------------------------
// y[k] = trapz(psc[k,:]) (unit spacing), ss[k] = sum_t psc[k,t]^2        caviar.py:28,30
// One wavefront per trial row; lanes stride the row so every wave-load is 512 contiguous bytes.
// HBM-bound: reads K*T*8 bytes once.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void psc_prologue_body(const double* __restrict__ psc, int K, int T,
                                                       double* __restrict__ y, double* __restrict__ ss) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / CM_WAVE;
  const int lane = threadIdx.x % CM_WAVE;
  if (wave >= K) return;
  const double* row = psc + (size_t)wave * T;
  double tz = 0.0, sq = 0.0;
  for (int t = lane; t < T; t += CM_WAVE) {
    const double x = row[t];
    sq = fma(x, x, sq);
    if (t + 1 < T) tz += (x + row[t + 1]) * 0.5;   // np.trapz: sum_i (x[i] + x[i+1]) / 2
  }
  tz = cm_wave_sum(tz);
  sq = cm_wave_sum(sq);
  if (lane == 0) { y[wave] = tz; ss[wave] = sq; }
}

__device__ __forceinline__ void sum_y_body(const double* __restrict__ y, int K, double* __restrict__ out) {
  __shared__ double s0[16], s1[16];
  double a = 0.0, b = 0.0;
  for (int k = threadIdx.x; k < K; k += blockDim.x) { a += y[k]; b = fma(y[k], y[k], b); }
  a = cm_wave_sum(a); b = cm_wave_sum(b);
  if (threadIdx.x % CM_WAVE == 0) { s0[threadIdx.x / CM_WAVE] = a; s1[threadIdx.x / CM_WAVE] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ta = 0.0, tb = 0.0;
    for (int w = 0; w < (int)(blockDim.x / CM_WAVE); ++w) { ta += s0[w]; tb += s1[w]; }
    out[0] = ta; out[1] = tb;
  }
}

__global__ __launch_bounds__(256) void k_psc_prologue(const double* __restrict__ psc, int K, int T, double* __restrict__ y, double* __restrict__ ss) {
  psc_prologue_body(psc, K, T, y, ss);
}
int cm_launch_psc_prologue(cm_ctx* ctx, int K, int T, const double* d_psc, double* d_y, double* d_ss) {
  const int waves_per_block = 256 / CM_WAVE;
  const int blocks = (K + waves_per_block - 1) / waves_per_block;
  hipLaunchKernelGGL(k_psc_prologue, dim3(blocks), dim3(256), 0, ctx->stream, d_psc, K, T, d_y, d_ss);
  CM_HIP(hipGetLastError());
  return CM_OK;
}

// ------------------------------------------------------------------------------------------------
// Distinct values of the stimulus matrix (np.unique(I), caviar.py:42).  Append-only set in global
// memory claimed by 64-bit CAS; each workgroup filters through an LDS copy so that the >99 % zero
// entries never touch the global table.
// ------------------------------------------------------------------------------------------------
#define CM_EMPTY 0xFFFFFFFFFFFFFFFFull

__device__ __forceinline__ bool cm_set_insert(unsigned long long* table, int cap, unsigned long long bits, bool* is_new) {
  for (int s = 0; s < cap; ++s) {
    unsigned long long cur = *(volatile unsigned long long*)&table[s];   // plain read first: same-address reads broadcast
    if (cur == bits) { *is_new = false; return true; }
    if (cur == CM_EMPTY) {
      cur = atomicCAS(&table[s], CM_EMPTY, bits);
      if (cur == CM_EMPTY) { *is_new = true; return true; }
      if (cur == bits) { *is_new = false; return true; }
    }
  }
  return false;
}

__device__ __forceinline__ void stim_distinct_body(const double* __restrict__ stim, size_t n,
                                                        unsigned long long* __restrict__ table, int* __restrict__ overflow) {
  __shared__ unsigned long long cache[CM_MAX_VALS];
  for (int i = threadIdx.x; i < CM_MAX_VALS; i += blockDim.x) cache[i] = CM_EMPTY;
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    double v = stim[i] + 0.0;   // -0.0 -> +0.0 (np.unique compares by value)
    unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    bool is_new = false;
    if (!cm_set_insert(cache, CM_MAX_VALS, bits, &is_new)) { *overflow = 1; continue; }
    if (is_new) {
      bool g_new;
      if (!cm_set_insert(table, CM_MAX_VALS, bits, &g_new)) *overflow = 1;
    }
  }
}

__device__ __forceinline__ void vals_finalize_body(const unsigned long long* __restrict__ table, double* __restrict__ vals, int* __restrict__ nvals) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int n = 0;
  for (int s = 0; s < CM_MAX_VALS; ++s) {
    if (table[s] == CM_EMPTY) continue;
    double v = __longlong_as_double((long long)table[s]);
    int j = n++;
    while (j > 0 && vals[j - 1] > v) { vals[j] = vals[j - 1]; --j; }
    vals[j] = v;
  }
  for (int s = n; s < CM_MAX_VALS; ++s) vals[s] = 0.0;
  *nvals = n;
}

__device__ __forceinline__ int cm_val_index(const double* vals, int V, double v) {
  for (int g = 0; g < V; ++g) if (vals[g] == v) return g;
  return 0;
}

// per-row counts of each distinct value and of targeted (stim > 0) trials; one wave per row
__device__ __forceinline__ void stim_rowcount_body(const double* __restrict__ stim, int N, int K, const double* __restrict__ vals_g,
                                                        const int* __restrict__ nvals, int* __restrict__ rowcnt, int* __restrict__ row_nnz) {
  __shared__ double vals[CM_MAX_VALS];
  const int V = *nvals;
  if (threadIdx.x < CM_MAX_VALS) vals[threadIdx.x] = vals_g[threadIdx.x];
  __syncthreads();
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / CM_WAVE;
  const int lane = threadIdx.x % CM_WAVE;
  if (wave >= N) return;
  const double* row = stim + (size_t)wave * K;
  int nz = 0;
  for (int g0 = 0; g0 < V; g0 += 8) {     // count 8 groups per pass over the (cache-resident) row
    int c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = lane; k < K; k += CM_WAVE) {
      const double v = row[k];
      if (g0 == 0) nz += (v > 0.0);
#pragma unroll
      for (int j = 0; j < 8; ++j) c[j] += (g0 + j < V && v == vals[g0 + j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int t = cm_wave_sum_i(c[j]);
      if (lane == 0 && g0 + j < V) rowcnt[(size_t)wave * CM_MAX_VALS + g0 + j] = t;
    }
  }
  nz = cm_wave_sum_i(nz);
  if (lane == 0) row_nnz[wave] = nz;
}

// exclusive scan of n counts into ptr[0..n]; single workgroup
__device__ __forceinline__ void exclusive_scan_body(const int* __restrict__ cnt, int n, int* __restrict__ ptr) {
  __shared__ int part[1024];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    int i = base + threadIdx.x;
    int v = i < n ? cnt[i] : 0;
    part[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      int t = (int)threadIdx.x >= o ? part[threadIdx.x - o] : 0;
      __syncthreads();
      part[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < n) ptr[i] = carry + part[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += part[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) ptr[n] = carry;
}

// CSR fill, one wave per row, ordered compaction by ballot prefix
__device__ __forceinline__ void stim_fill_csr_body(const double* __restrict__ stim, int N, int K, const double* __restrict__ vals_g,
                                                        const int* __restrict__ nvals, const int* __restrict__ row_ptr,
                                                        int* __restrict__ col, uint8_t* __restrict__ grp) {
  __shared__ double vals[CM_MAX_VALS];
  const int V = *nvals;
  if (threadIdx.x < CM_MAX_VALS) vals[threadIdx.x] = vals_g[threadIdx.x];
  __syncthreads();
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / CM_WAVE;
  const int lane = threadIdx.x % CM_WAVE;
  if (wave >= N) return;
  const double* row = stim + (size_t)wave * K;
  int base = row_ptr[wave];
  for (int k0 = 0; k0 < K; k0 += CM_WAVE) {
    const int k = k0 + lane;
    const double v = k < K ? row[k] : 0.0;
    const bool t = v > 0.0;
    const unsigned long long m = __ballot(t);
    if (t) {
      const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
      col[pos] = k;
      grp[pos] = (uint8_t)cm_val_index(vals, V, v + 0.0);
    }
    base += __popcll(m);
  }
}

// CSC from the CSR entries (nnz-sized work; the dense matrix is not read again): count the entries of every trial column,
// scan, scatter (neuron, CSR position) pairs through per-column cursors, then order each column by neuron so that the
// summation order downstream is deterministic (the cursor order is not).
__device__ __forceinline__ void csc_count_body(const int* __restrict__ row_ptr, const int* __restrict__ col, int N, int* __restrict__ col_cnt) {
  const int n = (blockIdx.x * blockDim.x + threadIdx.x) / CM_WAVE, lane = threadIdx.x % CM_WAVE;
  if (n >= N) return;
  for (int e = row_ptr[n] + lane; e < row_ptr[n + 1]; e += CM_WAVE) atomicAdd(&col_cnt[col[e]], 1);
}
__device__ __forceinline__ void csc_scatter_body(const int* __restrict__ row_ptr, const int* __restrict__ col, int N,
                                                      const int* __restrict__ col_ptr, int* __restrict__ cursor,
                                                      int* __restrict__ csc_row, int* __restrict__ csc_pos) {
  const int n = (blockIdx.x * blockDim.x + threadIdx.x) / CM_WAVE, lane = threadIdx.x % CM_WAVE;
  if (n >= N) return;
  for (int e = row_ptr[n] + lane; e < row_ptr[n + 1]; e += CM_WAVE) {
    const int k = col[e];
    const int w = col_ptr[k] + atomicAdd(&cursor[k], 1);
    csc_row[w] = n;
    csc_pos[w] = e;
  }
}
// power code of every CSC entry (csc_grp[ce] = grp[csc_pos[ce]]): lets a walk over a trial's column read the codes without the dependent gather
__device__ __forceinline__ void csc_grp_body(const int* __restrict__ csc_pos, const uint8_t* __restrict__ grp, long long nnz, uint8_t* __restrict__ csc_grp) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nnz) csc_grp[i] = grp[csc_pos[i]];
}
__device__ __forceinline__ void csc_sort_body(const int* __restrict__ col_ptr, int K, int* __restrict__ csc_row, int* __restrict__ csc_pos) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  const int b = col_ptr[k], e = col_ptr[k + 1];
  for (int i = b + 1; i < e; ++i) {                     // insertion sort by neuron (a neuron appears once per column)
    const int r = csc_row[i], q = csc_pos[i];
    int j = i - 1;
    while (j >= b && csc_row[j] > r) { csc_row[j + 1] = csc_row[j]; csc_pos[j + 1] = csc_pos[j]; --j; }
    csc_row[j + 1] = r; csc_pos[j + 1] = q;
  }
}

// longest row / column of a CSR / CSC pointer array -> *out (atomicMax; out zeroed by the caller)
__device__ __forceinline__ void max_len_body(const int* __restrict__ ptr, int n, int* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int v = i < n ? ptr[i + 1] - ptr[i] : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, CM_WAVE));
  if (threadIdx.x % CM_WAVE == 0 && v > 0) atomicMax(out, v);
}

// ------------------------------------------------------------------------------------------------
// The encoding of a GROUP of circuits in two stages around ONE host wait: every kernel above is launched once for the group (blockIdx.y
// = circuit, descriptor cm_enc), where one launch sequence per circuit cost ~25 host API calls and two waits each -- creating the 256
// circuits of a batch took as long as fitting them.  Stage 1 (everything whose size N and K give): PSC integrals, distinct values, per-row
// value counts, row pointers, longest row -> the host reads (too many values?, V, longest row, nnz) of all circuits with one copy.  Stage 2
// (sized by nnz, in the second slab): CSR entries, CSC, power codes of the CSC entries, longest column, cleared state.
// ------------------------------------------------------------------------------------------------
#define CM_ENC(K) const cm_enc& e = arr[blockIdx.y];
__global__ __launch_bounds__(256) void k_enc_clear1(const cm_enc* __restrict__ arr) {
  CM_ENC()
  const size_t n_rc = (size_t)e.N * CM_MAX_VALS;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rc; i += (size_t)gridDim.x * blockDim.x) e.rowcnt[i] = 0;
  if (blockIdx.x == 0) {
    if (threadIdx.x < CM_MAX_VALS) e.table[threadIdx.x] = CM_EMPTY;
    if (threadIdx.x < 8) e.flags[threadIdx.x] = 0;
  }
}
__global__ __launch_bounds__(256) void kb_psc_prologue(const cm_enc* __restrict__ arr) {
  CM_ENC()
  psc_prologue_body(e.psc, e.K, e.T, e.y, e.ss);
}
__global__ __launch_bounds__(256) void kb_stim_distinct(const cm_enc* __restrict__ arr) {
  CM_ENC()
  stim_distinct_body(e.stim, (size_t)e.N * e.K, e.table, e.flags);
}
__global__ void kb_vals_finalize(const cm_enc* __restrict__ arr) {
  CM_ENC()
  vals_finalize_body(e.table, e.vals, e.flags + 1);
}
__global__ __launch_bounds__(256) void kb_stim_rowcount(const cm_enc* __restrict__ arr) {
  CM_ENC()
  stim_rowcount_body(e.stim, e.N, e.K, e.vals, e.flags + 1, e.rowcnt, e.row_nnz);
}
__global__ __launch_bounds__(1024) void kb_scan_rows(const cm_enc* __restrict__ arr) {
  CM_ENC()
  exclusive_scan_body(e.row_nnz, e.N, e.row_ptr);
}
__global__ __launch_bounds__(256) void kb_max_row(const cm_enc* __restrict__ arr) {
  CM_ENC()
  max_len_body(e.row_ptr, e.N, e.flags + 2);
}
// (too many values, V, longest row, longest column) and nnz of every circuit, side by side for one copy to the host
__global__ __launch_bounds__(64) void k_enc_report(const cm_enc* __restrict__ arr, int n, int* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  const cm_enc& e = arr[c];
  for (int j = 0; j < 4; ++j) out[8 * c + j] = e.flags[j];
  out[8 * c + 4] = e.row_ptr[e.N];
}
__global__ __launch_bounds__(256) void k_enc_clear2(const cm_enc* __restrict__ arr) {
  CM_ENC()
  const size_t n_mc = (size_t)e.N * CM_MAX_VALS;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_mc; i += (size_t)gridDim.x * blockDim.x) e.mcE[i] = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)e.N + 1; i += (size_t)gridDim.x * blockDim.x) e.alive[i] = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)e.K; i += (size_t)gridDim.x * blockDim.x) { e.col_cnt[i] = 0; e.cursor[i] = 0; }
}
__global__ __launch_bounds__(256) void kb_stim_fill_csr(const cm_enc* __restrict__ arr) {
  CM_ENC()
  stim_fill_csr_body(e.stim, e.N, e.K, e.vals, e.flags + 1, e.row_ptr, e.col, e.grp);
}
__global__ __launch_bounds__(256) void kb_csc_count(const cm_enc* __restrict__ arr) {
  CM_ENC()
  csc_count_body(e.row_ptr, e.col, e.N, e.col_cnt);
}
__global__ __launch_bounds__(1024) void kb_scan_cols(const cm_enc* __restrict__ arr) {
  CM_ENC()
  exclusive_scan_body(e.col_cnt, e.K, e.col_ptr);
}
__global__ __launch_bounds__(256) void kb_csc_scatter(const cm_enc* __restrict__ arr) {
  CM_ENC()
  csc_scatter_body(e.row_ptr, e.col, e.N, e.col_ptr, e.cursor, e.csc_row, e.csc_pos);
}
__global__ __launch_bounds__(256) void kb_csc_sort(const cm_enc* __restrict__ arr) {
  CM_ENC()
  csc_sort_body(e.col_ptr, e.K, e.csc_row, e.csc_pos);
}
__global__ __launch_bounds__(256) void kb_csc_grp(const cm_enc* __restrict__ arr) {
  CM_ENC()
  csc_grp_body(e.csc_pos, e.grp, e.nnz, e.csc_grp);
}
__global__ __launch_bounds__(256) void kb_max_col(const cm_enc* __restrict__ arr) {
  CM_ENC()
  max_len_body(e.col_ptr, e.K, e.flags + 3);
}

int cm_encode_stage1(cm_ctx* ctx, const cm_enc* d_arr, int n, int N_max, int K_max, int* d_report) {
  hipStream_t st = ctx->stream;
  const dim3 b256(256);
  const size_t total = (size_t)N_max * K_max;
  int blocks = (int)((total + 255) / 256);
  const int cap = std::max(1, ctx->num_cus * 8 / std::max(1, std::min(n, 8)));      // grid-stride loops: enough workgroups to fill the chip, not one per 256 elements
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(k_enc_clear1, dim3(std::max(1, (N_max * CM_MAX_VALS + 255) / 256), n), b256, 0, st, d_arr);
  hipLaunchKernelGGL(kb_psc_prologue, dim3((K_max + 3) / 4, n), b256, 0, st, d_arr);
  hipLaunchKernelGGL(kb_stim_distinct, dim3(blocks, n), b256, 0, st, d_arr);
  hipLaunchKernelGGL(kb_vals_finalize, dim3(1, n), dim3(64), 0, st, d_arr);
  hipLaunchKernelGGL(kb_stim_rowcount, dim3((N_max + 3) / 4, n), b256, 0, st, d_arr);
  hipLaunchKernelGGL(kb_scan_rows, dim3(1, n), dim3(1024), 0, st, d_arr);
  hipLaunchKernelGGL(kb_max_row, dim3((N_max + 255) / 256, n), b256, 0, st, d_arr);
  hipLaunchKernelGGL(k_enc_report, dim3((n + 63) / 64), dim3(64), 0, st, d_arr, n, d_report);
  CM_HIP(hipGetLastError());
  return CM_OK;
}

int cm_encode_stage2(cm_ctx* ctx, const cm_enc* d_arr, int n, int N_max, int K_max, long long nnz_max, int* d_report) {
  hipStream_t st = ctx->stream;
  const dim3 b256(256);
  const int row_blocks = (N_max + 3) / 4;
  hipLaunchKernelGGL(k_enc_clear2, dim3(std::max(1, (N_max * CM_MAX_VALS + 255) / 256), n), b256, 0, st, d_arr);
  hipLaunchKernelGGL(kb_stim_fill_csr, dim3(row_blocks, n), b256, 0, st, d_arr);
  hipLaunchKernelGGL(kb_csc_count, dim3(row_blocks, n), b256, 0, st, d_arr);
  hipLaunchKernelGGL(kb_scan_cols, dim3(1, n), dim3(1024), 0, st, d_arr);
  hipLaunchKernelGGL(kb_csc_scatter, dim3(row_blocks, n), b256, 0, st, d_arr);
  hipLaunchKernelGGL(kb_csc_sort, dim3((K_max + 255) / 256, n), b256, 0, st, d_arr);
  hipLaunchKernelGGL(kb_csc_grp, dim3((unsigned)((std::max<long long>(nnz_max, 1) + 255) / 256), n), b256, 0, st, d_arr);
  hipLaunchKernelGGL(kb_max_col, dim3((K_max + 255) / 256, n), b256, 0, st, d_arr);
  hipLaunchKernelGGL(k_enc_report, dim3((n + 63) / 64), dim3(64), 0, st, d_arr, n, d_report);
  CM_HIP(hipGetLastError());
  return CM_OK;
}

// host mirror of the compressed rows: only the host-side false-negative scan (verbose runs, shapes the scan kernel does not take) and the
// dense expansion of a fetched lam need it -- copied on first use
int cm_host_mirror(cm_ctx* ctx, cm_problem* p) {
  if (p->have_mirror) return CM_OK;
  const int N = p->N, K = p->K;
  const size_t nnz_alloc = p->nnz > 0 ? (size_t)p->nnz : 1;
  hipStream_t st = ctx->stream;
  p->h_row_ptr.resize(N + 1); p->h_col.resize(nnz_alloc); p->h_grp.resize(nnz_alloc); p->h_vals.resize(CM_MAX_VALS);
  p->h_col_ptr.resize(K + 1); p->h_csc_row.resize(nnz_alloc);
  CM_HIP(hipMemcpyAsync(p->h_row_ptr.data(), p->row_ptr, (size_t)(N + 1) * sizeof(int), hipMemcpyDeviceToHost, st));
  CM_HIP(hipMemcpyAsync(p->h_col.data(), p->col, (size_t)p->nnz * sizeof(int), hipMemcpyDeviceToHost, st));
  CM_HIP(hipMemcpyAsync(p->h_grp.data(), p->grp, (size_t)p->nnz, hipMemcpyDeviceToHost, st));
  CM_HIP(hipMemcpyAsync(p->h_vals.data(), p->vals, CM_MAX_VALS * sizeof(double), hipMemcpyDeviceToHost, st));
  CM_HIP(hipMemcpyAsync(p->h_col_ptr.data(), p->col_ptr, (size_t)(K + 1) * sizeof(int), hipMemcpyDeviceToHost, st));
  CM_HIP(hipMemcpyAsync(p->h_csc_row.data(), p->csc_row, (size_t)p->nnz * sizeof(int), hipMemcpyDeviceToHost, st));
  CM_HIP(hipStreamSynchronize(st));
  p->have_mirror = true;
  return CM_OK;
}

// ------------------------------------------------------------------------------------------------
// state init, one launch for everything that is elementwise: lam_mask = ss > thresh; lam0 = 0.95 * [I > 0] * mask
// (caviar.py:30-34); z = 0 (caviar.py:50); mu, beta, phi, phi_cov from the uploaded priors (caviar.py:44-49); counters zeroed.
// sum(y), sum(y^2) keep their own single-workgroup launch (fixed summation order).
// ------------------------------------------------------------------------------------------------
template <bool BATCH>
__global__ __launch_bounds__(256) void k_init_state(cm_parg<BATCH> pa, double thresh) {
  const cm_pdev& p = pa.get();
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int N = p.N, K = p.K;
  if (i < K) { p.mask[i] = p.ss[i] > thresh; p.z[i] = 0.0; }
  if (i < p.nnz) {
    const int c = p.col[i];
    const bool on = p.ss[c] > thresh;
    p.lam[i] = 0.95 * (double)on;
    const_cast<uint8_t*>(p.me)[i] = (uint8_t)on;             // per-entry copies of mask and y for the row kernels
    const_cast<double*>(p.ye)[i] = p.y[c];
  }
  const double* __restrict__ pslab = p.pslab;              // mu_prior | beta_prior | phi_prior | phi_cov_prior | scalars
  if (i < N) { p.mu[i] = pslab[i]; p.beta[i] = pslab[N + i]; }
  if (i < 2 * (long long)N) p.phi[i] = pslab[2 * (size_t)N + i];
  if (i < 4 * (long long)N) p.phi_cov[i] = pslab[4 * (size_t)N + i];
  if (i < 4) p.n_act[i] = 0;
  if (i < 8) { p.seq_meta[i] = 0; p.tcnt[i] = 0; }
  if (i < 2 * (long long)K) p.tstamp[i] = -1;                 // (stamps are iteration numbers: those of an earlier run must not be taken for this one's)
  if (i == 0) p.alive[0] = 0;
}
template <bool BATCH>
__global__ __launch_bounds__(1024) void k_sum_y_set(cm_parg<BATCH> pa) {
  const cm_pdev& p = pa.get();
  sum_y_body(p.y, p.K, const_cast<double*>(p.sum_y));
}

int cm_launch_sum_y(cm_ctx* ctx, cm_set& s) {
  CM_LAUNCH_SET(k_sum_y_set, s, 1, 1024, 0, ctx->stream);
  CM_HIP(hipGetLastError());
  return CM_OK;
}

template <bool BATCH>
__global__ __launch_bounds__(256) void k_entry_copies(cm_parg<BATCH> pa) {
  const cm_pdev& p = pa.get();
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < p.nnz) { const int c = p.col[i]; const_cast<uint8_t*>(p.me)[i] = p.mask[c]; const_cast<double*>(p.ye)[i] = p.y[c]; }
}
static int launch_mask_counts(cm_ctx* ctx, cm_set& s);
int cm_launch_entry_copies(cm_ctx* ctx, cm_set& s) {
  CM_LAUNCH_SET(k_entry_copies, s, (unsigned)((std::max<long long>(s.nnz_max, 1) + 255) / 256), 256, 0, ctx->stream);
  CM_HIP(hipGetLastError());
  CM_TRY(launch_mask_counts(ctx, s));
  return CM_OK;
}

// unmasked trials of every row at each stimulus value (one wavefront per row), and the "revived" flags cleared: the closed-form gate of an
// all-zero row (cm_sweep.hip: quiescent_gate) needs the counts; they change only with the mask, i.e. once per run
template <bool BATCH>
__global__ __launch_bounds__(256) void k_mask_counts(cm_parg<BATCH> pa) {
  const cm_pdev& p = pa.get();
  const int n = (blockIdx.x * blockDim.x + threadIdx.x) / CM_WAVE, lane = threadIdx.x % CM_WAVE;
  if (n >= p.N) return;
  int* __restrict__ out = const_cast<int*>(p.mcnt) + (size_t)n * CM_MAX_VALS;
  int mine = 0;                                             // lane g: unmasked entries of value g
  for (int e0 = p.row_ptr[n]; e0 < p.row_ptr[n + 1]; e0 += CM_WAVE) {
    const int e = e0 + lane;
    const bool in = e < p.row_ptr[n + 1];
    const int g = in ? p.grp[e] : -1;
    const bool on = in && p.me[e] != 0;
    for (int v = 0; v < p.V; ++v) {
      const int c = __popcll(__ballot(on && g == v));
      if (lane == v) mine += c;
    }
  }
  if (lane < CM_MAX_VALS) out[lane] = lane < p.V ? mine : 0;
  if (lane >= 1 && lane <= 8) const_cast<int*>(p.qcnt)[(size_t)n * 8 + lane - 1] = lane < p.V ? ((mine << 16) | (p.rowcnt[(size_t)n * CM_MAX_VALS + lane] & 0xFFFF)) : 0;
  if (lane == 0) p.touch[n] = 0;
}
static int launch_mask_counts(cm_ctx* ctx, cm_set& s) {
  CM_LAUNCH_SET(k_mask_counts, s, (unsigned)((s.lead->N + 3) / 4), 256, 0, ctx->stream);
  CM_HIP(hipGetLastError());
  return CM_OK;
}

int cm_launch_init_state(cm_ctx* ctx, cm_set& s, double y_xcorr_thresh) {
  hipStream_t st = ctx->stream;
  const cm_problem* p = s.lead;
  const long long n = std::max<long long>(std::max<long long>(4LL * p->N, 2LL * p->K), std::max<long long>(s.nnz_max, 8));
  CM_LAUNCH_SET(k_init_state, s, (unsigned)((n + 255) / 256), 256, 0, st, y_xcorr_thresh);
  CM_LAUNCH_SET(k_sum_y_set, s, 1, 1024, 0, st);
  CM_HIP(hipGetLastError());
  return launch_mask_counts(ctx, s);
}

// dense <-> compressed lam (targeted entries only; everything else is identically zero, caviar.py:216)
__global__ __launch_bounds__(256) void k_scatter_dense(const double* __restrict__ dense, int N, int K, const int* __restrict__ row_ptr,
                                                        const int* __restrict__ col, double* __restrict__ lam) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / CM_WAVE;
  const int lane = threadIdx.x % CM_WAVE;
  if (wave >= N) return;
  for (int e = row_ptr[wave] + lane; e < row_ptr[wave + 1]; e += CM_WAVE) lam[e] = dense[(size_t)wave * K + col[e]];
}
__global__ __launch_bounds__(256) void k_expand_dense(const double* __restrict__ lam, int N, int K, const int* __restrict__ row_ptr,
                                                       const int* __restrict__ col, double* __restrict__ dense) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / CM_WAVE;
  const int lane = threadIdx.x % CM_WAVE;
  if (wave >= N) return;
  for (int e = row_ptr[wave] + lane; e < row_ptr[wave + 1]; e += CM_WAVE) dense[(size_t)wave * K + col[e]] = lam[e];
}

int cm_launch_scatter_dense_lam(cm_ctx* ctx, cm_problem* p, const double* d_lam_dense) {
  hipLaunchKernelGGL(k_scatter_dense, dim3((p->N + 3) / 4), dim3(256), 0, ctx->stream, d_lam_dense, p->N, p->K, p->row_ptr, p->col, p->lam);
  CM_HIP(hipGetLastError());
  return CM_OK;
}
int cm_launch_expand_lam(cm_ctx* ctx, cm_problem* p, const double* d_lam_sparse, double* d_dense) {
  CM_HIP(hipMemsetAsync(d_dense, 0, (size_t)p->N * p->K * sizeof(double), ctx->stream));
  hipLaunchKernelGGL(k_expand_dense, dim3((p->N + 3) / 4), dim3(256), 0, ctx->stream, d_lam_sparse, p->N, p->K, p->row_ptr, p->col, d_dense);
  CM_HIP(hipGetLastError());
  return CM_OK;
}
