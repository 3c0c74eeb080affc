// Weight posterior (mu, beta): replaces block_update_mu, reference circuitmap/optimise/caviar.py:166-172.
//
//   Sigma^-1 = sigma * (D + L) + diag(1/beta0^2),  L = lam lam^T,  D = diag(sum_k lam(1-lam)),  sigma = shape/rate
//   mu = Sigma (sigma * lam y + mu0/beta0^2),  beta = diag(Sigma)          (the reference's "beta" is a variance)
//
// MI355X design: rows with lam == 0 decouple exactly (mu = mu0, beta = beta0^2), so the dense fp64 system is built
// only over the ACTIVE set (compacted, ascending neuron order).  L is assembled from the compressed stimulus
// encoding (each trial column couples only the neurons targeted together), then factored by a right-looking
// blocked Cholesky on 32x32 tiles; L^-1 comes from recursive doubling of triangular blocks (GEMM-shaped),
// and mu, diag(Sigma) are column reductions of L^-1.  Everything fp64: the early iterations are
// ill-conditioned enough (SURVEY 8c / tests) that fp32 would not hold the 1e-4 parity target.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "cm_bodies.h"

#define TS CM_TILE

// Every kernel of the tiled solve serves one circuit (descriptor by value, active-set size `na_arg` known to the host) or a lock-step
// batch (blockIdx.y = circuit): then each circuit reads its own active-set size from HBM, derives its own tile count and leading
// dimension, and leaves at once if the tiled solve is not the solver of that size (cm_solve_kind) or if the block lies outside
// its (smaller) matrix -- the grid is sized for the largest matrix of the batch.  (The size arrives through a vector load; readfirstlane
// tells the compiler it is uniform, so the tile addresses derived from it are scalar.)
#define CM_SOLVE_PROLOGUE(KIND) CM_SOLVE_PROLOGUE_AT(KIND, pa.get())
#define CM_SOLVE_PROLOGUE_AT(KIND, PDEV)                                          \
  const cm_pdev& p = PDEV;                                                        \
  const int na = BATCH ? __builtin_amdgcn_readfirstlane(p.n_act[0]) : na_arg;     \
  if (BATCH && cm_solve_kind(na, p.small_cap, (skip_own & 1) != 0) != (KIND)) return;   \
  const int nblk = (na + TS - 1) / TS, nrow = nblk * TS, ld = nrow + CM_LD_PAD;  \
  (void)nblk; (void)nrow; (void)ld;
// Leading dimension of the solve matrices = rows + CM_LD_PAD: a power-of-two row stride (1024 doubles at N = 1000) maps the 32 rows
// of a tile -- and the same tile of every circuit of a batch -- onto a fraction of the memory channels.
#define CM_LARGE_TILES 48     // more tile rows than this: panel solve + lean trailing-update kernels (bandwidth-bound regime)

#define CM_LINV_THREADS 1024  // threads that run the diagonal-tile elimination (256 was measured slower: 4 elements per thread)
#define TP (CM_TILE + 2)   // padded LDS row (doubles): even, so that a thread's pair of neighbours moves with one 16-byte LDS store.  The matrix-core operand reads
                          // are tile[row = lane & 15][k = lane >> 4] (ds_read_b64, banked over the 32 lanes of a half-wave on 64 dword banks): a pitch of
                          // 68 dwords puts the 32 (row, k) pairs on 32 distinct bank pairs, where 32 + 1 made (row, 1) collide with (row + 1, 0)
#define TPN (CM_TILE + 2)  // pitch of a tile read as B[k = lane >> 4][col = lane & 15] (blk_mfma_nn).  The conflict-free pitch for that read is 48 (96 dwords = 32 mod 64
                           // keeps k = 0 / 1 on disjoint banks); it was measured no faster than this one (the 2-way conflict is hidden) and its 12 KB cost the
                           // inverse kernels a workgroup per CU: 1.212 ms at 48, 1.195 ms at 34 for 128 x n = 1000

// ------------------------------------------------------------------------------------------------
// precision matrix over the active set; one wave per (padded) row, accumulator row in LDS.
//
// Row a (neuron n) receives, for every trial e of n with lam[e] != 0 and every neuron m targeted in the same trial, lam[e] * lam[m, trial]
// at column act_inv[m] -- accumulated in the fixed order of the row's trials (the summation order the results are pinned to).  The
// operands sit behind a four-deep chain of dependent gathers (col -> col_ptr -> csc_row / csc_pos -> act_inv / lam); walked trial by
// trial that is ~20 x 3 memory round trips per row.  So the wave first GATHERS: one lane per trial fetches (lam, col, column range), a
// wave scan lays the column mates of all trials out as one item list, and the lanes fetch the items' (active index, lam) side by side
// (four per lane in flight) into LDS; then it ACCUMULATES from LDS in the original order -- operation for operation the sum of the
// trial-by-trial walk, with the memory latency paid once per level instead of once per trial and level.
// ------------------------------------------------------------------------------------------------
#define CM_BP_ITEMS 256           // staged items per gather round = one round of 4 per lane (a trial with more targeted neurons than this is walked
                                  // directly); small on purpose: the kernel is latency-bound and LDS is what limits its wavefronts per CU
__host__ __device__ static inline size_t build_prec_lds(int ld) { return (size_t)ld * sizeof(double) + CM_BP_ITEMS * (sizeof(double) + sizeof(int)); }

// Per-solve staging for the build (k_bp_prepare, one thread per entry / active row), laid out in the circuit's `packed` buffer -- the record
// stream of the Gauss-Seidel chain, dead between two sweeps: [N] BpRow | [nnz] BpEnt (CSR order) | [nnz] BpMate (CSC order).  With them a row's
// wavefront reaches its operands in THREE dependent round trips (row record -> its trials' records -> their mates' records) instead of six
// (act_idx -> row_ptr -> lam / col -> col_ptr -> csc_row / csc_pos -> act_inv / lam): the kernel is a chain of dependent gathers, its time
// is its wavefronts times the length of that chain over its occupancy.
struct __align__(16) BpRow { int e_beg, e_end, n, pad; };
struct __align__(16) BpEnt { double le; int cs, cl; };          // lam of the entry; CSC range of its trial (cl = 0 when lam == 0: contributes nothing)
struct __align__(16) BpMate { double l; int b, pad; };          // lam of the column mate, its active index (-1: not in the active set)
__device__ __forceinline__ BpRow* bp_rows(const cm_pdev& p) { return reinterpret_cast<BpRow*>(p.packed); }
__device__ __forceinline__ BpEnt* bp_ents(const cm_pdev& p) { return reinterpret_cast<BpEnt*>(reinterpret_cast<char*>(p.packed) + (size_t)p.N * 16); }
__device__ __forceinline__ BpMate* bp_mates(const cm_pdev& p) { return reinterpret_cast<BpMate*>(reinterpret_cast<char*>(p.packed) + (size_t)p.N * 16 + (size_t)p.nnz * 16); }

template <bool BATCH>
__global__ __launch_bounds__(256) void k_bp_prepare(cm_parg<BATCH> pa, int na_arg, int skip_own) {
  CM_SOLVE_PROLOGUE(2)
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < na) {
    const int n = p.act_idx[j];
    BpRow r; r.e_beg = p.row_ptr[n]; r.e_end = p.row_ptr[n + 1]; r.n = n; r.pad = 0;
    bp_rows(p)[j] = r;
  }
  if (j < p.nnz) {
    const double le = p.lam[j];
    const int k = p.col[j];
    BpEnt e; e.le = le; e.cs = 0; e.cl = 0;
    if (le != 0.0) { e.cs = p.col_ptr[k]; e.cl = p.col_ptr[k + 1] - e.cs; }
    bp_ents(p)[j] = e;
    BpMate m; m.b = p.act_inv[p.csc_row[j]]; m.l = p.lam[p.csc_pos[j]]; m.pad = 0;
    bp_mates(p)[j] = m;
  }
}

template <bool BATCH>
__global__ __launch_bounds__(64) void k_build_precision(cm_parg<BATCH> pa, int na_arg, int skip_own, int row0, int acc_len) {
  int bx = (int)blockIdx.x, circ = 0;
  if (BATCH) cm_xcd_remap(bx, circ, 2);              // a circuit's rows on one XCD, circuit after circuit: its staged records (0.7 MB) stay in that L2
  CM_SOLVE_PROLOGUE_AT(2, pa.get(circ))
  const int n_act = na;
  if (row0 + bx >= nrow) return;
  const double* __restrict__ rs = p.rs;
  const double* __restrict__ scal = p.scal;
  const double* __restrict__ mu_prior = p.mu_prior;
  const double* __restrict__ beta_prior = p.beta_prior;
  const BpEnt* __restrict__ ents = bp_ents(p);
  const BpMate* __restrict__ mates = bp_mates(p);
  double* __restrict__ A = p.A;
  double* __restrict__ rhs = p.rhs;
  extern __shared__ double acc[];                  // [acc_len] accumulator row (the launch covers rows whose diagonal tile ends at or before acc_len), then the staged items
  double* il = acc + acc_len;                      // [CM_BP_ITEMS] lam of the column mate
  int* ib = reinterpret_cast<int*>(il + CM_BP_ITEMS);   // [CM_BP_ITEMS] its active index (-1: not in the active set)
  __shared__ int s_off[CM_WAVE], s_cs[CM_WAVE];
  __shared__ double s_le[CM_WAVE];
  const int a = row0 + bx;
  const int lane = threadIdx.x;
  double* Arow = A + (size_t)a * ld;
  // Only the tiles on and below the diagonal are ever read (the factorisation works on the lower triangle, the inverse uses the
  // strictly-upper tiles as scratch it writes first): the row is produced up to the end of its diagonal tile.
  const int b_end = (a / TS + 1) * TS;
  if (a >= n_act) {                 // identity padding up to the tile multiple
    for (int b = lane; b < b_end; b += CM_WAVE) Arow[b] = (b == a) ? 1.0 : 0.0;
    if (lane == 0) rhs[a] = 0.0;
    return;
  }
  const BpRow row = bp_rows(p)[a];
  for (int b = lane; b < b_end; b += CM_WAVE) acc[b] = 0.0;
  const int n = row.n;
  const double sig = scal[CM_S_SHAPE] / scal[CM_S_RATE];
  const int e_beg = row.e_beg, e_end = row.e_end;
  for (int e0 = e_beg; e0 < e_end; e0 += CM_WAVE) {          // 64 trials of the row at a time, one per lane
    const int e = e0 + lane;
    const bool have = e < e_end;
    BpEnt en; en.le = 0.0; en.cs = 0; en.cl = 0;
    if (have) en = ents[e];
    const double le = en.le;
    const int cs = en.cs, cl = en.cl;
    int incl = cl;                                            // inclusive wave scan of the column lengths
#pragma unroll
    for (int d = 1; d < CM_WAVE; d <<= 1) { const int v = __shfl_up(incl, d, CM_WAVE); if (lane >= d) incl += v; }
    s_off[lane] = incl - cl; s_cs[lane] = cs; s_le[lane] = le;                 // (the gather's search and the direct walk read these)
    __builtin_amdgcn_wave_barrier();
    const int nv = min(CM_WAVE, e_end - e0);                  // trials in this chunk
    const int off = incl - cl;
    int j0 = 0;
    while (j0 < nv) {
      const int base = __builtin_amdgcn_readlane(off, j0);
      // as many whole trials from j0 on as fit into the staging area (the offsets are monotone: the fitting lanes form a run from j0)
      const bool fits = lane >= j0 && lane < nv && incl - base <= CM_BP_ITEMS;
      const int n_take = __popcll(__ballot(fits));
      if (n_take == 0) {                                      // one trial with more mates than the staging area: walk it directly
        const double lj = s_le[j0];
        const int c0 = s_cs[j0], c1 = c0 + __builtin_amdgcn_readlane(cl, j0);
        for (int ce = c0 + lane; ce < c1; ce += CM_WAVE) {
          const BpMate m = mates[ce];
          if (m.b >= 0 && m.b < b_end) acc[m.b] = fma(lj, m.l, acc[m.b]);   // (columns beyond the diagonal tile are never stored: not accumulated either)
        }
        __builtin_amdgcn_wave_barrier();
        ++j0;
        continue;
      }
      const int j1 = j0 + n_take, cnt = __builtin_amdgcn_readlane(incl, j1 - 1) - base;
      // gather: item g of the list belongs to the trial j with s_off[j] <= g < s_off[j + 1]
      for (int it0 = 0; it0 < cnt; it0 += 4 * CM_WAVE) {
        int ce[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int it = it0 + q * CM_WAVE + lane;
          ce[q] = -1;
          if (it < cnt) {
            const int g = base + it;
            int lo = j0, hi = j1 - 1;                         // last trial whose offset is <= g
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_off[mid] <= g) lo = mid; else hi = mid - 1; }
            ce[q] = s_cs[lo] + (g - s_off[lo]);
          }
        }
        BpMate mq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { mq[q].b = -1; mq[q].l = 0.0; if (ce[q] >= 0) mq[q] = mates[ce[q]]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int it = it0 + q * CM_WAVE + lane;
          if (it < cnt) { ib[it] = mq[q].b; il[it] = mq[q].l; }
        }
      }
      __builtin_amdgcn_wave_barrier();
      // accumulate: trial by trial in the row's order; the mates of one trial are distinct neurons -> no conflicts within a step
      for (int j = j0; j < j1; ++j) {
        const int c = __builtin_amdgcn_readlane(cl, j);
        if (c == 0) continue;                                 // lam[e] == 0: the trial contributes nothing
        const int o = __builtin_amdgcn_readlane(off, j) - base;
        const double lj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(le), j), __builtin_amdgcn_readlane(__double2loint(le), j));
        for (int q = lane; q < c; q += CM_WAVE) {
          const int b = ib[o + q];
          if (b >= 0 && b < b_end) acc[b] = fma(lj, il[o + q], acc[b]);
        }
        __builtin_amdgcn_wave_barrier();
      }
      j0 = j1;
    }
    __builtin_amdgcn_wave_barrier();
  }
  const double b0 = beta_prior[n];
  const double prec0 = 1.0 / (b0 * b0);
  for (int b = lane; b < b_end; b += CM_WAVE) {
    double v = 0.0;
    if (b < n_act) v = (b == a) ? sig * (rs[(size_t)n * 4 + 3] + acc[b]) + prec0 : sig * acc[b];
    Arow[b] = v;
  }
  if (lane == 0) rhs[a] = sig * rs[(size_t)n * 4 + 2] + prec0 * mu_prior[n];
}

// ------------------------------------------------------------------------------------------------
// 32x32 tile helpers (256 threads, each owns a 2x2 patch)
// ------------------------------------------------------------------------------------------------
// Global-memory accesses as uniform base + 32-bit byte offset of the lane (`global_load v, v_off, s[base]`): the offsets of a
// lane are the same for every tile of a kernel, where generic 64-bit addresses cost two registers per access in flight.
__device__ __forceinline__ double cm_gload(const double* base, unsigned byte_off) {
  return *(const __attribute__((address_space(1))) double*)((const __attribute__((address_space(1))) char*)base + byte_off);
}
__device__ __forceinline__ void cm_gstore(double* base, unsigned byte_off, double v) {
  *(__attribute__((address_space(1))) double*)((__attribute__((address_space(1))) char*)base + byte_off) = v;
}
// A tile always moves as "every load in flight, then consume": the 1024 / NT elements of a thread go through registers, and a
// kernel that needs several tiles fetches all of them before the first tile_put (a load-and-store loop per tile costs one memory
// round trip per element and tile -- measured: 8 dependent round trips in front of a 0.2 us tile product).  NT = threads of the
// workgroup taking part (compile time).
// NT = 256: a thread moves two pairs of neighbours -- 16 B per lane and load (a wave-instruction covers four 256-byte tile rows; the texture
// addresser works per lane, 8-byte lanes need twice the instructions for the same bytes) and one 16-byte LDS store per pair.  Measured on 128 x
// n = 1000: Cholesky 1.65 -> 1.55 ms, inverse 1.29 -> 1.21 ms (the change of the LDS pitch alone: nothing -- the operand reads are not the bound).
typedef double cm_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cm_d2 cm_gload2(const double* base, unsigned byte_off) {
  return *(const __attribute__((address_space(1))) cm_d2*)((const __attribute__((address_space(1))) char*)base + byte_off);
}
template <int NT = 256>
__device__ __forceinline__ void tile_fetch(double (&r)[TS * TS / NT], const double* __restrict__ src, int ld) {
  constexpr int E = TS * TS / NT;
  if constexpr (E % 2 == 0) {
#pragma unroll
    for (int q = 0; q < E / 2; ++q) {
      const unsigned i = threadIdx.x + NT * q;
      const cm_d2 v = cm_gload2(src, ((i / (TS / 2)) * (unsigned)ld + 2 * (i % (TS / 2))) * (unsigned)sizeof(double));   // (a tile spans < 32 rows: far below 4 GB)
      r[2 * q] = v.x; r[2 * q + 1] = v.y;
    }
  } else {
#pragma unroll
    for (int q = 0; q < E; ++q) {
      const unsigned i = threadIdx.x + NT * q;
      r[q] = cm_gload(src, ((i / TS) * (unsigned)ld + (i % TS)) * (unsigned)sizeof(double));
    }
  }
}
template <int NT = 256, int PITCH = TP>
__device__ __forceinline__ void tile_put(double (*dst)[PITCH], const double (&r)[TS * TS / NT]) {
  constexpr int E = TS * TS / NT;
  static_assert(PITCH % 2 == 0, "16-byte LDS stores need an even pitch");
  if constexpr (E % 2 == 0) {
#pragma unroll
    for (int q = 0; q < E / 2; ++q) {
      const int i = threadIdx.x + NT * q;
      cm_d2 v; v.x = r[2 * q]; v.y = r[2 * q + 1];
      *reinterpret_cast<cm_d2*>(&dst[i / (TS / 2)][2 * (i % (TS / 2))]) = v;
    }
  } else {
#pragma unroll
    for (int q = 0; q < E; ++q) { const int i = threadIdx.x + NT * q; dst[i / TS][i % TS] = r[q]; }
  }
}
__device__ __forceinline__ void tile_store(double* __restrict__ dst, int ld, const double (*src)[TP]) {
  for (int i = threadIdx.x; i < TS * TS; i += blockDim.x) dst[(size_t)(i / TS) * ld + (i % TS)] = src[i / TS][i % TS];
}
// fp64 matrix-core tile products.  A 32x32x32 tile product is 4 output blocks of 16x16, one wavefront each (waves 0-3 of the
// workgroup), 8 x v_mfma_f64_16x16x4_f64 per block.  Operand layout (gfx950): lane l feeds A[row = l & 15][k = l >> 4] and
// B[k = l >> 4][col = l & 15]; the 4 accumulator registers of lane l hold D[row = (l >> 4) + 4 * reg][col = l & 15].
// Same peak rate as the vector FMA pipe on CDNA4, but each operand is read from LDS once per 16 products instead of once
// per 2 -- the register-tiled FMA version of these products was LDS-bound.
typedef double cm_d4 __attribute__((ext_vector_type(4)));
// acc += A[16 rb .., :] * B[16 cb .., :]^T        (rows of B are output columns)
__device__ __forceinline__ cm_d4 blk_mfma_nt(cm_d4 acc, const double (*A)[TP], const double (*B)[TP], int rb, int cb, int lane) {
  const int r = lane & 15, kk = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < TS / 4; ++ks)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[16 * rb + r][4 * ks + kk], B[16 * cb + r][4 * ks + kk], acc, 0, 0, 0);
  return acc;
}
// acc += A[16 rb .., :] * B[:, 16 cb ..]
__device__ __forceinline__ cm_d4 blk_mfma_nn(cm_d4 acc, const double (*A)[TP], const double (*B)[TPN], int rb, int cb, int lane) {
  const int r = lane & 15, kk = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < TS / 4; ++ks)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[16 * rb + r][4 * ks + kk], B[4 * ks + kk][16 * cb + r], acc, 0, 0, 0);
  return acc;
}
// element (reg) of a lane's accumulator <-> tile coordinates
#define BLK_ROW(rb, lane, reg) (16 * (rb) + ((lane) >> 4) + 4 * (reg))
#define BLK_COL(cb, lane) (16 * (cb) + ((lane) & 15))

// L^-1 of a 32x32 SPD tile (C = L L^T) by forward elimination of the augmented matrix [C | I] in LDS, all 256 threads,
// ONE barrier per pivot: after the sweep the right half holds Ltilde^-1 (unit lower, C = Ltilde D Ltilde^T) and
// L^-1 = D^-1/2 Ltilde^-1.  Only L^-1 is needed downstream (panel solve and inverse seed), so L itself is not formed.
// C is destroyed; Rt is scratch; Li receives L^-1 (lower triangular).
// Executed by the first `nt` threads of the workgroup; the caller retires all other waves first (a retired wave no longer
// counts at s_barrier), so each of the 33 barriers only has to collect nt / 64 waves.
__device__ __forceinline__ void tile_linv(double (*C)[TP], double (*Rt)[TP], double (*Li)[TP], int* __restrict__ flag, int nt) {
  __shared__ double dk[TS];
  const int tid = threadIdx.x;
  for (int i = tid; i < TS * TS; i += nt) Rt[i / TS][i % TS] = (i / TS == i % TS) ? 1.0 : 0.0;
  bool bad = false;
  for (int k = 0; k < TS; ++k) {
    __syncthreads();
    const double d = C[k][k];
    bad |= !(d > 0.0);                             // not SPD (also catches NaN)
    // 1/d sits on the 32-pivot dependency chain: hardware reciprocal estimate + two Newton steps (full double precision for
    // the normal-range pivots of an SPD tile) instead of the ~3x longer IEEE division sequence
    double rd = __builtin_amdgcn_rcp(d);
    rd = fma(fma(-d, rd, 1.0), rd, rd);
    rd = fma(fma(-d, rd, 1.0), rd, rd);
    if (tid == 0) dk[k] = d;
    const int rem = TS - 1 - k;                    // rows below the pivot
    for (int idx = tid; idx < rem * TS; idx += nt) {
      const int i = k + 1 + idx / TS, q = idx % TS;
      const double m = C[i][k] * rd;
      if (q < rem) { const int j = k + 1 + q; C[i][j] = fma(-m, C[k][j], C[i][j]); }
      else { const int c = q - rem; Rt[i][c] = fma(-m, Rt[k][c], Rt[i][c]); }
    }
  }
  __syncthreads();
  for (int i = tid; i < TS * TS; i += nt) {
    const int r = i / TS, c = i % TS;
    Li[r][c] = (c <= r) ? Rt[r][c] / sqrt(dk[r]) : 0.0;
  }
  if (bad && tid == 0) *flag = 1;
  __syncthreads();
}

// Same elimination with the tile in REGISTERS: thread (r = tid & 31, g = tid >> 5) of the first 256 threads owns columns
// 4 g .. 4 g + 3 of row r.  It runs in place: once column c of C is eliminated its slot holds column c of the right-hand
// block (slot k of row i becomes -m_i at pivot k), so a row is always 32 live values.  Per pivot the pivot row and the pivot
// column are broadcast through a double-buffered 64 doubles of LDS -- ONE 4-wave barrier, 4 LDS operations and 4 FMAs per
// thread per pivot instead of an LDS read-modify-write per element.  C is read, Li written.  All of the first 256 threads of
// the workgroup must call it and every other wave must have retired (a retired wave no longer counts at s_barrier).
__device__ __forceinline__ double cm_recip(double d) {
  // hardware reciprocal estimate (~26 bits) + two Newton steps: full double precision for the normal-range pivots of an SPD
  // tile, a third of the latency of the IEEE division sequence on the 32-pivot dependency chain
  double rd = __builtin_amdgcn_rcp(d);
  rd = fma(fma(-d, rd, 1.0), rd, rd);
  return fma(fma(-d, rd, 1.0), rd, rd);
}
// FOUR pivots (k .. k + 3, one 4-column group) per broadcast round: the four pivot rows and, for every row, its four entries
// in the pivot columns go through LDS together; every thread then replays the four elimination steps on the 4 x 4 pivot
// block, on the pivot rows' entries in its own columns and on its own four elements -- operation for operation what the
// one-pivot-per-round elimination does, so the result is bitwise the same, with 8 barriers / LDS round trips instead of 32.
__device__ __forceinline__ void tile_linv_reg(const double (*C)[TP], double (*Li)[TP], int* __restrict__ flag) {
  __shared__ __attribute__((aligned(16))) double prow[2][4][TS];     // pivot rows as they stand before the round
  __shared__ __attribute__((aligned(16))) double pcol[2][TS][4];     // a[r][k .. k+3] of every row r
  __shared__ double dk[TS];
  const int r = threadIdx.x & 31, g = threadIdx.x >> 5;
  double v[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) v[c] = C[r][4 * g + c];
#pragma unroll
  for (int k = 0; k < TS; k += 4) {
    const int b = (k >> 2) & 1, gk = k >> 2;
    if (r >= k && r < k + 4) {
#pragma unroll
      for (int c = 0; c < 4; ++c) prow[b][r - k][4 * g + c] = v[c];
    }
    if (g == gk) {
#pragma unroll
      for (int c = 0; c < 4; ++c) pcol[b][r][c] = v[c];
    }
    __syncthreads();
    double B[4][4], pr[4][4], ac[4];
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
#pragma unroll
      for (int c = 0; c < 4; ++c) { B[s_][c] = prow[b][s_][k + c]; pr[s_][c] = prow[b][s_][4 * g + c]; }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) ac[c] = pcol[b][r][c];
    const bool own = (g == gk);                            // this thread's columns ARE the pivot columns: v == ac, pr == B
    double dloc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const double d = B[t][t];
      dloc[t] = d;
      const double rd = cm_recip(d);
      // pivot rows below pivot t: inside the block, and in this thread's columns
#pragma unroll
      for (int s_ = t + 1; s_ < 4; ++s_) {
        const double ms = B[s_][t] * rd;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          B[s_][u] = (u == t) ? -ms : fma(-ms, B[t][u], B[s_][u]);
          const double x = fma(-ms, pr[t][u], pr[s_][u]);
          pr[s_][u] = (own && u == t) ? -ms : x;
        }
      }
      // this thread's own row
      const double m = (r > k + t) ? ac[t] * rd : 0.0;
      const bool below = r > k + t;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double xa = fma(-m, B[t][u], ac[u]);
        ac[u] = (u == t && below) ? -m : xa;
        const double xv = fma(-m, pr[t][u], v[u]);
        v[u] = (own && u == t && below) ? -m : xv;
      }
    }
    if (threadIdx.x == 0) {
#pragma unroll
      for (int t = 0; t < 4; ++t) { dk[k + t] = dloc[t]; if (!(dloc[t] > 0.0)) *flag = 1; }   // not SPD (also catches NaN)
    }
  }
  __syncthreads();
  const double rs = 1.0 / sqrt(dk[r]);               // L^-1 = D^-1/2 Ltilde^-1
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int col = 4 * g + c;
    Li[r][col] = (col < r) ? v[c] * rs : ((col == r) ? rs : 0.0);
  }
  __syncthreads();
}

// The same with TWO pivots (k, q = k + 1) per broadcast round -- 54 instead of 100 VGPRs, for the bandwidth-bound kernel whose
// occupancy matters more than the last microsecond of its single factoring workgroup: both pivot rows and both pivot columns go through LDS together, every thread
// derives the second pivot (row q after step k) itself, operation for operation as its owner would, and applies both
// elimination steps to its four elements.  16 barriers and LDS round trips instead of 32; results are bitwise those of the
// one-pivot-per-round elimination.
__device__ __forceinline__ void tile_linv_reg2(const double (*C)[TP], double (*Li)[TP], int* __restrict__ flag) {
  __shared__ __attribute__((aligned(16))) double rowk[2][TS], rowq[2][TS];
  __shared__ double colk[2][TS], colq[2][TS], dk[TS];
  const int r = threadIdx.x & 31, g = threadIdx.x >> 5;
  double v[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) v[c] = C[r][4 * g + c];
#pragma unroll
  for (int k = 0; k < TS; k += 2) {
    const int q = k + 1, b = (k >> 1) & 1, gk = k >> 2;      // k is even: both pivot columns live in the same 4-column group
    if (r == k) {
#pragma unroll
      for (int c = 0; c < 4; ++c) rowk[b][4 * g + c] = v[c];
    }
    if (r == q) {
#pragma unroll
      for (int c = 0; c < 4; ++c) rowq[b][4 * g + c] = v[c];
    }
    if (g == gk) { colk[b][r] = v[k & 3]; colq[b][r] = v[q & 3]; }
    __syncthreads();
    const double dkk = rowk[b][k], akq = rowk[b][q], aqk = colk[b][q], aqq = rowq[b][q];
    const double cvk = colk[b][r], cvq = colq[b][r];
    double rk[4], rq[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { rk[c] = rowk[b][4 * g + c]; rq[c] = rowq[b][4 * g + c]; }
    // step k
    const double rdk = cm_recip(dkk);
    const double mq = aqk * rdk;                           // multiplier of row q at step k
    const double dq = fma(-mq, akq, aqq);                  // second pivot = a[q][q] after step k
    const double rdq = cm_recip(dq);
    const double mi = (r > k) ? cvk * rdk : 0.0;
    const double aiq = fma(-mi, akq, cvq);                 // a[r][q] after step k
    const double mi2 = (r > q) ? aiq * rdq : 0.0;
    if (threadIdx.x == 0) { dk[k] = dkk; dk[q] = dq; if (!(dkk > 0.0) || !(dq > 0.0)) *flag = 1; }   // not SPD (also catches NaN)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bool slot_k = (g == gk) && (c == (k & 3)), slot_q = (g == gk) && (c == (q & 3));
      double rqp = fma(-mq, rk[c], rq[c]);                 // row q after step k ...
      rqp = slot_k ? -mq : rqp;                            // ... whose slot k holds -m_q
      double x = fma(-mi, rk[c], v[c]);                    // step k
      x = (slot_k && r > k) ? -mi : x;
      x = fma(-mi2, rqp, x);                               // step q
      x = (slot_q && r > q) ? -mi2 : x;
      v[c] = x;
    }
  }
  __syncthreads();
  const double rs = 1.0 / sqrt(dk[r]);               // L^-1 = D^-1/2 Ltilde^-1
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int col = 4 * g + c;
    Li[r][col] = (col < r) ? v[c] * rs : ((col == r) ? rs : 0.0);
  }
  __syncthreads();
}

// factor the first diagonal tile
template <bool BATCH>
__global__ __launch_bounds__(256) void k_chol_first(cm_parg<BATCH> pa, int na_arg, int skip_own) {
  CM_SOLVE_PROLOGUE(2)
  if (na == 0) return;
  double* __restrict__ A = p.A;
  double* __restrict__ Dinv = p.Dinv;
  int* __restrict__ flag = p.n_act + 1;
  __shared__ __attribute__((aligned(16))) double C[TS][TP], Li[TS][TP];
  double rc[4];
  tile_fetch(rc, A, ld);
  tile_put(C, rc);
  __syncthreads();
  tile_linv_reg(C, Li, flag);
  for (int i = threadIdx.x; i < TS * TS; i += 256) Dinv[(size_t)(i / TS) * TS + (i % TS)] = Li[i / TS][i % TS];
}

// panel solve for large matrices: Lf[i][j] = A[i][j] * L_jj^-T for i > j   (one 256-thread workgroup per tile row)
template <bool BATCH>
__global__ __launch_bounds__(256) void k_chol_trsm(cm_parg<BATCH> pa, int na_arg, int skip_own, int j, int large_min) {
  CM_SOLVE_PROLOGUE(2)
  if (nblk <= large_min || j + 1 + (int)blockIdx.x >= nblk) return;
  const double* __restrict__ A = p.A;
  double* __restrict__ Lf = p.Lf;
  const double* __restrict__ Dinv = p.Dinv;
  __shared__ __attribute__((aligned(16))) double P[TS][TP], Li[TS][TP];
  const int i = j + 1 + blockIdx.x;
  double rp[4], rl[4];
  tile_fetch(rp, A + (size_t)i * TS * ld + (size_t)j * TS, ld);
  tile_fetch(rl, Dinv + (size_t)j * TS * TS, TS);
  tile_put(P, rp);
  tile_put(Li, rl);
  __syncthreads();
  const int wave = threadIdx.x / CM_WAVE, lane = threadIdx.x % CM_WAVE, rb = wave >> 1, cb = wave & 1;
  cm_d4 c = {0.0, 0.0, 0.0, 0.0};
  c = blk_mfma_nt(c, P, Li, rb, cb, lane);       // X = P * Li^T
  double* out = Lf + (size_t)i * TS * ld + (size_t)j * TS;
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) out[(size_t)BLK_ROW(rb, lane, reg) * ld + BLK_COL(cb, lane)] = c[reg];
}

// Large matrices (more than 48 tiles) and every batch: the trailing update is bandwidth-bound -- every pass re-reads and re-writes
// the trailing matrix -- so the panels go in PAIRS (j, j + 1) and the trailing matrix is read and rewritten once per pair:
//   k_chol_bulk<thin>(j)  X^j_i   = A[i][j] L_jj^-T  (-> Lf[i][j]),  A[i][j+1] -= X^j_i X^j_{j+1}^T      i >= j + 1   (one tile column: the panel
//                         solve of column j and the update of column j + 1 in one launch; block 0 factors the diagonal tile (j+1, j+1))
//   k_chol_trsm(j + 1)    X^{j+1}_i = A[i][j+1] L_{j+1,j+1}^-T                        i > j + 1
//   k_chol_bulk<wide>(j)  A[i][l] = (A[i][l] - X^j_i X^j_l^T) - X^{j+1}_i X^{j+1}_l^T   i >= l >= j + 2  (block 0 factors the diagonal tile (j+2, j+2))
// Each product is its own matrix-core accumulation from zero and the two subtractions happen in panel order, so every tile holds
// bitwise what one rank-32 update per panel leaves (k_chol_panel).  256-thread workgroups, one 16x16 output block per wave, the
// output tile prefetched into registers, all panel tiles of the workgroup staged in LDS by one round of loads.  A circuit whose last
// panel has no partner (one trailing tile row) is complete after the thin pass: its wide pass has no tiles.
template <bool BATCH, bool WIDE>
__global__ __launch_bounds__(256) void k_chol_bulk(cm_parg<BATCH> pa, int na_arg, int skip_own, int j, int large_min) {
  int bx = (int)blockIdx.x, circ = 0;
  if (BATCH) cm_xcd_remap(bx, circ, skip_own >> 8);  // all tiles of a circuit on one XCD: its panel tiles go through one L2
  CM_SOLVE_PROLOGUE_AT(2, pa.get(circ))
  const int tr = nblk - j - 1;                       // trailing tile rows of panel j
  if (nblk <= large_min || tr <= 0 || bx >= (WIDE ? (tr - 1) * tr / 2 : tr)) return;
  double* __restrict__ A = p.A;
  const double* __restrict__ Lf = p.Lf;
  double* __restrict__ Dinv = p.Dinv;
  int* __restrict__ flag = p.n_act + 1;
  __shared__ __attribute__((aligned(16))) double Sm[WIDE ? 2 : 3][TS][TP];        // two tiles (17 KB; the thin pass also stages L_jj^-1): LDS is not what limits the workgroups per CU
  const int t = bx;                                  // id 0 is the pass's diagonal tile
  int i, l;
  if (WIDE) {                                        // linear id -> (i, l), j + 2 <= l <= i
    int ip = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    ip = __builtin_amdgcn_readfirstlane(ip);        // (the square root runs on the vector unit; the result is uniform: scalar addresses below)
    while ((ip + 1) * (ip + 2) / 2 <= t) ++ip;
    while (ip * (ip + 1) / 2 > t) --ip;
    i = j + 2 + ip; l = j + 2 + (t - ip * (ip + 1) / 2);
  } else { i = j + 1 + t; l = j + 1; }
  const int wave = threadIdx.x / CM_WAVE, lane = threadIdx.x % CM_WAVE, rb = wave >> 1, cb = wave & 1;
  double* out = A + (size_t)i * TS * ld + (size_t)l * TS;
  double o[4];
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) o[reg] = cm_gload(out, ((unsigned)BLK_ROW(rb, lane, reg) * (unsigned)ld + (unsigned)BLK_COL(cb, lane)) * 8u);
  // every load of the workgroup is issued before the first one is consumed (one memory round trip, not one per tile): the panel
  // tiles go through registers, and a diagonal output tile (l == i) simply fetches its panel tile twice -- no branch between loads
  double rx[WIDE ? 4 : 3][4];
  if (WIDE) {
    tile_fetch(rx[0], Lf + (size_t)i * TS * ld + (size_t)j * TS, ld);
    tile_fetch(rx[1], Lf + (size_t)l * TS * ld + (size_t)j * TS, ld);
    tile_fetch(rx[2], Lf + (size_t)i * TS * ld + (size_t)(j + 1) * TS, ld);
    tile_fetch(rx[3], Lf + (size_t)l * TS * ld + (size_t)(j + 1) * TS, ld);
    tile_put(Sm[0], rx[0]);
    tile_put(Sm[1], rx[1]);
    __syncthreads();
  } else {
    // thin pass: the panel solve of column j rides along (no launch of its own) -- X_i = A[i][j] L_jj^-T and X_{j+1} are formed here
    // from the unsolved panel tiles, exactly as k_chol_trsm / k_chol_panel form them, and X_i is the tile of the factor this
    // workgroup owns (Lf[i][j], read by the rank-64 pass and the inverse)
    tile_fetch(rx[0], A + (size_t)i * TS * ld + (size_t)j * TS, ld);
    tile_fetch(rx[1], A + (size_t)l * TS * ld + (size_t)j * TS, ld);
    tile_fetch(rx[2], Dinv + (size_t)j * TS * TS, TS);
    tile_put(Sm[0], rx[0]);
    tile_put(Sm[1], rx[1]);
    tile_put(Sm[2], rx[2]);
    __syncthreads();
    cm_d4 xi = {0.0, 0.0, 0.0, 0.0}, xl = {0.0, 0.0, 0.0, 0.0};
    xi = blk_mfma_nt(xi, Sm[0], Sm[2], rb, cb, lane);   // X_i = P_i Linv^T
    xl = blk_mfma_nt(xl, Sm[1], Sm[2], rb, cb, lane);   // X_{j+1}
    __syncthreads();                                     // every wave has read the unsolved tiles: the solved ones replace them
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      Sm[0][BLK_ROW(rb, lane, reg)][BLK_COL(cb, lane)] = xi[reg];
      Sm[1][BLK_ROW(rb, lane, reg)][BLK_COL(cb, lane)] = xl[reg];
    }
    __syncthreads();
    tile_store(const_cast<double*>(Lf) + (size_t)i * TS * ld + (size_t)j * TS, ld, Sm[0]);
  }
  {
    cm_d4 c = {0.0, 0.0, 0.0, 0.0};
    c = blk_mfma_nt(c, Sm[0], Sm[1], rb, cb, lane);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) o[reg] -= c[reg];
  }
  if (WIDE) {                                        // the second panel's tiles (waiting in registers) take the same two LDS tiles
    __syncthreads();
    tile_put(Sm[0], rx[2]);
    tile_put(Sm[1], rx[3]);
    __syncthreads();
    cm_d4 c = {0.0, 0.0, 0.0, 0.0};
    c = blk_mfma_nt(c, Sm[0], Sm[1], rb, cb, lane);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) o[reg] -= c[reg];
  }
  if (t != 0) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) cm_gstore(out, ((unsigned)BLK_ROW(rb, lane, reg) * (unsigned)ld + (unsigned)BLK_COL(cb, lane)) * 8u, o[reg]);
    return;
  }
  // block 0: the updated diagonal tile never goes back to HBM -- factor it here, next to the bulk of the update, and
  // publish L^-1 for the next panel's solve (the tile's own copy in A is dead after this)
  __syncthreads();                                 // every wave is done with the panel tiles: tile 0 receives the diagonal tile, tile 1 L^-1
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) Sm[0][BLK_ROW(rb, lane, reg)][BLK_COL(cb, lane)] = o[reg];
  __syncthreads();
  tile_linv_reg2(Sm[0], Sm[1], flag);
  tile_store(Dinv + (size_t)(WIDE ? j + 2 : j + 1) * TS * TS, TS, Sm[1]);
}

// ------------------------------------------------------------------------------------------------
// GROUPED family (every batch, and single matrices above CM_LARGE_TILES tile rows): the panels go in groups of CM_GRP = 4 tile columns and
// the trailing matrix is read and rewritten once per 128 columns, by workgroups that own a 64 x 64 SUPER-TILE of it:
//   k_chol_diag(0)        factors the first 4 x 4 block of tiles on the diagonal (one workgroup; tiles in registers, operands through LDS)
//   per group g (first tile column j0 = 4 g), while tile rows remain below the group:
//     k_chol_gpanel(j0)   one workgroup per trailing tile row i: X^p_i = (A[i][j0+p] - sum_{p' < p} X^p'_i L[j0+p][j0+p']^T) L_pp^-T, p = 0..3
//     k_chol_wide(j0)     one workgroup per super-tile (2 x 2 tiles, one 32 x 32 tile = four 16 x 16 matrix-core blocks per wavefront):
//                         A[i][l] = (((A[i][l] - X^0_i X^0_l^T) - X^1_i X^1_l^T) - X^2_i X^2_l^T) - X^3_i X^3_l^T; the panel tiles of a super-tile
//                         are staged once for its four output tiles (double-buffered in LDS, the next panel's loads in flight behind 32 matrix
//                         instructions per wavefront) -- per tile product 12 KB move between L2 and a CU where the pair family moves 24 KB,
//                         and the trailing matrix goes through HBM once per four panels instead of once per two.  Workgroup 0 takes the
//                         three super-tiles that hold the NEXT group's diagonal block and then factors that block (chol_diag_block), so the
//                         serial part of the next group runs beside the bulk of this one.
// Every tile product is its own matrix-core accumulation from zero over the tile's 32 columns and the subtractions happen in panel order:
// each tile holds bitwise what one rank-32 update per panel leaves (k_chol_panel, the pair family).
// ------------------------------------------------------------------------------------------------
#define CM_GRP 4
#ifndef CM_WIDE_BUFS
#define CM_WIDE_BUFS 2                                // LDS stages of the rank-128 pass: 2 = double-buffered (70 KB: two workgroups per CU), 1 = one stage + register prefetch (44 KB: three)
#endif
#define CM_GSLOTS 8                                   // tile slots of dynamic LDS (TS x TP doubles each)
static constexpr size_t CM_GROUP_LDS = (size_t)CM_GSLOTS * TS * TP * sizeof(double);
static constexpr size_t CM_WIDE_LDS = (size_t)(CM_WIDE_BUFS == 2 ? 8 : 5) * TS * TP * sizeof(double);   // (the diagonal-block factorisation of workgroup 0 needs five slots)
typedef double (*cm_tile_t)[TP];
__device__ __forceinline__ cm_tile_t cm_slot(double* sm, int s) { return reinterpret_cast<cm_tile_t>(sm + (size_t)s * TS * TP); }
// a wavefront's 16 x 16 block (rb, cb) of a 32 x 32 tile <-> 4 registers (the matrix-core accumulator layout)
__device__ __forceinline__ void blk_gload(double (&o)[4], const double* tile, int ld, int rb, int cb, int lane) {
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) o[reg] = cm_gload(tile, ((unsigned)BLK_ROW(rb, lane, reg) * (unsigned)ld + (unsigned)BLK_COL(cb, lane)) * 8u);
}
__device__ __forceinline__ void blk_gstore(double* tile, int ld, int rb, int cb, int lane, const double (&o)[4]) {
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) cm_gstore(tile, ((unsigned)BLK_ROW(rb, lane, reg) * (unsigned)ld + (unsigned)BLK_COL(cb, lane)) * 8u, o[reg]);
}
__device__ __forceinline__ void blk_put(cm_tile_t dst, int rb, int cb, int lane, const double (&o)[4]) {
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) dst[BLK_ROW(rb, lane, reg)][BLK_COL(cb, lane)] = o[reg];
}
__device__ __forceinline__ void blk_sub(double (&o)[4], const cm_d4& c) {
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) o[reg] -= c[reg];
}

// Factorisation of the gs x gs block of tiles (gs <= 4) on the diagonal at tile (j0, j0), whose tiles have received every update of the
// earlier panels and are read from A: the right-looking sequence of the other families restricted to the block -- L_pp^-1 by the
// register-resident elimination, X_q = A[q][p] L_pp^-T, A[q][r] -= X_q X_r^T -- by ONE 256-thread workgroup.  The (up to) ten tiles live in
// registers in the accumulator layout (a 16 x 16 block of every tile per wavefront: the products subtract in place), matrix-core
// operands go through five LDS slots.  Writes Dinv[j0 ..] and the block's tiles of Lf.
__device__ __forceinline__ void chol_diag_block(const cm_pdev& p, double* sm, int j0, int gs, int ld) {
  const double* __restrict__ A = p.A;
  double* __restrict__ Lf = p.Lf;
  double* __restrict__ Dinv = p.Dinv;
  int* __restrict__ flag = p.n_act + 1;
  const int wave = threadIdx.x / CM_WAVE, lane = threadIdx.x % CM_WAVE, rb = wave >> 1, cb = wave & 1;
  cm_tile_t C = cm_slot(sm, 0), Li = cm_slot(sm, 1);
  double D[10][4];                                                  // tile (q, r), r <= q, at index q (q + 1) / 2 + r
#pragma unroll
  for (int q = 0; q < CM_GRP; ++q)
#pragma unroll
    for (int r = 0; r <= q; ++r)
      if (q < gs) blk_gload(D[q * (q + 1) / 2 + r], A + (size_t)(j0 + q) * TS * ld + (size_t)(j0 + r) * TS, ld, rb, cb, lane);
#pragma unroll
  for (int pp = 0; pp < CM_GRP; ++pp) {
    if (pp >= gs) break;
    blk_put(C, rb, cb, lane, D[pp * (pp + 1) / 2 + pp]);
    __syncthreads();
    tile_linv_reg(C, Li, flag);                                      // (ends on a barrier)
    {
      double* dst = Dinv + (size_t)(j0 + pp) * TS * TS;
      for (int i = threadIdx.x; i < TS * TS; i += 256) dst[i] = Li[i / TS][i % TS];
    }
#pragma unroll
    for (int q = pp + 1; q < CM_GRP; ++q)
      if (q < gs) blk_put(cm_slot(sm, 2 + q - pp - 1), rb, cb, lane, D[q * (q + 1) / 2 + pp]);
    __syncthreads();
    double x[CM_GRP - 1][4];
#pragma unroll
    for (int q = pp + 1; q < CM_GRP; ++q)
      if (q < gs) {
        cm_d4 c = {0.0, 0.0, 0.0, 0.0};
        c = blk_mfma_nt(c, cm_slot(sm, 2 + q - pp - 1), Li, rb, cb, lane);       // X_q = A[q][p] L_pp^-T
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) x[q - pp - 1][reg] = c[reg];
      }
    __syncthreads();                                                 // every wave has read the unsolved tiles: the solved ones replace them
#pragma unroll
    for (int q = pp + 1; q < CM_GRP; ++q)
      if (q < gs) {
        blk_put(cm_slot(sm, 2 + q - pp - 1), rb, cb, lane, x[q - pp - 1]);
        blk_gstore(Lf + (size_t)(j0 + q) * TS * ld + (size_t)(j0 + pp) * TS, ld, rb, cb, lane, x[q - pp - 1]);
      }
    __syncthreads();
#pragma unroll
    for (int q = pp + 1; q < CM_GRP; ++q)
#pragma unroll
      for (int r = pp + 1; r <= q; ++r)
        if (q < gs) {
          cm_d4 c = {0.0, 0.0, 0.0, 0.0};
          c = blk_mfma_nt(c, cm_slot(sm, 2 + q - pp - 1), cm_slot(sm, 2 + r - pp - 1), rb, cb, lane);
          blk_sub(D[q * (q + 1) / 2 + r], c);
        }
    __syncthreads();                                                 // the solved tiles are dead: the next pivot tile may overwrite the slots
  }
}

template <bool BATCH>
__global__ __launch_bounds__(256) void k_chol_diag(cm_parg<BATCH> pa, int na_arg, int skip_own, int j0, int large_min) {
  CM_SOLVE_PROLOGUE(2)
  (void)large_min;                                   // (a one-tile matrix of a batch is factored here too: nothing else touches it)
  if (na == 0 || j0 >= nblk) return;
  extern __shared__ __attribute__((aligned(16))) double sm_dyn[];
  chol_diag_block(p, sm_dyn, j0, min(CM_GRP, nblk - j0), ld);
}

// Panel solve of a group for one trailing tile row (see above).  LDS slots: 0-2 the solved tiles X^0..X^2 of the row, 3 the tile being
// solved, 4-6 the block's tiles L[j0+q][j0+p] of the current column q, 7 L_qq^-1.  All fourteen tiles are requested before the first is used.
template <bool BATCH>
__global__ __launch_bounds__(256, 2) void k_chol_gpanel(cm_parg<BATCH> pa, int na_arg, int skip_own, int j0, int large_min) {
  int bx = (int)blockIdx.x, circ = 0;
  if (BATCH) cm_xcd_remap(bx, circ, skip_own >> 8);
  CM_SOLVE_PROLOGUE_AT(2, pa.get(circ))
  const int t = nblk - j0 - CM_GRP;                  // trailing tile rows (a group with rows below it is CM_GRP panels wide)
  if (nblk <= large_min || bx >= t) return;
  const double* __restrict__ A = p.A;
  double* __restrict__ Lf = p.Lf;
  const double* __restrict__ Dinv = p.Dinv;
  extern __shared__ __attribute__((aligned(16))) double sm_dyn[];
  const int i = j0 + CM_GRP + bx;
  const int wave = threadIdx.x / CM_WAVE, lane = threadIdx.x % CM_WAVE, rb = wave >> 1, cb = wave & 1;
  double a[CM_GRP][4], rl[CM_GRP * (CM_GRP - 1) / 2][4], rd[CM_GRP][4];
#pragma unroll
  for (int q = 0; q < CM_GRP; ++q) blk_gload(a[q], A + (size_t)i * TS * ld + (size_t)(j0 + q) * TS, ld, rb, cb, lane);
#pragma unroll
  for (int q = 0; q < CM_GRP; ++q) {
    tile_fetch(rd[q], Dinv + (size_t)(j0 + q) * TS * TS, TS);
#pragma unroll
    for (int pp = 0; pp < q; ++pp) tile_fetch(rl[q * (q - 1) / 2 + pp], Lf + (size_t)(j0 + q) * TS * ld + (size_t)(j0 + pp) * TS, ld);
  }
  cm_tile_t P = cm_slot(sm_dyn, 3), Dv = cm_slot(sm_dyn, 7);
#pragma unroll
  for (int q = 0; q < CM_GRP; ++q) {
#pragma unroll
    for (int pp = 0; pp < q; ++pp) tile_put(cm_slot(sm_dyn, 4 + pp), rl[q * (q - 1) / 2 + pp]);
    tile_put(Dv, rd[q]);
    __syncthreads();
#pragma unroll
    for (int pp = 0; pp < q; ++pp) {                 // updates of tile (i, j0 + q) by the earlier panels of the group, in panel order
      cm_d4 c = {0.0, 0.0, 0.0, 0.0};
      c = blk_mfma_nt(c, cm_slot(sm_dyn, pp), cm_slot(sm_dyn, 4 + pp), rb, cb, lane);
      blk_sub(a[q], c);
    }
    blk_put(P, rb, cb, lane, a[q]);
    __syncthreads();
    cm_d4 x = {0.0, 0.0, 0.0, 0.0};
    x = blk_mfma_nt(x, P, Dv, rb, cb, lane);         // X^q_i = P L_qq^-T
    double xr[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) xr[reg] = x[reg];
    blk_gstore(Lf + (size_t)i * TS * ld + (size_t)(j0 + q) * TS, ld, rb, cb, lane, xr);
    if (q + 1 < CM_GRP) blk_put(cm_slot(sm_dyn, q), rb, cb, lane, xr);
    __syncthreads();                                 // X^q is in place; P, Dv and the L slots may be overwritten
  }
}

// one super-tile (I, L) of the trailing matrix below group j0: see above.  Slots: stage s holds the row-side tiles (2 s', s' = 0, 1 ->
// slots 4 s + s') and the column-side tiles (slots 4 s + 2 + s').
__device__ __forceinline__ void chol_wide_supertile(const cm_pdev& p, double* sm, int j0, int nblk, int ld, int I, int L) {
  double* __restrict__ A = p.A;
  const double* __restrict__ Lf = p.Lf;
  const int r0 = j0 + CM_GRP;
  const int wave = threadIdx.x / CM_WAVE, lane = threadIdx.x % CM_WAVE, qi = wave >> 1, ql = wave & 1;
  const int ti = r0 + 2 * I + qi, tl = r0 + 2 * L + ql;
  const bool valid = ti < nblk && tl <= ti;          // the wavefront's tile exists and lies on or below the diagonal
  // tile rows the stage loads read (a missing second row of the last super-row re-reads the last row: loaded, never used)
  const int ri0 = r0 + 2 * I, ri1 = min(ri0 + 1, nblk - 1), rl0 = r0 + 2 * L, rl1 = min(rl0 + 1, nblk - 1);
  double o[2][2][4];
  double* out = A + (size_t)ti * TS * ld + (size_t)tl * TS;
  if (valid) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) blk_gload(o[rb][cb], out, ld, rb, cb, lane);
  }
  double st[4][4];                                   // the stage in flight: 4 tiles x 4 elements per thread
  auto fetch = [&](int pp) {
    const size_t cj = (size_t)(j0 + pp) * TS;
    tile_fetch(st[0], Lf + (size_t)ri0 * TS * ld + cj, ld);
    tile_fetch(st[1], Lf + (size_t)ri1 * TS * ld + cj, ld);
    tile_fetch(st[2], Lf + (size_t)rl0 * TS * ld + cj, ld);
    tile_fetch(st[3], Lf + (size_t)rl1 * TS * ld + cj, ld);
  };
  auto put = [&](int s) {
#pragma unroll
    for (int k = 0; k < 4; ++k) tile_put(cm_slot(sm, 4 * s + k), st[k]);
  };
  fetch(0);
#if CM_WIDE_BUFS == 2
  put(0);
#endif
#pragma unroll
  for (int pp = 0; pp < CM_GRP; ++pp) {
#if CM_WIDE_BUFS == 2
    if (pp + 1 < CM_GRP) fetch(pp + 1);              // in flight behind this panel's 32 matrix instructions
    __syncthreads();
#else
    __syncthreads();                                 // (every wavefront is done with the last stage: one buffer, the next stage waits in registers)
    put(0);
    __syncthreads();
    if (pp + 1 < CM_GRP) fetch(pp + 1);              // in flight behind this panel's 32 matrix instructions
#endif
    if (valid) {
      const double (*Ai)[TP] = cm_slot(sm, (CM_WIDE_BUFS == 2 ? 4 * (pp & 1) : 0) + qi);
      const double (*Bl)[TP] = cm_slot(sm, (CM_WIDE_BUFS == 2 ? 4 * (pp & 1) : 0) + 2 + ql);
      const int r = lane & 15, kk = lane >> 4;
      cm_d4 c00 = {0.0, 0.0, 0.0, 0.0}, c01 = c00, c10 = c00, c11 = c00;
#pragma unroll
      for (int ks = 0; ks < TS / 4; ++ks) {
        const double a0 = Ai[r][4 * ks + kk], a1 = Ai[16 + r][4 * ks + kk], b0 = Bl[r][4 * ks + kk], b1 = Bl[16 + r][4 * ks + kk];
        c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c00, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, c01, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, c10, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c11, 0, 0, 0);
      }
      blk_sub(o[0][0], c00); blk_sub(o[0][1], c01); blk_sub(o[1][0], c10); blk_sub(o[1][1], c11);
    }
#if CM_WIDE_BUFS == 2
    if (pp + 1 < CM_GRP) put((pp + 1) & 1);          // (that buffer was last read before this iteration's barrier)
#endif
  }
  if (valid) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) blk_gstore(out, ld, rb, cb, lane, o[rb][cb]);
  }
}

template <bool BATCH>
__global__ __launch_bounds__(256, CM_WIDE_BUFS == 2 ? 2 : 3) void k_chol_wide(cm_parg<BATCH> pa, int na_arg, int skip_own, int j0, int large_min) {
  int bx = (int)blockIdx.x, circ = 0;
  if (BATCH) cm_xcd_remap(bx, circ, skip_own >> 8);  // all super-tiles of a circuit on one XCD: its panel tiles go through one L2
  CM_SOLVE_PROLOGUE_AT(2, pa.get(circ))
  const int t = nblk - j0 - CM_GRP;                  // trailing tile rows
  if (nblk <= large_min || t <= 0) return;
  const int ns = (t + 1) >> 1, nst = ns * (ns + 1) / 2;   // super-rows, super-tiles (lower triangle, row-major: id = I (I + 1) / 2 + L)
  extern __shared__ __attribute__((aligned(16))) double sm_dyn[];
  if (bx > 0) {
    const int id = bx + 2;                           // workgroup 0 owns super-tiles 0, 1, 2
    if (id >= nst) return;
    int I = (int)((sqrt(8.0 * (double)id + 1.0) - 1.0) * 0.5);
    I = __builtin_amdgcn_readfirstlane(I);
    while ((I + 1) * (I + 2) / 2 <= id) ++I;
    while (I * (I + 1) / 2 > id) --I;
    chol_wide_supertile(p, sm_dyn, j0, nblk, ld, I, id - I * (I + 1) / 2);
    return;
  }
  // workgroup 0: the super-tiles (0, 0), (1, 0), (1, 1) hold the next group's diagonal block (tile rows j0 + 4 .. j0 + 7) ...
  chol_wide_supertile(p, sm_dyn, j0, nblk, ld, 0, 0);
  if (ns >= 2) {
    chol_wide_supertile(p, sm_dyn, j0, nblk, ld, 1, 0);
    chol_wide_supertile(p, sm_dyn, j0, nblk, ld, 1, 1);
  }
  __syncthreads();                                   // ... which this workgroup now factors (its stores are visible to itself)
  chol_diag_block(p, sm_dyn, j0 + CM_GRP, min(CM_GRP, t), ld);
}

// Panel j in ONE launch (1024-thread workgroups, one per trailing tile (i, l), j < l <= i):
//   X_i = A[i][j] L_jj^-T, X_l = A[l][j] L_jj^-T     (panel solve, recomputed per workgroup from the read-only panel tiles)
//   A[i][l] -= X_i X_l^T                             (trailing update, in place: the tile is owned by this workgroup)
//   l == j + 1: Lf[i][j] = X_i                       (the factor's panel goes to a separate matrix: A[i][j] is still being read)
//   i == l == j + 1: factor the next diagonal tile on the spot -> Dinv[j + 1]
// The 32x32x32 tile products run on waves 0-3 (one 16x16 block each); all 1024 threads move the tiles (one element each, every
// load of the workgroup in flight before the first is consumed; a diagonal output tile fetches its panel tile twice rather than branch).
template <bool BATCH>
__global__ __launch_bounds__(1024) void k_chol_panel(cm_parg<BATCH> pa, int na_arg, int skip_own, int j, int large_min) {
  CM_SOLVE_PROLOGUE(2)
  {
    const int tr = nblk - j - 1;
    if (nblk > large_min || tr <= 0 || (int)blockIdx.x >= tr * (tr + 1) / 2) return;
  }
  double* __restrict__ A = p.A;
  double* __restrict__ Lf = p.Lf;
  double* __restrict__ Dinv = p.Dinv;
  int* __restrict__ flag = p.n_act + 1;
  __shared__ __attribute__((aligned(16))) double Pi[TS][TP], Pl[TS][TP], Li[TS][TP], Xi[TS][TP], Xl[TS][TP];
  const int t = blockIdx.x;
  int ip = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((ip + 1) * (ip + 2) / 2 <= t) ++ip;
  while (ip * (ip + 1) / 2 > t) --ip;
  const int lp = t - ip * (ip + 1) / 2;
  const int i = j + 1 + ip, l = j + 1 + lp;
  const bool mm = threadIdx.x < 256;                 // waves 0-3 own the four 16x16 blocks of the tile products
  const int wave = threadIdx.x / CM_WAVE, lane = threadIdx.x % CM_WAVE, rb = (wave >> 1) & 1, cb = wave & 1;
  double rpi[1], rpl[1], rli[1];
  tile_fetch<1024>(rpi, A + (size_t)i * TS * ld + (size_t)j * TS, ld);
  tile_fetch<1024>(rpl, A + (size_t)l * TS * ld + (size_t)j * TS, ld);
  tile_fetch<1024>(rli, Dinv + (size_t)j * TS * TS, TS);
  double* out = A + (size_t)i * TS * ld + (size_t)l * TS;
  double oprev[4] = {0.0, 0.0, 0.0, 0.0};
  if (mm) {                                          // the tile to be updated: requested now, consumed after the products
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) oprev[reg] = out[(size_t)BLK_ROW(rb, lane, reg) * ld + BLK_COL(cb, lane)];
  }
  tile_put<1024>(Pi, rpi);
  tile_put<1024>(Pl, rpl);
  tile_put<1024>(Li, rli);
  __syncthreads();
  if (mm) {
    cm_d4 c = {0.0, 0.0, 0.0, 0.0};
    c = blk_mfma_nt(c, Pi, Li, rb, cb, lane);          // X_i = P_i Linv^T
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) Xi[BLK_ROW(rb, lane, reg)][BLK_COL(cb, lane)] = c[reg];
    if (l != i) {
      cm_d4 e = {0.0, 0.0, 0.0, 0.0};
      e = blk_mfma_nt(e, Pl, Li, rb, cb, lane);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) Xl[BLK_ROW(rb, lane, reg)][BLK_COL(cb, lane)] = e[reg];
    }
  }
  __syncthreads();
  if (l == j + 1) tile_store(Lf + (size_t)i * TS * ld + (size_t)j * TS, ld, Xi);
  const bool diag = (i == j + 1 && l == j + 1);
  if (mm) {
    cm_d4 c = {0.0, 0.0, 0.0, 0.0};
    c = (l != i) ? blk_mfma_nt(c, Xi, Xl, rb, cb, lane) : blk_mfma_nt(c, Xi, Xi, rb, cb, lane);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int rr = BLK_ROW(rb, lane, reg), cc = BLK_COL(cb, lane);
      const double o = oprev[reg] - c[reg];
      if (diag) Pi[rr][cc] = o; else out[(size_t)rr * ld + cc] = o;
    }
  }
  if (diag) {
    __syncthreads();                                 // Pi holds the updated SPD tile; Lf store of this workgroup is issued
    if (threadIdx.x >= 256) return;
    tile_linv_reg(Pi, Li, flag);
    double* dst = Dinv + (size_t)(j + 1) * TS * TS;
    for (int q = threadIdx.x; q < TS * TS; q += 256) dst[(size_t)(q / TS) * TS + (q % TS)] = Li[q / TS][q % TS];
  }
}

// ------------------------------------------------------------------------------------------------
// X = L^-1 by recursive doubling.  Level with half-size b tiles, pair q = rows/cols [2qb, 2qb+2b):
//    T   = L21 * X11        (stored in the unused strictly-upper tile (c, j2) of A)
//    X21 = -X22 * T
// ------------------------------------------------------------------------------------------------
template <bool BATCH>
__global__ __launch_bounds__(256) void k_inv_seed(cm_parg<BATCH> pa, int na_arg, int skip_own) {
  CM_SOLVE_PROLOGUE(2)
  if ((int)blockIdx.x >= nblk) return;
  const double* __restrict__ Dinv = p.Dinv;
  double* __restrict__ X = p.X;
  const int d = blockIdx.x;
  double r[4];
  tile_fetch(r, Dinv + (size_t)d * TS * TS, TS);
  double* dst = X + (size_t)d * TS * ld + (size_t)d * TS;
#pragma unroll
  for (int q = 0; q < 4; ++q) { const int i = threadIdx.x + 256 * (q >> 1); dst[(size_t)(i / (TS / 2)) * ld + 2 * (i % (TS / 2)) + (q & 1)] = r[q]; }   // (tile_fetch's element map)
}

template <bool BATCH>
__global__ __launch_bounds__(256) void k_inv_T(cm_parg<BATCH> pa, int na_arg, int skip_own, int b) {
  int bx = (int)blockIdx.x, circ = 0;
  if (BATCH) cm_xcd_remap(bx, circ, skip_own >> 8);  // the tile rows / columns the workgroups of a circuit share: one L2
  CM_SOLVE_PROLOGUE_AT(2, pa.get(circ))
  double* __restrict__ A = p.A;
  const double* __restrict__ Lf = p.Lf;
  const double* __restrict__ X = p.X;
  __shared__ __attribute__((aligned(16))) double Lt[TS][TP], Xt[TS][TPN];
  const int per = b * b;
  const int q = bx / per, rem = bx % per;
  const int j2 = (2 * q + 1) * b + rem / b;     // tile row in the second half
  const int c = 2 * q * b + rem % b;            // tile col in the first half
  if (j2 >= nblk) return;
  const int wave = threadIdx.x / CM_WAVE, lane = threadIdx.x % CM_WAVE, rb = wave >> 1, cb = wave & 1;
  cm_d4 acc = {0.0, 0.0, 0.0, 0.0};
  const int jend = (2 * q + 1) * b;                  // X11[j1][c] is non-zero only for j1 >= c
  double ra[4], rx[4];
  tile_fetch(ra, Lf + (size_t)j2 * TS * ld + (size_t)c * TS, ld);
  tile_fetch(rx, X + (size_t)c * TS * ld + (size_t)c * TS, ld);
  for (int j1 = c; j1 < jend; ++j1) {
    __syncthreads();
    tile_put(Lt, ra);
    tile_put<256, TPN>(Xt, rx);
    __syncthreads();
    if (j1 + 1 < jend) {
      tile_fetch(ra, Lf + (size_t)j2 * TS * ld + (size_t)(j1 + 1) * TS, ld);
      tile_fetch(rx, X + (size_t)(j1 + 1) * TS * ld + (size_t)c * TS, ld);
    }
    acc = blk_mfma_nn(acc, Lt, Xt, rb, cb, lane);
  }
  double* out = A + (size_t)c * TS * ld + (size_t)j2 * TS;    // scratch: strictly-upper tile (c, j2)
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) out[(size_t)BLK_ROW(rb, lane, reg) * ld + BLK_COL(cb, lane)] = acc[reg];
}

template <bool BATCH>
__global__ __launch_bounds__(256) void k_inv_X21(cm_parg<BATCH> pa, int na_arg, int skip_own, int b) {
  int bx = (int)blockIdx.x, circ = 0;
  if (BATCH) cm_xcd_remap(bx, circ, skip_own >> 8);
  CM_SOLVE_PROLOGUE_AT(2, pa.get(circ))
  const double* __restrict__ A = p.A;
  double* __restrict__ X = p.X;
  __shared__ __attribute__((aligned(16))) double Xa[TS][TP], Tt[TS][TPN];
  const int per = b * b;
  const int q = bx / per, rem = bx % per;
  const int i = (2 * q + 1) * b + rem / b;
  const int c = 2 * q * b + rem % b;
  if (i >= nblk) return;
  const int wave = threadIdx.x / CM_WAVE, lane = threadIdx.x % CM_WAVE, rb = wave >> 1, cb = wave & 1;
  cm_d4 acc = {0.0, 0.0, 0.0, 0.0};
  const int jbeg = (2 * q + 1) * b;                  // X22[i][j2] is non-zero only for j2 <= i
  double ra[4], rx[4];
  tile_fetch(ra, X + (size_t)i * TS * ld + (size_t)jbeg * TS, ld);
  tile_fetch(rx, A + (size_t)c * TS * ld + (size_t)jbeg * TS, ld);
  for (int j2 = jbeg; j2 <= i; ++j2) {
    __syncthreads();
    tile_put(Xa, ra);
    tile_put<256, TPN>(Tt, rx);
    __syncthreads();
    if (j2 + 1 <= i) {
      tile_fetch(ra, X + (size_t)i * TS * ld + (size_t)(j2 + 1) * TS, ld);
      tile_fetch(rx, A + (size_t)c * TS * ld + (size_t)(j2 + 1) * TS, ld);
    }
    acc = blk_mfma_nn(acc, Xa, Tt, rb, cb, lane);
  }
  double* out = X + (size_t)i * TS * ld + (size_t)c * TS;
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) out[(size_t)BLK_ROW(rb, lane, reg) * ld + BLK_COL(cb, lane)] = -acc[reg];
}

// The two products of a level on 64 x 64 SUPER-TILES (levels b >= 2: the b x b blocks of tiles split into 2 x 2 groups): a workgroup computes
// four output tiles, a wavefront one 32 x 32 tile as four 16 x 16 matrix-core blocks -- every operand fragment read from LDS feeds two
// products (one LDS read per matrix instruction where the one-tile kernels above need two, which put them at the LDS port's limit), the four
// operand tiles of a step are fetched once for four tile products (8 KB through the CU's fetch path per product instead of 16), and a
// wavefront issues 32 matrix instructions between two barriers instead of 8.  Every 16 x 16 block still accumulates its products in the
// same order over the same k range (a tile whose k range starts later -- the triangular operand -- skips the steps before it): bitwise the
// X of the one-tile kernels.
template <bool BATCH>
__global__ __launch_bounds__(256) void k_inv_T2(cm_parg<BATCH> pa, int na_arg, int skip_own, int b) {
  int bx = (int)blockIdx.x, circ = 0;
  if (BATCH) cm_xcd_remap(bx, circ, skip_own >> 8);
  CM_SOLVE_PROLOGUE_AT(2, pa.get(circ))
  double* __restrict__ A = p.A;
  const double* __restrict__ Lf = p.Lf;
  const double* __restrict__ X = p.X;
  __shared__ __attribute__((aligned(16))) double Lt[2][TS][TP], Xt[2][TS][TPN];
  const int hb = b >> 1, per = hb * hb;
  const int q = bx / per, rem = bx % per;
  const int J2 = (2 * q + 1) * b + 2 * (rem / hb);   // first of the two tile rows (second half of the pair)
  const int C = 2 * q * b + 2 * (rem % hb);          // first of the two tile columns (first half)
  if (J2 >= nblk) return;
  const int wave = threadIdx.x / CM_WAVE, lane = threadIdx.x % CM_WAVE, wr = wave >> 1, wc = wave & 1;
  const int j2 = J2 + wr, c = C + wc;
  const bool valid = j2 < nblk;
  const int J2b = min(J2 + 1, nblk - 1);             // (a missing second tile row re-reads the first: loaded, never used)
  const int jend = (2 * q + 1) * b;                  // X11[j1][c] is non-zero only for j1 >= c
  cm_d4 c00 = {0.0, 0.0, 0.0, 0.0}, c01 = c00, c10 = c00, c11 = c00;
  double ra[2][4], rx[2][4];
  auto fetch = [&](int j1) {
    tile_fetch(ra[0], Lf + (size_t)J2 * TS * ld + (size_t)j1 * TS, ld);
    tile_fetch(ra[1], Lf + (size_t)J2b * TS * ld + (size_t)j1 * TS, ld);
    tile_fetch(rx[0], X + (size_t)j1 * TS * ld + (size_t)C * TS, ld);
    tile_fetch(rx[1], X + (size_t)j1 * TS * ld + (size_t)(C + 1) * TS, ld);   // (j1 == C: a strictly-upper tile -- loaded, never used)
  };
  fetch(C);
  const int r = lane & 15, kk = lane >> 4;
  for (int j1 = C; j1 < jend; ++j1) {
    __syncthreads();
    tile_put(Lt[0], ra[0]); tile_put(Lt[1], ra[1]);
    tile_put<256, TPN>(Xt[0], rx[0]); tile_put<256, TPN>(Xt[1], rx[1]);
    __syncthreads();
    if (j1 + 1 < jend) fetch(j1 + 1);
    if (valid && j1 >= c) {
      const double (*Aw)[TP] = Lt[wr];
      const double (*Bw)[TPN] = Xt[wc];
#pragma unroll
      for (int ks = 0; ks < TS / 4; ++ks) {
        const double a0 = Aw[r][4 * ks + kk], a1 = Aw[16 + r][4 * ks + kk], b0 = Bw[4 * ks + kk][r], b1 = Bw[4 * ks + kk][16 + r];
        c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c00, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, c01, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, c10, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c11, 0, 0, 0);
      }
    }
  }
  if (!valid) return;
  double* out = A + (size_t)c * TS * ld + (size_t)j2 * TS;    // scratch: strictly-upper tile (c, j2)
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    out[(size_t)BLK_ROW(0, lane, reg) * ld + BLK_COL(0, lane)] = c00[reg];
    out[(size_t)BLK_ROW(0, lane, reg) * ld + BLK_COL(1, lane)] = c01[reg];
    out[(size_t)BLK_ROW(1, lane, reg) * ld + BLK_COL(0, lane)] = c10[reg];
    out[(size_t)BLK_ROW(1, lane, reg) * ld + BLK_COL(1, lane)] = c11[reg];
  }
}

template <bool BATCH>
__global__ __launch_bounds__(256) void k_inv_X212(cm_parg<BATCH> pa, int na_arg, int skip_own, int b) {
  int bx = (int)blockIdx.x, circ = 0;
  if (BATCH) cm_xcd_remap(bx, circ, skip_own >> 8);
  CM_SOLVE_PROLOGUE_AT(2, pa.get(circ))
  const double* __restrict__ A = p.A;
  double* __restrict__ X = p.X;
  __shared__ __attribute__((aligned(16))) double Xa[2][TS][TP], Tt[2][TS][TPN];
  const int hb = b >> 1, per = hb * hb;
  const int q = bx / per, rem = bx % per;
  const int I = (2 * q + 1) * b + 2 * (rem / hb);
  const int C = 2 * q * b + 2 * (rem % hb);
  if (I >= nblk) return;
  const int wave = threadIdx.x / CM_WAVE, lane = threadIdx.x % CM_WAVE, wr = wave >> 1, wc = wave & 1;
  const int i = I + wr, c = C + wc;
  const bool valid = i < nblk;
  const int Ib = min(I + 1, nblk - 1);
  const int jbeg = (2 * q + 1) * b;                  // X22[i][j2] is non-zero only for j2 <= i
  const int jlast = Ib;                              // the longer of the two tile rows' k ranges
  cm_d4 c00 = {0.0, 0.0, 0.0, 0.0}, c01 = c00, c10 = c00, c11 = c00;
  double ra[2][4], rx[2][4];
  auto fetch = [&](int j2) {
    tile_fetch(ra[0], X + (size_t)I * TS * ld + (size_t)j2 * TS, ld);        // (j2 == I + 1: a strictly-upper tile -- loaded, never used)
    tile_fetch(ra[1], X + (size_t)Ib * TS * ld + (size_t)j2 * TS, ld);
    tile_fetch(rx[0], A + (size_t)C * TS * ld + (size_t)j2 * TS, ld);
    tile_fetch(rx[1], A + (size_t)(C + 1) * TS * ld + (size_t)j2 * TS, ld);
  };
  fetch(jbeg);
  const int r = lane & 15, kk = lane >> 4;
  for (int j2 = jbeg; j2 <= jlast; ++j2) {
    __syncthreads();
    tile_put(Xa[0], ra[0]); tile_put(Xa[1], ra[1]);
    tile_put<256, TPN>(Tt[0], rx[0]); tile_put<256, TPN>(Tt[1], rx[1]);
    __syncthreads();
    if (j2 + 1 <= jlast) fetch(j2 + 1);
    if (valid && j2 <= i) {
      const double (*Aw)[TP] = Xa[wr];
      const double (*Bw)[TPN] = Tt[wc];
#pragma unroll
      for (int ks = 0; ks < TS / 4; ++ks) {
        const double a0 = Aw[r][4 * ks + kk], a1 = Aw[16 + r][4 * ks + kk], b0 = Bw[4 * ks + kk][r], b1 = Bw[4 * ks + kk][16 + r];
        c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c00, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, c01, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, c10, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c11, 0, 0, 0);
      }
    }
  }
  if (!valid) return;
  double* out = X + (size_t)i * TS * ld + (size_t)c * TS;
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    out[(size_t)BLK_ROW(0, lane, reg) * ld + BLK_COL(0, lane)] = -c00[reg];
    out[(size_t)BLK_ROW(0, lane, reg) * ld + BLK_COL(1, lane)] = -c01[reg];
    out[(size_t)BLK_ROW(1, lane, reg) * ld + BLK_COL(0, lane)] = -c10[reg];
    out[(size_t)BLK_ROW(1, lane, reg) * ld + BLK_COL(1, lane)] = -c11[reg];
  }
}

// ------------------------------------------------------------------------------------------------
// mu = X^T (X rhs), beta = colsum(X^2)  (X = L^-1: Sigma = X^T X, caviar.py:170-172) in ONE pass over X.
//
// One 1024-thread workgroup per HALF tile row h (16 rows of X, columns through the diagonal tile): thread t holds column t of the 16 rows in
// registers (a slab of 1024 columns; matrices beyond 1024 rows walk their slabs twice).  Phase A: w_r = sum_c X[r][c] rhs[c].  The 16 row
// sums of a wavefront come out of ONE transposing butterfly instead of 16 reductions: at every step a lane hands the half of its values
// that its partner keeps to that partner and adds what it receives (DPP: mirror within 16, mirror within 8, xor 1, xor 2: 8 + 4 + 2 + 1
// exchanges), after which lane l of every 16-lane row holds the sum of row mb_row(l) over its 16 lanes; two cross-row adds finish the 64
// lanes, the 16 wave partials are added in wave order.  Phase B, from the same registers: the half tile row's contribution to
// mu_c = sum_r X[r][c] w_r and beta_c = sum_r X[r][c]^2, rows ascending, written to the factor's storage (dead by now) as partial sums per
// half tile row; k_colsum_final adds them in order.  Every summation order is fixed: results are reproducible and identical for a circuit
// alone and in a batch.  (Before: k_tri_matvec + k_colsum_partial streamed X twice -- 0.98 GB per 256 x n = 1000 solve.)
// ------------------------------------------------------------------------------------------------
#define CM_MB_THREADS 1024
#define CM_MB_ROWS 16
template <int CTRL, int HALF>
__device__ __forceinline__ void mb_step(double (&v)[CM_MB_ROWS], bool upper) {
#pragma unroll
  for (int q = 0; q < HALF; ++q) {
    const double keep = upper ? v[HALF + q] : v[q], send = upper ? v[q] : v[HALF + q];
    v[q] = keep + cm_dpp_f64<CTRL>(send);
  }
}
// row (0..15) whose sum lane l of a 16-lane row holds after the four steps: bit 3 and bit 2 from the lane's, bit 1 from lane bit 0, bit 0 from lane bit 1
__device__ __forceinline__ int mb_row(int l) { return (l & 12) | ((l & 1) << 1) | ((l >> 1) & 1); }

template <bool BATCH>
__global__ __launch_bounds__(CM_MB_THREADS, BATCH ? 8 : 4) void k_mu_beta_rows(cm_parg<BATCH> pa, int na_arg, int skip_own) {
  CM_SOLVE_PROLOGUE(2)
  const int h = 2 * nblk - 1 - (int)blockIdx.x;  // (the long rows first)
  if (h < 0) return;
  const double* __restrict__ X = p.X;
  const double* __restrict__ rhs = p.rhs;
  double* __restrict__ part = p.A;               // the factor is no longer needed: its storage takes the partial column sums [2 nblk][2][ld]
  __shared__ double wpart[CM_MB_THREADS / CM_WAVE][CM_MB_ROWS];
  __shared__ double wsh[CM_MB_ROWS];
  const int tid = threadIdx.x, wave = tid / CM_WAVE, lane = tid % CM_WAVE;
  const int r0 = h * CM_MB_ROWS, ncol = (h / 2 + 1) * TS;
  if (r0 >= na) return;                          // (rows of the identity padding: nothing to add)
  const int nslab = (ncol + CM_MB_THREADS - 1) / CM_MB_THREADS;
  double x[CM_MB_ROWS];
  auto load_slab = [&](int sl) {
    const int c = sl * CM_MB_THREADS + tid;
    const double* src = X + (size_t)r0 * ld + c;
#pragma unroll
    for (int rr = 0; rr < CM_MB_ROWS; ++rr) x[rr] = (c <= r0 + rr && r0 + rr < na) ? src[(size_t)rr * ld] : 0.0;   // lower triangle, rows of the active set
  };
  // ---- phase A: w_r = sum_c X[r][c] rhs[c]
  for (int sl = 0; sl < nslab; ++sl) {
    const int c = sl * CM_MB_THREADS + tid;
    double tot = 0.0;
    if (sl * CM_MB_THREADS + wave * CM_WAVE < ncol) {            // (wave-uniform: a wavefront whose columns lie beyond the diagonal tile adds nothing)
      load_slab(sl);
      const double rc = c < ncol ? rhs[c] : 0.0;
      double v[CM_MB_ROWS];
#pragma unroll
      for (int rr = 0; rr < CM_MB_ROWS; ++rr) v[rr] = x[rr] * rc;
      mb_step<CM_DPP_MIRROR, 8>(v, (lane & 8) != 0);
      mb_step<CM_DPP_HALF_MIRROR, 4>(v, (lane & 4) != 0);
      mb_step<CM_DPP_XOR1, 2>(v, (lane & 1) != 0);
      mb_step<CM_DPP_XOR2, 1>(v, (lane & 2) != 0);
      tot = v[0];
      tot += __shfl_xor(tot, 16, CM_WAVE);
      tot += __shfl_xor(tot, 32, CM_WAVE);
    }
    if (lane < CM_MB_ROWS) wpart[wave][mb_row(lane)] = (sl == 0) ? tot : wpart[wave][mb_row(lane)] + tot;
  }
  __syncthreads();
  if (tid < CM_MB_ROWS) {
    double w = 0.0;
#pragma unroll
    for (int wv = 0; wv < CM_MB_THREADS / CM_WAVE; ++wv) w += wpart[wv][tid];
    wsh[tid] = w;
  }
  __syncthreads();
  // ---- phase B: this half tile row's share of the column sums (one slab: straight from the registers of phase A)
  for (int sl = 0; sl < nslab; ++sl) {
    if (sl * CM_MB_THREADS + wave * CM_WAVE >= ncol) continue;
    if (nslab > 1) load_slab(sl);
    const int c = sl * CM_MB_THREADS + tid;
    double sm = 0.0, sb = 0.0;
#pragma unroll
    for (int rr = 0; rr < CM_MB_ROWS; ++rr) { sm = fma(x[rr], wsh[rr], sm); sb = fma(x[rr], x[rr], sb); }
    if (c < ncol) {
      part[((size_t)h * 2 + 0) * ld + c] = sm;
      part[((size_t)h * 2 + 1) * ld + c] = sb;
    }
  }
}
// 16 lanes per column: lane l adds the half tile rows h0 + l, h0 + l + 16, ... in order, the 16 partial sums are combined by the fixed DPP tree
// (one thread per column walked up to 63 dependent loads: 13.7 us for one circuit at n = 1000)
template <bool BATCH>
__global__ __launch_bounds__(256) void k_colsum_final(cm_parg<BATCH> pa, int na_arg, int skip_own) {
  CM_SOLVE_PROLOGUE(2)
  const double* __restrict__ part = p.A;
  const int n = na;
  const int c = (int)blockIdx.x * 16 + (int)(threadIdx.x >> 4), l = threadIdx.x & 15;
  const bool on = c < n;
  const int cc = on ? c : 0;
  const int hend = (n + CM_MB_ROWS - 1) / CM_MB_ROWS;     // half tile rows that hold rows of the active set
  double sm = 0.0, sb = 0.0;
  if (on)
    for (int h = (cc / TS) * 2 + l; h < hend; h += 16) {   // half tile rows that hold column c
      sm += part[((size_t)h * 2 + 0) * ld + cc];
      sb += part[((size_t)h * 2 + 1) * ld + cc];
    }
  sm = cm_seg_sum<16>(sm); sb = cm_seg_sum<16>(sb);
  if (on && l == 0) {
    const int nn = p.act_idx[cc];
    p.mu[nn] = sm;
    p.beta[nn] = sb;
  }
}

template <bool BATCH>
__global__ __launch_bounds__(256) void k_fill_prior(cm_parg<BATCH> pa, int na_arg, int skip_own) {
  CM_SOLVE_PROLOGUE(2)
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < p.N) { p.mu[n] = p.mu_prior[n]; const double b0 = p.beta_prior[n]; p.beta[n] = b0 * b0; }
}

// stand-alone launch of the small solve (per-phase test hook: the same body the fused mid-iteration kernel runs for small sets)
template <bool BATCH>
__global__ __launch_bounds__(1024) void k_solve_small(cm_parg<BATCH> pa, int na_arg, int skip_own) {
  extern __shared__ double sm[];
  CM_SOLVE_PROLOGUE(1)
  solve_small_body(sm, p.lam, p.row_ptr, p.col, p.act_idx, p.rs, p.scal, p.mu_prior, p.beta_prior, na, p.N, na * p.max_row_len, p.mu, p.beta,
                   p.n_act + 1, (int)blockDim.x);
}

// Expects cm_launch_stats() + a compaction (k_post_single or k_iter_mid) to have been enqueued after the last change of lam.
int cm_solve_mu(cm_ctx* ctx, cm_set& s, int* n_active_out) {
  CM_TRY(cm_wait_active(ctx, s));                   // the only host wait of an iteration: size of the active set
  if (n_active_out) *n_active_out = s.na_max;
  return cm_solve_launch(ctx, s, -1);
}

// Which family factors matrices above `large_min` tile rows: batches take the grouped one (throughput: 26 against 19 TFLOP/s at 128 x
// n = 1000); one large matrix is bound by the chain of diagonal blocks, which the pair family walks in slightly shorter steps (cfg5: 11.0
// against 12.0 ms per fit), so it stays there.  CAVIAR_HIP_CHOL_FAMILY: 0 = pairs everywhere, 1 = this default, 2 = grouped everywhere.
static inline int cm_large_tiles() { static const int v = getenv("CAVIAR_HIP_LARGE_TILES") ? atoi(getenv("CAVIAR_HIP_LARGE_TILES")) : CM_LARGE_TILES; return v; }   // (diagnostic)
static inline bool cm_use_grouped(const cm_ctx* ctx, bool single) { return ctx->chol_family == 2 || (ctx->chol_family == 1 && !single); }

// One group of circuits through the tiled solve; `parts`: 1 = precision matrices, 2 = Cholesky, 4 = inverse of the factor, 8 = mu / beta.
struct cm_solve_sub { int B; bool single; cm_pdev h0; const cm_pdev* d_arr; long long nnz_max = 0; };
static int cm_solve_group_launch(cm_ctx* ctx, const cm_solve_sub& g, int nblk, int na_g, int na1, int skip_le, int xflags, int large_min, int parts,
                                 hipEvent_t fact_begin, hipEvent_t fact_end) {
  hipStream_t st = ctx->stream;
  const bool large_g = nblk > large_min;
  const int ld = nblk * TS;                        // rows (the row stride is ld + CM_LD_PAD)
  const unsigned gy = g.single ? 1u : (unsigned)g.B;
#define CM_LAUNCH_G(KERN, gridx, block, lds, ...)                                                                                           \
  do {                                                                                                                                    \
    if (g.single) hipLaunchKernelGGL((KERN<false>), dim3(gridx), dim3(block), lds, st, cm_parg<false>{g.h0}, ##__VA_ARGS__);                \
    else hipLaunchKernelGGL((KERN<true>), dim3(gridx, gy), dim3(block), lds, st, cm_parg<true>{g.d_arr}, ##__VA_ARGS__);                    \
  } while (0)
#define CM_LAUNCH_GT(KERN, TARG, gridx, block, lds, ...)                                                                                    \
  do {                                                                                                                                    \
    if (g.single) hipLaunchKernelGGL((KERN<false, TARG>), dim3(gridx), dim3(block), lds, st, cm_parg<false>{g.h0}, ##__VA_ARGS__);          \
    else hipLaunchKernelGGL((KERN<true, TARG>), dim3(gridx, gy), dim3(block), lds, st, cm_parg<true>{g.d_arr}, ##__VA_ARGS__);              \
  } while (0)
  if (parts & 1) {
    // Row a accumulates columns up to the end of its diagonal tile only, and the kernel's wavefronts per CU are set by the LDS row they hold: a
    // batch launches the rows in four bands, each with the accumulator its last row needs (5 / 7 / 9 / 11 KB instead of 11 for all at n = 1000:
    // 20 instead of 14 wavefronts per CU on average -- the kernel is a chain of dependent gathers, its time is its wavefronts over its occupancy)
    CM_LAUNCH_G(k_bp_prepare, (unsigned)((std::max<long long>(g.nnz_max, na_g) + 255) / 256), 256, 0, na1, skip_le);
    const int bands = (!g.single && ld >= 16 * TS) ? 4 : 1;
    for (int q = 0; q < bands; ++q) {
      const int r0 = ((ld / TS) * q / bands) * TS, r1 = ((ld / TS) * (q + 1) / bands) * TS;
      if (r1 > r0) CM_LAUNCH_G(k_build_precision, r1 - r0, 64, build_prec_lds(r1 + CM_LD_PAD), na1, skip_le, r0, r1 + CM_LD_PAD);
    }
  }
  if (parts & 2) {
    if (fact_begin) CM_HIP(hipEventRecord(fact_begin, st));
    if (large_g && cm_use_grouped(ctx, g.single)) {    // grouped family: four panels per pass over the trailing matrix, 64 x 64 super-tiles
      if (g.single) {
        CM_TRY(cm_raise_lds(ctx, CM_LDS_CHOL_DIAG, reinterpret_cast<const void*>(k_chol_diag<false>), CM_GROUP_LDS, 48 * 1024));
        CM_TRY(cm_raise_lds(ctx, CM_LDS_CHOL_GPANEL, reinterpret_cast<const void*>(k_chol_gpanel<false>), CM_GROUP_LDS, 48 * 1024));
        CM_TRY(cm_raise_lds(ctx, CM_LDS_CHOL_WIDE, reinterpret_cast<const void*>(k_chol_wide<false>), CM_WIDE_LDS, 48 * 1024));
      } else {
        CM_TRY(cm_raise_lds(ctx, CM_LDS_CHOL_DIAG_B, reinterpret_cast<const void*>(k_chol_diag<true>), CM_GROUP_LDS, 48 * 1024));
        CM_TRY(cm_raise_lds(ctx, CM_LDS_CHOL_GPANEL_B, reinterpret_cast<const void*>(k_chol_gpanel<true>), CM_GROUP_LDS, 48 * 1024));
        CM_TRY(cm_raise_lds(ctx, CM_LDS_CHOL_WIDE_B, reinterpret_cast<const void*>(k_chol_wide<true>), CM_WIDE_LDS, 48 * 1024));
      }
      CM_LAUNCH_G(k_chol_diag, 1, 256, CM_GROUP_LDS, na1, skip_le, 0, large_min);
      for (int j0 = 0; j0 + CM_GRP < nblk; j0 += CM_GRP) {
        const int t = nblk - j0 - CM_GRP, ns = (t + 1) / 2, nst = ns * (ns + 1) / 2;
        CM_LAUNCH_G(k_chol_gpanel, t, 256, CM_GROUP_LDS, na1, xflags, j0, large_min);
        CM_LAUNCH_G(k_chol_wide, ns >= 2 ? nst - 2 : 1, 256, CM_WIDE_LDS, na1, xflags, j0, large_min);
      }
    } else if (large_g) {                              // panels in pairs: panel solve, thin column, panel solve, rank-64 trailing update
      CM_LAUNCH_G(k_chol_first, 1, 256, 0, na1, skip_le);
      for (int j = 0; j + 1 < nblk; j += 2) {
        const int t = nblk - j - 1;
        CM_LAUNCH_GT(k_chol_bulk, false, t, 256, 0, na1, xflags, j, large_min);        // (carries the panel solve of column j)
        if (t < 2) break;
        CM_LAUNCH_G(k_chol_trsm, t - 1, 256, 0, na1, skip_le, j + 1, large_min);
        CM_LAUNCH_GT(k_chol_bulk, true, (t - 1) * t / 2, 256, 0, na1, xflags, j, large_min);
      }
    } else {                                           // small / medium (one circuit): one launch per panel
      CM_LAUNCH_G(k_chol_first, 1, 256, 0, na1, skip_le);
      for (int j = 0; j + 1 < nblk; ++j) {
        const int tp = nblk - j - 1;
        CM_LAUNCH_G(k_chol_panel, tp * (tp + 1) / 2, 1024, 0, na1, skip_le, j, large_min);
      }
    }
    if (fact_end) CM_HIP(hipEventRecord(fact_end, st));   // after EVERY group that factorised: the last record stands
  }
  if (parts & 4) {
    // inverse of the factor (strictly-upper tiles of X are never read: no clearing needed)
    static const bool inv_super = getenv("CAVIAR_HIP_INV_ONE_TILE") == nullptr;   // (diagnostic / test hook: the one-tile kernels at every level)
    CM_LAUNCH_G(k_inv_seed, nblk, 256, 0, na1, skip_le);
    for (int b = 1; b < nblk; b *= 2) {
      const int npairs = (nblk + 2 * b - 1) / (2 * b);
      if (b >= 2 && inv_super && !g.single) {          // batches: 64 x 64 super-tiles (bitwise the X of the one-tile kernels; one circuit keeps the one-tile
                                                       // kernels -- four times the workgroups with shorter chains: 5.9 against 12.4 us per launch)
        CM_LAUNCH_G(k_inv_T2, npairs * (b / 2) * (b / 2), 256, 0, na1, xflags, b);
        CM_LAUNCH_G(k_inv_X212, npairs * (b / 2) * (b / 2), 256, 0, na1, xflags, b);
      } else {
        CM_LAUNCH_G(k_inv_T, npairs * b * b, 256, 0, na1, xflags, b);
        CM_LAUNCH_G(k_inv_X21, npairs * b * b, 256, 0, na1, xflags, b);
      }
    }
  }
  if (parts & 8) {
    CM_LAUNCH_G(k_mu_beta_rows, 2 * nblk, CM_MB_THREADS, 0, na1, skip_le);
    CM_LAUNCH_G(k_colsum_final, (na_g + 15) / 16, 256, 0, na1, skip_le);
  }
#undef CM_LAUNCH_G
#undef CM_LAUNCH_GT
  CM_HIP(hipGetLastError());
  return CM_OK;
}

// Measurement aid (include/caviar_hip.h): B synthetic SPD matrices of n rows through the tiled Cholesky (and, separately timed, the
// inverse of the factor) exactly as a lock-step batch of B circuits with n active rows each launches them; no CAVIaR state involved.
__global__ __launch_bounds__(256) void k_bench_fill_spd(double* __restrict__ A, int n, int ld, double shift) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)ld * ld) return;
  const int r = (int)(i / ld), c = (int)(i % ld);
  double v = 0.0;
  if (r < n && c < n) { const int lo = r < c ? r : c, hi = r < c ? c : r; v = (r == c) ? 2.0 + shift + 0.001 * (r % 7) : (0.1 + (double)((lo * 7 + hi * 13) % 11) / 11.0) / (double)n; }   // diagonally dominant
  else if (r == c) v = 1.0;
  A[i] = v;
}
// Measurement aid: the fp64 matrix-core rate this device sustains with operands in registers and no memory traffic -- four independent
// accumulation chains of v_mfma_f64_16x16x4_f64 per wavefront, eight wavefronts per SIMD (tools/micro/mfma_peak.hip is the stand-alone form;
// profiles/r04_mfma_peak.txt).  On the MI355X boxes of this pool: 48.8 TFLOP/s, 62 % of the 78.6 TFLOP/s the guide lists (the vector fp64 FMA
// pipe reaches 69.6) -- the practical roof of every tile-product kernel above; bench.py reports it beside the nominal peak.
__global__ __launch_bounds__(256) void k_mfma_peak(double* out, int iters, double a0, double b0) {
  cm_d4 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = cm_d4{0.0, 0.0, 0.0, 0.0};
  const double a = a0 + threadIdx.x * 1e-9, b = b0 + threadIdx.x * 1e-9;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
  }
  double sum = 0.0;
#pragma unroll
  for (int c = 0; c < 4; ++c) sum += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  if (sum == 12345.678) out[0] = sum;                // (never true: keeps the chains alive)
}
extern "C" int cm_k_mfma_peak(cm_ctx* ctx, int32_t iters, double* tflops) {
  if (!ctx || !tflops || iters < 1) { cm_set_error("cm_k_mfma_peak: bad argument"); return CM_ERR_INVALID; }
  CM_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  double* d = nullptr;
  CM_HIP(hipMalloc(&d, 64));
  const int blocks = ctx->num_cus * 8;
  float ms = 0.0f;
  for (int r = 0; r < 2; ++r) {                      // (first repetition: warm-up)
    hipEventRecord(ctx->ev[0], st);
    hipLaunchKernelGGL(k_mfma_peak, dim3(blocks), dim3(256), 0, st, d, iters, 1.0, 1.0);
    hipEventRecord(ctx->ev[1], st);
    if (hipStreamSynchronize(st) != hipSuccess) { hipFree(d); cm_set_error("cm_k_mfma_peak: %s", hipGetErrorString(hipGetLastError())); return CM_ERR_HIP; }
    hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]);
  }
  hipFree(d);
  *tflops = (double)blocks * 4.0 * 4.0 * (double)iters * 2048.0 / ((double)ms * 1e-3) / 1e12;
  return CM_OK;
}

extern "C" int cm_k_solve_bench(cm_ctx* ctx, int32_t n, int32_t B, int32_t reps, float* ms_chol, float* ms_inv) {
  if (!ctx || n < 1 || B < 1 || B > CM_MAX_BATCH || reps < 1 || !ms_chol || !ms_inv) { cm_set_error("cm_k_solve_bench: bad argument"); return CM_ERR_INVALID; }
  CM_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int nblk = (n + TS - 1) / TS, ld = nblk * TS + CM_LD_PAD;
  const size_t msz = (size_t)(nblk * TS) * ld;
  std::vector<cm_pdev> h(B);
  std::vector<void*> owned;
  auto alloc = [&](size_t bytes) -> void* { void* q = nullptr; if (hipMalloc(&q, bytes) != hipSuccess) return nullptr; owned.push_back(q); return q; };
  int rc = CM_OK;
  for (int c = 0; c < B && rc == CM_OK; ++c) {
    cm_pdev& d = h[c];
    ::memset(&d, 0, sizeof(d));
    d.N = n; d.small_cap = 0; d.ld_alloc = nblk * TS;
    d.A = (double*)alloc(msz * 8); d.X = (double*)alloc(msz * 8); d.Lf = (double*)alloc(msz * 8);
    d.Dinv = (double*)alloc((size_t)nblk * TS * TS * 8); d.n_act = (int*)alloc(16);
    if (!d.A || !d.X || !d.Lf || !d.Dinv || !d.n_act) { cm_set_error("cm_k_solve_bench: out of memory"); rc = CM_ERR_HIP; break; }
    const int na4[4] = {n, 0, 0, 0};
    if (hipMemcpy(d.n_act, na4, 16, hipMemcpyHostToDevice) != hipSuccess) rc = CM_ERR_HIP;
  }
  cm_pdev* d_arr = nullptr;
  if (rc == CM_OK && B > 1) {
    d_arr = (cm_pdev*)alloc(sizeof(cm_pdev) * B);
    if (!d_arr || hipMemcpy(d_arr, h.data(), sizeof(cm_pdev) * B, hipMemcpyHostToDevice) != hipSuccess) rc = CM_ERR_HIP;
  }
  double tc = 0.0, ti = 0.0;
  for (int r = 0; r < reps + 1 && rc == CM_OK; ++r) {          // (first repetition: warm-up)
    for (int c = 0; c < B; ++c) hipLaunchKernelGGL(k_bench_fill_spd, dim3((unsigned)((msz + 255) / 256)), dim3(256), 0, st, h[c].A, n, ld, 0.01 * c);
    cm_solve_sub g;
    g.B = B; g.single = B == 1; g.h0 = h[0]; g.d_arr = d_arr;
    const int large_min = B > 1 ? 1 : cm_large_tiles();
    const int xflags = 0 | (ctx->xcd_mode << 8);
    rc = cm_solve_group_launch(ctx, g, nblk, n, B == 1 ? n : 0, 0, xflags, large_min, 2, ctx->ev[0], ctx->ev[1]);
    if (rc == CM_OK) rc = cm_solve_group_launch(ctx, g, nblk, n, B == 1 ? n : 0, 0, xflags, large_min, 4, nullptr, nullptr);
    if (rc != CM_OK) break;
    if (hipEventRecord(ctx->ev[2], st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { cm_set_error("cm_k_solve_bench: %s", hipGetErrorString(hipGetLastError())); rc = CM_ERR_HIP; break; }
    float a = 0, b = 0;
    hipEventElapsedTime(&a, ctx->ev[0], ctx->ev[1]);
    hipEventElapsedTime(&b, ctx->ev[1], ctx->ev[2]);
    if (r > 0) { tc += a; ti += b; }
  }
  if (rc == CM_OK) {                                           // the factor must have been positive definite
    int flag[4] = {0, 0, 0, 0};
    if (hipMemcpy(flag, h[0].n_act, 16, hipMemcpyDeviceToHost) == hipSuccess && flag[1]) { cm_set_error("cm_k_solve_bench: test matrix not positive definite"); rc = CM_ERR_NOT_SPD; }
  }
  hipStreamSynchronize(st);
  for (void* q : owned) hipFree(q);
  *ms_chol = (float)(tc / reps); *ms_inv = (float)(ti / reps);
  return rc;
}

// block_update_mu for active sets of known sizes (s.na): small sets in one workgroup each, larger ones by the tiled factorisation.
// skip_le: circuits whose active set is no larger than this were already solved (by the fused mid-iteration kernel) and are left
// alone; -1 = solve every circuit of the set.  In a batch each circuit reads its own size on the device; the host only sizes the
// grids for the largest one.
int cm_solve_launch(cm_ctx* ctx, cm_set& s, int skip_mode, hipEvent_t fact_begin, hipEvent_t fact_end) {
  hipStream_t st = ctx->stream;
  s.last_fact = false;
  // what the set needs
  // Two families of trailing-update kernels with bitwise the same results: one launch per panel, every workgroup re-deriving the
  // panel solve (shortest dependent chain: one circuit, up to CM_LARGE_TILES tile rows), or panel solve once + lean 256-thread
  // trailing update (least traffic per tile: large matrices, and every batch -- there the launches are throughput-bound).
  const int large_min = s.B > 1 ? 1 : cm_large_tiles();
  bool any_small = false, any_tiled = false, any_panel = false, any_large = false;
  int na_small = 0, nblk_max = 0, nblk_panel = 0, na_tiled = 0;
  size_t lds_small = 0;
  for (int c = 0; c < s.B; ++c) {
    const cm_problem* q = s.ps[c];
    const int na = s.na[c];
    const int kind = cm_solve_kind(na, q->small_cap, skip_mode >= 0);
    if (kind == 1) { any_small = true; na_small = std::max(na_small, na); lds_small = std::max(lds_small, small_lds_bytes(na, na * q->max_row_len)); }
    if (kind == 2) {
      any_tiled = true;
      const int nb = (na + TS - 1) / TS;
      nblk_max = std::max(nblk_max, nb); na_tiled = std::max(na_tiled, na);
      if (nb > large_min) any_large = true; else { any_panel = true; nblk_panel = std::max(nblk_panel, nb); }
    }
  }
  const int skip_le = skip_mode < 0 ? 0 : 1;                   // device side: 1 = leave circuits at or below their own small_cap alone
  const int na1 = s.B == 1 ? s.na[0] : 0;
  const int N = s.lead->N;
  if (any_small) {
    if (s.B == 1) CM_TRY(cm_raise_lds(ctx, CM_LDS_SOLVE_SMALL, reinterpret_cast<const void*>(k_solve_small<false>), lds_small, 48 * 1024));
    else CM_TRY(cm_raise_lds(ctx, CM_LDS_SOLVE_SMALL_B, reinterpret_cast<const void*>(k_solve_small<true>), lds_small, 48 * 1024));
    CM_LAUNCH_SET(k_solve_small, s, 1, 1024, lds_small, st, na1, skip_le);
  }
  if (!any_tiled) { CM_HIP(hipGetLastError()); return CM_OK; }
  CM_LAUNCH_SET(k_fill_prior, s, (N + 255) / 256, 256, 0, st, na1, skip_le);
  if (nblk_max == 0) { CM_HIP(hipGetLastError()); return CM_OK; }
  {
    const size_t lds_bp = build_prec_lds(nblk_max * TS + CM_LD_PAD);
    if (lds_bp > 150 * 1024) { cm_set_error("active set of %d neurons exceeds the LDS row accumulator", na_tiled); return CM_ERR_INVALID; }
    if (s.B == 1) CM_TRY(cm_raise_lds(ctx, CM_LDS_BUILD_PREC, reinterpret_cast<const void*>(k_build_precision<false>), lds_bp, 48 * 1024));
    else CM_TRY(cm_raise_lds(ctx, CM_LDS_BUILD_PREC_B, reinterpret_cast<const void*>(k_build_precision<true>), lds_bp, 48 * 1024));
  }
  s.last_fact = true;
  for (int c = 0; c < s.B; ++c) {                  // bookkeeping for cm_timings: trailing-update passes and the tiles they read and rewrote
    const int nb = (s.na[c] + TS - 1) / TS;
    if (cm_solve_kind(s.na[c], s.ps[c]->small_cap, skip_mode >= 0) != 2) continue;
    s.fact_tri_tiles += (long long)nb * (nb + 1) / 2;
    for (long long t = 0; t < nb; ++t) s.fact_products += t + t * (t + 1) / 2;     // panel j with t = nb - j - 1 rows below it: t solves, t (t + 1) / 2 updates
    for (int b = 1; b < nb; b *= 2)                                                 // recursive doubling: per pair, T = L21 X11 and X21 = -X22 T over the existing rows
      for (int q = 0; (2 * q + 1) * b < nb; ++q)
        for (int r = (2 * q + 1) * b; r < std::min(nb, (2 * q + 2) * b); ++r) {
          s.inv_products += (long long)b * (b + 1) / 2;                             // row r of T: sum over the b columns c of (b - c') steps
          s.inv_products += (long long)b * (r - (2 * q + 1) * b + 1);               // row r of X21: (r - jbeg + 1) steps for each of the b columns
        }
    if (nb > large_min && cm_use_grouped(ctx, s.B == 1)) {
      for (int j0 = 0; j0 + CM_GRP < nb; j0 += CM_GRP) {   // group: panel solve of t rows x 4 columns, rank-128 update of the t (t + 1) / 2 trailing tiles
        const long long t = nb - j0 - CM_GRP;
        s.fact_tile_updates += t * (t + 1) / 2;
        s.fact_panel_tiles += 12 * t + 10;             // k_chol_gpanel: 4 t unsolved tiles in, 4 t solved out, the block's 10 tiles; k_chol_wide: 4 t panel tiles
      }
    } else if (nb > large_min) {
      for (int j = 0; j + 1 < nb; j += 2) {          // panel pair: two panel solves, the thin column, the rank-64 update
        const long long t = nb - j - 1;
        s.fact_tile_updates += t + (t - 1) * t / 2;
        s.fact_panel_tiles += (2 * t + 1);                                 // thin pass: t unsolved panel tiles in, t solved out, L^-1
        if (t >= 2) s.fact_panel_tiles += (2 * (t - 1) + 1) + 2 * (t - 1);  // k_chol_trsm(j + 1); rank-64 pass: both panels' tiles
      }
    } else {
      for (int j = 0; j + 1 < nb; ++j) {               // k_chol_panel: panel tiles in, solved tiles out, L^-1
        const long long t = nb - j - 1;
        s.fact_tile_updates += t * (t + 1) / 2;
        s.fact_panel_tiles += 2 * t + 1;
      }
    }
  }
  // A batch is solved in groups of `solve_group` circuits, each group from its precision matrices to mu / beta before the next one
  // starts: the matrices of a group (3 x 8 MB per circuit at N = 1000) then stay in the 256 MB Infinity Cache from one pass to the
  // next, where the lock-step batch as a whole streams ~1 GB per pass through it.
  const int G = (s.B > 1 && ctx->solve_group > 0) ? std::min(ctx->solve_group, s.B) : s.B;
  const int xflags = skip_le | (ctx->xcd_mode << 8);
  bool first_group = true;
  for (int c0 = 0; c0 < s.B; c0 += G) {
    cm_solve_sub g;
    g.B = std::min(G, s.B - c0);
    g.single = s.B == 1;
    if (s.B == 1) g.h0 = s.h0;
    g.d_arr = s.d_arr ? s.d_arr + c0 : nullptr;
    g.nnz_max = s.nnz_max;
    int nblk = 0, na_g = 0;
    for (int c = c0; c < c0 + g.B; ++c) {
      if (cm_solve_kind(s.na[c], s.ps[c]->small_cap, skip_mode >= 0) != 2) continue;
      nblk = std::max(nblk, (s.na[c] + TS - 1) / TS); na_g = std::max(na_g, s.na[c]);
    }
    if (nblk == 0) continue;
    s.fact_launches += nblk > large_min ? (cm_use_grouped(ctx, s.B == 1) ? (nblk - 1) / CM_GRP : nblk / 2) : std::max(0, nblk - 1);
    CM_TRY(cm_solve_group_launch(ctx, g, nblk, na_g, na1, skip_le, xflags, large_min, 15, first_group ? fact_begin : nullptr, fact_end));
    first_group = false;
  }
  CM_HIP(hipGetLastError());
  return CM_OK;
}
