// cm_nwd.hip -- NWD demixer forward pass (include/nwd_hip.h): the reference's NeuralDemixer.__call__
// (circuitmap/neural_waveform_demixing.py:36-54) as ONE fused kernel, one 1024-thread workgroup per trace.
//
// Every activation of the 1-D U-Net lives in LDS (119 KB at T = 900); HBM sees the trace once (float64 in) and the
// result once (float64 out).  The eight inner layers (4 strided-pool convolutions, 4 transposed convolutions) are
// implicit GEMMs on v_mfma_f32_16x16x4_f32: M = output channels (16 per block), N = 16 output positions, K = input
// channel x tap, the B operand read straight out of the LDS activation rows (im2col never materialised), the A operand
// from BatchNorm-folded, lane-packed weights that the workgroup stages through a 32 KB LDS window once per layer (each
// wavefront keeps a weight fragment in a register across its <= 4 output tiles).  Transposed convolutions run as
// ordinary correlations over zero-padded rows with flipped taps; the stride-2 one (ublock4) as two 16-tap polyphase
// branches; the last layer (1 output channel, 256 dilated taps) as a Toeplitz product so that it too fills the 16 rows of
// the matrix instruction.  Skip connections are written by the encoder directly into their slot of the decoder's
// concatenated, padded input rows.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/nwd_hip.h"
#include "cm_internal.h"

#define NWD_THREADS 1024
#define NWD_WAVES (NWD_THREADS / CM_WAVE)
#define NWD_LDS_LIMIT (160 * 1024)
#define NWD_WBUF 8192                   // floats of LDS the packed weights are staged through (32 KB)
#define NWD_WBUF_MIN 4096
#define NWD_MAXT 4                      // output tiles one wavefront carries through a layer

typedef float cm_f4 __attribute__((ext_vector_type(4)));

// lengths of every stage for a trace of T samples + LDS offsets (in floats)
struct NwdPlan {
  int T, P0, E1, P1, E2, P2, E3, P3, E4, U1, U2, U3, U4;
  int oB1, sB1;      // enc4, padded by 15:                  32 rows
  int oC1, sC1;      // [interp(ublock1) | enc3], padded 15: 48 rows
  int oC2, sC2;      // [interp(ublock2) | enc2], padded 31: 32 rows
  int oC3, sC3;      // [interp(ublock3) | enc1], padded 15: 32 rows
  int oC4, sC4;      // interp(ublock4), padded 255, split by sample parity, 16 samples per 17-float row: 8 rows (overlays C2)
  int oW, nW;        // weight staging window: nW floats (NWD_WBUF, less for long traces, never under NWD_WBUF_MIN)
  int oS;            // scratch: normalised trace + pooled rows / deconv outputs before interpolation
  int oOut;          // raw network output [T]                        (overlays C1)
  int oScan;         // float64 scan buffer [2][NWD_THREADS], as a float offset (overlays B1/C1 tail, 8-byte aligned)
  int lds_floats;
};

struct NwdDev {
  const float* wp[8];    // packed MFMA weights: dblock1..4, ublock1..4
  const float* bias[8];  // folded bias per output row (16-padded)
  const float* wfin;     // folded final conv weights [4][256]
  float bfin;
};

struct cm_nwd {
  cm_ctx* ctx = nullptr;
  float* d_blob = nullptr;
  NwdDev dev = {};
};

static int pool_len(int L) { return (L - 3) / 2 + 1; }

static bool nwd_make_plan(int T, NwdPlan* pl) {
  NwdPlan p = {};
  p.T = T;
  p.P0 = pool_len(T);      p.E1 = p.P0 - 62;
  p.P1 = pool_len(p.E1);   p.E2 = p.P1 - 31;
  p.P2 = pool_len(p.E2);   p.E3 = p.P2 - 15;
  p.P3 = pool_len(p.E3);   p.E4 = p.P3 - 15;
  if (p.E1 < 3 || p.E2 < 3 || p.E3 < 3 || p.E4 < 1) return false;
  p.U1 = p.E4 + 15; p.U2 = p.E3 + 15; p.U3 = p.E2 + 31; p.U4 = 2 * (p.E1 + 15);
  auto al = [](int x) { return (x + 3) & ~3; };
  int o = 0;
  p.sC3 = al(p.E1 + 30); p.oC3 = o; o += 32 * p.sC3;
  p.sC2 = al(p.E2 + 62); p.oC2 = o; o += 32 * p.sC2;
  p.sC4 = al((((T + 1) / 2 + 15) / 16 + 17) * 17);     p.oC4 = p.oC2;     // rows of 16 samples: one per 16-column group + 17 of look-ahead (K = 272)
  const int c4_end = p.oC4 + 8 * p.sC4;
  p.sC1 = al(p.E3 + 30); p.oC1 = o; o += 48 * p.sC1;
  p.sB1 = al(p.E4 + 30); p.oB1 = o; o += 32 * p.sB1;
  if (c4_end > o) o = c4_end;                      // very short C2: let C4 run over C1/B1 (dead too) -- keep the total honest
  // scratch: max over stages of what is live there at once
  int s = al(T) + al(p.P0);
  s = std::max(s, 16 * al(p.P1));
  s = std::max(s, 16 * al(p.P2));
  s = std::max(s, 32 * al(p.P3));
  s = std::max(s, 16 * al(p.U3));
  s = std::max(s, 4 * al(p.U4));
  p.oS = o; o += s;
  // raw output + scan buffer reuse the C1/B1 region only if it is large enough; otherwise append
  const int need_tail = al(T) + 2 * 2 * NWD_THREADS;          // T floats + 2 x NWD_THREADS doubles
  if (48 * p.sC1 + 32 * p.sB1 >= need_tail && c4_end <= p.oC1) {
    p.oOut = p.oC1; p.oScan = p.oC1 + al(T);
  } else {
    p.oOut = o; p.oScan = o + al(T); o += need_tail;
  }
  p.oScan = (p.oScan + 1) & ~1;
  o = (o + 2 + 3) & ~3;
  const int budget = (NWD_LDS_LIMIT - 256) / (int)sizeof(float);                       // 256 B left for static LDS
  int nw = budget - o;
  nw = nw > NWD_WBUF ? NWD_WBUF : (nw & ~1023);
  p.oW = o; p.nW = nw; o += nw > 0 ? nw : 0;
  p.lds_floats = o;
  *pl = p;
  return nw >= NWD_WBUF_MIN;
}

// ------------------------------------------------------------------------------------------------------------
// device pieces
// ------------------------------------------------------------------------------------------------------------
// AvgPool1d(kernel 3, stride 2) of C rows
__device__ __forceinline__ void nwd_pool(const float* in, int in_stride, int C, int Lout, float* out, int out_stride) {
  for (int i = threadIdx.x; i < C * Lout; i += NWD_THREADS) {
    const int c = i / Lout, q = i - c * Lout;
    const float* r = in + c * in_stride + 2 * q;
    out[c * out_stride + q] = (r[0] + r[1] + r[2]) / 3.0f;
  }
}

// F.interpolate(mode='linear', align_corners=False) of C rows, Lin -> Lout (PyTorch's area_pixel source index in float32)
__device__ __forceinline__ void nwd_interp(const float* in, int in_stride, int C, int Lin, float* out, int out_stride, int Lout) {
  const float scale = (float)Lin / (float)Lout;
  for (int i = threadIdx.x; i < C * Lout; i += NWD_THREADS) {
    const int c = i / Lout, q = i - c * Lout;
    float src = scale * ((float)q + 0.5f) - 0.5f;
    src = src < 0.0f ? 0.0f : src;
    const int i0 = (int)src;
    const int i1 = i0 + ((i0 < Lin - 1) ? 1 : 0);
    const float l1 = src - (float)i0, l0 = 1.0f - l1;
    const float* r = in + c * in_stride;
    out[c * out_stride + q] = l0 * r[i0] + l1 * r[i1];
  }
}

// Implicit-GEMM correlation on the fp32 matrix cores:
//     out[row][p] = relu( bias[row] + sum_ic sum_t W[row][ic][t] * in[ic][p + t * dil] ),   0 <= p < out_len
// Operand layout of v_mfma_f32_16x16x4_f32: lane l feeds A[row = l & 15][k = l >> 4] and B[k = l >> 4][col = l & 15];
// accumulator register i of lane l holds D[row = 4 (l >> 4) + i][col = l & 15].
// Packed weights wp[((ic * n_mblk + mb) * (TAPS / 4) + ks) * 64 + l]: a run of input channels is contiguous, so the workgroup
// copies as many channels as fit into the LDS window `wbuf`, every wavefront multiplies them, and the next run follows.
// Wavefront w owns row block mb = w % n_mblk and the tiles (w / n_mblk) + j * (NWD_WAVES / n_mblk), j < NWD_MAXT: a weight fragment
// read from LDS is used for all of them (independent accumulators, which also hides the 40-cycle dependent-MFMA latency).
// STORE(row, p, value) places one finished element.
template <int TAPS, class Store>
__device__ __forceinline__ void nwd_conv_mfma(const float* in, int in_stride, int n_ic, int dil, const float* __restrict__ wp,
                                              const float* __restrict__ bias, int n_mblk, int m_valid, int out_len, float* wbuf,
                                              int wfloats, Store store) {
  constexpr int KS = TAPS / 4;
  const int wave = threadIdx.x / CM_WAVE, lane = threadIdx.x % CM_WAVE;
  const int col = lane & 15, kk = lane >> 4;
  const int ntiles = (out_len + 15) >> 4;
  const int mb = wave % n_mblk, slot = wave / n_mblk, nslots = NWD_WAVES / n_mblk;
  const int nt = slot < ntiles ? (ntiles - slot + nslots - 1) / nslots : 0;            // tiles of this wavefront (uniform)
  const float* bsrc[NWD_MAXT];
  cm_f4 acc[NWD_MAXT];
#pragma unroll
  for (int j = 0; j < NWD_MAXT; ++j) {
    const int p = 16 * (slot + j * nslots) + col;
    bsrc[j] = in + (p < out_len ? p : out_len - 1) + kk * dil;                          // clamp the read, drop the store
    acc[j] = cm_f4{0.0f, 0.0f, 0.0f, 0.0f};
  }
  const int per_ic = n_mblk * KS * 64;
  const int chunk = wfloats / per_ic < n_ic ? wfloats / per_ic : n_ic;
  for (int c0 = 0; c0 < n_ic; c0 += chunk) {
    const int nc = (n_ic - c0 < chunk) ? n_ic - c0 : chunk;
    __syncthreads();                                                                    // the window is free again
    {
      const float4* src = reinterpret_cast<const float4*>(wp + (size_t)c0 * per_ic);
      float4* dst = reinterpret_cast<float4*>(wbuf);
      for (int i = threadIdx.x; i < nc * per_ic / 4; i += NWD_THREADS) dst[i] = src[i];
    }
    __syncthreads();
    if (nt > 0) {
      const float* asrc = wbuf + mb * KS * 64 + lane;
      for (int ic = 0; ic < nc; ++ic) {
        float a[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a[ks] = asrc[ic * per_ic + ks * 64];
#pragma unroll
        for (int j = 0; j < NWD_MAXT; ++j) {
          if (j < nt) {
            float b[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) b[ks] = bsrc[j][(c0 + ic) * in_stride + 4 * ks * dil];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], b[ks], acc[j], 0, 0, 0);
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NWD_MAXT; ++j) {
    const int p = 16 * (slot + j * nslots) + col;
    if (j < nt && p < out_len) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = 16 * mb + 4 * kk + i;
        if (row < m_valid) {
          const float v = acc[j][i] + bias[row];
          store(row, p, v < 0.0f ? 0.0f : v);                       // ReLU that lets NaN through, like torch.relu
        }
      }
    }
  }
}

__global__ __launch_bounds__(NWD_THREADS) void k_nwd_forward(NwdPlan pl, NwdDev w, int K, const double* __restrict__ traces,
                                                             int monotone_start, double* __restrict__ out, float* __restrict__ raw, int stop) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ double red[NWD_WAVES];
  const int k = blockIdx.x;
  if (k >= K) return;
  const int tid = threadIdx.x, T = pl.T;
  const double* tr = traces + (size_t)k * T;

  // ---- zero the whole plan (padding must be zero), find the trace maximum
  for (int i = tid; i < pl.lds_floats; i += NWD_THREADS) lds[i] = 0.0f;
  double m = -INFINITY;
  bool has_nan = false;
  for (int t = tid; t < T; t += NWD_THREADS) { const double v = tr[t]; has_nan |= (v != v); m = v > m ? v : m; }
  for (int off = 32; off; off >>= 1) {
    const double o = __shfl_xor(m, off);
    m = o > m ? o : m;
  }
  const bool any_nan = __any(has_nan);
  if ((tid % CM_WAVE) == 0) red[tid / CM_WAVE] = any_nan ? NAN : m;
  __syncthreads();
  double tmax = red[0];
  for (int i = 1; i < NWD_WAVES; ++i) { const double o = red[i]; tmax = (o != o || tmax != tmax) ? NAN : (o > tmax ? o : tmax); }   // np.max propagates NaN

  float* X = lds + pl.oS;
  float* Q = lds + pl.oS + ((T + 3) & ~3);
  for (int t = tid; t < T; t += NWD_THREADS) X[t] = (float)(tr[t] / tmax);     // float32(traces / tmax), :45-47
  __syncthreads();

  float* C3 = lds + pl.oC3; float* C2 = lds + pl.oC2; float* C1 = lds + pl.oC1; float* B1 = lds + pl.oB1; float* C4 = lds + pl.oC4;
  float* S = lds + pl.oS;
  float* WB = lds + pl.oW;
  const int sC3 = pl.sC3, sC2 = pl.sC2, sC1 = pl.sC1, sB1 = pl.sB1, sC4 = pl.sC4;

  // ---- encoder: pool -> conv -> BN -> ReLU, each result written into its skip slot
  nwd_pool(X, 0, 1, pl.P0, Q, 0);
  __syncthreads();
  nwd_conv_mfma<32>(Q, 0, 1, 2, w.wp[0], w.bias[0], 1, 16, pl.E1, WB, pl.nW,
                    [&](int r, int p, float v) { C3[(16 + r) * sC3 + 15 + p] = v; });
  __syncthreads();
  if (stop == 1) return;                             // diagnostic (CAVIAR_NWD_STOP): phase timing by early exit
  const int sQ1 = (pl.P1 + 3) & ~3;
  nwd_pool(C3 + 16 * sC3 + 15, sC3, 16, pl.P1, S, sQ1);
  __syncthreads();
  nwd_conv_mfma<32>(S, sQ1, 16, 1, w.wp[1], w.bias[1], 1, 16, pl.E2, WB, pl.nW,
                    [&](int r, int p, float v) { C2[(16 + r) * sC2 + 31 + p] = v; });
  __syncthreads();
  if (stop == 2) return;                             // diagnostic (CAVIAR_NWD_STOP): phase timing by early exit
  const int sQ2 = (pl.P2 + 3) & ~3;
  nwd_pool(C2 + 16 * sC2 + 31, sC2, 16, pl.P2, S, sQ2);
  __syncthreads();
  nwd_conv_mfma<16>(S, sQ2, 16, 1, w.wp[2], w.bias[2], 2, 32, pl.E3, WB, pl.nW,
                    [&](int r, int p, float v) { C1[(16 + r) * sC1 + 15 + p] = v; });
  __syncthreads();
  if (stop == 3) return;                             // diagnostic (CAVIAR_NWD_STOP): phase timing by early exit
  const int sQ3 = (pl.P3 + 3) & ~3;
  nwd_pool(C1 + 16 * sC1 + 15, sC1, 32, pl.P3, S, sQ3);
  __syncthreads();
  nwd_conv_mfma<16>(S, sQ3, 32, 1, w.wp[3], w.bias[3], 2, 32, pl.E4, WB, pl.nW,
                    [&](int r, int p, float v) { B1[r * sB1 + 15 + p] = v; });
  __syncthreads();
  if (stop == 4) return;                             // diagnostic (CAVIAR_NWD_STOP): phase timing by early exit

  // ---- decoder: deconv -> BN -> ReLU into scratch, linear interpolation into rows 0..15 of the next concatenated input
  const int sU1 = (pl.U1 + 3) & ~3, sU2 = (pl.U2 + 3) & ~3, sU3 = (pl.U3 + 3) & ~3, sU4 = (pl.U4 + 3) & ~3;
  nwd_conv_mfma<16>(B1, sB1, 32, 1, w.wp[4], w.bias[4], 1, 16, pl.U1, WB, pl.nW, [&](int r, int p, float v) { S[r * sU1 + p] = v; });
  __syncthreads();
  nwd_interp(S, sU1, 16, pl.U1, C1 + 15, sC1, pl.E3);
  __syncthreads();
  if (stop == 5) return;                             // diagnostic (CAVIAR_NWD_STOP): phase timing by early exit
  nwd_conv_mfma<16>(C1, sC1, 48, 1, w.wp[5], w.bias[5], 1, 16, pl.U2, WB, pl.nW, [&](int r, int p, float v) { S[r * sU2 + p] = v; });
  __syncthreads();
  nwd_interp(S, sU2, 16, pl.U2, C2 + 31, sC2, pl.E2);
  __syncthreads();
  if (stop == 6) return;                             // diagnostic (CAVIAR_NWD_STOP): phase timing by early exit
  nwd_conv_mfma<32>(C2, sC2, 32, 1, w.wp[6], w.bias[6], 1, 16, pl.U3, WB, pl.nW, [&](int r, int p, float v) { S[r * sU3 + p] = v; });
  __syncthreads();
  nwd_interp(S, sU3, 16, pl.U3, C3 + 15, sC3, pl.E1);
  // C2 is dead from here on: its space becomes C4, whose padding must be zero again
  for (int i = tid; i < 8 * sC4; i += NWD_THREADS) C4[i] = 0.0f;
  __syncthreads();
  if (stop == 7) return;                             // diagnostic (CAVIAR_NWD_STOP): phase timing by early exit
  // stride-2 deconv as two polyphase branches: row r = 2 oc + phase, output sample 2 q + phase
  nwd_conv_mfma<16>(C3, sC3, 32, 1, w.wp[7], w.bias[7], 1, 8, pl.E1 + 15, WB, pl.nW,
                    [&](int r, int q, float v) { S[(r >> 1) * sU4 + 2 * q + (r & 1)] = v; });
  __syncthreads();
  // linear interpolation to T samples, written for the Toeplitz product below: padded sample i = 255 + x of channel c goes to
  // row (c, parity i & 1), element q = i >> 1 at (q / 16) * 17 + q % 16   (16 samples per 17-float row: conflict-free B reads)
  {
    const float scale = (float)pl.U4 / (float)T;
    for (int i = tid; i < 4 * T; i += NWD_THREADS) {
      const int c = i / T, x = i - c * T;
      float src = scale * ((float)x + 0.5f) - 0.5f;
      src = src < 0.0f ? 0.0f : src;
      const int i0 = (int)src;
      const int i1 = i0 + ((i0 < pl.U4 - 1) ? 1 : 0);
      const float l1 = src - (float)i0, l0 = 1.0f - l1;
      const float* r = S + c * sU4;
      const int ip = 255 + x, q = ip >> 1;
      C4[(2 * c + (ip & 1)) * sC4 + (q >> 4) * 17 + (q & 15)] = l0 * r[i0] + l1 * r[i1];
    }
    // compact folded weights of the last layer, zero-padded by 16 on both sides: WB[ic * 288 + 16 + t]
    for (int i = tid; i < 4 * 288; i += NWD_THREADS) {
      const int ic = i / 288, t = i - ic * 288 - 16;
      WB[i] = (t >= 0 && t < 256) ? w.wfin[ic * 256 + t] : 0.0f;
    }
  }
  __syncthreads();
  if (stop == 8) return;                             // diagnostic (CAVIAR_NWD_STOP): phase timing by early exit

  // ---- final ConvolutionBlock(4, 1, 256, padding 255, dilation 2) as a Toeplitz product on the matrix cores.
  // For output parity pi, s[q] = padded[2 q + pi] and out[2 m + pi] = sum_t s[m + t] w[t].  With m = 16 a + r:
  //     D[r][a] = sum_{j < 272} Wtoe[r][j] * s[16 a + j],   Wtoe[r][j] = w[j - r]  (0 outside 0 <= j - r < 256)
  // 8 work items = parity x 16-column tile (a) x channel pair, one per wavefront (the first 8); the two channel-pair partial sums are
  // combined in a fixed order afterwards (no atomics: bitwise reproducible).
  float* OUT = lds + pl.oOut;
  float* PART = S;                                   // [2][T] partial sums (the scratch rows are dead)
  {
    const int wave = tid / CM_WAVE, lane = tid % CM_WAVE, col = lane & 15, kk = lane >> 4;
    const int pi = wave & 1, ntile = (wave >> 1) & 1, pair = wave >> 2;
    const int na = ((T + 1) / 2 + 15) / 16;          // 16-sample groups per parity
    for (int nt0 = ntile; wave < 8 && 16 * nt0 < na; nt0 += 2) {
      const int a = 16 * nt0 + col;
      const int ac = a < na ? a : na - 1;
      cm_f4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
      const float* b0 = C4 + (2 * (2 * pair) + pi) * sC4 + ac * 17 + kk;
      const float* b1 = b0 + 2 * sC4;
      const float* a0 = WB + (2 * pair) * 288 + 16 + kk - col;
      const float* a1 = a0 + 288;
#pragma unroll 4
      for (int ks = 0; ks < 68; ++ks) {
        const int boff = (ks >> 2) * 17 + 4 * (ks & 3);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[4 * ks], b0[boff], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[4 * ks], b1[boff], acc1, 0, 0, 0);
      }
      if (a < na) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int p = 2 * (16 * a + 4 * kk + i) + pi;
          if (p < T) PART[pair * T + p] = acc0[i] + acc1[i];
        }
      }
    }
  }
  __syncthreads();
  for (int p = tid; p < T; p += NWD_THREADS) {
    const float v = PART[p] + PART[T + p] + w.bfin;
    OUT[p] = v < 0.0f ? 0.0f : v;
  }
  __syncthreads();
  if (stop == 9) return;                             // diagnostic (CAVIAR_NWD_STOP): phase timing by early exit

  // ---- rescale in float64 and enforce monotone decay from `monotone_start` on (running minimum seeded at start - 1)
  double* scan = reinterpret_cast<double*>(lds + pl.oScan);
  if (raw != nullptr)
    for (int t = tid; t < T; t += NWD_THREADS) raw[(size_t)k * T + t] = OUT[t];
  double* o = out + (size_t)k * T;
  const int s0 = monotone_start - 1;                                    // first column of the running minimum
  for (int t = tid; t < T && t < s0; t += NWD_THREADS) o[t] = (double)OUT[t] * tmax;
  if (s0 >= T) return;
  // each thread owns a contiguous chunk of the tail; chunk minima are scanned across the workgroup
  const int n = T - s0, per = (n + NWD_THREADS - 1) / NWD_THREADS;
  const int b = s0 + tid * per, e = (b + per < T) ? b + per : T;
  double run = INFINITY;
  bool run_nan = false;
  for (int t = b; t < e; ++t) { const double v = (double)OUT[t] * tmax; run_nan |= (v != v); run = v < run ? v : run; }
  // np.min([a, b]) propagates NaN: once a NaN enters the running minimum it stays
  scan[tid] = run_nan ? NAN : run;
  __syncthreads();
  int src = 0;
  for (int off = 1; off < NWD_THREADS; off <<= 1) {
    const double mine = scan[src * NWD_THREADS + tid];
    double r = mine;
    if (tid >= off) {
      const double oth = scan[src * NWD_THREADS + tid - off];
      r = (oth != oth || mine != mine) ? NAN : (oth < mine ? oth : mine);
    }
    scan[(src ^ 1) * NWD_THREADS + tid] = r;
    src ^= 1;
    __syncthreads();
  }
  double carry = tid ? scan[src * NWD_THREADS + tid - 1] : INFINITY;    // minimum over all earlier chunks
  for (int t = b; t < e; ++t) {
    const double v = (double)OUT[t] * tmax;
    carry = (v != v || carry != carry) ? NAN : (v < carry ? v : carry);
    o[t] = carry;
  }
}

// ------------------------------------------------------------------------------------------------------------
// host side: fold BatchNorm into the convolutions, pack for the MFMA operand layout
// ------------------------------------------------------------------------------------------------------------
namespace {
struct Folded {
  std::vector<float> scale, bias;                 // per output channel
};
Folded fold_bn(const cm_nwd_block& b, int cout, float eps) {
  Folded f; f.scale.resize(cout); f.bias.resize(cout);
  for (int c = 0; c < cout; ++c) {
    const double s = (double)b.bn_weight[c] / std::sqrt((double)b.bn_var[c] + (double)eps);
    f.scale[c] = (float)s;
    f.bias[c] = (float)(((double)b.bias[c] - (double)b.bn_mean[c]) * s + (double)b.bn_bias[c]);
  }
  return f;
}
// pack W(row, ic, tap) -> [ic][mb][taps/4][64 lanes]
template <class W>
void pack_mfma(std::vector<float>& dst, int m_valid, int n_ic, int taps, W wf) {
  const int n_mblk = (m_valid + 15) / 16;
  dst.assign((size_t)n_mblk * n_ic * (taps / 4) * 64, 0.0f);
  for (int mb = 0; mb < n_mblk; ++mb)
    for (int ic = 0; ic < n_ic; ++ic)
      for (int ks = 0; ks < taps / 4; ++ks)
        for (int l = 0; l < 64; ++l) {
          const int row = 16 * mb + (l & 15), tap = 4 * ks + (l >> 4);
          if (row < m_valid) dst[(((size_t)ic * n_mblk + mb) * (taps / 4) + ks) * 64 + l] = wf(row, ic, tap);
        }
}
}  // namespace

extern "C" int cm_nwd_max_trace_len(void) {
  int T = cm_nwd_min_trace_len();
  NwdPlan pl;
  while (T > 0 && nwd_make_plan(T + 1, &pl)) ++T;
  return T;
}
extern "C" int cm_nwd_min_trace_len(void) {
  NwdPlan pl;
  for (int T = 8; T < 4096; ++T)
    if (nwd_make_plan(T, &pl)) return T;
  return -1;
}

extern "C" int cm_nwd_create(cm_ctx* ctx, const cm_nwd_params* prm, cm_nwd** out) {
  if (!ctx || !prm || !out) { cm_set_error("cm_nwd_create: NULL argument"); return CM_ERR_INVALID; }
  const cm_nwd_block* all[9] = {&prm->dblock[0], &prm->dblock[1], &prm->dblock[2], &prm->dblock[3], &prm->ublock[0],
                                &prm->ublock[1], &prm->ublock[2], &prm->ublock[3], &prm->conv};
  for (const cm_nwd_block* b : all)
    if (!b->weight || !b->bias || !b->bn_weight || !b->bn_bias || !b->bn_mean || !b->bn_var) {
      cm_set_error("cm_nwd_create: NULL parameter tensor");
      return CM_ERR_INVALID;
    }
  CM_HIP(hipSetDevice(ctx->device));
  const float eps = prm->bn_eps > 0.0f ? prm->bn_eps : 1e-5f;
  // (cin, cout, k) of the eight MFMA layers (neural_waveform_demixing.py:258-266)
  static const int dn[4][3] = {{1, 16, 32}, {16, 16, 32}, {16, 32, 16}, {32, 32, 16}};
  static const int up[4][3] = {{32, 16, 16}, {48, 16, 16}, {32, 16, 32}, {32, 4, 32}};
  std::vector<float> packed[8], biasv[8];
  for (int i = 0; i < 4; ++i) {                                       // Conv1d weight [cout][cin][k]
    const int cin = dn[i][0], cout = dn[i][1], k = dn[i][2];
    const Folded f = fold_bn(prm->dblock[i], cout, eps);
    const float* W = prm->dblock[i].weight;
    pack_mfma(packed[i], cout, cin, k, [&](int oc, int ic, int t) { return f.scale[oc] * W[((size_t)oc * cin + ic) * k + t]; });
    biasv[i].assign(((cout + 15) / 16) * 16, 0.0f);
    for (int c = 0; c < cout; ++c) biasv[i][c] = f.bias[c];
  }
  for (int i = 0; i < 3; ++i) {                                       // ConvTranspose1d weight [cin][cout][k], stride 1: flipped taps
    const int cin = up[i][0], cout = up[i][1], k = up[i][2];
    const Folded f = fold_bn(prm->ublock[i], cout, eps);
    const float* W = prm->ublock[i].weight;
    pack_mfma(packed[4 + i], cout, cin, k,
              [&](int oc, int ic, int t) { return f.scale[oc] * W[((size_t)ic * cout + oc) * k + (k - 1 - t)]; });
    biasv[4 + i].assign(16, 0.0f);
    for (int c = 0; c < cout; ++c) biasv[4 + i][c] = f.bias[c];
  }
  {                                                                   // ublock4, stride 2: rows 2 oc + phase, 16 taps per phase
    const int cin = 32, cout = 4, k = 32;
    const Folded f = fold_bn(prm->ublock[3], cout, eps);
    const float* W = prm->ublock[3].weight;
    pack_mfma(packed[7], 8, cin, 16, [&](int r, int ic, int t) {
      const int oc = r >> 1, ph = r & 1;
      return f.scale[oc] * W[((size_t)ic * cout + oc) * k + 2 * (15 - t) + ph];
    });
    biasv[7].assign(16, 0.0f);
    for (int r = 0; r < 8; ++r) biasv[7][r] = f.bias[r >> 1];
  }
  const Folded ff = fold_bn(prm->conv, 1, eps);
  std::vector<float> wfin(4 * 256);
  for (int i = 0; i < 4 * 256; ++i) wfin[i] = ff.scale[0] * prm->conv.weight[i];

  size_t total = wfin.size();
  for (int i = 0; i < 8; ++i) total += packed[i].size() + biasv[i].size();
  std::vector<float> blob;
  blob.reserve(total);
  size_t off_w[8], off_b[8];
  for (int i = 0; i < 8; ++i) {
    off_w[i] = blob.size(); blob.insert(blob.end(), packed[i].begin(), packed[i].end());
    off_b[i] = blob.size(); blob.insert(blob.end(), biasv[i].begin(), biasv[i].end());
  }
  const size_t off_fin = blob.size();
  blob.insert(blob.end(), wfin.begin(), wfin.end());

  cm_nwd* n = new cm_nwd();
  n->ctx = ctx;
  if (hipMalloc(&n->d_blob, blob.size() * sizeof(float)) != hipSuccess) {
    delete n; cm_set_error("cm_nwd_create: hipMalloc failed"); return CM_ERR_HIP;
  }
  if (hipMemcpy(n->d_blob, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
    hipFree(n->d_blob); delete n; cm_set_error("cm_nwd_create: upload failed"); return CM_ERR_HIP;
  }
  for (int i = 0; i < 8; ++i) { n->dev.wp[i] = n->d_blob + off_w[i]; n->dev.bias[i] = n->d_blob + off_b[i]; }
  n->dev.wfin = n->d_blob + off_fin;
  n->dev.bfin = ff.bias[0];
  *out = n;
  return CM_OK;
}

extern "C" void cm_nwd_destroy(cm_nwd* n) {
  if (!n) return;
  if (n->ctx) hipSetDevice(n->ctx->device);
  if (n->d_blob) hipFree(n->d_blob);
  delete n;
}

extern "C" int cm_nwd_demix(cm_ctx* ctx, const cm_nwd* nwd, int K, int T, const double* traces, int traces_mem,
                            int monotone_filter_start, double* out, float* raw, int out_mem, float* kernel_ms) {
  if (!ctx || !nwd || !traces || !out) { cm_set_error("cm_nwd_demix: NULL argument"); return CM_ERR_INVALID; }
  if (K < 0 || monotone_filter_start < 1) {
    cm_set_error("cm_nwd_demix: need K >= 0 and monotone_filter_start >= 1 (got %d, %d)", K, monotone_filter_start);
    return CM_ERR_INVALID;
  }
  NwdPlan pl;
  if (!nwd_make_plan(T, &pl)) {
    cm_set_error("cm_nwd_demix: trace length %d outside the fused kernel's range [%d, %d]", T, cm_nwd_min_trace_len(), cm_nwd_max_trace_len());
    return CM_ERR_INVALID;
  }
  if (kernel_ms) *kernel_ms = 0.0f;
  if (K == 0) return CM_OK;
  CM_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t n = (size_t)K * T;
  const double* d_in = traces;
  double* d_out = out;
  float* d_raw = raw;
  double* tmp_in = nullptr; double* tmp_out = nullptr; float* tmp_raw = nullptr;
  int rc = CM_OK;
  auto cleanup = [&]() { if (tmp_in) hipFree(tmp_in); if (tmp_out) hipFree(tmp_out); if (tmp_raw) hipFree(tmp_raw); };
#define NWD_HIP(call) do { hipError_t _e = (call); if (_e != hipSuccess) { cm_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(_e)); cleanup(); return CM_ERR_HIP; } } while (0)
  if (traces_mem == CM_MEM_HOST) {
    NWD_HIP(hipMalloc(&tmp_in, n * sizeof(double)));
    NWD_HIP(hipMemcpyAsync(tmp_in, traces, n * sizeof(double), hipMemcpyHostToDevice, st));
    d_in = tmp_in;
  }
  if (out_mem == CM_MEM_HOST) {
    NWD_HIP(hipMalloc(&tmp_out, n * sizeof(double)));
    d_out = tmp_out;
    if (raw) { NWD_HIP(hipMalloc(&tmp_raw, n * sizeof(float))); d_raw = tmp_raw; }
  }
  if (cm_raise_lds(ctx, CM_LDS_NWD, reinterpret_cast<const void*>(k_nwd_forward), NWD_LDS_LIMIT - 256, 0) != CM_OK) { cleanup(); return CM_ERR_HIP; }
  const size_t lds_bytes = (size_t)pl.lds_floats * sizeof(float);
  NWD_HIP(hipEventRecord(ctx->ev[0], st));
  static const int stop = getenv("CAVIAR_NWD_STOP") ? atoi(getenv("CAVIAR_NWD_STOP")) : 0;
  hipLaunchKernelGGL(k_nwd_forward, dim3(K), dim3(NWD_THREADS), lds_bytes, st, pl, nwd->dev, K, d_in, monotone_filter_start, d_out, d_raw, stop);
  NWD_HIP(hipGetLastError());
  NWD_HIP(hipEventRecord(ctx->ev[1], st));
  if (out_mem == CM_MEM_HOST) {
    NWD_HIP(hipMemcpyAsync(out, tmp_out, n * sizeof(double), hipMemcpyDeviceToHost, st));
    if (raw) NWD_HIP(hipMemcpyAsync(raw, tmp_raw, n * sizeof(float), hipMemcpyDeviceToHost, st));
  }
  if (out_mem == CM_MEM_HOST || traces_mem == CM_MEM_HOST || kernel_ms) {
    NWD_HIP(hipStreamSynchronize(st));
    if (kernel_ms) NWD_HIP(hipEventElapsedTime(kernel_ms, ctx->ev[0], ctx->ev[1]));
  }
#undef NWD_HIP
  cleanup();
  return rc;
}
