/*
 * nwd_hip.h -- C ABI of the NWD demixer forward pass in libcaviar_hip.so (MI355X, gfx950).
 *
 * The step immediately upstream of CAVIaR in the reference's pipeline (scripts/run_circuitmap_main.py:44-46):
 *     demix = NeuralDemixer(path=...);  psc_dem = demix(psc)          reference circuitmap/neural_waveform_demixing.py:17-54
 * i.e. per trace: divide by the trace maximum, run the 1-D U-Net `NWDUNet` (:255-290) in float32 in eval mode, multiply
 * back, enforce monotone decay after `monotone_filter_start` (:337-348).  Inference only: training (`train`,
 * `generate_training_data`) stays with the reference's PyTorch-Lightning trainer.
 *
 * Conventions as in caviar_hip.h: plain pointers and sizes, row-major, no exceptions, CM_OK or a negative cm_status with
 * cm_last_error(); the library never frees caller memory.  Parameter tensors are float32 in the layouts of the
 * reference's state dict (torch Conv1d weight [out, in, k]; ConvTranspose1d weight [in, out, k]).
 */
#ifndef NWD_HIP_H
#define NWD_HIP_H

#include "caviar_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cm_nwd cm_nwd;           /* one demixer's parameters, BatchNorm-folded and packed for the kernels, in HBM */

/* conv (or deconv) + BatchNorm1d of one block; host pointers.  State-dict names: <block>.conv.weight / <block>.deconv.weight,
 * .bias, <block>.bn.weight, .bn.bias, .bn.running_mean, .bn.running_var */
typedef struct cm_nwd_block {
  const float* weight;
  const float* bias;
  const float* bn_weight;
  const float* bn_bias;
  const float* bn_mean;
  const float* bn_var;
} cm_nwd_block;

/* NWDUNet (neural_waveform_demixing.py:258-268):
 *   dblock1..4 = DownsamplingBlock(1,16,32,dil 2), (16,16,32,1), (16,32,16,1), (32,32,16,1)     AvgPool1d(3,2) -> Conv1d -> BN -> ReLU
 *   ublock1..4 = UpsamplingBlock(32,16,16,s 1), (48,16,16,1), (32,16,32,1), (32,4,32,2)         ConvTranspose1d -> BN -> ReLU -> linear interp (+ skip cat)
 *   conv       = ConvolutionBlock(4,1,256,padding 255,stride 1,dilation 2)                       Conv1d -> BN -> ReLU */
typedef struct cm_nwd_params {
  cm_nwd_block dblock[4];
  cm_nwd_block ublock[4];
  cm_nwd_block conv;
  float bn_eps;                         /* 1e-5 (torch default; the reference does not override it) */
} cm_nwd_params;

/* Upload + fold + pack the parameters (replaces NWDUNet().load_from_checkpoint(path).eval().to(device), :27-34). */
int cm_nwd_create(cm_ctx* ctx, const cm_nwd_params* params, cm_nwd** out);
void cm_nwd_destroy(cm_nwd* nwd);

/* Largest trace length the fused kernel's LDS plan accommodates, smallest it accepts (U-Net bottleneck >= 1 sample). */
int cm_nwd_max_trace_len(void);
int cm_nwd_min_trace_len(void);

/* NeuralDemixer.__call__ (:36-54) for K traces of length T (float64, [K, T]):
 *   out[k, :] = monotone_decay_filter( float64(unet(float32(traces[k] / max(traces[k])))) * max(traces[k]) )
 * traces / out / raw live on the host or on the device according to traces_mem / out_mem (cm_mem_kind); raw (optional,
 * may be NULL; float32 [K, T], same memory kind as out) receives the network output before rescaling and filtering.
 * monotone_filter_start >= 1 (the reference's loop reads column start-1); >= T disables the filter.
 * kernel_ms (optional) receives the kernel's duration measured with HIP events on the ctx stream. */
int cm_nwd_demix(cm_ctx* ctx, const cm_nwd* nwd, int K, int T, const double* traces, int traces_mem,
                 int monotone_filter_start, double* out, float* raw, int out_mem, float* kernel_ms);

#ifdef __cplusplus
}
#endif
#endif
