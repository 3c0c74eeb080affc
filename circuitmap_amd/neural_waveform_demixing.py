"""MI355X drop-in for the inference half of ``circuitmap.neural_waveform_demixing`` (reference
circuitmap/neural_waveform_demixing.py:17-54, 255-290, 337-348).

    from circuitmap_amd.neural_waveform_demixing import NeuralDemixer
    demix = NeuralDemixer(path='demixers/nwd_ee_ChroME1.ckpt')
    psc_dem = demix(psc)                       # float64 [K, T], as the reference returns

`NeuralDemixer.__call__` runs the fused HIP kernel of libcaviar_hip.so (include/nwd_hip.h): same name, arguments and
return as the reference's.  There is no CPU fallback and no PyTorch in the forward pass; PyTorch is only used to read
a checkpoint file.  Training (`train`, `generate_training_data`) is not part of this package -- train with the
reference's PyTorch-Lightning loop and load the resulting checkpoint here.
"""
import ctypes as C
import time

import numpy as np

from . import _lib

# reference parameter names, in the order of cm_nwd_params (neural_waveform_demixing.py:258-268)
_DOWN = ("dblock1", "dblock2", "dblock3", "dblock4")
_UP = ("ublock1", "ublock2", "ublock3", "ublock4")
# expected weight shapes: Conv1d [out, in, k]; ConvTranspose1d [in, out, k]
_SHAPES = {"dblock1.conv": (16, 1, 32), "dblock2.conv": (16, 16, 32), "dblock3.conv": (32, 16, 16), "dblock4.conv": (32, 32, 16),
           "ublock1.deconv": (32, 16, 16), "ublock2.deconv": (48, 16, 16), "ublock3.deconv": (32, 16, 32), "ublock4.deconv": (32, 4, 32),
           "conv.conv": (1, 4, 256)}
BN_EPS = 1e-5


def load_checkpoint_state(path):
    """state dict (name -> float32 ndarray) of a PyTorch-Lightning checkpoint of the reference's NWDUNet, or of a plain
    ``torch.save(state_dict)`` file.  Loaded with ``weights_only=True``: tensors and plain containers only, no pickled code."""
    import torch
    try:
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:          # deliberately no unsafe retry
        raise RuntimeError("could not read %s with torch.load(weights_only=True): %s.  Re-save the parameters as a plain "
                           "state dict (torch.save(model.state_dict(), path)) in a trusted environment." % (path, e))
    sd = ckpt["state_dict"] if isinstance(ckpt, dict) and "state_dict" in ckpt else ckpt
    return {k: v.detach().cpu().numpy() for k, v in sd.items() if hasattr(v, "detach")}


def random_parameters(seed=0):
    """Seeded random parameters under the reference's names and shapes (NumPy only): inputs for benchmarks and smoke runs when no
    trained checkpoint is at hand (PyTorch-style uniform(-1/sqrt(fan_in), 1/sqrt(fan_in)) weights, non-trivial BatchNorm statistics;
    the single output channel is kept alive so the last ReLU does not return all zeros)."""
    rng = np.random.default_rng(seed)
    sd = {}
    for conv, shape in _SHAPES.items():
        block = conv.split(".")[0]
        is_conv = conv.endswith(".conv")
        cout = shape[0] if is_conv else shape[1]
        fan_in = shape[1] * shape[2]                      # Conv1d [out, in, k] -> in * k; ConvTranspose1d [in, out, k] -> out * k
        bound = 1.0 / np.sqrt(fan_in)
        sd[conv + ".weight"] = rng.uniform(-bound, bound, shape).astype(np.float32)
        sd[conv + ".bias"] = rng.uniform(-bound, bound, cout).astype(np.float32)
        sd[block + ".bn.weight"] = rng.uniform(0.75, 1.25, cout).astype(np.float32)
        sd[block + ".bn.bias"] = (0.1 * rng.standard_normal(cout)).astype(np.float32)
        sd[block + ".bn.running_mean"] = (0.2 * rng.standard_normal(cout)).astype(np.float32)
        sd[block + ".bn.running_var"] = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sd["conv.bn.bias"] = np.abs(sd["conv.bn.bias"]) + np.float32(0.25)
    sd["conv.bn.running_mean"] = np.zeros(1, dtype=np.float32)
    return sd


class NeuralDemixer:
    """Mirror of the reference class (:17-34).  `path` = checkpoint; alternatively `state` = dict of parameter arrays under
    the reference's names.  `device` = 'cuda', 'cuda:<i>' or an int; anything that is not a GPU raises (no CPU path)."""

    def __init__(self, path=None, eval_mode=True, device=None, state=None):
        if isinstance(device, str):
            if not device.startswith("cuda"):
                raise _lib.CaviarHipError("circuitmap_amd.NeuralDemixer runs on an MI355X only (device=%r): there is no CPU path" % device)
            device = int(device.split(":")[1]) if ":" in device else 0
        self.device = int(device or 0)
        if not eval_mode:
            raise NotImplementedError("training-mode BatchNorm (batch statistics) is not implemented: inference only")
        if path is not None:
            state = load_checkpoint_state(path)
        if state is None:
            raise ValueError("NeuralDemixer needs a checkpoint `path` or a `state` dict (an untrained network is only useful for "
                             "training, which this package leaves to the reference)")
        self.ctx = _lib.default_context(self.device)
        self.lib = self.ctx.lib
        self._keep = []
        self.handle = C.c_void_p()
        prm = _lib.CmNwdParams()
        for i, name in enumerate(_DOWN):
            self._fill(prm.dblock[i], state, name, name + ".conv")
        for i, name in enumerate(_UP):
            self._fill(prm.ublock[i], state, name, name + ".deconv")
        self._fill(prm.conv, state, "conv", "conv.conv")
        prm.bn_eps = BN_EPS
        with self.ctx.lock:
            _lib.check(self.lib.cm_nwd_create(self.ctx.handle, C.byref(prm), C.byref(self.handle)))
        self._keep = []             # the library has copied (and folded) everything

    def _fill(self, blk, state, block, conv):
        def arr(key, shape=None):
            if key not in state:
                raise KeyError("demixer state has no %r" % key)
            a = np.ascontiguousarray(np.asarray(state[key]), dtype=np.float32)
            if shape is not None and a.shape != shape:
                raise ValueError("%s has shape %s, expected %s" % (key, a.shape, shape))
            self._keep.append(a)
            return a.ctypes.data_as(C.POINTER(C.c_float))
        shape = _SHAPES[conv]
        cout = shape[0] if conv.endswith(".conv") else shape[1]
        blk.weight = arr(conv + ".weight", shape)
        blk.bias = arr(conv + ".bias", (cout,))
        blk.bn_weight = arr(block + ".bn.weight", (cout,))
        blk.bn_bias = arr(block + ".bn.bias", (cout,))
        blk.bn_mean = arr(block + ".bn.running_mean", (cout,))
        blk.bn_var = arr(block + ".bn.running_var", (cout,))

    def __call__(self, traces, monotone_filter_start=500, monotone_filter_inplace=True, verbose=True, return_raw=False):
        """Run the demixer over a PSC trace batch and apply the monotone decay filter (reference :36-54).
        traces: [K, T] array.  Returns float64 [K, T] (`monotone_filter_inplace` only chose between two equivalent code
        paths in the reference and is accepted for compatibility)."""
        if verbose:
            print("Demixing PSC traces... ", end="")
        t1 = time.time()
        traces = np.ascontiguousarray(traces, dtype=np.float64)
        if traces.ndim != 2:
            raise ValueError("traces must be [K, T]")
        K, T = traces.shape
        out = np.empty((K, T), dtype=np.float64)
        raw = np.empty((K, T), dtype=np.float32) if return_raw else None
        ms = C.c_float(0.0)
        with self.ctx.lock:
            _lib.check(self.lib.cm_nwd_demix(self.ctx.handle, self.handle, K, T, traces.ctypes.data_as(C.c_void_p), _lib.CM_MEM_HOST,
                                             int(monotone_filter_start), out.ctypes.data_as(C.c_void_p),
                                             raw.ctypes.data_as(C.c_void_p) if return_raw else None, _lib.CM_MEM_HOST, C.byref(ms)))
        self.last_kernel_ms = float(ms.value)
        t2 = time.time()
        if verbose:
            print("complete (elapsed time %.2fs, device=cuda:%d)." % (t2 - t1, self.device))
        return (out, raw) if return_raw else out

    def demix_device(self, traces_ptr, K, T, out_ptr, monotone_filter_start=500, raw_ptr=None):
        """Device-resident variant: `traces_ptr` / `out_ptr` are device addresses of float64 [K, T] buffers (e.g.
        ``tensor.data_ptr()``); the result can be handed to ``cm_problem_create(..., CM_MEM_DEVICE)`` without touching
        the host.  Returns the kernel time in ms."""
        ms = C.c_float(0.0)
        with self.ctx.lock:
            _lib.check(self.lib.cm_nwd_demix(self.ctx.handle, self.handle, int(K), int(T), C.c_void_p(traces_ptr), _lib.CM_MEM_DEVICE,
                                             int(monotone_filter_start), C.c_void_p(out_ptr),
                                             C.c_void_p(raw_ptr) if raw_ptr else None, _lib.CM_MEM_DEVICE, C.byref(ms)))
        return float(ms.value)

    def train(self, *args, **kwargs):
        raise NotImplementedError("training stays with the reference's PyTorch-Lightning loop; load its checkpoint here")

    generate_training_data = train

    def close(self):
        if getattr(self, "handle", None):
            with self.ctx.lock:
                self.lib.cm_nwd_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
