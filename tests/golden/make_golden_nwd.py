#!/usr/bin/env python3
"""Generate golden vectors for the NWD demixer forward pass by executing the reference's own module.

BUILD-CONTAINER TOOLING ONLY (needs /root/reference; never runs on the GPU box, never imported by the product).
Loads ``/root/reference/circuitmap/neural_waveform_demixing.py`` BY PATH with a two-line stand-in for the absent
``pytorch_lightning`` (``LightningModule = torch.nn.Module``; the forward pass uses nothing else of it) and real
PyTorch (CPU, float32) underneath, so the arithmetic IS the reference's.  No checkpoint is read: the network gets
seeded random parameters and seeded, non-trivial BatchNorm running statistics.  Stored: the parameter tensors,
seeded input traces, the raw network output and the output of ``NeuralDemixer.__call__`` (max-normalisation,
rescaling, monotone decay filter).  Only data is written; no reference text is copied.

    python tests/golden/make_golden_nwd.py
"""
import contextlib
import importlib.util
import io
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/circuitmap/neural_waveform_demixing.py"


def load_reference():
    pl = types.ModuleType("pytorch_lightning")
    pl.LightningModule = torch.nn.Module
    pl.Trainer = object
    sys.modules.setdefault("pytorch_lightning", pl)
    spec = importlib.util.spec_from_file_location("ref_nwd", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def seeded_traces(rng, K, T):
    """PSC-like test traces: sums of difference-of-exponential events + noise, plus degenerate rows."""
    t = np.arange(T)[None, :]
    out = np.zeros((K, T))
    for k in range(K):
        for _ in range(rng.integers(0, 4)):
            tr, td = rng.uniform(10, 60), rng.uniform(80, 300)
            d = rng.uniform(-200, 800)
            kern = (np.exp(-(t[0] - d) / td) - np.exp(-(t[0] - d) / tr)) * (t[0] >= d)
            out[k] += rng.uniform(0.2, 3.0) * kern
        out[k] += rng.normal(0, 0.02, T)
    return out


def main():
    ref = load_reference()
    torch.manual_seed(1234)
    torch.set_num_threads(1)
    dm = ref.NeuralDemixer(path=None, device="cpu")
    net = dm.demixer
    g = torch.Generator().manual_seed(99)
    with torch.no_grad():
        for name, buf in net.named_buffers():
            if name.endswith("running_mean"):
                buf.copy_(0.2 * torch.randn(buf.shape, generator=g))
            elif name.endswith("running_var"):
                buf.copy_(0.5 + torch.rand(buf.shape, generator=g))
        for name, p in net.named_parameters():
            if ".bn." in name and name.endswith("weight"):
                p.copy_(0.75 + 0.5 * torch.rand(p.shape, generator=g))
            elif ".bn." in name and name.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
    net.eval()

    rng = np.random.default_rng(7)
    cases = {}
    for tag, K, T in (("t900", 6, 900), ("t800", 3, 800)):
        traces = seeded_traces(rng, K, T)
        if tag == "t900":
            traces[4] = -np.abs(traces[4]) - 0.1          # all-negative row: tmax < 0 flips the sign of the input
        with torch.no_grad():
            tmax = np.max(traces, axis=1)[:, None]
            raw = net(torch.Tensor((traces / tmax).copy()[:, None, :])).numpy().squeeze()
        with contextlib.redirect_stdout(io.StringIO()):
            dem = dm(traces.copy(), verbose=False)
            dem_late = dm(traces.copy(), monotone_filter_start=100, monotone_filter_inplace=False, verbose=False)
        cases[tag + "_traces"] = traces
        cases[tag + "_raw"] = raw.astype(np.float32)
        cases[tag + "_dem"] = np.asarray(dem, dtype=np.float64)
        cases[tag + "_dem_start100"] = np.asarray(dem_late, dtype=np.float64)

    state = {"sd__" + k: v.detach().numpy() for k, v in net.state_dict().items()}
    np.savez_compressed(os.path.join(HERE, "nwd_forward.npz"), **state, **cases)
    for k, v in net.state_dict().items():
        print(k, tuple(v.shape), v.dtype)
    print({k: v.shape for k, v in cases.items()})


if __name__ == "__main__":
    main()
