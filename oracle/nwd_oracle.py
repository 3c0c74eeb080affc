"""CPU oracle for the NWD demixer forward pass (reference circuitmap/neural_waveform_demixing.py).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the
product path (circuitmap_amd/ calls the HIP library and fails loudly without it).

A float32 restatement on plain PyTorch (CPU) of the reference's inference path, written from the module definitions
rather than by instantiating them, so that it needs neither pytorch_lightning nor the reference:

    NeuralDemixer.__call__                   neural_waveform_demixing.py:36-54
    NWDUNet.forward                          :255-290
    DownsamplingBlock (pool -> conv -> BN -> ReLU)                  :204-218
    UpsamplingBlock   (deconv -> BN -> ReLU -> linear interp, cat)  :220-238
    ConvolutionBlock  (conv -> BN -> ReLU)                          :240-252
    _monotone_decay_filter                   :337-348

Parity pin: tests/golden/nwd_forward.npz was produced by the reference's own module running on real PyTorch in the
build container (tests/golden/make_golden_nwd.py); tests/test_nwd_oracle.py checks this restatement against it.
The state dict uses the reference's parameter names (`dblock1.conv.weight`, `ublock2.deconv.weight`, `conv.bn.running_var` ...).
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5                      # torch.nn.BatchNorm1d default, not overridden by the reference

# (name, in_channels, out_channels, kernel_size, dilation)          NWDUNet.__init__ :258-261
DOWN = (("dblock1", 1, 16, 32, 2), ("dblock2", 16, 16, 32, 1), ("dblock3", 16, 32, 16, 1), ("dblock4", 32, 32, 16, 1))
# (name, in_channels, out_channels, kernel_size, stride)            :263-266
UP = (("ublock1", 32, 16, 16, 1), ("ublock2", 48, 16, 16, 1), ("ublock3", 32, 16, 32, 1), ("ublock4", 32, 4, 32, 2))
# final ConvolutionBlock(4, 1, 256, padding=255, stride=1, dilation=2)   :268
FINAL = ("conv", 4, 1, 256, 255, 1, 2)


def _t(state, key):
    return torch.as_tensor(np.asarray(state[key]), dtype=torch.float32)


def _bn_relu(x, state, name):
    return F.relu(F.batch_norm(x, _t(state, name + ".bn.running_mean"), _t(state, name + ".bn.running_var"),
                               _t(state, name + ".bn.weight"), _t(state, name + ".bn.bias"), False, 0.0, BN_EPS))


def unet_forward(x, state):
    """x: float32 tensor [B, 1, T] -> [B, 1, T]   (eval mode: BatchNorm uses its running statistics)"""
    enc = []
    h = x
    for name, _, _, _, dil in DOWN:
        h = F.avg_pool1d(h, kernel_size=3, stride=2)
        h = F.conv1d(h, _t(state, name + ".conv.weight"), _t(state, name + ".conv.bias"), dilation=dil)
        h = _bn_relu(h, state, name)
        enc.append(h)
    skips = (enc[2], enc[1], enc[0], None)
    for (name, _, _, _, stride), skip in zip(UP, skips):
        h = F.conv_transpose1d(h, _t(state, name + ".deconv.weight"), _t(state, name + ".deconv.bias"), stride=stride)
        h = _bn_relu(h, state, name)
        size = skip.shape[-1] if skip is not None else x.shape[-1]
        h = F.interpolate(h, size=size, mode="linear", align_corners=False)
        if skip is not None:
            h = torch.cat([h, skip], dim=1)
    name, _, _, _, pad, stride, dil = FINAL
    h = F.conv1d(h, _t(state, name + ".conv.weight"), _t(state, name + ".conv.bias"), stride=stride, padding=pad, dilation=dil)
    return _bn_relu(h, state, name)


def monotone_decay_filter(arr, monotone_start=500):
    """running minimum from column `monotone_start` onwards (:337-348; in-place and copying variants agree)"""
    out = np.array(arr, dtype=np.float64, copy=True)
    for t in range(monotone_start, out.shape[1]):
        out[:, t] = np.minimum(out[:, t], out[:, t - 1])
    return out


def demix(traces, state, monotone_filter_start=500, return_raw=False):
    """NeuralDemixer.__call__ (:36-54): normalise each trace by its maximum, run the U-Net in float32, rescale in
    float64, enforce monotone decay.  traces: float64 [K, T]."""
    traces = np.asarray(traces, dtype=np.float64)
    tmax = np.max(traces, axis=1)[:, None]
    with torch.no_grad():
        raw = unet_forward(torch.Tensor((traces / tmax).copy()[:, None, :]), state).numpy()[:, 0, :]
    dem = monotone_decay_filter(raw * tmax, monotone_filter_start)
    return (dem, raw) if return_raw else dem


def random_state(seed=0):
    """Seeded random parameters with the reference's names and shapes (stand-in for a trained checkpoint)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv_like(name, shape, fan_in):
        bound = 1.0 / np.sqrt(fan_in)
        sd[name + ".weight"] = ((torch.rand(shape, generator=g) * 2 - 1) * bound).numpy()
        n_out = shape[0] if ".conv" in name else shape[1]
        sd[name + ".bias"] = ((torch.rand(n_out, generator=g) * 2 - 1) * bound).numpy()

    def bn(name, c):
        sd[name + ".bn.weight"] = (0.75 + 0.5 * torch.rand(c, generator=g)).numpy()
        sd[name + ".bn.bias"] = (0.1 * torch.randn(c, generator=g)).numpy()
        sd[name + ".bn.running_mean"] = (0.2 * torch.randn(c, generator=g)).numpy()
        sd[name + ".bn.running_var"] = (0.5 + torch.rand(c, generator=g)).numpy()

    for name, cin, cout, k, _ in DOWN:
        conv_like(name + ".conv", (cout, cin, k), cin * k)
        bn(name, cout)
    for name, cin, cout, k, _ in UP:
        conv_like(name + ".deconv", (cin, cout, k), cout * k)
        bn(name, cout)
    name, cin, cout, k = FINAL[:4]
    conv_like(name + ".conv", (cout, cin, k), cin * k)
    bn(name, cout)
    # keep the single output channel alive: a random BatchNorm shift below zero would make the last ReLU return all zeros,
    # and a comparison of two all-zero outputs tests nothing
    sd[name + ".bn.bias"] = np.abs(sd[name + ".bn.bias"]) + np.float32(0.25)
    sd[name + ".bn.running_mean"] = np.zeros(cout, dtype=np.float32)
    return sd
