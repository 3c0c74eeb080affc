"""Host-side logic of the NeuralDemixer mirror (no GPU): checkpoint reading, argument checks, no CPU fallback."""
import os

import numpy as np
import pytest


def test_checkpoint_reader_roundtrip(tmp_path):
    """a Lightning-style checkpoint ({'state_dict': ...}) and a plain state dict, both written by us, read back safely"""
    import torch
    from oracle import nwd_oracle as no
    from circuitmap_amd.neural_waveform_demixing import load_checkpoint_state
    state = no.random_state(seed=1)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in state.items()}
    sd["dblock1.bn.num_batches_tracked"] = torch.tensor(7)
    p1, p2 = str(tmp_path / "lightning.ckpt"), str(tmp_path / "plain.pt")
    torch.save({"epoch": 3, "global_step": 10, "pytorch-lightning_version": "1.5.0", "state_dict": sd}, p1)
    torch.save(sd, p2)
    for p in (p1, p2):
        got = load_checkpoint_state(p)
        for k, v in state.items():
            np.testing.assert_array_equal(got[k], v)


def test_unsafe_checkpoint_is_refused(tmp_path):
    """a checkpoint that needs arbitrary unpickling is rejected with an explanation, never loaded unsafely"""
    import pickle
    import torch
    from circuitmap_amd.neural_waveform_demixing import load_checkpoint_state

    class Evil:
        def __reduce__(self):
            return (os.path.join, ("a", "b"))
    p = str(tmp_path / "evil.ckpt")
    torch.save({"state_dict": {}, "callbacks": Evil()}, p, pickle_module=pickle)
    with pytest.raises(RuntimeError, match="weights_only"):
        load_checkpoint_state(p)


def test_no_cpu_path_and_argument_checks():
    from circuitmap_amd import _lib
    from circuitmap_amd.neural_waveform_demixing import NeuralDemixer
    from oracle import nwd_oracle as no
    with pytest.raises(_lib.CaviarHipError, match="no CPU path"):
        NeuralDemixer(state=no.random_state(0), device="cpu")
    with pytest.raises(ValueError):
        NeuralDemixer()
    with pytest.raises(NotImplementedError):
        NeuralDemixer(state=no.random_state(0), eval_mode=False)
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(_lib.CaviarHipError):
            NeuralDemixer(state=no.random_state(0))          # context creation fails loudly without a GPU


def test_trace_length_range():
    import __graft_entry__ as ge
    ge.build()
    from circuitmap_amd import _lib
    lib = _lib.load()
    assert lib.cm_nwd_min_trace_len() == 639        # shortest trace that leaves one sample at the U-Net bottleneck
    assert lib.cm_nwd_max_trace_len() >= 1000
