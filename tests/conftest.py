import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def tiny_setup():
    """Tiny-dims model: spec, seeded weights, CPU oracle (test infrastructure)."""
    from vietvoice_tts_amd.model_spec import ModelSpec, make_synthetic_weights
    from oracle.vv_oracle import Oracle
    spec = ModelSpec.tiny()
    w = make_synthetic_weights(spec, seed=9527)
    return spec, w, Oracle(spec, w, nfe_step=8)


@pytest.fixture(scope="session")
def hip_tiny(tiny_setup):
    """HIP engines (fp32 and bf16 acoustic) for the tiny model.  Fails loudly without GPU/library."""
    import torch
    from vietvoice_tts_amd.runtime import HipSynth
    assert torch.cuda.is_available(), "gpu-marked tests need a HIP device"
    spec, w, _ = tiny_setup
    return {"f32": HipSynth(spec, w, acoustic_dtype="fp32", nfe_step=8),
            "bf16": HipSynth(spec, w, acoustic_dtype="bf16", nfe_step=8)}
