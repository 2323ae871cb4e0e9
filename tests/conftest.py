import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))

    return load


@pytest.fixture(scope="session")
def sd_static():
    from avcer_amd import synth
    return synth.to_torch(synth.static_state_dict(42))


@pytest.fixture(scope="session")
def sd_dynamic():
    from avcer_amd import synth
    return synth.to_torch(synth.dynamic_state_dict(42))


@pytest.fixture(scope="session")
def sd_audio():
    from avcer_amd import synth
    return synth.to_torch(synth.audio_state_dict(42))


# ----------------------------------------------------------------------------- GPU fixtures (used only by -m gpu tests)
@pytest.fixture(scope="session")
def engine():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from avcer_amd.engine import Engine

    return Engine(0)


@pytest.fixture(scope="session")
def engine_static(engine, sd_static):
    engine.load_static(sd_static)
    return engine


@pytest.fixture(scope="session")
def engine_dynamic(engine, sd_dynamic):
    engine.load_dynamic(sd_dynamic)
    return engine


@pytest.fixture(scope="session")
def engine_audio(engine, sd_audio):
    engine.load_audio(sd_audio)
    return engine
