import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    path = os.path.join(ROOT, "tests", "golden", "stft_golden.npz")
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle

    oracle.build_oracle()
    return oracle


GOLDEN_GEOMETRIES = [(256, 64), (1024, 256), (2048, 512)]
GOLDEN_SIGNALS = ["impulse", "dc", "nyquist", "sine_on_bin", "sine_off_bin", "lcg_noise", "silence",
                  "corpus"]

# Tolerance of the path (BASELINE.json north_star): <= 1e-5 relative for float bins,
# measured per frame as max_k|out-ref| / max_k|ref|.  Against the f64
# truth we hold the tighter 5e-6 that implies 1e-5 against any correct f32 CPU FFT.
TOL_VS_TRUTH = 5e-6
TOL_VS_ORACLE = 1e-5
