import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json

    here = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(here, "constants.json")) as f:
        consts = json.load(f)
    with open(os.path.join(here, "varuna_circuit0.json")) as f:
        varuna = json.load(f)
    with open(os.path.join(here, "srs_g1_1024.bin"), "rb") as f:
        srs = f.read()
    with open(os.path.join(here, "beta_h_g2.bin"), "rb") as f:
        beta_h = f.read()
    with open(os.path.join(here, "srs_g1_32768.bin"), "rb") as f:
        srs_full = f.read()
    return {"constants": consts, "varuna": varuna, "srs_g1": srs, "srs_g1_full": srs_full, "beta_h_g2": beta_h}


@pytest.fixture(scope="session", autouse=True)
def _native_libraries_built():
    """A fresh checkout has no built artefacts (they are git-ignored): compile the HIP backend (hipcc cross-compiles
    without a GPU, a few minutes once) and the CPU oracle before the first test needs them."""
    from snarkvm_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        from snarkvm_amd import build as hip_build

        hip_build.build()
    yield


@pytest.fixture(scope="session", autouse=True)
def _oracle_threads():
    """The CPU oracle parallelises with OpenMP; on a many-core GPU host (256 hardware threads) the fork/join cost of
    hundreds of tiny parallel regions dominates small transforms, so the checker is capped at 16 threads."""
    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
    try:
        from oracle import cpu as oracle

        oracle.set_threads(min(16, os.cpu_count() or 1))
    except Exception:
        pass
    yield
