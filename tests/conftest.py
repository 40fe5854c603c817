import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")
    # the oracle runs thousands of tiny torch ops; on a 128-core host the default intra-op pool makes
    # each of them ~ms slow
    import torch
    torch.set_num_threads(min(8, torch.get_num_threads()))
    # a fresh checkout has no libmmk_hip.so (built artefacts stay out of the history), and one left over from other sources must not be
    # tested: (re)build when the digest compiled into the library is not the digest of these sources (~80 s with hipcc, 8 cores)
    try:
        from mimikit_amd import build as hip_build
        if hip_build._stale():
            print("[conftest] libmmk_hip.so is missing or was built from other sources: building it", file=sys.stderr)
            hip_build.build()
    except Exception as err:      # (no hipcc here: the tests that need the library say so themselves)
        print(f"[conftest] could not build libmmk_hip.so: {err}", file=sys.stderr)


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda", 0)
