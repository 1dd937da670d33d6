import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def golden_names(prefix):
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith(".npz"))


@pytest.fixture(scope="session")
def hip_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


# ---- parity record: what the GPU tests OBSERVED (per fixture: rel-Fro, flipped codes, rows of the packed buffer that are
# bit-identical to the reference's), written to gpurun_out/parity.json at the end of every `-m gpu` session; the copy
# under profiles/rNN_parity.json is the tracked one.
PARITY = {}


def record_parity(name, **fields):
    PARITY[name] = fields


def pytest_sessionfinish(session, exitstatus):
    if not PARITY:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity.json"), "w") as f:
            json.dump({"exitstatus": int(exitstatus), "fixtures": PARITY}, f, indent=1, sort_keys=True)
    except OSError:
        pass
