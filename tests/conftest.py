import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle (torch ops) is the slow half of every parity test.  On the GPU box torch defaults to one thread per host
    # CPU (256): the full-size oracle forward then takes ~25 s instead of the ~5 s it takes on 32 threads (bench.py's
    # cpu_baseline uses the same cap) -- the 8-block C3 test went from 11 minutes to under 3.
    import torch

    torch.set_num_threads(min(32, os.cpu_count() or 1))


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def rel_l2(a, b):
    """relative L2 error ||a-b|| / ||b|| on CPU double."""
    import torch

    a = a.detach().to("cpu", torch.float64) if not torch.is_complex(a) else torch.view_as_real(a.detach().cpu()).double()
    b = b.detach().to("cpu", torch.float64) if not torch.is_complex(b) else torch.view_as_real(b.detach().cpu()).double()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))
