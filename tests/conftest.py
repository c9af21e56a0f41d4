import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import sarssl_boot  # noqa: E402,F401  (makes `import sar_ssl_amd` work even without the symlink)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gold_dir():
    return GOLD


def has_gpu():
    import torch
    return torch.cuda.is_available()


_PARITY_LOG = os.path.join(ROOT, "gpurun_out", "parity_measured.jsonl")


def check(name, measured, tol):
    """Parity gate that also records what was measured: prints ``PARITY <name> measured=<m> tol=<t>`` (visible with -s / on
    failure) and appends it to gpurun_out/parity_measured.jsonl when that directory exists, so tolerances can be kept at a small
    multiple of the measured deviation instead of a guess."""
    import json
    measured = float(measured)
    print("PARITY %s measured=%.3e tol=%.3e" % (name, measured, tol))
    try:
        if os.path.isdir(os.path.dirname(_PARITY_LOG)):
            with open(_PARITY_LOG, "a") as f:
                f.write(json.dumps({"name": name, "measured": measured, "tol": tol}) + "\n")
    except OSError:
        pass
    assert measured < tol, "%s: measured %.3e >= tolerance %.3e" % (name, measured, tol)
