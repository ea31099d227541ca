import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure), built on demand with gcc."""
    from oracle import oracle as orc
    orc.build()
    return orc


class Cfg:
    """Attribute bag standing in for the namedtuple parse_ini returns (consumers use getattr(cfg, k, default))."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


# ---------------------------------------------------------------------------------------------- parity report
# Every parity assert of the GPU tests goes through `parity(...)`: it records the ACHIEVED error next to the bound it
# is asserted against (and, where there is one, the fp32-vs-fp64 gap of the reference / fp32 oracle on the same inputs),
# then asserts.  The table is printed in the terminal summary and written to gpurun_out/parity_report_{gpu,cpu}.md.
_PARITY_ROWS = []


class _Parity:
    def __init__(self, test):
        self.test = test

    def __call__(self, quantity, achieved, bound, yardstick=None):
        achieved, bound = float(achieved), float(bound)
        _PARITY_ROWS.append((self.test, quantity, achieved, bound, None if yardstick is None else float(yardstick)))
        assert achieved <= bound, "%s: %s achieved %.3e > bound %.3e" % (self.test, quantity, achieved, bound)


@pytest.fixture
def parity(request):
    return _Parity(request.node.name)


def pytest_terminal_summary(terminalreporter):
    if not _PARITY_ROWS:
        return
    lines = ["| test | quantity | achieved | asserted <= | yardstick: the reference's own fp32-vs-fp64 gap / self-noise |", "|---|---|---|---|---|"]
    for t, q, a, b, y in _PARITY_ROWS:
        lines.append("| %s | %s | %.2e | %.2e | %s |" % (t, q, a, b, "" if y is None else "%.2e" % y))
    terminalreporter.write_line("")
    terminalreporter.write_line("parity report (achieved error vs asserted bound):")
    for ln in lines:
        terminalreporter.write_line(ln)
    try:
        out = os.path.join(REPO, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        import torch
        name = "parity_report_gpu.md" if torch.cuda.is_available() else "parity_report_cpu.md"
        with open(os.path.join(out, name), "w") as f:
            f.write("\n".join(lines) + "\n")
    except OSError:
        pass
