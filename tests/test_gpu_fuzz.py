"""Short runs of the randomised parity sweeps (tools/fuzz_front_end.py, tools/fuzz_homography.py: thousands of cases per run when
called by hand, profiles/r03_fuzz_*.txt) so that the suite also covers shapes nobody thought of listing: every output byte
identical to the oracle's.  PARITY UNPINNED (the oracle restates closed-source behaviour)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

TOOLS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")


def _run(module, argv, monkeypatch):
    monkeypatch.syspath_prepend(TOOLS)
    monkeypatch.setattr(sys, "argv", [module] + [str(a) for a in argv])
    return __import__(module).main()


@pytest.mark.parametrize("seed", [11, 12])
def test_thin_layer_randomised_sweep(gpu, monkeypatch, seed):
    assert _run("fuzz_front_end", [250, seed], monkeypatch) == 0


def test_find_homography_randomised_sweep(gpu, monkeypatch):
    assert _run("fuzz_homography", [80, 5], monkeypatch) == 0
