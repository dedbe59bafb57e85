"""tools/parity_sweep.py --quick under -m gpu: 100 random configurations (grid geometry, blur, lattice shape, search
centre, dense / sparse / clustered clouds, both cell widths) through the branch-and-bound matcher, the kernel that
performs every add of the same width, and the CPU oracle -- grids, best-pose indices and integer sums bit-exact.
(The builder's longer runs of the same script: 600 configurations per kernel change, 3,000 per round.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_quick_parity_sweep(gpu):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "parity_sweep.py"), "--quick"], capture_output=True,
                       text=True, timeout=600)
    print(p.stdout[-2000:], p.stderr[-2000:])
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "sweep ok: 100 configurations" in p.stdout
