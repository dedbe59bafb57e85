"""The C++ drop-ins (nautilus_amd/adapters/): CorrelativeScanMatcher used as solver.cc:633-644
uses it, and the slam_residuals replacement driven like Ceres' evaluation loop.  The binary is
built by __graft_entry__.build() (g++, links libnautilus_hip.so) and needs a GPU to run."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "nautilus_amd", "adapters", "adapter_test")


def test_adapter_headers_mirror_reference_signatures():
    h = open(os.path.join(ROOT, "nautilus_amd", "adapters", "CorrelativeScanMatcher.h")).read()
    assert "class CorrelativeScanMatcher" in h and "GetTransformation(" in h
    assert "double scanner_range, double trans_range, double low_res, double high_res" in h
    r = open(os.path.join(ROOT, "nautilus_amd", "adapters", "slam_residuals_hip.h")).read()
    for s in ("struct OdometryResidual", "struct LIDARNormalResidual", "struct LIDARPointResidual",
              "struct PointToLineResidual", "namespace nautilus {", "static nautilus_hip::BatchedCost *create(",
              "double translation_weight, double rotation_weight", "line_segment, const std::vector<"):
        assert s in r


@pytest.mark.gpu
def test_adapter_binary_on_gpu(gpu):
    if not os.path.exists(BIN):
        subprocess.check_call(["make", "-C", os.path.dirname(BIN)])
    p = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    print(p.stdout, p.stderr)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "ADAPTER_OK" in p.stdout
