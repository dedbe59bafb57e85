"""nautilus_amd/adapters/nautilus_hip_io.h (the pose file, the map file and the HITL message of SURVEY 8f row 4 for a
C++ host) against the Python mirror in nautilus_amd/hostside.py: files written by one are read by the other and come
back byte for byte.  Host code only (g++), runs without a GPU."""
import os
import subprocess

import numpy as np

from nautilus_amd import hostside

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ADAPTERS = os.path.join(ROOT, "nautilus_amd", "adapters")


def test_pose_map_and_hitl_files_round_trip_through_the_cpp_header(tmp_path):
    subprocess.check_call(["make", "-C", ADAPTERS, "io_test"], stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(5)
    n = 200
    ts = 1583000000.0 + np.cumsum(rng.uniform(0.01, 0.2, n))
    ts[7] = 1583000000.1234565                       # a timestamp that std::fixed rounds at the sixth decimal
    poses = np.concatenate([rng.uniform(-50, 50, (n, 2)), rng.uniform(-3.14, 3.14, (n, 1))], axis=1)
    poses[3] = [1e-7, -123456.789, 0.0]
    # the file holds every node but two, plus a pose nobody asks for
    keep = np.ones(n, bool)
    keep[[10, 150]] = False
    hostside.write_poses(tmp_path / "poses_py.txt", list(ts[keep]) + [5.0], list(poses[keep]) + [[1.0, 2.0, 3.0]])
    with open(tmp_path / "nodes.txt", "w") as f:
        f.write("".join("%.17g\n" % t for t in ts))
    lines = rng.uniform(-30, 30, (60, 4)).astype(np.float32)
    lines[0] = [0, 0, 1.5, 2]
    lines[1] = [-3.25, 1e-5, 123456.7, 1e7]          # %g switches to exponent form as the stream's default format does
    hostside.write_map_lines(tmp_path / "map_py.txt", lines)
    msg = {"line_a_start": (0.5, -1.25, 9.0), "line_a_end": (2.0, 2.5, 9.0), "line_b_start": (1.0, 0.0, -1.0),
           "line_b_end": (1.0, 5.125, 0.0)}
    with open(tmp_path / "hitl.txt", "w") as f:
        for k in ("line_a_start", "line_a_end", "line_b_start", "line_b_end"):
            f.write("%r %r %r\n" % msg[k])
    p = subprocess.run([os.path.join(ADAPTERS, "io_test"), str(tmp_path)], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0 and "IO_OK" in p.stdout, p.stdout + p.stderr
    # LoadSolutionFromFile (main.cc:131-157) in C++ == hostside.load_solution, then WriteCallback == hostside.write_poses
    want, missing = hostside.load_solution(tmp_path / "poses_py.txt", ts, np.zeros((n, 3)))
    assert missing == [10, 150]
    hostside.write_poses(tmp_path / "poses_want.txt", ts, want)
    assert open(tmp_path / "poses_cpp.txt", "rb").read() == open(tmp_path / "poses_want.txt", "rb").read()
    # the reader keeps float precision (main.cc:138: float pose_x, pose_y, theta)
    got = hostside.read_poses(tmp_path / "poses_cpp.txt")
    assert len(got) == n and abs(got[float("%.6f" % ts[20])][0] - poses[20, 0]) < 1e-5
    # Vectorize's map file (solver.cc:608-618): byte for byte
    assert open(tmp_path / "map_cpp.txt", "rb").read() == open(tmp_path / "map_py.txt", "rb").read()
    assert open(tmp_path / "map_py.txt").readline().strip() == "0,0,1.5,2"
    assert np.allclose(hostside.read_map_lines(tmp_path / "map_cpp.txt"), lines, rtol=1e-5)
    rep = open(tmp_path / "report.txt").read().splitlines()
    assert rep[0] == "nodes 200 lines 60 missing 10 150"
    seg = np.array([[float.fromhex(v) for v in r.split()[1:]] for r in rep[1:]], dtype=np.float32)
    assert seg.tobytes() == hostside.hitl_segments(msg).tobytes()      # LineSegmentsFromHitlMsg, solver.cc:467-478
