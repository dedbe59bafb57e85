"""One-off: bench.parity_vs_f64 on a larger sample (5,000 pairs of configs[1] over 500 targets + 1,000 configs[3]-style over 10
targets) with the build in the tree: indices against the double table's argmax, exact scores.  -> gpurun_out/r06_parity_6000_pairs.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from nautilus_amd import _lib
_lib.load()
wl = bench.Workload("weak", 1)
out = bench.parity_vs_f64(wl, n_config2=5000, n_config4=1000, cell_bits=16, n_threads=bench._omp_threads())
out["kernel_source_hash"] = bench.kernel_source_hash()
txt = json.dumps(bench._round_floats(out, 9), indent=1)
open(os.path.join(ROOT, "gpurun_out", "r06_parity_6000_pairs.json"), "w").write(txt + "\n")
print({k: v for k, v in out.items() if not isinstance(v, (dict, list))})
