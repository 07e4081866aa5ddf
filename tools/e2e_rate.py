"""tools/e2e_rate.py -- bench.py's end_to_end (SURVEY 8(d)'s rate: pinned frames up, kernels, every result down, two contexts pipelined) alone, for A/B runs."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
torch.cuda.init()
wl = sys.argv[1] if len(sys.argv) > 1 else "euroc752x480_8lvl_1000feat"
sub = int(sys.argv[2]) if len(sys.argv) > 2 else bench.SHAPES[wl][0]
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 2
p = bench.Pipeline(0, wl, sub, 1, 3, 1000, distinct=min(3 * sub, 24 if "752" not in wl else 3 * sub))
n, sec, link = bench.end_to_end(p, min_seconds=1.0, depth=depth)
print(json.dumps({"workload": wl, "sub": sub, "depth": depth, "frames_per_s": round(n / sec, 1), "pcie": bench.pcie_roofline(link, n / sec)}))
