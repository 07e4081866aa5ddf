"""CPU tier: the N > 1 path of bench.py (one process per GPU, barrier, max-over-ranks reduction, rank-0 JSON) exercised
with world_size 2 on the gloo backend, with real data: the ranks share a clip round-robin, run the CPU oracle on their shards and
all-gather the per-frame keypoint counts, which must equal an unsharded run.  The line is marked as not a measurement."""
import json
import os
import socket
import subprocess
import sys

from tests.conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_plumbing():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--plumbing-selftest"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # only rank 0 prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["plumbing_selftest"] and not d["valid_measurement"]
    assert d["max_elapsed_s"] >= 0.02           # the MAX over ranks (rank 1 sleeps 20 ms)
    # the sharded counts equal an unsharded oracle run over the same clip
    sys.path.insert(0, ROOT)
    from oracle import oracle_py as O
    from orb_ygz_slam_amd.synth import synth_frame
    oex = O.Extractor(500, 1.2, 4, 20, 7)
    want = [len(oex.extract(synth_frame(4000 + i, 320, 240))[0]) for i in range(8)]
    assert d["shard_keypoint_counts"] == want and min(want) > 50


def test_algorithmic_bytes_match_survey():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.algorithmic_bytes(752, 480, 8, 1.2, 1000)[0] == 6518735 + 72000
    assert bench.algorithmic_bytes(640, 480, 8, 1.2, 1000)[0] == 5742474 + 72000
    assert bench.algorithmic_bytes(1920, 1080, 8, 1.2, 4000)[0] == 35145669 + 4 * 72000
    assert bench.level_sizes(3840, 2160, 12, 1.2)[-1] == (517, 291)
