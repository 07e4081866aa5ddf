"""CPU tier: the oracle replays tests/golden/paths_golden.npz (made by tools/make_golden_paths.py): matcher modes, frustum,
distinctive descriptors, DSO, stereo, aligner, direct projection.  PARITY UNPINNED w.r.t. the reference (it ships no vectors for these
paths); the fixture pins the oracle's definition so that it cannot drift between rounds."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "paths_golden.npz"))


def check_against_golden(results, float_tol=None):
    """results: iterable of (name, dict); float_tol: {(name, key): tolerance} for entries compared by value instead of digest."""
    from make_golden_paths import sha
    float_tol = float_tol or {}
    seen = 0
    for name, res in results:
        for key, arr in res.items():
            arr = np.asarray(arr)
            tag = "%s_%s" % (name, key)
            if (name, key) in float_tol:
                assert np.abs(arr.astype(np.float64) - GOLD[tag].astype(np.float64)).max() <= float_tol[(name, key)], tag
            else:
                assert sha(arr).tobytes() == GOLD[tag + "_sha"].tobytes(), tag
            seen += 1
    assert seen == sum(1 for k in GOLD.files if k.endswith("_sha"))


def test_oracle_replays_path_goldens(oracle):
    from make_golden_paths import cases
    check_against_golden(cases(oracle))
