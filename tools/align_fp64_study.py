"""tools/align_fp64_study.py [n_seeds] -- the aligner's tolerance argument with a third party, on the CPU (no GPU needed).

For every seed of tests/test_gpu_fuzz.py::test_fuzz_sparse_img_align (same scene, same parameters): the oracle's reference-order mode (fp32, the
reference's own pixel-by-pixel sums), its device-order mode (fp32 in k_sia_run's formulation; the GPU test demands the kernel's bits equal it) and
the fp64 evaluation of the same Gauss-Newton (oracle/oracle_align.cpp, sparse_img_align_f64).  Prints |device - fp64| beside |reference - fp64| on
the seven SE3 parameters and the reference-order mode's own band under feature permutation; writes gpurun_out/align_fp64_study.json.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O  # noqa: E402
from orb_ygz_slam_amd.capi import EUROC  # noqa: E402
from orb_ygz_slam_amd.scene import two_view_scene  # noqa: E402


def case(seed):
    """the parameters of tests/test_gpu_fuzz.py::test_fuzz_sparse_img_align, evaluated with the oracle's extractor (bit-equal to the device's)"""
    rng = np.random.default_rng(1400 + seed)
    w, h = 752, 480
    nl = int(rng.integers(3, 9))
    wild = seed % 4 == 3
    nf = int(rng.choice([60, 300, 1000, 2000] if wild else [300, 1000, 2000]))
    rv = tuple(rng.uniform(-0.01, 0.01, 3))
    tr = tuple(rng.uniform(-0.05, 0.05, 3))
    imgA, imgB, _, backproject = two_view_scene(1500 + seed, w, h, EUROC, Z=float(rng.uniform(2, 8)), rotvec=rv, trans=tr)
    oex = O.Extractor(nf, 1.2, nl, 20, 7)
    k, _ = oex.extract(imgA)
    pyrA, pyrB = oex.pyramid(imgA), oex.pyramid(imgB)
    world = backproject(k["x"], k["y"])
    inv = oex.tables()["inv_scale"]
    if wild:
        max_level = int(rng.integers(1, nl))
        min_level = int(rng.integers(0, max_level + 1))
        n_iter = int(rng.choice([1, 3, 10]))
    else:
        max_level = int(rng.integers(2, nl))
        min_level = int(rng.integers(0, max_level))
        n_iter = int(rng.choice([3, 10, 10]))
    valid = (rng.uniform(size=len(k)) > 0.2).astype(np.uint8)
    outl = (rng.uniform(size=len(k)) > 0.9).astype(np.uint8)
    return dict(k=k, world=world, pyrA=pyrA, pyrB=pyrB, inv=inv, max_level=max_level, min_level=min_level, n_iter=n_iter, valid=valid, outl=outl, rng=rng,
                desc=dict(seed=seed, levels=nl, features=int(len(k)), max_level=max_level, min_level=min_level, n_iter=n_iter))


def evaluate(c):
    ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    a = (c["k"], c["world"], ident, c["pyrA"], ident, c["pyrB"], c["inv"], EUROC, c["max_level"], c["min_level"], c["n_iter"])
    o = O.sparse_img_align(*a, mp_valid=c["valid"], outlier=c["outl"])
    d = O.sparse_img_align(*a, mp_valid=c["valid"], outlier=c["outl"], device_order=True)
    f = O.sparse_img_align_f64(*a, mp_valid=c["valid"], outlier=c["outl"])
    band = 0.0
    rng, n = c["rng"], len(c["k"])
    for perm in (np.arange(n)[::-1], rng.permutation(n), rng.permutation(n)):
        op = O.sparse_img_align(c["k"][perm], c["world"][perm], ident, c["pyrA"], ident, c["pyrB"], c["inv"], EUROC, c["max_level"], c["min_level"], c["n_iter"],
                                mp_valid=c["valid"][perm], outlier=c["outl"][perm])
        band = max(band, float(np.abs(op[1] - o[1]).max()))
    r = dict(c["desc"])
    r.update(band=band, dev_vs_ref=float(np.abs(d[1] - o[1]).max()), dev_vs_f64=float(np.abs(d[1].astype(np.float64) - f[1]).max()),
             ref_vs_f64=float(np.abs(o[1].astype(np.float64) - f[1]).max()), n_meas=(int(o[0]), int(d[0]), int(f[0])),
             iters=(int(o[2][0]), int(d[2][0]), int(f[2][0])))
    return r


def summarise(rows):
    out = {}
    for name, sel in (("well_conditioned", [r for r in rows if r["band"] < 1e-6]), ("ill_conditioned", [r for r in rows if r["band"] >= 1e-6])):
        if not sel:
            continue
        dv, rf = np.array([r["dev_vs_f64"] for r in sel]), np.array([r["ref_vs_f64"] for r in sel])
        out[name] = {"cases": len(sel), "device_vs_fp64_median": float(np.median(dv)), "device_vs_fp64_max": float(dv.max()),
                     "reference_order_vs_fp64_median": float(np.median(rf)), "reference_order_vs_fp64_max": float(rf.max()),
                     "device_no_further_than_reference_order": int((dv <= rf).sum()),
                     "device_within_2x_of_reference_order_or_1e-6": int((dv <= np.maximum(2 * rf, 1e-6)).sum())}
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    rows = []
    for seed in range(n):
        r = evaluate(case(seed))
        rows.append(r)
        print("seed %3d  L=%d N=%4d lv %d..%d it=%2d  band %.2e  dev-ref %.2e | dev-f64 %.2e  ref-f64 %.2e  iters %s" %
              (seed, r["levels"], r["features"], r["max_level"], r["min_level"], r["n_iter"], r["band"], r["dev_vs_ref"], r["dev_vs_f64"], r["ref_vs_f64"], r["iters"]))
    s = summarise(rows)
    print(json.dumps(s, indent=1))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump({"summary": s, "cases": rows}, open(os.path.join(ROOT, "gpurun_out", "align_fp64_study.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
