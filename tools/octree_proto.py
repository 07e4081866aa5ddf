#!/usr/bin/env python3
"""Array/range formulation of ORBextractor::DistributeOctTree (prototype of the HIP kernel k_octree).

Points get a 'path key' (root index, then 2 bits per subdivision: bit0 = x >= midX, bit1 = y >= midY, following
ExtractorNode::DivideNode's ceil-halving, reference src/ORBextractor.cc:479-531).  After sorting by key every octree
node is a contiguous range, so the breadth-first passes of :578-700 operate on (lo, cnt, depth) triples only.
Validated against the oracle's literal std::list restatement in tests/test_octree_proto.py.
"""
import math
import numpy as np


def path_keys(xs, ys, W, H, nIni, hX, D):
    keys = np.zeros(len(xs), np.int64)
    for i, (x, y) in enumerate(zip(xs.tolist(), ys.tolist())):
        root = int(np.float32(x) / np.float32(hX))
        root = min(root, nIni - 1)
        xl = int(np.float32(hX) * np.float32(root))
        xr = int(np.float32(hX) * np.float32(root + 1))
        yl, yr = 0, H
        k = root
        for _ in range(D):
            mx = xl + ((xr - xl + 1) >> 1)
            my = yl + ((yr - yl + 1) >> 1)
            bx = 1 if x >= mx else 0
            by = 1 if y >= my else 0
            if bx: xl = mx
            else: xr = mx
            if by: yl = my
            else: yr = my
            k = (k << 2) | (by << 1) | bx
        keys[i] = k
    return keys


def distribute(xs, ys, resp, minX, maxX, minY, maxY, N):
    xs = np.asarray(xs); ys = np.asarray(ys); resp = np.asarray(resp)
    M = len(xs)
    W, H = maxX - minX, maxY - minY
    nIni = int(math.floor(np.float32(W) / np.float32(H) + 0.5))  # std::round of a positive float
    nIni = max(nIni, 1)
    hX = np.float32(W) / np.float32(nIni)
    rootW = max(int(np.float32(hX) * np.float32(i + 1)) - int(np.float32(hX) * np.float32(i)) for i in range(nIni))
    D = max(1, math.ceil(math.log2(max(rootW, H, 2)))) + 1
    keys = path_keys(xs, ys, W, H, nIni, hX, D)
    order = np.argsort(keys, kind="stable")
    skeys = keys[order]

    def children(lo, cnt, depth):
        shift = 2 * (D - (depth + 1))
        dig = (skeys[lo:lo + cnt] >> shift) & 3
        b = [lo + int(np.searchsorted(dig, c, side="left")) for c in (1, 2, 3)]
        bounds = [lo] + b + [lo + cnt]
        return [(bounds[c], bounds[c + 1] - bounds[c], depth + 1) for c in range(4)]

    # initial roots (list order = root order), empty ones dropped
    rootdig = skeys >> (2 * D)
    nodes = []
    for r in range(nIni):
        lo = int(np.searchsorted(rootdig, r, side="left"))
        hi = int(np.searchsorted(rootdig, r, side="right"))
        if hi > lo:
            nodes.append((lo, hi - lo, 0))
    finish = False
    while not finish:
        prev = len(nodes)
        kids_blocks, singles, E = [], [], []
        for (lo, cnt, d) in nodes:           # front to back
            if cnt == 1:
                singles.append((lo, cnt, d))
            else:
                ch = [c for c in children(lo, cnt, d) if c[1] > 0]
                kids_blocks.append(ch)
                E.extend(c for c in ch if c[1] > 1)   # creation order: n1..n4
        newnodes = []
        for ch in reversed(kids_blocks):
            newnodes.extend(reversed(ch))     # push_front order: n4 ... n1 at the front
        nodes = newnodes + singles
        nToExpand = len(E)
        if len(nodes) >= N or len(nodes) == prev:
            finish = True
        elif len(nodes) + 3 * nToExpand > N:
            while not finish:
                prev = len(nodes)
                # descending (size, creation seq)
                idx = sorted(range(len(E)), key=lambda j: (E[j][1], j))
                newE, front, processed = [], [], set()
                size = len(nodes)
                for j in reversed(idx):
                    ch = [c for c in children(*E[j]) if c[1] > 0]
                    front = list(reversed(ch)) + front
                    newE.extend(c for c in ch if c[1] > 1)
                    processed.add(E[j])
                    size += len(ch) - 1
                    if size >= N:
                        break
                nodes = front + [n for n in nodes if n not in processed]
                E = newE
                if len(nodes) >= N or len(nodes) == prev:
                    finish = True
    out = []
    for (lo, cnt, d) in nodes:
        ids = order[lo:lo + cnt]
        best = max(ids.tolist(), key=lambda i: (resp[i], -i))
        out.append(best)
    return np.array(out, np.int32)


if __name__ == "__main__":
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle_py as O
    from orb_ygz_slam_amd.synth import synth_frame
    ex = O.Extractor(1000, 1.2, 8, 20, 7)
    tot = 0
    for seed in range(6):
        for (w, h) in ((640, 480), (752, 480), (333, 517)):
            img = synth_frame(seed, w, h)
            pyr = ex.pyramid(img)
            for l in range(8):
                xs, ys, sc = ex.cell_candidates(l)
                if len(xs) == 0: continue
                lw, lh = pyr[l].shape[1], pyr[l].shape[0]
                for N in (ex.tables()["nfeat"][l], 5, 1, 40, 3000):
                    a = ex.octree(xs, ys, sc, 16, lw - 16, 16, lh - 16, int(N))
                    b = distribute(xs, ys, sc, 16, lw - 16, 16, lh - 16, int(N))
                    assert len(a) == len(b) and (a == b).all(), (seed, w, h, l, N, len(a), len(b))
                    tot += 1
    print("octree prototype == oracle on", tot, "cases")
