"""One-off fuzz of the whole loop against the oracle: random sizes, shapes, gates, weights, cells, starts; 6 iterations each
(GPU box).  Prints the cases that disagree beyond the parity bars.     python tools/fuzz_icp.py [cases] [seed] [look_at_case]
With a third argument: that case alone, iteration by iteration — and where the two sides part, the correspondences that differ
(the device ranks candidates by fp32 squared distances of fp32 positions, the oracle in fp64: near-equidistant candidates)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-localisation-and-mapping_amd")); sys.path.insert(0, ROOT)
import numpy as np, mapalign
from mapalign import synthetic as syn
from oracle import icp_oracle as orc
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
eng = mapalign.MapAligner(0)
bad = done = degenerate = 0
look = int(sys.argv[3]) if len(sys.argv) > 3 else -1
for case in range(int(sys.argv[1]) if len(sys.argv) > 1 else 160):
    m, n = int(rng.integers(50, 20000)), int(rng.integers(50, 20000))
    kind = case % 4
    if kind == 0:
        gold = rng.uniform(-1, 1, (m, 3)) * rng.uniform(0.3, 5.0, 3)
    elif kind == 1:
        gold = rng.uniform(-3, 3, (m, 3)); gold[:, rng.integers(0, 3)] *= 0.01
    elif kind == 2:
        c = rng.uniform(-2, 2, (max(m // 300, 2), 3)); gold = c[rng.integers(0, len(c), m)] + rng.normal(0, 0.05, (m, 3))
    else:
        gold = syn.sample_rectangles(syn.carriage_rectangles(rng), m, rng)
    gold = gold.astype(np.float32)
    pick = rng.integers(0, m, n)
    comp_in_gold = gold[pick].astype(np.float64) + rng.normal(0, 0.003, (n, 3))
    R0 = syn.rotation_about_axis(rng.normal(size=3), rng.uniform(0, 0.02)); t0 = rng.normal(0, 0.01, 3)
    comp = ((comp_in_gold - t0) @ R0).astype(np.float32)
    gate = [0.03, 0.1, 0.5, np.inf][int(rng.integers(0, 4))]
    w = rng.uniform(0.1, 3.0, n).astype(np.float32) if case % 3 == 0 else None
    cell = [0.0, 0.0, 0.02, 0.2, 1.5][int(rng.integers(0, 5))]
    if look >= 0 and case != look: continue
    if look >= 0:
        eng.set_gold(gold, cell); eng.set_source(comp, w)
        tree = orc.build_index(gold) if hasattr(orc, "build_index") else None
        for k in range(1, 7):
            res = eng.icp(k, 0.0, gate)
            ref = orc.icp_oracle(gold, comp, max_iter=k, max_corr_dist=gate, weights=w, n_jobs=4)
            eR, et = orc.rt_error(res.R, res.t, ref["R"], ref["t"])
            print("after %d iterations: eR %.2e et %.2e inliers %s vs %s" % (k, eR, et, res.inliers, ref["inliers"]))
        # the correspondences at the oracle's transform after k iterations, k = 0..5: device (ma_nn) against brute force in fp64
        for k in range(0, 6):
            T = np.eye(4)
            if k:
                ref = orc.icp_oracle(gold, comp, max_iter=k, max_corr_dist=gate, weights=w, n_jobs=4)
                T[:3, :3] = ref["R"]; T[:3, 3] = ref["t"]
            idx, d2 = eng.nn(T, gate)
            pts = comp.astype(np.float64) @ T[:3, :3].T + T[:3, 3]
            bj, bd2 = orc.nn_bruteforce(gold.astype(np.float64), pts)
            inside = bd2 <= gate * gate if np.isfinite(gate) else np.ones(len(bd2), bool)
            diff = np.nonzero((idx.astype(np.int64) != np.where(inside, bj, idx.astype(np.int64))) & inside)[0]
            gaps = []
            for i in diff[:2000]:
                if idx[i] == 0xFFFFFFFF: gaps.append(np.inf); continue
                dg = ((pts[i] - gold[idx[i]].astype(np.float64)) ** 2).sum()
                gaps.append((np.sqrt(dg) - np.sqrt(bd2[i])))
            gaps = np.array(gaps)
            print("transform after %d oracle iterations: %d of %d queries name another gold point than the fp64 brute force; "
                  "their distance is larger by at most %.3e m (median %.1e); distance between the two gold points: median %.3f m" % (
                      k, len(diff), len(idx), gaps.max() if len(gaps) else 0.0, np.median(gaps) if len(gaps) else 0.0,
                      float(np.median([np.linalg.norm(gold[idx[i]].astype(np.float64) - gold[bj[i]]) for i in diff[:2000] if idx[i] != 0xFFFFFFFF])) if len(diff) else 0.0))
        break
    try:
        eng.set_gold(gold, cell); eng.set_source(comp, w)
        res = eng.icp(6, 0.0, gate)
    except mapalign.MapAlignError as e:
        ref = orc.icp_oracle(gold, comp, max_iter=6, max_corr_dist=gate, weights=w, n_jobs=4)
        degenerate += 1
        if ref["ok"]:
            bad += 1; print("case %d: GPU error %s but oracle ok" % (case, e))
        continue
    ref = orc.icp_oracle(gold, comp, max_iter=6, max_corr_dist=gate, weights=w, n_jobs=4)
    done += 1
    eR, et = orc.rt_error(res.R, res.t, ref["R"], ref["t"])
    if not ref["ok"] or eR > 1e-6 or et > 1e-5 or abs(res.inliers - ref["inliers"]) > 1e-3 * max(ref["inliers"], 1):
        bad += 1
        print("case %d (kind %d m %d n %d gate %s cell %s w %s): eR %.2e et %.2e inliers %s vs %s ok %s" % (case, kind, m, n, gate, cell, w is not None, eR, et, res.inliers, ref["inliers"], ref["ok"]))
print("%d cases compared, %d degenerate on both sides, %d disagree" % (done, degenerate - 0, bad))
