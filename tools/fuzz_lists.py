"""One-off fuzz of the candidate lists (DESIGN.md 5 "List launches"): random shapes, sizes, gates (down to ones that leave most
points unmatched, up to none), weights, cell edges and starts; 40-64 iterations with lists on and with every launch searching
(MA_FLAG_NO_LISTS) must give the same transform and the same 17 sums in every iteration, bit for bit.  The 200 cases of seed 31
are tests/test_gpu_round4.py::test_lists_on_and_off_bitwise_over_200_random_cases; this tool runs more of them.
    python tools/fuzz_lists.py [cases] [seed]      (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-localisation-and-mapping_amd")); sys.path.insert(0, ROOT)
import numpy as np, mapalign
from mapalign import synthetic as syn
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 31)
eng = mapalign.MapAligner(0)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = used = done = degenerate = 0
uses_total = 0
for case in range(cases):
    m, n = int(rng.integers(300, 120000)), int(rng.integers(300, 120000))
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
    if case % 5 == 0:       # strays and far points: queries without a gold point inside the gate
        k = n // 10
        comp_in_gold[:k] += rng.normal(0, 0.08, (k, 3))
        comp_in_gold[k:k + k // 2] += 3.0
    R0 = syn.rotation_about_axis(rng.normal(size=3), rng.uniform(0, 0.01)); t0 = rng.normal(0, 0.006, 3)
    comp = ((comp_in_gold - t0) @ R0).astype(np.float32)
    gate = [0.02, 0.05, 0.1, 0.5, np.inf][int(rng.integers(0, 5))]
    w = rng.uniform(0.0, 3.0, n).astype(np.float32) if case % 3 == 0 else None
    cell = [0.0, 0.0, 0.0, 0.02, 0.2][int(rng.integers(0, 5))]
    iters = int(rng.integers(40, 65))
    out = {}
    try:
        eng.set_gold(gold, cell); eng.set_source(comp, w)
        for mode in ("1", "0"):
            r = eng.icp(iters, 0.0, gate, with_trace=True, lists=mode == "1")
            out[mode] = (r.T.copy(), r.trace.copy(), eng.list_stats(), "".join("SWL"[int(v)] for v in eng.list_plan()))
    except mapalign.MapAlignError:
        degenerate += 1
        continue
    done += 1
    st = out["1"][2]
    used += st["uses"] > 0
    uses_total += st["uses"]
    if not (np.array_equal(out["1"][0], out["0"][0]) and np.array_equal(out["1"][1][:, :18], out["0"][1][:, :18])):
        bad += 1
        d = np.abs(out["1"][0] - out["0"][0]).max()
        first = int(np.argmax((out["1"][1][:, :18] != out["0"][1][:, :18]).any(1)))
        a, b = out["1"][1][first, :18], out["0"][1][first, :18]
        cols = [(int(c), float((a[c] - b[c]) / max(abs(b[c]), 1e-300))) for c in np.nonzero(a != b)[0]]
        plan = eng.list_plan() if False else None
        print("case %d (kind %d m %d n %d gate %s cell %s w %s iters %d, %s): max|dT| %.2e, first differing iteration %d, columns (rel diff) %s" % (
            case, kind, m, n, gate, cell, w is not None, iters, st, d, first, cols), flush=True)
        print("   plan", out["1"][3], flush=True)
        # a tie or a miss?  E = sum w d^2: a list that missed a nearer neighbour makes E larger by the difference of two
        # squared distances (>= 1e-7 m^2 here); two gold points at the same fp32 distance differ by rounding (~1e-12 m^2)
        print("   E on - off at that iteration: %.3e m^2 (E %.3e, W %.1f / %.1f); sum(w q) on - off: %s" % (
            a[16] - b[16], b[16], a[0], b[0], (a[4:7] - b[4:7])), flush=True)
print("%d cases compared (%d degenerate), lists used in %d of them (%d list launches in total), %d differ between lists on and off" % (
    done, degenerate, used, uses_total, bad))
