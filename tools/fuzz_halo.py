"""One-off fuzz of the HALO search (grid over the bulk of the gold map, stray points in its boundary cells): random scenes
(boxes, planes, lines of different aspect), 1 ... 600 stray points at 1.3 ... 2000 scene-widths — some of them inside the
scene's range along one or two axes — queries inside, around the strays and all over; ma_nn gated and ungated, identity
and a random rigid start, against an exhaustive search on the coordinates the device stores (tests/test_gpu_parity.py
assert_nn_exact_on_stored: every index exact up to fp32 ties checked one by one).  Prints the number of cases and of
cases in which the grid covered the bulk only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-localisation-and-mapping_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mapalign
from mapalign import synthetic as syn
from test_gpu_parity import assert_nn_exact_on_stored

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
eng = mapalign.MapAligner(0)


def loop_fuzz(n_loop):
    """Whole alignments of pairs with stray points in both clouds: candidate lists on against off — transform and the 17
    sums of every iteration bitwise (a list that missed a neighbour, or a stray's list trusted beyond its rounding
    allowance, would show)."""
    from mapalign import synthetic as syn
    bad = halo = lists_read = 0
    for case in range(n_loop):
        rng = np.random.default_rng(5000 + case)
        n = int(rng.choice([20_000, 60_000, 150_000]))
        gold, comp, _, _ = syn.make_pair(100 + case, n, n, angle_deg=float(rng.uniform(0.2, 1.5)), t0=rng.normal(0, 0.03, size=3))
        lo, hi = gold.min(0), gold.max(0)
        ctr, ext = 0.5 * (lo + hi), float((hi - lo).max())
        kg, kc = int(rng.choice([0, 5, 100])), int(rng.choice([0, 5, 100]))
        scale = float(rng.choice([3.0, 40.0, 1500.0]))
        mk = lambda k: (ctr + rng.uniform(-0.5, 0.5, size=(k, 3)) * ext * scale).astype(np.float32)
        g = np.concatenate([gold, mk(kg)]) if kg else gold
        c = np.concatenate([comp, mk(kc)])[rng.permutation(len(comp) + kc)] if kc else comp
        eng.set_gold(g)
        halo += int(eng.grid_box()["bulk_only"])
        eng.set_source(c)
        gate = float(rng.choice([0.05, 0.10, 0.25]))
        on = eng.icp(60, 0.0, gate, with_trace=True)
        lists_read += int((eng.list_plan()[:60] == 2).sum())
        off = eng.icp(60, 0.0, gate, with_trace=True, lists=False)
        if not (np.array_equal(on.T, off.T) and np.array_equal(on.trace[:, :17], off.trace[:, :17])):
            bad += 1
            print(f"loop case {case}: lists on / off differ", flush=True)
    print(f"{n_loop} alignments (60 iterations; 0 / 5 / 100 stray points per cloud at 3 / 40 / 1500 scene-widths), {halo} with the grid over "
          f"the bulk only, {lists_read} list launches: {bad} differ between lists on and off")
    return bad


halo = bad = 0
for case in range(n_cases):
    rng = np.random.default_rng(1000 + case)
    size = rng.uniform(0.05, 1.0, size=3) * rng.choice([1.0, 5.0, 30.0])
    if case % 5 == 0:
        size[rng.integers(0, 3)] = 0.0                       # a plane
    n = int(rng.choice([800, 3000, 12000]))
    scene = (rng.uniform(-0.5, 0.5, size=(n, 3)) * size + rng.uniform(-50, 50, size=3))
    if case % 3 == 0:                                        # a shell: points on the faces
        ax = rng.integers(0, 3, size=n)
        scene[np.arange(n), ax] = (np.round(rng.uniform(0, 1, size=n)) - 0.5) * size[ax] + scene[:, ax].mean() * 0
    ctr, ext = 0.5 * (scene.min(0) + scene.max(0)), max(float((scene.max(0) - scene.min(0)).max()), 1e-3)
    k = int(rng.choice([1, 3, 20, 200, 600]))
    k = min(k, max(1, n // 40))                               # (<= 2.5 % of the map: the box of the bulk is taken up to 3 %)
    scale = float(rng.choice([1.3, 3.0, 20.0, 300.0, 2000.0]))
    st = ctr + rng.uniform(-0.5, 0.5, size=(k, 3)) * ext * scale
    # some strays share the scene's range along one or two axes (they lie beyond the box along the others only)
    for j in range(k):
        if rng.uniform() < 0.5:
            axes = rng.choice(3, size=int(rng.integers(1, 3)), replace=False)
            st[j, axes] = scene[rng.integers(0, n), axes]
    gold = np.concatenate([scene, st]).astype(np.float32)
    gold = gold[rng.permutation(len(gold))]
    eng.set_gold(gold)
    halo += int(eng.grid_box()["bulk_only"])
    gate = float(rng.choice([0.02, 0.1, 0.5])) * ext
    near = st[rng.integers(0, k, size=600)] + rng.normal(0, gate * 0.7, size=(600, 3))
    wide = ctr + rng.uniform(-0.5, 0.5, size=(600, 3)) * ext * scale * 1.3
    inside = scene[rng.integers(0, n, size=800)] + rng.normal(0, gate * 0.5, size=(800, 3))
    q = np.concatenate([inside, near, wide, st]).astype(np.float32)
    eng.set_source(q)
    T = np.eye(4)
    if case % 2:
        T[:3, :3] = syn.rotation_about_axis(rng.normal(size=3), rng.uniform(0, 0.2))
        T[:3, 3] = rng.normal(0, 0.05 * ext, size=3)
    for g in (gate, np.inf):
        try:
            idx, d2 = eng.nn(T, g)
            assert_nn_exact_on_stored(eng, T, g, idx, d2)
        except AssertionError as e:
            bad += 1
            print(f"case {case} gate {g}: {str(e)[:200]}", flush=True)
print(f"{n_cases} cases x 2 gates, {halo} with the grid over the bulk only, {bad} disagree with the exhaustive search")
bad += loop_fuzz(int(sys.argv[2]) if len(sys.argv) > 2 else 40)
sys.exit(1 if bad else 0)
