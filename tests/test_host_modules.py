"""CPU tests of the host-side modules around the path: transform containers
(visualise.py:82-96 formats), PLY ingest (mapping.py:63-68 flags), box transform +
matching (task_def.py:182-197), coarse-stage candidate generation."""
import numpy as np
import pytest

from mapalign import boxes, coarse, ply, synthetic as syn, transforms


def test_rotation_about_centre_matches_open3d_semantics():
    R = syn.rotation_about_axis([0.1, 0.7, -0.3], 0.9)
    c = np.array([1.0, -2.0, 0.5])
    p = np.random.default_rng(0).normal(size=(50, 3))
    assert np.allclose(transforms.apply((R, c), p), (p - c) @ R.T + c)      # p' = R (p - c) + c
    T = transforms.rotation_about_centre(R, c)
    assert transforms.is_rigid(T) and np.allclose(transforms.apply(T, c[None]), c[None])


def test_compose_applies_in_list_order():
    R = syn.rotation_about_axis([0, 0, 1], 0.5)
    stages = [(R, np.array([1.0, 0, 0])), transforms.rt_to_matrix(np.eye(3), [0, 2.0, 0]),
              transforms.rt_to_matrix(syn.rotation_about_axis([1, 0, 0], -0.2), [0.1, 0, 0])]
    p = np.random.default_rng(1).normal(size=(20, 3))
    q = p
    for s in stages:
        q = transforms.apply(s, q)
    assert np.allclose(transforms.apply(transforms.compose(stages), p), q)
    with pytest.raises(ValueError):
        transforms.to_matrix(np.eye(3))


@pytest.mark.parametrize("binary", [True, False])
def test_ply_round_trip_and_filters(tmp_path, binary):
    rng = np.random.default_rng(2)
    pts = rng.normal(size=(200, 3)).astype(np.float32)
    pts[3, 1] = np.nan
    pts[7, 0] = np.inf
    cols = rng.integers(0, 255, size=(200, 3), dtype=np.uint8)
    path = tmp_path / "cloud.ply"
    ply.write_ply(path, pts, cols, binary=binary)
    out, c = ply.read_ply(path, with_colors=True)
    keep = np.isfinite(pts).all(axis=1)
    assert out.shape == (198, 3) and out.dtype == np.float64
    assert np.allclose(out, pts[keep].astype(np.float64), rtol=1e-6) and np.array_equal(c, cols[keep])
    assert ply.read_ply(path, remove_nan_points=False, remove_infinite_points=False).shape == (200, 3)


def test_ply_big_endian_double_and_extra_properties(tmp_path):
    pts = np.arange(30, dtype=np.float64).reshape(10, 3) / 7.0
    rec = np.zeros(10, dtype=np.dtype([("nx", ">f4"), ("x", ">f8"), ("y", ">f8"), ("z", ">f8"), ("label", ">u2")]))
    rec["x"], rec["y"], rec["z"] = pts.T
    head = ("ply\nformat binary_big_endian 1.0\ncomment made by test\nelement vertex 10\nproperty float nx\n"
            "property double x\nproperty double y\nproperty double z\nproperty ushort label\n"
            "element face 0\nproperty list uchar int vertex_indices\nend_header\n")
    path = tmp_path / "be.ply"
    path.write_bytes(head.encode() + rec.tobytes())
    assert np.array_equal(ply.read_ply(path), pts)


def test_ply_coordinates_in_the_files_own_type(tmp_path):
    """dtype=None (what Alignment.compare uses): float32 files come back as float32 — a view of the bytes read for an
    x-y-z-only record, a strided view with colours in the record — with the same values and the same filters; a
    big-endian double file as native float64; scattered or mixed-type coordinates still work."""
    rng = np.random.default_rng(9)
    pts = rng.normal(size=(300, 3)).astype(np.float32)
    plain, coloured = tmp_path / "plain.ply", tmp_path / "coloured.ply"
    ply.write_ply(plain, pts)
    ply.write_ply(coloured, pts, rng.integers(0, 255, size=(300, 3), dtype=np.uint8))
    for path in (plain, coloured):
        out = ply.read_ply(path, dtype=None)
        assert out.dtype == np.float32 and np.array_equal(out, pts)
        assert np.array_equal(ply.read_ply(path), pts.astype(np.float64))       # default: float64, as before
        assert np.array_equal(ply.read_ply(path, dtype=np.float32), pts)
    view = ply.read_ply(plain, dtype=None)
    assert not view.flags.owndata and view.flags.writeable                     # no copy was made, and the caller may edit it
    holes = pts.copy()
    holes[5, 2] = np.nan
    holes[9, 0] = -np.inf
    ply.write_ply(plain, holes)
    out = ply.read_ply(plain, dtype=None)
    assert out.dtype == np.float32 and np.array_equal(out, holes[np.isfinite(holes).all(axis=1)])
    assert len(ply.read_ply(plain, dtype=None, remove_nan_points=False)) == 299
    # big-endian doubles -> native float64; x, y, z not adjacent in the record -> gathered column by column
    p64 = np.arange(30, dtype=np.float64).reshape(10, 3) / 7.0
    rec = np.zeros(10, dtype=np.dtype([("x", ">f8"), ("nx", ">f4"), ("y", ">f8"), ("z", ">f8")]))
    rec["x"], rec["y"], rec["z"] = p64.T
    be = tmp_path / "be.ply"
    be.write_bytes(("ply\nformat binary_big_endian 1.0\nelement vertex 10\nproperty double x\nproperty float nx\n"
                    "property double y\nproperty double z\nend_header\n").encode() + rec.tobytes())
    out = ply.read_ply(be, dtype=None)
    assert out.dtype == np.float64 and out.dtype.isnative and np.array_equal(out, p64)
    rec2 = np.zeros(10, dtype=np.dtype([("x", ">f8"), ("y", ">f8"), ("z", ">f8")]))
    rec2["x"], rec2["y"], rec2["z"] = p64.T
    be.write_bytes(("ply\nformat binary_big_endian 1.0\nelement vertex 10\nproperty double x\n"
                    "property double y\nproperty double z\nend_header\n").encode() + rec2.tobytes())
    out = ply.read_ply(be, dtype=None)
    assert out.dtype == np.float64 and out.dtype.isnative and np.array_equal(out, p64)


def test_ply_errors(tmp_path):
    bad = tmp_path / "bad.ply"
    bad.write_bytes(b"plx\n")
    with pytest.raises(ply.PlyError):
        ply.read_ply(bad)
    trunc = tmp_path / "trunc.ply"
    trunc.write_bytes(b"ply\nformat binary_little_endian 1.0\nelement vertex 5\nproperty float x\nproperty float y\n"
                      b"property float z\nend_header\n" + b"\0" * 10)
    with pytest.raises(ply.PlyError):
        ply.read_ply(trunc)
    empty = tmp_path / "empty.ply"
    ply.write_ply(empty, np.zeros((0, 3)))
    assert ply.read_ply(empty).shape == (0, 3)


def _box(centre, cls=0, conf=0.9, label="exit"):
    c = np.asarray(centre, dtype=np.float64)
    d = 0.1
    return [c + [-d, -d, 0], c + [d, -d, 0], c + [d, d, 0], c + [-d, d, 0], cls, conf, label]


def test_transform_and_match_bboxes(tmp_path):
    gold = {0: [_box([0, 0, 0]), _box([1, 0, 0], label="fire")], 3: [_box([2, 1, 0])]}
    R = syn.rotation_about_axis([0, 0, 1], 0.3)
    T = transforms.rt_to_matrix(R, [0.5, -0.2, 0.1])
    Tinv = np.linalg.inv(T)
    # maintenance scan: same signs seen in its own frame, one missing, one damaged, one new
    comp = {5: [_box(transforms.apply(Tinv, [[0, 0, 0]])[0], cls=2), _box(transforms.apply(Tinv, [[2, 1, 0]])[0])],
            9: [_box(transforms.apply(Tinv, [[5, 5, 0]])[0])]}
    aligned = boxes.transform_bboxes(comp, T)
    assert aligned[5][0][4:] == [2, 0.9, "exit"]                      # metadata untouched (pose_processor.py:208)
    centres, _ = boxes.flatten(aligned)
    # a rotated square keeps its centre: centres land on the gold ones
    assert np.allclose(centres[0], [0, 0, 0], atol=1e-9) and np.allclose(centres[1], [2, 1, 0], atol=1e-9)
    out = tmp_path / "report.csv"
    res = boxes.match_bboxes(gold, aligned, max_dist=0.3, csv_output_file=out)
    assert sorted((g, a) for g, a, _ in res["matches"]) == [(0, 0), (2, 1)]
    assert res["missing"] == [1] and res["unexpected"] == [2] and res["damaged"] == [0]
    rows = out.read_text().strip().splitlines()
    assert rows[0].startswith("gold_index") and len(rows) == 4 and ",missing," in rows[3]
    assert boxes.match_bboxes({}, aligned)["matches"] == [] and boxes.match_bboxes(gold, {})["missing"] == [0, 1, 2]
    # list-of-stages form is accepted too
    staged = boxes.transform_bboxes(comp, [(R, np.zeros(3)), transforms.rt_to_matrix(np.eye(3), [0.5, -0.2, 0.1])])
    assert np.allclose(staged[9][0][0], aligned[9][0][0])


def test_coarse_candidates_contain_the_true_rotation():
    gold, comp, _, _ = syn.make_pair(40, 5000, resample=False, angle_deg=140.0, t0=np.array([6.0, -3.0, 1.0]))
    cg, ag, vg = coarse.principal_frame(gold)
    cc, ac, vc = coarse.principal_frame(comp)
    assert np.allclose(vg, vc, rtol=1e-3)                    # rigid motion preserves the spectrum
    R_true = syn.rotation_about_axis(syn.GT_AXIS, np.deg2rad(140.0))
    cands = coarse.candidate_rotations(ag, ac)
    assert all(abs(np.linalg.det(R) - 1) < 1e-9 for R in cands)
    assert min(np.linalg.norm(R - R_true) for R in cands) < 1e-3


def test_detector_row_schema_helpers():
    from mapalign import lifting
    row = lifting.parse_yolo_label_line("3 0.5 0.5 0.25 0.5 0.87", 640, 480)
    assert row == [240.0, 120.0, 400.0, 360.0, 0.87, 3]              # detector.py:155
    assert lifting.insert_damage_class(row, 2) == [240.0, 120.0, 400.0, 360.0, 2, 0.87, 3]   # detector.py:126-127
    assert lifting.scale_bounding_box([100, 50, 300, 200], (1000, 500), (100, 50)) == [10.0, 5.0, 30.0, 20.0]
    K = lifting.rescaled_intrinsics({"image_width": 1920, "image_height": 1440, "fx": 1500.0, "fy": 1500.0,
                                     "cx": 960.0, "cy": 720.0}, 192)
    assert np.allclose(K, (150.0, 150.0, 96.0, 72.0))                 # pose_processor.py:133-137


def test_suppress_bboxes_3d_nms():
    # the same sign seen in three frames + a different sign nearby + the same label far away
    data = {0: [_box([0, 0, 0], conf=0.7, label="exit")],
            1: [_box([0.05, 0.02, 0], conf=0.9, label="exit"), _box([0.1, 0, 0], conf=0.8, label="fire")],
            2: [_box([0.02, -0.03, 0.01], conf=0.6, label="exit"), _box([5, 0, 0], conf=0.5, label="exit")]}
    kept = boxes.suppress_bboxes(data, dist_thresh=0.3)
    assert sorted(kept) == [1, 2]
    assert [r[5] for r in kept[1]] == [0.9, 0.8] and [r[5] for r in kept[2]] == [0.5]
    assert boxes.suppress_bboxes({}) == {}


def test_bbox_comparison_class_has_the_reference_call_shape(tmp_path):
    """task_def.py:190-197: BBoxComparison(gold, aligned, goldstd_mesh, visualise=, csv_output_file=).match_bboxes()"""
    import mapalign
    sq = lambda c: [np.array(c) + d for d in ([0, 0, 0], [0.2, 0, 0], [0.2, 0.2, 0], [0, 0.2, 0])]   # noqa: E731
    gold = {0: [sq([0, 0, 0]) + [0, 0.9, 3], sq([5, 0, 0]) + [0, 0.8, 4]], 2: [sq([9, 1, 0]) + [0, 0.7, 5]]}
    aligned = {1: [sq([0.05, 0, 0]) + [1, 0.9, 3]], 4: [sq([9.1, 1, 0]) + [0, 0.6, 5]]}
    csv_path = tmp_path / "out.csv"
    cmp_ = mapalign.BBoxComparison(gold, aligned, np.zeros((4, 3)), visualise=False, csv_output_file=str(csv_path))
    res = cmp_.match_bboxes()
    assert res is cmp_.result
    assert sorted((g, a) for g, a, _ in res["matches"]) == [(0, 0), (2, 1)]
    assert res["missing"] == [1] and res["damaged"] == [0] and res["unexpected"] == []
    assert res == boxes.match_bboxes(gold, aligned)
    lines = csv_path.read_text().strip().splitlines()
    assert len(lines) == 4 and "missing" in lines[-1]


def _pose_table(xyz):
    """Array in the reference's pose layout: timestamp tx ty tz qx qy qz qw (database_query.py:22)."""
    k = len(xyz)
    return np.concatenate([np.arange(k)[:, None] * 0.1, xyz, np.tile([0, 0, 0, 1.0], (k, 1))], axis=1)


def test_trajectory_helpers_and_half_turn_disambiguation():
    # a walk along the carriage, 0.4 m off the centre line, recorded at uneven speed
    s = np.sort(np.random.default_rng(0).uniform(0, 18, 80))
    base = np.stack([1 + s, np.full_like(s, 1.9), np.full_like(s, 1.5)], axis=1)
    assert np.allclose(coarse.trajectory_xyz(_pose_table(base)), base)
    assert np.allclose(coarse.trajectory_xyz({"tx": base[:, 0], "ty": base[:, 1], "tz": base[:, 2]}), base)
    assert coarse.trajectory_xyz(None) is None and coarse.trajectory_xyz(base[:1]) is None
    poly = coarse.resample_polyline(base, 10)
    assert np.allclose(np.diff(poly[:, 0]), np.diff(poly[:, 0])[0]) and np.allclose(poly[[0, -1]], base[[0, -1]])
    # the comparison scan's frame: turned by 90 deg about z and shifted
    Rz = syn.rotation_about_axis([0, 0, 1], np.pi / 2)
    T_true = transforms.rt_to_matrix(Rz, [3.0, -2.0, 0.5])                 # comparison -> gold
    comp = transforms.apply(np.linalg.inv(T_true), base)[::-1]               # walked from the other end
    assert coarse.trajectory_residual(T_true, base, comp) < 1e-9
    # the half-turn twin of T_true (about the vertical axis through the shell's centre)
    c = np.array([10.0, 1.5, 1.25])
    half = transforms.rotation_about_centre(syn.rotation_about_axis([0, 0, 1], np.pi), c)
    assert coarse.trajectory_residual(half @ T_true, base, comp) > 0.5      # camera lands on the wrong side of the centre line


def test_detector_adaptor_rows_and_batched_lift_bookkeeping():
    """Row schema of detector.py:155 + :126-127 out of the torch adaptor (CPU, tiny input)."""
    import torch
    from mapalign import detect

    torch.manual_seed(0)
    model = detect.build_yolov10_like(num_classes=5, width=(8, 8, 16, 16, 32), max_det=20)

    class TwoClass(torch.nn.Module):
        def forward(self, x):
            return torch.stack([x.mean(dim=(1, 2, 3)), 1.0 - x.mean(dim=(1, 2, 3))], dim=1)

    det = detect.Detector(model, TwoClass(), batch_size=2, conf_thresh=0.0, device="cpu", half=False, crop_size=32)
    imgs = (np.random.default_rng(0).random((3, 64, 96, 3)) * 255).astype(np.uint8)
    raw, _ = det.infer(imgs)
    assert raw.shape == (3, 20, 6)
    preds = det(imgs)
    assert sorted(preds) == [0, 1, 2] and all(len(preds[k]) == 20 for k in preds)
    row = preds[1][0]
    assert len(row) == 7 and isinstance(row[6], int) and row[4] in (0, 1)    # x1 y1 x2 y2 damage conf label
    assert np.allclose(row[:4], raw[1, 0, :4].numpy()) and abs(row[5] - float(raw[1, 0, 4])) < 1e-6
    # a stricter threshold drops rows, never reorders them
    det.conf_thresh = float(np.quantile(raw[:, :, 4].numpy(), 0.7))
    fewer = det(imgs)
    assert 0 < sum(len(v) for v in fewer.values()) < 60

    # the crops the damage classifier sees are image[y1:y2, x1:x2] resized bilinearly (classifier.py:69-88),
    # cut and resized a whole classifier batch at a time
    class Recorder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.seen = []

        def forward(self, x):
            self.seen.append(x.clone())
            return torch.zeros(x.shape[0], 2)

    rec = Recorder()
    det2 = detect.Detector(model, rec, batch_size=4, conf_thresh=0.0, device="cpu", half=False, crop_size=16)
    x = det2._prepare(imgs)
    boxes6 = torch.tensor([[[3.7, 2.2, 40.9, 30.1, 0.9, 1], [-5.0, 10.0, 8.0, 11.2, 0.8, 0], [90.0, 60.0, 300.0, 300.0, 0.7, 2],
                            [20.0, 20.0, 20.5, 20.5, 0.6, 1], [0.0, 0.0, 96.0, 64.0, 0.5, 0]]] * 3)
    keep = torch.ones(3, 5, dtype=torch.bool)
    det2._damage(x, boxes6, keep)
    got = torch.cat(rec.seen)
    assert got.shape == (15, 3, 16, 16) and len(rec.seen) == 4                      # 15 crops in batches of 4
    H, W = x.shape[2], x.shape[3]
    k = 0
    for b in range(3):
        for x1, y1, x2, y2 in boxes6[b, :, :4].tolist():
            xa, ya = int(min(max(x1, 0), W - 1)), int(min(max(y1, 0), H - 1))
            xb, yb = int(min(max(x2, xa + 1), W)), int(min(max(y2, ya + 1), H))
            want = torch.nn.functional.interpolate(x[b:b + 1, :, ya:yb, xa:xb], size=(16, 16), mode="bilinear", align_corners=False)
            assert torch.allclose(got[k], want[0], atol=1e-5), (b, x1, y1, x2, y2)
            k += 1

    class FakeEngine:   # records what lift_batch hands to the ONE device call
        def lift_boxes_batch(self, b, bf, d, K, T, scale, z_rule=0):
            self.args = (b, bf, d.shape, K, T.shape, scale)
            return np.zeros((len(b), 4, 3)), np.ones((len(b), 4), bool)

    eng = FakeEngine()
    intr = {"fx": 100.0, "fy": 100.0, "cx": 48.0, "cy": 32.0, "image_width": 96, "image_height": 64}
    poses = np.tile([0, 0, 0, 0, 0, 0, 1.0], (3, 1))
    out = detect.lift_batch(eng, fewer, np.ones((3, 32, 48), np.float32), intr, poses)
    b, bf, dshape, K, Tshape, _ = eng.args
    assert len(b) == sum(len(v) for v in fewer.values()) and dshape == (3, 32, 48) and Tshape == (3, 4, 4)
    assert K == (50.0, 50.0, 24.0, 16.0)                                      # intrinsics / (image_width / depth_width)
    assert list(bf) == sorted(bf) and set(out) == {f for f in fewer if fewer[f]}
    f0 = min(out)
    assert np.allclose(b[0], np.array(fewer[f0][0][:4]) * 0.5)                # scale_bounding_box to depth resolution
    assert out[f0][0][4:] == fewer[f0][0][-3:]


def test_align_many_scheduling_with_stand_in_contexts(monkeypatch):
    """Host logic of mapalign.align_many without a GPU: results come back in input order whatever the
    worker interleaving, a gold array shared by consecutive pairs is indexed once per worker, and the
    first failing pair's exception is raised after the workers have stopped taking pairs."""
    import threading
    import time as _time
    from types import SimpleNamespace

    from mapalign import api

    class FakeEngine:
        def __init__(self):
            self.gold_builds = 0
            self.gold = self.src = None
            self.gold_token = None

        def set_gold(self, g, cell):
            self.gold_builds += 1
            self.gold = g
            self.gold_token = None

        def set_pair(self, g, c, weights, cell):
            self.set_gold(g, cell)
            self.set_source(c)

        def set_source(self, c):
            if np.isnan(c).any():
                raise ValueError("bad scan")
            self.src = c

        def icp(self, max_iter, tol, max_corr_dist):
            _time.sleep(0.002 * float(self.src[0, 0] % 3))        # uneven durations: workers finish out of order
            T = np.eye(4)
            T[0, 3] = float(self.gold[0, 0]) + 10.0 * float(self.src[0, 0])
            return SimpleNamespace(T=T)

    fakes = {}
    monkeypatch.setattr(api, "cached_engine", lambda device=0, slot=0: fakes.setdefault(slot, (FakeEngine(), threading.Lock())))
    gold = np.full((4, 3), 7.0)
    pairs = [(gold, np.full((2, 3), float(k))) for k in range(17)]
    for workers in (1, 2, 3):
        fakes.clear()
        Ts = api.align_many(pairs, workers=workers)
        assert [T[0, 3] for T in Ts] == [7.0 + 10.0 * k for k in range(17)]
        assert len(fakes) == workers and all(e.gold_builds == 1 for e, _ in fakes.values())   # the shared gold: once per worker
    fakes.clear()
    distinct = [(gold.copy(), s) for _, s in pairs]
    api.align_many(distinct, workers=2)
    assert sum(e.gold_builds for e, _ in fakes.values()) == 17
    bad = list(pairs)
    bad[5] = (gold, np.full((2, 3), np.nan))
    with pytest.raises(ValueError, match="bad scan"):
        api.align_many(bad, workers=3)
    assert api.align_many([], workers=3) == []
    # another user of the same cached context installs a different gold map BETWEEN two jobs of a worker
    # (align() shares slot 0): the worker must notice and index its own map again
    fakes.clear()
    api.align_many(pairs[:2], workers=1)
    eng0 = fakes[0][0]
    assert eng0.gold_builds == 1 and eng0.gold_token is gold
    eng0.set_gold(np.full((4, 3), 99.0), 0.0)                   # what a concurrent align() on slot 0 does
    Ts = api.align_many(pairs[:3], workers=1)
    assert [T[0, 3] for T in Ts] == [7.0 + 10.0 * k for k in range(3)] and eng0.gold_builds == 3


def _pil_bilinear_weights(n_in, n_out):
    """PIL's ImagingResample coefficients for the triangle (bilinear) filter, which widens its support
    by the reduction factor (anti-aliasing): row i of the [n_out, n_in] matrix."""
    scale = n_in / n_out
    fs = max(scale, 1.0)
    Wm = np.zeros((n_out, n_in))
    for i in range(n_out):
        centre = (i + 0.5) * scale
        lo, hi = max(int(centre - fs + 0.5), 0), min(int(centre + fs + 0.5), n_in)
        k = np.arange(lo, hi)
        w = np.clip(1.0 - np.abs((k + 0.5 - centre) / fs), 0.0, None)
        Wm[i, lo:hi] = w / w.sum()
    return Wm


def _full_frame_crop_numpy(img, box, S):
    """NumPy restatement of the reference's two-step resample for ONE box (detector.py:197-231, then the BEiT
    processor's resize): img [3,H,W] in 0..1 -> [3,S,S] normalised to mean = std = 0.5."""
    _, H, W = img.shape
    x1, y1, x2, y2 = (float(v) for v in box)
    sx = x1 + np.arange(W) / (W - 1) * (x2 - x1)
    sy = y1 + np.arange(H) / (H - 1) * (y2 - y1)
    pad = np.zeros((3, H + 2, W + 2))
    pad[:, 1:-1, 1:-1] = img                                    # BORDER_CONSTANT 0: one ring of zeros is all bilinear can reach

    def taps(s, n):
        f = np.floor(s)
        a = s - f
        i0 = np.clip(f.astype(int) + 1, 0, n + 1)               # index into the padded axis
        i1 = np.clip(f.astype(int) + 2, 0, n + 1)
        inside0 = (f >= -1) & (f <= n - 1)
        inside1 = (f + 1 >= -1) & (f + 1 <= n)
        return i0, i1, np.where(inside0, 1.0 - a, 0.0), np.where(inside1, a, 0.0)

    xi0, xi1, xw0, xw1 = taps(sx, W)
    yi0, yi1, yw0, yw1 = taps(sy, H)
    rows = pad[:, yi0, :] * yw0[None, :, None] + pad[:, yi1, :] * yw1[None, :, None]         # [3,H,W+2]
    warped = rows[:, :, xi0] * xw0[None, None, :] + rows[:, :, xi1] * xw1[None, None, :]     # [3,H,W]
    Wy, Wx = _pil_bilinear_weights(H, S), _pil_bilinear_weights(W, S)
    small = np.einsum("ih,chw,jw->cij", Wy, warped, Wx)
    return (small - 0.5) / 0.5


def test_full_frame_crop_matches_a_numpy_restatement_of_the_references_two_step_resample():
    """crop="full_frame" (detector.py:197-231 + the image processor of classifier.py:76): the box is warped onto
    the WHOLE frame (bilinear, zeros outside) and that frame-sized image is reduced to 224 (PIL-style bilinear with
    its anti-aliasing support) and normalised — against an independent NumPy restatement, on the CPU."""
    import torch
    from mapalign import detect

    rng = np.random.default_rng(4)
    H, W, S = 48, 64, 16
    imgs = rng.random((3, 3, H, W)).astype(np.float32)
    boxes = np.array([[10.3, 5.8, 40.9, 30.2],         # an ordinary box
                      [-6.0, -4.0, 20.5, 18.0],        # reaches outside the frame: zeros come in
                      [50.0, 40.0, 70.0, 55.0]],       # over the far corner
                     dtype=np.float32)

    class Identity(torch.nn.Module):
        def forward(self, x):
            return x

    det = detect.Detector(Identity(), None, device="cpu", half=False, crop_size=S, crop="full_frame")
    got = det.full_frame_crops(torch.from_numpy(imgs), torch.from_numpy(boxes)).numpy()
    for k in range(3):
        want = _full_frame_crop_numpy(imgs[k].astype(np.float64), boxes[k], S)
        assert np.abs(got[k] - want).max() < 2e-5, (k, np.abs(got[k] - want).max())
    # the reference's uint8 wrap-around, on request: v -> (256 - v) mod 256
    detw = detect.Detector(Identity(), None, device="cpu", half=False, crop_size=S, crop="full_frame", pixel_wrap=True)
    q = (np.round(rng.random((1, 3, H, W)) * 255) / 255).astype(np.float32)
    full = np.array([[0.0, 0.0, W - 1.0, H - 1.0]], dtype=np.float32)            # the identity warp
    gw = detw.full_frame_crops(torch.from_numpy(q), torch.from_numpy(full)).numpy()
    wrapped = np.mod(256.0 - np.round(q[0].astype(np.float64) * 255), 256.0) / 255.0
    Wy, Wx = _pil_bilinear_weights(H, S), _pil_bilinear_weights(W, S)
    assert np.abs(gw[0] - (np.einsum("ih,chw,jw->cij", Wy, wrapped, Wx) - 0.5) / 0.5).max() < 2e-5
    with pytest.raises(ValueError):
        detect.Detector(Identity(), None, device="cpu", crop="centre")


def test_damage_labels_through_the_full_frame_crop_with_a_toy_classifier():
    """Detector.forward with crop="full_frame": rows keep the reference's schema, and the class comes from the
    classifier's view of the warped box (a toy classifier that thresholds the mean of what it is shown)."""
    import torch
    from mapalign import detect

    class OneBox(torch.nn.Module):                    # one fixed box per image, confidence 0.9, label 3
        def forward(self, x):
            b = x.shape[0]
            row = torch.tensor([8.0, 8.0, 23.0, 23.0, 0.9, 3.0])
            return row.repeat(b, 1, 1)

    class Bright(torch.nn.Module):
        def forward(self, crops):
            m = crops.mean(dim=(1, 2, 3))
            return torch.stack([-m, m], 1)            # class 1 if the (normalised) crop is bright

    imgs = np.zeros((2, 32, 32, 3), dtype=np.uint8)
    imgs[1, 8:24, 8:24] = 255                         # image 1: the box is white, everything else black
    det = detect.Detector(OneBox(), Bright(), device="cpu", half=False, crop_size=8, crop="full_frame", conf_thresh=0.5)
    preds = det(imgs)
    assert [len(preds[0][0]), preds[0][0][4], preds[1][0][4]] == [7, 0, 1]
    assert preds[1][0][:4] == [8.0, 8.0, 23.0, 23.0] and preds[1][0][5:] == [pytest.approx(0.9), 3]


def test_host_side_principal_frame_ignores_stray_points():
    """coarse.principal_frame (the host-side coarse stage, N1): 20 stray points 200 scene-widths away among 100 000 must
    not move the frame — it is taken over the cloud's bulk, by the rule of the engine's grid box."""
    from mapalign import coarse, synthetic as syn
    rng = np.random.default_rng(22)
    gold, _, _, _ = syn.make_pair(18, 100_000, 10)
    lo, hi = gold.min(0), gold.max(0)
    st = (0.5 * (lo + hi) + rng.uniform(-0.5, 0.5, size=(20, 3)) * float((hi - lo).max()) * 200.0).astype(np.float32)
    c0, a0, w0 = coarse.principal_frame(gold)
    c1, a1, w1 = coarse.principal_frame(np.concatenate([gold, st]))
    assert np.linalg.norm(c1 - c0) < 0.05 and np.allclose(w1, w0, rtol=0.02)
    assert np.abs(np.abs(np.sum(a0 * a1, axis=0)) - 1.0).max() < 1e-3          # the same axes (up to sign)
