"""Steps immediately downstream of the alignment: move the comparison
scan's 3-D sign boxes onto the gold map and match them against the gold boxes.

Box container = the reference's: ``dict[frame_index] -> list of rows``, a row being
``[c0, c1, c2, c3, damage_cls, conf, label]`` with four xyz corner arrays
(``src/mapper/pose_processor.py:208``; corners are sliced ``[:4]`` at ``mapping.py:172``).
``src/map_alignment/comparison.py`` itself is absent from the snapshot; the matching below
is the bipartite scheme the README describes (``README.md:67``, ``readme_img/bp_match_algo.png``):
minimum-cost assignment on box-centre distance, unmatched gold boxes = missing signs.
"""
from __future__ import annotations

import csv

import numpy as np

from . import transforms


def transform_bboxes(bboxes, T):
    """Apply a rigid transform (4x4, (R,centre) tuple or reference-style list) to every corner."""
    T = transforms.compose(T) if isinstance(T, list) else transforms.to_matrix(T)
    out = {}
    for frame, rows in bboxes.items():
        new_rows = []
        for row in rows:
            corners = [transforms.apply(T, np.asarray(c, dtype=np.float64)) for c in row[:4]]
            new_rows.append(corners + list(row[4:]))
        out[frame] = new_rows
    return out


def flatten(bboxes):
    """-> (centres [K,3], rows list) over all frames, frame order preserved."""
    rows = [row for frame in bboxes for row in bboxes[frame]]
    if not rows:
        return np.zeros((0, 3)), rows
    centres = np.array([np.mean([np.asarray(c, dtype=np.float64) for c in row[:4]], axis=0) for row in rows])
    return centres, rows


def suppress_bboxes(global_bboxes_data, dist_thresh=0.3):
    """3-D non-maximum suppression of the per-frame global boxes (the step
    ``BoundingBoxProcessor(global_bboxes_data, pose_df).suppress_bboxes()`` performs at
    ``task_def.py:145-149``; its body is absent from the snapshot).  Greedy by confidence: a box
    is dropped when a kept box with the same label has its centre within ``dist_thresh`` metres.
    Returns the same ``dict[frame] -> rows`` container with the survivors."""
    centres, rows = flatten(global_bboxes_data)
    frames = [frame for frame in global_bboxes_data for _ in global_bboxes_data[frame]]
    conf = np.array([float(r[5]) if len(r) > 5 else 1.0 for r in rows]) if rows else np.zeros(0)
    keep = []
    for i in np.argsort(-conf, kind="stable"):
        lab = rows[i][6] if len(rows[i]) > 6 else None
        if all((rows[j][6] if len(rows[j]) > 6 else None) != lab or
               np.linalg.norm(centres[i] - centres[j]) > dist_thresh for j in keep):
            keep.append(int(i))
    out = {}
    for i in sorted(keep):
        out.setdefault(frames[i], []).append(rows[i])
    return out


def match_bboxes(gold_bboxes, aligned_bboxes, max_dist=0.5, csv_output_file=None):
    """Minimum-cost bipartite matching on centre distance, gated at ``max_dist`` metres.

    Returns dict(matches=[(gold_idx, aligned_idx, dist)], missing=[gold_idx],
    unexpected=[aligned_idx], damaged=[gold_idx whose match carries a non-zero damage class]).
    """
    from scipy.optimize import linear_sum_assignment

    gc, grows = flatten(gold_bboxes)
    ac, arows = flatten(aligned_bboxes)
    matches = []
    if len(gc) and len(ac):
        cost = np.linalg.norm(gc[:, None, :] - ac[None, :, :], axis=2)
        big = 1e6
        gi, ai = linear_sum_assignment(np.where(cost <= max_dist, cost, big))
        matches = [(int(g), int(a), float(cost[g, a])) for g, a in zip(gi, ai) if cost[g, a] <= max_dist]
    mg = {g for g, _, _ in matches}
    ma = {a for _, a, _ in matches}
    missing = [g for g in range(len(gc)) if g not in mg]
    unexpected = [a for a in range(len(ac)) if a not in ma]

    def _damaged(row):
        try:
            return float(row[4]) != 0.0
        except (TypeError, ValueError):
            return str(row[4]).lower() not in ("", "0", "none", "undamaged", "ok")

    damaged = [g for g, a, _ in matches if len(arows[a]) > 4 and _damaged(arows[a])]
    if csv_output_file:
        with open(csv_output_file, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["gold_index", "label", "status", "aligned_index", "distance_m", "x", "y", "z"])
            lab = lambda row: row[6] if len(row) > 6 else ""   # noqa: E731
            for g, a, d in matches:
                w.writerow([g, lab(grows[g]), "damaged" if g in damaged else "present", a, f"{d:.4f}", *gc[g]])
            for g in missing:
                w.writerow([g, lab(grows[g]), "missing", "", "", *gc[g]])
    return {"matches": matches, "missing": missing, "unexpected": unexpected, "damaged": damaged}


class BBoxComparison:
    """The reference's call shape for the step the alignment feeds
    (``/root/reference/task_def.py:190-197``)::

        compare_bboxes = BBoxComparison(goldstd_bboxes, aligned_maintenance_bboxes, goldstd_mesh,
                                        visualise=cfg.comparison_vis, csv_output_file=cfg.csv_output)
        compare_bboxes.match_bboxes()

    ``src/map_alignment/comparison.py`` is absent from the snapshot; the matching is
    :func:`match_bboxes` (bipartite on centre distance, ``README.md:67``).  ``goldstd_mesh`` is
    only shown by the reference's viewer (out of scope) and is kept for signature parity, as is
    ``visualise``.  The result dict is returned and kept in ``self.result``.
    """

    def __init__(self, gold_bboxes, aligned_bboxes, goldstd_mesh=None, visualise=False, csv_output_file=None,
                 max_dist=0.5):
        self.gold_bboxes = gold_bboxes or {}
        self.aligned_bboxes = aligned_bboxes or {}
        self.goldstd_mesh = goldstd_mesh
        self.visualise = visualise
        self.csv_output_file = csv_output_file
        self.max_dist = float(max_dist)
        self.result = None

    def match_bboxes(self):
        self.result = match_bboxes(self.gold_bboxes, self.aligned_bboxes, max_dist=self.max_dist,
                                   csv_output_file=self.csv_output_file)
        return self.result
