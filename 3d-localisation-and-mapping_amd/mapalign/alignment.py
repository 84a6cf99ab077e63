"""``Alignment(...).compare(data_folder)`` — the reference's call surface for the path.

Mirrors the call site ``/root/reference/task_def.py:175-187``::

    map_alignment = Alignment(base_pose_df=..., comparison_pose_df=..., base_bboxes=...,
                              comparison_bboxes=..., visualise=cfg.alignment_vis)
    aligned_maintenance_bboxes, _, goldstd_mesh, _ = map_alignment.compare(data_folder)

and leaves behind what the reference's offline visualiser reads from the scan's pickle
(``src/map_alignment/visualise.py:233-238``): ``transformations`` (list of ``(R, centre)``
tuples and 4x4 matrices, applied in order), ``base_map_filepath``,
``comparison_map_filepath``.  The body of the reference class is absent from the snapshot;
this adaptor runs coarse pre-alignment + ICP on the MI355X engine.

The second and fourth elements of the reference's 4-tuple are unused by its caller and
unknown; here they carry the composed 4x4 and an info dict.  ``goldstd_mesh`` is the gold
map's points (the reference returns an Open3D mesh for its viewer, which is out of scope).
"""
from __future__ import annotations

import glob
import os
import pickle

import numpy as np

from . import boxes, coarse, ply, transforms
from ._lib import MA_ERR_DEGENERATE, MapAlignError
from .api import cached_engine

DATA_ROOT = os.path.join("src", "common", "data")     # visualise.py:241
GOLD_FOLDER = "gold_std"                               # visualise.py:224-225


def default_map_resolver(data_folder):
    """(gold .ply, comparison .ply): first ``*.ply`` under src/common/data/<folder>/."""
    def first(folder):
        hits = sorted(glob.glob(os.path.join(DATA_ROOT, folder, "**", "*.ply"), recursive=True))
        if not hits:
            raise FileNotFoundError(f"no .ply under {os.path.join(DATA_ROOT, folder)}")
        return hits[0]
    return first(GOLD_FOLDER), first(data_folder)


class Alignment:
    def __init__(self, base_pose_df=None, comparison_pose_df=None, base_bboxes=None, comparison_bboxes=None,
                 visualise=False, *, base_map_filepath=None, comparison_map_filepath=None,
                 map_resolver=default_map_resolver, pickle_path=None, device=0, max_iter=50, tol=0.0,
                 max_corr_dist=0.10, coarse_stage=True, gate_schedule="auto"):
        self.base_pose_df = base_pose_df
        self.comparison_pose_df = comparison_pose_df
        self.base_bboxes = base_bboxes or {}
        self.comparison_bboxes = comparison_bboxes or {}
        self.visualise = visualise      # rendering is out of scope; kept for signature parity
        self.base_map_filepath = base_map_filepath
        self.comparison_map_filepath = comparison_map_filepath
        self.map_resolver = map_resolver
        self.pickle_path = pickle_path
        self.device = device
        self.max_iter, self.tol, self.max_corr_dist = max_iter, tol, max_corr_dist
        self.coarse_stage = coarse_stage
        # Gates tried before the final one.  The principal-axes start of the coarse stage is good to
        # a few degrees at best on real scans, and a 0.10 m gate alone only recovers starts within
        # about 2 degrees / 15 cm of the answer (tools/basin_probe.py: 3 deg / 0.37 m already fails),
        # whereas 0.8 -> 0.4 -> 0.2 -> 0.1 m recovers 6 deg / 1 m for 5-20 ms more.  "auto": 8x, 4x,
        # 2x max_corr_dist, up to 15 iterations each (stopping when the rmse settles); None: the final
        # gate only; or an explicit list of (gate_m, max_iterations).
        if gate_schedule == "auto":
            gate_schedule = [(8.0 * max_corr_dist, 15), (4.0 * max_corr_dist, 15), (2.0 * max_corr_dist, 15)] if coarse_stage else None
        self.gate_schedule = list(gate_schedule) if gate_schedule else []
        self.transformations = []

    def align_clouds(self, gold_cloud, comparison_cloud):
        """Staged alignment of two point arrays -> (transformations list, composed 4x4, info)."""
        eng, lock = cached_engine(self.device)     # one context per device for the process: warm buffers
        with lock:
            eng.set_pair(gold_cloud, comparison_cloud)     # both clouds staged at once (the scan beside the gold build)
            stages, T0, info = [], np.eye(4), {}
            if self.coarse_stage:                          # frames and scores from the device-resident clouds
                stages, T0, info = coarse.coarse_align(eng, gold_cloud, comparison_cloud,
                                                       base_pose_df=self.base_pose_df,
                                                       comparison_pose_df=self.comparison_pose_df, resident=True)
            T, gpu_ms, widened = T0, 0.0, []
            for gate, iters in self.gate_schedule:          # wide gates first: pulls a rough start into the final gate's basin
                try:
                    r = eng.icp(int(iters), 1e-5, float(gate), init=T)
                except MapAlignError as e:
                    if e.status != MA_ERR_DEGENERATE:       # only "nothing inside this gate (or collinear)" may be skipped
                        raise
                    continue                                # keep T, try the next gate
                T, gpu_ms = r.T, gpu_ms + r.gpu_ms
                widened.append((float(gate), r.iterations, r.rmse))
            res = eng.icp(self.max_iter, self.tol, self.max_corr_dist, init=T)
        T_icp = res.T @ np.linalg.inv(T0)          # refinement (all gates) applied AFTER the coarse stages
        stages = list(stages) + [T_icp]
        info.update(iterations=res.iterations, rmse=res.rmse, inliers=res.inliers, gpu_ms=gpu_ms + res.gpu_ms,
                    gate_schedule=widened)
        return stages, res.T, info

    def compare(self, data_folder):
        if self.base_map_filepath is None or self.comparison_map_filepath is None:
            self.base_map_filepath, self.comparison_map_filepath = self.map_resolver(data_folder)
        # both files at once (NumPy's file reads release the interpreter lock): NaN / inf removed, as mapping.py:63-68;
        # coordinates in the files' own type (float32 from RTAB-Map: a view of the bytes read, no conversion pass)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=2) as pool:
            fg = pool.submit(ply.read_ply, self.base_map_filepath, dtype=None)
            fc = pool.submit(ply.read_ply, self.comparison_map_filepath, dtype=None)
            gold, comp = fg.result(), fc.result()
        self.transformations, T, info = self.align_clouds(gold, comp)
        aligned = boxes.transform_bboxes(self.comparison_bboxes, T)
        if self.pickle_path:
            state = {}
            if os.path.exists(self.pickle_path):
                with open(self.pickle_path, "rb") as f:
                    state = pickle.load(f)
            state.update(transformations=self.transformations, base_map_filepath=self.base_map_filepath,
                         comparison_map_filepath=self.comparison_map_filepath)
            with open(self.pickle_path, "wb") as f:
                pickle.dump(state, f)
        return aligned, T, gold, info
