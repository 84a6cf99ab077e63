"""Host-side handle on one libmapalign context (= one MI355X + one HIP stream).

Thin by design: argument checking, NumPy <-> pointer plumbing and error translation.
All arithmetic of rows a4-a8 happens in the HIP kernels behind the C ABI.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from ._lib import (MA_ACC_LEN, MA_COMM_ID_BYTES, MA_ERR_DEGENERATE, MA_TRACE_STRIDE, MapAlignError, ma_grid_info,
                   ma_mem_info, ma_params, ma_stats)


def _as_T(init):
    """None | 4x4 | (R, t) -> contiguous 4x4 float64 (or None)."""
    if init is None:
        return None
    if isinstance(init, tuple):
        R, t = init
        T = np.eye(4)
        T[:3, :3] = np.asarray(R, dtype=np.float64)
        T[:3, 3] = np.asarray(t, dtype=np.float64).reshape(3)
        return np.ascontiguousarray(T)
    T = np.ascontiguousarray(init, dtype=np.float64)
    if T.shape != (4, 4):
        raise ValueError("transform must be None, a 4x4 matrix or an (R, t) tuple")
    return T


def _f64p(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _check_cloud(a, name):
    a = np.asarray(a)
    if a.ndim != 2 or a.shape[1] != 3:
        raise ValueError(f"{name} must have shape [N,3], got {a.shape}")
    return a


def _device_array(a, name, device, shape_tail, dtypes=("<f4",)):
    """``a`` if it is an array already in GPU memory (a torch CUDA tensor, or any object with
    ``__cuda_array_interface__``) -> ``(pointer, shape, typestr, keepalive)``; ``None`` for host
    arrays.  The array must be C-contiguous, on ``device``, and complete: for a torch tensor the
    producing stream is synchronised here; other producers synchronise themselves (or share
    the context's stream through :meth:`MapAligner.set_stream`)."""
    keep = a
    if type(a).__module__.startswith("torch") and hasattr(a, "is_cuda"):
        if not a.is_cuda:
            return None
        import torch
        if a.device.index != device:
            raise ValueError(f"{name} lives on {a.device}, the context on device {device}")
        a = a.detach()
        if a.is_floating_point() and a.dtype not in (torch.float32, torch.float64):
            a = a.to(torch.float32)                      # half / bfloat16: widened on the device
        elif a.dtype == torch.float64 and "<f8" not in dtypes:
            a = a.to(torch.float32)
        keep = a = a.contiguous()
        torch.cuda.current_stream(a.device).synchronize()
    iface = getattr(a, "__cuda_array_interface__", None)
    if iface is None:
        return None
    shape, typestr = tuple(iface["shape"]), iface["typestr"]
    if iface.get("strides") is not None:
        raise ValueError(f"{name} must be C-contiguous in device memory")
    if typestr not in dtypes:
        raise ValueError(f"{name} in device memory must have dtype {' or '.join(dtypes)}, got {typestr}")
    if shape_tail is not None and (len(shape) != 1 + len(shape_tail) or shape[1:] != shape_tail):
        raise ValueError(f"{name} must have shape [N{''.join(',%d' % d for d in shape_tail)}], got {shape}")
    return int(iface["data"][0]), shape, typestr, keep


def _cloud(a, name, shift=None):
    """[N,3] array-like -> contiguous float32, after subtracting the host-side shift in fp64."""
    a = _check_cloud(a, name)
    if shift is not None and np.any(shift != 0.0):
        a = np.asarray(a, dtype=np.float64) - shift
    return np.ascontiguousarray(a, dtype=np.float32)


# Georeferenced maps (UTM-like coordinates of 1e5..1e7 m) lose centimetres to decimetres when
# squeezed into fp32.  The device library re-centres internally, but its ABI takes fp32, so the
# host subtracts an integer-metre shift in fp64 FIRST whenever coordinates exceed this bound,
# and maps every result back (T_world = S T_local S^-1).
LARGE_COORDINATE_M = 1000.0


def _conj_shift(T_local, shift, inverse=False):
    """Local-frame 4x4 -> world frame (or back): p_world = p_local + shift."""
    if T_local is None or not np.any(shift != 0.0):
        return T_local
    T = np.array(T_local, dtype=np.float64)
    R, t = T[:3, :3], T[:3, 3]
    T[:3, 3] = (t - shift + R @ shift) if inverse else (t + shift - R @ shift)
    return T


def _acc_shift(acc, shift):
    """17 accumulators from the shifted frame to the world frame (p = p_l + s, q = q_l + s)."""
    if not np.any(shift != 0.0):
        return acc
    a = np.array(acc, dtype=np.float64)
    W, sp, sq = a[0], a[1:4].copy(), a[4:7].copy()
    a[1:4] = sp + W * shift
    a[4:7] = sq + W * shift
    a[7:16] = (a[7:16].reshape(3, 3) + np.outer(sp, shift) + np.outer(shift, sq) + W * np.outer(shift, shift)).ravel()
    return a


@dataclass
class AlignResult:
    T: np.ndarray                  # 4x4, comparison -> gold
    iterations: int
    rmse: float
    inliers: float
    converged: bool
    gpu_ms: float
    trace: np.ndarray = field(default=None, repr=False)   # [iterations, 20] or None

    @property
    def R(self):
        return self.T[:3, :3].copy()

    @property
    def t(self):
        return self.T[:3, 3].copy()


def _depth_image(depth, device):
    """2-D depth image (host array-like, or float32 / uint16 in this GPU's memory) ->
    ``(pointer, dtype code, width, height, keepalive)``; dtype code 0 = float32, 1 = uint16."""
    dev = _device_array(depth, "depth", device, None, ("<f4", "<u2"))
    if dev is not None:
        ptr, shape, typestr, keep = dev
        if len(shape) != 2:
            raise ValueError("depth must be a 2-D image")
        return C.c_void_p(ptr), 1 if typestr == "<u2" else 0, shape[1], shape[0], keep
    d = np.asarray(depth)
    if d.ndim != 2:
        raise ValueError("depth must be a 2-D image")
    if d.dtype == np.uint16:
        dtype = 1
    else:
        d = d.astype(np.float32, copy=False)
        dtype = 0
    d = np.ascontiguousarray(d)
    return d.ctypes.data_as(C.c_void_p), dtype, d.shape[1], d.shape[0], d


class MapAligner:
    """One alignment context.  Not thread-safe; use one per host thread / GPU."""

    def __init__(self, device=0):
        self._lib = _lib.load()
        self._ctx = C.c_void_p()
        st = self._lib.ma_create(int(device), C.byref(self._ctx))
        if st < 0:
            msg = (self._lib.ma_last_error(None) or b"").decode("utf-8", "replace")
            self._ctx = None
            raise MapAlignError(st, msg)
        self.device = int(device)
        self.n_gold = 0
        self._n_source = 0
        self._shift = np.zeros(3)
        self._map_shift = None
        self._depth_keep = None         # a depth frame whose copy into the context is enqueued but not waited for yet
        self._comm_ranks = 0
        self.gold_token = None          # set by callers that want to know whether "their" gold map is still installed (api.align_many)

    @property
    def n_source(self):
        """Points of the comparison cloud (valid pixels of a depth frame; asked of the library if not known yet)."""
        if self._n_source is None:
            n = C.c_uint64(0)
            self._chk(self._lib.ma_get_source(self._ctx, None, C.byref(n)))
            self._n_source = int(n.value)
        return self._n_source

    @n_source.setter
    def n_source(self, v):
        self._n_source = v

    # -- lifetime -------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.ma_destroy(self._ctx)
            self._ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, status):
        return _lib.check(self._lib, self._ctx, status)

    @staticmethod
    def _params(max_iter, tol, max_corr_dist, count_work=False, time_kernels=False, lists=None):
        flags = (_lib.MA_FLAG_COUNT_WORK if count_work else 0) | (_lib.MA_FLAG_TIME_KERNELS if time_kernels else 0)
        if lists is not None and not lists:
            flags |= _lib.MA_FLAG_NO_LISTS
        return ma_params(int(max_iter), flags, float(tol), float(max_corr_dist))

    def set_stream(self, hip_stream, external=True):
        """Run on a caller-owned hipStream_t given as an int handle, e.g.
        ``torch.cuda.current_stream().cuda_stream`` (0 = the legacy default stream, which is
        what torch returns for its default stream).  ``external=False`` returns to the
        context's own stream."""
        self._chk(self._lib.ma_set_stream(self._ctx, C.c_void_p(int(hip_stream) if hip_stream else None),
                                          1 if external else 0))

    def use_own_stream(self):
        self.set_stream(None, external=False)

    def sync(self):
        self._chk(self._lib.ma_sync(self._ctx))
        self._depth_keep = None

    # -- clouds ---------------------------------------------------------------------
    def set_gold(self, gold_cloud, cell=0.0):
        """``gold_cloud``: [M,3] host array-like, or a float32 / float64 array already in this GPU's
        memory (torch CUDA tensor / ``__cuda_array_interface__``) — then nothing crosses PCIe."""
        self.gold_token = None
        dev = _device_array(gold_cloud, "gold_cloud", self.device, (3,), ("<f4", "<f8"))
        if dev is not None:
            ptr, shape, typestr, _keep = dev
            self._shift = np.zeros(3)
            if typestr == "<f8":                         # fp64 on the device: bounds, shift and conversion there too
                shift = np.zeros(3)
                self._chk(self._lib.ma_set_gold_f64(self._ctx, C.cast(C.c_void_p(ptr), C.POINTER(C.c_double)), shape[0],
                                                    float(cell), _f64p(shift)))
                self._shift = shift
            else:
                self._chk(self._lib.ma_set_gold(self._ctx, C.cast(C.c_void_p(ptr), C.POINTER(C.c_float)), shape[0], float(cell)))
            self.n_gold = shape[0]
            self.n_source = 0
            return self.grid_info()
        raw = _check_cloud(gold_cloud, "gold_cloud")
        self._shift = np.zeros(3)
        if raw.dtype == np.float64 and raw.shape[0]:
            # fp64 clouds (Open3D's and NumPy's default): bounds, the large-coordinate shift and the
            # conversion to fp32 happen on the device; the host does not walk the array
            g = np.ascontiguousarray(raw)
            shift = np.zeros(3)
            self._chk(self._lib.ma_set_gold_f64(self._ctx, _f64p(g), g.shape[0], float(cell), _f64p(shift)))
            self._shift = shift
            self.n_gold = g.shape[0]
            self.n_source = 0
            return self.grid_info()
        if raw.shape[0] and raw.dtype != np.float32:
            # whole-array extrema first (2 ms per million points; per-axis reductions of an [N,3]
            # array cost 40 ms): the per-axis box is only needed for georeferenced coordinates
            big = max(abs(float(raw.max())), abs(float(raw.min())))
            if np.isfinite(big) and big > LARGE_COORDINATE_M:
                cols = np.ascontiguousarray(raw.T)
                lo, hi = cols.min(axis=1).astype(np.float64), cols.max(axis=1).astype(np.float64)
                self._shift = np.round(0.5 * (lo + hi))
        g = _cloud(raw, "gold_cloud", self._shift)
        self._chk(self._lib.ma_set_gold(self._ctx, g.ctypes.data_as(C.POINTER(C.c_float)), g.shape[0], float(cell)))
        self.n_gold = g.shape[0]
        self.n_source = 0
        return self.grid_info()

    def grid_info(self):
        gi = ma_grid_info()
        self._chk(self._lib.ma_grid_info_get(self._ctx, C.byref(gi)))
        return {"cell": float(gi.cell), "dims": (gi.nx, gi.ny, gi.nz), "n_cells": int(gi.n_cells),
                "n_occupied": int(gi.n_occupied), "origin": np.array(list(gi.origin)) + self._shift,
                "build_ms": float(gi.build_ms)}

    def grid_box(self):
        """The box the gold map's grid covers (world frame) and whether it is the box of the map's bulk only — stray
        points far from the scene are then kept in the grid's boundary cells (``ma_grid_box_get``)."""
        lo, hi, bulk = (C.c_double * 3)(), (C.c_double * 3)(), C.c_int(0)
        self._chk(self._lib.ma_grid_box_get(self._ctx, lo, hi, C.byref(bulk)))
        return {"lo": np.array(list(lo)) + self._shift, "hi": np.array(list(hi)) + self._shift, "bulk_only": bool(bulk.value)}

    def set_source(self, comparison_cloud, weights=None):
        """``comparison_cloud`` / ``weights``: host array-likes or arrays in this GPU's memory (as in
        :meth:`set_gold`).  fp64 clouds, host or device, are shifted and converted on the device;
        an fp32 DEVICE cloud cannot follow a gold map that needed a large-coordinate shift."""
        fp = C.POINTER(C.c_float)
        dev = _device_array(comparison_cloud, "comparison_cloud", self.device, (3,), ("<f4", "<f8"))
        f64 = False
        if dev is not None:
            f64 = dev[2] == "<f8"
            if not f64 and np.any(self._shift != 0.0):
                raise ValueError("an fp32 device-resident comparison cloud cannot be combined with a gold map in a shifted "
                                 "frame (coordinates above %g m given in fp64): pass it as fp64" % LARGE_COORDINATE_M)
            xyz_p = C.cast(C.c_void_p(dev[0]), C.POINTER(C.c_double) if f64 else fp)
            n, keep = dev[1][0], dev[3]
        else:
            raw = _check_cloud(comparison_cloud, "comparison_cloud")
            f64 = raw.dtype == np.float64 and raw.shape[0] > 0
            if f64:
                s = keep = np.ascontiguousarray(raw)
                xyz_p, n = _f64p(s), s.shape[0]
            else:
                s = keep = _cloud(raw, "comparison_cloud", self._shift)
                xyz_p, n = s.ctypes.data_as(fp), s.shape[0]
        w_p = None
        if weights is not None:
            wdev = _device_array(weights, "weights", self.device, ())
            if wdev is not None:
                w_p, nw, keep_w = C.cast(C.c_void_p(wdev[0]), fp), wdev[1][0], wdev[3]
            else:
                w = keep_w = np.ascontiguousarray(weights, dtype=np.float32).reshape(-1)
                w_p, nw = w.ctypes.data_as(fp), w.shape[0]
            if nw != n:
                raise ValueError("weights must have one entry per comparison point")
        if f64:
            self._chk(self._lib.ma_set_source_f64(self._ctx, xyz_p, w_p, n, _f64p(np.ascontiguousarray(self._shift))))
        else:
            self._chk(self._lib.ma_set_source(self._ctx, xyz_p, w_p, n))
        self.n_source = n

    def set_pair(self, gold_cloud, comparison_cloud, weights=None, cell=0.0):
        """``set_gold`` + ``set_source`` of one pair.  Two arrays of the same dtype — float32, or float64 as Open3D /
        NumPy hand them over — that are both host arrays or both already in this GPU's memory go through
        ``ma_set_pair`` / ``ma_set_pair_f64``, which stage the comparison cloud (copy, bounds, sort) on a second stream
        while the gold map is being indexed (same state afterwards, bitwise); anything else (mixed dtypes or places)
        takes the two calls."""
        fp = C.POINTER(C.c_float)

        def view(a):
            """(pointer, n, typestr, keepalive) of an [N,3] float32 / float64 array, host or device; None if it is neither."""
            dev = _device_array(a, "cloud", self.device, (3,), ("<f4", "<f8"))
            if dev is not None:
                return dev[0], dev[1][0], dev[2], dev[3], True
            if isinstance(a, np.ndarray) and a.dtype in (np.float32, np.float64) and a.ndim == 2 and a.shape[1:] == (3,) and a.shape[0] > 0:
                h = np.ascontiguousarray(a)
                return h.ctypes.data, h.shape[0], "<f4" if h.dtype == np.float32 else "<f8", h, False
            return None

        w_ok = weights is None or (isinstance(weights, np.ndarray) and weights.ndim == 1)
        vg, vc = view(gold_cloud), view(comparison_cloud)
        if vg is None or vc is None or vg[2] != vc[2] or vg[4] != vc[4] or not w_ok:
            grid = self.set_gold(gold_cloud, cell)
            self.set_source(comparison_cloud, weights)
            return grid
        w_p = None
        if weights is not None:
            w = np.ascontiguousarray(weights, dtype=np.float32)
            if w.shape[0] != vc[1]:
                raise ValueError("weights must have one entry per comparison point")
            w_p = w.ctypes.data_as(fp)
        self._shift = np.zeros(3)
        self.n_gold = self.n_source = 0
        self.gold_token = None
        if vg[2] == "<f4":
            self._chk(self._lib.ma_set_pair(self._ctx, C.cast(C.c_void_p(vg[0]), fp), vg[1], float(cell),
                                            C.cast(C.c_void_p(vc[0]), fp), w_p, vc[1]))
        else:
            dp = C.POINTER(C.c_double)
            shift = np.zeros(3)
            self._chk(self._lib.ma_set_pair_f64(self._ctx, C.cast(C.c_void_p(vg[0]), dp), vg[1], float(cell), _f64p(shift),
                                                C.cast(C.c_void_p(vc[0]), dp), w_p, vc[1]))
            self._shift = shift
        self.n_gold, self.n_source = vg[1], vc[1]
        return self.grid_info()

    def set_source_from_depth(self, depth, K, T_cam_to_world, depth_scale=1000.0, count=True):
        """Fused depth -> world unprojection into the source buffer (row a8).  Returns #valid pixels — or, with
        ``count=False``, nothing: the call then only enqueues the upload and the kernel (no wait for the device; an
        ``icp()`` may follow at once) and ``n_source`` is settled when first asked for."""
        d_p, dtype, width, height, _keep = _depth_image(depth, self.device)
        Kf = np.ascontiguousarray(K, dtype=np.float32).reshape(4)
        T = np.array(T_cam_to_world, dtype=np.float64).reshape(4, 4)
        T[:3, 3] -= self._shift
        T = np.ascontiguousarray(T)
        nv = C.c_uint64(0)
        self._chk(self._lib.ma_set_source_from_depth(
            self._ctx, d_p, dtype, width, height,
            Kf.ctypes.data_as(C.POINTER(C.c_float)), _f64p(T), float(depth_scale), C.byref(nv) if count else None))
        self._n_source = int(nv.value) if count else None
        # count=False only enqueued the copy of the frame (include/mapalign.h: the buffer must outlive it): what was made
        # here for the call — an astype / contiguous copy, a tensor's owner — stays referenced until the next wait
        self._depth_keep = None if count else (_keep, depth)
        return self._n_source

    # -- map builder (N4): depth frames + poses -> one voxel-deduplicated cloud on the device
    def map_begin(self, voxel, capacity_points):
        self._chk(self._lib.ma_map_begin(self._ctx, float(voxel), int(capacity_points)))
        self._map_shift = None

    def map_add_depth(self, depth, K, T_cam_to_world, depth_scale=1000.0):
        d_p, dtype, width, height, _keep = _depth_image(depth, self.device)
        Kf = np.ascontiguousarray(K, dtype=np.float32).reshape(4)
        T = np.array(T_cam_to_world, dtype=np.float64).reshape(4, 4)
        if self._map_shift is None:      # georeferenced poses: integer-metre host shift, as for clouds
            t = T[:3, 3]
            self._map_shift = np.round(t) if np.abs(t).max() > LARGE_COORDINATE_M else np.zeros(3)
        T[:3, 3] -= self._map_shift
        T = np.ascontiguousarray(T)
        nv = C.c_uint64(0)
        self._chk(self._lib.ma_map_add_depth(self._ctx, d_p, dtype, width, height,
                                             Kf.ctypes.data_as(C.POINTER(C.c_float)), _f64p(T), float(depth_scale),
                                             C.byref(nv)))
        return int(nv.value)

    def map_add_depth_batch(self, depths, K, poses, depth_scale=1000.0):
        """``map_add_depth`` for a run of frames in one call: ``depths`` [F,H,W] (float32 / uint16; host, or this GPU's
        memory), ``poses`` [F,4,4].  One upload, one kernel launch, one synchronisation; the map holds what F single calls
        would have left, bit for bit.  Returns the F counts of valid pixels."""
        dev = _device_array(depths, "depths", self.device, None, ("<f4", "<u2"))
        if dev is not None:
            ptr, shape, typestr, _keep = dev
            if len(shape) != 3:
                raise ValueError("depths must be [frames, height, width]")
            d_p, dtype = C.c_void_p(ptr), 1 if typestr == "<u2" else 0
        else:
            d = np.asarray(depths)
            if d.ndim != 3:
                raise ValueError("depths must be [frames, height, width]")
            dtype = 1 if d.dtype == np.uint16 else 0
            d = np.ascontiguousarray(d if dtype == 1 else d.astype(np.float32, copy=False))
            d_p, shape, _keep = d.ctypes.data_as(C.c_void_p), d.shape, d
        n_frames, height, width = (int(v) for v in shape)
        T = np.array(poses, dtype=np.float64).reshape(n_frames, 4, 4)
        if self._map_shift is None:      # georeferenced poses: integer-metre host shift, as for clouds
            t = T[0, :3, 3]
            self._map_shift = np.round(t) if np.abs(t).max() > LARGE_COORDINATE_M else np.zeros(3)
        T[:, :3, 3] -= self._map_shift
        T = np.ascontiguousarray(T)
        Kf = np.ascontiguousarray(K, dtype=np.float32).reshape(4)
        counts = np.zeros(n_frames, dtype=np.uint64)
        self._chk(self._lib.ma_map_add_depth_batch(self._ctx, d_p, dtype, width, height, n_frames,
                                                   Kf.ctypes.data_as(C.POINTER(C.c_float)), _f64p(T), float(depth_scale),
                                                   counts.ctypes.data_as(C.POINTER(C.c_uint64))))
        return counts.astype(np.int64)

    def map_finish(self, as_gold=False, cell=0.0, fetch=True):
        """Voxel centroids of everything added since map_begin.  ``as_gold`` installs them as this
        context's gold map on the device.  Returns the [V,3] array (float64 if shifted) or the count."""
        n = C.c_uint64(0)
        shift = self._map_shift if self._map_shift is not None else np.zeros(3)
        if not fetch:
            self._chk(self._lib.ma_map_finish(self._ctx, 1 if as_gold else 0, float(cell), None, C.byref(n)))
            out = int(n.value)
        else:
            self._chk(self._lib.ma_map_finish(self._ctx, 0, float(cell), None, C.byref(n)))    # size query
            arr = np.empty((int(n.value), 3), dtype=np.float32)
            self._chk(self._lib.ma_map_finish(self._ctx, 1 if as_gold else 0, float(cell),
                                              arr.ctypes.data_as(C.POINTER(C.c_float)), C.byref(n)))
            out = arr if not np.any(shift != 0.0) else arr.astype(np.float64) + shift
        if as_gold:
            self.gold_token = None
            self._shift = shift.copy()
            self.n_gold = int(n.value)
            self.n_source = 0
        return out

    def lift_boxes(self, boxes_xyxy, K, T_cam_to_world, depth=None, depth_scale=1000.0, z_rule=0):
        """Row a10: corners (x1,y1) (x2,y1) (x2,y2) (x1,y2) of each box -> global 3-D points.

        ``boxes_xyxy`` [B,4] in depth-image pixels; ``depth=None`` reuses the frame uploaded by the
        last set_source_from_depth / lift_boxes call.  ``z_rule``: 0 = depth at the corner pixel,
        1 = median of the valid depths over the box rectangle (``include/mapalign.h``: MA_Z_CORNER /
        MA_Z_MEDIAN).  Returns (corners f64 [B,4,3], valid bool [B,4])."""
        b = np.ascontiguousarray(boxes_xyxy, dtype=np.float32).reshape(-1, 4)
        Kf = np.ascontiguousarray(K, dtype=np.float32).reshape(4)
        T = np.array(T_cam_to_world, dtype=np.float64).reshape(4, 4)
        T[:3, 3] -= self._shift
        T = np.ascontiguousarray(T)
        out = np.empty((b.shape[0], 4, 3), dtype=np.float64)
        valid = np.zeros((b.shape[0], 4), dtype=np.uint8)
        dptr, dtype, w, h = None, 0, 0, 0
        if depth is not None:
            d = np.asarray(depth)
            if d.ndim != 2:
                raise ValueError("depth must be a 2-D image")
            dtype = 1 if d.dtype == np.uint16 else 0
            d = np.ascontiguousarray(d if dtype == 1 else d.astype(np.float32, copy=False))
            dptr, w, h = d.ctypes.data_as(C.c_void_p), d.shape[1], d.shape[0]
        self._chk(self._lib.ma_lift_boxes(self._ctx, dptr, dtype, w, h, Kf.ctypes.data_as(C.POINTER(C.c_float)),
                                          _f64p(T), float(depth_scale), b.ctypes.data_as(C.POINTER(C.c_float)),
                                          b.shape[0], int(z_rule), _f64p(out), valid.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out + self._shift, valid.astype(bool)

    def lift_boxes_batch(self, boxes_xyxy, box_frame, depth_frames, K, T_cam_to_world, depth_scale=1000.0, z_rule=0):
        """Row a10 for a whole detector batch in ONE device call: ``depth_frames`` [F,H,W]
        (float32 or uint16), ``T_cam_to_world`` [F,4,4], ``boxes_xyxy`` [B,4] in depth-resolution
        pixels, ``box_frame`` [B] = frame index of each box.  Returns (corners f64 [B,4,3], valid bool [B,4])."""
        d = np.asarray(depth_frames)
        if d.ndim != 3:
            raise ValueError("depth_frames must have shape [F,H,W]")
        dtype = 1 if d.dtype == np.uint16 else 0
        d = np.ascontiguousarray(d if dtype == 1 else d.astype(np.float32, copy=False))
        F, h, w = d.shape
        b = np.ascontiguousarray(boxes_xyxy, dtype=np.float32).reshape(-1, 4)
        bf = np.ascontiguousarray(box_frame, dtype=np.uint32).reshape(-1)
        if bf.shape[0] != b.shape[0]:
            raise ValueError("box_frame must have one entry per box")
        T = np.array(T_cam_to_world, dtype=np.float64).reshape(F, 4, 4)
        T[:, :3, 3] -= self._shift
        T = np.ascontiguousarray(T)
        Kf = np.ascontiguousarray(K, dtype=np.float32).reshape(4)
        out = np.empty((b.shape[0], 4, 3), dtype=np.float64)
        valid = np.zeros((b.shape[0], 4), dtype=np.uint8)
        self._chk(self._lib.ma_lift_boxes_batch(
            self._ctx, d.ctypes.data_as(C.c_void_p), dtype, w, h, F, Kf.ctypes.data_as(C.POINTER(C.c_float)), _f64p(T),
            float(depth_scale), b.ctypes.data_as(C.POINTER(C.c_float)), bf.ctypes.data_as(C.POINTER(C.c_uint32)),
            b.shape[0], int(z_rule), _f64p(out), valid.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out + self._shift, valid.astype(bool)

    def last_device_ms(self):
        """Device time of the last set_source_from_depth / map_add_depth (the single-pass unprojection)."""
        v = C.c_double(0.0)
        self._chk(self._lib.ma_last_device_ms(self._ctx, C.byref(v)))
        return float(v.value)

    def mem_info(self):
        """Device allocations of this context since it was created (hipMalloc / hipFree counts, bytes)."""
        mi = ma_mem_info()
        self._chk(self._lib.ma_mem_info_get(self._ctx, C.byref(mi)))
        return {"device_allocs": int(mi.device_allocs), "device_frees": int(mi.device_frees),
                "bytes_live": int(mi.bytes_live), "bytes_peak": int(mi.bytes_peak)}

    def search_geometry(self):
        """(workgroups per search launch, threads = points per workgroup) for the current source cloud."""
        g, b = C.c_uint32(0), C.c_uint32(0)
        self._chk(self._lib.ma_search_geometry(self._ctx, C.byref(g), C.byref(b)))
        return int(g.value), int(b.value)

    def cloud_moments(self, which):
        """(number of valid points, centroid [3], covariance [3,3]) of the gold map (``which=0``) or the source cloud
        (``which=1``) as stored on the device — one reduction over data that is in HBM anyway (coarse stage, N1)."""
        out = np.empty(13, dtype=np.float64)
        self._chk(self._lib.ma_cloud_moments(self._ctx, int(which), _f64p(out)))
        return int(out[0]), out[1:4] + self._shift, out[4:13].reshape(3, 3).copy()

    def truncated_rms(self, T, bound):
        """sqrt(mean(min(d, bound)^2)) over the source cloud's nearest-neighbour distances d under ``T`` — from the
        17 sums of ONE search + accumulate launch gated at ``bound`` (E = sum of matched d^2, W = matches), nothing
        per point comes back to the host.  Unweighted clouds only (the weights would enter E and W)."""
        p = self._params(1, 0.0, float(bound), False)
        Tin = _conj_shift(_as_T(T), self._shift, inverse=True)
        acc = np.empty(MA_ACC_LEN, dtype=np.float64)
        Tout = np.empty((4, 4), dtype=np.float64)
        rc = self._lib.ma_icp_step(self._ctx, C.byref(p), _f64p(Tin), _f64p(acc), _f64p(Tout))
        if rc != MA_ERR_DEGENERATE:        # "nothing (or too little) inside the bound" still fills the sums
            self._chk(rc)
        n = float(self.n_source)
        return float(np.sqrt(max(acc[16] + (n - acc[0]) * float(bound) ** 2, 0.0) / n))

    def get_source(self):
        n = C.c_uint64(0)
        self._chk(self._lib.ma_get_source(self._ctx, None, C.byref(n)))
        out = np.empty((int(n.value), 3), dtype=np.float32)
        self._chk(self._lib.ma_get_source(self._ctx, out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(n)))
        return out if not np.any(self._shift != 0.0) else out.astype(np.float64) + self._shift

    def _stored(self, fn):
        n = C.c_uint64(0)
        self._chk(fn(self._ctx, None, C.byref(n)))
        out = np.empty((int(n.value), 3), dtype=np.float64)
        self._chk(fn(self._ctx, _f64p(out), C.byref(n)))
        return out + self._shift

    def get_gold(self):
        """The gold map as the device stores it (fp32 about the centring origin), exactly, in fp64 and
        in the caller's order: what the nearest-neighbour search compares against."""
        return self._stored(self._lib.ma_get_gold)

    def get_source_stored(self):
        """The comparison cloud as stored (see :meth:`get_gold`), caller's order."""
        return self._stored(self._lib.ma_get_source_stored)

    # -- the loop -------------------------------------------------------------------
    def icp(self, max_iter=50, tol=0.0, max_corr_dist=0.10, init=None, count_work=False, with_trace=False,
            time_kernels=False, lists=None):
        """The whole loop (rows a4-a7).  ``lists=False``: every launch of this run searches (``MA_FLAG_NO_LISTS``);
        ``None`` / ``True``: the context's setting (:meth:`list_config`).  Same bits either way."""
        p = self._params(max_iter, tol, max_corr_dist, count_work, time_kernels, lists)
        Tin = _conj_shift(_as_T(init), self._shift, inverse=True)
        Tout = np.empty((4, 4), dtype=np.float64)
        st = ma_stats()
        try:
            self._chk(self._lib.ma_icp(self._ctx, C.byref(p), _f64p(Tin), _f64p(Tout), C.byref(st)))
        finally:
            self._depth_keep = None          # ma_icp waits for the stream: an enqueued depth-frame copy is done
        Tout = _conj_shift(Tout, self._shift)
        tr = self.trace(st.iterations) if with_trace else None
        return AlignResult(Tout, int(st.iterations), float(st.rmse), float(st.inliers), bool(st.converged),
                           float(st.gpu_ms), tr)

    def list_config(self, enabled=None, skin_div=None, min_uses=None):
        """Per-context setting of the candidate lists (``ma_list_config``): on / off, skin = cell edge / ``skin_div``,
        reads that must be predicted to follow a write.  ``None`` keeps a value.  Speed only, never results."""
        self._chk(self._lib.ma_list_config(self._ctx, -1 if enabled is None else int(bool(enabled)),
                                           0.0 if skin_div is None else float(skin_div), 0 if min_uses is None else int(min_uses)))

    def planner_diag(self):
        """{waits, yielded, timeouts, enqueued} of the host-side planner of the last icp() / icp_sharded()."""
        out = (C.c_int32 * 4)()
        self._chk(self._lib.ma_get_planner_diag(self._ctx, out))
        return {"waits": int(out[0]), "yielded": int(out[1]), "timeouts": int(out[2]), "enqueued": int(out[3])}

    def list_stats(self):
        """{builds, uses}: launches of the last icp() that wrote / read the per-query candidate lists."""
        out = (C.c_int32 * 2)()
        self._chk(self._lib.ma_get_list_stats(self._ctx, out))
        return {"builds": int(out[0]), "uses": int(out[1])}

    def list_plan(self, max_iterations=4096):
        """Per iteration of the last icp(): 0 searched, 1 searched and wrote the candidate lists, 2 read them."""
        buf = np.zeros(max(int(max_iterations), 1), dtype=np.uint8)
        n = self._chk(self._lib.ma_get_list_plan(self._ctx, buf.ctypes.data_as(C.POINTER(C.c_uint8)), buf.shape[0]))
        return buf[:n].copy()

    def kernel_times(self):
        """{nn_ms, nn_launches, solve_ms, solve_launches} of the last icp(time_kernels=True)."""
        out = np.zeros(4, dtype=np.float64)
        self._chk(self._lib.ma_get_kernel_times(self._ctx, _f64p(out)))
        return {"nn_ms": out[0], "nn_launches": int(out[1]), "solve_ms": out[2], "solve_launches": int(out[3])}

    def iteration_times(self, max_iterations=4096):
        """[iterations, 2] milliseconds of the last icp(time_kernels=True): search launch, reduce + solve launch."""
        buf = np.zeros((max(int(max_iterations), 1), 2), dtype=np.float32)
        n = self._chk(self._lib.ma_get_iteration_times(self._ctx, buf.ctypes.data_as(C.POINTER(C.c_float)), buf.shape[0]))
        return buf[:n].astype(np.float64)

    def trace(self, max_rows=4096):
        buf = np.zeros((max(int(max_rows), 1), MA_TRACE_STRIDE), dtype=np.float64)
        rows = self._chk(self._lib.ma_get_trace(self._ctx, _f64p(buf), buf.shape[0]))
        out = buf[:rows].copy()
        if np.any(self._shift != 0.0):
            for r in out:
                r[:MA_ACC_LEN] = _acc_shift(r[:MA_ACC_LEN], self._shift)
        return out

    def icp_step(self, T_in=None, max_corr_dist=0.10, count_work=False):
        """One iteration at T_in: (acc17 world-frame, T_next)."""
        p = self._params(1, 0.0, max_corr_dist, count_work)
        Tin = _conj_shift(_as_T(T_in), self._shift, inverse=True)
        acc = np.empty(MA_ACC_LEN, dtype=np.float64)
        Tout = np.empty((4, 4), dtype=np.float64)
        self._chk(self._lib.ma_icp_step(self._ctx, C.byref(p), _f64p(Tin), _f64p(acc), _f64p(Tout)))
        return _acc_shift(acc, self._shift), _conj_shift(Tout, self._shift)

    def nn(self, T=None, max_corr_dist=np.inf):
        """Nearest gold index (original order; 0xFFFFFFFF = none within the gate) and d^2 per source point."""
        Tin = _conj_shift(_as_T(T), self._shift, inverse=True)
        idx = np.empty(self.n_source, dtype=np.uint32)
        d2 = np.empty(self.n_source, dtype=np.float32)
        self._chk(self._lib.ma_nn(self._ctx, _f64p(Tin), float(max_corr_dist),
                                  idx.ctypes.data_as(C.POINTER(C.c_uint32)),
                                  d2.ctypes.data_as(C.POINTER(C.c_float))))
        return idx, d2

    # -- split stepping (sharded-source multi-GPU mode) -------------------------------
    def state_set(self, T=None):
        self._chk(self._lib.ma_state_set(self._ctx, _f64p(_conj_shift(_as_T(T), self._shift, inverse=True))))

    def state_get(self):
        Tout = np.empty((4, 4), dtype=np.float64)
        st = ma_stats()
        self._chk(self._lib.ma_state_get(self._ctx, _f64p(Tout), C.byref(st)))
        return _conj_shift(Tout, self._shift), {"iterations": int(st.iterations), "converged": bool(st.converged),
                      "degenerate": bool(st.degenerate), "rmse": float(st.rmse), "inliers": float(st.inliers)}

    # -- exact sums: the bounds every shard of one comparison cloud must agree on (include/mapalign.h "Exact sums")
    def sum_bounds(self):
        """This context's own ``(n, wmax, rs)``: points, largest weight, largest distance of a point from the origin."""
        out = np.zeros(3, dtype=np.float64)
        self._chk(self._lib.ma_sum_bounds_get(self._ctx, _f64p(out)))
        return out

    def set_sum_bounds(self, bounds):
        """Install the WHOLE cloud's bounds (``combine_sum_bounds`` of every shard's); ``None`` clears them."""
        if bounds is None:
            self._chk(self._lib.ma_sum_bounds_set(self._ctx, None))
        else:
            self._chk(self._lib.ma_sum_bounds_set(self._ctx, _f64p(np.ascontiguousarray(bounds, dtype=np.float64).reshape(3))))

    @staticmethod
    def combine_sum_bounds(all_bounds):
        """Bounds of the union of shards: ``[sum n, max wmax, max rs]``."""
        b = np.asarray(all_bounds, dtype=np.float64).reshape(-1, 3)
        return np.array([b[:, 0].sum(), b[:, 1].max(), b[:, 2].max()])

    def sum_scale(self, max_corr_dist=0.10):
        """Exponents ``e[5]`` (W, s_p, s_q, S, E) of the fixed-point sums for a run gated at ``max_corr_dist``."""
        p = self._params(1, 0.0, max_corr_dist)
        out = (C.c_int32 * 5)()
        self._chk(self._lib.ma_sum_scale_get(self._ctx, C.byref(p), out))
        return np.array(out[:], dtype=np.int64)

    def sums_to_float(self, sums_i64, max_corr_dist=0.10):
        """The 17 fixed-point sums (what ``step_accumulate`` writes: int64) as numbers, centred frame."""
        e = self.sum_scale(max_corr_dist)
        group = np.array([0, 1, 1, 1, 2, 2, 2] + [3] * 9 + [4])
        return np.ldexp(np.asarray(sums_i64, dtype=np.int64).astype(np.float64), -e[group])

    def step_accumulate(self, acc_dev_ptr, max_corr_dist=0.10, count_work=False):
        p = self._params(1, 0.0, max_corr_dist, count_work)
        self._chk(self._lib.ma_step_accumulate(self._ctx, C.byref(p), C.c_void_p(int(acc_dev_ptr))))

    def step_solve(self, acc_dev_ptr, tol=0.0, max_corr_dist=0.10):
        p = self._params(1, tol, max_corr_dist)
        self._chk(self._lib.ma_step_solve(self._ctx, C.byref(p), C.c_void_p(int(acc_dev_ptr))))

    # -- RCCL inside the library (sharded-source mode) ----------------------
    def comm_unique_id(self):
        """Rank 0: the MA_COMM_ID_BYTES bootstrap id to hand to the other ranks (any channel)."""
        buf = (C.c_ubyte * MA_COMM_ID_BYTES)()
        st = self._lib.ma_comm_unique_id(buf)
        if st < 0:
            raise MapAlignError(st, (self._lib.ma_last_error(None) or b"").decode("utf-8", "replace"))
        return bytes(buf)

    def comm_init(self, unique_id, n_ranks, rank):
        """Collective: join the RCCL communicator described by ``unique_id`` as ``rank`` of ``n_ranks``."""
        if len(unique_id) != MA_COMM_ID_BYTES:
            raise ValueError(f"unique_id must be {MA_COMM_ID_BYTES} bytes")
        buf = (C.c_ubyte * MA_COMM_ID_BYTES).from_buffer_copy(unique_id)
        self._chk(self._lib.ma_comm_init(self._ctx, buf, int(n_ranks), int(rank)))
        self._comm_ranks = int(n_ranks)

    def comm_destroy(self):
        self._chk(self._lib.ma_comm_destroy(self._ctx))

    # -- peer-to-peer exchange of the sums (no library collective on the data path; include/mapalign.h) ----------
    def p2p_export(self):
        """This context's rows, allocated and zeroed: the MA_P2P_BLOB_BYTES to hand to every rank (any channel, rank order)."""
        buf = (C.c_ubyte * _lib.MA_P2P_BLOB_BYTES)()
        self._chk(self._lib.ma_p2p_export(self._ctx, buf))
        # (bytes 28-31 of the blob: bit 0 = the rows are fine-grained device memory, bit 1 = no IPC handle could be made)
        flags = int.from_bytes(bytes(buf[28:32]), "little")
        self.p2p_rows = {"fine_grained": bool(flags & 1), "ipc_handle": not (flags & 2)}
        return bytes(buf)

    def p2p_init(self, blobs, rank):
        """Map every rank's rows (``blobs``: the exports of all ranks, in rank order) and join as ``rank``."""
        blobs = [bytes(b) for b in blobs]
        if any(len(b) != _lib.MA_P2P_BLOB_BYTES for b in blobs):
            raise ValueError(f"every blob must be {_lib.MA_P2P_BLOB_BYTES} bytes")
        raw = b"".join(blobs)
        buf = (C.c_ubyte * len(raw)).from_buffer_copy(raw)
        self._chk(self._lib.ma_p2p_init(self._ctx, buf, len(blobs), int(rank)))

    def p2p_destroy(self):
        self._chk(self._lib.ma_p2p_destroy(self._ctx))

    def comm_config(self, transport):
        """Which exchange :meth:`icp_sharded` uses when both exist: ``"rccl"`` (default) or ``"p2p"``."""
        self._chk(self._lib.ma_comm_config(self._ctx, {"rccl": _lib.MA_TRANSPORT_RCCL, "p2p": _lib.MA_TRANSPORT_P2P}[transport]))

    def icp_sharded(self, max_iter=50, tol=0.0, max_corr_dist=0.10, init=None, count_work=False, with_trace=False, lists=None):
        """The whole sharded-source loop in the library: per iteration search (or list launch) -> 19-way sum ->
        ncclAllReduce(17 x int64) -> solve — or, over the peer-to-peer exchange, search -> sum + exchange + solve in one
        launch — all enqueued from C on the context stream; planner, candidate lists and early exit as in :meth:`icp`.
        The sums are exact integers: the result is :meth:`icp`'s of the whole cloud on one GPU, bit for bit."""
        p = self._params(max_iter, tol, max_corr_dist, count_work, False, lists)
        Tin = _conj_shift(_as_T(init), self._shift, inverse=True)
        Tout = np.empty((4, 4), dtype=np.float64)
        st = ma_stats()
        self._chk(self._lib.ma_icp_sharded(self._ctx, C.byref(p), _f64p(Tin), _f64p(Tout), C.byref(st)))
        Tout = _conj_shift(Tout, self._shift)
        tr = self.trace(st.iterations) if with_trace else None
        return AlignResult(Tout, int(st.iterations), float(st.rmse), float(st.inliers), bool(st.converged),
                           float(st.gpu_ms), tr)

    def comm_allgather(self, vec):
        """Gather a small float64 host vector from every rank -> [n_ranks, len(vec)] (RCCL all-gather)."""
        v = np.ascontiguousarray(vec, dtype=np.float64).reshape(-1)
        out = np.empty((self._comm_ranks, v.shape[0]), dtype=np.float64)
        self._chk(self._lib.ma_comm_allgather_f64(self._ctx, _f64p(v), v.shape[0], _f64p(out)))
        return out

    def comm_allreduce(self, vec, op="sum"):
        v = np.array(vec, dtype=np.float64).reshape(-1)
        self._chk(self._lib.ma_comm_allreduce_f64(self._ctx, _f64p(v), v.shape[0], {"sum": 0, "max": 1}[op]))
        return v
