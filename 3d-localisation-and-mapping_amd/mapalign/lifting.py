"""2-D detections -> 3-D global sign boxes (SURVEY.md §8 rows a10 / a11's output schema).

Host-side mirror of ``ProcessPose._3d_processing`` (``/root/reference/src/mapper/pose_processor.py:124-240``)
for the part that touches the registered cloud: per frame

    intrinsics / (image_width / depth_width)                       (:133-137)
    scaled = Transforms.scale_bounding_box(bbox, (W_img,H_img), (W_d,H_d))   (:174-178)
    corners = Transforms.bbox_to_3d(scaled, img_size)              (:181)
    corner_3d = Transforms._depth_to_3d(int(x), int(y), depth, fx, fy, cx, cy, scale_depth)  (:184-196)
    global    = _transform_to_global(corner_3d, pose)              (:199-201, :242-260)
    row       = global_corners + bbox[-3:]                         (:208)

The ``Transforms`` bodies are absent from the snapshot; the arithmetic follows the call
shapes above (linear box scaling, the four corners in (x1,y1) (x2,y1) (x2,y2) (x1,y2)
order, pinhole unprojection through the integer corner pixel).  The unprojection + pose
transform of all corners runs on the GPU (``ma_lift_boxes``).

WHICH DEPTH a corner gets is the one thing the surviving lines do not settle, so it is a switch
(``z_rule``, ``include/mapalign.h``):

* ``Z_CORNER`` (default) — the depth at the corner pixel: ``_depth_to_3d`` is called with
  ``int(x), int(y)`` and the whole depth image, once per corner (``:184-196``), and is handed no box;
* ``Z_MEDIAN`` — the comment right above that call, "Generate 3D corners with z-values from median
  over bbox (x, y) range" (``:183``): the median of the valid depths over the box's pixel rectangle,
  the same z for its four corners (a sign is flat; a corner pixel often falls just OFF the sign, on the
  wall metres behind it).

``bbox_to_3d(scaled_bbox, self.img_size)`` (``:181``) also receives ``img_size``; its only use
consistent with the following ``int(x), int(y)`` pixel reads is clamping the corners into the image,
which both rules do (at the depth image's size, the array those pixels index).

Detector rows (``src/detector/detector.py:155`` then ``:126-127``):
``[x1, y1, x2, y2, conf, label]`` -> after the damage classifier ``[x1, y1, x2, y2, damage_cls, conf, label]``.
"""
from __future__ import annotations

import numpy as np

from .transforms import pose_to_matrix


def insert_damage_class(pred_row, damage_cls):
    """detector.py:126-127: ``bbox.insert(-2, cls)``."""
    row = list(pred_row)
    row.insert(-2, damage_cls)
    return row


def parse_yolo_label_line(line, img_w, img_h):
    """One line of a YOLO txt label as ultralytics writes it with save_conf
    (``cls cx cy w h conf``, normalised) -> ``[x1, y1, x2, y2, conf, label]`` in pixels
    (the row layout of ``detector.py:136-157``)."""
    tok = line.split()
    label = int(float(tok[0]))
    cx, cy, w, h = (float(v) for v in tok[1:5])
    conf = float(tok[5]) if len(tok) > 5 else 1.0
    x1, y1 = (cx - w / 2) * img_w, (cy - h / 2) * img_h
    x2, y2 = (cx + w / 2) * img_w, (cy + h / 2) * img_h
    return [x1, y1, x2, y2, conf, label]


def scale_bounding_box(bbox, from_size, to_size):
    """Linear rescale of [x1,y1,x2,y2] from (W,H) = from_size to to_size (pose_processor.py:174-178)."""
    sx, sy = to_size[0] / from_size[0], to_size[1] / from_size[1]
    return [bbox[0] * sx, bbox[1] * sy, bbox[2] * sx, bbox[3] * sy]


def rescaled_intrinsics(camera_intrinsics, depth_width):
    """pose_processor.py:133-137."""
    s = camera_intrinsics["image_width"] / depth_width
    return (camera_intrinsics["fx"] / s, camera_intrinsics["fy"] / s,
            camera_intrinsics["cx"] / s, camera_intrinsics["cy"] / s)


Z_CORNER, Z_MEDIAN = 0, 1      # include/mapalign.h: MA_Z_CORNER / MA_Z_MEDIAN


def lift_frame_bboxes(engine, bboxes, depth_image, camera_intrinsics, pose_data, depth_width, depth_height,
                      scale_depth=1000.0, reuse_uploaded_depth=False, z_rule=Z_CORNER):
    """One frame of ``_3d_processing``: list of detector rows -> list of
    ``[g0, g1, g2, g3, damage_cls, conf, label]`` (corners as float64 xyz arrays).

    ``pose_data`` = ``[tx,ty,tz,qx,qy,qz,qw]`` (database_query.py:22 with the timestamp sliced off,
    pose_processor.py:109) or a 4x4 camera->world matrix."""
    if not bboxes:
        return []
    K = rescaled_intrinsics(camera_intrinsics, depth_width)
    T = np.asarray(pose_data, dtype=np.float64)
    T = T if T.shape == (4, 4) else pose_to_matrix(T.reshape(-1)[:7])
    size_img = (camera_intrinsics["image_width"], camera_intrinsics["image_height"])
    scaled = np.array([scale_bounding_box(b[:4], size_img, (depth_width, depth_height)) for b in bboxes], dtype=np.float32)
    corners, _ = engine.lift_boxes(scaled, K, T, None if reuse_uploaded_depth else depth_image, scale_depth, z_rule=z_rule)
    return [[corners[i, 0].copy(), corners[i, 1].copy(), corners[i, 2].copy(), corners[i, 3].copy()] + list(b[-3:])
            for i, b in enumerate(bboxes)]
