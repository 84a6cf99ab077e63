"""Batched sign detection feeding the box lift (BASELINE configs[3]:
"batch-32 YOLOv10 inference on PyTorch-ROCm + 2D->3D bbox projection into registered cloud").

Reference flow (``/root/reference/src/detector/detector.py``):

    self.model(source=..., batch=self.batch_size, conf=..., iou=...)            (:82-90)
    row = [x1, y1, x2, y2, conf, label]   in image pixels                       (:136-157)
    damage = self.damage_classifier(crops)                                      (:119-123, classifier.py:69-88)
    bbox.insert(-2, damage)  ->  [x1, y1, x2, y2, damage_cls, conf, label]      (:126-127)

and then per frame ``ProcessPose._3d_processing`` lifts the four corners of every box
(``src/mapper/pose_processor.py:172-208``).  north_star keeps the networks on stock
PyTorch-ROCm ("PyTorch-ROCm used only for the YOLOv10/BEiT inference"): this module is the
adaptor around ANY ``torch.nn.Module`` with YOLOv10's output contract, plus the one piece
that is ours — lifting the boxes of ALL frames of a batch in ONE device call
(``ma_lift_boxes_batch``) instead of one upload + one sync per frame.

Neither ``ultralytics`` nor any weights exist in this environment (SURVEY.md §8a row a11:
"weights unavailable => random-init YOLOv10 is acceptable for throughput only"), so
:func:`build_yolov10_like` constructs a YOLOv10-n-shaped network (CSP/C2f backbone, SPPF,
PAN neck, three decoupled one-to-one heads with DFL box regression over 8400 anchors at
640x640, NMS-free top-300 output — the v10 contract ``[B, 300, 6] = x1 y1 x2 y2 conf cls``)
with random weights, and :func:`build_beit_like` a BEiT classifier from ``transformers``'
config (random init).  With ``ultralytics`` installed, pass ``YOLOv10(weights).model``.
"""
from __future__ import annotations

import numpy as np

from . import lifting

MAX_DET = 300          # YOLOv10's NMS-free head keeps the top 300 candidates per image
REG_MAX = 16           # DFL bins per box side


def _torch():
    import torch
    return torch


def build_yolov10_like(num_classes=80, width=(16, 32, 64, 128, 256), max_det=MAX_DET):
    """Random-init network with YOLOv10-n's shape and output contract (throughput stand-in)."""
    torch = _torch()
    nn = torch.nn
    F = torch.nn.functional

    class ConvBnAct(nn.Module):
        def __init__(self, c1, c2, k=1, s=1):
            super().__init__()
            self.conv = nn.Conv2d(c1, c2, k, s, k // 2, bias=False)
            self.bn = nn.BatchNorm2d(c2)

        def forward(self, x):
            return F.silu(self.bn(self.conv(x)))

    class Bottleneck(nn.Module):
        def __init__(self, c):
            super().__init__()
            self.a, self.b = ConvBnAct(c, c, 3), ConvBnAct(c, c, 3)

        def forward(self, x):
            return x + self.b(self.a(x))

    class C2f(nn.Module):
        def __init__(self, c1, c2, n=1):
            super().__init__()
            self.c = c2 // 2
            self.cv1 = ConvBnAct(c1, 2 * self.c)
            self.m = nn.ModuleList(Bottleneck(self.c) for _ in range(n))
            self.cv2 = ConvBnAct((2 + n) * self.c, c2)

        def forward(self, x):
            y = list(self.cv1(x).chunk(2, 1))
            for m in self.m:
                y.append(m(y[-1]))
            return self.cv2(torch.cat(y, 1))

    class SPPF(nn.Module):
        def __init__(self, c):
            super().__init__()
            self.cv1, self.cv2 = ConvBnAct(c, c // 2), ConvBnAct(c * 2, c)

        def forward(self, x):
            y = [self.cv1(x)]
            for _ in range(3):
                y.append(F.max_pool2d(y[-1], 5, 1, 2))
            return self.cv2(torch.cat(y, 1))

    class Head(nn.Module):
        def __init__(self, c, nc):
            super().__init__()
            cb, cc = max(16, c // 4, 4 * REG_MAX // 4), max(c, min(nc, 100))
            self.box = nn.Sequential(ConvBnAct(c, cb, 3), ConvBnAct(cb, cb, 3), nn.Conv2d(cb, 4 * REG_MAX, 1))
            self.cls = nn.Sequential(ConvBnAct(c, cc, 3), ConvBnAct(cc, cc, 3), nn.Conv2d(cc, nc, 1))

        def forward(self, x):
            return self.box(x), self.cls(x)

    class YoloV10Like(nn.Module):
        strides = (8, 16, 32)

        def __init__(self):
            super().__init__()
            w = width
            self.nc, self.max_det = num_classes, max_det
            self.stem = nn.Sequential(ConvBnAct(3, w[0], 3, 2), ConvBnAct(w[0], w[1], 3, 2), C2f(w[1], w[1], 1))
            self.s3 = nn.Sequential(ConvBnAct(w[1], w[2], 3, 2), C2f(w[2], w[2], 2))
            self.s4 = nn.Sequential(ConvBnAct(w[2], w[3], 3, 2), C2f(w[3], w[3], 2))
            self.s5 = nn.Sequential(ConvBnAct(w[3], w[4], 3, 2), C2f(w[4], w[4], 1), SPPF(w[4]))
            self.up4, self.up3 = C2f(w[4] + w[3], w[3], 1), C2f(w[3] + w[2], w[2], 1)
            self.dn3, self.dn4 = ConvBnAct(w[2], w[2], 3, 2), ConvBnAct(w[3], w[3], 3, 2)
            self.pan4, self.pan5 = C2f(w[2] + w[3], w[3], 1), C2f(w[3] + w[4], w[4], 1)
            self.heads = nn.ModuleList(Head(c, num_classes) for c in (w[2], w[3], w[4]))
            self.register_buffer("bins", torch.arange(REG_MAX, dtype=torch.float32), persistent=False)

        def forward(self, x):
            p3 = self.s3(self.stem(x))
            p4 = self.s4(p3)
            p5 = self.s5(p4)
            n4 = self.up4(torch.cat([F.interpolate(p5, scale_factor=2.0, mode="nearest"), p4], 1))
            n3 = self.up3(torch.cat([F.interpolate(n4, scale_factor=2.0, mode="nearest"), p3], 1))
            o4 = self.pan4(torch.cat([self.dn3(n3), n4], 1))
            o5 = self.pan5(torch.cat([self.dn4(o4), p5], 1))
            boxes, scores = [], []
            for feat, head, stride in zip((n3, o4, o5), self.heads, self.strides):
                reg, cls = head(feat)
                b, _, h, w_ = reg.shape
                dist = (reg.view(b, 4, REG_MAX, h * w_).float().softmax(2) * self.bins.view(1, 1, -1, 1)).sum(2)
                ys, xs = torch.meshgrid(torch.arange(h, device=x.device), torch.arange(w_, device=x.device), indexing="ij")
                cx, cy = (xs.reshape(-1).float() + 0.5), (ys.reshape(-1).float() + 0.5)
                xyxy = torch.stack([cx - dist[:, 0], cy - dist[:, 1], cx + dist[:, 2], cy + dist[:, 3]], 2) * stride
                boxes.append(xyxy)
                scores.append(cls.view(b, self.nc, h * w_).float().sigmoid().transpose(1, 2))
            boxes, scores = torch.cat(boxes, 1), torch.cat(scores, 1)        # [B, 8400, 4], [B, 8400, nc]
            conf, label = scores.max(2)
            k = min(self.max_det, conf.shape[1])
            conf, idx = conf.topk(k, dim=1)
            boxes = boxes.gather(1, idx.unsqueeze(2).expand(-1, -1, 4))
            label = label.gather(1, idx)
            return torch.cat([boxes, conf.unsqueeze(2), label.unsqueeze(2).float()], 2)   # [B, k, 6]

    net = YoloV10Like()
    # He-normal weights keep the activations alive through the depth of the net, so that the
    # random-init stand-in produces varied boxes and scores (PyTorch's default conv init decays
    # to the biases after ~20 layers in eval mode).  Throughput is unaffected.
    for m in net.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_in", nonlinearity="relu")
            if m.bias is not None:
                nn.init.normal_(m.bias, 0.0, 0.5)
    return net


def build_beit_like(num_labels=2, image_size=224):
    """Random-init BEiT classifier from ``transformers`` (the reference loads a fine-tuned
    ``BeitForImageClassification``, classifier.py:40-53); ``None`` if transformers lacks it."""
    try:
        from transformers import BeitConfig, BeitForImageClassification
    except Exception:      # pragma: no cover - optional dependency
        return None
    cfg = BeitConfig(image_size=image_size, num_labels=num_labels)
    return BeitForImageClassification(cfg)


class Detector:
    """Adaptor: batched inference -> the reference's prediction rows.

    ``model(images [B,3,H,W] in 0..1) -> [B, K, 6]`` rows ``x1 y1 x2 y2 conf cls`` in input
    pixels (YOLOv10's NMS-free output).  ``damage_classifier`` (optional) maps crops
    ``[N,3,224,224]`` to logits (or an object with ``.logits``).

    ``crop`` — what the damage classifier is shown for a box:

    * ``"box"``: the box's pixels resized straight to ``crop_size`` (one bilinear resample);
    * ``"full_frame"``: the reference's geometry (``src/detector/detector.py:197-231`` then
      ``src/damage/classifier.py:69-88``).  ``_perform_homography`` maps the box corners
      ``(x1,y1) (x1,y2) (x2,y2) (x2,y1)`` onto the FRAME's corners ``(0,0) (0,H-1) (W-1,H-1) (W-1,0)``
      and ``cv2.warpPerspective`` resamples the whole frame to ``W x H`` (bilinear, pixels from outside
      the frame are 0) — for an axis-aligned box that homography is the scale-and-shift
      ``x = x1 + X (x2-x1)/(W-1)``, ``y = y1 + Y (y2-y1)/(H-1)``; the written image is then read back
      and the BEiT image processor resizes it to ``crop_size`` (PIL bilinear, i.e. with the
      anti-aliasing triangle filter of a reduction; ``microsoft/beit-*`` preprocessor configs:
      ``resample = 2``, no centre crop) and normalises with mean = std = 0.5.  Two resamples, not one:
      a 60-pixel sign is first blown up to the frame and then reduced again.
      (The reference writes ``(tf_img * 255).astype(np.uint8)`` of an image that is already uint8,
      which wraps every value v to (256 - v) mod 256; ``pixel_wrap=True`` reproduces that too —
      a fine-tuned model that was trained through the same code expects it.)"""

    def __init__(self, model, damage_classifier=None, batch_size=32, conf_thresh=0.25, device="cuda:0",
                 half=True, crop_size=224, crop="box", pixel_wrap=False):
        if crop not in ("box", "full_frame"):
            raise ValueError("crop must be 'box' or 'full_frame'")
        self.crop, self.pixel_wrap = crop, bool(pixel_wrap)
        torch = _torch()
        self.device = torch.device(device)
        self.dtype = torch.float16 if (half and self.device.type == "cuda") else torch.float32
        self.model = model.to(self.device, self.dtype).eval()
        self.damage_classifier = damage_classifier.to(self.device, self.dtype).eval() if damage_classifier is not None else None
        self.batch_size, self.conf_thresh, self.crop_size = int(batch_size), float(conf_thresh), int(crop_size)

    def _prepare(self, images):
        torch = _torch()
        x = torch.as_tensor(images)
        if x.ndim == 4 and x.shape[-1] == 3:                       # [B,H,W,3] uint8, as cv2.imread gives them
            x = x.permute(0, 3, 1, 2)
        x = x.to(self.device, non_blocking=True)
        return (x.to(self.dtype) / 255.0) if x.dtype == torch.uint8 else x.to(self.dtype)

    def infer(self, images):
        """Raw batched forward: [B,K,6] float32 tensor on the device (one launch sequence per batch of batch_size)."""
        torch = _torch()
        outs = []
        with torch.no_grad():
            x = self._prepare(images)
            for s in range(0, x.shape[0], self.batch_size):
                outs.append(self.model(x[s:s + self.batch_size]).float())
        return torch.cat(outs, 0), x

    def full_frame_crops(self, frames, boxes):
        """The reference's two-step resample for ``n`` boxes: ``frames`` [n,3,H,W] float32 in 0..1 (the
        frame each box was detected in), ``boxes`` [n,4] float ``x1 y1 x2 y2`` ->
        [n,3,crop_size,crop_size], normalised as the BEiT processor does (``(v - 0.5) / 0.5``).
        Step 1 = detector.py:197-231 (warp of the box onto the whole frame), step 2 = the image
        processor's resize inside classifier.py:76."""
        torch = _torch()
        F = torch.nn.functional
        n, _, H, W = frames.shape
        S = self.crop_size
        dev = frames.device
        x1, y1, x2, y2 = (boxes[:, k].to(torch.float32) for k in range(4))
        X = torch.arange(W, device=dev, dtype=torch.float32) / max(W - 1, 1)
        Y = torch.arange(H, device=dev, dtype=torch.float32) / max(H - 1, 1)
        sx = x1[:, None] + X[None, :] * (x2 - x1)[:, None]            # source column of output column X
        sy = y1[:, None] + Y[None, :] * (y2 - y1)[:, None]
        gx = sx * (2.0 / max(W - 1, 1)) - 1.0                          # align_corners=True: integer coordinates are pixel centres
        gy = sy * (2.0 / max(H - 1, 1)) - 1.0
        grid = torch.stack([gx[:, None, :].expand(n, H, W), gy[:, :, None].expand(n, H, W)], dim=-1)
        # cv2.warpPerspective(..., INTER_LINEAR, BORDER_CONSTANT 0): neighbours outside the frame count as 0
        warped = F.grid_sample(frames, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
        if self.pixel_wrap:                                            # (tf_img * 255).astype(np.uint8) of a uint8 image
            v = torch.round(warped * 255.0)
            warped = torch.remainder(256.0 - v, 256.0) / 255.0
        small = F.interpolate(warped, size=(S, S), mode="bilinear", antialias=True, align_corners=False)
        return (small - 0.5) / 0.5

    def _damage(self, x, det, keep):
        """BEiT-style damage class of every kept box: crop, resize to crop_size, classify (classifier.py:69-88)."""
        torch = _torch()
        F = torch.nn.functional
        if self.damage_classifier is None or not int(keep.sum()):
            return torch.zeros(int(keep.sum()), dtype=torch.long, device=det.device)
        b_idx, k_idx = keep.nonzero(as_tuple=True)
        H, W = x.shape[2], x.shape[3]
        S = self.crop_size
        if self.crop == "full_frame":
            box = det[b_idx, k_idx, :4]
            labels = []
            with torch.no_grad():
                for s in range(0, box.shape[0], self.batch_size):
                    e = min(s + self.batch_size, box.shape[0])
                    crops = self.full_frame_crops(x[b_idx[s:e]].float(), box[s:e])
                    out = self.damage_classifier(crops.to(self.dtype))
                    logits = out.logits if hasattr(out, "logits") else out
                    labels.append(logits.argmax(-1))
            return torch.cat(labels)
        # crop rectangles in whole pixels, as slicing image[y1:y2, x1:x2] gives them (at least 1 x 1)
        box = det[b_idx, k_idx, :4]
        xa = box[:, 0].clamp(0, W - 1).floor()
        ya = box[:, 1].clamp(0, H - 1).floor()
        xb = torch.minimum(torch.maximum(box[:, 2], xa + 1), torch.full_like(xa, W)).floor()
        yb = torch.minimum(torch.maximum(box[:, 3], ya + 1), torch.full_like(ya, H)).floor()
        # every crop of a classifier batch is cut and resized by ONE grid_sample: output pixel o
        # samples the crop at (o + 0.5) * w / S - 0.5, clamped to the crop (edge replication) —
        # the positions of F.interpolate(crop, bilinear, align_corners=False) — then shifted into
        # the image; align_corners=True makes integer coordinates pixel centres
        o = (torch.arange(S, device=det.device, dtype=torch.float32) + 0.5) / S
        labels = []
        with torch.no_grad():
            for s in range(0, box.shape[0], self.batch_size):
                e = min(s + self.batch_size, box.shape[0])
                w, h = (xb - xa)[s:e, None], (yb - ya)[s:e, None]
                sx = (o[None, :] * w - 0.5).clamp(min=0.0).minimum(w - 1.0) + xa[s:e, None]      # [n, S]
                sy = (o[None, :] * h - 0.5).clamp(min=0.0).minimum(h - 1.0) + ya[s:e, None]
                gx = sx * (2.0 / max(W - 1, 1)) - 1.0
                gy = sy * (2.0 / max(H - 1, 1)) - 1.0
                grid = torch.stack([gx[:, None, :].expand(-1, S, -1), gy[:, :, None].expand(-1, -1, S)], dim=-1)
                # (sampled in fp32: half-precision grid coordinates would be off by up to half a pixel)
                crops = F.grid_sample(x[b_idx[s:e]].float(), grid, mode="bilinear", padding_mode="border",
                                      align_corners=True)
                out = self.damage_classifier(crops.to(self.dtype))
                logits = out.logits if hasattr(out, "logits") else out
                labels.append(logits.argmax(-1))
        return torch.cat(labels)

    def forward(self, images):
        """-> ``dict[image_index] -> [[x1,y1,x2,y2,damage_cls,conf,label], ...]`` (detector.py:73-131)."""
        det, x = self.infer(images)
        keep = det[:, :, 4] >= self.conf_thresh
        damage = self._damage(x, det, keep).cpu().numpy()
        det_h, keep_h = det.cpu().numpy(), keep.cpu().numpy()
        preds, j = {}, 0
        for b in range(det_h.shape[0]):
            sel = det_h[b][keep_h[b]]                     # [k, 6] = x1 y1 x2 y2 conf label   (detector.py:155)
            k = sel.shape[0]
            dmg = damage[j:j + k]
            j += k
            # detector.py:126-127: bbox.insert(-2, damage) -> [x1, y1, x2, y2, damage_cls, conf, label]
            preds[b] = [[float(r[0]), float(r[1]), float(r[2]), float(r[3]), int(c), float(r[4]), int(r[5])]
                        for r, c in zip(sel, dmg)]
        return preds

    __call__ = forward


def lift_batch(engine, predictions, depth_frames, camera_intrinsics, poses, scale_depth=1000.0, z_rule=lifting.Z_CORNER):
    """Rows of ALL frames -> global 3-D boxes in ONE device call (``ma_lift_boxes_batch``).

    ``predictions``: the dict :meth:`Detector.forward` returns (frame index -> rows);
    ``depth_frames`` [F,H_d,W_d]; ``poses`` [F,7] (``tx ty tz qx qy qz qw``,
    database_query.py:22) or [F,4,4]; ``camera_intrinsics`` as in pose_processor.py:133-137.
    Returns ``dict[frame] -> [[g0,g1,g2,g3,damage_cls,conf,label], ...]`` (pose_processor.py:208);
    frames without detections are absent, as in the reference's loop."""
    d = np.asarray(depth_frames)
    F_, hd, wd = d.shape
    K = lifting.rescaled_intrinsics(camera_intrinsics, wd)
    P = np.asarray(poses, dtype=np.float64)
    T = P if P.shape[1:] == (4, 4) else np.stack([lifting.pose_to_matrix(p[:7]) for p in P])
    size_img = (camera_intrinsics["image_width"], camera_intrinsics["image_height"])
    frames = [f for f in sorted(predictions) if predictions[f]]
    if not frames:
        return {}
    counts = [len(predictions[f]) for f in frames]
    rows = [row for f in frames for row in predictions[f]]
    # Transforms.scale_bounding_box for every row at once (pose_processor.py:174-178)
    scale = np.array([wd / size_img[0], hd / size_img[1], wd / size_img[0], hd / size_img[1]], dtype=np.float64)
    boxes = (np.array([row[:4] for row in rows], dtype=np.float64) * scale).astype(np.float32)
    frame_of = np.repeat(np.asarray(frames, dtype=np.uint32), counts)
    corners, _ = engine.lift_boxes_batch(boxes, frame_of, d, K, T, scale_depth, z_rule=z_rule)
    out, i = {}, 0
    for f, k in zip(frames, counts):
        out[f] = [[corners[i + r, 0], corners[i + r, 1], corners[i + r, 2], corners[i + r, 3]] + list(predictions[f][r][-3:])
                  for r in range(k)]
        i += k
    return out
