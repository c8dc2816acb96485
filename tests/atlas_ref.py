"""Test infrastructure: numpy restatement of the atlas path (src/renderer/atlas.ts + potpack@2.0.0 +
renderer.ts:246-261), written independently of host/atlas.js so the two can be compared byte for byte.
Scaling rule (bilinear at pixel centres, premultiplied, round half up) is this build's definition — the
browser's drawImage filter is implementation-defined ("parity unpinned")."""
import math

import numpy as np

RATIO = 0.5     # atlas.ts:10


def potpack(boxes):
    """boxes: list of dicts with w, h; sets x, y. Returns (w, h). potpack@2.0.0 (mapbox/potpack index.js)."""
    area = sum(b["w"] * b["h"] for b in boxes)
    widest = max([b["w"] for b in boxes], default=0)
    boxes.sort(key=lambda b: -b["h"])                 # Python's sort is stable, like V8's
    start = max(math.ceil(math.sqrt(area / 0.95)), widest)
    spaces = [[0, 0, start, math.inf]]                # x, y, w, h
    width = height = 0
    for b in boxes:
        for i in reversed(range(len(spaces))):
            sx, sy, sw, sh = spaces[i]
            if b["w"] > sw or b["h"] > sh:
                continue
            b["x"], b["y"] = sx, sy
            height = max(height, sy + b["h"])
            width = max(width, sx + b["w"])
            if b["w"] == sw and b["h"] == sh:
                last = spaces.pop()
                if i < len(spaces):
                    spaces[i] = last
            elif b["h"] == sh:
                spaces[i] = [sx + b["w"], sy, sw - b["w"], sh]
            elif b["w"] == sw:
                spaces[i] = [sx, sy + b["h"], sw, sh - b["h"]]
            else:
                spaces.append([sx + b["w"], sy, sw - b["w"], b["h"]])
                spaces[i] = [sx, sy + b["h"], sw, sh - b["h"]]
            break
    return width, height


def _half_up(v):
    return np.clip(np.floor(np.asarray(v, np.float64) + 0.5), 0, 255)


def _clamped_u8(v):
    """Uint8ClampedArray store: round to nearest, ties to even, clamp."""
    return np.clip(np.rint(np.asarray(v, np.float64)), 0, 255)


def _span(start, extent):
    """canvas pixels whose centres lie inside [start, start + extent)"""
    if not extent > 0:
        return 0, 0
    first, end = math.ceil(start - 0.5), math.ceil(start + extent - 0.5)
    return first, max(0, end - first)


def resample(img, dx, dy, dw, dh):
    """img (H, W, 4) uint8 straight alpha drawn into the rectangle (dx, dy, dw, dh) ->
    (x0, y0, (nh, nw, 4) float64 premultiplied rgb + alpha)."""
    sh, sw = img.shape[:2]
    cx, nw = _span(dx, dw)
    cy, nh = _span(dy, dh)
    src = img.astype(np.float64)
    pre = src.copy()
    pre[..., :3] = src[..., :3] * src[..., 3:4] / 255
    fx = (cx + np.arange(nw) + 0.5 - dx) * sw / dw - 0.5 if nw else np.zeros(0)
    fy = (cy + np.arange(nh) + 0.5 - dy) * sh / dh - 0.5 if nh else np.zeros(0)
    x0, y0 = np.floor(fx), np.floor(fy)
    tx, ty = (fx - x0)[None, :, None], (fy - y0)[:, None, None]
    xa, xb = np.clip(x0, 0, sw - 1).astype(int), np.clip(x0 + 1, 0, sw - 1).astype(int)
    ya, yb = np.clip(y0, 0, sh - 1).astype(int), np.clip(y0 + 1, 0, sh - 1).astype(int)
    top = (1 - tx) * pre[ya][:, xa] + tx * pre[ya][:, xb]
    bot = (1 - tx) * pre[yb][:, xa] + tx * pre[yb][:, xb]
    return cx, cy, (1 - ty) * top + ty * bot


def build(materials, images):
    """materials: list of dicts {albedo, normal, pbr, emissive: image index or None} in glTF material order;
    images: list of (H, W, 4) uint8. Returns (rects [per material dict name -> (x, y, w, h) floats],
    canvas (S, S, 4) uint8, atlas (S, S, 4) float16)."""
    boxes, per_material = [], []
    for m in materials:
        entry = {}
        for key in ("normal", "albedo", "pbr", "emissive"):             # push order of atlas.ts:56-59
            idx = m.get(key)
            if idx is None:
                box = {"w": 0, "h": 0, "x": 0, "y": 0}
            else:
                box = {"w": images[idx].shape[1] * RATIO, "h": images[idx].shape[0] * RATIO, "x": 0, "y": 0}
            box["image"], box["albedo"] = idx, key == "albedo"
            entry[key] = box
            boxes.append(box)
        per_material.append(entry)
    w, h = potpack(boxes)
    m = max(w, h)
    size = int(max(1, 2 ** math.ceil(math.log2(m)))) if m > 0 else 1
    canvas = np.zeros((size, size, 4), np.uint8)
    canvas[..., 3] = 255
    for entry in per_material:
        for key in ("albedo", "normal", "pbr", "emissive"):             # draw order of atlas.ts:169-181
            box = entry[key]
            if box["image"] is None:
                continue
            x, y, s = resample(images[box["image"]], box["x"], box["y"], box["w"], box["h"])
            rgb, a = s[..., :3], s[..., 3:4]
            if box["albedo"]:
                a8 = _half_up(a)
                with np.errstate(divide="ignore", invalid="ignore"):
                    straight = np.where(a8 > 0, _half_up(_half_up(rgb) * 255 / a8), 0.0)
                lin = _clamped_u8(np.power(straight / 255, 2.2) * 255)
                rgb = lin * a8 / 255
            px = _half_up(rgb).astype(np.uint8)
            hh, ww = min(px.shape[0], size - y), min(px.shape[1], size - x)
            canvas[y:y + hh, x:x + ww, :3] = px[:hh, :ww]
    atlas = (canvas.astype(np.float32) / np.float32(255)).astype(np.float16)
    rects = [{k: (e[k]["x"], e[k]["y"], e[k]["w"], e[k]["h"]) for k in e} for e in per_material]
    return rects, canvas, atlas
