"""Corner labels of VideoRenderer.make_split_frame (reference: renderers/video.py:160-196 `_draw_label`, :241-245).

Host side: the GEOMETRY of the reference, restated -- font scale rule, text size, origin clamping, box corners -- and the
text as stroke segments in pixel coordinates; the pixels are drawn by csrc/labels.hip on the device.

Glyphs: the public-domain Hershey "Roman Simplex" strokes (A. V. Hershey, NBS 1967; the vector tables circulated by the
Usenet distribution and P. Bourke's transcription), entered here in the form `advance | polyline ; polyline ...`
with x to the right from the glyph's left bound, y UP from the baseline (capitals are 21 units tall, descenders reach
-7).  cv2.FONT_HERSHEY_SIMPLEX is this font: a glyph's advance is its right bound minus its left bound, and OpenCV's
getTextSize constants for it are cap_line = 12, base_line = 9 (units above / below the 'R' origin: 21 in all).
OpenCV itself is not installed anywhere in this pipeline: text SIZE and BOX follow its published formulas
(cv::getTextSize), the anti-aliased stroke pixels do not claim parity with its LINE_AA rasteriser (unpinned)."""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional, Tuple

import numpy as np

_G = {
    " ": "16|",
    "!": "10|5,21 5,7;5,2 4,1 5,0 6,1 5,2",
    '"': "16|4,21 4,14;12,21 12,14",
    "#": "21|11,25 4,-7;17,25 10,-7;4,12 18,12;3,6 17,6",
    "$": "20|8,25 8,-4;12,25 12,-4;17,18 15,20 12,21 8,21 5,20 3,18 3,16 4,14 5,13 7,12 13,10 15,9 16,8 17,6 17,3 15,1 12,0 8,0 5,1 3,3",
    "%": "24|21,21 3,0;8,21 10,19 10,17 9,15 7,14 5,14 3,16 3,18 4,20 6,21 8,21 10,20 13,19 16,19 19,20 21,21;17,7 15,6 14,4 14,2 16,0 18,0 20,1 21,3 21,5 19,7 17,7",
    "&": "26|23,12 23,13 22,14 21,14 20,13 19,11 17,6 15,3 13,1 11,0 7,0 5,1 4,2 3,4 3,6 4,8 5,9 12,13 13,14 14,16 14,18 13,20 11,21 9,20 8,18 8,16 9,13 11,10 16,3 18,1 20,0 22,0 23,1 23,2",
    "'": "10|5,19 4,20 5,21 6,20 6,18 5,16 4,15",
    "(": "14|11,25 9,23 7,20 5,16 4,11 4,7 5,2 7,-2 9,-5 11,-7",
    ")": "14|3,25 5,23 7,20 9,16 10,11 10,7 9,2 7,-2 5,-5 3,-7",
    "*": "16|8,21 8,9;3,18 13,12;13,18 3,12",
    "+": "26|13,18 13,0;4,9 22,9",
    ",": "10|6,1 5,0 4,1 5,2 6,1 6,-1 5,-3 4,-4",
    "-": "26|4,9 22,9",
    ".": "10|5,2 4,1 5,0 6,1 5,2",
    "/": "22|20,25 2,-7",
    "0": "20|9,21 6,20 4,17 3,12 3,9 4,4 6,1 9,0 11,0 14,1 16,4 17,9 17,12 16,17 14,20 11,21 9,21",
    "1": "20|6,17 8,18 11,21 11,0",
    "2": "20|4,16 4,17 5,19 6,20 8,21 12,21 14,20 15,19 16,17 16,15 15,13 13,10 3,0 17,0",
    "3": "20|5,21 16,21 10,13 13,13 15,12 16,11 17,8 17,6 16,3 14,1 11,0 8,0 5,1 4,2 3,4",
    "4": "20|13,21 3,7 18,7;13,21 13,0",
    "5": "20|15,21 5,21 4,12 5,13 8,14 11,14 14,13 16,11 17,8 17,6 16,3 14,1 11,0 8,0 5,1 4,2 3,4",
    "6": "20|16,18 15,20 12,21 10,21 7,20 5,17 4,12 4,7 5,3 7,1 10,0 11,0 14,1 16,3 17,6 17,7 16,10 14,12 11,13 10,13 7,12 5,10 4,7",
    "7": "20|17,21 7,0;3,21 17,21",
    "8": "20|8,21 5,20 4,18 4,16 5,14 7,13 11,12 14,11 16,9 17,7 17,4 16,2 15,1 12,0 8,0 5,1 4,2 3,4 3,7 4,9 6,11 9,12 13,13 15,14 16,16 16,18 15,20 12,21 8,21",
    "9": "20|16,14 15,11 13,9 10,8 9,8 6,9 4,11 3,14 3,15 4,18 6,20 9,21 10,21 13,20 15,18 16,14 16,9 15,4 13,1 10,0 8,0 5,1 4,3",
    ":": "10|5,14 4,13 5,12 6,13 5,14;5,2 4,1 5,0 6,1 5,2",
    ";": "10|5,14 4,13 5,12 6,13 5,14;6,1 5,0 4,1 5,2 6,1 6,-1 5,-3 4,-4",
    "<": "24|20,18 4,9 20,0",
    "=": "26|4,12 22,12;4,6 22,6",
    ">": "24|4,18 20,9 4,0",
    "?": "18|3,16 3,17 4,19 5,20 7,21 11,21 13,20 14,19 15,17 15,15 14,13 13,12 9,10 9,7;9,2 8,1 9,0 10,1 9,2",
    "@": "27|18,13 17,15 15,16 12,16 10,15 9,14 8,11 8,8 9,6 11,5 14,5 16,6 17,8;12,16 10,14 9,11 9,8 10,6 11,5;18,16 17,8 17,6 19,5 21,5 23,7 24,10 24,12 23,15 22,17 20,19 18,20 15,21 12,21 9,20 7,19 5,17 4,15 3,12 3,9 4,6 5,4 7,2 9,1 12,0 15,0 18,1 20,2 21,3;19,16 18,8 18,6 19,5",
    "A": "18|9,21 1,0;9,21 17,0;4,7 14,7",
    "B": "21|4,21 4,0;4,21 13,21 16,20 17,19 18,17 18,15 17,13 16,12 13,11;4,11 13,11 16,10 17,9 18,7 18,4 17,2 16,1 13,0 4,0",
    "C": "21|18,16 17,18 15,20 13,21 9,21 7,20 5,18 4,16 3,13 3,8 4,5 5,3 7,1 9,0 13,0 15,1 17,3 18,5",
    "D": "21|4,21 4,0;4,21 11,21 14,20 16,18 17,16 18,13 18,8 17,5 16,3 14,1 11,0 4,0",
    "E": "19|4,21 4,0;4,21 17,21;4,11 12,11;4,0 17,0",
    "F": "18|4,21 4,0;4,21 17,21;4,11 12,11",
    "G": "21|18,16 17,18 15,20 13,21 9,21 7,20 5,18 4,16 3,13 3,8 4,5 5,3 7,1 9,0 13,0 15,1 17,3 18,5 18,8;13,8 18,8",
    "H": "22|4,21 4,0;18,21 18,0;4,11 18,11",
    "I": "8|4,21 4,0",
    "J": "16|12,21 12,5 11,2 10,1 8,0 6,0 4,1 3,2 2,5 2,7",
    "K": "21|4,21 4,0;18,21 4,7;9,12 18,0",
    "L": "17|4,21 4,0;4,0 16,0",
    "M": "24|4,21 4,0;4,21 12,0;20,21 12,0;20,21 20,0",
    "N": "22|4,21 4,0;4,21 18,0;18,21 18,0",
    "O": "22|9,21 7,20 5,18 4,16 3,13 3,8 4,5 5,3 7,1 9,0 13,0 15,1 17,3 18,5 19,8 19,13 18,16 17,18 15,20 13,21 9,21",
    "P": "21|4,21 4,0;4,21 13,21 16,20 17,19 18,17 18,14 17,12 16,11 13,10 4,10",
    "Q": "22|9,21 7,20 5,18 4,16 3,13 3,8 4,5 5,3 7,1 9,0 13,0 15,1 17,3 18,5 19,8 19,13 18,16 17,18 15,20 13,21 9,21;12,4 18,-2",
    "R": "21|4,21 4,0;4,21 13,21 16,20 17,19 18,17 18,15 17,13 16,12 13,11 4,11;11,11 18,0",
    "S": "20|17,18 15,20 12,21 8,21 5,20 3,18 3,16 4,14 5,13 7,12 13,10 15,9 16,8 17,6 17,3 15,1 12,0 8,0 5,1 3,3",
    "T": "16|8,21 8,0;1,21 15,21",
    "U": "22|4,21 4,6 5,3 7,1 10,0 12,0 15,1 17,3 18,6 18,21",
    "V": "18|1,21 9,0;17,21 9,0",
    "W": "24|2,21 7,0;12,21 7,0;12,21 17,0;22,21 17,0",
    "X": "20|3,21 17,0;17,21 3,0",
    "Y": "18|1,21 9,11 9,0;17,21 9,11",
    "Z": "20|17,21 3,0;3,21 17,21;3,0 17,0",
    "[": "14|4,25 4,-7;5,25 5,-7;4,25 11,25;4,-7 11,-7",
    "\\": "14|0,21 14,-3",
    "]": "14|9,25 9,-7;10,25 10,-7;3,25 10,25;3,-7 10,-7",
    "^": "16|6,15 8,18 10,15;3,12 8,17 13,12;8,17 8,0",
    "_": "16|0,-2 16,-2",
    "`": "10|6,21 5,20 4,18 4,16 5,15 6,16 5,17",
    "a": "19|15,14 15,0;15,11 13,13 11,14 8,14 6,13 4,11 3,8 3,6 4,3 6,1 8,0 11,0 13,1 15,3",
    "b": "19|4,21 4,0;4,11 6,13 8,14 11,14 13,13 15,11 16,8 16,6 15,3 13,1 11,0 8,0 6,1 4,3",
    "c": "18|15,11 13,13 11,14 8,14 6,13 4,11 3,8 3,6 4,3 6,1 8,0 11,0 13,1 15,3",
    "d": "19|15,21 15,0;15,11 13,13 11,14 8,14 6,13 4,11 3,8 3,6 4,3 6,1 8,0 11,0 13,1 15,3",
    "e": "18|3,8 15,8 15,10 14,12 13,13 11,14 8,14 6,13 4,11 3,8 3,6 4,3 6,1 8,0 11,0 13,1 15,3",
    "f": "12|10,21 8,21 6,20 5,17 5,0;2,14 9,14",
    "g": "19|15,14 15,-2 14,-5 13,-6 11,-7 8,-7 6,-6;15,11 13,13 11,14 8,14 6,13 4,11 3,8 3,6 4,3 6,1 8,0 11,0 13,1 15,3",
    "h": "19|4,21 4,0;4,10 7,13 9,14 12,14 14,13 15,10 15,0",
    "i": "8|3,21 4,20 5,21 4,22 3,21;4,14 4,0",
    "j": "10|5,21 6,20 7,21 6,22 5,21;6,14 6,-3 5,-6 3,-7 1,-7",
    "k": "17|4,21 4,0;14,14 4,4;8,8 15,0",
    "l": "8|4,21 4,0",
    "m": "30|4,14 4,0;4,10 7,13 9,14 12,14 14,13 15,10 15,0;15,10 18,13 20,14 23,14 25,13 26,10 26,0",
    "n": "19|4,14 4,0;4,10 7,13 9,14 12,14 14,13 15,10 15,0",
    "o": "19|8,14 6,13 4,11 3,8 3,6 4,3 6,1 8,0 11,0 13,1 15,3 16,6 16,8 15,11 13,13 11,14 8,14",
    "p": "19|4,14 4,-7;4,11 6,13 8,14 11,14 13,13 15,11 16,8 16,6 15,3 13,1 11,0 8,0 6,1 4,3",
    "q": "19|15,14 15,-7;15,11 13,13 11,14 8,14 6,13 4,11 3,8 3,6 4,3 6,1 8,0 11,0 13,1 15,3",
    "r": "13|4,14 4,0;4,8 5,11 7,13 9,14 12,14",
    "s": "17|14,11 13,13 10,14 7,14 4,13 3,11 4,9 6,8 11,7 13,6 14,4 14,3 13,1 10,0 7,0 4,1 3,3",
    "t": "12|5,21 5,4 6,1 8,0 10,0;2,14 9,14",
    "u": "19|4,14 4,4 5,1 7,0 10,0 12,1 15,4;15,14 15,0",
    "v": "16|2,14 8,0;14,14 8,0",
    "w": "22|3,14 7,0;11,14 7,0;11,14 15,0;19,14 15,0",
    "x": "17|3,14 14,0;14,14 3,0",
    "y": "16|2,14 8,0;14,14 8,0 6,-4 4,-6 2,-7 1,-7",
    "z": "17|14,14 3,0;3,14 14,14;3,0 14,0",
    "{": "14|9,25 7,24 6,23 5,21 5,19 6,17 7,16 8,14 8,12 6,10;7,24 6,22 6,20 7,18 8,17 9,15 9,13 8,11 4,9 8,7 9,5 9,3 8,1 7,0 6,-2 6,-4 7,-6;6,8 8,6 8,4 7,2 6,1 5,-1 5,-3 6,-5 7,-6 9,-7",
    "|": "8|4,25 4,-7",
    "}": "14|5,25 7,24 8,23 9,21 9,19 8,17 7,16 6,14 6,12 8,10;7,24 8,22 8,20 7,18 6,17 5,15 5,13 6,11 10,9 6,7 5,5 5,3 6,1 7,0 8,-2 8,-4 7,-6;8,8 6,6 6,4 7,2 8,1 9,-1 9,-3 8,-5 7,-6 5,-7",
    "~": "24|3,6 3,8 4,11 6,12 8,12 10,11 14,8 16,7 18,7 20,8 21,10;3,8 4,10 6,11 8,11 10,10 14,7 16,6 18,6 20,7 21,10 21,12",
}


def _parse() -> Dict[str, Tuple[int, List[np.ndarray]]]:
    out = {}
    for ch, spec in _G.items():
        adv, strokes = spec.split("|")
        lines = [np.array([[float(v) for v in pt.split(",")] for pt in s.split()], np.float64) for s in strokes.split(";") if s.strip()]
        out[ch] = (int(adv), lines)
    return out


HERSHEY_SIMPLEX = _parse()
CAP_LINE, BASE_LINE = 12, 9  # cv::getTextSize's constants for FONT_HERSHEY_SIMPLEX


def cv_round(x: float) -> int:
    """cvRound: round half to even."""
    return int(np.rint(x))


def get_text_size(text: str, font_scale: float, thickness: int) -> Tuple[Tuple[int, int], int]:
    """cv2.getTextSize(text, FONT_HERSHEY_SIMPLEX, font_scale, thickness) -> ((width, height), baseline): width =
    cvRound(sum of advances * scale + thickness), height = cvRound((cap + base) * scale + (thickness + 1) / 2), baseline =
    cvRound(base * scale + thickness / 2).  Characters outside 32..126 count as '?' (OpenCV's substitution)."""
    view_x = 0.0
    for ch in text:
        view_x += HERSHEY_SIMPLEX.get(ch, HERSHEY_SIMPLEX["?"])[0] * font_scale
    w = cv_round(view_x + thickness)
    h = cv_round((CAP_LINE + BASE_LINE) * font_scale + (thickness + 1) / 2)
    return (w, h), cv_round(BASE_LINE * font_scale + thickness * 0.5)


def label_font_scale(H: int) -> float:
    """renderers/video.py:171."""
    return max(0.5, min(1.2, H / 900.0))


def label_layout(text: str, org: Tuple[int, int], H: int, W: int):
    """renderers/video.py:168-187 -> (font_scale, thickness, (x, y) text origin after clamping, (x0, y0, x1, y1) box, inclusive)."""
    font_scale, thickness, pad = label_font_scale(H), 2, 8
    (tw, th), baseline = get_text_size(text, font_scale, thickness)
    x, y = org
    if x + tw + pad > W:
        x = W - tw - pad
    if y - th - baseline - pad < 0:
        y = th + baseline + pad
    x0, y0 = max(x - pad, 0), max(y - th - baseline - pad, 0)
    x1, y1 = min(x + tw + pad, W - 1), min(y + baseline + pad, H - 1)
    return font_scale, thickness, (x, y), (x0, y0, x1, y1)


def right_label_origin(text: str, H: int, W: int) -> Tuple[int, int]:
    """renderers/video.py:243-244: measured with thickness 1 and a 0.45 floor on the scale (not the drawing's 0.5 / 2)."""
    (tw, _), _ = get_text_size(text, max(0.45, min(1.2, H / 900.0)), 1)
    return max(W - tw - 10, 10), 24


def text_segments(text: str, origin: Tuple[int, int], font_scale: float) -> np.ndarray:
    """Stroke segments of `text` with its baseline-left at `origin`, in pixel coordinates: (n, 6) float32 rows
    {ax, ay, bx - ax, by - ay, 1 / |b - a|^2 (0 for a zero-length segment), 0} (csrc/labels.hip)."""
    segs = []
    pen = 0.0
    ox, oy = origin
    for ch in text:
        adv, lines = HERSHEY_SIMPLEX.get(ch, HERSHEY_SIMPLEX["?"])
        for ln in lines:
            px = ox + (pen + ln[:, 0]) * font_scale
            py = oy - ln[:, 1] * font_scale
            for k in range(len(ln) - 1):
                segs.append((px[k], py[k], px[k + 1] - px[k], py[k + 1] - py[k]))
            if len(ln) == 1:
                segs.append((px[0], py[0], 0.0, 0.0))
        pen += adv
    s = np.zeros((len(segs), 6), np.float32)
    if segs:
        s[:, :4] = np.asarray(segs, np.float64).astype(np.float32)
        l2 = s[:, 2] * s[:, 2] + s[:, 3] * s[:, 3]
        s[:, 4] = np.where(l2 > 0, np.float32(1.0) / np.where(l2 > 0, l2, np.float32(1.0)), np.float32(0.0))
    return s


def draw_label_device(ctx, d_img, H: int, W: int, text: str, org: Tuple[int, int], slot: int, stream=None) -> None:
    """_draw_label (renderers/video.py:160-196) on a device-resident uint8 HxWx3 frame, in place."""
    from .._lib import lib

    font_scale, thickness, origin, box = label_layout(text, org, H, W)
    segs = np.ascontiguousarray(text_segments(text, origin, font_scale))
    box_c = (ctypes.c_int * 4)(*box)
    ctx._check(lib.avx_draw_label_u8(ctx._h, d_img.ptr if hasattr(d_img, "ptr") else d_img, H, W, box_c, segs.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                     len(segs), float(thickness + 2), float(thickness), slot, ctx._s(stream)))
    # (the library copies the segment table into its own mirror before it returns; the box travels in the kernel arguments)


def draw_split_labels_device(ctx, d_img, H: int, W: int, left_label: Optional[str], right_label: Optional[str], stream=None) -> None:
    """renderers/video.py:241-245 on the device: left label at (10, 24), right label right-aligned 10 px from the edge."""
    if left_label is not None:
        draw_label_device(ctx, d_img, H, W, left_label, (10, 24), 0, stream)
    if right_label is not None:
        draw_label_device(ctx, d_img, H, W, right_label, right_label_origin(right_label, H, W), 1, stream)


def draw_split_labels(frame: np.ndarray, left_label: Optional[str], right_label: Optional[str]) -> np.ndarray:
    """NumPy in / out: uploads the composed frame, draws both labels on the device, downloads."""
    from ..runtime import get_context

    if frame.dtype != np.uint8:
        raise TypeError("labels are drawn on uint8 frames (what get_image() yields)")
    ctx = get_context()
    H, W, _ = frame.shape
    d = ctx.upload(np.ascontiguousarray(frame))
    try:
        draw_split_labels_device(ctx, d, H, W, left_label, right_label)
        return ctx.download(d, frame.shape, np.uint8)
    finally:
        d.free()
