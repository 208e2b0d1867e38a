"""make_split_frame's labels and resize branch (reference: renderers/video.py:160-196, :225-245).

Pinned here: the GEOMETRY the reference computes -- font scale rule, cv2.getTextSize's formulas for FONT_HERSHEY_SIMPLEX
(advances from the public Hershey tables; cap_line 12, base_line 9), origin clamping, box corners, right-label origin -- with
literal numbers worked out by hand from those lines.  The label PIXELS are compared device <-> oracle mirror only: OpenCV's
LINE_AA rasteriser is not restated (parity unpinned; OpenCV is not installed anywhere in this pipeline)."""
import numpy as np
import pytest


def test_text_size_follows_getTextSize_formulas():
    from animal_vision_amd.renderers import labels as L

    # advances (Hershey simplex, right bound - left bound): O22 r13 i8 g19 i8 n19 a19 l8 = 116; T16 r13 a19 n19 s17 f12 o19 r13 m30 e18 d19 = 195
    assert sum(L.HERSHEY_SIMPLEX[c][0] for c in "Original") == 116
    assert sum(L.HERSHEY_SIMPLEX[c][0] for c in "Transformed") == 195
    # width = cvRound(116 * s + t); height = cvRound(21 * s + (t + 1) / 2); baseline = cvRound(9 * s + t / 2)
    assert L.get_text_size("Original", 1.0, 2) == ((118, 22), 10)          # 118, round(22.5) = 22 (half to even), 10
    assert L.get_text_size("Original", 0.5, 2) == ((60, 12), 6)            # 60, round(12.0), round(5.5) = 6
    assert L.get_text_size("Transformed", 1.2, 2) == ((236, 27), 12)       # 236, round(26.7), round(11.8)
    assert L.get_text_size("Transformed", 0.45, 1) == ((89, 10), 5)        # round(88.75), round(10.45), round(4.55)
    assert L.get_text_size("", 1.0, 2) == ((2, 22), 10)
    assert L.get_text_size("é", 1.0, 1) == L.get_text_size("?", 1.0, 1)  # outside 32..126 -> '?'
    assert len(L.HERSHEY_SIMPLEX) == 95 and all(chr(c) in L.HERSHEY_SIMPLEX for c in range(32, 127))
    for ch, (adv, lines) in L.HERSHEY_SIMPLEX.items():  # strokes stay inside the glyph's advance box and the font's vertical extent
        for ln in lines:
            assert ln[:, 0].min() >= 0 and ln[:, 0].max() <= adv and ln[:, 1].min() >= -7 and ln[:, 1].max() <= 25, ch


def test_label_layout_follows_the_reference_rules():
    from animal_vision_amd.renderers import labels as L

    # font_scale = max(0.5, min(1.2, h / 900))  (video.py:171), thickness 2, pad 8
    assert L.label_font_scale(360) == 0.5 and L.label_font_scale(900) == 1.0 and L.label_font_scale(2160) == 1.2
    assert L.label_font_scale(720) == pytest.approx(0.8)
    # H = 900, "Original" at (10, 24): tw 118, th 22, baseline 10.  y - th - baseline - pad = -16 < 0 -> y = th + baseline + pad = 40.
    # box: x0 = max(10 - 8, 0) = 2, y0 = max(40 - 22 - 10 - 8, 0) = 0, x1 = min(10 + 118 + 8, W - 1) = 136, y1 = min(40 + 10 + 8, H - 1) = 58
    assert L.label_layout("Original", (10, 24), 900, 1600) == (1.0, 2, (10, 40), (2, 0, 136, 58))
    # 4K (scale 1.2): tw = cvRound(139.2 + 2) = 141, th = cvRound(25.2 + 1.5) = 27, baseline = cvRound(10.8 + 1) = 12 -> y = 47
    assert L.label_layout("Original", (10, 24), 2160, 3840) == (1.2, 2, (10, 47), (2, 0, 159, 67))
    # right label: measured at max(0.45, ..) and thickness 1 (video.py:243): 1080p -> scale 1.2, tw = cvRound(234 + 1) = 235 -> x = 1920 - 235 - 10
    assert L.right_label_origin("Transformed", 1080, 1920) == (1675, 24)
    assert L.right_label_origin("Transformed", 360, 640) == (640 - 89 - 10, 24)
    assert L.right_label_origin("Transformed", 100, 60) == (10, 24)  # max(W - tw - 10, 10)
    # a label that would leave the frame on the right is pulled in: x = W - tw - pad
    assert L.label_layout("Transformed", (1675, 24), 1080, 1920) == (1.2, 2, (1675, 47), (1667, 0, 1919, 67))  # 1675 + 236 + 8 = 1919: fits
    # H = 360 (scale 0.5): tw = cvRound(97.5 + 2) = 100 (99.5 -> even), th = 12, baseline = cvRound(5.5) = 6; in a 150-px-wide frame
    # 100 + 100 + 8 > 150 -> x = 150 - 100 - 8 = 42; y = 24 - 12 - 6 - 8 < 0 -> y = 26
    assert L.label_layout("Transformed", (100, 24), 360, 150) == (0.5, 2, (42, 26), (34, 0, 149, 40))
    segs = L.text_segments("Hi", (100, 50), 2.0)
    assert segs.shape == (3 + 4 + 1, 6) and segs.dtype == np.float32  # H: three strokes of one segment; i: its dot (4 segments) and its stem
    assert np.allclose(segs[0, :4], [100 + 8, 50 - 42, 0, 42])  # H's left stem: (4,21)->(4,0), scaled by 2, y down


@pytest.mark.gpu
def test_device_labels_equal_the_oracle_mirror(oracle):
    from animal_vision_amd.renderers import VideoRenderer, labels as L, split_compose
    from animal_vision_amd.synthetic import noise_frame, structured_frame

    vr = VideoRenderer()
    for (H, W), pair in (((360, 640), ("Original", "Transformed")), ((96, 128), ("Original", "Transformed")), ((1080, 1920), ("Human", "Honeybee (UV)")),
                         ((48, 40), ("a much too long label", "Q{}|~"))):
        a, b = structured_frame(0, H, W), noise_frame(1, H, W)
        got = vr.make_split_frame(a, b, left_label=pair[0], right_label=pair[1])
        want = oracle.make_split_frame_nolabel(a, b)
        for text, org in ((pair[0], (10, 24)), (pair[1], L.right_label_origin(pair[1], H, W))):
            fs, th, origin, box = L.label_layout(text, org, H, W)
            oracle.draw_label_pixels(want, box, L.text_segments(text, origin, fs), th + 2, th)
        assert got.shape == a.shape and got.dtype == np.uint8
        assert np.array_equal(got, want), ((H, W), int(np.abs(got.astype(int) - want.astype(int)).max()))
        # properties of the reference's drawing that do not depend on the rasteriser: outside both boxes the composition is
        # untouched; inside a box but > 4 px from any stroke the pixel is cvRound(0.4 * composed) (video.py:189-191)
        bare = split_compose(a, b)
        fs, th, origin, box = L.label_layout(pair[0], (10, 24), H, W)
        x0, y0, x1, y1 = box
        outside = np.ones((H, W), bool)
        outside[y0 : y1 + 1, x0 : x1 + 1] = False
        fs2, th2, origin2, box2 = L.label_layout(pair[1], L.right_label_origin(pair[1], H, W), H, W)
        outside[box2[1] : box2[3] + 1, box2[0] : box2[2] + 1] = False
        grow = 4
        far = np.zeros((H, W), bool)
        far[max(y0 - grow, 0) : y1 + 1 + grow, max(x0 - grow, 0) : x1 + 1 + grow] = True
        far[max(box2[1] - grow, 0) : box2[3] + 1 + grow, max(box2[0] - grow, 0) : box2[2] + 1 + grow] = True
        assert np.array_equal(got[~far], bare[~far])
        if H >= 96:
            # the white core of 'O' in "Original": its leftmost vertical run passes through (3, 8..13) glyph units
            cx, cy = origin[0] + 3 * fs, origin[1] - 10.5 * fs
            iy, ix = int(round(cy)), int(round(cx))
            assert got[iy - 1 : iy + 2, ix - 1 : ix + 2].min(axis=2).max() >= 200  # the stroke's white core passes within a pixel
            # a dimmed pixel well inside the box and away from the text: bottom-left pad corner
            px, py = x0 + 1, y1 - 1
            if not (box2[0] <= px <= box2[2] and box2[1] <= py <= box2[3]):
                assert np.array_equal(got[py, px], np.rint(bare[py, px].astype(np.float32) * np.float32(0.4)).astype(np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("src_hw", [(180, 320), (97, 131), (36, 64), (720, 1280)])
def test_split_frame_resizes_modified_like_the_reference(oracle, src_hw):
    """video.py:228-231: a `modified` of another size is cv2.resize(..., INTER_AREA)d to the original's size first: integer
    ratio (2x2 special case and 4x4), general ratio, and enlarging (INTER_AREA then behaves as INTER_LINEAR)."""
    from animal_vision_amd.renderers import split_compose
    from animal_vision_amd.synthetic import noise_frame, structured_frame

    H, W = 90, 160
    a = structured_frame(2, H, W)
    if src_hw == (720, 1280):
        a = structured_frame(2, 180, 320)
        H, W = 180, 320
    b = noise_frame(3, *src_hw)
    got = split_compose(a, b)
    want = oracle.make_split_frame_nolabel(a, b)
    assert got.shape == (H, W, 3) and np.array_equal(got, want)
    with pytest.raises(AssertionError):
        split_compose(a[..., 0], b)
