"""GPU: the pipelined frame loop (depth-3 slots, one HIP stream each, pinned staging) returns exactly what the
one-frame-at-a-time species call returns, in stream order, for this rank's round-robin shard."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_run_video_matches_per_frame_visualize(tmp_path):
    from animal_vision_amd.animals import Dog, HoneyBee
    from animal_vision_amd.dichromat import DichromatOp
    from animal_vision_amd.pipeline import run_video
    from animal_vision_amd.renderers import VideoRenderer

    for species, op in ((Dog(), DichromatOp(Dog.SPEC)), (HoneyBee(), HoneyBee()._operator())):
        for world, rank in ((1, 0), (2, 1)):
            path = str(tmp_path / f"o_{type(species).__name__}_{world}_{rank}.npy")
            vr = VideoRenderer(read_path="synthetic:160x96:9:structured", write_path=path)
            vr.open()
            stats = run_video(op, vr, rank=rank, world=world, depth=3)
            vr.close()
            src = VideoRenderer(read_path="synthetic:160x96:9:structured")
            src.open()
            want = []
            i = 0
            while True:
                f = src.get_image()
                if f is None:
                    break
                if i % world == rank:
                    want.append(species.visualize(f)[1])
                i += 1
            got = np.load(path)
            assert stats.frames == len(want) == got.shape[0]
            assert np.array_equal(got, np.stack(want))


def test_split_compare_stream(tmp_path):
    from animal_vision_amd.animals import Wolf
    from animal_vision_amd.dichromat import DichromatOp
    from animal_vision_amd.pipeline import run_video
    from animal_vision_amd.renderers import VideoRenderer, split_compose
    from animal_vision_amd.synthetic import SyntheticVideoSource

    path = str(tmp_path / "split.npy")
    vr = VideoRenderer(read_path="synthetic:128x64:4", write_path=path)
    vr.open()
    run_video(DichromatOp(Wolf.SPEC), vr, split_compare=True)
    vr.close()
    got = np.load(path)
    src = SyntheticVideoSource(64, 128, 4)
    for k in range(4):
        f = src.get_image()
        assert np.array_equal(got[k], split_compose(f, Wolf().visualize(f)[1]))


def test_uv_species_stream_through_pipeline():
    """run_video-style streaming of a plane-program species: one recorded plan per slot, 3 frames in flight."""
    from animal_vision_amd.animals import Reindeer
    from animal_vision_amd.animals._uv_species import SpeciesStreamOp
    from animal_vision_amd.pipeline import FramePipeline
    from animal_vision_amd.renderers import split_compose
    from animal_vision_amd.synthetic import structured_frame

    H, W = 96, 128
    sp = Reindeer()
    frames = [structured_frame(i, H, W) for i in range(7)]
    want = [sp.visualize(f)[1] for f in frames]
    for split in (False, True):
        op = SpeciesStreamOp(sp, H, W, depth=3)
        pipe = FramePipeline(op, H, W, depth=3, split_compare=split)
        got = {}
        pipe.run(((i, f) for i, f in enumerate(frames)), lambda i, o: got.__setitem__(i, o))
        pipe.close()
        op.close()
        for i, f in enumerate(frames):
            assert np.array_equal(got[i], split_compose(f, want[i]) if split else want[i]), (split, i)


def test_baseline_config0_png_through_image_renderer(tmp_path, oracle):
    """BASELINE.json configs[0]: dog.py on one 640x480 PNG via ImageRenderer (the reference's own CPU-runnable case),
    here PNG -> get_image -> Dog().visualize on the device -> save PNG -> read back == the oracle's frame, bit for bit."""
    from animal_vision_amd.animals import Dog
    from animal_vision_amd.renderers import ImageRenderer
    from animal_vision_amd.synthetic import noise_frame

    src, dst = str(tmp_path / "in.png"), str(tmp_path / "out.png")
    ImageRenderer(save_to=src).render(noise_frame(0, 480, 640))
    frame = ImageRenderer(src).get_image()
    assert frame.shape == (480, 640, 3) and frame.dtype == np.uint8
    base, out = Dog().visualize(frame)
    assert base is frame
    ImageRenderer(save_to=dst).render(out)
    back = ImageRenderer(dst).get_image()
    _, want = oracle.dichromat_visualize(oracle.DICHROMATS["dog"], frame)
    assert np.array_equal(back, want)
