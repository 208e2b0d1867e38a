"""GPU: the pipelined frame loop (depth-3 slots, one HIP stream each, pinned staging) against the ORACLE (not against the
device's own one-frame-at-a-time path): stream order, round-robin shards that reassemble into the one ordered stream,
BASELINE config 4's frame size (3840x2160) through FramePipeline, split-compare with labels, and a real two-process run."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bee_close(got, want, what):
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert d.max() <= 1 and (d > 0).mean() < 5e-3, (what, int(d.max()), float((d > 0).mean()))


def _source_frames(spec):
    from animal_vision_amd.renderers import VideoRenderer

    src = VideoRenderer(read_path=spec)
    src.open()
    out = []
    while True:
        f = src.get_image()
        if f is None:
            return out
        out.append(f)


@pytest.mark.parametrize("world", [1, 2, 3])
def test_run_video_vs_oracle_and_shards_reassemble(tmp_path, oracle, world):
    """dog (bit-exact) and honeybee (+-1 code) streams: every rank of `world` run in turn on this GPU, each touching only
    its own frames (strided source) and writing its shard of ONE sink path; merge_shards() must give the oracle's stream in
    frame order."""
    from animal_vision_amd.animals import HoneyBee
    from animal_vision_amd.dichromat import DichromatOp
    from animal_vision_amd.animals import Dog
    from animal_vision_amd.pipeline import run_video
    from animal_vision_amd.renderers import VideoRenderer

    spec = "synthetic:160x96:10:structured"
    frames = _source_frames(spec)
    assert len(frames) == 10
    for name, make_op, want_fn, exact in (("dog", lambda: DichromatOp(Dog.SPEC), lambda f: oracle.dichromat_visualize(oracle.DICHROMATS["dog"], f)[1], True),
                                          ("honeybee", lambda: HoneyBee()._operator(), lambda f: oracle.honeybee_visualize(f)[1], False)):
        path = str(tmp_path / f"{name}_{world}.npy")
        seen = 0
        rends = []
        for rank in range(world):
            vr = VideoRenderer(read_path=spec, write_path=path, rank=rank, world=world)
            vr.open()
            st = run_video(make_op(), vr, rank=rank, world=world, depth=3)
            vr.close()
            assert st.frames == len(range(rank, 10, world))
            seen += st.frames
            rends.append(vr)
        assert seen == 10
        rends[0].merge_shards()
        got = np.load(path)
        assert got.shape == (10, 96, 160, 3)
        for i, f in enumerate(frames):
            if exact:
                assert np.array_equal(got[i], want_fn(f)), (name, world, i)
            else:
                _bee_close(got[i], want_fn(f), (name, world, i))


def test_png_sink_carries_global_frame_indices(tmp_path, oracle):
    from animal_vision_amd.animals import Wolf
    from animal_vision_amd.dichromat import DichromatOp
    from animal_vision_amd.pipeline import run_video
    from animal_vision_amd.renderers import ImageRenderer, VideoRenderer

    spec = "synthetic:96x64:5"
    frames = _source_frames(spec)
    out_dir = str(tmp_path / "frames")
    for rank in range(2):
        vr = VideoRenderer(read_path=spec, write_path=out_dir, rank=rank, world=2)
        vr.open()
        run_video(DichromatOp(Wolf.SPEC), vr, rank=rank, world=2)
        vr.close()
    assert sorted(os.listdir(out_dir)) == [f"frame_{i:06d}.png" for i in range(5)]
    for i, f in enumerate(frames):
        got = ImageRenderer(os.path.join(out_dir, f"frame_{i:06d}.png")).get_image()
        assert np.array_equal(got, oracle.dichromat_visualize(oracle.DICHROMATS["wolf"], f)[1]), i


def test_4k_frames_through_frame_pipeline_vs_oracle(oracle):
    """BASELINE config 4's frame size: three 3840x2160 frames through FramePipeline (3 in flight), dog and honeybee."""
    from animal_vision_amd.animals import Dog, HoneyBee
    from animal_vision_amd.dichromat import DichromatOp
    from animal_vision_amd.pipeline import FramePipeline
    from animal_vision_amd.synthetic import noise_frame, structured_frame

    H, W = 2160, 3840
    for name, op, frames in (("dog", DichromatOp(Dog.SPEC), [noise_frame(0, H, W), structured_frame(1, H, W), noise_frame(2, H, W)]),
                             ("honeybee", HoneyBee()._operator(), [structured_frame(k, H, W) for k in range(3)])):
        pipe = FramePipeline(op, H, W, depth=3)
        got = {}
        st = pipe.run(((i, f) for i, f in enumerate(frames)), lambda i, o: got.__setitem__(i, o))
        pipe.close()
        assert st.frames == 3 and sorted(got) == [0, 1, 2]
        for i, f in enumerate(frames):
            if name == "dog":
                assert np.array_equal(got[i], oracle.dichromat_visualize(oracle.DICHROMATS["dog"], f)[1]), (name, i)
            else:
                _bee_close(got[i], oracle.honeybee_visualize(f)[1], (name, i))


def _oracle_split(oracle, original, modified, left="Original", right="Transformed"):
    """make_split_frame as the reference composes it (video.py:225-245), label pixels by the oracle's mirror of the device
    rasteriser, label geometry from the host rules (pinned with literal numbers in tests/test_labels.py)."""
    from animal_vision_amd.renderers import labels as L

    out = oracle.make_split_frame_nolabel(original, modified)
    H, W, _ = out.shape
    for text, org in ((left, (10, 24)), (right, L.right_label_origin(right, H, W))):
        fs, th, origin, box = L.label_layout(text, org, H, W)
        oracle.draw_label_pixels(out, box, L.text_segments(text, origin, fs), th + 2, th)
    return out


def test_split_compare_stream_with_labels(tmp_path, oracle):
    from animal_vision_amd.animals import Wolf
    from animal_vision_amd.dichromat import DichromatOp
    from animal_vision_amd.pipeline import run_video
    from animal_vision_amd.renderers import VideoRenderer

    for spec, n in (("synthetic:128x64:4", 4), ("synthetic:640x360:3:structured", 3)):
        path = str(tmp_path / f"split_{n}.npy")
        vr = VideoRenderer(read_path=spec, write_path=path)
        vr.open()
        run_video(DichromatOp(Wolf.SPEC), vr, split_compare=True)
        vr.close()
        got = np.load(path)
        for k, f in enumerate(_source_frames(spec)):
            want = _oracle_split(oracle, f, oracle.dichromat_visualize(oracle.DICHROMATS["wolf"], f)[1])
            assert np.array_equal(got[k], want), (spec, k)
    # and without labels: the bare composition
    vr = VideoRenderer(read_path="synthetic:128x64:2", write_path=str(tmp_path / "bare.npy"))
    vr.open()
    run_video(DichromatOp(Wolf.SPEC), vr, split_compare=True, labels=None)
    vr.close()
    for k, f in enumerate(_source_frames("synthetic:128x64:2")):
        assert np.array_equal(np.load(str(tmp_path / "bare.npy"))[k],
                              oracle.make_split_frame_nolabel(f, oracle.dichromat_visualize(oracle.DICHROMATS["wolf"], f)[1]))


def test_uv_species_stream_through_pipeline():
    """run_video-style streaming of a plane-program species: one recorded plan per slot, 3 frames in flight; the stream must
    equal the one-frame-at-a-time call (whose parity with the oracle is tests/test_uv_species_gpu.py's subject)."""
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
        pipe = FramePipeline(op, H, W, depth=3, split_compare=split, labels=None)
        got = {}
        pipe.run(((i, f) for i, f in enumerate(frames)), lambda i, o: got.__setitem__(i, o))
        pipe.close()
        op.close()
        for i, f in enumerate(frames):
            assert np.array_equal(got[i], split_compose(f, want[i]) if split else want[i]), (split, i)


def test_mst_honeybee_stream_op_cold_start_equals_one_frame_route(oracle):
    """The north-star stream route (ml/predict.py::MstHoneybeeStreamOp: MST++ cube -> honeybee tail, one slot stream per frame in
    flight) from a COLD predictor: the model's derived weights (packed fragments, stacked QKV, folded up-fuse weights) are built
    by torch kernels, and frames 1 and 2 run on other streams than frame 0 -- every frame of the stream, those included, must equal the
    one-frame-at-a-time route `predictor.honeybee(frame)` byte for byte, and that route's tail the oracle's on the same cube.
    A second pipeline over the same op (new slot streams; the first ones are destroyed) must give the same frames."""
    from animal_vision_amd.animals import HoneyBee
    from animal_vision_amd.ml import MSTPlusPlusPredictor, MstHoneybeeStreamOp
    from animal_vision_amd.pipeline import FramePipeline
    from animal_vision_amd.synthetic import structured_frame

    H, W = 96, 160
    frames = [structured_frame(20 + i, H, W) for i in range(7)]
    pred = MSTPlusPlusPredictor(None, seed=0, half=True)  # cold: nothing derived yet
    op = MstHoneybeeStreamOp(pred, HoneyBee()._operator(), H, W, depth=3)
    got = {}
    pipe = FramePipeline(op, H, W, depth=3)
    pipe.run(((i, f) for i, f in enumerate(frames)), lambda i, o: got.__setitem__(i, o))
    pipe.close()
    assert not op._streams  # the wrappers of the destroyed slot streams are gone
    bee = HoneyBee()._operator()
    want = [pred.honeybee(f, bee) for f in frames]
    for i in range(len(frames)):
        assert np.array_equal(got[i], want[i]), i
    # the one-frame route's tail against the oracle tail on the device's own cube (the forward pass is tests/test_mstpp.py's subject)
    lam = np.linspace(400.0, 700.0, 31, dtype=np.float32)
    w0, _ = oracle.honeybee_tail(*oracle.honeybee_catches(pred.predict(frames[1]), lam), np.uint8)
    _bee_close(want[1], w0, "tail vs oracle")
    got2 = {}
    pipe = FramePipeline(op, H, W, depth=3)
    pipe.run(((i, f) for i, f in enumerate(frames[:4])), lambda i, o: got2.__setitem__(i, o))
    pipe.close()
    for i in range(4):
        assert np.array_equal(got2[i], want[i]), ("second pipeline", i)


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


WORKER = textwrap.dedent(
    """
    import os, sys, json
    sys.path.insert(0, {root!r})
    import torch.distributed as dist
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from animal_vision_amd.animals import Dog
    from animal_vision_amd.dichromat import DichromatOp
    from animal_vision_amd.pipeline import run_video
    from animal_vision_amd.renderers import VideoRenderer
    vr = VideoRenderer(read_path="synthetic:320x180:13:structured", write_path={path!r}, rank=rank, world=world)
    vr.open()
    dist.barrier()
    tot = run_video(DichromatOp(Dog.SPEC), vr, rank=rank, world=world, dist=dist)
    vr.close()
    dist.barrier()
    if rank == 0:
        print(json.dumps({{"frames": tot.frames, "ranks": tot.ranks, "pixels": tot.pixels}}))
    dist.destroy_process_group()
    """
)


def test_two_process_stream_on_one_gpu(tmp_path, oracle):
    """Two real ranks (one process each, gloo for the barriers and the statistics, both on this box's one GPU): each touches
    only its frames, rank 0 reassembles the ordered stream behind the closing collective; checked against the oracle."""
    path = str(tmp_path / "two.npy")
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, path=path))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29581", str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert r == {"frames": 13, "ranks": 2, "pixels": 13 * 320 * 180}
    got = np.load(path)
    frames = _source_frames("synthetic:320x180:13:structured")
    assert got.shape[0] == 13
    for i, f in enumerate(frames):
        assert np.array_equal(got[i], oracle.dichromat_visualize(oracle.DICHROMATS["dog"], f)[1]), i


def test_stream_wait_forks_and_joins_lanes(oracle):
    """avx_stream_wait: frames enqueued on lane streams forked from a main stream, joined back into it, downloaded on the main
    stream without any host wait in between -- every frame must be complete (bench.py's UV / MST++ steps, FramePipeline slots)."""
    from animal_vision_amd import get_context
    from animal_vision_amd.animals import Dog
    from animal_vision_amd.dichromat import DichromatOp
    from animal_vision_amd.synthetic import noise_frame

    ctx = get_context()
    H, W, n = 540, 960, 6
    frames = [noise_frame(100 + i, H, W) for i in range(n)]
    want = [oracle.dichromat_visualize(oracle.DICHROMATS["dog"], f)[1] for f in frames]
    op = DichromatOp(Dog.SPEC, ctx)
    main = ctx.stream_create()
    lanes = [ctx.stream_create() for _ in range(3)]
    d_in = [ctx.malloc(frames[0].nbytes) for _ in range(n)]
    d_out = [ctx.malloc(frames[0].nbytes) for _ in range(n)]
    try:
        for rep in range(3):
            for i in range(n):
                ctx.upload(frames[i], d_in[i], stream=main)  # on the main stream: the lanes must wait for it
                ctx.memset(d_out[i], 0, stream=main)
            for ls in lanes:
                ctx.stream_wait(ls, main)
            for i in range(n):
                op.run_device(d_in[i], d_out[i], 1, H, W, lanes[i % len(lanes)])
            for ls in lanes:
                ctx.stream_wait(main, ls)
            got = [ctx.download(d_out[i], frames[i].shape, np.uint8, stream=main, sync=False) for i in range(n)]
            ctx.sync(main)
            for i in range(n):
                assert np.array_equal(got[i], want[i]), (rep, i)
    finally:
        for b in d_in + d_out:
            b.free()
        for s in lanes + [main]:
            ctx.stream_destroy(s)


def test_image_renderer_surface(tmp_path):
    """renderers/image.py's surface: render() remembers and saves, send_image() is its alias, render_split_compare() composes the same
    labelled half-and-half frame as VideoRenderer.make_split_frame, open() / close() exist; the GUI preview raises."""
    from animal_vision_amd.renderers import ImageRenderer, VideoRenderer
    from animal_vision_amd.synthetic import noise_frame, structured_frame

    a, b = noise_frame(1, 120, 200), structured_frame(2, 120, 200)
    out = os.path.join(str(tmp_path), "split.png")
    r = ImageRenderer(save_to=out)
    r.open()
    r.render_split_compare(a, b, left_label="Human", right_label="Cat")
    want = VideoRenderer(read_path=None, write_path=None).make_split_frame(a, b, left_label="Human", right_label="Cat")
    assert np.array_equal(r.visualized_image, want)
    assert np.array_equal(ImageRenderer(out).get_image(), want)
    assert not np.array_equal(want[:40, :100], a[:40, :100])  # the left label is drawn
    r.send_image(a)
    assert r.visualized_image is a
    r.close()
    with pytest.raises(NotImplementedError):
        ImageRenderer(show_window=True).render(a)
