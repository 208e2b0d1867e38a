"""CPU-only: frame sharding / ordering / statistics reduction of the frame loop, including a world_size-2
`gloo` run (the N > 1 path of bench.py and pipeline.run_video without a GPU), and the renderer surface."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT


def test_round_robin_sharding_partitions_the_stream():
    from animal_vision_amd.pipeline import merge_in_order, owner_of, shard_indices

    for n in (0, 1, 7, 64, 257):
        for world in (1, 2, 4, 8):
            shards = [shard_indices(n, r, world) for r in range(world)]
            assert sorted(i for s in shards for i in s) == list(range(n))
            assert all(owner_of(i, world) == r for r, s in enumerate(shards) for i in s)
            assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
            merged = merge_in_order([[(i, f"f{i}") for i in s] for s in shards])
            assert merged == [f"f{i}" for i in range(n)]
    with pytest.raises(ValueError):
        shard_indices(4, 2, 2)
    with pytest.raises(ValueError):
        merge_in_order([[(0, "a")], [(2, "b")]])


def test_video_renderer_surface(tmp_path):
    from animal_vision_amd.renderers import ImageRenderer, Renderer, VideoRenderer, split_compose
    from oracle import cpu_ref

    vr = VideoRenderer(read_path="synthetic:64x48:5", write_path=str(tmp_path / "out.npy"), window_name="AnimalCam")
    assert isinstance(vr, Renderer) and vr.fps == 30
    vr.open()
    frames = []
    while True:
        f = vr.get_image()
        if f is None:
            break
        assert f.shape == (48, 64, 3) and f.dtype == np.uint8
        frames.append(f)
        vr.render_split_compare(f, 255 - f, left_label=None, right_label=None)  # labels are drawn on the device (tests/test_labels.py)
    vr.close()
    assert len(frames) == 5
    out = np.load(tmp_path / "out.npy")
    assert out.shape == (5, 48, 64, 3)
    assert np.array_equal(out[0], cpu_ref.make_split_frame_nolabel(frames[0], 255 - frames[0]))
    assert np.array_equal(split_compose(frames[1], frames[2], draw_seam=False)[:, 32:], frames[2][:, 32:])
    with pytest.raises(AssertionError):
        split_compose(frames[0][..., 0], frames[0])
    # image renderer round trip through Pillow
    p = str(tmp_path / "a.png")
    ImageRenderer(save_to=p).render(frames[0])
    assert np.array_equal(ImageRenderer(p).get_image(), frames[0])


WORKER = textwrap.dedent(
    """
    import os, sys, json
    sys.path.insert(0, {root!r})
    import torch.distributed as dist
    from animal_vision_amd.pipeline import StreamStats, reduce_stats, shard_indices
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    mine = shard_indices(25, rank, world)
    local = StreamStats(frames=len(mine), pixels=len(mine) * 1920 * 1080, seconds=1.0 + rank)
    dist.barrier()
    tot = reduce_stats(local, dist)
    if rank == 0:
        print(json.dumps({{"frames": tot.frames, "pixels": tot.pixels, "seconds": tot.seconds, "ranks": tot.ranks, "mine": mine}}))
    dist.destroy_process_group()
    """
)


def test_two_rank_gloo_statistics(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29573", str(script)], capture_output=True, text=True, timeout=240, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    r = json.loads(line)
    assert r["frames"] == 25 and r["pixels"] == 25 * 1920 * 1080 and r["ranks"] == 2
    assert r["seconds"] == 2.0  # max over ranks
    assert r["mine"] == list(range(0, 25, 2))


def test_bench_gpus_flag_starts_that_many_ranks(tmp_path):
    """`python bench.py --gpus 2` without a torchrun environment: the launcher itself starts two rank processes (it touches no
    GPU), they rendezvous at 127.0.0.1 and rank 0's JSON line says n_gpus = 2.  --dry-run keeps the ranks off the GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "7"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert r["n_gpus"] == 2 and r["ranks_seen"] == 2 and r["dry_run"] is True and r["steps"] == 7
    # under a torchrun-style environment the process is a rank, not a launcher
    env1 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29591")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dry-run"], capture_output=True, text=True, timeout=300, env=env1)
    assert out.returncode == 0 and json.loads(out.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_bench_ranks_fail_loudly_without_a_gpu():
    """On a box without GPUs the ranks of `bench.py --gpus 2` must fail (no CPU fallback, no silent single-rank run) and the
    launcher must pass the failure on."""
    import torch

    if torch.cuda.device_count() > 0:
        pytest.skip("this box has a GPU: the real run is the driver's")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert "rank exit codes" in out.stderr
