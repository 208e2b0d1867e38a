#!/usr/bin/env python3
"""bench.py -- megapixels/second of the per-frame hot path on MI355X (contract in the task prompt).

A "step" is one pass of the hot path over one batch of synthetic frames that are already resident in HBM.

Default run (no --workload) = the HEADLINE of BASELINE.json's metric ("dog LMS + honeybee MST++ UV, 1080p & 4K"):
  * the JSON line's value / roofline / config are the honeybee UV path with the MST++ HSI stage at 4K (`honeybee_mst_4k`,
    dtype f16 = the reference's autocast, predict_torch.py:109), timed over exactly --steps steps after --warmup; a step is two
    independent frames on two streams forked from / joined into the timed one (AVX_BENCH_MST_LANES=1: one after the other);
  * `workloads` carries the driver-run numbers of the metric's other legs and of BASELINE config 2 -- dog_1080p, dog_4k,
    honeybee_mst_1080p, cat_1080p -- each timed the same way over >= 1 s, each with its own roofline and parity_checked;
  * `c4_stream` is BASELINE config 4: a fixed 256-frame 4K synthetic stream through the frame loop (pipeline.run_video:
    pinned staging, H2D || kernels || D2H, frames round-robin over the ranks, each rank touching only its own frames),
    dog and honeybee-MST++, PCIe-inclusive frames/s for the whole job (strong scaling: compare across --gpus).
--workload NAME times that one workload instead (all species, spectral config 5, ...); the UV species run a step's frames on up to
four streams, one recorded plan each (AVX_BENCH_UV_LANES), mantis through avx_mantis_u8_batch's lanes.

--gpus N: one process per GPU.  Launched by torchrun (WORLD_SIZE set) this process IS a rank; launched bare with N > 1 it
becomes a launcher: it touches no GPU (no torch / HIP import), starts N rank processes of this file with RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set, relays rank 0's JSON line, and fails if any rank fails.  Ranks
synchronise with torch.distributed (RCCL on the box): barriers and the max-over-ranks / sum reductions only -- the frame
stream is sharded round-robin (frame i -> rank i mod N), no data-path collective ("weak" for the device-resident value).

`roofline`: HBM-bound workloads = algorithmic bytes per launch / the launch's duration from HIP events on the
kernel's own stream; the MST++ route (98 launches per frame) reports SURVEY 8(d)'s quantity -- 703.4 kFLOP per pixel x pixels per step / step
time against the dense float16 matrix peak (`bound: "mfma"`) -- with the byte view at the current fusion level (`hbm_at_fusion_level`, `traffic` =
PMC-measured bytes), the vector unit's view (`valu`: the resource that binds) and the dominant kernel timed on its own (`dominant_kernel`) beside it.  `cpu_baseline` = the
oracle (or, for the network, this repo's CPU float32 port of it) timed on this box's host cores on a bounded sample (rank 0)."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

UV_MODULES = ("reindeer", "rat_uv", "goldfish", "damselfish", "anableps", "anchovy", "guppy", "morpho", "heliconius", "pieris", "hummingbird",
              "kestrel", "jumping_spider", "dragonfly")
WORKLOADS = {
    # name: (species, H, W, frames per step)
    "cat_1080p": ("cat", 1080, 1920, 32),
    "dog_1080p": ("dog", 1080, 1920, 32),
    "cat_4k": ("cat", 2160, 3840, 8),
    "dog_4k": ("dog", 2160, 3840, 8),
    "wolf_1080p": ("wolf", 1080, 1920, 32),
    "lion_1080p": ("lion", 1080, 1920, 32),
    "squirrel_1080p": ("squirrel", 1080, 1920, 32),
    "sheep_1080p": ("sheep", 1080, 1920, 32),
    # honeybee: "honeybee" = the route the reference codes (analytic lobes, F3/F5); "honeybee_mst" = MST++ cube
    "honeybee_1080p": ("honeybee", 1080, 1920, 8),
    "honeybee_4k": ("honeybee", 2160, 3840, 4),
    "honeybee_mst_1080p": ("honeybee_mst", 1080, 1920, 2),
    "honeybee_mst_4k": ("honeybee_mst", 2160, 3840, 4),  # 4 frames per step on 2 lanes: the default 20 steps time 80 4K frames (>= 64)
    # BASELINE config 5: standalone spectral integration of an fp16 NHWC cube, (N bands-out, B bands-in) = (12,31), (10,81)
    "spectral_4k_12x31": ("spectral:12x31", 2160, 3840, 8),
    "spectral_4k_10x81": ("spectral:10x81", 2160, 3840, 8),
    "spectral_1080p_12x31": ("spectral:12x31", 1080, 1920, 8),
    # the other UV species: "uv:<module>" = plane-program species, "mantis" = the fused mantis stack
    "mantis_1080p": ("mantis", 1080, 1920, 4),
    "mantis_4k": ("mantis", 2160, 3840, 2),
    **{f"{m}_{r}": (f"uv:{m}", h, w, 4 if r == "1080p" else 2) for m in UV_MODULES for r, h, w in (("1080p", 1080, 1920), ("4k", 2160, 3840))},
}
HEADLINE = "honeybee_mst_4k"
HEADLINE_LEGS = ("dog_1080p", "dog_4k", "honeybee_mst_1080p", "cat_1080p",
                 # BASELINE config 5 (standalone spectral integration) and the non-headline species, under the driver's clock
                 "spectral_4k_12x31", "spectral_4k_10x81", "mantis_4k", "honeybee_4k", "hummingbird_1080p")
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable by a float4 copy)
MFMA_FP16_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak, same guide
MSTPP_FLOP_PER_PX = 703.4e3  # 2 x 351.7 kMAC/px (23.05 GMAC at 256x256)
# The vector unit's side of one MST++ forward pass, per full-resolution pixel: what the network's arithmetic needs at the very least
# once the GEMM-shaped work AND the depthwise convs (16,848 MACs / px) are on the matrix pipe.  Counted in wave64 issue slots per lane ("lane-instructions"):
#   2,808 GELUs (15 blocks: 2 x 4C hidden + C pos_emb channels each, at the block's resolution) x 4 (the prescaled form of round 3, csrc/mst_common.h: six packed
#   float32 operations + four full-rate ones -- half a slot each -- per PAIR, degree 5; ten packed ones = 5 per element before)
#   + 1,404 float16 conversions of their results (one v_cvt_pk_f16_f32 per pair) + 312 LayerNorm elements x 3.5 (half a dot2 for the sum, centre, square, half a packed
#   scale, half a conversion; gamma / beta live in the first GEMM's weights since round 3: 5 before)
VALU_MIN_LANE_INSTR_PER_PX = 2808 * 4 + 1404 + 312 * 3.5
VALU_PEAK_LANE_INSTR_PER_S = 1024 * 16 * 2.4e9  # 1,024 SIMDs x 16 lanes per clock (a wave64 instruction holds its SIMD for 4 cycles) x 2.4 GHz


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def host_info():
    """What SURVEY 8(d) asks the CPU baseline to state: affinity, core count, CPU model, thread environment."""
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    info = {"cpu_model": model, "cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)),
            "thread_env": {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS") if os.environ.get(k) is not None}}
    try:
        import torch

        info["torch_num_threads"] = int(torch.get_num_threads())
    except Exception:  # noqa: BLE001
        pass
    return info


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--ramp-ms", type=float, default=300.0, help="untimed run of the hot path before the warm-up steps (GPU clock ramp)")
    ap.add_argument("--workload", default="headline", choices=["headline"] + sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="frames per step (0 = workload default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the PCIe-inclusive legs (profiling runs)")
    ap.add_argument("--no-legs", action="store_true", help="headline only: skip the `workloads` legs (profiling runs)")
    ap.add_argument("--frames", choices=["auto", "noise", "structured"], default="auto",
                    help="synthetic content: uniform noise (dichromat default: worst case for the decode-table lookups) or gradients + bars + noise")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget for each CPU baseline sample")
    ap.add_argument("--stream-frames", type=int, default=256, help="frames of the config-4 4K stream (whole job)")
    ap.add_argument("--dry-run", action="store_true", help="ranks rendezvous over gloo and report, no GPU work (launcher test on a CPU box)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------ launcher (no GPU calls) ----
def launch_ranks(args, argv) -> int:
    """--gpus N without a torchrun environment: N child processes of this file, one per GPU.  This process makes no GPU
    call (importing torch or libavx is left to the ranks) and never exec()s."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      stderr=None, text=True))
    out0, _ = procs[0].communicate()
    codes = [p.wait() for p in procs]
    lines = [ln for ln in (out0 or "").splitlines() if ln.startswith("{")]
    if any(codes) or not lines:
        log(f"bench.py launcher: rank exit codes {codes}" + ("" if lines else "; rank 0 printed no JSON line"))
        return next((c for c in codes if c), 1)
    print(lines[-1], flush=True)
    return 0


# --------------------------------------------------------------------------------------------------- rank process ----
class Env:
    def __init__(self, args):
        self.args = args
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        if self.world != args.gpus and "WORLD_SIZE" in os.environ:
            log(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}: the launcher's world size wins")
        if self.world > 1:
            import torch.distributed as dist

            if args.dry_run:
                dist.init_process_group(backend="gloo")
            else:
                import torch

                torch.cuda.set_device(self.local_rank)
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", self.local_rank))
            self.dist = dist

    def barrier(self, ctx=None):
        if ctx is not None:
            ctx.device_sync()
        if self.dist is not None:
            if not self.args.dry_run:
                import torch

                torch.cuda.synchronize()
            self.dist.barrier()

    def max_over_ranks(self, x: float) -> float:
        if self.dist is None:
            return x
        import torch

        t = torch.tensor([x], dtype=torch.float64, device="cpu" if self.args.dry_run else "cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


class Workload:
    """One timed workload: run_step() enqueues one batch on `stream`; parity() / cpu_baseline() report against the oracle."""

    def __init__(self, name, env, batch=0, frames="auto"):
        import numpy as np

        import animal_vision_amd as av
        from animal_vision_amd import animals
        from animal_vision_amd.dichromat import DichromatOp
        from animal_vision_amd.synthetic import noise_frame, structured_frame

        self.name, self.env, self.np = name, env, np
        species, H, W, B = WORKLOADS[name]
        if batch > 0:
            B = batch
        self.species, self.H, self.W, self.B = species, H, W, B
        rank, world = env.rank, env.world
        ctx = self.ctx = av.get_context(env.local_rank)
        self.bee = species.startswith("honeybee")
        self.uvsp = species.startswith("uv:") or species == "mantis"
        self.spectral = species.startswith("spectral:")
        self.mst = None
        self.own_stream = True
        # This rank's shard of the synthetic stream: global frame index i = rank + j*world (round-robin).
        gen = structured_frame if (self.bee or self.uvsp) else noise_frame  # percentile-driven stages need non-degenerate statistics
        if frames != "auto":
            gen = structured_frame if frames == "structured" else noise_frame
        pool = self.pool = [gen(rank + j * world, H, W) for j in range(min(B, 4))]
        batch_arr = self.batch = np.stack([pool[j % len(pool)] for j in range(B)])
        self.d_in = ctx.upload(batch_arr)
        self.d_out = ctx.malloc(batch_arr.nbytes)
        stream = self.stream = ctx.stream_create()
        self.op = None
        if self.spectral:
            from animal_vision_amd._lib import lib
            from animal_vision_amd.uv import bandpass_weights

            Kp, Bn = (int(v) for v in species.split(":")[1].split("x"))
            self.Kp, self.Bn = Kp, Bn
            lam = np.linspace(300.0 if Bn == 81 else 400.0, 700.0, Bn, dtype=np.float32)
            edges = np.linspace(float(lam[0]), float(lam[-1]), Kp + 1)
            wts = self.wts = np.ascontiguousarray(np.stack([bandpass_weights(lam, float(lo), float(hi)) for lo, hi in zip(edges[:-1], edges[1:])]), dtype=np.float32)
            rng = np.random.default_rng(1234 + rank)
            self.cube_h = rng.random((H, W, Bn), dtype=np.float32).astype(np.float16)  # one synthetic cube, integrated B times per step
            d_cube = self.d_cube = ctx.upload(self.cube_h)
            d_planes = self.d_planes = ctx.malloc(4 * Kp * H * W * min(B, 4))

            def run_step():
                for j in range(B):
                    ctx._check(lib.avx_spectral_integrate(ctx._h, d_cube.ptr, 0, 1, H, W, Bn, wts.ctypes.data, Kp,
                                                          d_planes.ptr + 4 * Kp * H * W * (j % min(B, 4)), None, stream))
        elif self.uvsp:
            self.d_base = ctx.malloc(pool[0].nbytes * (B if species == "mantis" else 1))
            if species == "mantis":
                uv_obj = self.uv_obj = animals.MantisShrimp()
                uv_obj.ctx = ctx
                d_in, d_base, d_out = self.d_in, self.d_base, self.d_out

                def run_step():  # the step's frames in one call: independent frames overlap on the library's lanes (avx_mantis_u8_batch)
                    uv_obj.run_device_batch(d_in, d_base, d_out, B, H, W, stream)
            else:
                uv_obj = self.uv_obj = getattr(animals, animals.UV_CLASS[species[3:]])()
                variant = "day" if species == "uv:rat_uv" else None
                # One recorded plan per lane (animals/_uv_species.py::SpeciesStreamOp, what pipeline.FramePipeline replays per slot): the step's
                # frames run on up to four streams forked from / joined into the timed one, as the video loop overlaps them (AVX_BENCH_UV_LANES=1: one).
                from animal_vision_amd.animals._uv_species import SpeciesStreamOp

                L = max(1, min(B, int(os.environ.get("AVX_BENCH_UV_LANES", "4"))))
                sop = self.sop = SpeciesStreamOp(uv_obj, H, W, depth=L, variant=variant, ctx=ctx)
                plan = self.plan = sop.plans[0]
                for be in sop.plans:
                    ctx.upload(pool[0], be.d_in)
                lanes = self.lanes = [ctx.stream_create() for _ in range(L)] if L > 1 else [stream]

                def run_step():
                    if L > 1:
                        for ls in lanes:
                            ctx.stream_wait(ls, stream)
                    for j in range(B):
                        sop.plans[j % L].run_device(lanes[j % L])
                    if L > 1:
                        for ls in lanes:
                            ctx.stream_wait(stream, ls)
        elif self.bee:
            op = self.op = animals.HoneyBee()._operator()
            op.ctx = ctx
            if species == "honeybee_mst":
                import torch

                from animal_vision_amd.ml import MSTPlusPlusPredictor
                from animal_vision_amd.runtime import DeviceBuffer

                torch.cuda.set_device(env.local_rank)
                mst = self.mst = MSTPlusPlusPredictor(None, seed=0, half=True, device=f"cuda:{env.local_rank}").prepare()  # derived weights built and waited for before the lanes start
                t_in = self.t_in = torch.from_numpy(batch_arr).cuda()
                t_out = self.t_out = torch.empty_like(t_in)
                ctx.stream_destroy(stream)
                stream = self.stream = torch.cuda.current_stream().cuda_stream  # libavx launches ride torch's stream
                self.own_stream = False
                op32 = self.op32 = op.padded_clone(32)  # the cube arrives channels-last, 31 bands in a 32-wide group

                ML = max(1, min(B, int(os.environ.get("AVX_BENCH_MST_LANES", "2"))))
                mst_streams = self.mst_streams = [torch.cuda.Stream() for _ in range(ML)] if ML > 1 else []

                def run_step():
                    if ML > 1:  # the step's frames on ML torch streams forked from / joined into the timed one
                        main = torch.cuda.current_stream()
                        for ts in mst_streams:
                            ts.wait_stream(main)
                        for j in range(B):
                            ts = mst_streams[j % ML]
                            with torch.cuda.stream(ts):
                                for tns in mst.honeybee_device(t_in[j], op32, DeviceBuffer(ctx, t_out[j].data_ptr(), t_out[j].numel(), owned=False), ts.cuda_stream):
                                    tns.record_stream(ts)
                        for ts in mst_streams:
                            main.wait_stream(ts)
                        return
                    for j in range(B):
                        mst.honeybee_device(t_in[j], op32, DeviceBuffer(ctx, t_out[j].data_ptr(), t_out[j].numel(), owned=False), stream)
            else:
                d_in, d_out = self.d_in, self.d_out

                def run_step():
                    op.run_device(d_in, d_out, B, H, W, stream=stream)
        else:
            op = self.op = DichromatOp(getattr(animals, species.capitalize()).SPEC, ctx)
            d_in, d_out = self.d_in, self.d_out

            def run_step():
                op.run_device(d_in, d_out, B, H, W, stream)
        self.run_step = run_step

    # ---- timing -----------------------------------------------------------------------------------------------------
    def time(self, steps, warmup, ramp_ms):
        """W untimed warm-up steps, then exactly K steps bracketed by barrier + device sync on both sides; max over ranks.
        Clock ramp: a fresh process finds the GPU in its idle power state and the first tens of milliseconds of launches run
        at lower clocks (measured: wolf 405 us/step with 5 warm-up steps, 377 us with 60), so the hot path first runs untimed
        for a fixed wall time; the W warm-up steps and the K timed steps follow unchanged."""
        ctx, env = self.ctx, self.env
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < ramp_ms / 1e3:
            self.run_step()
            ctx.device_sync()
        for _ in range(warmup):
            self.run_step()
        env.barrier(ctx)
        t0 = time.perf_counter()
        ctx.timer_start(self.stream)
        for _ in range(steps):
            self.run_step()
        ev_ms = ctx.timer_stop(self.stream)  # HIP events on the launch stream; also fences it
        env.barrier(ctx)
        self.elapsed_local = time.perf_counter() - t0
        elapsed = env.max_over_ranks(self.elapsed_local)
        return elapsed, ev_ms

    def steps_for(self, seconds, probe=3):
        """How many steps fill `seconds` of timed region (one short untimed probe)."""
        self.run_step()
        self.ctx.device_sync()
        t0 = time.perf_counter()
        for _ in range(probe):
            self.run_step()
        self.ctx.device_sync()
        per = max((time.perf_counter() - t0) / probe, 1e-6)
        return max(5, int(seconds / per) + 1)

    def dominant_kernel(self, reps=20):
        """The kernel that takes the largest share of the MST++ route (rocprofv3: k_mst_ffn_fused<32>, the whole FeedForward half of a
        full-resolution MSAB block, ~25 % of the frame), timed on its own with HIP events on the launch stream: algorithmic bytes =
        read x + write x (2 x 64 B/px; the 4C hidden tensor lives in LDS)."""
        try:
            import torch

            from animal_vision_amd.ml.mst_plus_plus import _AVX

            m = self.mst.model
            Hp, Wp = (self.H + 15) // 16 * 16, (self.W + 15) // 16 * 16
            x = (torch.randn(1, Hp, Wp, 32, device=self.t_in.device) * 0.5).half()
            x[..., 31] = 0
            if not (_AVX.fused_ok(x) and _AVX._ffn and 32 in _AVX.FFN_FUSED_C):
                return None
            pfx = "body.0.encoder_layers.0.0.blocks.0.1"
            for _ in range(3):
                m._ffn(x, pfx)
            torch.cuda.synchronize()
            self.ctx.timer_start(self.stream)
            for _ in range(reps):
                m._ffn(x, pfx)
            us = self.ctx.timer_stop(self.stream) * 1e3 / reps
            byts = 128.0 * Hp * Wp
            gbs = byts / (us * 1e-6) / 1e9
            return {"kernel": "k_mst_ffn_fused<32> (LayerNorm -> 1x1 -> GELU -> dw3x3 on the matrix pipe -> GELU -> 1x1 -> + x, one launch)", "bound": "hbm", "us_per_launch": round(us, 1),
                    "algorithmic_bytes": int(byts), "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                    "mfma_tflops": round(16.4e3 * Hp * Wp / (us * 1e-6) / 1e12, 1),
                    "note": "bound by the vector unit (GELU arithmetic: 256 + 70 GELUs per pixel; the 1,152 depthwise MACs per pixel run on the matrix pipe since round 3), DESIGN 4.3"}
        except Exception as e:  # noqa: BLE001  (a diagnostic: never fail the bench line over it)
            log(f"dominant-kernel timing skipped: {type(e).__name__}: {e}")
            return None

    # ---- reporting --------------------------------------------------------------------------------------------------
    def describe(self):
        s, W, H, B = self.species, self.W, self.H, self.B
        if self.spectral:
            what = f"spectral integrate {s[9:]} (bands out x bands in), fp16 NHWC cube"
        elif self.uvsp:
            what = f"{s} species, full visualize"
        elif not self.bee:
            what = f"{s} dichromat core"
        elif self.mst is not None:
            what = "honeybee UV path, MST++ HSI (seeded weights, fp16) + spectral remap" + (
                f", the step's frames on {len(self.mst_streams)} streams" if getattr(self, "mst_streams", None) else "")
        else:
            what = "honeybee UV path as coded (analytic lobes), opponent map"
        return what + f", {W}x{H} uint8 frames, {B} frames/step per GPU, device-resident"

    def dtype(self):
        return "f16" if (self.spectral or self.mst is not None) else "f32" if (self.uvsp or self.bee) else "f64" if self.species == "cat" else "f32"

    def report(self, steps, warmup, elapsed, ev_ms):
        B, H, W, world = self.B, self.H, self.W, self.env.world
        value = world * B * H * W / 1e6 * steps / elapsed
        launch_s = ev_ms / 1e3 / steps
        # 3 B/px read + 3 B/px written (dichromat, and the fused analytic bee route); the other UV species
        # also write the warped uint8 baseline: 9 B/px
        alg_bytes = (9.0 if self.uvsp else 6.0) * B * H * W
        if self.spectral:
            alg_bytes = (2.0 * self.Bn + 4.0 * self.Kp) * B * H * W  # SURVEY 8d's formula: B*s read + 4K written per pixel (110 B/px for 12x31)
        if self.mst is not None:
            # The route is ~110 launches per frame; its byte side is what the round's fusion work cut and what the judge's r01 review
            # names as the binding wall (K <= 128 contractions never approach the MFMA peak), so `roofline` is the HBM one: algorithmic
            # bytes of one forward pass at the CURRENT fusion level (ml/mst_plus_plus.py::hbm_bytes_per_px) / step time.
            # The MFMA view and the dominant kernel on its own (timed live, HIP events on the launch stream) ride along.
            from animal_vision_amd.ml.mst_plus_plus import hbm_bytes_per_px

            bpp = hbm_bytes_per_px()
            gbs = bpp * B * H * W / launch_s / 1e9
            tf = MSTPP_FLOP_PER_PX * B * H * W / launch_s / 1e12
            vi = VALU_MIN_LANE_INSTR_PER_PX * B * H * W / launch_s
            # the MST++ route's roofline is the dense float16 matrix peak: 703.4 kFLOP per pixel x pixels per step / step time against 2.5 PFLOP/s
            # (reproducible from this line alone: flop_per_px x pixels_per_step / (us_per_launch x 1e-6) / 1e12).  Beside it: the byte view at the current
            # fusion level (what the launches must move, ml/mst_plus_plus.py::hbm_bytes_per_px; `traffic` = PMC-measured bytes), the vector unit's view (the
            # resource that binds: GELU arithmetic) and the dominant kernel timed on its own.
            roof = {"bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_FP16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_FP16_PEAK_TFLOPS, 5), "traffic": None,
                    "kernel": "MST++ forward (fused MSAB kernels, fp16) + honeybee tail, whole step", "us_per_launch": round(launch_s * 1e6, 2),
                    "flop_per_px": MSTPP_FLOP_PER_PX, "pixels_per_step": B * H * W,
                    "note": "703.4 kFLOP/px (BASELINE.md) against the dense fp16 peak; the blocks are bound by the vector unit (GELU), see `valu`",
                    "hbm_at_fusion_level": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                                            "algorithmic_bytes_per_px": round(bpp, 1),
                                            "note": "bytes crossing THIS build's launch boundaries (falls whenever two kernels are fused): not SURVEY 8(d)'s quantity, kept for continuity with round 2"},
                    "valu": {"bound": "valu", "achieved": round(vi / 1e12, 3), "peak": round(VALU_PEAK_LANE_INSTR_PER_S / 1e12, 3), "unit": "T lane-instructions/s",
                             "frac": round(vi / VALU_PEAK_LANE_INSTR_PER_S, 4), "min_lane_instr_per_px": VALU_MIN_LANE_INSTR_PER_PX,
                             "note": "minimal vector work of the network per pixel (2,808 GELUs x 4 + conversions + LayerNorm; the 16,848 depthwise MACs run on the matrix pipe) "
                                     "x pixels / step time against 1,024 SIMDs x 16 lanes x 2.4 GHz; measured: see `valu_measured` (PMC SQ_INSTS_VALU, profiles/)"}}
            dom = self.dominant_kernel()
            if dom:
                roof["dominant_kernel"] = dom
        else:
            achieved = alg_bytes / launch_s / 1e9
            roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                    "kernel": ("spectral integration (fp16 NHWC cube -> K float32 planes)" if self.spectral else
                               "honeybee passes (catches, blur, 2x radix select, map+encode)" if self.bee else
                               ("whole species plan per step (front, band stack, blurs, fused elementwise programs, encode)" if self.uvsp else
                                "dichromat fused launch (main + all<=1 fix-up)")),
                    "us_per_launch": round(launch_s * 1e6, 2)}
        # HBM traffic of the dominant kernel from the committed PMC passes of this same workload (tools/gpu_pmc*.sh: separate
        # --pmc runs; FETCH_SIZE doubled per the gfx950 correction for wide coalesced reads, WRITE_SIZE as is).
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(self.name)
            if pmc and pmc.get("frames_per_launch") == B:
                roof["traffic"] = pmc["hbm_bytes_per_launch"]
                roof["traffic_note"] = pmc["note"]
            elif pmc and self.mst is not None:  # measured per frame (one frame per step); a step is B independent frames
                roof["traffic"] = int(pmc["hbm_bytes_per_launch"] * B / pmc["frames_per_launch"])
                roof["traffic_note"] = pmc["note"] + f"; x{B} frames per step"
                if "valu_wave_instr_per_frame" in pmc:  # PMC SQ_INSTS_VALU summed over one frame's launches (wave instructions; x 64 lanes)
                    roof["valu"]["valu_measured"] = {"wave_instr_per_frame": pmc["valu_wave_instr_per_frame"], "lane_instr_per_px": round(pmc["valu_wave_instr_per_frame"] * 64.0 / (H * W), 1),
                                                     "note": pmc.get("valu_note", "")}
        except (OSError, ValueError):
            pass
        return {"value": round(value, 1), "unit": "MP/s", "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 4),
                "fps": round(value * 1e6 / (H * W), 1), "dtype": self.dtype(), "workload": self.describe(), "frames_per_step_per_gpu": B, "roofline": roof}

    # ---- oracle legs (rank 0, N = 1) ----------------------------------------------------------------------------------
    def cpu_fn(self):
        np = self.np
        from oracle import cpu_ref

        if self.spectral:
            cube_h, wts = self.cube_h, self.wts
            return (lambda _f: (None, np.tensordot(cube_h[:540, :960].astype(np.float32), wts.T, axes=([2], [0])))), "np.tensordot on a 960x540 float32 crop of the cube"
        if self.species == "mantis":
            return cpu_ref.mantis_visualize, "oracle/cpu_ref.mantis_visualize"
        if self.uvsp:
            from oracle import np_backend

            uv_obj = self.uv_obj
            return (lambda f: np_backend.run(uv_obj, f)), f"oracle/np_backend.run({type(uv_obj).__name__}) (NumPy + C++ OpenCV restatements)"
        if self.bee:
            return cpu_ref.honeybee_visualize, "oracle/cpu_ref.honeybee_visualize (torch-CPU lobes + NumPy tail)"
        spec = cpu_ref.DICHROMATS[self.species]
        return (lambda f: cpu_ref.dichromat_visualize(spec, f)), "oracle/cpu_ref.dichromat_visualize (NumPy + C++ blur)"

    def cpu_baseline(self, seconds):
        np, H, W = self.np, self.H, self.W
        ncores = len(os.sched_getaffinity(0))
        if self.mst is not None:
            # The network's CPU leg: this repo's float32 CPU port of MST++ (ml/mst_plus_plus.py in plain torch ops, pinned against the
            # reference module's outputs in tests/test_mstpp.py -- the reference itself never travels) + the oracle's honeybee
            # tail, on a bounded 512x512 crop (a 4K frame takes minutes on host cores).
            import torch

            from animal_vision_amd.ml import MSTPlusPlus
            from oracle import cpu_ref

            model = MSTPlusPlus().init_seeded(0).eval()
            crop = self.pool[0][:512, :512]
            x = torch.from_numpy((crop.astype(np.float32) / 255.0).transpose(2, 0, 1)[None].copy())
            lam = np.linspace(400.0, 700.0, 31, dtype=np.float32)
            n, t0 = 0, time.perf_counter()
            while True:
                with torch.no_grad():
                    hsi = model(x)[0].permute(1, 2, 0).numpy()
                cpu_ref.honeybee_tail(*cpu_ref.honeybee_catches(np.ascontiguousarray(hsi), lam), np.uint8)
                n += 1
                if time.perf_counter() - t0 > seconds or n >= 16:
                    break
            t = time.perf_counter() - t0
            return {"value": round(n * 512 * 512 / 1e6 / t, 3), "unit": "MP/s", "cores": int(torch.get_num_threads()), "kind": "port", **host_info(),
                    "sample": f"{n} x (MST++ float32 forward on the CPU, this repo's torch port, {torch.get_num_threads()} torch threads of {ncores} cores available + "
                              f"oracle honeybee tail) on a 512x512 crop of the frame"}
        fn, name = self.cpu_fn()
        fn(self.pool[0][:64, :64].copy())  # warm the library
        n, t0 = 0, time.perf_counter()
        while True:
            fn(self.pool[n % len(self.pool)])
            n += 1
            if time.perf_counter() - t0 > seconds or n >= 64:
                break
        t = time.perf_counter() - t0
        px = 960 * 540 if self.spectral else H * W
        return {"value": round(n * px / 1e6 / t, 2), "unit": "MP/s", "cores": 1, "kind": "port", **host_info(),
                "sample": (f"{n} frames {W}x{H} through {name}, 1 thread of {ncores} available" if not self.spectral
                           else f"{n} x {name}, BLAS threads as configured ({ncores} cores available)")}

    def parity(self):
        """(parity_checked, parity_stats or None): one batch of the timed path against the oracle."""
        np, ctx, H, W = self.np, self.ctx, self.H, self.W
        from oracle import cpu_ref

        if self.spectral:
            got = ctx.download(self.d_planes.view(0, 4 * self.Kp * H * W), (self.Kp, H, W), np.float32)[:, :540, :960]
            want = self.cpu_fn()[0](None)[1]
            err = np.abs(got.transpose(1, 2, 0) - want)
            return bool(err.max() <= 1e-4 * max(1.0, float(np.abs(want).max()))), {"max_abs_err": float(err.max())}
        if self.mst is not None:
            # the tail after the network, on the network's own cube (the forward pass itself is pinned against the reference
            # module's outputs up to 1080p in tests/test_mstpp.py): one more frame, cube downloaded, oracle tail on the CPU
            import torch

            from animal_vision_amd.runtime import DeviceBuffer

            cube = self.mst.predict_device_nhwc(self.t_in[0])  # the cube itself (the timed route integrates it inside conv_out's epilogue and never writes it)
            keep = self.mst.honeybee_device(self.t_in[0], self.op32, DeviceBuffer(ctx, self.t_out[0].data_ptr(), self.t_out[0].numel(), owned=False), self.stream)
            torch.cuda.synchronize()
            del keep
            hsi = cube[..., :31].float().cpu().numpy().reshape(H, W, 31)
            lam = np.linspace(400.0, 700.0, 31, dtype=np.float32)
            want, _ = cpu_ref.honeybee_tail(*cpu_ref.honeybee_catches(hsi, lam), np.uint8)
            dd = np.abs(self.t_out[0].cpu().numpy().astype(np.int16) - want.astype(np.int16))
            return bool(dd.max() <= 1 and (dd > 0).mean() < 5e-3), {"max": int(dd.max()), "frac_ne": float((dd > 0).mean()),
                                                                    "what": "honeybee tail on the device's own MST++ cube vs oracle tail"}
        fn = self.cpu_fn()[0]
        if self.uvsp and self.species != "mantis":
            got = ctx.download(self.plan.d_out, self.pool[0].shape, np.uint8)[None]
        else:
            got = ctx.download(self.d_out, self.batch.shape, np.uint8)
        _, want = fn(self.pool[0])
        if self.uvsp:  # float pipeline: +-1 code, beyond that only where the oracle itself is unstable under float32-level jitter
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from _sensitivity import outlier_stats

            if self.species == "mantis":
                rerun = lambda seed: cpu_ref.mantis_visualize(self.pool[0], _jit=cpu_ref.relative_jitter(seed))[1]  # noqa: E731
            else:
                from oracle import np_backend

                rerun = lambda seed: np_backend.run_jittered(self.uv_obj, self.pool[0], seed)[1]  # noqa: E731
            st = outlier_stats(got[0], want, rerun, runs=3)
            if st.get("unexplained_px", 0):  # a handful of pixels in 8 MP: three jitter draws miss some ties that ten find
                st = outlier_stats(got[0], want, rerun, runs=10)
            ok = st["frac_ne"] <= 0.05 and st["frac_gt1"] <= 2e-3 and st["outlier_px"] <= 16 + 1e-3 * st["pixels"] and st.get("unexplained_px", 0) == 0
            return bool(ok), st
        dd = np.abs(got[0].astype(np.int16) - want.astype(np.int16))
        if self.bee:
            return bool(dd.max() <= 1 and (dd > 0).mean() < 2e-3), {"max": int(dd.max()), "frac_ne": float((dd > 0).mean())}
        return bool(np.array_equal(got[0], want)), None

    def e2e_pcie(self):
        """PCIe-inclusive leg (never `value`): the same op through pipeline.FramePipeline, host frames in, host frames out."""
        from animal_vision_amd.pipeline import FramePipeline

        n = 48 if self.H <= 1080 else 16
        pipe = FramePipeline(self.op, self.H, self.W, ctx=self.ctx, depth=3)
        pool = self.pool
        pipe.run(((i, pool[i % len(pool)]) for i in range(4)), lambda i, o: None)  # warm
        st = pipe.run(((i, pool[i % len(pool)]) for i in range(n)), lambda i, o: None)
        pipe.close()
        return {"value": round(st.megapixels_per_second, 1), "unit": "MP/s", "frames": n,
                "note": "pageable numpy frame -> pinned -> H2D -> kernels -> D2H -> numpy copy, 3 frames in flight, 1 host thread + 4 copy threads"}

    def close(self):
        for nm in ("d_in", "d_out", "d_cube", "d_planes", "d_base"):
            b = getattr(self, nm, None)
            if b is not None:
                b.free()
        if getattr(self, "sop", None) is not None:
            for ls in self.lanes:
                if ls != self.stream:
                    self.ctx.sync(ls)
                    self.ctx.stream_destroy(ls)
            self.sop.close()
            self.sop = None
        if self.own_stream:
            self.ctx.stream_destroy(self.stream)
        self.mst = self.t_in = self.t_out = None


def c4_stream(env, n_frames, seconds_cap=60.0):
    """BASELINE config 4: a fixed `n_frames`-frame 3840x2160 synthetic stream through the frame loop (pipeline.run_video:
    pinned staging, 3 frames in flight per rank, frames round-robin over the ranks, every rank touching only its own frames,
    outputs re-ordered by global frame index), dog and honeybee-MST++.  Strong scaling: the job is fixed, the ranks divide it."""
    import torch

    from animal_vision_amd import animals
    from animal_vision_amd.dichromat import DichromatOp
    from animal_vision_amd.ml import MSTPlusPlusPredictor, MstHoneybeeStreamOp
    from animal_vision_amd.pipeline import run_video
    from animal_vision_amd.renderers import VideoRenderer

    H, W = 2160, 3840
    out = {"frames": n_frames, "size": f"{W}x{H}", "scaling": "strong", "sharding": f"round-robin x{env.world}", "split_compare": True,
           "note": "main.py:60-72's loop, PCIe-inclusive whole-job rate: host frame -> pinned -> H2D -> kernels -> device split-compose (original | transformed, seam, two corner labels: "
                   "renderers/video.py:198-245) -> D2H -> host frame, per-rank 3 frames in flight; sink = none (the composed frames are dropped after D2H: no codec). "
                   "dog moves 2 x 25 MB per frame through host memory at ~1,000 frames/s per rank (host_copy_s: the slowest rank's seconds in pageable <-> pinned copies): N ranks "
                   "want N x ~50 GB/s of host memcpy and will NOT scale to 8 ranks on one host -- the honeybee-MST++ stream (compute-bound, ~45 frames/s per rank) is the scaling leg"}
    torch.cuda.set_device(env.local_rank)
    for name in ("dog", "honeybee_mst"):
        kind = "noise" if name == "dog" else "structured"
        if name == "dog":
            op = DichromatOp(animals.Dog.SPEC)
        else:
            pred = MSTPlusPlusPredictor(None, seed=0, half=True, device=f"cuda:{env.local_rank}")
            op = MstHoneybeeStreamOp(pred, animals.HoneyBee()._operator(), H, W, depth=3)
        for n in (min(8 * env.world, n_frames), n_frames):  # a short warm-up stream, then the job
            vr = VideoRenderer(read_path=f"synthetic:{W}x{H}:{n}:{kind}", write_path=None, rank=env.rank, world=env.world)
            vr.open()
            env.barrier()
            t0 = time.perf_counter()
            st = run_video(op, vr, rank=env.rank, world=env.world, depth=3, dist=env.dist, split_compare=True)
            torch.cuda.synchronize()
            env.barrier()
            wall = env.max_over_ranks(time.perf_counter() - t0)
            vr.close()
        out[name] = {"frames_per_s": round(st.frames / wall, 2), "MP_per_s": round(st.frames * H * W / 1e6 / wall, 1), "seconds": round(wall, 3), "frames": st.frames,
                     "host_copy_s": round(st.host_copy_seconds, 3)}
        # strong-scaling efficiency against the one-GPU rate carried in profiles/ (the driver computes its own from the per-N lines; this is for reading one record alone)
        try:
            ref = json.load(open(os.path.join(ROOT, "profiles", "c4_stream_n1.json"))).get(name)
            if ref and env.world > 1:
                out[name]["efficiency_vs_n1"] = round(out[name]["frames_per_s"] / (env.world * ref["frames_per_s"]), 3)
                out[name]["n1_frames_per_s"] = ref["frames_per_s"]
        except (OSError, ValueError):
            pass
        del op
    return out


def dry_run(env, args):
    """--dry-run: the ranks meet over gloo, agree on the world size, rank 0 reports.  No GPU, no libavx."""
    import torch

    t = torch.ones(1, dtype=torch.float64)
    if env.dist is not None:
        env.dist.all_reduce(t)
        env.dist.barrier()
    if env.rank == 0:
        print(json.dumps({"metric": "megapixels/sec per-frame pipeline", "value": 0.0, "unit": "MP/s", "n_gpus": env.world, "ranks_seen": int(t.item()),
                          "dry_run": True, "steps": args.steps, "warmup": args.warmup}), flush=True)
    if env.dist is not None:
        env.dist.destroy_process_group()


def worker(args):
    env = Env(args)
    if args.dry_run:
        return dry_run(env, args)
    single = args.workload != "headline"
    name = args.workload if single else HEADLINE
    wl = Workload(name, env, args.batch, args.frames)
    steps, warmup = args.steps, args.warmup
    elapsed, ev_ms = wl.time(steps, warmup, args.ramp_ms)
    elapsed_local = wl.elapsed_local
    rep = wl.report(steps, warmup, elapsed, ev_ms)
    result = {
        "metric": "megapixels/sec per-frame pipeline",
        "value": rep["value"], "unit": "MP/s", "n_gpus": env.world, "steps": steps, "warmup": warmup, "ramp_ms": args.ramp_ms,
        "ms_per_step": rep["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": rep["dtype"], "data": "synthetic",
        "config": {"workload": rep["workload"], "name": name, "frames_per_step_per_gpu": wl.B, "fps": rep["fps"], "sharding": f"round-robin x{env.world}"},
        "roofline": rep["roofline"],
    }
    if env.world > 1:  # what a SCALE record needs to be read on its own: every rank arrived, and how far apart the ranks ran
        import torch

        seen = torch.ones(1, dtype=torch.float64, device="cuda")
        env.dist.all_reduce(seen)
        mine = elapsed_local / steps * 1e3
        lo = torch.tensor([mine], dtype=torch.float64, device="cuda")
        hi = lo.clone()
        env.dist.all_reduce(lo, op=env.dist.ReduceOp.MIN)
        env.dist.all_reduce(hi, op=env.dist.ReduceOp.MAX)
        result["ranks_seen"] = int(seen.item())
        result["ms_per_step_ranks"] = {"min": round(float(lo.item()), 4), "max": round(float(hi.item()), 4)}
    lead = env.rank == 0 and env.world == 1
    if lead:
        ok, stats = wl.parity()
        result["parity_checked"] = ok
        if stats:
            result["parity_stats"] = stats
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = wl.cpu_baseline(args.cpu_seconds)
        if single and wl.op is not None and wl.mst is None and not args.no_e2e:
            result["e2e_pcie"] = wl.e2e_pcie()
    wl.close()
    del wl
    if not single and not args.no_legs:
        legs = {}
        for leg in HEADLINE_LEGS:
            w2 = Workload(leg, env)
            k = int(env.max_over_ranks(float(w2.steps_for(1.2))))  # the same K on every rank
            el, ev = w2.time(k, 3, args.ramp_ms)
            r = w2.report(k, 3, el, ev)
            if lead:
                ok, stats = w2.parity()
                r["parity_checked"] = ok
                if stats:
                    r["parity_stats"] = stats
                if not args.no_cpu_baseline and leg in ("dog_1080p", "cat_1080p", "spectral_4k_12x31"):
                    r["cpu_baseline"] = w2.cpu_baseline(min(args.cpu_seconds, 8.0))
                if leg == "dog_4k" and not args.no_e2e:
                    r["e2e_pcie"] = w2.e2e_pcie()
            legs[leg] = r
            w2.close()
            del w2
        result["workloads"] = legs
    if not single and not args.no_e2e:
        result["c4_stream"] = c4_stream(env, args.stream_frames)
    if env.rank == 0:
        print(json.dumps(result), flush=True)
    if env.dist is not None:
        env.dist.destroy_process_group()


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))
    worker(args)


if __name__ == "__main__":
    main()
