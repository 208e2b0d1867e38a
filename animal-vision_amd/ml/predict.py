"""Predictor harness around MSTPlusPlus: the semantics of the reference's
ml/MST_plus_plus/predict_code/predict_torch.py (predict_rgb_to_hsi_torch :249-310): float01 ->
reflect-pad H,W to a multiple of `stride` split on both sides (:171-183) -> NCHW -> model under fp16
-> crop -> HxWx31 float32.  Full-frame only: MI355X has 288 GB of HBM, and MST++'s spectral attention
contracts over every pixel of the frame, so the reference's OOM tile fallback (:199-235) is neither
needed nor equivalent."""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np


def to_float01(img: np.ndarray) -> np.ndarray:
    """predict_torch.py:12-19."""
    if np.issubdtype(img.dtype, np.integer):
        return img.astype(np.float32) / 255.0
    x = img.astype(np.float32)
    if x.max() > 1.001:
        x = np.clip(x / 255.0, 0.0, 1.0)
    return x


def pad_amounts(H: int, W: int, mult: int) -> Tuple[int, int, int, int]:
    """(top, bottom, left, right) of predict_torch.py:22-34: total padding split floor/ceil."""
    if mult is None or mult <= 0:
        return 0, 0, 0, 0
    Hn, Wn = ((H + mult - 1) // mult) * mult, ((W + mult - 1) // mult) * mult
    py, px = Hn - H, Wn - W
    return py // 2, py - py // 2, px // 2, px - px // 2


def pad_to_multiple_reflect(x: np.ndarray, mult: int):
    pads = pad_amounts(x.shape[0], x.shape[1], mult)
    if pads == (0, 0, 0, 0):
        return x, pads
    t, b, l, r = pads
    return np.pad(x, ((t, b), (l, r), (0, 0)), mode="reflect"), pads


def crop_pads(x: np.ndarray, pads) -> np.ndarray:
    """predict_torch.py:37-40."""
    t, b, l, r = pads
    H, W = x.shape[:2]
    return x[t : H - b if b else H, l : W - r if r else W, :]


class MSTPlusPlusPredictor:
    """RGB frame -> 31-band cube on the GPU.  weights: None (seeded random init: no checkpoint ships with the
    reference and there is no network), a path to a local .pth, or a state_dict."""

    def __init__(self, weights=None, *, seed: int = 0, half: bool = True, stride: int = 16, device: Optional[str] = None):
        import torch

        from .mst_plus_plus import MSTPlusPlus

        self.torch = torch
        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"
        self.device = torch.device(device)
        self.half = bool(half) and self.device.type == "cuda"
        self.stride = stride
        model = MSTPlusPlus()
        if weights is None:
            model.init_seeded(seed)
        elif isinstance(weights, (str, bytes)):
            model.load_reference_state_dict(torch.load(weights, map_location="cpu", weights_only=True), strict=False)
        else:
            model.load_reference_state_dict(weights)
        self._padded_ops = {}
        self.model = model.to(self.device).eval()
        if self.half:
            self.model = self.model.half()
        self._prepared = False

    def prepare(self) -> "MSTPlusPlusPredictor":
        """Build every derived weight tensor of the model (packed MFMA fragments, stacked QKV, folded up-fuse weights, gather
        indices: MSTPlusPlus._prep fills its cache lazily with torch kernels on whichever stream runs the first frame) NOW, on
        the current stream, and wait for them: afterwards frames may be enqueued on any number of streams without one of them
        reading a cache entry another stream is still writing.  The derived tensors do not depend on the frame size, so one
        small frame through the same route builds them all.  Idempotent; a no-op on the CPU."""
        if self._prepared or self.device.type != "cuda":
            return self
        torch = self.torch
        with torch.cuda.device(self.device):
            probe = torch.zeros((32, 32, 3), dtype=torch.uint8, device=self.device)
            self.predict_device_nhwc(probe)
            torch.cuda.synchronize(self.device)
        self._prepared = True
        return self

    def predict_device_nhwc(self, frame_dev):
        """uint8 (H,W,3) torch tensor on the device -> (H, W, 32) contiguous channels-last cube: bands 0..30 and one
        zero padding channel (64-byte pixels; fp16 when half).  This is what the libavx hand-off consumes."""
        torch = self.torch
        H, W, _ = frame_dev.shape
        t, b, l, r = pad_amounts(H, W, self.stride)
        if self.half and frame_dev.dtype == torch.uint8 and self.stride % 8 == 0 and max(t, b) < H and max(l, r) < W and H > 1 and W > 1 and self.model.can_fuse_conv_in():
            y = self.model.forward_from_u8(frame_dev, (t, b, l, r))  # / 255, float16, reflect pad and conv_in in ONE kernel
            return y[0, t : t + H, l : l + W, :].contiguous()
        x = frame_dev.to(torch.float32).div_(255.0).permute(2, 0, 1).unsqueeze(0)  # 1x3xHxW
        if t or b or l or r:
            x = torch.nn.functional.pad(x, [l, r, t, b], mode="reflect")
        y = self.model.forward_nhwc(x.half() if self.half else x)
        return y[0, t : t + H, l : l + W, :].contiguous()

    def honeybee_device(self, frame_dev, op32, d_out, stream_handle):
        """uint8 (H, W, 3) device frame -> honeybee frame in d_out (a DeviceBuffer of H*W*3 bytes), everything enqueued on torch's current stream
        (= stream_handle).  Where the fused kernels apply, the 31-band cube is never written: conv_out's epilogue integrates it into the three catch
        planes (+ their statistics) that the honeybee tail starts from; else the cube is handed over by data_ptr.  op32: the HoneybeeOp padded to
        32 bands (op.padded_clone(32)).  Returns the tensors that must stay alive until the stream has run (the caller records / keeps them)."""
        torch = self.torch
        H, W, _ = frame_dev.shape
        t, b, l, r = pad_amounts(H, W, self.stride)
        fuse = (self.half and frame_dev.dtype == torch.uint8 and self.stride % 8 == 0 and max(t, b) < H and max(l, r) < W and H > 1 and W > 1
                and self.model.can_fuse_spectral() and op32.weights.shape == (3, 32))
        if fuse:
            planes, partials, n = self.model.forward_from_u8(frame_dev, (t, b, l, r), spectral=op32.weights)
            op32.run_device(None, d_out, 1, H, W, catches=(planes.data_ptr(), partials.data_ptr(), n), stream=stream_handle)
            return planes, partials
        cube = self.predict_device_nhwc(frame_dev)
        op32.run_device(None, d_out, 1, H, W, hsi_ptr=cube.data_ptr(), hsi_layout=0, hsi_dtype=1 if cube.dtype == torch.float16 else 0, stream=stream_handle)
        return (cube,)

    def predict_device(self, frame_dev):
        """uint8 (H,W,3) torch tensor on the device -> (31, H, W) contiguous tensor (fp16 when half)."""
        return self.predict_device_nhwc(frame_dev)[..., :31].permute(2, 0, 1).contiguous()

    def predict(self, image: np.ndarray) -> np.ndarray:
        """predict_rgb_to_hsi_torch for one image: HxWx3 (uint8 or float) -> HxWx31 float32."""
        torch = self.torch
        if self.device.type == "cuda" and image.dtype == np.uint8:  # the device route (what the honeybee hand-off runs): same kernels, same cube
            frame = torch.from_numpy(np.ascontiguousarray(image)).to(self.device)
            return self.predict_device_nhwc(frame)[..., :31].float().cpu().numpy()
        x01, pads = pad_to_multiple_reflect(to_float01(image), self.stride)
        xt = torch.from_numpy(np.ascontiguousarray(x01.transpose(2, 0, 1))[None]).to(self.device)
        y = self.model(xt.half() if self.half else xt.float())
        hsi = y[0].permute(1, 2, 0).float().cpu().numpy()
        return crop_pads(hsi, pads).astype(np.float32, copy=False)

    def honeybee(self, image: np.ndarray, op) -> np.ndarray:
        """uint8 frame -> MST++ cube -> HoneybeeOp (csrc/uv.hip), the cube handed over on the device:
        the (H,W,32) channels-last tensor's data_ptr goes straight into avx_honeybee_u8 on torch's current stream."""
        torch = self.torch
        if self.device.type != "cuda":
            raise RuntimeError("MST++ -> libavx hand-off needs the GPU (no CPU path)")
        from ..runtime import DeviceBuffer

        H, W, _ = image.shape
        frame = torch.from_numpy(np.ascontiguousarray(image)).to(self.device)
        out = torch.empty((H, W, 3), dtype=torch.uint8, device=self.device)
        ctx = op._ctx()
        stream = torch.cuda.current_stream().cuda_stream
        op32 = self._padded_ops.get(id(op))
        if op32 is None:
            op32 = self._padded_ops.setdefault(id(op), op.padded_clone(32))  # the cube is channels-last, 31 bands in a 32-wide group
        keep = self.honeybee_device(frame, op32, DeviceBuffer(ctx, out.data_ptr(), out.numel(), owned=False), stream)
        res = out.cpu().numpy()  # synchronises: `keep` may go
        del keep
        return res


class MstHoneybeeStreamOp:
    """The north-star route (uint8 frame -> MST++ cube -> honeybee tail -> uint8 frame) as a frame-loop operator
    (pipeline.FramePipeline's protocol: slot_buffers(k) lends a slot's device frames, run_device(...) enqueues one frame on
    the slot's HIP stream).  torch's kernels and libavx's ride the SAME stream: the slot's stream is made torch's current
    stream for the duration of the call, and the cube goes from the network to csrc/uv.hip by data_ptr (no copy, no sync)."""

    def __init__(self, predictor: "MSTPlusPlusPredictor", bee_op, H: int, W: int, depth: int = 3):
        torch = predictor.torch
        if predictor.device.type != "cuda":
            raise RuntimeError("MST++ -> libavx hand-off needs the GPU (no CPU path)")
        from ..runtime import DeviceBuffer

        self.pred, self.H, self.W = predictor, H, W
        self.ctx = bee_op._ctx()
        predictor.prepare()  # every derived weight exists and is complete before the slot streams start (they share the cache, unsynchronised)
        self._t_in = [torch.empty((H, W, 3), dtype=torch.uint8, device=predictor.device) for _ in range(depth)]
        self._t_out = [torch.empty((H, W, 3), dtype=torch.uint8, device=predictor.device) for _ in range(depth)]
        self._bufs = [(DeviceBuffer(self.ctx, a.data_ptr(), a.numel(), owned=False), DeviceBuffer(self.ctx, b.data_ptr(), b.numel(), owned=False))
                      for a, b in zip(self._t_in, self._t_out)]
        self._by_in = {bi.ptr: k for k, (bi, _) in enumerate(self._bufs)}
        self._bee = bee_op
        self._op32 = None
        self._streams = {}

    def slot_buffers(self, k: int):
        return self._bufs[k]

    def run_device(self, d_in, d_out, n_frames: int, H: int, W: int, stream=None):
        assert n_frames == 1 and (H, W) == (self.H, self.W)
        torch = self.pred.torch
        k = self._by_in[d_in.ptr]
        ext = self._streams.get(stream)
        if ext is None:
            ext = self._streams.setdefault(stream, torch.cuda.ExternalStream(stream, device=self.pred.device))
        with torch.cuda.stream(ext):
            if self._op32 is None:
                self._op32 = self._bee.padded_clone(32)  # host-side tables only (uploaded per call through the workspace of `stream`)
            for tns in self.pred.honeybee_device(self._t_in[k], self._op32, d_out, stream):
                tns.record_stream(ext)

    def release_streams(self):
        """Called by pipeline.FramePipeline.close(): its slot streams are about to be destroyed, so the ExternalStream wrappers
        (and the allocator pools torch keyed by them) must not outlive them."""
        torch = self.pred.torch
        if self._streams:
            torch.cuda.synchronize(self.pred.device)
            self._streams.clear()
            torch.cuda.empty_cache()
