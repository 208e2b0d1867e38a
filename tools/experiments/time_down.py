"""Time avx_mst_down4x4 alone vs MIOpen's conv at 4K / 1080p (device events, 20 launches)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch.nn.functional as F
from animal_vision_amd.ml.mst_plus_plus import _AVX, pack_down4x4

dev = torch.device("cuda")
for (h, w, c) in [(2160, 3840, 32), (1080, 1920, 32), (1080, 1920, 64), (540, 960, 64)]:
    x = torch.randn(1, h, w, c, device=dev).half()
    wt = (torch.randn(2 * c, c, 4, 4, device=dev) * 0.08).half()
    wq = pack_down4x4(wt)
    xn = x.permute(0, 3, 1, 2)
    wn = wt.contiguous(memory_format=torch.channels_last)
    for name, fn in [("down4x4", lambda: _AVX.down4x4(x, wq)), ("miopen", lambda: F.conv2d(xn, wn, stride=2, padding=1))]:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        gb = (h * w * 2 * c + h * w // 4 * 4 * c) / 1e9
        print(f"{h}x{w}x{c} {name}: {ms*1e3:.1f} us  {gb/ms*1e3:.0f} GB/s  {os.environ.get('AVX_DOWN_ABLATE','')}")
