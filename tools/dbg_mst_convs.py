"""Time the conv shapes of one MST++ forward at 1080p (NHWC fp16) to see which ones MIOpen serves with naive kernels."""
import torch, torch.nn.functional as F
H, W = 1080, 1920
dev = "cuda"
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def nhwc(c, h, w): return torch.randn(1, c, h, w, device=dev, dtype=torch.half).contiguous(memory_format=torch.channels_last)
cases = [("conv_in 3->31 3x3", 3, 31, 3, 1, 1, H, W), ("3x3 31->31 full", 31, 31, 3, 1, 1, H, W), ("4x4s2 31->62", 31, 62, 4, 2, 1, H, W),
         ("4x4s2 62->124", 62, 124, 4, 2, 1, H // 2, W // 2)]
for name, ci, co, k, s, p, h, w in cases:
    x = nhwc(ci, h, w); wt = torch.randn(co, ci, k, k, device=dev, dtype=torch.half)
    print(f"{name:24s} {t(lambda: F.conv2d(x, wt, stride=s, padding=p)):8.2f} ms")
    x32 = nhwc(32 if ci == 31 else ci, h, w) if ci in (31,) else None
for name, ci, co, h, w in (("convT 124->62", 124, 62, H // 4, W // 4), ("convT 62->31", 62, 31, H // 2, W // 2)):
    x = nhwc(ci, h, w); wt = torch.randn(ci, co, 2, 2, device=dev, dtype=torch.half); b = torch.randn(co, device=dev, dtype=torch.half)
    print(f"{name:24s} {t(lambda: F.conv_transpose2d(x, wt, b, stride=2)):8.2f} ms")
    x2 = x.permute(0, 2, 3, 1).reshape(-1, ci); wc = wt.permute(0, 2, 3, 1).reshape(ci, 4 * co).contiguous()
    def gemm_path():
        y = (x2 @ wc).reshape(1, h, w, 2, 2, co) + b
        return y.permute(0, 1, 3, 2, 4, 5).reshape(1, 2 * h, 2 * w, co)
    print(f"{'  as GEMM + shuffle':24s} {t(gemm_path):8.2f} ms")
