"""Times the FeedForward / attention-tail kernels of MST++ on their own at a given size (GPU box): python tools/experiments/time_ffn.py [H W]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import animal_vision_amd as av
from animal_vision_amd.ml import MSTPlusPlusPredictor
from animal_vision_amd.ml.mst_plus_plus import _AVX

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2160, 3840)
pred = MSTPlusPlusPredictor(None, seed=0, half=True)
m = pred.model
pred.prepare()
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for c, div, pfx in ((32, 1, "body.0.encoder_layers.0.0"), (64, 2, "body.0.encoder_layers.1.0"), (128, 4, "body.0.bottleneck")):
    x = (torch.randn(1, H // div, W // div, c, device="cuda") * 0.5).half()
    x[..., 31::32] = 0
    us = t(lambda: m._ffn(x, pfx + ".blocks.0.1"))
    us2 = t(lambda: m._ms_msa(x, pfx + ".blocks.0.0", c // 32))
    line = f"C={c}: ffn {us:8.1f} us   msa (gram + pack + tail) {us2:8.1f} us"
    if c in (32, 64, 128):
        from animal_vision_amd.ml.mst_plus_plus import pack_dw_mfma, pack_fragments16
        p = pfx + ".blocks.0.0"
        heads = c // 32
        wqkv = torch.cat([m._w(p + ".to_q.weight", (0, 1)), m._w(p + ".to_k.weight", (0, 1)), m._w(p + ".to_v.weight", (0, 1))], 0).t().contiguous()
        wv16 = pack_fragments16(wqkv[:, 2 * c:].contiguous())
        gram = torch.randn(heads, 32, 32, device="cuda"); nq = torch.rand(c, device="cuda") + 0.5; nk = torch.rand(c, device="cuda") + 0.5
        resc = torch.ones(heads, device="cuda"); wpt = m._w(p + ".proj.weight", (0, 1)).t().float().contiguous(); b32 = m._w(p + ".proj.bias", (0,)).float().contiguous()
        d1 = pack_dw_mfma(m._w(p + ".pos_emb.0.weight", (0,))); d2 = pack_dw_mfma(m._w(p + ".pos_emb.2.weight", (0,)))
        mp = _AVX.attn_pack_mx(gram, nq, nk, resc, wpt)
        o = torch.empty_like(x[0])
        wv16h = pack_fragments16(wqkv[:, 2 * c:].contiguous(), halfrow=True)
        us3 = t(lambda: _AVX.attn_tail_mx(x[0], wv16h, mp, d1, d2, b32, o))
        us4 = float("nan")
        if c != 128:
            t1 = m._w(p + ".pos_emb.0.weight", (0,)).reshape(c, 9).t().contiguous(); t2 = m._w(p + ".pos_emb.2.weight", (0,)).reshape(c, 9).t().contiguous()
            m16 = _AVX.attn_pack16(gram, nq, nk, resc, wpt)
            us4 = t(lambda: _AVX.attn_tail_x(x[0], wv16, m16, t1, t2, b32, o))
        line += f"   tail_mx alone {us3:8.1f} us (round-2 tail {us4:8.1f})"
        _AVX._dwmx = False
        us6 = t(lambda: m._ms_msa(x, pfx + ".blocks.0.0", c // 32))
        _AVX._dwmx = True
        line += f"   msa round-2 {us6:8.1f}"
    _AVX._dwmx = False
    us5 = t(lambda: m._ffn(x, pfx + ".blocks.0.1"))
    _AVX._dwmx = True
    line += f"   ffn round-2 {us5:8.1f}"
    print(line)
if os.environ.get("AVX_FFN_STAMPS"):
    pass
