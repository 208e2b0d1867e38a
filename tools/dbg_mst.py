import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_golden
from animal_vision_amd.animals import HoneyBee
from animal_vision_amd.ml import MSTPlusPlusPredictor
from animal_vision_amd.synthetic import structured_frame
from oracle import cpu_ref as O
w = load_golden("mstpp_weights_fp16")
sd = {k: torch.from_numpy(w[k].astype(np.float32)) for k in w.files}
pred = MSTPlusPlusPredictor(sd, half=False)
frame = structured_frame(1, 72, 88)
hsi = pred.predict(frame)
print("cube range", hsi.min(), hsi.max())
lam = np.linspace(400.0, 700.0, 31, dtype=np.float32)
U, B, G = O.honeybee_catches(hsi, lam)
print("catches ranges", [(float(p.min()), float(p.max())) for p in (U, B, G)])
want, _ = O.honeybee_tail(U, B, G, np.uint8)
bee = HoneyBee()
op = bee._operator()
out_np, planes = op(frame, hsi=hsi, hsi_layout="nhwc", return_planes=True)
Uw, Bw, Gw = O.von_kries_white_patch(U, B, G)
wantp = np.stack([O.gaussian_blur(p, 0.2) for p in (Uw, Bw, Gw)])
print("planes max abs err", np.abs(planes - wantp).max(), "rel", (np.abs(planes - wantp) / (np.abs(wantp) + 1e-6)).max())
d = np.abs(out_np.astype(int) - want.astype(int)); print("numpy-cube route: max", d.max(), "frac", (d > 0).mean())
bee2 = HoneyBee(hsi_model=pred)
out_dev = bee2.visualize(frame)[1]
d = np.abs(out_dev.astype(int) - want.astype(int)); print("device hand-off route: max", d.max(), "frac", (d > 0).mean())
d = np.abs(out_dev.astype(int) - out_np.astype(int)); print("hand-off vs numpy-cube: max", d.max(), "frac", (d > 0).mean())
cube_dev = pred.predict_device(torch.from_numpy(frame).cuda()).float().cpu().numpy().transpose(1, 2, 0)
print("predict vs predict_device cube diff", np.abs(cube_dev - hsi).max())
