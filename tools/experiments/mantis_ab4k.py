import os, numpy as np, sys
sys.path.insert(0, '/root/repo')
from animal_vision_amd.animals import MantisShrimp
from animal_vision_amd.synthetic import structured_frame
f = structured_frame(0, 2160, 3840)
m = MantisShrimp()
os.environ["AVX_MANTIS_UP"] = "0"
b0, o0 = m.visualize(f)
os.environ["AVX_MANTIS_UP"] = "1"
b1, o1 = m.visualize(f)
d = np.abs(o0.astype(int) - o1.astype(int))
print("max diff", d.max(), "count", (d > 0).sum(), "first", np.argwhere(d > 0)[:5].tolist())
