"""Test tool (uses the oracle as the checker). Mantis at 3840x2160: how many jitter runs of the oracle it takes to reproduce the device's >1-code pixels (tests/_sensitivity.py)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # tests/tools/ -> repo root
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from animal_vision_amd.animals import MantisShrimp
from animal_vision_amd.synthetic import structured_frame
from oracle import cpu_ref
from _sensitivity import outlier_stats
f = structured_frame(0, 2160, 3840)
_, got = MantisShrimp().visualize(f)
t = time.time(); _, want = cpu_ref.mantis_visualize(f); print("oracle s", time.time() - t, flush=True)
for runs in (3, 8, 16):
    st = outlier_stats(got, want, lambda seed: cpu_ref.mantis_visualize(f, _jit=cpu_ref.relative_jitter(seed))[1], runs=runs)
    print(runs, st, flush=True)
    if st.get("unexplained_px", 0) == 0: break
