"""Print the recorded device plan (stage list, program sizes) of UV species; needs the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from animal_vision_amd import animals
from animal_vision_amd.synthetic import structured_frame

f = structured_frame(0, 270, 480)
for mod in sys.argv[1:] or sorted(animals.UV_CLASS):
    sp = getattr(animals, animals.UV_CLASS[mod])()
    be = sp._plan(f, "day" if mod == "rat_uv" else None)
    print(f"== {mod}: {len(be.plan)} calls, {be.n_programs} programs, {be.n_insn} instructions")
    for s in be.stages:
        print("   ", s)
