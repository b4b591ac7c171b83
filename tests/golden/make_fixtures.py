"""Generates tests/golden/ref_bitmaps.npz from the reference's own test
bitmaps (Tests/Tests/*.bmp).  Run ONCE in the build container (the reference
tree does not exist on the GPU box):

    python tests/golden/make_fixtures.py

The npz holds DATA only: the four cost rasters (cost = ~pixel, 0 -> 1, the
convention of Simulator/simulator/run_simulator.py:106-111 for the
high-resolution layer) and the start/goal encoded in each file name
(<name>_<fromx>_<fromy>_<tox>_<toy>_.bmp).
"""
import os
import numpy as np
from PIL import Image

REF = "/root/reference/Tests/Tests"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_bitmaps.npz")

out = {}
for fn in sorted(os.listdir(REF)):
    if not fn.endswith(".bmp"):
        continue
    name, fx, fy, tx, ty = fn[:-5].split("_")[:5]
    img = np.array(Image.open(os.path.join(REF, fn)).convert("L"), dtype=np.uint8)
    cost = (~img).astype(np.uint8)
    cost = cost + (cost == 0).astype(np.uint8)
    out[name + "_cost"] = cost
    out[name + "_startgoal"] = np.array([fx, fy, tx, ty], dtype=np.float32)
    print(fn, cost.shape, cost.min(), cost.max())
np.savez_compressed(OUT, **out)
print("wrote", OUT, os.path.getsize(OUT), "bytes")
