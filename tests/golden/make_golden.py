#!/usr/bin/env python3
"""Regenerates the fixtures of this directory from the reference checkout (run in the build container, where
/root/reference exists; the GPU box only ever sees the committed .npy files).

The reference ships no numeric test vectors; what it does ship are its test IMAGES.  They are data, stored here
losslessly as uint8 arrays (imread of the 8-bit greyscale PNGs, run_Gaussian_demo.m:117):
  man_512.npy   <- images/man.png   (512 x 512)   config[1] of BASELINE.json, tiled 4 x 4 for the 2048^2 bench
  cman_256.npy  <- image/cman.png   (256 x 256)   config[0]
"""
import os
import sys

import numpy as np
from PIL import Image

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
for out, src in (("man_512.npy", "images/man.png"), ("cman_256.npy", "image/cman.png")):
    a = np.asarray(Image.open(os.path.join(REF, src)).convert("L"), dtype=np.uint8)
    np.save(os.path.join(HERE, out), a)
    print(out, a.shape, a.dtype, int(a.min()), int(a.max()))
