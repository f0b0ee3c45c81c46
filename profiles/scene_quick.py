#!/usr/bin/env python3
"""Times of the frames of profiles/scene_ab.py only (no buffers kept): for sweeps over the knobs of
the scene kernel (environment variables read by lf_set_scene, rebuilt libraries via LF_LIB)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import scene_ab

np.savez_compressed = lambda out, **kw: None
scene_ab.main("/dev/null")
