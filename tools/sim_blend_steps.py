"""Step counts of alternative blend-kernel decompositions on the config-3 frame, replayed on the CPU (no GPU needed).
Renders the frame with the oracle (host threads), dumps what the replay needs, builds tools/sim_blend_steps.c and runs it:
    python tools/sim_blend_steps.py  > profiles/<tag>_sim_blend_steps.log
For the backward (quirk Q1: contributors counted from the END of each list) and the forward it prints the wave-steps and
live lanes of the shipped 8x8-quad waves, of four 4x4 cells per wave walking their own hit lists, of two 8x4 halves, and
the perfect-packing bound."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge   # noqa: E402


def main():
    pkg, orc = ge.load_package(), ge.load_oracle()
    wl = pkg.scene.CONFIGS["config3"]
    arrays = pkg.scene.make_gaussians(wl.n, wl.width, wl.height, sh_degree=wl.sh_degree)
    cam = pkg.scene.make_camera(wl.width, wl.height)
    K = cam.intrinsics
    ref = orc.render(arrays, cam.rotation, cam.translation, K.fx, K.fy, K.cx, K.cy, wl.width, wl.height,
                     threads=orc.host_threads())
    with tempfile.TemporaryDirectory() as tmp:
        for name, key in (("n_contrib", "n_contrib"), ("values", "values"), ("tile_ranges", "tile_ranges"),
                          ("means_2d", "means_2d"), ("cov_2d_inv", "cov_2d_inv"), ("opa", "opacities_act")):
            np.ascontiguousarray(ref[key]).tofile(os.path.join(tmp, name + ".bin"))
        exe = os.path.join(tmp, "sim")
        subprocess.run(["gcc", "-O2", "-fopenmp", "-o", exe, os.path.join(ROOT, "tools", "sim_blend_steps.c"), "-lm"], check=True)
        print("config 3: %d Gaussians, %dx%d, %d pairs" % (wl.n, wl.width, wl.height, ref["total_pairs"]), flush=True)
        subprocess.run([exe], cwd=tmp, check=True)


if __name__ == "__main__":
    main()
