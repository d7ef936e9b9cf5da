#!/usr/bin/env python3
"""The C++ host (cuda-gaussian-splatting_amd/adapter: what links under the reference's render() / render_backward() /
FusedAdam) timed on the benchmark workloads: writes the scene to /dev/shm as raw arrays and runs adapter_bench.bin.
    python tools/bench_cpp_host.py [config3|config4] [steps] [warmup]"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
cfg = sys.argv[1] if len(sys.argv) > 1 else "config3"
steps = sys.argv[2] if len(sys.argv) > 2 else "200"
warmup = sys.argv[3] if len(sys.argv) > 3 else "150"
wl = pkg.scene.CONFIGS[cfg]
arrays = pkg.scene.make_gaussians(wl.n, wl.width, wl.height, sh_degree=wl.sh_degree, mu_s=wl.mu_s)
cam = pkg.scene.make_camera(wl.width, wl.height)
exe = os.path.join(ROOT, "cuda-gaussian-splatting_amd", "adapter", "adapter_bench.bin")
with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
    for k, v in arrays.items():
        np.ascontiguousarray(v, np.float32).tofile(os.path.join(d, f"{k}.bin"))
    pkg.scene.make_dl_dcolor(wl.width, wl.height, seed=pkg.scene.GRAD_SEED).tofile(os.path.join(d, "dl_dcolor.bin"))
    abi = cam.to_abi()
    np.array(list(abi.view) + [abi.fx, abi.fy, abi.cx, abi.cy] + list(abi.cam_center) + [0.0, 0.0, 0.0], np.float32).tofile(
        os.path.join(d, "camera.bin"))
    c = str(pkg.sh_coeff_count(wl.sh_degree))
    for extra in ([], ["adam"]):
        res = subprocess.run([exe, d, str(wl.n), c, str(wl.width), str(wl.height), steps, warmup] + extra,
                             capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
        print(res.stdout.strip() or res.stderr.strip()[-2000:], flush=True)
