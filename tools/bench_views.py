#!/usr/bin/env python3
"""Times SURVEY §8(f) N4 at 1080p: an iteration's target from the device-resident view cache, beside what the
reference does per iteration after decoding (float conversion + CPU resize + 12 B/pixel upload, restated in
oracle/views_oracle.py; its stb_image decode comes on top)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
from test_views_oracle import load_views_oracle
pkg = ge.load_package(); vo = load_views_oracle(); dev = torch.device("cuda:0")
raw_same = np.random.default_rng(0).integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
raw_big = np.random.default_rng(1).integers(0, 256, (2160, 3840, 3), dtype=np.uint8)
cache = pkg.ViewCache(dev); a = cache.add(raw_same); b = cache.add(raw_big)
def gpu(i):
    for _ in range(3): cache.target(i, 1920, 1080)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [cache.target(i, 1920, 1080) for _ in range(20)]; e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20
def cpu(raw):
    t0 = time.perf_counter(); t = torch.from_numpy(vo.target(raw, 1920, 1080)).to(dev); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
print("target 1920x1080 from a cached 1920x1080 view: HIP %.3f ms   host convert + upload %.1f ms" % (gpu(a), cpu(raw_same)))
print("target 1920x1080 from a cached 3840x2160 view: HIP %.3f ms   host convert + resize + upload %.1f ms" % (gpu(b), cpu(raw_big)))
print("cache: %d views, %.1f MB on the device (a 1080p view is 6.2 MB: 40,000 of them fit in 288 GB at 85 %%)" % (len(cache), cache.bytes() / 1e6))
