"""The driver's contract with bench.py, checked on the GPU box: `python bench.py --gpus 1 --steps K --warmup W` prints
exactly ONE line on stdout, a JSON object with the keys the contract names, measured on the headline workload
(BASELINE.json configs[2]) through the product's own render / render_backward sequence, with the roofline object for
the dominant kernel and the parity probe against the oracle.  (The CPU baseline leg is skipped here: ~15 s of host work
that `tests/test_host_logic.py` and the default `python bench.py` run cover.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "8", "--warmup", "2",
                          "--spinup-ms", "0", "--no-cpu-baseline", *flags], capture_output=True, text=True, timeout=600,
                         cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout[-2000:]                     # ONE line on stdout
    return json.loads(lines[0])


def test_bench_line_keeps_the_contract():
    out = _bench()
    assert out["metric"] == "fwd+bwd Mpixels/s @1080p, 1M Gaussians, SH3" and out["unit"] == "Mpixels/s"
    assert out["n_gpus"] == 1 and out["steps"] == 8 and out["warmup"] == 2
    assert out["higher_is_better"] is True and out["scaling"] == "weak" and out["vs_baseline"] is None
    assert out["dtype"] == "f32" and out["data"] == "synthetic"
    cfg = out["config"]
    assert cfg["workload"] == "1M/1920x1080/SH3 fwd+bwd" and cfg["n_gaussians"] == 1000000
    assert (cfg["width"], cfg["height"], cfg["sh_degree"]) == (1920, 1080, 3)
    assert cfg["pairs"] == 8376524                                  # the scene of SURVEY 8d, seed 1234
    # value = pixels of all ranks' views / time; ms_per_step is the same time
    assert out["value"] > 0 and abs(out["value"] - 1920 * 1080 / (out["ms_per_step"] * 1e-3) / 1e6) <= 1e-3 * out["value"]
    r = out["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["algorithmic_bytes"] > 0 and ("traffic" in r)
    assert set(out["stages_ms"]) == {"project_forward", "sort", "raster_forward", "raster_backward", "project_backward"}
    p = out["parity"]                                               # the probe against the oracle rides along
    assert p["rgb"] == 0.0 and p["sort_order_equal"] and p["n_contrib_equal"] and p["tiles_touched_equal"]
    assert p["grad_max_rel_err"] <= 1e-4


def test_keyed_and_unkeyed_routes_agree_on_the_workload():
    """`--unkeyed-sort` (stage-by-stage hand-over) and the default (the projection keys the sort) time the same frame."""
    a, b = _bench("--no-parity"), _bench("--no-parity", "--unkeyed-sort")
    assert a["config"]["pairs"] == b["config"]["pairs"] == 8376524


def test_launched_bench_line_on_one_rank():
    """The driver's N > 1 command shape (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) with
    the ONE rank this box can hold: RCCL comes up on the GPU, the exchange modes are calibrated and timed (their
    collectives run, over a world of one), stdout carries exactly one JSON line, and the line diagnoses its exchange -
    bytes per mode, exposed time against the compute-only step, what RCCL's log said (the box exports
    NCCL_DEBUG=VERSION, which the bench overrides for its own process)."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "1", "--steps", "6", "--warmup", "2", "--rehearse-calibration"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["steps"] == 6 and out["config"]["parallelism"].startswith("dp1-views+rccl-")
    ex = out["exchange"]
    assert set(ex["per_mode"]) == {"compact", "compact-early", "allreduce"} and ex["compute_only_ms"] > 0
    assert ex["per_mode"]["allreduce"]["collectives"][0]["bytes"] == 236 * 1000000      # 59 floats per Gaussian
    assert ex["rccl"] is not None and "unparsed_tail" not in ex["rccl"] or ex["rccl"].get("channels")
    assert "cpu_baseline" not in out or out["n_gpus"] == 1
