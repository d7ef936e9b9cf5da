"""bench.py --gpus N must start its N ranks itself when no launcher did: the launch, the
rendezvous, MAX over ranks and the single JSON line from rank 0, exercised on CPU over gloo
(--launcher-dry-run: no GPU work); and the two failure modes - a rank that dies takes the run down with a
non-zero exit code, --gpus that disagrees with an external launcher's WORLD_SIZE is an error."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(extra)
    return env


def test_self_launch_two_ranks_one_json_line():
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--launcher-dry-run"],
                         capture_output=True, text=True, timeout=300, env=_env())
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout                         # rank 0's JSON line and nothing else (no backend banners)
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["max_over_ranks"] == 2.0 and out["steps"] == 3 and out["warmup"] == 1
    # the N > 1 line diagnoses itself - per exchange mode the bytes a rank sends and the exposed exchange
    # time, the compute-only step it is measured against, and RCCL's algorithm / protocol (null in the dry run)
    ex = out["exchange"]
    assert set(ex) >= {"mode", "compute_only_ms", "per_mode", "rccl"}
    assert set(ex["per_mode"]) == {"compact", "compact-early", "allreduce"}
    for m, e in ex["per_mode"].items():
        assert set(e) >= {"collectives", "bytes_sent_per_rank", "bytes_received_per_rank", "exposed_ms"}, m
    n = 1_000_000                                              # world 2: all-reduce sends 2 (W-1)/W B, all-gather W-1 blocks
    assert ex["per_mode"]["allreduce"]["bytes_sent_per_rank"] == 236 * n
    assert ex["per_mode"]["compact"]["bytes_sent_per_rank"] == 12 * n + 44 * n


def test_rccl_log_lines_are_parsed():
    """The RCCL INFO wording is version dependent; the parser takes any line naming a collective with an algorithm and
    a protocol, by id or by name, and the channel count."""
    import importlib.util
    import tempfile
    spec = importlib.util.spec_from_file_location("cugs_bench", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    with tempfile.NamedTemporaryFile("w", suffix=".log", delete=False) as f:
        f.write("host:1:2 [0] NCCL INFO Connected all rings\n"
                "host:1:2 [0] NCCL INFO 16 coll channels, 0 collnet channels, 0 nvls channels, 16 p2p channels\n"
                "host:1:2 [0] NCCL INFO AllReduce: opCount 3 sendbuff 0x1 recvbuff 0x2 count 11000000 datatype 7 op 0 -> algo 1 proto 2\n"
                "host:1:2 [0] NCCL INFO AllGather: 12000000 Bytes -> Algo Ring proto LL128 time 1.0\n")
        path = f.name
    got = bench.parse_rccl_log([path])
    os.unlink(path)
    assert got["AllReduce"] == ["Ring/Simple"] and got["AllGather"] == ["Ring/LL128"] and got["channels"] == 16
    assert bench.parse_rccl_log(["/nonexistent"]) == {}


def test_gpus_must_match_external_world_size():
    res = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--launcher-dry-run"], capture_output=True, text=True,
                         timeout=120, env=_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_PORT="29999",
                                               MASTER_ADDR="127.0.0.1"))
    assert res.returncode != 0 and "WORLD_SIZE=2" in res.stderr


def test_failing_rank_fails_the_run():
    # an unknown flag makes every child exit 2 (argparse) before the rendezvous: the parent must report failure
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launcher-dry-run", "--config", "nope"],
                         capture_output=True, text=True, timeout=120, env=_env())
    assert res.returncode != 0
