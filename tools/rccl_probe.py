import os, sys, glob
os.environ["NCCL_DEBUG"] = "INFO"
os.environ["NCCL_DEBUG_SUBSYS"] = "INIT,TUNING"
os.environ["NCCL_DEBUG_FILE"] = "/tmp/probe_rccl_%h_%p.log"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.ones(1 << 20, device="cuda")
dist.all_reduce(x); y = torch.empty(1 << 20, device="cuda"); dist.all_gather_into_tensor(y, x)
torch.cuda.synchronize()
dist.destroy_process_group()
files = glob.glob("/tmp/probe_rccl_*")
print("files", files, file=sys.stderr)
for f in files:
    lines = open(f, errors="replace").read().splitlines()
    print(len(lines), "lines", file=sys.stderr)
    for l in lines[:12] + [l for l in lines if "lgo" in l or "roto" in l][:12]:
        print("   ", l[:200], file=sys.stderr)
