"""Does ONE large all_to_all_single deliver its bytes?  (dist.exchange_fingerprints sends its buckets in rounds of at most
256 MiB per peer because a single 1.6 GB message once came back different - round 2, a 1-rank RCCL group on a one-GPU
box, `bench.py --dedup-extra` with 100 M names; every round has carried a checksum per slice since.)  This repeats the
observation's shape: a 1-rank group, messages of 0.25 .. 3.2 GB, the received tensor against the sent one.
  python3 tools/rccl_big_message.py"""
import os

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29671")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
print("torch", torch.__version__, "nccl/rccl", torch.cuda.nccl.version())
for gb in (0.25, 0.8, 1.6, 3.2):
    n = int(gb * 1e9) // 16 * 16
    bad = 0
    for rep in range(3):
        g = torch.Generator(device=dev)
        g.manual_seed(100 + rep)
        send = torch.randint(0, 256, (n,), dtype=torch.uint8, device=dev, generator=g)
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send, [n], [n])
        torch.cuda.synchronize()
        bad += int((recv != send).sum().item() != 0)
    print(f"{gb:4.2f} GB in one all_to_all_single, 3 repetitions: {'identical' if not bad else str(bad) + ' of 3 DIFFER'}", flush=True)
dist.destroy_process_group()
