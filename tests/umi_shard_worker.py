"""Worker of tests/test_gpu_dist_umi.py: one rank of dist.umi_count_sharded over gloo, all ranks on GPU 0.
argv: <npz with the shards> <out json>.  Launched by torch.distributed.run."""
import json
import os
import sys

import numpy as np
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastq_utils_amd as fq  # noqa: E402
from fastq_utils_amd import dist as fdist  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    data = np.load(sys.argv[1])
    stream = data["shard%d" % rank].tobytes()
    with fq.Context(0) as ctx:
        got = fdist.umi_count_sharded(ctx, stream)
    outs = [None] * world
    dist.all_gather_object(outs, got["entries"])
    if rank == 0:
        with open(sys.argv[2], "w") as f:
            json.dump({"entries_u": [e for o in outs for e in o[0]], "entries_r": [e for o in outs for e in o[1]],
                       "n_entries": got["n_entries"], "total": got["total"], "tot_reads": got["tot_reads"],
                       "tot_umi": got["tot_umi"], "rl_undefined": got["rl_undefined"]}, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
