"""1-rank RCCL sanity: replicate_index() over backend nccl (meta broadcast on torch tensors, bulk broadcast on the
library-owned device buffers through zero-copy views)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import instant_distance_amd as ida  # noqa: E402
from instant_distance_amd import dist as idd  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
pts = np.random.default_rng(0).random((20000, 300), dtype=np.float32)
h = ida.Hnsw.from_ordered_points(pts, ida.Builder())
h2 = idd.replicate_index(h, ida.Builder(), src=0)
q = pts[:50]
a = h2.search_batch(q, ida.Search())
assert np.array_equal(a.pid[:, 0], np.arange(50))
dist.barrier()
dist.destroy_process_group()
print("nccl selftest ok")
