"""GPU probe 26: placement effect, part 3: one index; (a) one context launched on several streams, (b) several
contexts launched on one stream — is it the hardware queue or the memory of the context?"""
import json
import os
import sys

import numpy as np
import torch

torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import instant_distance_amd as ida  # noqa: E402
from bench import synth  # noqa: E402

n, dim, nq, ef = 1_000_000, 300, 10000, 100
dev = torch.device("cuda", 0)
d_pts = synth(torch, n, dim, 123456789, dev)
h = ida.Hnsw.from_device_points(d_pts.data_ptr(), n, dim, ida.Builder())
d_q = synth(torch, nq, dim, 123456790, dev)
pid = torch.empty(nq, ef, dtype=torch.int32, device=dev); dd = torch.empty(nq, ef, dtype=torch.float32, device=dev)
cnt = torch.empty(nq, dtype=torch.int32, device=dev)


def run(s, stream):
    h.search_batch_device(s, d_q.data_ptr(), nq, pid.data_ptr(), dd.data_ptr(), cnt.data_ptr(), 0, stream.cuda_stream)
    stream.synchronize()
    return float(s.kernel_times_ms(1)[0])


streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(5)]
ctx = ida.Search()
t = [[] for _ in streams]
for rep in range(8):
    for i, st in enumerate(streams):
        t[i].append(run(ctx, st))
print(json.dumps({"one context, six streams (first = null stream): ms_median": [round(float(np.median(x[2:])), 3) for x in t]}), flush=True)
ctxs = [ida.Search() for _ in range(10)]
t = [[] for _ in ctxs]
for rep in range(8):
    for i, c in enumerate(ctxs):
        t[i].append(run(c, streams[0]))
print(json.dumps({"ten contexts, null stream: ms_median": [round(float(np.median(x[2:])), 3) for x in t]}), flush=True)
