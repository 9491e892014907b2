"""GPU probe 30: does the placement of the INDEX matter as well?  One context (fixed visited array), the index's
buffers moved into fresh allocations between measurements (idist_index_rehome); spacer allocations in between so
that successive homes are far apart."""
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
st = torch.cuda.current_stream()


def run(s, reps=6):
    t = []
    for _ in range(reps):
        h.search_batch_device(s, d_q.data_ptr(), nq, pid.data_ptr(), dd.data_ptr(), cnt.data_ptr(), 0, st.cuda_stream)
        st.synchronize()
        t.append(float(s.kernel_times_ms(1)[0]))
    return round(float(np.median(t[1:])), 3)


for ctx_no in range(2):
    s = ida.Search()
    res = [run(s)]
    spacers = []
    for i in range(10):
        spacers.append(torch.empty(6 << 30, dtype=torch.uint8, device=dev))   # push the next home elsewhere
        h.rehome()
        res.append(run(s))
        if len(spacers) > 4:
            spacers.pop(0)
    print(json.dumps({"context": ctx_no, "ms_median after each re-home of the index (first = built in place)": res}), flush=True)
    del spacers, s
    torch.cuda.empty_cache()
