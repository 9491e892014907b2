import json, os, sys
import numpy as np
sys.path.insert(0, '/root/repo')
import torch, bench
import instant_distance_amd as ida
dev = torch.device("cuda", 0)
n, dim, nq = 1_000_000, 300, 10_000
d_pts = bench.synth(torch, n, dim, 123456789, dev); d_q = bench.synth(torch, nq, dim, 123456790, dev)
torch.cuda.synchronize()
h = ida.Hnsw.from_device_points(d_pts.data_ptr(), n, dim, ida.Builder())
st = torch.cuda.current_stream().cuda_stream
o = (torch.empty(nq, 100, dtype=torch.int32, device=dev), torch.empty(nq, 100, dtype=torch.float32, device=dev), torch.empty(nq, dtype=torch.int32, device=dev), torch.zeros(nq, 3, dtype=torch.int32, device=dev))
for w in (128, 256, 320, 384, 512, 640, 768, 1024):
    row = {"nq": w}
    for nm, env in (("single", {"IDIST_QUAD_NQ": "0"}), ("quad", {"IDIST_QUAD_NQ": "4000000000"})):
        os.environ.update(env)
        s = ida.Search()
        for i in range(10):
            h.search_batch_device(s, d_q[i * 512:].data_ptr(), w, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), st)
        torch.cuda.synchronize()
        row[nm + "_ms"] = round(float(np.median(s.kernel_times_ms(8))), 4)
        del s
    print(json.dumps(row), flush=True)
