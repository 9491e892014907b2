"""Round-2 GPU probe of the search kernel on C3 (1M x 300): (a) does the kernel time still depend on where a
context's visited memory lands (ten contexts in a row, one index), (b) the experimental walk variants of the
tuning build (libidist_tune.so, -DIDIST_TUNE) on ONE index and matching results, (c) resident slots.
usage: python scripts/probe_r02_search.py [out.jsonl]   (GPU box)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import instant_distance_amd as ida  # noqa: E402
from instant_distance_amd import _capi  # noqa: E402

out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "probe_r02_search.jsonl")
tune = os.path.join(ROOT, "instant-distance_amd", "csrc", "libidist_tune.so")
if os.path.exists(tune):
    _capi._singleton = _capi.Lib(tune)
fo = open(out_path, "a")


def emit(**kw):
    print(json.dumps(kw), flush=True)
    fo.write(json.dumps(kw) + "\n")
    fo.flush()


dev = torch.device("cuda", 0)
n, dim, nq = 1_000_000, 300, 10_000
d_pts = bench.synth(torch, n, dim, 123456789, dev)
d_q = bench.synth(torch, nq, dim, 123456790, dev)
torch.cuda.synchronize()
h = ida.Hnsw.from_device_points(d_pts.data_ptr(), n, dim, ida.Builder())
st = h.build_stats()
emit(what="build", seconds=st.seconds, lib=os.path.basename(_capi.lib().path))
outs = (torch.empty(nq, 100, dtype=torch.int32, device=dev), torch.empty(nq, 100, dtype=torch.float32, device=dev),
        torch.empty(nq, dtype=torch.int32, device=dev), torch.empty(nq, 3, dtype=torch.int32, device=dev))


def time_ctx(search, reps=5, nq_=nq):
    for _ in range(reps + 1):
        h.search_batch_device(search, d_q.data_ptr(), nq_, outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(),
                              outs[3].data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    search.check_status()
    kt = search.kernel_times_ms(reps)
    return [round(float(x), 3) for x in kt]


# (a) ten contexts, one index, all kept alive
ctxs = []
times = []
for i in range(10):
    s = ida.Search()
    times.append(float(np.median(time_ctx(s, 3))))
    ctxs.append(s)
emit(what="ten contexts in a row, one index (median ms per 10k-query launch)", ms=times)
ref_pid = outs[0].clone()
ref_d = outs[1].clone()
ref_ctr = outs[3].clone()
ctr = ref_ctr.cpu().numpy().astype(np.int64)
alg = int((ctr[:, 0] * 4 * dim + ctr[:, 1] * 256 + ctr[:, 2] * 128 + 8 * 100).sum())
emit(what="algorithmic bytes per launch", bytes=alg)
del ctxs

# (b) walk variants on the same index
names = ["rif4 qreg (production)", "rif3 qreg", "rif6 qreg", "rif8 qreg", "rif4 qlds", "rif6 qlds", "rif8 qlds", "rif2 qreg",
         "rif4 qreg, compiler's occupancy", "rif5 qreg"]
for i, nm in enumerate(names):
    os.environ["IDIST_TUNE"] = str(i)
    s = ida.Search()
    try:
        kt = time_ctx(s, 5)
    except Exception as e:  # noqa: BLE001
        emit(what="variant", i=i, name=nm, error=repr(e))
        continue
    same = bool(torch.equal(outs[0], ref_pid) and torch.equal(outs[1].view(torch.int32), ref_d.view(torch.int32))
                and torch.equal(outs[3], ref_ctr))
    med = float(np.median(kt))
    emit(what="variant", i=i, name=nm, ms=kt, median=med, TBps=round(alg / med / 1e9, 3), identical=same)
    del s
os.environ.pop("IDIST_TUNE", None)
for env, nm in (({"IDIST_VISITED": "bitmap"}, "bitmap + Bloom filter, overlap walk, 16 waves/CU"), ({"IDIST_WALK": "classic"}, "on-chip set, classic walk"),
                ({"IDIST_TAB_LOG2": "12"}, "on-chip set of 4096 ids, then bitmap")):
    os.environ.update(env)
    s = ida.Search()
    kt = time_ctx(s, 5)
    same = bool(torch.equal(outs[0], ref_pid) and torch.equal(outs[3], ref_ctr))
    emit(what=nm, ms=kt, TBps=round(alg / float(np.median(kt)) / 1e9, 3), identical=same)
    for k in env:
        os.environ.pop(k)
    del s
# (c) narrow batches
for w in (1, 32, 256, 1024, 4096):
    s = ida.Search()
    kt = time_ctx(s, 8, w)
    emit(what="batch width", nq=w, ms_median=round(float(np.median(kt)), 4), qps=round(w / float(np.median(kt)) * 1e3))
    del s
# (d) ef sweep: on-chip set (spills to the bitmap beyond ~7k visited ids) vs bitmap walk
nd = ref_ctr[:, 0].cpu().numpy()
emit(what="n_dist per query at ef=100", mean=float(nd.mean()), p50=int(np.percentile(nd, 50)), p90=int(np.percentile(nd, 90)),
     p99=int(np.percentile(nd, 99)), max=int(nd.max()), over_7168=int((nd > 7168).sum()))
for ef in (100, 128, 160, 200, 400):
    h.set_ef_search(ef)
    o2 = (torch.empty(nq, ef, dtype=torch.int32, device=dev), torch.empty(nq, ef, dtype=torch.float32, device=dev),
          torch.empty(nq, dtype=torch.int32, device=dev), torch.empty(nq, 3, dtype=torch.int32, device=dev))
    row = {"what": "ef sweep", "ef": ef}
    for env, nm in (({}, "on_chip"), ({"IDIST_VISITED": "bitmap"}, "bitmap")):
        os.environ.update(env)
        s = ida.Search()
        for _ in range(4):
            h.search_batch_device(s, d_q.data_ptr(), nq, o2[0].data_ptr(), o2[1].data_ptr(), o2[2].data_ptr(), o2[3].data_ptr(),
                                  torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        s.check_status()
        row[nm + "_ms"] = round(float(np.median(s.kernel_times_ms(3))), 3)
        row[nm + "_n_dist"] = float(o2[3][:, 0].float().mean().item())
        for k in env:
            os.environ.pop(k)
        del s
    emit(**row)
h.set_ef_search(100)
os.environ["IDIST_VISITED"] = "bitmap"
s = ida.Search()
emit(what="nq=1, bitmap latency walk", ms_median=round(float(np.median(time_ctx(s, 16, 1))), 4))
os.environ.pop("IDIST_VISITED")
