#!/usr/bin/env python3
"""The headline kernel alone, for the PMC passes behind ``roofline.traffic`` (bench.py): the k=3 dilated Conv1d forward of
highwayConv C=256 (M=512), L=325, B=32 in the current arithmetic mode, launched as the training step launches it (resident
pre-split weights, the input's scale list prepared, 20 rotating operand sets).  Run under
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d DIR -- python3 tools/pmc_headline.py
and again with WRITE_SIZE (they do not fit one pass); tools/pmc_headline.py --json F W prints the profiles/traffic.json entry."""
import ctypes, csv, glob, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def entry(fetch_dir, write_dir, profile):
    def avg(d, name):
        path = (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0]
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Counter_Name"] == name and "gemm_nn_bf3_kernel<3, 1, 7" in r["Kernel_Name"]]
        return sum(v) / len(v), len(v)
    f, n = avg(fetch_dir, "FETCH_SIZE")
    w, _ = avg(write_dir, "WRITE_SIZE")
    corr = 2 * 512 * 256 * 3 * 2 / 1024.0 / 2          # hi + lo planes of the (512, 256, 3) weight = 1.5 MiB, fetched by 16-byte-per-lane loads: counted at half
    # everything the kernel's loads, scaling, tile and launch choice come from: a change in any of them withholds the figure
    src = ["spoofsv_amd/csrc/conv_nn.hip", "spoofsv_amd/csrc/bf3_common.h", "spoofsv_amd/csrc/bf3_tuning.h", "spoofsv_amd/csrc/ssv_common.h", "spoofsv_amd/csrc/api.hip"]
    h = hashlib.sha256()
    for s in src:
        h.update(open(os.path.join(ROOT, s), "rb").read())
    return {"kernel": "gemm_nn_bf3_kernel<3,1,7,0,1,16>", "shape": "Conv1d fwd B=32 C=256->512 L=325 k=3 (+ column statistics of the output)", "launches": n,
            "fetch_kib": round(f, 1), "write_kib": round(w, 1), "fetch_correction_kib": corr, "traffic_bytes": round((f + w) * 1024.0, 1), "traffic_bytes_corrected": round((f + w + corr) * 1024.0, 1),
            "profile": profile,
            "note": "traffic = raw FETCH_SIZE + WRITE_SIZE per launch of the kernel alone (tools/pmc_headline.py), separate --pmc passes; traffic_corrected adds 768 KiB on the ASSUMPTION (MI355X_MICROARCH.md, not re-verified here) that gfx950 tallies 16-byte-per-lane loads -- the 1.5 MiB of pre-split weight fragments -- at half their bytes",
            "sources": src, "sources_sha16": h.hexdigest()[:16]}


if len(sys.argv) > 1 and sys.argv[1] == "--json":
    print(json.dumps(entry(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "profiles/round4_bench_kernel_stats.txt"), indent=1))
    sys.exit(0)

import torch
from spoofsv_amd import _lib, ops, resident
dev = "cuda:0"
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
B, C, L, k, nset = 32, 256, 325, 3, 20
xs = [torch.randn(B, C, L, device=dev) for _ in range(nset)]
hs = [torch.empty(B, 2 * C, L, device=dev) for _ in range(nset)]
ys = [torch.empty(B, C, L, device=dev) for _ in range(nset)]
f16 = _lib.precision() == 2
xa = [ops.amax_of(x) if f16 else None for x in xs]
w = torch.randn(2 * C, C, k, device=dev) * 0.05
bias = torch.randn(2 * C, device=dev)
g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
rw = resident.ResidentWeights([w]); rw.refresh(st)
nb = _lib.query("ssv_highway_conv1d_fwd_workspace", B, C, L, k)
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
stats = torch.empty(B, 4, L, device=dev)
for rep in range(2):
    for i in range(nset):       # the highway forward = this conv (with the column statistics the step asks for) + the streaming LayerNorm / gate
        _lib.call("ssv_highway_conv1d_fwd", P(xs[i]), C * L, P(xa[i]), ops._AMAX_PIECES if f16 else 0, P(w), resident.lookup(w), P(bias), P(g), P(b), P(g), P(b),
                  P(hs[i]), P(stats), P(ys[i]), C * L, None, B, C, L, k, 1, 1, P(ws), nb, st)
torch.cuda.synchronize()
print("ok")
