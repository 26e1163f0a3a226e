#!/usr/bin/env python
"""profiles/pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md prescribes).
usage: pmc_traffic.py <fetch results.db> <write results.db> > traffic.json
traffic_bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB * 1024 — the gfx950 FETCH_SIZE x2 correction."""
import json
import sqlite3
import sys
from collections import defaultdict

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from rocprof_summary import short

# substrings of the shipped kernels' names -> bench.py kernel ids (first match wins)
IDS = [("conv1_fwd_planes_kernel", 0), ("ConvFwd<IgemmTile<64, 64, 32, 2, 2, 4>, 4, 4, 2", 1), ("ConvFwd<IgemmTile<128, 64, 16, 2, 2, 4>, 3, 3, 1", 2),
       ("igemm_dma_kernel<DenseFwd", 3), ("MatWgrad<IgemmTile<128, 32", 4), ("DenseDgrad", 5), ("MatWgrad<IgemmTile<128, 256", 6), ("Conv3DgradPos", 7),
       ("conv3_wgrad_frames_kernel", 8), ("Conv2DgradMergedPos", 9), ("conv2_wgrad_frames_kernel", 10), ("conv1_wgrad_frames_kernel", 11)]


def avg(path, counter):
    con = sqlite3.connect(path)
    acc = defaultdict(list)
    for name, did, val in con.execute("select name, dispatch_id, sum(counter_value) from pmc_events where counter_name = ? group by name, dispatch_id", (counter,)):
        acc[name].append(val)
    return {n: sum(v) / len(v) for n, v in acc.items()}


fetch, write = avg(sys.argv[1], "FETCH_SIZE"), avg(sys.argv[2], "WRITE_SIZE")
out = {}
for name, f in fetch.items():
    for pat, kid in IDS:
        if pat in name and name in write and str(kid) not in out:
            out[str(kid)] = {"kernel": short(name), "FETCH_SIZE_KiB_raw": round(f, 1), "WRITE_SIZE_KiB_raw": round(write[name], 1),
                             "traffic_bytes": int((2 * f + write[name]) * 1024)}
print(json.dumps(dict(sorted(out.items(), key=lambda kv: int(kv[0]))), indent=1))
