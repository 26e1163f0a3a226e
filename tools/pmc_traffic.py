#!/usr/bin/env python
"""profiles/pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md prescribes).
usage: pmc_traffic.py <fetch results.db> <write results.db> [names.json] > traffic.json
traffic_bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB * 1024 — the gfx950 FETCH_SIZE x2 correction.
names.json (tools/microbench.py --names-out): {id: "<kernel symbol> <functor type>"} as the LIBRARY reports what it launched for every id
(cbm_profile_kernel_name); with it the id <-> kernel mapping is the library's own, and every entry carries the full profiled kernel name and the
git revision so that bench.py can refuse numbers that belong to another kernel."""
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


def git_rev():
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        return subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:  # noqa: BLE001  (the GPU box's snapshot has no .git: GIT_REV, or the .build_rev file tools/r06_artifacts.sh's caller writes before gpurun)
        if os.environ.get("GIT_REV"):
            return os.environ["GIT_REV"]
        try:
            return open(os.path.join(root, ".build_rev")).read().strip() or "unknown"
        except OSError:
            return "unknown"


def same_kernel(launched, full):
    full = full.replace(" ", "")
    return bool(launched) and all(part.replace(" ", "") in full for part in launched.split(" ", 1))


fetch, write = avg(sys.argv[1], "FETCH_SIZE"), avg(sys.argv[2], "WRITE_SIZE")
names = json.load(open(sys.argv[3])) if len(sys.argv) > 3 else None
rev = git_rev()
out = {}
for name, f in fetch.items():
    if name not in write:
        continue
    if names is not None:
        kids = [int(k) for k, launched in names.items() if same_kernel(launched, name)]
    else:
        kids = [kid for pat, kid in IDS if pat in name][:1]
    for kid in kids:
        if str(kid) not in out:
            out[str(kid)] = {"kernel": short(name), "kernel_full": name, "git": rev, "FETCH_SIZE_KiB_raw": round(f, 1),
                             "WRITE_SIZE_KiB_raw": round(write[name], 1), "traffic_bytes": int((2 * f + write[name]) * 1024)}
print(json.dumps(dict(sorted(out.items(), key=lambda kv: int(kv[0]))), indent=1))
