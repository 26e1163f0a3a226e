"""The native all-reduce (csrc/comm.hip) in lockstep: n rank processes on GPU 0, 200 gradient all-reduces (6.78 MB: tail + head kernels) back to back,
microseconds per all-reduce as each rank sees it.  usage: python tools/native_allreduce_bench.py [n ...]   (default 2 3 4 8)"""
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ns = [int(x) for x in sys.argv[1:]] or [2, 3, 4, 8]
for n in ns:
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    tmp = tempfile.mkdtemp()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CBM_NATIVE_BENCH="1", CBM_NATIVE_TIMEOUT_S="60")
    outs = [os.path.join(tmp, f"r{r}.npz") for r in range(n)]
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "native_comm_worker.py"), str(r), str(n), str(port), outs[r]], env=env,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for r in range(n)]
    rc = [p.wait(timeout=600) for p in procs]
    if any(rc):
        print(f"n={n}: a rank failed {rc}")
        continue
    us = [float(np.load(o)["us_per_allreduce"]) for o in outs]
    b = int(np.load(outs[0])["bytes"])
    print(f"native all-reduce, {n} rank processes on one GPU, {b / 1e6:.2f} MB: {np.mean(us):.1f} us per all-reduce (per rank: {[round(u, 1) for u in us]})")
