import copy, os, sys, threading
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from test_gpu_split import LoopbackDist, _Shared
from cleanba_amd.args import parse_args
from cleanba_amd.trainer import train
os.chdir("/tmp")
algo = sys.argv[1]
for updates in (3,) * int(os.environ.get('REPS', '6')):
    E, T = 8, 8
    base = ["--local-num-envs", str(E), "--num-actor-threads", "2", "--num-steps", str(T), "--env-backend", "device", "--network", "nature",
            "--total-timesteps", str(updates * E * 2 * T), "--log-frequency", "1000", "--update-epochs", "1"]
    r1 = train(parse_args(base, algo), algo)["params"]
    r2 = train(parse_args(base, algo), algo)["params"]
    split_argv = base + ["--distributed", "--actor-device-ids", "0", "--learner-device-ids", "1"]
    shared, results = _Shared(), {}
    def run(rank):
        results[rank] = train(parse_args(split_argv, algo), algo, rendezvous=(2, rank, 0, None, None), dist_module=LoopbackDist(shared, rank))
    ths = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(2)]
    [t.start() for t in ths]; [t.join(60) for t in ths]
    if len(results) < 2:
        print('split run failed/hung', flush=True); os._exit(1)
    sp = results[1]["params"]
    print(algo, "updates", updates, "ref deterministic", np.array_equal(r1, r2), "split==ref", np.array_equal(sp, r1), "maxdiff", np.abs(sp - r1).max(),
          "actor==learner", np.array_equal(results[0]["params"], sp), flush=True)
