"""The driver's multi-GPU command shapes, exercised on ONE GPU: `python bench.py --gpus N` launches its own N ranks (bench.self_launch) — the
exact path the SCALE run takes — and `--topology a0-l1` runs one process per role.  CBM_FORCE_DEVICE=0 puts every rank on GPU 0 and
CBM_COMM_LOOPBACK=1 replaces RCCL (which needs distinct devices per rank) by the library's self-test communicator; everything else — the
rendezvous, the per-rank contexts, the barriers and max-over-ranks timing through cbm_comm_*, the IPC shard hand-off, the ONE JSON line on the
parent's stdout — is the code the 8-GPU node runs."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, loopback=True):
    env = dict(os.environ, CBM_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if loopback:
        env["CBM_COMM_LOOPBACK"] = "1"
    else:
        env.pop("CBM_COMM_LOOPBACK", None)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args, "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-host-env"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    out = p.stdout.decode().strip()
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [ln for ln in out.split("\n") if ln.strip()]
    assert len(lines) == 1, f"stdout must carry exactly one line, got {len(lines)}:\n{out[-2000:]}"
    return json.loads(lines[0])


def test_bench_gpus_2_self_launch_emits_one_line():
    d = _bench("--gpus", "2")
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and "error" not in d
    assert d["allreduce"]["ranks"] == 2 and d["allreduce"]["minibatches"] > 0
    assert d["value"] > 0 and abs(d["value"] - 2 * 120 * 128 * 2 / (d["ms_per_step"] * 2 / 1e3)) / d["value"] < 1e-3   # whole-job: both ranks' env-steps
    assert d["config"]["parallelism"] == "dp2" and d["roofline"]["kernel"]


def test_bench_topology_a0_l1_emits_one_line():
    d = _bench("--gpus", "2", "--topology", "a0-l1")
    assert d["role_processes"] == 2 and d["updates"] == 3 and d["value"] > 0 and "error" not in d
    assert d["config"]["parallelism"] == "1x(actor1+dp1)"


def test_bench_gpus_4_native_allreduce_then_the_configs3_topology_line():
    """The SCALE command at N = 4 with every rank on GPU 0 and NO loopback: the data-parallel line runs on the library's native all-reduce (four
    ranks, four different gradients — RCCL cannot put two ranks on one device), then the same four processes run BASELINE configs[3]
    (`a0-l1,2,3`) and its line rides along as `baseline_config`."""
    d = _bench("--gpus", "4", loopback=False)
    assert d["n_gpus"] == 4 and "error" not in d and d["value"] > 0
    assert d["allreduce"]["backend"] == "native" and d["allreduce"]["ranks"] == 4
    bc = d["baseline_config"]
    assert bc.get("value") and bc["value"] > 0, bc
    assert "a0-l1,2,3" in bc["config"] and bc["allreduce"]["backend"] == "native" and bc["allreduce"]["ranks"] == 3


def test_bench_gpus_8_native_allreduce_then_the_configs4_topology_line():
    """The SCALE command at N = 8 on one GPU: eight data-parallel ranks on the native all-reduce, then BASELINE configs[4] (`2x(a0-l1,2,3)`, Atari-57
    synthetic frame mix: two actor + six learner role processes, one six-rank communicator) rides along as `baseline_config`."""
    d = _bench("--gpus", "8", loopback=False)
    assert d["n_gpus"] == 8 and "error" not in d and d["value"] > 0
    assert d["allreduce"]["backend"] == "native" and d["allreduce"]["ranks"] == 8
    bc = d["baseline_config"]
    assert bc.get("value") and bc["value"] > 0, bc
    assert "2x(a0-l1,2,3)" in bc["config"] and "Atari-57" in bc["config"] and bc["allreduce"]["backend"] == "native" and bc["allreduce"]["ranks"] == 6
