"""Split actor/learner path between REAL processes on one GPU: an actor role process and a learner role process (both pinned to cuda:0)
exchange rollout shards and parameters exactly as they would across two GPUs — the learner exports HIP IPC handles of its ring fields, the
actor maps them and writes its shards with strided 2-D copies on its io stream, learner 0 writes parameters into the actor's version buffers,
'landed' messages travel over the TCP store.  With one learner nothing is re-sharded, so the run must reproduce the ordinary single-process
a0-l0 run bit for bit.  (More than one learner needs RCCL between distinct GPUs; those shapes run on CPU in tests/test_host_cpu.py.)"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


@pytest.mark.parametrize("algo", ["ppo", "impala"])
def test_split_processes_equal_single_process(tmp_path, algo):
    from cleanba_amd.args import parse_args
    from cleanba_amd.trainer import train
    os.chdir(str(tmp_path))
    E, T, updates, threads = 8, 8, 4, 2
    base = ["--local-num-envs", str(E), "--num-actor-threads", str(threads), "--num-steps", str(T), "--env-backend", "device", "--network", "nature",
            "--total-timesteps", str(updates * E * threads * T), "--log-frequency", "1000", "--update-epochs", "1"]
    ref = train(parse_args(base, algo), algo)

    port = _free_port()
    env = dict(os.environ, CBM_TEST_TMP=str(tmp_path))
    outs, procs = [], []
    for r in range(2):
        out = os.path.join(str(tmp_path), f"split_{r}.npz")
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "gpu_split_worker.py"), str(r), "2", str(port), out, algo, str(E), str(T),
                                       str(updates), str(threads)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        o, _ = p.communicate(timeout=600)
        assert p.returncode == 0, o.decode()[-3000:]
    a, l0 = (np.load(o) for o in outs)
    assert str(a["role"]) == "actor" and str(l0["role"]) == "learner0"
    assert int(l0["updates"]) == updates == ref["updates"]
    assert np.array_equal(l0["params"], ref["params"])
    assert np.array_equal(a["params"], l0["params"])     # the actor holds the version learner 0 wrote last


def test_ship_shard_is_the_reference_column_split():
    """cbm_actor_ship_shard against numpy: jnp.split(x, L, axis=1) per field, shard li of thread s lands in columns [port*E/L, +E/L) of the
    learner's hstack (ppo:358-363,587).  Two contexts in one process: the destination pointers are the learner context's own buffers."""
    import cleanba_amd.lib as L
    import cleanba_amd.model as M
    import cleanba_amd.prng as prng
    E, T, S, NL = 12, 5, 2, 3
    El = E // NL
    cfg = L.default_config(L.ALGO_IMPALA)
    cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps, cfg.num_minibatches = E, S, T, 2
    actor = L.Context(cfg)
    lcfg = L.default_config(L.ALGO_IMPALA)
    lcfg.local_num_envs, lcfg.num_actor_slots, lcfg.num_steps, lcfg.num_minibatches = El, S, T, 2
    learner = L.Context(lcfg)
    key = prng.prng_key(3)
    key, nk, ak, ck = prng.split(key, 4)
    actor.set_params(M.init_nature_params(18, nk, ak, ck))
    fields = {"obs": np.uint8, "actions": np.int32, "logits": np.float32, "rewards": np.float32, "dones": np.uint8, "firststeps": np.uint8}
    for s in range(S):
        actor.actor_set_key(s, key)
        actor.actor_env_reset_device(s, 5 + s)
        actor.actor_begin_rollout(s, True)
        actor.actor_rollout_device(s, T + 1)
        actor.actor_commit(s)
    li = 1
    pr = L.PeerRing()
    for f in fields:
        setattr(pr, f, learner.buffer(f, 0)[0])
    for s in range(S):
        actor.actor_ship_shard(s, 0, li, NL, pr, S * El, s * El)
    actor.io_sync()
    for f, dt in fields.items():
        src = actor.read(f, dt).reshape(T + 1, S * E, -1)
        dst = learner.read(f, dt).reshape(T + 1, S * El, -1)
        want = np.concatenate([src[:, s * E + li * El:s * E + (li + 1) * El] for s in range(S)], axis=1)
        assert np.array_equal(dst, want), f
    assert learner.read("obs", np.uint8).any()
    actor.close()
    learner.close()
