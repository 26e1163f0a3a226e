"""CPU stand-in for cleanba_amd.lib.Context with the SAME method surface, built on the oracle.

Test infrastructure only (lives under tests/, imports oracle/): it lets the host logic of
cleanba_amd.trainer — actor threads, ring/sequence hand-off, policy-version skew, per-minibatch gradient
all-reduce across processes — run on a CPU box with the gloo backend, where the HIP library cannot create a
context.  It mirrors csrc/api.hip's orchestration (row layout [T+1][B], ring depth, params versions)."""
import threading
from multiprocessing import shared_memory

import numpy as np

import oracle

FRAME = 4 * 84 * 84


class _Shm:
    """numpy arrays in named shared memory: the CPU stand-in for device buffers another process maps through a HIP IPC handle."""

    def __init__(self):
        self.owned, self.mapped = [], []

    def alloc(self, dtype, *shape):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        m = shared_memory.SharedMemory(create=True, size=max(n, 8))
        self.owned.append(m)
        a = np.ndarray(shape, dtype, buffer=m.buf)
        a[...] = 0
        return a, (m.name, tuple(shape), np.dtype(dtype).str)

    def open(self, handle):
        name, shape, dt = handle
        m = shared_memory.SharedMemory(name=name)
        try:   # Python 3.10 registers ATTACHED segments with this process's resource tracker too, which then unlinks the owner's segment
            from multiprocessing import resource_tracker
            resource_tracker.unregister(m._name, "shared_memory")
        except Exception:  # noqa: BLE001
            pass
        self.mapped.append(m)
        return np.ndarray(shape, np.dtype(dt), buffer=m.buf)

    def close(self):
        for m in self.mapped:
            try:
                m.close()
            except BufferError:
                pass
        for m in self.owned:
            try:
                m.close()
            except BufferError:
                pass
            try:
                m.unlink()
            except FileNotFoundError:
                pass
        self.owned, self.mapped = [], []


class OracleEngine:
    def __init__(self, cfg):
        self.cfg = cfg
        assert int(getattr(cfg, "network", 0)) == 0, "OracleEngine restates the Nature-CNN trainer (the ResNet torso is checked per call: oracle.resnet_*)"
        self.ppo = cfg.algo == 0
        # conv1 of learner-size passes the way the configuration asks for it: cbm_config.conv1_fp32_chain bit 0 clear (the default) = exact products summed by the
        # measured rule of the bf16 matrix instruction (oracle.set_conv1_exact), set = the fmaf chain.  Process-wide, like oracle.set_threads.
        oracle.set_conv1_exact((int(getattr(cfg, "conv1_fp32_chain", 3)) & 1) == 0)
        self.A, self.E, self.S = cfg.num_actions, cfg.local_num_envs, cfg.num_actor_slots
        self.T = cfg.num_steps
        self.asyncB = int(getattr(cfg, "async_batch_size", 0) or 0)
        if self.asyncB:   # legacy async mode: rows of asyncB samples, num_steps*async_update of them (mirrors csrc/api.hip)
            assert self.ppo and self.S == 1 and self.E % self.asyncB == 0
            self.NE, self.E, self.T = self.E, self.asyncB, cfg.num_steps * (self.E // self.asyncB)
        self.B = self.E * self.S
        self.T1 = self.T + 1
        self.nmb = cfg.num_minibatches
        self.accum = max(1, cfg.grad_accum_steps)
        self.nmicro = self.nmb * self.accum
        self.epochs = cfg.update_epochs if self.ppo else 1
        self.depth = cfg.ring_depth
        self.P = oracle.nature_param_count(self.A)
        self.shm = _Shm()
        self.ring, self.ring_handles = [], []
        spec = dict(obs=(np.uint8, self.T1, self.B, 4, 84, 84), actions=(np.int32, self.T1, self.B), logprobs=(np.float32, self.T1, self.B),
                    values=(np.float32, self.T1, self.B), rewards=(np.float32, self.T1, self.B), logits=(np.float32, self.T1, self.B, self.A),
                    dones=(np.uint8, self.T1, self.B), firststeps=(np.uint8, self.T1, self.B), env_ids=(np.int32, self.T1, self.B))
        for _ in range(self.depth):
            arrs, hs = {}, {}
            for k, sp in spec.items():
                arrs[k], hs[k] = self.shm.alloc(*sp)
            self.ring.append(arrs)
            self.ring_handles.append(hs)
        self.apv, self.apv_handles = [], []          # the three versioned actor parameter buffers (version v lives in v % 3)
        for _ in range(3):
            a, h = self.shm.alloc(np.float32, self.P)
            self.apv.append(a)
            self.apv_handles.append(h)
        self.group, self.nranks, self.aborted = None, 1, False
        self.params = np.zeros(self.P, np.float32)
        self.grads = np.zeros(self.P, np.float32)
        self.m = np.zeros(self.P, np.float32)
        self.v = np.zeros(self.P, np.float32)
        self.gacc = np.zeros(self.P, np.float32)
        self.keys = [np.zeros(2, np.uint32) for _ in range(self.S)]
        self.slot = [dict(t=0, rollout=0, ring=0, pver=0) for _ in range(self.S)]
        self.committed = [0] * self.S
        self.updates_done = 0
        self.cv = threading.Condition()
        self.stats = []
        self.h = True

    # ---- params
    def set_params(self, p):
        self.params = np.ascontiguousarray(p, np.float32).copy()
        for a in self.apv:
            a[:] = self.params
        self.m[:] = 0
        self.v[:] = 0

    def get_params(self):
        return self.params.copy()

    def sync(self):
        pass

    def close(self):
        self.ring, self.apv = [], []
        self.shm.close()

    def abort(self):
        with self.cv:
            self.aborted = True
            self.cv.notify_all()

    def _wait(self, pred):
        with self.cv:
            self.cv.wait_for(lambda: self.aborted or pred())
            if self.aborted:
                raise RuntimeError("context aborted")

    # ---- the learner communicator: gloo over a private TCP store stands in for RCCL (same call order as HipEngine)
    def wants_comm_at_world_one(self):
        return False

    def comm_unique_id(self):
        import socket
        from torch.distributed import TCPStore
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        self._comm_store = TCPStore("127.0.0.1", port, -1, True, wait_for_workers=False)
        return f"127.0.0.1:{port}".encode()

    def comm_init(self, uid, nranks, rank):
        import datetime
        import torch.distributed as dist
        from torch.distributed import TCPStore
        host, port = uid.decode().split(":")
        store = getattr(self, "_comm_store", None) or TCPStore(host, int(port), -1, False)
        dist.init_process_group("gloo", store=store, rank=rank, world_size=nranks, timeout=datetime.timedelta(seconds=600))
        self.group, self.nranks = dist, nranks

    def _allreduce(self, arr):
        """pmean's SUM half; the caller divides (like grad_div in the HIP optimizer kernel)."""
        if self.group is None:
            return 1.0
        import torch
        t = torch.from_numpy(arr)
        self.group.all_reduce(t, op=self.group.ReduceOp.SUM)
        return float(self.nranks)

    # ---- actor
    def actor_set_key(self, s, key):
        self.keys[s] = np.ascontiguousarray(key, np.uint32).copy()

    def actor_get_key(self, s):
        return self.keys[s].copy()

    def actor_begin_rollout(self, s, concurrency):
        sl = self.slot[s]
        sl["rollout"] += 1
        u = sl["rollout"]
        need = (u - 2 if u >= 2 else 0) if concurrency else u - 1
        self._wait(lambda: self.updates_done >= need and self.updates_done >= u - self.depth)
        sl["pver"], sl["ring"], sl["t"] = need, (u - 1) % self.depth, 0
        if u >= 2 and not self.ppo:
            prev, cur = self.ring[(u - 2) % self.depth], self.ring[sl["ring"]]
            c = slice(s * self.E, (s + 1) * self.E)
            for k in cur:
                cur[k][0, c] = prev[k][self.T, c]
            sl["t"] = 1
        return need + 1

    def actor_step_host(self, s, obs, done, firststep=None, reward_with_obs=None, actions_out=None):
        sl = self.slot[s]
        R, t, c = self.ring[sl["ring"]], sl["t"], slice(s * self.E, (s + 1) * self.E)
        R["obs"][t, c] = obs
        R["dones"][t, c] = done
        if firststep is not None:
            R["firststeps"][t, c] = firststep
        if reward_with_obs is not None:
            R["rewards"][t, c] = reward_with_obs
        logits, value = oracle.nature_forward(self.apv[sl["pver"] % 3].copy(), self.A, R["obs"][t, c], ksplit=self.cfg.actor_dense_ksplit)
        a, lp, self.keys[s] = oracle.sample_actions(logits, self.keys[s])
        R["actions"][t, c] = a
        if self.ppo:
            R["logprobs"][t, c], R["values"][t, c] = lp, value
        else:
            R["logits"][t, c] = logits
        sl["t"] += 1
        if actions_out is not None:
            actions_out[:] = a
            return actions_out
        return a

    def actor_step_async(self, s, obs, reward, done, env_id, actions_out=None):
        sl = self.slot[s]
        R, t = self.ring[sl["ring"]], sl["t"]
        assert t < self.T, "rollout overrun"
        R["obs"][t], R["rewards"][t], R["dones"][t], R["env_ids"][t] = obs, reward, done, env_id
        logits, value = oracle.nature_forward(self.apv[sl["pver"] % 3].copy(), self.A, R["obs"][t], ksplit=self.cfg.actor_dense_ksplit)
        a, lp, self.keys[s] = oracle.sample_actions(logits, self.keys[s])
        R["actions"][t], R["logprobs"][t], R["values"][t] = a, lp, value
        sl["t"] += 1
        if actions_out is not None:
            actions_out[:] = a
            return actions_out
        return a

    def actor_record_host(self, s, reward):
        sl = self.slot[s]
        self.ring[sl["ring"]]["rewards"][sl["t"] - 1, s * self.E:(s + 1) * self.E] = reward

    def actor_commit(self, s, next_obs=None, next_done=None):
        sl = self.slot[s]
        if next_obs is not None:
            R, c = self.ring[sl["ring"]], slice(s * self.E, (s + 1) * self.E)
            R["obs"][self.T, c] = next_obs
            R["dones"][self.T, c] = next_done
        with self.cv:
            self.committed[s] = sl["rollout"]
            self.cv.notify_all()

    def actor_episode_stats(self, s):
        return 0.0, 0.0

    # ---- split topologies (same surface as HipEngine)
    def actor_ring_index(self, s):
        return self.slot[s]["ring"]

    def export_ring(self, fields, tag=-1):
        return {"cols": self.B, "entries": [{f: self.ring_handles[r][f] for f in fields} for r in range(self.depth)]}

    def open_peer_ring(self, desc, what=""):
        return [{f: self.shm.open(h) for f, h in entry.items()} for entry in desc["entries"]]

    def actor_ship_shard(self, slot, ring, li, n_learners, peer_ring, dst_cols, dst_col0):
        El = self.E // n_learners
        c0 = slot * self.E + li * El
        R = self.ring[ring]
        for f, dst in peer_ring.items():
            assert dst.shape[1] == dst_cols
            dst[:, dst_col0:dst_col0 + El] = R[f][:, c0:c0 + El]

    def io_sync(self):
        pass

    def export_actor_params(self, tag=-1):
        return list(self.apv_handles)

    def open_peer_params(self, handles, what=""):
        return [self.shm.open(h) for h in handles]

    def params_push(self, peer_versions):
        peer_versions[self.updates_done % 3][:] = self.params

    def params_mark_published(self):
        with self.cv:
            self.updates_done += 1
            self.cv.notify_all()

    def ingest_begin(self, s):
        sl = self.slot[s]
        sl["rollout"] += 1
        u = sl["rollout"]
        self._wait(lambda: self.updates_done >= u - self.depth)
        sl["ring"] = (u - 1) % self.depth
        return sl["ring"]

    def ingest_commit(self, s):
        with self.cv:
            self.committed[s] = self.slot[s]["rollout"]
            self.cv.notify_all()

    def get_actor_params(self):
        return self.apv[self.updates_done % 3].copy()

    # ---- learner
    def _cur(self):
        return self.ring[self.updates_done % self.depth]

    def learner_wait(self):
        v = self.updates_done + 1
        self._wait(lambda: all(c >= v for c in self.committed))

    def learner_prepare(self, key):
        if self.asyncB:
            R = self._cur()
            self.adv, self.tgt = oracle.gae_async(R["env_ids"][:self.T], R["rewards"][:self.T], R["values"][:self.T], R["dones"][:self.T], self.NE,
                                                  self.cfg.gamma, self.cfg.gae_lambda)
        elif self.ppo:
            R = self._cur()
            _, nv = oracle.nature_forward(self.params, self.A, R["obs"][self.T], ksplit=self.cfg.actor_dense_ksplit)
            adv, tgt = oracle.gae(R["rewards"][:self.T], R["values"][:self.T], R["dones"][:self.T], nv, R["dones"][self.T], self.cfg.gamma,
                                  self.cfg.gae_lambda)
            self.adv = oracle.advnorm(adv, self.nmb) if self.cfg.norm_adv else adv
            self.tgt = tgt
        self.stats = []
        return np.ascontiguousarray(key, np.uint32).copy()

    def learner_epoch_begin(self, key):
        key = np.ascontiguousarray(key, np.uint32)
        if self.ppo:
            ks = oracle.split(key, 2)
            key, sub = ks[0], ks[1]
            self.perm = oracle.permutation(sub, self.T * self.B)
        return key.copy()

    def learner_minibatch_grad(self, e, mb):
        R = self._cur()
        c = self.cfg
        if self.ppo:
            N = self.T * self.B
            MB = N // self.nmicro
            idx = self.perm[mb * MB:(mb + 1) * MB]
            fo = R["obs"][:self.T].reshape(N, 4, 84, 84)
            mb_adv = self.adv.reshape(N)[idx]
            if self.asyncB and self.cfg.norm_adv:
                mb_adv = oracle.mb_advnorm(mb_adv)   # naturecnn:540-541
            st, g, _, _ = oracle.ppo_loss_grad(self.params, self.A, fo, idx, R["actions"][:self.T].reshape(N)[idx],
                                               R["logprobs"][:self.T].reshape(N)[idx], mb_adv, self.tgt.reshape(N)[idx],
                                               c.clip_coef, c.ent_coef, c.vf_coef)
        else:
            Bm = self.B // self.nmicro
            cs = slice(mb * Bm, (mb + 1) * Bm)
            st, g = oracle.impala_loss_grad(self.params, self.A, R["obs"][:, cs].reshape(-1, 4, 84, 84), None, self.T1, Bm, R["logits"][:, cs],
                                            R["actions"][:, cs], R["rewards"][:, cs], R["dones"][:, cs], R["firststeps"][:, cs], c.gamma,
                                            c.vf_coef, c.ent_coef)
        self.grads[:] = g
        self.stats.append(st)

    def learner_accumulate(self, mini_step, grad_div=1.0):
        g = self.grads / np.float32(grad_div) if grad_div != 1.0 else self.grads
        a = (g - self.gacc) / np.float32(mini_step + 1) + self.gacc   # optax.MultiSteps running mean
        if mini_step == self.accum - 1:
            self.grads[:] = a
            self.gacc[:] = 0
        else:
            self.gacc[:] = a

    def learner_optimizer_step(self, lr, bc1, bc2, grad_div=1.0):
        g = self.grads / np.float32(grad_div) if grad_div != 1.0 else self.grads
        c = self.cfg
        if self.ppo:
            oracle.adam_step(self.params, g, self.m, self.v, c.max_grad_norm, lr, c.adam_b1, c.adam_b2, c.adam_eps, bc1=bc1, bc2=bc2)
        else:
            oracle.rmsprop_step(self.params, g, self.m, c.max_grad_norm, lr, c.rms_decay, c.rms_eps)

    def learner_finish(self, n_rows, want_stats=True):
        v = self.updates_done + 1
        stats = np.array(self.stats, np.float32) if want_stats else None
        if want_stats and self.group is not None:      # pmean of the loss statistics over the learners (ppo:649-653)
            stats = stats / np.float32(self._allreduce(stats))
        with self.cv:
            self.apv[v % 3][:] = self.params
            self.updates_done = v
            self.cv.notify_all()
        return stats

    def learner_update(self, key, lrs, bc1, bc2, want_stats=True):
        key = self.learner_prepare(key)
        i = 0
        for e in range(self.epochs):
            key = self.learner_epoch_begin(key)
            for mb in range(self.nmicro):
                self.learner_minibatch_grad(e, mb)
                div = self._allreduce(self.grads)          # pmean(grads) over the learner communicator (ppo:628)
                if self.accum > 1:
                    self.learner_accumulate(mb % self.accum, div)
                    if mb % self.accum != self.accum - 1:
                        continue
                    div = 1.0
                self.learner_optimizer_step(float(lrs[i]), float(bc1[i]), float(bc2[i]), div)
                i += 1
        return key, self.learner_finish(len(lrs), want_stats)
