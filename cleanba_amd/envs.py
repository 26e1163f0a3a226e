"""Environment adapters behind the envpool gym API the reference uses (make_env ppo:126-146):
`reset()`, `step(actions)`, `async_reset()`, `recv()`, `send(actions, env_id)`, `action_space.n`,
`observation_space.sample()`, `spec.config.max_episode_steps`, `close()`; info carries `env_id`,
`reward` (unclipped), `terminated`, `elapsed_step`, `TimeLimit.truncated` (ppo:321-340).

SyntheticAtariEnv is the host twin of the device env compiled into libcleanba_mi.so: pure-integer
Breakout-shaped 84x84x4 frame stacks, identical bytes on CPU and GPU.  With `env_backend=envpool`
and envpool installed, the real ALE env is returned instead (not available in this image).
"""
from types import SimpleNamespace

import ctypes as C

import numpy as np

from . import lib as L

ATARI_MAX_FRAMES = int(108000 / 4)  # ppo:121-123


class _Space:
    def __init__(self, n=None, shape=None):
        self.n, self.shape = n, shape

    def sample(self):
        if self.n is not None:
            return np.random.randint(self.n)
        return np.random.randint(0, 256, size=self.shape, dtype=np.uint8)


def is_atari57_mix(env_id):
    """`--env-id Atari57Mix-v5`: env e plays preset e % 57 of the synthetic family (BASELINE configs[4]; benchmark.sh:5 lists the 57 games)."""
    return str(env_id).lower().startswith("atari57")


class SyntheticAtariEnv:
    def __init__(self, env_id="Breakout-v5", num_envs=8, seed=1, max_episode_steps=ATARI_MAX_FRAMES, num_actions=18, batch_size=None,
                 reuse_buffers=False, **_):
        self.env_id, self.num_envs, self.seed = env_id, int(num_envs), int(seed)
        # reuse_buffers=False (default, the envpool contract): every step() returns arrays nobody else writes again — a reference-style rollout keeps
        # 128 steps of them (ppo:329-342).  True (what the trainer's own loop asks for: it copies a step's results into the ring at once): results
        # live in OBS_RING rotating slots and are overwritten OBS_RING - 1 steps later.
        self.reuse_buffers = bool(reuse_buffers)
        self.batch_size = int(batch_size or num_envs)   # envpool async mode: recv() hands back this many envs (naturecnn:119-125)
        assert 0 < self.batch_size <= self.num_envs
        self.action_space = _Space(n=num_actions)
        self.observation_space = _Space(shape=(4, 84, 84))
        self.single_action_space, self.single_observation_space = self.action_space, self.observation_space
        self.is_vector_env = True
        self.spec = SimpleNamespace(config=SimpleNamespace(max_episode_steps=max_episode_steps))
        self._st = None
        self._obs = None
        self._pending = None
        self._ring = None
        self._env_id = np.arange(self.num_envs, dtype=np.int32)

    def _info(self, reward, terminated, elapsed):
        return {"env_id": self._env_id, "reward": reward.copy(), "terminated": terminated.astype(np.int32),
                "elapsed_step": elapsed, "TimeLimit.truncated": elapsed >= self.spec.config.max_episode_steps}

    def reset(self):
        self._st, self._obs = L.synth_env_reset_host(self.seed, self.num_envs, atari57_mix=is_atari57_mix(self.env_id))
        self._ring = None   # (step() caches the addresses of the state block and of the current observations: both are new now)
        return self._obs.copy()

    OBS_RING = 4   # observation buffers handed out round robin

    def step(self, actions):
        # Like envpool's state-buffer queue, the results of a step land in one of a few PRE-ALLOCATED slots (a different array object and address
        # than the previous step's; a slot's content stays valid for OBS_RING - 1 further steps): no 3.4 MB allocation — mmap, page faults,
        # munmap — on the step's critical path, which two actor threads of one process would serialise on the kernel's address-space lock, and the
        # slots' addresses are taken once (a numpy -> pointer conversion is 1-3 us of GIL time, a step passes ten).
        n = self.num_envs

        def slot():
            arrs = (np.empty_like(self._obs), np.zeros(n, np.float32), np.zeros(n, np.uint8), np.zeros(n, np.uint8), np.zeros(n, np.int32))
            ptrs = tuple(a.ctypes.data for a in arrs)
            arrs[0].flags.writeable = False   # handed out read-only (the env itself writes through the address): it is next step's older planes
            return arrs, ptrs
        if self._ring is None:
            self._ring = [slot() for _ in range(self.OBS_RING)] if self.reuse_buffers else []
            self._ring_i = 0
            self._fn = L.load().cbm_synth_env_step_host_to
            self._st_p = C.addressof(self._st)
            self._seed32 = int(self.seed) & 0xFFFFFFFF
            self._obs_p = self._obs.ctypes.data
        if self.reuse_buffers:
            (out, r, d, term, el), (out_p, r_p, d_p, term_p, el_p) = self._ring[self._ring_i]
            self._ring_i = (self._ring_i + 1) % self.OBS_RING
        else:
            (out, r, d, term, el), (out_p, r_p, d_p, term_p, el_p) = slot()
        actions = np.asarray(actions)   # (lists and foreign array types, as envpool takes them)
        a = actions if actions.dtype == np.int32 and actions.flags.c_contiguous else np.ascontiguousarray(actions, np.int32)
        if self._fn(self._seed32, self.num_envs, int(self.spec.config.max_episode_steps), a.ctypes.data, self._st_p, self._obs_p, out_p, r_p, d_p,
                    term_p, el_p) != 0:
            raise L.CbmError(L.load().cbm_last_error().decode())
        self._obs, self._obs_p = out, out_p   # the env reads it once more (next step's older planes)
        return out, r, d.view(np.bool_), self._info(r, term, el)

    # async API (impala:308,352,365).  batch_size == num_envs: every recv() returns all envs sorted by env_id (what cleanba_impala.py
    # relies on).  batch_size < num_envs (legacy --async-batch-size, naturecnn:119-133): recv() returns the batch_size envs whose step
    # finished first, in completion order; completion times come from a deterministic integer latency per (seed, env, step), so runs are
    # reproducible while the env-id pattern is irregular like a real pool's.
    def async_reset(self):
        obs = self.reset()
        n = self.num_envs
        z = np.zeros(n, np.float32)
        if self.batch_size == n:
            self._pending = (obs, z, np.zeros(n, bool), self._info(z, np.zeros(n, np.uint8), np.zeros(n, np.int32)))
            return
        self._res = dict(reward=z.copy(), done=np.zeros(n, bool), term=np.zeros(n, np.uint8), elapsed=np.zeros(n, np.int32))
        self._waiting = np.ones(n, bool)            # env has a (possibly still running) step whose result nobody received yet
        self._futs = {}                             # env id -> future of the send() batch that is stepping it
        if getattr(self, "_pool", None) is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(1)      # like envpool, send() returns at once and the envs step in the background
        self._nsteps = np.zeros(n, np.int64)
        self._clock = 0
        self._ready_at = self._latency(np.arange(n, dtype=np.int64), 0)

    def _latency(self, e, k):
        return ((e * 2654435761 + k * 40503 + self.seed * 97) >> 7) % 5

    def recv(self):
        if self.batch_size == self.num_envs:
            out, self._pending = self._pending, None
            return out
        cand = np.nonzero(self._waiting)[0]
        assert cand.size >= self.batch_size, "recv() without enough outstanding send()s"
        ids = cand[np.argsort(self._ready_at[cand], kind="stable")[:self.batch_size]].astype(np.int32)
        for f in {id(f): f for f in (self._futs.pop(int(e), None) for e in ids) if f is not None}.values():
            self._apply(f)                          # WHO is returned is decided by the simulated latencies above (deterministic);
        self._waiting[ids] = False                  # the real threads only have to be done with those envs
        self._clock = max(self._clock + 1, int(self._ready_at[ids].max()))
        r = self._res
        rw, el = r["reward"][ids], r["elapsed"][ids]   # fancy indexing copies
        info = {"env_id": ids, "reward": rw.copy(), "terminated": r["term"][ids].astype(np.int32), "elapsed_step": el,
                "TimeLimit.truncated": el >= self.spec.config.max_episode_steps}
        return self._obs[ids], rw, r["done"][ids], info

    def send(self, actions, env_id=None):
        a = np.asarray(actions, np.int32)
        if self.batch_size == self.num_envs:
            if env_id is not None:
                b = np.zeros_like(a)
                b[np.asarray(env_id)] = a
                a = b
            self._pending = self.step(a)
            return
        ids, a = np.array(env_id, np.int32), np.array(a, np.int32)   # copies: the caller reuses its buffers while the envs step
        assert not self._waiting[ids].any(), "send() for an env whose last result was not received"
        fut = self._pool.submit(lambda: (ids,) + tuple(L.synth_env_step_host_ids(self.seed, self._st, self._obs, ids, a,
                                                                                    self.spec.config.max_episode_steps)))
        for e in ids:
            self._futs[int(e)] = fut
        self._nsteps[ids] += 1
        self._ready_at[ids] = self._clock + 1 + self._latency(ids.astype(np.int64), self._nsteps[ids])
        self._waiting[ids] = True

    def _apply(self, fut):
        """Fold one finished send() batch into the per-env result table — once, even when its envs are received in different recv()s."""
        ids, rw, d, term, el = fut.result()
        if not getattr(fut, "_applied", False):
            fut._applied = True
            r = self._res
            r["reward"][ids], r["done"][ids], r["term"][ids], r["elapsed"][ids] = rw, d.astype(bool), term, el

    def close(self):
        if getattr(self, "_pool", None) is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        self._st = None


def make_env(env_id, seed, num_envs, backend="host", num_actions=18, async_batch_size=None, reuse_buffers=False):
    """Same thunk contract as the reference's make_env (ppo:126-146); async_batch_size = envpool's batch_size (naturecnn:119-125)."""
    def thunk():
        if backend == "envpool":
            import envpool  # noqa: F401  (not installed in this image; kept for drop-in use elsewhere)
            kw = dict(batch_size=async_batch_size) if async_batch_size else {}
            envs = envpool.make(env_id, env_type="gym", num_envs=num_envs, episodic_life=False, repeat_action_probability=0.25,
                                noop_max=1, full_action_space=True, max_episode_steps=ATARI_MAX_FRAMES, reward_clip=True, seed=seed, **kw)
            envs.num_envs = num_envs
            envs.single_action_space = envs.action_space
            envs.single_observation_space = envs.observation_space
            envs.is_vector_env = True
            return envs
        return SyntheticAtariEnv(env_id, num_envs, seed, num_actions=num_actions, batch_size=async_batch_size, reuse_buffers=reuse_buffers)
    return thunk
