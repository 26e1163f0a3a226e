"""DryRunEngine: the CPU stand-in `bench.py --dry-run` runs its multi-rank code on (launcher, rendezvous, communicator bring-up, timing protocol,
watchdog, JSON merge of the N = 2 / 4 / 8 lines) when no GPU is there: OracleEngine (the oracle-backed engine of the CPU tests, gloo for the
collectives) plus the handful of HipEngine calls only the bench makes — a device-env rollout (here: the host twin of the device env stepped through
actor_step_host), the f64 all-reduce / barrier of the timing protocol, and no-op profilers.  Test infrastructure: nothing in cleanba_amd/ imports it."""
import numpy as np

from oracle_engine import OracleEngine


class DryRunEngine(OracleEngine):
    def __init__(self, cfg):
        super().__init__(cfg)
        self._envs, self._next = {}, {}

    # ---- what run_dp / run_topology ask of a HipEngine beyond the trainer's calls
    def actor_env_reset_device(self, slot, seed, atari57_mix=False):
        from cleanba_amd.envs import SyntheticAtariEnv
        env = SyntheticAtariEnv("Atari57Mix-v5" if atari57_mix else "Breakout-v5", self.E, int(seed), num_actions=self.A)
        self._envs[slot] = env
        self._next[slot] = (env.reset(), np.zeros(self.E, dtype=bool))

    def actor_rollout_device(self, slot, nsteps):
        env = self._envs[slot]
        obs, done = self._next[slot]
        actions = np.empty(self.E, np.int32)
        for _ in range(nsteps):
            self.actor_step_host(slot, obs, done, None, None, actions)
            obs, reward, done, _ = env.step(actions)
            self.actor_record_host(slot, reward)
        self._next[slot] = (obs, done)

    def actor_commit(self, s, next_obs=None, next_done=None):
        if next_obs is None and s in self._next and self.ppo:
            next_obs, next_done = self._next[s]
        return super().actor_commit(s, next_obs, next_done)

    def comm_size(self, which=0):
        return self.nranks if self.group is not None else 0

    def comm_backend(self, which=0):
        return "gloo (dry run)" if self.group is not None else ""

    def comm_allreduce_f64(self, values, op="sum", which=0):
        import torch
        a = np.ascontiguousarray(values, np.float64).copy()
        if self.group is not None:
            t = torch.from_numpy(a)
            self.group.all_reduce(t, op={"sum": self.group.ReduceOp.SUM, "max": self.group.ReduceOp.MAX, "min": self.group.ReduceOp.MIN}[op])
        return a

    def comm_barrier(self, which=0):
        self.comm_allreduce_f64([1.0])

    def comm_profile(self, on=True):
        pass

    def comm_profile_read(self):
        return 0.0, 0.0, 0

    def profile_select(self, k):
        pass

    def grad_tail_offset(self):
        return 0

    def unmap_peers(self):
        pass

    def close(self):
        if self.group is not None:   # the topology phase of the same process brings up its own process group
            try:
                self.group.destroy_process_group()
            except Exception:  # noqa: BLE001
                pass
            self.group = None
        super().close()
