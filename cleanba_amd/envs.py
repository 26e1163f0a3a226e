"""Environment adapters behind the envpool gym API the reference uses (make_env ppo:126-146):
`reset()`, `step(actions)`, `async_reset()`, `recv()`, `send(actions, env_id)`, `action_space.n`,
`observation_space.sample()`, `spec.config.max_episode_steps`, `close()`; info carries `env_id`,
`reward` (unclipped), `terminated`, `elapsed_step`, `TimeLimit.truncated` (ppo:321-340).

SyntheticAtariEnv is the host twin of the device env compiled into libcleanba_mi.so: pure-integer
Breakout-shaped 84x84x4 frame stacks, identical bytes on CPU and GPU.  With `env_backend=envpool`
and envpool installed, the real ALE env is returned instead (not available in this image).
"""
from types import SimpleNamespace

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
    def __init__(self, env_id="Breakout-v5", num_envs=8, seed=1, max_episode_steps=ATARI_MAX_FRAMES, num_actions=18, **_):
        self.env_id, self.num_envs, self.seed = env_id, int(num_envs), int(seed)
        self.action_space = _Space(n=num_actions)
        self.observation_space = _Space(shape=(4, 84, 84))
        self.single_action_space, self.single_observation_space = self.action_space, self.observation_space
        self.is_vector_env = True
        self.spec = SimpleNamespace(config=SimpleNamespace(max_episode_steps=max_episode_steps))
        self._st = None
        self._obs = None
        self._pending = None
        self._env_id = np.arange(self.num_envs, dtype=np.int32)

    def _info(self, reward, terminated, elapsed):
        return {"env_id": self._env_id, "reward": reward.copy(), "terminated": terminated.astype(np.int32),
                "elapsed_step": elapsed, "TimeLimit.truncated": elapsed >= self.spec.config.max_episode_steps}

    def reset(self):
        self._st, self._obs = L.synth_env_reset_host(self.seed, self.num_envs, atari57_mix=is_atari57_mix(self.env_id))
        return self._obs.copy()

    def step(self, actions):
        r, d, term, el = L.synth_env_step_host(self.seed, self._st, self._obs, actions, self.spec.config.max_episode_steps)
        return self._obs.copy(), r, d.astype(bool), self._info(r, term, el)

    # async API (impala:308,352,365): batch_size == num_envs, results sorted by env_id
    def async_reset(self):
        obs = self.reset()
        z = np.zeros(self.num_envs, np.float32)
        self._pending = (obs, z, np.zeros(self.num_envs, bool), self._info(z, np.zeros(self.num_envs, np.uint8), np.zeros(self.num_envs, np.int32)))

    def recv(self):
        out, self._pending = self._pending, None
        return out

    def send(self, actions, env_id=None):
        a = np.asarray(actions, np.int32)
        if env_id is not None:
            b = np.zeros_like(a)
            b[np.asarray(env_id)] = a
            a = b
        self._pending = self.step(a)

    def close(self):
        self._st = None


def make_env(env_id, seed, num_envs, backend="host", num_actions=18):
    """Same thunk contract as the reference's make_env (ppo:126-146)."""
    def thunk():
        if backend == "envpool":
            import envpool  # noqa: F401  (not installed in this image; kept for drop-in use elsewhere)
            envs = envpool.make(env_id, env_type="gym", num_envs=num_envs, episodic_life=False, repeat_action_probability=0.25,
                                noop_max=1, full_action_space=True, max_episode_steps=ATARI_MAX_FRAMES, reward_clip=True, seed=seed)
            envs.num_envs = num_envs
            envs.single_action_space = envs.action_space
            envs.single_observation_space = envs.observation_space
            envs.is_vector_env = True
            return envs
        return SyntheticAtariEnv(env_id, num_envs, seed, num_actions=num_actions)
    return thunk
