"""Split actor / learner topologies (`--actor-device-ids 0 --learner-device-ids 1 2 3`, README.md:62; with `--distributed`,
several such groups, benchmark.sh:80) as one process per GPU.

Reference semantics kept (ppo:97-100, 358-363, 435-439, 587, 628, 721-725):
  * each actor slot's [T+1, E] rollout is cut along the env axis into L contiguous shards; learner l receives columns
    [l*E/L, (l+1)*E/L) of every slot and hstacks them slot-major;
  * GAE / adv-norm / shuffle / minibatching are local to the shard; gradients are averaged over ALL learner ranks of ALL
    groups once per minibatch (one flat all-reduce);
  * learner 0 of each group sends the new parameters to its actor after every update.
What replaces `jax.device_put_sharded` / `device_put`: point-to-point sends of the shard (RCCL over xGMI on GPUs, gloo in
the CPU tests) straight from / into the HBM rings, ordered on the actor slot's stream.

Rank layout: group g occupies ranks [g*G, (g+1)*G), G = len(actor_device_ids) + len(learner_device_ids); inside a group
the first len(actor_device_ids) ranks are actors, the rest learners, and local rank == position in that list.
"""
import queue
import threading

import numpy as np

PPO_FIELDS = ("obs", "actions", "logprobs", "values", "rewards", "dones")
IMPALA_FIELDS = ("obs", "actions", "logits", "rewards", "dones", "firststeps")


def is_split(args):
    return sorted(args.actor_device_ids) != sorted(args.learner_device_ids)


class Layout:
    def __init__(self, args, world_size, rank):
        self.na, self.nl = len(args.actor_device_ids), len(args.learner_device_ids)
        assert self.na == 1, "one actor device per group (the reference's published topologies: a0-l1, a0-l1,2, a0-l1,2,3)"
        assert not set(args.actor_device_ids) & set(args.learner_device_ids), "split topology needs disjoint device lists"
        self.G = self.na + self.nl
        assert world_size % self.G == 0, f"world_size {world_size} must be a multiple of {self.G} (actor + learner GPUs per group)"
        self.groups = world_size // self.G
        self.group, self.pos = rank // self.G, rank % self.G
        self.is_actor = self.pos < self.na
        self.learner_index = self.pos - self.na
        base = self.group * self.G
        self.actor_rank = base
        self.learner_ranks = [base + self.na + i for i in range(self.nl)]
        self.all_learner_ranks = [g * self.G + self.na + i for g in range(self.groups) for i in range(self.nl)]
        self.device_id = (list(args.actor_device_ids) + list(args.learner_device_ids))[self.pos]


class Groups:
    """Process groups of a split run.  Every rank builds them in the same order (torch.distributed requirement).  Rollout shards and
    parameters travel on DIFFERENT communicators: with `--concurrency` the actor is sending rollout u+1 while learner 0 is sending
    parameters u the other way, and two in-flight point-to-point kernels on one RCCL communicator would deadlock."""

    def __init__(self, dist, lay):
        self.learners = dist.new_group(ranks=lay.all_learner_ranks)
        self.data, self.params = {}, {}
        for g in range(lay.groups):
            a = g * lay.G
            for li in range(lay.nl):
                self.data[(g, li)] = dist.new_group(ranks=[a, a + lay.na + li])
            self.params[g] = dist.new_group(ranks=[a, a + lay.na])


class ActorShipper:
    """Sends every committed rollout of every slot to the group's learners, in (update, slot) order, on the engine's io stream so that
    the slot's own stream keeps stepping the next rollout while the shards are in flight."""

    def __init__(self, engine, layout, groups, args, algo, dist, num_rollouts):
        self.engine, self.lay, self.groups, self.args, self.dist, self.n = engine, layout, groups, args, dist, num_rollouts
        self.fields = PPO_FIELDS if algo == "ppo" else IMPALA_FIELDS
        self.slots = args.num_actor_threads * len(args.actor_device_ids)
        self.q = [queue.Queue() for _ in range(self.slots)]
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.error = None

    def on_commit(self, slot, update, ring_index):
        """Called by the slot's rollout thread right after cbm_actor_commit: the fence marks the end of this rollout's writes."""
        self.q[slot].put((update, ring_index, self.engine.actor_fence(slot)))

    def _run(self):
        try:
            E, L = self.args.local_num_envs, self.lay.nl
            El = E // L
            for u in range(1, self.n + 1):
                for s in range(self.slots):
                    upd, ring, fence = self.q[s].get()
                    assert upd == u
                    t = self.engine.ring_tensors(ring)
                    with self.engine.io_context():
                        self.engine.io_wait(fence)
                        for li, dst in enumerate(self.lay.learner_ranks):
                            lo = s * E + li * El
                            for f in self.fields:
                                self.dist.send(t[f][:, lo:lo + El].contiguous(), dst=dst, group=self.groups.data[(self.lay.group, li)])
                        self.engine.io_sync()
        except BaseException as e:  # noqa: BLE001
            self.error = e
            raise


class ParamReceiver:
    """Actor side of ppo:721-725: a new parameter version arrives from learner 0 after every update."""

    def __init__(self, engine, layout, groups, dist, num_updates):
        self.engine, self.lay, self.groups, self.dist, self.n = engine, layout, groups, dist, num_updates
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.error = None

    def _run(self):
        try:
            with self.engine.io_context():
                buf = self.engine.params_staging_tensor()
                for _ in range(self.n):
                    self.dist.recv(buf, src=self.lay.learner_ranks[0], group=self.groups.params[self.lay.group])
                    self.engine.io_sync()
                    self.engine.params_publish_external_tensor(buf)
        except BaseException as e:  # noqa: BLE001
            self.error = e
            raise


class LearnerReceiver:
    """Learner side of ppo:358-363 (device_put_sharded): fills one ring entry per slot with this learner's column shard, running ahead
    of the update loop by up to ring_depth rollouts (cbm_ingest_begin blocks when the ring is full)."""

    def __init__(self, engine, layout, groups, args, algo, dist, num_rollouts):
        self.engine, self.lay, self.groups, self.args, self.dist, self.n = engine, layout, groups, args, dist, num_rollouts
        self.fields = PPO_FIELDS if algo == "ppo" else IMPALA_FIELDS
        self.slots = args.num_actor_threads * len(args.actor_device_ids)
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.error = None

    def _run(self):
        try:
            El = self.args.local_num_envs // self.lay.nl
            grp = self.groups.data[(self.lay.group, self.lay.learner_index)]
            for _ in range(self.n):
                for s in range(self.slots):
                    ring = self.engine.ingest_begin(s)
                    t = self.engine.ring_tensors(ring)
                    with self.engine.io_context():
                        for f in self.fields:
                            dst = t[f][:, s * El:(s + 1) * El]
                            tmp = dst.contiguous()
                            self.dist.recv(tmp, src=self.lay.actor_rank, group=grp)
                            dst.copy_(tmp)
                        self.engine.io_sync()
                    self.engine.ingest_commit(s)
        except BaseException as e:  # noqa: BLE001
            self.error = e
            raise
