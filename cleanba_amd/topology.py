"""Split actor / learner topologies (`--actor-device-ids 0 --learner-device-ids 1 2 3`, README.md:62; `--learner-device-ids 0 1` sharing
GPU 0 with the actor, README.md:58; two actor GPUs feeding two learner GPUs, benchmark.sh:90; several such groups with `--distributed`,
benchmark.sh:80) as one process per ROLE: every entry of the two id lists is a process bound to that GPU (an actor and a learner may
share one GPU).

Reference semantics kept (ppo:97-100, 358-363, 435-439, 587, 628, 721-725):
  * each actor thread's [T+1, E] rollout is cut along the env axis into L contiguous shards; learner l receives columns
    [l*E/L, (l+1)*E/L) of every (actor device, thread) and hstacks them device-thread-major;
  * GAE / adv-norm / shuffle / minibatching are local to the shard; gradients are averaged over ALL learner ranks of ALL
    groups once per minibatch;
  * learner 0 of each group hands the new parameters to its actors after every update.
What replaces `jax.device_put_sharded` / `device_put`: the actor WRITES each shard straight into the learner's HBM ring through a HIP IPC
mapping (one strided 2-D copy per field on a side stream — xGMI peer writes, no staging), learner 0 writes parameters straight into the
actors' version buffers the same way, and only "it has landed" travels as a host message (cleanba_amd.topology.Rendezvous, a TCP
key-value store) — the points where the reference blocks on queue.put / queue.get.

Rank layout: group g occupies ranks [g*G, (g+1)*G), G = len(actor_device_ids) + len(learner_device_ids); inside a group the actor
roles come first, then the learners, and LOCAL_RANK == the GPU id from the list.
"""
import datetime
import os
import pickle
import queue
import threading
import sys
import time

PPO_FIELDS = ("obs", "actions", "logprobs", "values", "rewards", "dones")
IMPALA_FIELDS = ("obs", "actions", "logits", "rewards", "dones", "firststeps")


def is_split(args):
    """One process does everything only for the a0-l0 shape; every other pair of id lists is a set of role processes."""
    return not (len(args.actor_device_ids) == 1 and list(args.actor_device_ids) == list(args.learner_device_ids))


def validate(args):
    """Fail before any worker is spawned, with the supported shapes spelled out."""
    a, l = list(args.actor_device_ids), list(args.learner_device_ids)
    if not a or not l or len(set(a)) != len(a) or len(set(l)) != len(l) or min(a + l) < 0:
        raise SystemExit(f"--actor-device-ids {a} / --learner-device-ids {l}: each list needs distinct, non-negative GPU ids")
    if args.local_num_envs % len(l):
        raise SystemExit("local_num_envs must be divisible by len(learner_device_ids) (ppo:412)")
    if getattr(args, "async_batch_size", 0) and is_split(args):
        raise SystemExit("--async-batch-size runs a0-l0 only (naturecnn:105)")


class Layout:
    def __init__(self, args, world_size, rank):
        validate(args)
        self.na, self.nl = len(args.actor_device_ids), len(args.learner_device_ids)
        self.G = self.na + self.nl
        if world_size % self.G:
            raise SystemExit(f"world_size {world_size} must be a multiple of {self.G} (actor + learner roles per group)")
        self.groups = world_size // self.G
        self.group, self.pos = rank // self.G, rank % self.G
        self.is_actor = self.pos < self.na
        self.actor_index = self.pos if self.is_actor else -1
        self.learner_index = self.pos - self.na
        base = self.group * self.G
        self.actor_ranks = [base + i for i in range(self.na)]
        self.learner_ranks = [base + self.na + i for i in range(self.nl)]
        self.all_learner_ranks = [g * self.G + self.na + i for g in range(self.groups) for i in range(self.nl)]
        self.device_id = (list(args.actor_device_ids) + list(args.learner_device_ids))[self.pos]
        self.threads = args.num_actor_threads
        self.ports = self.na * self.threads          # ingest ports of a learner = (actor device, thread) pairs, device-major (ppo:668-686)
        self.shard_envs = args.local_num_envs // self.nl


class Rendezvous:
    """Host control plane of a multi-process run: torch's TCPStore (a C++ key-value server on rank 0; no process group, no GPU).  Carries
    the RCCL unique id, the IPC handles and the 'landed' notifications.  `get` polls so that a failed peer (which posts 'abort') or a
    vanished server ends the wait instead of hanging it.  Construct it BEFORE the process creates its first HIP context: the constructor imports torch, and
    importing torch into a process whose HIP runtime is already live has been seen to stall inside torch's extension load (round 6, two rank processes)."""

    _base = {}   # (addr, port, world, rank) -> TCPStore: one connection (and, on rank 0, one server) per process, whatever the number of runs
    _runs = {}   # (store key, prefix) -> how many Rendezvous objects of this process used that prefix so far

    def __init__(self, world, rank, addr, port, timeout_s=1800.0, prefix=None):
        from torch.distributed import TCPStore
        self.world, self.rank, self.timeout_s = world, rank, timeout_s
        # rank 0 hosts the server — unless torchrun's agent already does on this port (TORCHELASTIC_USE_AGENT_STORE: TCPStore then joins
        # that one); every key carries a prefix so the job's keys never meet the launcher's, and a process that makes several runs in a row
        # (bench.py: the data-parallel line, then the BASELINE topology line) gives each run its own prefix (CBM_RDV_PREFIX)
        from torch.distributed import PrefixStore
        k = (addr, int(port), world, rank)
        if k not in Rendezvous._base:
            Rendezvous._base[k] = TCPStore(addr, int(port), world, rank == 0, timeout=datetime.timedelta(seconds=timeout_s), wait_for_workers=False)
        # a second run of the same process under the same prefix must not inherit the first one's keys (a sticky 'abort', barrier counters that
        # already stand at `world`): the n-th use of a prefix gets "<prefix>#n" — every rank builds its Rendezvous objects in the same order
        prefix = prefix or os.environ.get("CBM_RDV_PREFIX", "cbm")
        n = Rendezvous._runs.get((k, prefix), 0)
        Rendezvous._runs[(k, prefix)] = n + 1
        self.store = PrefixStore(prefix if n == 0 else f"{prefix}#{n}", Rendezvous._base[k])

    def put(self, key, value=b"1"):
        self.store.set(key, value)

    def get(self, key, take=False):
        """Blocks until `key` exists.  Polls with non-blocking checks (50 us backing off to 1 ms) instead of a blocking wait, so that a failed
        peer's 'abort' key or a vanished server ends the wait, and the 'landed' notifications of the split path cost well under a millisecond."""
        t0 = time.time()
        nap, since_abort_check = 5e-5, 0.0
        while not self.store.check([key]):
            time.sleep(nap)
            since_abort_check += nap
            nap = min(nap * 1.5, 1e-3)
            if since_abort_check >= 1.0:
                since_abort_check = 0.0
                if self.store.check(["abort"]):
                    why = self.store.get("abort").decode(errors="replace")
                    raise RuntimeError(f"rank {self.rank}: a peer aborted while this rank waited for '{key}' ({why})")
                if time.time() - t0 > self.timeout_s:
                    raise TimeoutError(f"rank {self.rank}: '{key}' did not arrive within {self.timeout_s} s")
        v = self.store.get(key)
        if take:
            self.store.delete_key(key)
        return v

    def abort(self, reason=None):
        """Tells every peer blocked in get() to stop; the key carries who failed and with what (the exception being handled, if any): the FIRST
        failure is what the peers report, later aborts (peers that failed because of it) do not overwrite it."""
        try:
            if reason is None:
                exc = sys.exc_info()[1]
                reason = repr(exc) if exc is not None else "abort requested"
            self.store.compare_set("abort", "", f"rank {self.rank}: {reason}"[:500])
        except Exception:  # noqa: BLE001
            pass

    def barrier(self, tag):
        n = self.store.add(f"barrier/{tag}", 1)
        if n == self.world:
            self.put(f"barrier/{tag}/go")
        self.get(f"barrier/{tag}/go")
        # rank 0 hosts the store: it may only leave (and possibly exit) once every other rank has finished talking to it
        if self.rank != 0:
            self.store.add(f"barrier/{tag}/ack", 1)
        else:
            t0 = time.time()
            while self.store.add(f"barrier/{tag}/ack", 0) < self.world - 1:
                time.sleep(1e-3)
                if time.time() - t0 > 60.0:
                    break   # a peer died after the barrier: nothing left to protect

    def share(self, key, make, owner_rank):
        """The owner computes a value once, everybody receives it."""
        if self.rank == owner_rank:
            v = make()
            self.put(key, v)
            return v
        return self.get(key)


def comm_backend(nranks=2):
    """CBM_COMM=native|rccl.  Default: RCCL — except when several ranks were told to share one GPU (CBM_FORCE_DEVICE, the one-GPU test shape),
    where RCCL refuses two ranks per device and the library's native all-reduce is the only backend that can run."""
    v = os.environ.get("CBM_COMM", "").lower()
    if v in ("native", "rccl"):
        return v
    return "native" if os.environ.get("CBM_FORCE_DEVICE") is not None and nranks > 1 else "rccl"


def setup_learner_comm(engine, rdv, ranks, rank, tag="learners"):
    """The communicator of the gradient / statistics all-reduce over `ranks` (pmap's device list, ppo:435-439,656-660)."""
    if len(ranks) < 2 and not engine.wants_comm_at_world_one():
        return
    if os.environ.get("CBM_COMM_LOOPBACK") == "1" and hasattr(engine, "comm_init_loopback"):
        # testing aid: the self-test communicator (SUM over identical ranks) instead of RCCL, so the multi-process plumbing — launcher,
        # rendezvous, one result line — can run with several ranks on a ONE-GPU box, where RCCL refuses two ranks per device
        rdv.barrier(f"comm/{tag}") if rdv is not None else None
        engine.comm_init_loopback(len(ranks))
        return
    if comm_backend(len(ranks)) == "native" and hasattr(engine, "comm_native_export"):
        # the library's own all-reduce kernels over IPC-mapped peer buffers (csrc/comm.hip): every rank publishes its blob, reads everybody's
        me = ranks.index(rank)
        blob = engine.comm_native_export()
        if rdv is not None:
            rdv.put(f"comm/{tag}/native/{me}", blob)
            blobs = [blob if i == me else bytes(rdv.get(f"comm/{tag}/native/{i}")) for i in range(len(ranks))]
        else:
            blobs = [blob]
        engine.comm_native_init(blobs, me)
        return
    uid = rdv.share(f"comm/{tag}/uid", engine.comm_unique_id, ranks[0]) if len(ranks) > 1 else engine.comm_unique_id()
    engine.comm_init(uid, len(ranks), ranks.index(rank))


class ActorShipper:
    """Actor role: writes every committed rollout of every local thread into the group's learners, in (update, thread) order, on the engine's
    io stream — the thread's own stream keeps stepping the next rollout meanwhile — and posts 'landed' once the copies are complete."""

    def __init__(self, engine, layout, rdv, args, algo, num_rollouts):
        self.engine, self.lay, self.rdv, self.n = engine, layout, rdv, num_rollouts
        self.fields = PPO_FIELDS if algo == "ppo" else IMPALA_FIELDS
        self.q = [queue.Queue() for _ in range(layout.threads)]
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.error = None
        self.stop = threading.Event()   # set by a failing sibling (a rollout thread that died will never commit): _run must not wait for it forever
        g = layout.group
        # map every learner's ring (ppo:358-363's device_put_sharded targets)
        self.peer = [engine.open_peer_ring(pickle.loads(rdv.get(f"g{g}/ring/{li}")), f"actor rank {rdv.rank} (group {g}, actor {layout.actor_index}) maps the ring of learner {li} (rank {layout.learner_ranks[li]})")
                     for li in range(layout.nl)]

    def on_commit(self, slot, update, ring_index):
        """Called by the thread's rollout loop right after cbm_actor_commit (the commit event orders the copies after the rollout)."""
        self.q[slot].put((update, ring_index))

    def _next(self, s):
        """Queue.get that gives up when a sibling thread failed (self.stop) instead of blocking the actor process for ever."""
        while True:
            try:
                return self.q[s].get(timeout=0.2)
            except queue.Empty:
                if self.stop.is_set():
                    raise RuntimeError(f"actor thread {s} stopped before committing its next rollout") from None

    def _run(self):
        try:
            lay = self.lay
            cols = lay.ports * lay.shard_envs
            for u in range(1, self.n + 1):
                for s in range(lay.threads):
                    upd, ring = self._next(s)
                    assert upd == u
                    port = lay.actor_index * lay.threads + s
                    for li in range(lay.nl):
                        self.engine.actor_ship_shard(s, ring, li, lay.nl, self.peer[li][ring], cols, port * lay.shard_envs)
                    self.engine.io_sync()
                    for li in range(lay.nl):
                        self.rdv.put(f"g{lay.group}/shard/{li}/{u}/{port}", str(ring).encode())
        except BaseException as e:  # noqa: BLE001
            self.error = e
            self.rdv.abort()
            self.engine.abort()
            raise


class ParamReceiver:
    """Actor role, ppo:721-725: learner 0 has written a new parameter version into this actor's version buffer."""

    def __init__(self, engine, layout, rdv, num_updates):
        self.engine, self.lay, self.rdv, self.n = engine, layout, rdv, num_updates
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.error = None
        rdv.put(f"g{layout.group}/aparams/{layout.actor_index}", pickle.dumps(engine.export_actor_params(tag=rdv.rank)))

    def _run(self):
        try:
            for v in range(1, self.n + 1):
                self.rdv.get(f"g{self.lay.group}/params/{self.lay.actor_index}/{v}", take=True)
                self.engine.params_mark_published()
        except BaseException as e:  # noqa: BLE001
            self.error = e
            self.rdv.abort()
            self.engine.abort()
            raise


class LearnerReceiver:
    """Learner role: one ring entry per rollout, one ingest port per (actor device, thread); an entry is handed to the update loop when
    every port's shard has landed.  Runs ahead of the update loop by up to ring_depth rollouts (cbm_ingest_begin blocks on a full ring)."""

    def __init__(self, engine, layout, rdv, args, algo, num_rollouts):
        self.engine, self.lay, self.rdv, self.n = engine, layout, rdv, num_rollouts
        self.fields = PPO_FIELDS if algo == "ppo" else IMPALA_FIELDS
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.error = None
        rdv.put(f"g{layout.group}/ring/{layout.learner_index}", pickle.dumps(engine.export_ring(self.fields, tag=rdv.rank)))

    def _run(self):
        try:
            lay = self.lay
            for u in range(1, self.n + 1):
                for port in range(lay.ports):
                    ring = self.engine.ingest_begin(port)
                    got = int(self.rdv.get(f"g{lay.group}/shard/{lay.learner_index}/{u}/{port}", take=True))
                    assert got == ring, f"actor wrote ring entry {got}, learner expected {ring}"
                    self.engine.ingest_commit(port)
        except BaseException as e:  # noqa: BLE001
            self.error = e
            self.rdv.abort()
            self.engine.abort()
            raise


class ParamSender:
    """Learner 0: after every update, write the parameters into every actor of the group and tell it (ppo:721-725)."""

    def __init__(self, engine, layout, rdv):
        self.engine, self.lay, self.rdv = engine, layout, rdv
        g = layout.group
        self.peers = [engine.open_peer_params(pickle.loads(rdv.get(f"g{g}/aparams/{ai}")), f"learner rank {rdv.rank} (group {g}, learner 0) maps the parameter buffers of actor {ai} (rank {layout.actor_ranks[ai]})")
                      for ai in range(layout.na)]

    def __call__(self, version):
        for ai, peer in enumerate(self.peers):
            self.engine.params_push(peer)
            self.rdv.put(f"g{self.lay.group}/params/{ai}/{version}")
