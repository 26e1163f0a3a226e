"""ctypes binding of libcleanba_mi.so (include/cleanba_mi.h).

This is the only way the Python host reaches the GPU: there is no CPU fallback.  If the
shared library is missing, or no MI355X is visible, importing/constructing fails loudly.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("CBM_SO", os.path.join(_HERE, "libcleanba_mi.so"))  # CBM_SO: ablation builds (tools/)

NET_NATURE, NET_IMPALA_RESNET = 0, 1
ALGO_PPO, ALGO_IMPALA = 0, 1
FRAME = 4 * 84 * 84


class CbmError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("device", C.c_int32), ("network", C.c_int32), ("algo", C.c_int32),
                ("num_actions", C.c_int32), ("local_num_envs", C.c_int32), ("num_actor_slots", C.c_int32),
                ("num_steps", C.c_int32), ("num_minibatches", C.c_int32), ("update_epochs", C.c_int32),
                ("norm_adv", C.c_int32), ("ring_depth", C.c_int32), ("gamma", C.c_float), ("gae_lambda", C.c_float),
                ("clip_coef", C.c_float), ("ent_coef", C.c_float), ("vf_coef", C.c_float), ("max_grad_norm", C.c_float),
                ("adam_b1", C.c_float), ("adam_b2", C.c_float), ("adam_eps", C.c_float), ("rms_decay", C.c_float),
                ("rms_eps", C.c_float), ("actor_dense_ksplit", C.c_int32), ("forward_bf16", C.c_int32), ("grad_accum_steps", C.c_int32), ("async_batch_size", C.c_int32),
                ("backward_split", C.c_int32), ("num_channels", C.c_int32), ("channels", C.c_int32 * 4), ("num_hiddens", C.c_int32),
                ("hiddens", C.c_int32 * 4), ("conv1_fp32_chain", C.c_int32), ("reserved", C.c_int32 * 2)]


class EnvState(C.Structure):
    _fields_ = [("elapsed", C.c_int32), ("needs_reset", C.c_int32), ("paddle_x", C.c_int32), ("ball_x", C.c_int32),
                ("ball_y", C.c_int32), ("ball_dx", C.c_int32), ("ball_dy", C.c_int32), ("bricks", C.c_uint32 * 3),
                ("episode", C.c_uint32), ("ep_return", C.c_float), ("ep_length", C.c_float), ("ret_return", C.c_float),
                ("ret_length", C.c_float), ("game", C.c_int32)]


# every symbol include/cleanba_mi.h declares (checked by tests/test_abi.py against the header text)
SYMBOLS = [
    "cbm_default_config", "cbm_config_size", "cbm_ctx_create", "cbm_ctx_destroy", "cbm_last_error", "cbm_build_info", "cbm_param_count", "cbm_param_count_hidden",
    "cbm_params_set", "cbm_params_get", "cbm_actor_params_get", "cbm_buffer", "cbm_copy_to_host", "cbm_copy_to_device",
    "cbm_dev_alloc", "cbm_dev_free", "cbm_learner_stream", "cbm_sync", "cbm_actor_set_key", "cbm_actor_get_key",
    "cbm_actor_begin_rollout", "cbm_actor_step_host", "cbm_actor_record_host", "cbm_actor_rollout_device",
    "cbm_actor_commit", "cbm_actor_episode_stats", "cbm_learner_wait", "cbm_learner_update", "cbm_learner_prepare",
    "cbm_learner_epoch_begin", "cbm_learner_minibatch_grad", "cbm_learner_accumulate", "cbm_learner_optimizer_step", "cbm_learner_finish",
    "cbm_forward", "cbm_sample", "cbm_gae", "cbm_advnorm", "cbm_permutation", "cbm_ppo_loss_grad",
    "cbm_impala_loss_grad", "cbm_adam_step", "cbm_rmsprop_step", "cbm_synth_env_reset_host", "cbm_synth_env_reset_host_games", "cbm_synth_env_step_host", "cbm_actor_env_reset_device_games",
    "cbm_actor_env_reset_device", "cbm_profile_select", "cbm_profile_read", "cbm_ingest_begin", "cbm_ingest_commit",
    "cbm_params_publish_external", "cbm_actor_stream", "cbm_actor_ring_index", "cbm_actor_step_async", "cbm_gae_async", "cbm_mb_advnorm",
    "cbm_synth_env_step_host_ids", "cbm_synth_env_step_host_to", "cbm_synth_env_render_host", "cbm_learner_grad_tail_offset", "cbm_vtrace", "cbm_comm_init_loopback",
    "cbm_comm_load", "cbm_comm_unique_id", "cbm_comm_init", "cbm_comm_size", "cbm_comm_allreduce_f64", "cbm_comm_barrier",
    "cbm_learner_allreduce_grads", "cbm_comm_profile", "cbm_comm_profile_read", "cbm_ipc_export_window", "cbm_ipc_window_offset", "cbm_ipc_open_window",
    "cbm_ipc_close_window", "cbm_ipc_close_all",
    "cbm_host_register", "cbm_host_unregister", "cbm_actor_ship_shard", "cbm_io_sync", "cbm_params_push", "cbm_params_mark_published", "cbm_ctx_abort", "cbm_profile_read_all", "cbm_profile_kernel_name",
    "cbm_comm_native_export", "cbm_comm_native_init", "cbm_comm_backend", "cbm_comm_allreduce_grads", "cbm_comm_overlap_probe",
]

COMM_LEARNERS, COMM_WORLD = 0, 1
COMM_ID_BYTES, IPC_HANDLE_BYTES, IPC_WINDOW_BYTES = 128, 64, 128
NATIVE_BLOB_BYTES, NATIVE_MAX_RANKS = 320, 16
RING_FIELDS = ("obs", "actions", "logprobs", "values", "rewards", "dones", "firststeps", "logits")


class PeerRing(C.Structure):
    """cbm_peer_ring: device pointers of one ring entry of the destination context (None = field not shipped)."""
    _fields_ = [(f, C.c_void_p) for f in RING_FIELDS]

_lib = None


def build(force=False):
    """Compile libcleanba_mi.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j4"] + (["-B"] if force else [])
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return SO_PATH


_TORCH_RCCL = None   # torch's bundled librccl.so (bound by cbm_comm_load so the process holds ONE HIP runtime and ONE RCCL)


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so / libhsa-runtime64.so (torch/lib, DT_NEEDED without the .7 suffix), which the
    loader does not unify with /opt/rocm's copy when OUR library is loaded first: the process then holds two HIP runtimes and torch finds
    "No HIP GPUs".  The distributed paths share streams and buffers with torch (RCCL all-reduce, shard sends), so bind to torch's runtime
    up front — by path, without importing torch.  CBM_SYSTEM_HIP=1 keeps /opt/rocm's runtime (torch-free deployments)."""
    global _TORCH_RCCL
    if os.environ.get("CBM_SYSTEM_HIP") == "1":
        return
    import importlib.util
    spec = importlib.util.find_spec("torch")
    for loc in (spec.submodule_search_locations or []) if spec else []:
        p = os.path.join(loc, "lib", "libamdhip64.so")
        if os.path.exists(p):
            C.CDLL(p, mode=C.RTLD_GLOBAL)
            r = os.path.join(loc, "lib", "librccl.so")
            _TORCH_RCCL = r if os.path.exists(r) else None
            return


def load():
    global _lib
    if _lib is not None:
        return _lib
    _share_hip_runtime_with_torch()
    if not os.path.exists(SO_PATH):
        raise CbmError(f"{SO_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no CPU fallback for the hot path)")
    lib = C.CDLL(SO_PATH)
    lib.cbm_last_error.restype = C.c_char_p
    lib.cbm_build_info.restype = C.c_char_p
    lib.cbm_param_count.restype = C.c_int64
    lib.cbm_param_count_hidden.restype = C.c_int64
    if lib.cbm_config_size() != C.sizeof(Config):
        raise CbmError(f"cbm_config is {lib.cbm_config_size()} bytes in {SO_PATH}, {C.sizeof(Config)} in cleanba_amd/lib.py: rebuild the library")
    lib.cbm_learner_grad_tail_offset.restype = C.c_int64
    lib.cbm_learner_stream.restype = C.c_void_p
    lib.cbm_learner_stream.argtypes = [C.c_void_p]
    lib.cbm_actor_stream.restype = C.c_void_p
    lib.cbm_actor_stream.argtypes = [C.c_void_p, C.c_int32]
    vp = C.c_void_p
    lib.cbm_actor_step_host.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp]
    lib.cbm_actor_record_host.argtypes = [vp, C.c_int32, vp]
    lib.cbm_synth_env_step_host_to.argtypes = [C.c_uint32, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.cbm_comm_load.argtypes = [C.c_char_p]
    lib.cbm_comm_backend.restype = C.c_char_p
    _lib = lib
    return lib


def _chk(rc):
    if rc != 0:
        raise CbmError(load().cbm_last_error().decode())


def _p(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return C.c_void_p(a.ctypes.data)   # (data_as() costs 2.7 us a call, this 1.3: a 120-env host step makes a dozen of them under the GIL)
    return C.c_void_p(a)


def default_config(algo=ALGO_PPO):
    cfg = Config()
    _chk(load().cbm_default_config(int(algo), C.byref(cfg)))
    return cfg


def param_count(network, A, hidden=0):
    """hidden: width of the IMPALA-ResNet's hidden layer (`--hiddens`, ppo:94); 0 = the network's default."""
    if hidden:
        return int(load().cbm_param_count_hidden(int(network), int(A), int(hidden)))
    return int(load().cbm_param_count(int(network), int(A)))


class DevBuf:
    """A raw device allocation (hipMalloc) with numpy upload/download; used by tests and the host."""

    def __init__(self, ctx, arr=None, nbytes=None, dtype=np.uint8, shape=None):
        self.ctx = ctx
        if arr is not None:
            arr = np.ascontiguousarray(arr)
            nbytes, dtype, shape = arr.nbytes, arr.dtype, arr.shape
        self.nbytes, self.dtype, self.shape = int(nbytes), np.dtype(dtype), shape
        p = C.c_void_p()
        _chk(load().cbm_dev_alloc(C.c_int64(max(self.nbytes, 16)), C.byref(p)))
        self.ptr = p.value
        if arr is not None:
            self.upload(arr)

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        _chk(load().cbm_copy_to_device(self.ctx.h, C.c_void_p(self.ptr), _p(arr), C.c_int64(arr.nbytes)))

    def download(self):
        out = np.empty(self.nbytes // self.dtype.itemsize, self.dtype)
        _chk(load().cbm_copy_to_host(self.ctx.h, _p(out), C.c_void_p(self.ptr), C.c_int64(self.nbytes)))
        return out.reshape(self.shape) if self.shape is not None else out

    def free(self):
        if self.ptr:
            load().cbm_dev_free(C.c_void_p(self.ptr))
            self.ptr = None


class Context:
    """Owns one cbm_ctx (one GPU: actor slots + learner)."""

    def __init__(self, cfg):
        self.lib = load()
        self.cfg = cfg
        h = C.c_void_p()
        _chk(self.lib.cbm_ctx_create(C.byref(cfg), C.byref(h)))
        self.h = h
        self.A = cfg.num_actions
        self._ao = {}          # slot -> (that actor thread's action array of its last host step, its address)
        self.P = param_count(cfg.network, cfg.num_actions, cfg.hiddens[0] if cfg.network == NET_IMPALA_RESNET else 0)

    def close(self):
        if self.h:
            self.lib.cbm_ctx_destroy(self.h)
            self.h = None

    # ---- params / buffers
    def set_params(self, p):
        p = np.ascontiguousarray(p, np.float32)
        _chk(self.lib.cbm_params_set(self.h, _p(p), C.c_int64(p.size)))

    def get_params(self):
        out = np.empty(self.P, np.float32)
        _chk(self.lib.cbm_params_get(self.h, _p(out), C.c_int64(out.size)))
        return out

    def buffer(self, name, ring=0):
        p, n = C.c_void_p(), C.c_int64()
        _chk(self.lib.cbm_buffer(self.h, name.encode(), int(ring), C.byref(p), C.byref(n)))
        return p.value, n.value

    def read(self, name, dtype, ring=0, shape=None):
        ptr, n = self.buffer(name, ring)
        out = np.empty(n // np.dtype(dtype).itemsize, dtype)
        _chk(self.lib.cbm_copy_to_host(self.h, _p(out), C.c_void_p(ptr), C.c_int64(n)))
        return out.reshape(shape) if shape is not None else out

    def write(self, name, arr, ring=0, offset_bytes=0):
        ptr, n = self.buffer(name, ring)
        arr = np.ascontiguousarray(arr)
        assert offset_bytes + arr.nbytes <= n
        _chk(self.lib.cbm_copy_to_device(self.h, C.c_void_p(ptr + offset_bytes), _p(arr), C.c_int64(arr.nbytes)))

    def sync(self):
        _chk(self.lib.cbm_sync(self.h))

    def learner_stream(self):
        return self.lib.cbm_learner_stream(self.h)

    # ---- actor
    def actor_set_key(self, slot, key):
        k = np.ascontiguousarray(key, np.uint32)
        _chk(self.lib.cbm_actor_set_key(self.h, int(slot), _p(k)))

    def actor_get_key(self, slot):
        k = np.zeros(2, np.uint32)
        _chk(self.lib.cbm_actor_get_key(self.h, int(slot), _p(k)))
        return k

    def actor_env_reset_device(self, slot, seed, atari57_mix=False):
        _chk(self.lib.cbm_actor_env_reset_device_games(self.h, int(slot), C.c_uint32(int(seed) & 0xFFFFFFFF), int(bool(atari57_mix))))

    def actor_begin_rollout(self, slot, concurrency):
        v = C.c_int32()
        _chk(self.lib.cbm_actor_begin_rollout(self.h, int(slot), int(bool(concurrency)), C.byref(v)))
        return v.value

    def actor_step_host(self, slot, obs, done, firststep=None, reward_with_obs=None, actions_out=None):
        E = self.cfg.local_num_envs
        if actions_out is None:
            actions_out = np.empty(E, np.int32)
        # per-step call of the envpool-API loop: everything here runs under the GIL, so addresses of arrays that come back every step (the
        # caller's action buffer) are cached and the rest is taken with one attribute read each
        # (per SLOT, and read into a local: several actor threads step the same context, each with its own buffer — one shared field let a thread
        # switch between the check and the call hand thread A's call thread B's buffer)
        ao = self._ao.get(slot)
        if ao is None or actions_out is not ao[0]:
            assert actions_out.dtype == np.int32 and actions_out.flags.c_contiguous and actions_out.size >= E
            ao = (actions_out, actions_out.ctypes.data)
            self._ao[slot] = ao
        if obs.dtype != np.uint8 or not obs.flags.c_contiguous:
            obs = np.ascontiguousarray(obs, np.uint8)
        done = done.view(np.uint8) if done.dtype == np.bool_ and done.flags.c_contiguous else np.ascontiguousarray(done, np.uint8)
        fs = None if firststep is None else np.ascontiguousarray(firststep, np.uint8)
        rw = None if reward_with_obs is None else np.ascontiguousarray(reward_with_obs, np.float32)
        _chk(self.lib.cbm_actor_step_host(self.h, int(slot), obs.ctypes.data, done.ctypes.data, None if fs is None else fs.ctypes.data,
                                          None if rw is None else rw.ctypes.data, ao[1]))
        return actions_out

    def actor_step_async(self, slot, obs, reward, done, env_id, actions_out=None):
        """One envpool.recv() batch of the legacy async loop (naturecnn:346-367) -> actions for envs.send(actions, env_id)."""
        Ba = self.cfg.async_batch_size
        if actions_out is None:
            actions_out = np.empty(Ba, np.int32)
        obs = np.ascontiguousarray(obs, np.uint8)
        rw = np.ascontiguousarray(reward, np.float32)
        done = np.ascontiguousarray(done, np.uint8)
        eid = np.ascontiguousarray(env_id, np.int32)
        if obs.shape[0] != Ba or eid.size != Ba:
            raise ValueError(f"async step expects batches of async_batch_size={Ba} envs, got {obs.shape[0]}")
        _chk(self.lib.cbm_actor_step_async(self.h, int(slot), _p(obs), _p(rw), _p(done), _p(eid), _p(actions_out)))
        return actions_out

    def actor_record_host(self, slot, reward):
        r = reward if reward.dtype == np.float32 and reward.flags.c_contiguous else np.ascontiguousarray(reward, np.float32)
        _chk(self.lib.cbm_actor_record_host(self.h, int(slot), r.ctypes.data))

    def host_register(self, arr):
        """Page-locks a numpy array the env reuses for its observations (cbm_host_register)."""
        assert arr.flags["C_CONTIGUOUS"]
        _chk(self.lib.cbm_host_register(self.h, _p(arr), C.c_int64(arr.nbytes)))

    def host_unregister(self, arr):
        _chk(self.lib.cbm_host_unregister(self.h, _p(arr)))

    def actor_rollout_device(self, slot, nsteps):
        _chk(self.lib.cbm_actor_rollout_device(self.h, int(slot), int(nsteps)))

    def actor_commit(self, slot, next_obs=None, next_done=None):
        no = None if next_obs is None else np.ascontiguousarray(next_obs, np.uint8)
        nd = None if next_done is None else np.ascontiguousarray(next_done, np.uint8)
        _chk(self.lib.cbm_actor_commit(self.h, int(slot), _p(no), _p(nd)))

    def actor_episode_stats(self, slot):
        r, l = C.c_float(), C.c_float()
        _chk(self.lib.cbm_actor_episode_stats(self.h, int(slot), C.byref(r), C.byref(l)))
        return r.value, l.value

    def profile_select(self, kernel_id):
        _chk(self.lib.cbm_profile_select(self.h, int(kernel_id)))

    def profile_read(self):
        ms, n = C.c_double(), C.c_int32()
        _chk(self.lib.cbm_profile_read(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_kernel_name(self, kernel_id):
        """'<kernel symbol> <problem functor>' of the kernel launched for `kernel_id` the last time it was timed ('' if never)."""
        buf = C.create_string_buffer(512)
        _chk(self.lib.cbm_profile_kernel_name(self.h, int(kernel_id), buf, 512))
        return buf.value.decode()

    def profile_read_all(self, n_ids=12):
        ms, n = np.zeros(n_ids, np.float64), np.zeros(n_ids, np.int32)
        _chk(self.lib.cbm_profile_read_all(self.h, _p(ms), _p(n), int(n_ids)))
        return ms, n

    # ---- split topologies
    def ingest_begin(self, slot):
        r = C.c_int32()
        _chk(self.lib.cbm_ingest_begin(self.h, int(slot), C.byref(r)))
        return r.value

    def ingest_commit(self, slot):
        _chk(self.lib.cbm_ingest_commit(self.h, int(slot)))

    def params_publish_external(self, dev_ptr):
        _chk(self.lib.cbm_params_publish_external(self.h, C.c_void_p(dev_ptr), C.c_int64(self.P)))

    def actor_stream(self, slot):
        return self.lib.cbm_actor_stream(self.h, int(slot))

    def actor_ring_index(self, slot):
        return int(self.lib.cbm_actor_ring_index(self.h, int(slot)))

    def abort(self):
        """Every blocking wait of this context fails from now on (a host thread died: do not hang the others)."""
        if self.h:
            self.lib.cbm_ctx_abort(self.h)

    # ---- collectives among the learner GPUs (RCCL behind the C ABI)
    def comm_init(self, which, uid, nranks, rank):
        if _TORCH_RCCL:
            _chk(self.lib.cbm_comm_load(_TORCH_RCCL.encode()))
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(bytes(uid))
        _chk(self.lib.cbm_comm_init(self.h, int(which), buf, int(nranks), int(rank)))

    def comm_init_loopback(self, nranks, which=COMM_LEARNERS):
        _chk(self.lib.cbm_comm_init_loopback(self.h, int(which), int(nranks)))

    # native backend (no RCCL): kernels working on the peers' buffers through IPC mappings; ranks may share one GPU
    def comm_native_export(self, which=COMM_LEARNERS):
        blob = (C.c_uint8 * NATIVE_BLOB_BYTES)()
        _chk(self.lib.cbm_comm_native_export(self.h, int(which), blob))
        return bytes(blob)

    def comm_native_init(self, blobs, rank, which=COMM_LEARNERS):
        """blobs: every rank's comm_native_export(), in rank order."""
        assert all(len(b) == NATIVE_BLOB_BYTES for b in blobs)
        table = (C.c_uint8 * (NATIVE_BLOB_BYTES * len(blobs))).from_buffer_copy(b"".join(blobs))
        _chk(self.lib.cbm_comm_native_init(self.h, int(which), len(blobs), int(rank), table))

    def comm_backend(self, which=COMM_LEARNERS):
        return self.lib.cbm_comm_backend(self.h, int(which)).decode()

    def comm_size(self, which=COMM_LEARNERS):
        return int(self.lib.cbm_comm_size(self.h, int(which)))

    def comm_allreduce_f64(self, values, op="sum", which=COMM_LEARNERS):
        a = np.ascontiguousarray(values, np.float64).copy()
        _chk(self.lib.cbm_comm_allreduce_f64(self.h, int(which), _p(a), int(a.size), {"sum": 0, "max": 1, "min": 2}[op]))
        return a

    def comm_barrier(self, which=COMM_LEARNERS):
        _chk(self.lib.cbm_comm_barrier(self.h, int(which)))

    def comm_allreduce_grads(self, which=COMM_LEARNERS):
        """all-reduce(SUM) of the whole flat gradient through communicator `which`, blocking (cbm_comm_allreduce_grads)."""
        _chk(self.lib.cbm_comm_allreduce_grads(self.h, int(which)))

    def comm_overlap_probe(self, which=COMM_LEARNERS, iters=8):
        """(ms per backward pass alone, ms beside one whole-gradient all-reduce through `which`, us per all-reduce beside it): cbm_comm_overlap_probe."""
        o = (C.c_double * 3)()
        _chk(self.lib.cbm_comm_overlap_probe(self.h, int(which), int(iters), o))
        return o[0], o[1], o[2]

    def learner_allreduce_grads(self):
        d = C.c_float()
        _chk(self.lib.cbm_learner_allreduce_grads(self.h, C.byref(d)))
        return d.value

    def comm_profile(self, on=True):
        _chk(self.lib.cbm_comm_profile(self.h, int(bool(on))))

    def comm_profile_read(self):
        a, b, n = C.c_double(), C.c_double(), C.c_int32()
        _chk(self.lib.cbm_comm_profile_read(self.h, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value

    # ---- peer writes (split topologies)
    def ipc_export_window(self, window, tag=-1):
        """The blob of export window `window` (0: ring fields + actor parameter versions, 1: gradient / statistics / scratch); `tag` is quoted by
        the importing side's error messages (the exporting rank)."""
        b = (C.c_uint8 * IPC_WINDOW_BYTES)()
        _chk(self.lib.cbm_ipc_export_window(self.h, int(window), int(tag), b))
        return bytes(b)

    def ipc_window_offset(self, name, ring=0):
        w, off, n = C.c_int32(), C.c_int64(), C.c_int64()
        _chk(self.lib.cbm_ipc_window_offset(self.h, name.encode(), int(ring), C.byref(w), C.byref(off), C.byref(n)))
        return w.value, off.value, n.value

    def ipc_open_window(self, blob, what=""):
        p = C.c_void_p()
        buf = (C.c_uint8 * IPC_WINDOW_BYTES).from_buffer_copy(bytes(blob))
        _chk(self.lib.cbm_ipc_open_window(self.h, buf, what.encode(), C.byref(p)))
        return p.value

    def ipc_close_window(self, base):
        _chk(self.lib.cbm_ipc_close_window(self.h, C.c_void_p(base)))

    def unmap_peers(self):
        """cbm_ipc_close_all: every peer window this context mapped and the native communicators' peer mappings.  Multi-process teardown =
        unmap_peers() on every process, a host barrier, close()."""
        if self.h:
            _chk(self.lib.cbm_ipc_close_all(self.h))

    def actor_ship_shard(self, slot, ring, li, n_learners, peer_ring, dst_cols, dst_col0):
        _chk(self.lib.cbm_actor_ship_shard(self.h, int(slot), int(ring), int(li), int(n_learners), C.byref(peer_ring), int(dst_cols), int(dst_col0)))

    def io_sync(self):
        _chk(self.lib.cbm_io_sync(self.h))

    def params_push(self, peer_versions):
        arr = (C.c_void_p * 3)(*[C.c_void_p(p) for p in peer_versions])
        _chk(self.lib.cbm_params_push(self.h, arr))

    def params_mark_published(self):
        _chk(self.lib.cbm_params_mark_published(self.h))

    # ---- learner
    def learner_wait(self):
        _chk(self.lib.cbm_learner_wait(self.h))

    def learner_update(self, key, lrs, bc1, bc2, want_stats=True):
        key = np.ascontiguousarray(key, np.uint32).copy()
        lrs = np.ascontiguousarray(lrs, np.float32)
        bc1 = np.ascontiguousarray(bc1, np.float32)
        bc2 = np.ascontiguousarray(bc2, np.float32)
        w = 5 if self.cfg.algo == ALGO_PPO else 4
        stats = np.zeros((len(lrs) * max(1, self.cfg.grad_accum_steps), w), np.float32) if want_stats else None   # one row per micro-batch
        _chk(self.lib.cbm_learner_update(self.h, _p(key), _p(lrs), _p(bc1), _p(bc2), int(len(lrs)), _p(stats)))
        return key, stats

    def learner_prepare(self, key):
        key = np.ascontiguousarray(key, np.uint32).copy()
        _chk(self.lib.cbm_learner_prepare(self.h, _p(key)))
        return key

    def learner_epoch_begin(self, key):
        key = np.ascontiguousarray(key, np.uint32).copy()
        _chk(self.lib.cbm_learner_epoch_begin(self.h, _p(key)))
        return key

    def learner_minibatch_grad(self, epoch, mb):
        _chk(self.lib.cbm_learner_minibatch_grad(self.h, int(epoch), int(mb)))

    def grad_tail_offset(self):
        return int(self.lib.cbm_learner_grad_tail_offset(self.h))

    def learner_accumulate(self, mini_step, grad_div=1.0):
        _chk(self.lib.cbm_learner_accumulate(self.h, int(mini_step), C.c_float(grad_div)))

    def learner_optimizer_step(self, lr, bc1, bc2, grad_div=1.0):
        _chk(self.lib.cbm_learner_optimizer_step(self.h, C.c_float(lr), C.c_float(bc1), C.c_float(bc2), C.c_float(grad_div)))

    def learner_finish(self, n_rows, want_stats=True):
        w = 5 if self.cfg.algo == ALGO_PPO else 4
        stats = np.zeros((n_rows * max(1, self.cfg.grad_accum_steps), w), np.float32) if want_stats else None
        _chk(self.lib.cbm_learner_finish(self.h, _p(stats)))
        return stats


# ---- host twin of the synthetic env (CPU code inside the same library; no GPU needed)
def synth_env_reset_host(seed, n, atari57_mix=False):
    st = (EnvState * n)()
    obs = np.zeros((n, 4, 84, 84), np.uint8)
    _chk(load().cbm_synth_env_reset_host_games(C.c_uint32(int(seed) & 0xFFFFFFFF), int(n), int(bool(atari57_mix)), st, _p(obs)))
    return st, obs


def synth_env_step_host(seed, st, obs, actions, max_episode_steps=27000):
    n = obs.shape[0]
    actions = np.ascontiguousarray(actions, np.int32)
    reward = np.zeros(n, np.float32)
    done = np.zeros(n, np.uint8)
    term = np.zeros(n, np.uint8)
    elapsed = np.zeros(n, np.int32)
    _chk(load().cbm_synth_env_step_host(C.c_uint32(int(seed) & 0xFFFFFFFF), int(n), int(max_episode_steps), _p(actions), st,
                                        _p(obs), _p(reward), _p(done), _p(term), _p(elapsed)))
    return reward, done, term, elapsed


def synth_env_render_host(state, layered):
    plane = np.empty((84, 84), np.uint8)
    _chk(load().cbm_synth_env_render_host(C.byref(state), int(layered), _p(plane)))   # 0 per pixel, 1 the host's layers, 2 the kernels' word painter
    return plane


def synth_env_step_host_to(seed, st, obs_prev, actions, max_episode_steps=27000, out=None):
    """Out-of-place step: returns (obs_next, reward, done, terminated, elapsed) with obs_next a fresh array (envpool's recv() contract) or `out`."""
    n = obs_prev.shape[0]
    actions = np.ascontiguousarray(actions, np.int32)
    obs_next = np.empty_like(obs_prev) if out is None else out
    reward = np.zeros(n, np.float32)
    done = np.zeros(n, np.uint8)
    term = np.zeros(n, np.uint8)
    elapsed = np.zeros(n, np.int32)
    _chk(load().cbm_synth_env_step_host_to(C.c_uint32(int(seed) & 0xFFFFFFFF), int(n), int(max_episode_steps), _p(actions), st,
                                           _p(obs_prev), _p(obs_next), _p(reward), _p(done), _p(term), _p(elapsed)))
    return obs_next, reward, done, term, elapsed


def synth_env_step_host_ids(seed, st, obs, env_ids, actions, max_episode_steps=27000):
    """Steps only the listed envs of the (st, obs) arrays that hold all of them — envpool's send(action, env_id) in async mode."""
    env_ids = np.ascontiguousarray(env_ids, np.int32)
    actions = np.ascontiguousarray(actions, np.int32)
    k = env_ids.size
    reward = np.zeros(k, np.float32)
    done = np.zeros(k, np.uint8)
    term = np.zeros(k, np.uint8)
    elapsed = np.zeros(k, np.int32)
    _chk(load().cbm_synth_env_step_host_ids(C.c_uint32(int(seed) & 0xFFFFFFFF), int(obs.shape[0]), int(k), int(max_episode_steps), _p(env_ids),
                                            _p(actions), st, _p(obs), _p(reward), _p(done), _p(term), _p(elapsed)))
    return reward, done, term, elapsed
