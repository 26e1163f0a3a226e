"""`.cleanrl_model` writer / reader (ppo:753-771; loader cleanrl_utils/evals/ppo_envpool_jax_eval.py:35-38).

The reference writes flax.serialization.to_bytes([vars(args), [network_params, actor_params, critic_params]]):
msgpack of the state dict (lists become {"0":..,"1":..} dicts), every ndarray as ExtType(1) whose payload is
msgpack((shape, dtype.name, C-order bytes)) (flax 0.6.8 serialization._ndarray_to_bytes).  Parameter tree
names are flax's auto names, so a model trained here loads in the reference's eval script.
"""
import msgpack
import numpy as np

from . import model as M


def _to_state(x):
    if isinstance(x, (list, tuple)):
        return {str(i): _to_state(v) for i, v in enumerate(x)}
    if isinstance(x, dict):
        return {str(k): _to_state(v) for k, v in x.items()}
    return x


def _pack_ext(x):
    if isinstance(x, np.ndarray):
        return msgpack.ExtType(1, msgpack.packb((list(x.shape), x.dtype.name, x.tobytes("C")), use_bin_type=True))
    if isinstance(x, np.generic):
        a = np.asarray(x)
        return msgpack.ExtType(3, msgpack.packb((list(a.shape), a.dtype.name, a.tobytes("C")), use_bin_type=True))
    raise TypeError(f"cannot serialise {type(x)}")


def _unpack_ext(code, data):
    if code in (1, 3):
        shape, dtype, buf = msgpack.unpackb(data, raw=False)
        a = np.frombuffer(buf, dtype=np.dtype(dtype)).reshape(shape)
        return a if code == 1 else a[()]
    return msgpack.ExtType(code, data)


def save_cleanrl_model(path, args, flat_params, num_actions, network="nature"):
    to_tree = M.params_to_flax_tree if network == "nature" else M.resnet_params_to_flax_tree
    tree = [dict(vars(args)), to_tree(np.asarray(flat_params, np.float32), num_actions)]
    with open(path, "wb") as f:
        f.write(msgpack.packb(_to_state(tree), default=_pack_ext, strict_types=True))


def load_cleanrl_model(path, num_actions, network="nature"):
    with open(path, "rb") as f:
        state = msgpack.unpackb(f.read(), ext_hook=_unpack_ext, raw=False)
    args_dict, params = state["0"], state["1"]
    tree = [params["0"], params["1"], params["2"]]
    from_tree = M.flax_tree_to_params if network == "nature" else M.resnet_flax_tree_to_params
    return args_dict, from_tree(tree, num_actions)
