"""stable-baselines3 2.0.0 zip checkpoint reader / writer (no SB3, no gymnasium needed).

Format (SURVEY.md §5 "Checkpoint / resume", verified by unzipping /root/reference/data/policies/*.zip):
an uncompressed (STORED) zip with
    data                        JSON; non-JSON values as {":type:", ":serialized:" base64(cloudpickle), ...}
    pytorch_variables.pth       torch.save({})
    policy.pth                  torch.save(OrderedDict of the 13 tensors, SB3 key order)
    policy.optimizer.pth        torch.save({"state": {i: {step, exp_avg, exp_avg_sq}}, "param_groups": [...]})
    _stable_baselines3_version  "2.0.0"
    system_info.txt
Reference call sites: `PPO.save` via src/mobrob/rl_control/ppo.py:76-77 and examples/train.py:36-41,49;
`PPO.load` via src/mobrob/utils.py:15-16 and examples/train.py:30-33.

Writing blobs that real SB3 can unpickle without having SB3/gymnasium installed:
  * policy_class          by-reference pickle of stable_baselines3.common.policies.ActorCriticPolicy (70 bytes)
  * observation/action_space   by-value pickles of gymnasium.spaces.box.Box, emitted opcode-by-opcode in the
                          same layout the reference blobs have (NEWOBJ + state dict), with ndarray payloads going
                          through `numpy.core.numeric._frombuffer` so that NumPy 1.24 (the reference's) and 2.x
                          both load them
  * learning_rate / clip_range are written as plain JSON floats (SB3 turns floats into constant schedules in
    `_setup_model`); `lr_schedule` is omitted (SB3 rebuilds it from `learning_rate` on load)
"""
from __future__ import annotations

import base64
import collections
import io
import json
import pickle
import struct
import time
import warnings
import zipfile
from collections import OrderedDict

import numpy as np

SB3_VERSION = "2.0.0"
POLICY_KEYS = ["log_std",
               "mlp_extractor.policy_net.0.weight", "mlp_extractor.policy_net.0.bias",
               "mlp_extractor.policy_net.2.weight", "mlp_extractor.policy_net.2.bias",
               "mlp_extractor.value_net.0.weight", "mlp_extractor.value_net.0.bias",
               "mlp_extractor.value_net.2.weight", "mlp_extractor.value_net.2.bias",
               "action_net.weight", "action_net.bias", "value_net.weight", "value_net.bias"]


def policy_keys(n_hidden_pi=2, n_hidden_vf=2):
    """SB3's registration order for `net_arch` depths other than the reference's two hidden layers (`nn.Sequential` indices 0, 2,
    4 ...: every Linear is followed by its activation module); POLICY_KEYS == policy_keys(2, 2)."""
    keys = ["log_std"]
    for net, n in (("policy_net", n_hidden_pi), ("value_net", n_hidden_vf)):
        for i in range(n):
            keys += [f"mlp_extractor.{net}.{2 * i}.weight", f"mlp_extractor.{net}.{2 * i}.bias"]
    return keys + ["action_net.weight", "action_net.bias", "value_net.weight", "value_net.bias"]


def depth_of_keys(keys):
    """(policy hidden layers, value hidden layers) of a state-dict key list."""
    return (sum(1 for k in keys if k.startswith("mlp_extractor.policy_net.") and k.endswith(".weight")),
            sum(1 for k in keys if k.startswith("mlp_extractor.value_net.") and k.endswith(".weight")))


# ---------------------------------------------------------------------------------------------------
# tiny pickle assembler (protocol 5, same opcodes as the reference blobs; payloads are bytearrays)
# ---------------------------------------------------------------------------------------------------
def _s(x: str) -> bytes:  # SHORT_BINUNICODE, BINUNICODE for 256 bytes and more
    b = x.encode()
    if len(b) < 256:
        return b"\x8c" + bytes([len(b)]) + b
    return b"X" + struct.pack("<I", len(b)) + b


def _glob(mod: str, name: str) -> bytes:  # STACK_GLOBAL
    return _s(mod) + _s(name) + b"\x93"


def _int(i: int) -> bytes:
    if 0 <= i < 256:
        return b"K" + bytes([i])
    return b"J" + struct.pack("<i", i)


def _dtype(code: str, byteorder: str) -> bytes:
    # numpy.dtype(code, False, True) then BUILD with (3, byteorder, None, None, None, -1, -1, 0)
    return (_glob("numpy", "dtype") + _s(code) + b"\x89\x88\x87R" + b"(" + _int(3) + _s(byteorder) + b"NNN" +
            _int(-1) + _int(-1) + _int(0) + b"t" + b"b")


def _ndarray(a: np.ndarray) -> bytes:
    a = np.ascontiguousarray(a)
    code = {"float32": ("f4", "<"), "float64": ("f8", "<"), "bool": ("b1", "|"), "int64": ("i8", "<")}[str(a.dtype)]
    raw = a.tobytes()
    shape = b"(" + b"".join(_int(int(s)) for s in a.shape) + b"t"
    return (_glob("numpy.core.numeric", "_frombuffer") + b"(" + b"\x96" + struct.pack("<Q", len(raw)) + raw +
            _dtype(*code) + shape + _s("C") + b"t" + b"R")


def _wrap(body: bytes) -> bytes:
    return b"\x80\x05" + body + b"."


def pickle_ndarray(a) -> bytes:
    return _wrap(_ndarray(np.asarray(a)))


def short_repr(x) -> str:
    """gymnasium 0.28.1 `spaces.box._short_repr`: the scalar when every bound is the same number, else str(array)."""
    x = np.asarray(x)
    return str(np.min(x)) if x.size != 0 and np.min(x) == np.max(x) else str(x)


def box_state(low, high, dtype=np.float32) -> "OrderedDict":
    """The nine-key `__dict__` of a gymnasium 0.28.1 Box(low, high) (verified against the by-value pickles inside
    /root/reference/data/policies/*.zip, tests/test_checkpoint.py)."""
    low, high = np.asarray(low, dtype), np.asarray(high, dtype)
    return OrderedDict([("dtype", np.dtype(dtype)), ("bounded_below", np.isfinite(low)), ("bounded_above", np.isfinite(high)),
                        ("_shape", tuple(int(s) for s in low.shape)), ("low", low), ("high", high),
                        ("low_repr", short_repr(low)), ("high_repr", short_repr(high)), ("_np_random", None)])


def pickle_box(low, high, dtype=np.float32) -> bytes:
    """gymnasium.spaces.box.Box(low, high) by value: NEWOBJ + the 9-key state dict of gymnasium 0.28.1."""
    low = np.asarray(low, dtype)
    high = np.asarray(high, dtype)
    rep = short_repr

    items = [(_s("dtype"), _dtype("f4", "<")),
             (_s("bounded_below"), _ndarray(np.isfinite(low))),
             (_s("bounded_above"), _ndarray(np.isfinite(high))),
             (_s("_shape"), b"(" + b"".join(_int(int(s)) for s in low.shape) + b"t"),
             (_s("low"), _ndarray(low)), (_s("high"), _ndarray(high)),
             (_s("low_repr"), _s(rep(low))), (_s("high_repr"), _s(rep(high))),
             (_s("_np_random"), b"N")]
    body = _glob("gymnasium.spaces.box", "Box") + b")\x81" + b"}" + b"(" + b"".join(k + v for k, v in items) + b"u" + b"b"
    return _wrap(body)


def pickle_policy_class() -> bytes:
    return _wrap(_glob("stable_baselines3.common.policies", "ActorCriticPolicy"))


def _blob(type_repr: str, payload: bytes, **extra) -> dict:
    d = {":type:": type_repr, ":serialized:": base64.b64encode(payload).decode()}
    d.update(extra)
    return d


def box_entry(low, high) -> dict:
    """The JSON entry SB3's `data_to_json` writes for a Box: the by-value pickle plus str() of every state item."""
    st = box_state(low, high)
    extra = {k: (str(v) if isinstance(v, (np.ndarray, np.dtype)) else (list(v) if isinstance(v, tuple) else v))
             for k, v in st.items()}
    return _blob("<class 'gymnasium.spaces.box.Box'>", pickle_box(low, high), **extra)


class _BoxStandIn:
    """What a `gymnasium.spaces.box.Box` pickle is rebuilt into when gymnasium is absent: just its state."""

    def __setstate__(self, state):
        self.__dict__.update(state)


# Exactly the globals the blobs of an SB3 2.0.0 checkpoint reference (listed from the five reference zips with
# pickletools) plus what this module's own writer emits: array / dtype / scalar reconstructors, the PCG64 generator state
# a gymnasium Box carries, the deque of Monitor records.  Nothing else resolves -- in particular none of numpy's file or
# nested-pickle entry points (numpy.load / save / fromfile ...), which a "numpy.*" prefix rule would let a crafted blob
# REDUCE into.
_ALLOWED_GLOBALS = {
    ("numpy", "dtype"), ("numpy", "ndarray"),
    ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    ("numpy.random._pickle", "__generator_ctor"), ("numpy.random._pickle", "__bit_generator_ctor"),
    ("numpy.random._pickle", "__randomstate_ctor"),
    ("numpy.random._pcg64", "PCG64"), ("numpy.random._generator", "Generator"),
    ("numpy.random.bit_generator", "SeedSequence"), ("numpy.random._mt19937", "MT19937"),
    ("collections", "deque"), ("collections", "OrderedDict"),
    # `policy_kwargs` holding an activation class is cloudpickled as a whole by SB3: the class travels by reference
    *(("torch.nn.modules.activation", cls) for cls in ("Tanh", "ReLU", "ELU", "LeakyReLU", "Sigmoid", "Softplus", "Softsign",
                                                        "Hardtanh", "ReLU6", "SiLU", "GELU", "Mish")),
}


class _SpaceUnpickler(pickle.Unpickler):
    """Restricted unpickler for EVERY blob of a checkpoint's `data` JSON: an exact (module, name) allow-list and the Box
    stand-in.  Blobs that need anything else (SB3's cloudpickled schedules, `policy_class`) are not decoded at all."""

    def find_class(self, module, name):
        if (module, name) == ("gymnasium.spaces.box", "Box"):
            return _BoxStandIn
        if (module, name) in _ALLOWED_GLOBALS:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                return super().find_class(module, name)
        raise pickle.UnpicklingError(f"{module}.{name} is not allowed in a checkpoint blob")


def unpickle_box(entry) -> dict:
    """A Box JSON entry (or raw pickle bytes) -> its state dict (low, high, dtype, _shape, bounded_*, *_repr)."""
    raw = base64.b64decode(entry[":serialized:"]) if isinstance(entry, dict) else entry
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        obj = _SpaceUnpickler(io.BytesIO(raw)).load()
    return dict(obj.__dict__)


def _unblob(entry):
    """`_last_obs`, `_last_episode_starts`, `ep_info_buffer` ...: arrays and deques of plain records, through the same
    allow-list as the spaces (a checkpoint is not trusted more in one entry than in another)."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return _SpaceUnpickler(io.BytesIO(base64.b64decode(entry[":serialized:"]))).load()


# ---------------------------------------------------------------------------------------------------
# read
# ---------------------------------------------------------------------------------------------------
def load_zip(path):
    """-> dict(data=<plain hyper-parameters + decoded arrays>, params=OrderedDict[str, np.ndarray],
               optimizer=dict(exp_avg, exp_avg_sq, step, param_groups) | None)"""
    import torch
    if not str(path).endswith(".zip"):
        path = str(path) + ".zip"
    with zipfile.ZipFile(path) as z:
        names = set(z.namelist())
        raw = json.loads(z.read("data").decode())
        data = {}
        for k, v in raw.items():
            if isinstance(v, dict) and ":serialized:" in v:
                if k in ("_last_obs", "_last_episode_starts", "_last_original_obs", "ep_info_buffer", "ep_success_buffer"):
                    try:
                        data[k] = _unblob(v)
                    except Exception:  # blobs that need gymnasium/SB3/py3.11 code objects are skipped, like SB3 does
                        data[k] = None
                elif k == "policy_kwargs":  # a dict with a class in it (activation_fn) is ONE blob in SB3's JSON
                    try:
                        pk = dict(_unblob(v))
                        if "activation_fn" in pk:
                            pk["activation_fn"] = getattr(pk["activation_fn"], "__name__", str(pk["activation_fn"]))
                        data[k] = pk
                    except Exception:  # noqa: BLE001 - something this build cannot rebuild: keep the readable side
                        data[k] = {kk: vv for kk, vv in v.items() if not kk.startswith(":")}
                elif k in ("observation_space", "action_space"):
                    try:
                        st = unpickle_box(v)
                        data[k] = {"shape": tuple(st["_shape"]), "low": np.asarray(st["low"]), "high": np.asarray(st["high"]),
                                   "dtype": np.dtype(st["dtype"]), "low_repr": st["low_repr"], "high_repr": st["high_repr"]}
                    except Exception:  # noqa: BLE001 - unknown blob layout: fall back to the JSON side
                        data[k] = {"shape": tuple(v.get("_shape", ())), "low": None, "high": None,
                                   "low_repr": v.get("low_repr"), "high_repr": v.get("high_repr")}
                else:
                    data[k] = None
            else:
                data[k] = v
        sd = torch.load(io.BytesIO(z.read("policy.pth")), map_location="cpu", weights_only=True)
        params = OrderedDict((k, t.detach().cpu().numpy().astype(np.float32).copy()) for k, t in sd.items())
        opt = None
        if "policy.optimizer.pth" in names:
            o = torch.load(io.BytesIO(z.read("policy.optimizer.pth")), map_location="cpu", weights_only=True)
            if o.get("state"):
                keys = list(params.keys())
                opt = dict(exp_avg=OrderedDict((k, o["state"][i]["exp_avg"].numpy().copy()) for i, k in enumerate(keys)),
                           exp_avg_sq=OrderedDict((k, o["state"][i]["exp_avg_sq"].numpy().copy()) for i, k in enumerate(keys)),
                           step=int(float(o["state"][0]["step"])), param_groups=o["param_groups"])
            else:
                opt = dict(exp_avg=None, exp_avg_sq=None, step=0, param_groups=o.get("param_groups"))
        version = z.read("_stable_baselines3_version").decode() if "_stable_baselines3_version" in names else None
    data["_sb3_version"] = version
    return dict(data=data, params=params, optimizer=opt)


# ---------------------------------------------------------------------------------------------------
# write
# ---------------------------------------------------------------------------------------------------
def save_zip(path, *, params, optimizer, hyper, obs_dim, act_dim, net_arch=None, counters=None, last_obs=None,
             last_episode_starts=None, ep_info_buffer=None, action_low=-1.0, action_high=1.0, verbose=1, seed=0,
             tensorboard_log=None, obs_low=None, obs_high=None, extra_policy_kwargs=None):
    """Write an SB3-2.0.0-layout zip.  `hyper` carries n_steps, batch_size, n_epochs, gamma, gae_lambda,
    ent_coef, vf_coef, max_grad_norm, learning_rate, clip_range, n_envs; `optimizer` = dict(exp_avg, exp_avg_sq,
    step, lr, betas, eps) with per-key arrays.  obs_low / obs_high: the bounds of `env.observation_space` (what
    `PPO.save` stores and `PPO.load(path, env=...)` checks with `check_for_correct_spaces`); default unbounded."""
    import torch
    if not str(path).endswith(".zip"):
        path = str(path) + ".zip"
    counters = dict(counters or {})
    keys = list(params.keys())
    assert keys == policy_keys(*depth_of_keys(keys)), "parameters must be in SB3 registration order"
    obs_low = np.broadcast_to(np.asarray(-np.inf if obs_low is None else obs_low, np.float32), (obs_dim,)).copy()
    obs_high = np.broadcast_to(np.asarray(np.inf if obs_high is None else obs_high, np.float32), (obs_dim,)).copy()
    act_low = np.broadcast_to(np.asarray(action_low, np.float32), (act_dim,)).copy()
    act_high = np.broadcast_to(np.asarray(action_high, np.float32), (act_dim,)).copy()
    n_envs = int(hyper["n_envs"])
    if last_obs is None:
        last_obs = np.zeros((n_envs, obs_dim), np.float32)
    if last_episode_starts is None:
        last_episode_starts = np.zeros((n_envs,), bool)
    policy_kwargs = {} if net_arch is None else {"net_arch": {"pi": list(net_arch[0]), "vf": list(net_arch[1])}}
    policy_kwargs.update(extra_policy_kwargs or {})  # JSON-able ones the user passed: log_std_init, ortho_init, optimizer_kwargs
    act = policy_kwargs.get("activation_fn")
    if act is not None:  # SB3 pickles a policy_kwargs dict that holds a class as ONE blob (the class by reference) + readable keys
        from .engine import ACTIVATIONS, activation_name
        cls = getattr(torch.nn, ACTIVATIONS[activation_name(act)][1])
        real = dict(policy_kwargs, activation_fn=cls)
        readable = {k: (str(cls) if k == "activation_fn" else v) for k, v in policy_kwargs.items()}
        policy_kwargs = _blob("<class 'dict'>", pickle.dumps(real, protocol=4), **readable)
    data = OrderedDict()
    data["policy_class"] = _blob("<class 'abc.ABCMeta'>", pickle_policy_class(),
                                 __module__="stable_baselines3.common.policies")
    data["verbose"] = int(verbose)
    data["policy_kwargs"] = policy_kwargs
    data["num_timesteps"] = int(counters.get("num_timesteps", 0))
    data["_total_timesteps"] = int(counters.get("_total_timesteps", 0))
    data["_num_timesteps_at_start"] = int(counters.get("_num_timesteps_at_start", 0))
    data["seed"] = seed
    data["action_noise"] = None
    data["start_time"] = int(counters.get("start_time", time.time_ns()))
    data["learning_rate"] = float(hyper["learning_rate"])
    data["tensorboard_log"] = tensorboard_log
    data["_last_obs"] = _blob("<class 'numpy.ndarray'>", pickle_ndarray(np.asarray(last_obs, np.float32)))
    data["_last_episode_starts"] = _blob("<class 'numpy.ndarray'>", pickle_ndarray(np.asarray(last_episode_starts, bool)))
    data["_last_original_obs"] = None
    data["_episode_num"] = int(counters.get("_episode_num", 0))
    data["use_sde"] = bool(hyper.get("use_sde", False))
    data["sde_sample_freq"] = int(hyper.get("sde_sample_freq", -1))
    data["_current_progress_remaining"] = float(counters.get("_current_progress_remaining", 1.0))
    data["_stats_window_size"] = 100
    buf = collections.deque(ep_info_buffer or [], maxlen=100)
    data["ep_info_buffer"] = _blob("<class 'collections.deque'>", pickle.dumps(buf, protocol=4))
    data["ep_success_buffer"] = _blob("<class 'collections.deque'>", pickle.dumps(collections.deque(maxlen=100), protocol=4))
    data["_n_updates"] = int(counters.get("_n_updates", 0))
    for k in ("n_steps", "gamma", "gae_lambda", "ent_coef", "vf_coef", "max_grad_norm", "batch_size", "n_epochs"):
        data[k] = hyper[k]
    data["clip_range"] = float(hyper["clip_range"])
    data["clip_range_vf"] = hyper.get("clip_range_vf")  # float or None (schedules are stored by their current value)
    data["normalize_advantage"] = bool(hyper.get("normalize_advantage", True))
    data["target_kl"] = hyper.get("target_kl")
    data["observation_space"] = box_entry(obs_low, obs_high)
    data["action_space"] = box_entry(act_low, act_high)
    data["n_envs"] = n_envs

    sd = OrderedDict((k, torch.from_numpy(np.ascontiguousarray(params[k], dtype=np.float32)).clone()) for k in keys)
    state = {}
    if optimizer.get("exp_avg") is not None and int(optimizer.get("step", 0)) > 0:
        for i, k in enumerate(keys):
            state[i] = {"step": torch.tensor(float(optimizer["step"])),
                        "exp_avg": torch.from_numpy(np.ascontiguousarray(optimizer["exp_avg"][k], np.float32)).clone(),
                        "exp_avg_sq": torch.from_numpy(np.ascontiguousarray(optimizer["exp_avg_sq"][k], np.float32)).clone()}
    opt_sd = {"state": state,
              "param_groups": [{"lr": float(optimizer.get("lr", hyper["learning_rate"])),
                                "betas": tuple(optimizer.get("betas", (0.9, 0.999))), "eps": float(optimizer.get("eps", 1e-5)),
                                "weight_decay": 0, "amsgrad": False, "maximize": False, "foreach": None,
                                "capturable": False, "differentiable": False, "fused": None,
                                "params": list(range(len(keys)))}]}

    def tsave(obj):
        b = io.BytesIO()
        torch.save(obj, b)
        return b.getvalue()

    import platform
    info = (f"- OS: {platform.platform()}\n- Python: {platform.python_version()}\n- Stable-Baselines3: {SB3_VERSION} "
            f"(written by mobrob_amd, MI355X-native engine)\n- PyTorch: {torch.__version__}\n- GPU Enabled: True\n"
            f"- Numpy: {np.__version__}\n")
    with zipfile.ZipFile(path, "w", compression=zipfile.ZIP_STORED) as z:
        z.writestr("data", json.dumps(data, indent=4))
        z.writestr("pytorch_variables.pth", tsave({}))
        z.writestr("policy.pth", tsave(sd))
        z.writestr("policy.optimizer.pth", tsave(opt_sd))
        z.writestr("_stable_baselines3_version", SB3_VERSION)
        z.writestr("system_info.txt", info)
    return path
