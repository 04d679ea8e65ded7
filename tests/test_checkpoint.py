"""CPU: SB3-zip checkpoint reader/writer (mobrob_amd/checkpoint.py)."""
import io
import json
import os
import pickle
import sys
import types
import zipfile
from collections import OrderedDict

import numpy as np
import pytest

from mobrob_amd import checkpoint as ck
from tests.util import ENVS, load_golden
from oracle import ppo_oracle as O

REF = "/root/reference/data/policies"


@pytest.mark.parametrize("env", ENVS)
def test_reads_reference_zip(env):
    if not os.path.isdir(REF):
        pytest.skip("reference checkpoints are only present in the build container")
    g = load_golden(env)
    c = ck.load_zip(f"{REF}/{env}-ppo.zip")
    assert list(c["params"].keys()) == ck.POLICY_KEYS == O.param_keys()
    for k in ck.POLICY_KEYS:
        assert np.array_equal(c["params"][k], g["p/" + k])
        assert np.array_equal(c["optimizer"]["exp_avg"][k], g["m/" + k])
        assert np.array_equal(c["optimizer"]["exp_avg_sq"][k], g["v/" + k])
    assert c["optimizer"]["step"] == int(g["adam_step"])
    assert c["data"]["n_steps"] == int(g["hyper/n_steps"]) and c["data"]["_sb3_version"] == "2.0.0"
    assert np.allclose(np.asarray(c["data"]["_last_obs"], np.float32), g["last_obs"])
    assert len(c["data"]["ep_info_buffer"]) == 100
    assert c["data"]["observation_space"]["shape"] == (g["last_obs"].shape[1],)


def test_write_read_round_trip(tmp_path):
    g = load_golden("doggo")
    params = OrderedDict((k, g["p/" + k]) for k in ck.POLICY_KEYS)
    m = OrderedDict((k, g["m/" + k]) for k in ck.POLICY_KEYS)
    v = OrderedDict((k, g["v/" + k]) for k in ck.POLICY_KEYS)
    hyper = dict(n_steps=1000, batch_size=100, n_epochs=5, gamma=0.99, gae_lambda=0.95, ent_coef=0.01, vf_coef=0.5,
                 max_grad_norm=0.5, learning_rate=3e-4, clip_range=0.2, n_envs=16)
    path = ck.save_zip(str(tmp_path / "doggo-ppo"), params=params,
                       optimizer=dict(exp_avg=m, exp_avg_sq=v, step=1499200), hyper=hyper, obs_dim=58, act_dim=12,
                       net_arch=((64, 64), (64, 64)), counters=dict(num_timesteps=123, _n_updates=7),
                       last_obs=g["last_obs"], ep_info_buffer=[{"r": 1.0, "l": 5, "t": 0.1}])
    assert path.endswith(".zip")
    with zipfile.ZipFile(path) as z:
        assert [i.filename for i in z.infolist()] == ["data", "pytorch_variables.pth", "policy.pth",
                                                      "policy.optimizer.pth", "_stable_baselines3_version",
                                                      "system_info.txt"]
        assert all(i.compress_type == zipfile.ZIP_STORED for i in z.infolist())
        d = json.loads(z.read("data"))
        assert z.read("_stable_baselines3_version") == b"2.0.0"
    ref_keys = ['policy_class', 'verbose', 'policy_kwargs', 'num_timesteps', '_total_timesteps',
                '_num_timesteps_at_start', 'seed', 'action_noise', 'start_time', 'learning_rate', 'tensorboard_log',
                '_last_obs', '_last_episode_starts', '_last_original_obs', '_episode_num', 'use_sde',
                'sde_sample_freq', '_current_progress_remaining', '_stats_window_size', 'ep_info_buffer',
                'ep_success_buffer', '_n_updates', 'n_steps', 'gamma', 'gae_lambda', 'ent_coef', 'vf_coef',
                'max_grad_norm', 'batch_size', 'n_epochs', 'clip_range', 'clip_range_vf', 'normalize_advantage',
                'target_kl', 'observation_space', 'action_space', 'n_envs']
    assert list(d.keys()) == ref_keys  # the reference zips' key order minus lr_schedule (rebuilt by SB3 on load)
    c = ck.load_zip(path)
    for k in ck.POLICY_KEYS:
        assert np.array_equal(c["params"][k], params[k])
        assert np.array_equal(c["optimizer"]["exp_avg"][k], m[k])
    assert c["optimizer"]["step"] == 1499200 and c["data"]["num_timesteps"] == 123
    assert np.array_equal(c["data"]["_last_obs"], g["last_obs"])
    assert c["data"]["policy_kwargs"] == {"net_arch": {"pi": [64, 64], "vf": [64, 64]}}


REFERENCE_SPACES = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_spaces.json")))


def _assert_same_box_state(mine, ref, tag):
    assert sorted(mine) == sorted(ref) == sorted(["dtype", "bounded_below", "bounded_above", "_shape", "low", "high",
                                                  "low_repr", "high_repr", "_np_random"]), tag
    for k in ref:
        if k == "_np_random":        # a seeded generator in some reference action spaces: state, not identity of a Box
            continue
        a, b = mine[k], ref[k]
        if isinstance(b, np.ndarray):
            assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b), (tag, k)
        else:
            assert a == b, (tag, k, a, b)


@pytest.mark.parametrize("env", ENVS)
def test_space_blobs_equal_the_reference_held_ones(env):
    """The Box pickles the writer emits for each robot rebuild EXACTLY the state gymnasium 0.28.1 / SB3 2.0.0 wrote
    into the reference checkpoints (tests/golden/reference_spaces.json = the `observation_space` / `action_space`
    entries of /root/reference/data/policies/<env>-ppo.zip): dtype, shape, low / high arrays -- finite for drone and
    turtlebot3 --, bounded masks and the repr strings; the JSON side of the entry matches key by key as well."""
    from mobrob_amd.envs.wrapper import get_env, observation_space_of
    robot = get_env(env)
    for key, space in (("observation_space", robot.observation_space), ("action_space", robot.action_space)):
        ref_entry = REFERENCE_SPACES[env][key]
        ref = ck.unpickle_box(ref_entry)
        entry = ck.box_entry(space.low, space.high)
        _assert_same_box_state(ck.unpickle_box(entry), ref, (env, key))
        for k, v in ref_entry.items():
            if k not in (":serialized:", "_np_random"):
                assert entry[k] == v, (env, key, k)
    assert np.array_equal(observation_space_of(env).low, robot.observation_space.low)
    if env in ("drone", "turtlebot3"):
        assert np.isfinite(robot.observation_space.low).all() and np.isfinite(robot.observation_space.high).all()


def test_space_unpickler_refuses_foreign_globals():
    evil = pickle.dumps(os.getcwd)   # a by-reference global outside numpy / the Box stand-in
    with pytest.raises(pickle.UnpicklingError):
        ck.unpickle_box(evil)


def test_zip_keeps_the_observation_bounds_of_the_robot(tmp_path):
    """save -> load keeps finite bounds (they used to be overwritten with +-inf), also through a reference-zip round
    trip when the reference checkpoints are present."""
    from mobrob_amd.envs.wrapper import observation_space_of
    g = load_golden("drone")
    params = OrderedDict((k, g["p/" + k]) for k in ck.POLICY_KEYS)
    zeros = OrderedDict((k, np.zeros_like(v)) for k, v in params.items())
    hyper = dict(n_steps=1000, batch_size=100, n_epochs=5, gamma=0.99, gae_lambda=0.99, ent_coef=0.01, vf_coef=0.5,
                 max_grad_norm=0.5, learning_rate=3e-4, clip_range=0.2, n_envs=16)
    sp = observation_space_of("drone")
    path = ck.save_zip(str(tmp_path / "drone-ppo"), params=params, optimizer=dict(exp_avg=zeros, exp_avg_sq=zeros, step=0),
                       hyper=hyper, obs_dim=12, act_dim=18, obs_low=sp.low, obs_high=sp.high)
    c = ck.load_zip(path)
    assert np.array_equal(c["data"]["observation_space"]["low"], sp.low) and c["data"]["observation_space"]["low"][2] == -50.0
    assert np.array_equal(c["data"]["action_space"]["high"], np.ones(18, np.float32))
    with zipfile.ZipFile(path) as z:
        entry = json.loads(z.read("data"))["observation_space"]
    _assert_same_box_state(ck.unpickle_box(entry), ck.unpickle_box(REFERENCE_SPACES["drone"]["observation_space"]), "drone")
    if os.path.isdir(REF):
        r = ck.load_zip(f"{REF}/turtlebot3-ppo.zip")
        assert abs(float(r["data"]["observation_space"]["low"][2]) + 2 ** 0.5) < 1e-6


def test_ndarray_pickle_round_trip():
    assert pickle.loads(ck.pickle_ndarray(np.array([[1.5, 2.5]], np.float32))).tolist() == [[1.5, 2.5]]


def test_checkpoint_blobs_resolve_only_an_exact_allow_list():
    """A blob may name only the reconstructors SB3 checkpoints really use: `numpy.load`, `numpy.save`, `os.system` ... do
    not resolve, whichever entry of the zip carries them (spaces and `_last_obs`-style blobs go through the same unpickler)."""
    import base64
    import pickle
    from mobrob_amd import checkpoint as ck

    class _Evil:
        def __init__(self, fn, *args):
            self.fn, self.args = fn, args

        def __reduce__(self):
            return self.fn, self.args

    import os as _os
    for fn, args in ((np.load, ("/etc/passwd",)), (np.save, ("/tmp/x", 1)), (np.fromfile, ("/etc/passwd",)), (_os.system, ("true",))):
        blob = pickle.dumps(_Evil(fn, *args))
        entry = {":serialized:": base64.b64encode(blob).decode()}
        with pytest.raises(pickle.UnpicklingError):
            ck.unpickle_box(entry)
        with pytest.raises(pickle.UnpicklingError):
            ck._unblob(entry)
    # what the reference zips do contain still decodes: arrays, deques of Monitor records
    from collections import deque
    ok = {":serialized:": base64.b64encode(pickle.dumps(deque([{"r": 1.5, "l": 7, "t": 0.1}], maxlen=100))).decode()}
    assert list(ck._unblob(ok)) == [{"r": 1.5, "l": 7, "t": 0.1}]
    arr = {":serialized:": base64.b64encode(pickle.dumps(np.arange(6, dtype=np.float32).reshape(2, 3))).decode()}
    assert ck._unblob(arr).tolist() == [[0.0, 1.0, 2.0], [3.0, 4.0, 5.0]]


def test_policy_kwargs_with_an_activation_class_round_trip(tmp_path):
    """SB3 pickles a `policy_kwargs` dict that holds a class (activation_fn=nn.ReLU) as ONE blob with the class by
    reference; the writer emits that form and the reader gets net_arch / activation name back (allow-listed globals only)."""
    import base64
    import pickle
    import torch
    p = O.init_params(14, 2, seed=0)
    zeros = OrderedDict((k, np.zeros_like(v)) for k, v in p.items())
    hyper = dict(n_envs=2, n_steps=8, gamma=0.99, gae_lambda=0.95, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, batch_size=4,
                 n_epochs=1, clip_range=0.2, learning_rate=3e-4)
    path = str(tmp_path / "relu.zip")
    ck.save_zip(path, params=p, optimizer=dict(exp_avg=zeros, exp_avg_sq=zeros, step=0, lr=3e-4, betas=(0.9, 0.999), eps=1e-5),
                hyper=hyper, obs_dim=14, act_dim=2, net_arch=((64, 64), (64, 64)),
                extra_policy_kwargs={"activation_fn": "relu", "log_std_init": -0.5})
    raw = json.loads(zipfile.ZipFile(path).read("data"))["policy_kwargs"]
    assert raw[":type:"] == "<class 'dict'>" and "ReLU" in raw["activation_fn"]
    real = pickle.loads(base64.b64decode(raw[":serialized:"]))          # what real SB3 would rebuild
    assert real["activation_fn"] is torch.nn.ReLU and real["net_arch"] == {"pi": [64, 64], "vf": [64, 64]}
    back = ck.load_zip(path)["data"]["policy_kwargs"]
    assert back["activation_fn"] == "ReLU" and back["log_std_init"] == -0.5 and back["net_arch"] == {"pi": [64, 64], "vf": [64, 64]}


def test_event_file_writer_round_trip(tmp_path):
    """tensorboard_log without the tensorboard package (mobrob_amd/tb_events.py): CRC-32C known answer, SB3's run-directory
    numbering, and a write / parse round trip with every record CRC checked."""
    from mobrob_amd import tb_events as tb
    assert tb.crc32c(b"123456789") == 0xE3069283          # the check value of CRC-32C (Castagnoli)
    assert tb.crc32c(b"") == 0
    root = str(tmp_path / "tensorboard")
    first = tb.next_run_dir(root, "PPO")
    assert first.endswith("PPO_1")
    w = tb.EventFileWriter(first)
    w.add_scalars([("train/loss", 0.25), ("time/fps", 1234), ("train/explained_variance", float("nan"))], 4096)
    w.add_scalars([("train/loss", -1.5)], 8192)
    w.close()
    assert tb.next_run_dir(root, "PPO").endswith("PPO_2") and tb.next_run_dir(root, "PPO", continue_latest=True).endswith("PPO_1")
    assert tb.next_run_dir(root, "other").endswith("other_1")
    ev = tb.read_events(w.path)
    assert ev[0]["file_version"] == "brain.Event:2" and ev[0]["step"] == 0
    assert ev[1]["step"] == 4096 and ev[1]["scalars"] == {"train/loss": 0.25, "time/fps": 1234.0}   # the NaN is not written
    assert ev[2]["step"] == 8192 and ev[2]["scalars"] == {"train/loss": -1.5}
    # the framing is TFRecord's: a flipped payload byte is caught by the payload CRC
    raw = bytearray(open(w.path, "rb").read())
    raw[30] ^= 1
    bad = tmp_path / "bad"
    bad.write_bytes(bytes(raw))
    with pytest.raises(ValueError):
        tb.read_events(str(bad))
    # the exact bytes of one event, written out from the protobuf wire format by hand: Event{wall_time=1.5, step=3,
    # summary{value{tag="a", simple_value=2.0}}}
    want = bytes([0x09]) + __import__("struct").pack("<d", 1.5) + bytes([0x10, 0x03, 0x2A, 0x0A, 0x0A, 0x08, 0x0A, 0x01, 0x61, 0x15]) \
        + __import__("struct").pack("<f", 2.0)
    assert tb.encode_event(1.5, 3, [("a", 2.0)]) == want


@pytest.mark.parametrize("pi,vf", [((64,), (64,)), ((64, 48, 32), (64, 48, 32)), ((40,), (64, 32, 16))])
def test_zip_of_other_depths_round_trips(pi, vf, tmp_path):
    """net_arch with one or three hidden layers per network: SB3's registration order (nn.Sequential indices 0, 2, 4) in the
    state dict and in the optimizer's parameter numbering, torch-loadable, round trip exact (torch-CPU builds the same modules
    and must agree on every key and shape)."""
    import torch
    from oracle import ppo_oracle as O
    D, A = 14, 2
    params = O.init_params(D, A, pi, vf, seed=1)
    assert list(params.keys()) == ck.policy_keys(len(pi), len(vf)) == O.param_keys(len(pi), len(vf))
    assert ck.policy_keys(2, 2) == ck.POLICY_KEYS and ck.depth_of_keys(list(params.keys())) == (len(pi), len(vf))
    rng = np.random.default_rng(0)
    m = OrderedDict((k, rng.standard_normal(v.shape).astype(np.float32)) for k, v in params.items())
    v_ = OrderedDict((k, rng.random(v.shape).astype(np.float32)) for k, v in params.items())
    hyper = dict(n_steps=32, batch_size=64, n_epochs=2, gamma=0.99, gae_lambda=0.95, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5,
                 learning_rate=3e-4, clip_range=0.2, n_envs=4)
    path = ck.save_zip(str(tmp_path / "deep"), params=params, optimizer=dict(exp_avg=m, exp_avg_sq=v_, step=12), hyper=hyper,
                       obs_dim=D, act_dim=A, net_arch=(pi, vf))
    c = ck.load_zip(path)
    assert list(c["params"].keys()) == list(params.keys())
    assert all(np.array_equal(c["params"][k], params[k]) for k in params)
    assert all(np.array_equal(c["optimizer"]["exp_avg"][k], m[k]) for k in params) and c["optimizer"]["step"] == 12
    assert c["data"]["policy_kwargs"] == {"net_arch": {"pi": list(pi), "vf": list(vf)}}
    # the torch modules SB3's MlpExtractor builds for this net_arch accept the state dict key for key
    def seq(widths):
        mods, prev = [], D
        for w in widths:
            mods += [torch.nn.Linear(prev, w), torch.nn.Tanh()]
            prev = w
        return torch.nn.Sequential(*mods)
    pol, val = seq(pi), seq(vf)
    with zipfile.ZipFile(path) as z:
        sd = torch.load(io.BytesIO(z.read("policy.pth")), map_location="cpu", weights_only=True)
    pol.load_state_dict({k.split("policy_net.")[1]: t for k, t in sd.items() if "policy_net" in k})
    val.load_state_dict({k.split("value_net.")[1]: t for k, t in sd.items() if ".value_net." in k})


def test_config_accepts_one_to_eight_hidden_layers_and_nothing_else():
    """Host-only sizing pass (mobrob_ppo_device_bytes runs check_cfg without touching a device)."""
    from mobrob_amd.engine import PPOEngine
    base = dict(obs_dim=14, act_dim=2, n_envs=8, n_steps=16, batch_size=32)
    sizes = {arch: PPOEngine.device_bytes(pi=arch, vf=arch, **base) for arch in [(64,), (64, 64), (64, 64, 64), (64,) * 5, (64,) * 8]}
    assert all(v > 0 for v in sizes.values()) and sizes[(64,)] < sizes[(64, 64, 64)] < sizes[(64,) * 5] < sizes[(64,) * 8]   # (generic chain; two layers: fused kernels)
    assert PPOEngine.device_bytes(pi=(40,), vf=(64, 32, 16, 8, 24), **base) > 0
    for bad in [(), (64,) * 9]:
        with pytest.raises(ValueError):
            PPOEngine.device_bytes(pi=bad, vf=(64, 64), **base)
    with pytest.raises(Exception, match="multiples of 8"):
        PPOEngine.device_bytes(pi=(64, 30), vf=(64, 64), **base)
    cfg = PPOEngine.make_config(pi=(64, 64), vf=(64, 64), **base)
    cfg.pi_hidden[1], cfg.pi_hidden3 = 0, 64                      # a hole in the list
    from mobrob_amd import _lib
    import ctypes as C
    n = C.c_size_t(0)
    assert _lib.load().mobrob_ppo_device_bytes(C.byref(cfg), C.byref(n)) != 0
    cfg = PPOEngine.make_config(pi=(64, 64, 64), vf=(64, 64), **base)
    cfg.pi_hidden_ext[1] = 64                                      # a hole behind the third layer
    assert _lib.load().mobrob_ppo_device_bytes(C.byref(cfg), C.byref(n)) != 0
    for act in ("tanh", "ReLU", "elu", "leaky_relu", "Sigmoid", "softplus", "softsign", "hardtanh", "relu6"):
        assert PPOEngine.device_bytes(pi=(64, 64), vf=(64, 64), activation=act, **base) > 0
    for act in ("SiLU", "gelu", "Mish"):
        assert PPOEngine.device_bytes(pi=(64, 64), vf=(64, 64), activation=act, **base) > 0
    with pytest.raises(NotImplementedError, match="PReLU"):
        PPOEngine.make_config(pi=(64, 64), vf=(64, 64), activation="PReLU", **base)   # (has a parameter of its own)
