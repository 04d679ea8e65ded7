"""CPU: SB3-zip checkpoint reader/writer (mobrob_amd/checkpoint.py)."""
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


def test_space_blobs_unpickle_like_the_reference_ones():
    """The handcrafted Box pickles rebuild the same state dict gymnasium 0.28.1 wrote into the reference zips."""
    mods = {n: types.ModuleType(n) for n in ("gymnasium", "gymnasium.spaces", "gymnasium.spaces.box")}

    class Box:
        pass

    Box.__module__ = "gymnasium.spaces.box"
    mods["gymnasium.spaces.box"].Box = Box
    saved = {n: sys.modules.get(n) for n in mods}
    sys.modules.update(mods)
    try:
        b = pickle.loads(ck.pickle_box(np.full(12, -1.0), np.full(12, 1.0)))
        assert sorted(b.__dict__) == sorted(["dtype", "bounded_below", "bounded_above", "_shape", "low", "high",
                                             "low_repr", "high_repr", "_np_random"])
        assert b._shape == (12,) and b.low_repr == "-1.0" and b.high_repr == "1.0" and b.dtype == np.float32
        assert b.bounded_below.all() and b.low.dtype == np.float32 and b.low.flags.writeable
        o = pickle.loads(ck.pickle_box(np.full(58, -np.inf), np.full(58, np.inf)))
        assert o.low_repr == "-inf" and o.high_repr == "inf" and not o.bounded_above.any()
    finally:
        for n, m in saved.items():
            if m is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = m
    assert pickle.loads(ck.pickle_ndarray(np.array([[1.5, 2.5]], np.float32))).tolist() == [[1.5, 2.5]]
