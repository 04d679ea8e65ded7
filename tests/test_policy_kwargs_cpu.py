"""CPU: the keyword surface of PPO(...) as a reference YAML would carry it (examples/train.py loads `ppo_kwargs` with yaml.FullLoader and
splats it into PPO, /root/reference/src/mobrob/rl_control/ppo.py:58) -- parsed without a device (`_init_setup_model=False`)."""
import pytest
import yaml

from mobrob_amd.engine import ACTIVATIONS, MAX_HIDDEN, PPOEngine, activation_name, param_shapes
from mobrob_amd.rl_control.ppo import PPO

YAML = """
ppo_kwargs:
  policy: MlpPolicy
  n_steps: 1000
  batch_size: 100
  n_epochs: 5
  use_sde: true
  sde_sample_freq: 8
  policy_kwargs:
    net_arch: {pi: [64, 64, 32], vf: [64]}
    activation_fn: !!python/name:torch.nn.ELU
    log_std_init: -2.0
    full_std: false
    share_features_extractor: false
"""


def test_yaml_with_a_torch_activation_class_and_gsde_is_parsed():
    kw = yaml.load(YAML, Loader=yaml.FullLoader)["ppo_kwargs"]
    ppo = PPO(env=None, _dims=(16, 58, 12), _init_setup_model=False, **kw)
    assert ppo.activation == "elu" and ppo.net_arch == ((64, 64, 32), (64,))
    assert ppo.use_sde and ppo.sde_sample_freq == 8 and not ppo.sde_full_std and not ppo.sde_use_expln and ppo.log_std_init == -2.0
    cfg = PPOEngine.make_config(58, 12, 16, 1000, batch_size=100, n_epochs=5, pi=ppo.net_arch[0], vf=ppo.net_arch[1], activation=ppo.activation,
                                use_sde=True, sde_sample_freq=8, sde_full_std=False)
    assert (cfg.activation, cfg.use_sde, cfg.sde_sample_freq, cfg.sde_full_std, cfg.sde_use_expln) == (ACTIVATIONS["elu"][0], 1, 8, 0, 0)
    assert (cfg.pi_hidden[0], cfg.pi_hidden[1], cfg.pi_hidden3, cfg.vf_hidden[0], cfg.vf_hidden[1]) == (64, 64, 32, 64, 0)
    shapes = param_shapes(58, 12, *ppo.net_arch, use_sde=True, full_std=False)
    assert shapes["log_std"] == (32, 1) and shapes["action_net.weight"] == (12, 32) and shapes["value_net.weight"] == (1, 64)
    assert list(shapes)[:3] == ["log_std", "mlp_extractor.policy_net.0.weight", "mlp_extractor.policy_net.0.bias"]
    assert PPOEngine.device_bytes(obs_dim=58, act_dim=12, n_envs=16, n_steps=1000, batch_size=100, n_epochs=5, pi=(64, 64, 32), vf=(64,),
                                  activation="elu", use_sde=True, sde_full_std=False) > 0


def test_activation_names_resolve_from_classes_names_and_checkpoint_strings():
    import torch
    for key, (_, cls) in ACTIVATIONS.items():
        assert activation_name(getattr(torch.nn, cls)) == key == activation_name(cls) == activation_name(f"<class 'torch.nn.modules.activation.{cls}'>")
    assert activation_name(None) == "tanh" and activation_name("leaky_relu") == "leakyrelu"
    assert len(ACTIVATIONS) == 12 and [v[0] for v in ACTIVATIONS.values()] == list(range(12))


@pytest.mark.parametrize("kw,word", [(dict(policy_kwargs=dict(activation_fn="PReLU")), "PReLU"),
                                     (dict(policy_kwargs=dict(squash_output=True), use_sde=True), "squash_output"),
                                     (dict(policy_kwargs=dict(features_extractor_class="NatureCNN")), "features_extractor_class"),
                                     (dict(policy_kwargs=dict(optimizer_class="SGD")), "SGD"),
                                     (dict(policy_kwargs=dict(optimizer_kwargs=dict(weight_decay=0.1))), "weight"),
                                     (dict(policy="CnnPolicy"), "CnnPolicy")])
def test_what_is_not_served_is_refused_by_name(kw, word):
    with pytest.raises((NotImplementedError, ValueError), match=word):
        PPO(env=None, _dims=(4, 6, 2), _init_setup_model=False, **{"policy": "MlpPolicy", **kw})


def test_depth_limit_is_the_engine_s():
    with pytest.raises(ValueError, match=f"one to {MAX_HIDDEN}"):
        PPOEngine.make_config(6, 2, 4, 8, pi=(8,) * (MAX_HIDDEN + 1), vf=(8,))
    assert PPOEngine.device_bytes(obs_dim=6, act_dim=2, n_envs=4, n_steps=8, pi=(8,) * MAX_HIDDEN, vf=(16,)) > 0


def test_keys_other_sb3_writers_leave_in_policy_kwargs_are_tolerated():
    ppo = PPO(env=None, _dims=(4, 6, 2), _init_setup_model=False, use_sde=True, policy_kwargs=dict(use_sde=True, sde_net_arch=None, net_arch=[32]))
    assert ppo.use_sde and "use_sde" not in ppo.policy_kwargs and "sde_net_arch" not in ppo.policy_kwargs and ppo.net_arch == ((32,), (32,))
    with pytest.raises(ValueError, match="use_sde"):
        PPO(env=None, _dims=(4, 6, 2), _init_setup_model=False, use_sde=False, policy_kwargs=dict(use_sde=True))
