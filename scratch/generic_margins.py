"""Margins of the generic-chain parity tests whose sums go through float atomics (run-to-run order differs): max scaled error per
run of tests/test_arch_gpu.py::test_full_size_minibatch_on_the_generic_chain's comparison, several runs.  gpurun -- python scratch/generic_margins.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ppo_oracle as O
from tests.util import scaled_err
from tests.test_arch_gpu import _rollout_for, _engine

for act, sde, pi, vf in [("tanh", False, (128, 128), (128, 96)), ("silu", False, (128, 128), (128, 96)), ("tanh", True, (128, 128), (128, 96)),
                         ("elu", False, (128, 128, 64), (96,))]:
    D, A, T, N = 26, 3, 64, 1024
    B = T * N
    rng = np.random.default_rng(3)
    p = O.init_params(D, A, pi, vf, seed=7)
    p["log_std"] = (rng.normal(-1.5, 0.3, (pi[-1], A)) if sde else rng.normal(-0.3, 0.2, A)).astype(np.float32)
    p["action_net.weight"] *= 30
    buf, lv, dones = _rollout_for(p, act, D, A, T, N, rng, sde)
    h = O.Hyper(ent_coef=0.01, n_epochs=1, batch_size=B, activation=act, use_sde=sde)
    buf["advantages"], buf["returns"] = O.gae(buf["rewards"], buf["values"], buf["episode_starts"], lv, dones, h.gamma, h.gae_lambda)
    perm = rng.permutation(B)
    _, og, _ = O.loss_and_grads(p, *O.gather_minibatch(buf, perm), h, acc=np.float64)
    worst = []
    for rep in range(4):
        e = _engine(D, A, N, T, pi, vf, batch_size=B, n_epochs=1, ent_coef=h.ent_coef, activation=act, use_sde=sde)
        e.set_params(p)
        e.load_rollout(buf, lv, dones)
        e.epoch_begin(perm)
        e.minibatch_grad(0)
        got = e.unflatten(e.read("grads"))
        errs = {k: scaled_err(got[k], og[k]) for k in og}
        k = max(errs, key=errs.get)
        worst.append((errs[k], k))
        e.close()
    print(act, "sde" if sde else "", pi, vf, " worst scaled error per run:", ", ".join(f"{w:.2e} ({k.split('.')[-2]}.{k.split('.')[-1]})" if '.' in k else f"{w:.2e} ({k})" for w, k in worst), flush=True)
