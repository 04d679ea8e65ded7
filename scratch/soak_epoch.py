"""Soak of k_epoch64's in-launch hand-offs (csrc/kernels_epoch64.h): PPO iterations (device rollout -> update) at the reference YAML shape (16 envs x 1000 steps, batch
100, 2x64, 5 epochs: 800 optimizer steps = 2 400 grid barriers per iteration) for ITERS iterations with the co-operative epoch kernel and
with three launches per optimizer step, same seeds.  A stale read at any hand-off -- a weight pack, a slab, a norm record, a moment -- would
change a bit somewhere downstream, so the check is the strongest one there is: parameters, both Adam moments and the rollout buffers of the
LAST iteration must be BIT-equal between the two runs.  With LOAD=1 a second engine (doggo 2x256, 4096 envs) keeps rolling out on another
stream meanwhile: uneven load on the CUs, the L2s and the fabric, workgroups of the epoch kernel dispatched late (MI355X_MICROARCH.md: "test
every hand-off under UNEVEN load, consumer L1-warm, checking every word").
    python scratch/soak_epoch.py [ITERS=150] [LOAD=1]"""
import hashlib
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobrob_amd.engine import PPOEngine  # noqa: E402
from mobrob_amd.rl_control.init import orthogonal_policy_init  # noqa: E402

ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 150
LOAD = int(sys.argv[2]) if len(sys.argv) > 2 else 1
D, A, H, N, T, B, E = 58, 12, 64, 16, 1000, 100, 5


def run(epoch_kernel):
    e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=E, pi=(H, H), vf=(H, H), ent_coef=0.01, seed=7)
    e.set_hyper(epoch_kernel=int(epoch_kernel))
    e.set_params(orthogonal_policy_init(D, A, (H, H), (H, H), 0))
    stop = threading.Event()
    bg = None
    if LOAD:
        big = PPOEngine(obs_dim=58, act_dim=12, n_envs=4096, n_steps=200, batch_size=65536, n_epochs=1, pi=(256, 256), vf=(256, 256), seed=1)
        big.set_params(orthogonal_policy_init(58, 12, (256, 256), (256, 256), 1))

        def load():
            while not stop.is_set():
                big.collect_synthetic()
                big.synchronize()
        bg = threading.Thread(target=load, daemon=True)
        bg.start()
    t0 = time.time()
    try:
        for _ in range(ITERS):
            e.collect_synthetic()          # device-resident env source: the rollout depends on the policy, the update on the rollout
            e.train(None)
    except BaseException:
        stop.set()
        raise
    dt = time.time() - t0
    assert e.update_mode() == (1 if epoch_kernel else 0)
    m, v, step = e.get_optimizer_state()
    h = hashlib.sha256()
    for arr in [e.get_flat_params()] + [m[k] for k in m] + [v[k] for k in v] + [e.read(k) for k in ("obs", "actions", "rewards", "values", "log_probs", "advantages")]:
        h.update(np.ascontiguousarray(arr).tobytes())
    stop.set()
    if bg is not None:
        bg.join()
        big.close()
    fin = bool(np.isfinite(e.get_flat_params()).all())
    e.close()
    return h.hexdigest(), step, dt, fin


a = run(True)
b = run(False)
print(f"{ITERS} iterations x 800 optimizer steps at the reference YAML shape, background rollouts of a 2x256 engine: {'on' if LOAD else 'off'}")
print(f"  k_epoch64           : {a[2]:6.1f} s, Adam step {a[1]}, finite {a[3]}, sha256 {a[0][:16]}")
print(f"  three launches/step : {b[2]:6.1f} s, Adam step {b[1]}, finite {b[3]}, sha256 {b[0][:16]}")
print("  BIT-EQUAL" if a[0] == b[0] and a[1] == b[1] else "  DIFFERENT")
sys.exit(0 if a[0] == b[0] else 1)
