"""Per-launch times of the per-step update of 64-wide nets over minibatch sizes: default kernel choice (split tile <= 64 tiles,
two-wave pairs above), without the pair kernel (MOBROB_PAIR64_MIN_TILES=0), block kernel only (+ MOBROB_SPLIT64_MAX_TILES=0).  Usage: python scratch/time64_split.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
which = sys.argv[1] if len(sys.argv) > 1 else None
if which is None:
    for k in ("default", "no-pair", "block-only"):
        env = dict(os.environ)
        if k == "no-pair":
            env["MOBROB_PAIR64_MIN_TILES"] = "0"
        if k == "block-only":
            env["MOBROB_PAIR64_MIN_TILES"] = "0"
            env["MOBROB_SPLIT64_MAX_TILES"] = "0"
        subprocess.run([sys.executable, __file__, k], check=True, env=env)
    sys.exit(0)
import time
from mobrob_amd.engine import PPOEngine
from mobrob_amd.rl_control.init import orthogonal_policy_init
for (D, A, N, T, B) in [(58, 12, 16, 1000, 64), (58, 12, 16, 1000, 100), (14, 2, 2, 4000, 100), (58, 12, 64, 256, 512), (58, 12, 64, 256, 2048), (58, 12, 64, 256, 2080),
                        (58, 12, 128, 256, 4096), (58, 12, 256, 256, 8192), (58, 12, 256, 256, 16384), (14, 2, 256, 256, 2080), (14, 2, 256, 256, 8192)]:
    e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=2, pi=(64, 64), vf=(64, 64), ent_coef=0.01)
    e.set_params(orthogonal_policy_init(D, A, (64, 64), (64, 64), 0))
    e.collect_synthetic()
    e.train(None)
    e.synchronize() if hasattr(e, "synchronize") else None
    t0 = time.perf_counter()
    e.train(None)
    e.read("params")
    wall = time.perf_counter() - t0
    steps = 2 * e.n_minibatches
    e.profile(True)
    e.train(None)
    pr = e.profile_read()
    print(which, D, A, "B", B, "train us/launch %.1f" % (1e3 * pr["train_grad"][0] / pr["train_grad"][1]),
          "reduce %.1f" % (1e3 * pr["grad_reduce"][0] / pr["grad_reduce"][1]), "apply %.1f" % (1e3 * pr["apply"][0] / pr["apply"][1]),
          "| un-instrumented: %.1f us per optimizer step" % (1e6 * wall / steps), flush=True)
    e.close()
