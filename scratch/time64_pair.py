"""k_pair64_train experiments at BASELINE config 2's minibatch (65 536 rows, 14/2 and 58/12, 2x64): against the block kernel.  Usage: python scratch/time64_pair.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
variants = {"block": {"MOBROB_PAIR64_MIN_TILES": "0"}, "pair": {}}
which = sys.argv[1] if len(sys.argv) > 1 else None
if which is None:
    for k, v in variants.items():
        subprocess.run([sys.executable, __file__, k], check=True, env=dict(os.environ, **v))
    sys.exit(0)
from mobrob_amd.engine import PPOEngine
from mobrob_amd.rl_control.init import orthogonal_policy_init
for (D, A, N, T, B) in [(14, 2, 1024, 256, 65536), (26, 2, 1024, 256, 65536), (43, 2, 1024, 128, 65536), (58, 12, 1024, 128, 65536), (58, 12, 1024, 32, 16384)]:
    e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=4, pi=(64, 64), vf=(64, 64), ent_coef=0.01)
    e.set_params(orthogonal_policy_init(D, A, (64, 64), (64, 64), 0))
    e.collect_synthetic()
    e.train(None)
    e.profile(True)
    e.train(None)
    pr = e.profile_read()
    print("%-11s %2d/%-2d B %d: train %.1f us/launch, reduce %.1f, apply %.1f" % (which, D, A, B, 1e3 * pr["train_grad"][0] / pr["train_grad"][1],
          1e3 * pr["grad_reduce"][0] / pr["grad_reduce"][1], 1e3 * pr["apply"][0] / pr["apply"][1]), flush=True)
    e.close()
