"""k_pair64_train per-launch time for builds with phases ablated (scratch/build_variant.sh pair_<mask> -DPAIR_SKIP=<mask>):
python scratch/time_pair_ablate.py [D A] -- <lib> <lib> ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "--" in sys.argv:
    i = sys.argv.index("--")
    for lib in sys.argv[i + 1:]:
        subprocess.run([sys.executable, __file__] + sys.argv[1:i], env=dict(os.environ, MOBROB_PPO_LIB=os.path.abspath(lib) if lib != "default" else ""), check=True)
    sys.exit(0)
from mobrob_amd.engine import PPOEngine
from mobrob_amd.rl_control.init import orthogonal_policy_init
D, A = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (14, 2)
N, T, B = 1024, 256, int(os.environ.get('PAIR_B', 65536))
e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=4, pi=(64, 64), vf=(64, 64), ent_coef=0.01)
e.set_params(orthogonal_policy_init(D, A, (64, 64), (64, 64), 0))
e.collect_synthetic()
e.train(None)
e.profile(True)
for _ in range(3):
    e.train(None)
pr = e.profile_read()
print("B %6d " % B + "%-28s %2d/%-2d train %.2f us/launch, reduce %.2f, apply %.2f" % (os.path.basename(os.environ.get("MOBROB_PPO_LIB") or "default"), D, A,
      1e3 * pr["train_grad"][0] / pr["train_grad"][1], 1e3 * pr["grad_reduce"][0] / pr["grad_reduce"][1], 1e3 * pr["apply"][0] / pr["apply"][1]), flush=True)
e.close()
