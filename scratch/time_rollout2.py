"""run rollouts with a ROLL_SKIP build (for rocprofv3 --kernel-trace --stats)"""
import sys, os
sys.path.insert(0, '.')
from mobrob_amd import _lib
_lib.LIB_PATH = os.path.abspath(f"scratch/lib_roll_{sys.argv[1]}.so")
from mobrob_amd.engine import PPOEngine
from mobrob_amd.rl_control.init import orthogonal_policy_init
D, A, H, N, T = 58, 12, 256, 4096, 1000
e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=65536, n_epochs=1, pi=(H, H), vf=(H, H))
e.set_params(orthogonal_policy_init(D, A, (H, H), (H, H), 0))
for _ in range(3): e.collect_synthetic()
e.synchronize()
