"""Host-only timing of ShmVecEnv: vector step and empty command round trip.  usage: python scratch/time_shm_env.py n_envs n_workers"""
import functools, time, sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobrob_amd.envs.shm_vec_env import ShmVecEnv
from mobrob_amd.envs.vec_env import HostVecEnv, make_vec_env
from mobrob_amd.envs.wrapper import get_env
if __name__ == "__main__":
    n = int(sys.argv[1]); W = int(sys.argv[2])
    kw = dict(env_name="doggo", enable_gui=False, terminate_on_goal=True, time_limit=1000)
    env = make_vec_env(get_env, n, kw, functools.partial(ShmVecEnv, n_workers=W), seed=0)
    env.reset()
    a = np.zeros((n, 12), np.float32)
    for _ in range(3): env.step_arrays(a)
    t0 = time.time()
    for _ in range(20): env.step_arrays(a)
    print("n", n, "W", W, "step ms", 1e3 * (time.time() - t0) / 20, flush=True)
    t0 = time.time()
    for _ in range(200): env._command(1, 0, 0)
    print("empty command ms", 1e3 * (time.time() - t0) / 200)
    env.close()
