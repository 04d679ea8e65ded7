"""Native host env (csrc/host_env.c) step cost on this box: thread-count sweep for the full batch and for half ranges
(what the pipelined collector steps)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobrob_amd.envs.native_env import NativeGoalVecEnv

N = 4096
for robot in ("doggo", "point"):
    env = NativeGoalVecEnv.for_robot(robot, N, time_limit=1000, seed=0)
    env.reset()
    a = np.random.default_rng(0).standard_normal((N, env.act_dim)).astype(np.float32)
    for thr in (1, 4, 8, 16, 32, 64):
        env.set_threads(thr)
        out = []
        for i0, i1 in ((0, N), (0, N // 2)):
            for _ in range(50):
                env.step_range(i0, i1, a)
            t0 = time.perf_counter()
            for _ in range(500):
                env.step_range(i0, i1, a)
            out.append((time.perf_counter() - t0) / 500 * 1e6)
        print(f"{robot:6s} threads {thr:3d}: full {out[0]:7.1f} us   half {out[1]:7.1f} us")
    env.close()
