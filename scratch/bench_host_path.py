"""Host-env path (act/store through PCIe): engine cost per step and the whole loop with (a) the NumPy synthetic source,
(b) the native multi-threaded C environment writing into pinned staging."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from mobrob_amd.engine import PPOEngine
from mobrob_amd.envs.native_env import NativeGoalVecEnv
from mobrob_amd.envs.vec_env import SyntheticVecEnv
from mobrob_amd.rl_control.init import orthogonal_policy_init
D, A, N, T, H = 58, 12, 4096, 256, 256
e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=65536, n_epochs=5, pi=(H, H), vf=(H, H))
e.set_params(orthogonal_policy_init(D, A, (H, H), (H, H), 0))

env = SyntheticVecEnv(N, D, A, seed=0)
obs_buf, clip_buf = e.pinned((N, D)), e.pinned((N, A))
obs_buf[:] = env.reset()
import os
for rep in range(0 if os.environ.get('HOSTENV_ONLY') else 2):
    e.rollout_begin()
    t_eng = t_env = 0.0
    t0 = time.perf_counter()
    for t in range(T):
        a = time.perf_counter()
        e.act(obs_buf, out_clipped=clip_buf, want_all=False)
        b = time.perf_counter()
        o, r, d, infos = env.step(clip_buf)
        obs_buf[:] = o
        c = time.perf_counter()
        e.store(r, d)
        t_eng += (b - a) + (time.perf_counter() - c); t_env += c - b
    e.finish_rollout(obs_buf, d)
    dt = time.perf_counter() - t0
if not os.environ.get("HOSTENV_ONLY"): print(f"numpy synthetic source: {N*T/dt/1e6:.2f} M env-steps/s rollout; engine {1e6*t_eng/T:.0f} us/step, host env {1e6*t_env/T:.0f} us/step")

env = NativeGoalVecEnv.for_robot("doggo", N, time_limit=1000)
b = dict(obs=obs_buf, rew=e.pinned((N,)), done=e.pinned((N,), np.uint8), trunc=e.pinned((N,), np.uint8), term=e.pinned((N, D)))
env.use_buffers(obs=b["obs"], rewards=b["rew"], dones=b["done"], truncated=b["trunc"], terminal_obs=b["term"])
env.reset()
import os
for threads in [int(x) for x in os.environ.get('HOSTENV_THREADS', '16,32').split(',')]:
    env.set_threads(threads)
    for rep in range(2):
        e.rollout_begin()
        t_eng = t_env = 0.0
        t0 = time.perf_counter()
        for t in range(T):
            a = time.perf_counter()
            e.act(b["obs"], out_clipped=clip_buf, want_all=False)
            bb = time.perf_counter()
            nt = env.step_arrays(clip_buf)[5]
            c = time.perf_counter()
            e.store(b["rew"], b["done"], b["trunc"] if nt else None, b["term"] if nt else None)
            t_eng += (bb - a) + (time.perf_counter() - c); t_env += c - bb
        e.finish_rollout(b["obs"], b["done"])
        e.synchronize()
        dt = time.perf_counter() - t0
    t1 = time.perf_counter(); e.train(None); tt = time.perf_counter() - t1
    print(f"native C env, {threads:3d} threads: rollout {N*T/dt/1e6:.2f} M env-steps/s (engine {1e6*t_eng/T:.0f} us/step, env {1e6*t_env/T:.0f} us/step); "
          f"with the update ({tt*1e3:.0f} ms for {T} steps x {N} envs x 5 epochs): {N*T/(dt+tt)/1e6:.2f} M env-steps/s end to end")
