"""Host-env path (act/store through PCIe): engine cost per step vs the host env source."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from mobrob_amd.engine import PPOEngine
from mobrob_amd.envs.vec_env import SyntheticVecEnv
from mobrob_amd.rl_control.init import orthogonal_policy_init
D, A, N, T, H = 58, 12, 4096, 64, 256
e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=65536, n_epochs=1, pi=(H, H), vf=(H, H))
e.set_params(orthogonal_policy_init(D, A, (H, H), (H, H), 0))
env = SyntheticVecEnv(N, D, A, seed=0)
for pinned in (False, True):
    obs_buf = e.pinned((N, D)) if pinned else np.empty((N, D), np.float32)
    clip_buf = e.pinned((N, A)) if pinned else np.empty((N, A), np.float32)
    obs_buf[:] = env.reset()
    rew = np.zeros(N, np.float32); dones = np.zeros(N, np.uint8)
    for rep in range(2):
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
            dd = time.perf_counter()
            t_eng += (b - a) + (dd - c); t_env += c - b
        e.finish_rollout(obs_buf, d)
        dt = time.perf_counter() - t0
    print(f"pinned={pinned}: {N*T/dt/1e6:.2f} M env-steps/s total; engine {1e6*t_eng/T:.0f} us/step "
          f"({N*T/t_eng/1e6:.1f} M env-steps/s engine-only, PCIe inclusive); host env {1e6*t_env/T:.0f} us/step")
