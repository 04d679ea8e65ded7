"""Bit-level comparison of two builds of libmobrob_ppo.so on the same inputs (refactoring check): three PPO
iterations on three shapes, then parameters / Adam moments / step statistics must be identical.
    python scratch/compare_builds.py /path/to/old.so /path/to/new.so"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) == 3 and sys.argv[1] == "--run":
    import numpy as np, hashlib
    from mobrob_amd import _lib
    _lib.LIB_PATH = sys.argv[2]
    from mobrob_amd.engine import PPOEngine
    from mobrob_amd.rl_control.init import orthogonal_policy_init
    out = {}
    for (D, A, H, N, T, B) in [(58, 12, 256, 512, 64, 4096), (14, 2, 64, 256, 64, 2048), (26, 2, 48, 64, 32, 256)]:
        e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=3, pi=(H, H), vf=(H, H), ent_coef=0.01, seed=5)
        e.set_params(orthogonal_policy_init(D, A, (H, H), (H, H), 0))
        for _ in range(3):
            e.collect_synthetic(p_term=0.02, time_limit=40)
            st = e.train(None)
        m, v, step = e.get_optimizer_state()
        h = hashlib.sha256(e.get_flat_params().tobytes())
        for k in sorted(m):
            h.update(m[k].tobytes()); h.update(v[k].tobytes())
        out[f"{D}x{A}x{H}"] = [h.hexdigest()[:16], step, repr(st["grad_norm"]), repr(st["loss"])]
        e.close()
    print(json.dumps(out))
else:
    res = [json.loads(subprocess.run([sys.executable, __file__, "--run", p], capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])
           for p in sys.argv[1:3]]
    for k in res[0]:
        print(k, "IDENTICAL" if res[0][k] == res[1][k] else "DIFFERENT", res[0][k], res[1][k])
