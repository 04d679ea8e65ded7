"""Experiment: forward pass of the value network with 8 waves per workgroup (two per SIMD; kernels_fused8.h, build
with -DMOBROB_VALUE8 as scratch/lib_value8.so) against the product's 4-wave k_value_batch: values must be identical,
the 'act' phase of a persistent rollout is the final value pass (51 steps x 4096 rows on the whole device)."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) == 2:
    import hashlib, numpy as np
    from mobrob_amd import _lib
    if sys.argv[1] != "product":
        _lib.LIB_PATH = sys.argv[1]
    from mobrob_amd.engine import PPOEngine
    from mobrob_amd.rl_control.init import orthogonal_policy_init
    D, A, H, N, T = 58, 12, 256, 4096, 1000
    e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=65536, n_epochs=1, pi=(H, H), vf=(H, H), seed=0)
    e.set_params(orthogonal_policy_init(D, A, (H, H), (H, H), 0))
    e.collect_synthetic(); e.synchronize()
    h = hashlib.sha256(e.read("values").tobytes()).hexdigest()[:16]
    e.profile(True, only=["act", "env"])
    for _ in range(5): e.collect_synthetic()
    e.synchronize()
    pr = e.profile_read()
    print(json.dumps({"lib": sys.argv[1], "values": h, "final_value_pass_us": 1e3 * pr["act"][0] / pr["act"][1],
                      "rollout_ms": pr["env"][0] / pr["env"][1]}))
else:
    for lib in ("product", f"{ROOT}/scratch/lib_value8.so", "product", f"{ROOT}/scratch/lib_value8.so"):
        print(subprocess.run([sys.executable, __file__, lib], capture_output=True, text=True).stdout.strip().splitlines()[-1])
