"""per-step time of k_rollout_persistent with phases ablated (ROLL_SKIP builds)"""
import sys, os, subprocess, json
if len(sys.argv) == 2:
    sys.path.insert(0, '.')
    from mobrob_amd import _lib
    _lib.LIB_PATH = os.path.abspath(f"scratch/lib_roll_{sys.argv[1]}.so")
    from mobrob_amd.engine import PPOEngine
    from mobrob_amd.rl_control.init import orthogonal_policy_init
    D, A, H, N, T = 58, 12, 256, 4096, 1000
    e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=65536, n_epochs=1, pi=(H, H), vf=(H, H))
    e.set_params(orthogonal_policy_init(D, A, (H, H), (H, H), 0))
    e.collect_synthetic(); e.synchronize()
    e.profile(True)
    for _ in range(3): e.collect_synthetic()
    e.synchronize()
    ms, calls = e.profile_read()["env"]  # the persistent rollout chunks (the "act" phase is the final value pass)
    print(json.dumps({"m": sys.argv[1], "ms": ms / calls}))
else:
    names = {"0": "full", "1": "normals draw", "2": "L1 gemm", "4": "L2 gemm", "8": "sample math", "16": "env phase", "31": "all of those", "128": "global stores"}
    base = None
    for m in (sys.argv[2:] or ["0", "1", "2", "4", "8", "16", "31", "0"]):
        out = subprocess.run([sys.executable, __file__, m], capture_output=True, text=True).stdout.strip().splitlines()[-1]
        v = json.loads(out)["ms"]
        base = base or v
        print(f"{names.get(m, m):16s} rollout {v:7.2f} ms  ({v:5.2f} us/step of 4096 envs)  delta {base - v:6.2f} us/step")
