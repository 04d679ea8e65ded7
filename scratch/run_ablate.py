"""Timing-only ablation of k_fused_train phases (each variant in its own process)."""
import subprocess, sys, os, json
if len(sys.argv) == 2:
    m = sys.argv[1]
    sys.path.insert(0, '.')
    from mobrob_amd import _lib
    _lib.LIB_PATH = os.path.abspath(f"scratch/lib_skip_{m}.so")
    from mobrob_amd.engine import PPOEngine
    from mobrob_amd.rl_control.init import orthogonal_policy_init
    D, A, H, N, T, B = 58, 12, 256, 4096, 128, 65536
    e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=2, pi=(H, H), vf=(H, H), ent_coef=0.01)
    e.set_params(orthogonal_policy_init(D, A, (H, H), (H, H), 0))
    e.collect_synthetic()
    e.train(None)
    e.profile(True)
    e.train(None)
    ms, calls = e.profile_read()["train_grad"]
    print(json.dumps({"mask": m, "ms_per_launch": ms / calls}))
else:
    base = None
    names = {0: "full", 1: "gather loads", 2: "L1 gemm", 4: "L2 gemm", 8: "head gemm+reduce", 16: "loss stage", 32: "dW3+RMW",
             64: "dh2 gemm", 128: "dW2", 256: "dh1 gemm", 512: "dW1+RMW", 1024: "all epilogues", 2048: "column sums", 4096: "dW1 slab loads", 8192: "dW1 slab stores", 12288: "dW1 slab ld+st"}
    ms = [0, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048] if len(sys.argv) < 2 else sys.argv[2:]
    for m in ms:
        out = subprocess.run([sys.executable, __file__, str(m)], capture_output=True, text=True).stdout.strip().splitlines()[-1]
        v = json.loads(out)["ms_per_launch"]
        if base is None:
            base = v
        print(f"{names.get(m, str(m)):20s} {v*1e3:8.1f} us   delta {1e3*(base - v):7.1f} us  ({100*(base-v)/base:5.1f}%)")
