"""Per-phase cycle shares of k_pair64_train (build with -DMOBROB_PAIR_STAMPS): python scratch/pair_stamps.py [D A]"""
import ctypes as C, sys, os, subprocess, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib_path = os.path.join(ROOT, "gpurun_out", "libmobrob_pair_stamps.so")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-pass-failed", "-mllvm",
                "-amdgpu-mfma-vgpr-form", "-DMOBROB_PAIR_STAMPS", "-o", lib_path, os.path.join(ROOT, "mobrob_amd/csrc/engine.hip")], check=True)
from mobrob_amd import _lib
_lib.LIB_PATH = lib_path
from mobrob_amd.engine import PPOEngine
from mobrob_amd.rl_control.init import orthogonal_policy_init
names = ["tile start: X -> LDS, index loads, f1 issue, barrier", "layer 1 + loss-operand issue", "barrier", "layer 2 + barrier", "head", "barrier",
         "loss (role 0) + backward fragment / next-row issue", "barrier", "dW3 + dh2 + dz2 + bias", "barrier", "dW2 + dh1 + dz1 + bias + dW1", "end barrier", "layer 1 alone (f1 wait + GEMM + tanh)", "loss: log-prob", "loss: ratio, clip, g_logp", "loss: gradient loop + row sums"]
for (D, A) in ([(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(14, 2), (58, 12)]):
    N, T, B = 1024, 128, 65536
    e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=2, pi=(64, 64), vf=(64, 64), ent_coef=0.01)
    e.set_params(orthogonal_policy_init(D, A, (64, 64), (64, 64), 0))
    e.collect_synthetic()
    e.train(None)
    out = (C.c_ulonglong * 32)()
    e.lib.mobrob_dbg_read_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
    e.lib.mobrob_dbg_read_stamps(e._h, out, 1)
    e.profile(True)
    e.train(None)
    pr = e.profile_read()
    us = 1e3 * pr["train_grad"][0] / pr["train_grad"][1]
    e.lib.mobrob_dbg_read_stamps(e._h, out, 1)
    v = np.array(list(out), dtype=np.float64)
    launches = 2 * e.n_minibatches
    tiles = 2 * 2048 * launches            # tile-nets per role
    print(f"--- {D}/{A}: shader-clock cycles per tile and wave")
    print(f" launch {us:.1f} us (HIP events); cycles per launch and pair (4 tiles) {v[:16].sum() / tiles * 4:.0f} -> implied clock {v[:16].sum() / tiles * 4 / us / 1e3:.2f} GHz if the loop were the whole launch")
    print(f" tile loop: {v[16 + 13] / v[16 + 14] * 100:.0f} MHz shader clock (s_memtime / s_memrealtime), {v[16 + 14] / (tiles / 4) / 100:.1f} us per wave")
    v[16 + 13] = v[16 + 14] = 0
    for role in (0, 1):
        tot = v[16 * role:16 * role + 16].sum()
        print(f" role {role}: total {tot / tiles:8.1f} ticks per tile")
        for k in range(16):
            print(f"   {k:2d} {names[k]:58s} {v[16 * role + k] / tiles:8.2f}  {100 * v[16 * role + k] / tot:5.1f}%")
    e.close()
