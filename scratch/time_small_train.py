"""Phase times of the persistent small-batch kernel (build with -DMOBROB_SMALL_STAMPS): cycles per phase of thread 0 of
each workgroup: 0 setup, 1 tile, 2 wave reduction, 3 gradient scatter, 4 norms, 5 hand-off, 6 Adam + re-pack."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
lib = os.path.join(ROOT, "gpurun_out", "libstamp_small.so")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-pass-failed", "-mllvm",
                "-amdgpu-mfma-vgpr-form", "-DMOBROB_SMALL_STAMPS", "-o", lib, os.path.join(ROOT, "mobrob_amd/csrc/engine.hip")], check=True)
import mobrob_amd._lib as L
L.LIB_PATH = lib
from mobrob_amd.engine import PPOEngine
from mobrob_amd.rl_control.init import orthogonal_policy_init
D, A, H, N, T, B, E = 58, 12, 64, 16, 1000, 100, 5
e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=E, pi=(H, H), vf=(H, H), ent_coef=0.01, persistent_train=True)
e.set_params(orthogonal_policy_init(D, A, (H, H), (H, H), 0))
e.collect_synthetic()
e.train(None)
t0 = time.perf_counter(); e.train(None); dt = time.perf_counter() - t0
nmb = e.n_minibatches
print("train() %.2f ms, %.1f us per optimizer step" % (1e3 * dt, 1e6 * dt / (nmb * E)))
e.train_enqueue()
rows = e.fetch_step_stats(nmb)
names = ["setup", "tile", "wave reduction", "gradient scatter", "norms", "hand-off", "Adam + re-pack"]
for net in (0, 1):
    cyc = rows[net].astype(np.float64) / (nmb)          # cycles per optimizer step (last epoch)
    print("net", net, {n: "%.2f us" % (c / 100.0) for n, c in zip(names, cyc)}, "sum %.1f us" % (cyc.sum() / 100.0), "(100 MHz counter)")
