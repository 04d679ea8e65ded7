"""k_fused64_train at the reference YAML shape (100-row minibatches): launch time of the normal build and of the
`-DMOBROB64_EMPTY` build (tile loop skipped: launch + weight mirror + wave reduction + slab store)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mobrob_amd._lib as L
libs = {"full": L.LIB_PATH}
empty = os.path.join(ROOT, "gpurun_out", "lib64_empty.so")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-pass-failed", "-mllvm",
                "-amdgpu-mfma-vgpr-form", "-DMOBROB64_EMPTY", "-o", empty, os.path.join(ROOT, "mobrob_amd/csrc/engine.hip")], check=True)
libs["empty tile loop"] = empty
which = sys.argv[1] if len(sys.argv) > 1 else None
if which is None:
    for k in libs:
        subprocess.run([sys.executable, __file__, k], check=True)
    sys.exit(0)
L.LIB_PATH = libs[which]
from mobrob_amd.engine import PPOEngine
from mobrob_amd.rl_control.init import orthogonal_policy_init
for (D, A, N, T, B) in [(58, 12, 16, 1000, 100), (14, 2, 2, 4000, 100), (58, 12, 1024, 64, 65536)]:
    e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=2, pi=(64, 64), vf=(64, 64), ent_coef=0.01)
    e.set_params(orthogonal_policy_init(D, A, (64, 64), (64, 64), 0))
    e.collect_synthetic()
    e.train(None)
    e.profile(True)
    e.train(None)
    pr = e.profile_read()
    print(which, D, A, "B", B, "train us/launch %.1f" % (1e3 * pr["train_grad"][0] / pr["train_grad"][1]),
          "reduce %.1f" % (1e3 * pr["grad_reduce"][0] / pr["grad_reduce"][1]), "apply %.1f" % (1e3 * pr["apply"][0] / pr["apply"][1]))
    e.close()
