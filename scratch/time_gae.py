"""k_gae timing at the headline shape (T=1000, N=4096; 20 B/transition = 82 MB per launch) and a timing-only
ablation: each variant is the library rebuilt with -DGAE_SKIP=<mask> (1 serial scan, 2 advantage/return stores,
4 global loads after the first two tiles, 8 delta/coef staging).  Build the variants first (no GPU needed):
    python scratch/time_gae.py build
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MASKS = [0, 1, 2, 4, 6, 7, 15]
if sys.argv[1:] == ["build"]:
    for m in MASKS:
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-pass-failed",
                        "-mllvm", "-amdgpu-mfma-vgpr-form", f"-DGAE_SKIP={m}", "-o", f"{ROOT}/scratch/lib_gae_{m}.so",
                        f"{ROOT}/mobrob_amd/csrc/engine.hip"], check=True)
elif len(sys.argv) == 2:
    import time
    from mobrob_amd import _lib
    _lib.LIB_PATH = f"{ROOT}/scratch/lib_gae_{sys.argv[1]}.so"
    from mobrob_amd.engine import PPOEngine
    T, N = 1000, 4096
    e = PPOEngine(obs_dim=4, act_dim=2, n_envs=N, n_steps=T, batch_size=4096, n_epochs=1)
    e.collect_synthetic()
    for _ in range(3): e.compute_gae()
    e.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): e.compute_gae()
    e.synchronize()
    dt = (time.perf_counter() - t0) / 200
    print(f"GAE_SKIP={sys.argv[1]:>2s}  {dt*1e6:6.1f} us  ({T*N*20/dt/1e12:.2f} TB/s if it were the full kernel)")
else:
    for m in MASKS:
        subprocess.run([sys.executable, __file__, str(m)])
