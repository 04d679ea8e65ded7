"""How close is test_benchmarked_epoch_matches_oracle[doggo] to its 1e-4 bound, with and without the x3 forward kernels?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ppo_oracle as O
from tests.test_full_size_gpu import _bench_like_engine, _device_perm_key

for seed, rs in ((23, 6), (24, 7), (25, 8)):
    D, A, H, n_envs, T, B = 58, 12, 256, 4096, 1000, 65536
    rng = np.random.default_rng(rs)
    e, p, st, buf, h = _bench_like_engine(D, A, H, n_envs, T, B, seed, rng)
    perm = O.feistel_permutation(T * n_envs, _device_perm_key(seed, 0))
    e.train(None)
    O.train(p, st, buf, h, perm[None])
    newp = e.get_params()
    errs = {k: float(np.max(np.abs(newp[k] - p[k]))) for k in p}
    worst = max(errs, key=errs.get)
    print("x3" if not os.environ.get("MOBROB_NO_X3") else "f32", "seed", seed, "worst", worst, f"{errs[worst]:.2e}", "log_std", f"{errs['log_std']:.2e}", flush=True)
    e.close()
