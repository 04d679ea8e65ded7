"""Diagnostic: per-phase cycle shares of k_fused_train (stamps build)."""
import ctypes as C, sys, os, numpy as np
sys.path.insert(0, '.')
from mobrob_amd import _lib
_lib.LIB_PATH = os.path.abspath("scratch/libmobrob_ppo_stamps.so")
from mobrob_amd.engine import PPOEngine
from mobrob_amd.rl_control.init import orthogonal_policy_init
D, A, H, N, T, B = 58, 12, 256, 4096, 64, 65536
e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=1, pi=(H, H), vf=(H, H), ent_coef=0.01)
e.set_params(orthogonal_policy_init(D, A, (H, H), (H, H), 0))
e.collect_synthetic()
e.train(None)
lib = e.lib
out = (C.c_ulonglong * 32)()
lib.mobrob_dbg_read_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
lib.mobrob_dbg_read_stamps(e._h, out, 1)
e.train(None)
lib.mobrob_dbg_read_stamps(e._h, out, 1)
names = {0: "gather+barrier", 1: "L1 gemm", 2: "L1 tanh epi", 3: "barrier", 4: "L2 gemm", 5: "L2 tanh epi", 6: "barrier",
         7: "head gemm", 8: "head reduce(2 barriers)", 9: "loss + barrier", 10: "dW3 gemm + slab RMW", 11: "dh2 gemm",
         12: "barrier", 13: "dz2 epi", 14: "barrier", 15: "dW2 gemm", 16: "dh1 gemm", 17: "barrier", 18: "dz1 epi",
         19: "barrier", 20: "dW1 gemm + slab RMW", 21: "end barrier", 22: "loop exit"}
v = np.array(list(out), dtype=np.float64)
tot = v.sum()
for i in range(23):
    print(f"{i:2d} {names[i]:28s} {100 * v[i] / tot:6.2f}%   {v[i] / (1024 * 8 * 4):12.0f} cycles/wave/tile")
print("total cycles/wave/tile", tot / (1024 * 8 * 4))
