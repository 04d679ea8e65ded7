"""Diagnostic: per-phase cycle shares of k_chain_train (stamps build: scratch/libmobrob_ppo_stamps.so, -DMOBROB_STAMPS)."""
import ctypes as C, sys, os, numpy as np
sys.path.insert(0, '.')
from mobrob_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ.get("STAMPS_LIB", "scratch/libmobrob_ppo_stamps.so"))
from mobrob_amd.engine import PPOEngine
from mobrob_amd.rl_control.init import orthogonal_policy_init
D, A, H, N, T, B = 58, 12, 256, 4096, int(os.environ.get('STAMPS_T', '1000')), 65536
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
names = {0: "X image + split", 1: "layer 1 (ring)", 2: "layer 2 (ring) + end barrier", 3: "h2 tanh + head + loss + dh2", 4: "barrier",
         5: "dW3", 6: "barrier + dz2 image + barrier", 7: "dW2", 8: "barrier", 9: "dh1 (ring) + dz1", 10: "gather issue + barrier",
         11: "dW1 stores", 12: "end barrier", 13: "  dW1 prologue (4 fragments)", 14: "  dW1 slab-load wait", 15: "  dW1 loop",
         16: "  dh1 ring prologue (first DMA wait)", 17: "  dh1 loop", 18: "  fwd ring prologue (first DMA wait)"}
v = np.array(list(out), dtype=np.float64)
tot = v[:19].sum()
for i in range(19):
    print(f"{i:2d} {names[i]:34s} {100 * v[i] / tot:6.2f}%   {v[i] / ((N * T + B - 1) // B * 256 * 8 * 4):12.0f} cycles/wave/tile")
launches = (N * T + B - 1) // B
print("total cycles/wave/tile", v[:19].sum() / (launches * 256 * 8 * 4))
print(f"in-kernel clock {v[24] / v[25] * 0.1:.3f} GHz; kernel span per wave {v[25] / (launches * 256 * 4) / 100:.1f} us ({launches} launches)")
