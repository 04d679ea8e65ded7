"""Per-phase wall-clock shares of k_epoch64 at the reference YAML shape (stamps build: scratch/build_variant.sh epst -DMOBROB_EPOCH_STAMPS;
STAMPS_LIB=scratch/lib_epst.so python scratch/epoch_stamps.py).  Workgroup 0 (a gradient, reduction and Adam workgroup) sums
wall_clock64() intervals (100 MHz) per phase over the launches of one train()."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, '.')
from mobrob_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ.get("STAMPS_LIB", "scratch/lib_epst.so"))
from mobrob_amd.engine import PPOEngine
from mobrob_amd.rl_control.init import orthogonal_policy_init
D, A, H, N, T, B, E = 58, 12, 64, 16, 1000, 100, 5
e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=E, pi=(H, H), vf=(H, H), ent_coef=0.01)
e.set_params(orthogonal_policy_init(D, A, (H, H), (H, H), 0))
e.collect_synthetic()
e.train(None)
out = (C.c_ulonglong * 32)()
e.lib.mobrob_dbg_read_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
e.lib.mobrob_dbg_read_stamps(e._h, out, 1)
e.collect_synthetic()
e.train(None)
e.lib.mobrob_dbg_read_stamps(e._h, out, 1)
assert e.update_mode() == 1
v = np.array(list(out)[:7], dtype=np.float64)
w = np.array(list(out)[8:14], dtype=np.float64)
steps = v[6]
names = ["A gradient (k_split64_train's body)", "barrier 1 (slabs -> reduction)", "B slab reduction + norm records", "barrier 2 (gradient, records -> Adam)",
         "C clip + Adam + packs", "barrier 3 (weights -> next gradient)"]
print(f"k_epoch64, doggo reference YAML shape ({N} envs x {T} steps, batch {B}, 2x{H}, {E} epochs): workgroup 0, microseconds per optimizer step over {int(steps)} steps")
print(f"  {'phase':44s} {'wg 0':>10s} {'wg 81 (statistics duties)':>28s}")
for k in range(6):
    print(f"  {names[k]:44s} {v[k] / steps / 100:7.2f} us {w[k] / steps / 100:25.2f} us")
print(f"  {'sum':44s} {v[:6].sum() / steps / 100:7.2f} us {w[:6].sum() / steps / 100:25.2f} us")
