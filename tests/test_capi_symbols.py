"""CPU: the C-ABI library loads and exports every symbol include/mobrob_ppo.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "mobrob_ppo.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mobrob_(?:ppo|ctrl)_[a-z_0-9]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    from mobrob_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = header_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mobrob_ppo.h but not exported"
        assert n in _lib.SYMBOLS, f"{n} has no ctypes prototype in mobrob_amd/_lib.py"
    assert set(_lib.SYMBOLS) == set(names)
    assert lib.mobrob_ppo_abi_version() == _lib.ABI_VERSION


def test_config_struct_layout_matches_header():
    """default_config() needs no GPU: check a few fields land where the ctypes mirror expects them."""
    from mobrob_amd import _lib
    lib = _lib.load()
    cfg = _lib.Config()
    lib.mobrob_ppo_default_config(ctypes.byref(cfg))
    assert (cfg.abi_version, cfg.n_steps, cfg.batch_size, cfg.n_epochs) == (_lib.ABI_VERSION, 2048, 64, 10)
    assert (cfg.gamma, cfg.gae_lambda, cfg.clip_range, cfg.vf_coef, cfg.max_grad_norm) == (0.99, 0.95, 0.2, 0.5, 0.5)
    assert (cfg.learning_rate, cfg.adam_eps, cfg.action_low, cfg.action_high) == (3e-4, 1e-5, -1.0, 1.0)
    assert (cfg.world_size, cfg.fast_kernels, cfg.normalize_advantage) == (1, 1, 1)
    assert list(cfg.pi_hidden) == [64, 64]
    assert cfg.pi_hidden3 == 0 and list(cfg.pi_hidden_ext) == [0] * 5 and list(cfg.vf_hidden_ext) == [0] * 5


def test_config_struct_size_and_offsets_are_the_c_compiler_s(tmp_path):
    """sizeof / offsetof of mobrob_ppo_config_t as gcc lays the header out == the ctypes mirror (every field, by name)."""
    import subprocess
    from mobrob_amd import _lib
    names = [n for n, _ in _lib.Config._fields_]
    prog = ('#include <stdio.h>\n#include <stddef.h>\n#include "mobrob_ppo.h"\nint main(void) {\n'
            '  printf("%zu\\n", sizeof(mobrob_ppo_config_t));\n'
            + "".join(f'  printf("%zu\\n", offsetof(mobrob_ppo_config_t, {n}));\n' for n in names) + "  return 0;\n}\n")
    src, exe = tmp_path / "layout.c", tmp_path / "layout"
    src.write_text(prog)
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    out = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert out[0] == ctypes.sizeof(_lib.Config)
    assert out[1:] == [getattr(_lib.Config, n).offset for n in names]


def test_no_device_fails_loudly():
    """Without a GPU the product must raise, never fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mobrob_amd.engine import PPOEngine
    with pytest.raises(Exception) as ei:
        PPOEngine(obs_dim=4, act_dim=2, n_envs=2, n_steps=2)
    assert "no CPU fallback" in str(ei.value) or "HIP" in str(ei.value) or "device" in str(ei.value)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mobrob_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f"{f} imports the oracle"
    for f in ("examples/train.py", "examples/control.py"):
        p = os.path.join(ROOT, f)
        if os.path.exists(p):
            assert "oracle" not in open(p).read()


def test_build_fails_on_register_spills():
    """__graft_entry__.build() parses hipcc's kernel-resource-usage remarks and refuses a library in which any kernel spills
    a vector register or touches scratch memory; the audit of the shipped binary sits beside it."""
    import __graft_entry__ as g
    ok = ("engine.hip:10:1: remark: Function Name: _Zk_good [-Rpass-analysis=kernel-resource-usage]\n"
          "engine.hip:10:1: remark:     SGPRs: 40 [-Rpass-analysis=kernel-resource-usage]\n"
          "engine.hip:10:1: remark:     VGPRs: 242 [-Rpass-analysis=kernel-resource-usage]\n"
          "engine.hip:10:1: remark:     AGPRs: 256 [-Rpass-analysis=kernel-resource-usage]\n"
          "engine.hip:10:1: remark:     ScratchSize [bytes/lane]: 0 [-Rpass-analysis=kernel-resource-usage]\n"
          "engine.hip:10:1: remark:     Occupancy [waves/SIMD]: 1 [-Rpass-analysis=kernel-resource-usage]\n"
          "engine.hip:10:1: remark:     SGPRs Spill: 12 [-Rpass-analysis=kernel-resource-usage]\n"
          "engine.hip:10:1: remark:     VGPRs Spill: 0 [-Rpass-analysis=kernel-resource-usage]\n")
    rows = g.check_kernel_resources(ok)
    assert rows == [("_Zk_good", 242, 256, 0, 12, 0, 1)]
    for bad in (ok.replace("VGPRs Spill: 0", "VGPRs Spill: 7"), ok.replace("ScratchSize [bytes/lane]: 0", "ScratchSize [bytes/lane]: 32")):
        with pytest.raises(RuntimeError, match="_Zk_good"):
            g.check_kernel_resources(bad)
    audit = os.path.join(ROOT, "mobrob_amd", "kernel_resources.txt")
    if os.path.exists(audit):     # written by the build that produced the shipped library
        lines = open(audit).read().strip().splitlines()
        assert len(lines) > 100 and any("k_fused_train" in l for l in lines)


def test_build_reuse_is_keyed_on_a_hash_of_the_sources(tmp_path):
    """build() reuses a library only while the sidecar beside it holds the sha256 of csrc/ + include/ + the compile flags
    (VERDICT r3 #11): a changed source, a changed flag or a missing sidecar force a compile; mtimes play no part."""
    import time
    import __graft_entry__ as G
    src, lib = tmp_path / "a.hip", str(tmp_path / "lib.so")
    src.write_text("int f() { return 1; }\n")
    flags = ["-O3"]
    assert G.needs_build(lib, [str(src)], flags)                 # nothing built yet
    open(lib, "wb").write(b"\x7fELF")
    assert G.needs_build(lib, [str(src)], flags)                 # a library without a sidecar is not trusted
    G._stamp(lib, [str(src)], flags)
    assert not G.needs_build(lib, [str(src)], flags)
    os.utime(src, (time.time() + 100, time.time() + 100))         # touched, content unchanged: still current
    assert not G.needs_build(lib, [str(src)], flags)
    src.write_text("int f() { return 2; }\n")
    os.utime(src, (1, 1))                                         # changed content with an OLD mtime (a reverted edit): rebuild
    assert G.needs_build(lib, [str(src)], flags)
    src.write_text("int f() { return 1; }\n")
    assert not G.needs_build(lib, [str(src)], flags)
    assert G.needs_build(lib, [str(src)], flags + ["-g"])        # the flags are part of the key
    # the shipped libraries carry their sidecars and they match the tree
    G.build()
    assert not G.needs_build(G.LIB, G.engine_sources(), G.HIPCC_FLAGS)
    assert not G.needs_build(G.ENV_LIB, [G.ENV_SRC], G.ENV_FLAGS)
