"""Static checks on the SHIPPED code object of k_chain_train (no GPU needed: the disassembly of libmobrob_ppo.so's gfx950 image).

The kernel waits for its dW1 running sums with a hand-counted `s_waitcnt vmcnt(N)` (csrc/kernels_chain.h, dW1 phase): the sums are
loaded by inline-asm `global_load_dwordx4 ... sc1` statements the compiler does not track, a whole phase ahead of their use.  That is
correct only while
  * nothing reads or copies a loaded register between its load and a wait that covers it (the quads must have been coalesced into
    the accumulator tuples), and
  * exactly 2 K1 + 2 gather loads and twelve ring DMAs -- and no other vector-memory instruction -- lie between the slab loads and
    the counted wait (the wait lets exactly those stay in flight).
Both are properties of the compiled binary, so they are checked on the binary: a compiler upgrade that breaks one fails here, on the
CPU, before any gradient depends on it."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(ROOT, "mobrob_amd", "libmobrob_ppo.so")

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-objdump")), reason="no llvm-objdump")


@pytest.fixture(scope="module")
def code_object(tmp_path_factory):
    d = tmp_path_factory.mktemp("isa")
    fat, co = str(d / "fat.bin"), str(d / "gfx950.co")
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", LIB], check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
    return co


def disassemble(co, symbol):
    r = subprocess.run([f"{LLVM}/llvm-objdump", "-d", f"--disassemble-symbols={symbol}", co], check=True, capture_output=True, text=True)
    out = []
    for line in r.stdout.splitlines():
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//", line)
        if m:
            out.append((m.group(1), m.group(2)))
    assert len(out) > 1000, "disassembly of %s is empty" % symbol
    return out


def regs(operands):
    """vector registers named in an operand string: {'v12', 'v13', ...} (a[..] accumulators are a separate file)"""
    s = set()
    for m in re.finditer(r"(?<![a-z_])v\[(\d+):(\d+)\]", operands):
        s.update("v%d" % i for i in range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"(?<![a-z_\[:])v(\d+)\b", operands):
        s.add("v" + m.group(1))
    return s


VMEM = re.compile(r"^(global_|buffer_|flat_|scratch_)")


@pytest.mark.parametrize("dp", [16, 32, 64])
def test_counted_wait_of_the_dw1_sums(code_object, dp):
    k1 = (dp + 31) // 32
    ngl, nslab = 2 * k1 + 2, 16 if dp > 32 else 8
    ins = disassemble(code_object, "_ZN6mobrob13k_chain_trainILi%dEEEvNS_14FusedTrainArgsE" % dp)
    # the slab loads: the only sc1 vector loads of the kernel, one block
    idx = [i for i, (op, args) in enumerate(ins) if op == "global_load_dwordx4" and args.rstrip().endswith("sc1")]
    assert len(idx) == nslab, "expected %d slab loads (sc1), found %d" % (nslab, len(idx))
    assert idx[-1] - idx[0] < 4 * nslab, "the slab loads are not one block"
    dest = set()
    for i in idx:
        dest |= regs(ins[i][1].split(",")[0])
    assert len(dest) == 4 * nslab
    # the counted waits: vmcnt(NGL) (next tile not primed) and vmcnt(NGL + 12) (primed), the first ones of those counts behind the loads
    want = {ngl, ngl + 12}
    waits = {}
    for i in range(idx[-1] + 1, len(ins)):
        if ins[i][0] == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", ins[i][1])
            if m and int(m.group(1)) in want and int(m.group(1)) not in waits:
                waits[int(m.group(1))] = i
        if len(waits) == 2:
            break
    assert set(waits) == want, "counted waits vmcnt(%d) / vmcnt(%d) not found behind the slab loads: %r" % (ngl, ngl + 12, waits)
    end = max(waits.values())
    assert end - idx[-1] < 1500, "the counted waits are implausibly far from the slab loads"
    gathers = dmas = 0
    covered = False    # a full wait (vmcnt(0)) in between covers the slab loads by itself
    for i in range(idx[-1] + 1, end):
        op, args = ins[i]
        if op == "s_waitcnt" and "vmcnt(0)" in args:
            covered = True
        if VMEM.match(op):
            if op == "global_load_lds_dwordx4":
                dmas += 1
            elif op == "global_load_dwordx4":
                gathers += 1
                assert not (regs(args.split(",")[0]) & dest), "a gather overwrites a slab-sum register: %s %s" % (op, args)
            else:
                raise AssertionError("unexpected vector-memory instruction between the slab loads and their wait: %s %s" % (op, args))
            continue
        if not covered and op not in ("s_waitcnt",):
            hit = regs(args) & dest
            assert not hit, "%s %s touches %s before the wait that covers the slab loads" % (op, args, sorted(hit))
    assert gathers == ngl, "expected %d gather loads between the slab loads and the counted wait, found %d" % (ngl, gathers)
    assert dmas == 12, "expected 12 priming ring DMAs between the slab loads and the counted wait, found %d" % dmas


@pytest.mark.parametrize("dp", [16, 32, 64])
def test_ring_dma_waits_count_six(code_object, dp):
    """the ring protocol's segment boundaries wait with vmcnt(6): six DMA instructions per wave and segment.  Every wave issues
    3 pieces x 2 units per segment; the count of LDS-DMA instructions in the kernel must be a multiple of six per stream."""
    ins = disassemble(code_object, "_ZN6mobrob13k_chain_trainILi%dEEEvNS_14FusedTrainArgsE" % dp)
    n_dma = sum(1 for op, _ in ins if op == "global_load_lds_dwordx4")
    assert n_dma % 6 == 0 and n_dma > 0, n_dma
    # no spill code: a scratch access would also sit in the vmcnt queue and break every counted wait
    assert not any(op.startswith("scratch_") for op, _ in ins)
