"""ISA-level pins of the gfx950 kernels inside librt_amd.so (no GPU needed: the code objects are disassembled here).

Round 3's k_shade let whole waves `return` in front of the block's __syncthreads() -- fine on CDNA hardware (s_barrier does
not count ended waves), undefined in HIP.  Round 4 removed the construct: the shading kernels are barrier-free persistent
waves.  These tests fail if a later change (or a compiler upgrade) brings a block barrier back into them, or lets the one
barrier of the traversal kernel -- behind the LDS staging of the top of the tree, reached by every wave of a block that
takes part -- move behind a wave-level exit.
"""
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from kernel_hash import _device_images  # noqa: E402

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
LIB = os.path.join(ROOT, "rustraytracer_amd", "librt_amd.so")


def _disassembly():
    """{kernel symbol: [(address, instruction text)]} over every gfx950 code object of the library."""
    out = {}
    blob = open(LIB, "rb").read()
    for img in _device_images(blob):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(img)
            f.flush()
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], stdout=subprocess.PIPE, text=True, check=True).stdout
        cur = None
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
            if m:
                cur = m.group(1)
                out[cur] = []
            elif cur is not None and line.startswith("\t"):
                body, _, tail = line.partition("//")
                am = re.match(r"\s*([0-9A-Fa-f]+):", tail)
                out[cur].append((int(am.group(1), 16) if am else -1, body.strip()))
    return out


@pytest.fixture(scope="module")
def isa():
    if not (os.path.exists(OBJDUMP) and os.path.exists(LIB)):
        pytest.skip("llvm-objdump or librt_amd.so missing")
    return _disassembly()


def _kernels(isa, needle):
    ks = {k: v for k, v in isa.items() if needle in k and ".kd" not in k}
    assert ks, needle
    return ks


def test_shading_kernels_have_no_block_barrier(isa):
    # (k_classify_scan is one block that scans through LDS: block barriers are its job)
    for needle in ("k_shade_cls", "k_shade_light", "k_classify_count", "k_classify_scatter", "k_tail", "k_generate"):
        for name, ins in _kernels(isa, needle).items():
            assert not any(s.startswith("s_barrier") for _, s in ins), name


def test_traversal_kernel_barrier_is_reached_by_every_wave_that_stays(isa):
    """k_trace: exactly one s_barrier (after the top of the tree is staged in LDS).  In front of it no wave may end on
    its own: no s_endpgm in line, and every branch that leaves the region in front of the barrier is taken on a SCALAR
    condition (block-uniform: `the queue gives this block nothing`), never on EXEC / VCC (some lanes or waves only)."""
    for name, ins in _kernels(isa, "k_traceIL").items():
        bars = [i for i, (_, s) in enumerate(ins) if s.startswith("s_barrier")]
        assert len(bars) == 1, (name, len(bars))
        bar_addr = ins[bars[0]][0]
        for addr, s in ins[:bars[0]]:
            assert not s.startswith("s_endpgm"), name
            m = re.match(r"s_cbranch_(execz|execnz|vccz|vccnz)\s+(-?\d+)", s)
            if m:
                target = addr + 4 + 4 * int(m.group(2))
                assert addr < target <= bar_addr, (name, s, hex(addr), hex(target), hex(bar_addr))


def test_no_kernel_uses_mfma(isa):
    """BASELINE north_star: there is no dense contraction on this path, so no matrix-core instruction may appear."""
    for name, ins in isa.items():
        assert not any("mfma" in s for _, s in ins), name
