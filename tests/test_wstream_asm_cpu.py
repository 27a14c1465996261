"""CPU: k_gemm_wstream issues its weight loads from inline asm and retires them with counted `s_waitcnt vmcnt(N)`, which
hipcc cannot see: between a load and the wait that covers it the compiler must not touch the destination registers (a copy
or an early reuse would read / clobber bytes still in flight).  This compiles ze_gemm.hip to gfx950 assembly (hipcc
cross-compiles without a GPU) and walks every instance's control-flow graph with tools/check_wstream_asm.py."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_no_instruction_touches_a_weight_register_in_flight(tmp_path):
    src = os.path.join(ROOT, "zoomearth_amd", "csrc", "ze_gemm.hip")
    out = tmp_path / "ze_gemm.s"
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-strict-aliasing", "-fno-slp-vectorize",
                        "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(src), "--cuda-device-only", "-S", src, "-o", str(out)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    c = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_wstream_asm.py"), str(out)], capture_output=True, text=True)
    assert c.returncode == 0, c.stdout[-3000:]
    lines = [ln for ln in c.stdout.splitlines() if "weight loads" in ln]
    assert len(lines) >= 6 and all(" 0 early touches" in ln and " 96 weight loads" in ln for ln in lines), c.stdout[-2000:]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_no_instruction_touches_an_attention_load_in_flight(tmp_path):
    """The per-wave decode attention kernels (k_attn_decode_wave, k_attn_decode_wave_long) issue their Q / K loads from inline
    asm and retire them inside the asm statement that consumes them: tools/check_attn_asm.py walks the generated code of both
    with the queue of in-flight vector-memory operations as state (every path of the pipelined kernel's six straight-line bodies)
    and reports any instruction that reads or writes a register whose load has not been retired."""
    src = os.path.join(ROOT, "zoomearth_amd", "csrc", "ze_attn_batch.hip")
    out = tmp_path / "ze_attn_batch.s"
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-strict-aliasing", "-fno-slp-vectorize",
                        "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(src), "--cuda-device-only", "-S", src, "-o", str(out)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    c = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_attn_asm.py"), str(out)], capture_output=True, text=True)
    assert c.returncode == 0, c.stdout[-3000:]
    lines = [ln for ln in c.stdout.splitlines() if "vector-memory operations" in ln]
    assert len(lines) == 2 and all(" 0 early touches" in ln for ln in lines), c.stdout[-2000:]
    assert any("wave_long" in ln for ln in lines)
