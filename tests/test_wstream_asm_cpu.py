"""CPU: static checks on the assembly hipcc generates for the library (it cross-compiles gfx950 without a GPU).

Three properties that inline asm hides from the compiler and that therefore have to be proved on the generated code:
  * k_gemm_wstream issues its weight loads from inline asm and retires them with counted `s_waitcnt vmcnt(N)`: between a load
    and the wait that covers it nothing may touch the destination registers (tools/check_wstream_asm.py);
  * the per-wave decode attention kernels do the same with their Q / K loads (tools/check_attn_asm.py);
  * no VALU result reaches an MFMA in fewer than two wait states, no asm statement of MFMAs ends unpadded, no vector-memory
    instruction reads an SGPR a VALU instruction wrote fewer than five wait states earlier -- over EVERY kernel of EVERY unit
    (tools/check_mfma_hazards.py; the first rule is the root cause of round 4's mis-scheduled instantiations).
The units are compiled once per session, in parallel."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "zoomearth_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
UNITS = ["ze_frontend", "ze_elementwise", "ze_gemm", "ze_gemm_oneshot", "ze_gemv", "ze_gemv8", "ze_gemv_logits", "ze_attention",
         "ze_attn_decode", "ze_attn_batch", "ze_quant", "ze_sample"]
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")


def _flags(unit):
    fl = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-strict-aliasing", "-fno-slp-vectorize", "-I" + os.path.join(ROOT, "include"),
          "-I" + CSRC, "--cuda-device-only", "-S"]
    if unit == "ze_attn_decode":  # the Makefile keeps the SLP vectoriser for this unit
        fl.remove("-fno-slp-vectorize")
    return fl


def _compile(unit, out, extra=()):
    r = subprocess.run([HIPCC] + _flags(unit) + list(extra) + [os.path.join(CSRC, unit + ".hip"), "-o", out], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (unit, r.stderr[-2000:])
    return out


@pytest.fixture(scope="session")
def lib_asm(tmp_path_factory):
    d = tmp_path_factory.mktemp("libasm")
    with ThreadPoolExecutor(max_workers=6) as ex:  # ze_gemm.hip is the long pole (~2 min); the rest finish beside it
        outs = list(ex.map(lambda u: _compile(u, str(d / (u + ".s"))), UNITS))
    return dict(zip(UNITS, outs))


def _run(tool, *paths):
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)] + list(paths), capture_output=True, text=True)


def test_no_instruction_touches_a_weight_register_in_flight(lib_asm):
    c = _run("check_wstream_asm.py", lib_asm["ze_gemm"])
    assert c.returncode == 0, c.stdout[-3000:]
    lines = [ln for ln in c.stdout.splitlines() if "weight loads" in ln]
    assert len(lines) >= 6 and all(" 0 early touches" in ln and " 96 weight loads" in ln for ln in lines), c.stdout[-2000:]


def test_no_instruction_touches_an_attention_load_in_flight(lib_asm):
    """The per-wave decode attention kernels (k_attn_decode_wave, k_attn_decode_wave_long) issue their Q / K loads from inline
    asm and retire them inside the asm statement that consumes them: tools/check_attn_asm.py walks the generated code of both
    with the queue of in-flight vector-memory operations as state (every path of the pipelined kernel's six straight-line bodies)
    and reports any instruction that reads or writes a register whose load has not been retired."""
    c = _run("check_attn_asm.py", lib_asm["ze_attn_batch"])
    assert c.returncode == 0, c.stdout[-3000:]
    lines = [ln for ln in c.stdout.splitlines() if "vector-memory operations" in ln]
    # (three kernels since round 6: the pipelined kernel has a second instantiation for the split-row / paired form)
    assert len(lines) == 3 and all(" 0 early touches" in ln for ln in lines), c.stdout[-2000:]
    assert any("wave_long" in ln for ln in lines)


def test_no_mfma_hazard_hidden_by_inline_asm_in_any_kernel_of_the_library(lib_asm):
    """Rules A / B / C of tools/check_mfma_hazards.py over every kernel of every unit (about 20,000 MFMAs)."""
    c = _run("check_mfma_hazards.py", *lib_asm.values())
    assert c.returncode == 0, c.stdout[-4000:]
    tot = [ln for ln in c.stdout.splitlines() if ln.endswith(" hazards")]
    assert len(tot) == len(UNITS) and all(ln.endswith(" 0 hazards") for ln in tot), c.stdout[-2000:]
    n_mfma = sum(int(ln.split(" kernels, ")[1].split(" MFMAs")[0]) for ln in tot)
    assert n_mfma > 15000, c.stdout[-2000:]


@pytest.mark.parametrize("rounds", [3, 4])
def test_the_walker_flags_round_4s_broken_instantiations(tmp_path, rounds):
    """Negative control.  Round 4's form of the pipelined decode attention -- P converted by `asm("v_cvt_pk_bf16_f32")`, no scheduler
    fences, 192- / 256-key parts -- returned wrong rows on the GPU (tools/probes/fence_hunt.sh, gpurun_out of round 5: variants
    nofence / pre64 / post64 wrong, cvtpost / cvtbuiltin / either fence right).  The walker must find that bug in the assembly
    alone: a VALU write inside an asm statement one wait state in front of the MFMA that reads it."""
    out = _compile("ze_attn_batch", str(tmp_path / "bad.s"),
                   ["-include", os.path.join(ROOT, "tools", "probes", "fence_variants.h"), "-DFH=1", f"-DAW_LONG_ROUNDS={rounds}"])
    c = _run("check_mfma_hazards.py", out)
    assert c.returncode == 1, c.stdout[-2000:]
    hits = [ln for ln in c.stdout.splitlines() if ln.strip().startswith("A: `v_cvt_pk_bf16_f32") and "[asm]" in ln and "v_mfma_f32_16x16x16_bf16" in ln]
    assert hits and all(": 1 wait states" in ln or ": 0 wait states" in ln for ln in hits), c.stdout[-2000:]
    # the same unit with the conversion left to the compiler (what ships), still unfenced: clean
    ok = _compile("ze_attn_batch", str(tmp_path / "good.s"), [f"-DAW_LONG_ROUNDS={rounds}"])
    c2 = _run("check_mfma_hazards.py", ok)
    assert c2.returncode == 0 and " 0 hazards" in c2.stdout, c2.stdout[-2000:]
