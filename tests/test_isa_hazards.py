"""The hand-written DPP adds of the integer reductions against the gfx9 data hazard they must respect.

``csrc/ma_internal.h`` spells the 64-bit DPP add in assembly (v_add_co_u32_dpp + v_addc_co_u32_dpp); the compiler's
hazard recogniser does not look inside an asm block, so the distance rule — a VGPR written by a VALU instruction may be read
THROUGH DPP only two wait states later — is the source's to keep: the general form carries an ``s_nop 1``, the "far" form
used by reduce_chunk's two passes relies on the 17 accumulators lying between a value's producer and its DPP read.  This
test disassembles the gfx950 device code of the kernels that use them (hipcc cross-compiles here, no GPU needed) and checks
every such instruction: none of the two preceding wait states may write the register it reads through DPP, and none of the
five preceding ones may be a VALU write of EXEC (v_cmpx).
"""
import os
import re
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "3d-localisation-and-mapping_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"


def _disassemble(unit, bdir):
    obj = os.path.join(bdir, f"isa_{unit}.o")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"),
                    "-mllvm", "-amdgpu-kernarg-preload-count=16", "--cuda-device-only", "-c", "-o", obj,
                    os.path.join(CSRC, f"{unit}.hip")], check=True)
    subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={obj}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={obj}.elf"], check=True)
    out = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", "--no-leading-addr", f"{obj}.elf"],
                         check=True, capture_output=True, text=True).stdout
    kernels, cur = {}, None
    for line in out.splitlines():
        m = re.match(r"^<(.*)>:$", line.strip())
        if m:
            cur = m.group(1)
            kernels[cur] = []
        elif cur is not None and line.strip():
            kernels[cur].append(re.sub(r"\s*//.*$", "", line.strip()))
    return kernels


def _regs(tok):
    """'v12' -> {12}; 'v[4:5]' -> {4, 5}; anything else -> empty."""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def _written(ins):
    """VGPRs an instruction writes (its first operand; loads and LDS reads complete asynchronously and are guarded by s_waitcnt,
    which the compiler places itself — they are not this hazard)."""
    op, _, rest = ins.partition(" ")
    if not op.startswith("v_") or op.startswith(("v_cmp", "v_nop")):
        return set()
    first = rest.split(",")[0].strip()
    return _regs(first)


@pytest.mark.parametrize("unit", ["kernels_search", "kernels_solve"])
def test_hand_written_dpp_adds_keep_two_wait_states_from_their_producers(unit):
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc on this box")
    bdir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(bdir, exist_ok=True)
    kernels = _disassemble(unit, bdir)
    checked = 0
    for name, ins in kernels.items():
        for i, line in enumerate(ins):
            op, _, rest = line.partition(" ")
            if op not in ("v_add_co_u32_dpp", "v_addc_co_u32_dpp"):
                continue
            ops = [t.strip() for t in rest.split(",")]
            src0 = _regs(ops[2].split(" ")[0])            # vD, vcc, vSRC0(dpp), vSRC1 ...
            assert src0, line
            # walk back over two wait states (an instruction is one; s_nop N is N + 1)
            states, j = 0, i - 1
            while states < 2 and j >= 0:
                prev = ins[j]
                pop = prev.split(" ")[0]
                if pop == "v_add_co_u32_dpp" and j == i - 1 and op == "v_addc_co_u32_dpp":
                    assert not (_written(prev) & src0), (name, prev, line)      # the pair's low half writes another register
                else:
                    assert not (_written(prev) & src0), f"{name}: `{prev}` writes a register `{line}` reads through DPP {states} wait state(s) later"
                states += int(prev.split(" ")[1]) + 1 if pop == "s_nop" else 1
                j -= 1
            # second rule: a VALU write of EXEC (v_cmpx*) needs five wait states before a DPP instruction
            states, j = 0, i - 1
            while states < 5 and j >= 0:
                prev = ins[j]
                pop = prev.split(" ")[0]
                assert not pop.startswith("v_cmpx"), f"{name}: `{prev}` writes EXEC {states} wait state(s) before `{line}`"
                states += int(prev.split(" ")[1]) + 1 if pop == "s_nop" else 1
                j -= 1
            checked += 1
    assert checked >= (34 * 6 if unit == "kernels_search" else 12), checked
