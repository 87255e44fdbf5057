"""The contract of the hand-scheduled loops of the 2-bit kernel with the compiler.

The loops' read-write operands are not early-clobber (that register numbering runs 2 % slower), so an input-only operand whose
value the compiler can prove equal to a read-write operand's at the loop's entry may be given the SAME register, which the loop
then overwrites -- round 3 hit exactly that with a zero mask beside `sl` = 0.  Every asm statement therefore prints, as a
comment in the generated code, the registers of its read-write and of its input-only operands; this test compiles the device
code (hipcc -S, gfx950: no GPU needed) and fails when the two sets meet in any instantiation."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def _regs(tokens):
    out = set()
    for t in tokens:
        m = re.fullmatch(r"([vs])\[(\d+):(\d+)\]", t)
        if m:
            out |= {f"{m.group(1)}{k}" for k in range(int(m.group(2)), int(m.group(3)) + 1)}
            continue
        assert re.fullmatch(r"[vs]\d+|vcc|exec|m0|0x[0-9a-f]+|-?\d+", t), f"unexpected operand text {t!r}"
        if re.fullmatch(r"[vs]\d+", t):
            out.add(t)
    return out


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    out = tmp_path_factory.mktemp("asm") / "snacc_hip.s"
    src = ROOT / "snacc_amd" / "csrc"
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", f"-I{ROOT / 'include'}", "--offload-arch=gfx950", "--cuda-device-only", "-S",
                           "-o", str(out), "snacc_hip.hip"], cwd=src)
    return out.read_text()


def test_input_only_operands_never_share_a_register_with_a_read_write_operand(device_asm):
    lines = [ln for ln in device_asm.splitlines() if "snk-asm-contract" in ln]
    # every instantiation of the four loop forms (one lane / two lanes, with and without the class window, far chains) in every
    # kernel that carries them
    assert len(lines) >= 20, len(lines)
    spec = 0
    for ln in lines:
        m = re.search(r"snk-asm-contract inout (.*) \| in (.*)$", ln)
        assert m, ln
        inout, inp = _regs(m.group(1).split()), _regs(m.group(2).split())
        assert len(inout) >= 16 and len(inp) >= 10, ln
        assert not (inout & inp), f"an input-only operand shares {sorted(inout & inp)} with a read-write operand: {ln.strip()}"
        # the loops' hard temporaries (clobbers: v90..v127, the two-lane loops v88..v127) are not operands either
        two_lanes = len(m.group(1).split()) == 18
        hard = {f"v{k}" for k in range(88 if two_lanes else 90, 128)}
        assert not ((inout | inp) & hard), f"an operand sits in a clobbered temporary: {ln.strip()}"
        spec += two_lanes
    assert spec >= 8, spec                     # the two-lane loops (18 read-write operands) are among them


def test_every_loop_carries_its_contract(device_asm):
    """No hand-scheduled loop goes without a contract line: each table read of a loop (`ds_or_rtn_b32 ... offset:1792`, the
    first LDS instruction of a trip) has the contract comment of its asm statement a few lines above it."""
    lines = device_asm.splitlines()
    # (the hand-scheduled loops read through their hard temporaries: v94 <- [v92] | v93 / v99; compiled C++ statements of the same
    # table read use whatever registers the compiler picked)
    reads = [i for i, ln in enumerate(lines) if re.search(r"ds_or_rtn_b32 v94, v92, v9[39] offset:1792", ln)]
    assert len(reads) >= 14, len(reads)
    for i in reads:
        assert any("snk-asm-contract" in ln for ln in lines[max(0, i - 40):i]), lines[max(0, i - 40):i + 1]


def test_every_loop_head_sits_on_a_64_byte_boundary(device_asm):
    """Where a hand-scheduled loop lies in the code decides 2.3 % of the two-lane loop's rate (heads 24 bytes past a 32-byte
    boundary are the slow ones; round 4), so every loop head is aligned: between a contract comment and
    the table read of the loop's first trip there is a `.p2align 6` followed directly by the loop's label."""
    lines = [ln.strip() for ln in device_asm.splitlines()]
    heads = [i for i, ln in enumerate(lines) if "snk-asm-contract" in ln]
    assert len(heads) >= 20
    for i in heads:
        body = lines[i + 1:i + 40]
        k = next((j for j, ln in enumerate(body) if ln.startswith(".p2align")), None)
        assert k is not None and body[k].split()[1].rstrip(",") == "6", body[:12]
        rest = [ln for ln in body[k + 1:k + 4] if not ln.startswith(".fill 0,")]
        assert rest[0] == "1:", body[k:k + 4]


def test_the_2bit_kernels_use_no_scratch(device_asm):
    """Every 2-bit kernel keeps its state in registers: a kernel that spills loses the schedule the loops were written for, and
    the one time a build of the exception kernels spilled (round 4: 12 bytes per lane, while the table swaps of the other-case
    mode were being written) it faulted on the card.  The compiler's own figure, from the generated code."""
    sizes = dict(re.findall(r"\.set (_Z\d+snk_fast\w+)\.private_seg_size, (\d+)", device_asm))
    assert len(sizes) >= 14, sorted(sizes)
    assert all(v == "0" for v in sizes.values()), {k: v for k, v in sizes.items() if v != "0"}
