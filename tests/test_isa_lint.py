"""ISA lint for the hand-scheduled scalar loads of the mask-head kernels (vnext_amd/csrc/mask_head.hip).

Those kernels issue `s_load_dwordx16 / x8 / x4 / x2` in one asm statement and wait for them (`s_waitcnt lgkmcnt(0)`) in
another, so that the next group of parameters is in flight while the current one is consumed.  The compiler does not
know that the destination registers are not valid until the wait: if it copied, moved or spilled one of them in
between, the kernel would read stale values.  This test compiles the file for gfx950 (no GPU needed) and checks,
instruction by instruction, that nothing touches the destination of a hand-issued scalar load before the next
`s_waitcnt lgkmcnt(0)`, that the forward kernels spill nothing at all, and that no kernel uses scratch memory."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "vnext_amd", "csrc", "mask_head.hip")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "mask_head.s"
    # the development build: the product's kernels (runs forward, backward) + the strip forward kernels kept for A/B
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-DVNX_DEV_VARIANTS", "-S",
                           "--cuda-device-only", "-o", str(out), SRC], stderr=subprocess.DEVNULL)
    text = open(out).read()
    funcs = {}
    for m in re.finditer(r"^(_ZN3vnx\w*mask_head\w*):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M):
        funcs[m.group(1)] = m.group(2)
    assert len(funcs) >= 4, list(funcs)
    return text, funcs


def sregs(token_text):
    """scalar registers named in an operand string: s7 -> {7}, s[8:23] -> {8..23}"""
    regs = set()
    for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", token_text):
        regs.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bs(\d+)\b", token_text):
        regs.add(int(a))
    return regs


def test_nothing_touches_a_scalar_load_destination_before_the_wait(asm):
    _, funcs = asm
    checked = 0
    for name, body in funcs.items():
        pending = set()          # destination registers of scalar loads issued since the last full wait
        for line in body.splitlines():
            ins = line.split(";")[0].strip()
            if not ins or ins.startswith(".") or ins.endswith(":"):
                continue
            op, _, rest = ins.partition(" ")
            if op.startswith("s_load_dword"):
                dst = rest.split(",")[0]
                assert not (sregs(rest.split(",", 1)[1]) & pending), (name, ins)      # base / offset still in flight
                pending |= sregs(dst)
                checked += 1
                continue
            if op == "s_waitcnt":
                if "lgkmcnt(0)" in rest:
                    pending.clear()
                continue
            if pending and op not in ("s_nop",):
                assert not (sregs(rest) & pending), f"{name}: `{ins}` touches a scalar load still in flight"
    assert checked > 50        # the kernels do issue their loads this way


def test_forward_kernels_spill_nothing_and_nobody_uses_scratch(asm):
    text, funcs = asm
    for name, body in funcs.items():
        if "bwd" not in name:
            assert "v_readlane_b32" not in body and "v_writelane_b32" not in body, name
        assert "scratch_" not in body, name
        # the only buffer stores are the runs kernel's output stream (dwordx2, `nt`): anything else would be a spill
        for line in body.splitlines():
            if "buffer_store_dword" in line:
                assert "runs_kernel" in name and "buffer_store_dwordx2" in line and line.rstrip().endswith(" nt"), (name, line)
    for m in re.finditer(r"\.private_segment_fixed_size:\s*(\d+)", text):
        assert int(m.group(1)) == 0


# ---- the slab forward's samples requested one tile ahead (vnext_amd/csrc/msda_d32.hip: msda_fwd_slab_kernel) ------------------
# The next tile's locations and weights are loaded by an asm statement and waited for, one tile later, by another: in between
# (the whole gather phase, at 126 VGPRs) the compiler believes the destination registers hold values.  If it moved, copied or
# spilled one of them there, the decode would read what the register held BEFORE the load landed.  Same class of hazard as the
# mask head's scalar loads above; same kind of check.
D32 = os.path.join(ROOT, "vnext_amd", "csrc", "msda_d32.hip")


def vregs(token_text):
    regs = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", token_text):
        regs.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", token_text):
        regs.add(int(a))
    return regs


def test_nothing_touches_the_slab_forwards_prefetched_samples_before_the_wait(tmp_path):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path / "msda_d32.s"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-I", os.path.dirname(D32), "-S",
                           "--cuda-device-only", "-o", str(out), D32], stderr=subprocess.DEVNULL)
    text = open(out).read()
    # every unfused instantiation with fp32 locations prefetches its samples this way: fp32, bf16 and f16 values (round 6)
    # -- and, for 16-bit values, in both slab sizes (8 waves x 320 rows of 128 B; 16 waves x 600: the 720p slab)
    for tv, want in (("f", 1), ("NS_6bf16_tE", 2), ("NS_5f16_tE", 2)):
        found = list(re.finditer(r"^(_ZN3vnx20msda_fwd_slab_kernelI" + tv + r"fLb0E(?:Li\d+E)*E\w*):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M))
        assert len(found) == want, "the unfused slab kernels (%s): %d instantiations, expected %d" % (tv, len(found), want)
        for m in found:
            _check_slab_prefetch(m)


def _check_slab_prefetch(m):
    lines = [l.split(";")[0].strip() for l in m.group(2).splitlines()]
    # hand-issued loads: the instructions between ;;#ASMSTART / ;;#ASMEND markers (the split above removed the markers'
    # text, so find them in the raw body)
    raw = m.group(2).splitlines()
    in_asm, asm_loads, asm_waits = False, [], []
    for i, l in enumerate(raw):
        if "#ASMSTART" in l:
            in_asm = True
        elif "#ASMEND" in l:
            in_asm = False
        elif in_asm and "global_load_dword" in l:
            asm_loads.append(i)
        elif in_asm and "s_waitcnt vmcnt(0)" in l:
            asm_waits.append(i)
    assert len(asm_loads) == 8 and len(asm_waits) == 1, (asm_loads, asm_waits)      # 2 x 2 before the loop, 2 x 2 inside; one wait
    wait = asm_waits[0]
    dst = set()
    for i in asm_loads:
        dst |= vregs(lines[i].split(",")[0])
    assert 4 <= len(dst) <= 6, dst
    before, inside = [i for i in asm_loads if i < wait], [i for i in asm_loads if i > wait]
    assert len(before) == 4 and len(inside) == 4
    # forbidden zones: (last load before the loop, the wait) and (last load inside the loop, end of the kernel's loop body)
    zones = list(range(before[-1] + 1, wait)) + list(range(inside[-1] + 1, len(lines)))
    for i in zones:
        ins = lines[i]
        if not ins or ins.startswith(".") or ins.endswith(":") or ins.startswith("s_"):
            continue
        op, _, rest = ins.partition(" ")
        assert not (vregs(rest) & dst), f"`{ins}` (line {i}) touches a prefetched sample register {sorted(dst)} before its wait"
