"""
Properties of the gfx950 code object that the kernels' correctness or speed arguments lean on, read from
the ISA hipcc writes for the shipped sources (cross-compiles without a GPU).
"""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def isa():
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    tmp = tempfile.mkdtemp(prefix="prosstt_isa_")
    try:
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize",
                               "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-fPIC", "-shared",
                               "-fvisibility=hidden", "-save-temps", "-o", os.path.join(tmp, "lib.so"),
                               os.path.join(ROOT, "prosstt_amd", "csrc", "prosstt_amd.hip")],
                              cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        text = open(os.path.join(tmp, "prosstt_amd-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return text


def _body(text, mangled_part):
    m = re.search(r"^(_Z\w*%s\w*):[^\n]*\n(.*?)\n\s+s_endpgm" % mangled_part, text, re.S | re.M)
    assert m, mangled_part
    return m.group(2)


def _meta(text, mangled_part, key):
    for blk in re.split(r"\n  - \.agpr_count", text)[1:]:
        if re.search(r"\.name:\s+\S*%s" % mangled_part, blk):
            return int(re.search(r"\." + key + r":\s+(\d+)", blk).group(1))
    raise AssertionError(mangled_part)


@pytest.mark.parametrize("kernel", ["sample_counts_stream_kernelILb1ELb1E", "sample_counts_stream_kernelILb0ELb0E", "sample_counts_stream_kernelILb1ELb0E"])
def test_stream_kernel_memory_operations_are_global(isa, kernel):
    # a late 4-byte store must land after its row's 16-byte store: global_* and buffer_* operations of a wave are
    # performed in issue order, flat_* are not (k3_stream.h, flush_late)
    body = _body(isa, kernel)
    assert "flat_" not in body
    assert re.search(r"global_store_dword\b", body)


def test_stream_kernel_rows_are_stored_non_temporally_and_nothing_spills(isa):
    body = _body(isa, "sample_counts_stream_kernelILb1ELb1E")
    assert re.search(r"buffer_store_dwordx4 .* offen nt\b", body)       # rows: the row in soffset, the lane's 16 bytes in voffset
    assert _meta(isa, "sample_counts_stream_kernelILb1ELb1E", "private_segment_fixed_size") == 0
    assert _meta(isa, "sample_counts_stream_kernelILb1ELb1E", "vgpr_spill_count") == 0
    # five blocks of 256 threads per CU (the fifth is worth 11 %: profiles/r04_ablation.txt): 96 VGPRs and
    # 32 000 B of LDS each at most
    assert _meta(isa, "sample_counts_stream_kernelILb1ELb1E", "vgpr_count") <= 96
    # (gfx950 hands out LDS in 1 280-byte granules: 25 of them per block is the most that five blocks leave)
    assert _meta(isa, "sample_counts_stream_kernelILb1ELb1E", "group_segment_fixed_size") <= 32000
    # packed binary32 instructions take two issue slots and cost moves to pair their operands (-fno-slp-vectorize)
    assert "v_pk_mul_f32" not in body and "v_pk_add_f32" not in body and "v_pk_fma_f32" not in body


def test_second_kernel_has_no_scratch(isa):
    assert _meta(isa, "sample_counts_heavy_kernel", "private_segment_fixed_size") == 0


_SGPR = r"(?:s\[\d+:\d+\]|s\d+|vcc(?:_lo|_hi)?|exec(?:_lo|_hi)?)"


def _sgpr_set(tok):
    """The 32-bit scalar registers a token like s[4:5], s7, vcc, exec_lo names."""
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return {"s%d" % i for i in range(int(m.group(1)), int(m.group(2)) + 1)}
    if tok in ("vcc", "exec"):
        return {tok + "_lo", tok + "_hi"}
    return {tok}


@pytest.mark.parametrize("kernel", ["sample_counts_stream_kernelILb1ELb1E", "sample_counts_heavy_kernel"])
def test_no_hand_written_valu_reads_a_mask_straight_behind_the_valu_that_wrote_it(isa, kernel):
    """gfx950: a VALU instruction that reads an SGPR (vcc, exec) as data within two instructions of the VALU
    instruction that wrote it sees the old value (tools/cmpx_probe.hip).  The compiler separates such pairs in its
    own code but does not look into inline asm: every VALU instruction of an asm statement is checked here against
    the two instructions in front of it."""
    lines = [ln.strip() for ln in _body(isa, kernel).splitlines()]
    code = []          # (text, inside_asm)
    inside = False
    for ln in lines:
        if ln.startswith(";;#ASMSTART"):
            inside = True
        elif ln.startswith(";;#ASMEND"):
            inside = False
        elif ln and not ln.startswith((";", ".")) and not ln.endswith(":"):
            code.append((ln.split(";")[0].strip(), inside))
    checked = 0
    for i, (ins, in_asm) in enumerate(code):
        if not in_asm or not ins.startswith("v_"):
            continue
        ops = [o.strip() for o in ins.split(None, 1)[1].split(",")] if " " in ins else []
        reads = set()
        for o in ops[1:]:
            for tok in re.findall(_SGPR, o):
                reads |= _sgpr_set(tok)
        if not reads:
            continue
        checked += 1
        for prev, _ in code[max(0, i - 2):i]:
            if not prev.startswith("v_"):
                continue
            pops = [o.strip() for o in prev.split(None, 1)[1].split(",")]
            written = set()
            if re.fullmatch(_SGPR, pops[0]):
                written |= _sgpr_set(pops[0])
            if prev.startswith("v_cmpx"):
                written |= {"exec_lo", "exec_hi"}
            if re.match(r"v_(add|sub|subrev)_co_|v_addc_co|v_subb_co|v_mad_[ui]64", prev) and len(pops) > 1 and re.fullmatch(_SGPR, pops[1]):
                written |= _sgpr_set(pops[1])
            assert not (written & reads), "%s reads %s written by %s" % (ins, sorted(written & reads), prev)
    assert checked > 0 or "heavy" in kernel          # (K3h has no such asm statement today)
