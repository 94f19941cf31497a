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
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
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


@pytest.mark.parametrize("kernel", ["sample_counts_stream_kernelILb1", "sample_counts_stream_kernelILb0"])
def test_stream_kernel_memory_operations_are_global(isa, kernel):
    # a late 4-byte store must land after its row's 16-byte store: global_* operations of a wave are performed
    # in issue order, flat_* are not (k3_stream.h, flush_late)
    body = _body(isa, kernel)
    assert "flat_" not in body
    assert re.search(r"global_store_dword\b", body)


def test_stream_kernel_rows_are_stored_non_temporally_and_nothing_spills(isa):
    body = _body(isa, "sample_counts_stream_kernelILb1")
    assert re.search(r"global_store_dwordx4 .* nt\b", body)
    assert _meta(isa, "sample_counts_stream_kernelILb1", "private_segment_fixed_size") == 0
    assert _meta(isa, "sample_counts_stream_kernelILb1", "vgpr_spill_count") == 0
    # four blocks of 256 threads per CU: 128 VGPRs and 40 KB of LDS each at most
    assert _meta(isa, "sample_counts_stream_kernelILb1", "vgpr_count") <= 128
    assert _meta(isa, "sample_counts_stream_kernelILb1", "group_segment_fixed_size") <= 40 * 1024


def test_second_kernel_has_no_scratch(isa):
    assert _meta(isa, "sample_counts_heavy_kernel", "private_segment_fixed_size") == 0
