#!/bin/bash
# Register / LDS use of every kernel of the library, from the code object's own metadata (hipcc -save-temps).
# usage: tools/kernel_resources.sh > profiles/rNN_kernel_resources.txt
R="$(cd "$(dirname "$0")/.." && pwd)"; T=$(mktemp -d); cd $T
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp -fPIC -shared -fvisibility=hidden -save-temps \
    -o $T/lib.so $R/prosstt_amd/csrc/prosstt_amd.hip > /dev/null 2>&1
python3 - <<PY
import re, subprocess
txt = open("$T/prosstt_amd-hip-amdgcn-amd-amdhsa-gfx950.s").read()
print("# kernel_source_sha: %s   (fields of the .amdgpu_metadata note of the gfx950 code object)" % subprocess.check_output(
    ["python3", "-c", "import sys; sys.path.insert(0, '$R'); import bench; print(bench.kernel_source_sha())"]).decode().strip())
for blk in re.split(r"\n  - \.agpr_count", txt)[1:]:
    get = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk)
    name = get("name").group(1)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    print("%-52s vgpr %3s  sgpr %3s  lds %6s B  scratch %3s B  vgpr spills %s  sgpr spills %s  max threads/block %s" % (
        dem[-52:], get("vgpr_count").group(1), get("sgpr_count").group(1), get("group_segment_fixed_size").group(1),
        get("private_segment_fixed_size").group(1), get("vgpr_spill_count").group(1), get("sgpr_spill_count").group(1),
        get("max_flat_workgroup_size").group(1)))
PY
rm -rf $T
