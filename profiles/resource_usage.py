#!/usr/bin/env python3
"""profiles/<prefix>_kernel_resource_usage.txt: registers, scratch, spills and occupancy of every kernel instantiation,
from hipcc's own remarks (no GPU needed):  python profiles/resource_usage.py r02"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prefix = sys.argv[1] if len(sys.argv) > 1 else "r02"
# the library's translation units (founder-sequences_amd/build.py SRCS)
units = ["fseq_api.hip", "fseq_api_join.hip", "fseq_reduced.hip", "fseq_kernelsets.hip", "fseq_kernelsets_stream.hip"]
txt = ""
for u in units:
    src = os.path.join(ROOT, "founder-sequences_amd", "csrc", u)
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-Rpass-analysis=kernel-resource-usage",
                        "-o", "/dev/null", src], capture_output=True, text=True)
    txt += r.stderr
rows = []
for b in re.split(r"(?=remark: Function Name:)", txt):
    m = re.match(r"remark: Function Name: (\S+)", b)
    if not m:
        continue

    def g(k):
        mm = re.search(re.escape(k) + r": (\d+)", b)
        return mm.group(1) if mm else "-"
    dem = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"^void ", "", re.sub(r"\(.*", "", dem))
    rows.append([dem, g("VGPRs"), g("AGPRs"), g("TotalSGPRs"), g("ScratchSize [bytes/lane]"), g("Occupancy [waves/SIMD]"),
                 g("VGPRs Spill"), g("SGPRs Spill"), g("LDS Size [bytes/block]")])
out = ["# hipcc --offload-arch=gfx950 -O3 -std=c++17 -Rpass-analysis=kernel-resource-usage founder-sequences_amd/csrc/{fseq_api,fseq_api_join}.hip",
       "# kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | occupancy waves/SIMD | VGPR spill | SGPR spill | static LDS B (dynamic LDS not included)"]
out += [" | ".join(r_) for r_ in sorted(rows)]
path = os.path.join(ROOT, "profiles", "%s_kernel_resource_usage.txt" % prefix)
open(path, "w").write("\n".join(out) + "\n")
print(path, len(rows), "kernels")
