#!/usr/bin/env python3
"""Per-kernel register / spill / scratch table of the gfx950 code objects, straight from the compiler
(`-Rpass-analysis=kernel-resource-usage`, device-only compile of one csrc/*.hip with the Makefile's flags).
    python tools/kernel_resources.py ba_kernels [filter]   ->  stdout (commit under profiles/ when it backs a claim)
The same numbers sit in the code object's notes (llvm-readelf --notes: .vgpr_count, .vgpr_spill_count,
.private_segment_fixed_size)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "orb-slam3-mac_amd")
EXTRA = os.environ.get("EXTRA", "").split()
FILEFLAGS = {"match_kernels": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "orb_kernels": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
             "ba_kernels": ["-mllvm", "-simplifycfg-sink-common=false"]}        # as orb-slam3-mac_amd/Makefile


def main():
    name = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off",
           "-fhip-fp32-correctly-rounded-divide-sqrt", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage"] + FILEFLAGS.get(name, []) + EXTRA + \
          ["-c", "-o", "/dev/null", os.path.join(PKG, "csrc", name + ".hip")]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    rows, cur = [], None
    for line in r.stdout.splitlines():
        m = re.search(r"remark:\s+(.*?)\s+\[-Rpass-analysis", line)
        if not m:
            continue
        t = m.group(1)
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    try:
        dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + [r_["name"] for r_ in rows], stdout=subprocess.PIPE, text=True).stdout.splitlines()
    except OSError:
        dem = [r_["name"] for r_ in rows]
    print("%-58s %6s %6s %6s %9s %9s %8s %5s" % ("kernel", "VGPRs", "AGPRs", "SGPRs", "VGPRspill", "SGPRspill", "scratchB", "occ"))
    for r_, d in zip(rows, dem):
        d = re.sub(r"\(.*$", "", d).replace("void ", "")
        if flt and flt not in d:
            continue
        print("%-58s %6s %6s %6s %9s %9s %8s %5s" % (d[:58], r_.get("VGPRs", "?"), r_.get("AGPRs", "?"), r_.get("TotalSGPRs", "?"), r_.get("VGPRs Spill", "?"),
                                                    r_.get("SGPRs Spill", "?"), r_.get("ScratchSize [bytes/lane]", "?"), r_.get("Occupancy [waves/SIMD]", "?")))


if __name__ == "__main__":
    main()
