"""Register / LDS / scratch budget of every kernel in the built library, read from the code objects' metadata
(llvm-objdump --offloading + llvm-readelf --notes; no GPU needed).  Occupancy on gfx950 follows from the VGPR count
(512 per SIMD lane: <= 64 -> 8 wavefronts per SIMD, <= 128 -> 4, <= 168 -> 3), so a kernel that silently crosses a
boundary loses a quarter of its latency hiding -- tests/test_kernel_resources.py pins the budgets of the hot kernels.

usage: python tools/kernel_resources.py [out.json]
"""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
LLVM = "/opt/rocm/lib/llvm/bin"


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.strip().split("\n")


def kernel_resources(lib=None):
    lib = lib or os.path.join(ROOT, "graspqp_amd", "lib", "libgraspqp_hip.so")
    tmp = tempfile.mkdtemp(prefix="gq_co_")
    try:
        shutil.copy(lib, os.path.join(tmp, "lib.so"))
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=tmp, capture_output=True, check=True)
        rows = {}
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f], cwd=tmp, capture_output=True, text=True).stdout
            for blk in notes.split("  - .agpr_count:")[1:]:
                g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk)
                name = g("name").group(1)
                rows[name] = {"vgpr": int(g("vgpr_count").group(1)), "agpr": int(blk.split()[0]), "sgpr": int(g("sgpr_count").group(1)),
                              "lds_static": int(g("group_segment_fixed_size").group(1)),
                              "scratch": int(g("private_segment_fixed_size").group(1)),
                              "vgpr_spills": int(g("vgpr_spill_count").group(1)),
                              "max_threads": int(g("max_flat_workgroup_size").group(1))}
        names = list(rows)
        out = {}
        for n, d in zip(names, demangle(names)):
            r = rows[n]
            regs = r["vgpr"] + r["agpr"]
            r["waves_per_simd"] = 8 if regs <= 64 else 512 // ((regs + 7) // 8 * 8)
            out[re.sub(r"\(.*", "", d.replace("void ", ""))] = r
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    res = kernel_resources()
    if len(sys.argv) > 1:
        json.dump(res, open(sys.argv[1], "w"), indent=1, sort_keys=True)
    for k in sorted(res, key=lambda k: -res[k]["vgpr"]):
        r = res[k]
        print(f"{k[:64]:64s} vgpr {r['vgpr']:4d} waves/SIMD {r['waves_per_simd']} lds {r['lds_static']:6d} scratch {r['scratch']}")
