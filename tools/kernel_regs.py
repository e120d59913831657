#!/usr/bin/env python3
"""Register / scratch use of the kernels in the SHIPPED library (the code objects inside vbnn_amd/lib/libvbnn_hip.so):
    python tools/kernel_regs.py [substring ...]
prints, per kernel whose (demangled-ish) name contains every substring: VGPRs, AGPRs, SGPRs, spilled VGPRs, scratch bytes, LDS.
A kernel that spills in its main loop is a regression whatever the tests say; this is the check before a GPU run."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(lib=None):
    lib = lib or os.path.join(ROOT, "vbnn_amd", "lib", "libvbnn_hip.so")
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], cwd=tmp, capture_output=True, check=True)
        for co in sorted(f for f in os.listdir(tmp) if "amdgcn" in f and "gfx950" in f):
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, co)], capture_output=True, text=True).stdout
            for blk in notes.split("- .agpr_count:")[1:]:
                def g(key, blk=blk):
                    m = re.search(r"\." + key + r":\s+(\S+)", blk)
                    return m.group(1) if m else "?"
                agpr = blk.split()[0]
                out.append(dict(name=g("name"), vgpr=g("vgpr_count"), agpr=agpr, sgpr=g("sgpr_count"), spill=g("vgpr_spill_count"),
                                scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size")))
    return out


def main():
    pats = sys.argv[1:]
    for k in kernels():
        name = k["name"]
        filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
        if filt:
            name = subprocess.run([filt, name], capture_output=True, text=True).stdout.strip() or name
        if all(p in name for p in pats):
            print(f"vgpr {k['vgpr']:>3} agpr {k['agpr']:>3} sgpr {k['sgpr']:>3} spill {k['spill']:>3} scratch {k['scratch']:>5} lds {k['lds']:>6}  {name[:150]}")


if __name__ == "__main__":
    main()
