#!/usr/bin/env python3
"""Per-kernel resources of the gfx950 code objects in .o / .so files: VGPRs, AGPRs, SGPRs, spilled VGPRs, scratch bytes, LDS bytes.

    python tools/kres.py sepreformer_amd/_native/sepr_gcfn_fused.o [name-regex]
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = os.environ.get("LLVM_OBJDUMP") or shutil.which("llvm-objdump") or "/opt/rocm/lib/llvm/bin/llvm-objdump"
READELF = os.path.join(os.path.dirname(OBJDUMP), "llvm-readelf")
FILT = shutil.which("c++filt") or "c++filt"


def resources(path, pattern="."):
    """[{name, vgpr, agpr, sgpr, spill, scratch, lds}] of the kernels in `path` whose demangled name matches `pattern`."""
    pat, out = re.compile(pattern), []
    with tempfile.TemporaryDirectory() as tmp:
        base = os.path.join(tmp, os.path.basename(path))
        shutil.copy(path, base)
        subprocess.run([OBJDUMP, "--offloading", base], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
        for o in sorted(f for f in (os.path.join(tmp, n) for n in os.listdir(tmp)) if f.startswith(base + ".") and "amdgcn" in f):
            notes = subprocess.run([READELF, "--notes", o], stdout=subprocess.PIPE, text=True, check=True).stdout
            for blk in notes.split("- .agpr_count:")[1:]:
                g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]   # noqa: E731
                name = g("name")
                dem = subprocess.run([FILT, name], stdout=subprocess.PIPE, text=True).stdout.strip()
                if not pat.search(dem):
                    continue
                out.append({"name": dem, "vgpr": g("vgpr_count"), "agpr": blk.split("\n", 1)[0].strip(), "sgpr": g("sgpr_count"), "spill": g("vgpr_spill_count"),
                            "scratch": g("private_segment_fixed_size"), "lds": g("group_segment_fixed_size")})
    return out


def main(argv):
    for r in resources(argv[0], argv[1] if len(argv) > 1 else "."):
        print(f"vgpr {r['vgpr']:>4} agpr {r['agpr']:>3} sgpr {r['sgpr']:>4} spill {r['spill']:>4} scratch {r['scratch']:>5} lds {r['lds']:>6}  {r['name'][:150]}")


if __name__ == "__main__":
    main(sys.argv[1:])
