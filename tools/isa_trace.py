#!/usr/bin/env python3
"""Compressed instruction trace of one kernel of a .o / .so: waits, barriers, LDS-DMA, LDS / global accesses, MFMAs, branches - the view that shows
where hipcc put its s_waitcnt vmcnt(0) in a software-pipelined loop.

    python tools/isa_trace.py sepreformer_amd/_native/sepr_gcfn_fused.o 'gcfn_fused3_kernel<128, 1, 6, 0, false, false, 3>'
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = os.environ.get("LLVM_OBJDUMP") or shutil.which("llvm-objdump") or "/opt/rocm/lib/llvm/bin/llvm-objdump"
PAT = re.compile(r"\b(s_waitcnt [a-z]+cnt\(\d+\)(?: [a-z]+cnt\(\d+\))*|s_barrier|global_load_lds_dwordx4|ds_read\w*|ds_write\w*|v_mfma\w+|global_load_dword\w*|"
                 r"global_store_dword\w*|scratch_\w+|s_cbranch\w+)")


def main(argv):
    path, want = argv[0], argv[1]
    with tempfile.TemporaryDirectory() as tmp:
        base = os.path.join(tmp, os.path.basename(path))
        shutil.copy(path, base)
        subprocess.run([OBJDUMP, "--offloading", base], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
        for o in sorted(f for f in (os.path.join(tmp, n) for n in os.listdir(tmp)) if f.startswith(base + ".") and "amdgcn" in f):
            dis = subprocess.run([OBJDUMP, "-d", "--demangle", o], stdout=subprocess.PIPE, text=True, check=True).stdout
            cur, res = None, []
            for line in dis.splitlines():
                if line.endswith(">:"):
                    cur = line.split("<", 1)[1][:-2]
                    continue
                if cur is None or want not in cur:
                    continue
                m = PAT.search(line)
                if not m:
                    continue
                t = m.group(1)
                if res and res[-1][0] == t:
                    res[-1][1] += 1
                else:
                    res.append([t, 1])
            print(" ".join(f"{t}x{n}" if n > 1 else t for t, n in res))


if __name__ == "__main__":
    main(sys.argv[1:])
