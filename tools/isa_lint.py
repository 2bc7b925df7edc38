#!/usr/bin/env python3
"""ISA lint of the gfx950 code objects (round 5; DESIGN.md section 10, tools/probe/pk_opsel.hip).

On gfx950 a packed-f32 VALU instruction (VOP3P v_pk_{add,mul,fma,...}_f32) whose op_sel modifier makes the LOW result half read the HIGH
dword of src1 (or src2) returns wrong low halves in lanes 16-31 / 48-63 while another wave of the SIMD executes bf16 MFMAs.  hipcc's SLP
vectoriser emits exactly that form for "splat of an odd register" and for horizontal adds - it caused the wrong weight gradients of round
3 and the nondeterministic attention of round 4.  This script disassembles every device code object of the given .o / .so files and fails
(exit 1) if such an instruction exists anywhere, so a source change that makes the compiler produce one stops the build.

    python tools/isa_lint.py sepreformer_amd/_native/*.o          (the csrc Makefile runs it on every object it compiles)

Allowed: no modifier, op_sel_hi in any combination, op_sel on src0 only (measured clean by the probe)."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = os.environ.get("LLVM_OBJDUMP") or shutil.which("llvm-objdump") or "/opt/rocm/lib/llvm/bin/llvm-objdump"
PK_F32 = re.compile(r"\bv_pk_[a-z0-9]+_f32\b")
BAD_SEL = re.compile(r"\bop_sel:\[[01],1")            # src1 bit set (covers [x,1] and [x,1,y])
BAD_SEL2 = re.compile(r"\bop_sel:\[[01],[01],1")      # src2 bit set


HOST_ONLY = ("sepr_api.o", "sepr_train_api.o")       # objects WITHOUT kernels: for anything else "no device code object found" is a failure of the lint itself


def device_objects(path, tmp):
    """Extract the gfx950 code objects bundled into a host object / shared library."""
    base = os.path.join(tmp, os.path.basename(path))
    shutil.copy(path, base)
    p = subprocess.run([OBJDUMP, "--offloading", base], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, check=False)
    objs = sorted(f for f in (os.path.join(tmp, n) for n in os.listdir(tmp)) if f.startswith(base + ".") and "amdgcn" in f)
    if not objs and os.path.basename(path) not in HOST_ONLY:
        # fail CLOSED (round 6): an objdump without --offloading support, or one that names its outputs differently, must not pass every object
        raise SystemExit(f"isa_lint: {path}: no gfx950 code object extracted (llvm-objdump --offloading rc={p.returncode}: {p.stderr.strip()[:200]}) - "
                         "the lint cannot see the kernels; set LLVM_OBJDUMP or list the object in HOST_ONLY if it has none")
    return objs


def lint(path):
    bad, n_pk = [], 0
    with tempfile.TemporaryDirectory() as tmp:
        objs = device_objects(path, tmp)
        if not objs:            # a host-only object (HOST_ONLY: sepr_api.o has no kernels); the CPU test lints the linked .so and checks the instruction count
            return 0, []
        for o in objs:
            dis = subprocess.run([OBJDUMP, "-d", o], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, check=True).stdout
            kernel = "?"
            for line in dis.splitlines():
                if line.endswith(">:"):
                    kernel = line.split("<", 1)[-1][:-2]
                    continue
                if PK_F32.search(line):
                    n_pk += 1
                    if BAD_SEL.search(line) or BAD_SEL2.search(line):
                        bad.append((kernel, line.split("//")[0].strip()))
    return n_pk, bad


def main(argv):
    if not argv:
        raise SystemExit(__doc__)
    rc = 0
    for path in argv:
        n_pk, bad = lint(path)
        if bad:
            rc = 1
            print(f"isa_lint: {path}: {len(bad)} packed-f32 instruction(s) with a low-half select on src1 / src2 (gfx950 fault, see tools/probe/pk_opsel.hip):", file=sys.stderr)
            seen = {}
            for k, ins in bad:
                seen.setdefault(k, []).append(ins)
            for k, v in seen.items():
                print(f"  {k}: {len(v)} x, e.g. {v[0]}", file=sys.stderr)
        elif os.environ.get("ISA_LINT_VERBOSE"):
            print(f"isa_lint: {path}: ok ({n_pk} packed-f32 instructions)")
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
