"""Static check of the shipped code objects for the packed-FP32 operand form that gives wrong results on gfx950 under GPU sharing.

Finding (round 4; scripts/probe/rowdot_variants.hip, build_rowdot_patched.sh, profiles/r04_rowdot_*):
`v_pk_fma_f32 D, A, B, C op_sel:[0,1,0]` — the LOW result taking the HIGH dword of src1 — occasionally leaves the low dword of D
unwritten in lanes 48-63 (the value of an in-place accumulator chain with exactly that FMA missing) when waves of OTHER processes
share the CU.  Wait states before or after the instruction do not help; the same instruction with src0 and src1 exchanged
(`op_sel:[1,0,0]`), or split into two v_fma_f32, never failed (0 of 27 process-runs against 31 of 42).  The compiler picks the form
by itself (SLP-vectorised fmaf chains with a broadcast operand), so the check runs on the binaries, not the sources.

Flags every packed FP32 arithmetic instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) whose op_sel selects the high dword
of src1 or src2 for the low result.  src0 selection ([1,0,0]) is the form the shipped kernels use and is not flagged.

    python scripts/check_pk_opsel.py [library.so ...]        exit status 1 when anything is flagged
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = os.environ.get("VF_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PK = re.compile(r"\b(v_pk_(?:fma|mul|add)_f32)\b.*?\bop_sel:\[([01](?:,[01])*)\]")


def code_objects(so, workdir):
    """The gfx950 code objects bundled in a host library, as files in workdir."""
    local = os.path.join(workdir, os.path.basename(so))
    shutil.copy(so, local)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    return sorted(os.path.join(workdir, f) for f in os.listdir(workdir) if "amdgcn" in f)


def scan(so):
    """[(kernel symbol, instruction text)] for every flagged instruction; also the number of kernels seen."""
    flagged, kernels = [], set()
    with tempfile.TemporaryDirectory() as wd:
        for co in code_objects(so, wd):
            out = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], check=True, capture_output=True,
                                 text=True).stdout
            sym = None
            for line in out.split("\n"):
                m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
                if m:
                    sym = m.group(1)
                    if not sym.startswith("L") and "$local" not in sym:
                        kernels.add(sym)
                    continue
                m = PK.search(line)
                if m:
                    sel = [int(b) for b in m.group(2).split(",")]
                    if any(sel[1:]):
                        flagged.append((sym, line.split("//")[0].strip()))
    return flagged, len(kernels)


def main(argv):
    libs = argv or [os.path.join(ROOT, "video-filler_amd", "lib", "libvf_hip.so")]
    bad = 0
    for so in libs:
        flagged, n = scan(so)
        print("%s: %d kernels, %d flagged instructions" % (os.path.relpath(so, ROOT), n, len(flagged)))
        per = {}
        for sym, ins in flagged:
            per.setdefault(sym, []).append(ins)
        for sym, ins in per.items():
            print("  %s: %d, e.g. %s" % (sym, len(ins), ins[0]))
        bad += len(flagged)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
