"""Static check of the shipped code objects for the packed-FP32 operand form that gives wrong results on gfx950 under GPU sharing.

Finding (round 4; scripts/probe/rowdot_variants.hip, build_rowdot_patched.sh, profiles/r04_rowdot_*):
`v_pk_fma_f32 D, A, B, C op_sel:[0,1,0]` — the LOW result taking the HIGH dword of src1 — occasionally leaves the low dword of D
unwritten in lanes 48-63 (the value of an in-place accumulator chain with exactly that FMA missing) when waves of OTHER processes
share the CU.  Wait states before or after the instruction do not help; the same instruction with src0 and src1 exchanged
(`op_sel:[1,0,0]`), or split into two v_fma_f32, never failed (0 of 27 process-runs against 31 of 42).  The compiler picks the form
by itself (SLP-vectorised fmaf chains with a broadcast operand), so the check runs on the binaries, not the sources.

Which form the compiler emits depends on register allocation (the form came back once from an unrelated refactoring), so since the
end of round 4 the library is built with `-Xclang -target-feature -Xclang -packed-fp32-ops` (video-filler_amd/build.py) and holds NO
packed FP32 arithmetic at all.  The check mirrors that intent: by default it flags EVERY v_pk_fma_f32 / v_pk_mul_f32 /
v_pk_add_f32 — if the feature flag were ever dropped silently, the harmless src0-select form would reappear first and the check
fails on it — and reports separately (`scan(...)[0]`) the instructions of the dangerous form, the LOW result selecting the HIGH dword
of src1 or src2.

    python scripts/check_pk_opsel.py [--opsel-only] [library.so ...]        exit status 1 when anything is flagged
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
PK_ANY = re.compile(r"\bv_pk_(?:fma|mul|add)_f32\b")


def code_objects(so, workdir):
    """The gfx950 code objects bundled in a host library, as files in workdir."""
    local = os.path.join(workdir, os.path.basename(so))
    shutil.copy(so, local)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    return sorted(os.path.join(workdir, f) for f in os.listdir(workdir) if "amdgcn" in f)


def scan(so, any_packed=None):
    """[(kernel symbol, instruction text)] for every instruction of the dangerous form; also the number of kernels seen.  With a
    list in `any_packed`, every packed FP32 arithmetic instruction (whatever its op_sel) is appended to it as well."""
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
                if any_packed is not None and PK_ANY.search(line):
                    any_packed.append((sym, line.split("//")[0].strip()))
                m = PK.search(line)
                if m:
                    sel = [int(b) for b in m.group(2).split(",")]
                    if any(sel[1:]):
                        flagged.append((sym, line.split("//")[0].strip()))
    return flagged, len(kernels)


def main(argv):
    opsel_only = "--opsel-only" in argv
    libs = [a for a in argv if not a.startswith("--")] or [os.path.join(ROOT, "video-filler_amd", "lib", "libvf_hip.so")]
    bad = 0
    for so in libs:
        packed = []
        flagged, n = scan(so, packed)
        print("%s: %d kernels, %d packed FP32 arithmetic instructions, %d of them with a high-dword src1 / src2 select"
              % (os.path.relpath(so, ROOT), n, len(packed), len(flagged)))
        if not opsel_only:
            flagged = packed
        per = {}
        for sym, ins in flagged:
            per.setdefault(sym, []).append(ins)
        for sym, ins in per.items():
            print("  %s: %d, e.g. %s" % (sym, len(ins), ins[0]))
        bad += len(flagged)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
