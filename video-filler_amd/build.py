"""Build the gfx950 HIP library in-tree: video-filler_amd/lib/libvf_hip.so (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = ["vf_core.hip", "vf_bn.hip", "vf_conv.hip", "vf_conv_generic.hip", "vf_pipeline.hip", "vf_pgemm.hip", "vf_conv_thin.hip", "vf_comm.hip", "vf_wgrad_small.hip", "vf_smallm.hip", "vf_trace.hip", "vf_net.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -packed-fp32-ops: no v_pk_{fma,mul,add}_f32 in the device code.  One operand form of those instructions (`op_sel` taking the HIGH
# dword of src1 for the low result) gives wrong results on gfx950 when other processes share the CU, and which form
# the compiler picks depends on register allocation (round 4: a refactoring of the fused Adam epilogue brought it back at once).
# Same-box A/B of the whole library with and without the packed forms: configs[1] +0.3 %, configs[2] / [4] within noise — they issue
# at half rate on this part and every kernel that used them is memory-bound.  scripts/check_pk_opsel.py is the guard on the binary:
# it fails on ANY packed FP32 arithmetic, so a silently dropped flag cannot pass.  The flag cannot be scoped to the device pass
# (`-Xarch_device` refuses options that take an argument), so the host half of each compile prints "'-packed-fp32-ops' is not a
# recognized feature for this target (ignoring feature)"; build() drops those lines from its verbose output.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


def lib_path():
    return os.path.join(HERE, "lib", "libvf_hip.so")


def build(force=False, verbose=False):
    csrc = os.path.join(HERE, "csrc")
    objdir = os.path.join(HERE, "lib", "obj")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(csrc, "vf_common.h"), os.path.join(HERE, "..", "include", "vf_hip.h")]
    hdr_m = max(os.path.getmtime(h) for h in hdrs)
    objs, rebuilt = [], False
    procs = []
    for s in SRC:
        src = os.path.join(csrc, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_m):
            cmd = [HIPCC, *FLAGS, "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
            rebuilt = True
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed on " + s)
        if verbose and out:
            print("\n".join(ln for ln in out.decode().split("\n") if "-packed-fp32-ops' is not a recognized feature" not in ln))
    so = lib_path()
    if rebuilt or not os.path.exists(so):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, *objs, "-ldl"])
    return so


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
