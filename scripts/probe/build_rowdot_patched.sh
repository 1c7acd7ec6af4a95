#!/bin/bash
# Builds variants of the PRE-FIX row-dot kernel (scripts/probe/rowdot_variants.hip, RED=0 LOADS=0) from its own assembly with wait
# states inserted around the packed FMAs that select the HIGH dword of src1 (v_pk_fma_f32 ... op_sel:[0,1,0]) — the one instruction
# form every failing variant has and no passing one.  Same kernel, same registers, same schedule; only s_nop added.
#   nopafter0 : s_nop 0 after each such instruction (its consumer is then >= 2 wait states away)
#   nopafter3 : s_nop 3 after
#   nopbefore3: s_nop 3 before (its producers are then >= 4 wait states away)
#   scalarfma : each such instruction replaced by the two v_fma_f32 it stands for (same arithmetic per lane)
#   swap      : src0 and src1 exchanged, op_sel:[1,0,0] (the product commutes; the HIGH-dword select moves to src0)
#   identity  : the unpatched assembly through the same pipeline (control)
set -e
LL=/opt/rocm/lib/llvm/bin; HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
root=$PWD; obj=$root/video-filler_amd/lib/obj; alt=$root/video-filler_amd/lib/alt; w=/tmp/rdpatch
mkdir -p $alt $w
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -I$root/video-filler_amd/csrc -DRED=0 -DLOADS=0 -DDIAG=0"
SRC=scripts/probe/rowdot_variants.hip
$HIPCC $F -S --cuda-device-only $SRC -o $w/dev.s 2>/dev/null
objs=$(ls $obj/*.o | grep -v vf_smallm.o)
mk() {  # tag sed-expression | py:<mode>
  case "$2" in
    py:*) python3 - "$w/dev.s" "$w/dev_$1.s" "${2#py:}" <<'PY'
import re, sys
src, dst, mode = sys.argv[1:4]
pat = re.compile(r'^\s*v_pk_fma_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\] op_sel:\[0,1,0\]\s*$')
out = []
for line in open(src):
    m = pat.match(line.rstrip('\n'))
    if not m:
        out.append(line)
        continue
    d0, d1, a0, a1, b0, b1, c0, c1 = map(int, m.groups())
    if mode == "scalarfma":
        assert d0 not in (a1, b1, c1) or d0 == c0, line          # the first write must not clobber a source of the second
        out.append("\tv_fma_f32 v%d, v%d, v%d, v%d\n" % (d0, a0, b1, c0))
        out.append("\tv_fma_f32 v%d, v%d, v%d, v%d\n" % (d1, a1, b1, c1))
    else:
        out.append("\tv_pk_fma_f32 v[%d:%d], v[%d:%d], v[%d:%d], v[%d:%d] op_sel:[1,0,0]\n" % (d0, d1, b0, b1, a0, a1, c0, c1))
open(dst, "w").writelines(out)
PY
    ;;
    *) sed "$2" $w/dev.s > $w/dev_$1.s ;;
  esac
  $LL/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $w/dev_$1.s -o $w/dev_$1.o
  $LL/ld.lld -shared $w/dev_$1.o -o $w/dev_$1.hsaco
  $LL/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$w/dev_$1.hsaco -output=$w/dev_$1.hipfb
  $HIPCC $F -c $SRC --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $w/dev_$1.hipfb -o $w/rd_$1.o 2>/dev/null
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o $alt/libvf_hip_rd$1.so $objs $w/rd_$1.o -ldl
  echo "$1: s_nop $(grep -c s_nop $w/dev_$1.s), op_sel:[0,1,0] $(grep -c 'op_sel:\[0,1,0\]' $w/dev_$1.s), op_sel:[1,0,0] $(grep -c 'op_sel:\[1,0,0\]' $w/dev_$1.s), v_fma_f32 $(grep -c 'v_fma_f32' $w/dev_$1.s)"
}
mk identity 's/^$//'
mk scalarfma py:scalarfma
mk swap py:swap
mk nopafter0 's/^\(\s*v_pk_fma_f32 .*op_sel:\[0,1,0\].*\)$/\1\n\ts_nop 0/'
mk nopafter3 's/^\(\s*v_pk_fma_f32 .*op_sel:\[0,1,0\].*\)$/\1\n\ts_nop 3/'
mk nopbefore3 's/^\(\s*v_pk_fma_f32 .*op_sel:\[0,1,0\].*\)$/\ts_nop 3\n\1/'
