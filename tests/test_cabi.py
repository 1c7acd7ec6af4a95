"""CPU-side checks of the drop-in boundary: the HIP shared library loads without a GPU, exports every symbol that
include/vf_hip.h declares, the ctypes table mirrors the header one to one, and the product fails loudly — no CPU
fallback — when no GPU is present."""
import ctypes as C
import os
import re

import pytest
import torch

import video_filler_amd  # noqa: F401
from video_filler_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_is_built_in_tree():
    assert os.path.exists(_lib.lib_path()), "run `python -c 'import __graft_entry__ as g; g.build()'` first"
    assert os.path.realpath(_lib.lib_path()).startswith(os.path.realpath(ROOT))


def test_every_header_symbol_is_exported_and_bound():
    lib = _lib.load()
    syms = _lib.header_symbols()
    assert len(syms) >= 45
    for s in syms:
        assert hasattr(lib, s), "libvf_hip.so does not export %s" % s
        assert s in _lib.SIGNATURES, "no ctypes signature for %s" % s
    assert sorted(_lib.SIGNATURES) == syms


def test_signature_arity_matches_header():
    with open(os.path.join(ROOT, "include", "vf_hip.h")) as fh:
        text = re.sub(r"/\*.*?\*/", "", fh.read(), flags=re.S)
    for name, (res, args) in _lib.SIGNATURES.items():
        m = re.search(r"\b%s\s*\(([^;]*?)\)\s*;" % name, text, flags=re.S)
        assert m, name
        params = m.group(1).strip()
        n = 0 if params in ("", "void") else len(params.split(","))
        assert n == len(args), "%s: header has %d parameters, binding %d" % (name, n, len(args))


def test_gfx950_code_object_present():
    with open(_lib.lib_path(), "rb") as fh:
        blob = fh.read()
    assert b"gfx950" in blob


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_cpu_fallback():
    lib = _lib.load()
    ctx = C.c_void_p()
    rc = lib.vf_ctx_create(C.byref(ctx), 0, None)
    assert rc != 0
    assert len(lib.vf_last_error()) > 0
    from video_filler_amd.backend import HipBackend
    with pytest.raises(RuntimeError):
        HipBackend()


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "video-filler_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".lua")):
                with open(os.path.join(dp, f)) as fh:
                    src = fh.read()
                assert "import oracle" not in src and "from oracle" not in src and "vf_oracle" not in src, os.path.join(dp, f)


def test_lua_binding_declares_the_header_faithfully():
    """video-filler_amd/lua/hipnn.lua cannot be executed here (no LuaJIT/Torch7); at least its ffi.cdef must declare
    only functions the header declares, with the same number of parameters, and every C.vf_* call in the file must be
    declared in that cdef."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lua = open(os.path.join(root, "video-filler_amd", "lua", "hipnn.lua")).read()
    cdef = re.search(r"ffi\.cdef\[\[(.*?)\]\]", lua, flags=re.S).group(1)
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", "vf_hip.h")).read(), flags=re.S)

    def protos(text):
        out = {}
        for name, args in re.findall(r"\b(vf_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
            args = args.strip()
            out[name] = 0 if args in ("", "void") else args.count(",") + 1
        return out
    h, l = protos(hdr), protos(cdef)
    assert len(l) >= 35
    for name, n in l.items():
        assert name in h, "hipnn.lua declares %s, which include/vf_hip.h does not" % name
        assert n == h[name], "%s: %d parameters in hipnn.lua, %d in the header" % (name, n, h[name])
    called = set(re.findall(r"\bC\.(vf_[a-z0-9_]+)\s*\(", lua))
    assert called and called <= set(l), "called but not declared in the cdef: %s" % sorted(called - set(l))


def test_descriptor_mirror_matches_the_device_struct():
    """backend.COLSUM_DESC mirrors struct VfColsumDesc of csrc/vf_bn.hip field for field (64 bytes, no padding)."""
    import re
    import numpy as np
    from video_filler_amd.backend import COLSUM_DESC
    dt = np.dtype(COLSUM_DESC)
    assert dt.itemsize == 64
    src = open(os.path.join(ROOT, "video-filler_amd", "csrc", "vf_bn.hip")).read()
    body = re.search(r"struct VfColsumDesc \{(.*?)\};", src, flags=re.S).group(1)
    body = re.sub(r"//[^\n]*", "", body)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(","):
            names.append(re.sub(r"^.*[\s\*]", "", part.strip()))
    assert names == list(dt.names), (names, dt.names)


def test_range_helpers_balance_without_a_gpu():
    """vf_range_push / vf_range_pop / vf_mark (roctx when a roctx library is there, no-ops otherwise): host-only entries, so they are
    exercised here — pushes and pops balance, a stray pop is an error with a message, not a crash."""
    from video_filler_amd import _lib
    lib = _lib.load()
    assert lib.vf_trace_available() in (0, 1)
    d0 = lib.vf_range_depth()
    assert lib.vf_range_push(b"outer") == 0 and lib.vf_range_push(b"inner") == 0
    assert lib.vf_range_depth() == d0 + 2
    assert lib.vf_mark(b"a point in time") == 0
    assert lib.vf_range_pop() == 0 and lib.vf_range_pop() == 0
    assert lib.vf_range_depth() == d0
    if d0 == 0:
        assert lib.vf_range_pop() != 0 and b"no range" in lib.vf_last_error()


def test_shipped_code_objects_hold_no_packed_fma_with_a_high_dword_src1_select():
    """scripts/check_pk_opsel.py over the built library: `v_pk_fma_f32 ... op_sel:[0,1,0]` (and its mul / add / src2 relatives) gives
    wrong low results in lanes 48-63 on gfx950 when other processes share the CU — the cause of round 3's run-to-run differences
    (DESIGN.md 4.9).  The compiler chooses the form by itself, so the guard is on the binary: a kernel that acquires it after a
    source change fails HERE, before it reaches a GPU.  Since the library is built without packed FP32 ops altogether
    (build.py: -packed-fp32-ops), the guard asserts exactly that: NO v_pk_{fma,mul,add}_f32 of any form — a silently dropped feature
    flag would bring the harmless src0-select form back first, and must fail here too."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_pk_opsel", os.path.join(ROOT, "scripts", "check_pk_opsel.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    packed = []
    flagged, kernels = chk.scan(os.path.join(ROOT, "video-filler_amd", "lib", "libvf_hip.so"), packed)
    assert kernels >= 150, "the scan did not find the library's kernels (%d)" % kernels
    assert not flagged, "packed FP32 instructions selecting the high dword of src1/src2: %s" % flagged[:5]
    assert not packed, "the library is built with -packed-fp32-ops and must hold no packed FP32 arithmetic: %s" % packed[:5]
    assert chk.PK_ANY.search("v_pk_mul_f32 v[0:1], v[2:3], s[8:9]") and not chk.PK_ANY.search("v_pk_add_f16 v0, v1, v2")
    # the detector itself: the instruction text of the round-3 kernel is flagged, the shipped form is not
    assert chk.PK.search("v_pk_fma_f32 v[12:13], v[22:23], v[18:19], v[12:13] op_sel:[0,1,0]")
    m = chk.PK.search("v_pk_fma_f32 v[4:5], v[26:27], v[6:7], v[4:5] op_sel:[1,0,0]")
    assert m and not any(int(b) for b in m.group(2).split(",")[1:])
