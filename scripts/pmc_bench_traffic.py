"""Fold the two rocprofv3 --pmc passes of scripts/pmc_bench_traffic.sh into bytes per launch per kernel."""
import csv
import glob
import json
import os
import re
import sys



# rocprof kernel name -> the name bench.py's kernel table uses (tests/test_host_logic.py pins the ones a bench line can choose)
def bench_name(k):
    suf = {"0": "", "1": "_bf16", "3": "_bf16x3"}
    m = re.match(r"void k_wgrad_group<(\d+), \w+, \w+, (\d)>", k)
    if m:
        return "wgrad_group_%sx128%s" % (m.group(1), suf.get(m.group(2), ""))
    m = re.match(r"void k_wgrad<(\d+), \w+, \w+(?:, (\d))?>", k)
    if m:
        return "wgrad_%sx128%s" % (m.group(1), suf.get(m.group(2) or "0", ""))
    # k_igemm<BM, BN, WM, WN, kmajorB, V, mode, KLIN, DB>
    m = re.match(r"void k_igemm<(\d+), (\d+), \d+, \d+, (\w+), (\d)(?:, (\d))?(?:, (\w+))?(?:, (\w+))?>", k)
    if m:
        return "igemm_%sx%s_%s_v%s%s%s" % (m.group(1), m.group(2), "kmajorB" if m.group(3) == "true" else "rowB", m.group(4),
                                          suf.get(m.group(5) or "0", ""), "_db" if m.group(7) == "true" else "")
    # k_pconv_dma<BM, BN, NTAPS> / k_pconv<BM, BN, WM, WN, NTAPS, CH, PAIR>
    m = re.match(r"void k_pwgrad_group<(\d+)(?:, (\d+))?(?:, (\d+))?>", k)          # <stages, planes per operand, fragment sets>
    if m:
        return "pwgrad_group_128x128x32" + ("_bf16" if m.group(2) == "1" else "")
    # k_pconv_dma<BM, BN, NTAPS, LDS stages (1: the two-blocks-per-CU form), planes per operand (1: the bf16-operand mode)>
    m = re.match(r"void k_pconv_dma<(\d+), (\d+), (\d+)(?:, (\w+))?(?:, (\w+))?>", k)
    if m:
        if m.group(5) == "1":
            return "pconv_dma_%sx%sx64_t%s_bf16" % (m.group(1), m.group(2), m.group(3))
        return "pconv_dma_%sx%sx64_t%s%s" % (m.group(1), m.group(2), m.group(3), "_1stage" if m.group(4) == "1" else "")
    m = re.search(r"k_pconv_patch_tr<(\d)>", k)                                   # <parity classes per block>
    if m:
        return "pconv_patch_128x64_t4_c" + m.group(1)
    m = re.search(r"k_pconv_patch_g<(\d), \w+, (\w+)>", k)                       # <MODE, stamps, TR>
    if m:
        return "pconv_patchg_128x64_%s_m%s" % ("t4" if m.group(2) == "true" else "t16", m.group(1))
    if "k_pconv_patch_h" in k:
        return "pconv_patchh_128x64_t16"
    if "k_adam_fused_multi" in k:      # the bottleneck pair's update as one launch (vf_internal_adam_fused_multi)
        return "adam_fused_wgrad"
    if "k_conv_thin_in" in k:          # (the nets' thin-input layers feed planes consumers: bench.py's name carries the suffix)
        return "conv_thin_in_planes"
    if "k_deconv_thin_out" in k:
        return "deconv_thin_out"
    if "k_wgrad_smallk" in k:      # <NJ, FUSE>: FUSE = optim.adam in the epilogue (vf_wgrad_adam_outer)
        return "adam_fused_wgrad" if re.search(r"k_wgrad_smallk<\d+, true", k) else "wgrad_smallk_f32"
    m = re.search(r"k_smallm_(rowdot|axpy)", k)
    if m:
        return "smallm_" + m.group(1)
    m = re.match(r"void k_pconv<(\d+), (\d+), \d+, \d+, (\d+), (\d+), (\w+)>", k)
    if m:
        return "pconv_%sx%sx%s_t%s%s" % (m.group(1), m.group(2), m.group(4), m.group(3), "_pair" if m.group(5) == "true" else "")
    m = re.match(r"(?:void )?k_([a-z0-9_]+)", k)      # k_adam -> adam, k_bn_stats -> bn_stats ...
    return m.group(1) if m else k

def main(src, dst):
    acc = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        files = sorted(glob.glob(os.path.join(src, counter, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
        if not files:
            raise SystemExit("no counter_collection.csv under %s/%s" % (src, counter))
        with open(files[-1]) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                name = bench_name(row["Kernel_Name"])
                e = acc.setdefault(name, {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
                e[counter][0] += float(row["Counter_Value"])
                e[counter][1] += 1
    out = {}
    for name, e in sorted(acc.items()):
        nf, nw = e["FETCH_SIZE"][1], e["WRITE_SIZE"][1]
        if not nf or not nw:
            continue
        fetch_kib, write_kib = e["FETCH_SIZE"][0] / nf, e["WRITE_SIZE"][0] / nw
        out[name] = dict(launches_counted=nf, FETCH_SIZE_KiB_per_launch=round(fetch_kib, 1), WRITE_SIZE_KiB_per_launch=round(write_kib, 1),
                         hbm_read_MB_per_launch_x2=round(2 * fetch_kib * 1024 / 1e6, 2), hbm_write_MB_per_launch=round(write_kib * 1024 / 1e6, 2),
                         hbm_MB_per_launch=round((2 * fetch_kib + write_kib) * 1024 / 1e6, 2))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import csrc_digest  # noqa: E402
    head = os.environ.get("VF_GIT_HEAD", "")
    json.dump(dict(csrc_sha256=csrc_digest(), git_head=head, workload=os.environ.get("VF_PMC_WORKLOAD", "center"), note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --no-graph --no-overlap --steps 2 "
                        "--warmup 1`; averages over every launch of the kernel in that run; FETCH_SIZE x2 (gfx950 correction, "
                        "MI355X_MICROARCH.md); kernels keyed as in bench.py's kernel table",
                   kernels=out), open(dst, "w"), indent=1)
    print(json.dumps({k: v["hbm_MB_per_launch"] for k, v in out.items()}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
