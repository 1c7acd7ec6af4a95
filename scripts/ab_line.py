import json, sys
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l)
        k = d["kernels"]
        g = sum(v["ms_per_step"] for n, v in k.items() if n.startswith("igemm"))
        extra = " ".join("%s=%.1f" % (n.replace("igemm_", "").replace("_bf16x3", ""), 1e3 * v["ms_per_step"]) for n, v in k.items()
                         if n.startswith("igemm_64x128") or n.startswith("slab_reduce_igemm"))
        print(sys.argv[1], d["value"], d["ms_per_step"], "igemm_ms=%.3f" % g, extra)
