import json, sys
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l)
        k = d["kernels"]
        g = sum(v["ms_per_step"] for n, v in k.items() if n.startswith("pconv"))
        extra = " ".join("%s=%.1f" % (n.replace("pconv_dma_128x64x64_", ""), 1e3 * v["ms_per_step"]) for n, v in k.items() if n.startswith("pconv"))
        print(sys.argv[1], d["value"], d["ms_per_step"], "pconv_ms=%.3f" % g, extra)
