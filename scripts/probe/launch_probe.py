#!/usr/bin/env python3
"""Dependent-launch cost probe: a chain of N tiny kernels, eager vs captured in a HIP graph, under a few HIP runtime
knobs (each variant in a child process, the knob set before the runtime loads).  Prints us per kernel."""
import os
import subprocess
import sys
import time

CHILD = r'''
import os, sys, time, torch
n = int(sys.argv[1])
x = torch.zeros(4096, device="cuda")
big = torch.zeros(16 << 20, device="cuda")      # 64 MB: a kernel that dirties L2 before each tiny one (mode "dirty")
def chain(dirty):
    for i in range(n):
        if dirty and i % 8 == 0:
            big.add_(1.0)
        x.add_(1.0)
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6
out = {}
for dirty in (False, True):
    tag = "dirty" if dirty else "tiny"
    out["eager_" + tag] = timed(lambda: chain(dirty)) / n
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        chain(dirty); torch.cuda.synchronize()
        with torch.cuda.graph(g):
            chain(dirty)
    out["graph_" + tag] = timed(g.replay) / n
print(" ".join("%s=%.2f" % kv for kv in out.items()))
'''

VARIANTS = [
    {},
    {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "0"},
    {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "1"},
    {"AMD_OPT_FLUSH": "0"},
    {"AMD_OPT_FLUSH": "1"},
    {"HIP_FORCE_DEV_KERNARG": "1"},
    {"DEBUG_HIP_GRAPH_BATCH_SIZE": "1"},
    {"DEBUG_HIP_GRAPH_BATCH_SIZE": "64"},
    {"ROC_SYSTEM_SCOPE_SIGNAL": "0"},
    {"GPU_FLUSH_ON_EXECUTION": "0"},
    {"DEBUG_HIP_FORCE_GRAPH_QUEUES": "1"},
]

if __name__ == "__main__":
    n = sys.argv[1] if len(sys.argv) > 1 else "200"
    for v in VARIANTS:
        env = dict(os.environ)
        env.update(v)
        t0 = time.time()
        r = subprocess.run([sys.executable, "-c", CHILD, n], env=env, capture_output=True, text=True, timeout=300)
        print("%-45s %s   (%.0fs)" % (v or "default", r.stdout.strip() or ("ERR " + r.stderr.strip()[-300:]), time.time() - t0), flush=True)
