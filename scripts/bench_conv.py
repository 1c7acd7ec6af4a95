"""Micro-benchmark of single conv passes (HIP events via the library profiler).  Usage:
   python scripts/bench_conv.py [B]      -> table of layer passes with TFLOP/s"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from video_filler_amd.backend import get_backend

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(os.environ.get("REPS", "5"))
only = os.environ.get("ONLY", "")
hb = get_backend()
if os.environ.get("MFMA"):
    hb.set_mfma_mode(os.environ["MFMA"])      # f32_3xbf16 (default) | f32 (native) | bf16 (opt-in)
# (name, Cin, H, Cout, stride, pad, full)
LAYERS = [("E1", 3, 128, 64, 2, 1, 0), ("E2", 64, 64, 64, 2, 1, 0), ("E3", 64, 32, 128, 2, 1, 0), ("E4", 128, 16, 256, 2, 1, 0),
          ("E5", 256, 8, 512, 2, 1, 0), ("E6", 512, 4, 4000, 1, 0, 0), ("D1", 4000, 1, 512, 1, 0, 1), ("D2", 512, 4, 256, 2, 1, 1),
          ("D3", 256, 8, 128, 2, 1, 1), ("D4", 128, 16, 64, 2, 1, 1), ("D5", 64, 32, 3, 2, 1, 1)]
for name, Cin, H, Cout, s, p, full in LAYERS:
    if only and name not in only.split(","):
        continue
    Ho = (H - 1) * s - 2 * p + 4 if full else (H + 2 * p - 4) // s + 1
    x = hb.empty_act(B, Cin, H, H).normal_()
    y = hb.empty_act(B, Cout, Ho, Ho).normal_()
    gx = hb.empty_act(B, Cin, H, H)
    w = (hb.empty(Cin if full else Cout, 4, 4, Cout if full else Cin).normal_() * 0.02).permute(0, 3, 1, 2)
    gw = torch.zeros_like(w)
    bias = hb.zeros(Cout)
    gb = hb.zeros(Cout)
    fns = {
        "fwd": (lambda: (hb.deconv2d_fwd if full else hb.conv2d_fwd)(x, w, bias, y, 4, s, p)),
        "bwd_data": (lambda: (hb.deconv2d_bwd_data if full else hb.conv2d_bwd_data)(y, w, gx, 4, s, p)),
        "bwd_weight": (lambda: (hb.deconv2d_bwd_weight if full else hb.conv2d_bwd_weight)(x, y, gw, None, 4, s, p, 0.0)),
    }
    for pname, fn in fns.items():
        fn()
        hb.prof_begin()
        fn()
        r = hb.prof_end()
        fl = sum(v["flops"] for v in r.values())
        ks = ",".join("%s:%.1fus" % (k, v["ms"] * 1e3) for k, v in r.items())
        # back-to-back launches, one event pair around all of them (steady clocks, no host gaps)
        nb = 50
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            fn()
        e0.record()
        for _ in range(nb):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / nb
        print("%-3s %-10s B=%d  %8.1f us  %6.1f TFLOP/s   [single launch: %s]" % (name, pname, B, ms * 1e3, fl / ms / 1e9 if ms else 0, ks))
