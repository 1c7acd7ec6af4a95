"""Sample rocm-smi (clocks, power, temperature) while the bench iteration replays: is the sustained shader clock the
2.4 GHz the MFMA peak is quoted at?  Usage: python scripts/probe/clock_watch.py [seconds]"""
import subprocess, sys, time, threading, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
samples = []
stop = False
def watch():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--showuse"], capture_output=True, text=True, timeout=10).stdout
        except Exception as e:
            out = "ERR %r" % e
        samples.append((time.time(), out))
        time.sleep(0.2)
import torch
from video_filler_amd.trainers import CenterTrainer
tr = CenterTrainer(dict(batchSize=64, nBottleneck=4000, wtl2=0.999, overlapPred=4), seed=1)
tr.set_batch(torch.rand(64, 3, 128, 128) * 2 - 1)
tr.capture(warmup=3)
torch.cuda.synchronize()
th = threading.Thread(target=watch); th.start()
time.sleep(1.0)
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    for _ in range(50):
        tr.replay()
    torch.cuda.synchronize(); n += 50
dt = time.time() - t0
time.sleep(0.5)
stop = True; th.join()
print("replayed %d iterations in %.2f s -> %.3f ms/step, %.0f images/s" % (n, dt, dt / n * 1e3, 64 * n / dt))
for ts, out in samples[:: max(1, len(samples) // 12)]:
    keep = [l.strip() for l in out.splitlines() if re.search(r"sclk|mclk|Power|Temperature \(Sensor (junction|edge)|GPU use", l)]
    print("t=%5.1fs " % (ts - t0) + " | ".join(re.sub(r"\s+", " ", k.split("]")[-1].strip(": ")) for k in keep)[:300])
