import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_filler_amd.trainers import CenterTrainer
from video_filler_amd.backend import get_backend
B = get_backend()
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
tr = CenterTrainer(dict(batchSize=int(os.environ.get("BS","8")), nBottleneck=int(sys.argv[2]) if len(sys.argv) > 2 else 4000, wtl2=0.999, overlapPred=4), seed=1, overlap=(mode == "overlap"))
tr.set_batch(torch.rand(int(os.environ.get("BS","8")), 3, 128, 128) * 2 - 1)
print("built", flush=True)
x = tr.netG.forward(tr.input_ctx)
torch.cuda.synchronize(); print("netG fwd ok", flush=True)
tr.step()
torch.cuda.synchronize(); print("step ok", flush=True)
