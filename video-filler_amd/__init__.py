"""video-filler_amd — MI355X (gfx950) backend for the context-encoder GAN hot path of MKimiSH/video-filler.

Import as `video_filler_amd` (the repo-root shim video_filler_amd.py maps the hyphenated directory).
  nn        Torch7-`nn`-shaped modules and criteria over the HIP C-ABI (include/vf_hip.h)
  optim     optim.adam with the reference's call shape
  util      backend-swap hook (util.cudnn analogue) and checkpoints
  trainers  train.lua / train_vid_weighted.lua / train_wholeim_input.lua closures and loop body
  data      batch preparation on the device (train.lua:284-298, datavid/donkey_folder.lua trainHook)
  inference test_vid.lua / test_vid_wholeim.lua drivers (evaluate-mode generator, tile loop in one batch)
  build     compiles csrc/*.hip for gfx950 into lib/libvf_hip.so
"""
from . import _lib, backend, build  # noqa: F401

__all__ = ["nn", "optim", "util", "trainers", "data", "inference", "backend", "build", "_lib"]


def __getattr__(name):
    # nn / optim / util / trainers import torch and touch the backend lazily
    if name in ("nn", "optim", "util", "trainers", "data", "inference"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
