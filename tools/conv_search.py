#!/usr/bin/env python3
"""MIOpen find (cudnn.benchmark=True) over the U-Net's convolution shapes at a given batch, forward only or forward +
backward, so that the results land in MIOpen's user cache (tools/build_miopen_cache.sh).  Exits cleanly on SIGINT so a
`timeout -s INT` still lets MIOpen flush what it has found.   python3 tools/conv_search.py multi 16 5 fwd|fwdbwd"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from svbrdf_estimation_amd import training
training.use_in_tree_miopen_cache()
from svbrdf_estimation_amd.training import models

kind, batch, views, what = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
torch.backends.cudnn.deterministic = False
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
net = (models.MultiViewModel() if kind == "multi" else models.SingleViewModel()).to(dev).train()
x = torch.rand((batch, views, 3, 256, 256) if kind == "multi" else (batch, 3, 256, 256), device=dev)
t0 = time.time()
try:
    for it in range(2):
        if what == "fwd":
            with torch.no_grad():
                net(x)
        else:
            net(x).mean().backward()
        torch.cuda.synchronize()
        print("pass %d done after %.0f s" % (it, time.time() - t0), flush=True)
except KeyboardInterrupt:
    print("interrupted after %.0f s" % (time.time() - t0), flush=True)
    sys.exit(0)
