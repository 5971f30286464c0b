"""Times igs_amd.losses' one-launch L1 (l1_mean_kernel) at the bench image size: python tools/ubench/l1_mean_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from igs_amd import _cabi
ext = _cabi.ext()
dev = torch.device("cuda:0")
for shape in [(3, 1014, 1352), (3, 1014, 1352)]:
    a = torch.rand(shape, device=dev); b = torch.rand(shape, device=dev)
    for _ in range(5):
        ext.l1_mean(a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        m, g = ext.l1_mean(a, b)
    e1.record(); torch.cuda.synchronize()
    print(shape, "l1_mean %.1f us/call" % (e0.elapsed_time(e1) * 1000 / 50), float(m), float((a - b).abs().mean()))
    e0.record()
    for _ in range(50):
        g2 = torch.sign(a - b)
    e1.record(); torch.cuda.synchronize()
    print("torch sub+sign %.1f us/call" % (e0.elapsed_time(e1) * 1000 / 50))
