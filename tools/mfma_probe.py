"""Runs tools/mfma_probe.hip: TFLOP/s of the fp32-MFMA main loop as its pieces are added, at 1-3 blocks per CU."""
import ctypes
import os
import subprocess
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "mfma_probe.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(here, "mfma_probe.hip"), "-o", so])
lib = ctypes.CDLL(so)
lib.probe_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
src = torch.randn(72 * 1024 * 1024, device=dev)
out = torch.empty(4096 * 256, device=dev)
st = torch.cuda.current_stream().cuda_stream
steps = 2000
print("stage  blocks/CU   ms     TFLOP/s")
for stage in range(5):
    for bpc in (1, 2, 3, 4):
        blocks = 256 * bpc
        for _ in range(2):
            assert lib.probe_launch(stage, src.data_ptr(), out.data_ptr(), blocks, steps, st) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            lib.probe_launch(stage, src.data_ptr(), out.data_ptr(), blocks, steps, st)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 3
        flops = blocks * 4 * steps * 32 * 4096.0
        print("%5d  %9d  %6.3f  %7.1f" % (stage, bpc, ms, flops / ms / 1e9), flush=True)
