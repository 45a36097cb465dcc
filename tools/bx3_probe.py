"""Runs tools/bx3_probe.hip: algorithmic TFLOP/s (1 of the 3 MFMAs counted) of the bf16x3 main loop as its pieces are added."""
import ctypes
import os
import subprocess

import torch

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "bx3_probe.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(here, "bx3_probe.hip"), "-o", so])
lib = ctypes.CDLL(so)
lib.probe_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p]
dev = torch.device("cuda:0")
src = torch.randn(1 << 30, device=dev)        # 4 GiB
out = torch.empty(4096 * 256, device=dev)
st = torch.cuda.current_stream().cuda_stream
steps = 1000
print("stage  blocks/CU  footprint/block   ms     TFLOP/s (algorithmic)")
for stage, foot in ((0, 0), (1, 0), (2, 0), (3, 0), (4, 1 << 16), (4, 1 << 20), (4, 1 << 22)):
    for bpc in (1, 2):
        blocks = 256 * bpc
        f = max(foot, 1 << 16)
        for _ in range(2):
            assert lib.probe_launch(stage, src.data_ptr(), out.data_ptr(), blocks, steps, f, st) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            lib.probe_launch(stage, src.data_ptr(), out.data_ptr(), blocks, steps, f, st)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 3
        flops = blocks * steps * 128.0 * 128 * 32 * 2
        print("%5d  %9d  %14d  %6.3f  %7.1f" % (stage, bpc, foot, ms, flops / ms / 1e9), flush=True)
