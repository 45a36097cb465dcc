"""Time of re-packing one layer's fp16-piece panels: two-pass form against the direct form (weights' largest magnitude given), per
direction.  usage (GPU box): python tools/pack_bench.py [kind,Cin,Cout,k,stride,pad,H ...]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from locate_amd import ops  # noqa: E402
from locate_amd._lib import check, lib  # noqa: E402

SHAPES = ["convT,768,768,4,2,1,4", "convT,384,384,4,2,1,8", "conv,256,256,5,2,2,4", "conv,512,512,5,2,2,2", "conv,48,48,3,1,1,64",
          "convT,768,384,1,1,0,8"]


def main():
    L = lib()
    dev = torch.device("cuda:0")
    S = lambda: torch.cuda.current_stream().cuda_stream
    for t in (sys.argv[1:] or SHAPES):
        f = t.split(",")
        kind, (cin, cout, k, s, p, H) = f[0], [int(v) for v in f[1:]]
        spec = ops.ConvSpec(kind, k, k, s, p, p)
        wshape = (cout, cin, k, k) if kind == "conv" else (cin, cout, k, k)
        w = torch.randn(wshape, device=dev)
        geom, _ = spec.geometry((64, cin, H, H), wshape)
        garr = (ctypes.c_int * 12)(*geom)
        nw = L.locate_absmax_words()
        amax = torch.zeros(nw, dtype=torch.int32, device=dev)
        check(L.locate_absmax(w.data_ptr(), w.numel(), amax.data_ptr(), S()))
        line = "%-26s" % t
        for adjoint in (0, 1):
            nbytes = max(L.locate_conv_panel_bytes(garr, adjoint | 2), 16)
            buf = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
            check(L.locate_conv_pack_panel(garr, adjoint | 2, w.data_ptr(), buf.data_ptr(), S()))
            times = []
            for wm in (None, amax.data_ptr()):
                job = ctypes.create_string_buffer(L.locate_conv_pack_job_bytes())
                nb = ctypes.c_int(0)
                check(L.locate_conv_pack_job(garr, adjoint | 2, w.data_ptr(), buf.data_ptr(), 0, job, ctypes.byref(nb), int(wm is not None), wm))
                table = torch.frombuffer(job, dtype=torch.uint8).clone().to(dev)
                fn = lambda: check(L.locate_conv_pack_panels(table.data_ptr(), 1, nb.value, int(wm is None), int(wm is None), S()))
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    fn()
                e1.record()
                e1.synchronize()
                times.append(e0.elapsed_time(e1) / 20 * 1e3)
            line += "  %s: %6.1f -> %6.1f us (%5.1f MB)" % ("Rt" if adjoint else "R ", times[0], times[1], nbytes / 1e6)
        print(line, flush=True)


if __name__ == "__main__":
    main()
