"""Do the fork/join branches of ONE captured hipGraph run side by side?  Two chains of small dependent kernels (launch-floor
bound), captured (a) back to back on one stream, (b) as two branches of one graph (second stream forked inside the
capture), (c) as two graphs replayed on two streams.  Usage (GPU box): python tools/graph_branch_probe.py"""
import torch

N = 300


def chain(x):
    for _ in range(N):
        x.mul_(1.0001)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device("cuda:0")
    for numel in (1 << 12, 1 << 20, 1 << 23):
        a, b = torch.ones(numel, device=dev), torch.ones(numel, device=dev)
        cap, side = torch.cuda.Stream(), torch.cuda.Stream()
        g_serial = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_serial, stream=cap):
            chain(a)
            chain(b)
        g_fork = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_fork, stream=cap):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                chain(b)
            chain(a)
            torch.cuda.current_stream().wait_stream(side)
        g_a, g_b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        pool_b = torch.cuda.graph_pool_handle()
        with torch.cuda.graph(g_a, stream=cap):
            chain(a)
        with torch.cuda.graph(g_b, stream=cap, pool=pool_b):
            chain(b)
        s2 = torch.cuda.Stream()

        def two_streams():
            main_s = torch.cuda.current_stream()
            s2.wait_stream(main_s)
            with torch.cuda.stream(s2):
                g_b.replay()
            g_a.replay()
            main_s.wait_stream(s2)
        print("%8d elements x 2 chains of %d kernels:  one stream %.3f ms | fork inside one graph %.3f ms | two graphs on two streams %.3f ms"
              % (numel, N, timed(g_serial.replay), timed(g_fork.replay), timed(two_streams)))


if __name__ == "__main__":
    main()
