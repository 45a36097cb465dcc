"""bench.py - images/sec of one G+D training step (BASELINE.json metric) on N MI355X of one node.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path (loop body of the reference's main.py:142-172: D-step with three D
forwards + one backward + Nadam, G-step with G forward + D forward + backward + Nadam) over one synthetic
batch; workload = BASELINE.json configs[1]: 64x64 RGB, batch 64 per GPU.  Weak scaling: per-GPU batch fixed,
gradients averaged with RCCL all-reduce.  Inputs are resident in HBM before the timed region.
Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, timed live with HIP events on the launch
stream) and, at N=1, `cpu_baseline` (the CPU oracle timed on this host)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--image-size", type=int, default=64)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch")
    ap.add_argument("--dtype", choices=("fp32", "bf16", "fp8"), default="fp32",
                    help="fp32 (headline): the reference's arithmetic.  bf16: contraction operands rounded to bf16, one MFMA per "
                         "slice, fp32 accumulation and storage (BASELINE configs[1] as named; tolerances in tests/test_gpu_bf16.py).  "
                         "fp8: operands as e4m3 on the fp8 matrix instruction (BASELINE configs[4]'s arithmetic; tests/test_gpu_fp8.py)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of a captured hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--step-only", action="store_true",
                    help="profiling runs: time the step only - no roofline / hbm_bound / cpu_baseline blocks (their launches would "
                         "mix into a rocprofv3 trace of the step)")
    ap.add_argument("--concurrent-d", action="store_true",
                    help="run the three discriminator passes of the D-step on three streams (measured: no gain under "
                         "hipGraph replay on ROCm 7.0 - parallel branches are replayed almost serially)")
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--wgrad-overlap", choices=("auto", "on", "off"), default="auto",
                    help="weight gradients on a second stream beside the backward pass's chain of input gradients (same values; the "
                         "data-parallel reducer picks such gradients up at the end of the pass - tests/test_gpu_dp.py).  auto = off: "
                         "since the small weight gradients and all split reductions of a pass run as two batched launches at its end, "
                         "the ~60 forks into the second stream cost more than the concurrency returns (same-box A/B, "
                         "profiles/r03_wgrad_overlap_ab.txt: 9.88 -> 9.72 ms; in the one-rank RCCL rehearsal of round 4 9.91 / 9.85 "
                         "on, 9.68 / 9.71 off)")
    ap.add_argument("--no-wgrad-overlap", action="store_true", help="same as --wgrad-overlap off")
    ap.add_argument("--f16-min-gflop", type=float, default=None,
                    help="work threshold (GFLOP per launch) above which a fp32-faithful contraction takes its fp16-piece form "
                         "(default: locate_amd.ops.F16_MIN_FLOPS); a huge value turns the form off")
    ap.add_argument("--no-overlap", action="store_true",
                    help="replay the G-step's generator pass in line instead of on a second stream beside the D-step")
    return ap.parse_args()


PMC_RECORD = os.path.join(ROOT, "profiles", "r04_conv_pmc_mem.json")


def _pmc_traffic(flops_by_stage):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 --pmc record (tools/pmc_summary.py --json
    over tools/roofline_stages.py: the same four launches, FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes) - reported only while the record's SHA-256 of conv.hip matches the source this run uses.
    Counters need the profiler, so this is the one figure of the line that is not measured live."""
    import hashlib
    try:
        rec = json.load(open(PMC_RECORD))
        srcs = [os.path.join(ROOT, "locate_amd", "csrc", n) for n in ("conv.hip", "convwin.hip", "igemm.h")]
        if rec.get("conv_hip_sha256") != hashlib.sha256(b"".join(open(x, "rb").read() for x in srcs)).hexdigest():
            return None, "stale: the contraction kernels' sources changed since %s was taken" % os.path.relpath(PMC_RECORD, ROOT)
        stages = rec["stages"]
        if len(stages) != len(flops_by_stage):
            return None, "record has %d stages, this run %d" % (len(stages), len(flops_by_stage))
        tot = sum(flops_by_stage)
        return sum(st["hbm_bytes_per_launch"] * f for st, f in zip(stages, flops_by_stage)) / tot, os.path.relpath(PMC_RECORD, ROOT)
    except (OSError, ValueError, KeyError) as e:
        return None, "no PMC record (%s)" % type(e).__name__


def conv_roofline(cfg, batch, dev, reps=10, bf16=False, fp8=False):
    """Dominant kernel: the implicit GEMM behind the generator's four C>=96 ConvTranspose 4x4 s2 stages (75 % of
    the step's FLOPs).  Algorithmic FLOPs per launch
    = 2 * B * (2H * 2W) * C_out * C_in * 4 taps (each output pixel of a 4x4 s2 p1 transposed conv has 2x2 taps);
    algorithmic bytes per launch = weights + input + output once each, fp32 (4 (16 C^2 + B C H W + B C 2H 2W));
    time = HIP events on the launch stream around `reps` launches of each stage's forward.
    fp32 line: the kernel computes fp32-faithful products on the fp16 matrix cores - both operands scaled by a power of two and
    split into two fp16 pieces (22 significant bits), three v_mfma_f32_32x32x16_f16 per slice (hh, hl, lh) -
    so its roof is the dense fp16 MFMA peak divided by the three instructions each algorithmic multiply-add costs:
    2500 / 3 = 833.3 TFLOP/s algorithmic, i.e. `frac` = executed MFMA TFLOP/s / 2500.  (Round 2's three-piece bf16 form executed
    six: 141 TF algorithmic = 0.34 of ITS roof 416.7; the two-piece form is ~1.35x faster and sits LOWER against its higher
    roof - the frac falls while the time improves.)  bf16 line: one MFMA per slice, roof 2500."""
    from locate_amd import ops
    from locate_amd.models import generator_features
    feats = generator_features(cfg)
    rt = ops.Runtime()
    rt.precision = 3 if fp8 else (1 if bf16 else 0)
    bf16 = bf16 or fp8          # one matrix instruction per slice either way; the unscaled fp8 MFMA runs at the bf16 rate (MI355X_MICROARCH.md)
    total_flops, total_ms, rows, flops_list, alg_bytes = 0.0, 0.0, [], [], 0.0
    size = 2
    calls0 = ops.F16_CALLS["fwd"]
    for i in range(len(feats) - 1):
        c = feats[i]
        if c >= 96 and i >= 1:
            w = torch.randn(c, c, 4, 4, device=dev) * 0.05
            u = torch.randn(c, device=dev)
            v = torch.randn(c * 16, device=dev)
            x = torch.randn(batch, c, size, size, device=dev)
            if not bf16:
                ops.tag_amax(x)      # in the step the producing kernel (norm + RootTanh) leaves this word; it selects the fp16-piece form
            spec = ops.ConvSpec("convT", 4, 4, 2, 1, 1)
            pre = ops.sn_power_iteration(w, u, v)
            with torch.no_grad():
                for _ in range(2):
                    ops.sn_conv(x, w, u, v, None, spec, pre, rt)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    ops.sn_conv(x, w, u, v, None, spec, pre, rt)
                e1.record()
                e1.synchronize()
            ms = e0.elapsed_time(e1) / reps
            flops = 2.0 * batch * (2 * size) * (2 * size) * c * c * 4
            nbytes = 4.0 * (16 * c * c + batch * c * size * size + batch * c * 4 * size * size)
            rows.append({"C": c, "in": size, "ms": round(ms, 4), "tflops": round(flops / ms / 1e9, 2)})
            total_flops += flops
            total_ms += ms
            flops_list.append(flops)
            alg_bytes += nbytes * flops
        size *= 2
    f16 = (not bf16) and ops.F16_CALLS["fwd"] > calls0
    achieved = total_flops / total_ms / 1e9 if total_ms > 0 else 0.0
    alg_bytes = alg_bytes / total_flops if total_flops else 0.0          # FLOP-weighted mean per launch, like `traffic`
    traffic, source = (None, "not collected for the bf16 variant") if bf16 else _pmc_traffic(flops_list)
    per_madd = 1 if bf16 else (3 if f16 else 6)
    peak = round(2500.0 / per_madd, 1)
    if fp8:
        kernel = "conv_igemm_fp8_kernel (implicit GEMM, e4m3 operands with per-tensor power-of-two scales, one v_mfma_f32_32x32x16_fp8_fp8 per slice - the bf16 rate -, fp32 accumulate, ConvTranspose 4x4 s2 fwd)"
    elif bf16:
        kernel = "conv_igemm_bx6_kernel<NP=1> (implicit GEMM, bf16 operands, one bf16 MFMA per 32x32x16 slice, fp32 accumulate, ConvTranspose 4x4 s2 fwd)"
    elif f16:
        kernel = ("conv_igemm_bx6_kernel<NP=2> (implicit GEMM, fp32-faithful: 2 x fp16 scaled operand pieces, 3 fp16 MFMAs per 32x32x16 "
                  "slice, ConvTranspose 4x4 s2 fwd; the weight-streaming stage C = 768 through the window form conv_win_kernel<NP=2>)")
    else:
        kernel = ("conv_igemm_bx6_kernel<NP=3> (implicit GEMM, 3 x bf16 exact operand splits, 6 bf16 MFMAs per 32x32x16 slice, "
                  "ConvTranspose 4x4 s2 fwd)")
    out = {"bound": "mfma", "kernel": kernel,
           "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
           "traffic": None if traffic is None else round(traffic), "traffic_source": source,
           "algorithmic_bytes": round(alg_bytes), "per_stage": rows}
    if not bf16:
        out.update({"mfma_per_multiply_add": per_madd, "executed_mfma_tflops": round(per_madd * achieved, 1), "dense_mfma_peak": 2500.0,
                    "fp32_mfma_peak": 157.3,
                    "note": "achieved = ALGORITHMIC TFLOP/s; frac = achieved / (2500 / mfma_per_multiply_add) = executed matrix-core TFLOP/s / "
                            "2500.  Round 2 (six bf16 MFMAs per multiply-add): 141 TF algorithmic, 846 executed, frac 0.34; the two-piece "
                            "fp16 form executes half the instructions, so the time drops (~0.13 -> ~0.10 ms per stage) while frac drops too",
                    # what a bare MFMA loop sustains on RANDOM operands (the chip lowers its clock under matrix load:
                    # MI355X_MICROARCH.md "DVFS give-back" measures 1247 TF there, 1483 on zeros)
                    "sustained_mfma_on_random_data": 1247.0, "frac_of_sustained": round(per_madd * achieved / 1247.0, 4)})
    return out


def hbm_bound_block(batch, dev, reps=10):
    """north_star's ">= 40 % achieved HBM bandwidth" figure: the HBM-bound kernels of the step (RootTanh, InPlaceNorm incl. its
    statistics pass, residual gate, softmax over positions, bilinear upsample, average pool, FeaturePooling), forward and
    backward, at the activation shapes config 2 actually runs them on, each weighted by how often the step runs it; timed live
    with HIP events through the C ABI.  Algorithmic bytes = every distinct operand read or written once, fp32 (DESIGN.md
    section 4.2); achieved = sum of bytes / sum of time; peak 8.0 TB/s (MI355X_MICROARCH.md; ~6.3 TB/s is what a copy reaches)."""
    from locate_amd._lib import check, lib
    L = lib()
    st = torch.cuda.current_stream().cuda_stream

    def P(t):
        return t.data_ptr()

    def timed(fn):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    rows, tot_b, tot_t = [], 0.0, 0.0

    def add(name, shape, count, nbytes, fn):
        nonlocal tot_b, tot_t
        t = timed(fn)
        rows.append({"op": name, "shape": "x".join(map(str, shape)), "per_step": count, "GBps": round(nbytes / t / 1e9, 1)})
        tot_b += count * nbytes
        tot_t += count * t

    # (shape, how many RootTanh fwd / bwd, norm fwd / bwd, gate fwd / bwd per step run on it) - generator stages of config 2;
    # the step runs the generator forward twice and its backward once
    for shape, n_act, n_norm, n_gate in (((batch, 96, 64, 64), 1, 0, 0), ((batch, 48, 64, 64), 3, 3, 3), ((batch, 96, 32, 32), 1, 1, 1),
                                         ((batch, 192, 32, 32), 1, 0, 0), ((batch, 192, 16, 16), 2, 3, 3)):
        Bn, C, H, W = shape
        hw, planes, n = H * W, Bn * C, Bn * C * H * W
        x, g, y, gx, a = (torch.randn(shape, device=dev) for _ in range(5))
        add("RootTanh fwd", shape, 2 * n_act, 8 * n, lambda: check(L.locate_roottanh_fwd(P(x), P(y), n, None, st)))
        add("RootTanh bwd", shape, n_act, 12 * n, lambda: check(L.locate_roottanh_bwd(P(x), P(g), P(gx), n, 0, None, st)))
        if n_norm:
            w_, b_ = torch.ones(C, device=dev), torch.zeros(C, device=dev)
            dw, db, stats = torch.empty(C, device=dev), torch.empty(C, device=dev), torch.empty(2, device=dev)
            ws_f = torch.empty(max(L.locate_norm_stats_workspace_bytes(), 16), dtype=torch.uint8, device=dev)
            ws_b = torch.empty(max(L.locate_norm_bwd_workspace_bytes(Bn, C), 16), dtype=torch.uint8, device=dev)
            add("InPlaceNorm fwd (stats + apply)", shape, 2 * n_norm, 12 * n,
                lambda: check(L.locate_norm_fwd(P(x), P(w_), 0, P(b_), P(y), 0, P(stats), Bn, C, hw, 1, P(ws_f), None, None, st)))
            add("InPlaceNorm bwd", shape, n_norm, 20 * n,
                lambda: check(L.locate_norm_bwd(P(x), P(g), P(stats), P(w_), 0, P(b_), 0, P(gx), P(dw), P(db), Bn, C, hw, 1, P(ws_b), 0, st)))
        if n_gate:
            da = torch.empty_like(a)
            gam, dgam = torch.full((1,), 2.0, device=dev), torch.empty(1, device=dev)
            ws_g = torch.empty(max(L.locate_gate_bwd_workspace_bytes(planes), 16), dtype=torch.uint8, device=dev)
            add("gate fwd", shape, 2 * n_gate, 12 * n, lambda: check(L.locate_gate_fwd(P(x), P(a), 0, P(gam), P(y), planes, hw, st)))
            add("gate bwd", shape, n_gate, 20 * n,
                lambda: check(L.locate_gate_bwd(P(x), P(a), 0, P(gam), P(g), P(gx), P(da), P(dgam), planes, hw, P(ws_g), 0, None, st)))
        if (C, H) in ((48, 64), (192, 16)):
            add("softmax over H*W fwd", shape, 2, 8 * n, lambda: check(L.locate_softmax_fwd(P(x), P(y), planes, hw, st)))
            add("softmax over H*W bwd", shape, 1, 12 * n, lambda: check(L.locate_softmax_bwd(P(y), P(g), P(gx), planes, hw, st)))
        if H == 32 and C == 96:
            hp = torch.empty(Bn, C // 2, H, W, device=dev)
            up = torch.empty(Bn, C // 2, 2 * H, 2 * W, device=dev)
            gup = torch.randn_like(up)
            add("FeaturePooling / 2", shape, 2, 6 * n, lambda: check(L.locate_feature_pool_fwd(P(x), P(hp), n // 2, 2, st)))
            add("upsample x2 fwd", (Bn, C // 2, H, W), 2, 10 * n, lambda: check(L.locate_upsample2x_fwd(P(hp), P(up), planes // 2, H, W, st)))
            add("upsample x2 bwd", (Bn, C // 2, H, W), 1, 10 * n, lambda: check(L.locate_upsample2x_bwd(P(gup), P(hp), planes // 2, H, W, st)))
    # discriminator side: the stacked [3B] pass at 32x32 (stem output) - average pool and its gate / norm
    shape = (3 * batch, 32, 32, 32)
    Bn, C, H, W = shape
    n, planes = Bn * C * H * W, Bn * C
    x, gx = torch.randn(shape, device=dev), torch.empty(shape, device=dev)
    q = torch.empty(Bn, C, H // 2, W // 2, device=dev)
    gq = torch.randn_like(q)
    add("avgpool 2x2 fwd", shape, 1, 5 * n, lambda: check(L.locate_avgpool2_fwd(P(x), P(q), planes, H, W, st)))
    add("avgpool 2x2 bwd", shape, 1, 5 * n, lambda: check(L.locate_avgpool2_bwd(P(gq), P(gx), planes, H, W, 0, st)))
    achieved = tot_b / tot_t / 1e9 if tot_t > 0 else 0.0
    return {"bound": "hbm", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 4),
            "what": "byte-weighted over the step's HBM-bound kernels (element-wise, norm, gate, softmax, resampling) at config 2's "
                    "activation shapes and multiplicities; algorithmic fp32 bytes, HIP events, this run", "kernels": rows}


def cpu_baseline(cfg, batch, steps):
    """The CPU oracle (a port of the reference's step, parity-pinned by tests/golden) on this host's cores."""
    from oracle import locate_oracle as O
    from locate_amd import Discriminator, Generator
    torch.manual_seed(999)
    G, D = Generator(cfg), Discriminator(cfg)
    ocfg = O.NetConfig(image_size=cfg.image_size, base_feature_factor=cfg.base_feature_factor)
    PG = O.make_params({k: v.clone() for k, v in G.state_dict().items()})
    PD = O.make_params({k: v.clone() for k, v in D.state_dict().items()})
    og, od = O.Nadam(ocfg.glr, (ocfg.beta1, ocfg.beta2)), O.Nadam(ocfg.dlr, (ocfg.beta1, ocfg.beta2))
    S = cfg.image_size
    latent = torch.randn(batch, S)
    real = torch.randn(batch, 3, S, S).clamp(-1, 1)
    aug = torch.randn(batch, 3, S, S).clamp(-1, 1)
    torch.set_num_threads(min(16, os.cpu_count() or 1))     # the GPU box grants one GPU's share of host cores
    O.train_step(PG, PD, G.noise.clone(), og, od, latent, real, aug, ocfg)   # warm-up
    t0 = time.time()
    for _ in range(steps):
        O.train_step(PG, PD, G.noise.clone(), og, od, latent, real, aug, ocfg)
    dt = (time.time() - t0) / steps
    return {"value": round(batch / dt, 3), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d warm-up + %d timed G+D steps of the same workload (%dx%d, batch %d, fp32) with this repo's CPU oracle - a port "
                      "of the reference's step pinned to it by tests/golden, NOT the reference itself (which cannot travel to the GPU "
                      "box; its own figure, 13.5 images/sec on 8 threads, is in BASELINE.md)" % (1, steps, S, S, batch),
            "s_per_step": round(dt, 3)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # LOCATE_DP_FORCE=1: the data-parallel machinery at world size 1 - process group, both reducers, the discriminator's cut
    # backward, buckets on the side stream between the graph replays, RCCL all-reduce (a self-copy), unpack, join - as a one-rank
    # rehearsal of the multi-GPU path on a single GPU (launch: python -m torch.distributed.run --nproc-per-node 1 bench.py)
    force_dp = os.environ.get("LOCATE_DP_FORCE") == "1" and "RANK" in os.environ
    dp = world > 1 or force_dp
    if dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # rehearsal hook: LOCATE_BENCH_BACKEND=gloo lets several ranks share one GPU (RCCL refuses that) to exercise the
    # data-parallel control flow on a single-GPU box; the driver's runs use one GPU per rank over RCCL
    backend = os.environ.get("LOCATE_BENCH_BACKEND", "nccl")
    local_dev = local_rank % max(torch.cuda.device_count(), 1) if backend != "nccl" else local_rank
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if dp:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from locate_amd import Discriminator, Generator, NetConfig, TrainStep, get_model
    from locate_amd._lib import require_gpu
    from locate_amd.parallel import GradAllReducer, broadcast_module_state
    from locate_amd.graph import GraphedTrainStep
    require_gpu()

    if args.f16_min_gflop is not None:
        from locate_amd import ops as _ops
        _ops.F16_MIN_FLOPS = args.f16_min_gflop * 1e9
    cfg = NetConfig(image_size=args.image_size)
    torch.manual_seed(cfg.seed)
    G, GO = get_model(Generator(cfg), cfg.glr, dev)
    D, DO = get_model(Discriminator(cfg), cfg.dlr, dev)
    G.batched_spectral_norm = D.batched_spectral_norm = True
    G.set_precision(args.dtype)
    D.set_precision(args.dtype)
    red_g = red_d = None
    d_cut = None
    if dp:
        broadcast_module_state(G, 0, extra_tensors=[G.noise])
        broadcast_module_state(D, 0)
        late_v = [p for n, p in D.named_parameters() if n.endswith("weight_v")]
        # the D-step's backward runs in two segments cut behind the discriminator's block 2 (maps >= 8x8 | <= 4x4): the deep
        # segment holds ~95 % of D's parameters and is on the wire while the high-resolution segment's backward runs
        d_cut = 3 if (len(D.main[1].blocks) > 3 and os.environ.get("LOCATE_DP_NOCUT") != "1") else None      # (experiment switch)
        solo = int(os.environ.get("LOCATE_DP_SOLO_BYTES", "0"))     # > 0: gradients from this size up are all-reduced in place
        bucket = int(os.environ.get("LOCATE_DP_BUCKET_MB", "128")) << 20   # one collective per backward segment at config 2
        red_g = GradAllReducer(G.parameters(), force=force_dp, solo_bytes=solo, bucket_bytes=bucket)
        red_d = GradAllReducer(D.parameters(), late=late_v, groups=D.segment_parameters(d_cut) if d_cut else None, force=force_dp,
                               solo_bytes=solo, bucket_bytes=bucket)
    wgrad_overlap = args.wgrad_overlap == "on" and not args.no_wgrad_overlap
    step = TrainStep(G, D, GO, DO, reducer_g=red_g, reducer_d=red_d, concurrent_d=args.concurrent_d,
                     overlap_wgrad=wgrad_overlap, d_cut=d_cut)
    B, S = args.batch, args.image_size
    gen = torch.Generator(device="cpu").manual_seed(1234 + rank)
    latent = torch.randn(B, S, generator=gen).to(dev)
    real = torch.randn(B, 3, S, S, generator=gen).clamp(-1, 1).to(dev)
    aug = torch.randn(B, 3, S, S, generator=gen).clamp(-1, 1).to(dev)

    use_graph = not args.no_graph
    runner = GraphedTrainStep(step, latent, real, aug, warmup=2, overlap=False if args.no_overlap else None) if use_graph else None

    def one_step():
        if runner is not None:
            return runner.replay()
        return step(latent, real, aug)

    def barrier():
        if dp:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = one_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_step()
    barrier()
    elapsed = time.perf_counter() - t0
    comm = None
    if dp:
        # a few extra steps with HIP events around the exchange: how much of it hides behind backward work
        red_g.timing = red_d.timing = True
        for _ in range(5):
            one_step()
        tg, xg = red_g.pop_timing()
        td, xd = red_d.pop_timing()
        red_g.timing = red_d.timing = False
        comm = {"buckets_sent": {"D": {"in_place": red_d.sent_in_place, "packed": red_d.sent_packed},
                                 "G": {"in_place": red_g.sent_in_place, "packed": red_g.sent_packed}},
                "collectives": {"D": red_d.collectives, "G": red_g.collectives},
                "allreduce_ms_per_step": {"D": round(td / 5, 4), "G": round(tg / 5, 4)},
                "exposed_ms_per_step": {"D": round(xd / 5, 4), "G": round(xg / 5, 4)},
                "payload_MB": {"D": round(sum(p.numel() for p in D.parameters()) * 4 / 1e6, 1),
                               "G": round(sum(p.numel() for p in G.parameters()) * 4 / 1e6, 1)},
                "note": "buckets_sent / collectives: counts over the whole run (eager warm-up included; a replayed step sends a "
                        "backward segment's resident buckets as ONE collective, handed to the process group from the compute stream). "
                        "allreduce_ms: from the hand-over to the compute stream's join (an upper bound of the exchange: the overlapped "
                        "segment is inside) - for packed buckets the side stream's pack + all-reduce + unpack; exposed_ms: what "
                        "the compute stream waited at the join; D's deep segment (blocks 3.. + head, ~95 % of its parameters) is sent while "
                        "the high-resolution segment's backward runs, G's widest layers are the last its backward produces"}
    if dp:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if dp:
        # replicas must still be identical after the timed steps (same averaged gradients, same deterministic kernels)
        chk = torch.stack([sum(p.detach().double().sum() for p in net.parameters()) for net in (G, D)])
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert torch.equal(lo, hi), "replicas diverged: parameter checksums %s vs %s" % (lo.tolist(), hi.tolist())
    d_error = float(out["d_error"])
    g_error = float(out["g_error"])
    assert d_error == d_error and g_error == g_error, "non-finite loss in the benchmark"

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        line = {
            "metric": "images/sec (G+D step) %dx%d bs=%d" % (S, S, B), "value": round(world * B * args.steps / elapsed, 2),
            "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (contractions fp32-faithful on the matrix cores: 2 x scaled fp16 operand pieces / 3 MFMAs per slice where the operand's "
                     "largest magnitude comes with it, else 3 x bf16 pieces / 6 MFMAs; fp32 accumulate)" if args.dtype == "fp32" else
                     ("bf16 (contraction operands rounded to bf16, one MFMA per slice, fp32 accumulate; storage, statistics, "
                      "sigma, activations and Nadam fp32) - NOT the headline line, see --dtype fp32" if args.dtype == "bf16" else
                      "fp8 (contraction operands as OCP e4m3 with per-tensor power-of-two scales on v_mfma_f32_32x32x16_fp8_fp8, fp32 "
                      "accumulate; storage, statistics, sigma, activations and Nadam fp32; BASELINE configs[4]'s arithmetic, tolerances "
                      "in tests/test_gpu_fp8.py) - NOT the headline line, see --dtype fp32"), "data": "synthetic",
            "config": {"workload": "LocAtE G+D step, %dx%d RGB, batch %d per GPU (%s of BASELINE.json configs[1]: 64x64 RGB bs 64), "
                                   "self/feature attention at 16x16 and 64x64, random-init weights"
                                   % (S, S, B, "the fp32 variant - the reference's own precision -" if args.dtype == "fp32" else "the %s variant" % args.dtype),
                       "global_batch": world * B, "image_size": S, "parallelism": "dp%d" % world,
                       "launch": ("hipGraph replay" + ("" if args.no_overlap else ", G-step generator pass on a second stream")
                                  + (", weight gradients on a second stream" if step.overlap_wgrad else "")
                                  + (" + bucketed RCCL all-reduce on a side stream between the graphs of the segmented backward" if dp else "")
                                  + (" [LOCATE_DP_FORCE: one-rank rehearsal of the data-parallel path]" if force_dp and world == 1 else ""))
                                 if use_graph else "eager (all-reduce overlapped with backward)"},
            "losses": {"d_error": round(d_error, 5), "g_error": round(g_error, 5)},
        }
        if comm is not None:
            line["data_parallel"] = comm
        if not args.step_only:
            line["roofline"] = conv_roofline(cfg, B, dev, bf16=args.dtype == "bf16", fp8=args.dtype == "fp8")
            if S == 64:
                line["hbm_bound"] = hbm_bound_block(B, dev)
        if S == 64:
            # step-level figures with SURVEY.md section 8(d)'s op-by-op accounting of the REFERENCE graph (7.5 GFLOP and
            # 356 MB fp32 per image at 64x64): effective rates - fusion that never materialises an intermediate counts
            step_s = elapsed / args.steps
            line["step_level"] = {"algorithmic_tflops": round(7.5e9 * B / step_s / 1e12, 2),
                                  "reference_graph_GBps": round(356e6 * B / step_s / 1e9, 1),
                                  "frac_of_hbm_peak": round(356e6 * B / step_s / 8.0e12, 4)}
        if world == 1 and not args.no_cpu_baseline and not args.step_only:
            line["cpu_baseline"] = cpu_baseline(cfg, B, args.cpu_steps)
        print(json.dumps(line), flush=True)
    if dp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
