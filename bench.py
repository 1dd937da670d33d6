#!/usr/bin/env python3
"""Headline benchmark: Mparams/s quantized (4-bit) on BASELINE.json configs[1].

A step = one pass of the GPTQ hot path over one OPT-1.3b decoder block's six Linears
(q,k,v,out 2048x2048, fc1 8192x2048, fc2 2048x8192; 4-bit, groupsize 128, static groups as
opt.py:584-587 forces), exactly as the reference drives it for one block (opt.py:177-214):
  1. Hessian accumulation: nsamples x add_batch per Linear (one 2048-token sample per call, fp16), in the order the
     reference's hooks fire; q/k/v are fed one tensor, as in the model (--no-shared-inputs: private tensors);
  2. fasterquant of every Linear (damped inverse factor + column loop + trailing updates) through
     gptq_amd.fasterquant_many (--serial-solve: one by one);
  3. 4-bit pack of the integer codes (the reference packs on the host; its own TODO, opt.py:361).
Inputs (fp16 weights + fp16 calibration activations) are resident in HBM before the timed
region.  value = params quantized by all ranks / max-over-ranks wall time.  All the work of 1-3 happens inside the
timed region, but not always in that order on the GPU: add_batch defers hook inputs by reference (--hessian-defer per
launch), and the narrow Linears' Hessian updates run beside the widest Linear's solve (--no-lazy-hessians: as the hooks
fire); `phases` reports where the time went (the "hessian" phase then holds only what was folded while the hooks fired).

N > 1 (one process per GPU, launched by torch.distributed.run): weak scaling -- the job is N
blocks' worth of Linears dealt to ranks by cost (gptq_amd.parallel.assign_units); the only
exchange is the all-gather of packed weights + grids at the step boundary (RCCL over xGMI).

Besides the contract keys the JSON line carries
  roofline      -- dominant kernel (the f16-MFMA Hessian SYRK, fp32 accumulate): algorithmic flops
                   (S*C^2 per launch, the symmetric half) / measured kernel time against the 2.5 PFLOP/s
                   dense f16 MFMA peak;
  cpu_baseline  -- the oracle (reference algorithm, torch CPU fp32) timed on this box's host
                   cores on a bounded sample of the same workload (rank 0, N = 1 only);
  phases        -- per-phase milliseconds and the solve-only Mparams/s (the scope the reference's
                   own timer prints, gptq.py:139-293).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

SHAPES = [("q_proj", 2048, 2048), ("k_proj", 2048, 2048), ("v_proj", 2048, 2048), ("out_proj", 2048, 2048),
          ("fc1", 8192, 2048), ("fc2", 2048, 8192)]
BITS, GROUPSIZE, SEQLEN = 4, 128, 2048
PEAK_F16_MFMA_TFLOPS = 2500.0      # MI355X_MICROARCH.md, dense f16/bf16 MFMA peak (the Hessian's products are exact in fp32)
PEAK_HBM_GBS = 8000.0


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nsamples", type=int, default=128, help="calibration samples per step (reference default 128)")
    ap.add_argument("--hessian-defer", type=int, default=16,
                    help="hook inputs folded into H per launch (gptq_amd.gptq.HESSIAN_DEFER; 1 = per call like the reference)")
    ap.add_argument("--serial-solve", action="store_true", help="solve the Linears one by one instead of on concurrent streams")
    ap.add_argument("--no-lazy-hessians", action="store_true",
                    help="fold every Linear's inputs into its Hessian as the hooks fire (gptq_amd.gptq.LAZY_HESSIANS = False)")
    ap.add_argument("--no-shared-inputs", action="store_true",
                    help="give q/k/v private calibration tensors (their Hessians are then accumulated three times)")
    ap.add_argument("--solve-streams", type=int, default=0, help="concurrent solves (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-samples", type=int, default=128, help="calibration samples in the CPU baseline sample")
    return ap.parse_args()


def main():
    args = parse()
    # stdout carries exactly one JSON line: everything libraries print there meanwhile (RCCL announces its version on
    # stdout when the first communicator is created) is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    # GPTQ_BENCH_BACKEND=gloo lets several ranks rehearse the N > 1 path on a box with fewer GPUs
    # (ranks then share devices and the packed-weight all-gather is staged through host memory)
    backend = os.environ.get("GPTQ_BENCH_BACKEND", "nccl")
    dev = torch.device("cuda", local_rank % max(ndev, 1))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(dev)

    import gptq_amd
    import gptq_amd.gptq as gmod
    from gptq_amd import parallel as par
    from gptq_amd import _lib
    _lib.load()
    gmod.VERBOSE = False
    gmod.HESSIAN_DEFER = args.hessian_defer
    gmod.LAZY_HESSIANS = not args.no_lazy_hessians

    # ---- unit list: `world` blocks' worth of Linears, dealt by cost -----------------------------
    units = [par.Unit(f"b{b}.{n}", r, c) for b in range(world) for (n, r, c) in SHAPES]
    costs = [par.unit_cost(u, args.nsamples, SEQLEN) for u in units]
    # q/k/v of a block are fed one tensor and share one Hessian: they travel together
    bundles = [] if args.no_shared_inputs else [[6 * b, 6 * b + 1, 6 * b + 2] for b in range(world)]
    assignment = par.assign_units(costs, world, bundles,
                                  [par.hessian_cost(units[m[0]], args.nsamples, SEQLEN) for m in bundles])
    mine = assignment[rank]
    total_params = sum(u.params for u in units)

    # ---- synthetic inputs, resident in HBM (W ~ N(0, 0.02^2), X ~ N(0,1)*(1 + c mod 7)) ----
    gen = torch.Generator(device=dev).manual_seed(1000 + rank)
    weights = {i: (torch.randn(units[i].rows, units[i].cols, device=dev, generator=gen) * 0.02).half() for i in mine}
    # Calibration inputs per Linear.  As in the model, q/k/v of a block are fed the SAME tensor (opt.py:184-185 hooks
    # fire on the one LayerNorm output); out_proj, fc1 and fc2 each see their own.  --no-shared-inputs gives every
    # Linear a private tensor (then no Hessian is shared, see gptq_amd.gptq.SHARE_INPUT_HESSIANS).
    def make_acts(C):
        chan = (1 + torch.arange(C, device=dev) % 7).half()
        return torch.randn(args.nsamples, SEQLEN, C, device=dev, generator=gen, dtype=torch.float16) * chan
    acts = {}
    shared = {}
    for i in mine:
        u = units[i]
        blk = i // 6                                   # units are laid out block by block: q,k,v,out,fc1,fc2
        if u.name.split(".")[-1] in ("q_proj", "k_proj", "v_proj") and not args.no_shared_inputs:
            key = (blk, "qkv")
            if key not in shared:
                shared[key] = make_acts(u.cols)
            acts[i] = shared[key]
        else:
            acts[i] = make_acts(u.cols)
    torch.cuda.synchronize()

    ev = lambda: torch.cuda.Event(enable_timing=True)
    phase_ms = {"hessian": 0.0, "solve": 0.0, "pack": 0.0, "allgather": 0.0}
    hess_flops = 0.0
    hess_launches = 0
    big = max(mine, key=lambda i: (units[i].cols, units[i].rows)) if mine else None   # dominant launch shape
    flush_events = []

    def step(record):
        nonlocal hess_flops, hess_launches
        packed = {}
        solvers = {}
        for i in mine:
            u = units[i]
            lin = torch.nn.Linear(u.cols, u.rows, bias=False, device=dev, dtype=torch.float16)
            lin.weight.data = weights[i].clone()
            g = gptq_amd.GPTQ(lin)
            g.quantizer = gptq_amd.Quantizer()
            g.quantizer.configure(BITS, perchannel=True, sym=False, mse=False)
            solvers[i] = g
        # 1. Hessians, in the order the reference's hooks fire (opt.py:184-185): per calibration sample,
        #    every Linear of the block gets its add_batch
        e0, e1 = ev(), ev()
        gmod.FLUSH_EVENTS = flush_events if record else None
        e0.record()
        for j in range(args.nsamples):
            for i in mine:
                solvers[i].add_batch(acts[i][j:j + 1], None)
        if not gmod.LAZY_HESSIANS:                 # lazy: the narrow Linears' updates are folded beside fc2's solve
            gmod.flush_pending()
        e1.record()
        gmod.FLUSH_EVENTS = None
        # 2. solve + pack for every Linear of the block (opt.py:189-214); the solves are independent and go out
        #    on separate streams (gptq_amd.fasterquant_many), --serial-solve restores the one-by-one loop
        s0, s1, s2 = ev(), ev(), ev()
        s0.record()
        if args.serial_solve:
            for i in mine:
                solvers[i].fasterquant(blocksize=128, percdamp=0.01, groupsize=GROUPSIZE, actorder=False,
                                       static_groups=True)
        else:
            gptq_amd.fasterquant_many([solvers[i] for i in mine], blocksize=128, percdamp=0.01, groupsize=GROUPSIZE,
                                      actorder=False, static_groups=True, max_concurrent=args.solve_streams or None)
        s1.record()
        for i in mine:
            g = solvers[i]
            packed[i] = (gptq_amd.pack_codes(g.codes, BITS), g.group_scale, g.group_zero)
            g.free()
        s2.record()
        solve_ev = [(s0, s1, s2)]
        if record:
            torch.cuda.synchronize()
            phase_ms["hessian"] += e0.elapsed_time(e1)
            for s0, s1, s2 in solve_ev:
                phase_ms["solve"] += s0.elapsed_time(s1)
                phase_ms["pack"] += s1.elapsed_time(s2)
        if world > 1:
            a0, a1 = ev(), ev()
            a0.record()
            par.allgather_packed(packed, units, assignment, BITS, GROUPSIZE)
            a1.record()
            if record:
                torch.cuda.synchronize()
                phase_ms["allgather"] += a0.elapsed_time(a1)
        return packed

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if rank == 0:
        log(f"inputs resident; {len(mine)} Linears on rank 0; warmup {args.warmup}, steps {args.steps}")
    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        log(f"timed region done: {elapsed:.3f} s for {args.steps} steps")
    ms_per_step = elapsed / args.steps * 1e3
    value = total_params / (elapsed / args.steps) / 1e6

    out = {
        "metric": "Mparams/sec quantized (4bit)", "value": round(value, 2), "unit": "Mparams/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "OPT-1.3b decoder block (BASELINE configs[1]): q,k,v,out 2048x2048, fc1 8192x2048, "
                               "fc2 2048x8192; 4-bit asym, groupsize 128 (static groups), blocksize 128, percdamp 0.01",
                   "nsamples": args.nsamples, "seqlen": SEQLEN, "blocks_per_step": world, "hessian_defer": args.hessian_defer,
                   "scope": "add_batch x nsamples + fasterquant + 4-bit pack for every Linear",
                   "calibration_inputs": "private tensor per Linear" if args.no_shared_inputs else "q,k,v share one input tensor (as in the model): their common Hessian is accumulated once; out_proj, fc1, fc2 private",
                   "parallelism": "1 GPU" if world == 1 else f"module-sharded over {world} GPUs, all-gather of packed weights"},
    }
    if rank == 0:
        steps = args.steps
        # roofline of the dominant kernel AT its dominant launch shape: the Hessian of the widest Linear
        # (fc2, C = 8192), `hessian_defer` samples of 2048 tokens per launch.  Algorithmic flops per
        # launch = samples * S * C^2 (the symmetric half actually needed); duration = HIP events around
        # each such launch (rocprofv3 summary under profiles/ agrees).
        ub = units[big]
        per_launch = max(1, args.hessian_defer)
        # executed algorithmic flops: S*C^2 (upper-triangle SYRK) per problem and slab of every Hessian launch.
        hess_flops = sum(nslab * float(SEQLEN) * sum(float(c) * c for c in Cs) for (Cs, nslab, a, b) in flush_events)
        sel = [(Cs, a.elapsed_time(b)) for (Cs, nslab, a, b) in flush_events if max(Cs) == ub.cols and nslab == per_launch]
        n_launch = len(sel)
        launch_ms = sum(d for _, d in sel) / max(n_launch, 1)
        shape_cs = sel[0][0] if sel else [ub.cols]
        flops_launch = per_launch * float(SEQLEN) * sum(float(c) * c for c in shape_cs)
        achieved = flops_launch / (launch_ms / 1e3) / 1e12 if launch_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_hessian16_big_pmc.json")   # tools/pmc_traffic.py, same launch shape
        if os.path.exists(pmc) and shape_cs == [8192] and per_launch == json.load(open(pmc)).get("samples_per_launch", 8):
            traffic = json.load(open(pmc))["hbm_bytes_per_launch"]      # FETCH_SIZE x2 (gfx950) + WRITE_SIZE
        out["roofline"] = {
            "kernel": "hessian16_big16_kernel<f16> + hessian16_big16_fixup (v_mfma_f32_16x16x32_f16 SYRK, 256x256 upper-triangle "
                      "tiles, K-split last round, fp32 accumulate; GPTQ_HESS_SHAPE=32 selects the 32x32x16 variant)",
            "launch_shape": f"Hessians of C = {sorted(shape_cs, reverse=True)} in one launch, {per_launch} samples x {SEQLEN} tokens each", "bound": "mfma",
            "achieved": round(achieved, 2), "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_F16_MFMA_TFLOPS, 4), "traffic": traffic,
            "algorithmic_flops_per_launch": flops_launch, "avg_launch_ms": round(launch_ms, 4), "launches": n_launch,
            "all_hessian_launches_tflops": round(hess_flops / (phase_ms["hessian"] / 1e3) / 1e12, 2) if phase_ms["hessian"] else None,
        }
        solve_ms = (phase_ms["solve"] + phase_ms["pack"]) / steps
        rank_params = sum(units[i].params for i in mine)
        out["phases"] = {k: round(v / steps, 3) for k, v in phase_ms.items()}
        out["phases"]["solve_only_mparams_per_s"] = round(rank_params / (solve_ms / 1e3) / 1e6, 1) if solve_ms > 0 else None
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_samples)
        sys.stdout.flush()
        with os.fdopen(json_fd, "w") as real_stdout:
            real_stdout.write(json.dumps(out) + "\n")
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(nsamples):
    """The oracle (the reference's algorithm restated on torch CPU fp32, bit-identical to the
    reference on the golden vectors) on ONE q_proj-shaped Linear of the same workload:
    nsamples x add_batch (2048 tokens each) + fasterquant (4-bit, g128 static) + pack."""
    from oracle import gptq_oracle as O
    # the box's CPU share, not the host's core count (oversubscribing MKL stalls for minutes)
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    log(f"cpu baseline: {nsamples} x add_batch + fasterquant on {torch.get_num_threads()} threads ...")
    R = C = 2048
    gen = torch.Generator().manual_seed(0)
    W = (torch.randn(R, C, generator=gen) * 0.02).half()
    chan = (1 + torch.arange(C) % 7).float()
    x = (torch.randn(1, SEQLEN, C, generator=gen) * chan).half()
    H = torch.zeros(C, C)
    t0 = time.perf_counter()
    n = 0
    for k in range(nsamples):
        n = O.hessian_add_batch(H, n, x)
        if k % 32 == 31:
            log(f"cpu baseline: add_batch {k + 1}/{nsamples} at {time.perf_counter() - t0:.1f} s")
    t1 = time.perf_counter()
    r = O.fasterquant(W, H, bits=BITS, sym=False, blocksize=128, percdamp=0.01, groupsize=GROUPSIZE,
                      actorder=False, static_groups=True)
    t2 = time.perf_counter()
    O.pack4(r.codes.t().contiguous().numpy().astype("uint32"))
    t3 = time.perf_counter()
    total = t3 - t0
    return {"value": round(R * C / total / 1e6, 4), "unit": "Mparams/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"one q_proj-shaped Linear ({R}x{C}) of the workload: {nsamples} x add_batch (2048 tokens) "
                      f"+ fasterquant (4-bit, g128 static) + pack; {total:.1f} s "
                      f"(hessian {t1 - t0:.1f} s, solve {t2 - t1:.1f} s, pack {t3 - t2:.1f} s)",
            "solve_only_mparams_per_s": round(R * C / (t2 - t1) / 1e6, 3)}


if __name__ == "__main__":
    main()
