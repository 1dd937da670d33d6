#!/usr/bin/env python3
"""Headline benchmark: Mparams/s quantized (4-bit) by the GPTQ hot path on one transformer block.

Workloads (--workload; shapes are the HF architecture constants):
  llama7b  (default; BASELINE configs[2], the largest single-GPU 4-bit config): q,k,v,o 4096x4096, gate,up 11008x4096,
           down 4096x11008; 4-bit asym per-channel, --act-order, --true-sequential: the four groups
           [k,v,q] -> [o] -> [up,gate] -> [down] (llama.py:97-105) are calibrated and solved ONE AFTER THE OTHER, as
           the real pipeline must (each group's inputs depend on the previous group's quantized weights);
  opt6.7b  (north_star's >= 10x target config): q,k,v,out 4096x4096, fc1 16384x4096, fc2 4096x16384; 4-bit, one group
           (opt.py:189-214), static groups forced like opt.py:585 (a no-op without --groupsize);
  opt1.3b  (BASELINE configs[1], round-1 continuity): 2048 / 8192, groupsize 128 static;
  llama65b (BASELINE configs[4]): 8192 / 22016, 4-bit act-order true-sequential.
A step = one pass of the hot path over one block, exactly as the reference drives it (opt.py:177-214,
llama.py:97-190), per group:
  1. Hessian accumulation: nsamples x add_batch per Linear (one 2048-token fp16 sample per call) in the order the
     hooks fire; Linears fed the same tensor in the model (q/k/v; gate/up) are fed one tensor here;
  2. fasterquant of the group's Linears (damped inverse factor + column loop + trailing updates) through
     gptq_amd.fasterquant_many;
  3. 4-bit pack of the integer codes (the reference packs on the host; its own TODO, opt.py:361).
Inputs (fp16 weights + fp16 calibration activations) are resident in HBM before the timed region.
value = params of the block / max-over-ranks wall time per step.

N > 1 (one process per GPU, launched by torch.distributed.run): STRONG scaling of the same block -- data-parallel over
the calibration samples (each rank folds nsamples/N samples into its Hessians), one all-reduce of H per group
(RCCL over xGMI), the factorization chain replicated, the rows of W split over the ranks, one all-gather of the packed
rows + grids per group (gptq_amd.parallel.fasterquant_sharded).

Besides the contract keys the JSON line carries
  roofline      -- dominant kernel (the f16-MFMA Hessian SYRK, fp32 accumulate) at its dominant launch shape:
                   algorithmic flops (samples*S*C^2 per launch, the symmetric half) / measured launch time (HIP events
                   on the launching stream) against the 2.5 PFLOP/s dense f16 MFMA peak; `traffic` = HBM bytes per
                   launch from the PMC passes committed under profiles/ (same launch shape) or null;
  cpu_baseline  -- the oracle (reference algorithm, torch CPU fp32, pinned bit-for-bit to the reference by the golden
                   fixtures) timed on this box's host cores on a bounded sample of the same workload's shapes
                   (rank 0, N = 1 only);
  phases        -- per-phase milliseconds and the solve-only Mparams/s (the scope the reference's own timer prints,
                   gptq.py:139-293);
  also          -- the same measurement (fewer steps) on the other headline configuration (opt6.7b when the workload is
                   llama7b), with the GPU / CPU ratio north_star's ">= 10x" refers to.
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

SEQLEN = 2048
PEAK_F16_MFMA_TFLOPS = 2500.0      # MI355X_MICROARCH.md, dense f16/bf16 MFMA peak (the Hessian's products are exact in fp32)
PEAK_HBM_GBS = 8000.0


def _llama(h, f):
    return [[("k_proj", h, h, "ln1"), ("v_proj", h, h, "ln1"), ("q_proj", h, h, "ln1")], [("o_proj", h, h, "attn")],
            [("up_proj", f, h, "ln2"), ("gate_proj", f, h, "ln2")], [("down_proj", h, f, "act")]]


def _opt(h, f):
    return [[("q_proj", h, h, "ln1"), ("k_proj", h, h, "ln1"), ("v_proj", h, h, "ln1"), ("out_proj", h, h, "attn"),
             ("fc1", f, h, "ln2"), ("fc2", h, f, "act")]]


# name -> groups of (Linear, out_features R, in_features C, input key); Linears of a group with one key share a tensor
WORKLOADS = {
    "llama7b": dict(groups=_llama(4096, 11008), bits=4, groupsize=-1, actorder=True, static_groups=False,
                    desc="Llama-7B decoder block (BASELINE configs[2]): q,k,v,o 4096x4096, gate,up 11008x4096, down "
                         "4096x11008; 4-bit asym per-channel, act-order, true-sequential groups [k,v,q]->[o]->[up,gate]->[down]"),
    "opt6.7b": dict(groups=_opt(4096, 16384), bits=4, groupsize=-1, actorder=False, static_groups=True,
                    desc="OPT-6.7B decoder block (north_star target config): q,k,v,out 4096x4096, fc1 16384x4096, fc2 "
                         "4096x16384; 4-bit asym per-channel, one group"),
    "opt1.3b": dict(groups=_opt(2048, 8192), bits=4, groupsize=128, actorder=False, static_groups=True,
                    desc="OPT-1.3b decoder block (BASELINE configs[1]): q,k,v,out 2048x2048, fc1 8192x2048, fc2 2048x8192; "
                         "4-bit asym, groupsize 128 (static groups)"),
    "llama65b": dict(groups=_llama(8192, 22016), bits=4, groupsize=-1, actorder=True, static_groups=False,
                     desc="Llama-65B decoder block (BASELINE configs[4]): q,k,v,o 8192x8192, gate,up 22016x8192, down "
                          "8192x22016; 4-bit asym per-channel, act-order, true-sequential groups"),
    # configs[4] as BASELINE words it ("per-block Linear layers sharded", no --true-sequential): ONE hooked group with four
    # distinct Hessians (q/k/v share one, gate/up share one) whose chains run on disjoint rank sets (parallel.plan_rows)
    "llama65b-1group": dict(groups=[[l for g in _llama(8192, 22016) for l in g]], bits=4, groupsize=-1, actorder=False,
                            static_groups=False,
                            desc="Llama-65B decoder block, all seven Linears hooked in one pass (BASELINE configs[4] as worded, "
                                 "llama.py:97-105 without --true-sequential): q,k,v,o 8192x8192, gate,up 22016x8192, down "
                                 "8192x22016; 4-bit asym per-channel"),
}


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="llama7b", choices=sorted(WORKLOADS))
    ap.add_argument("--wbits", type=int, default=0, help="override the workload's bit width (3 or 4)")
    ap.add_argument("--groupsize", type=int, default=0, help="override the workload's groupsize (-1 = per row)")
    ap.add_argument("--nsamples", type=int, default=128, help="calibration samples per step (reference default 128)")
    ap.add_argument("--hessian-defer", type=int, default=16,
                    help="hook inputs folded into H per launch (gptq_amd.gptq.HESSIAN_DEFER; 1 = per call like the reference)")
    ap.add_argument("--serial-solve", action="store_true", help="solve the Linears one by one instead of on concurrent streams")
    ap.add_argument("--no-lazy-hessians", action="store_true",
                    help="fold every Linear's inputs into its Hessian as the hooks fire (gptq_amd.gptq.LAZY_HESSIANS = False)")
    ap.add_argument("--no-shared-inputs", action="store_true",
                    help="give q/k/v (gate/up) private calibration tensors (their Hessians are then accumulated per Linear)")
    ap.add_argument("--solve-streams", type=int, default=0, help="concurrent solves (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary measurement on the other headline config")
    ap.add_argument("--also-steps", type=int, default=2)
    ap.add_argument("--project-gpus", type=int, default=0, metavar="N",
                    help="N = 1 only: add `projection` for N GPUs from measured single-GPU pieces (per distinct Hessian: the "
                         "solve at full and at half the rows -> replicated chain + per-row cost; parallel.plan_rows)")
    ap.add_argument("--end-to-end", type=int, default=-1, metavar="BLOCKS",
                    help="add `end_to_end`: gptq_amd.sequential.quantize_sequential on a random-init Llama-7B-architecture "
                         "model of BLOCKS decoder blocks (the reference's 'full quantization time' scope, opt.py:686-691); "
                         "default: 2 blocks when the workload is llama7b on one GPU, 0 = off")
    ap.add_argument("--ppl-proxy", action="store_true",
                    help="add `ppl_proxy`: our OPT driver + evaluator vs the reference's numbers on the random-init OPT-125m "
                         "architecture of tests/golden/g6_opt125m.npz (a proxy: no Wiki2 / checkpoints offline)")
    return ap.parse_args()


class Block:
    """One workload instance on this rank: weights + calibration inputs resident in HBM, and the step."""

    def __init__(self, name, args, dev, rank, world, group=None, dist_on=False):
        import gptq_amd
        import gptq_amd.gptq as gmod
        self.G, self.gmod = gptq_amd, gmod
        self.wl = dict(WORKLOADS[name])
        if args.wbits:
            self.wl["bits"] = args.wbits
        if args.groupsize:
            self.wl["groupsize"] = args.groupsize
        self.name, self.args, self.dev, self.rank, self.world, self.group = name, args, dev, rank, world, group
        self.dist_on = dist_on or world > 1
        self.groups = self.wl["groups"]
        self.params = sum(r * c for g in self.groups for (_, r, c, _) in g)
        # data-parallel over the calibration samples: rank r holds samples r, r + world, ...
        self.local_samples = len(range(rank, args.nsamples, world))
        gen = torch.Generator(device=dev).manual_seed(1000)         # weights: identical on every rank
        self.weights = {n: (torch.randn(r, c, device=dev, generator=gen) * 0.02).half()
                        for g in self.groups for (n, r, c, _) in g}
        gen = torch.Generator(device=dev).manual_seed(2000 + rank)  # activations: this rank's samples
        self.acts = {}
        for gi, g in enumerate(self.groups):
            for (n, r, c, key) in g:
                k = (gi, n if args.no_shared_inputs else key)
                if k not in self.acts:
                    chan = (1 + torch.arange(c, device=dev) % 7).half()
                    self.acts[k] = torch.randn(self.local_samples, SEQLEN, c, device=dev, generator=gen,
                                               dtype=torch.float16) * chan
        torch.cuda.synchronize()
        self.phase_ms = {"hessian": 0.0, "solve": 0.0, "pack": 0.0, "exchange": 0.0}
        self.flush_events = []

    def act_of(self, gi, lin):
        n, _, _, key = lin
        return self.acts[(gi, n if self.args.no_shared_inputs else key)]

    def step(self, record):
        G, gmod, args, dev = self.G, self.gmod, self.args, self.dev
        wl = self.wl
        ev = lambda: torch.cuda.Event(enable_timing=True)
        marks = []
        out = {}
        for gi, g in enumerate(self.groups):
            solvers = []
            for lin in g:
                n, r, c, _ = lin
                # (no random init of a weight that is replaced at once: 0.24 ms of RNG kernels per step)
                m = torch.nn.utils.skip_init(torch.nn.Linear, c, r, bias=False, device=dev, dtype=torch.float16)
                m.weight.data = self.weights[n].clone()
                s = G.GPTQ(m)
                s.quantizer = G.Quantizer()
                s.quantizer.configure(wl["bits"], perchannel=True, sym=False, mse=False)
                solvers.append(s)
            # 1. Hessians, in the order the reference's hooks fire: per calibration sample, every Linear of the group
            e0, e1, e2, e3 = ev(), ev(), ev(), ev()
            gmod.FLUSH_EVENTS = self.flush_events if record else None
            e0.record()
            for j in range(self.local_samples):
                for s, lin in zip(solvers, g):
                    s.add_batch(self.act_of(gi, lin)[j:j + 1], None)
            if not gmod.LAZY_HESSIANS or self.dist_on:
                gmod.flush_pending()
            e1.record()
            gmod.FLUSH_EVENTS = None
            # 2. solve (opt.py:189-214)
            kw = dict(blocksize=128, percdamp=0.01, groupsize=wl["groupsize"], actorder=wl["actorder"],
                      static_groups=wl["static_groups"])
            if self.dist_on:
                from gptq_amd import parallel as par
                packed = par.fasterquant_sharded(solvers, bits=wl["bits"], group=self.group, timings=self.phase_ms if record else None, **kw)
                e2.record()
                for (n, _, _, _), s, p in zip(g, solvers, packed):
                    out[n] = p
                    s.free()
                e3.record()
            else:
                if args.serial_solve:
                    for s in solvers:
                        s.fasterquant(**kw)
                else:
                    G.fasterquant_many(solvers, max_concurrent=args.solve_streams or None, **kw)
                e2.record()
                # 3. pack
                for (n, _, _, _), s in zip(g, solvers):
                    out[n] = (G.pack_codes(s.codes, wl["bits"]), s.group_scale, s.group_zero)
                    s.free()
                e3.record()
            marks.append((e0, e1, e2, e3))
        if record:
            torch.cuda.synchronize()
            for e0, e1, e2, e3 in marks:
                self.phase_ms["hessian"] += e0.elapsed_time(e1)
                self.phase_ms["solve"] += e1.elapsed_time(e2)
                self.phase_ms["pack"] += e2.elapsed_time(e3)
        return out

    def run(self, warmup, steps, barrier):
        for _ in range(warmup):
            self.step(False)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step(True)
        barrier()
        return time.perf_counter() - t0

    def roofline(self, steps):
        """Dominant kernel at its dominant launch shape: the Hessian of the widest Linear, `hessian_defer` samples of 2048
        tokens per launch.  Algorithmic flops per launch = samples * S * C^2 (the symmetric half actually needed);
        duration = HIP events around each such launch on the launching stream."""
        per_launch = max(1, self.args.hessian_defer)
        cmax = max(c for g in self.groups for (_, _, c, _) in g)
        fe = self.flush_events
        hess_flops = sum(nslab * float(SEQLEN) * sum(float(c) * c for c in Cs) for (Cs, nslab, a, b) in fe)
        sel = [(Cs, a.elapsed_time(b)) for (Cs, nslab, a, b) in fe if max(Cs) == cmax and nslab == per_launch]
        n_launch = len(sel)
        launch_ms = sum(d for _, d in sel) / max(n_launch, 1)
        shape_cs = sel[0][0] if sel else [cmax]
        flops_launch = per_launch * float(SEQLEN) * sum(float(c) * c for c in shape_cs)
        achieved = flops_launch / (launch_ms / 1e3) / 1e12 if launch_ms > 0 else 0.0
        traffic, traffic_src = None, None
        for fn in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
            if fn.endswith("_pmc.json") and "hessian" in fn:
                j = json.load(open(os.path.join(ROOT, "profiles", fn)))
                if j.get("C") == shape_cs[0] and len(shape_cs) == 1 and j.get("samples_per_launch") == per_launch:
                    traffic, traffic_src = j["hbm_bytes_per_launch"], "profiles/" + fn
                    break
        hess_ms = self.phase_ms["hessian"]
        return {
            "kernel": "hessian16_big16_kernel<f16> + hessian16_big16_fixup (v_mfma_f32_16x16x32_f16 SYRK, 256x256 upper-triangle "
                      "tiles, K-split last round, fp32 accumulate)",
            "launch_shape": f"Hessian of C = {sorted(shape_cs, reverse=True)}, {per_launch} samples x {SEQLEN} tokens per launch",
            "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_F16_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_flops_per_launch": flops_launch,
            "algorithmic_bytes_per_launch": per_launch * SEQLEN * shape_cs[0] * 2 + float(shape_cs[0]) ** 2 * 4,
            "avg_launch_ms": round(launch_ms, 4), "launches": n_launch,
            "all_hessian_launches_tflops": round(hess_flops / (hess_ms / 1e3) / 1e12, 2) if hess_ms else None,
        }


def main():
    args = parse()
    from gptq_amd.build import ensure_built
    ensure_built()                     # (before anything touches the GPU; a no-op when the library is there)
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
    # (ranks then share devices and the exchanges are staged through host memory)
    backend = os.environ.get("GPTQ_BENCH_BACKEND", "nccl")
    dev = torch.device("cuda", local_rank % max(ndev, 1))
    # GPTQ_BENCH_FORCE_DIST=1: take the N > 1 code path (process group, all-reduce of H, all-gather of packed rows) with
    # whatever world size there is -- with ONE rank this rehearses the RCCL calls on a 1-GPU box
    dist_on = world > 1 or os.environ.get("GPTQ_BENCH_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(dev)

    import gptq_amd.gptq as gmod
    from gptq_amd import _lib
    _lib.load()
    gmod.VERBOSE = False
    gmod.HESSIAN_DEFER = args.hessian_defer
    gmod.LAZY_HESSIANS = not args.no_lazy_hessians
    gmod.SHARE_INPUT_HESSIANS = not args.no_shared_inputs

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    blk = Block(args.workload, args, dev, rank, world, dist_on=dist_on)
    if rank == 0:
        log(f"{args.workload}: inputs resident ({blk.local_samples} samples on this rank); warmup {args.warmup}, steps {args.steps}")
    elapsed = blk.run(args.warmup, args.steps, barrier)
    if dist_on:
        t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        log(f"timed region done: {elapsed:.3f} s for {args.steps} steps")
    ms_per_step = elapsed / args.steps * 1e3
    value = blk.params / (elapsed / args.steps) / 1e6
    wl = blk.wl
    out = {
        "metric": "Mparams/sec quantized (4bit)", "value": round(value, 2), "unit": "Mparams/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32 (fp16 weights and activations in, fp32 accumulate and solve)", "data": "synthetic",
        "config": {"workload": wl["desc"] + "; blocksize 128, percdamp 0.01", "workload_key": args.workload,
                   "bits": wl["bits"], "groupsize": wl["groupsize"], "actorder": wl["actorder"],
                   "params_per_step": blk.params, "nsamples": args.nsamples, "seqlen": SEQLEN,
                   "hessian_defer": args.hessian_defer,
                   "scope": "add_batch x nsamples + fasterquant + pack for every Linear of the block, group after group",
                   "calibration_inputs": "private tensor per Linear" if args.no_shared_inputs else
                                         "Linears fed one tensor in the model (q/k/v; gate/up) are fed one tensor: their common Hessian is accumulated once",
                   "parallelism": "1 GPU" if world == 1 else
                                  f"{world} GPUs: calibration samples data-parallel, all-reduce of H, rows of W sharded, all-gather of packed rows"},
    }
    if rank == 0:
        steps = args.steps
        out["roofline"] = blk.roofline(steps)
        solve_ms = (blk.phase_ms["solve"] + blk.phase_ms["pack"]) / steps
        out["phases"] = {k: round(v / steps, 3) for k, v in blk.phase_ms.items()}
        out["phases"]["solve_only_mparams_per_s"] = round(blk.params / (solve_ms / 1e3) / 1e6, 1) if solve_ms > 0 else None
        if world == 1 and args.project_gpus > 1:
            out["projection"] = projection(blk, args.project_gpus, blk.phase_ms["hessian"] / steps, dev)
    del blk
    torch.cuda.empty_cache()
    if world == 1:
        cpu = None
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(args.workload, WORKLOADS[args.workload], args.nsamples)
            out["cpu_baseline"] = cpu
        other = {"llama7b": "opt6.7b", "opt6.7b": "llama7b"}.get(args.workload)
        if other and not args.no_also:
            b2 = Block(other, args, dev, 0, 1)
            log(f"also: {other}, {args.also_steps} steps")
            el = b2.run(1, args.also_steps, barrier)
            entry = {"workload": b2.wl["desc"], "workload_key": other, "steps": args.also_steps,
                     "ms_per_step": round(el / args.also_steps * 1e3, 3),
                     "value": round(b2.params / (el / args.also_steps) / 1e6, 2), "unit": "Mparams/s",
                     "phases": {k: round(v / args.also_steps, 3) for k, v in b2.phase_ms.items()},
                     "roofline": b2.roofline(args.also_steps)}
            sm = (b2.phase_ms["solve"] + b2.phase_ms["pack"]) / args.also_steps
            entry["phases"]["solve_only_mparams_per_s"] = round(b2.params / (sm / 1e3) / 1e6, 1) if sm > 0 else None
            del b2
            torch.cuda.empty_cache()
            if not args.no_cpu_baseline:
                entry["cpu_baseline"] = cpu_baseline(other, WORKLOADS[other], args.nsamples)
                c = entry["cpu_baseline"]
                entry["gpu_over_cpu"] = {"whole_path": round(entry["value"] / c["value"], 1),
                                         "solve_only": round(entry["phases"]["solve_only_mparams_per_s"] / c["solve_only_mparams_per_s"], 1)}
            out["also"] = [entry]
        if cpu is not None:
            out["gpu_over_cpu"] = {"whole_path": round(out["value"] / cpu["value"], 1),
                                   "solve_only": round(out["phases"]["solve_only_mparams_per_s"] / cpu["solve_only_mparams_per_s"], 1)}
    if rank == 0 and world == 1 and not args.no_also:
        # the library's SAFE defaults (what a drop-in caller gets without opting in): add_batch launches its update before it
        # returns (HESSIAN_DEFER = 1) and every Linear folds its inputs as the hooks fire (no lazy Hessians)
        gmod.HESSIAN_DEFER, gmod.LAZY_HESSIANS = 1, False
        b3 = Block(args.workload, args, dev, 0, 1)
        el = b3.run(1, args.also_steps, barrier)
        out["default_mode"] = {"hessian_defer": 1, "lazy_hessians": False, "stage_inputs": int(gmod.STAGE_INPUTS),
                               "steps": args.also_steps,
                               "ms_per_step": round(el / args.also_steps * 1e3, 3),
                               "value": round(b3.params / (el / args.also_steps) / 1e6, 2), "unit": "Mparams/s",
                               "phases": {k: round(v / args.also_steps, 3) for k, v in b3.phase_ms.items()}}
        del b3
        torch.cuda.empty_cache()
        gmod.HESSIAN_DEFER, gmod.LAZY_HESSIANS = args.hessian_defer, not args.no_lazy_hessians
    e2e_blocks = args.end_to_end if args.end_to_end >= 0 else (2 if (world == 1 and args.workload == "llama7b") else 0)
    if rank == 0 and world == 1 and e2e_blocks > 0:
        gmod.HESSIAN_DEFER = 1
        out["end_to_end"] = end_to_end(dev, e2e_blocks, args.nsamples, args.hessian_defer)
    if rank == 0 and world == 1 and args.ppl_proxy:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_gpu_driver import run_ppl_proxy
        gmod.HESSIAN_DEFER = 1
        r = run_ppl_proxy(dev)
        out["ppl_proxy"] = {"model": "OPT-125m architecture, random init (seed 0), vocab 2048, 32 x 2048 synthetic calibration tokens, 4-bit",
                            "ppl_fp": r["ppl_fp"], "ppl_rtn4": r["ppl_rtn4"], "ppl_gptq4": r["ppl_gptq4"],
                            "reference": r["ref"], "abs_delta_ppl_gptq4": abs(r["ppl_gptq4"] - r["ref"]["ppl_gptq4"]),
                            "note": "reference numbers = /root/reference opt_sequential + opt_eval on CPU (oracle/gen_golden_opt125m.py); proxy, not Wiki2"}
    if rank == 0:
        sys.stdout.flush()
        with os.fdopen(json_fd, "w") as real_stdout:
            real_stdout.write(json.dumps(out) + "\n")
    if dist_on:
        dist.destroy_process_group()


def projection(blk, n_gpus, hessian_ms, dev):
    """What `gptq_amd.parallel.fasterquant_sharded` would take on n_gpus, from pieces MEASURED on this one GPU (no 8-GPU
    node was available to the builder; the driver's SCALE run is the real number).  Per hooked group and distinct
    Hessian (bundle): the solve of its stacked rows at R and at R / 2 rows gives T(R) = chain + rows * R; the chain (it
    depends on H only) is replicated on the bundle's ranks, the rows are split (plan_rows).  Hessians: samples are
    sharded, so hessian_ms / n; exchange: ring all-reduce of the upper trapezoids and direct all-gather of packed rows
    over xGMI at 300 GB/s effective per GPU (7 links x 153 GB/s nominal, MI355X guide)."""
    import gptq_amd
    from gptq_amd import parallel as par
    G, wl = gptq_amd, blk.wl
    kw = dict(blocksize=128, percdamp=0.01, groupsize=wl["groupsize"], actorder=wl["actorder"], static_groups=wl["static_groups"])
    total, detail, exch = hessian_ms / n_gpus, [], 0.0
    for g in blk.groups:
        bundles = {}
        for (n, r, c, key) in g:
            bundles.setdefault((key, c), []).append((n, r))
        shapes, fits = [], []
        for (key, c), members in bundles.items():
            R = sum(r for _, r in members)
            gen = torch.Generator(device=dev).manual_seed(7)
            X = torch.randn(2048, c, device=dev, generator=gen, dtype=torch.float16) * (1 + torch.arange(c, device=dev) % 7).half()
            t = {}
            for rows in (R, max(128, R // 2 // 128 * 128)):
                best = 1e9
                for _ in range(2):
                    m = torch.nn.Linear(c, rows, bias=False, device=dev, dtype=torch.float16)
                    m.weight.data = (torch.randn(rows, c, device=dev, generator=gen) * 0.02).half()
                    s = G.GPTQ(m)
                    s.quantizer = G.Quantizer(); s.quantizer.configure(wl["bits"], perchannel=True, sym=False, mse=False)
                    s.add_batch(X.unsqueeze(0), None); s.add_batch(X.flip(0).unsqueeze(0), None)
                    G.gptq.flush_pending()
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    s.fasterquant(**kw)
                    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) * 1e3)
                    s.free(); del m, s
                t[rows] = best
            (R1, T1), (R2, T2) = sorted(t.items(), reverse=True)
            per_row = max(0.0, (T1 - T2) / (R1 - R2))
            fits.append((max(0.0, T1 - per_row * R1), per_row))
            shapes.append((c, R))
            del X
        plan = par.plan_rows(shapes, n_gpus)
        load = [0.0] * n_gpus
        shard_note = []
        for bi, ((chain, per_row), slabs) in enumerate(zip(fits, plan)):
            k, (c, R) = len(slabs), shapes[bi]
            if par.SHARD_CHOL and k > 1 and c >= par.SHARD_CHOL_MIN_C and c % 128 == 0:
                # the factorization itself is spread over the bundle's k ranks (parallel.rfactor_sharded): its rank-512
                # updates (timed here call by call on this one GPU) divide by k, its serial panels do not, and every
                # outer panel is broadcast once (C^2 / 2 floats in all, at 300 GB/s)
                g_ms = _time_chol_updates(c, dev)
                bcast_ms = (c * c / 2 * 4) / 300e9 * 1e3
                shard_note.append({"C": c, "ranks": k, "replicated_chain_ms": round(chain, 2), "far_updates_ms": round(g_ms, 2),
                                   "sharded_chain_ms": round(chain - g_ms + g_ms / k + bcast_ms, 2), "broadcast_ms": round(bcast_ms, 2)})
                chain = chain - g_ms + g_ms / k + bcast_ms
            for (rk, a, e) in slabs:
                load[rk] += chain + per_row * (e - a)
        trap = sum(par.tri_numel(c) * 4 for c, _ in shapes)
        packed = sum(c * R * wl["bits"] // 8 for c, R in shapes)
        ex = (2 * (n_gpus - 1) / n_gpus * trap + (n_gpus - 1) / n_gpus * packed) / 300e9 * 1e3 if n_gpus > 1 else 0.0
        total += max(load) + ex
        exch += ex
        detail.append({"bundles": [{"C": c, "rows": R, "chain_ms": round(f[0], 2), "ms_per_1k_rows": round(f[1] * 1e3, 3),
                                    "ranks": len(sl)} for (c, R), f, sl in zip(shapes, fits, plan)],
                       "sharded_factorizations": shard_note,
                       "solve_ms": round(max(load), 2), "exchange_ms": round(ex, 2)})
    return {"n_gpus": n_gpus, "ms_per_step": round(total, 2), "mparams_per_s": round(blk.params / total / 1e3, 1),
            "hessian_ms": round(hessian_ms / n_gpus, 2), "exchange_ms": round(exch, 2), "groups": detail,
            "basis": "measured on 1 GPU: per distinct Hessian the solve at R and R/2 rows (rows split by parallel.plan_rows; the "
                     "chain replicated, or -- bundles on several ranks -- its rank-512 updates, timed call by call, divided by "
                     "the ranks + one broadcast per outer panel), Hessian time / n (samples sharded), exchanges at 300 GB/s per "
                     "GPU; NOT a hardware number"}


def _time_chol_updates(C, dev):
    """Milliseconds the rank-512 updates of ONE factorization of width C take on this GPU: the part of the chain that
    gptq_amd.parallel.rfactor_sharded divides among the ranks of a bundle (gptq_chol_update, timed call by call)."""
    from gptq_amd import _lib
    lib = _lib.load()
    gen = torch.Generator(device=dev).manual_seed(3)
    X = torch.randn(2048, C, device=dev, generator=gen)
    H = (X.t() @ X) * (2.0 / 2048)
    H.diagonal().add_(1.0)
    nbytes = lib.gptq_hinv_workspace_bytes(C)
    ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    info = torch.zeros(1, device=dev, dtype=torch.int32)
    st = _lib.stream(dev)
    nblk = C // 128
    evs = []
    _lib.call("gptq_chol_begin", _lib.ptr(H), H.stride(0), C, 0.01, None, _lib.ptr(info), _lib.ptr(ws), nbytes, st)
    for p0 in range(0, nblk, 4):
        _lib.call("gptq_chol_panel", _lib.ptr(ws), C, p0, _lib.ptr(info), st)
        if p0 + 4 < nblk:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            _lib.call("gptq_chol_update", _lib.ptr(ws), C, p0, p0 + 4, nblk, st)
            b.record()
            evs.append((a, b))
    torch.cuda.synchronize()
    assert int(info.item()) == 0
    return sum(a.elapsed_time(b) for a, b in evs)


def end_to_end(dev, blocks, nsamples, hessian_defer):
    """SURVEY 8(d) second scope, the reference's own `full quantization time` (opt.py:686-691, llama.py:31-207): capture of
    the first block's inputs + per block {hooked calibration forwards of every true-sequential group, fasterquant, the
    forward with quantized weights, block upload / download}.  Llama-7B architecture (hidden 4096, ffn 11008, 32 heads,
    vocab 32000) with `blocks` decoder blocks, random init (no checkpoints offline), 128 x 2048 synthetic tokens, 4-bit,
    --act-order --true-sequential.  Run twice: blocks moved synchronously like the reference, and prefetched."""
    from transformers import LlamaConfig, LlamaForCausalLM
    from gptq_amd.sequential import QuantArgs, quantize_sequential
    cfg = LlamaConfig(vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=blocks,
                      num_attention_heads=32, num_key_value_heads=32, max_position_embeddings=SEQLEN)
    t0 = time.perf_counter()
    # random init without the CPU's normal_() over a billion parameters (20 s for two blocks): meta construction, weights
    # drawn on the GPU (N(0, 0.02) like HF's initializer_range; norms = 1), rotary buffers rebuilt on the host
    with torch.device("meta"):
        model = LlamaForCausalLM(cfg).half()
    model = model.to_empty(device="cpu").eval()
    gen = torch.Generator(device=dev).manual_seed(0)
    for _, p in model.named_parameters():
        if p.dim() >= 2:
            p.data.copy_(torch.randn(p.shape, device=dev, dtype=torch.float16, generator=gen) * 0.02)
        else:
            p.data.fill_(1.0)
    model.model.rotary_emb = type(model.model.rotary_emb)(config=cfg)
    model.seqlen = SEQLEN
    log(f"end-to-end: {blocks}-block Llama-7B-architecture model built in {time.perf_counter() - t0:.1f} s")
    saved = {k: v.clone() for k, v in model.state_dict().items()}
    gen = torch.Generator().manual_seed(1)
    calib = [(torch.randint(0, cfg.vocab_size, (1, SEQLEN), generator=gen), None) for _ in range(nsamples)]
    params = sum(p.numel() for layer in model.model.layers for n, p in layer.named_parameters() if p.dim() == 2)
    out = {"model": f"Llama-7B architecture, {blocks} decoder blocks ({params / 1e6:.0f} M Linear params), random init, "
                    f"{nsamples} x {SEQLEN} synthetic tokens, 4-bit --act-order --true-sequential",
           "scope": "capture + per block: 4 hooked calibration passes (one per true-sequential group) + fasterquant + "
                    "forward with quantized weights + block transfer (opt.py:686-691)"}
    for key, prefetch, fbatch in (("reference_like", False, 1), ("driver_defaults", True, 1), ("forward_batch_8", True, 8)):
        model.load_state_dict(saved)
        tm = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        # first run: the reference's behaviour (blocks moved on the compute stream, full hooked passes, every Linear recomputed
        # in every pass); second: this driver's defaults (copy-stream prefetch / download, hooked passes left once the
        # group's hooks have fired, outputs of solved Linears kept per sample);
        # third: the defaults with 8 calibration samples per block forward (QuantArgs.forward_batch, opt-in)
        quantize_sequential(model, calib, dev, QuantArgs(wbits=4, nsamples=nsamples, act_order=True, true_sequential=True,
                                                         hessian_defer=hessian_defer if fbatch == 1 else max(1, hessian_defer // fbatch),
                                                         prefetch_blocks=prefetch, early_exit=prefetch, forward_batch=fbatch,
                                                         cache_outputs=prefetch),
                            timings=tm)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        per = lambda k: round(tm.get(k, 0.0) / blocks, 2)
        out[key] = {
            "prefetch_blocks": prefetch, "early_exit_of_hooked_passes": prefetch, "cached_linear_outputs": prefetch,
            "forward_batch": fbatch,
            "wall_s": round(wall, 3), "s_per_block": round(wall / blocks, 3),
            "mparams_per_s": round(params / wall / 1e6, 2),
            "gpu_ms_per_block": {"forward_hooked_incl_hessian": per("forward_hooked"), "hessian": per("hessian"),
                                 "forwards_alone": round(per("forward_hooked") - per("hessian") + per("forward_final"), 2),
                                 "solve": per("solve"), "forward_final": per("forward_final"), "transfer_waited": per("transfer")}}
        log(f"end-to-end ({key}): {wall:.2f} s for {blocks} blocks")
    del model, saved
    torch.cuda.empty_cache()
    return out


def cpu_baseline(name, wl, nsamples):
    """The oracle (the reference's algorithm restated on torch CPU fp32, bit-identical to the reference on the golden
    vectors) on a bounded sample of THIS workload's shapes: the square attention projection (median of 3 solves) and
    the widest Linear (one solve), same flags as the GPU run; add_batch timed on 2 calls per width and scaled to
    nsamples (the full 128 x add_batch of the widest Linear alone is minutes of CPU time).  Mparams/s over the
    sample = sample params / (scaled Hessian time + solve time + pack time)."""
    from oracle import gptq_oracle as O
    # the box's CPU share, not the host's core count (oversubscribing MKL stalls for minutes)
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    lins = [l for g in wl["groups"] for l in g]
    sq = next(l for l in lins if l[1] == l[2])
    wide = max(lins, key=lambda l: (l[2], l[1]))
    bits, gs = wl["bits"], wl["groupsize"]
    log(f"cpu baseline ({name}): {sq[1]}x{sq[2]} x3 and {wide[1]}x{wide[2]} x1 on {torch.get_num_threads()} threads ...")
    gen = torch.Generator().manual_seed(0)
    t_h, t_s, t_p, parts = 0.0, 0.0, 0.0, []
    for (n, R, C, _), reps in ((sq, 3), (wide, 1)):
        W = (torch.randn(R, C, generator=gen) * 0.02).half()
        chan = (1 + torch.arange(C) % 7).float()
        H = torch.zeros(C, C)
        cnt, th = 0, []
        for k in range(3):                                      # first call warms MKL up
            x = (torch.randn(1, SEQLEN, C, generator=gen) * chan).half()
            t0 = time.perf_counter()
            cnt = O.hessian_add_batch(H, cnt, x)
            th.append(time.perf_counter() - t0)
        hess = sum(th[1:]) / 2 * nsamples
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            r = O.fasterquant(W, H.clone(), bits=bits, sym=False, blocksize=128, percdamp=0.01, groupsize=gs,
                              actorder=wl["actorder"], static_groups=wl["static_groups"])
            ts.append(time.perf_counter() - t0)
            log(f"cpu baseline: {n} {R}x{C} solve {ts[-1]:.1f} s")
        solve = sorted(ts)[len(ts) // 2]
        t0 = time.perf_counter()
        (O.pack4 if bits == 4 else O.pack3)(r.codes.t().contiguous().numpy().astype("uint32"))
        pk = time.perf_counter() - t0
        t_h += hess; t_s += solve; t_p += pk
        parts.append(f"{n} {R}x{C}: add_batch {sum(th[1:]) / 2 * 1e3:.0f} ms/call x {nsamples} = {hess:.1f} s (scaled from 2 calls), "
                     f"fasterquant {solve:.1f} s ({'median of 3' if reps == 3 else 'one run'}), pack {pk:.1f} s")
    params = sq[1] * sq[2] + wide[1] * wide[2]
    return {"value": round(params / (t_h + t_s + t_p) / 1e6, 4), "unit": "Mparams/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{name}: " + "; ".join(parts) + f"; {bits}-bit, groupsize {gs}, act-order {wl['actorder']}",
            "solve_only_mparams_per_s": round(params / t_s / 1e6, 3)}


if __name__ == "__main__":
    main()
