"""bench.py - image-pairs/sec of the RoMa match() hot path on MI355X.

Workload (BASELINE.json metric / configs[2]; reference timing script
tests/test_roma_upsample_inference_time.py:7-47): roma_outdoor, coarse 560 -> upsample 864,
batch = 8 pairs per GPU, symmetric, synthetic N(0,1) images and seeded synthetic weights (no
pretrained weights / datasets offline).  Precision = the arithmetic that timing script really
runs (--dtype mixed, the default): its amp_dtype=bfloat16 reaches DINOv2 only
(roma_models.py:183-188), VGG / decoder / refiners autocast to float16 (encoders.py:7,
matcher.py:46,341) - ROMA_MIXED.  All-bfloat16 (--dtype bf16) is timed under other_configs.  A "step" = one match() over one batch,
inputs already resident in HBM.  N > 1: one process per GPU, pairs sharded 8 per GPU (weak
scaling), the only collective is the RCCL gather of the results.

    python bench.py                                           # N = 1, 50 timed steps after 10 warm-ups
    python bench.py --gpus 8 --steps 20 --warmup 3            # re-executes itself under torch.distributed.run
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 20 --warmup 3
    python bench.py --config coarse --batch 1                 # BASELINE config 2 (coarse-only 560, B = 1)
    python bench.py --gpus 2 --dry                            # CPU / gloo: launch + shard + gather logic only

Rank 0 prints ONE JSON line.  Besides the driver contract it carries
  roofline      dominant kernel: algorithmic FLOP/s (or B/s) from per-launch HIP events on the launch stream
  kernels       every instrumented kernel (GEMMs, attention, local correlation, grid-sample warp, ...)
  parity        the timed configuration's outputs against the committed reference golden (tools/parity_metrics.py)
  cpu_baseline  the CPU oracle on the host cores: 1 warm-up + 3 timed calls, median
"""
import argparse
import ctypes as C
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}  # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
PMC_SUMMARIES = ("profiles/r06_pmc_summary.json", "profiles/r05_pmc_summary.json", "profiles/r04_pmc_summary.json", "profiles/r03_pmc_summary.json", "profiles/r02_pmc_summary.json", "profiles/r01_pmc_summary.json")


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (tools/pmc_round.sh: separate
    --pmc FETCH_SIZE / WRITE_SIZE runs of this same bench; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
    wide coalesced reads on gfx950).  Returns (bytes or None, source file or None)."""
    # "gemm_kernel<bf16,f32,2,4,4,2,dense>" -> "void roma::gemm_kernel<unsigned short, float, 2, 4, 4, 2, false>"
    base, _, targs = kernel.partition("<")
    targs = targs.rstrip(">").split(",") if targs else []
    conv = {"bf16": "unsigned short", "f16": "unsigned short", "f32": "float", "dense": "false", "conv3x3": "true"}
    want = "void roma::" + base + ("<" + ", ".join(conv.get(t, t) for t in targs) + ">" if targs else "")
    if base == "gemm8p_kernel" and len(targs) == 4:    # profile scope <in,out,form,epilogue> -> <TOUT, CONV, EPI, DMAMF>
        epi = {"none": 0, "relu": 1, "gelu": 2, "res_bf16": 3, "qkv": 4}[targs[3]]
        want = f"void roma::gemm8p_kernel<{conv[targs[1]]}, {conv[targs[2]]}, {epi}, false"  # (+ ", SCHED, ABL>" since round 4)
    elif base == "gemm6p_kernel" and len(targs) == 4:  # -> <TOUT, ACT>
        act = {"none": 0, "relu": 1}[targs[3]]
        want = f"void roma::gemm6p_kernel<{conv[targs[1]]}, {act}>"
    elif base == "ws1x1_kernel" and len(targs) == 2:   # -> <ACT>
        want = "void roma::ws1x1_kernel<%d>" % {"none": 0, "relu": 1}[targs[1]]
    elif base == "conv3x3_patch_kernel":               # not a template: the profile scope only names the storage format
        want = "roma::conv3x3_patch_kernel"
    for rel in PMC_SUMMARIES:
        path = os.path.join(ROOT, rel)
        if not os.path.exists(path):
            continue
        pm = json.load(open(path))
        def pick(table):  # exact name, else the instantiation(s) that start with it (trailing template arguments added later)
            for exact in (want, want + ">"):
                if exact in table:
                    return table[exact]
            hits = [v for k, v in table.items() if k.startswith(want + ",")]
            if not hits:
                return None
            return {"launches": sum(h["launches"] for h in hits), "sum_kb": sum(h["sum_kb"] for h in hits)}
        f, w = pick(pm.get("FETCH_SIZE", {})), pick(pm.get("WRITE_SIZE", {}))
        if f and w and f["launches"]:
            return (2.0 * f["sum_kb"] / f["launches"] + w["sum_kb"] / w["launches"]) * 1024.0, rel
    return None, None


def respawn_distributed(args):
    """`python bench.py --gpus N` with N > 1 outside a launcher: run ourselves as N ranks (one per GPU)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL needs dmabuf IPC on this driver
    env["ROMA_BENCH_SPAWNED"] = "1"
    return subprocess.call(cmd, env=env)


class DryMatcher:
    """CPU stand-in used by --dry (no GPU, gloo): same output shapes and pair order as match(), no arithmetic."""

    def __init__(self, res):
        self.res = res

    def match(self, a, b, **kw):
        n = a.shape[0]
        tag = a.reshape(n, -1)[:, 0]
        warp = tag[:, None, None, None].expand(n, self.res, 2 * self.res, 4).contiguous()
        return warp, warp[..., 0].contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)   # SURVEY 8d protocol: >= 50 timed steps after 10 warm-ups (~6 s)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=None, help="image pairs per GPU per step (default 8; 1 for --config coarse)")
    ap.add_argument("--dtype", default="mixed", choices=["bf16", "f16", "f32", "mixed"],
                    help="mixed (default) = what the reference's bf16 timing script really computes: bf16 DINOv2 + binary16 "
                         "VGG / decoder / refiners (ROMA_MIXED) - the parity-bearing 16-bit mode; bf16 = one 16-bit format "
                         "everywhere; f16 = the reference's default amp_dtype (IEEE binary16 storage, libroma_hip_f16.so); "
                         "f32 = the exact parity mode")
    ap.add_argument("--config", default="full", choices=["full", "coarse"],
                    help="full = 560 -> 864 upsample path (the metric); coarse = coarse-only 560 (BASELINE config 2)")
    ap.add_argument("--coarse", type=int, default=560)
    ap.add_argument("--upsample", type=int, default=864)
    ap.add_argument("--streams", type=int, default=2, choices=[1, 2],
                    help="sub-batch HIP streams per GPU (2 = the library default: two half-batches, DESIGN.md section 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the extra legs of the default run: BASELINE configs 2 (coarse-only B = 1) and 5 (f32 B = 8, "
                         "indoor weights) and the f16 storage mode, each with its own roofline object")
    ap.add_argument("--cpu-baseline-reps", type=int, default=3)
    ap.add_argument("--dry", action="store_true", help="CPU-only launch-logic check: gloo backend, stand-in match()")
    ap.add_argument("--compact-gather", action="store_true",
                    help="N > 1: gather only the predicted warp channels + certainty, the root rebuilds the constant grid "
                         "channels (-40 %% of the root's ingress; byte-identical result, roma_amd/distributed.py)")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 1 if args.config == "coarse" else 8
    full = args.config == "full"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(respawn_distributed(args))

    import torch.distributed as dist
    from roma_amd import synthetic
    from roma_amd.distributed import gather_results

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.dry:
        dev = torch.device("cpu")
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("nccl", device_id=dev)  # RCCL on ROCm

    if args.dry:
        res = 16
        model = DryMatcher(res)
        inp = {"im_A": torch.full((args.batch, 3, 4, 4), float(rank)), "im_B": torch.zeros(args.batch, 3, 4, 4)}
        sd = dsd = None
    else:
        from roma_amd import _lib, roma_outdoor
        sd, dsd = synthetic.make_matcher_state_dict(0), synthetic.make_dinov2_state_dict(0)
        amp = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32, "mixed": torch.bfloat16}[args.dtype]
        model = roma_outdoor(device=dev, weights=sd, dinov2_weights=dsd, coarse_res=args.coarse, upsample_res=args.upsample,
                             amp_dtype=amp, symmetric=True, upsample_preds=full, max_batch=args.batch,
                             decoder_dtype=torch.float16 if args.dtype == "mixed" else None)
        model.dual_stream = args.streams == 2
        inp = {k: v.to(dev) for k, v in synthetic.make_inputs(args.batch, args.coarse, args.upsample if full else None,
                                                              seed=1 + rank).items()}
    n_pairs = args.batch * world
    kw = dict(im_A_high_res=inp["im_A_high_res"], im_B_high_res=inp["im_B_high_res"]) if (full and not args.dry) else {}

    pending = [None]  # N > 1: the result gather of step i runs on RCCL's stream under the match() of step i + 1
    gather_ms = []

    def drain():
        if pending[0] is None:
            return None
        t = time.perf_counter()
        res = pending[0].wait()
        gather_ms.append(1e3 * (time.perf_counter() - t))
        pending[0] = None
        return res

    def sync():
        if not args.dry:
            torch.cuda.synchronize()

    def step():
        warp, cert = model.match(inp["im_A"], inp["im_B"], **kw)
        if world > 1:
            drain()  # queued behind this step's kernels: the previous gather has had the whole match() to finish
            pending[0] = gather_results(warp, cert, n_pairs, async_op=True, compact_grid=args.compact_gather)
        return warp, cert

    if args.warmup == 0 and not args.dry:
        step()  # one-time work of the very first call (second-stream workspace allocation, function attributes) is never timed
    for _ in range(args.warmup):
        step()
    solo = None
    if world > 1:
        # the N = 1 figure of THIS binary on THIS rank's GPU, for the scaling check: the same steps with no gather at all
        # (every rank runs it at the same time, so host-side contention between the processes is included)
        drain()
        sync()
        ns = max(1, min(args.steps, 5))
        ts = time.perf_counter()
        for _ in range(ns):
            model.match(inp["im_A"], inp["im_B"], **kw)
        sync()
        solo = args.batch * ns / (time.perf_counter() - ts)
    if world > 1:
        drain()
        dist.barrier()
    sync()
    gather_ms.clear()
    evs = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if not args.dry:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()  # torch's current stream IS the stream roma_match launches on (matcher.py passes it down)
        out = step()
        if not args.dry:
            e1.record()
            evs.append((e0, e1))
    if world > 1:
        gathered = drain()  # every step's results are on rank 0 before the clock stops
        out = gathered if rank == 0 else out
    sync()
    if world > 1:
        dist.barrier()
        sync()
    dt = time.perf_counter() - t0
    dt_rank = dt
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        per_rank = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(per_rank, torch.tensor([n_pairs / world * args.steps / dt_rank], device=dev, dtype=torch.float64))
        per_rank = [float(x.item()) for x in per_rank]
        solo_all = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(solo_all, torch.tensor([solo], device=dev, dtype=torch.float64))
        solo_all = [float(x.item()) for x in solo_all]
    step_ms = [e0.elapsed_time(e1) for e0, e1 in evs]
    finite = bool(torch.isfinite(out[1]).all()) if out[1] is not None else True

    what = (f"{args.coarse}->{args.upsample}, symmetric, upsample_preds" if full else f"{args.coarse} coarse-only, symmetric")
    policy = {"mixed": "precision policy of the reference's bf16 timing script: bfloat16 DINOv2 + float16 autocast for VGG / decoder / "
                       "refiners (ROMA_MIXED), f32 accumulate",
              "bf16": "bfloat16 storage everywhere, f32 accumulate", "f16": "IEEE binary16 storage everywhere, f32 accumulate",
              "f32": "exact f32 (f32-input MFMA)"}[args.dtype]
    result = {
        "metric": "image-pairs/sec, roma_outdoor 560->864, batch=8 per GPU" if full else
                  "image-pairs/sec, roma_outdoor coarse-only 560, batch=1 (BASELINE config 2; not the headline metric)",
        "value": n_pairs * args.steps / dt,
        "unit": "image-pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        # the reference script's "bf16" run IS this mix (16-bit MFMA rate for both formats); --dtype bf16 = all-bfloat16
        "dtype": "bf16" if args.dtype == "mixed" else args.dtype, "data": "synthetic",
        "config": {"workload": f"roma_outdoor match() {what}, {args.batch} pairs/GPU/step, seeded synthetic weights + N(0,1) images; {policy}",
                   "precision_policy": args.dtype,
                   "global_batch": n_pairs, "streams_per_gpu": int(os.environ.get("ROMA_STREAMS", args.streams)),
                   "parallelism": (f"pairs sharded x{world}, {'gloo (dry run)' if args.dry else 'RCCL'} gather of results "
                                   "(step i's gather overlaps step i+1's match)") if world > 1 else "single GPU",
                   "outputs_finite": finite},
    }
    if step_ms:
        result["ms_per_step_median_hip_events"] = statistics.median(step_ms)
    if world > 1:
        result["rccl_ranks"] = world
        result["pairs_per_s_per_rank"] = per_rank
        result["ms_per_step_per_rank"] = [1e3 * args.batch / v for v in per_rank]
        # what the root receives per step: (world - 1) shards of warp [b, H, 2W, 4] + certainty [b, H, 2W] f32
        result["gather_bytes_per_step_into_rank0"] = int((world - 1) * (out[0][:args.batch].numel() // (2 if args.compact_gather else 1)
                                                                       + out[1][:args.batch].numel()) * 4) if rank == 0 else None
        result["compact_gather"] = bool(args.compact_gather)
        result["single_gpu_no_gather_pairs_per_s_per_rank"] = solo_all  # same binary, same GPUs, gather off (N = 1 yardstick)
        result["gather_overhead_frac"] = 1.0 - (n_pairs * args.steps / dt) / sum(solo_all)
        result["gather_wait_ms_rank0"] = {"median": statistics.median(gather_ms) if gather_ms else None,
                                          "last": gather_ms[-1] if gather_ms else None}
        result["launched_by"] = "bench.py self-spawn" if os.environ.get("ROMA_BENCH_SPAWNED") else "external launcher"
    if args.dry:
        if rank == 0 and world > 1:  # the gathered result must hold every rank's shard in pair order
            tags = out[0][:, 0, 0, 0].tolist()
            assert tags == [float(r) for r in range(world) for _ in range(args.batch)], tags
        result["dry"] = True
        if rank == 0:
            print(json.dumps(result))
        if world > 1:
            dist.destroy_process_group()
        return

    def instrumented_pass(mdl, run_step, nprof, two_streams):
        """roofline of the dominant kernel + the per-kernel table: per-launch HIP events on the launch stream in a separate
        instrumented pass.  Every kernel is timed owning the chip on the full-batch launch: the sub-batch stream split is
        switched off for this pass only."""
        lib = mdl._lib  # the library of this dtype (bf16 / f16 storage builds)
        mdl.dual_stream = False
        lib.roma_profile_enable(1)
        for _ in range(nprof):
            run_step()
        torch.cuda.synchronize()
        mdl.dual_stream = two_streams
        n = lib.roma_profile_report(None, 0)
        buf = C.create_string_buffer(int(n))
        lib.roma_profile_report(buf, n)
        lib.roma_profile_enable(0)
        prof = json.loads(buf.value.decode())
        tot_ms = sum(v["total_ms"] for v in prof.values())
        name, v = max(prof.items(), key=lambda kv: kv[1]["total_ms"])
        if v["unit"] == "flop":
            ach = v["work"] / (v["total_ms"] * 1e-3) / 1e12
            head = name.split(",")[0]  # "gemm6p_kernel<bf16", "attn_f32_kernel<64>", ...: the INPUT type names the MFMA rate
            peak = PEAK_TFLOPS["bf16" if ("bf16" in head or "f16" in head) else "f32"]  # one 16-bit rate for bf16 / f16
            roof = {"kernel": name, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak}
        else:
            ach = v["work"] / (v["total_ms"] * 1e-3) / 1e9
            roof = {"kernel": name, "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS}
        traffic, src = pmc_traffic(name)
        roof.update({"traffic": traffic,
                     "traffic_source": (f"{src}: committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench "
                                        "(tools/pmc_round.sh), not measured in this run") if src else None,
                     "work_counted": "algorithmic: un-padded M, N, K of the reference's layer (the padded channels of the "
                                     "refiner buffers are not counted)" if v["unit"] == "flop" else "algorithmic bytes (DESIGN.md section 4)",
                     "launches_per_step": v["calls"] / nprof, "avg_launch_ms": v["total_ms"] / v["calls"],
                     "share_of_instrumented_time": v["total_ms"] / tot_ms})
        if two_streams:
            roof["mode"] = ("instrumented pass with the sub-batch stream split off: full-batch launches, one kernel on the "
                            "chip at a time; the timed region overlaps two half-batch streams")
        kernels = {k: {"ms_per_step": x["total_ms"] / nprof, "calls_per_step": x["calls"] / nprof,
                       ("TFLOP/s" if x["unit"] == "flop" else "GB/s"):
                           x["work"] / (x["total_ms"] * 1e-3) / (1e12 if x["unit"] == "flop" else 1e9)}
                   for k, x in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])}
        return roof, kernels

    if rank == 0 and world == 1 and not args.no_roofline:
        result["roofline"], result["kernels"] = instrumented_pass(model, step, max(1, min(3, args.steps)), args.streams == 2)
        if full and args.dtype != "f32":
            # ---- the regime a TRAINED matcher produces: smooth (coherent) warps.  The random-weight benchmark model's coarse
            # matches are incoherent (neighbouring tokens point ~12 of 40 tokens apart), which sends every local-correlation
            # tile to the gather work list.  Here a smooth coarse match (identity + a low-frequency displacement) is injected
            # behind cls_to_flow_refine (roma_debug_inject) and the same instrumented pass is repeated: the local-correlation
            # and grid-sample-warp kernels below then run on the warps they were designed for.
            import numpy as np
            T = (args.coarse // 14) ** 2
            th = args.coarse // 14
            ys, xs = np.meshgrid(np.linspace(-1 + 1 / th, 1 - 1 / th, th), np.linspace(-1 + 1 / th, 1 - 1 / th, th), indexing="ij")
            gx = 0.92 * xs + 0.05 * np.sin(2.5 * ys) + 0.02
            gy = 0.95 * ys - 0.04 * np.cos(2.0 * xs) - 0.01
            flow = np.stack([gx, gy], -1).reshape(1, T, 2).astype(np.float32).repeat(2 * args.batch, 0)
            model.debug = True
            model.debug_inject("gm_flow16", flow)
            model.debug_inject("gm_cert16", np.full((2 * args.batch, T, 1), 2.0, np.float32))
            try:
                _, kc = instrumented_pass(model, step, 1, args.streams == 2)
            finally:
                model.debug = False
                model.debug_inject("gm_flow16", None)
                model.debug_inject("gm_cert16", None)
            result["kernels_coherent"] = {
                "what": "one instrumented match() with a smooth coarse match injected (identity + low-frequency displacement): "
                        "the warp regime of a trained matcher; only the warp-dependent kernels are listed",
                "kernels": {k: v for k, v in kc.items() if k.startswith("local_corr") or k.startswith("refiner_input")}}

    if rank == 0 and not args.no_parity:
        # ---- parity of the timed configuration (rank 0's shard = seeds 0 / 1) against the reference's own output
        gold = os.path.join(ROOT, "tests", "golden", "match_full8.npz" if full else "match_full_coarse.npz")
        default_cfg = args.coarse == 560 and (not full or args.upsample == 864) and args.batch == (8 if full else 1)
        if default_cfg and os.path.exists(gold):
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import numpy as np
            import parity_metrics as PM
            g = np.load(gold)
            tol = 1e-3

            def run(inject):
                model.debug = True
                model.debug_inject("gm_flow16", PM.nchw_to_tokens(g["gm_flow16"]) if inject else None)
                model.debug_inject("gm_cert16", PM.nchw_to_tokens(g["gm_cert16"]) if inject else None)
                w, c = model.match(inp["im_A"], inp["im_B"], **kw)
                torch.cuda.synchronize()
                own = model.debug_fetch("gm_flow16_own").reshape(-1, 1600, 2).copy()
                model.debug = False
                model.debug_inject("gm_flow16", None)
                model.debug_inject("gm_cert16", None)
                return w.cpu().numpy()[:, ::8, ::8], c.cpu().numpy()[:, ::8, ::8], own

            w, c, own = run(False)
            par = {"golden": os.path.relpath(gold, ROOT) + " (unmodified reference, CPU fp32, same seeds; 1/8 sub-sampled)",
                   "tolerance_f32": tol,
                   "coarse_argmax": PM.coarse_flips(own, PM.nchw_to_tokens(g["gm_flow16"]),
                                                    PM.nchw_to_tokens(g["cls16_top2gap"][:, None])),
                   "outputs": PM.output_errors(w, c, g["warp_sub"], g["cert_sub"], tol=tol)}
            if args.dtype != "f32":  # continuous part of the pipeline: the reference's coarse match injected
                w, c, _ = run(True)
                par["outputs_with_reference_coarse_match_injected"] = PM.output_errors(w, c, g["warp_sub"], g["cert_sub"], tol=tol)
            result["parity"] = par
        else:
            result["parity"] = None

    default_run = full and args.dtype == "mixed" and args.coarse == 560 and args.upsample == 864 and args.batch == 8
    if rank == 0 and world == 1 and default_run and not args.no_other_configs:
        # ---- the other single-GPU configurations of BASELINE.json, as nested objects of the same JSON line (they are NOT
        # the metric): each builds its own handle, is timed like the main loop (synchronise / K steps / synchronise) and
        # gets its own instrumented pass.
        import numpy as np
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import parity_metrics as PM

        def side_config(dtype, is_full, batch, steps, warmup, seed_w, seed_in, golden):
            amp_ = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32, "mixed": torch.bfloat16}[dtype]
            sd_, dsd_ = (sd, dsd) if seed_w == 0 else (synthetic.make_matcher_state_dict(seed_w), synthetic.make_dinov2_state_dict(seed_w))
            m_ = roma_outdoor(device=dev, weights=sd_, dinov2_weights=dsd_, coarse_res=560, upsample_res=864, amp_dtype=amp_,
                              symmetric=True, upsample_preds=is_full, max_batch=batch,
                              decoder_dtype=torch.float16 if dtype == "mixed" else None)
            i_ = {k: v.to(dev) for k, v in synthetic.make_inputs(batch, 560, 864 if is_full else None, seed=seed_in).items()}
            kw_ = dict(im_A_high_res=i_["im_A_high_res"], im_B_high_res=i_["im_B_high_res"]) if is_full else {}
            run = lambda: m_.match(i_["im_A"], i_["im_B"], **kw_)  # noqa: E731
            for _ in range(max(1, warmup)):
                run()
            torch.cuda.synchronize()
            t_ = time.perf_counter()
            for _ in range(steps):
                out_ = run()
            torch.cuda.synchronize()
            dt_ = time.perf_counter() - t_
            r_ = {"value": batch * steps / dt_, "unit": "image-pairs/s", "ms_per_step": 1e3 * dt_ / steps, "steps": steps,
                  "dtype": dtype, "batch": batch, "workload": "560 -> 864 full" if is_full else "560 coarse-only"}
            r_["roofline"], kern = instrumented_pass(m_, run, 1, batch >= 2)
            r_["kernels_top5"] = dict(list(kern.items())[:5])
            gpath = os.path.join(ROOT, "tests", "golden", golden)
            if os.path.exists(gpath):
                g_ = np.load(gpath)
                if dtype == "f32":
                    r_["parity"] = {"golden": "tests/golden/" + golden, "tolerance": 1e-3,
                                    "outputs": PM.output_errors(out_[0].cpu().numpy()[:, ::8, ::8], out_[1].cpu().numpy()[:, ::8, ::8],
                                                                g_["warp_sub"], g_["cert_sub"], tol=1e-3)}
                else:
                    m_.debug = True
                    m_.debug_inject("gm_flow16", PM.nchw_to_tokens(g_["gm_flow16"]))
                    m_.debug_inject("gm_cert16", PM.nchw_to_tokens(g_["gm_cert16"]))
                    wi, ci = run()
                    torch.cuda.synchronize()
                    m_.debug = False
                    m_.debug_inject("gm_flow16", None)
                    m_.debug_inject("gm_cert16", None)
                    r_["parity"] = {"golden": "tests/golden/" + golden, "outputs_with_reference_coarse_match_injected":
                                    PM.output_errors(wi.cpu().numpy()[:, ::8, ::8], ci.cpu().numpy()[:, ::8, ::8], g_["warp_sub"], g_["cert_sub"])}
            del m_
            torch.cuda.empty_cache()
            return r_

        result["other_configs"] = {
            "config2_coarse_only_b1_bf16": side_config("bf16", False, 1, 30, 5, 0, 1, "match_full_coarse.npz"),
            "config5_indoor_f32_b8": side_config("f32", True, 8, 3, 1, 2, 3, "match_full8_indoor.npz"),
            "f16_storage_b8 (the reference's default amp_dtype)": side_config("f16", True, 8, 10, 3, 0, 1, "match_full8.npz"),
            # one 16-bit format everywhere (narrower than what the reference computes: its VGG / decoder / refiners autocast
            # to float16 whatever amp_dtype is) - kept as a timed, parity-gated side line since round 6
            "all_bf16_storage_b8": side_config("bf16", True, 8, 10, 3, 0, 1, "match_full8.npz"),
        }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # ---- CPU baseline: the oracle (CPU restatement of the reference; the reference itself is not on the GPU box) on
        # the host cores, ONE symmetric pair of the same workload per call: 1 warm-up + N timed calls, median
        from oracle import roma_oracle
        torch.set_num_threads(min(os.cpu_count(), 32))  # MKL/oneDNN stop scaling (and regress) beyond ~32 threads at these sizes
        cin = synthetic.make_inputs(1, args.coarse, args.upsample if full else None, seed=1)
        ckw = dict(upsample_preds=full)
        times = []
        for i in range(1 + max(1, args.cpu_baseline_reps)):
            t0 = time.perf_counter()
            cpu_out = roma_oracle.match(cin["im_A"], cin["im_B"], sd, dsd, cin.get("im_A_high_res"), cin.get("im_B_high_res"), **ckw)
            if i > 0:
                times.append(time.perf_counter() - t0)
        med = statistics.median(times)
        checked = None
        gold1 = os.path.join(ROOT, "tests", "golden", "match_full.npz" if full else "match_full_coarse.npz")
        if args.coarse == 560 and (not full or args.upsample == 864) and os.path.exists(gold1):
            # the timed CPU leg is also a checked one: same seeds as the reference-generated golden (weights 0, inputs 1)
            import numpy as np
            g1 = np.load(gold1)
            checked = {"golden": os.path.relpath(gold1, ROOT),
                       "max_abs_warp": float(np.abs(cpu_out[0].numpy()[:, ::8, ::8] - g1["warp_sub"]).max()),
                       "max_abs_certainty": float(np.abs(cpu_out[1].numpy()[:, ::8, ::8] - g1["cert_sub"]).max())}
        result["cpu_baseline"] = {"value": 1.0 / med, "unit": "image-pairs/s", "cores": torch.get_num_threads(), "kind": "port",
                                  "checked_against_reference_golden": checked,
                                  "sample": f"1 symmetric pair {what}, fp32, torch CPU oracle: 1 warm-up + {len(times)} timed calls, "
                                            f"median {med:.1f} s (all: {', '.join(f'{t:.1f}' for t in times)})",
                                  "host_cores_available": os.cpu_count(),
                                  "why_not_all_cores": "more threads are SLOWER for this workload: the same call takes ~250 s with all "
                                                       "256 hardware threads against 21-28 s with 32 (MKL / oneDNN at these sizes; "
                                                       "measured in round 3, DESIGN.md section 5), so 32 is the fastest CPU "
                                                       "configuration, not a handicap"}

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
