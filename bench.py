"""bench.py - image-pairs/sec of the RoMa match() hot path on MI355X.

Workload (BASELINE.json metric / configs[2]; reference timing script
tests/test_roma_upsample_inference_time.py:7-47): roma_outdoor, coarse 560 -> upsample 864,
batch = 8 pairs per GPU, symmetric, bf16 compute, synthetic N(0,1) images and seeded synthetic
weights (no pretrained weights / datasets offline).  A "step" = one match() over one batch,
inputs already resident in HBM.  N > 1: one process per GPU (torch.distributed.run), pairs
sharded 8 per GPU (weak scaling), the only collective is the RCCL gather of the results.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 20 --warmup 3
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}  # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (tools/pmc_round.sh: separate
    --pmc FETCH_SIZE / WRITE_SIZE runs of this same bench; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
    wide coalesced reads on gfx950).  None when no summary matches."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
    if not os.path.exists(path):
        return None
    pm = json.load(open(path))
    # "gemm_kernel<bf16,f32,2,4,4,2,dense>" -> "void roma::gemm_kernel<unsigned short, float, 2, 4, 4, 2, false>"
    base, _, targs = kernel.partition("<")
    targs = targs.rstrip(">").split(",")
    conv = {"bf16": "unsigned short", "f32": "float", "dense": "false", "conv3x3": "true"}
    want = "void roma::" + base + "<" + ", ".join(conv.get(t, t) for t in targs) + ">"
    f, w = pm.get("FETCH_SIZE", {}).get(want), pm.get("WRITE_SIZE", {}).get(want)
    if not f or not w or not f["launches"]:
        return None
    return (2.0 * f["sum_kb"] / f["launches"] + w["sum_kb"] / w["launches"]) * 1024.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="image pairs per GPU per step")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--coarse", type=int, default=560)
    ap.add_argument("--upsample", type=int, default=864)
    ap.add_argument("--streams", type=int, default=1, choices=[1, 2],
                    help="sub-batch HIP streams per GPU (2 = experimental stream split, see DESIGN.md)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    import torch.distributed as dist
    from roma_amd import _lib, roma_outdoor, synthetic
    from roma_amd.distributed import gather_results

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)  # RCCL on ROCm

    sd, dsd = synthetic.make_matcher_state_dict(0), synthetic.make_dinov2_state_dict(0)
    amp = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model = roma_outdoor(device=dev, weights=sd, dinov2_weights=dsd, coarse_res=args.coarse, upsample_res=args.upsample,
                         amp_dtype=amp, symmetric=True, upsample_preds=True, max_batch=args.batch)
    model.dual_stream = args.streams == 2
    inp = {k: v.to(dev) for k, v in synthetic.make_inputs(args.batch, args.coarse, args.upsample, seed=1 + rank).items()}
    n_pairs = args.batch * world

    pending = [None]  # N > 1: the result gather of step i runs on RCCL's stream under the match() of step i + 1

    def drain():
        res, pending[0] = (pending[0].wait() if pending[0] is not None else None), None
        return res

    def step():
        warp, cert = model.match(inp["im_A"], inp["im_B"], im_A_high_res=inp["im_A_high_res"], im_B_high_res=inp["im_B_high_res"])
        if world > 1:
            drain()  # queued behind this step's kernels: the previous gather has had the whole match() to finish
            pending[0] = gather_results(warp, cert, n_pairs, async_op=True)
        return warp, cert

    for _ in range(args.warmup):
        step()
    if world > 1:
        drain()
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    if world > 1:
        gathered = drain()  # every step's results are on rank 0 before the clock stops
        out = gathered if rank == 0 else out
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    finite = bool(torch.isfinite(out[1]).all()) if out[1] is not None else True

    result = {
        "metric": "image-pairs/sec, roma_outdoor 560->864, batch=8 per GPU",
        "value": n_pairs * args.steps / dt,
        "unit": "image-pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"roma_outdoor match() {args.coarse}->{args.upsample}, symmetric, upsample_preds, "
                               f"{args.batch} pairs/GPU/step, seeded synthetic weights + N(0,1) images",
                   "global_batch": n_pairs, "streams_per_gpu": int(os.environ.get("ROMA_STREAMS", args.streams)), "parallelism": f"pairs sharded x{world}, RCCL gather of results (step i's gather overlaps step i+1's match)" if world > 1 else "single GPU",
                   "outputs_finite": finite},
    }

    if rank == 0 and world == 1 and not args.no_roofline:
        # ---- roofline of the dominant kernel: per-launch HIP events on the launch stream (separate instrumented pass)
        # every kernel is timed owning the chip on the full-batch launch: with --streams 2 the split is switched off for
        # this pass only
        lib = _lib.load()
        model.dual_stream = False
        lib.roma_profile_enable(1)
        nprof = max(1, min(3, args.steps))
        for _ in range(nprof):
            step()
        torch.cuda.synchronize()
        model.dual_stream = args.streams == 2
        n = lib.roma_profile_report(None, 0)
        buf = C.create_string_buffer(int(n))
        lib.roma_profile_report(buf, n)
        lib.roma_profile_enable(0)
        prof = json.loads(buf.value.decode())
        tot_ms = sum(v["total_ms"] for v in prof.values())
        dom = max(prof.items(), key=lambda kv: kv[1]["total_ms"])
        name, v = dom
        if v["unit"] == "flop":
            ach = v["work"] / (v["total_ms"] * 1e-3) / 1e12
            peak = PEAK_TFLOPS["bf16" if "bf16" in name.split(",")[0] else "f32"]
            roof = {"kernel": name, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak}
        else:
            ach = v["work"] / (v["total_ms"] * 1e-3) / 1e9
            roof = {"kernel": name, "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS}
        roof.update({"traffic": pmc_traffic(name), "launches_per_step": v["calls"] / nprof, "avg_launch_ms": v["total_ms"] / v["calls"],
                     "share_of_instrumented_time": v["total_ms"] / tot_ms})
        if args.streams == 2:
            roof["mode"] = ("instrumented pass with the sub-batch stream split off: full-batch launches, one kernel on the "
                            "chip at a time; the timed region overlaps two half-batch streams")
        result["roofline"] = roof
        result["kernels"] = {k: {"ms_per_step": x["total_ms"] / nprof, "calls_per_step": x["calls"] / nprof,
                                 ("TFLOP/s" if x["unit"] == "flop" else "GB/s"):
                                     x["work"] / (x["total_ms"] * 1e-3) / (1e12 if x["unit"] == "flop" else 1e9)}
                             for k, x in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # ---- CPU baseline: the oracle (CPU restatement of the reference) on the host cores, ONE pair of the same workload
        from oracle import roma_oracle
        torch.set_num_threads(min(os.cpu_count(), 32))  # MKL/oneDNN stop scaling (and regress) beyond ~32 threads at these sizes
        cin = synthetic.make_inputs(1, args.coarse, args.upsample, seed=1)
        t0 = time.perf_counter()
        roma_oracle.match(cin["im_A"], cin["im_B"], sd, dsd, cin["im_A_high_res"], cin["im_B_high_res"])
        cdt = time.perf_counter() - t0
        result["cpu_baseline"] = {"value": 1.0 / cdt, "unit": "image-pairs/s", "cores": torch.get_num_threads(), "kind": "port",
                                  "sample": f"1 symmetric pair {args.coarse}->{args.upsample}, fp32, torch CPU, no warm-up ({cdt:.1f} s)"}

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
