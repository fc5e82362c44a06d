"""Who runs next to whom: concurrency analysis of a rocprofv3 --kernel-trace CSV of the two-stream bench (tools/r04_final.sh).

    python tools/stream_overlap.py <..._kernel_trace.csv[.gz]>

For the last complete step of the trace (between the final_epilogue kernels of consecutive match() calls):
  * per HIP queue: number of kernels, sum of their durations, gaps between consecutive kernels;
  * wall-clock split: both queues inside a kernel / exactly one / none;
  * the same split by kernel class pairs (MFMA GEMM, attention, stencil, gather, GP chain, small rest): how much of the time a
    latency-bound phase of one sub-batch (the GP's blocked Cholesky: ~175 launches of 10..50 us on a few CUs) really runs under a
    throughput kernel of the other one;
  * per class: sum of durations in this trace, to set against the one-stream kernel statistics of the same visit
    (a persistent 256-workgroup GEMM that shares the chip with the other stream's GEMM takes ~2x as long per launch; a class
    whose two-stream sum is well above 2x half its one-stream sum is being stretched by waiting for CUs).
"""
import collections
import csv
import gzip
import io
import re
import sys


def klass(name):
    n = name
    if re.search(r"gemm8p|gemm6p|ws1x1|conv3x3_c64|conv3x3_c128|conv3x3_patch", n) or re.search(r"gemm_kernel<unsigned short", n):
        return "mfma_gemm"
    if "attn" in n:
        return "attention"
    if re.search(r"dwconv|refiner_block", n):
        return "stencil"
    if re.search(r"local_corr|refiner_input|refiner_out", n):
        return "gather"
    if re.search(r"gemm_kernel<float|chol_diag|chol_col|transpose_kernel|pad_identity|gp_basis|rownorm|copyBuffer|fillBuffer", n):
        return "gp_chain"
    return "rest"


def main(path):
    ev = []
    fh = io.TextIOWrapper(gzip.open(path, "rb")) if path.endswith(".gz") else open(path)
    for r in csv.DictReader(fh):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Queue_Id"])))
    ev.sort()
    fin = [e for s, e, n, q in ev if "final_epilogue" in n]
    if len(fin) < 4:
        print("trace too short")
        return 1
    # two final_epilogue kernels per step (one per sub-batch): the last step lies between the ends of pairs -2 and -1
    a, b = max(fin[-4:-2]), max(fin[-2:])
    step = [x for x in ev if x[0] >= a and x[1] <= b]
    queues = sorted({q for *_, q in step})
    print(f"last step: {(b - a) / 1e6:.2f} ms wall (under the profiler), {len(step)} kernels on queues {queues}")
    for q in queues:
        k = [x for x in step if x[3] == q]
        busy = sum(e - s for s, e, *_ in k)
        gaps = sum(max(0, k[i + 1][0] - k[i][1]) for i in range(len(k) - 1))
        print(f"  queue {q}: {len(k)} kernels, inside kernels {busy / 1e6:.2f} ms, between kernels {gaps / 1e6:.2f} ms")
    pts = []
    for s, e, n, q in step:
        pts.append((s, 1, q, klass(n)))
        pts.append((e, -1, q, klass(n)))
    pts.sort(key=lambda p: (p[0], p[1]))
    cur = {q: collections.Counter() for q in queues}
    last = a
    split = collections.Counter()
    pairs = collections.Counter()
    for t, d, q, c in pts:
        dt = t - last
        if dt > 0:
            act = {qq: [k for k, v in cur[qq].items() if v > 0] for qq in queues}
            n_act = sum(1 for qq in queues if act[qq])
            split[n_act] += dt
            key = tuple(sorted((act[qq][0] if act[qq] else "-") for qq in queues))
            pairs[key] += dt
        last = t
        cur[q][c] += d
    print("  wall-clock: " + ", ".join(f"{k} queue(s) in a kernel {v / 1e6:.2f} ms" for k, v in sorted(split.items())))
    print("  by class pair (ms):")
    for k, v in pairs.most_common(16):
        print(f"    {' + '.join(k):32s} {v / 1e6:7.2f}")
    per = collections.defaultdict(lambda: [0, 0])
    for s, e, n, q in step:
        per[klass(n)][0] += 1
        per[klass(n)][1] += e - s
    print("  sum of kernel durations by class (both queues):")
    for k, (c, d) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f"    {k:12s} {c:5d} launches {d / 1e6:8.2f} ms")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
