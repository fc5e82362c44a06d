"""ISA audit of every counted `s_waitcnt vmcnt(N)`, N > 0, that guards an LDS-DMA load (`global_load_lds_*` or `buffer_load_* ... lds`): the N youngest
vector-memory instructions in front of the wait must all be LOADS.

The rule comes from round 3 (profiles/r03_v20_determinism_stress.log): vmcnt counts loads and stores in one
counter; loads return in order among themselves, but on gfx950 a store can be acknowledged before an OLDER LDS-DMA load has
landed.  A counted wait "leave the N youngest in flight" therefore only proves that the awaited DMA piece is in LDS if those N
youngest operations are loads: a store among them may retire early and let the wait pass with the awaited piece still in flight
(dwconv_ring / refiner_block* read a ring slot before its DMA had landed, 1 .. 5 of 3 000 two-stream runs).  Stores OLDER than the
awaited load are harmless.

The audit walks the objdump listing of every kernel of every object: from each counted wait backwards in program order, around
the back edge of the innermost loop that contains the wait (one wrap).  The awaited operation is the youngest load OLDER than
the N youngest vector-memory instructions; if it is an LDS-DMA load and a store / atomic / scratch access is among the N
youngest, the wait is reported.  Conditional issues are counted as issued (conservative).  Waits whose awaited operation is a
plain register load are the compiler's own (hipcc's wait-count pass treats loads and stores of gfx9 targets as one in-order
stream); they are counted and listed with --all, not failed: nothing in the source controls them, and no kernel of this library
feeds such a load into a hand-counted wait.

Limits: the walk follows the LISTING, so a kernel whose blocks hipcc moved out of line (the two-barrier refiner_block_kernel<>:
its border-handling blocks sit behind the row loop and branch back) can be reported although its program order is fine; such
kernels are named in KNOWN_LAYOUT below and reported as "layout", not failed - their waits are pinned by the GPU race screens
(bit-identical to the wave-private / one-barrier kernels over 150 launches under load, tests/test_gpu_ops.py).

    python tools/audit_vmcnt.py [--all] [objects...]          # default: every object of both builds
"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from audit_asm_reads import disassemble  # noqa: E402

KNOWN_LAYOUT = ("refiner_block_kernelILi24E", "refiner_block_kernelILi144E")
LOAD = ("global_load", "buffer_load", "flat_load")
OTHER = ("global_store", "buffer_store", "flat_store", "global_atomic", "buffer_atomic", "flat_atomic", "scratch_")


def is_dma(c):
    """LDS-DMA load: `global_load_lds_dwordx4 ...` or (round 6) `buffer_load_dwordx4 ... offen lds`"""
    return "_lds_" in c or (c.startswith("buffer_load") and re.search(r"\blds\b", c) is not None)


def kernels(text):
    """{name: [(address, instruction text, branch target address or None)]}"""
    out, name, base = {}, None, 0
    for ln in text.split("\n"):
        m = re.match(r"^([0-9a-f]+) <(.*)>:", ln)
        if m:
            if not m.group(2).startswith("L"):
                name, base = m.group(2), int(m.group(1), 16)
                out[name] = []
            continue
        if name is None or "//" not in ln:
            continue
        code, _, cm = ln.partition("//")
        code = code.strip()
        ma = re.match(r"\s*([0-9A-Fa-f]+):", cm)
        if not code or not ma:
            continue
        tgt = None
        if code.startswith(("s_cbranch", "s_branch")):
            mt = re.search(r"<[^>]*\+0x([0-9a-f]+)>", cm)
            if mt:
                tgt = base + int(mt.group(1), 16)
            elif re.search(r"<[^+>]*>", cm):
                tgt = base
        out[name].append((int(ma.group(1), 16), code, tgt))
    return out


def audit_kernel(body):
    """[(index, N, offending instruction)] for the counted waits of one kernel"""
    addr_to_idx = {a: i for i, (a, _, _) in enumerate(body)}
    loops = []  # (head index, back-edge index)
    for i, (_, c, t) in enumerate(body):
        if t is not None and t in addr_to_idx and addr_to_idx[t] <= i:
            loops.append((addr_to_idx[t], i))
    bad, soft, counted = [], [], 0
    for w, (_, c, _) in enumerate(body):
        m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", c)
        if not m or int(m.group(1)) == 0:
            continue
        n = int(m.group(1))
        counted += 1
        inner = [lp for lp in loops if lp[0] <= w <= lp[1]]
        lp = min(inner, key=lambda x: x[1] - x[0]) if inner else None
        order = list(range(w - 1, (lp[0] if lp else 0) - 1, -1))
        if lp:
            order += list(range(lp[1], w, -1))
        young, awaited = [], None
        for i in order:
            ci = body[i][1]
            if not (ci.startswith(LOAD) or ci.startswith(OTHER)):
                continue
            if len(young) < n:
                young.append(ci)
            elif ci.startswith(LOAD):
                awaited = ci
                break
        stores = [ci for ci in young if ci.startswith(OTHER)]
        if stores:
            if awaited is not None and is_dma(awaited):
                bad.append((w, n, stores[0]))
            else:
                soft.append((w, n, stores[0]))
    return bad, soft, counted


def main(argv):
    objs = [a for a in argv if not a.startswith("--")]
    show_all = "--all" in argv
    if not objs:
        for b in ("build", "build_f16"):
            objs += sorted(glob.glob(os.path.join(ROOT, "roma_amd", "csrc", b, "*.o")))
    nbad = 0
    for obj in objs:
        try:
            ks = kernels(disassemble(obj))
        except RuntimeError:
            continue  # host-only object
        total = nsoft = ndma = 0
        for name, body in ks.items():
            bad, soft, counted = audit_kernel(body)
            total += counted
            nsoft += len(soft)
            ndma += sum(is_dma(c) for _, c, _ in body)
            layout = any(k in name for k in KNOWN_LAYOUT)
            for tag, lst in (("layout" if layout else "FAIL", bad), ("compiler", soft if show_all else [])):
                for w, n, ci in lst:
                    nm = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:110]
                    print(f"{os.path.relpath(obj, ROOT)}: [{tag}] {nm} @{w}: vmcnt({n}) with a non-load among its {n} youngest: {ci}")
            nbad += 0 if layout else len(bad)
        print(f"{os.path.relpath(obj, ROOT)}: {len(ks)} kernels, {ndma} LDS-DMA issues, {total} counted waits, {nsoft} of them the compiler's "
              f"with a store in the allowance (register loads)")
    print("AUDIT OK" if nbad == 0 else f"AUDIT FAILED: {nbad} wait(s)")
    return 0 if nbad == 0 else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
