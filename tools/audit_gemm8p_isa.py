"""Static audit of the hand-scheduled main loop of roma_amd/csrc/gemm8p.hip in the ISA hipcc emitted.

    make -C roma_amd/csrc AUDIT=1 build/gemm8p.o      # or: hipcc ... -save-temps=obj -c gemm8p.hip
    python tools/audit_gemm8p_isa.py roma_amd/csrc/build/gemm8p-hip-amdgcn-amd-amdhsa-gfx950.s

The fragment reads of that loop are inline-asm `ds_read_b128` whose completion the compiler does not track
(/opt/skills/guides/cdna_hip_programming.md section 5.7 item 1): the destination registers hold garbage until OUR
`s_waitcnt lgkmcnt(0)`.  The construct is only safe if, between a read and the wait that covers it, no instruction
reads, copies, spills or overwrites those registers.  The round-1 f32 kernel with a carried k-group produced wrong sums
exactly there.  This script checks it for every kernel in the file, plus three performance properties of the loop:

  1. no instruction outside ;;#ASMSTART / ;;#ASMEND touches a register with an asm LDS read in flight;
  2. no scratch (spill) traffic between the first and the last MFMA of the kernel;
  3. no compiler-inserted s_waitcnt inside the K loop except the one lgkmcnt(0) we place at the top of every K tile (one per
     copy of the body: the STEADY and the general inner loop) and lgkmcnt(0) waits in the once-per-tile code between them;
  4. every K-loop `s_waitcnt vmcnt(N)` is ours (inside an asm block).

Exit status 1 if any check fails.
"""
import re
import sys


def regs_of(text):
    """all VGPR numbers named in an operand string: v12, v[4:7]"""
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", text):
        out.add(int(a))
    return out


def audit(name, body):
    problems = []
    mf = [i for i, l in enumerate(body) if "v_mfma" in l]
    if not mf:
        return ["no MFMA found"], {}
    # K loop = from the label of the loop that holds the first MFMA to the last MFMA
    lo = mf[0]
    while lo > 0 and not re.match(r"^\.LBB\d+_\d+:.*", body[lo]):
        lo -= 1
    # walk back to the inner loop header (a label whose comment says "Inner Loop Header")
    k = mf[0]
    hdr = lo
    while k > 0:
        if re.match(r"^\.LBB\d+_\d+:", body[k]) and any("Inner Loop Header" in body[j] for j in range(k, min(k + 4, len(body)))):
            hdr = k
            break
        k -= 1
    hi = mf[-1]
    in_asm = False
    pending = {}  # vgpr -> line of the asm read
    compiler_waits, scratch, foreign_vm = [], [], []
    for i in range(hdr, hi + 1):
        l = body[i]
        s = l.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        code = s.split(";")[0].strip()
        if not code:
            continue
        op, _, operands = code.partition(" ")
        if in_asm:
            if op.startswith("ds_read"):
                dst = operands.split(",")[0]
                for r in regs_of(dst):
                    pending[r] = i
            elif op == "s_waitcnt" and "lgkmcnt(0)" in operands:
                pending.clear()
            continue
        if op == "s_waitcnt":
            compiler_waits.append((i, code))
            if "lgkmcnt(0)" in operands:
                pending.clear()
            if "vmcnt" in operands:
                foreign_vm.append((i, code))
            continue
        if op.startswith("scratch_"):
            scratch.append((i, code))
        touched = regs_of(operands) & set(pending)
        if touched:
            problems.append(f"line {i}: `{code}` touches v{sorted(touched)} while an asm ds_read (line {min(pending[r] for r in touched)}) is in flight")
    if scratch:
        problems.append(f"{len(scratch)} scratch access(es) inside the K loop, first: {scratch[0]}")
    # Round 6: the K loop is two inner loops (the STEADY copy and the general copy of gemm8p_ktile.inc) with once-per-tile
    # code between them.  Allowed: ONE lgkmcnt(0) per inner loop (ours, at the top of every K tile) and lgkmcnt(0) waits in
    # the once-per-tile code outside the inner loops; nothing else.
    # the compiler's block comments say which loop a basic block belongs to: "=> This Inner Loop Header: Depth=2" on the header,
    # "in Loop: Header=BBx_y Depth=2" on the other blocks of an inner loop, "Depth=1" on the once-per-tile code
    def loop_of(line_no):
        j = line_no
        while j > 0 and not (re.match(r"^\.LBB\d+_\d+:", body[j]) or re.match(r"^; %bb\.\d+:", body[j])):
            j -= 1
        head = body[j:j + 5]
        lab = re.match(r"^\.(LBB\d+_\d+):", body[j])
        for h in head[:4] if lab else head[:3]:
            if "Inner Loop Header" in h and lab:
                return lab.group(1)[1:]
            m = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", h)
            if m:
                return m.group(1) if int(m.group(2)) >= 2 else None
            if h is not head[0] and (re.match(r"^\.LBB\d+_\d+:", h) or not h.lstrip().startswith(";")):
                break
        return None
    per_loop = {}
    bad_in_loop = []
    for i, c in compiler_waits:
        lp = loop_of(i)
        if lp is not None:
            per_loop[lp] = per_loop.get(lp, 0) + 1
            if "lgkmcnt(0)" not in c:
                bad_in_loop.append((i, c))
        elif "lgkmcnt" not in c:  # once-per-tile code: scalar-load waits of any count are fine, nothing else
            bad_in_loop.append((i, c))
    loops = sorted(per_loop)
    if any(n > 1 for n in per_loop.values()) or bad_in_loop:
        problems.append(f"compiler-inserted waits inside the K loop: {compiler_waits} (per inner loop: {per_loop})")
    if foreign_vm:
        problems.append(f"compiler-inserted vmcnt waits inside the K loop: {foreign_vm}")
    info = {"k_loop_lines": hi - hdr + 1, "mfma": len(mf), "inner_loops": len(loops), "compiler_waits": [c for _, c in compiler_waits]}
    return problems, info


def main():
    path = sys.argv[1]
    lines = open(path).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN4roma13gemm8p_kernel\w+:", l)]
    starts.append(len(lines))
    bad = 0
    for a, b in zip(starts[:-1], starts[1:]):
        name = lines[a].split(":")[0]
        end = next((j for j in range(a, b) if "s_endpgm" in lines[j]), b)
        problems, info = audit(name, lines[a:end + 1])
        print(f"{name}: {info}")
        for p in problems:
            bad += 1
            print("   PROBLEM:", p)
    meta = re.findall(r"\.vgpr_spill_count:\s+(\d+)", "\n".join(lines))
    print("vgpr_spill_count per kernel:", meta)
    print("AUDIT", "FAILED" if bad else "OK")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
