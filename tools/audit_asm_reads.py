"""ISA audit of every hand-scheduled K loop in libroma_hip (gemm.hip, gemm8p.hip, gemm6p.hip, conv64.hip): no instruction may touch a register
that has an inline-asm LDS read in flight.

Background (the round-1 "f32 carried k-group produced wrong sums" defect, root-caused in round 2): the 8-wave GEMM
loops read their MFMA fragments with `asm volatile("ds_read_b128 %0, ...": "=v"(frag))` and wait with a separate
`s_waitcnt lgkmcnt(N)`, because hipcc drains the LDS-DMA queue before any LDS read it can see.  For the compiler the
asm's output is valid at ;;#ASMEND; the hardware writes it tens of cycles later.  In the bf16 kernels each fragment is
consumed whole (a 128-bit MFMA operand) and nothing happens to it before the wait.  In the f32 kernels every MFMA operand
is ONE 32-bit component; with the carried k-group the register file is full, and hipcc then gives consecutive asm reads
OVERLAPPING destination tuples (v[128:131], v[130:133], ...) and saves the components with `v_mov_b32` right after each
ds_read - i.e. it copies registers whose data has not landed (profiles/r02_f32_carry_isa_excerpt.txt).  Nothing in
the source can forbid such a copy, so the construct is guarded here instead: this audit runs on the code objects the
library is linked from and fails on the first such instruction.  tests/test_cpu_oracle.py runs it on every build.

Works on an llvm-objdump listing of the gfx950 code object, so it needs no special build: for each kernel, the span
from the second-to-last s_barrier before the first MFMA to the last MFMA is scanned; every ds_read is in flight until the
`s_waitcnt lgkmcnt(N)` that retires it (LDS returns in order; N = reads still allowed in flight).

    python tools/audit_asm_reads.py            # audits roma_amd/csrc/build/{gemm,gemm8p,gemm6p,conv64}.o
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def regs_of(text):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", text):
        out.add(int(a))
    return out


def extract_code_object(obj):
    """host object with an embedded hip fat binary -> path of its gfx950 code object (in a temp dir)"""
    tmp = tempfile.mkdtemp(prefix="roma_audit_")
    co = os.path.join(tmp, "dev.co")
    r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={obj}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
        # newer bundles: let llvm-objdump split them next to a copy of the object
        cp = os.path.join(tmp, os.path.basename(obj))
        subprocess.run(["cp", obj, cp], check=True)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", cp], capture_output=True, text=True)
        cands = [f for f in os.listdir(tmp) if "gfx950" in f]
        if not cands:
            raise RuntimeError(f"no gfx950 code object found in {obj}: {r.stderr}")
        co = os.path.join(tmp, cands[0])
    return co


def disassemble(obj):
    """objdump text of the gfx950 code object embedded in a host object"""
    co = extract_code_object(obj)
    return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout


def audit_listing(text):
    kern, name = {}, None
    for l in text.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.*)>:", l)
        if m:
            if not m.group(1).startswith("L"):
                name = m.group(1)
                kern[name] = []
            continue
        if name:
            c = l.split("//")[0].strip()
            if c:
                kern[name].append(c)
    report = []
    for k, body in kern.items():
        mf = [i for i, c in enumerate(body) if c.startswith("v_mfma")]
        if not mf:
            continue
        lo = mf[0]
        for _ in range(2):  # two barriers back: the 8-phase loop issues its first reads BEFORE the barrier that precedes its MFMAs
            lo = max(lo - 1, 0)
            while lo > 0 and not body[lo].startswith("s_barrier"):
                lo -= 1
        pend, probs, nreads = [], [], 0
        for i in range(lo, mf[-1] + 1):
            code = body[i]
            op, _, ops = code.partition(" ")
            if op.startswith("ds_read"):
                nreads += 1
                allp = set().union(*pend) if pend else set()
                if regs_of(",".join(ops.split(",")[1:])) & allp:
                    probs.append(code)
                pend.append(regs_of(ops.split(",")[0]))
                continue
            if op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load"):
                pend.append(set())
                continue
            if op == "s_waitcnt":
                m = re.search(r"lgkmcnt\((\d+)\)", ops)
                if m:
                    n = int(m.group(1))
                    pend = pend[len(pend) - n:] if n > 0 else []
                continue
            allp = set().union(*pend) if pend else set()
            if regs_of(ops) & allp:
                probs.append(code)
        nm = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        nm = nm.replace("void roma::", "").replace("(roma::GemmArgs)", "").replace("unsigned short", "bf16")
        report.append((nm, mf[-1] - lo, nreads, probs))
    return report


def main(objs=None):
    objs = objs or [os.path.join(ROOT, "roma_amd", "csrc", "build", f) for f in ("gemm_f32.o", "gemm_f32_conv.o", "gemm_h16.o", "gemm_h16_conv.o", "gemm_h16f32.o", "gemm8p.o", "gemm6p.o", "conv64.o")]
    bad = 0
    for obj in objs:
        rep = audit_listing(disassemble(obj))
        for nm, span, nreads, probs in rep:
            state = "clean" if not probs else f"{len(probs)} TOUCHES of in-flight registers, first: {probs[0]}"
            print(f"{os.path.basename(obj):10s} {nm:48s} K-loop span {span:5d} instr, {nreads:4d} LDS reads: {state}")
            bad += bool(probs)
    print("AUDIT", "FAILED" if bad else "OK", f"({bad} kernels with problems)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:] or None))
