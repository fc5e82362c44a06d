"""ISA audit: no packed-f32 VALU instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) may read an SGPR that one of the
next few instructions overwrites.

Found in round 4 (profiles/r04_v10_pk_sgpr_hazard.md): in refiner_input_pix_kernel hipcc SLP-packed the displacement
embedding into

        v_pk_fma_f32 v[0:1], v[6:7], s[4:5], v[0:1] op_sel:[0,0,1] op_sel_hi:[1,1,0]
        s_mov_b32 s4, s11                       ; next instruction: the register allocator reuses s4

and on gfx950 the LAST 16-lane pass of the packed instruction occasionally saw the NEW s4 - one output channel of 16
consecutive pixels (lanes 48 .. 63 of a wave) off by a few per cent, only while a second stream kept the SIMD's VALU busy
(refiner_block24_wave_kernel, itself v_pk_fma_f32 bound), i.e. once in ~50 two-stream match() calls.  The packed-f32
instructions read their 64-bit scalar operand pass by pass; a scalar write one issue slot later is not interlocked against the
later passes.  Nothing in the source controls the register allocator, so the rule is: scalar values that feed f32 arithmetic the
compiler may pack are moved to VGPRs first (`asm volatile("" : "+v"(x))`) where the scalars are short-lived, and this audit
checks the emitted ISA of every object of the library: a v_pk_*_f32 with an SGPR source is accepted only if none of the next
WINDOW instructions (up to the next branch) writes that SGPR - long-lived scalars such as the GEMM epilogues' alpha are fine.

    python tools/audit_pk_sgpr.py [objects...]        # default: every object of both builds
"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from audit_asm_reads import disassemble  # noqa: E402

PK = re.compile(r"^(v_pk_(?:fma|mul|add)_f32)\s+(.*)$")
WINDOW = 12


def sregs(text):
    out = set()
    for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bs(\d+)\b", text):
        out.add(int(a))
    return out


def writes_sgpr(c):
    """SGPRs an instruction writes (first operand of scalar / scalar-memory / lane-read instructions, SGPR results of VOP3 compares)"""
    op, _, rest = c.partition(" ")
    if not rest:
        return set()
    first = rest.split(",")[0]
    if op.startswith(("s_load", "s_buffer_load", "s_mov", "s_cmov", "s_add", "s_sub", "s_mul", "s_and", "s_or", "s_xor", "s_andn2", "s_orn2",
                      "s_lshl", "s_lshr", "s_ashr", "s_bfe", "s_bfm", "s_min", "s_max", "s_cselect", "s_not", "s_abs", "s_sext", "s_ff",
                      "s_flbit", "s_bcnt", "s_brev", "s_getpc", "s_getreg", "s_memtime", "s_nand", "s_nor", "s_xnor", "s_pack", "s_wqm",
                      "s_quadmask", "s_movk", "s_addk", "s_mulk", "s_lshl1", "s_lshl2", "s_lshl3", "s_lshl4", "s_addc", "s_subb")):
        return sregs(first)
    if op.startswith(("v_readlane", "v_readfirstlane", "v_cmp", "v_div_scale", "v_mad_u64", "v_mad_i64", "v_add_co", "v_sub_co", "v_addc_co",
                      "v_subb_co")):
        return sregs(first) | (sregs(rest.split(",")[1]) if op.startswith(("v_div_scale", "v_mad_u64", "v_mad_i64", "v_add_co", "v_sub_co")) and
                               "," in rest else set())
    return set()


def audit(obj):
    """[(kernel, instruction index, text)] of packed-f32 instructions with a scalar source"""
    hits = []
    kern, name = {}, None
    for ln in disassemble(obj).split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.*)>:", ln)
        if m:
            if not m.group(1).startswith("L"):
                name = m.group(1)
                kern[name] = []
            continue
        c = ln.split("//")[0].strip()
        if c and name is not None:
            kern[name].append(c)
    for name, body in kern.items():
        for i, c in enumerate(body):
            m = PK.match(c)
            if not m:
                continue
            ops = m.group(2).split(" op_sel")[0].split(" neg_")[0]
            srcs = ",".join([o.strip() for o in ops.split(",")][1:])  # operand 0 is the destination
            src = sregs(srcs)
            if not src:
                continue
            for j in range(i + 1, min(i + 1 + WINDOW, len(body))):
                if body[j].startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
                    break
                w = writes_sgpr(body[j]) & src
                if w:
                    hits.append((name, i, f"{c}   <- s{sorted(w)} rewritten {j - i} instruction(s) later by: {body[j]}"))
                    break
    return hits


def main(argv):
    objs = [a for a in argv if not a.startswith("--")]
    if not objs:
        for b in ("build", "build_f16"):
            objs += sorted(glob.glob(os.path.join(ROOT, "roma_amd", "csrc", b, "*.o")))
    bad = 0
    for obj in objs:
        try:
            hits = audit(obj)
        except RuntimeError:
            continue  # host-only object
        for k, i, c in hits:
            nm = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:100]
            print(f"{os.path.relpath(obj, ROOT)}: {nm} @{i}: {c}")
        bad += len(hits)
        print(f"{os.path.relpath(obj, ROOT)}: {len(hits)} packed-f32 instruction(s) whose scalar source is rewritten within {WINDOW} instructions")
    print("AUDIT OK" if bad == 0 else f"AUDIT FAILED: {bad} instruction(s)")
    return 0 if bad == 0 else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
