"""Per-kernel register / scratch / spill figures of built HIP objects, read from the code-object metadata (no recompile).

    python tools/kernel_resources.py [roma_amd/csrc/build/gemm.o ...]     # default: every object of the library
    python tools/kernel_resources.py --spills                             # only kernels that spill

The GEMM epilogues preload per-column vectors next to 96-128 accumulator registers; a change that tips a hot kernel into
scratch costs 30-50 % of its speed without failing any test (round 2: the 256 x 192 tile went from 247 to 335 us), so
tests/test_cpu_oracle.py holds the hot kernels to zero spills with this reader.
"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
sys.path.insert(0, os.path.join(ROOT, "tools"))


from audit_asm_reads import extract_code_object as code_object  # noqa: E402


def kernels(obj):
    """[{name, vgpr, agpr, sgpr, spill, scratch, lds}] of one host object"""
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", code_object(obj)], capture_output=True, text=True,
                           check=True).stdout
    out = []
    for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
        blk = ".agpr_count:" + blk

        def g(key, default="0"):
            m = re.search(r"\." + key + r":\s*(\S+)", blk)
            return m.group(1) if m else default
        sym = g("name", "?")
        nm = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
        nm = nm.replace("void roma::", "").replace("unsigned short", "bf16")
        nm = re.sub(r"\(.*\)$", "", nm)
        out.append({"name": nm, "vgpr": int(g("vgpr_count")), "agpr": int(g("agpr_count")), "sgpr": int(g("sgpr_count")),
                    "spill": int(g("vgpr_spill_count")), "sgpr_spill": int(g("sgpr_spill_count")),
                    "scratch": int(g("private_segment_fixed_size")), "lds": int(g("group_segment_fixed_size"))})
    return out


def main(argv):
    only_spills = "--spills" in argv
    objs = [a for a in argv if not a.startswith("--")] or sorted(glob.glob(os.path.join(ROOT, "roma_amd", "csrc", "build", "*.o")))
    n = 0
    for obj in objs:
        try:
            ks = kernels(obj)
        except RuntimeError:  # host-only object (api.o): no gfx950 code object inside
            continue
        for k in ks:
            if only_spills and k["spill"] == 0:
                continue
            n += 1
            print(f"{os.path.basename(obj):18s} {k['name'][:86]:86s} vgpr {k['vgpr']:3d} agpr {k['agpr']:3d} sgpr {k['sgpr']:3d} "
                  f"spill {k['spill']:3d} scratch {k['scratch']:4d} B/lane")
    print(f"{n} kernels listed")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
