"""Instruction mix of one kernel of a built object:  python tools/isa_mix.py <obj> <substring of the demangled name> [--dump]"""
import collections
import re
import subprocess
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from audit_asm_reads import extract_code_object  # noqa: E402


def main(obj, pat, dump=False):
    co = extract_code_object(obj)
    dis = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", co], capture_output=True, text=True).stdout
    for blk in re.split(r"\n(?=[0-9a-f]+ <)", dis):
        m = re.match(r"[0-9a-f]+ <([^>]+)>", blk)
        if not m:
            continue
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        if pat not in name:
            continue
        lines = [re.sub(r"//.*", "", ln).strip() for ln in blk.split("\n")[1:]]
        if dump:
            print("\n".join(f"{i}: {ln}" for i, ln in enumerate(lines)))
            return
        ops = collections.Counter(ln.split()[0] for ln in lines if ln and not ln.startswith("<"))
        print(name[:110], sum(ops.values()))
        for k, v in ops.most_common(40):
            print(f"   {k:32s} {v}")
        return


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], "--dump" in sys.argv)
