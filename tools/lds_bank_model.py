"""LDS bank-conflict model of the fused refiner blocks' lane maps (MI355X_MICROARCH.md, LDS table) - no GPU needed.

    python tools/lds_bank_model.py          # extra LDS cycles per image row and wave, per access stream

A wave64 LDS instruction is served in fixed lane groups, one cycle per group when the group's addresses fall on distinct
banks (or are equal: broadcast); every further distinct address on a busy bank costs one more cycle.  The model reproduces
the measured SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE shares of the kernels it was built for (block144: model 0.229 / measured
0.205-0.229, then 0.045 / 0.038; block24: 0.155 / 0.152), and is what the lane maps of refiner_block.hip and
refiner_block24w.hip were derived with.  tests/test_cpu_oracle.py runs it on the maps the sources hold.
"""
import collections
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_G = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS_READ_B128 = _G + [[l + 32 for l in g] for g in _G]
GROUPS_READ_B64 = [list(range(32)), list(range(32, 64))]
GROUPS_WRITE_B64 = [list(range(i, i + 16)) for i in range(0, 64, 16)]


def cycles(addrs, groups, banks, width):
    """addrs: {lane: byte address} of the active lanes -> (cycles, extra cycles) of one wave instruction"""
    cyc = ext = 0
    for g in groups:
        hit = collections.defaultdict(set)
        for l in g:
            if l in addrs:
                d0 = addrs[l] // 4
                for d in range(width // 4):
                    hit[(d0 + d) % banks].add(d0 + d)
        w = max([len(v) for v in hit.values()] or [1])
        cyc += w
        ext += w - 1
    return cyc, ext


def read_b128(a): return cycles(a, GROUPS_READ_B128, 64, 16)
def read_b64(a): return cycles(a, GROUPS_READ_B64, 64, 8)
def write_b64(a): return cycles(a, GROUPS_WRITE_B64, 32, 8)


# ---- refiner_block144_1b_kernel (refiner_block.hip): lane -> (quad, channel group), 4 waves
def block144_lane(wv, lane):
    if wv < 3:
        return 2 * wv + ((lane >> 4) & 1), (lane & 15) + 16 * (lane >> 5), True
    if lane < 32:
        return 6, lane, True
    k = lane - 32
    return (0x55643120 >> (4 * (k >> 2))) & 7, 32 + (k & 3), lane < 60


def block144(lane_map=block144_lane):
    CP, XROW = 144, 304
    tot, ext, cover = collections.Counter(), collections.Counter(), collections.Counter()
    for wv in range(4):
        act = {l: lane_map(wv, l) for l in range(64)}
        act = {l: v for l, v in act.items() if v[2]}
        for v in act.values():
            cover[(v[0], v[1])] += 1
        streams = [("ring", read_b64, [{l: (v[0] * 4 * CP + v[1] * 4) * 2 + j * CP * 2 for l, v in act.items()} for j in range(8)]),
                   ("taps", read_b128, [{l: (k * CP + v[1] * 4) * 4 for l, v in act.items()} for k in range(25)]),
                   ("xt_write", write_b64, [{l: (v[0] * 4 + px) * XROW + v[1] * 8 for l, v in act.items()} for px in range(4)]),
                   ("xt_read", read_b128, [{l: (l & 31) * XROW + ks * 32 + (l >> 5) * 16 for l in range(64)} for ks in range(9)])]
        for name, fn, instrs in streams:
            for a in instrs:
                c, e = fn(a)
                tot[name] += c
                ext[name] += e
    return dict(tot), dict(ext), cover


# ---- refiner_block24_wave_kernel (refiner_block24w.hip): the table in the source, one wave
def block24_table():
    src = open(os.path.join(ROOT, "roma_amd", "csrc", "refiner_block24w.hip")).read()
    body = re.search(r"g_rbw_lane_map\[64\]\s*=\s*\{(.*?)\};", src, re.S).group(1)
    body = re.sub(r"//[^\n]*", "", body)
    vals = [int(x) for x in re.findall(r"\d+", body)]
    assert len(vals) == 64, len(vals)
    return vals


def block24_xt_row(p):
    return p * 80 + 16 * ((0x96 >> ((p >> 2) & 7)) & 1) + (32 if p >= 32 else 0)


def block24(table=None):
    table = table or block24_table()
    act = {l: (e >> 3, e & 7) for l, e in enumerate(table) if e != 255}
    cover = collections.Counter(act.values())
    tot, ext = collections.Counter(), collections.Counter()
    streams = [("ring", read_b64, [{l: v[0] * 208 + v[1] * 8 + off for l, v in act.items()} for off in (0, 48, 96, 144, 208, 256, 304, 352)]),
               ("taps", read_b128, [{l: k * 96 + v[1] * 16 for l, v in act.items()} for k in range(25)]),
               ("xt_write", write_b64, [{l: block24_xt_row(v[0] * 4) + px * 80 + v[1] * 8 for l, v in act.items()} for px in range(4)]),
               ("xt_read", read_b128, [{l: block24_xt_row(u * 32 + (l & 31)) + ks * 32 + (l >> 5) * 16 for l in range(64)}
                                       for u in range(2) for ks in range(2)])]
    for name, fn, instrs in streams:
        for a in instrs:
            c, e = fn(a)
            tot[name] += c
            ext[name] += e
    return dict(tot), dict(ext), cover


if __name__ == "__main__":
    for name, fn in (("refiner_block144_1b", block144), ("refiner_block24_wave", block24)):
        tot, ext, cover = fn()
        t, e = sum(tot.values()), sum(ext.values())
        print(f"{name}: cycles {tot}  extra {ext}  ->  conflict share {e / t:.3f}  ({len(cover)} (quad, group) items)")
