"""Which kernels of a gfx950 assembly file (hipcc --save-temps) wait for a scalar (kernarg) load before they issue their first
vector-memory load?  With kernarg preload (csrc/gemv.h GEMV_HOT_PARAMS) the hot kernels of the decode chain must not: an `s_waitcnt
lgkmcnt` in front of the first global load is a memory round trip in front of every weight load of the launch.
usage: python tools/isa_prologue_check.py file.s [name-substring ...]"""
import re
import sys


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    S = open(path).read().split("\n")
    res = []
    for i, line in enumerate(S):
        m = re.match(r"^(_Z\S+):", line)
        if not m or (pats and not any(p in m.group(1) for p in pats)):
            continue
        j = i + 1
        while j < len(S) and ".p2align\t8" not in S[j] and not S[j].startswith("\t.section") and j < i + 40:
            j += 1
        preload = ".p2align\t8" in S[j] if j < len(S) else False
        if not preload:
            j = i
        sl, first = 0, None
        for k in range(j, min(len(S), j + 600)):
            t = S[k].strip()
            if t.startswith("s_load"):
                sl += 1
            if t.startswith("s_waitcnt") and "lgkmcnt" in t and sl:
                first = f"SMEM WAIT at +{k - j}"
                break
            if t.startswith(("global_load", "buffer_load", "flat_load")):
                first = f"first load at +{k - j}"
                break
            if t.startswith("s_endpgm"):
                break
        res.append((m.group(1), preload, first, sl))
    bad = [r for r in res if r[2] and r[2].startswith("SMEM")]
    print(f"{len(res)} kernels, {sum(1 for r in res if r[1])} with a preload header, {len(bad)} wait for a scalar load before their first vector load")
    for r in bad:
        print("  ", r[0][:110], r[2], f"({r[3]} s_loads before)")


if __name__ == "__main__":
    main()
