#!/usr/bin/env python3
"""Static instruction mix of one kernel of a gfx950 assembly file (hipcc --cuda-device-only -S).

    python3 exp/tools/isa_stats.py /tmp/isa/gat_compact.s 'rgat_aggregate_runs_packedILi16ELi4ELb1' [--dump out.s] [--loops]

Counts by class (VALU / SALU / VMEM / LDS / branch), the 64-bit address arithmetic, moves and conditional branches, and the
register / occupancy figures of the kernel descriptor.  --loops prints the same per loop body (a backward branch target to its
branch), which is what the gather kernels' time follows."""
import re
import sys
from collections import Counter


def kernels(path):
    cur, body, out = None, [], {}
    for line in open(path):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m and not line.startswith("."):
            cur, body = m.group(1), []
            out[cur] = body
            continue
        if cur is not None:
            body.append(line.rstrip("\n"))
            if line.strip().startswith(".end_amdhsa_kernel") or line.strip().startswith(".Lfunc_end"):
                cur = None
    return out


def classify(op):
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("s_cbranch", "s_branch")):
        return "BRANCH"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "WAIT"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("v_"):
        return "VALU"
    return "OTHER"


def stats(lines):
    ops = []
    for l in lines:
        s = l.strip()
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        ops.append(s.split()[0])
    c = Counter(ops)
    cls = Counter(classify(o) for o in ops)
    return ops, c, cls


def report(name, lines, top=28):
    ops, c, cls = stats(lines)
    print(f"== {name}: {len(ops)} instructions  " + "  ".join(f"{k} {v}" for k, v in sorted(cls.items())))
    a64 = sum(v for k, v in c.items() if k in ("v_lshl_add_u64", "v_lshlrev_b64", "v_mad_u64_u32", "v_addc_co_u32", "v_add_co_u32", "v_ashrrev_i64", "v_mad_i64_i32", "v_lshl_add_u64"))
    print(f"   64-bit address arithmetic {a64}   v_mov_b32 {c['v_mov_b32']}   v_cndmask_b32 {c['v_cndmask_b32']}   s_cbranch_execz {c['s_cbranch_execz']}"
          f"   v_pk_* {sum(v for k, v in c.items() if k.startswith('v_pk_'))}   dpp {sum(1 for l in lines if 'quad_perm' in l or 'row_' in l and 'dpp' in l)}")
    print("   " + "  ".join(f"{k}:{v}" for k, v in c.most_common(top)))


def main():
    path, pat = sys.argv[1], sys.argv[2]
    ks = kernels(path)
    hit = [k for k in ks if pat in k]
    if not hit:
        print("no kernel matches; have:")
        for k in ks:
            print("  ", k)
        return 1
    for k in hit:
        lines = ks[k]
        report(k, lines)
        for l in lines:
            if any(t in l for t in (".amdhsa_next_free_vgpr", ".amdhsa_next_free_sgpr", ".amdhsa_group_segment", ".amdhsa_private_segment_fixed", ".amdhsa_accum_offset")):
                print("   " + l.strip())
        for l in lines:
            if re.search(r"; (Occupancy|NumVgprs|NumAgprs|ScratchSize|codeLenInByte)", l):
                print("   " + l.strip())
        if "--loops" in sys.argv:
            labels = {}
            for i, l in enumerate(lines):
                m = re.match(r"^(\.LBB\w+):", l)
                if m:
                    labels[m.group(1)] = i
            seen = set()
            for i, l in enumerate(lines):
                m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\w+)|\s+s_branch\s+(\.LBB\w+)", l)
                if m:
                    t = m.group(1) or m.group(2)
                    if t in labels and labels[t] < i and (labels[t], i) not in seen:
                        seen.add((labels[t], i))
                        report(f"loop {t} lines {labels[t]}..{i}", lines[labels[t]:i + 1], top=18)
        if "--dump" in sys.argv:
            out = sys.argv[sys.argv.index("--dump") + 1]
            open(out, "w").write("\n".join(lines))
    return 0


if __name__ == "__main__":
    sys.exit(main())
