#!/usr/bin/env python3
"""ISA census of one kernel of csrc/stmpc.hip (gfx950 assembly from `hipcc --cuda-device-only -S`).
usage: scripts/isa/census.py <stmpc.s> <substring of the mangled kernel name> [--blocks]
Prints the kernel's resource footer, and per loop depth the instruction count, the spill moves (v_readlane / v_writelane that implement
SGPR spills, scratch_ loads / stores that implement VGPR spills) -- the evidence that spills sit outside the hot loops -- and, with
--blocks, one row per basic block."""
import collections, re, sys


def kernel_body(lines, pat):
    st = next(k for k, l in enumerate(lines) if re.match(r'^_Z\w*:', l) and pat in l.split(':')[0])
    en = next(k for k in range(st, len(lines)) if lines[k].startswith('.Lfunc_end'))
    return lines[st].split(':')[0], lines[st:en], st, en


def main():
    path, pat = sys.argv[1], sys.argv[2]
    lines = open(path).read().split('\n')
    name, body, st, en = kernel_body(lines, pat)
    print("kernel", name)
    for l in lines[en:en + 80]:
        m = re.match(r'^; (codeLenInByte|NumSgprs|NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|Occupancy|SGPRSpill|VGPRSpill|LDSByteSize|sgpr_spill_count|vgpr_spill_count)\b.*', l)
        if m:
            print("  " + l[2:].strip())
    depth, blk = 0, 'entry'
    tot, lane, scr, f64, bar = (collections.Counter() for _ in range(5))
    blocks = collections.OrderedDict()
    for l in body[1:]:
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        m2 = re.match(r'^; %bb\.(\d+):', l)
        if m or m2:
            blk = m.group(1) if m else 'bb' + m2.group(1)
            d = re.search(r'Depth=(\d+)', l)
            depth = int(d.group(1)) if d else 0
            blocks[blk] = [depth, 0, 0, 0, 0, 0]
            continue
        if 'Loop Header' in l:
            d = re.search(r'Depth=(\d+)', l)
            if d:
                depth = int(d.group(1))
                if blk in blocks:
                    blocks[blk][0] = depth
            continue
        s = l.strip()
        if not s or s.startswith(';') or s.startswith('.'):
            continue
        op = s.split()[0]
        tot[depth] += 1
        b = blocks.setdefault(blk, [depth, 0, 0, 0, 0, 0])
        b[1] += 1
        if op in ('v_readlane_b32', 'v_writelane_b32'):
            lane[depth] += 1; b[2] += 1
        if op.startswith('scratch_'):
            scr[depth] += 1; b[3] += 1
        if re.match(r'v_\w+_f64', op):
            f64[depth] += 1; b[4] += 1
        if op == 's_barrier':
            bar[depth] += 1; b[5] += 1
    print("  %5s %7s %10s %8s %6s %8s" % ("depth", "instrs", "lane-moves", "scratch", "f64", "barriers"))
    for d in sorted(tot):
        print("  %5d %7d %10d %8d %6d %8d" % (d, tot[d], lane[d], scr[d], f64[d], bar[d]))
    if '--blocks' in sys.argv:
        for k, v in blocks.items():
            print("  %-12s depth %d n=%4d lane=%3d scratch=%3d f64=%3d bar=%d" % (k, *v))


if __name__ == "__main__":
    main()
