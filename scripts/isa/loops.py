#!/usr/bin/env python3
"""Loop tree of one kernel of csrc/stmpc.hip with the spill traffic of every loop's OWN blocks (sub-loops excluded).
usage: scripts/isa/loops.py <stmpc.s> <substring of the mangled kernel name>
For each loop: depth, header, parent, instructions / v_readlane+v_writelane (SGPR spill moves) / scratch_ (VGPR spill) / barriers in the
blocks that belong to this loop and to none of its sub-loops, and markers that identify the loop in the source:
  MIN64r ds_min_rtn_u64 (exact pass, candidate stage A)   MIN64 ds_min_u64 (bounding passes: one atomic per candidate)
  MIN32 ds_min_u32 (tie stage)   DIV v_div_* / v_rcp_f64 (cell penalties: the per-cell IEEE division)   GST global_store   GLD global_load"""
import collections, re, sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    lines = open(path).read().split('\n')
    st = next(k for k, l in enumerate(lines) if re.match(r'^_Z\w*:', l) and pat in l.split(':')[0])
    en = next(k for k in range(st, len(lines)) if lines[k].startswith('.Lfunc_end'))
    print("kernel", lines[st].split(':')[0])
    loops = collections.OrderedDict()      # header -> dict
    cur = None                             # header of the innermost loop of the current block (None = not in a loop)
    def loop(h, depth=None, parent=None):
        d = loops.setdefault(h, dict(depth=depth, parent=parent, n=0, lane=0, scr=0, bar=0, marks=collections.Counter()))
        if depth is not None: d["depth"] = depth
        if parent is not None: d["parent"] = parent
        return d
    k = st + 1
    while k < en:
        l = lines[k]
        m = re.match(r'^(\.LBB\d+_\d+):\s*;(.*)', l) or re.match(r'^; %bb\.(\d+):\s*;(.*)', l)
        if m:
            label = m.group(1) if l.startswith('.') else 'BB%s_%s' % (re.search(r'BB(\d+)_', lines[st + 1] + ''.join(lines[st:st + 400])).group(1), m.group(1))
            label = label.lstrip('.L')
            rest = m.group(2)
            # gather the comment block (continuation lines start with spaces + ';')
            j = k + 1
            while j < en and re.match(r'^\s+;', lines[j]): rest += ' ' + lines[j].strip(); j += 1
            mh = re.search(r'This (?:Inner )?Loop Header: Depth=(\d+)', rest)
            if mh:
                parents = re.findall(r'Parent Loop (BB\d+_\d+) Depth=(\d+)', rest)
                par = parents[-1][0] if parents else None
                loop(label, int(mh.group(1)), par)
                cur = label
            else:
                mi = re.search(r'in Loop: Header=(BB\d+_\d+) Depth=(\d+)', rest)
                cur = mi.group(1) if mi else None
                if mi: loop(cur, int(mi.group(2)))
            k = j
            continue
        s = l.strip()
        if s and not s.startswith(';') and not s.startswith('.') and cur is not None:
            op = s.split()[0]
            d = loops[cur]
            d["n"] += 1
            if op in ('v_readlane_b32', 'v_writelane_b32'): d["lane"] += 1
            if op.startswith('scratch_'): d["scr"] += 1
            if op == 's_barrier': d["bar"] += 1
            if op == 'ds_min_rtn_u64': d["marks"]["MIN64r"] += 1
            elif op == 'ds_min_u64': d["marks"]["MIN64"] += 1
            elif op == 'ds_min_u32': d["marks"]["MIN32"] += 1
            elif op.startswith('v_div_') or op == 'v_rcp_f64_e32': d["marks"]["DIV"] += 1
            elif op.startswith('global_store'): d["marks"]["GST"] += 1
            elif op.startswith('global_load'): d["marks"]["GLD"] += 1
        k += 1
    print("  %-5s %-12s %-12s %6s %6s %7s %4s  %s" % ("depth", "header", "parent", "instrs", "lane", "scratch", "bar", "markers"))
    tot = collections.Counter()
    for h, d in loops.items():
        print("  %-5s %-12s %-12s %6d %6d %7d %4d  %s" % (d["depth"], h, d["parent"] or "-", d["n"], d["lane"], d["scr"], d["bar"],
                                                          " ".join("%s:%d" % kv for kv in sorted(d["marks"].items()))))
        tot[(d["depth"], "lane")] += d["lane"]; tot[(d["depth"], "scr")] += d["scr"]; tot[(d["depth"], "n")] += d["n"]
    print("  by depth:", ", ".join("d%d: %d instrs, %d lane moves, %d scratch" % (dd, tot[(dd, "n")], tot[(dd, "lane")], tot[(dd, "scr")]) for dd in sorted({k_[0] for k_ in tot if k_[0] is not None})))


if __name__ == "__main__":
    main()
