#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite file: per-kernel dispatch stats and PMC sums.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db [kernel-substring]
"""
import sqlite3
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]

    def T(prefix):
        return [t for t in tabs if t.startswith(prefix)][0]

    ksym = {r[0]: r[1] for r in c.execute("select id, kernel_name from %s" % T("rocpd_info_kernel_symbol"))}
    disp = list(c.execute("select id, kernel_id, start, end, event_id, grid_size_x, workgroup_size_x, group_segment_size "
                          "from %s" % T("rocpd_kernel_dispatch")))
    per = defaultdict(list)
    ev2k = {}
    for did, kid, s, e, ev, gx, wx, lds in disp:
        name = ksym.get(kid, str(kid))
        per[name].append((e - s, gx, wx, lds))
        ev2k[ev] = name
    print("%-90s %6s %12s %12s %12s  grid/wg/lds" % ("kernel", "calls", "avg_us", "min_us", "max_us"))
    for name, v in sorted(per.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
        if filt and filt not in name:
            continue
        d = [x[0] / 1e3 for x in v]
        print("%-90s %6d %12.2f %12.2f %12.2f  %d/%d/%d" % (name[:90], len(d), sum(d) / len(d), min(d), max(d), v[0][1], v[0][2], v[0][3]))
    pmc = {r[0]: r[1] for r in c.execute("select id, name from %s" % T("rocpd_info_pmc"))}
    sums = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(set)
    for ev, pid, val in c.execute("select event_id, pmc_id, value from %s" % T("rocpd_pmc_event")):
        k = ev2k.get(ev, "?")
        sums[k][pmc.get(pid, str(pid))] += val
        cnt[k].add(ev)
    for k, d in sums.items():
        if filt and filt not in k:
            continue
        n = max(len(cnt[k]), 1)
        print("\nPMC per dispatch (avg over %d dispatches): %s" % (n, k[:100]))
        for name, v in sorted(d.items()):
            print("   %-28s %18.1f" % (name, v / n))


if __name__ == "__main__":
    main()
