"""Summarise a rocprofv3 --kernel-trace result (rocpd sqlite .db): per-kernel stats as CSV
(the same columns as rocprofv3's kernel_stats.csv) plus the busy time (union of kernel
intervals) so overlap between the two pipeline streams is visible.

    python scripts/prof_summary.py gpurun_out/prof_x/x_results.db [out.csv] [--skip-first N]
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:110]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    db = sqlite3.connect(args[0])
    rows = db.execute("select name, start, end, stream_id, grid_x*grid_y*grid_z, lds_size, vgpr_count, accum_vgpr_count, scratch_size "
                      "from kernels order by start").fetchall()
    stats = {}
    for name, s, e, st, grid, lds, vg, ag, scr in rows:
        d = stats.setdefault(short(name), dict(n=0, tot=0, mn=1 << 62, mx=0, grid=grid, lds=lds, vgpr=vg, agpr=ag, scratch=scr))
        dur = e - s
        d["n"] += 1
        d["tot"] += dur
        d["mn"] = min(d["mn"], dur)
        d["mx"] = max(d["mx"], dur)
    total = sum(d["tot"] for d in stats.values())
    # union of intervals
    busy, cur_s, cur_e = 0, None, None
    for _, s, e, *_ in rows:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        busy += cur_e - cur_s
    lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,LastGrid,LdsBytes,VGPR,AGPR,Scratch"]
    for k, d in sorted(stats.items(), key=lambda kv: -kv[1]["tot"]):
        lines.append('"%s",%d,%d,%.1f,%.2f,%d,%d,%s,%s,%s,%s,%s' % (k, d["n"], d["tot"], d["tot"] / d["n"], 100.0 * d["tot"] / total,
                                                                d["mn"], d["mx"], d["grid"], d["lds"], d["vgpr"], d["agpr"], d["scratch"]))
    lines.append('"# sum of kernel durations (ns)",,%d' % total)
    lines.append('"# GPU busy = union of kernel intervals (ns)",,%d' % busy)
    lines.append('"# span first start .. last end (ns)",,%d' % (rows[-1][2] - rows[0][1]))
    out = "\n".join(lines) + "\n"
    if len(args) > 1:
        open(args[1], "w").write(out)
    sys.stdout.write(out)


if __name__ == "__main__":
    main()
