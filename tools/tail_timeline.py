"""Step-tail anatomy from a rocprofv3 kernel trace of bench.py: what runs on the scoring stream between the end of one
k_feat_reduce and the start of the next (durations and the gaps between the launches).
    python tools/tail_timeline.py <dir with *_kernel_trace.csv>"""
import collections, csv, glob, os, sys


def main(d):
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    feat = [r for r in rows if "k_feat_reduce<" in r["Kernel_Name"]]
    if not feat:
        print("no k_feat_reduce in", f); return
    qkey = "Queue_Id" if "Queue_Id" in rows[0] else None
    feat.sort(key=lambda r: r["s"])
    per = collections.defaultdict(list)
    gaps = []
    for a, b in zip(feat[4:-1], feat[5:]):      # skip the first launches (warm-up)
        between = sorted((r for r in rows if r["s"] >= a["e"] and r["e"] <= b["s"] and (qkey is None or r[qkey] == a[qkey])), key=lambda r: r["s"])
        t = a["e"]
        busy = 0
        for r in between:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
            per[name].append((r["e"] - r["s"]) / 1e3)
            busy += r["e"] - r["s"]
            t = r["e"]
        gaps.append(((b["s"] - a["e"]) / 1e3, busy / 1e3, len(between)))
    n = len(gaps)
    print("file", f)
    print("steps analysed", n, " feat avg ms", sum((r["e"] - r["s"]) for r in feat[4:]) / 1e6 / len(feat[4:]))
    print("tail (feat end -> next feat start) avg us %.1f, of which kernels on that queue %.1f us in %.1f launches" %
          (sum(g[0] for g in gaps) / n, sum(g[1] for g in gaps) / n, sum(g[2] for g in gaps) / n))
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        print("  %-62s calls/step %.2f  avg %.1f us" % (k, len(v) / n, sum(v) / len(v)))


if __name__ == "__main__":
    main(sys.argv[1])
