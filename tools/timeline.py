"""Step timeline out of a rocprofv3 kernel trace: which queue every launch of one replayed step ran on, the gaps between
launches and how much of the step two queues overlapped.  python tools/timeline.py <kernel_trace.csv> [--dump]"""
import collections
import csv
import sys


def main(path, dump=False, steps=10):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_kernel")]
    t_end = [int(rows[i]["End_Timestamp"]) for i in ends]
    best = min(range(len(ends) - steps), key=lambda a: t_end[a + steps] - t_end[a])
    print("wall per step %.1f us over %d steps" % ((t_end[best + steps] - t_end[best]) / steps / 1e3, steps))
    seg = rows[ends[best] + 1: ends[best + 1] + 1]
    t0 = int(seg[0]["Start_Timestamp"])
    iv = [((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r["Queue_Id"], r["Kernel_Name"]) for r in seg]
    busy = sum(e - s for s, e, _, _ in iv)
    # union of intervals
    cover, cur_s, cur_e = 0.0, None, None
    for s, e, _, _ in sorted(iv):
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                cover += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    cover += cur_e - cur_s
    print("launches %d, sum of kernel times %.1f us, covered %.1f us, idle %.1f us, overlapped %.1f us" %
          (len(iv), busy, cover, iv[-1][1] - cover, busy - cover))
    print("queues:", dict(collections.Counter(q for _, _, q, _ in iv)))
    aten = collections.Counter(k.split("(")[0][:90] for _, _, _, k in iv if "at::native" in k)
    for k, v in aten.items():
        print("ATen: %d x %s" % (v, k))
    if dump:
        for s, e, q, k in iv:
            print("%8.1f %8.1f %6.1f q%s %s" % (s, e, e - s, q, k[:60]))


if __name__ == "__main__":
    main(sys.argv[1], "--dump" in sys.argv)
