#!/usr/bin/env python3
"""Summarise rocprofv3 output (kernel_stats.csv, optional FETCH_SIZE / WRITE_SIZE counter passes) into a short text table."""
import csv, glob, sys, collections
def stats(path, top=60):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    out = ["%7s %7s %11s  %s" % ("time%", "calls", "avg_us", "kernel")]
    for r in rows[:top]:
        out.append("%6.2f%% %7s %11.1f  %s" % (100 * float(r["TotalDurationNs"]) / tot, r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:120]))
    out.append("total kernel time: %.3f ms over %d kernels" % (tot / 1e6, len(rows)))
    return "\n".join(out)
def counters(path, name):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name: continue
        k = r["Kernel_Name"]
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    return acc
def counters_by_shape(path, name):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name: continue
        k = (r["Kernel_Name"].split("(")[0][:60], r["Grid_Size"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    return acc
def by_shape(path, top=22):
    """Kernel time per (kernel, grid): separates the launch shapes that share one template instantiation."""
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = (r["Kernel_Name"][:60], "%sx%sx%s" % (r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"]))
        acc[k][0] += 1
        acc[k][1] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
    tot = sum(v[1] for v in acc.values())
    out = ["\nper launch shape (grid in threads):", "%7s %7s %11s  %s" % ("time%", "calls", "avg_us", "kernel  grid")]
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:top]:
        out.append("%6.2f%% %7d %11.1f  %s  %s" % (100 * v[1] / tot, v[0], v[1] / v[0], k[0], k[1]))
    return "\n".join(out)
def idle_share(path):
    """How much of a replayed training step is NOT covered by any kernel: the steps are delimited by the Adam launches (two per
    benchmark step: Text2Mel, SSRN); per step the union of the kernels' busy intervals is compared with the step's span."""
    rows = [(float(r["Start_Timestamp"]), float(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
    rows.sort()
    adam = [i for i, r in enumerate(rows) if r[2].startswith("adam_multi_kernel")]
    spans = []
    for a, b in zip(adam[:-2:2], adam[2::2]):               # Adam k .. Adam k+2 = one Text2Mel + one SSRN iteration
        seg = rows[a + 1:b + 1]
        if len(seg) < 200:
            continue
        t0, t1 = rows[a][1], seg[-1][1]
        busy, cur_s, cur_e = 0.0, None, None
        for s_, e_, _ in seg:
            s_ = max(s_, t0)
            if cur_e is None or s_ > cur_e:
                if cur_e is not None:
                    busy += cur_e - cur_s
                cur_s, cur_e = s_, e_
            else:
                cur_e = max(cur_e, e_)
        busy += cur_e - cur_s
        spans.append(((t1 - t0) / 1e6, busy / 1e6, len(seg), sum(e_ - s_ for s_, e_, _ in seg) / 1e6))
    if not spans:
        return ""
    spans.sort()
    sp, busy, n, tot = spans[len(spans) // 2]
    return ("\nmedian replayed step (Adam to Adam, %d steps found): span %.3f ms, %d kernels, device busy (union of kernel intervals) %.3f ms = %.1f %%, "
            "idle %.3f ms (%.2f us per kernel), sum of kernel durations %.3f ms (overlap of the two encoder streams %.3f ms)"
            % (len(spans), sp, n, busy, 100 * busy / sp, sp - busy, (sp - busy) / n * 1e3, tot, tot - busy))


if __name__ == "__main__":
    d = sys.argv[1]
    print(stats((glob.glob(d + "/*/*kernel_stats.csv") + glob.glob(d + "/*kernel_stats.csv"))[0]))
    tr = glob.glob(d + "/*/*kernel_trace.csv") + glob.glob(d + "/*kernel_trace.csv")
    if tr:
        print(by_shape(tr[0]))
        print(idle_share(tr[0]))
    if len(sys.argv) > 3:
        f = counters((glob.glob(sys.argv[2] + "/*/*counter_collection.csv") + glob.glob(sys.argv[2] + "/*counter_collection.csv"))[0], "FETCH_SIZE")
        w = counters((glob.glob(sys.argv[3] + "/*/*counter_collection.csv") + glob.glob(sys.argv[3] + "/*counter_collection.csv"))[0], "WRITE_SIZE")
        print("\nPMC (separate passes; units as reported by rocprofv3 = KiB; per-launch averages)")
        print("%12s %12s %7s  %s" % ("FETCH_KiB", "WRITE_KiB", "calls", "kernel"))
        for k in sorted(f, key=lambda k: -f[k][0])[:14]:
            print("%12.1f %12.1f %7d  %s" % (f[k][0] / f[k][1], w[k][0] / max(1, w[k][1]), f[k][1], k[:110]))
        fs = counters_by_shape((glob.glob(sys.argv[2] + "/*/*counter_collection.csv") + glob.glob(sys.argv[2] + "/*counter_collection.csv"))[0], "FETCH_SIZE")
        wsz = counters_by_shape((glob.glob(sys.argv[3] + "/*/*counter_collection.csv") + glob.glob(sys.argv[3] + "/*counter_collection.csv"))[0], "WRITE_SIZE")
        print("\nPMC per launch shape (grid = total threads), the GEMM and reduction kernels")
        print("%12s %12s %7s  %s" % ("FETCH_KiB", "WRITE_KiB", "calls", "kernel  grid"))
        for k in sorted(fs, key=lambda k: -fs[k][0]):
            if "gemm_n" in k[0] or "reduce_" in k[0]:
                print("%12.1f %12.1f %7d  %s  %s" % (fs[k][0] / fs[k][1], wsz[k][0] / max(1, wsz[k][1]) if k in wsz else -1, fs[k][1], k[0], k[1]))
