#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output into a short text table.

    summarize_prof.py TRACE_DIR [FETCH_DIR WRITE_DIR] [--shapes shapes.tsv] [--sq SQ_DIR]

TRACE_DIR  : rocprofv3 --kernel-trace --stats --output-format csv
FETCH_DIR / WRITE_DIR : separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (the guide: they do not fit one pass)
--shapes   : the library's shape log (env SSV_SHAPE_LOG while the traced command ran): algorithmic FLOP / bytes per (kernel, grid),
             from the formulas of DESIGN.md section 4 -- adds the "achieved / roof" columns (MFMA roof for the GEMM kernels, HBM roof
             for the LayerNorm kernels)
--sq       : a --pmc pass with SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES
             SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAVES -- adds the SQ table with the MFMA-busy column
"""
import collections
import csv
import glob
import re
import sys

PEAK_SPLIT_TFLOPS = 2500.0 / 3.0      # three 16-bit MFMAs per algorithmic fp32 product (bf16 or fp16 operands: same dense peak)
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_TBS = 8.0


def _find(d, pat):
    hits = glob.glob(d + "/*/*" + pat) + glob.glob(d + "/*" + pat)
    return hits[0] if hits else None


def short(name, n=58):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name)[:n]


def stats(path, top=40):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    out = ["%7s %7s %11s  %s" % ("time%", "calls", "avg_us", "kernel")]
    for r in rows[:top]:
        out.append("%6.2f%% %7s %11.1f  %s" % (100 * float(r["TotalDurationNs"]) / tot, r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:120]))
    out.append("total kernel time: %.3f ms over %d kernels" % (tot / 1e6, len(rows)))
    return "\n".join(out)


def load_shapes(path):
    """(kernel name as logged, 'XxYxZ' in threads) -> (flops, bytes, note)"""
    shapes = {}
    if not path:
        return shapes
    acc = collections.defaultdict(list)
    for ln in open(path):
        f = ln.rstrip("\n").split("\t")
        if len(f) >= 4:
            acc[(f[0], f[1])].append((float(f[2]), float(f[3]), f[4] if len(f) > 4 else "", float(f[5]) if len(f) > 5 else 1.0))
    for k, v in acc.items():        # several problems can share a (kernel, grid): launch-weighted means, notes joined
        w = sum(e[3] for e in v)
        shapes[k] = (sum(e[0] * e[3] for e in v) / w, sum(e[1] * e[3] for e in v) / w, " | ".join("%s x%d" % (e[2], e[3]) if len(v) > 1 else e[2] for e in v))
    return shapes


def by_shape(path, shapes, top=40):
    """Kernel time per (kernel, grid) -- separates the launch shapes that share one template instantiation -- with, where the
    shape log knows the launch, algorithmic TFLOP/s or TB/s and the fraction of the roof that bounds the kernel."""
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = (short(r["Kernel_Name"]), "%sx%sx%s" % (r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"]))
        acc[k][0] += 1
        acc[k][1] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
    tot = sum(v[1] for v in acc.values())
    out = ["\nper launch shape (grid in threads); achieved = algorithmic FLOP (or bytes) of the launch / average duration;",
           "roof: split-MFMA GEMMs 833 TFLOP/s (2500 / 3), fp32-MFMA GEMMs 157.3 TFLOP/s, LayerNorm kernels 8 TB/s of HBM",
           "%7s %6s %9s %10s %6s  %-46s %-14s %s" % ("time%", "calls", "avg_us", "achieved", "frac", "kernel", "grid", "problem")]
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:top]:
        avg = v[1] / v[0]
        ach, frac, note = "", "", ""
        hit = shapes.get((k[0], k[1]))
        if hit:
            flops, nbytes, note = hit
            if flops > 0:
                t = flops / (avg * 1e-6) / 1e12
                roof = PEAK_SPLIT_TFLOPS if ("bf3" in k[0] or "pwln" in k[0] or "nt3r" in k[0]) else PEAK_F32_TFLOPS
                ach, frac = "%.0f TF/s" % t, "%.3f" % (t / roof)
            elif nbytes > 0:
                t = nbytes / (avg * 1e-6) / 1e12
                ach, frac = "%.2f TB/s" % t, "%.3f" % (t / PEAK_HBM_TBS)
        out.append("%6.2f%% %6d %9.1f %10s %6s  %-46s %-14s %s" % (100 * v[1] / tot, v[0], avg, ach, frac, k[0][:46], k[1], note))
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


def busy_share(path):
    """Device-busy share of the whole trace (union of kernel intervals / span from the first to the last kernel), and the launches
    of torch's own element-wise / reduction kernels in it (the adversarial cycle's glue)."""
    rows = sorted((float(r["Start_Timestamp"]), float(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path)))
    if not rows:
        return ""
    busy, cs, ce = 0.0, rows[0][0], rows[0][1]
    for s_, e_, _ in rows[1:]:
        if s_ > ce:
            busy += ce - cs
            cs, ce = s_, e_
        else:
            ce = max(ce, e_)
    busy += ce - cs
    span = rows[-1][1] - rows[0][0]
    aten = [r for r in rows if "at::native" in r[2]]
    out = ("\nwhole trace: %d kernels over %.3f ms, device busy %.1f %%; torch element-wise / reduction kernels (at::native::*): %d launches, %.2f %% of kernel time"
           % (len(rows), span / 1e6, 100 * busy / span, len(aten), 100 * sum(e - s for s, e, _ in aten) / max(1.0, sum(e - s for s, e, _ in rows))))
    # the replayed part: from the last host-side gap of more than 2 ms (warm-up / capture end there) to the end of the trace
    i = len(rows) - 1
    hi = rows[i][0]
    while i > 0:
        prev_end = max(r[1] for r in rows[max(0, i - 64):i])
        if rows[i][0] - prev_end > 2e6:
            break
        i -= 1
    tail = rows[i:]
    if len(tail) > 100:
        b2, cs, ce = 0.0, tail[0][0], tail[0][1]
        for s_, e_, _ in tail[1:]:
            if s_ > ce:
                b2 += ce - cs
                cs, ce = s_, e_
            else:
                ce = max(ce, e_)
        b2 += ce - cs
        sp2 = max(r[1] for r in tail) - tail[0][0]
        at2 = [r for r in tail if "at::native" in r[2]]
        out += ("\nreplayed part (after the last host-side gap > 2 ms): %d kernels over %.3f ms, device busy %.1f %%, idle %.2f us per kernel; at::native::* %d launches, %.2f %% of its kernel time"
                % (len(tail), sp2 / 1e6, 100 * b2 / sp2, (sp2 - b2) / len(tail) / 1e3, len(at2), 100 * sum(e - s for s, e, _ in at2) / max(1.0, sum(e - s for s, e, _ in tail))))
    return out


def counters(path, name):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name:
            continue
        k = (short(r["Kernel_Name"], 60), r["Grid_Size"])
        acc[k][0] += float(r["Counter_Value"])
        acc[k][1] += 1
    return acc


def pmc_table(fetch_dir, write_dir):
    f = counters(_find(fetch_dir, "counter_collection.csv"), "FETCH_SIZE")
    w = counters(_find(write_dir, "counter_collection.csv"), "WRITE_SIZE")
    out = ["\nPMC, separate passes (FETCH_SIZE, WRITE_SIZE; units as rocprofv3 reports them = KiB), per-launch averages per (kernel, total threads);",
           "gfx950: 16-byte-per-lane loads are tallied at half their bytes (MI355X_MICROARCH.md) -- the pre-split weight fragments of the GEMM kernels",
           "%12s %12s %7s  %s" % ("FETCH_KiB", "WRITE_KiB", "calls", "kernel  grid")]
    for k in sorted(f, key=lambda k: -f[k][0])[:30]:
        out.append("%12.1f %12.1f %7d  %s  %s" % (f[k][0] / f[k][1], w[k][0] / max(1, w[k][1]) if k in w else -1, f[k][1], k[0], k[1]))
    return "\n".join(out)


def sq_table(sq_dir):
    path = _find(sq_dir, "counter_collection.csv")
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for r in csv.DictReader(open(path)):
        k = (short(r["Kernel_Name"], 60), r["Grid_Size"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES":
            n[k] += 1
    out = ["\nSQ counters (one --pmc pass), per (kernel, total threads), summed over the launches and divided by SQ_WAVE_CYCLES:",
           "issue_stall = SQ_WAIT_INST_ANY, parked = SQ_WAIT_ANY (s_waitcnt / barrier), valu = SQ_ACTIVE_INST_VALU, lds = SQ_ACTIVE_INST_LDS,",
           "mfma/wave = SQ_VALU_MFMA_BUSY_CYCLES per wave-cycle; waves/SIMD = SQ_WAVE_CYCLES / SQ_BUSY_CU_CYCLES (average resident waves per SIMD:",
           "the CU counter advances once per 4 cycles; reads 1.9 / 2.8 / 4.0 for the kernels that hold 2 / 3 / 4 waves per SIMD);",
           "MFMA busy = mfma/wave x waves/SIMD / 4 = share of the matrix pipes' cycles with an MFMA in flight (the busy counter is summed over the 4 SIMDs)",
           "%-50s %-9s %5s | %11s %7s %6s %6s | %9s %10s %9s" % ("kernel", "grid", "n", "issue_stall", "parked", "valu", "lds", "mfma/wave", "waves/SIMD", "MFMA busy")]
    for k in sorted(acc, key=lambda k: -acc[k]["SQ_WAVE_CYCLES"])[:30]:
        c = acc[k]
        wc = c["SQ_WAVE_CYCLES"] or 1.0
        occ = wc / c["SQ_BUSY_CU_CYCLES"] if c.get("SQ_BUSY_CU_CYCLES") else float("nan")
        mf = c["SQ_VALU_MFMA_BUSY_CYCLES"] / wc
        out.append("%-50s %-9s %5d | %11.3f %7.3f %6.3f %6.3f | %9.3f %10.2f %9.3f" % (
            k[0][:50], k[1], n[k], c["SQ_WAIT_INST_ANY"] / wc, c["SQ_WAIT_ANY"] / wc, c["SQ_ACTIVE_INST_VALU"] / wc, c["SQ_ACTIVE_INST_LDS"] / wc,
            mf, occ, mf * occ / 4.0))
    return "\n".join(out)


if __name__ == "__main__":
    args = sys.argv[1:]
    shapes_path = sq_dir = None
    if "--shapes" in args:
        i = args.index("--shapes")
        shapes_path = args[i + 1]
        del args[i:i + 2]
    if "--sq" in args:
        i = args.index("--sq")
        sq_dir = args[i + 1]
        del args[i:i + 2]
    d = args[0]
    st = _find(d, "kernel_stats.csv")
    if st:
        print(stats(st))
    tr = _find(d, "kernel_trace.csv")
    if tr:
        print(by_shape(tr, load_shapes(shapes_path)))
        print(idle_share(tr))
        print(busy_share(tr))
    if len(args) > 2:
        print(pmc_table(args[1], args[2]))
    if sq_dir:
        print(sq_table(sq_dir))
