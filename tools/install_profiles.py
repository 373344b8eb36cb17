#!/usr/bin/env python3
"""Copy the summaries tools/profile_round4.sh left under gpurun_out/r4/profiles/ (and the full bench line gpurun_out/r4/bench_full.json) into
profiles/ with a header that says what was run, and refresh profiles/traffic.json (the PMC record bench.py's roofline.traffic reads)."""
import json, os, shutil, sys
N = sys.argv[1] if len(sys.argv) > 1 else "5"
R = 'gpurun_out/r%s/profiles' % N
ms = json.loads(open(R + '/trace.json').read().strip().splitlines()[-1])['ms_per_step']
d = json.loads(open('gpurun_out/r%s/bench_full.json' % N).read().strip().splitlines()[-1])
hdr = ("# Round %s, MI355X (gfx950), split-fp16 arithmetic (the default): `python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ge2e --no-adversarial\n"
       "# --no-fp32 --no-roofline --no-stock` under rocprofv3 (tools/profile_round.sh): (1) --kernel-trace --stats, with the library's shape log (SSV_SHAPE_LOG) giving\n"
       "# the algorithmic FLOP / bytes of every launch shape -> \"achieved\" and \"frac\" of the roof per (kernel, grid); (2) --pmc FETCH_SIZE and (3) --pmc WRITE_SIZE\n"
       "# in separate passes; (4) one SQ pass with the MFMA-busy column.  Profiled runs hold a lower clock than un-profiled ones, and boxes differ by +-4 %% (%.1f ms per step here; the\n"
       "# un-profiled line profiles/round%s_bench_line.json, %.1f ms, was taken in another call, possibly on another box): compare rows of this file with each other, not with bench.py's wall clock.\n"
       "# Summary by tools/summarize_prof.py.\n" % (N, ms, N, d['ms_per_step']))
open('profiles/round%s_bench_kernel_stats.txt' % N, 'w').write(hdr + open(R + '/bench_f16x2.txt').read())
adv_h = ("# Round " + N + ", MI355X: the WGAN-GP cycle of train_ssrn --adversarial (1 G + 5 D iterations, B = 32, hipGraph replay), split-fp16 arithmetic:\n"
         "# rocprofv3 --kernel-trace --stats -- python3 tools/bench_adversarial.py   (the last line gives the device-busy share of the replayed part)\n")
open('profiles/round%s_adversarial_kernel_stats.txt' % N, 'w').write(adv_h + open(R + '/adversarial.txt').read())
shutil.copy(R + '/shapes.tsv', 'profiles/round%s_shapes.tsv' % N)
if os.path.exists(R + '/step_kernels.txt'):
    shutil.copy(R + '/step_kernels.txt', 'profiles/round%s_step_kernels.txt' % N)
t = json.load(open('profiles/traffic.json'))
t['f16x2'] = json.load(open(R + '/traffic_f16x2.json'))
json.dump(t, open('profiles/traffic.json', 'w'), indent=1)
# the raw per-launch counter rows the traffic figure was averaged from (so that roofline.traffic can be recomputed from a committed file)
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    src = R + '/headline_%s.csv' % c
    if os.path.exists(src):
        shutil.copy(src, 'profiles/round%s_headline_%s.csv' % (N, c))
open('profiles/round%s_bench_line.json' % N, 'w').write(json.dumps(d) + "\n")
c = d['config']
print("profiled %.2f ms; line %.3f ms = %.0f; roofline %.3f (%.2f us, traffic %s); bf16x3 %.2f; fp32 %.1f; adv %.2f / %.2f = %.0f; ge2e %.2f ms %.0f utt/s frac %.3f; cpu %.0f x%.0f" % (
    ms, d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['us_per_launch'], d['roofline']['traffic'], c['fast_bf16x3_ms_per_step'], c['fp32_exact_ms_per_step'],
    c['adversarial_text2mel_ms'], c['adversarial_ssrn_ms'], c['adversarial_combined_fps'], c['ge2e_ms'], c['ge2e_utt_per_s'], c['ge2e_roofline_frac'],
    d['cpu_baseline']['value'], c['speedup_vs_cpu_baseline']))
