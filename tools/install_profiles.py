#!/usr/bin/env python3
"""Copy the summaries tools/profile_round3.sh left under gpurun_out/r3/profiles/ (and the full bench line gpurun_out/r3/bench_full.json) into
profiles/, keeping the hand-written headers of the tracked files and refreshing the figures they quote."""
import json, re, shutil
ms = json.loads(open('gpurun_out/r3/profiles/trace.json').read().strip().splitlines()[-1])['ms_per_step']
d = json.loads(open('gpurun_out/r3/bench_full.json').read().strip().splitlines()[-1])
hdr = open('profiles/round3_bench_kernel_stats_f16x2.txt').read().split('  time%   calls')[0]
hdr = re.sub(r'\(\d+\.\d ms per step here, \d+\.\d\n# un-profiled', '(%.1f ms per step here, %.1f\n# un-profiled' % (ms, d['ms_per_step']), hdr)
open('profiles/round3_bench_kernel_stats_f16x2.txt', 'w').write(hdr + open('gpurun_out/r3/profiles/bench_f16x2.txt').read())
adv_h = open('profiles/round3_adversarial_kernel_stats.txt').read().split('  time%   calls')[0]
open('profiles/round3_adversarial_kernel_stats.txt', 'w').write(adv_h + open('gpurun_out/r3/profiles/adversarial.txt').read())
shutil.copy('gpurun_out/r3/profiles/shapes.tsv', 'profiles/round3_shapes.tsv')
t = json.load(open('profiles/traffic.json'))
t['f16x2'] = json.load(open('gpurun_out/r3/profiles/traffic_f16x2.json'))
json.dump(t, open('profiles/traffic.json', 'w'), indent=1)
open('profiles/round3_bench_line.json', 'w').write(json.dumps(d) + "\n")
c = d['config']
print("profiled %.2f ms; line %.3f ms = %.0f; roofline %.3f (%.2f us, traffic %s); bf16x3 %.2f; fp32 %.1f; adv %.2f / %.2f = %.0f; ge2e %.2f ms %.0f utt/s frac %.3f; cpu %.0f x%.0f" % (
    ms, d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['us_per_launch'], d['roofline']['traffic'], c['fast_bf16x3_ms_per_step'], c['fp32_exact_ms_per_step'],
    c['adversarial_text2mel_ms'], c['adversarial_ssrn_ms'], c['adversarial_combined_fps'], c['ge2e_ms'], c['ge2e_utt_per_s'], c['ge2e_roofline_frac'],
    d['cpu_baseline']['value'], c['speedup_vs_cpu_baseline']))
