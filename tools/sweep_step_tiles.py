#!/usr/bin/env python3
"""Tuning aid: force the tile of ONE conv problem shape (SSV_NNB_FORCE) and time the whole training step with bench.py.
Isolated kernel timings miss what a tile does to its neighbours in the step (cache state, clocks); this measures it.
usage: python tools/sweep_step_tiles.py [steps]      prints ms per step for every (shape, tile) candidate"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = sys.argv[1] if len(sys.argv) > 1 else "60"
SHAPES = [(3, 512, 325), (3, 256, 325), (3, 1024, 186), (3, 512, 186), (3, 512, 650), (3, 256, 650), (3, 512, 1300), (3, 256, 1300),
          (3, 1024, 1300)]
TILES = ["2,7", "2,6", "2,4", "1,7", "1,6", "1,4"]
def run(force):
    env = dict(os.environ)
    env.pop("SSV_NNB_FORCE", None)
    if force: env["SSV_NNB_FORCE"] = force
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-adversarial", "--no-fp32", "--no-ge2e", "--steps", steps, "--warmup", "3"],
                         env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    return d["ms_per_step"], d["config"]["text2mel_ms"], d["config"]["ssrn_ms"]
base = [run(None) for _ in range(2)]
print("baseline", base, flush=True)
for (kt, M, N) in SHAPES:
    line = "kt%d M%d N%d:" % (kt, M, N)
    for t in TILES:
        ms = run("%d:%d:%d=%s" % (kt, M, N, t))
        line += "  %s %.3f" % (t, ms[0])
    print(line, flush=True)
