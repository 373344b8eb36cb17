#!/usr/bin/env python3
"""Command line of the reference's generate_test_utterances.py (:44-52 there) for its synthesis part (:56-139): every
speaker's evaluation sentences through Text2Mel, SSRN and the vocoder on the HIP path.

    python generate_test_utterances.py -C config.json -T <tag> [--eval_utt_num 20]
"""
import argparse
import json


def cli():
    ap = argparse.ArgumentParser(description="Synthesize the spoofing test utterances (MI355X hot path)")
    ap.add_argument("-C", "--configuration", type=str, required=True)
    ap.add_argument("--eval_utt_num", type=int, default=20)
    ap.add_argument("-T", "--current_time", type=str, required=True)
    return ap.parse_args()


if __name__ == "__main__":
    a = cli()
    from spoofsv_amd import harness
    with open(a.configuration) as f:
        cfg = json.load(f)
    done = harness.generate_test_utterances(cfg, a.current_time, a.eval_utt_num)
    print("wrote %d utterances for %d speakers" % (sum(len(v) for v in done.values()), len(done)))
