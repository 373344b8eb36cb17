#!/usr/bin/env python3
"""Command line of the reference's main.py (main.py:10-16 there), dispatching to the HIP-backed trainers:

    python main.py {train_text2mel,train_ssrn,synthesize} -C config.json -T <tag>
                   [-P {universal,conditional,ubm-finetune}] [-R checkpoint] [--adversarial] [--save_spectrogram]
"""
import argparse
import json
import os


def cli():
    ap = argparse.ArgumentParser(description="Adversarial Conditional Text-to-speech (MI355X hot path)")
    ap.add_argument("step", choices=["train_text2mel", "train_ssrn", "synthesize"], metavar="s")
    ap.add_argument("-P", "--pattern", choices=["universal", "conditional", "ubm-finetune"], default="conditional", metavar="m")
    ap.add_argument("-R", "--resume", type=str, default=None, metavar="checkpoint")
    ap.add_argument("-C", "--configuration", type=str, default=None)
    ap.add_argument("--adversarial", action="store_true")
    ap.add_argument("--save_spectrogram", action="store_true")
    ap.add_argument("-T", "--current_time", type=str, required=True, metavar="T")
    return ap.parse_args()


def run(args):
    from spoofsv_amd import harness
    with open(args.configuration) as f:
        cfg = json.load(f)
    spec_dir = None
    if args.save_spectrogram:
        spec_dir = os.path.join(cfg["SRC_ROOT_DIR"], "spec") + os.sep
        os.makedirs(spec_dir, exist_ok=True)
    common = dict(cfg=cfg, spec_dir=spec_dir, current_time=args.current_time)
    if args.step == "synthesize":
        return harness.synthesize(pattern=args.pattern, **common)
    trainer = harness.adversarial_train if args.adversarial else harness.ordinary_train
    return trainer(train_step=args.step, train_pattern=args.pattern, resume_checkpoints=args.resume, **common)


if __name__ == "__main__":
    run(cli())
