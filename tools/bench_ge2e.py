#!/usr/bin/env python3
"""BASELINE config 5 (GE2E d-vector extraction and one training iteration): runs `python bench.py --ge2e`."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.exit(subprocess.call([sys.executable, os.path.join(root, "bench.py"), "--ge2e"]))
