#!/usr/bin/env python3
"""`run_scripts/infer.sh` of the reference calls `python src/infer.py` (the file lives at src/eval/infer.py there,
so the reference script is broken as shipped); this shim makes that path work."""
import os
import runpy
import sys

if __name__ == "__main__":
    target = os.path.join(os.path.dirname(os.path.abspath(__file__)), "eval", "infer.py")
    sys.argv[0] = target
    runpy.run_path(target, run_name="__main__")
