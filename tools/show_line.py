"""One-line digest of a bench.py JSON line: python tools/show_line.py file.json [label]"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2] if len(sys.argv) > 2 else "", round(d["value"], 2), d["unit"], "| phases ms/q", d.get("phase_ms_per_question"),
      "| decode ms/step", d.get("decode_ms_per_step"), "| chains/step", d.get("mean_chains_per_step"))
