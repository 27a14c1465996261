#!/bin/bash
# same-box A/B of the prefill pass budget (rows per pass): tools/ab_prefill_rows.sh ROWS ROWS ...  (26624 = the default 32 x 832)
mkdir -p gpurun_out/abrows
for rows in "$@"; do
  ZE_PREFILL_ROWS=$rows python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batch64 --no-configs1 --no-reuse-sensitivity > gpurun_out/abrows/$rows.json 2> gpurun_out/abrows/$rows.err
  python tools/show_line.py gpurun_out/abrows/$rows.json "prefill rows $rows:" || tail -3 gpurun_out/abrows/$rows.err
  python - <<P
import json
d = json.loads([l for l in open("gpurun_out/abrows/$rows.json") if l.startswith("{")][-1])
s = d["scheduler"]
print("   steps by live chains:", {k[6:]: v for k, v in sorted(s.items(), key=lambda kv: (len(kv[0]), kv[0])) if k.startswith("steps_")})
print("   phases:", {k: (round(v["ms_per_question"], 3), round(v["frac"], 3)) for k, v in d["roofline_phases"].items() if isinstance(v, dict) and "ms_per_question" in v})
P
done
