#!/usr/bin/env python3
"""Digest of the round's rocprofv3 summaries in profiles/: writes profiles/traffic_latest.json -- HBM bytes per launch of
the kernels whose roofline bench.py reports, from the PMC passes of tools/pmc_kernel.py (FETCH_SIZE and WRITE_SIZE collected
in separate runs; FETCH_SIZE x 1024 x 2 because gfx950 tallies 128-B read requests at 64 B, WRITE_SIZE x 1024:
MI355X_MICROARCH.md, HBM) -- and prints the per-kernel duration and MFMA-utilisation tables quoted in profiles/README.md.

usage: tools/profile_digest.py [profiles_dir] [round]
"""
import csv
import json
import os
import sys

D = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
ROUND = int(sys.argv[2]) if len(sys.argv) > 2 else 3
P = f"r{ROUND:02d}"
# bench.py's kernel keys -> substring of the kernel's name, per measured configuration (tools/pmc_kernel.py modes)
KEYS = {
    "configs1": {"gate_up": "k_gemv<2, 1, 1, 4, 16>", "down": "k_gemv<1, 1, 4, ", "lm_head": "k_gemv<3, 2, 1, 4, 16>"},
    # (down at 220 chains = two launches: 128 x 128 tiles park the K slices, k_splitk_reduce adds them -- summed)
    "wide": {"attention": "k_attn_decode_wave<8>", "gate_up": "k_gemm_wstream<256, 96",
             "down": ("k_gemm_ring<128, 128, 4, 0, 2, 4, true>", "k_splitk_reduce<128, 128")},
    "wide_shared": {"attention": "k_attn_decode_wave<8>"},
    "batch64": {"attention": "k_attn_decode_wave<8>", "gate_up": "k_gemm_skinny<6, 3", "down": "k_gemm_ring<64, 64, 4, 2, 4, 2, false>"},
}
SECTION = {"configs1": "configs1", "wide": "stream", "wide_shared": "stream_shared", "batch64": "batch64"}
if ROUND >= 4:  # round 4: the stream's rows at the line's own mean chain count (410) and every GEMM of the layer; 580 = the bucket
    # (513-640 chains) in which the stream runs most of its chain-steps
    del KEYS["wide"], KEYS["wide_shared"]
    KEYS["wide410"] = {"attention": "k_attn_decode_wave_long<8, 6", "gate_up": "k_gemm_wstream<512, 96",
                       "down": ("k_gemm_ring<128, 256, 3, 0, 2, 4, true, false", "k_splitk_reduce<128, 256"),  # (…, true, true: a prefill GEMM of the set-up)
                       "qkv": "k_gemm_ring<64, 64, 4, 5", "o_proj": "k_gemm_ring<64, 64, 4, 2, 4, 2, false, true", "lm_head": "k_gemm_p8<4, true>"}
    KEYS["wide410_shared"] = {"attention": "k_attn_decode_wave_long<8, 6"}
    KEYS["wide580"] = {"attention": "k_attn_decode_wave_long<8, 6", "gate_up": "k_gemm_ring<320, 192, 4, 3",
                       "down": ("k_gemm_ring<320, 128, 4, 0", "k_splitk_reduce<320, 128"),
                       "qkv": "k_gemm_ring<64, 64, 4, 5", "o_proj": "k_gemm_ring<64, 128, 4, 2, 2, 4, false, true", "lm_head": "k_gemm_p8<4, true>"}
    KEYS["wide580_shared"] = {"attention": "k_attn_decode_wave_long<8, 6"}
    KEYS["batch64"]["attention"] = "k_attn_decode_wave_long<8, 6"
    SECTION.update(wide410="stream", wide410_shared="stream_shared", wide580="stream580", wide580_shared="stream580_shared")

if ROUND >= 5:  # round 5: tall one-round tiles for qkv / o, K-steps of 64 in two stages for the 256- / 320-row gate/up and down tiles
    KEYS["wide410"].update(gate_up="k_gemm_ring<256, 192, 2, 3", qkv="k_gemm_ring<80, 64, 6, 5", o_proj="k_gemm_ring<64, 64, 6, 2, 4, 2, false, true")
    KEYS["wide580"].update(gate_up="k_gemm_ring<320, 192, 2, 3", down=("k_gemm_ring<320, 128, 2, 0", "k_splitk_reduce<320, 128"),
                           qkv="k_gemm_ring<112, 64, 6, 5", o_proj="k_gemm_ring<80, 64, 6, 2")


    # second pass of round 5 (tools/profile_round5b.sh): the scheduler's hold raised the line's mean step to ~490 chains; the 410-chain
    # files stay as sections stream410 / stream410_shared
    KEYS["wide490"] = {"attention": "k_attn_decode_wave_long<8, 6", "gate_up": "k_gemm_ring<256, 192, 2, 3",
                       "down": ("k_gemm_ring<128, 256, 3, 0, 2, 4, true, false", "k_splitk_reduce<128, 256"),
                       "qkv": "k_gemm_ring<96, 64, 6, 5", "o_proj": "k_gemm_ring<64, 64, 6, 2, 4, 2, false, true", "lm_head": "k_gemm_p8<4, true>"}
    KEYS["wide490_shared"] = {"attention": "k_attn_decode_wave_long<8, 6"}
    SECTION.update(wide490="stream", wide490_shared="stream_shared", wide410="stream410", wide410_shared="stream410_shared")


def rows(name, source=None):
    with open(os.path.join(D, f"{P}_{name}")) as fh:
        return [r for r in csv.DictReader(fh) if source is None or r["source"] == source]


def find(table, sub):
    hits = [(k, v) for k, v in table.items() if sub in k]
    return max(hits, key=lambda kv: kv[1][1]) if hits else (None, None)  # the most-dispatched match


out = {"round": ROUND, "correction": "FETCH_SIZE x 1024 x 2 (gfx950 counts 128-B read requests at 64 B) + WRITE_SIZE x 1024; "
       "separate --pmc passes around tools/pmc_kernel.py (the launches bench.py times)"}
for mode, keys in KEYS.items():
    try:
        fetch = {r["kernel"]: (float(r["mean"]), int(r["dispatches"])) for r in rows(f"pmc_{mode}_fetch_size.csv", "pmc") if r["quantity"] == "FETCH_SIZE"}
        write = {r["kernel"]: (float(r["mean"]), int(r["dispatches"])) for r in rows(f"pmc_{mode}_write_size.csv", "pmc") if r["quantity"] == "WRITE_SIZE"}
        with open(os.path.join(D, f"{P}_pmc_{mode}_launches.json")) as fh:
            launches = json.loads(fh.read().strip().splitlines()[-1])
    except FileNotFoundError:
        continue
    sec = out.setdefault(SECTION[mode], {})
    if launches.get("kernel_sources_sha16"):   # round 6: the tree the passes ran on (bench.py compares it with the tree that quotes them)
        out["kernel_sources_sha16"] = launches["kernel_sources_sha16"]
    for key, sub in keys.items():
        subs = sub if isinstance(sub, tuple) else (sub,)
        parts = [(find(fetch, x), find(write, x)) for x in subs]
        if any(pf[0] is None or pw[1] is None for pf, pw in parts) or key not in launches:
            continue
        name = " + ".join(pf[0].split("(")[0] for pf, _ in parts)
        f = (sum(pf[1][0] for pf, _ in parts), parts[0][0][1][1])
        w = (sum(pw[1][0] for _, pw in parts), parts[0][1][1][1])
        hbm = f[0] * 1024 * 2 + w[0] * 1024
        alg = launches[key]["bytes_per_launch"]
        sec[key] = {"kernel": name.split("(")[0], "fetch_size_kb_mean": f[0], "write_size_kb_mean": w[0], "dispatches": f[1],
                    "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "ratio": hbm / alg,
                    "chains": launches.get("chains")}
        print(f"{SECTION[mode]:9s} {key:10s} {name.split('(')[0][:60]:60s} fetch {2 * f[0] * 1024 / 1e6:8.2f} MB + write {w[0] * 1024 / 1e6:7.2f} MB"
              f" = {hbm / 1e6:8.2f} MB per launch = {hbm / alg:.3f} x algorithmic ({alg / 1e6:.2f} MB)")
out["source"] = [f"profiles/{P}_pmc_{m}_{c}.csv" for m in KEYS for c in ("fetch_size", "write_size")]
with open(os.path.join(D, "traffic_latest.json"), "w") as fh:
    json.dump(out, fh, indent=1)

for name in ("stream", "configs1", "batch64"):
    try:
        stats = {r["kernel"]: r for r in rows(f"{name}_kernel_stats.csv", "kernel_stats")}
    except FileNotFoundError:
        continue
    tot = sum(float(r["mean"]) * int(r["dispatches"]) for r in stats.values())
    print(f"\n[{name}] kernel | calls | mean us | % of GPU time")
    for k, r in sorted(stats.items(), key=lambda kv: -float(kv[1]["mean"]) * int(kv[1]["dispatches"]))[:18]:
        print(f"{k.split('(')[0][:78]:78s} | {r['dispatches']:>7s} | {float(r['mean']) / 1e3:8.2f} | {100 * float(r['mean']) * int(r['dispatches']) / tot:5.2f}")

try:
    busy, gui, dur = {}, {}, {}
    for r in rows("stream_pmc_mfma_busy.csv", "pmc"):
        (busy if r["quantity"] == "SQ_VALU_MFMA_BUSY_CYCLES" else gui)[r["kernel"]] = float(r["mean"])
    for r in rows("stream_pmc_mfma_busy.csv", "kernel_trace"):
        dur[r["kernel"]] = (float(r["mean"]) / 1e3, int(r["dispatches"]))
    print("\n[stream, eager] kernel | calls | mean us | MFMA utilisation = busy / (1024 SIMDs x GUI_ACTIVE / 8)")
    for k in sorted(busy, key=lambda k: -busy[k] * dur.get(k, (0, 0))[1])[:16]:
        if busy[k] > 0 and gui.get(k, 0) > 0:
            print(f"{k.split('(')[0][:78]:78s} | {dur.get(k, (0, 0))[1]:6d} | {dur.get(k, (float('nan'), 0))[0]:8.1f} | {busy[k] / (1024 * gui[k] / 8):.3f}")
except FileNotFoundError:
    pass
