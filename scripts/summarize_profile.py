#!/usr/bin/env python
"""Condense a gpurun_out/prof_rN directory (rocprofv3 csv output of `bench.py --steps 5 --warmup 2
--samples 0 --no-cpu-baseline`) into the small files committed under profiles/:
  rN_kernel_stats.csv      rocprofv3 --kernel-trace --stats summary (verbatim)
  rN_traffic.json          per-kernel HBM bytes per launch from the separate --pmc passes
                           (FETCH_SIZE x 2 on gfx950, MI355X_MICROARCH.md §HBM; WRITE_SIZE as is; KiB units)
  rN_sq_counters.json      SQ counters per kernel (MFMA busy, wait/issue split)
usage: python scripts/summarize_profile.py gpurun_out/prof_r1 r1
"""
import collections
import csv
import glob
import json
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]


def counters(sub):
    found = sorted(glob.glob(f"{src}/{sub}/**/*_counter_collection.csv", recursive=True))
    if not found:                                   # optional pass (e.g. no SQ pass for the Krylov profile)
        return collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0.0, 0]))
    f = found[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0.0, 0]))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        t = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += t; a[2] += 1
    return acc


shutil.copy(sorted(glob.glob(f"{src}/stats/**/*_kernel_stats.csv", recursive=True))[0], f"profiles/{tag}_kernel_stats.csv")
fetch, write, sq = counters("fetch"), counters("write"), counters("sq")
traffic = {}
for k in fetch:
    if "lip::" not in k:
        continue
    f = fetch[k]["FETCH_SIZE"]
    w = write[k].get("WRITE_SIZE", [0, 0, 1])
    h = write[k].get("TCC_HIT_sum", [0, 0, 1])[0]
    m = write[k].get("TCC_MISS_sum", [0, 0, 1])[0]
    n = f[2]
    rd = 2.0 * f[0] * 1024 / n          # FETCH_SIZE is in KiB and reads 1/2 of a wide coalesced stream on gfx950
    wr = w[0] * 1024 / max(w[2], 1)
    traffic[k] = dict(launches=n, avg_launch_ms=1e3 * f[1] / n, hbm_read_bytes_per_launch=rd,
                      hbm_write_bytes_per_launch=wr, hbm_bytes_per_launch=rd + wr,
                      hbm_tb_per_s=(rd + wr) / (f[1] / n) / 1e12, l2_hit_rate=h / max(h + m, 1.0))
what = sys.argv[3] if len(sys.argv) > 3 else "python bench.py --steps 5 --warmup 2 --samples 0 --no-cpu-baseline"
json.dump(dict(command="rocprofv3 --kernel-trace --pmc FETCH_SIZE  /  --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum  -- " + what,
               correction="read bytes = 2 x FETCH_SIZE (gfx950 counts 128-B requests as 64 B); KiB -> bytes",
               kernels=traffic), open(f"profiles/{tag}_traffic.json", "w"), indent=1)
sqo = {}
for k in (sq if sq else {}):
    if "lip::" not in k:
        continue
    d = {c: v[0] for c, v in sq[k].items()}
    t = sq[k]["SQ_WAVE_CYCLES"][1]
    d["total_seconds"] = t
    d["mfma_busy_fraction_at_2p4GHz"] = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (t * 2.4e9 * 1024)
    sqo[k] = d
if sqo:
    json.dump(sqo, open(f"profiles/{tag}_sq_counters.json", "w"), indent=1)
for k, v in sorted(traffic.items(), key=lambda kv: -kv[1]["avg_launch_ms"] * kv[1]["launches"])[:8]:
    print(f"{k[:52]:52s} n={v['launches']:4d} {v['avg_launch_ms']:.2f} ms  {v['hbm_bytes_per_launch']/1e9:.2f} GB/launch "
          f"{v['hbm_tb_per_s']:.2f} TB/s  L2 hit {v['l2_hit_rate']:.2f}  mfma busy {sqo.get(k, {}).get('mfma_busy_fraction_at_2p4GHz', 0):.2f}")
