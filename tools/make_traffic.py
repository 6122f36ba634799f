#!/usr/bin/env python3
"""profiles/traffic.json + profiles/rNN_pmc_summary.txt from the three PMC passes of tools/prof_round.sh
(gpurun_out/r2p/pmc_FETCH_SIZE.txt, pmc_WRITE_SIZE.txt, pmc_sq.txt: one line per (kernel, grid) with the mean counter
values per dispatch).  Usage: python3 tools/make_traffic.py gpurun_out/r2p r02"""
import json
import os
import re
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_hash  # noqa: E402  (bench.py prints traffic_stale when csrc/ no longer matches)


def rows(name):
    out = {}
    for line in open(os.path.join(src, name)):
        m = re.match(r"\s*(?:void )?(k_\w+(?:<[^>]*>?)?)[^|]*\|g(\d+)\s+n=\s*(\d+)\s+(.*)", line)
        if not m:
            continue
        vals = {k: float(v) for k, v in re.findall(r"(\w+)=([0-9.eE+-]+)", m.group(4))}
        out[(m.group(1).split("(")[0], int(m.group(2)))] = (int(m.group(3)), vals)
    return out


fetch, write, sq = rows("pmc_FETCH_SIZE.txt"), rows("pmc_WRITE_SIZE.txt"), rows("pmc_sq.txt")
# the update's five launches, identified by kernel + grid (config 2: 240 / 400 forward tiles; backward tiles 256 / 500 / 210 since round 5's
# re-split, 312 / 464 / 190 before it and with DDRL_BQ_SPLIT=0)
bwd = (256, 500, 210) if any(k[0].startswith("k_dg<10>") and k[1] == 500 * 256 for k in fetch) else (312, 464, 190)
want = [("k_dfwd<0>", "k_dfwd<0,", 240 * 256), ("k_dfwd<1>", "k_dfwd<1,", 400 * 256), ("k_dg bq", "k_dg<10>", bwd[0] * 256),
        ("k_dg mid", "k_dg<10>", bwd[1] * 256), ("k_dg pi", "k_dg<10>", bwd[2] * 256)]
per, lines = {}, []
for label, kname, grid in want:
    key = next((k for k in fetch if k[0].startswith(kname.rstrip(",")) and k[1] == grid), None)
    if key is None:
        print("missing", label, file=sys.stderr)
        continue
    f = fetch[key][1]["FETCH_SIZE"] * 1024 * 2      # KB; x2: the gfx950 counter tallies 64 B per 128-B request (MI355X guide)
    w = write[key][1]["WRITE_SIZE"] * 1024
    s = sq[key][1]
    cyc = s["BUSY_CYCLES"] / 32.0                   # SQ_BUSY_CYCLES is summed over the 32 shader engines
    util = s["VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)
    per[label] = {"fetch_bytes": f, "write_bytes": w, "bytes": f + w, "mfma_busy_cycles_all_simds": s["VALU_MFMA_BUSY_CYCLES"],
                  "kernel_cycles": cyc, "mfma_utilisation": util, "wave_cycles_waiting_frac": s["WAIT_ANY"] / s["WAVE_CYCLES"]}
    lines.append("%-10s grid %6d  dispatches %3d  fetch %.2f MB  write %.2f MB  total %.2f MB | kernel %.0f cycles  MFMA busy %.3f of 1024 SIMDs  "
                 "waves waiting %.2f of their cycles" % (label, grid, fetch[key][0], f / 1e6, w / 1e6, (f + w) / 1e6, cyc, util, s["WAIT_ANY"] / s["WAVE_CYCLES"]))
mean = sum(v["bytes"] for v in per.values()) / max(1, len(per))
srcnote = ("profiles/%s_pmc_summary.txt: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_* / --pmc TCC_* in separate passes over "
           "`python3 tools/insitu.py 40` (40 eager updates back to back: every dispatch in the update sequence; PMC collection does not "
           "survive hipGraph launches on this ROCm, and the profiler runs every dispatch in isolation), tools/prof_round.sh + "
           "tools/make_traffic.py" % tag)
json.dump({"source": srcnote, "kernel_source_sha256": kernel_source_hash(),
           "fetch_correction": "FETCH_SIZE x2 (MI355X guide: the counter tallies 64 B per 128-B request for 16 B/lane coalesced loads on gfx950)",
           "write_note": "WRITE_SIZE as reported (16-B/lane stores are calibrated; the 4-B/lane epilogue stores are not)",
           "what": "L2 <-> fabric traffic (Infinity Cache + HBM behind it), mean per dispatch",
           "per_kernel": per, "bytes_per_launch_mean": mean,
           "algorithmic_bytes_per_launch": "operands read once + outputs written once, per launch: k_dfwd<0> ~6.0 MB, k_dfwd<1> ~3.5 MB, k_dg bq ~6.5 MB, "
                                           "k_dg mid ~12 MB (the two Q layer-2 optimizer jobs: m, v, main, target read + written = 9.6 MB), k_dg pi ~7 MB (the policy's: 4.8 MB): "
                                           "mean ~7 MB (DESIGN.md section 4)"},
          open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
with open(os.path.join(ROOT, "profiles", "%s_pmc_summary.txt" % tag), "w") as fo:
    fo.write("PMC passes (separate rocprofv3 runs, eager launches, 40 dispatches per kernel): %s\n" % srcnote)
    fo.write("\n".join(lines) + "\n")
    fo.write("mean over the five launches: %.2f MB per launch\n\n" % (mean / 1e6))
    for name in ("pmc_FETCH_SIZE.txt", "pmc_WRITE_SIZE.txt", "pmc_sq.txt", "pmc_tcc.txt", "pmc_tcc_warm.txt"):
        if os.path.exists(os.path.join(src, name)):
            fo.write("---- %s%s\n%s\n" % (name, "  (L2 = TCC: requests / hits / misses / fabric read requests of 64 B per dispatch; 'warm' = the same "
                                                 "launch repeated back to back)" if "tcc" in name else "", open(os.path.join(src, name)).read()))
print("\n".join(lines))
print("mean %.2f MB per launch" % (mean / 1e6))
