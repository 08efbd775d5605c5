"""Summarise the rocprofv3 passes of scripts/pmc_collect.sh into profiles/ (round 2).

usage: pmc_summary2.py <dir written by pmc_collect.sh> [tag]   ->  profiles/<tag>_pmc_summary.json, profiles/traffic_<tag>.json,
                                                                 profiles/<tag>_train_kernel_stats.csv

Per kernel and pass: launches, mean duration, mean counter values per dispatch, and the derived figures the north star
asks for:
  * MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs)   (MI355X_MICROARCH.md: the
    counter counts cycles summed over SIMDs -- 32 per v_mfma_f32_32x32x16_bf16; GRBM_GUI_ACTIVE is summed over the 8 XCDs);
  * issue-stall / parked shares of the wave cycles (SQ_WAIT_INST_ANY, SQ_WAIT_ANY, SQ_ACTIVE_INST_* over SQ_WAVE_CYCLES:
    quad-cycle units, disjoint buckets);
  * HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 (gfx950 reports half of wide coalesced reads) and the
    achieved GB/s = bytes / mean duration, next to the 8 TB/s HBM peak."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
csv.field_size_limit(1 << 30)


def short(name):
    n = name.split("(")[0].replace("void ", "")
    return n if len(n) < 120 else n[:117] + "..."


def collect(passdir):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0, 0.0]))
    for f in glob.glob(os.path.join(src, passdir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = short(r["Kernel_Name"])
            if "asr::" not in n:
                continue
            a = agg[n][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            a[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    out = {}
    for n, cs in agg.items():
        any_c = next(iter(cs.values()))
        out[n] = {"launches": any_c[0], "avg_ns": any_c[2] / any_c[0],
                  "counters_per_launch": {c: v[1] / v[0] for c, v in cs.items()}}
    return out


summary = {"note": __doc__.split("\n\n")[2] if False else "see scripts/pmc_summary2.py for the formulas; one counter group per rocprofv3 "
           "pass (--kernel-trace only), passes listed by directory"}
NSIMD = 256 * 4
for passdir in sorted(os.listdir(src)):
    if not os.path.isdir(os.path.join(src, passdir)):
        continue
    d = collect(passdir)
    if not d:
        continue
    for n, v in d.items():
        c = v["counters_per_launch"]
        der = {}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE", 0) > 0:
            cyc = c["GRBM_GUI_ACTIVE"] / 8.0
            der["mfma_busy_fraction_of_simd_cycles"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * NSIMD)
            der["effective_clock_GHz"] = cyc / v["avg_ns"]
            der["counter_saturated"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] >= 2147483648.0
        if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"] > 0:
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
                if k in c:
                    der[k.lower() + "_share_of_wave_cycles"] = c[k] / c["SQ_WAVE_CYCLES"]
        if der:
            v["derived"] = der
    summary[passdir] = d

# HBM traffic of whole train steps: FETCH and WRITE passes joined per kernel
fetch, write = summary.get("step_fetch", {}), summary.get("step_write", {})
traffic_rows = {}
for n, v in fetch.items():
    f = v["counters_per_launch"].get("FETCH_SIZE", 0.0)
    w = write.get(n, {}).get("counters_per_launch", {}).get("WRITE_SIZE", 0.0)
    nbytes = (2.0 * f + w) * 1024.0
    traffic_rows[n] = {"launches": v["launches"], "avg_us": v["avg_ns"] / 1e3, "hbm_bytes_per_launch": nbytes,
                       "achieved_GBps": nbytes / v["avg_ns"], "frac_of_8TBps": nbytes / v["avg_ns"] / 8000.0}
summary["step_hbm_per_kernel"] = traffic_rows
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
json.dump(summary, open(os.path.join(ROOT, "profiles", "%s_pmc_summary.json" % tag), "w"), indent=1, sort_keys=True)


def family(prefix):
    """Launch-weighted mean over ALL instantiations of a kernel family: (bytes per launch, launches, {instantiation: row}).
    (Round 4's summary took the FIRST matching instantiation: for the forward that was lstm_rec_fwd4_kernel<10> -- the first layer
    alone, T = 800, 402 MB -- set against the mean of all four launches, 270 MB: the "1.49x" of VERDICT r04 was that mismatch.)"""
    rows = {n: v for n, v in traffic_rows.items() if prefix in n}
    nl = sum(v["launches"] for v in rows.values())
    if not nl:
        return None, 0, {}
    return int(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in rows.values()) / nl), nl, rows


B, H, F = 32, 256, 80
T_layers = [800, 400, 200, 100]
unit_steps = [t * B * 2 * H for t in T_layers]
# bytes per unit-step since round 4 (20-byte split records).  Forward: writes out, h_prev, {i,j,f,o}, c = 7 floats; layers >= 2 also
# read the projection x.K_x + b (4 floats); the first layer's projection runs inside the kernel (<10>): it reads the frames instead
# (80 floats per row and step).  BPTT: reads {i,j,f,o} + c + dout (6 floats) and writes dG (4 floats).
alg_fwd = unit_steps[0] * 7 * 4 + B * T_layers[0] * F * 4 + sum(u * 11 * 4 for u in unit_steps[1:])
alg_bwd = sum(u * 10 * 4 for u in unit_steps)
fb, fn, frows = family("lstm_rec_fwd4_kernel")
# (the encoder's four launches only: <true, ..> = REC32 is the decoder's LM-chain BPTT on the same kernel since round 5 -- one more
#  launch per step over 120 steps of 32 rows, reported under per_instantiation but not part of the family bench.py times)
bb, bn, brows = family("lstm_rec_bwd4_kernel<false")
_, _, lmrows = family("lstm_rec_bwd4_kernel<true")
brows = dict(brows, **lmrows)
traffic = {
    "note": "HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB * 1024 from separate rocprofv3 --pmc FETCH_SIZE / --pmc "
            "WRITE_SIZE passes (--kernel-trace only) over `bench.py --steps 3 --warmup 2 --no-cpu-baseline`; FETCH_SIZE doubled per "
            "MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads; mixed access widths are uncalibrated).  Launch-weighted "
            "mean over the 4 launches per step (T = 800/400/200/100) and over every instantiation of the family.",
    "lstm_rec_bwd_bytes_per_launch": bb,
    "lstm_rec_fwd_bytes_per_launch": fb,
    "algorithmic_bytes_per_launch": {"lstm_rec_bwd": alg_bwd // 4, "lstm_rec_fwd": alg_fwd // 4},
    "ratio_measured_over_algorithmic": {"lstm_rec_bwd": (bb / (alg_bwd / 4.0)) if bb else None, "lstm_rec_fwd": (fb / (alg_fwd / 4.0)) if fb else None},
    "per_instantiation": {n: {"launches": v["launches"], "hbm_bytes_per_launch": int(v["hbm_bytes_per_launch"]), "avg_us": v["avg_us"]}
                          for n, v in list(frows.items()) + list(brows.items())},
    "algorithmic_bytes_by_layer": {"lstm_rec_fwd": [unit_steps[0] * 28 + B * T_layers[0] * F * 4] + [u * 44 for u in unit_steps[1:]],
                                   "lstm_rec_bwd": [u * 40 for u in unit_steps]},
}
json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic_%s.json" % tag), "w"), indent=1)
for f in glob.glob(os.path.join(src, "step_stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(ROOT, "profiles", "%s_train_kernel_stats.csv" % tag))
print(json.dumps({k: v for k, v in summary.items() if k in ("gemm_split_sq", "gemm_exact_sq", "lstm_sq")}, indent=1)[:6000])
print(json.dumps(traffic, indent=1))
