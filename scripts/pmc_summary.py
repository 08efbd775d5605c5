"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only, separate runs) into
profiles/r01_pmc_fetch_write_summary.json and profiles/traffic_r01.json.
usage: pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv>
HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB * 1024 (MI355X_MICROARCH.md: gfx950 reports half of wide
coalesced reads; counters are in KiB)."""
import csv, json, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def collect(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"].split("(")[0]
        a = agg[n]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        a[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return agg


fetch = collect(sys.argv[1], "FETCH_SIZE")
write = collect(sys.argv[2], "WRITE_SIZE")
out = {}
for n, (cnt, tot, ns) in fetch.items():
    if not n.startswith("void asr::") and not n.startswith("asr::"):
        continue
    w = write.get(n, [1, 0.0, 0.0])
    out[n] = {"launches": cnt, "fetch_KiB_per_launch": tot / cnt, "write_KiB_per_launch": w[1] / max(1, w[0]),
              "avg_ns": ns / cnt}
json.dump(out, open(os.path.join(ROOT, "profiles", "r01_pmc_fetch_write_summary.json"), "w"), indent=1)


def per_launch(prefix):
    for n, v in out.items():
        if prefix in n:
            return int((2 * v["fetch_KiB_per_launch"] + v["write_KiB_per_launch"]) * 1024)
    return None


B, H = 32, 256
steps = 800 + 400 + 200 + 100
# algorithmic bytes per launch, averaged over the 4 launches of a step (2 directions):
#   forward: read gates 4H + write out H + act 8H + hprev H floats per (b,t,dir)  = 14H floats
#   backward: read act 8H + dout H, write dG 4H                                   = 13H floats
alg_fwd = steps * B * 2 * 14 * H * 4 // 4
alg_bwd = steps * B * 2 * 13 * H * 4 // 4
# exchange bytes per launch: forward {tag32,value32} granules (8 B) of R*H values per group-step = B*2*H per step; backward:
# all-gather of dG = 4x the values, as 1-bit-tagged floats (4 B each)
gran_fwd = steps * B * 2 * H * 8 // 4
gran_bwd = 2 * gran_fwd
traffic = {
    "note": "HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB * 1024: separate --pmc passes (rocprofv3 --pmc FETCH_SIZE / "
            "--pmc WRITE_SIZE, --kernel-trace only), FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide "
            "coalesced reads; mixed access widths here are uncalibrated). Average over the 4 launches per step (T=800/400/200/100). "
            "The exchange granules are written with sc1 / plain stores and polled with sc1 loads: they are L2 traffic that the "
            "memory-side counters also see when lines are written through.",
    "lstm_rec_bwd_bytes_per_launch": per_launch("lstm_rec_bwd_ag_kernel<256, 2") or per_launch("lstm_rec_bwd_kernel<256, 2>"),
    "lstm_rec_fwd_bytes_per_launch": per_launch("lstm_rec_fwd_kernel<256, 32, 2"),
    "algorithmic_bytes_per_launch": {"lstm_rec_bwd": alg_bwd, "lstm_rec_fwd": alg_fwd},
    "exchange_granule_bytes_per_launch": {"lstm_rec_fwd": gran_fwd, "lstm_rec_bwd": gran_bwd},
}
json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic_r01.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
