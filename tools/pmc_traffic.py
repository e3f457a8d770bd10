"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counter_collection CSVs) into per-kernel HBM traffic.
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per 128-B fabric read request, i.e. half the
bytes of a coalesced stream -> doubled; WRITE_SIZE is exact.  Units of both counters: KiB.
    python tools/pmc_traffic.py <fetch.csv> <write.csv> <out.json> ["<command that was profiled>"] [<mfma pass csv>]
The output records the sha of the kernel sources it was measured on (`_meta.csrc_sha`): bench.py quotes
`roofline.traffic` from the file only while that sha matches the sources it runs.
"""
import collections, csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arfe_amd.build import sources_sha

def per_kernel(path, ctr):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == ctr:
            d[r['Kernel_Name']].append(float(r['Counter_Value']))
    return d

f = per_kernel(sys.argv[1], 'FETCH_SIZE')
w = per_kernel(sys.argv[2], 'WRITE_SIZE')
out = {}
for k in f:
    fk = sum(f[k]) / len(f[k]); wk = sum(w.get(k, [0.0])) / max(len(w.get(k, [0.0])), 1)
    out[k] = {"launches": len(f[k]), "fetch_size_kib_raw": round(fk, 1), "write_size_kib": round(wk, 1),
              "hbm_bytes_per_launch": int((2.0 * fk + wk) * 1024)}
# optional third pass: matrix-pipe counters (SQ_VALU_MFMA_BUSY_CYCLES counts cycles of all SIMDs, SQ_BUSY_CYCLES the cycles
# a shader engine was busy, summed over the 32 SEs; MI355X_MICROARCH.md "rocprofv3 PMC slots")
if len(sys.argv) > 5:
    ctrs = ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_INSTS_VALU_MFMA_MOPS_BF16', 'SQ_INSTS_VALU_MFMA_MOPS_F32',
            'SQ_WAVE_CYCLES', 'SQ_INSTS_VALU')
    vals = {c: per_kernel(sys.argv[5], c) for c in ctrs}
    for k in out:
        m = {c: (sum(vals[c][k]) / len(vals[c][k]) if vals[c].get(k) else None) for c in ctrs}
        out[k]["mfma"] = {c.lower(): (round(v, 1) if v is not None else None) for c, v in m.items()}
        if m['SQ_VALU_MFMA_BUSY_CYCLES'] and m['SQ_BUSY_CYCLES']:
            # matrix pipe busy, as a fraction of (kernel duration x 1024 SIMDs): SQ_BUSY_CYCLES / 32 SEs ~ kernel cycles
            out[k]["mfma"]["busy_frac_of_all_simds"] = round(m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['SQ_BUSY_CYCLES'] / 32.0 * 1024.0), 4)
kern = dict(sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]))
meta = {"csrc_sha": sources_sha(), "command": sys.argv[4] if len(sys.argv) > 4 else "",
        "formula": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB, separate --pmc passes; mfma.busy_frac_of_all_simds = "
                   "SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 x 1024 SIMDs)"}
json.dump({"_meta": meta, "kernels": kern}, open(sys.argv[3], 'w'), indent=1)
print("wrote", sys.argv[3], len(out), "kernels")
