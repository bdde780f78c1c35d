"""Reduce the per-kernel PMC summaries of tools/profile_round.sh to profiles/rNN_traffic.json (bytes per launch; read by bench.py).
usage: traffic_json.py <dir with *_pmc_*_per_kernel.csv> <out.json>"""
import csv, json, os, sys
src, out = sys.argv[1], sys.argv[2]

def load(name):
    p = os.path.join(src, name)
    if not os.path.exists(p): return {}
    d = {}
    for r in csv.reader(open(p)):
        if len(r) < 3 or r[0] == "Kernel_Name": continue
        d[r[0]] = (int(r[1]), float(r[2]))
    return d

def pick(d, key):
    hits = [(k, v) for k, v in d.items() if key in k]
    return max(hits, key=lambda kv: kv[1][0])[1] if hits else (0, None)

res = {"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only; tools/profile_round.sh), mean per dispatch, "
                "counter unit KB (x 1024 = bytes).  On gfx950 FETCH_SIZE reads exactly 1/2 of the bytes of a wide (16 B/lane) coalesced stream "
                "(MI355X_MICROARCH.md section HBM); other access widths are uncalibrated.  k_pdhg_* (cfg3: LP resident in the Infinity Cache, 8-byte "
                "and 4-byte loads) and k_spmv_tiled (8-byte value and 2-byte index loads) are reported RAW; k_sep_eval_blk (16 B/lane double2 "
                "stream) is reported with the x2 correction, as in round 1."}
for tag, prefix, kern, corr in (("k_pdhg_x", "bench", "k_pdhg_x_packed", 1.0), ("k_pdhg_y", "bench", "k_pdhg_y_packed", 1.0),
                                ("k_spmv_tiled", "spmv_hbm", "k_spmv_tiled", 1.0), ("k_y_epilogue", "spmv_hbm", "k_y_epilogue", 1.0),
                                ("k_x_epilogue", "spmv_hbm", "k_x_epilogue", 1.0), ("k_sep_eval_blk", "sweep_hbm", "k_sep_eval_blk", 2.0)):
    nf, f = pick(load(prefix + "_pmc_FETCH_SIZE_per_kernel.csv"), kern)
    nw, w = pick(load(prefix + "_pmc_WRITE_SIZE_per_kernel.csv"), kern)
    if f is None: continue
    res[tag] = {"kernel": kern, "dispatches": nf, "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "fetch_correction": corr,
                "bytes_per_launch": 1024.0 * (corr * f + (w or 0.0))}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v["bytes_per_launch"] for k, v in res.items() if k != "_note"}))
