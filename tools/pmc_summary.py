#!/usr/bin/env python3
"""tools/pmc_summary.py — turns rocprofv3 counter-collection CSVs into the tracked summaries under profiles/.

  python tools/pmc_summary.py traffic <fetch.csv> <write.csv> [log_n]   -> profiles/ntt_traffic.json (read by bench.py `roofline.traffic`)
  python tools/pmc_summary.py sq <out.json> <counter csv> [<counter csv> ...] -> per-kernel averages of every counter found

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are collected in SEPARATE passes
(TCC slots), values are KiB per dispatch, FETCH_SIZE is doubled on gfx950 (128-B requests tallied at 64 B), WRITE_SIZE is exact.
The first dispatch of each kernel is dropped (cold tables)."""
import csv
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)(<[^>(]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def rows(path):
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            yield r


def per_kernel(paths):
    acc = defaultdict(lambda: defaultdict(list)); dur = defaultdict(list); meta = {}
    for p in paths:
        seen = defaultdict(int)
        for r in rows(p):
            k = short(r["Kernel_Name"]); c = r["Counter_Name"]
            seen[(k, c)] += 1
            if seen[(k, c)] == 1 and k.startswith(("k_ntt", "k_leaf", "k_hash")):
                continue                                     # warm-up dispatch
            acc[k][c].append(float(r["Counter_Value"]))
            dur[(k, c)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            meta[k] = {"grid": int(r["Grid_Size"]), "workgroup": int(r["Workgroup_Size"]), "lds_bytes": int(r["LDS_Block_Size"]), "vgprs": int(r["VGPR_Count"]), "sgprs": int(r["SGPR_Count"])}
    return acc, dur, meta


def ntt_dispatches(path, counter):
    """(kernel, value, ns) of every k_ntt_* dispatch in launch order."""
    out = []
    for r in rows(path):
        k = short(r["Kernel_Name"])
        if k.startswith("k_ntt") and r["Counter_Name"] == counter:
            out.append((int(r["Dispatch_Id"]), k.split("<")[0], float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return [x[1:] for x in sorted(out)]


def traffic(fetch_csv, write_csv, log_n=23):
    per = (2 if log_n > 20 else (1 if log_n > 10 else 0)) + 1          # launches per transform: strided passes + the last pass
    kib = 1024.0

    def per_transform(path, counter):
        d = ntt_dispatches(path, counter)
        chunks = [d[i:i + per] for i in range(0, len(d) - per + 1, per)]
        chunks = chunks[1:] if len(chunks) > 1 else chunks               # the first transform also fills the plan's tables: dropped
        assert chunks and all(c[-1][0] == "k_ntt_last" for c in chunks), "unexpected dispatch sequence"
        names = [c[0] for c in chunks[0]]
        vals = [sum(c[i][1] for c in chunks) / len(chunks) for i in range(per)]
        ns = [sum(c[i][2] for c in chunks) / len(chunks) for i in range(per)]
        return names, vals, ns, len(chunks)
    names, fk, _, nf = per_transform(fetch_csv, "FETCH_SIZE")
    _, wk, ns, nw = per_transform(write_csv, "WRITE_SIZE")
    fetch = 2.0 * sum(fk) * kib; write = sum(wk) * kib
    out = {f"ntt_2^{log_n}_bytes_per_transform": fetch + write,
           "detail": {"launches_per_transform": names, "FETCH_SIZE_KiB_per_launch_raw": fk, "WRITE_SIZE_KiB_per_launch": wk, "avg_ns_per_launch": ns, "transforms_averaged": [nf, nw],
                      "correction": "FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE exact",
                      "fetch_bytes": fetch, "write_bytes": write, "algorithmic_bytes": 64 * (1 << log_n),
                      "ratio_to_algorithmic": (fetch + write) / (64 * (1 << log_n)),
                      "source": f"{os.path.basename(fetch_csv)}, {os.path.basename(write_csv)} (rocprofv3 --pmc, separate passes, tools/kern_once.py ntt); derived by tools/pmc_summary.py"}}
    path = os.path.join(ROOT, "profiles", "ntt_traffic.json")
    prev = {}
    if os.path.exists(path):
        try:
            prev = {k: v for k, v in json.load(open(path)).items() if k.startswith("ntt_2^") and k != f"ntt_2^{log_n}_bytes_per_transform"}
        except Exception:
            prev = {}
    json.dump({**prev, **out}, open(path, "w"), indent=1)
    print(json.dumps(out))


def sq(out_json, paths):
    acc, dur, meta = per_kernel(paths)
    res = {}
    for k, cs in acc.items():
        if not k.startswith(("k_ntt", "k_leaf", "k_hash", "k_fri", "k_ali", "k_tr_hash")):
            continue
        d = {c: sum(v) / len(v) for c, v in cs.items()}
        d["dispatches_averaged"] = max(len(v) for v in cs.values())
        ds = [x for (kk, c), vs in dur.items() if kk == k for x in vs]
        d["avg_ns"] = sum(ds) / len(ds)
        d.update(meta.get(k, {}))
        if "SQ_ACTIVE_INST_VALU" in d and "SQ_BUSY_CYCLES" in d and d["SQ_BUSY_CYCLES"]:
            d["note"] = "SQ_ACTIVE_INST_* / SQ_WAIT_* / SQ_WAVE_CYCLES count quad-cycles summed over waves; SQ_BUSY_CYCLES over shader engines"
        if d.get("SQ_WAVE_CYCLES"):
            for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_LDS"):
                if c in d:
                    d[c + "_per_WAVE_CYCLES"] = d[c] / d["SQ_WAVE_CYCLES"]
        if d.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_bank_conflict_frac"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
        res[k] = d
    json.dump(res, open(out_json, "w"), indent=1, sort_keys=True)
    print(json.dumps({k: {c: round(v, 4) if isinstance(v, float) else v for c, v in d.items() if "per_WAVE" in c or c in ("avg_ns", "lds_bank_conflict_frac", "SQ_INSTS_VALU")} for k, d in res.items()}))


if __name__ == "__main__":
    if sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 23)
    elif sys.argv[1] == "sq":
        sq(sys.argv[2], sys.argv[3:])
