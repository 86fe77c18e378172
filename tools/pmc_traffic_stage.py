"""HBM traffic of the rasteriser / crop kernels from two rocprofv3 PMC passes over tools/stage_workload.py
(`--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, each with `--kernel-trace` only), units and corrections as
MI355X_MICROARCH.md prescribes (1024-B units; FETCH_SIZE x 2 on gfx950 for wide reads; WRITE_SIZE as is):

    python tools/pmc_traffic_stage.py <fetch_dir> <write_dir> <stage_workload.json> <out.json>

Per kernel: launches, bytes per launch; per stage and workload: counter bytes per CALL next to the algorithmic bytes."""
import json
import sys

import pandas as pd

PAT = "raster|crop"


def per_kernel(d, counter):
    df = pd.read_csv(f"{d}/p_counter_collection.csv")
    df = df[(df.Counter_Name == counter) & df.Kernel_Name.str.contains(PAT)]
    return df


def main(fetch_dir, write_dir, stage_json, out):
    f, w = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    stage = json.load(open(stage_json))
    rows = {}
    for name, g in f.groupby("Kernel_Name"):
        gw = w[w.Kernel_Name == name]
        short = name.replace("hp::(anonymous namespace)::", "").replace("void hp::", "").split("(")[0]
        rows[short] = {"launches": int(len(g)), "fetch_bytes_per_launch": 2.0 * 1024.0 * float(g.Counter_Value.mean()),
                       "write_bytes_per_launch": 1024.0 * float(gw.Counter_Value.mean()) if len(gw) else None,
                       "fetch_bytes_total": 2.0 * 1024.0 * float(g.Counter_Value.sum()),
                       "write_bytes_total": 1024.0 * float(gw.Counter_Value.sum()) if len(gw) else None}
    res = {"method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/stage_workload.py; FETCH_SIZE x 2 "
                     "(gfx950 wide-read correction), units of 1024 B; the launches of C2 and C3 calls are pooled per kernel "
                     "(per-launch means), the per-call split below uses the dispatch order",
           "per_kernel": rows, "stage_workload": stage}
    # per call: dispatches in order; a raster call = xform + bin + raster_kernel, a crop call = one crop kernel
    for tag, df in (("fetch", f), ("write", w)):
        df = df.sort_values("Dispatch_Id") if "Dispatch_Id" in df.columns else df
        res[f"{tag}_dispatch_count"] = int(len(df))
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "stage_workload"}, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:5])
