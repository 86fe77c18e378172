"""HBM traffic of the conv kernels from two rocprofv3 PMC passes over the bench command
(collected on their own, `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` with `--kernel-trace` only):

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/tr_fetch -o p --output-format csv -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/tr_write -o p --output-format csv -- python3 bench.py ...
    python tools/pmc_traffic.py gpurun_out/tr_fetch gpurun_out/tr_write profiles/r01c_conv_hbm_traffic.json

Units and corrections as MI355X_MICROARCH.md (HBM / rocprofv3) prescribes: both counters are in
KiB-scale units of 1024 B; on gfx950 FETCH_SIZE tallies the 128-B requests of wide (16 B / lane)
reads at 64 B, i.e. reports half the bytes -> doubled; WRITE_SIZE is uncalibrated and taken as is.
Infinity-Cache hits are included (memory-side L2 requests), so this is traffic below the L2."""
import json
import sys

import pandas as pd


def per_kernel(d, counter):
    df = pd.read_csv(f"{d}/p_counter_collection.csv")
    df = df[(df.Counter_Name == counter) & df.Kernel_Name.str.contains("conv")]
    return df.groupby("Kernel_Name").Counter_Value.agg(["sum", "count"])


def main(fetch_dir, write_dir, out, layer_launches=None):
    f, w = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    launches = int(f["count"].sum())
    assert launches == int(w["count"].sum()), "the two passes must profile the same command"
    fetch = 2.0 * float(f["sum"].sum()) * 1024.0  # gfx950 wide-read correction
    write = float(w["sum"].sum()) * 1024.0
    rows = {}
    for k in f.index:
        rows[k] = {"launches": int(f.loc[k, "count"]), "fetch_bytes_per_launch": 2.0 * 1024.0 * f.loc[k, "sum"] / f.loc[k, "count"],
                   "write_bytes_per_launch": 1024.0 * w.loc[k, "sum"] / w.loc[k, "count"] if k in w.index else None}
    # a conv LAYER may run as two kernels (the Winograd kernel launches the last, partially filled round
    # separately): bench.py counts layer launches, so the per-launch figure is normalised by those
    layer_launches = int(layer_launches) if layer_launches else launches
    res = {"conv_kernel_launches": launches, "conv_launches": layer_launches, "fetch_bytes_per_launch": fetch / layer_launches,
           "write_bytes_per_launch": write / layer_launches, "hbm_bytes_per_launch": (fetch + write) / layer_launches,
           "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) over `bench.py --steps 4 --warmup 1`; "
                     "FETCH_SIZE x 2 (gfx950 wide-read correction), units of 1024 B, summed over all conv kernels and divided by the number of conv launches of the profiled command (timed steps, warm-up and the estimator pass alike: every launch of a layer moves the same bytes)",
           "per_kernel": rows}
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "per_kernel"}, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:5])
