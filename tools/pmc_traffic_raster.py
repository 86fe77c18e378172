"""Per-workload HBM traffic of the reference-state rasteriser: counter bytes per CALL against the algorithmic bytes.

    python tools/pmc_traffic_raster.py <out.json> <workload>:<fetch_dir>:<write_dir>:<stage_workload.json> [...]

Each workload was profiled on its own (``HP_STAGE_WORKLOADS=<wl> HP_STAGE_ONLY=raster HP_STAGE_MSAA=1 HP_STAGE_ANISO=1
tools/stage_workload.py`` under ``rocprofv3 --kernel-trace --pmc FETCH_SIZE`` / ``WRITE_SIZE``, separate passes), so every
raster_* dispatch of a pass belongs to that workload; besides the timed calls the script renders one depth-only pass
(coverage count), which is subtracted by kernel name where possible and otherwise counted (stated in the output).  Units and
corrections as MI355X_MICROARCH.md prescribes: 1024-B units, FETCH_SIZE x 2 (gfx950 wide reads), WRITE_SIZE as is."""
import json
import sys

import pandas as pd


def load(d, counter):
    df = pd.read_csv(f"{d}/p_counter_collection.csv")
    return df[(df.Counter_Name == counter) & df.Kernel_Name.str.contains("raster")]


def main():
    out, specs = sys.argv[1], sys.argv[2:]
    res = {"method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, one pass per counter AND per workload over tools/stage_workload.py "
                     "(HP_STAGE_ONLY=raster, reference render state); FETCH_SIZE x 2 (gfx950 wide-read correction), 1024-B units; "
                     "bytes per call = sum over the raster_xform / raster_setup / raster_kernel dispatches of the timed calls / calls",
           "workloads": {}}
    for spec in specs:
        wl, fd, wd, sj = spec.split(":")
        stage = json.load(open(sj))[wl]
        calls = stage["calls_each"]
        f, w = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
        per_kernel = {}
        tot_f = tot_w = 0.0
        for name, g in f.groupby("Kernel_Name"):
            short = name.replace("void hp::", "").replace("hp::", "").split("(")[0]
            gw = w[w.Kernel_Name == name]
            n = len(g)
            # the reference-state band kernel is raster_kernel<5, ...>; the depth-only coverage pass launches raster_kernel<1, ...> once
            timed = not short.startswith("raster_kernel<1")
            fb, wb = 2.0 * 1024.0 * float(g.Counter_Value.sum()), 1024.0 * float(gw.Counter_Value.sum())
            per_kernel[short] = {"dispatches": int(n), "fetch_bytes": fb, "write_bytes": wb, "counted": timed}
            if timed:
                tot_f += fb; tot_w += wb
        # xform / bin dispatches of the one depth-only pass are in the totals: (calls + 1) passes launched them
        scale = calls / (calls + 1.0)
        xb = sum(v["fetch_bytes"] + v["write_bytes"] for k, v in per_kernel.items() if k.startswith(("raster_xform", "raster_bin", "raster_setup")))
        band = sum(v["fetch_bytes"] + v["write_bytes"] for k, v in per_kernel.items() if k.startswith("raster_kernel<5"))
        counter_per_call = (band + xb * scale) / calls
        alg = stage["raster"]["algorithmic_MB"] * 1e6
        res["workloads"][wl] = {"views": stage["views"], "calls": calls, "us_per_call": stage["raster"]["us"], "algorithmic_bytes_per_call": alg,
                                "counter_bytes_per_call": counter_per_call, "fetch_bytes_per_call": (tot_f - (1 - scale) * sum(
                                    v["fetch_bytes"] for k, v in per_kernel.items() if k.startswith(("raster_xform", "raster_bin", "raster_setup")))) / calls,
                                "counter_over_algorithmic": counter_per_call / alg, "per_kernel": per_kernel}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "per_kernel"} for k, v in res["workloads"].items()}, indent=1))


if __name__ == "__main__":
    main()
