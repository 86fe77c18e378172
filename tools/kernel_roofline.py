"""profiles/<tag>_kernel_roofline.json: everything a reader needs to RECOMPUTE the rooflines of this repository from profiles/
alone -- per conv layer the algorithmic FLOPs and bytes of one launch with its duration alone, per kernel of
the benchmark step the launches and durations as they ran, and the rasteriser stage's algorithmic / counter bytes.

  python tools/kernel_roofline.py <out.json> --layers C2:<layers.txt> [C3:<...> C5:<...>] --kernel-stats <bench_kernel_stats.csv>
         [--conv-traffic <conv_hbm_traffic.json>] [--raster-traffic <raster_hbm_traffic.json>] [--bench-line <bench_line.json>]

<layers.txt> = stderr of ``tools/backbone_layers.py <arch> <cin> <prec> <batch>`` (HP_PROFILE_LAYERS=1: one line per conv launch:
name, k x k, stride, cin -> cout @ H x W (input size), K (padded reduction), microseconds alone, algorithmic TFLOP/s).
Algorithmic bytes of a conv launch (fp32 tensors; fp16 plan: 2 B): input N H W Cin + output N Ho Wo Cout (+ the residual read
of a block's second conv) + the weights once.  Peaks: MI355X_MICROARCH.md (fp16 MFMA 2516.6 TFLOP/s dense at 2.4 GHz x 256 CUs,
fp32 MFMA 157.3, HBM 8 TB/s)."""
import argparse
import csv
import json
import re

PEAK_F16, PEAK_F32, PEAK_HBM = 2516.6e12, 157.3e12, 8.0e12
LINE = re.compile(r"\[hp conv\]\s+(\S+)\s+(\d)x(\d) s(\d)\s+(\d+)->\s*(\d+) @\s*(\d+)x\s*(\d+)\s+K=\s*(\d+)\s+([\d.]+) us\s+([\d.]+) TFLOP/s")


def layers(path, batch, elem):
    rows = []
    for ln in open(path):
        m = LINE.search(ln)
        if not m:
            continue
        name, kh, kw, s, cin, cout, h, w, kpad, us, tf = m.groups()
        kh, kw, s, cin, cout, h, w = int(kh), int(kw), int(s), int(cin), int(cout), int(h), int(w)
        ho, wo = (h + s - 1) // s, (w + s - 1) // s
        flops = 2.0 * batch * ho * wo * cout * cin * kh * kw
        residual = name.endswith("conv2.weight")  # the second conv of a basic block adds the identity / shortcut
        nbytes = elem * batch * (h * w * cin + ho * wo * cout * (2 if residual else 1)) + elem * cout * cin * kh * kw
        us = float(us)
        rows.append(dict(layer=name, kernel=f"{kh}x{kw} s{s}", cin=cin, cout=cout, in_hw=[h, w], out_hw=[ho, wo], k_padded=int(kpad),
                         algorithmic_flops=flops, algorithmic_bytes=nbytes, residual_read=residual, us_alone=us,
                         algorithmic_tflops=flops / us / 1e6, algorithmic_gbs=nbytes / us / 1e3,
                         frac_f16_mfma_peak_algorithmic=flops / (us * 1e-6) / PEAK_F16, frac_hbm_peak_algorithmic=nbytes / (us * 1e-6) / PEAK_HBM))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--layers", nargs="*", default=[])  # WL:path:batch:elem_bytes:mfma_per_product
    ap.add_argument("--kernel-stats")
    ap.add_argument("--conv-traffic")
    ap.add_argument("--raster-traffic")
    ap.add_argument("--bench-line")
    a = ap.parse_args()
    res = {"peaks": {"fp16_mfma_flops": PEAK_F16, "fp32_mfma_flops": PEAK_F32, "hbm_bytes_per_s": PEAK_HBM,
                     "source": "/opt/skills/guides/MI355X_MICROARCH.md (chip-level parameters)"},
           "conv_layers": {}, "note": __doc__.split("\n\n")[2] if False else "see tools/kernel_roofline.py for the definitions"}
    for spec in a.layers:
        wl, path, batch, elem, per_prod = spec.split(":")
        rows = layers(path, int(batch), int(elem))
        tot_f, tot_b, tot_us = sum(r["algorithmic_flops"] for r in rows), sum(r["algorithmic_bytes"] for r in rows), sum(r["us_alone"] for r in rows)
        res["conv_layers"][wl] = {
            "batch_per_launch": int(batch), "bytes_per_element": int(elem), "fp16_mfma_per_product": int(per_prod), "launches_per_forward": len(rows),
            "algorithmic_flops_per_forward": tot_f, "algorithmic_bytes_per_forward": tot_b, "algorithmic_bytes_per_launch_mean": tot_b / max(len(rows), 1),
            "us_per_forward_layers_alone": tot_us, "algorithmic_tflops": tot_f / tot_us / 1e6 if tot_us else None,
            "frac_f16_mfma_peak_algorithmic": tot_f / (tot_us * 1e-6) / PEAK_F16 if tot_us else None,
            "frac_f16_mfma_peak_executed": int(per_prod) * tot_f / (tot_us * 1e-6) / PEAK_F16 if tot_us else None, "layers": rows}
    if a.kernel_stats:
        ks = []
        for r in csv.DictReader(open(a.kernel_stats)):
            ks.append(dict(kernel=r["Name"], calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3, total_ms=float(r["TotalDurationNs"]) / 1e6,
                           share=float(r["Percentage"]) / 100.0))
        res["bench_step_kernels"] = {"source": a.kernel_stats, "note": "rocprofv3 --kernel-trace --stats over bench.py (C2, two lanes): durations AS RUN beside the other lane",
                                     "kernels": sorted(ks, key=lambda k: -k["total_ms"])[:40]}
    if a.conv_traffic:
        t = json.load(open(a.conv_traffic))
        c2 = res["conv_layers"].get("C2")
        if c2:
            t["algorithmic_bytes_per_launch"] = c2["algorithmic_bytes_per_launch_mean"]
            t["counter_over_algorithmic"] = t["hbm_bytes_per_launch"] / c2["algorithmic_bytes_per_launch_mean"]
            json.dump(t, open(a.conv_traffic, "w"), indent=1)
        res["conv_hbm_traffic"] = {k: v for k, v in t.items() if k != "per_kernel"}
    if a.raster_traffic:
        res["rasteriser"] = json.load(open(a.raster_traffic))
    if a.bench_line:
        b = json.load(open(a.bench_line))
        res["bench_line"] = {k: b.get(k) for k in ("metric", "value", "unit", "ms_per_step", "dtype", "arithmetic", "config", "roofline", "stages") if k in b}
    json.dump(res, open(a.out, "w"), indent=1)
    print("wrote", a.out, {k: (len(v["layers"]), round(v["algorithmic_tflops"], 1)) for k, v in res["conv_layers"].items()})


if __name__ == "__main__":
    main()
