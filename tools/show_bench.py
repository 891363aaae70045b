"""The figures of one bench.py line that the notes quote: python tools/show_bench.py line.json"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"])
print("roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "traffic", "avg_launch_us", "valu_busy_frac", "valu_busy_source") if k in d["roofline"]})
sj = d.get("scale_job")
if sj:
    print("fill", sj["fill"]["ms"], "grid", sj["grid_aterms"]["ms"], sj["grid_aterms"]["ms_evaluation_median"], "fresh",
          {k: sj["grid_aterms_fresh"][k] for k in ("ms", "ms_update_pairs", "ms_first_evaluation", "fused_fallbacks")})
rs = d.get("roofline_sweep")
if rs:
    print("sweep", rs["kernel"][:40], rs["device_ms"], rs["frac"], rs.get("valu_busy_frac"), rs["traffic"])
ss = d.get("extra", {}).get("sampler_sweep")
if ss:
    print("samplea", {k: ss["samplea"][k] for k in ("seconds", "seconds_best", "seconds_same_pairs_median", "seconds_cache_hit_median", "aterms_evaluations")})
    print("sampleb", ss["sampleb"])
    print("cpu", (d.get("cpu_baseline") or {}).get("value"), ss["samplea"].get("cpu_reference", {}).get("seconds"))
print(d.get("extra", {}).get("sampler_error"))
