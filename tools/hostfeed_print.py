"""Print the step times of a bench_detail.json next to where each workload's ids came from (HBM ring / host arrays)."""
import json
import sys

d = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "bench_detail.json"))
print("head", round(d["ms_per_step"] * 1e3, 1), "us", d["config"].get("ids_fed_from"))
for k, v in d.get("sub", {}).items():
    c = v.get("config", {})
    print(k, None if v.get("ms_per_step") is None else round(v["ms_per_step"] * 1e3, 1), "us", v.get("value"),
          c.get("ids_fed_from"), "host enqueue ms/step:", c.get("host_enqueue_ms_per_step"), v.get("error"))
