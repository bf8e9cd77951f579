import json
v=None
for l in open("gpurun_out/ab_strip.jsonl"):
    d=json.loads(l)
    if "variant" in d: v=d["variant"]
    else: print(v, "strip %.2f us (dev %.2f)  whole %.2f" % (d["us_per_sweep_wall"], d["us_per_sweep_device"], d["whole_grid_us_per_sweep"]))
