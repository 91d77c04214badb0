import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f))
    print(f, round(d.get("value")), round(d.get("kernel_ms")["pusch_decode_batch"], 3))
    for k, v in d.get("legs", {}).items():
        print("  ", k, round(v.get("value")), round(v.get("kernel_ms")["pusch_decode_batch"], 3), round(v.get("mean_iterations"), 2))
