import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d["value"], d["checksum"], d["roofline"]["kernel"], round(d["roofline"]["frac"],4), d["train"]["value"])
for k in d["kernels"][:5]: print("  ", k["kernel"], k["launches_per_frame"], round(k["avg_us"],1), round(k["tflops"] or 0,1))
