import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["steps"], d["warmup"], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d.get("roofline_burst",{}).get("avg_launch_ms"), d["roofline_fp64"]["frac"])
print(d["voice_mix"]["ms_per_block"], d["supersaw_mix"]["ms_per_block"], d["supersaw_mix"]["roofline_fp64"]["frac"], d["voice_mix"]["roofline_fp64"]["frac"])
print(d.get("rendered"), d.get("value_with_d2h"))
