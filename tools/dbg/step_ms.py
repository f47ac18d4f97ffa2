import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1], "it/s", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 4), "sweep ms", round(d["roofline"]["avg_launch_ms"], 4))
