import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d["cdgp_same_size"]
print(sys.argv[1], round(c["prior_kl_64_probes_ms"], 2), round(c["logdet_gradient_64_probes_ms"], 2), c["probe_cg_iterations"], repr(c["prior_kl"]), round(c["cg_ms"], 3))
