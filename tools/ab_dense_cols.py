"""In-run A/B of the dense CG with 2..8 right-hand sides: the tile scheme with BT columns (csrc/cg_dense1.hip, d1m_*,
default) against round 3's route -- skinny MFMA product + fused update (MGP_CG_DENSE1_COLS=1).  One child process per
variant (switches are read at handle creation), two alternations.  usage: python tools/ab_dense_cols.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cases = [(int(a), int(b)) for a, b in (c.split('x') for c in sys.argv[1:])] or [(4096, 2), (4096, 5), (4096, 8), (2048, 5), (2048, 8), (8192, 5)]
for rnd in range(1 if len(sys.argv) > 1 else 2):
    for name, env in (json.loads(os.environ["AB_VARIANTS"]) if os.environ.get("AB_VARIANTS") else
                      (("skinny product + fused update (r03)", {"MGP_CG_DENSE1_COLS": "1"}), ("default route", {}))):
        for n, bt in cases:
            p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_probe_cg.py"), str(n), str(bt), "300"],
                               env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
            print(f"round {rnd} [{name}] " + (p.stdout.strip().splitlines()[-1] if p.returncode == 0 else "FAILED " + p.stderr[-500:]), flush=True)
