"""dev: cfg3 solve-time spread over seeds for a list of (env, solver-option) variants, each in a fresh process.
    python tools/dev_knobs.py cfg3 0 8 "KTN_NEAR_CHUNK=15" "lp_check_every=32" ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, n0, n1 = sys.argv[1:4]
for var in sys.argv[4:] or [""]:
    env = dict(os.environ)
    opts = []
    for tok in var.split():
        k, v = tok.split("=")
        if k.startswith("KTN_"):
            env[k] = v
        else:
            opts.append(tok)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dev_seeds2.py"), name, n0, n1] + opts, env=env, capture_output=True, text=True)
    print("[%s] %s" % (var, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1]), flush=True)
