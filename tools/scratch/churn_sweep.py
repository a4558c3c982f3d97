import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for k in ("1000", "2000", "4000", "8000", "16000", "40000"):
    env = dict(os.environ, CSF_REBIN_CHURN=k)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "churn_rate.py"), "600"], env=env, capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print(k, "static", round(d["static"]["us_per_tick"], 1), "incremental", round(d["incremental"]["us_per_tick"], 1), "host", round(d["incremental"]["host_us_in_population_calls"], 1), "ratio", round(d["incremental_over_static"], 3), flush=True)
    except Exception as ex:
        print(k, "failed", r.stdout[-300:], r.stderr[-300:], flush=True)
