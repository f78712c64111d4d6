# small-batch timings: config 2 (cum, count, hits), config 3 (cum, count, hits)
import sys, os, subprocess
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for cfg in ("2","3"):
    for mode in ("cum","count","hits"):
        r=subprocess.run([sys.executable, os.path.join(ROOT,"profiles","recipes","prof_cfg.py"), cfg, mode, "7"], capture_output=True, text=True)
        print([l for l in r.stdout.splitlines() if l.startswith("config")][-1], flush=True)
