"""Scratch: cProfile of bench.py's host side (where does a step spend host time?).  A short capture makes the
GPU work negligible, so ms_per_step ~ host time per step."""
import cProfile, pstats, sys, io
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.argv = ["bench.py", "--steps", "400", "--warmup", "20", "--no-cpu-baseline", "--sample-rate", "2.5e6", "--seconds", "5"]
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(s.getvalue()[:9000], file=sys.stderr)
