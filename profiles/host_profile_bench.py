"""Scratch: cProfile of bench.py's host side (where does a step spend host time?)."""
import cProfile, pstats, sys, io
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.argv = ["bench.py", "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--seconds", "60"]
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
print(s.getvalue()[:4000], file=sys.stderr)
