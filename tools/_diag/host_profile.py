"""Where does the host spend a 6x128 training step?  cProfile over bench.py's step loop (the configuration is bound by the launch rate)."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = ["bench.py", "--workload", os.environ.get("HP_WORKLOAD", "6x128"), "--steps", "40", "--warmup", "5", "--no-cpu-baseline", "--no-fp32", "--no-secondary", "--no-kernel-events"]
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "bench.py"), run_name="__main__")
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
