import os, sys, time, json
sys.argv = ["bench.py", "--docs", "1250000", "--no-cpu-baseline", "--no-text-paths", "--no-stream-side", "--steps", "100", "--latency-batches", "5", "--latency-warmup", "1", "--exchange", "native", "--lanes", "2"]
os.environ["OI_BENCH_FORCE_DIST"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import openintel_amd.retriever as R
acc = {"c": 0.0, "py": 0.0, "n": 0}
orig = R.NativePipeline.submit
def timed(self, *a, **k):
    t0 = time.perf_counter()
    lib_submit = self.lib.oi_pipeline_submit
    class W:
        def __call__(s, *args):
            t1 = time.perf_counter()
            r = lib_submit(*args)
            acc["c"] += time.perf_counter() - t1
            return r
    # swap the bound function for this call
    self.lib.__dict__["oi_pipeline_submit"] = W()
    try:
        return orig(self, *a, **k)
    finally:
        self.lib.__dict__["oi_pipeline_submit"] = lib_submit
        acc["py"] += time.perf_counter() - t0
        acc["n"] += 1
R.NativePipeline.submit = timed
import runpy
try:
    runpy.run_path(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "bench.py"), run_name="__main__")
finally:
    sys.stderr.write("DBG submits %d: ctypes call %.1f us, whole Python submit %.1f us\n" % (acc["n"], acc["c"] / max(1, acc["n"]) * 1e6, acc["py"] / max(1, acc["n"]) * 1e6))
