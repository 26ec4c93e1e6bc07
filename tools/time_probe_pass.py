"""Wall time of the probe pass (rt_collect_gather_points, counters only) of the given scenes."""
import importlib, sys, time
sys.path.insert(0, '.')
rt = importlib.import_module("ray-tracer_amd")
for name in sys.argv[1:]:
    scene = rt.Scene("scenes/%s.json" % name)
    tracer = rt.Tracer(scene, device=0)
    best = 1e9
    for _ in range(3):
        tracer = rt.Tracer(scene, device=0)
        t0 = time.perf_counter()
        n = 0
        for c in range(len(tracer.cameras)):
            n += tracer.count_gather_points(c)
        tracer.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(name, "gather points", n, "probe pass %.2f ms" % (1e3 * best), flush=True)
