import sys, time, json
sys.path.insert(0, '/root/repo')
import manytor_amd as m
out = {}
for n in (64, 1024):
    e = m.StepEngine(n, 7)
    e.reset_random(1, 0)
    for name, fn in (("sample_actions", lambda t: e.sample_actions(1, t)), ("step_random", lambda t: e.step_random(1, t)),
                     ("check_done", lambda t: e.check_done()), ("observe", lambda t: e.observe())):
        for t in range(300):
            fn(t)
        e.sync()
        t0 = time.perf_counter()
        for t in range(1000):
            fn(t)
        t1 = time.perf_counter()
        e.sync()
        t2 = time.perf_counter()
        out[f"{name}_n{n}"] = {"enqueue_us": round((t1 - t0) * 1e3, 2), "complete_us": round((t2 - t0) * 1e3, 2)}
    e.close()
print(json.dumps(out, indent=1))
