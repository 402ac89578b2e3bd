import sys, time, json
sys.path.insert(0, '/root/repo')
import manytor_amd as m
out = {}
for name, kw in (("ref4", dict()), ("dh7", dict(dh_table=m.DH7_TABLE, radius=92.6)), ("rt4", dict(specialize=False))):
    e = m.StepEngine(1 << 20, 7, **kw)
    e.reset_random(1, 0)
    for _ in range(3):
        e.rollout_fused(50, 1, 0)
        e.reset_random(1, 1)
    e.sync()
    tot = 0.0
    for r in range(6):
        e.reset_random(1, r)
        e.sync()
        e.timer_start()
        e.rollout_fused(50, 1, 0)
        tot += e.timer_stop()
    out[name] = round(tot * 1e3 / 300, 2)
    e.close()
print(json.dumps(out))
