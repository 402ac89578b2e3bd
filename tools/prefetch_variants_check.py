import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
import manytor_amd as m
# bit-identity of the three variants (static + runtime tables, K = 7 and K = 32, ragged n)
for kw, n, k in ((dict(), 100003, 7), (dict(specialize=False), 5000, 32), (dict(dh_table=m.DH7_TABLE, radius=92.6), 70001, 7)):
    outs = []
    for pf in (0, 1):
        os.environ["MT_PREFETCH"] = str(pf)
        e = m.StepEngine(n, k, pickup_tol=20.0, **kw)
        e.reset_random(3, 0)
        e.rollout(6, 3, 0)
        acts = np.random.RandomState(1).randint(-180, 180, size=(n, e.dof)).astype(np.float32)
        e.step(acts)
        outs.append({f: e.get(getattr(m.lib, f)) for f in ("F_GOALS", "F_ALIVE", "F_TOTAL_REWARD", "F_POINTS", "F_OBS", "F_REWARD", "F_DONE", "F_EE", "F_DONE_BITS")})
        e.close()
    for f in outs[0]:
        assert np.array_equal(outs[0][f], outs[1][f]), (kw, f, 1)
print("variants bit-identical")
