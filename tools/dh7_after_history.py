#!/usr/bin/env python3
"""Why does the 7-joint 1 M-arm step read 3 x slower at the end of bench.py's secondary list than in a fresh process?
Replays pieces of bench.py's allocation history and times the same configuration after each."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import manytor_amd as m  # noqa: E402


def dh7(label):
    us, name = bench.time_step_launches(m, 1048576, m.DH7_TABLE, 92.6, 7, 0, 0x5EED, steps=600)
    print(f"{label:50s} dh7 1M: {us:7.2f} us", flush=True)


def ref(n, steps=600):
    us, _ = bench.time_step_launches(m, n, m.REF_DH_TABLE, 51.3, 7, 0, 0x5EED, steps=steps)
    return us


dh7("fresh process")
m.stream_probe(4194304, 4, 7, reps=20)
dh7("after stream_probe(4 M)")
m.stream_probe(1048576, 4, 7, reps=20)
dh7("after stream_probe(1 M)")
print("ref 1M", round(ref(1048576), 2))
dh7("after a 1 M reference engine")
print("ref 524288", round(ref(524288), 2))
dh7("after a 524 288 engine")
print("ref 4M", round(ref(4194304, 300), 2))
dh7("after a 4 M engine")
dh7("again")
time.sleep(2.0)
dh7("after 2 s of idle")
e = m.StepEngine(1048576, 7, dh_table=m.DH7_TABLE, radius=92.6)
d = e.dispatch()
print(d["chains"], d["step"])
e.close()
