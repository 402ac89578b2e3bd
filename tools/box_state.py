#!/usr/bin/env python3
"""What state is this box's GPU in while the 4 194 304-arm step runs?  rocm-smi clocks / power / perf level sampled while a
background thread keeps the kernel busy, next to the measured us per step (two allocations) and a device copy rate.
    python tools/box_state.py > gpurun_out/box_state.json"""
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import manytor_amd as m  # noqa: E402


def smi(*args):
    try:
        return subprocess.run(["rocm-smi", *args], capture_output=True, text=True, timeout=30).stdout[-3000:]
    except Exception as e:  # noqa: BLE001
        return repr(e)


out = {"idle": smi("--showclocks", "--showpower", "--showperflevel", "--showmemuse")}
times = []
for k in range(2):
    e = m.StepEngine(4194304, 7)
    e.reset_random(1, 0)
    stop = False

    def spin():
        while not stop:
            e.rollout(50, 1, 0)
            e.sync()
    t = threading.Thread(target=spin)
    t.start()
    time.sleep(1.0)
    out[f"busy_{k}"] = smi("--showclocks", "--showpower")
    stop = True
    t.join()
    e.timer_start()
    e.rollout(50, 1, 0)
    times.append(round(e.timer_stop() * 1e3 / 50, 2))
    e.close()
out["us_per_step_4m"] = times
out["partition"] = smi("--showcomputepartition", "--showmemorypartition")
print(json.dumps(out, indent=1))
