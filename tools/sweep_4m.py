#!/usr/bin/env python3
"""Why does the 4 194 304-arm step time differ by 20 % between boxes / runs?
Same process, same binary: (a) repeat the measurement on several freshly allocated arenas (physical placement),
(b) vary the row padding (HBM channel aliasing of the 64 concurrent row streams), (c) a plain 1 GiB copy beside each.
    python tools/sweep_4m.py > gpurun_out/sweep_4m.json"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import manytor_amd as m  # noqa: E402


def step_us(n, pad=None, steps=300, reps=3):
    if pad is None:
        os.environ.pop("MT_LD_PAD", None)
    else:
        os.environ["MT_LD_PAD"] = str(pad)
    e = m.StepEngine(n, 7)
    e.reset_random(1, 0)
    t0 = time.perf_counter()
    s = 0
    while time.perf_counter() - t0 < 0.2:
        e.rollout(50, 1, s % 50)
        e.reset_random(1, s)
        s += 50
        e.sync()
    out = []
    for _ in range(reps):
        e.reset_random(1, 0)
        e.sync()
        e.timer_start()
        e.rollout(50, 1, 0)
        out.append(e.timer_stop() * 1e3 / 50)
    ptr = e.device_ptr(m.lib.F_REWARD)[0]
    ld = e.ld
    e.close()
    return {"us": [round(v, 2) for v in out], "arena_reward_ptr": hex(ptr), "ld": ld}


def copy_gbs(nbytes=1 << 30):
    src = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda").normal_()
    dst = torch.empty_like(src)
    for _ in range(3):
        dst.copy_(src)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        dst.copy_(src)
    b.record()
    torch.cuda.synchronize()
    return round(2.0 * nbytes * 10 / (a.elapsed_time(b) * 1e-3) / 1e9, 1)


def smi():
    try:
        return subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True,
                              timeout=20).stdout[-1500:]
    except Exception as exc:  # noqa: BLE001
        return repr(exc)


def main():
    n = 4194304
    res = {"smi_before": smi(), "copy_gbs": [copy_gbs()], "fresh_arenas": [], "pads": {}, "sizes": {}}
    keep = []
    for k in range(4):                              # fresh arenas at different addresses (the blocker shifts them)
        res["fresh_arenas"].append(step_us(n))
        keep.append(torch.empty((64 << 20) * (k + 1), dtype=torch.uint8, device="cuda"))
    del keep
    torch.cuda.empty_cache()
    res["copy_gbs"].append(copy_gbs())
    for pad in (0, 64, 256, 512, 1024, 1088, 2048, 2112, 4096, 4160, 8256, 16448):
        res["pads"][pad] = step_us(n, pad=pad, reps=2)
    res["copy_gbs"].append(copy_gbs())
    for n2 in (1048576, 2097152, 3145728, 4194304, 8388608):
        res["sizes"][n2] = step_us(n2, reps=2)
    res["smi_after"] = smi()
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
