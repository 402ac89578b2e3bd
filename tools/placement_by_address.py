#!/usr/bin/env python3
"""4 194 304 arms: us per step of successive fresh engines of one process next to the device address of their arena
(MT_F_REWARD's row: a fixed offset into it) -- the fast / slow alternation of profiles/r03_variants.md section 5 by address.
    python tools/placement_by_address.py [engines]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import manytor_amd as m  # noqa: E402

L = m.lib
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    e = m.StepEngine(4194304, 7)
    ptr = e.device_ptr(L.F_REWARD)[0]
    e.reset_random(1, 0)
    for _ in range(3):
        e.rollout(50, 1, 0)
    e.sync(); e.lap_times()
    for ep in range(4):
        e.reset_random(1, ep + 1)
        e.lap_begin(); e.rollout(50, 1, 0); e.lap_end()
    e.sync()
    us = sum(e.lap_times()) * 1e3 / 200
    print(f"engine {rep}: reward row at 0x{ptr:x}  (mod 2 MiB 0x{ptr % (1 << 21):x}, mod 1 GiB 0x{ptr % (1 << 30):x}, mod 4 GiB 0x{ptr % (1 << 32):x})  {us:6.1f} us per step", flush=True)
    e.close()
