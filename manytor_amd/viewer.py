"""Optional host-side link to the reference's vispy viewer (plotting.py) -- SURVEY.md 8(f) rank 3.

The reference streams one UDP/JSON datagram per sub-step per env to localhost:5001
(manytor.py:94-101, :194-202, :246-249, :271-279; consumer plotting.py:27-87).  This module speaks that wire format
for a *sampled subset* of a batch, entirely off the step path: frames are built from `engine.route_trace` (the 25
sub-step joints_coordinates of the routes just taken) after the step has run.  It never spawns the viewer (vispy is
not a dependency); start `plotting.py` yourself if you want the picture.

Datagrams (JSON lists, NaN spelled the way Python's json does):
  init   [env_number, obj_number, 3, [rows, cols]]      multi-env   (manytor.py:94)
         [1, obj_number, 3]                             single env  (manytor.py:271)
  stop   [NaN, NaN, 2]                                  (manytor.py:277; Multienv's own stop sends raw bytes, :100 --
                                                         an upstream bug the viewer cannot parse; the JSON form is used)
  clear  [NaN, NaN, 4]                                  trajectory reset (manytor.py:247)
  frame  [id, NaN, flag,  jc(4x3)...,  points(Kx3)...,  ee(3)]  flattened (manytor.py:197-200).  The viewer restarts a
         trajectory when flag == 1 (plotting.py:84), but the reference itself never sends 1: it tests
         `trajectory.size == 3` AFTER appending the sub-step's row (manytor.py:190,196), so every frame it emits
         carries 0 (fixture F9, captured from the reference) and trajectories restart only on `clear`.  The drop-in
         classes do the same; `first=True` remains available to callers who want the viewer's restart.

Pinned by tests/golden/f9_viewer_frames.npz: the datagrams a reference Environment emitted for two steps.
"""
from __future__ import annotations

import json
import socket
import time

import numpy as np

MAX_DATAGRAM = 1024     # plotting.py:28 reads at most this many bytes


def encode_init(env_number, obj_number, env_shape=None) -> bytes:
    msg = [int(env_number), int(obj_number), 3]
    if env_shape is not None:
        msg.append([int(env_shape[0]), int(env_shape[1])])
    return json.dumps(msg).encode()


def encode_stop() -> bytes:
    return json.dumps([float("nan"), float("nan"), 2]).encode()


def encode_clear() -> bytes:
    return json.dumps([float("nan"), float("nan"), 4]).encode()


def encode_frame(env_id, joints, points, ee, first=False, digits=4) -> bytes:
    """One sub-step frame.  joints (4, 3) = joints_coordinates, points (K, 3), ee (3,).
    Coordinates are rounded to `digits` decimals so that K = 10 still fits the viewer's 1024-byte read."""
    rows = np.vstack([[float(env_id), np.nan, 1.0 if first else 0.0], np.asarray(joints, dtype=np.float64).reshape(-1, 3),
                      np.asarray(points, dtype=np.float64).reshape(-1, 3), np.asarray(ee, dtype=np.float64).reshape(1, 3)])
    flat = np.round(rows.reshape(-1), digits).tolist()
    flat[0] = int(env_id)
    data = json.dumps(flat).encode()
    if len(data) > MAX_DATAGRAM:
        raise ValueError(f"frame of {len(data)} bytes exceeds the viewer's {MAX_DATAGRAM}-byte datagram limit")
    return data


class ViewerLink:
    """UDP sender for the datagrams above."""

    def __init__(self, host="localhost", port=5001, frame_delay=0.0):
        self.dest = (host, port)
        self.frame_delay = float(frame_delay)      # the reference sleeps 6 ms per sub-step (manytor.py:202)
        self.sock = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        self.sent = 0

    def _send(self, data: bytes):
        self.sock.sendto(data, self.dest)
        self.sent += 1

    def init(self, env_number, obj_number, env_shape=None):
        self._send(encode_init(env_number, obj_number, env_shape))

    def clear(self):
        self._send(encode_clear())

    def stop(self):
        self._send(encode_stop())

    def frames(self, env_ids, traces, points, first=False):
        """traces (n, S, D, 3) from engine.route_trace, points (n, K, 3); sends S frames per env, sub-step major
        (all envs advance together, like the reference's lock-step loop)."""
        traces = np.asarray(traces)
        n, steps = traces.shape[0], traces.shape[1]
        for k in range(steps):
            for j in range(n):
                jc = traces[j, k]
                self._send(encode_frame(env_ids[j], jc[-4:] if jc.shape[0] >= 4 else jc, points[j], jc[-1],
                                        first=first and k == 0))
            if self.frame_delay:
                time.sleep(self.frame_delay)

    def close(self):
        self.sock.close()
