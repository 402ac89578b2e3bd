"""CPU-only: the UDP/JSON wire format of the reference's viewer (manytor.py:94-101,194-202,246-249,271-279 as
consumed by plotting.py:27-87), produced by manytor_amd/viewer.py."""
import json
import math
import socket

import numpy as np

from manytor_amd import viewer as V


def parse(data):
    """What plotting.py:29-30 does with a datagram."""
    return np.array(json.loads(data), dtype=object)


def test_control_messages():
    m = json.loads(V.encode_init(6, 7, (3, 2)))
    assert m == [6, 7, 3, [3, 2]]                       # manytor.py:94; plotting.py:36-43 reads msg[0], msg[1], msg[3]
    assert json.loads(V.encode_init(1, 10)) == [1, 10, 3]   # manytor.py:271
    for enc, code in ((V.encode_stop, 2), (V.encode_clear, 4)):
        m = json.loads(enc())
        assert math.isnan(m[0]) and math.isnan(m[1]) and m[2] == code


def test_frame_layout_matches_the_consumer():
    k = 7
    jc = np.arange(12, dtype=np.float64).reshape(4, 3) + 0.123456789
    pts = np.arange(3 * k, dtype=np.float64).reshape(k, 3) - 20.5
    data = V.encode_frame(5, jc, pts, jc[3], first=True)
    msg = np.array(json.loads(data), dtype=np.float64)
    assert msg[2] == 1 and int(msg[0]) == 5 and np.isnan(msg[1])
    rows = msg.reshape(-1, 3)                            # plotting.py:80
    assert rows.shape == (1 + 4 + k + 1, 3)
    np.testing.assert_allclose(rows[1:5], jc, atol=1e-4)           # plotting.py:82
    np.testing.assert_allclose(rows[5:5 + k], pts, atol=1e-4)      # plotting.py:83
    np.testing.assert_allclose(rows[-1], jc[3], atol=1e-4)         # plotting.py:86
    assert json.loads(V.encode_frame(0, jc, pts, jc[3]))[2] == 0


def test_frames_fit_the_1024_byte_datagram_for_both_harness_sizes():
    rng = np.random.RandomState(0)
    for k in (7, 10):
        jc = rng.uniform(-55.6, 55.6, size=(4, 3))
        pts = rng.uniform(-51.3, 51.3, size=(k, 3))
        assert len(V.encode_frame(123456, jc, pts, jc[3])) <= V.MAX_DATAGRAM


def test_link_sends_datagrams_in_lockstep_order():
    recv = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    recv.bind(("127.0.0.1", 0))
    recv.settimeout(5.0)
    link = V.ViewerLink("127.0.0.1", recv.getsockname()[1])
    n, s, k = 2, 3, 2
    traces = np.arange(n * s * 4 * 3, dtype=np.float64).reshape(n, s, 4, 3)
    pts = np.zeros((n, k, 3))
    link.init(n, k, (1, 2))
    link.frames([0, 1], traces, pts, first=True)
    link.clear()
    link.stop()
    got = [recv.recvfrom(2048)[0] for _ in range(1 + n * s + 2)]
    assert json.loads(got[0])[2] == 3
    order = [(int(json.loads(g)[0]), json.loads(g)[2]) for g in got[1:1 + n * s]]
    assert order == [(0, 1), (1, 1), (0, 0), (1, 0), (0, 0), (1, 0)]          # sub-step major, flag on the first only
    ee_last = np.array(json.loads(got[n * s])).reshape(-1, 3)[-1].astype(float)
    np.testing.assert_allclose(ee_last, traces[1, s - 1, 3], atol=1e-4)
    assert json.loads(got[-2])[2] == 4 and json.loads(got[-1])[2] == 2
    assert link.sent == len(got)
    link.close()
    recv.close()
