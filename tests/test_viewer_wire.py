"""CPU-only: the UDP/JSON wire format of the reference's viewer (manytor.py:94-101,194-202,246-249,271-279 as
consumed by plotting.py:27-87), produced by manytor_amd/viewer.py."""
import json
import math
import socket

import numpy as np
import pytest

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


def test_f9_reference_datagrams_pin_the_frame_format(golden):
    """Fixture F9 = the datagrams a reference Environment(7, index=3) sent over two steps (captured by giving it a
    socket stand-in, manytor.py:194-201) and on the following reset (:246-249).  encode_frame fed with the pieces of a
    reference frame must give back the same numbers in the same order, and fit the viewer's 1024-byte read."""
    g = golden("f9_viewer_frames")
    k, env_id = int(g["obj_number"]), int(g["env_id"])
    frames = g["frames"]
    assert frames.shape == (2 * 25, 3 * (1 + 4 + k + 1))
    for f in frames:
        rows = f.reshape(-1, 3)
        assert int(rows[0, 0]) == env_id and math.isnan(rows[0, 1]) and rows[0, 2] == 0     # the reference never sends 1
        mine = np.array(json.loads(V.encode_frame(env_id, rows[1:5], rows[5:5 + k], rows[-1])), dtype=np.float64)
        assert mine.shape == f.shape
        assert int(mine[0]) == env_id and math.isnan(mine[1]) and mine[2] == 0
        np.testing.assert_allclose(mine[3:], f[3:], atol=1e-4)
        np.testing.assert_array_equal(rows[-1], rows[4])          # last row = the end effector = joints_coordinates[3]
    assert g["datagram_bytes"].max() <= V.MAX_DATAGRAM
    clear = np.array(json.loads(V.encode_clear()), dtype=np.float64)
    np.testing.assert_array_equal(np.isnan(clear), np.isnan(g["clear"]))
    assert clear[2] == g["clear"][2] == 4


@pytest.mark.gpu
def test_f9_environment_streams_the_reference_frames(golden):
    """The same two steps through the drop-in Environment with a viewer link: 50 datagrams, each within 1e-4 of the
    reference's (positions from the GPU's route trace), then the clear message on reset."""
    import manytor_amd as m
    g = golden("f9_viewer_frames")
    k, env_id = int(g["obj_number"]), int(g["env_id"])
    recv = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    recv.setsockopt(socket.SOL_SOCKET, socket.SO_RCVBUF, 1 << 22)
    recv.bind(("127.0.0.1", 0))
    recv.settimeout(5.0)
    np.random.seed(int(g["seed"]))
    env = m.Environment(k, index=env_id, viewer=("127.0.0.1", recv.getsockname()[1]))
    env.reset()
    env.render()                                       # init datagram [1, K, 3] (manytor.py:271)
    for a in g["action"]:
        assert list(env.action_sample()) == list(a)     # same R2 stream as the reference
        env.step(list(a))
    env.reset()
    got = [recv.recvfrom(2048)[0] for _ in range(1 + 50 + 1)]
    assert json.loads(got[0]) == [1, k, 3]
    for data, ref in zip(got[1:51], g["frames"]):
        mine = np.array(json.loads(data), dtype=np.float64)
        assert int(mine[0]) == env_id and math.isnan(mine[1]) and mine[2] == ref[2] == 0
        np.testing.assert_allclose(mine[3:], ref[3:], atol=2e-4)
    last = json.loads(got[51])
    assert math.isnan(last[0]) and last[2] == 4
    recv.close()
