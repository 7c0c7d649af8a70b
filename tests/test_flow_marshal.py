"""Flow.marshal (host side of Flow.predict / MultiTracker.predict_async): the closest-first order equals the reference's
`tracks.sort(reverse=True)` with Track.__lt__ (fastmot/track.py:160-162), ties included, and the packed arrays are the
tracks' boxes / keypoints in that order.  No GPU needed: the method only reads `frame_rect`."""
import numpy as np

from fastmot_amd.flow import Flow


class RefTrack:
    def __init__(self, trk_id, tlbr, age, n_kp, rng):
        self.trk_id, self.age = trk_id, age
        self._tlbr = np.asarray(tlbr, float)
        self.keypoints = rng.uniform(0, 100, (n_kp, 2)).astype(np.float32)

    @property
    def tlbr(self):
        return self._tlbr

    def __lt__(self, other):                    # the reference's ordering
        return (self.tlbr[-1], -self.age) < (other.tlbr[-1], -other.age)


def test_marshal_order_and_packing():
    rng = np.random.default_rng(3)
    flow = object.__new__(Flow)
    flow.frame_rect = np.array([0., 0., 639., 359.])
    for trial in range(50):
        n = int(rng.integers(0, 40))
        tracks = []
        for i in range(n):
            x0, y0 = rng.uniform(-20, 600), rng.uniform(-20, 300)
            bottom = float(rng.choice([100., 150., 200., y0 + rng.uniform(5, 80)]))      # many equal bottoms: ties
            bottom = max(bottom, max(y0, 0.) + 1)                                         # (boxes reach into the frame)
            tracks.append(RefTrack(i, [x0, y0, max(x0, 0.) + rng.uniform(5, 60), bottom], int(rng.integers(0, 3)),
                                   int(rng.integers(0, 30)), rng))
        want = sorted(tracks, reverse=True)                                               # reference semantics
        got, inside, tlbrs, kps, kp_off = flow.marshal(list(tracks))
        assert [t.trk_id for t in got] == [t.trk_id for t in want]
        assert tlbrs.shape == (n, 4) and inside.shape == (n, 4) and kp_off.shape == (n + 1,)
        if n:
            np.testing.assert_array_equal(tlbrs, np.array([t.tlbr for t in want]))
            np.testing.assert_array_equal(inside[:, :2], np.maximum(tlbrs[:, :2], 0.))
            np.testing.assert_array_equal(inside[:, 2:], np.minimum(tlbrs[:, 2:], [639., 359.]))
            np.testing.assert_array_equal(np.diff(kp_off), [len(t.keypoints) for t in want])
            if kp_off[-1]:
                np.testing.assert_array_equal(kps, np.concatenate([t.keypoints for t in want]))
        assert kps.dtype == np.float32 and kp_off.dtype == np.int32
