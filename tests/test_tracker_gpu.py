"""GPU parity (end to end, oracle=reference): fastmot_amd.MultiTracker vs the golden runs of the
reference MultiTracker on the scripted scenes of tests/scenes.py.  Track IDs, dict order,
rounded boxes, lifecycle flags and history order must be IDENTICAL on every frame; final Kalman
states agree to fp64 round-off."""
import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu


# default configuration on every scene + the all-device configuration (device LAP kernels, blit copies
# instead of pinned zero-copy I/O) on two of them
CONFIGS = [(name, None) for name in scenes.SCENES] + [
    ('s50_skip1_cosine', dict(host_lap_elems=0, zero_copy_tracks=0)),
    ('s300_4k_multiclass', dict(host_lap_elems=0)),
]


@pytest.mark.parametrize('name,options', CONFIGS)
def test_scene_matches_reference(golden_dir, ctx, name, options):
    from fastmot_amd import MultiTracker, Track
    for key, value in (options or {}).items():
        ctx.set_option(key, value)
    try:
        _run_scene(golden_dir, name)
    finally:
        ctx.set_option('host_lap_elems', 262144)
        ctx.set_option('zero_copy_tracks', 2048)


def _run_scene(golden_dir, name):
    from fastmot_amd import MultiTracker, Track
    g = np.load(golden_dir / f'tracker_{name}.npz')
    scene = scenes.Scene(name)
    Track._count = 0
    tracker = MultiTracker(scene.size, scene.metric, **scenes.tracker_kwargs(name))
    records, final = scenes.run_scene(tracker, scene)
    out = scenes.pack_records(records, final)
    exp, got = g['tracks'], out['tracks']
    for frame_id in range(scene.n_frames):
        e = exp[exp[:, 0] == frame_id]
        t = got[got[:, 0] == frame_id]
        assert e[:, 2].tolist() == t[:, 2].tolist(), f'frame {frame_id}: track ids / order differ'
        np.testing.assert_array_equal(e, t, err_msg=f'frame {frame_id}')
    np.testing.assert_array_equal(out['hist'], g['hist'])
    np.testing.assert_array_equal(out['final_ids'], g['final_ids'])
    np.testing.assert_allclose(out['final_mean'], g['final_mean'], rtol=1e-9, atol=1e-7)
    np.testing.assert_allclose(out['final_cov'], g['final_cov'], rtol=1e-7, atol=1e-6)
    # every slot handed out is accounted for
    tracker._clear_tracks()
    tracker.reset(1 / 30.)


def test_kalman_filter_mirror_api(ctx):
    """KalmanFilter's ndarray methods (create/warp/predict/update/motion_distance) keep working."""
    import np_oracle as o
    from fastmot_amd import KalmanFilter, MeasType
    kf = KalmanFilter()
    kf.reset_dt(1 / 30.)
    p = o.KFParams(1 / 30.)
    box = np.array([100., 200., 180., 420.])
    m, c = kf.create(box)
    em, ec = o.kf_create(p, box)
    np.testing.assert_allclose(m, em[0]); np.testing.assert_allclose(c, ec[0])
    m2, c2 = kf.predict(m, c)
    em2, ec2 = o.kf_predict(p, em, ec)
    np.testing.assert_allclose(m2, em2[0], rtol=1e-12); np.testing.assert_allclose(c2, ec2[0], rtol=1e-10)
    m3, c3 = kf.update(m2, c2, box + 3, MeasType.FLOW, 2.0)
    em3, ec3 = o.kf_update(p, em2, ec2, box + 3, 'flow', 2.0)
    np.testing.assert_allclose(m3, em3[0], rtol=1e-11); np.testing.assert_allclose(c3, ec3[0], rtol=1e-9)
    d = kf.motion_distance(m3, c3, np.stack([box, box + 10]))
    np.testing.assert_allclose(d, o.kf_maha(p, em3, ec3, np.stack([box, box + 10]))[0], rtol=1e-10)


def test_foreign_gallery_reid(ctx):
    """Opt-in cross-stream ReID: a detection whose embedding matches a FOREIGN gallery entry starts
    a local track tagged with the foreign identity; with no sync object nothing changes."""
    from fastmot_amd import MultiTracker, Track

    class FakeSync:                      # stands in for gallery.GallerySync (no process group here)
        def __init__(self, entries):
            self.entries, self.calls = entries, 0

        def exchange(self, local):
            self.calls += 1
            self.local = list(local)
            return self.entries

        def consume(self, rank, trk_id):          # a re-identified foreign identity is not offered again
            self.entries = [e for e in self.entries if (e['rank'], e['trk_id']) != (rank, trk_id)]

    rng = np.random.default_rng(0)
    feat = rng.normal(0, 1, 512).astype(np.float32)
    feat /= np.linalg.norm(feat)
    other = rng.normal(0, 1, 512).astype(np.float32)
    other /= np.linalg.norm(other)
    sync = FakeSync([dict(rank=3, trk_id=17, label=1, count=4, feat=feat)])
    Track._count = 0
    trk = MultiTracker((1920, 1080), 'cosine', gallery_sync=sync, **scenes.tracker_kwargs())
    trk.flow = scenes.Scene('s8_flowfail').make_flow()
    trk.reset(1 / 30.)
    dets = np.zeros(2, scenes.DET_DTYPE).view(np.recarray)
    dets.tlbr = [[100, 100, 160, 280], [900, 300, 960, 480]]
    dets.label = 1
    dets.conf = 0.9
    emb = np.stack([feat + 0.01 * other, other]).astype(np.float32)
    emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    trk.update(1, dets, emb)
    assert sync.calls == 1
    tagged = [t for t in trk.tracks.values() if getattr(t, 'global_id', None) == (3, 17)]
    assert len(tagged) == 1 and tagged[0].avg_feat.count == 5 and tagged[0].confirmed
    np.testing.assert_array_equal(tagged[0].tlbr, dets.tlbr[0])
    assert len(trk.tracks) == 2 and sync.entries == []
    trk._clear_tracks()
