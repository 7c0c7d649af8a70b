"""End to end through the command line (the reference's app.py flow): image sequence on disk -> VideoIO ->
MOT (public detections, OSNet with seeded weights, KLT, Kalman, association) -> MOTChallenge result file ->
CLEAR-MOT / IDF1 against the synthetic ground truth."""
import json
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import pytest

from fastmot_amd.utils import motchallenge as mc
from synthetic import SyntheticVideo
from fastmot_amd.videoio import VideoIO, resize_bgr


def test_videoio_sources_and_queue(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(0)
    frames = rng.integers(0, 256, (7, 36, 64, 3), dtype=np.uint8)
    (tmp_path / 'img1').mkdir()
    for i, f in enumerate(frames):
        Image.fromarray(f[:, :, ::-1]).save(tmp_path / 'img1' / f'{i + 1:06d}.png')
    np.save(tmp_path / 'stack.npy', frames)
    for uri in (str(tmp_path / 'img1' / '%06d.png'), str(tmp_path / 'stack.npy')):
        stream = VideoIO((64, 36), uri, frame_rate=25, buffer_size=3)
        assert stream.cap_dt == 1 / 25 and stream.resolution == (64, 36) and not stream.do_resize
        stream.start_capture()
        got = []
        while True:
            f = stream.read()
            if f is None:
                break
            got.append(f)
        stream.release()
        np.testing.assert_array_equal(np.stack(got), frames)
    # resizing source: 2x decimation = area average, arbitrary ratio = fixed-point bilinear
    stream = VideoIO((32, 18), str(tmp_path / 'stack.npy'))
    f = stream.read()
    ref = (frames[0].astype(int).reshape(18, 2, 32, 2, 3).sum((1, 3)) + 2) >> 2
    np.testing.assert_array_equal(f, ref.astype(np.uint8))
    stream.release()
    up = resize_bgr(frames[0], (100, 50))
    assert up.shape == (50, 100, 3) and up.dtype == np.uint8
    np.testing.assert_array_equal(up[0, 0], frames[0][0, 0])
    with pytest.raises(NotImplementedError):
        VideoIO((64, 36), 'rtsp://camera/stream')
    with pytest.raises(NotImplementedError):
        VideoIO((64, 36), str(tmp_path / 'movie.mp4'))


@pytest.mark.gpu
def test_app_end_to_end(tmp_path):
    from PIL import Image
    size, n_frames, n_ids = (960, 540), 24, 8
    video = SyntheticVideo(size, n_ids=n_ids, n_frames=n_frames, seed=5)
    seq = tmp_path / 'SYN-01'
    (seq / 'img1').mkdir(parents=True)
    (seq / 'det').mkdir()
    (seq / 'seqinfo.ini').write_text(f'[Sequence]\nname=SYN-01\nimDir=img1\nframeRate=30\nseqLength={n_frames}\n'
                                     f'imWidth={size[0]}\nimHeight={size[1]}\nimExt=.png\n')
    det_rows, gt = [], {}
    for f in range(n_frames):
        Image.fromarray(video.frames[f][:, :, ::-1]).save(seq / 'img1' / f'{f + 1:06d}.png')
        d = video.detections(f)
        for box in d.tlbr:
            det_rows.append(f'{f + 1},-1,{box[0]},{box[1]},{box[2] - box[0] + 1},{box[3] - box[1] + 1},1,-1,-1,-1')
        gt[f + 1] = [(i + 1, np.array([b[0], b[1], b[2] - b[0] + 1, b[3] - b[1] + 1])) for i, b in enumerate(video.gt[f])]
    (seq / 'det' / 'det.txt').write_text('\n'.join(det_rows) + '\n')

    import fastmot_amd
    from fastmot_amd.readahead import track_stream
    from fastmot_amd.utils import ConfigDecoder
    cfg = json.load(open(Path(fastmot_amd.__file__).parent / 'cfg' / 'mot.json'))
    cfg['resize_to'] = list(size)
    cfg['stream_cfg']['resolution'] = list(size)
    cfg['mot_cfg']['detector_type'] = 'PUBLIC'
    cfg['mot_cfg']['detector_frame_skip'] = 1
    cfg['mot_cfg']['public_detector_cfg']['sequence_path'] = str(seq)
    (tmp_path / 'mot.json').write_text(json.dumps(cfg))

    def run(txt_path, output_uri=None):
        """What the reference's app.py does with `-i seq -c mot.json -m -t txt [-o uri]` (app.py:55-104), the frame
        loop with one frame of read-ahead (fastmot_amd/readahead.py)."""
        with open(tmp_path / 'mot.json') as f:
            config = json.load(f, cls=ConfigDecoder, object_hook=lambda d: SimpleNamespace(**d))
        stream = fastmot_amd.VideoIO(config.resize_to, str(seq / 'img1' / '%06d.png'), output_uri,
                                     **vars(config.stream_cfg))
        mot = fastmot_amd.MOT(config.resize_to, **vars(config.mot_cfg), draw=output_uri is not None)
        mot.reset(stream.cap_dt)
        txt_path.parent.mkdir(parents=True, exist_ok=True)
        stream.start_capture()
        try:
            with open(txt_path, 'w') as txt:
                n = track_stream(stream, mot, txt, config.resize_to, write_frames=output_uri is not None)
        finally:
            stream.release()
        return n

    out = tmp_path / 'out' / 'SYN-01.txt'
    assert run(out) == n_frames
    res = mc.read_txt(out)
    score = mc.evaluate({f: v for f, v in gt.items() if f in res or f > 1}, res)
    # public detections are the ground truth + 1 px jitter: everything is tracked, identities are stable
    assert score['mota'] > 0.9 and score['idf1'] > 0.9 and score['idsw'] <= 2, score

    # -o: the written frames carry the overlays (reference app.py:70-71 draws whenever an output is requested)
    out2 = tmp_path / 'out2' / 'SYN-01.txt'
    assert run(out2, str(tmp_path / 'annotated' / '%06d.png')) == n_frames
    assert out2.read_text() == out.read_text()          # drawing does not change the tracks
    last = np.asarray(Image.open(tmp_path / 'annotated' / f'{n_frames - 1:06d}.png'))[:, :, ::-1]
    assert last.shape == video.frames[-1].shape and (last != video.frames[-1]).any(axis=2).sum() > 400
