"""The frame loop of the reference application (app.py:85-98) with ONE frame of read-ahead.

The reference's `app.py` is the command line -- it runs unmodified against this tree (`import fastmot`,
tests/test_dropin_app.py) and its loop calls `mot.step(frame)`.  A caller that already holds the following frame (a
file source, the capture queue of VideoIO) can hand it over as `next_frame`: the detector pass of frame t+1 then
overlaps the ReID / association stages of frame t, with identical results.  This function is
that loop; everything around it (arguments, configuration file, logging) is the reference's."""
from .utils.motchallenge import write_rows


def track_stream(stream, mot=None, txt=None, resize_to=None, write_frames=False):
    """stream: a started VideoIO; mot: a reset MOT (None: frames are only passed through); txt: an open text file for
    MOT Challenge result rows (app.py:91-97), needs `resize_to`; write_frames: stream.write(frame) after each step
    (the frame carries the overlays when the MOT draws).  Returns the number of frames."""
    n = 0
    frame = stream.read()
    while frame is not None:
        upcoming = stream.read()
        if mot is not None:
            mot.step(frame, next_frame=upcoming)
            if txt is not None:
                write_rows(txt, mot.frame_count, mot.visible_tracks(), resize_to, stream.resolution)
        if write_frames:
            stream.write(frame)
        frame = upcoming
        n += 1
    return n
