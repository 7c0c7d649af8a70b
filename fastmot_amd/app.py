"""Command line of the reference (app.py:15-116) on top of fastmot_amd:

    python -m fastmot_amd.app -i 'MOT20-01/img1/%06d.jpg' -c cfg/mot.json -m -t out/MOT20-01.txt

Same arguments, same configuration file format (the reference's cfg/mot.json works unchanged), same result
rows.  `--output-uri` writes the frames with the tracker overlays drawn (image sequence or .npy), as the
reference does (app.py:70-71: draw = show or output); `--show` needs a display toolkit (no GUI in this image)
and is rejected."""
from pathlib import Path
from types import SimpleNamespace
import argparse
import json
import logging

import fastmot_amd
from fastmot_amd.utils import ConfigDecoder, Profiler
from fastmot_amd.utils.motchallenge import write_rows


def parse_args(argv=None):
    parser = argparse.ArgumentParser(formatter_class=argparse.RawTextHelpFormatter)
    optional = parser._action_groups.pop()
    required = parser.add_argument_group('required arguments')
    group = parser.add_mutually_exclusive_group()
    required.add_argument('-i', '--input-uri', metavar="URI", required=True, help=
                          'URI to input stream\n'
                          "1) image sequence (e.g. %%06d.jpg)\n"
                          '2) frame stack (e.g. frames.npy, [N, H, W, 3] uint8 BGR)\n'
                          'video files, cameras and network streams need a decoder (not available here)')
    optional.add_argument('-c', '--config', metavar="FILE",
                          default=Path(__file__).parent / 'cfg' / 'mot.json',
                          help='path to JSON configuration file')
    optional.add_argument('-l', '--labels', metavar="FILE",
                          help='path to label names (e.g. coco.names)')
    optional.add_argument('-o', '--output-uri', metavar="URI",
                          help="URI to output frames (e.g. out/%%06d.png or out.npy)")
    optional.add_argument('-t', '--txt', metavar="FILE",
                          help='path to output MOT Challenge format results (e.g. MOT20-01.txt)')
    optional.add_argument('-m', '--mot', action='store_true', help='run multiple object tracker')
    optional.add_argument('-s', '--show', action='store_true', help='show visualizations (needs a display: not supported here)')
    group.add_argument('-q', '--quiet', action='store_true', help='reduce output verbosity')
    group.add_argument('-v', '--verbose', action='store_true', help='increase output verbosity')
    parser._action_groups.append(optional)
    args = parser.parse_args(argv)
    if args.txt is not None and not args.mot:
        raise parser.error('argument -t/--txt: not allowed without argument -m/--mot')
    if args.show:
        raise parser.error('argument -s/--show: no display toolkit in this build; use -o to write annotated frames')
    return args


def main(argv=None):
    args = parse_args(argv)

    # set up logging
    logging.basicConfig(format='%(asctime)s [%(levelname)8s] %(message)s', datefmt='%Y-%m-%d %H:%M:%S')
    logger = logging.getLogger(fastmot_amd.__name__)
    logger.setLevel(logging.WARNING if args.quiet else logging.DEBUG if args.verbose else logging.INFO)

    # load config file
    with open(args.config) as cfg_file:
        config = json.load(cfg_file, cls=ConfigDecoder, object_hook=lambda d: SimpleNamespace(**d))

    # load labels if given
    if args.labels is not None:
        with open(args.labels) as label_file:
            fastmot_amd.models.set_label_map(label_file.read().splitlines())

    stream = fastmot_amd.VideoIO(config.resize_to, args.input_uri, args.output_uri, **vars(config.stream_cfg))

    mot = None
    txt = None
    if args.mot:
        mot = fastmot_amd.MOT(config.resize_to, **vars(config.mot_cfg), draw=args.output_uri is not None)
        mot.reset(stream.cap_dt)
    if args.txt is not None:
        Path(args.txt).parent.mkdir(parents=True, exist_ok=True)
        txt = open(args.txt, 'w')

    logger.info('Starting video capture...')
    stream.start_capture()
    try:
        with Profiler('app') as prof:
            # one frame of read-ahead: the detector network of frame t+1 then overlaps the ReID / association
            # stages of frame t (MOT.step's next_frame; results are identical to strictly sequential steps)
            frame = stream.read()
            while frame is not None:
                upcoming = stream.read()
                if args.mot:
                    mot.step(frame, next_frame=upcoming)
                    if txt is not None:
                        write_rows(txt, mot.frame_count, mot.visible_tracks(), config.resize_to, stream.resolution)
                if args.output_uri is not None:
                    stream.write(frame)
                frame = upcoming
    finally:
        # clean up resources
        if txt is not None:
            txt.close()
        stream.release()

    # timing statistics
    if args.mot:
        avg_fps = round(mot.frame_count / prof.duration)
        logger.info('Average FPS: %d', avg_fps)
        mot.print_timing_info()
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
