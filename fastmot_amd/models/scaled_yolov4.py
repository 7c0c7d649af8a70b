"""Built-in Darknet cfg text of the Scaled-YOLOv4 detectors the reference lists as "supported but not provided"
(fastmot/models/yolo.py:166-253): YOLOv4-CSP (640x640, strides 8/16/32) and YOLOv4-P6 (1280x1280, strides
8/16/32/64, four anchors per head).

The reference ships neither cfg nor weights (scripts/download_models.sh only fetches its own CrowdHuman model); the
public AlexeyAB/darknet cfg files are not reachable offline.  The sections below are GENERATED from the published
architecture (Scaled-YOLOv4, Wang et al., CVPR 2021, Fig. 4 / Table 1: CSP-ised Darknet backbone, CSP-SPP, CSP-PAN
neck, mish everywhere, new_coords heads with logistic outputs, scale_x_y = 2) in Darknet's section order, with every
relative route / shortcut index computed from recorded section indices.  A Darknet cfg placed next to the weight
file always takes precedence (models/yolo.py); these tables exist so that BASELINE configs [2] and [4] can be built
and measured without it.  Checks (tests/test_scaled_yolov4.py): parameter and FLOP totals against the published
figures (52.9 M / ~120 GFLOP @640 for CSP; 127 M / ~718 GFLOP @1280 for P6)."""

_COCO9 = '12,16, 19,36, 40,28, 36,75, 76,55, 72,146, 142,110, 192,243, 459,401'
_P6_16 = ('13,17, 31,25, 24,51, 61,45, 61,45, 48,102, 119,96, 97,189, 97,189, 217,184, 171,384, 324,451, '
          '324,451, 545,357, 616,618, 1024,1024')


class _Cfg:
    def __init__(self, width, height):
        self.out = [f'[net]\nbatch=1\nwidth={width}\nheight={height}\nchannels=3\n\n']
        self.idx = -1

    def add(self, text):
        self.out.append(text)
        self.idx += 1
        return self.idx

    def conv(self, f, k, s=1, act='mish', bn=1):
        return self.add(f'[convolutional]\n' + ('batch_normalize=1\n' if bn else '') +
                        f'filters={f}\nsize={k}\nstride={s}\npad=1\nactivation={act}\n\n')

    def route(self, *targets):          # absolute section indices -> relative
        cur = self.idx + 1
        return self.add('[route]\nlayers = ' + ','.join(str(t - cur) for t in targets) + '\n\n')

    def shortcut(self, src):
        cur = self.idx + 1
        return self.add(f'[shortcut]\nfrom={src - cur}\nactivation=linear\n\n')

    def maxpool(self, k):
        return self.add(f'[maxpool]\nstride=1\nsize={k}\n\n')

    def upsample(self):
        return self.add('[upsample]\nstride=2\n\n')

    def yolo(self, mask, anchors, num, classes):
        return self.add(f'[yolo]\nmask = {mask}\nanchors = {anchors}\nclasses={classes}\nnum={num}\n'
                        'scale_x_y = 2.0\nnew_coords=1\n\n')

    def text(self):
        return ''.join(self.out)

    # ---- building blocks
    def csp_stage(self, cout, n_res, half=True):
        """Downsample 3x3/2 to `cout`, then a CSP block of n_res residual units (hidden width cout/2, or the
        full width in the first stage of the P models)."""
        d = self.conv(cout, 3, 2)
        h = cout // 2 if half else cout
        a = self.conv(h, 1)                     # route branch
        self.route(d)
        x = self.conv(h, 1)
        for _ in range(n_res):
            self.conv(h, 1)
            self.conv(h, 3)
            x = self.shortcut(x)
        post = self.conv(h, 1)
        self.route(post, a)
        return self.conv(cout, 1)

    def csp_spp(self, c):
        """CSP-SPP on the deepest level: (1x1, 3x3, 1x1, SPP 5/9/13, 1x1, 3x3) beside a 1x1 route branch."""
        src = self.idx
        a = self.conv(c, 1)
        self.route(src)
        self.conv(c, 1)
        self.conv(c, 3)
        x = self.conv(c, 1)
        p5 = self.maxpool(5)
        self.route(x)
        p9 = self.maxpool(9)
        self.route(x)
        p13 = self.maxpool(13)
        self.route(p13, p9, p5, x)
        self.conv(c, 1)
        y = self.conv(c, 3)
        self.route(y, a)
        return self.conv(c, 1)

    def rcsp(self, c, n_pairs=2):
        """CSP block of the neck (no shortcuts): 1x1 reduce, then n_pairs x (1x1, 3x3) beside a 1x1 route branch."""
        x = self.conv(c, 1)
        a = self.conv(c, 1)
        self.route(x)
        self.conv(c, 1)
        y = None
        for i in range(n_pairs):
            y = self.conv(c, 3)
            if i + 1 < n_pairs:
                self.conv(c, 1)
        self.route(y, a)
        return self.conv(c, 1)


def _pan(g, feats, spp_c, widths, classes, anchors, n_anchor, n_pairs):
    """feats: backbone outputs shallow -> deep; widths: neck width per level (same order)."""
    n = len(feats)
    cur = g.csp_spp(spp_c)
    tops = {n - 1: cur}
    for lvl in range(n - 2, -1, -1):                  # top-down
        c = widths[lvl]
        g.conv(c, 1)                                  # (reads the block just above: the previous section)
        up = g.upsample()
        g.route(feats[lvl])
        lat = g.conv(c, 1)
        g.route(lat, up)
        cur = tops[lvl] = g.rcsp(c, n_pairs)
    n_out = (classes + 5) * n_anchor
    prev = tops[0]
    for lvl in range(n):                              # heads, bottom-up path between them
        c = widths[lvl]
        if lvl > 0:
            g.route(prev)                             # back from the [yolo] section to the neck
            dn = g.conv(c, 3, 2)
            g.route(dn, tops[lvl])
            prev = g.rcsp(c, n_pairs)
        g.conv(2 * c, 3)
        g.conv(n_out, 1, act='logistic', bn=0)
        g.yolo(','.join(str(lvl * n_anchor + i) for i in range(n_anchor)), anchors, n_anchor * n, classes)
    return g.text()


def yolov4_csp_cfg(width=640, height=640, classes=80):
    g = _Cfg(width, height)
    g.conv(32, 3)
    # first stage: a plain Darknet residual unit (no CSP split), as in yolov4-csp.cfg
    d = g.conv(64, 3, 2)
    g.conv(32, 1)
    g.conv(64, 3)
    g.shortcut(d)
    feats = []
    for cout, n_res in ((128, 2), (256, 8), (512, 8), (1024, 4)):
        feats.append(g.csp_stage(cout, n_res))
    return _pan(g, feats[1:], 512, [128, 256, 512], classes, _COCO9, 3, 2)


def yolov4_p6_cfg(width=1280, height=1280, classes=80):
    g = _Cfg(width, height)
    g.conv(32, 3)
    feats = []
    for cout, n_res in ((64, 1), (128, 3), (256, 15), (512, 15), (1024, 7), (1024, 7)):
        feats.append(g.csp_stage(cout, n_res))
    return _pan(g, feats[2:], 512, [128, 256, 512, 512], classes, _P6_16, 4, 3)
