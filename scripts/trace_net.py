"""Runs one network a few times (graph replay) so that `rocprofv3 --kernel-trace` records per-dispatch
durations; scripts/rocpd_dispatches.py then lists the last replay dispatch by dispatch."""
import os as _os
_os.environ.setdefault('FASTMOT_RANDOM_WEIGHTS', '1')   # no weight files offline
import sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from fastmot_amd.runtime import get_context
from fastmot_amd.engine import HipNet
from fastmot_amd.models import YOLO, ReID

which = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ctx = get_context()
if which == 0:       # python scripts/trace_net.py 0 [YOLOv4_608 | YOLOv4CSP_640 | YOLOv4P6_1280]
    g, _ = YOLO.get_model(sys.argv[2] if len(sys.argv) > 2 else 'YOLOv4_608').build_graph(); batch = 1
else:
    ctx.feat_configure(512)
    g, _ = ReID.get_model('OSNet025').build_graph(); batch = int(sys.argv[2]) if len(sys.argv) > 2 else 50
net = HipNet(ctx, which, g, batch, reuse_buffers=True)
for _ in range(6):
    net.run(batch)
    ctx.synchronize()
