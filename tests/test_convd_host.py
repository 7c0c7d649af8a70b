"""Host side of the DMA-fed conv kernel (convd.hip): weight tile images and the table builder's choice of kernel (no GPU)."""
import numpy as np

from fastmot_amd.models import YOLO
from fastmot_amd.models import graph as G
from fastmot_amd.models.graph import Graph, RandomWeights


def test_convd_weight_image_cpu_layout():
    """_pack_tile64: slot s of row r of a (32-cout block, K step) image holds K chunk s ^ ((r / 2) % 8)."""
    rng = np.random.default_rng(0)
    w = rng.normal(0, 1, (64, 192)).astype(np.float16)
    img = Graph._pack_tile64(w)
    assert img.shape == (2, 3, 32, 8, 8)
    for blk, step, r, s in ((0, 0, 0, 0), (1, 2, 5, 3), (0, 1, 31, 7), (1, 0, 16, 1)):
        c = s ^ ((r >> 1) & 7)
        np.testing.assert_array_equal(img[blk, step, r, s], w[blk * 32 + r, step * 64 + c * 8:step * 64 + c * 8 + 8])


def test_table_builder_picks_the_dma_kernel_for_cin_multiple_of_64():
    g = Graph(RandomWeights(seed=0), (20, 20), 64)
    g.conv('a', g.input, 64, 1)
    g.conv('b', g.input, 96, 3, 2)
    y = g.conv('c', g.input, 32, 1)                     # 32 channels out ...
    g.conv('d', y, 64, 3)                               # ... so this one has cin = 32: LDS-tiled kernel
    g.conv('e', g.input, 64, 5)                         # 5 x 5: not a shape of the DMA kernel (here: the streamed one)
    assert [d['op'] for d in g.layers] == [G.OP_CONVD, G.OP_CONVS, G.OP_CONVD, G.OP_CONV, G.OP_CONVS]   # (b: small map, long K)
    g = Graph(RandomWeights(seed=0), (80, 80), 64)
    g.conv('a', g.input, 64, 3)                         # 6400 pixels: no streamed kernel -> DMA kernel
    g.conv('b', g.input, 64, 3, 2)                      # stride 2 into 1600 pixels: DMA kernel
    y = g.conv('c', g.input, 64, 3, 2)
    g.conv('d', y, 64, 3)                               # 1600 pixels, K = 576: DMA kernel (beyond convs_max_pixels)
    g.conv('e', g.new(30, 30, 64), 64, 3)               # 900 pixels, K = 576: the streamed kernel keeps it
    assert [d['op'] for d in g.layers] == [G.OP_CONVD] * 4 + [G.OP_CONVS]
    g = Graph(RandomWeights(seed=0), (20, 20), 64)
    g.convd_level = 2                                   # everything the DMA kernel can do
    g.conv('a', g.input, 64, 1)
    g.conv('b', g.input, 64, 3)
    assert [d['op'] for d in g.layers] == [G.OP_CONVD, G.OP_CONVD]
