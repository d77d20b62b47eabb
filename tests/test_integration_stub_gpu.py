"""INTEGRATION.md §3 shows the ctypes binding a maintainer of the reference would write against the C ABI (no Python of this
repository).  This test runs THAT code block, verbatim, and checks its render against the oracle — so the document cannot drift
from the ABI (struct layouts, argument order, call sequence)."""
import os
from collections import namedtuple

import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_ctypes_stub_of_integration_md_renders_what_the_oracle_renders(hip_device):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    start = text.index("```python\nimport ctypes as C, torch\nlib = C.CDLL")
    code = text[start + len("```python\n"):text.index("```", start + 10)]
    so = os.path.join(ROOT, "latentsplat_amd", "csrc", "liblsr_hip.so")
    assert 'C.CDLL("liblsr_hip.so")' in code
    ns: dict = {}
    exec(compile(code.replace('C.CDLL("liblsr_hip.so")', f"C.CDLL({so!r})"), "INTEGRATION.md", "exec"), ns)
    sc = util.make_scene(3000, image_size=64, views=1, color_sh_degree=2, feature_channels=4)
    bi = util.boundary_inputs(sc, 64, 64, bg=(0.1, 0.2, 0.3))
    dev, c = hip_device, bi["cams"]
    S = namedtuple("S", "image_height image_width tanfovx tanfovy bg viewmatrix projmatrix sh_degree campos")
    st = S(64, 64, float(c.tan_fov_x[0]), float(c.tan_fov_y[0]), bi["bg"][0].to(dev), c.view_matrix[0].to(dev).contiguous(),
           c.full_projection[0].to(dev).contiguous(), bi["sh_degree"], c.campos[0].to(dev))
    t = lambda x: x.to(dev).contiguous()
    with torch.no_grad():
        color, feat, mask, depth, radii, _keep = ns["rasterize_forward"](st, t(bi["means"][0]), t(bi["shs"]), t(bi["features"][0]), t(bi["opac"]), t(bi["cov6"][0]))
    torch.cuda.synchronize(dev)
    o = util.oracle_forward(bi, 0)
    assert np.array_equal(radii.cpu().numpy(), o["radii"])
    util.assert_close_except_fragile(color.cpu().numpy(), o["color"], o, 1e-4, "INTEGRATION.md stub colour")
    util.assert_close_except_fragile(feat.cpu().numpy(), o["feature"], o, 1e-4, "INTEGRATION.md stub feature")
    util.assert_close_except_fragile(mask[0].cpu().numpy(), o["mask"], o, 1e-4, "INTEGRATION.md stub mask")
