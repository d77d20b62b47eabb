import numpy as np, torch, sys
sys.path.insert(0,'.')
from tests import util
from latentsplat_amd.rasterizer import rasterize_views
dev=torch.device('cuda:0')
sc = util.make_scene(300_000, image_size=256, views=1, color_sh_degree=None, feature_channels=4)
bi = util.boundary_inputs(sc, 256, 256)
views = util.view_table(bi, dev)
req = lambda k: bi[k].to(dev).clone().requires_grad_(True)
m, c, o, f = req("means"), req("cov6"), req("opac"), req("features")
out = rasterize_views(views, 256, 256, 0, m, c, o, features=f)
g = torch.randn(out[1].shape, generator=torch.Generator().manual_seed(11))
grads = torch.autograd.grad((out[1] * g.to(dev)).sum(), (m, c, o, f))
ofw = util.oracle_forward(bi, 0)
b = util.oracle_backward(bi, 0, ofw, None, g[0].numpy())
fe = np.abs(out[1][0].detach().cpu().numpy()-ofw["feature"]); print("fwd feat err max", fe.max(), "n>1e-4", (fe>1e-4).sum(), "n>1e-5", (fe>1e-5).sum())
for name, got, want in (("means3D", grads[0][0], b["means3D"]), ("cov3D", grads[1][0], b["cov3D"]), ("opacities", grads[2], b["opacities"]), ("features", grads[3][0], b["features"])):
    got=got.cpu().numpy(); e=np.abs(got-want); s=max(1,np.abs(want).max())
    rowmax = e.reshape(e.shape[0],-1).max(1)
    print(name,"scale",s,"max err",e.max(),"rows>1e-4*s",(rowmax>1e-4*s).sum(),"rows>1e-5*s",(rowmax>1e-5*s).sum(), "median rowerr", np.median(rowmax), "p99.9", np.percentile(rowmax,99.9))
