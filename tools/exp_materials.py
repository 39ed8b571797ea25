"""Experiment: cost of the material branches of the shading pass.  Renders the 1M-triangle scene with its own
materials, with every material made Lambertian, and with every material made Metal."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import rrt_pkg  # noqa: E402

rrt = rrt_pkg.load()
from rust_ray_tracer_amd import scenes  # noqa: E402

s = scenes.synthetic_mesh(rrt, os.path.join(ROOT, "assets"), subdiv=512)
w, h, spp = 3840, 2160, 64
accum = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
orig = list(s.materials)
for name in ("own", "lambertian", "metal"):
    mats = []
    for m in orig:
        kind = type(m).__name__
        if name == "own" or kind == "DiffuseLight":
            mats.append(m)
        elif name == "lambertian":
            mats.append(rrt.Lambertian(0.7, 0.7, 0.7))
        else:
            mats.append(rrt.Metal((0.7, 0.7, 0.7), 0.3))
    s.materials = mats
    r = rrt.Renderer(device=0, seed=1)
    r.upload(s)
    for _ in range(2):
        r.render_device(w, h, spp, 0, accum.data_ptr(), torch.cuda.current_stream().cuda_stream)
        st = r.render_stats()
    print(name, "kernel_ms %.2f" % st.kernel_ms, "rays %d" % st.rays, "ns per ray %.4f" % (st.kernel_ms * 1e6 / st.rays), flush=True)
    r.close()
