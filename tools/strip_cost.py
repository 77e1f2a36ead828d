import sys; sys.path.insert(0,'.')
import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import scenes
rt.init()
flags = int(__import__('os').environ.get('RT_SC_FLAGS', '0'))
for cfg in sys.argv[1:]:
    sph, tri, rq = scenes.config_world(cfg)
    with rt.Scene(0, rt.World(sph, tri)) as sc:
        row=[]
        for k in range(rq.divisions):
            r=rq.copy(); r.division_no=k; r.flags=flags
            sc.render_tile(r)
            _,_,st = sc.render_tile(r)
            row.append((st.ray_segments/ (rq.width*(rq.height//rq.divisions)*rq.spp), st.kernel_ms))
        print(cfg, "segments per sample by strip (top to bottom):", " ".join(f"{a:.2f}" for a,_ in row))
        print(cfg, "kernel ms by strip:", " ".join(f"{b:.3f}" for _,b in row))
