import sys; sys.path.insert(0,'.')
import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import scenes, _abi
rt.init()
for cfg in sys.argv[1:]:
    sph, tri, rq = scenes.config_world(cfg)
    reqs=[]
    for k in range(rq.divisions):
        r=rq.copy(); r.division_no=k; r.flags=_abi.RT_FLAG_COUNT_STEPS; reqs.append(r)
    with rt.Scene(0, rt.World(sph, tri)) as sc:
        outs,_,st = sc.render_tiles(reqs)
        print(cfg, "engine", st.engine, "segments", st.ray_segments, "interior steps/seg %.2f"%(st.node_steps/st.ray_segments), "leaves/seg %.2f"%(st.broad_candidates/st.ray_segments), flush=True)
