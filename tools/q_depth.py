import sys
sys.path.insert(0,'wgpu-path-tracing_amd'); sys.path.insert(0,'tests')
from ptmi import native, scenes, layout
ctx=native.Context(0)
for name in ('cornell','cornell_spheres','grid_1m'):
    sc=scenes.make(name)
    for keep in (1,0):
        ctx.set_options(keep_reference_tree=keep)
        import time; t=time.time(); ctx.upload_scene(sc); dt=time.time()-t
        ctx.resize(64,64); ctx.dispatch(layout.make_camera(64,64),1); st=ctx.stats()
        print(name,'keep',keep,'depth',st.bvh_depth,'trav',st.traversal_used,'ref depth',sc.bvh_depth,'upload s',round(dt,2))
