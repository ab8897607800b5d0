import sys, time, numpy as np
sys.path.insert(0,'.')
from pyfocusr_amd import _hip, Graph
from pyfocusr_amd.meshgen import blob_mesh
ctx=_hip.default_context()
m=blob_mesh(250000,0)
m._pf_device_mesh=_hip.DeviceMesh(m.points,m.faces,ctx=ctx)
for it in range(4):
    t0=time.perf_counter(); g=Graph(m,n_spectral_features=5,n_rand_samples=5000,ctx=ctx,verbose=False); t1=time.perf_counter()
    d=g.device; ctx.sync(); t2=time.perf_counter()
    print('ctor %.2f ms  build %.2f ms  kernels(build_ms) %.2f'%(1e3*(t1-t0),1e3*(t2-t1),ctx.timing()['build_ms']))
    t0=time.perf_counter(); g.get_graph_spectrum(); t1=time.perf_counter()
    st=g.eigs_stats
    print('  spectrum %.2f ms matvecs %d steps %d deg %d'%(1e3*(t1-t0), st.matvecs, st.outer_steps, st.degree))
    d.close()
