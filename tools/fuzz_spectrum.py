#!/usr/bin/env python3
"""Randomised robustness sweep of the eigensolver: blob meshes of 120-150k vertices, k = 1-12, with holes, removed caps,
extra components + stray points, squashed and shifted coordinates; checks convergence, residuals against the
downloaded Laplacian, ordering and finiteness.   python tools/fuzz_spectrum.py SEED N_CASES [holes]
("holes": only large open meshes with a hole of random radius and k = 1-3, the combination with the smallest cut and the
strongest non-normality)"""
import sys, time, numpy as np
sys.path.insert(0,'.')
from pyfocusr_amd import _hip, Graph, PolyMesh
from pyfocusr_amd.meshgen import blob_mesh
ctx=_hip.default_context()
rng=np.random.default_rng(int(sys.argv[1]))
fails=0; worst=0.0; t0=time.time(); N=int(sys.argv[2]); HOLES=len(sys.argv)>3 and sys.argv[3]=='holes'
for it in range(N):
    n=int(rng.choice([120,300,700,1500,4000,9000,20000,45000,150000])); k=int(rng.integers(1,13)); seed=int(rng.integers(0,10**6))
    m=blob_mesh(n,seed=seed); pts, faces = m.points, m.faces
    mode=int(rng.integers(0,6)); radius=6.0
    if HOLES: n=int(rng.choice([45000,150000,300000])); k=int(rng.integers(1,4)); mode=2; radius=float(rng.uniform(2.0,14.0)); m=blob_mesh(n,seed=seed); pts, faces = m.points, m.faces
    if mode==1: faces=np.delete(faces, rng.choice(len(faces), size=max(1,len(faces)//500), replace=False), axis=0)
    if mode==2: faces=faces[np.linalg.norm(pts[faces].mean(1)-pts[0],axis=1)>radius]
    if mode==3:  # second component + stray points
        m2=blob_mesh(max(60,n//3),seed=seed+1); pts=np.concatenate([pts,m2.points+500.0,rng.normal(size=(3,3))]); faces=np.concatenate([faces,m2.faces+n])
    if mode==4: pts=pts*np.array([1.0,1e-3,1.0])   # squashed: tiny edge lengths, huge weights
    if mode==5: pts=pts*1e4+1e6                      # large offsets / scales
    try:
        g=Graph(PolyMesh(pts,faces),n_spectral_features=k,norm_eig_vecs=False,n_rand_samples=10**9,ctx=ctx,verbose=False)
        g.get_graph_spectrum(); g.get_laplacian_matrix(); L=g.laplacian_matrix
        R=L@g.eig_vecs-g.eig_vecs*g.eig_vals[None,:]
        res=float(np.max(np.linalg.norm(R,axis=0)))
        paired=np.isclose(g.eig_vals[:-1],g.eig_vals[1:],rtol=1e-9).any() if len(g.eig_vals)>1 else False
        if not paired: worst=max(worst,res)
        ok=(res<1e-8 or paired) and len(g.eig_vals)>=min(k, len(pts)-5) and np.all(np.diff(g.eig_vals)>=-1e-18) and np.all(np.isfinite(g.eig_vecs))
        if not ok:
            fails+=1; print("FAIL n=%d k=%d seed=%d mode=%d res=%.2e cols=%d sym=%s"%(n,k,seed,mode,res,len(g.eig_vals),g.device.symmetric),flush=True)
        g.device.close()
    except Exception as e:
        fails+=1; print("EXC n=%d k=%d seed=%d mode=%d: %s: %s"%(n,k,seed,mode,type(e).__name__,str(e)[:200]),flush=True)
print("done: %d failures of %d, worst residual %.2e, %.1fs"%(fails,N,worst,time.time()-t0))
