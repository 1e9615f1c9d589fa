#!/usr/bin/env python3
"""CPU study (scipy; hierarchy from the oracle's restatement of the reference's aggregation): what the paper's smoother would buy on the reference's
convection-diffusion class.  csky3d at N^3 (with the bundled file's row-sum margin), V(1,1) and K-cycle (GCR form, all levels) with damped Jacobi
(0.6 = the reference's, 0.8) against forward/backward Gauss-Seidel (docs/AGMG_For_Convection_Diffusion.pdf; Fortran `smoothtype = 1`,
src/CPU_Matlab/dagtwolev_mex.f90:50-54), inside scipy's BiCGSTAB and a plain FGCR(10).  Not product code, nothing here runs on the GPU.
usage: gs_probe_cpu.py [N=48]   (N = 64 takes a minute: the triangular solves are scipy's)"""
import os, sys, time, numpy as np, scipy.sparse as sps, scipy.sparse.linalg as spla
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigridsolver_amd.synthetic import csky3d, CSKY_ROWSUM_MARGIN
from oracle import oracle_py as orc
N=int(sys.argv[1]) if len(sys.argv)>1 else 48
n=N**3
rp,ci,v=csky3d(N, rowsum_floor=CSKY_ROWSUM_MARGIN)
A=sps.csr_matrix((v,ci,rp),shape=(n,n))
As=[A]; Ps=[]
while As[-1].shape[0]>2500 and len(As)<10:
    M=As[-1].tocsr(); M.sort_indices()
    Ao=orc.Csr.from_arrays(M.shape[0],M.shape[0],M.indptr.astype(np.int32),M.indices.astype(np.int32),M.data)
    P=Ao.agmg(10.0,2,8.0,strict=False).to_scipy().tocsr()
    Ps.append(P); As.append((P.T@M@P).tocsr())
print("levels",[a.shape[0] for a in As])
lu=spla.splu(As[-1].tocsc())
L=[sps.tril(a,0).tocsr() for a in As]; U=[sps.triu(a,0).tocsr() for a in As]
D=[a.diagonal() for a in As]
def smooth(kind,l,x,b,post):
    a=As[l]
    if kind=='jac':
        return x+0.6*(b-a@x)/D[l] if x is not None else 0.6*b/D[l]
    if kind=='jac08':
        return x+0.8*(b-a@x)/D[l] if x is not None else 0.8*b/D[l]
    r=b-(a@x) if x is not None else b
    x0=x if x is not None else 0
    if kind=='gs':      # forward pre, backward post (the paper's smoother)
        e=spla.spsolve_triangular(U[l] if post else L[l], r, lower=not post)
        return x0+e
    raise ValueError
def cyc(kind,l,b):
    if l==len(As)-1: return lu.solve(b)
    x=smooth(kind,l,None,b,False)
    r=b-As[l]@x
    x=x+Ps[l]@cyc(kind,l+1,Ps[l].T@r)
    return smooth(kind,l,x,b,True)
def kc(kind,l,b,klev):   # K-cycle GCR form on levels 1..klev
    def inner(l,rhs):
        if l==len(As)-1: return lu.solve(rhs)
        x=smooth(kind,l,None,rhs,False)
        r=rhs-As[l]@x
        x=x+Ps[l]@coarse(l+1,Ps[l].T@r)
        return smooth(kind,l,x,rhs,True)
    def coarse(l,rhs):
        if not(1<=l<=klev and l<len(As)-1): return inner(l,rhs)
        c1=inner(l,rhs); v1=As[l]@c1; rho1=v1@v1; a1=v1@rhs
        rp_=rhs-(a1/rho1)*v1
        c2=inner(l,rp_); v2=As[l]@c2; g=(v2@v1)/rho1; v2o=v2-g*v1; rho2=v2o@v2o; a2=v2o@rp_
        k1=a1/rho1;k2=0.0
        if rho2>0: k2=a2/rho2; k1-=g*k2
        return k1*c1+k2*c2
    return inner(l,b)
rng=np.random.default_rng(0); b=rng.random(n)
def count(prec,name,flex=False):
    its=[0]
    def cb(x): its[0]+=1
    M=spla.LinearOperator((n,n),matvec=prec)
    t=time.time()
    if flex:
        # simple FGCR(10)
        x=np.zeros(n); r=b.copy(); nb=np.linalg.norm(b); it=0
        while it<300:
            Cs=[];Vs=[];rh=[]
            for k in range(10):
                c=prec(r); vv=A@c
                for cj,vj,rj in zip(Cs,Vs,rh):
                    hh=(vj@vv)/rj; vv=vv-hh*vj; c=c-hh*cj
                rho=vv@vv; al=(vv@r)/rho; x+=al*c; r-=al*vv; Cs.append(c);Vs.append(vv);rh.append(rho); it+=1
                if np.linalg.norm(r)/nb<1e-10 or it>=300: break
            r=b-A@x
            if np.linalg.norm(r)/nb<1e-10: break
        print(f"  {name}: FGCR(10) {it} iterations, true residual {np.linalg.norm(b-A@x)/nb:.1e}, {time.time()-t:.1f}s",flush=True)
    else:
        x,info=spla.bicgstab(A,b,M=M,rtol=1e-10,maxiter=400,callback=cb)
        print(f"  {name}: BiCGSTAB {its[0]} iterations (info {info}), true residual {np.linalg.norm(b-A@x)/np.linalg.norm(b):.1e}, {time.time()-t:.1f}s",flush=True)
nl=len(As)
for kind in ('jac','jac08','gs'):
    print(kind)
    count(lambda r: cyc(kind,0,r), "V(1,1)")
    count(lambda r: kc(kind,0,r,nl-2), "K all levels", flex=True)
    count(lambda r: cyc(kind,0,r), "V(1,1) in FGCR(10)", flex=True)
