#!/usr/bin/env python3
"""CPU study, second part (see gs_probe_cpu.py): would a PARALLEL Gauss-Seidel do?  Multicolour sweeps (red-black on the 7-point fine level, Jones-Plassmann
colouring with random priorities below: 9-10 colours) against lexicographic Gauss-Seidel and the reference's damped Jacobi, V(1,1) inside scipy's BiCGSTAB on
csky3d.  48^3: Jacobi 76, lexicographic 27, multicolour 49 iterations - on this convection-dominated class most of Gauss-Seidel's gain is its downstream
ORDER, which a colouring gives up; at 1.5x the matrix passes per cycle the multicolour form does not pay.  Not product code.  usage: gs_multicolour_probe_cpu.py [N=48]"""
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
print("levels",[a.shape[0] for a in As],flush=True)
lu=spla.splu(As[-1].tocsc())
D=[a.diagonal() for a in As]
def greedy_color(a, seed=0):
    # parallel-style colouring: random priorities, Jones-Plassmann with smallest available colour (what a device kernel would do)
    m=a.shape[0]; S=(a+a.T).tocsr(); S.setdiag(0); S.eliminate_zeros()
    rng=np.random.default_rng(seed); pr=rng.permutation(m)
    col=-np.ones(m,dtype=np.int64); ip,ix=S.indptr,S.indices
    rows=np.repeat(np.arange(m),np.diff(ip))
    rounds=0
    while (col<0).any():
        un=col<0
        # local max among uncoloured neighbours
        nbr_pr=np.where(un[ix],pr[ix],-1)
        mx=np.zeros(m,dtype=np.int64)-1
        np.maximum.at(mx,rows,nbr_pr)
        pick=un&(pr>mx)
        idx=np.nonzero(pick)[0]
        # smallest colour not used by coloured neighbours
        for i in idx:
            used=set(col[ix[ip[i]:ip[i+1]]].tolist())
            c=0
            while c in used: c+=1
            col[i]=c
        rounds+=1
    return col,rounds
cols=[]
for l,a in enumerate(As[:-1]):
    if l==0:
        i=np.arange(n); kk=i%N; jj=(i//N)%N; ii=i//(N*N)
        cols.append(((ii+jj+kk)%2).astype(np.int64))
    else:
        c,r=greedy_color(a); cols.append(c); print("level",l,"colours",c.max()+1,"rounds",r,flush=True)
Arow=[a.tocsr() for a in As]
def mc_sweep(l,x,b,reverse):
    a=Arow[l]; c=cols[l]; nc=c.max()+1
    order=range(nc-1,-1,-1) if reverse else range(nc)
    for cc in order:
        idx=np.nonzero(c==cc)[0]
        r=b[idx]-a[idx]@x
        x[idx]+=r/D[l][idx]
    return x
L=[sps.tril(a,0).tocsr() for a in As]; U=[sps.triu(a,0).tocsr() for a in As]
def smooth(kind,l,x,b,post):
    a=As[l]
    if kind=='jac':
        return x+0.6*(b-a@x)/D[l] if x is not None else 0.6*b/D[l]
    if kind=='gs':
        r=b-(a@x) if x is not None else b
        e=spla.spsolve_triangular(U[l] if post else L[l], r, lower=not post)
        return (x if x is not None else 0)+e
    if kind=='mc':
        xx=x.copy() if x is not None else np.zeros(a.shape[0])
        return mc_sweep(l,xx,b,post)
    if kind=='mcgs0':   # lexicographic... n/a
        raise ValueError
def cyc(kind,l,b):
    if l==len(As)-1: return lu.solve(b)
    x=smooth(kind,l,None,b,False)
    r=b-As[l]@x
    x=x+Ps[l]@cyc(kind,l+1,Ps[l].T@r)
    return smooth(kind,l,x,b,True)
rng=np.random.default_rng(0); b=rng.random(n)
for kind in ('jac','gs','mc'):
    its=[0]
    def cb(x): its[0]+=1
    t=time.time()
    x,info=spla.bicgstab(A,b,M=spla.LinearOperator((n,n),matvec=lambda r: cyc(kind,0,r)),rtol=1e-10,maxiter=400,callback=cb)
    print(f"{kind}: BiCGSTAB {its[0]} iterations (info {info}), true residual {np.linalg.norm(b-A@x)/np.linalg.norm(b):.1e}, {time.time()-t:.1f}s",flush=True)
