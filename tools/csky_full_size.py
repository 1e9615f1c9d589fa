#!/usr/bin/env python3
"""bench.py's convection-diffusion leg at the HEADLINE size: csky3d at N^3 (default 512^3 = 134 M rows, 938 M entries; variable coefficients,
nonsymmetric) built on the host, uploaded once — SpMV / cycle time per million rows beside the constant-coefficient Poisson operator of the
same size, BiCGSTAB + V on three right-hand sides.  usage: csky_full_size.py [N=512]  → one JSON line"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MGS_ARENA_GB", "110")
import bench
import multigridsolver_amd as mg

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
args = argparse.Namespace(omega=0.6, nu1=1, nu2=1, ktg=10.0, npass=2, tou=8.0, coarse_rows=2500)
out = bench.convdiff_leg(mg, args, N)
print(json.dumps(out))
