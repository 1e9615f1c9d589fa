"""axpby / axpbypcz at 512^3 against the pairs each lane handles (option blas1_pairs; 0 = capped persistent grid).  usage: PAIRS=k blas1_grid.py"""
import os, sys, time
sys.path.insert(0, '/root/repo')
import multigridsolver_amd as mg
ctx = mg.Context(0); n = 512 ** 3
ctx.set_option('blas1_pairs', int(os.environ.get('PAIRS', '1')))
x = ctx.vec(n).rand(seed=1); y = ctx.vec(n).rand(seed=2); z = ctx.vec(n).rand(seed=3)
def timed(f, reps=20):
    f(); ctx.sync(); t0 = time.perf_counter()
    for _ in range(reps): f()
    ctx.sync(); return (time.perf_counter() - t0) / reps * 1e3
for r in range(2):
    t1 = timed(lambda: y.axpby(0.5, x, 0.25)); t2 = timed(lambda: z.axpbypcz(0.5, x, 0.25, y, 0.125))
    print(f"blas1_pairs={os.environ.get('PAIRS','1')}: axpby {t1:.3f} ms = {24*n/t1/1e9:.2f} TB/s; axpbypcz {t2:.3f} ms = {32*n/t2/1e9:.2f} TB/s", flush=True)
