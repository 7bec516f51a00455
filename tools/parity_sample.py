"""A larger parity sample than the test-suite affords: every k-th cell of the 1e5-cell tot benchmark batch through the
kernel and through the oracle (oracle/kpp_ros3.c, one process per host core).  Reports step-bookkeeping agreement and the
concentration differences.  GPU box: python tools/parity_sample.py [ncells_in_sample=4096] [mech=tot]"""
import multiprocessing as mp, os, sys, time
import numpy as np
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT)

def work(job):
    mech, v, f, r = job
    from oracle.oracle import Oracle
    return Oracle(mech).integrate_batch(v, f, r)

if __name__ == '__main__':
    nsample = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    mech = sys.argv[2] if len(sys.argv) > 2 else 'tot'
    from oracle.oracle import build_oracle
    build_oracle()
    from mistra_amd.workload import make_batch
    n = 100000
    idx = np.unique(np.linspace(0, n - 1, nsample).astype(np.int64))
    var, fix, rconst = (x.numpy() for x in make_batch(mech, 0, n, 'cpu'))
    v, f, r = var[idx], fix[idx], rconst[idx]
    cores = max(1, min(16, len(os.sched_getaffinity(0))))
    parts = np.array_split(np.arange(len(idx)), cores)
    t0 = time.time()
    with mp.get_context('fork').Pool(cores) as pool:          # before the GPU is touched
        out = pool.map(work, [(mech, v[p], f[p], r[p]) for p in parts])
    want = np.concatenate([o[0] for o in out]); st_want = np.concatenate([o[2] for o in out])
    print('oracle: %d cells on %d cores in %.1f s' % (len(idx), cores, time.time() - t0), flush=True)
    from mistra_amd import chem
    res = chem.integrate(mech, v, f, r)
    same = (res.stats == st_want).all(axis=1)
    floor = 1e-12 * np.abs(want).max(axis=1, keepdims=True)
    d = np.abs(res.var - want) / (np.abs(want) + floor)
    major = np.abs(want) >= 1e-4 * np.abs(want).max(axis=1, keepdims=True)
    print('%s: %d sampled cells of the 1e5 batch; /Statistics/ identical in %d (%.3f %%); ierr ok %d' %
          (mech, len(idx), same.sum(), 100.0 * same.mean(), int((res.ierr == 1).sum())))
    print('max rel diff over cells with identical bookkeeping: all species %.3e, major species %.3e; median of per-cell max %.3e' %
          (d[same].max(), np.where(major, d, 0)[same].max(), np.median(d[same].max(axis=1))))
    if (~same).any():
        bad = np.where(~same)[0]
        print('cells with different bookkeeping:', idx[bad][:20], 'kernel stats', res.stats[bad][:5].tolist(), 'oracle', st_want[bad][:5].tolist())
        print('their max rel diff: all %.3e major %.3e' % (d[bad].max(), np.where(major, d, 0)[bad].max()))
