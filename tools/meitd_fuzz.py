"""MEITD as one launch (itd_meitd_small_f64) against the host-driven loop (one launch per operator) on random signals: the same
components bit for bit, the same numbers of extractions and probes.  usage: python tools/meitd_fuzz.py [cases] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyitd_amd import meitd



def run(cases, seed, log=print):
    """returns (mismatches, handed to the host-driven loop, early returns)"""
    rng = np.random.default_rng(seed)
    bad = handed = early = 0
    msg1 = msg2 = ""
    for case in range(cases):
        n = int(rng.choice([1024, 1500, 2048, 3000, 4096, 4800, 5000, 8192]))
        t = np.arange(n) / 1000.0
        fam = case % 6
        if fam == 0:
            x = np.sin(2 * np.pi * rng.uniform(1, 8) * t) + rng.uniform(0.05, 0.5) * rng.standard_normal(n)
        elif fam == 1:
            x = np.cumsum(rng.standard_normal(n))
        elif fam == 2:
            x = np.sin(2 * np.pi * rng.uniform(1, 5) * t) * (1 + 0.5 * np.sin(2 * np.pi * rng.uniform(0.1, 1) * t)) + 0.3 * np.sin(2 * np.pi * rng.uniform(20, 60) * t)
        elif fam == 3:
            x = rng.standard_normal(n) * 10.0 ** rng.integers(-6, 7)
        elif fam == 4:
            x = np.round(np.sin(2 * np.pi * rng.uniform(1, 8) * t) * 50 + 5 * rng.standard_normal(n))      # plateaus, ties
        else:
            x = np.exp(-((t - t.mean()) * rng.uniform(0.5, 3)) ** 2) + 1e-3 * rng.standard_normal(n) * (case % 12 == 5)
        wpemax = float(rng.uniform(0.3, 0.9))
        with np.errstate(all="ignore"):
            try:
                got = meitd.MEITD(x.copy(), WPEMAX=wpemax)
                err = None
            except Exception as ex:                      # (the host-driven loop must raise the same)
                got, err, msg1 = None, type(ex), str(ex)
            wk = meitd._work_for(n, 0)
            last = dict(wk.last)
            calls = {"extract": 0, "probe": 0}
            orig = {k: getattr(wk, k) for k in calls}
            for k in calls:
                setattr(wk, k, (lambda kk: lambda *a, **kw: (calls.__setitem__(kk, calls[kk] + 1), orig[kk](*a, **kw))[1])(k))
            wk.one_launch = False
            try:
                ref = meitd.MEITD(x.copy(), WPEMAX=wpemax)
                err2 = None
            except Exception as ex:
                ref, err2, msg2 = None, type(ex), str(ex)
            finally:
                wk.one_launch = True
                for k in calls:
                    delattr(wk, k)
        if last.get("status", 0) >= 2 or last.get("status", 0) < 0:
            handed += 1
        if last.get("status") == 1:
            early += 1
        ok = err == err2 and (got is None or all(a.shape == b.shape and np.array_equal(a, b, equal_nan=True) for a, b in zip(got, ref)))
        if ok and last.get("status") == 0:
            ok = last["extractions"] == calls["extract"] and last["probes"] == calls["probe"]
        if not ok:
            bad += 1
            log("case %d (family %d, n %d, WPEMAX %.3f): MISMATCH  %s %s %s %s" % (case, fam, n, wpemax, {k: v for k, v in last.items() if k != "us"}, calls,
                                                                                     (err, err2), (msg1 if err else "", msg2 if err2 else "")))
            if os.environ.get("MEITD_FUZZ_DUMP"):
                np.savez(os.path.join(os.environ["MEITD_FUZZ_DUMP"], "case_%d.npz" % case), x=x, wpemax=wpemax)
    return bad, handed, early


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    t0 = time.time()
    bad, handed, early = run(cases, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("%d cases, %d mismatches, %d handed to the host-driven loop, %d early returns, %.1f s" % (cases, bad, handed, early, time.time() - t0))
    sys.exit(1 if bad else 0)
