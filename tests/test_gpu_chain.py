"""The one-launch chain (itd_set_chain_mode, pyitd_amd/csrc/itd_chain.hpp) against the golden vectors and the CPU oracle:
bit-exact rows, baselines and knot counts, the level-by-level repeat when the stop rule fires inside the requested levels,
and ITD_CHAIN_ONLY's refusal to repeat.  (PYITD_CHAIN_MODE=0 runs the whole -m gpu suite through the chain.)"""
import numpy as np
import pytest

from helpers import assert_bits_equal, fuzz_signal, load_golden, sha, sines_noise

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    from pyitd_amd import engine
    return engine


@pytest.fixture(scope="module")
def oracle():
    from oracle import cpu_oracle
    return cpu_oracle


def _run(E, x, M, mode, want_baselines=True):
    x = np.ascontiguousarray(x)
    eng = E.Engine(x.shape[-1], 1)
    eng.set_chain_mode(mode)
    out = eng.decompose_host(x, M, want_baselines=want_baselines)
    out["repeats"] = eng.chain_repeats
    eng.close()
    return out


@pytest.mark.parametrize("name", ["chirp65536_f32_m3", "sines131072_f32_m7", "sines16384_f64_m7", "radio_tiled_32768_f32_m9",
                                  "radio8000_m11", "edge_lead_plateau_nan", "edge_monotone", "edge_n3", "edge_noise_m0",
                                  "edge_noise_f32_odd", "edge_staircase", "demo400_m11", "edge_zigzag1024", "edge_noise_m20",
                                  "nanin_tile_edges", "nanin_f32"])
def test_chain_matches_golden(E, name):
    g = load_golden(name)
    out = _run(E, g["x"], int(g["max_iteration"]), E.CHAIN_AUTO)
    assert out["rows"].shape[0] == int(g["n_rows"])
    assert sha(out["rows"]) == str(g["rows_sha256"]), "rows are not bit-identical to the reference"
    assert tuple(out["baselines"].shape) == tuple(g["baselines_shape"])
    assert sha(out["baselines"]) == str(g["baselines_sha256"])
    if str(g["stop"]) != "timeout" and not name.startswith("nanin_"):
        assert out["repeats"] == 1    # the stop rule fired inside the requested levels: the chain had to be repeated level by level
    # (NaN in the input: itd_get_summary goes straight to the NaN-faithful level-by-level repeat)


def test_chain_alone_completes_a_long_signal(E, oracle):
    n, M = 1 << 21, 7
    x = sines_noise(n)
    out = _run(E, x, M, E.CHAIN_ONLY, want_baselines=False)     # CHAIN_ONLY: an incomplete chain would raise
    ref = oracle.itd_lean(x, M)
    assert out["repeats"] == 0
    assert_bits_equal(out["rows"], ref["rows"], "chain rows, 2^21 samples")
    kc = [int(v) for v in out["knot_counts"] if v >= 0]
    assert kc[: len(ref["knot_counts"])] == [int(v) for v in ref["knot_counts"]]


def test_chain_ragged_sizes_and_fuzz_slice(E, oracle):
    rng = np.random.default_rng(20260)
    for case in range(60):
        n = int(rng.choice([3, 5, 200, 1023, 1024, 1025, 2047, 2049, 3000, 9999, 65553, 150001]))
        kind = int(rng.integers(0, 8))
        M = int(rng.integers(0, 10))
        dt = np.float32 if rng.integers(0, 2) else np.float64
        x = fuzz_signal(rng, kind, n).astype(dt)
        if not np.isfinite(x).all():
            continue
        ref = oracle.itd(x, M)
        out = _run(E, x, M, E.CHAIN_AUTO)
        what = "case %d: n=%d kind=%d M=%d %s" % (case, n, kind, M, dt.__name__)
        assert_bits_equal(out["rows"], ref["rows"], what + " rows")
        assert_bits_equal(out["baselines"], ref["baselines"], what + " baselines")


def test_chain_batch_with_baselines(E, oracle):
    import torch
    B, n, M = 20, 1 << 15, 5
    xs = np.stack([sines_noise(n, seed=b, fscale=1 + b / 64) for b in range(B)])
    xs[3] = np.linspace(0, 1, n)            # monotone: the stop rule fires at once for this signal -> the whole call is repeated
    x = torch.from_numpy(xs).cuda()
    rows = torch.zeros((B, M + 2, n), dtype=torch.float64, device="cuda")
    bases = torch.zeros((B, M + 2, n), dtype=torch.float64, device="cuda")
    eng = E.Engine(n, B)
    eng.set_chain_mode(E.CHAIN_AUTO)
    eng.decompose_dev(x.data_ptr(), np.float32, n, B, n, M, rows.data_ptr(), bases.data_ptr(), None)
    s = eng.summary(B)
    assert eng.chain_repeats == 1
    for b in range(B):
        ref = oracle.itd(xs[b], M)
        nr, nb = int(s["n_rows"][b]), int(s["n_baselines"][b])
        assert nr == ref["rows"].shape[0]
        assert_bits_equal(rows[b, :nr].cpu().numpy(), ref["rows"], "signal %d rows" % b)
        assert_bits_equal(bases[b, :nb].cpu().numpy(), ref["baselines"], "signal %d baselines" % b)
    # the same batch without the monotone signal: the chain alone
    xs[3] = xs[2]
    x = torch.from_numpy(xs).cuda()
    eng.set_chain_mode(E.CHAIN_ONLY)
    eng.decompose_dev(x.data_ptr(), np.float32, n, B, n, M, rows.data_ptr(), None, None)
    s = eng.summary(B)
    for b in (0, 3, B - 1):
        ref = oracle.itd(xs[b], M)
        assert_bits_equal(rows[b, : int(s["n_rows"][b])].cpu().numpy(), ref["rows"], "chain-only signal %d" % b)
    eng.close()


def test_chain_only_refuses_the_repeat(E):
    import pyitd_amd
    x = np.linspace(0.0, 1.0, 5000)     # monotone: no knots, the reference stops at once
    with pytest.raises(pyitd_amd.ITDError):
        _run(E, x, 4, E.CHAIN_ONLY)
    out = _run(E, x, 4, E.CHAIN_AUTO)
    assert out["rows"].shape == (1, 5000) and not out["rows"].any() and out["repeats"] == 1
